#!/usr/bin/env python3
"""lib_compare.py (GPU box): kernel time of the C3-shaped batch with alternative builds of libfskhip.so
(tools/build/libfskhip_<tag>.so), one child process each.  Diagnostic aid."""
import os, sys, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import webaudio_modem_amd as wm
S, N = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
st = torch.cuda.current_stream().cuda_stream
x = torch.empty((S, N), dtype=torch.float32, device="cuda")
op = eng.max_bytes(N)
out = torch.empty((S, op), dtype=torch.uint8, device="cuda"); cnt = torch.empty(S, dtype=torch.int32, device="cuda")
eng.synth_device(x.data_ptr(), N, N, 100, 0xF5C0DE, 400, 0.1, 1.0, st)
torch.cuda.synchronize()
def step(): eng.demodulate_device(x.data_ptr(), N, N, out.data_ptr(), op, cnt.data_ptr(), 0, 0, st)
step(); torch.cuda.synchronize()
eng.timing_begin()
for _ in range(5): step()
torch.cuda.synchronize()
n, ms = eng.timing_end()
print("RESULT", ms / n, eng.last_kernel(), int(cnt.sum().item()))
''' % ROOT
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
libs = [("default", None)] + [(os.path.basename(p)[10:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "build", "libfskhip_*.so"))) if "ablate" not in p]
for tag, path in libs:
    env = dict(os.environ)
    if path:
        env["FSKHIP_LIB_OVERRIDE"] = path
    r = subprocess.run([sys.executable, "-c", CHILD, str(S), str(N)], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(tag, "FAILED", r.stderr[-300:]); continue
    f = line[0].split()
    ms = float(f[1])
    print("%-12s %8.3f ms  %7.1f Gsamples/s  %s bytes=%s" % (tag, ms, S * N / ms / 1e6, f[2], f[-1]), flush=True)
