#!/usr/bin/env python3
"""stage_times.py (GPU box): how long each wave of demod_pipe_kernel needs on its own.

Runs the FSK_ABLATE build (tools/build/libfskhip_ablate.so: `make -C webaudio_modem_amd/csrc` objects + fsk_pipe.hip compiled
with -DFSK_ABLATE) with FSK_ABLATE = bitmask of waves whose arithmetic is skipped, so that the remaining wave sets the pace.
Results are wrong by construction; only the kernel time is read.  Diagnostic aid, not part of the suite."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
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
for _ in range(4): step()
torch.cuda.synchronize()
n, ms = eng.timing_end()
print("RESULT", ms / n, eng.last_kernel())
''' % ROOT
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
for mask, what in ((0, "all three waves"), (6, "wave 0 only (staging + AGC + pre-filter)"), (5, "wave 1 only (mixer + low-pass + discriminator)"),
                   (3, "wave 2 only (frame logic)"), (7, "none (loads, LDS hand-off, polling)"), (4, "waves 0+1"), (1, "waves 1+2"), (2, "waves 0+2")):
    env = dict(os.environ, FSKHIP_LIB_OVERRIDE=os.path.join(ROOT, "tools", "build", "libfskhip_ablate.so"), FSK_ABLATE=str(mask))
    r = subprocess.run([sys.executable, "-c", CHILD, str(S), str(N)], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(mask, what, "FAILED", r.stderr[-300:]); continue
    ms = float(line[0].split()[1])
    cyc = ms * 1e-3 * 2.1e9 / N
    print("ablate=%d  %-50s %8.3f ms  %7.1f Gsamples/s  ~%5.0f cycles/sample at 2.1 GHz" % (mask, what, ms, S * N / ms / 1e6, cyc), flush=True)
