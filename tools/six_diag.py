#!/usr/bin/env python3
"""six_diag.py (GPU box): where the seven-wave kernel first leaves the four-wave kernel's results.  For growing call lengths
(whole tiles) it demodulates the same buffer from the reset state with both kernels and names the state words that differ:
the first length at which a stream's words differ localises the tile.   tools/six_diag.py S N [workload] [step]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import webaudio_modem_amd as wm
from state_fields import REAL, INT

S, N = int(sys.argv[1]), int(sys.argv[2])
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
step = int(sys.argv[4]) if len(sys.argv) > 4 else 16
cfg = dict(baudRate=300, markFrequency=1070, spaceFrequency=1270) if wl == "c2" else dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
st = torch.cuda.current_stream().cuda_stream
x = torch.zeros((S, N), dtype=torch.float32, device="cuda")
g = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
g.synth_device(x.data_ptr(), N, N, 32 if wl == "c2" else 100, 0xF5C0DE, 10 * (160 if wl == "c2" else 40), 0.1, 1.0, st)
torch.cuda.synchronize()
g.close()
extra = dict(kv.split("=", 1) for kv in filter(None, os.environ.get("SIX_OPTS", "").split(",")))


def run(kernel, n):
    eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32, options=dict(extra, kernel=kernel))
    op = eng.max_bytes(N)
    out = torch.zeros((S, op), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(S, dtype=torch.int32, device="cuda")
    eod = torch.zeros(S, dtype=torch.int32, device="cuda")
    eng.demodulate_device(x.data_ptr(), n, N, out.data_ptr(), op, cnt.data_ptr(), eod.data_ptr(), 0, st)
    torch.cuda.synchronize()
    k = eng.last_kernel()
    states = [eng.debug_state(s) for s in range(min(S, 64))]
    eng.close()
    return cnt.cpu().numpy(), eod.cpu().numpy(), states, k


def diff(n):
    a, b = run("auto-r04", n), run("seven-wave", n)
    bad = {}
    for s, ((ra, ia), (rb, ib)) in enumerate(zip(a[2], b[2])):
        names = [REAL[i] for i in range(len(ra)) if np.float64(ra[i]).view(np.uint64) != np.float64(rb[i]).view(np.uint64)]
        names += [INT[i] for i in range(len(ia)) if ia[i] != ib[i]]
        if names:
            bad[s] = (names, {nm: ((ra[REAL.index(nm)], rb[REAL.index(nm)]) if nm in REAL else (ia[INT.index(nm)], ib[INT.index(nm)])) for nm in names[:6]})
    return bad, a[3], b[3], (a[0], b[0])


lo, hi = 0, N // 16 * 16
bad, k4, k6, _ = diff(hi)
print("kernels:", k4, "|", k6)
if not bad:
    print("no difference at n = %d" % hi)
    sys.exit(0)
print("n = %d: %d streams differ, e.g. %s" % (hi, len(bad), list(bad.items())[:2]))
# bisect over whole tiles for the first length with any difference
lo_t, hi_t = 1, hi // 16
while lo_t < hi_t:
    mid = (lo_t + hi_t) // 2
    b, _, _, _ = diff(mid * 16)
    if b:
        hi_t = mid
    else:
        lo_t = mid + 1
b, _, _, cn = diff(lo_t * 16)
print("first difference with %d tiles (%d samples, %d decimated):" % (lo_t, lo_t * 16, lo_t * 8))
for s, (names, vals) in list(b.items())[:8]:
    print("  stream %d: %s  %s" % (s, names, vals))
b2, _, _, _ = diff((lo_t - 1) * 16) if lo_t > 1 else ({}, 0, 0, 0)
print("one tile fewer: %d streams differ" % len(b2))
