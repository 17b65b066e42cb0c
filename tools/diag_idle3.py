"""(GPU box) needs gpurun_out/idle_group_<g0>.npy from tools/diag_idle2.py on the same box.  Four-wave kernel, one lane:
which single cut of the call changes its eod total?  argv: g0 lane lo hi step"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm  # noqa: E402

BELL = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
g0, lane, lo, hi, step = [int(v) for v in sys.argv[1:6]]
x = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "idle_group_%d.npy" % g0))
S, N = x.shape


def run(schedule, kern="four-wave"):
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": kern})
    d_x = eng.device_malloc(S * N * 4)
    eng.h2d(d_x, x)
    op = eng.max_bytes(N)
    d_out = eng.device_malloc(S * op); d_cnt = eng.device_malloc(S * 4); d_eod = eng.device_malloc(S * 4)
    eod = np.zeros(S, np.uint32)
    tot = np.zeros(S, np.int64)
    off = 0
    for n in schedule:
        eng.demodulate_device(d_x + off * 4, n, N, d_out, op, d_cnt, d_eod)
        eng.synchronize()
        eng.d2h(eod, d_eod)
        tot += eod
        off += n
    st = eng.get_status(lane)
    eng.close()
    return tot, st


ref, st0 = run([N])
print("one call: lane eod", ref[lane], "gsc", st0["globalSampleCounter"])
for a in range(lo, hi, step):
    for sched in ([a, N - a], [a, 16, N - a - 16]):
        tot, st = run(sched)
        d = np.nonzero(tot != ref)[0]
        if len(d):
            print("cut", sched[:-1], "-> lanes differing", list(d), "lane", lane, "eod", tot[lane], "gsc", st["globalSampleCounter"])
print("done")
