"""(GPU box) needs gpurun_out/idle_group_<g0>.npy.  One lane's carried state after each call of a schedule, four-wave kernel
next to the two-wave kernel.  argv: g0 lane cut [more cuts...]"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import webaudio_modem_amd as wm  # noqa: E402
import state_fields as sf  # noqa: E402

BELL = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
g0, lane = int(sys.argv[1]), int(sys.argv[2])
cuts = [int(v) for v in sys.argv[3:]]
x = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "idle_group_%d.npy" % g0))
S, N = x.shape
sched = [b - a for a, b in zip([0] + cuts, cuts + [N])]


def run(kern):
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": kern})
    d_x = eng.device_malloc(S * N * 4)
    eng.h2d(d_x, x)
    op = eng.max_bytes(N)
    d_out = eng.device_malloc(S * op); d_cnt = eng.device_malloc(S * 4); d_eod = eng.device_malloc(S * 4)
    eod = np.zeros(S, np.uint32)
    off = 0
    states = []
    for n in sched:
        eng.demodulate_device(d_x + off * 4, n, N, d_out, op, d_cnt, d_eod)
        eng.synchronize()
        eng.d2h(eod, d_eod)
        off += n
        r, i = eng.debug_state(lane)
        states.append((off, int(eod[lane]), r, i, eng.last_kernel().split("::")[-1][:14]))
    eng.close()
    return states


a, b = run("two-wave"), run("four-wave")
for (off, ea, ra, ia, ka), (_, eb, rb, ib, kb) in zip(a, b):
    print("after sample %d: eod of the call %d (%s) / %d (%s)" % (off, ea, ka, eb, kb))
    for k, (u, v) in enumerate(zip(ra, rb)):
        fu, fv = struct.pack("<f", u), struct.pack("<f", v)
        if fu != fv or sf.REAL[k] in ("zq_ai", "zq_bi", "last_phase", "sil_thr"):
            print("    %-10s %-16.9g %-16.9g %s" % (sf.REAL[k], u, v, "" if fu == fv else "DIFFERS"))
    for k, (u, v) in enumerate(zip(ia, ib)):
        if u != v or sf.INT[k] in ("zr_dph", "sil_cnt", "gsc", "eod_total"):
            print("    %-10s %-16d %-16d %s" % (sf.INT[k], u, v, "" if u == v else "DIFFERS"))
