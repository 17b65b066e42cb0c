#!/usr/bin/env python3
"""diag_chunk.py (GPU box): which streams differ between chunk schedules of the whole-tile fp32 kernels, and who is
right according to the oracle.  Diagnostic aid, not part of the suite."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from oracle import pyoracle as po
from test_gpu_fullsize import _demod_schedule, BELL, SEED

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = 48000 // 128 * 128
gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
d_x = gen.device_malloc(S * N * 4)
gen.synth_device(d_x, N, N, 20, SEED, 400, 0.1, 1.0)
gen.synchronize()
res = {}
for name, sched in (("one", [N]), ("half", [N // 2]), ("q128", [128]), ("q16", [16 * 25])):
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    res[name] = _demod_schedule(eng, d_x, N, N, sched)
    print(name, "kernel", eng.last_kernel(), flush=True)
    eng.close()
row = np.empty(N, np.float32)
for name in ("half", "q128", "q16"):
    bad = [s for s in range(S) if res[name][0][s] != res["one"][0][s] or res[name][1][s] != res["one"][1][s]]
    print(name, "differs on", len(bad), "streams", bad[:10], flush=True)
    for s in bad[:6]:
        gen.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(BELL).demodulate(row)
        a, b = res["one"][0][s], res[name][0][s]
        print("  stream", s, "oracle len", len(ob), "eod", oe, "| one: len", len(a), "eod", int(res["one"][1][s]), "==oracle", a == ob,
              "|", name, ": len", len(b), "eod", int(res[name][1][s]), "==oracle", b == ob)
