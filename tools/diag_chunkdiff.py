#!/usr/bin/env python3
"""diag_chunkdiff.py (GPU box): localise a chunking-invariance failure of tests/test_gpu_fullsize.py: which kernel choice
and which schedule disagree on which streams, and who agrees with the CPU oracle."""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from oracle import pyoracle as po
from test_gpu_fullsize import _demod_schedule, BELL, SEED
S, N = 262144, 12000 // 128 * 128
gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
d_x = gen.device_malloc(S * N * 4)
gen.synth_device(d_x, N, N, 20, SEED, 400, 0.1, 1.0)
gen.synchronize()
res = {}
for split in ("a", "1", "0", "4"):
    for name, sched in (("one", [N]), ("ragged", [1000, 17, 4096, 3, 128, 2049])):
        os.environ["FSKHIP_SPLIT"] = split
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        os.environ.pop("FSKHIP_SPLIT")
        res[(split, name)] = _demod_schedule(eng, d_x, N, N, sched)
        eng.close()
base = res[("1", "one")]
bad = set()
for k, (rows, eod) in res.items():
    d = [s for s in range(S) if rows[s] != base[0][s] or eod[s] != base[1][s]]
    print(k, "differs from (pipe, one call) on", len(d), d[:8], flush=True)
    bad |= set(d)
row = np.empty(N, np.float32)
for s in sorted(bad)[:4]:
    gen.d2h(row, d_x + s * N * 4)
    ob, oe = po.OracleCore(BELL).demodulate(row)
    print("stream", s, "oracle", len(ob), zlib.crc32(ob), "eod", oe)
    for k, (rows, eod) in res.items():
        print("   ", k, len(rows[s]), zlib.crc32(rows[s]), int(eod[s]), "== oracle" if rows[s] == ob else "DIFFERS")
    np.save(os.path.join(ROOT, "gpurun_out", "chunkdiff_stream_%d.npy" % s), row)
