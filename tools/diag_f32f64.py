#!/usr/bin/env python3
"""diag_f32f64.py (GPU box): streams of the BASELINE-shaped batch on which the fp32 engine's bytes differ from the fp64
engine's; dumps their inputs (npz) for offline analysis.  Diagnostic aid."""
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
S, N = 65536, 48000
out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/diag"
os.makedirs(out_dir, exist_ok=True)
gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
d_x = gen.device_malloc(S * N * 4)
gen.synth_device(d_x, N, N, 20, SEED + 5, 400, 0.1, 1.0)
gen.synchronize()
res = {}
for name, prec, env in (("f32_pipe", wm.PRECISION_F32, {}), ("f64", wm.PRECISION_F64, {}), ("f32_generic", wm.PRECISION_F32, {"FSKHIP_FORCE_GENERIC": "1"}),
                        ("f32_fused", wm.PRECISION_F32, {"FSKHIP_SPLIT": "0"})):
    os.environ.update(env)
    e = wm.FSKEngine(S, BELL, precision=prec)
    for k in env: os.environ.pop(k)
    res[name] = _demod_schedule(e, d_x, N, N, [N])
    e.close()
bad = [s for s in range(S) if res["f32_pipe"][0][s] != res["f64"][0][s]]
print("f32_pipe vs f64 differ:", bad[:20], len(bad))
for name in ("f32_generic", "f32_fused"):
    b2 = [s for s in range(S) if res[name][0][s] != res["f64"][0][s]]
    print(name, "vs f64 differ:", b2[:20], len(b2))
row = np.empty(N, np.float32)
dump = {}
for s in bad[:4]:
    gen.d2h(row, d_x + s * N * 4)
    ob, oe = po.OracleCore(BELL).demodulate(row)
    print("stream", s, "oracle", ob.hex(), "eod", oe)
    for name in res:
        print("   %-12s %s eod %d" % (name, res[name][0][s].hex(), int(res[name][1][s])))
    dump["x%d" % s] = row.copy()
np.savez_compressed(os.path.join(out_dir, "f32f64_streams.npz"), **dump)
