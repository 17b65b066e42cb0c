#!/usr/bin/env python3
"""diag_bits.py (GPU box): per-decimated-sample trace (amplitude, post-filter output, slicer bit) of one recorded stream on
the fp64 engine, the generic fp32 kernel and fsk_pipe.hip's arithmetic (sample-granular kernel), next to each other.
usage: diag_bits.py <x.npy> <cfg-json> [first_push last_push]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
x = np.load(sys.argv[1]).astype(np.float32)
cfg = json.loads(sys.argv[2])
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
S = 64
def run(env, prec):
    os.environ.update(env)
    eng = wm.FSKEngine(S, cfg, precision=prec)
    for k in env: os.environ.pop(k)
    eng.trace_enable(S - 1, len(x))
    eng.demodulate_data(np.tile(x, (S, 1)))
    name = eng.last_kernel()
    t = eng.trace_read()
    t = (t['amp'], t['post_out'], t['bit'])
    eng.close()
    return name, t
n64, t64 = run({}, wm.PRECISION_F64)
ng, tg = run({"FSKHIP_FORCE_GENERIC": "1"}, wm.PRECISION_F32)
nt, tt = run({}, wm.PRECISION_F32)
print(n64, ng, nt)
a64, p64, b64 = t64
for name, (a, p, b) in (("generic f32", tg), ("pipe arithmetic", tt)):
    n = min(len(b), len(b64))
    bd = np.nonzero(b[:n] != b64[:n])[0]
    pe = np.abs(p[:n] - p64[:n])
    ae = np.abs(a[:n] - a64[:n]) / np.maximum(a64[:n], 1e-30)
    print("%-16s pushes %d  bit flips vs f64 %d  max|post err| %.3g at %d  median %.3g  max rel amp err %.3g" % (name, n, len(bd), pe.max(), int(pe.argmax()), float(np.median(pe)), float(ae[a64[:n] > 1e-6].max())))
    for i in bd[:40]:
        print("    push %6d  f64 post %+.3e  this %+.3e  amp %.3e" % (i, p64[i], p[i], a64[i]))
for i in range(lo, hi):
    print("%6d  amp %.6e %.6e %.6e   post %+.6e %+.6e %+.6e  bits %d %d %d" % (i, a64[i], tg[0][i], tt[0][i], p64[i], tg[1][i], tt[1][i], b64[i], tg[2][i], tt[2][i]))
