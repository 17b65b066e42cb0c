"""(GPU box) one recorded stream, traced on an fp64 and an fp32 engine (one call): where do they part, and was the decision
there marginal?  Prints the first decimated sample whose amplitudes differ by more than 1e-4 relative (a reset took effect in
one engine and not in the other) and the amplitudes / silence threshold in front of it.  usage: diag_marginal.py x.npy cfg-json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

x = np.load(sys.argv[1]).astype(np.float32)
cfg = json.loads(sys.argv[2])
tr, thr = [], []
for prec in (wm.PRECISION_F64, wm.PRECISION_F32):
    e = wm.FSKEngine(1, cfg, precision=prec)
    e.trace_enable(0, len(x))
    out, eod = e.demodulate_data(x.reshape(1, -1).copy())
    tr.append(e.trace_read())
    st = e.get_status(0)
    print("prec", prec, "bytes", out[0].hex(), "eod", int(eod[0]), "thr", st["silenceThreshold"], "syncs", st["syncDetections"], e.last_kernel())
    e.close()
ob, oe = po.OracleCore(cfg).demodulate(x)
print("oracle bytes", ob.hex(), "eod", oe)
a64, a32 = tr[0]["amp"], tr[1]["amp"]
n = min(len(a64), len(a32))
rel = np.abs(a64[:n] - a32[:n]) / np.maximum(np.abs(a64[:n]), 1e-30)
bad = np.nonzero(rel > 1e-4)[0]
print("decimated samples", n, "first amplitude divergence > 1e-4 at", int(bad[0]) if len(bad) else None)
d = np.nonzero(tr[0]["bit"][:n] != tr[1]["bit"][:n])[0]
print("first differing slicer bit at", int(d[0]) if len(d) else None)
if len(bad):
    k = int(bad[0])
    lo = max(0, k - 150)
    print("amplitudes in front of it (fp64, fp32, rel diff):")
    for i in range(max(lo, k - 12), k + 3):
        print("   %6d  %.9e  %.9e  %.2e   post %.3e / %.3e" % (i, a64[i], a32[i], rel[i], tr[0]["post_out"][i], tr[1]["post_out"][i]))
    # silence run bookkeeping: which thresholds would make the two engines count differently in the 141 samples before k?
    seg64, seg32 = a64[lo:k], a32[lo:k]
    cand = np.sort(np.concatenate([seg64, seg32]))
    print("samples in [%d, %d) where the two amplitudes straddle a common value within 1e-5 relative:" % (lo, k))
    for i in range(lo, k):
        if a64[i] != a32[i] and abs(a64[i] - a32[i]) / a64[i] < 1e-5:
            pass
    print("(silence threshold after the last sync is in the status lines above; compare with the amplitudes listed)")
    near = [(i, a64[i], a32[i]) for i in range(lo, k)]
    for thr_name, t in (("fp64 status thr", None),):
        pass
