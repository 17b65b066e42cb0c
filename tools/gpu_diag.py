"""Diagnostics on the GPU box: intermediate errors of the fp32/fp64 kernels vs the golden traces."""
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from conftest import golden

g = golden()
for c in g.manifest["cases"]:
    if "trace" not in c:
        continue
    x = g.case_input(c)
    ref_amp = g.array(c["trace"]["amp"]); ref_post = g.array(c["trace"]["post_out"]); ref_bit = g.array(c["trace"]["bit"])
    for pname, prec in (("f64", 1), ("f32", 0)):
        eng = wm.FSKEngine(1, c["config"], precision=prec)
        eng.trace_enable(0, x.size // 2 + 8)
        out, eod = eng.demodulate_data(x.reshape(1, -1))
        tr = eng.trace_read()
        n = min(tr["amp"].size, ref_amp.size)
        peak = ref_amp.max()
        da = np.abs(tr["amp"][:n] - ref_amp[:n])
        big = ref_amp[:n] >= 0.01 * peak
        rel = (da[big] / ref_amp[:n][big]).max()
        dp = np.abs(tr["post_out"][:n] - ref_post[:n])
        flips = int(np.count_nonzero(tr["bit"][:n] != ref_bit[:n]))
        # margin of flipped decisions
        fl = np.nonzero(tr["bit"][:n] != ref_bit[:n])[0]
        mflip = np.abs(ref_post[fl]).max() if fl.size else 0.0
        st = eng.get_status(0)
        print("%-28s %s n=%d/%d amp: max_abs=%.3e (peak %.3f) max_rel(>1%%peak)=%.3e | post max_abs=%.3e | bit flips=%d (max |ref post| at flips %.2e) | thr rel err %.2e | bytes_ok=%s" % (
            c["name"], pname, tr["amp"].size, ref_amp.size, da.max(), peak, rel, dp.max(), flips, mflip,
            abs(st["silenceThreshold"] - c["status"]["silenceThreshold"]) / c["status"]["silenceThreshold"], list(out[0]) == c["bytes"]))
        eng.close()
