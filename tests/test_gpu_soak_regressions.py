"""Streams on which tools/soak.py caught the round-2 whole-tile arithmetic decoding other bytes than the reference
(DESIGN.md section 7), kept as fixtures: every kernel path, the failing chunk schedule and a one-call run must now give
the oracle's bytes, per-call eod counts and status.

  soak_r02_default_s48   lead-in of exact zeros with an 'eod' reset 11 decimated samples before a weak frame: the
                         free-running frame's phase of the zero vector (Math.atan2(0, 0) = 0 is frame-dependent)
  soak_r02_sync075_s131  syncThreshold 0.75: a sync on the ringing after a frame; U - q cancelled to rounding noise
  soak_r02_v21_s8        V.21 300 baud: the same in the slower low-pass, a frame lost / gained after a false sync
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CASES = [
    ("soak_r02_default_s48.npy", {}, [1168] + [16] * 60),
    ("soak_r02_sync075_s131.npy", {"syncThreshold": 0.75}, [129, 17, 4096, 1000]),
    ("soak_r02_v21_s8.npy", {"baudRate": 300, "markFrequency": 1070, "spaceFrequency": 1270}, [1000, 128, 128, 1000]),
]
PATHS = [("seven-wave", {"FSKHIP_SPLIT": "6"}), ("four-wave", {"FSKHIP_SPLIT": "4"}), ("two-wave", {"FSKHIP_SPLIT": "1"}), ("one-wave", {"FSKHIP_SPLIT": "0"}),
         ("generic", {"FSKHIP_FORCE_GENERIC": "1"})]
KEYS = ["frameStarted", "globalSampleCounter", "receivedBitsLength", "syncDetections"]


@pytest.mark.parametrize("path,env", PATHS)
@pytest.mark.parametrize("fname,cfg,sched", CASES)
def test_soak_found_streams_decode_like_the_reference(fname, cfg, sched, path, env, monkeypatch):
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    x = np.load(os.path.join(ROOT, "tools", "diag_data", fname)).astype(np.float32)
    S = 64
    for schedule in (sched, []):                      # the schedule that failed, then one call
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
        for k in env:
            monkeypatch.delenv(k)
        o = po.OracleCore(cfg)
        off = 0
        for n in list(schedule) + [len(x) - sum(schedule)]:
            out, eod = eng.demodulate_data(np.tile(x[off:off + n], (S, 1)))
            ob, oe = o.demodulate(x[off:off + n])
            assert out[S - 1] == ob and int(eod[S - 1]) == oe, (fname, path, off, n, out[S - 1].hex(), ob.hex())
            assert out[0] == ob
            off += n
        st, ost = eng.get_status(S - 1), o.status()
        if cfg.get("syncThreshold", 0.85) >= 0.8:     # (counts inside zero tails: DESIGN.md section 2)
            for k in KEYS:
                assert st[k] == ost[k], (fname, path, k)
        eng.close()


def test_the_one_fp32_divergence_of_the_soak_is_a_marginal_slicer_decision():
    """soak_r02_stop2_s80: the one stream in ~600 000 randomised stream-runs on which an fp32 engine decoded other bytes
    than the reference after the round-2 fixes.  It is the generic fp32 kernel (fractional ring capacity), and the cause is
    one slicer decision the reference itself takes at |post-filter output| = 1.0e-8 (fp32: +1.1e-8) while its sync count
    sits exactly on the threshold.  tools/soak.py accepts a divergence only if it traces back to such a decision; this
    pins both that classification and that the fp64 engine follows the reference on the same stream."""
    import sys
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import soak
    cfg = {"stopBits": 2, "markFrequency": 900.0, "spaceFrequency": 1100.0}
    x = np.load(os.path.join(ROOT, "tools", "diag_data", "soak_r02_stop2_s80.npy")).astype(np.float32)
    ok, why = soak.fp32_mismatch_is_marginal(cfg, x, False)
    assert ok, why
    o = po.OracleCore(cfg)
    want, oe = o.demodulate(x)
    e64 = wm.FSKEngine(1, cfg, precision=wm.PRECISION_F64)
    got, eod = e64.demodulate_data(x.reshape(1, -1))
    assert got[0] == want and int(eod[0]) == oe and len(want) >= 1
    e64.close()
