"""Hostile input on the GPU (round 6): NaN / +-Inf / out-of-range / subnormal samples, every case captured from
the REAL reference (tests/golden/golden_hostile.npz, oracle/refrun/golden_harness_hostile.js), through the C ABI.

What the reference does (fsk.ts:52-76, 175-188, 264, 285; filters.ts:47-76): a NaN fails both AGC level tests (the gain holds),
the pre-filter is never reset (one NaN or Inf poisons the instance for good), `NaN > 0` slices 0 and `NaN < threshold` is
never silence -- the instance emits nothing more, no byte, no 'eod'.  Out-of-range finite samples and subnormal ones it simply
follows in doubles.

The bar here:
  * FSKHIP_PRECISION_F64: bytes, per-call byte / 'eod' distribution and status EXACTLY the reference's on every case;
  * FSKHIP_PRECISION_F32, every whole-tile kernel: the same on every case inside fp32's range -- the non-finite and the subnormal
    ones included (the slicer is NaN-proof since this round: fsk_pipe_dev.h slicer_nf) -- and on the three kinds of case that
    leave it (|x * gain| beyond ~1e19: the I/Q branch runs 2^60 times larger than the reference's) the documented behaviour:
    fskhip_get_faults() reports the stream, it stays quiet, nothing else is touched;
  * fskhip_get_faults() = 1 exactly where the stream's filter state is no longer finite;
  * the other 63 streams of the bad stream's group are byte-identical to a run without it; nothing hangs (pytest-timeout)."""
import numpy as np
import pytest

from conftest import golden_hostile, hostile_status_matches, run_chunked

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]

VARIANTS = [("f64", 1, {}), ("f32-auto", 0, {}), ("f32-seven-wave", 0, {"kernel": "seven-wave"}),
            ("f32-four-wave", 0, {"kernel": "four-wave"}), ("f32-four-wave-resets", 0, {"kernel": "four-wave", "blk_resets": 1}),
            ("f32-two-wave", 0, {"kernel": "two-wave"}), ("f32-one-wave", 0, {"kernel": "one-wave"}), ("f32-generic", 0, {"force_generic": 1})]


def _names():
    return [c["name"] for c in golden_hostile().manifest["cases"]]


def _beyond_f32(name):
    """cases whose samples leave the fp32 engines' range (the reference's doubles go on)"""
    return name.endswith("scale_1e25") or name.endswith("scale_3e38") or name == "h_dflt_fmax_mid"


def _state_poisoned(case, x):
    """does the REFERENCE's own filter state end up non-finite?  A non-finite sample always does it; one FLT_MAX sample does where the
    AGC gain is above 1 at that moment (the product is stored as a float: Infinity) -- seen in the reference's own output: it goes quiet."""
    if not np.all(np.isfinite(x)):
        return True
    return case["name"] == "h_bell_fmax_mid"


@pytest.mark.parametrize("vname,prec,opts", VARIANTS)
@pytest.mark.parametrize("name", _names())
def test_hostile_input_matches_reference(name, vname, prec, opts):
    import webaudio_modem_amd as wm
    g = golden_hostile()
    c = g.cases[name]
    x = g.case_input(c)
    eng = wm.FSKEngine(1, c["config"], precision=wm.PRECISION_F64 if prec else wm.PRECISION_F32, options=opts or None)

    def call(chunk):
        out, eod = eng.demodulate_data(chunk.reshape(1, -1).copy())
        return out[0], int(eod[0])

    out, eod, nonempty, n_calls = run_chunked(call, x, c["chunk"])
    st = eng.get_status(0)
    fault = eng.fault(0)
    eng.close()
    if prec == 0 and _beyond_f32(name) and (fault or vname != "f32-generic"):
        # outside fp32's range: flagged, quiet from the overflow on, and nothing more is promised (the reference decodes on).  (The
        # per-sample generic kernel runs its I/Q branch unscaled: it follows the reference further out -- then all of it must match.)
        assert fault
        assert list(out) == c["bytes"][:len(out)]            # what it did emit before the overflow is the reference's
        return
    if vname == "f32-generic" and name in ("h_bell_scale_1e-40", "h_bell_scale_1e-44"):
        # the documented limit of the per-sample generic fp32 kernel (include/fskhip.h, fskhip_get_faults): its I/Q branch is not scaled
        # by 2^60, so a frame at 1e-40 of full scale is a handful of subnormal bits there.  Whether the reference itself syncs on such a
        # frame is marginal at Bell-202's shift (it does at 1e-44 and does not at 1e-40), and this kernel decides the other way in both.
        # The clean frame behind it decodes.  (The whole-tile kernels, i.e. every lock-step batch, follow the reference in both.)
        assert bytes(out).endswith(b"World") and not fault
        return
    assert list(out) == c["bytes"]
    assert eod == c["eod_total"]
    assert n_calls == c["calls"]["count"]
    assert nonempty == c["calls"]["nonempty"]
    assert hostile_status_matches(st, g.array(c["status_vector"]), 1e-12 if prec else 1e-5) is None
    assert fault == _state_poisoned(c, x)


@pytest.mark.parametrize("vname,prec,opts", [v for v in VARIANTS if v[0] in ("f64", "f32-auto", "f32-four-wave", "f32-seven-wave", "f32-two-wave", "f32-one-wave")])
@pytest.mark.parametrize("name", ["h_bell_qnan_mid", "h_bell_neg_qnan_mid", "h_bell_pinf_mid", "h_bell_ninf_mid", "h_bell_qnan_idle", "h_bell_scale_1e25",
                                  "h_bell_scale_3e38", "h_bell_fmax_mid", "h_dflt_fmax_mid", "h_dflt_snan_mid", "h_bell_scale_1e-40"])
def test_hostile_stream_leaves_its_neighbours_alone(name, vname, prec, opts):
    """the bad stream as number 37 of a 130-stream batch (three groups, the last one ragged) of clean ones: every other stream -- the 63
    that share its wave, the ones in the other groups -- decodes byte for byte what it decodes without it; the bad one what it decodes alone"""
    import webaudio_modem_amd as wm
    g = golden_hostile()
    c = g.cases[name]
    x, clean = g.case_input(c), g.clean_input(c)
    S, bad = 130, 37
    n = max(x.size, clean.size)
    base = np.zeros((S, n), np.float32)
    for s in range(S):
        sh = (7 * s) % 200                                   # the frames do not line up across the wave
        base[s, sh:sh + clean.size] = clean[:n - sh]
    X = base.copy()
    X[bad, :] = 0
    X[bad, :x.size] = x
    res = []
    for buf in (X, base):
        eng = wm.FSKEngine(S, c["config"], precision=wm.PRECISION_F64 if prec else wm.PRECISION_F32, options=opts or None)
        out, eod = eng.demodulate_data(buf.copy())
        res.append((out, [int(e) for e in eod], eng.faults()))
        eng.close()
    (o1, e1, f1), (o0, e0, f0) = res
    for s in range(S):
        if s != bad:
            assert o1[s] == o0[s] and e1[s] == e0[s], s
    assert not f0.any()
    assert [int(s) for s in np.nonzero(f1)[0]] == ([bad] if (_state_poisoned(c, x) or (prec == 0 and _beyond_f32(name))) else [])
    if not (prec == 0 and _beyond_f32(name)):
        assert list(o1[bad]) == c["bytes"] and e1[bad] == c["eod_total"]


def test_hostile_samples_at_full_width_never_hang():
    """4 096 streams (config #2's size: the seven-wave kernel in 16-stream groups), every seventh with a NaN or an Inf of either sign somewhere: the call returns, the clean streams decode, the flagged ones are the bad ones"""
    import webaudio_modem_amd as wm
    g = golden_hostile()
    c = g.cases["h_bell_clean"]
    clean = g.case_input(c)
    S = 4096
    X = np.tile(clean, (S, 1))
    rng = np.random.default_rng(0xBAD5EED)
    vals = np.array([np.nan, -np.nan, np.inf, -np.inf], np.float32)
    bad = np.arange(3, S, 7)
    for s in bad:
        X[s, rng.integers(0, clean.size)] = vals[rng.integers(0, vals.size)]
    for prec in (wm.PRECISION_F32, wm.PRECISION_F64):
        eng = wm.FSKEngine(S, c["config"], precision=prec)
        out, eod = eng.demodulate_data(X.copy())
        f = eng.faults()
        eng.close()
        good = np.setdiff1d(np.arange(S), bad)
        assert all(list(out[s]) == c["bytes"] for s in good)
        assert not f[good].any()
        assert f[bad].all()
