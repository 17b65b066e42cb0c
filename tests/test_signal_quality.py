"""getSignalQuality().  The reference ships an all-zero stub (fsk.ts:471-479), which the host
classes keep; the opt-in estimates are an extension DEFINED in include/fskhip.h / oracle/fsk_oracle.h.  CPU: the oracle's
restatement of that definition behaves (known SNR, known carrier offset).  GPU: libfskhip against the oracle on the same
buffers, every kernel path that implements it."""
import numpy as np
import pytest

KEYS = ["snr", "ber", "eyeOpening", "phaseJitter", "frequencyOffset", "signalLevel", "noiseFloor", "frames", "bytes"]


def _burst(cfg_tx, payload, amp, noise_rms, seed, lead=300, tail=900):
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)
    sig = po.OracleCore(cfg_tx).modulate(payload)
    x = np.concatenate([np.zeros(lead, np.float32), sig * np.float32(amp), np.zeros(tail, np.float32)])
    return (x + noise_rms * rng.standard_normal(len(x))).astype(np.float32)


def test_oracle_estimates_track_noise_and_carrier_offset():
    from oracle import pyoracle as po
    payload = bytes(np.random.default_rng(3).integers(0, 256, 160, dtype=np.uint8))
    est = {}
    for name, off, noise in (("clean", 0, 0.002), ("noisy", 0, 0.00632), ("high", 15, 0.002), ("low", -15, 0.002)):
        o = po.OracleCore({})
        assert o.quality() == {k: 0.0 for k in KEYS}          # off: the reference's zeros
        o.enable_quality()
        x = _burst(dict(markFrequency=1650 + off, spaceFrequency=1850 + off), payload, 0.5, noise, 7)
        got, _ = o.demodulate(x)
        assert got == payload
        est[name] = o.quality()
        assert est[name]["frames"] == 1 and est[name]["bytes"] == len(payload)
    # 10 dB more noise power -> about 10 dB less SNR, and more jitter on the discriminator output
    assert 7.0 < est["clean"]["snr"] - est["noisy"]["snr"] < 13.0
    assert est["noisy"]["phaseJitter"] > est["clean"]["phaseJitter"]
    assert 0.5 < est["clean"]["eyeOpening"] <= 1.0 and 0.0 <= est["clean"]["ber"] < 0.25
    # a carrier that sits 15 Hz high / low moves the estimate by about that much, in that direction
    assert 8.0 < est["high"]["frequencyOffset"] - est["clean"]["frequencyOffset"] < 25.0
    assert -25.0 < est["low"]["frequencyOffset"] - est["clean"]["frequencyOffset"] < -8.0
    # digital silence after the frame: a floor of exactly zero is reported as the 200 dB cap, not as infinity
    o = po.OracleCore(dict(agcEnabled=False))
    o.enable_quality()
    o.demodulate(_burst(dict(agcEnabled=False), b"x" * 8, 0.5, 0.0, 1, tail=20000))
    assert o.quality()["snr"] > 100.0


def test_host_classes_keep_the_references_zero_stub():
    import webaudio_modem_amd as wm
    core = wm.FSKCore()
    assert core.getSignalQuality() == {"snr": 0, "ber": 0, "eyeOpening": 0, "phaseJitter": 0, "frequencyOffset": 0}
    assert core.getSignalQualityEstimates()["frames"] == 0
    with pytest.raises(RuntimeError, match="not configured"):
        core.enableSignalQualityEstimates()


@pytest.mark.gpu
@pytest.mark.parametrize("prec_name", ["f64", "f32", "f32-generic"])
def test_gpu_estimates_match_the_oracle(prec_name, monkeypatch):
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    prec = wm.PRECISION_F64 if prec_name == "f64" else wm.PRECISION_F32
    if prec_name == "f32-generic":
        monkeypatch.setenv("FSKHIP_FORCE_GENERIC", "1")
    S = 70
    rng = np.random.default_rng(11)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(20, 60)), dtype=np.uint8)) for _ in range(S)]
    xs = [_burst(dict(markFrequency=1650 + (s % 5 - 2) * 10, spaceFrequency=1850 + (s % 5 - 2) * 10), payloads[s],
                 0.2 + 0.01 * s, 0.003 * (1 + s % 4), 100 + s) for s in range(S)]
    N = max(len(v) for v in xs) + 500
    x = np.zeros((S, N), np.float32)
    for s, v in enumerate(xs):
        x[s, :len(v)] = v
    eng = wm.FSKEngine(S, {}, precision=prec)
    assert eng.get_signal_quality(3) == {k: 0.0 for k in KEYS}
    eng.enable_signal_quality()
    off = 0
    for n in (1000, 333, 4096, N):                            # any chunking
        n = min(n, N - off)
        if n:
            eng.demodulate_data(x[:, off:off + n].copy())
        off += n
    if prec_name == "f32":
        assert "tail" in eng.last_kernel()                    # fp32 engines run diagnostics on the sample-granular kernel
    tol = 1e-9 if prec_name == "f64" else 2e-3
    for s in range(0, S, 3):
        o = po.OracleCore({})
        o.enable_quality()
        ob, _ = o.demodulate(x[s])
        assert ob == payloads[s]
        want, got = o.quality(), eng.get_signal_quality(s)
        assert got["frames"] == want["frames"] == 1 and got["bytes"] == want["bytes"] == len(payloads[s])
        for k in KEYS:
            scale = {"snr": 60.0, "frequencyOffset": 100.0}.get(k, max(abs(want[k]), 1e-3))
            assert abs(got[k] - want[k]) <= tol * scale, (prec_name, s, k, got[k], want[k])
    # switching off freezes the values, switching on again clears them
    eng.enable_signal_quality(False)
    before = eng.get_signal_quality(0)
    assert before["bytes"] == len(payloads[0])
    eng.reset()
    eng.demodulate_data(x.copy())
    assert eng.get_signal_quality(0) == before
    eng.enable_signal_quality(True)
    assert eng.get_signal_quality(0) == {k: 0.0 for k in KEYS}
    eng.close()
