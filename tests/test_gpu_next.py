"""GPU parity tests of the SURVEY.md 8(f) rows (run with -m gpu on the MI355X box), through the C ABI of
libfskhip.so: CRC-16 / XModem packets / receive-grammar scan, batched FIR, ChunkedModulator and the FSKProcessor
quantum loop -- against the golden vectors captured from the real reference (tests/golden/golden_next.npz,
golden.npz) and against the CPU oracle on seeded inputs.  Integer/byte results must be identical; FIR outputs are
bit-identical on the fp64 path and within 1e-5 of full scale on the fp32 path."""
import numpy as np
import pytest

from conftest import golden, golden_next

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(seed)


# ---------------------------------------------------------------- CRC-16 / XModem ----------------
def test_crc16_golden_and_random_batch():
    import webaudio_modem_amd as wm
    from oracle import next_oracle as no
    g = golden_next()
    rows = g.ragged(g.manifest["crc"])
    vals = g.arrays[g.manifest["crc"]["values"]]
    got = wm.crc16_batch(rows)
    assert [int(v) for v in got] == [int(v) for v in vals]
    assert wm.CRC16.calculate(b"123456789") == 0x29B1 and wm.CRC16.calculate(b"") == 0xFFFF
    assert wm.CRC16.verify(b"123456789", 0x29B1) and not wm.CRC16.verify(b"123456789", 0x29B0)
    r = _rng(0xC4C)
    rows = [r.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in r.integers(0, 600, 3000)]
    got = wm.crc16_batch(rows)
    for i in range(0, len(rows), 7):
        assert int(got[i]) == no.crc16(rows[i]), i
    # unaligned pitch goes through the byte reader
    slab = np.zeros((5, 13), np.uint8)
    slab[:] = np.arange(13, dtype=np.uint8)
    lens = np.array([0, 1, 5, 12, 13], np.uint32)
    got = wm.crc16_batch((slab, lens))
    assert [int(v) for v in got] == [no.crc16(bytes(range(13))[:n]) for n in lens]


def test_xmodem_serialize_golden_and_errors():
    import webaudio_modem_amd as wm
    g = golden_next()
    p = g.manifest["packets"]
    payloads, wires = g.ragged(p["payload"]), g.ragged(p["wire"])
    got = wm.serialize_batch([m["seq"] for m in p["meta"]], payloads)
    assert got == wires
    for payload, m in zip(payloads[:6], p["meta"][:6]):
        pk = wm.XModemPacket.createData(m["seq"], payload)
        assert (pk["sequence"], pk["invSequence"], pk["length"], pk["checksum"]) == (m["seq"], m["inv"], m["len"], m["crc"])
        assert wm.XModemPacket.verify(pk) and wm.XModemPacket.serialize(pk) == wires[payloads.index(payload)]
    for e in p["errors"]:
        if e["seq"] < 0:
            continue
        with pytest.raises(ValueError) as ei:
            wm.serialize_batch([e["seq"]], [bytes(e["len"])])
        assert str(ei.value) == e["error"]
    assert list(wm.XModemPacket.serializeControl(wm.ControlType.EOT)) == p["control"]["EOT"]


def _check_scan(r, want, data, name):
    assert r["status_name"] == want["status"], name
    for k in ("expected_after", "packets", "dropped", "consumed", "err_seq", "err_len", "crc_rx", "crc_calc"):
        assert r[k] == want[k], (name, k)
    assert r["data"] == data, name


def test_xmodem_scan_golden_cases():
    import webaudio_modem_amd as wm
    g = golden_next()
    sc = g.manifest["scans"]
    bursts, datas = g.ragged(sc["bytes"]), g.ragged(sc["data"])
    res = wm.scan_bursts(bursts, [c["expected"] for c in sc["cases"]])
    for r, c, d in zip(res, sc["cases"], datas):
        _check_scan(r, c, d, c["name"])
    errs = {c["name"]: r["error"] for r, c in zip(res, sc["cases"])}
    assert errs["bad_crc_payload"] == "Invalid CRC" and errs["bad_nseq"] == "Invalid sequence number"
    assert errs["bad_seq_both_consistent"] == "Unexpected sequence number" and errs["one_ok"] is None


def test_xmodem_scan_random_batch_matches_oracle():
    import webaudio_modem_amd as wm
    from oracle import next_oracle as no
    r = _rng(0x5CA9)
    bursts, expected = [], []
    for _ in range(1500):
        seq = int(r.integers(1, 256))
        expected.append(seq if r.random() > 0.1 else int(r.integers(1, 256)))
        parts = []
        for _k in range(int(r.integers(0, 5))):
            if r.random() < 0.3:
                parts.append(r.integers(0, 256, int(r.integers(0, 4)), dtype=np.uint8).tobytes())
            w = bytearray(no.serialize(no.create_data(seq, r.integers(0, 256, int(r.integers(0, 256)), dtype=np.uint8).tobytes())))
            if r.random() < 0.2 and len(w):
                w[int(r.integers(0, len(w)))] ^= 1 << int(r.integers(0, 8))
            if r.random() < 0.1:
                w = w[:int(r.integers(0, len(w) + 1))]
            parts.append(bytes(w))
            if r.random() < 0.85:
                seq = seq % 255 + 1
        if r.random() < 0.4:
            parts.append(b"\x04")
        bursts.append(b"".join(parts))
    res = wm.scan_bursts(bursts, expected)
    seen = set()
    for i, (b, e, got) in enumerate(zip(bursts, expected, res)):
        want = no.scan_burst(b, e)
        seen.add(want["status"])
        for k in ("status", "expected_after", "packets", "dropped", "consumed", "err_seq", "err_len", "crc_rx", "crc_calc", "data"):
            assert got[k] == want[k], (i, k)
    assert seen == {0, 1, 2, 3, 4, 5}


def test_packet_round_trip_through_the_modem_on_gpu():
    """serialize -> modulate -> demodulate -> scan, all on the GPU: every stream gets its payload back."""
    import webaudio_modem_amd as wm
    S = 200
    r = _rng(0x900D)
    cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
    payloads = [r.integers(0, 256, int(r.integers(1, 129)), dtype=np.uint8).tobytes() for _ in range(S)]
    seqs = [int(v) for v in r.integers(1, 256, S)]
    wires = wm.serialize_batch(seqs, payloads)
    eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
    sigs = eng.modulate_data(wires)
    n = max(len(s) for s in sigs) + 64
    x = np.zeros((S, n), np.float32)
    for i, s in enumerate(sigs):
        x[i, :len(s)] = s
    out, _ = eng.demodulate_data(x)
    res = wm.scan_bursts(out, seqs)
    ok = sum(1 for got, p in zip(res, payloads) if got["data"] == p and got["packets"] == 1)
    assert ok == S, "%d/%d packets survived a clean channel" % (ok, S)
    eng.close()


# ---------------------------------------------------------------- FIR ---------------------------------
FIR_RUNS = [r for r in golden().manifest["filter_runs"] if isinstance(r["coeffs"], list)]


@pytest.mark.parametrize("run", FIR_RUNS, ids=lambda r: r["name"])
def test_fir_matches_reference_golden(run):
    import webaudio_modem_amd as wm
    g = golden()
    x = g.array(run["x"])
    ref = g.array(run["y_buffer"])
    f = wm.FIRFilter(run["coeffs"], precision=wm.PRECISION_F64)
    assert f.getCoefficients() == run["coeffs"]
    assert np.array_equal(f.processBuffer(x), ref)          # bit-identical Float32Array
    f.reset()
    parts = [f.processBuffer(x[a:b]) for a, b in ((0, 1), (1, 8), (8, 9), (9, 200), (200, 203), (203, 512))]
    assert np.array_equal(np.concatenate(parts), ref)       # the delay line carries across calls
    f32 = wm.FIRFilter(run["coeffs"], precision=wm.PRECISION_F32)
    y = f32.processBuffer(x)
    assert np.max(np.abs(y - ref)) <= 1e-5 * max(1.0, float(np.max(np.abs(ref))))


def test_fir_factories_and_per_sample_surface():
    import webaudio_modem_amd as wm
    g = golden()
    runs = {r["name"]: r for r in g.manifest["filter_runs"]}
    x = g.array("filt.x")
    for name, f in (("fir_lp_1000_51", wm.FilterFactory.createFIRLowpass(1000, 48000)),
                    ("fir_hp_1000_51", wm.FilterFactory.createFIRHighpass(1000, 48000)),
                    ("fir_bp_1750_800_51", wm.FilterFactory.createFIRBandpass(1750, 800, 48000))):
        ref = g.array(runs[name]["y_buffer"])
        y = f.processBuffer(x)
        # designs agree with V8 to the last ulp of sin/cos, so outputs agree to f32 rounding of ~1e-16 differences
        assert np.max(np.abs(y - ref)) <= 2e-7 * max(1.0, float(np.max(np.abs(ref)))), name
    # tests/dsp/filters.node.test.ts:190-206: the impulse response is the taps
    taps = [0.1, -0.2, 0.3, 0.25, -0.05]
    f = wm.FIRFilter(taps)
    y = [f.process(1.0 if i == 0 else 0.0) for i in range(8)]
    assert y == [float(np.float32(t)) for t in taps] + [0.0, 0.0, 0.0]


def test_fir_batch_matches_oracle():
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    r = _rng(0xF12)
    S, N = 70, 3000
    x = (r.random((S, N)) * 2 - 1).astype(np.float32)
    for n_taps in (1, 2, 5, 51, 64, 257):
        taps = list((r.random(n_taps) - 0.5))
        f = wm.FIRFilterBatch(taps, S, precision=wm.PRECISION_F64)
        y = np.concatenate([f.processBuffer(x[:, :1000]), f.processBuffer(x[:, 1000:1003]), f.processBuffer(x[:, 1003:])], axis=1)
        for s in (0, 1, 63, 64, 69):
            o = po.FIR(taps)
            assert np.array_equal(y[s], o.process_buffer(x[s])), (n_taps, s)
        f.reset(3)
        y2 = f.processBuffer(x[:, :100])
        o = po.FIR(taps)
        assert np.array_equal(y2[3], o.process_buffer(x[3, :100]))
        if n_taps > 1:
            assert not np.array_equal(y2[4], y[4, :100])  # stream 4 kept its delay line
        f.close()


# ---------------------------------------------------------------- ChunkedModulator ---------------------
def test_chunked_modulator_mirror_matches_reference():
    import webaudio_modem_amd as wm
    g = golden_next()
    for c in g.manifest["chunked"]:
        core = wm.FSKCore()
        core.configure(c["config"])
        cm = wm.ChunkedModulator(core)
        assert cm.isModulating() is False and cm.getProgress() == 0 and cm.getNextSamples(128) is None
        cm.startModulation(bytes(c["payload"]))
        direct = core.modulateData(bytes(c["payload"]))
        assert len(direct) == c["total"]
        steps, got = [], []
        while True:
            r = cm.getNextSamples(c["chunk"])
            if r is None:
                break
            steps.append([len(r["signal"]), int(r["isComplete"]), r["samplesConsumed"], r["totalSamples"],
                          cm.getProgress(), int(cm.isModulating())])
            got.append(r["signal"])
        keep = steps[:8] + steps[-8:] if c["steps_truncated"] else steps
        assert len(steps) == c["n_steps"] and keep == c["steps"], c["name"]
        assert np.array_equal(np.concatenate(got), direct)
        core.close()
    core = wm.FSKCore()
    core.configure({})
    cm = wm.ChunkedModulator(core)
    cm.startModulation(b"")
    assert cm.isModulating() is False and cm.getNextSamples(128) is None
    cm.startModulation(bytes([1, 2, 3]))
    cm.getNextSamples(128)
    assert cm.getProgress() == g.manifest["chunked_misc"]["mid"]["progress"]
    cm.cancel()
    assert cm.isModulating() is False and cm.getProgress() == 0
    core.close()


# ---------------------------------------------------------------- FSKProcessor quantum loop -------------
@pytest.mark.parametrize("use_graph", [False, True], ids=["launches", "graph"])
@pytest.mark.parametrize("name", [r["name"] for r in golden_next().manifest["processor"]])
def test_processor_loop_matches_reference(name, use_graph):
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    from test_oracle_next import rx_run_input
    g = golden_next()
    run = next(r for r in g.manifest["processor"] if r["name"] == name)
    if use_graph and run["quanta"] > 1200:
        pytest.skip("long run covered without the graph")
    S = 3  # three identical streams: every one must behave like the reference's single processor
    eng = wm.FSKEngine(S, run["config"], precision=wm.PRECISION_F64)
    if run["kind"] == "rx":
        proc = wm.FSKProcessorBatch(eng, rx_capacity=run["ring_capacity"], use_graph=use_graph)
        buf = rx_run_input(lambda: po.OracleCore(run["config"]), run)
        drains = {d["quantum"]: d["bytes"] for d in run["drains"]}
        probe = dict(map(tuple, run["length_probe"]))
        for q in range(run["quanta"]):
            proc.process(np.tile(buf[q * 128:(q + 1) * 128], (S, 1)), 0)
            if q in drains:
                for got in proc.demodulate():
                    assert list(got) == drains[q], (name, q)
            if q in probe and q % 512 == 0:
                assert list(proc.rx_lengths()) == [probe[q]] * S, (name, q)
        for got in proc.demodulate():
            assert list(got) == drains[run["quanta"]]
        assert proc.status(1)["demodulatedBufferLength"] == 0
    else:
        proc = wm.FSKProcessorBatch(eng, use_graph=use_graph)
        direct = po.OracleCore(run["config"]).modulate(bytes(run["payload"]))
        outs, complete_at = [], -1
        for q in range(run["quanta"]):
            if q == run["start_quantum"]:
                proc.modulate([bytes(run["payload"])] * S)
                st = proc.tx_state()
                assert list(st["totalSamples"]) == [run["total"]] * S and all(st["pendingModulation"])
                with pytest.raises(RuntimeError, match="Modulation already in progress"):
                    proc.modulate([b"x"] * S)
            before = proc.tx_state()["completed"].copy()
            outs.append(proc.process(None, 128))
            if proc.tx_state()["completed"][0] != before[0]:
                complete_at = q
        assert complete_at == run["complete_at"]
        out = np.concatenate(outs, axis=1)
        want = np.zeros(out.shape[1], np.float32)
        k = run["start_quantum"] * 128
        want[k:k + len(direct)] = direct
        for s in range(S):
            assert np.array_equal(out[s], want), (name, s)
        assert not any(proc.tx_state()["pendingModulation"])
    proc.close()
    eng.close()


@pytest.mark.parametrize("use_graph", [False, True], ids=["launches", "graph"])
def test_processor_batch_random_schedule_matches_oracle(use_graph):
    """70 streams (ragged wave), different payloads, modulations started at different quanta through the mask,
    RX fed with each stream's own frames; every stream must track its own ProcessorOracle."""
    import webaudio_modem_amd as wm
    from oracle import next_oracle as no
    from oracle import pyoracle as po
    cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
    S, Q, n = 70, 420, 128
    r = _rng(0xF1B)
    eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F64)
    proc = wm.FSKProcessorBatch(eng, rx_capacity=48, use_graph=use_graph)
    oracles = [no.ProcessorOracle(po.OracleCore(cfg), rx_capacity=48) for _ in range(S)]
    # RX input: lead + one frame of 20..60 bytes per stream
    x = np.zeros((S, Q * n), np.float32)
    for s in range(S):
        sig = po.OracleCore(cfg).modulate(r.integers(0, 256, int(r.integers(20, 61)), dtype=np.uint8).tobytes())
        lead = int(r.integers(0, 200)) * 2
        x[s, lead:lead + len(sig)] = sig[:Q * n - lead]
    tx_at = {int(q): None for q in r.choice(np.arange(5, 300), 6, replace=False)}
    drain_at = {100, 250, 419}
    for q in range(Q):
        if q in tx_at:
            pend = proc.tx_state()["pendingModulation"]
            mask = (r.random(S) < 0.4) & ~pend
            payloads = [r.integers(0, 256, int(r.integers(0, 12)), dtype=np.uint8).tobytes() for _ in range(S)]
            proc.modulate(payloads, mask=list(mask))
            for s in range(S):
                if mask[s]:
                    oracles[s].modulate(payloads[s])
        out = proc.process(x[:, q * n:(q + 1) * n], n)
        for s in range(S):
            want = oracles[s].process(x[s, q * n:(q + 1) * n], n)
            assert np.array_equal(out[s], want), (q, s)
        if q in drain_at:
            got = proc.demodulate()
            for s in range(S):
                assert got[s] == oracles[s].demodulate(), (q, s)
        if q % 97 == 0:
            assert list(proc.rx_lengths()) == [o.ring.length for o in oracles]
    st = proc.tx_state()
    assert list(st["completed"]) == [o.completed for o in oracles]
    assert sum(st["completed"]) > 20
    proc.reset(5)
    assert proc.rx_lengths()[5] == 0 and not proc.tx_state()["pendingModulation"][5]
    proc.close()
    eng.close()


def test_xmodem_scan_kernels_agree_across_layouts():
    """The tiled kernel (16-byte aligned rows, what the host entry point uses) and the row-walking kernels (4-byte
    aligned / unaligned rows, reachable through the device entry point) give identical results and payloads."""
    import ctypes as C
    import webaudio_modem_amd as wm
    from webaudio_modem_amd import _lib
    g = golden_next()
    sc = g.manifest["scans"]
    bursts = g.ragged(sc["bytes"])
    expected = np.array([c["expected"] for c in sc["cases"]], np.uint32)
    ref = wm.scan_bursts(bursts, expected)
    n = len(bursts)
    counts = np.array([len(b) for b in bursts], np.uint32)
    eng = wm.FSKEngine(1, {})
    L = _lib.lib()
    for pitch, data_pitch in ((int(counts.max()) + 1 | 1, 601), (int(counts.max()) + 4 & ~3, 604)):
        slab = np.zeros((n, pitch), np.uint8)
        for i, b in enumerate(bursts):
            slab[i, :len(b)] = np.frombuffer(b, np.uint8)
        d_b, d_c, d_e = eng.device_malloc(slab.nbytes), eng.device_malloc(4 * n), eng.device_malloc(4 * n)
        d_d, d_r = eng.device_malloc(data_pitch * n), eng.device_malloc(C.sizeof(_lib.XModemResult) * n)
        eng.h2d(d_b, slab); eng.h2d(d_c, counts); eng.h2d(d_e, expected)
        _lib.check(L.fskhip_xmodem_scan_device(d_b, pitch, d_c, d_e, n, d_d, data_pitch, d_r, None))
        eng.synchronize()
        data = np.zeros((n, data_pitch), np.uint8)
        res = np.zeros(n * 10, np.int32)
        eng.d2h(data, d_d); eng.d2h(res, d_r)
        res = res.reshape(n, 10)
        for i, want in enumerate(ref):
            got = dict(zip(("status", "expected_after", "packets", "dropped", "consumed", "data_len", "err_seq", "err_len", "crc_rx", "crc_calc"),
                           (int(v) for v in res[i])))
            for k, v in got.items():
                assert v == want[k], (pitch, sc["cases"][i]["name"], k)
            assert data[i, :got["data_len"]].tobytes() == want["data"]
        for p in (d_b, d_c, d_e, d_d, d_r):
            eng.device_free(p)
    eng.close()


# ---- batched generic IIRFilter (src/dsp/filters.ts:8-106, 325-344) ------------------------------------------------------------
IIR_RUNS = [r for r in golden().manifest["filter_runs"] if isinstance(r["coeffs"], dict)]


@pytest.mark.parametrize("run", IIR_RUNS, ids=lambda r: r["name"])
def test_iir_matches_reference_golden(run):
    """The six IIR runs captured from the real IIRFilter: process() per sample (doubles) and processBuffer() (Float32Array),
    bit for bit on the fp64 path, any chunking; the fp32 path within 1e-4 of the peak."""
    import webaudio_modem_amd as wm
    g = golden()
    x = g.array(run["x"])
    y64 = g.array(run["y_process"])
    y32 = g.array(run["y_buffer"])
    co = run["coeffs"]
    f = wm.IIRFilter(co["b"], co["a"], precision=wm.PRECISION_F64)
    assert f.getCoefficients() == {"b": co["b"], "a": co["a"]}
    got = f.processSamples(x.astype(np.float64))
    assert np.array_equal(got.view(np.uint64), y64.view(np.uint64))          # process(): the reference's doubles
    f.reset()
    assert np.array_equal(f.processBuffer(x).view(np.uint32), y32.view(np.uint32))   # processBuffer(): its Float32Array
    f.reset()
    parts = [f.processBuffer(x[a:b]) for a, b in ((0, 1), (1, 8), (8, 9), (9, 200), (200, 203), (203, x.size))]
    assert np.array_equal(np.concatenate(parts).view(np.uint32), y32.view(np.uint32))   # the histories carry across calls
    f.reset()
    assert [f.process(float(v)) for v in x[:40]] == [float(v) for v in y64[:40]]      # the per-sample surface
    f32 = wm.IIRFilter(co["b"], co["a"], precision=wm.PRECISION_F32)
    y = f32.processBuffer(x)
    assert np.max(np.abs(y.astype(np.float64) - y32)) <= 1e-4 * max(1.0, float(np.max(np.abs(y32))))


def test_iir_constructor_errors_and_normalisation():
    """filters.ts:19-21 (tests/dsp/filters-advanced.node.test.ts:115-143): the three constructor errors with the reference's
    messages -- from the host class and from the C ABI itself -- and the a[0] normalisation."""
    import ctypes as C
    import webaudio_modem_amd as wm
    from webaudio_modem_amd import _lib
    for b, a, msg in (([], [1.0], "Feedforward coefficients (b) cannot be empty"), ([1.0], [], "Feedback coefficients (a) cannot be empty"),
                      ([1.0], [0.0, 1.0], "First feedback coefficient (a[0]) cannot be zero")):
        with pytest.raises(ValueError, match=msg.replace("(", r"\(").replace(")", r"\)").replace("[", r"\[").replace("]", r"\]")):
            wm.IIRFilter(b, a)
        L = _lib.lib()
        h = C.c_void_p()
        bb, aa = (C.c_double * max(1, len(b)))(*b), (C.c_double * max(1, len(a)))(*a)
        rc = L.fskhip_iir_create(0, bb, len(b), aa, len(a), 1, wm.PRECISION_F64, C.byref(h))
        assert rc == -1 and L.fskhip_last_error().decode() == msg
    f = wm.IIRFilter([2.0, 1.0, 0.5], [2.0, -0.5, 0.25])
    assert f.getCoefficients() == {"b": [1.0, 0.5, 0.25], "a": [1.0, -0.25, 0.125]}
    with pytest.raises(wm.FskHipError) as ei:
        wm.IIRFilter([1.0] * 10, [1.0])
    assert ei.value.code == -3


def test_iir_factories_match_the_designs():
    import webaudio_modem_amd as wm
    g = golden()
    runs = {r["name"]: r for r in g.manifest["filter_runs"]}
    x = g.array("filt.x")
    for name, f in (("iir_lp_1200", wm.FilterFactory.createIIRLowpass(1200, 48000)), ("iir_hp_300", wm.FilterFactory.createIIRHighpass(300, 48000)),
                    ("iir_bp_1750_2600", wm.FilterFactory.createIIRBandpass(1750, 2600, 48000))):
        assert f.getCoefficients() == runs[name]["coeffs"], name
        assert np.array_equal(f.processBuffer(x).view(np.uint32), g.array(runs[name]["y_buffer"]).view(np.uint32)), name


def test_iir_batch_matches_oracle():
    """A ragged batch (70 streams: a full wave and a part of one), orders 0 .. 8 with unequal numerator / denominator lengths,
    odd chunk lengths, per-stream reset: every stream bit-identical to the oracle's IIR (fp64 path)."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    r = _rng(0x11F)
    S, N = 70, 2500
    x = (r.random((S, N)) * 2 - 1).astype(np.float32)
    for nb, na in ((1, 1), (2, 1), (1, 2), (3, 3), (2, 4), (5, 3), (9, 9), (9, 2)):
        b = list(r.random(nb) - 0.5)
        # a stable denominator: small feedback
        a = [1.0 + 0.5 * float(r.random())] + list((r.random(na - 1) - 0.5) * (0.8 / max(1, na - 1)))
        f = wm.IIRFilterBatch(b, a, S, precision=wm.PRECISION_F64)
        y = np.concatenate([f.processBuffer(x[:, :1000]), f.processBuffer(x[:, 1000:1003]), f.processBuffer(x[:, 1003:])], axis=1)
        for s in (0, 1, 31, 63, 64, 69):
            o = po.IIR(b, a)
            assert np.array_equal(y[s].view(np.uint32), o.process_buffer(x[s]).view(np.uint32)), (nb, na, s)
        f.reset(5)
        y2 = f.processBuffer(x[:, :300])
        o = po.IIR(b, a)
        assert np.array_equal(y2[5].view(np.uint32), o.process_buffer(x[5, :300]).view(np.uint32))     # stream 5 started over
        o = po.IIR(b, a)
        o.process_buffer(x[6])
        assert np.array_equal(y2[6].view(np.uint32), o.process_buffer(x[6, :300]).view(np.uint32))     # stream 6 carried on
        f.close()
