"""Pins oracle/next_oracle.py (CRC-16, XModem packets + receive grammar, ChunkedModulator, the FSKProcessor quantum
loop) against tests/golden/golden_next.npz, captured from the real reference classes under Node."""
import numpy as np
import pytest

from conftest import golden_next
from oracle import next_oracle as no
from oracle import pyoracle as po

STATUS = {v: k for k, v in no.XM_NAMES.items()}


def test_crc16_kats():
    g = golden_next()
    c = g.manifest["crc"]
    rows = g.ragged(c)
    vals = g.arrays[c["values"]]
    assert len(rows) == len(vals) >= 20
    for row, v in zip(rows, vals):
        assert no.crc16(row) == int(v)
    # the reference's own KATs (tests/utils/crc16.node.test.ts:12-60)
    assert no.crc16(b"") == 0xFFFF and no.crc16(b"A") == 0xB915 and no.crc16(b"123456789") == 0x29B1
    assert no.crc16(b"\x00") == 0xE1F0 and no.crc16(b"\xff") == 0xFF00 and no.crc16(b"\xaa\xaa") == 0xFB1A
    assert no.crc16(bytes(range(256))) == 0x3FBD
    assert c["verify_true"] is True and c["verify_false"] is False


def test_packets_serialize_like_reference():
    g = golden_next()
    p = g.manifest["packets"]
    payloads, wires = g.ragged(p["payload"]), g.ragged(p["wire"])
    for payload, wire, m in zip(payloads, wires, p["meta"]):
        pk = no.create_data(m["seq"], payload)
        assert pk["invSequence"] == m["inv"] and pk["checksum"] == m["crc"] and pk["length"] == m["len"]
        assert no.serialize(pk) == wire
        assert m["verify"] is True
    for e in p["errors"]:
        with pytest.raises(ValueError) as ei:
            no.create_data(e["seq"], bytes(e["len"]))
        assert str(ei.value) == e["error"]
    assert p["control"] == {"SOH": [no.SOH], "ACK": [no.ACK], "NAK": [no.NAK], "EOT": [no.EOT]}
    assert p["constants"]["MAX_PACKET_SIZE"] == 261 and p["constants"]["HEADER_SIZE"] == 4


def test_scan_grammar_matches_harness():
    g = golden_next()
    sc = g.manifest["scans"]
    bursts, datas = g.ragged(sc["bytes"]), g.ragged(sc["data"])
    assert len(bursts) == len(sc["cases"]) >= 45
    for burst, data, c in zip(bursts, datas, sc["cases"]):
        r = no.scan_burst(burst, c["expected"])
        assert no.XM_NAMES[r["status"]] == c["status"], c["name"]
        for k in ("expected_after", "packets", "dropped", "consumed", "err_seq", "err_len", "crc_rx", "crc_calc"):
            assert r[k] == c[k], (c["name"], k)
        assert r["data"] == data, c["name"]


def test_chunked_modulator_matches_reference():
    g = golden_next()
    for c in g.manifest["chunked"]:
        core = po.OracleCore(c["config"])
        cm = no.ChunkedModulator(core)
        assert cm.is_modulating() == c["pre"]["modulating"] and cm.get_progress() == c["pre"]["progress"]
        assert cm.get_next_samples(128) is None
        cm.start_modulation(bytes(c["payload"]))
        assert cm.is_modulating() == c["started"]
        direct = core.modulate(bytes(c["payload"]))
        assert len(direct) == c["total"]
        steps, got = [], []
        while True:
            r = cm.get_next_samples(c["chunk"])
            if r is None:
                break
            steps.append([len(r["signal"]), int(r["isComplete"]), r["samplesConsumed"], r["totalSamples"],
                          cm.get_progress(), int(cm.is_modulating())])
            got.append(r["signal"])
        assert len(steps) == c["n_steps"]
        keep = steps[:8] + steps[-8:] if c["steps_truncated"] else steps
        assert keep == c["steps"], c["name"]
        assert c["identical_to_direct"] and np.array_equal(np.concatenate(got), direct)
        assert cm.is_modulating() == c["after"]["modulating"] and cm.get_next_samples(c["chunk"]) is None
    m = g.manifest["chunked_misc"]
    core = po.OracleCore({})
    cm = no.ChunkedModulator(core)
    cm.start_modulation(b"")
    assert (cm.is_modulating(), cm.get_next_samples(128), cm.get_progress()) == (m["empty"]["modulating"], None, 0)
    cm.start_modulation(bytes([1, 2, 3]))
    cm.get_next_samples(128)
    assert cm.is_modulating() == m["mid"]["modulating"] and cm.get_progress() == m["mid"]["progress"]
    cm.cancel()
    assert (cm.is_modulating(), cm.get_next_samples(128), cm.get_progress()) == (False, None, 0)
    cm.start_modulation(bytes([1, 2, 3]))
    cm.get_next_samples(300)
    cm.start_modulation(bytes([9]))
    r = cm.get_next_samples(128)
    assert {"consumed": r["samplesConsumed"], "total": r["totalSamples"]} == m["restart"]
    cm.start_modulation(b"")
    assert cm.is_modulating() is False and cm.get_next_samples(128) is None


def rx_run_input(core_factory, run):
    """Rebuild the quantum-loop input of an rx run: lead zeros + back-to-back frames, zero padded to whole quanta
    (the modulator oracle is bit-identical to the reference's, pinned in test_oracle_golden.py)."""
    parts = [np.zeros(run["lead"], np.float32)]
    for p in run["payloads"]:
        parts.append(core_factory().modulate(bytes(p)))
    x = np.concatenate(parts)
    buf = np.zeros(run["quanta"] * 128, np.float32)
    buf[:x.size] = x
    return buf


@pytest.mark.parametrize("name", [r["name"] for r in golden_next().manifest["processor"]])
def test_processor_loop_matches_reference(name):
    g = golden_next()
    run = next(r for r in g.manifest["processor"] if r["name"] == name)
    if run["kind"] == "rx":
        buf = rx_run_input(lambda: po.OracleCore(run["config"]), run)
        proc = no.ProcessorOracle(po.OracleCore(run["config"]), rx_capacity=run["ring_capacity"])
        drains = {d["quantum"]: d["bytes"] for d in run["drains"]}
        probe = dict(map(tuple, run["length_probe"]))
        for q in range(run["quanta"]):
            proc.process(buf[q * 128:(q + 1) * 128], 128)
            if q in drains:
                assert list(proc.demodulate()) == drains[q], (name, q)
            if q in probe:
                assert proc.ring.length == probe[q], (name, q)
        assert list(proc.demodulate()) == drains[run["quanta"]]
    else:
        proc = no.ProcessorOracle(po.OracleCore(run["config"]))
        direct = po.OracleCore(run["config"]).modulate(bytes(run["payload"]))
        assert len(direct) == run["total"] and run["output_is_shifted_direct"]
        complete_at, out = -1, []
        for q in range(run["quanta"]):
            if q == run["start_quantum"]:
                proc.modulate(bytes(run["payload"]))
            before = proc.completed
            out.append(proc.process(None, 128))
            if proc.completed != before:
                complete_at = q
        assert complete_at == run["complete_at"]
        out = np.concatenate(out)
        want = np.zeros_like(out)
        k = run["start_quantum"] * 128
        want[k:k + len(direct)] = direct
        assert np.array_equal(out, want)


def test_v8_sin_restatement_is_bit_identical_to_node():
    """oracle/v8_sin.h (fdlibm's sin, the algorithm V8 ports) against Math.sin of the Node that ran the reference:
    5 000+ arguments over the modulator's phase range and the corners of the argument reduction, compared bit for bit."""
    g = golden_next()
    x, want = g.arrays[g.manifest["sin"]["x"]], g.arrays[g.manifest["sin"]["y"]]
    assert x.size > 5000
    got = po.v8_sin(x)
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    # beyond 2^19*pi/2 the restatement defers to the C library (Payne-Hanek range, not restated): allow 1 ulp there
    inside = np.abs(x) <= 823549.6
    assert not np.any(inside[bad]), (x[bad][:5], got[bad][:5], want[bad][:5])
    assert np.all(np.abs(got[bad] - want[bad]) <= 2.3e-16)
