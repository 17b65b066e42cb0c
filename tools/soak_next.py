#!/usr/bin/env python3
"""Randomised differential soak of the SURVEY 8(f) rows (GPU box): FSKProcessor quantum loops with random quantum
sizes, TX starts through random masks, drains, resets and RX-ring capacities against oracle/next_oracle.ProcessorOracle;
XModem scans of random bursts in random layouts against next_oracle.scan_burst; batched FIR with random taps, chunking
and per-stream resets against the oracle's FIR.  Everything must match exactly (fp64 engines / fp64 FIR).
usage: python tools/soak_next.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import webaudio_modem_amd as wm  # noqa: E402
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from oracle import next_oracle as no  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

CONFIGS = [{}, dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200), dict(baudRate=2400, markFrequency=2400, spaceFrequency=4800),
           dict(baudRate=300, markFrequency=1070, spaceFrequency=1270), dict(parity="odd")]


def processor_round(rng):
    cfg = CONFIGS[int(rng.integers(len(CONFIGS)))]
    S = int(rng.choice([1, 3, 64, 65, 70]))
    cap = int(rng.choice([8, 48, 1024]))
    clear = bool(rng.integers(2))
    graph = bool(rng.integers(2))
    n_in = int(rng.choice([64, 128, 128, 256, 100, 77]))
    n_out = int(rng.choice([128, 128, 64, 200]))
    os.environ["FSKHIP_SPLIT"] = str(int(rng.integers(2)))
    eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F64)
    os.environ.pop("FSKHIP_SPLIT")
    proc = wm.FSKProcessorBatch(eng, rx_capacity=cap, clear_rx_on_tx_complete=clear, use_graph=graph)
    oracles = [no.ProcessorOracle(po.OracleCore(cfg), rx_capacity=cap, clear_rx_on_tx_complete=clear) for _ in range(S)]
    Q = int(rng.integers(60, 260))
    x = np.zeros((S, Q * n_in), np.float32)
    for s in range(S):
        sig = po.OracleCore(cfg).modulate(bytes(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8)))
        lead = int(rng.integers(0, 150)) * 2
        m = min(len(sig), Q * n_in - lead)
        if m > 0:
            x[s, lead:lead + m] = sig[:m]
    for q in range(Q):
        r = rng.random()
        if r < 0.04:
            pend = proc.tx_state()["pendingModulation"]
            mask = (rng.random(S) < 0.5) & ~pend
            payloads = [bytes(rng.integers(0, 256, int(rng.integers(0, 6)), dtype=np.uint8)) for _ in range(S)]
            proc.modulate(payloads, mask=list(mask))
            for s in range(S):
                if mask[s]:
                    oracles[s].modulate(payloads[s])
        elif r < 0.06:
            got = proc.demodulate()
            for s in range(S):
                assert got[s] == oracles[s].demodulate(), ("drain", cfg, S, cap, q, s)
        elif r < 0.07:
            t = int(rng.integers(S))
            proc.reset(t)
            oracles[t].ring.clear()
            oracles[t].pending = None
        use_in = rng.random() < 0.95
        use_out = rng.random() < 0.9
        out = proc.process(x[:, q * n_in:(q + 1) * n_in] if use_in else None, n_out if use_out else 0)
        for s in range(S):
            if use_out:
                want = oracles[s].process(x[s, q * n_in:(q + 1) * n_in] if use_in else None, n_out)
                assert np.array_equal(out[s], want), ("tx", cfg, S, q, s, n_in, n_out, graph)
            elif use_in:
                got, _ = oracles[s].core.demodulate(x[s, q * n_in:(q + 1) * n_in])
                for b in got:
                    oracles[s].ring.put(b)
    assert list(proc.rx_lengths()) == [o.ring.length for o in oracles], ("rxlen", cfg, S, cap)
    got = proc.demodulate()
    for s in range(S):
        assert got[s] == oracles[s].demodulate(), ("final drain", cfg, S, cap, s)
    assert list(proc.tx_state()["completed"]) == [o.completed for o in oracles]
    proc.close()
    eng.close()
    return S


def scan_round(rng):
    n = int(rng.integers(1, 400))
    bursts, expected = [], []
    for _ in range(n):
        seq = int(rng.integers(1, 256))
        expected.append(seq if rng.random() > 0.15 else int(rng.integers(1, 256)))
        parts = []
        for _k in range(int(rng.integers(0, 4))):
            if rng.random() < 0.3:
                parts.append(bytes(rng.integers(0, 256, int(rng.integers(0, 5)), dtype=np.uint8)))
            w = bytearray(no.serialize(no.create_data(seq, bytes(rng.integers(0, 256, int(rng.integers(0, 256)), dtype=np.uint8)))))
            if rng.random() < 0.2:
                w[int(rng.integers(len(w)))] ^= 1 << int(rng.integers(8))
            if rng.random() < 0.1:
                w = w[:int(rng.integers(len(w) + 1))]
            parts.append(bytes(w))
            if rng.random() < 0.85:
                seq = seq % 255 + 1
        if rng.random() < 0.3:
            parts.append(b"\x04")
        bursts.append(b"".join(parts))
    res = wm.scan_bursts(bursts, expected)
    for b, e, got in zip(bursts, expected, res):
        want = no.scan_burst(b, e)
        for k in ("status", "expected_after", "packets", "dropped", "consumed", "err_seq", "err_len", "crc_rx", "crc_calc", "data"):
            assert got[k] == want[k], ("scan", k, len(b), e)
    crcs = wm.crc16_batch(bursts)
    for b, c in zip(bursts[:50], crcs[:50]):
        assert int(c) == no.crc16(b)
    return n


def fir_round(rng):
    S = int(rng.choice([1, 5, 64, 67]))
    T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 51, 64, 65, 200]))
    taps = list(rng.standard_normal(T))
    f = wm.FIRFilterBatch(taps, S, precision=wm.PRECISION_F64)
    oracles = [po.FIR(taps) for _ in range(S)]
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.choice([1, 2, 3, 5, 63, 1000, 1024, 1025, 3000]))
        x = (rng.random((S, n)) * 2 - 1).astype(np.float32)
        y = f.processBuffer(x)
        for s in range(S):
            assert np.array_equal(y[s], oracles[s].process_buffer(x[s])), ("fir", S, T, n, s)
        if rng.random() < 0.3:
            t = int(rng.integers(S))
            f.reset(t)
            oracles[t].reset()
    f.close()
    return S


def main(budget=None, seed=None, max_rounds=None):
    if budget is None:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0x4E58
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    counts = {"processor": 0, "scan": 0, "fir": 0}
    rounds = 0
    while time.time() < t_end and (max_rounds is None or rounds < max_rounds):
        rounds += 1
        r = rng.random()
        if r < 0.6:
            counts["processor"] += processor_round(rng)
        elif r < 0.8:
            counts["scan"] += scan_round(rng)
        else:
            counts["fir"] += fir_round(rng)
    print("soak_next ok: seed %#x, stream-runs %s" % (seed, counts))
    return counts


if __name__ == "__main__":
    main()
