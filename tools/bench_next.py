#!/usr/bin/env python3
"""Measurement of the SURVEY.md 8(f) rows on one MI355X (not the headline metric; bench.py is).

  python tools/bench_next.py [--streams 262144] [--quanta 200]

Prints one JSON line per row: the FSKProcessor quantum loop (128-sample process() calls over all streams, with
and without the captured hipGraph), the batched FIR (fp64 parity path and fp32), the XModem scan and CRC-16
kernels.  Inputs are resident in HBM; times are HIP events on the launch stream.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=262144)
    ap.add_argument("--quanta", type=int, default=200)
    ap.add_argument("--fir-streams", type=int, default=16384)
    ap.add_argument("--fir-samples", type=int, default=48000)
    args = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build()
    import webaudio_modem_amd as wm
    from webaudio_modem_amd import _lib
    L = _lib.lib()
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    sh = st.cuda_stream

    def timed(fn, reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            fn()
            st.synchronize()
            a.record(st)
            for _ in range(reps):
                fn()
            b.record(st)
        b.synchronize()
        return a.elapsed_time(b) / reps  # ms

    # ---- FSKProcessor quantum loop ------------------------------------------------------------------
    S, n = args.streams, 128
    cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
    eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
    Q = 64  # distinct quanta of input kept resident; the loop cycles through them
    x = torch.empty((S, Q * n), dtype=torch.float32, device="cuda")
    eng.synth_device(x.data_ptr(), Q * n, Q * n, 100, 0xF5C0DE, 400, 0.1, 1.0, sh)
    out = torch.empty((S, n), dtype=torch.float32, device="cuda")
    st.synchronize()
    for use_graph in (False, True):
        for tx_frac in (0.0, 0.25):
            proc = wm.FSKProcessorBatch(eng, use_graph=use_graph)
            if tx_frac:
                r = np.random.default_rng(1)
                mask = r.random(S) < tx_frac
                payload = bytes(range(128))
                pl = np.zeros((S, 128), np.uint8)
                pl[:] = np.frombuffer(payload, np.uint8)
                lens = np.full(S, 128, np.uint32)
                m = mask.astype(np.uint8)
                _lib.check(L.fskhip_processor_modulate_host(proc._h, pl.ctypes.data, lens.ctypes.data, 128, m.ctypes.data))
            state = {"q": 0}

            def quantum():
                q = state["q"] % Q
                state["q"] += 1
                # with the graph the input pointer must stay fixed: quantum 0 is replayed (same work per launch)
                off = 0 if use_graph else q * n * 4
                proc.process_device(x.data_ptr() + off, n, Q * n, out.data_ptr(), n, n, sh)

            ms = timed(quantum, args.quanta)
            rate = S * n / ms / 1e3
            print(json.dumps({"row": "f1 FSKProcessor.process() per 128-sample quantum", "streams": S, "graph": use_graph,
                              "tx_active_fraction": tx_frac, "ms_per_quantum": round(ms, 4), "Msamples_per_s": round(rate, 1),
                              "realtime_48k_streams": int(S * (128 / 48000 * 1e3) / ms)}))
            proc.close()
    eng.close()
    del x, out

    # ---- FIR -------------------------------------------------------------------------------------------
    Sf, Nf = args.fir_streams, args.fir_samples
    xin = (torch.rand((Sf, Nf), device="cuda") * 2 - 1).contiguous()
    yout = torch.empty_like(xin)
    taps = wm.FilterDesign.sincLowpass(1000, 48000, 51)
    for prec, name in ((wm.PRECISION_F64, "f64"), (wm.PRECISION_F32, "f32")):
        f = wm.FIRFilterBatch(taps, Sf, precision=prec)
        ms = timed(lambda: f.process_device(xin.data_ptr(), Nf, Nf, yout.data_ptr(), Nf, sh), 10)
        gbs = 8.0 * Sf * Nf / ms / 1e6
        print(json.dumps({"row": "f3 FIRFilter.processBuffer, 51 taps", "dtype": name, "streams": Sf, "samples": Nf,
                          "ms": round(ms, 3), "Msamples_per_s": round(Sf * Nf / ms / 1e3, 1),
                          "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}))
        f.close()
    # ---- IIR (round 5): the biquad low-pass the demodulator is wired from, as the generic batched IIRFilter ------------
    bw = wm.FilterDesign.butterworthLowpass(1200, 48000)
    for prec, name in ((wm.PRECISION_F64, "f64"), (wm.PRECISION_F32, "f32")):
        f = wm.IIRFilterBatch(bw["b"], bw["a"], Sf, precision=prec)
        ms = timed(lambda: f.process_device(xin.data_ptr(), Nf, Nf, yout.data_ptr(), Nf, sh), 10)
        gbs = 8.0 * Sf * Nf / ms / 1e6
        print(json.dumps({"row": "f3' IIRFilter.processBuffer, order 2", "dtype": name, "streams": Sf, "samples": Nf,
                          "ms": round(ms, 3), "Msamples_per_s": round(Sf * Nf / ms / 1e3, 1),
                          "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}))
        f.close()
    del xin, yout
    # (a lane per stream: 16 384 streams are one wave per compute unit; a full device)
    Sb = 65536
    xb = torch.randn((Sb, Nf), dtype=torch.float32, device="cuda")
    yb = torch.empty_like(xb)
    for prec, name in ((wm.PRECISION_F64, "f64"), (wm.PRECISION_F32, "f32")):
        f = wm.IIRFilterBatch(bw["b"], bw["a"], Sb, precision=prec)
        ms = timed(lambda: f.process_device(xb.data_ptr(), Nf, Nf, yb.data_ptr(), Nf, sh), 10)
        gbs = 8.0 * Sb * Nf / ms / 1e6
        print(json.dumps({"row": "f3' IIRFilter.processBuffer, order 2", "dtype": name, "streams": Sb, "samples": Nf,
                          "ms": round(ms, 3), "Msamples_per_s": round(Sb * Nf / ms / 1e3, 1),
                          "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}))
        f.close()
    del xb, yb

    # ---- XModem scan / CRC ------------------------------------------------------------------------------
    Sx = args.streams
    r = np.random.default_rng(7)
    payload_len = 128
    pl = torch.randint(0, 256, (Sx, payload_len), dtype=torch.uint8, device="cuda")
    lens = torch.full((Sx,), payload_len, dtype=torch.int32, device="cuda")
    seqs = torch.randint(1, 256, (Sx,), dtype=torch.int32, device="cuda")
    pitch = 144  # 16-byte rows: the tiled scan kernel applies
    wire = torch.zeros((Sx, pitch), dtype=torch.uint8, device="cuda")
    wlen = torch.zeros((Sx,), dtype=torch.int32, device="cuda")
    ms_ser = timed(lambda: _lib.check(L.fskhip_xmodem_serialize_device(pl.data_ptr(), payload_len, lens.data_ptr(), seqs.data_ptr(),
                                                                        Sx, wire.data_ptr(), pitch, wlen.data_ptr(), sh)), 10)
    data = torch.zeros((Sx, pitch), dtype=torch.uint8, device="cuda")
    res = torch.zeros((Sx, 10), dtype=torch.int32, device="cuda")
    ms_scan = timed(lambda: _lib.check(L.fskhip_xmodem_scan_device(wire.data_ptr(), pitch, wlen.data_ptr(), seqs.data_ptr(), Sx,
                                                                   data.data_ptr(), pitch, res.data_ptr(), sh)), 10)
    crc = torch.zeros((Sx,), dtype=torch.int16, device="cuda")
    ms_crc = timed(lambda: _lib.check(L.fskhip_crc16_device(pl.data_ptr(), payload_len, lens.data_ptr(), Sx, crc.data_ptr(), sh)), 10)
    st.synchronize()
    rr = res.cpu().numpy()
    ok = int(((rr[:, 0] == 0) & (rr[:, 2] == 1)).sum())
    same = bool(torch.equal(data[:, :payload_len], pl))
    for row, ms, nbytes in (("f2 XModemPacket.serialize", ms_ser, Sx * (payload_len + payload_len + 6)),
                            ("f2 XModem receive scan", ms_scan, Sx * (payload_len + 6 + payload_len)),
                            ("f2 CRC16.calculate", ms_crc, Sx * payload_len)):
        print(json.dumps({"row": row, "rows": Sx, "payload_bytes": payload_len, "ms": round(ms, 4),
                          "Mpackets_per_s": round(Sx / ms / 1e3, 1), "algorithmic_GBps": round(nbytes / ms / 1e6, 1),
                          "scan_accepted": ok, "payloads_identical": same}))


if __name__ == "__main__":
    main()
