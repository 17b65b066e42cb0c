"""Idle-regime chunking diagnosis (GPU box): one frame per stream, then a noise floor 30 dB under it; the same buffer through
several call schedules and kernels, every stream compared with the oracle.  usage: python tools/diag_idle.py [S] [N]"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

BELL = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
SEED = 0xF5C0DE + 21
payload, lead_max = 100, 400
gen = wm.FSKEngine(S, BELL)
frame_len = gen.modulated_length(payload)
n0 = (lead_max + frame_len + 31) // 32 * 32
d_x = gen.device_malloc(S * N * 4)
gen.synth_device(d_x, n0, N, payload, SEED, lead_max, 0.1, 1.0)
gen.synchronize()
row = np.empty(N, np.float32)
rng = np.random.RandomState(7)
x = np.zeros((S, N), np.float32)
for s in range(S):
    gen.d2h(row, d_x + s * N * 4)
    lead, _ = gen.synth_stream_params(SEED, s, lead_max, 0.1, 1.0)
    end = lead + frame_len
    x[s, :end] = row[:end]
    p = float(np.mean(row[lead:end].astype(np.float64) ** 2))
    x[s] += rng.normal(0.0, np.sqrt(p / 1000.0), N).astype(np.float32)
gen.h2d(d_x, x)


def run(schedule, opts, prec=wm.PRECISION_F32):
    eng = wm.FSKEngine(S, BELL, precision=prec, options=opts)
    out_pitch = eng.max_bytes(max(schedule))
    d_out = eng.device_malloc(S * out_pitch); d_cnt = eng.device_malloc(S * 4); d_eod = eng.device_malloc(S * 4)
    got = [bytearray() for _ in range(S)]
    eod_tot = np.zeros(S, np.int64)
    out = np.empty((S, out_pitch), np.uint8); cnt = np.empty(S, np.uint32); eod = np.empty(S, np.uint32)
    off = i = 0
    kernels = set()
    while off < N:
        n = min(schedule[i % len(schedule)], N - off)
        eng.demodulate_device(d_x + off * 4, n, N, d_out, out_pitch, d_cnt, d_eod)
        eng.synchronize()
        kernels.add(eng.last_kernel().split("<")[0])
        eng.d2h(cnt, d_cnt); eng.d2h(eod, d_eod)
        eod_tot += eod
        if cnt.any():
            eng.d2h(out, d_out)
            for s in np.nonzero(cnt)[0]:
                got[s] += out[s, :cnt[s]].tobytes()
        off += n; i += 1
    st = [eng.get_status(s) for s in range(S)]
    eng.close()
    return [bytes(g) for g in got], eod_tot, st, kernels


ref_b, ref_e = [], []
for s in range(S):
    ob, oe = po.OracleCore(BELL).demodulate(x[s])
    ref_b.append(ob); ref_e.append(oe)
ref_e = np.array(ref_e)
print("oracle: bytes", sum(len(b) for b in ref_b), "eod", int(ref_e.sum()))
for name, sched, opts in (("blk one call", [N], {"kernel": "four-wave"}), ("blk 48000", [48000], {"kernel": "four-wave"}), ("blk 128", [128], {"kernel": "four-wave"}),
                          ("blk 1024", [1024], {"kernel": "four-wave"}), ("blk 16", [16], {"kernel": "four-wave"}),
                          ("pipe one call", [N], {"kernel": "two-wave"}), ("pipe 128", [128], {"kernel": "two-wave"}),
                          ("fused 128", [128], {"kernel": "one-wave"}), ("fused one call", [N], {"kernel": "one-wave"}),
                          ("tail (generic off) 4097", [4097], {"kernel": "four-wave"}), ("generic", [N], {"force_generic": 1}),
                          ("f64", [N], None)):
    prec = wm.PRECISION_F64 if name == "f64" else wm.PRECISION_F32
    b, e, st, k = run(sched, opts, prec)
    bad_b = [s for s in range(S) if b[s] != ref_b[s]]
    bad_e = [s for s in range(S) if int(e[s]) != int(ref_e[s])]
    print("%-26s kernels %s: byte mismatches %d %s, eod mismatches %d %s" % (name, sorted(k), len(bad_b), bad_b[:6], len(bad_e), bad_e[:6]))
    for s in bad_e[:3]:
        print("      stream %d: eod %d vs oracle %d; bytes %d vs %d; syncDetections %d" % (s, e[s], ref_e[s], len(b[s]), len(ref_b[s]), st[s]["syncDetections"]))
