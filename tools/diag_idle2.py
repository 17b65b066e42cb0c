"""(GPU box) the 64-stream group of tools/diag_idle.py's 2048 x 192000 idle batch that holds stream argv[1] (default 1562):
four-wave kernel next to the two-wave kernel, both in argv[2]-sample calls (default 128); per call and lane (eod of the call,
globalSampleCounter); prints the first calls where a lane differs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm  # noqa: E402

BELL = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
TARGET = int(sys.argv[1]) if len(sys.argv) > 1 else 1562
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 128
S_ALL, N = 2048, 192000
SEED = 0xF5C0DE + 21
payload, lead_max = 100, 400
g0 = TARGET // 64 * 64
gen = wm.FSKEngine(S_ALL, BELL)
frame_len = gen.modulated_length(payload)
n0 = (lead_max + frame_len + 31) // 32 * 32
d_all = gen.device_malloc(S_ALL * N * 4)
gen.synth_device(d_all, n0, N, payload, SEED, lead_max, 0.1, 1.0)
gen.synchronize()
row = np.empty(N, np.float32)
rng = np.random.RandomState(7)
x = np.zeros((64, N), np.float32)
for s in range(g0 + 64):
    noise = rng.normal(0.0, 1.0, N)
    if s < g0:
        continue
    gen.d2h(row, d_all + s * N * 4)
    lead, _ = gen.synth_stream_params(SEED, s, lead_max, 0.1, 1.0)
    end = lead + frame_len
    x[s - g0, :end] = row[:end]
    p = float(np.mean(row[lead:end].astype(np.float64) ** 2))
    x[s - g0] += (noise * np.sqrt(p / 1000.0)).astype(np.float32)
gen.device_free(d_all)
gen.close()
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "idle_group_%d.npy" % g0), x)
S = 64
logs = {}
for kern in ("two-wave", "four-wave"):
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": kern})
    d_x = eng.device_malloc(S * N * 4)
    eng.h2d(d_x, x)
    op = eng.max_bytes(CH)
    d_out = eng.device_malloc(S * op); d_cnt = eng.device_malloc(S * 4); d_eod = eng.device_malloc(S * 4)
    cnt = np.zeros(S, np.uint32); eod = np.zeros(S, np.uint32)
    log = []
    for off in range(0, N, CH):
        n = min(CH, N - off)
        eng.demodulate_device(d_x + off * 4, n, N, d_out, op, d_cnt, d_eod)
        eng.synchronize()
        eng.d2h(eod, d_eod)
        log.append(eod.copy())
    logs[kern] = np.array(log)
    print(kern, eng.last_kernel(), "eod totals of lane", TARGET - g0, int(logs[kern][:, TARGET - g0].sum()))
    eng.close()
a, b = logs["two-wave"], logs["four-wave"]
d = np.argwhere(a != b)
print("calls x lanes differing:", len(d))
for c, l in d[:12]:
    print("  call %d (samples %d..%d) lane %d: two-wave eod %d, four-wave eod %d" % (c, c * CH, c * CH + CH, l, a[c, l], b[c, l]))
