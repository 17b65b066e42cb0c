import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import webaudio_modem_amd as wm
from state_fields import REAL, INT
import test_gpu_fullsize as T
S, payload = 64, 12
BELL = T.BELL; SEED = T.SEED
gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
fl = gen.modulated_length(payload)
N = (400 + fl + 2400 + 31) // 32 * 32
n0 = (400 + fl + 31) // 32 * 32
d_x = gen.device_malloc(S * N * 4)
gen.synth_device(d_x, n0, N, payload, SEED + 91, 400, 0.3, 1.0); gen.synchronize()
row = np.empty(N, np.float32); rng = np.random.RandomState(17); xh = np.zeros((S, N), np.float32); ends = np.zeros(S, np.int64)
for s_ in range(S):
    gen.d2h(row, d_x + s_ * N * 4)
    lead, _ = gen.synth_stream_params(SEED + 91, s_, 400, 0.3, 1.0)
    ends[s_] = lead + fl
    xh[s_, :ends[s_]] = row[:ends[s_]]
    p = float(np.mean(row[lead:ends[s_]].astype(np.float64) ** 2))
    xh[s_] += rng.normal(0.0, np.sqrt(p / 1e4), N).astype(np.float32)
gen.h2d(d_x, xh)
first = int(ends.min()) + 150
import collections
for cut in list(range(first, first + 900, 2)):
    if (cut - first) % 100 == 0: print('cut', cut, flush=True)
    st = {}
    for name in ("two-wave", "four-wave", "seven-wave", "one-wave"):
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": name})
        T._demod_schedule(eng, d_x, cut, N, [cut])
        st[name] = [eng.debug_state(s_) for s_ in range(S)]
        eng.close()
    for name in ("four-wave", "seven-wave", "one-wave"):
        names = collections.Counter()
        for s_ in range(S):
            (ra, ia), (rb, ib) = st["two-wave"][s_], st[name][s_]
            for i in range(len(ra)):
                if REAL[i].startswith("zq_") and REAL[i] not in ("zq_0i", "zq_0q") and ia[INT.index("zr_dph")] < 50: continue   # dead while the direct instance runs
                if np.float64(ra[i]).view(np.uint64) != np.float64(rb[i]).view(np.uint64): names[REAL[i]] += 1
            for i in range(len(ia)):
                if ia[i] != ib[i]: names[INT[i]] += 1
        if names: print(cut, name, dict(names))
cut = first
st = {}
for name in ("two-wave", "four-wave", "seven-wave", "one-wave"):
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": name})
    T._demod_schedule(eng, d_x, cut, N, [cut])
    st[name] = [eng.debug_state(s_) for s_ in range(S)]
    eng.close()
iq = [REAL.index(n) for n in ("zq_ai", "zq_aq", "zq_bi", "zq_bq")]
idph = INT.index("zr_dph")
for s_ in range(S):
    a = st["two-wave"][s_]; b = st["four-wave"][s_]
    if any(np.float64(a[0][i]).view(np.uint64) != np.float64(b[0][i]).view(np.uint64) for i in iq):
        for name in st:
            r, ii = st[name][s_]
            print(s_, name, [r[i] for i in iq], "dph", ii[idph], "ends", ends[s_], "cut", cut)
