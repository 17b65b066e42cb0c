#!/usr/bin/env python3
"""variants.py (GPU box): kernel time of one batch shape with alternative builds of libfskhip.so and/or environment
switches, one child process each, plus a checksum of the decoded bytes so that a variant that changes the result shows.

  tools/variants.py S N spec [spec ...]      spec = label[@libtag][:ENV=VAL[,ENV=VAL...]]
     libtag -> tools/build/libfskhip_<libtag>.so (tools/build_variant.sh); no tag = the shipped library
Diagnostic aid, not part of the suite."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, zlib
sys.path.insert(0, %r)
import webaudio_modem_amd._lib as L
if sys.argv[3] != "-": L.LIB_PATH = sys.argv[3]
import torch
import webaudio_modem_amd as wm
sys.path.insert(0, os.path.join(%r, "tools"))
import envopts  # (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
S, N = int(sys.argv[1]), int(sys.argv[2])
wl = os.environ.get("VAR_WORKLOAD", "c3")
cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200) if wl in ("c3", "idle") else dict(baudRate=300, markFrequency=1070, spaceFrequency=1270)
# per-stream tone pairs (round 6): c4 = BASELINE config #4, p3 = config #3's parameters with a tone pair per stream
if wl == "c4": cfg = [dict(baudRate=300, markFrequency=1000 + 10 * (s %% 100), spaceFrequency=1200 + 10 * (s %% 100)) for s in range(S)]
if wl == "p3": cfg = [dict(baudRate=1200, markFrequency=1200 + 7 * (s %% 13), spaceFrequency=2200 + 5 * (s %% 11)) for s in range(S)]
eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
st = torch.cuda.current_stream().cuda_stream
x = torch.empty((S, N), dtype=torch.float32, device="cuda")
op = eng.max_bytes(N)
out = torch.zeros((S, op), dtype=torch.uint8, device="cuda"); cnt = torch.empty(S, dtype=torch.int32, device="cuda")
if wl == "idle":      # bench.py --workload idle: one frame per stream, then a floor 30 dB under it
    import math, numpy as np
    fl = eng.modulated_length(100)
    n0 = min(N, (400 + fl + 31) // 32 * 32)
    x.zero_()
    eng.synth_device(x.data_ptr(), n0, N, 100, 0xF5C0DE, 400, 0.1, 1.0, st)
    torch.cuda.synchronize()
    ends = np.array([eng.synth_stream_params(0xF5C0DE, s_, 400, 0.1, 1.0)[0] for s_ in range(S)], np.int64) + fl
    c0 = int(ends.min())
    if c0 < n0:
        x[:, c0:n0] *= (torch.arange(c0, n0, device="cuda")[None, :] < torch.as_tensor(ends, device="cuda")[:, None])
    eng.add_awgn_device(x.data_ptr(), N, N, 30.0 - 10.0 * math.log10(N / float(fl)), 0xF5C0DE ^ 0xA36, st)
else:
    eng.synth_device(x.data_ptr(), N, N, 100 if wl in ("c3", "p3") else 32, 0xF5C0DE, int(os.environ.get("VAR_LEAD", "400")), 0.1, 1.0, st)
torch.cuda.synchronize()
def step():
    eng.demodulate_device(x.data_ptr(), N, N, out.data_ptr(), op, cnt.data_ptr(), 0, 0, st)
step(); torch.cuda.synchronize()     # first pass from the reset state: its bytes are the checksum
crc = zlib.crc32(out.cpu().numpy().tobytes()) ^ zlib.crc32(cnt.cpu().numpy().tobytes())
nb = int(cnt.sum().item())
step(); torch.cuda.synchronize()
eng.timing_begin()
for _ in range(int(os.environ.get("VAR_STEPS", "4"))): step()   # back to back, state carried on (as bench.py does)
torch.cuda.synchronize()
n, ms = eng.timing_end()
print("RESULT", ms / n, eng.last_kernel().replace(" ", "") + ("/%%d" %% eng.blk_lanes() if "blk" in eng.last_kernel() else ""), nb, "%%08x" %% crc)
if os.environ.get("VAR_STAMPS"):
    import ctypes, numpy as np
    six = "blk6" in eng.last_kernel()
    f = L.lib().fskdbg_read_stamps_blk6 if six else L.lib().fskdbg_read_stamps_blk if "blk" in eng.last_kernel() else L.lib().fskdbg_read_stamps
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    a = np.zeros((7 if six else 5 if "blk5" in eng.last_kernel() else 4, 2048, 8), np.uint64)
    NW = a.shape[0]
    assert f(a.ctypes.data, a.size) == 0
    g = min(2048, (S + eng.blk_lanes() - 1) // eng.blk_lanes() if eng.blk_lanes() else (S + 63) // 64)
    hw = a[:, :g, 2].astype(np.int64)
    simd = (hw >> 4) & 3
    if os.environ.get("VAR_STAMPS") == "3":   # which groups are slow: per compute unit (XCC, SE, SH, CU) and per SIMD
        import collections
        cu = ((hw >> 8) & 0xFF) | (((hw >> 32) & 7) << 8)
        tot3 = a[3, :g, 1].astype(np.float64) / N
        by = collections.defaultdict(list)
        for b in range(g): by[int(cu[3, b])].append(tot3[b])
        means = np.array([np.mean(v) for v in by.values()]); cnts = collections.Counter(len(v) for v in by.values())
        within = np.mean([np.max(v) - np.min(v) for v in by.values() if len(v) > 1])
        print("STAMP %%d compute units, groups per unit %%s; loop cycles/sample per unit: min %%.0f median %%.0f max %%.0f; mean spread inside a unit %%.0f"
              %% (len(by), dict(cnts), means.min(), np.median(means), means.max(), within))
        xcc = (hw[3] >> 32) & 7
        print("STAMP per XCC mean loop: " + " ".join("%%d:%%.0f" %% (x, tot3[xcc == x].mean()) for x in range(8) if (xcc == x).any()))
        # roles per (unit, simd)
        roles = collections.defaultdict(list)
        for w in range(NW):
            for b in range(g):
                if a[w, b, 1]: roles[(int(((hw[w, b] >> 8) & 0xFF) | (((hw[w, b] >> 32) & 7) << 8)), int((hw[w, b] >> 4) & 3))].append(w)
        pat = collections.Counter("".join(str(r) for r in sorted(v)) for v in roles.values())
        print("STAMP roles sharing a SIMD (sorted), how often: %%s" %% dict(pat.most_common(8)))
    if os.environ.get("VAR_STAMPS") == "2":   # where the waves of a group sit: SIMD id of wave 0..3 (HW_REG_HW_ID bits 5:4), first 16 groups
        print("STAMP simd of waves (first 16 groups): " + " ".join("".join(str(int(simd[w, b])) for w in range(NW) if a[w, b, 1]) for b in range(min(16, g))))
        cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
        import collections
        for w in range(NW):
            if a[w, :g, 1].max() == 0: continue
            cnt = collections.Counter(int(x) for x in simd[w, :g])
            print("STAMP wave %%d on SIMD 0..3: %%s" %% (w, [cnt.get(i, 0) for i in range(4)]))
    r0 = a[0, :g, 7].astype(np.float64)
    if r0.max() > 0:
        late = (r0 - r0.min()) / 100.0   # s_memrealtime ticks at 100 MHz -> microseconds
        print("STAMP loop start of wave 0 after the first group's: median %%.0f us, 90 %%%% %%.0f us, max %%.0f us; groups starting more than 1 ms late: %%d of %%d"
              %% (np.median(late), np.percentile(late, 90), late.max(), int((late > 1000).sum()), g))
    BW = 5 if six else 3
    if a[BW, :g, 3].sum():
        c = a[BW, :g, 3:7].astype(np.float64).sum(axis=0)
        print("STAMP back wave blocks: %%.0f per group; per sample because a lane is inside this wave's own span after a reset %%.1f %%%%, (unused) %%.1f %%%%, per sample for a rare event %%.1f %%%%"
              %% (c[0] / g, 100 * c[1] / c[0], 100 * c[2] / c[0], 100 * c[3] / c[0]))
    for w in range(NW):
        tot = a[w, :g, 1].astype(np.float64)
        if tot.max() == 0: continue
        wait = a[w, :g, 0].astype(np.float64)
        print("STAMP wave %%d: loop %%.0f cycles/sample (min %%.0f max %%.0f over groups), waiting %%.1f %%%% of it (min %%.1f max %%.1f), busy %%.0f cycles/sample"
              %% (w, tot.mean() / N, tot.min() / N, tot.max() / N, 100 * (wait / tot).mean(), 100 * (wait / tot).min(), 100 * (wait / tot).max(),
                 (tot - wait).mean() / N))
''' % (ROOT, ROOT)
S, N = int(sys.argv[1]), int(sys.argv[2])
for spec in sys.argv[3:]:
    head, _, envs = spec.partition(":")
    label, _, tag = head.partition("@")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    lib = os.path.join(ROOT, "tools", "build", "libfskhip_%s.so" % tag) if tag else "-"
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, str(S), str(N), lib], env=env, capture_output=True, text=True, timeout=int(os.environ.get("VAR_TIMEOUT", "300")))
    except subprocess.TimeoutExpired:
        print("%-28s TIMEOUT" % label, flush=True); continue
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print("%-28s FAILED %s" % (label, r.stderr[-400:].replace("\n", " | ")), flush=True); continue
    f = line[0].split()
    ms = float(f[1])
    print("%-28s %9.3f ms %8.1f Gsamples/s  %s bytes=%s crc=%s" % (label, ms, S * N / ms / 1e6, f[2], f[3], f[4]), flush=True)
    for l in r.stdout.splitlines():
        if l.startswith("STAMP"):
            print("    " + l, flush=True)
