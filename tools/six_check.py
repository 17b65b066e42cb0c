#!/usr/bin/env python3
"""six_check.py (GPU box): the seven-wave small-batch kernel (fsk_blk6.hip) against the four-wave kernel on the same synthetic
buffers -- decoded bytes, counts, 'eod' counts and the carried per-stream state words must be IDENTICAL (the same float
instruction sequence per decimated sample, whoever runs it) -- and their kernel times.

  tools/six_check.py [S:N[:workload[:chunks]] ...]     workload = c3 | c2 | idle | noisy | c4 | p3 (per-stream tone pairs); chunks = call lengths, '+'-separated
Each case runs in a child process under a timeout (a hand-off bug is a hung kernel, not an error code).  Diagnostic aid."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, zlib
sys.path.insert(0, %r)
import numpy as np, torch
import webaudio_modem_amd as wm
S, N, wl = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
chunks = [int(c) for c in sys.argv[4].split("+")] if sys.argv[4] != "-" else [N]
cfg = dict(baudRate=300, markFrequency=1070, spaceFrequency=1270) if wl == "c2" else dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
# per-stream tone pairs (round 6): c4 = BASELINE config #4 (300 baud, mark_s = 1000 + 10 (s mod 100), space_s = mark_s + 200), p3 = config #3's
# parameters with a tone pair per stream
if wl == "c4": cfg = [dict(baudRate=300, markFrequency=1000 + 10 * (s %% 100), spaceFrequency=1200 + 10 * (s %% 100)) for s in range(S)]
if wl == "p3": cfg = [dict(baudRate=1200, markFrequency=1200 + 7 * (s %% 13), spaceFrequency=2200 + 5 * (s %% 11)) for s in range(S)]
st = torch.cuda.current_stream().cuda_stream
x = torch.zeros((S, N), dtype=torch.float32, device="cuda")
g = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
if wl == "idle":
    fl = g.modulated_length(100)
    n0 = min(N, (400 + fl + 31) // 32 * 32)
    g.synth_device(x.data_ptr(), n0, N, 100, 0xF5C0DE, 400, 0.1, 1.0, st)
    torch.cuda.synchronize()
    import math
    g.add_awgn_device(x.data_ptr(), N, N, 30.0 - 10.0 * math.log10(N / float(fl)), 0xF5C0DE ^ 0xA36, st)
else:
    g.synth_device(x.data_ptr(), N, N, 32 if wl in ("c2", "c4") else 100, 0xF5C0DE, 10 * (160 if wl in ("c2", "c4") else 40), 0.1, 1.0, st)
    if wl == "noisy":
        g.add_awgn_device(x.data_ptr(), N, N, 10.0, 0xF5C0DE ^ 0xA36, st)
torch.cuda.synchronize()
g.close()
res = {}
for name, opts in (("four", {"kernel": "auto-r04"}), ("six", {"kernel": "seven-wave"})):
    opts = dict(opts)
    for kv in filter(None, os.environ.get("SIX_OPTS", "").split(",")):
        k, _, v = kv.partition("=")
        if name == "six" or not k.startswith("six"): opts[k] = v
    eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32, options=opts)
    op = eng.max_bytes(N)
    outs, kern = [], set()
    off = 0
    ci = 0
    while off < N:
        n = min(chunks[ci %% len(chunks)], N - off); ci += 1
        out = torch.zeros((S, op), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(S, dtype=torch.int32, device="cuda"); eod = torch.zeros(S, dtype=torch.int32, device="cuda")
        eng.demodulate_device(x.data_ptr() + 4 * off, n, N, out.data_ptr(), op, cnt.data_ptr(), eod.data_ptr(), 0, st)
        torch.cuda.synchronize()
        kern.add(eng.last_kernel().replace(" ", ""))
        outs.append((out.cpu().numpy(), cnt.cpu().numpy(), eod.cpu().numpy()))
        off += n
    rows = sorted(set(list(range(min(S, 70))) + list(np.linspace(0, S - 1, 40).astype(int))))
    state = [eng.debug_state(r) for r in rows]
    # timing: whole-buffer calls from wherever the state stands
    ms = None
    if os.environ.get("SIX_TIME", "1") != "0":
        out = torch.zeros((S, op), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(S, dtype=torch.int32, device="cuda")
        n16 = N // 16 * 16
        eng2 = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32, options=opts)
        eng2.demodulate_device(x.data_ptr(), n16, N, out.data_ptr(), op, cnt.data_ptr(), 0, 0, st); torch.cuda.synchronize()
        eng2.timing_begin()
        for _ in range(3): eng2.demodulate_device(x.data_ptr(), n16, N, out.data_ptr(), op, cnt.data_ptr(), 0, 0, st)
        torch.cuda.synchronize()
        nl, tms = eng2.timing_end(); ms = tms / nl
        kern.add("timed:" + eng2.last_kernel().replace(" ", ""))
        eng2.close()
    res[name] = (outs, state, kern, ms, eng.blk_lanes())
    eng.close()
a, b = res["four"], res["six"]
bad = []
for i, (u, v) in enumerate(zip(a[0], b[0])):
    if not np.array_equal(u[1], v[1]): bad.append("call %%d: counts differ in %%d streams (first %%d)" %% (i, int((u[1] != v[1]).sum()), int(np.nonzero(u[1] != v[1])[0][0])))
    elif not np.array_equal(u[0], v[0]): bad.append("call %%d: bytes differ in %%d streams" %% (i, int((u[0] != v[0]).any(axis=1).sum())))
    if not np.array_equal(u[2], v[2]): bad.append("call %%d: eod counts differ in %%d streams" %% (i, int((u[2] != v[2]).sum())))
ns = 0
for (ra, ia), (rb, ib) in zip(a[1], b[1]):
    if not (np.array_equal(np.asarray(ra).view(np.uint64), np.asarray(rb).view(np.uint64)) and np.array_equal(ia, ib)): ns += 1
if ns: bad.append("state words differ in %%d of %%d sampled streams" %% (ns, len(a[1])))
nb = int(sum(int(u[1].sum()) for u in a[0]))
t4, t6 = a[3], b[3]
print("RESULT %%s S=%%d N=%%d %%s chunks=%%s lanes=%%d bytes=%%d | four %%s %%s | six %%s %%s | %%s" %% (
    "OK " if not bad else "BAD", S, N, wl, sys.argv[4], b[4], nb,
    ("%%.3f ms %%.1f G/s" %% (t4, S * (N // 16 * 16) / t4 / 1e6)) if t4 else "-", sorted(a[2]),
    ("%%.3f ms %%.1f G/s" %% (t6, S * (N // 16 * 16) / t6 / 1e6)) if t6 else "-", sorted(b[2]),
    ("x%%.2f" %% (t4 / t6)) if t4 and t6 else ""))
for m in bad: print("   " + m)
''' % ROOT

specs = sys.argv[1:] or ["96:20000", "8192:96000", "4096:96000", "2048:96000", "1000:50000:c2", "8192:48000:idle", "300:40000:noisy:4096+1600+16+48+9000"]
rc = 0
for spec in specs:
    f = spec.split(":")
    S, N = f[0], f[1]
    wl = f[2] if len(f) > 2 else "c3"
    ch = f[3] if len(f) > 3 else "-"
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, S, N, wl, ch], capture_output=True, text=True, timeout=int(os.environ.get("SIX_TIMEOUT", "240")))
    except subprocess.TimeoutExpired:
        print("TIMEOUT %s" % spec, flush=True)
        rc = 1
        continue
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT") or l.startswith("   ")]
    if not lines:
        print("FAILED %s: %s" % (spec, r.stderr[-600:].replace("\n", " | ")), flush=True)
        rc = 1
        continue
    for l in lines:
        print(l, flush=True)
    if "RESULT OK" not in lines[0]:
        rc = 1
sys.exit(rc)
