#!/usr/bin/env python3
"""hostile_check.py (GPU box): every hostile-input golden (tests/golden/golden_hostile.npz: NaN / Inf / out-of-range / subnormal
samples through the REAL reference) through the engines -- alone and as stream 5 of a 64-stream batch whose other streams carry
a clean signal -- printing bytes / eod / status against the reference.  Each engine run in a child process under a timeout (a
kernel that cannot take a NaN must not hang the box).  Diagnostic aid; the assertions live in tests/test_gpu_hostile.py.
  tools/hostile_check.py [case-substring ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import webaudio_modem_amd as wm
from conftest import run_chunked
man = json.load(open(os.path.join(%r, "tests/golden/manifest_hostile.json")))
arr = np.load(os.path.join(%r, "tests/golden/golden_hostile.npz"))
name, prec, kern = sys.argv[1], sys.argv[2], sys.argv[3]
c = [c for c in man["cases"] if c["name"] == name][0]
x = arr[c["input"]]
clean = arr[[k for k in man["cases"] if k["name"] == name.split("_")[0] + "_" + name.split("_")[1] + "_clean"][0]["input"]]
opts = None if kern == "-" else {"kernel": kern}
P = wm.PRECISION_F64 if prec == "f64" else wm.PRECISION_F32
# alone
e = wm.FSKEngine(1, c["config"], precision=P, options=opts)
out, eod, nonempty, n = run_chunked(lambda a: (lambda o, ee: (o[0], int(ee[0])))(*e.demodulate_data(a[None, :].copy())), x, c["chunk"])
st = e.get_status(0)
flt = e.fault(0) if hasattr(e, "fault") else None
e.close()
# in a batch of 64 with clean neighbours
S = 64
n_ = max(x.size, clean.size)
X = np.zeros((S, n_), np.float32); X[:, :clean.size] = clean; X[5, :] = 0; X[5, :x.size] = x
e = wm.FSKEngine(S, c["config"], precision=P, options=opts)
o2, e2 = e.demodulate_data(X.copy())
e.close()
e = wm.FSKEngine(S, c["config"], precision=P, options=opts)
Xc = X.copy(); Xc[5, :] = 0; Xc[5, :clean.size] = clean
o3, e3 = e.demodulate_data(Xc)
e.close()
neigh = all(o2[s] == o3[s] and e2[s] == e3[s] for s in range(S) if s != 5)
sv = arr[c["status_vector"]]
okb = list(out) == c["bytes"]; oke = eod == c["eod_total"]
oks = bool(st["frameStarted"]) == bool(sv[0]) and st["globalSampleCounter"] == sv[1] and st["syncDetections"] == sv[5]
print("RESULT %%s %%-6s %%-12s bytes %%s %%r (ref %%r) eod %%s %%d (ref %%d) status %%s batch-of-64: same %%s neighbours %%s fault %%s" %% (
    name, prec, kern, "ok " if okb else "BAD", bytes(out), bytes(c["bytes"]), "ok " if oke else "BAD", eod, c["eod_total"], "ok " if oks else "BAD(%%s/%%s)" %% (st, list(sv)),
    "ok " if (bytes(o2[5]) == bytes(out) and int(e2[5]) == eod) else "BAD", "ok " if neigh else "BAD", flt))
''' % (ROOT, ROOT, ROOT, ROOT)
man = json.load(open(os.path.join(ROOT, "tests/golden/manifest_hostile.json")))
pats = sys.argv[1:]
variants = [("f64", "-"), ("f32", "-"), ("f32", "four-wave"), ("f32", "two-wave"), ("f32", "one-wave")]
for c in man["cases"]:
    if pats and not any(p in c["name"] for p in pats):
        continue
    for prec, kern in variants:
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, c["name"], prec, kern], capture_output=True, text=True, timeout=120)
        except subprocess.TimeoutExpired:
            print("TIMEOUT %s %s %s" % (c["name"], prec, kern), flush=True)
            continue
        l = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
        print(l[0] if l else "FAILED %s %s %s: %s" % (c["name"], prec, kern, r.stderr[-300:].replace("\n", " | ")), flush=True)
