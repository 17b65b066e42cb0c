#!/usr/bin/env python3
"""diag_replay.py (GPU box): one recorded stream through a chunk schedule, per kernel path, status after every call next to
the oracle's.  usage: diag_replay.py <x.npy> <cfg-json> <schedule comma list> [S]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from oracle import pyoracle as po
x = np.load(sys.argv[1]).astype(np.float32)
cfg = json.loads(sys.argv[2])
PER_STREAM = bool(cfg.pop("_per_stream", False))   # a list of S equal configs: the per-stream-constant kernel variants
sched = [int(v) for v in sys.argv[3].split(",")]
S = int(sys.argv[4]) if len(sys.argv) > 4 else 64
KEYS = ["frameStarted", "globalSampleCounter", "receivedBitsLength", "syncDetections", "silenceThreshold", "agcGain"]
def run(name, env, prec):
    os.environ.update(env)
    eng = wm.FSKEngine(S, [cfg] * S if PER_STREAM else cfg, precision=prec)
    for k in env: os.environ.pop(k)
    o = po.OracleCore(cfg)
    off = 0
    print("==", name)
    for n in sched + [len(x) - sum(sched)]:
        chunk = np.tile(x[off:off + n], (S, 1)).copy()
        out, eod = eng.demodulate_data(chunk)
        ob, oe = o.demodulate(x[off:off + n])
        st, ost = eng.get_status(S - 1), o.status()
        flag = "" if (out[S - 1] == ob and int(eod[S - 1]) == oe) else "   <<<<<< DIFF"
        print("  call %6d %6d %-22s gpu %s eod %d | oracle %s eod %d | %s%s" % (off, n, eng.last_kernel()[:22], out[S - 1].hex(), int(eod[S - 1]), ob.hex(), oe,
              " ".join("%s=%s/%s" % (k[:6], (round(st[k], 6) if isinstance(st[k], float) else st[k]), (round(ost[k], 6) if isinstance(ost[k], float) else int(ost[k]))) for k in KEYS), flag))
        off += n
    eng.close()
only = os.environ.get("REPLAY_ONLY", "generic,fused,pipe,blk").split(",")
if "generic" in only: run("generic f32", {"FSKHIP_FORCE_GENERIC": "1"}, wm.PRECISION_F32)
if "f64" in only: run("generic f64", {}, wm.PRECISION_F64)
if "fused" in only: run("fused", {"FSKHIP_SPLIT": "0"}, wm.PRECISION_F32)
if "pipe" in only: run("pipe", {"FSKHIP_SPLIT": "1"}, wm.PRECISION_F32)
