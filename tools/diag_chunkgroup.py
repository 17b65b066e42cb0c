#!/usr/bin/env python3
"""diag_chunkgroup.py (GPU box): per-call status of a 64-stream group, two-wave kernel next to the four-wave kernel, on the
ragged device-pointer schedule of tests/test_gpu_fullsize.py with the last call cut into chunks of argv[1] samples."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from test_gpu_fullsize import BELL
# the 64-stream group of tests/test_gpu_fullsize.py's 262144-stream batch in which stream 96606 lost an 'eod' with the
# four-wave kernel (round 3: the correction's hand-over was posted only 20 samples ahead while the discriminator wave can
# lead the back wave's tile by 23): regenerated here from the synthesiser
from test_gpu_fullsize import SEED
_S, N = 262144, 12000 // 128 * 128
_gen = wm.FSKEngine(_S, BELL, precision=wm.PRECISION_F32)
_d = _gen.device_malloc(_S * N * 4)
_gen.synth_device(_d, N, N, 20, SEED, 400, 0.1, 1.0)
_gen.synchronize()
x = np.empty((64, N), np.float32)
_gen.d2h(x, _d + 96576 * N * 4)
_gen.close()
S = 64
lane = 96606 - 96576
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pre = [1000, 17, 4096, 3, 128, 2049, 1000, 17]      # up to offset 8310
assert sum(pre) == 8310
last = [2] + [CH] * ((N - 8312) // CH) + ([(N - 8312) % CH] if (N - 8312) % CH else [])
logs = {}
for split in ("1", "4"):
    os.environ["FSKHIP_SPLIT"] = split
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    os.environ.pop("FSKHIP_SPLIT")
    d_x = eng.device_malloc(S * N * 4); eng.h2d(d_x, x)
    op = eng.max_bytes(N)
    d_out = eng.device_malloc(S * op); d_cnt = eng.device_malloc(S * 4); d_eod = eng.device_malloc(S * 4)
    cnt = np.zeros(S, np.uint32); eod = np.zeros(S, np.uint32)
    off = 0; log = []
    for n in pre + last:
        eng.demodulate_device(d_x + off * 4, n, N, d_out, op, d_cnt, d_eod)
        eng.synchronize()
        eng.d2h(cnt, d_cnt); eng.d2h(eod, d_eod)
        sts = [eng.get_status(l) for l in range(S)]
        log.append((off, n, [(int(eod[l]), int(cnt[l]), sts[l]["globalSampleCounter"], sts[l]["syncDetections"], sts[l]["frameStarted"], round(sts[l]["silenceThreshold"], 7)) for l in range(S)], eng.last_kernel().split("::")[-1][:16]))
        off += n
    logs[split] = log
    eng.close()
shown = 0
for a, b in zip(logs["1"], logs["4"]):
    d = [l for l in range(S) if a[2][l] != b[2][l]]
    if d or a[0] >= 8310 and shown < 0:
        print(a[0], a[1], a[3], b[3], "lanes differing:", d[:10])
        for l in d[:3]: print("    lane", l, "pipe", a[2][l], "blk", b[2][l])
        shown += 1
        if shown > 6: break
print("done; chunk", CH)
