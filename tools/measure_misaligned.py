"""(GPU box) what a tile path on dword-aligned (not 16-byte aligned) rows costs: BASELINE config #3's batch, 2 s per stream, one
call from a 16-byte aligned buffer against the same samples behind an odd-length call (the head realigns the decimator: the
tiles then start 4, 8 or 12 bytes off a 16-byte boundary).  usage: python tools/measure_misaligned.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webaudio_modem_amd as wm  # noqa: E402

BELL = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
S, N, P = 65536, 96000, 96064
gen = wm.FSKEngine(S, BELL)
d_x = gen.device_malloc(S * P * 4)
gen.synth_device(d_x, N + 32, P, 100, 0xF5C0DE, 400, 0.1, 1.0)
gen.synchronize()
op = gen.max_bytes(N)
d_out = gen.device_malloc(S * op); d_cnt = gen.device_malloc(S * 4); d_eod = gen.device_malloc(S * 4)
for first in (0, 1, 2, 3, 6, 17):
    eng = wm.FSKEngine(S, BELL)
    off = 0
    if first:
        eng.demodulate_device(d_x, first, P, d_out, op, d_cnt, d_eod)
        off = first
    eng.synchronize()
    best = 1e9
    for rep in range(3):
        eng.timing_begin()
        eng.demodulate_device(d_x + (off + rep * 0) * 4, N, P, d_out, op, d_cnt, d_eod)
        n, ms = eng.timing_end()
        best = min(best, ms)
        off_next = off + N
        # (keep the same alignment for every repetition: rewind by re-creating the engine would change nothing; the
        # stream content does not matter for the rate)
    print("after a %2d-sample call: tiles start %2d bytes off a 16-byte boundary, %s: %.2f ms = %.1f Gsamples/s"
          % (first, (4 * off) % 16, eng.last_kernel().split("::")[-1], best, S * N / best / 1e6))
    eng.close()
