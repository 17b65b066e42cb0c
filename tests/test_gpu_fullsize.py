"""BASELINE.json's full stream counts on the GPU (-m gpu), checked through size-independent properties:

* chunking invariance -- one call over the whole buffer, 128-sample quanta (FSKProcessor's cadence) and a ragged
  chunk schedule must give the SAME bytes and eod counts for EVERY stream (the reference is a streaming state
  machine: fsk-demodulation.node.test.ts:363-398, 668-753 pin this for one stream);
* a checksum of per-stream checksums equal across the schedules, and determinism of a repeated run;
* the oracle on a strided sample of the same buffers (byte-identical);
* decoded bytes contain the synthesised payload for the streams the reference decodes.

Sizes: config #3's 65 536 streams (Bell-202) and 262 144 (the bench's four-waves-per-SIMD point), config #4's
32 768 per-stream tone pairs, config #5's 16 384 streams through 10 dB AWGN.  Device buffers only (ctypes + the C ABI's
own allocator), so the host never holds more than the sampled rows.
"""
import zlib

import numpy as np
import os
import sys
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BELL = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
SEED = 0xF5C0DE


def _demod_schedule(eng, d_x, N, pitch, schedule):
    """Run the device buffer through `eng` in the given chunk sizes; returns (list of bytes per stream, eod totals)."""
    S = eng.n_streams
    out_pitch = eng.max_bytes(max(schedule))
    d_out = eng.device_malloc(S * out_pitch)
    d_cnt = eng.device_malloc(S * 4)
    d_eod = eng.device_malloc(S * 4)
    got = [bytearray() for _ in range(S)]
    eod_tot = np.zeros(S, np.int64)
    out = np.empty((S, out_pitch), np.uint8)
    cnt = np.empty(S, np.uint32)
    eod = np.empty(S, np.uint32)
    off, i = 0, 0
    while off < N:
        n = min(schedule[i % len(schedule)], N - off)
        eng.demodulate_device(d_x + off * 4, n, pitch, d_out, out_pitch, d_cnt, d_eod)
        eng.synchronize()
        eng.d2h(cnt, d_cnt)
        eng.d2h(eod, d_eod)
        eod_tot += eod
        if cnt.any():
            assert int(cnt.max()) <= out_pitch
            eng.d2h(out, d_out)
            for s in np.nonzero(cnt)[0]:
                got[s] += out[s, :cnt[s]].tobytes()
        off += n
        i += 1
    for p in (d_out, d_cnt, d_eod):
        eng.device_free(p)
    return [bytes(g) for g in got], eod_tot


def _digest(rows, eod):
    per_stream = np.array([zlib.crc32(r) for r in rows], dtype=np.uint32)
    return zlib.crc32(per_stream.tobytes() + eod.astype(np.int64).tobytes())


@pytest.mark.parametrize("S,seconds", [(65536, 1.0), (81920, 0.5), (262144, 0.25)], ids=["c3_65536", "one_and_a_quarter_rounds_81920", "bench_262144"])
def test_full_size_chunking_invariance_and_oracle_sample(S, seconds, monkeypatch):
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    N = int(seconds * 48000) // 128 * 128
    pitch = N
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * pitch * 4)
    gen.synth_device(d_x, N, pitch, 20, SEED, 400, 0.1, 1.0)
    gen.synchronize()
    results = {}
    for name, schedule in (("one_call", [N]), ("quanta_128", [128]), ("ragged", [1000, 17, 4096, 3, 128, 2049])):
        if name == "quanta_128" and S > 65536:
            schedule = [256]
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        results[name] = _demod_schedule(eng, d_x, N, pitch, schedule)
        if name == "one_call":
            # beyond one round of resident workgroups: the block kernel in time slices if the call is long enough for
            # two of them, round 2's kernels otherwise; within one round: one workgroup per group
            k = eng.last_kernel()
            if S <= 65536:
                assert k.startswith("fsk::demod_blk_kernel") and k.endswith("false>"), k
            elif N // 16 > 768:
                assert k.startswith("fsk::demod_blk_kernel") and k.endswith("true>"), k
            else:
                assert "demod_blk_kernel" not in k, k
        eng.close()
    if S > 65536:
        # beyond one round of resident workgroups the one-call launch is persistent and time-sliced (fsk_blk.hip); the same
        # call with slicing off must give the same bytes
        monkeypatch.setenv("FSKHIP_SLICE_TILES", "off")
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        results["one_call_unsliced"] = _demod_schedule(eng, d_x, N, pitch, [N])
        assert "demod_blk_kernel" not in eng.last_kernel(), eng.last_kernel()      # (round 2's kernel for this size)
        eng.close()
        monkeypatch.delenv("FSKHIP_SLICE_TILES")
    base_rows, base_eod = results["one_call"]
    base_digest = _digest(base_rows, base_eod)
    for name, (rows, eod) in results.items():
        if _digest(rows, eod) != base_digest:
            bad = [s for s in range(S) if rows[s] != base_rows[s] or eod[s] != base_eod[s]]
            raise AssertionError("%s differs from one_call on %d streams, first %s" % (name, len(bad), bad[:5]))
    # determinism, and the other whole-tile kernel (one wave per group <-> two-wave split) on the same buffer
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    again = _demod_schedule(eng, d_x, N, pitch, [N])
    eng.close()
    assert _digest(*again) == base_digest
    monkeypatch.setenv("FSKHIP_SPLIT", "0" if S <= 65536 else "1")
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    other = _demod_schedule(eng, d_x, N, pitch, [N // 2 // 16 * 16, N])
    eng.close()
    monkeypatch.delenv("FSKHIP_SPLIT")
    assert _digest(*other) == base_digest
    # the oracle on a strided sample of the same buffers, and payload round trips
    row = np.empty(pitch, np.float32)
    decoded = 0
    sample = list(range(0, S, S // 48)) + [S - 1]
    for s in sample:
        gen.d2h(row, d_x + s * pitch * 4)
        ob, oe = po.OracleCore(BELL).demodulate(row[:N])
        assert base_rows[s] == ob, s
        assert int(base_eod[s]) == oe, s
        if gen.synth_payload(SEED, s, 0, 20) in base_rows[s]:
            decoded += 1
    assert decoded >= len(sample) * 0.6  # sanity only: the reference misses frames at some odd lead-ins, and so must we
    total = sum(1 for s in range(S) if len(base_rows[s]) >= 20)
    assert total >= S * 0.6, "only %d/%d streams produced a frame" % (total, S)
    gen.device_free(d_x)
    gen.close()


def test_config4_per_stream_tones_full_size():
    """BASELINE config #4: 32 768 streams, mark_s = 1000 + 10*(s mod 100), space_s = mark_s + 200, 300 baud."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S = 32768
    cfgs = [dict(baudRate=300, markFrequency=1000 + 10 * (s % 100), spaceFrequency=1200 + 10 * (s % 100)) for s in range(S)]
    N = 48000  # one 16-byte frame at 300 baud is 30 560 samples
    eng = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32)
    d_x = eng.device_malloc(S * N * 4)
    eng.synth_device(d_x, N, N, 16, SEED, 1600, 0.1, 1.0)
    eng.synchronize()
    rows, eod = _demod_schedule(eng, d_x, N, N, [N])
    eng2 = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32)
    rows2, eod2 = _demod_schedule(eng2, d_x, N, N, [128 * 7])
    assert _digest(rows, eod) == _digest(rows2, eod2)
    row = np.empty(N, np.float32)
    ok = 0
    sample = list(range(0, S, 331))
    for s in sample:
        eng.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(cfgs[s]).demodulate(row)
        assert rows[s] == ob and int(eod[s]) == oe, s
        ok += eng.synth_payload(SEED, s, 0, 16) in rows[s]
    assert ok >= len(sample) * 0.6  # sanity only (bytes were already required to equal the oracle's)
    eng.device_free(d_x)
    eng.close()
    eng2.close()


def test_config5_awgn_full_size():
    """BASELINE config #5: 16 384 streams modulate -> AWGN 10 dB -> demodulate, all on the GPU."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S, N, P = 16384, 48000, 40
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = eng.device_malloc(S * N * 4)
    eng.synth_device(d_x, N, N, P, SEED, 400, 0.1, 1.0)
    eng.add_awgn_device(d_x, N, N, 10.0, 0xA36)
    eng.synchronize()
    rows, eod = _demod_schedule(eng, d_x, N, N, [N])
    eng2 = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    rows2, eod2 = _demod_schedule(eng2, d_x, N, N, [128])
    assert _digest(rows, eod) == _digest(rows2, eod2)
    row = np.empty(N, np.float32)
    sample = list(range(0, S, 257))
    for s in sample:
        eng.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(BELL).demodulate(row)
        assert rows[s] == ob and int(eod[s]) == oe, s
    frames_ok = sum(1 for s in range(S) if eng.synth_payload(SEED, s, 0, P) in rows[s]) if S <= 4096 else \
        sum(1 for s in sample if eng.synth_payload(SEED, s, 0, P) in rows[s]) * S // len(sample)
    assert frames_ok >= S * 0.5  # sanity: at 10 dB the reference itself loses frames
    eng.device_free(d_x)
    eng.close()
    eng2.close()


@pytest.mark.parametrize("S", [1, 63, 65, 1000, 4096])
def test_whole_tile_kernels_agree_on_ragged_batches(S, monkeypatch):
    """The one-wave kernel, the two-wave split kernel and the generic kernel (FSKHIP_FORCE_GENERIC) must produce the same
    bytes / eod for every stream of a batch whose size is not a multiple of the wave, under a ragged chunk schedule."""
    import webaudio_modem_amd as wm
    N = 48000
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, 12, SEED + 1, 300, 0.1, 1.0)
    gen.synchronize()
    digests = {}
    for name, env in (("split", {"FSKHIP_SPLIT": "1"}), ("one_wave", {"FSKHIP_SPLIT": "0"}),
                      ("four_wave", {"FSKHIP_SPLIT": "4"}), ("seven_wave", {"FSKHIP_SPLIT": "6"}), ("generic", {"FSKHIP_FORCE_GENERIC": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        for k in env:
            monkeypatch.delenv(k)
        rows, eod = _demod_schedule(eng, d_x, N, N, [4096, 16, 1000, 48, 20000])
        digests[name] = (_digest(rows, eod), sum(len(r) for r in rows))
        eng.close()
    assert digests["split"] == digests["one_wave"] == digests["four_wave"] == digests["seven_wave"] == digests["generic"], digests
    assert digests["split"][1] >= 12 * S * 0.5
    gen.device_free(d_x)
    gen.close()


@pytest.mark.parametrize("S", [200, 4096])
def test_y_ring_depth_does_not_change_results(S, monkeypatch):
    """demod_blk_kernel's y ring (wave 0 -> wave 1) takes whatever LDS the batch leaves (fsk_blk.hip, demod_blk_plan): 6 half
    tiles at least, an odd number and the maximum included here; bytes and eod counts must not depend on it."""
    import webaudio_modem_amd as wm
    N = 48000
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, 12, SEED + 11, 300, 0.1, 1.0)
    gen.synchronize()
    digests = {}
    monkeypatch.setenv("FSKHIP_SPLIT", "c")      # (auto-r04: the four-wave kernel is what this test is about; small batches default to seven waves since round 5)
    for y in ("auto", "6", "7", "12", "28"):
        if y != "auto":
            monkeypatch.setenv("FSKHIP_BLK_YSLOTS", y)
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        monkeypatch.delenv("FSKHIP_BLK_YSLOTS", raising=False)
        rows, eod = _demod_schedule(eng, d_x, N, N, [20000, 4096, 16, 1000, 48, 999, 8000])
        assert "demod_blk_kernel" in eng.last_kernel()
        digests[y] = (_digest(rows, eod), sum(len(r) for r in rows))
        eng.close()
    assert len(set(digests.values())) == 1, digests
    assert digests["auto"][1] >= 12 * S * 0.4
    gen.device_free(d_x)
    gen.close()


@pytest.mark.parametrize("S,resident,slice_tiles", [(1000, 5, 1), (1000, 3, 7), (4096, 16, 64), (4096, 63, 3)])
def test_time_sliced_persistent_launch_matches_one_workgroup_per_group(S, resident, slice_tiles, monkeypatch):
    """Batches beyond one round of resident workgroups run demod_blk_kernel persistently over (group, time slice) items
    (fsk_blk.hip, BlkSched).  With the device "shrunk" to a few resident workgroups and slices as short as one tile (16
    samples: shorter than every lag of the ZIR hand-over), bytes and eod counts must equal the plain launch's for every
    stream, under a ragged call schedule, and the state handed to the next call must be the same (the second half of the
    buffer goes through the per-sample kernels of an odd-length schedule)."""
    import webaudio_modem_amd as wm
    N = 48000
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, 12, SEED + 7, 300, 0.1, 1.0)
    gen.synchronize()
    digests = {}
    for name, env in (("plain", {"FSKHIP_SLICE_TILES": "off", "FSKHIP_SPLIT": "c"}),      # ("c" = auto-r04: never the seven-wave kernel, which is never sliced)
                      ("sliced", {"FSKHIP_BLK_RESIDENT": str(resident), "FSKHIP_SLICE_TILES": str(slice_tiles), "FSKHIP_SPLIT": "c"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        for k in env:
            monkeypatch.delenv(k)
        d_out = eng.device_malloc(S * eng.max_bytes(20000))
        d_cnt = eng.device_malloc(S * 4)
        d_eod = eng.device_malloc(S * 4)
        eng.demodulate_device(d_x, 1600, N, d_out, eng.max_bytes(20000), d_cnt, d_eod)
        eng.synchronize()
        assert eng.last_kernel().endswith("true>" if name == "sliced" else "false>"), (name, eng.last_kernel())
        rows, eod = _demod_schedule(eng, d_x + 1600 * 4, N - 1600, N, [20000, 4096, 16, 1000, 48, 999, 8000])
        digests[name] = (_digest(rows, eod), sum(len(r) for r in rows))
        for p_ in (d_out, d_cnt, d_eod):
            eng.device_free(p_)
        eng.close()
    assert digests["plain"] == digests["sliced"], digests
    assert digests["plain"][1] >= 12 * S * 0.4
    gen.device_free(d_x)
    gen.close()


def test_time_sliced_launch_with_agc_write_back(monkeypatch):
    """FSKHIP_DEMOD_WRITEBACK_AGC through the time-sliced launch: wave 0 writes the AGC'd samples over its input slice by
    slice; the buffer afterwards and the decoded bytes must equal the plain launch's."""
    import webaudio_modem_amd as wm
    from webaudio_modem_amd._lib import DEMOD_WRITEBACK_AGC
    S, N = 1000, 32000
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_src = gen.device_malloc(S * N * 4)
    gen.synth_device(d_src, N, N, 12, SEED + 13, 300, 0.1, 1.0)
    gen.synchronize()
    src = np.empty((S, N), np.float32)
    gen.d2h(src, d_src)
    results = {}
    for name, env in (("plain", {"FSKHIP_SLICE_TILES": "off", "FSKHIP_SPLIT": "c"}), ("sliced", {"FSKHIP_BLK_RESIDENT": "4", "FSKHIP_SLICE_TILES": "9", "FSKHIP_SPLIT": "c"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        for k in env:
            monkeypatch.delenv(k)
        d_x = eng.device_malloc(S * N * 4)
        eng.h2d(d_x, src)
        op = eng.max_bytes(N)
        d_out, d_cnt, d_eod = eng.device_malloc(S * op), eng.device_malloc(S * 4), eng.device_malloc(S * 4)
        eng.demodulate_device(d_x, N, N, d_out, op, d_cnt, d_eod, DEMOD_WRITEBACK_AGC)
        eng.synchronize()
        assert eng.last_kernel().startswith("fsk::demod_blk_kernel<true,") and eng.last_kernel().endswith("true>" if name == "sliced" else "false>"), eng.last_kernel()
        x = np.empty((S, N), np.float32)
        out = np.empty((S, op), np.uint8)
        cnt = np.empty(S, np.uint32)
        eng.d2h(x, d_x); eng.d2h(out, d_out); eng.d2h(cnt, d_cnt)
        results[name] = (x, [out[s, :cnt[s]].tobytes() for s in range(S)])
        for p_ in (d_x, d_out, d_cnt, d_eod):
            eng.device_free(p_)
        eng.close()
    assert results["plain"][1] == results["sliced"][1]
    assert np.array_equal(results["plain"][0].view(np.uint32), results["sliced"][0].view(np.uint32))
    assert not np.array_equal(results["plain"][0], src)      # (the AGC did write)
    assert sum(len(r) for r in results["plain"][1]) >= 12 * S * 0.4
    gen.device_free(d_src)
    gen.close()


@pytest.mark.parametrize("split", ["0", "1", "4"])
def test_partial_wave_lanes_stay_out_of_rare_paths(split, monkeypatch):
    """Regression (found by tools/soak.py): one stream in a 64-lane wave, lowered syncThreshold, a chunk schedule that
    hands a synced state to the whole-tile kernels.  The 63 lanes beyond the batch used to reach the sync path, whose
    amplitude-column read took their out-of-range row index as an address and faulted.  Bytes must equal the oracle's
    (fp64: eod and status too)."""
    import os
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    x = np.load(os.path.join(os.path.dirname(__file__), "golden", "soak_partial_wave_sync.npy"))
    cfg = dict(syncThreshold=0.75)
    schedule = [(0, 128), (128, 128), (256, 128), (384, 3), (387, 3), (390, 16), (406, 1000), (1406, 6038)]
    monkeypatch.setenv("FSKHIP_SPLIT", split)
    for prec in (wm.PRECISION_F32, wm.PRECISION_F64):
        eng = wm.FSKEngine(1, cfg, precision=prec)
        o = po.OracleCore(cfg)
        for off, n in schedule:
            out, eod = eng.demodulate_data(np.ascontiguousarray(x[:, off:off + n]))
            ob, oe = o.demodulate(x[0, off:off + n])
            assert out[0] == ob
            if prec == wm.PRECISION_F64:
                assert int(eod[0]) == oe
        if prec == wm.PRECISION_F64:
            st, ost = eng.get_status(0), o.status()
            for k in ("frameStarted", "globalSampleCounter", "receivedBitsLength", "syncDetections"):
                assert st[k] == ost[k], k
        eng.close()


V21 = dict(baudRate=300, markFrequency=1070, spaceFrequency=1270)


def test_config2_v21_300_baud_batch(monkeypatch):
    """BASELINE config #2's workload as written: 4 096 uniform 300-baud streams, V.21 tones in the polarity the reference
    decodes (1070/1270), dsSPB = 80 (20 KB of polyphase registers per group in LDS).  Chunking invariance, both whole-tile
    kernels (three... two-wave pipeline <-> one wave per group), the oracle on a strided sample."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S, N, P = 4096, 120000, 24   # 2.5 s: two 24-byte frames (45 120 samples each) per stream
    gen = wm.FSKEngine(S, V21, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, P, SEED + 2, 1600, 0.1, 1.0)
    gen.synchronize()
    results = {}
    for name, env, schedule in (("pipe_one_call", {"FSKHIP_SPLIT": "1"}, [N]), ("pipe_quanta", {"FSKHIP_SPLIT": "1"}, [128]),
                                ("pipe_ragged", {"FSKHIP_SPLIT": "1"}, [4096, 19, 128, 48000, 7]),
                                ("fused_ragged", {"FSKHIP_SPLIT": "0"}, [30000, 17, 4096, 3, 128, 2049]),
                                ("blk_one_call", {"FSKHIP_SPLIT": "4"}, [N]), ("blk_quanta", {"FSKHIP_SPLIT": "4"}, [128]),
                                ("blk_ragged", {"FSKHIP_SPLIT": "4"}, [4096, 19, 128, 48000, 7, 30000, 17, 3, 2049]),
                                ("six_one_call", {"FSKHIP_SPLIT": "6"}, [N]), ("six_ragged", {"FSKHIP_SPLIT": "6"}, [4096, 19, 128, 48000, 7, 30000, 17, 3, 2049]),
                                ("auto_ragged", {}, [16, 4096, 21, 128, 48000, 5, 1024])):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = wm.FSKEngine(S, V21, precision=wm.PRECISION_F32)
        for k in env:
            monkeypatch.delenv(k)
        results[name] = _demod_schedule(eng, d_x, N, N, schedule)
        if name == "pipe_one_call":
            assert "demod_pipe_kernel" in eng.last_kernel()
        if name == "blk_one_call":
            assert "demod_blk_kernel" in eng.last_kernel()
        if name == "six_one_call":
            assert "demod_blk6_kernel" in eng.last_kernel()
        eng.close()
    base = _digest(*results["pipe_one_call"])
    for name, r in results.items():
        assert _digest(*r) == base, name
    rows, eod = results["pipe_one_call"]
    row = np.empty(N, np.float32)
    sample = list(range(0, S, 97)) + [S - 1]
    hit = 0
    for s in sample:
        gen.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(V21).demodulate(row)
        assert rows[s] == ob and int(eod[s]) == oe, s
        hit += gen.synth_payload(SEED + 2, s, 0, P) in rows[s]
    assert hit >= len(sample) * 0.6
    gen.device_free(d_x)
    gen.close()


def test_config3_full_length_480000_samples():
    """BASELINE config #3 at the bench's own size: 65 536 Bell-202 streams x 480 000 samples (10 s, 126 GB resident) in ONE
    call, against the oracle on a strided sample, plus a second engine fed the same buffer in 1 s calls (chunking invariance
    at full length: a checksum of per-stream checksums)."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S, N = 65536, 480000
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    try:
        d_x = gen.device_malloc(S * N * 4)
    except Exception as ex:  # a smaller GPU than the one the metric is defined on
        pytest.skip("cannot hold 126 GB: %s" % ex)
    gen.synth_device(d_x, N, N, 100, SEED, 400, 0.1, 1.0)
    gen.synchronize()
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    rows, eod = _demod_schedule(eng, d_x, N, N, [N])
    assert "demod_blk_kernel" in eng.last_kernel()
    eng.close()
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    rows2, eod2 = _demod_schedule(eng, d_x, N, N, [48000])
    eng.close()
    assert _digest(rows, eod) == _digest(rows2, eod2)
    row = np.empty(N, np.float32)
    for s in list(range(0, S, S // 16)) + [S - 1]:
        gen.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(BELL).demodulate(row)
        assert rows[s] == ob and int(eod[s]) == oe, s
    assert sum(len(r) for r in rows) >= S * 100 * 8   # ~11 frames of 100 bytes per stream, most of them decoded
    gen.device_free(d_x)
    gen.close()


def test_fp32_vs_fp64_engines_deviation_rate_full_size():
    """What the fp32 path's documented approximations amount to on BASELINE's own batch: the fp64
    engine (op for op with the reference) and the fp32 engine on the same 65 536 x 48 000 buffer, with and without 10 dB of
    noise.  The fp32 path decides the same slicer bits except where the post-filter output is within ~1e-7 of zero; a flipped
    bit only matters where the reference's own decision was marginal (a sync count exactly at its threshold, say).  Streams
    whose bytes / eod counts / status words differ are COUNTED and bounded, so the deviation is a measured rate, not an
    anecdote (round 2, clean batch: 1 stream of 65 536 -- the reference missed a frame by one matching tap, the fp32 path
    synchronised on it)."""
    import webaudio_modem_amd as wm
    S, N = 65536, 48000
    # bounds: what was ever measured on this batch (0 and 0 in rounds 3 and 4; one clean stream in round 2) with a margin of
    # one more -- VERDICT r03 weak #1: the old bounds (8 / 32) were 8-32 times looser than anything observed
    for snr, max_byte_diff in ((None, 1), (10.0, 4)):
        gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        d_x = gen.device_malloc(S * N * 4)
        gen.synth_device(d_x, N, N, 20, SEED + 5, 400, 0.1, 1.0)
        if snr is not None:
            gen.add_awgn_device(d_x, N, N, snr, 0xBEE)
        gen.synchronize()
        e32 = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        r32, eod32 = _demod_schedule(e32, d_x, N, N, [N])
        e64 = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F64)
        r64, eod64 = _demod_schedule(e64, d_x, N, N, [N])
        byte_diff = sum(1 for s in range(S) if r32[s] != r64[s])
        eod_diff = int((eod32 != eod64).sum())
        status_diff = 0
        for s in range(0, S, 257):
            a, b = e32.get_status(s), e64.get_status(s)
            status_diff += any(a[k] != b[k] for k in ("frameStarted", "globalSampleCounter", "receivedBitsLength", "syncDetections"))
        print("fp32 vs fp64, snr=%s: %d/%d streams differ in bytes, %d in eod count, %d/%d sampled in status"
              % (snr, byte_diff, S, eod_diff, status_diff, len(range(0, S, 257))))
        assert byte_diff <= max_byte_diff, (snr, byte_diff)
        assert eod_diff <= 8, (snr, eod_diff)     # (an 'eod' moved with identical bytes: a silence compare within 1e-6 of the threshold)
        for e in (e32, e64, gen):
            pass
        e32.close(); e64.close()
        gen.device_free(d_x)
        gen.close()


def test_short_odd_calls_do_not_take_the_engine_off_the_block_kernel():
    """VERDICT r03 #6.  A 6-sample call leaves the amplitude ring three pushes into a quad, a 17-sample call then leaves the
    /2 decimator mid-pair and the next buffer on a 4-byte boundary.  Round 3 then kept the engine on round 2's kernels (or
    the per-sample kernel) for the rest of its life; now the head of the next call realigns both (fsk_api.hip) and BASELINE
    config #3's full-length call runs on the block kernel: its bytes are those of one call over the whole stream and of
    the oracle."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S, N = 65536, 480000 + 6 + 17
    P = N + 1                                   # row pitch in floats: a multiple of four (the tile path's only layout condition)
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    try:
        d_x = gen.device_malloc(S * P * 4)
    except Exception as ex:
        pytest.skip("cannot hold 126 GB: %s" % ex)
    gen.synth_device(d_x, N, P, 100, SEED + 3, 400, 0.1, 1.0)
    gen.synchronize()
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    rows, eod = _demod_schedule(eng, d_x, N, P, [6, 17, 480000])
    assert eng.last_kernel().startswith("fsk::demod_blk_kernel"), eng.last_kernel()
    eng.close()
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    rows1, eod1 = _demod_schedule(eng, d_x, N, P, [N])
    eng.close()
    assert _digest(rows, eod) == _digest(rows1, eod1)
    row = np.empty(P, np.float32)
    for s in list(range(0, S, S // 8)) + [S - 1]:
        gen.d2h(row, d_x + s * P * 4)
        ob, oe = po.OracleCore(BELL).demodulate(row[:N])
        assert rows[s] == ob and int(eod[s]) == oe, s
    gen.device_free(d_x)
    gen.close()


@pytest.mark.parametrize("S", [64, 200])
def test_odd_call_lengths_and_unaligned_buffers_stay_on_the_fp32_arithmetic(S):
    """Calls of odd lengths leave a decimator pair open and the next call's buffer on a 4-byte boundary: the head / tile /
    tail dispatch (fsk_api.hip) must give exactly the bytes of one call, and of the oracle, whatever the cut."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    N = 20011
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * (N + 8) * 4)
    gen.synth_device(d_x, N, N + 8, 10, SEED + 9, 300, 0.2, 1.0)
    gen.synchronize()
    digests = set()
    for schedule in ([N], [1], [7, 33, 1, 128, 5001], [4097], [15, 17]):
        if schedule == [1] and S > 64:
            continue
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
        rows, eod = _demod_schedule(eng, d_x, N, N + 8, schedule)
        digests.add(_digest(rows, eod))
        eng.close()
    assert len(digests) == 1, digests
    row = np.empty(N + 8, np.float32)
    for s in (0, S // 2, S - 1):
        gen.d2h(row, d_x + s * (N + 8) * 4)
        ob, oe = po.OracleCore(BELL).demodulate(row[:N])
        assert rows[s] == ob and int(eod[s]) == oe
    gen.device_free(d_x)
    gen.close()


def test_one_process_several_engines_sharded_host_matches_one_engine():
    """webaudio_modem_amd/sharded.py (one host process, one engine per device, calls fanned out on threads): on this
    1-GPU box the three shards are three engines on device 0, driven concurrently from three threads -- the streams'
    bytes, eod counts and status must be those of one engine holding the whole batch, on any chunk schedule."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S = 200
    cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
    rng = np.random.default_rng(0x5AAD)
    o = po.OracleCore(cfg)
    sigs = [np.concatenate([np.zeros(int(rng.integers(0, 400)), np.float32),
                            o.modulate(bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))) * np.float32(rng.uniform(0.1, 1.0))])
            for _ in range(S)]
    N = max(len(v) for v in sigs) + 300
    x = np.zeros((S, N), np.float32)
    for s, v in enumerate(sigs):
        x[s, :len(v)] = v
    one = wm.FSKEngine(S, cfg)
    many = wm.FSKEngineSharded(S, cfg, devices=[0, 0, 0])
    assert [c for _, c, _ in many.shards] == [67, 67, 66]
    got_one, got_many = [b""] * S, [b""] * S
    eod_one, eod_many = np.zeros(S, np.int64), np.zeros(S, np.int64)
    off = 0
    for n in (1000, 17, 4096, 1, N):
        n = min(n, N - off)
        a, ea = one.demodulate_data(x[:, off:off + n].copy())
        b, eb = many.demodulate_data(x[:, off:off + n].copy())
        for s in range(S):
            got_one[s] += a[s]
            got_many[s] += b[s]
        eod_one += ea
        eod_many += eb
        off += n
    assert got_one == got_many and np.array_equal(eod_one, eod_many)
    assert sum(len(v) for v in got_one) > 10 * S
    for s in (0, 66, 67, 133, 134, 199):
        assert many.get_status(s) == one.get_status(s)
    many.reset(70)
    one.reset(70)
    assert many.get_status(70) == one.get_status(70)
    pay = [bytes([s % 251] * (1 + s % 7)) for s in range(S)]
    for u, v in zip(one.modulate_data(pay), many.modulate_data(pay)):
        assert np.array_equal(u, v)
    one.close()
    many.close()


@pytest.mark.parametrize("writeback", [False, True])
def test_host_entry_point_pipelined_over_time_slabs_gives_the_one_shot_result(writeback, monkeypatch):
    """fskhip_demodulate_host cuts a long call into time slabs and overlaps the next slab's H2D copy with the current
    slab's kernels (include/fskhip.h).  Bytes, eod counts, status and the written-back AGC samples must be those of the
    one-shot call -- from pageable and from page-locked (fskhip_host_alloc) buffers, with a slab length that does not
    divide the call."""
    import webaudio_modem_amd as wm
    S, N = 96, 30011
    gen = wm.FSKEngine(S, BELL)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, 24, SEED + 9, 400, 0.1, 1.0)
    gen.synchronize()
    x = np.zeros((S, N), np.float32)
    gen.d2h(x, d_x)
    gen.device_free(d_x)
    gen.close()

    def run(slab, pinned):
        monkeypatch.setenv("FSKHIP_HOST_SLAB", str(slab))
        e = wm.FSKEngine(S, BELL)
        buf = wm.pinned_empty((S, N), np.float32) if pinned else np.empty((S, N), np.float32)
        buf[:] = x
        out, eod = e.demodulate_data(buf, writeback_agc=writeback)
        st = [e.get_status(s) for s in (0, 17, 95)]
        e.close()
        return out, eod, st, buf.copy()

    ref = run(0, False)                       # no pipeline
    assert sum(len(b) for b in ref[0]) > 20 * S
    for slab, pinned in ((4096, False), (4096, True), (10000, True)):
        got = run(slab, pinned)
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
        assert np.array_equal(got[3], ref[3])  # the input, or the AGC-scaled samples when written back
    if writeback:
        assert not np.array_equal(ref[3], x)


def _full_length_check(S, cfg_of, N, payload, lead_max, oracle_streams, chunk):
    """one call over N samples vs the same buffer in `chunk`-sample calls (checksum of per-stream checksums), and the oracle on
    a strided sample of the one-call result"""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    per_stream = not isinstance(cfg_of, dict)
    cfgs = [cfg_of(s) for s in range(S)] if per_stream else cfg_of
    gen = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32)
    try:
        d_x = gen.device_malloc(S * N * 4)
    except Exception as ex:  # a smaller GPU than the one the metric is defined on
        pytest.skip("cannot hold %.0f GB: %s" % (S * N * 4 / 1e9, ex))
    gen.synth_device(d_x, N, N, payload, SEED + 11, lead_max, 0.1, 1.0)
    gen.synchronize()
    eng = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32)
    rows, eod = _demod_schedule(eng, d_x, N, N, [N])
    kernel = eng.last_kernel()
    eng.close()
    eng = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32)
    rows2, eod2 = _demod_schedule(eng, d_x, N, N, [chunk])
    eng.close()
    assert _digest(rows, eod) == _digest(rows2, eod2)
    row = np.empty(N, np.float32)
    hit = 0
    sample = list(range(0, S, max(1, S // oracle_streams))) + [S - 1]
    for s in sample:
        gen.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(cfgs[s] if per_stream else cfgs).demodulate(row)
        assert rows[s] == ob and int(eod[s]) == oe, s
        hit += gen.synth_payload(SEED + 11, s, 0, payload) in rows[s]
    assert hit >= len(sample) * 0.6
    gen.device_free(d_x)
    gen.close()
    return kernel, sum(len(r) for r in rows)


def test_config2_full_length_480000_samples():
    """BASELINE config #2 as written: 4 096 V.21-tone 300-baud streams x 480 000 samples (10 s) in ONE call, against the oracle
    on a strided sample, and against the same buffer in 1 s calls."""
    kernel, nbytes = _full_length_check(4096, V21, 480000, 32, 1600, 12, 48000)
    assert "demod_blk6_kernel" in kernel      # (round 5: 4 096 streams = 256 groups of 16: the seven-wave small-batch kernel)
    assert nbytes >= 4096 * 32 * 4          # eight 32-byte frames fit into 10 s; most of them decode


def test_config4_full_length_480000_samples():
    """BASELINE config #4 at full length: 32 768 streams with per-stream tone pairs (mark_s = 1000 + 10 (s mod 100), space_s
    = mark_s + 200, 300 baud) x 480 000 samples = 63 GB resident, one call vs 1 s calls, oracle (each stream its own
    configuration) on a strided sample."""
    kernel, nbytes = _full_length_check(
        32768, lambda s: dict(baudRate=300, markFrequency=1000 + 10 * (s % 100), spaceFrequency=1200 + 10 * (s % 100)),
        480000, 16, 1600, 12, 48000)
    assert "demod_blk_kernel<false, false," in kernel      # the per-stream-constant instantiation
    assert nbytes >= 32768 * 16 * 6


def test_config5_roundtrip_modulate_awgn_demodulate_full_length():
    """BASELINE config #5 as BASELINE.md defines it, at 16 384 streams x 480 000 samples: every frame comes from
    fskhip_modulate_device (FSKCore.modulateData), Gaussian noise at 10 dB with the reference tests' power definition
    (fsk-demodulation.node.test.ts:1184-1205), demodulated frame slot by frame slot; reports frame success rate and BER
    against the transmitted payloads, and requires the strided oracle sample to be byte-identical slot by slot.  Runs
    `bench.py --workload c5` itself in a child interpreter: bench.py holds its buffers in torch tensors."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--streams", "16384", "--steps", "1",
                        "--warmup", "0", "--cpu-seconds", "4"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]      # (exit 3 = a sampled stream differed from the oracle)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    q = line["config"]["c5_quality"]
    print("config #5:", q)
    assert line["config"]["samples_per_stream"] == 480000 and line["config"]["streams_per_gpu"] == 16384
    assert q["oracle_streams_checked"] >= 8 and q["oracle_streams_byte_identical"] == q["oracle_streams_checked"]
    assert line["cpu_baseline"]["parity_ok"]
    assert q["frames"] == 11 * 16384
    # what the reference's own algorithm delivers at this SNR with noise everywhere (gaps included: its AGC opens up in
    # them and false syncs add bytes to about four slots in ten): measured frame_success_rate 0.61, delivery 0.71, no bit
    # error in any length-matched frame.  The parity claim is the oracle comparison above; these are sanity floors.
    assert q["frame_success_rate"] >= 0.4 and q["frame_delivery_rate"] >= q["frame_success_rate"], q
    assert q["ber_on_length_matched_frames"] <= 1e-4, q


def test_idle_receiver_bank_one_frame_then_a_noise_floor():
    """VERDICT r03 #3, bench.py --workload idle: every stream carries ONE frame and then 4 s of a Gaussian floor 30 dB under
    it.  After its frame a stream fires 'eod' every samplesForEOD decimated samples and resets (fsk.ts:285-295, 175-188) for
    the rest of the call, each stream on its own schedule -- the regime in which the block kernel's back wave takes its
    per-sample path in nearly every tile.  One call, 1 s calls and the FSKProcessor's 128-sample quanta agree for every
    stream, and a strided sample is the oracle's bytes and eod counts exactly."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S, N, payload, lead_max = 2048, 192000, 100, 400
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    frame_len = gen.modulated_length(payload)
    n0 = (lead_max + frame_len + 31) // 32 * 32
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, n0, N, payload, SEED + 21, lead_max, 0.1, 1.0)     # frames back to back up to n0 ...
    gen.synchronize()
    head = np.zeros((S, n0), np.float32)
    row = np.empty(N, np.float32)
    rng = np.random.RandomState(7)
    x = np.zeros((S, N), np.float32)
    for s in range(S):
        gen.d2h(row, d_x + s * N * 4)
        lead, _amp = gen.synth_stream_params(SEED + 21, s, lead_max, 0.1, 1.0)
        end = lead + frame_len
        x[s, :end] = row[:end]                                               # ... of which each stream keeps its first
        p_frame = float(np.mean(row[lead:end].astype(np.float64) ** 2))
        x[s] += rng.normal(0.0, np.sqrt(p_frame / 1000.0), N).astype(np.float32)   # the floor: 30 dB under the frame
    gen.h2d(d_x, x)
    digests = []
    # (round 4: this is the regime the block path that takes resets -- demod_blk_kernel_r -- was built for; "auto" starts on
    # the other kernel and moves to it once a call's statistics are in)
    for schedule, resets in (([N], 0), ([N], 1), ([48000], "auto"), ([128], "auto"), ([4800], 1), ([16000], 2)):
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"blk_resets": resets, "kernel": "auto-r04"})
        rows, eod = _demod_schedule(eng, d_x, N, N, schedule)
        if schedule == [N]:
            assert eng.last_kernel().startswith("fsk::demod_blk_kernel_r<" if resets else "fsk::demod_blk_kernel<"), eng.last_kernel()
            rows1, eod1 = rows, eod
        if schedule == [48000]:
            assert eng.last_kernel().startswith("fsk::demod_blk_kernel_r<"), eng.last_kernel()      # three calls of floor behind it
        digests.append(_digest(rows, eod))
        eng.close()
    # round 5's own choice for a batch this small: seven waves (its frame wave takes own-span tiles on the block path with resets too);
    # and the seven-wave kernel pinned, on another call schedule
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    rows, eod = _demod_schedule(eng, d_x, N, N, [48000])
    assert "demod_blk6_kernel" in eng.last_kernel(), eng.last_kernel()      # (narrow groups: seven waves also where resets are frequent)
    digests.append(_digest(rows, eod))
    eng.close()
    eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": "seven-wave"})
    rows, eod = _demod_schedule(eng, d_x, N, N, [30000, 128, 17, 48000])
    assert "demod_blk6_kernel" in eng.last_kernel(), eng.last_kernel()
    digests.append(_digest(rows, eod))
    eng.close()
    assert len(set(digests)) == 1, digests
    hit = 0
    sample = list(range(0, S, S // 12)) + [S - 1]
    for s in sample:
        ob, oe = po.OracleCore(BELL).demodulate(x[s])
        assert rows1[s] == ob and int(eod1[s]) == oe, s
        hit += gen.synth_payload(SEED + 21, s, 0, payload) in rows1[s]
    assert hit >= len(sample) * 0.6      # (the reference itself loses a frame here and there: the bytes above are its own)
    assert int(np.median(eod1)) >= 100     # up to (192 000 - 42 000) / 2 / 140 = 535 'eod' events per stream; fewer where the AGC lifts the floor to the threshold
    gen.device_free(d_x)
    gen.close()


def test_config1_polarity_bank_never_syncs():
    """bench.py --workload c1x: BASELINE config #1's tone pair as written (mark 1270 / space 1070 Hz, the polarity the
    reference does not decode: SURVEY section 8d) on a batch -- every stream searches for its preamble through the whole
    call (fsk.ts:297-328) and decodes nothing, exactly as the oracle on a strided sample; any cut of the call agrees."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    cfg = dict(baudRate=300, markFrequency=1270, spaceFrequency=1070)
    S, N = 4096, 96000
    gen = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, 32, SEED + 23, 1600, 0.1, 1.0)
    gen.synchronize()
    digests = []
    for schedule in ([N], [7001, 128, 40000]):
        eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
        rows, eod = _demod_schedule(eng, d_x, N, N, schedule)
        digests.append(_digest(rows, eod))
        eng.close()
    assert digests[0] == digests[1]
    row = np.empty(N, np.float32)
    for s in list(range(0, S, S // 10)) + [S - 1]:
        gen.d2h(row, d_x + s * N * 4)
        ob, oe = po.OracleCore(cfg).demodulate(row)
        assert rows[s] == ob and int(eod[s]) == oe, s
    assert sum(len(r) for r in rows) <= S // 8       # (nothing, but for the odd false sync the reference has too)
    gen.device_free(d_x)
    gen.close()


@pytest.mark.parametrize("kernel", ["seven-wave", "four-wave", "two-wave", "one-wave"])
def test_agc_write_back_on_dword_aligned_tiles(kernel):
    """Round 4: after an odd-length call the whole tiles of the next start 4, 8 or 12 bytes off a 16-byte boundary (the head only
    realigns the decimator and the amplitude ring).  With FSKHIP_DEMOD_WRITEBACK_AGC the kernels also STORE 16 bytes per lane
    there (fsk.ts:55: the input buffer holds the AGC-scaled samples afterwards): the written-back buffer and the bytes must be
    those of one call, for every whole-tile kernel."""
    import webaudio_modem_amd as wm
    S, N = 200, 20011
    P = N + 1
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_src = gen.device_malloc(S * P * 4)
    gen.synth_device(d_src, N, P, 10, SEED + 31, 300, 0.2, 1.0)
    gen.synchronize()
    x = np.zeros((S, P), np.float32)
    gen.d2h(x, d_src)
    results = []
    for schedule in ([N], [1, 4096, 3, 8000, 2, 17, 7892]):
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options={"kernel": kernel})
        d_x = eng.device_malloc(S * P * 4)
        eng.h2d(d_x, x)
        op = eng.max_bytes(N)
        d_out = eng.device_malloc(S * op); d_cnt = eng.device_malloc(S * 4); d_eod = eng.device_malloc(S * 4)
        out = np.empty((S, op), np.uint8); cnt = np.empty(S, np.uint32)
        got = [bytearray() for _ in range(S)]
        off = 0
        kernels = set()
        for n in schedule:
            eng.demodulate_device(d_x + off * 4, n, P, d_out, op, d_cnt, d_eod, flags=wm.DEMOD_WRITEBACK_AGC)
            eng.synchronize()
            kernels.add(eng.last_kernel().split("<")[0])
            eng.d2h(cnt, d_cnt); eng.d2h(out, d_out)
            for s in range(S):
                got[s] += out[s, :cnt[s]].tobytes()
            off += n
        assert off == N
        y = np.empty((S, P), np.float32)
        eng.d2h(y, d_x)
        results.append(([bytes(g) for g in got], y[:, :N].copy(), kernels))
        eng.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])
    assert not np.array_equal(results[0][1], x[:, :N])                    # (something was written back)
    want = {"seven-wave": "demod_blk6_kernel", "four-wave": "demod_blk_kernel", "two-wave": "demod_pipe_kernel", "one-wave": "demod_fused_kernel"}[kernel]
    assert any(want in k for k in results[1][2]), results[1][2]
    gen.device_free(d_src)
    gen.close()


@pytest.mark.parametrize("per_stream", [False, True], ids=["uniform", "per-stream-tones"])
@pytest.mark.parametrize("S,lanes", [(2048, None), (4096, None), (700, "32"), (150, "64"), (96, "16")])
def test_seven_wave_kernel_is_the_four_wave_kernel_bit_for_bit(S, lanes, per_stream, monkeypatch):
    """Round 5: demod_blk6_kernel (fsk_blk6.hip) -- the small-batch kernel whose stages that are not recurrences run on the idle
    lanes of a narrow group and whose post filter runs ahead of the frame logic and is rewound after a reset -- against
    demod_blk_kernel on the same buffers: config #3's signal at 10 dB SNR on top of random lead-ins and levels, a ragged call
    schedule (whole tiles, odd lengths, 128-sample quanta: launches that begin and end inside a reset's own span, a hand-over
    or a frame).  Decoded bytes, per-call byte and 'eod' counts, and EVERY carried state word of every sampled stream must be
    identical: the float sequence per decimated sample is the same whoever evaluates it.
    Round 6: the same with per-stream tone pairs (demod_blk6_kernel<., ., false>: per-lane phasor rotation in the iq wave,
    per-lane phasors and lastPhase in the frame wave's block path with resets) against demod_blk_kernel<., false, .>."""
    import webaudio_modem_amd as wm
    N = 96000
    BELL = dict(globals()["BELL"])
    if per_stream:
        BELL = [dict(BELL, markFrequency=1200 + 7 * (s % 13), spaceFrequency=2200 + 5 * (s % 11)) for s in range(S)]
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, N, N, 40, SEED + 77, 400, 0.1, 1.0)
    gen.add_awgn_device(d_x, N, N, 10.0, SEED + 78)
    gen.synchronize()
    sched = [4096, 19, 128, 128, 30000, 7, 2049, 16, 48000, 3]
    res = {}
    for name, opts in (("four", {"kernel": "four-wave"}), ("six", {"kernel": "seven-wave"})):
        if lanes:
            opts = dict(opts, blk_lanes=lanes)
        eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options=opts)
        rows, eod = _demod_schedule(eng, d_x, N, N, sched)
        want = "demod_blk6_kernel" if name == "six" else "demod_blk_kernel"
        if per_stream and name == "six":
            want = ", false>"                                                     # (<write-back, streams per workgroup, uniform = false>)
        state = [eng.debug_state(s) for s in sorted(set(list(range(min(S, 80))) + list(range(0, S, 61)) + [S - 1]))]
        res[name] = (rows, eod, state)
        eng2 = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options=opts)      # one long call: the kernel meant is the one that runs
        _demod_schedule(eng2, d_x, N, N, [N])
        assert want in eng2.last_kernel(), eng2.last_kernel()
        assert _digest(*_demod_schedule(wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options=opts), d_x, N, N, [N])) == _digest(rows, eod)
        eng.close(); eng2.close()
    assert _digest(res["four"][0], res["four"][1]) == _digest(res["six"][0], res["six"][1])
    assert sum(len(r) for r in res["six"][0]) > 20 * S
    for (ra, ia), (rb, ib) in zip(res["four"][2], res["six"][2]):
        assert np.array_equal(np.asarray(ra).view(np.uint64), np.asarray(rb).view(np.uint64))
        assert ia == ib
    gen.device_free(d_x)
    gen.close()


def test_a_call_may_end_anywhere_in_the_span_after_a_reset():
    """Round 5: after resetState() the direct instance runs kDirectPairs decimated samples, the correction stays un-retired until
    zr_dph reaches kHandPairs -- with the back wave of the four-wave kernel for kOwnLag4 samples and with its discriminator wave
    for the rest, with the frame wave of the seven-wave kernel throughout (fsk_params.h).  One frame per stream, then silence: the
    'eod' resets every stream once, each at its own sample (random lead-ins).  A call boundary is swept in steps of two input
    samples over 260 decimated samples around those resets -- so that for some stream it falls on every value of zr_dph from 0 to
    beyond kHandPairs, on the hand-over sample and on the samples either side of it -- and what the four-wave and the seven-wave
    kernel carry over it (every state word of every stream) and decode must be what the two-wave kernel, which never hands the
    correction to another wave, carries and decodes."""
    import webaudio_modem_amd as wm
    S, payload = 64, 12
    gen = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32)
    fl = gen.modulated_length(payload)
    N = (400 + fl + 2400 + 31) // 32 * 32
    n0 = (400 + fl + 31) // 32 * 32
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, n0, N, payload, SEED + 91, 400, 0.3, 1.0)
    gen.synchronize()
    row = np.empty(N, np.float32)
    rng = np.random.RandomState(17)
    xh = np.zeros((S, N), np.float32)
    ends = np.zeros(S, np.int64)
    for s_ in range(S):
        gen.d2h(row, d_x + s_ * N * 4)
        lead, _amp = gen.synth_stream_params(SEED + 91, s_, 400, 0.3, 1.0)
        ends[s_] = lead + fl
        xh[s_, :ends[s_]] = row[:ends[s_]]                                       # nothing behind a stream's one frame ...
        p_frame = float(np.mean(row[lead:ends[s_]].astype(np.float64) ** 2))
        xh[s_] += rng.normal(0.0, np.sqrt(p_frame / 1e4), N).astype(np.float32)  # ... but a floor 40 dB under it
    gen.h2d(d_x, xh)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from state_fields import REAL, INT
    import re
    with open(os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_params.h")) as fh:
        direct_pairs = int(re.search(r"#define FSK_ZLAG (\d+)", fh.read()).group(1)) + 2              # kDirectPairs
    i_dph, dead = INT.index("zr_dph"), [REAL.index(n) for n in ("zq_ai", "zq_aq", "zq_bi", "zq_bq")]
    first = int(ends.min()) + 150
    cuts = list(range(first, first + 520 + int(ends.max() - ends.min()), 5))                # (both parities of the decimator; 64 lead-ins: every zr_dph many times)
    ref = {}
    eod_seen = 0
    for cut in cuts:
        res = {}
        for name, opts in (("two", {"kernel": "two-wave"}), ("four", {"kernel": "four-wave"}), ("seven", {"kernel": "seven-wave"})):
            eng = wm.FSKEngine(S, BELL, precision=wm.PRECISION_F32, options=opts)
            rows1, eod1 = _demod_schedule(eng, d_x, cut, N, [cut])
            at_cut = [eng.debug_state(s_) for s_ in range(S)]                        # what is carried over the boundary
            rows2, eod2 = _demod_schedule(eng, d_x + cut * 4, N - cut, N, [N - cut])
            rows = [bytes(a_) + bytes(b_) for a_, b_ in zip(rows1, rows2)]
            res[name] = (_digest(rows, eod1 + eod2), at_cut)
            eod_seen = max(eod_seen, int((eod1 + eod2).min()))
            eng.close()
        for name in ("four", "seven"):
            assert res[name][0] == res["two"][0], (cut, name)
            for s_, ((ra, ia), (rb, ib)) in enumerate(zip(res["two"][1], res[name][1])):
                ua, ub = np.asarray(ra).view(np.uint64), np.asarray(rb).view(np.uint64)
                assert np.array_equal(ua, ub), (cut, name, s_, [REAL[i] for i in np.nonzero(ua != ub)[0]])
                if ia[i_dph] < direct_pairs:                 # (dead while the direct instance runs: every kernel stores zeros)
                    assert not ua[dead].any(), (cut, name, s_)
                assert ia == ib, (cut, name, s_)
        ref.setdefault("digest", res["two"][0])
        assert res["two"][0] == ref["digest"], cut                                   # ... and the cut does not matter at all
    assert eod_seen >= 1                                                             # every stream did fire its 'eod' (and reset)
    gen.device_free(d_x)
    gen.close()


def test_seven_wave_is_the_default_for_small_uniform_batches():
    """The engine's own choice: a uniform configuration, every workgroup a compute unit to itself (up to 64 x compute units
    streams), a long enough call -> seven waves (round 6: per-stream tone pairs too); bigger batches and short calls stay on four."""
    import webaudio_modem_amd as wm
    N = 48000
    for S, cfg, n, want in ((2048, BELL, N, "demod_blk6_kernel"), (8192, BELL, N, "demod_blk6_kernel"), (16384, BELL, N, "demod_blk6_kernel"),
                            (32768, BELL, N, "demod_blk_kernel<"), (2048, BELL, 64, "demod_blk_kernel<"),
                            (2048, [dict(BELL, markFrequency=1200 + s % 7) for s in range(2048)], N, "demod_blk6_kernel<false, 8, false>")):
        eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
        d_x = eng.device_malloc(S * N * 4)
        eng.synth_device(d_x, N, N, 12, SEED + 5, 300, 0.1, 1.0)
        _demod_schedule(eng, d_x, N, N, [n])
        assert want in eng.last_kernel(), (S, n, eng.last_kernel())
        eng.device_free(d_x)
        eng.close()


def test_idle_receiver_bank_with_per_stream_tone_pairs():
    """Round 5: the idle regime for streams that each have their OWN tone pair (BASELINE config #4's kind): one
    frame per stream, then a floor 30 dB under it, every stream firing 'eod' and resetting on its own schedule.  The block path
    that takes resets now evaluates the direct instance's NCO phasors and lastPhase per lane (demod_blk_kernel_rp); it, the plain
    four-wave kernel (per-sample path), the two-wave kernel and several call schedules must agree for every stream -- bytes,
    'eod' counts, carried state words -- and a strided sample with the oracle."""
    import webaudio_modem_amd as wm
    from oracle import pyoracle as po
    S, N, payload, lead_max = 700, 120000, 40, 400
    cfgs = [dict(BELL, markFrequency=1200 + 7 * (s % 13), spaceFrequency=2200 + 5 * (s % 11)) for s in range(S)]
    gen = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32)
    frame_len = gen.modulated_length(payload)
    n0 = (lead_max + frame_len + 31) // 32 * 32
    d_x = gen.device_malloc(S * N * 4)
    gen.synth_device(d_x, n0, N, payload, SEED + 41, lead_max, 0.1, 1.0)
    gen.synchronize()
    row = np.empty(N, np.float32)
    rng = np.random.RandomState(11)
    x = np.zeros((S, N), np.float32)
    for s in range(S):
        gen.d2h(row, d_x + s * N * 4)
        lead, _amp = gen.synth_stream_params(SEED + 41, s, lead_max, 0.1, 1.0)
        end = lead + frame_len
        x[s, :end] = row[:end]
        p_frame = float(np.mean(row[lead:end].astype(np.float64) ** 2))
        x[s] += rng.normal(0.0, np.sqrt(p_frame / 1000.0), N).astype(np.float32)
    gen.h2d(d_x, x)
    results = {}
    for name, opts, schedule in (("plain", {"kernel": "four-wave", "blk_resets": 0}, [N]), ("resets", {"kernel": "four-wave", "blk_resets": 1}, [N]),
                                 ("resets_quanta", {"kernel": "four-wave", "blk_resets": 1}, [4800, 128, 17, 30000]),
                                 ("resets_redo", {"kernel": "four-wave", "blk_resets": 2}, [N]), ("redo_calls", {"kernel": "four-wave", "blk_resets": 2}, [16000]),
                                 ("two_wave", {"kernel": "two-wave"}, [N]), ("auto", {}, [24000]), ("auto_r04", {"kernel": "auto-r04"}, [24000]),
                                 ("seven", {"kernel": "seven-wave"}, [N]), ("seven_quanta", {"kernel": "seven-wave"}, [4800, 128, 17, 30000])):
        eng = wm.FSKEngine(S, cfgs, precision=wm.PRECISION_F32, options=opts)
        rows, eod = _demod_schedule(eng, d_x, N, N, schedule)
        if name == "resets":
            assert eng.last_kernel().startswith("fsk::demod_blk_kernel_rp<"), eng.last_kernel()
        if name == "plain":
            assert eng.last_kernel().startswith("fsk::demod_blk_kernel<false, false"), eng.last_kernel()
        if name == "auto_r04":
            assert eng.last_kernel().startswith("fsk::demod_blk_kernel_rp<"), eng.last_kernel()      # (the statistics of four calls of floor behind it)
        if name in ("auto", "seven"):
            assert eng.last_kernel().startswith("fsk::demod_blk6_kernel<false, 8, false>"), eng.last_kernel()   # (round 6: 700 streams = 88 groups of 8: seven waves, per-stream)
        results[name] = (rows, eod, [eng.debug_state(s_) for s_ in range(0, S, 23)])
        eng.close()
    base = _digest(results["plain"][0], results["plain"][1])
    for name, r in results.items():
        assert _digest(r[0], r[1]) == base, name
    for name in ("resets", "resets_redo", "two_wave", "seven"):      # (the same call lengths: the state words must be the same too -- a parked bit clock counts from the call's start)
        for (ra, ia), (rb, ib) in zip(results["plain"][2], results[name][2]):
            assert np.array_equal(np.asarray(ra).view(np.uint64), np.asarray(rb).view(np.uint64)), name
            assert ia == ib, name
    rows1, eod1 = results["plain"][0], results["plain"][1]
    hit = 0
    sample = list(range(0, S, S // 10)) + [S - 1]
    for s in sample:
        ob, oe = po.OracleCore(cfgs[s]).demodulate(x[s])
        assert rows1[s] == ob and int(eod1[s]) == oe, s
        hit += gen.synth_payload(SEED + 41, s, 0, payload) in rows1[s]
    assert hit >= len(sample) * 0.6
    assert int((eod1 > 50).sum()) >= S // 4 and int(eod1.max()) >= 300      # (a good part of the bank resets all the time; elsewhere the AGC lifts the floor over the threshold)
    gen.device_free(d_x)
    gen.close()
