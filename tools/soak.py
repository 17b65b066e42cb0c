#!/usr/bin/env python3
"""Randomised differential soak (GPU box): random configurations, batch sizes, chunk schedules (odd lengths, 1-sample
calls, whole-tile calls), per-stream resets and noise levels; every stream's bytes / eod counts / integer status are
compared with the CPU oracle fed the same buffers.  usage: python tools/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import webaudio_modem_amd as wm  # noqa: E402
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from oracle import pyoracle as po  # noqa: E402

CONFIGS = [
    {}, dict(baudRate=300), dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200),
    dict(baudRate=300, markFrequency=1070, spaceFrequency=1270), dict(parity="even"), dict(stopBits=2),
    dict(preamblePattern=[0x55, 0x55, 0x55], sfdPattern=[0x7E]), dict(sampleRate=44100), dict(agcEnabled=False),
    dict(baudRate=2400, markFrequency=2400, spaceFrequency=4800), dict(syncThreshold=0.75),
    # dsSPB = 8, 12, 16: the block kernel's smallest bit cells (one decision per eight-sample block at most)
    dict(baudRate=3000, markFrequency=3000, spaceFrequency=6000), dict(baudRate=2000, markFrequency=2000, spaceFrequency=4000),
    dict(baudRate=1500, markFrequency=1500, spaceFrequency=3000),
]
KEYS = ["frameStarted", "globalSampleCounter", "receivedBitsLength", "syncDetections"]
# Known, documented divergence of the fp32 path ("fp32 in exact-zero tails"): with a lowered syncThreshold the
# reference syncs on its own filter ringing as it decays through 1e-27 .. 1e-300 after a frame followed by exact zeros; fp32
# cannot represent that tail, so sync / eod COUNTS there can differ (decoded bytes are still compared).
def counts_comparable(cfg, prec):
    return prec == wm.PRECISION_F64 or cfg.get("syncThreshold", 0.85) >= 0.8


# fp32 engines approximate the reference's doubles: where an amplitude crosses the silence threshold within ~1e-6 of it,
# the silence run can start one decimated sample earlier or later (expected about once per 1e5 frame ends).  Bytes must
# still match; such timing differences are counted and reported, not failed.  fp64 engines must match exactly.
SOFT = {"n": 0, "first": None, "marginal": 0, "marginal_first": None}
HOSTILE = {"rounds": 0, "soft": 0}


def fp32_mismatch_is_marginal(cfg_s, xs, resets_seen):
    """An fp32 engine decoded other bytes than the reference.  Legitimate only if it traces back to a slicer decision
    the reference itself took within fp32 rounding of zero (|post filter output| < 1e-6: the discriminator's phase
    noise in fp32 is ~1e-7) -- e.g. one preamble tap flipped while the reference's sync count sat exactly at its
    threshold.  Replays the stream (one call, traced) on an fp64 and an fp32 engine and looks at the FIRST bit that
    differs.  Streams that were reset mid-round cannot be replayed this way: those stay hard failures."""
    if resets_seen:
        return False, "stream was reset during the round"
    tr = []
    THR_SEEN = {0.01}            # silence.threshold: its initial value (fsk.ts:128) and what the replays end with
    for prec in (wm.PRECISION_F64, wm.PRECISION_F32):
        e = wm.FSKEngine(1, cfg_s, precision=prec)
        e.trace_enable(0, len(xs))
        e.demodulate_data(xs.reshape(1, -1).copy())
        tr.append(e.trace_read())
        THR_SEEN.add(float(e.get_status(0)["silenceThreshold"]))
        e.close()
    n = min(len(tr[0]["bit"]), len(tr[1]["bit"]))
    d = np.nonzero(tr[0]["bit"][:n] != tr[1]["bit"][:n])[0]
    if not len(d):
        return False, "no slicer bit differs in a one-call replay"
    k = int(d[0])
    p64, p32 = float(tr[0]["post_out"][k]), float(tr[1]["post_out"][k])
    # (round 6, seed 67001: the fp32 side of the test is the accuracy the fp32 post filter is HELD to -- 2e-5 absolute,
    # tests/test_gpu_parity.py::test_intermediates_match_reference -- not 1e-6: at 300 baud its output crosses zero at 8e-4 per decimated
    # sample and carries ~1e-5 of rounding; the reference sat 2.4e-8 below zero on a preamble tap, fp32 9.7e-6 above)
    if abs(p64) < 1e-6 and abs(p32 - p64) <= 2e-5:
        return True, "first differing bit at decimated sample %d: post filter %.3e (fp64) vs %.3e (fp32), magnitude %.3e" % (
            k, p64, p32, float(tr[0]["amp"][k]))
    # round 4 (idle tails): the other decision fp32 can take differently is the silence compare (fsk.ts:285): one amplitude
    # within rounding of the threshold, the 'eod' reset then falls in one engine and not in the other, and from that sample on
    # the two simply are in different states (with a lowered syncThreshold the noise behind it can then sync differently).
    # Accepted only if, inside the last samplesForEOD + 1 decimated samples in front of the point where the amplitudes part,
    # the two engines' amplitudes sit on different sides of a threshold either of them held, within 1e-5 of it.
    a64, a32 = tr[0]["amp"][:n], tr[1]["amp"][:n]
    rel = np.abs(a64 - a32) / np.maximum(np.abs(a64), 1e-300)
    part = np.nonzero(rel > 1e-3)[0]
    if len(part) and int(part[0]) <= k:
        k0 = int(part[0])
        lo = max(0, k0 - 700)   # (>= samplesForEOD at every configuration of this soak)
        for t in THR_SEEN:
            for i in range(lo, k0):
                # (both within 1e-5 of the threshold AND on different sides of it -- or one exactly on it)
                if abs(a64[i] - t) <= 1e-5 * t and abs(a32[i] - t) <= 1e-5 * t and (a64[i] - t) * (a32[i] - t) <= 0.0:
                    if i + 1 <= k0:
                        return True, ("silence compare within 1e-5 of the threshold %.9g at decimated sample %d (%.9g fp64, %.9g fp32); the engines "
                                      "part at sample %d" % (t, i, a64[i], a32[i], k0))
    # round 6 (seed 67001, the first in ~6 M stream-runs): the third decision fp32 can take differently is the discriminator's phase
    # WRAP (fsk.ts:254-256): two consecutive decimated I/Q samples pointing in opposite directions (noise at a low magnitude), the
    # reference's phase difference within rounding of -pi or +pi -- fp32 lands on the other side, the post filter's input differs by
    # 2 pi and its output is somewhere else for the next few dozen samples.  Accepted only if the reference (the oracle, traced: it
    # records the post filter's INPUT) has |phase difference| within 1e-5 of pi at a sample in the 64 in front of the first
    # differing bit, and the two engines' post filter outputs agree to 1e-4 just before that sample and differ after it.
    o = po.OracleCore(cfg_s)
    o.enable_trace(len(xs), len(xs))
    o.demodulate(np.ascontiguousarray(xs, np.float32))
    pin = o.trace()["post_in"][:n]
    q64, q32 = tr[0]["post_out"][:n], tr[1]["post_out"][:n]
    for i in range(max(1, k - 64), k + 1):
        if abs(abs(float(pin[i])) - np.pi) <= 1e-5 and abs(q64[i - 1] - q32[i - 1]) <= 1e-4 and abs(q64[i] - q32[i]) > 1e-3:
            return True, ("phase wrap within 1e-5 of pi at decimated sample %d (the reference's phase difference there: %.9f, magnitude %.3e); post "
                          "filter %.4e (fp64) vs %.4e (fp32) one sample earlier, %.4e vs %.4e there; first differing bit at %d" % (
                              i, float(pin[i]), float(tr[0]["amp"][i]), q64[i - 1], q32[i - 1], q64[i], q32[i], k))
    return False, "first differing bit at decimated sample %d: post filter %.3e (fp64) vs %.3e (fp32), magnitude %.3e" % (
        k, p64, p32, float(tr[0]["amp"][k]))


def main(budget=None, seed=None, max_rounds=None):
    if budget is None:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0x50A4
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed ^ 0x5EED4)
    t_end = time.time() + budget
    rounds = streams = 0
    while time.time() < t_end and (max_rounds is None or rounds < max_rounds):
        if os.environ.get("SOAK_ONLY_ROUND") and rounds > int(os.environ["SOAK_ONLY_ROUND"]):
            break
        cfg = CONFIGS[int(rng.integers(len(CONFIGS)))]
        prec = wm.PRECISION_F32 if rng.random() < 0.7 else wm.PRECISION_F64
        target = os.environ.get("SOAK_ONLY_ROUND") is not None and int(os.environ["SOAK_ONLY_ROUND"]) == rounds
        draw_prec = prec
        if target and os.environ.get("SOAK_FORCE_PREC"):
            prec = int(os.environ["SOAK_FORCE_PREC"])
        S = int(rng.choice([1, 2, 63, 64, 65, 100, 130, 192, 257]))
        os.environ["FSKHIP_SPLIT"] = "014"[int(rng.integers(3))]   # one / two / four waves per 64-stream group
        # round 4: which four-wave kernel (block path with resets: never / always / always + redo / by the statistics) and its
        # group width -- drawn from a generator of their own, so that earlier rounds' seeds still replay the same signals
        os.environ["FSKHIP_BLK_RESETS"] = ["auto", "0", "1", "2", "1"][int(rng2.integers(5))]
        os.environ["FSKHIP_BLK_LANES"] = ["auto", "64", "16"][int(rng2.integers(3))]
        # round 5: half of the four-wave rounds go to the seven-wave small-batch kernel instead (uniform configurations; others
        # fall back to four waves), drawn from the second generator like the choices above
        if os.environ["FSKHIP_SPLIT"] == "4" and rng2.random() < 0.5:
            os.environ["FSKHIP_SPLIT"] = "6"
        if target and os.environ.get("SOAK_FORCE_SPLIT"):
            os.environ["FSKHIP_SPLIT"] = os.environ["SOAK_FORCE_SPLIT"]
        os.environ["FSKHIP_SPLIT_LAST"] = os.environ["FSKHIP_SPLIT"]
        # the four-wave block kernel's launch shapes (its own generator: the main random stream stays replayable): a
        # "device" of 1-3 resident workgroups so that batches of 2-5 groups run persistent and time-sliced, slices down to
        # one tile, y rings of every depth class
        knobs = {}
        krng = np.random.default_rng((seed * 1000003 + rounds) & 0x7FFFFFFF)
        if os.environ["FSKHIP_SPLIT"] == "4" and krng.random() < 0.6:
            knobs = {"FSKHIP_BLK_RESIDENT": str(int(krng.integers(1, 4))), "FSKHIP_SLICE_TILES": str(int(krng.choice([1, 2, 5, 16, 64]))),
                     "FSKHIP_BLK_YSLOTS": str(int(krng.choice([6, 7, 8, 12, 28])))}
            os.environ.update(knobs)
        # a quarter of the rounds: per-stream tone pairs (BASELINE config #4) -> the per-stream-constant kernel variants
        per_stream = rng.random() < 0.25 and "markFrequency" not in cfg and cfg.get("sampleRate", 48000) == 48000
        if per_stream:
            base = float(rng.choice([900, 1300, 1700]))
            cfgs = [dict(cfg, markFrequency=base + 13 * (s % 40), spaceFrequency=base + 200 + 13 * (s % 40)) for s in range(S)]
        else:
            cfgs = [cfg] * S
        writeback = rng.random() < 0.3
        eng = wm.FSKEngine(S, cfgs if per_stream else cfg, precision=prec)
        os.environ.pop("FSKHIP_SPLIT")
        for k_ in knobs:
            os.environ.pop(k_)
        if os.environ.get("SOAK_VERBOSE"):
            print("round", rounds, cfg, "prec", prec, "S", S, "split", eng and os.environ.get("FSKHIP_SPLIT_LAST"), knobs, flush=True)
        if not eng.demod_supported():
            eng.close()
            continue
        oracles = [po.OracleCore(cfgs[s]) for s in range(S)]
        frames = int(rng.integers(1, 4))
        payloads = [[bytes(rng.integers(0, 256, int(rng.integers(1, 24)), dtype=np.uint8)) for _ in range(frames)] for _ in range(S)]
        sigs = [np.concatenate([np.zeros(int(rng.integers(0, 300)), np.float32)] +
                               [oracles[s].modulate(p) * np.float32(rng.uniform(0.05, 1.0)) for p in payloads[s]]) for s in range(S)]
        N = max(len(x) for x in sigs) + int(rng.integers(0, 200))
        # round 4: a third of the rounds get an idle tail behind the frames (with the noise below: a receiver that fires
        # 'eod' and resets every samplesForEOD decimated samples, on its own schedule per stream -- the regime whose launch
        # boundaries lost a ZIR hand-over in the four-wave kernel).  Drawn from the knob generator: the main stream stays as it was.
        N += int(krng.choice([0, 0, 3000, 9000]))
        x = np.zeros((S, N), np.float32)
        for s in range(S):
            x[s, :len(sigs[s])] = sigs[s]
        if rng.random() < 0.4:
            snr = rng.uniform(8, 25)
            p = np.mean(x.astype(np.float64) ** 2, axis=1, keepdims=True)
            x = (x + rng.standard_normal(x.shape) * np.sqrt(p / 10 ** (snr / 10))).astype(np.float32)
        # round 6: one round in eight poisons one of its streams with a NaN or an Inf sample (either sign) somewhere -- the reference's
        # instance is dead from there on (fsk.ts:175-188, 264, 285) and every engine must die the same death, its neighbours untouched.
        # Its own generator (the main random stream stays replayable); writeback rounds excepted (their float compare is not NaN-aware).
        hrng = np.random.default_rng((seed * 7919 + rounds) & 0x7FFFFFFF)
        hostile_here = False
        if hrng.random() < 0.125 and not writeback:
            hostile_here = True
            x[int(hrng.integers(S)), int(hrng.integers(N))] = np.float32([np.nan, -np.nan, np.inf, -np.inf][int(hrng.integers(4))])
            HOSTILE["rounds"] += 1
        got = [b""] * S
        want = [b""] * S
        off = 0
        log = []
        dead = set()   # fp32 streams whose divergence was traced to a marginal slicer decision
        only = os.environ.get("SOAK_ONLY_ROUND")
        dry = only is not None and int(only) != rounds  # replay the random stream, skip the work
        if only is not None and not dry and os.environ.get("SOAK_DUMP"):
            np.save(os.environ["SOAK_DUMP"], x)
            print("dumped", x.shape, cfg, prec)
            return
        while off < N:
            n = int(rng.choice([1, 3, 16, 17, 128, 129, 1000, 4096, 10 ** 9], p=[.05, .05, .1, .05, .25, .1, .2, .1, .1]))
            n = min(n, N - off)
            if dry:
                off += n
                if rng.random() < 0.03:
                    rng.integers(S)
                continue  # (replay mode: the reset-all / empty-call branches draw nothing)
            if os.environ.get("SOAK_VERBOSE"):
                print("  call", off, n, flush=True)
            chunk = x[:, off:off + n].copy()  # (ascontiguousarray would alias x when the chunk is the whole buffer)
            out, eod = eng.demodulate_data(chunk, writeback_agc=writeback)
            log.append(("call", off, n))
            for s in range(S):
                ob, oe = oracles[s].demodulate(x[s, off:off + n])
                if writeback and cfg.get("agcEnabled", True):
                    # fsk.ts:55: the input buffer holds the AGC-scaled samples afterwards (fp64 engines: the same floats)
                    ref = oracles[s].last_agc_out
                    if prec == wm.PRECISION_F64:
                        assert np.array_equal(chunk[s], ref), ("agc writeback", cfg, S, s, off, n)
                    else:
                        d = np.abs(chunk[s] - ref)
                        i = int(np.argmax(d))
                        # fp32: where |x*g| lands within an ulp of the 0.5 attack/release boundary (fsk.ts:60) the two
                        # paths can take different branches; the gains then differ by up to (1-g)*(attack-release),
                        # a few per cent for a few hundred samples, until the AGC has pulled them together again
                        assert d[i] <= 0.05 * max(1.0, float(np.max(np.abs(ref)))), (
                            "agc writeback", cfg, S, s, off, n, i, float(d[i]), float(chunk[s, i]), float(ref[i]), float(x[s, off + i]), per_stream)
                if s not in dead:
                    got[s] += out[s]
                    want[s] += ob
                if int(eod[s]) != oe and counts_comparable(cfg, prec) and prec == wm.PRECISION_F32 and out[s] == ob:
                    SOFT["n"] += 1
                    HOSTILE["soft"] += 1 if hostile_here else 0
                    SOFT["first"] = SOFT["first"] or ("eod", cfg, S, s, off, n, int(eod[s]), oe, "hostile round" if hostile_here else "")
                elif (int(eod[s]) != oe and counts_comparable(cfg, prec)) or out[s] != ob:
                    if s in dead:
                        continue
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.save("gpurun_out/soak_fail_%x.npy" % seed, x[s])
                    what = ("mismatch", "round %d" % rounds, cfgs[s], prec, S, s, off, n, int(eod[s]), oe, out[s], ob, log[-12:])
                    if prec == wm.PRECISION_F32:
                        ok, why = fp32_mismatch_is_marginal(cfgs[s], x[s], any(l[0].startswith("reset") and (l[0] == "reset all" or l[1] == s) for l in log))
                        if ok:
                            SOFT["marginal"] += 1
                            SOFT["marginal_first"] = SOFT["marginal_first"] or (why, cfgs[s], S, s)
                            print("fp32 marginal decision (counted, not failed):", why, cfgs[s], "stream", s, flush=True)
                            dead.add(s)   # its bytes legitimately differ from here on
                            continue
                        what = what + (why,)
                    raise AssertionError(what)
            off += n
            u = rng.random()
            if u < 0.03:
                r = int(rng.integers(S))
                eng.reset(r)
                oracles[r].reset()
                log.append(("reset", r, off))
            elif u < 0.04:
                eng.reset(-1)
                for o in oracles:
                    o.reset()
                log.append(("reset all", off))
            elif u < 0.05:
                out0, eod0 = eng.demodulate_data(np.zeros((S, 0), np.float32))  # empty call (fsk-demodulation.node.test.ts:38-41)
                for s in range(S):
                    ob, oe = oracles[s].demodulate(np.zeros(0, np.float32))
                    assert out0[s] == ob == b"" and int(eod0[s]) == oe == 0
                log.append(("empty", off))
        for s in range(S):
            assert s in dead or got[s] == want[s], ("bytes", cfg, prec, S, s, got[s][:8], want[s][:8])
        sel = rng.choice(S, min(S, 8), replace=False) if counts_comparable(cfg, draw_prec) else []
        for s in ([] if dry else [v for v in sel if int(v) not in dead]):
            st, ost = eng.get_status(int(s)), oracles[int(s)].status()
            for k in KEYS:
                if st[k] != ost[k] and prec == wm.PRECISION_F32:
                    SOFT["n"] += 1
                    SOFT["first"] = SOFT["first"] or ("status", k, cfg, S, int(s), st[k], ost[k])
                    break
                assert st[k] == ost[k], ("status", k, cfg, prec, S, int(s), st[k], ost[k])
        eng.close()
        rounds += 1
        streams += S
    # an excused stream stops being checked for the rest of its round, so excuses must stay the rare exception they were
    # measured to be (round 2: two in 1.1 M stream-runs): more than one per 100 000 stream-runs fails the soak
    assert SOFT["marginal"] <= 1 + streams // 100000, ("too many fp32 divergences excused as marginal", SOFT["marginal"], SOFT["marginal_first"])
    print("soak ok: %d rounds, %d stream-runs, seed %#x; fp32 timing differences with identical bytes: %d %s; fp32 streams "
          "diverging after a marginal decision (slicer input within 1e-6 of zero in the reference / silence compare / phase wrap): %d %s; rounds with a NaN / Inf sample in one stream: %d (timing differences in those: %d)"
          % (rounds, streams, seed, SOFT["n"], SOFT["first"] or "", SOFT["marginal"], SOFT["marginal_first"] or "", HOSTILE["rounds"], HOSTILE["soft"]))
    return rounds, streams, SOFT["n"]


if __name__ == "__main__":
    main()
