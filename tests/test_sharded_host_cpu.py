"""One process, several devices (webaudio_modem_amd/sharded.py): the shard / fan-out / gather logic, checked on the CPU
with a stand-in engine (the real one needs a GPU; tests/test_gpu_fullsize.py runs the real engines on device 0)."""
import threading
import time

import numpy as np
import pytest

import webaudio_modem_amd as wm
from webaudio_modem_amd.sharding import all_shards


class FakeEngine:
    """Decodes nothing: returns what identifies (device, local stream, data) so the test can see the routing."""
    live = 0
    peak = 0
    lock = threading.Lock()

    def __init__(self, count, cfg, dev, prec):
        self.count, self.cfg, self.dev, self.prec = count, cfg, dev, prec
        self.resets = []
        self.closed = False

    def demodulate_data(self, x, writeback_agc=False):
        with FakeEngine.lock:
            FakeEngine.live += 1
            FakeEngine.peak = max(FakeEngine.peak, FakeEngine.live)
        time.sleep(0.05)                      # long enough for the shards to overlap if they run concurrently
        assert x.shape[0] == self.count
        out = [bytes([self.dev, s % 256, int(x[s, 0]) % 256]) for s in range(self.count)]
        if writeback_agc:
            x *= 2.0
        with FakeEngine.lock:
            FakeEngine.live -= 1
        return out, np.full(self.count, self.dev, np.uint32)

    def modulate_data(self, payloads):
        assert len(payloads) == self.count
        return [np.full(len(p), self.dev, np.float32) for p in payloads]

    def reset(self, stream=-1):
        self.resets.append(stream)

    def get_status(self, stream=0):
        return {"device": self.dev, "local": stream}

    def demod_supported(self):
        return True

    def close(self):
        self.closed = True


def make(count, cfg, dev, prec):
    return FakeEngine(count, cfg, dev, prec)


def test_shards_are_contiguous_and_calls_run_concurrently():
    S, devs = 103, [0, 1, 2, 3]
    eng = wm.FSKEngineSharded(S, {}, devices=devs, engine_factory=make)
    assert [(f, c) for f, c, _ in eng.shards] == all_shards(S, 4) and [d for _, _, d in eng.shards] == devs
    x = np.zeros((S, 8), np.float32)
    x[:, 0] = np.arange(S)
    FakeEngine.peak = 0
    out, eod = eng.demodulate_data(x)
    assert FakeEngine.peak == 4               # one call in flight per device
    assert len(out) == S and eod.shape == (S,)
    for s in range(S):
        i, local = eng.locate(s)
        assert out[s] == bytes([devs[i], local % 256, s % 256]) and eod[s] == devs[i]
    held = list(eng.engines)
    eng.close()
    assert all(e.closed for e in held) and eng.engines == []


def test_per_stream_configs_are_sliced_per_shard_and_resets_routed():
    S = 10
    cfgs = [dict(markFrequency=1000 + s, spaceFrequency=1200 + s) for s in range(S)]
    eng = wm.FSKEngineSharded(S, cfgs, devices=[5, 7, 9], engine_factory=make)
    assert [e.count for e in eng.engines] == [4, 3, 3]
    assert eng.engines[1].cfg == cfgs[4:7] and eng.engines[2].dev == 9
    eng.reset(5)
    eng.reset(-1)
    assert eng.engines[1].resets == [1, -1] and eng.engines[0].resets == [-1]
    assert eng.get_status(9) == {"device": 9, "local": 2}
    sig = eng.modulate_data([b"x" * (s + 1) for s in range(S)])
    assert [len(v) for v in sig] == list(range(1, S + 1)) and sig[4][0] == 7.0
    with pytest.raises(ValueError):
        eng.locate(10)
    with pytest.raises(ValueError):
        eng.demodulate_data(np.zeros((9, 4), np.float32))
    eng.close()


def test_more_devices_than_streams_and_writeback():
    eng = wm.FSKEngineSharded(2, {}, devices=[0, 1, 2, 3], engine_factory=make)
    assert len(eng.engines) == 2              # the idle devices get no engine
    x = np.ones((2, 4), np.float32)
    eng.demodulate_data(x, writeback_agc=True)
    assert np.all(x == 2.0)                   # the engines worked on views of the caller's buffer (fsk.ts:55 side effect)
    eng.close()


def test_an_error_on_one_device_is_raised_after_all_calls_finished():
    class Boom(FakeEngine):
        def demodulate_data(self, x, writeback_agc=False):
            if self.dev == 1:
                raise wm.FskHipError(-5, "device 1 failed")
            return super().demodulate_data(x, writeback_agc)

    eng = wm.FSKEngineSharded(8, {}, devices=[0, 1, 2], engine_factory=lambda c, cfg, d, p: Boom(c, cfg, d, p))
    with pytest.raises(wm.FskHipError, match="device 1 failed"):
        eng.demodulate_data(np.zeros((8, 4), np.float32))
    assert FakeEngine.live == 0               # nothing left running
    eng.close()


def test_without_a_gpu_the_sharded_engine_fails_as_loudly_as_the_plain_one():
    from webaudio_modem_amd import _lib
    if _lib.lib().fskhip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(wm.FskHipError) as ei:
        wm.FSKEngineSharded(4, {})
    assert ei.value.code == -4 and "no CPU fallback" in str(ei.value)
