"""One host process driving several GPUs: FSKEngineSharded = one FSKEngine per device, each owning a contiguous block
of streams (sharding.stream_shard), calls fanned out on worker threads (libfskhip.so releases nothing Python-side: ctypes
drops the GIL for the duration of a call, every entry point selects its engine's device first, and the library's error
text is thread-local).  The data path still has no collective: streams are independent.

This is the in-process alternative to bench.py's one-process-per-GPU layout, for hosts that own all the node's streams in
one address space (the reference's FSKCore instances live in one JavaScript realm).  napi/fsk-core.js has the same class
(FSKBatchSharded) over the asynchronous N-API calls."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from ._lib import PRECISION_F32
from .engine import FSKEngine
from .sharding import all_shards


class FSKEngineSharded:
    """S independent FSKCore instances spread over `devices` (default: every HIP device of the node).

    Same methods as FSKEngine where a host-side call makes sense for a sharded batch; stream indices are global.
    engine_factory(count, configs, device, precision) -> engine is there for tests (no GPU needed to check the
    shard / gather logic)."""

    def __init__(self, n_streams, configs=None, devices=None, precision=PRECISION_F32, engine_factory=None):
        if devices is None:
            if engine_factory is not None:
                # a stand-in engine (CPU tests of the sharding / gather logic): the HIP library is not touched at all
                raise ValueError("pass `devices` together with an engine_factory")
            n_dev = int(_lib.lib().fskhip_device_count())
            if n_dev <= 0:
                # same loud failure as FSKEngine on a box without a GPU: there is no CPU path
                FSKEngine(1, configs if not isinstance(configs, (list, tuple)) else configs[0])
            devices = list(range(max(n_dev, 1)))
        devices = list(devices)
        if not devices:
            raise ValueError("no devices")
        if isinstance(configs, (list, tuple)) and len(configs) != n_streams:
            raise ValueError("need one config per stream")
        make = engine_factory or (lambda count, cfg, dev, prec: FSKEngine(count, cfg, device=dev, precision=prec))
        self.n_streams = n_streams
        self.precision = precision
        self.devices = devices
        # devices with nothing to do (more devices than streams) get no engine
        self.shards = [(first, count, dev) for (first, count), dev in zip(all_shards(n_streams, len(devices)), devices) if count]
        self.engines = []
        try:
            for first, count, dev in self.shards:
                cfg = list(configs[first:first + count]) if isinstance(configs, (list, tuple)) else configs
                self.engines.append(make(count, cfg, dev, precision))
        except Exception:
            self.close()
            raise
        self._pool = ThreadPoolExecutor(max_workers=max(1, len(self.engines)), thread_name_prefix="fskhip-dev")

    # ---- plumbing --------------------------------------------------------------------------------
    def close(self):
        for e in getattr(self, "engines", []):
            e.close()
        self.engines = []
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def locate(self, stream):
        """(shard index, local stream index) of a global stream index."""
        if not (0 <= stream < self.n_streams):
            raise ValueError("stream out of range")
        for i, (first, count, _dev) in enumerate(self.shards):
            if first <= stream < first + count:
                return i, stream - first
        raise AssertionError("unreachable")

    def _fan_out(self, fn):
        """fn(shard index, engine, first, count) on every shard concurrently; results in shard order.  The first
        exception is re-raised after every worker has finished (no call is left running on a device)."""
        futs = [self._pool.submit(fn, i, e, self.shards[i][0], self.shards[i][1]) for i, e in enumerate(self.engines)]
        results, err = [], None
        for f in futs:
            try:
                results.append(f.result())
            except Exception as ex:  # noqa: BLE001 -- collected, re-raised below
                err = err or ex
                results.append(None)
        if err is not None:
            raise err
        return results

    # ---- demodulateData / modulateData (fsk.ts:190-222, 377-424) ---------------------------------
    def demodulate_data(self, samples, writeback_agc=False):
        """samples: float32 [S, N] (host).  Returns (list of bytes per stream, eod counts ndarray), stream order."""
        x = samples if (isinstance(samples, np.ndarray) and samples.dtype == np.float32 and samples.flags.c_contiguous) \
            else np.ascontiguousarray(samples, dtype=np.float32)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        if x.shape[0] != self.n_streams:
            raise ValueError("expected %d streams, got %d" % (self.n_streams, x.shape[0]))
        parts = self._fan_out(lambda i, e, first, count: e.demodulate_data(x[first:first + count], writeback_agc=writeback_agc))
        out, eod = [], []
        for o, c in parts:
            out.extend(o)
            eod.append(np.asarray(c, dtype=np.uint32))
        if writeback_agc and x is not samples:
            np.copyto(samples, x.reshape(np.shape(samples)))
        return out, (np.concatenate(eod) if eod else np.zeros(0, np.uint32))

    def modulate_data(self, payloads):
        if len(payloads) != self.n_streams:
            raise ValueError("need one payload per stream")
        parts = self._fan_out(lambda i, e, first, count: e.modulate_data(payloads[first:first + count]))
        return [sig for p in parts for sig in p]

    # ---- reset / getStatus -----------------------------------------------------------------------
    def reset(self, stream=-1):
        if stream < 0:
            self._fan_out(lambda i, e, first, count: e.reset(-1))
        else:
            i, local = self.locate(stream)
            self.engines[i].reset(local)

    def get_status(self, stream=0):
        i, local = self.locate(stream)
        return self.engines[i].get_status(local)

    def demod_supported(self):
        return all(e.demod_supported() for e in self.engines)
