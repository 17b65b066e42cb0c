"""The streaming contract either side of FSKCore, above the C ABI (include/fskhip_next.h):

* `ChunkedModulator` -- src/webaudio/chunked-modulator.ts, same method names, over any object with the
  `modulateData(bytes)` of `FSKCore` (the GPU modulator);
* `FSKProcessorBatch` -- S instances of src/webaudio/processors/fsk-processor.ts on one GPU: `process()` once per
  quantum for all streams (RX byte rings and pending modulations stay on the device), `modulate`, `demodulate`,
  `reset`, `status`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PROC_CLEAR_RX_ON_TX_COMPLETE, PROC_GRAPH  # noqa: F401


class ChunkedModulator:
    """chunked-modulator.ts:22-88."""

    def __init__(self, modulator):
        self.modulator = modulator
        self.pendingSignal = None
        self.samplePosition = 0

    def startModulation(self, data):
        if len(data) == 0:
            self._reset()
            return
        self.pendingSignal = np.asarray(self.modulator.modulateData(bytes(data)), dtype=np.float32)
        self.samplePosition = 0

    def getNextSamples(self, sampleCount):
        if self.pendingSignal is None:
            return None
        remaining = len(self.pendingSignal) - self.samplePosition
        if remaining <= 0:
            return None
        n = min(sampleCount, remaining)
        signal = self.pendingSignal[self.samplePosition:self.samplePosition + n].copy()
        self.samplePosition += n
        total = len(self.pendingSignal)
        if self.samplePosition >= total:
            self._reset()
            return {"signal": signal, "isComplete": True, "samplesConsumed": total, "totalSamples": total}
        return {"signal": signal, "isComplete": False, "samplesConsumed": self.samplePosition, "totalSamples": total}

    def isModulating(self):
        return self.pendingSignal is not None

    def getProgress(self):
        return self.samplePosition / len(self.pendingSignal) if self.pendingSignal is not None else 0

    def cancel(self):
        self._reset()

    def _reset(self):
        self.pendingSignal = None
        self.samplePosition = 0


class FSKProcessorBatch:
    """S FSKProcessors over one FSKEngine.  rx_capacity 1024 is the reference's demodulatedBuffer size
    (fsk-processor.ts:84); clear_rx_on_tx_complete mirrors its 'modulate' handler (228-235)."""

    def __init__(self, engine, rx_capacity=1024, clear_rx_on_tx_complete=True, use_graph=False):
        self.engine = engine
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.fskhip_processor_create(engine._h, rx_capacity, C.byref(h)))
        self._h = h
        self.n_streams = engine.n_streams
        self.rx_capacity = rx_capacity
        self.flags = (PROC_CLEAR_RX_ON_TX_COMPLETE if clear_rx_on_tx_complete else 0) | (PROC_GRAPH if use_graph else 0)
        self.processDemodulationCallCount = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.fskhip_processor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- process(inputs, outputs) fsk-processor.ts:152-167 -----------------------------------------
    def process(self, inputs=None, n_out=0):
        """inputs: float32 [S, n_in] or None; returns float32 [S, n_out] (or None when n_out == 0)."""
        x = None
        n_in = 0
        if inputs is not None:
            x = np.ascontiguousarray(inputs, dtype=np.float32)
            if x.ndim != 2 or x.shape[0] != self.n_streams:
                raise ValueError("inputs must be [n_streams, n]")
            n_in = x.shape[1]
            self.processDemodulationCallCount += 1
        out = np.zeros((self.n_streams, n_out), dtype=np.float32) if n_out else None
        # the host form stages through the processor's own stream, which a graph capture needs anyway
        _lib.check(self._L.fskhip_processor_process_host(
            self._h, x.ctypes.data if x is not None else None, n_in, n_in,
            out.ctypes.data if out is not None else None, n_out, n_out, self.flags))
        return out

    def process_device(self, d_in, n_in, in_pitch, d_out, n_out, out_pitch, stream=None, flags=None):
        _lib.check(self._L.fskhip_processor_process_device(self._h, d_in, n_in, in_pitch, d_out, n_out, out_pitch,
                                                           self.flags if flags is None else flags, stream))

    # ---- 'modulate' fsk-processor.ts:87-113 ------------------------------------------------------------
    def modulate(self, payloads, mask=None):
        """payloads: list of S bytes-like; mask: optional list of S bools (streams to start)."""
        rows = [bytes(p) for p in payloads]
        if len(rows) != self.n_streams:
            raise ValueError("need one payload per stream")
        lens = np.array([len(r) for r in rows], dtype=np.uint32)
        pitch = max(1, int(lens.max()))
        slab = np.zeros((self.n_streams, pitch), dtype=np.uint8)
        for i, r in enumerate(rows):
            slab[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=bool).astype(np.uint8))
        rc = self._L.fskhip_processor_modulate_host(self._h, slab.ctypes.data, lens.ctypes.data, pitch,
                                                    m.ctypes.data if m is not None else None)
        if rc == _lib.E_BUSY:
            raise RuntimeError("Modulation already in progress")  # fsk-processor.ts:91
        _lib.check(rc)

    def tx_state(self):
        S = self.n_streams
        pos, total, done = (np.zeros(S, np.uint32) for _ in range(3))
        pend = np.zeros(S, np.uint8)
        _lib.check(self._L.fskhip_processor_tx_state_host(self._h, pos.ctypes.data, total.ctypes.data, pend.ctypes.data,
                                                          done.ctypes.data))
        return {"samplePosition": pos, "totalSamples": total, "pendingModulation": pend.astype(bool), "completed": done,
                "isModulating": total > 0,
                "progress": np.where(total > 0, pos / np.maximum(total, 1), 0.0)}

    # ---- 'demodulate' fsk-processor.ts:117-138 (without the wait) -----------------------------------------
    def demodulate(self):
        out = np.zeros((self.n_streams, self.rx_capacity), dtype=np.uint8)
        counts = np.zeros(self.n_streams, dtype=np.uint32)
        _lib.check(self._L.fskhip_processor_rx_drain_host(self._h, out.ctypes.data, self.rx_capacity, counts.ctypes.data))
        return [out[s, :counts[s]].tobytes() for s in range(self.n_streams)]

    def rx_lengths(self):
        lens = np.zeros(self.n_streams, dtype=np.uint32)
        _lib.check(self._L.fskhip_processor_rx_length_host(self._h, lens.ctypes.data))
        return lens

    def reset(self, stream=-1):
        _lib.check(self._L.fskhip_processor_reset(self._h, stream))

    def status(self, stream=0):
        """The 'status' reply (fsk-processor.ts:240-253) for one stream."""
        st = self.engine.get_status(stream)
        tx = self.tx_state()
        st.update(demodulatedBufferLength=int(self.rx_lengths()[stream]), pendingModulation=bool(tx["pendingModulation"][stream]),
                  fskCoreReady=True, processDemodulationCallCount=self.processDemodulationCallCount)
        return st
