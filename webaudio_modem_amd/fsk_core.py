"""FSKCore: single-stream mirror of the reference's IModulator surface on top of the HIP engine.

Same names, argument meaning and error behaviour as the reference class
(src/modems/fsk.ts:82-494, src/core.ts:88-117, 210-289): configure / getConfig / modulateData /
demodulateData / reset / isReady / getStatus / getSignalQuality / on / off / emit, with the
'configured', 'eod' and 'error' events.  Every compute call goes through libfskhip.so; nothing
here computes DSP on the CPU.
"""
import numpy as np

from .engine import FSKEngine, DEFAULT_FSK_CONFIG, PRECISION_F64
from ._lib import FskHipError


class Event:
    """core.ts:205-207"""

    def __init__(self, data=None):
        self.data = data


class EventEmitter:
    """core.ts:210-244"""

    def __init__(self):
        self._listeners = {}

    def on(self, event_name, callback):
        self._listeners.setdefault(event_name, []).append(callback)

    def off(self, event_name, callback):
        lst = self._listeners.get(event_name)
        if lst and callback in lst:
            lst.remove(callback)

    def emit(self, event_name, event=None):
        event = event if event is not None else Event()
        for cb in list(self._listeners.get(event_name, [])):
            cb(event)

    def removeAllListeners(self, event_name=None):
        if event_name:
            self._listeners.pop(event_name, None)
        else:
            self._listeners.clear()


class FSKCore(EventEmitter):
    """One FSK modem instance; the DSP runs on the GPU (one-stream engine).

    precision defaults to the fp64 parity path: a single stream cannot fill the machine anyway,
    and fp64 is op-for-op with the reference's arithmetic.  The batch engine (FSKEngine) is the
    throughput interface.
    """
    name = "FSK"
    type = "FSK"

    def __init__(self, device=0, precision=PRECISION_F64):
        super().__init__()
        self._device = device
        self._precision = precision
        self._engine = None
        self.config = None
        self.ready = False

    # configure() fsk.ts:133-157
    def configure(self, config=None):
        old = self._engine
        self._engine = FSKEngine(1, dict(config or {}), device=self._device, precision=self._precision)
        if old is not None:
            # the reference rebuilds in place and keeps silence.threshold and the debug counters (fsk.ts:133-157)
            self._engine.carry_over_from(old)
            old.close()
        self.config = dict(self._engine.config)
        self.ready = True
        self.emit("configured")

    def getConfig(self):
        return dict(self.config) if self.config else {}

    def isReady(self):
        return self.ready

    # demodulateData() fsk.ts:190-222: mutates `samples` in place when AGC is on (fsk.ts:55)
    def demodulateData(self, samples):
        if not self.ready or self._engine is None:
            raise RuntimeError("FSK demodulator not configured")
        try:
            x = samples
            if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags.c_contiguous):
                x = np.ascontiguousarray(samples, dtype=np.float32)
            view = x.reshape(1, -1)
            out, eod = self._engine.demodulate_data(view, writeback_agc=bool(self.config.get("agcEnabled", True)))
            for _ in range(int(eod[0])):
                self.emit("eod")
            return np.frombuffer(out[0], dtype=np.uint8).copy()
        except FskHipError as err:
            # fsk.ts:218-221: any exception inside demodulation -> 'error' event + empty result
            self.emit("error", Event({"data": err}))
            return np.zeros(0, dtype=np.uint8)

    # modulateData() fsk.ts:377-383
    def modulateData(self, data):
        if not self.ready or self._engine is None:
            raise RuntimeError("FSK modulator not configured")
        return self._engine.modulate_data([bytes(bytearray(data))])[0]

    # reset() fsk.ts:464-469 (ready stays true: FSKCore overrides BaseModulator.reset)
    def reset(self):
        if self._engine is not None:
            self._engine.reset(0)

    # getSignalQuality() fsk.ts:471-479: all-zero stub in the reference
    def getSignalQuality(self):       # fsk.ts:471-479: the reference's stub, kept
        return {"snr": 0, "ber": 0, "eyeOpening": 0, "phaseJitter": 0, "frequencyOffset": 0}

    # opt-in extension (include/fskhip.h): real estimates of the same five fields, from the demodulator's rare paths
    def enableSignalQualityEstimates(self, on=True):
        if self._engine is None:
            raise RuntimeError("FSK demodulator not configured")
        self._engine.enable_signal_quality(on)

    def getSignalQualityEstimates(self):
        if self._engine is None:
            return dict(self.getSignalQuality(), signalLevel=0, noiseFloor=0, frames=0, bytes=0)
        return self._engine.get_signal_quality(0)

    # getStatus() fsk.ts:481-493
    def getStatus(self):
        if self._engine is None:
            return {"ready": False, "frameStarted": False, "globalSampleCounter": 0, "receivedBitsLength": 0,
                    "byteBufferLength": 0, "demodulationCalls": 0, "syncDetections": 0, "silenceThreshold": 0.01,
                    "totalSamplesProcessed": 0}
        st = self._engine.get_status(0)
        st["ready"] = self.ready
        return st

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        self.ready = False


__all__ = ["FSKCore", "Event", "EventEmitter", "DEFAULT_FSK_CONFIG"]
