"""FilterDesign: the configure-time Butterworth designs the engine uses (reference
src/dsp/filters.ts:180-234), served by the C ABI so that Python, Node and the kernels' host
code share one implementation."""
import ctypes as C

from . import _lib


def _call(fn, *args):
    b = (C.c_double * 3)()
    a = (C.c_double * 3)()
    fn(*args, b, a)
    return {"b": list(b), "a": list(a)}


class FilterDesign:
    @staticmethod
    def butterworthLowpass(cutoffFreq, sampleRate):
        return _call(_lib.lib().fskhip_butterworth_lowpass, float(cutoffFreq), float(sampleRate))

    @staticmethod
    def butterworthHighpass(cutoffFreq, sampleRate):
        return _call(_lib.lib().fskhip_butterworth_highpass, float(cutoffFreq), float(sampleRate))

    @staticmethod
    def butterworthBandpass(centerFreq, bandwidth, sampleRate):
        return _call(_lib.lib().fskhip_butterworth_bandpass, float(centerFreq), float(bandwidth), float(sampleRate))

    # windowed-sinc FIR designs (filters.ts:243-314)
    @staticmethod
    def _sinc(fn, numTaps, *args):
        taps = (C.c_double * (int(numTaps) + 1))()
        n = fn(*[float(a) for a in args], int(numTaps), taps)
        if n < 0:
            _lib.check(n)
        return list(taps)[:n]

    @staticmethod
    def sincLowpass(cutoffFreq, sampleRate, numTaps):
        return FilterDesign._sinc(_lib.lib().fskhip_sinc_lowpass, numTaps, cutoffFreq, sampleRate)

    @staticmethod
    def sincHighpass(cutoffFreq, sampleRate, numTaps):
        return FilterDesign._sinc(_lib.lib().fskhip_sinc_highpass, numTaps, cutoffFreq, sampleRate)

    @staticmethod
    def sincBandpass(centerFreq, bandwidth, sampleRate, numTaps):
        return FilterDesign._sinc(_lib.lib().fskhip_sinc_bandpass, numTaps, centerFreq, bandwidth, sampleRate)


class FIRFilterBatch:
    """`new FIRFilter(coefficients)` (filters.ts:112-167) for n_streams streams on one GPU: `processBuffer` on a
    float32 [S, N] block, delay lines carried across calls, `reset`, `getCoefficients`."""

    def __init__(self, coefficients, n_streams=1, device=0, precision=_lib.PRECISION_F64):
        import numpy as np
        self._np = np
        self.coefficients = [float(c) for c in coefficients]
        if not self.coefficients:
            raise ValueError("FIR needs at least one coefficient")
        taps = (C.c_double * len(self.coefficients))(*self.coefficients)
        h = C.c_void_p()
        self._L = _lib.lib()
        _lib.check(self._L.fskhip_fir_create(device, taps, len(self.coefficients), n_streams, precision, C.byref(h)))
        self._h = h
        self.n_streams = n_streams

    def close(self):
        if getattr(self, "_h", None):
            self._L.fskhip_fir_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def processBuffer(self, x):
        np = self._np
        x = np.ascontiguousarray(x, dtype=np.float32)
        single = x.ndim == 1
        if single:
            x = x[None, :]
        if x.shape[0] != self.n_streams:
            raise ValueError("input must be [n_streams, n]")
        out = np.empty_like(x)
        if x.shape[1]:
            _lib.check(self._L.fskhip_fir_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data,
                                                       x.shape[1]))
        return out[0] if single else out

    def process_device(self, d_in, n, in_pitch, d_out, out_pitch, stream=None):
        _lib.check(self._L.fskhip_fir_process_device(self._h, d_in, n, in_pitch, d_out, out_pitch, stream))

    def reset(self, stream=-1):
        _lib.check(self._L.fskhip_fir_reset(self._h, stream))

    def getCoefficients(self):
        return list(self.coefficients)


class FIRFilter(FIRFilterBatch):
    """One FIRFilter with the reference's per-sample surface: process(x) is a batch of one sample."""

    def __init__(self, coefficients, device=0, precision=_lib.PRECISION_F64):
        super().__init__(coefficients, 1, device, precision)

    def process(self, x):
        return float(self.processBuffer(self._np.array([x], dtype=self._np.float32))[0])


class IIRFilterBatch:
    """`new IIRFilter(b, a)` (filters.ts:8-106) for n_streams streams on one GPU: `processBuffer` on a float32 [S, N]
    block (Float32Array in, Float32Array out), `processSamples` on a float64 block (what `process(x)` returns sample by
    sample, nothing rounded to float), histories carried across calls, `reset`, `getCoefficients` (normalised by a[0]).
    The constructor's three errors are the reference's (filters.ts:19-21), raised as ValueError with its messages."""

    def __init__(self, b, a, n_streams=1, device=0, precision=_lib.PRECISION_F64):
        import numpy as np
        self._np = np
        b = [float(v) for v in (b if b is not None else [])]
        a = [float(v) for v in (a if a is not None else [])]
        # filters.ts:19-21 (checked here too so that the errors do not need a GPU; the library checks again)
        if len(b) == 0:
            raise ValueError("Feedforward coefficients (b) cannot be empty")
        if len(a) == 0:
            raise ValueError("Feedback coefficients (a) cannot be empty")
        if a[0] == 0:
            raise ValueError("First feedback coefficient (a[0]) cannot be zero")
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.fskhip_iir_create(device, (C.c_double * len(b))(*b), len(b), (C.c_double * len(a))(*a), len(a),
                                             n_streams, precision, C.byref(h)))
        self._h = h
        self.n_streams = n_streams

    def close(self):
        if getattr(self, "_h", None):
            self._L.fskhip_iir_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _run(self, x, dtype, fn):
        np = self._np
        x = np.ascontiguousarray(x, dtype=dtype)
        single = x.ndim == 1
        if single:
            x = x[None, :]
        if x.shape[0] != self.n_streams:
            raise ValueError("input must be [n_streams, n]")
        out = np.empty_like(x)
        if x.shape[1]:
            _lib.check(fn(self._h, x.ctypes.data, x.shape[1], x.shape[1], out.ctypes.data, x.shape[1]))
        return out[0] if single else out

    def processBuffer(self, x):
        return self._run(x, self._np.float32, self._L.fskhip_iir_process_host)

    def processSamples(self, x):
        return self._run(x, self._np.float64, self._L.fskhip_iir_process_f64_host)

    def process_device(self, d_in, n, in_pitch, d_out, out_pitch, stream=None):
        _lib.check(self._L.fskhip_iir_process_device(self._h, d_in, n, in_pitch, d_out, out_pitch, stream))

    def reset(self, stream=-1):
        _lib.check(self._L.fskhip_iir_reset(self._h, stream))

    def getCoefficients(self):
        b, a = (C.c_double * 9)(), (C.c_double * 9)()
        nb, na = C.c_uint32(), C.c_uint32()
        _lib.check(self._L.fskhip_iir_get_coefficients(self._h, b, C.byref(nb), a, C.byref(na)))
        return {"b": list(b[:nb.value]), "a": list(a[:na.value])}


class IIRFilter(IIRFilterBatch):
    """One IIRFilter with the reference's per-sample surface: process(x) takes and returns a number."""

    def __init__(self, b, a, device=0, precision=_lib.PRECISION_F64):
        super().__init__(b, a, 1, device, precision)

    def process(self, x):
        return float(self.processSamples(self._np.array([x], dtype=self._np.float64))[0])


class FilterFactory:
    """FilterFactory.createIIR* (filters.ts:325-344) and createFIR* (346-368)."""

    @staticmethod
    def createIIRLowpass(cutoffFreq, sampleRate, **kw):
        d = FilterDesign.butterworthLowpass(cutoffFreq, sampleRate)
        return IIRFilter(d["b"], d["a"], **kw)

    @staticmethod
    def createIIRHighpass(cutoffFreq, sampleRate, **kw):
        d = FilterDesign.butterworthHighpass(cutoffFreq, sampleRate)
        return IIRFilter(d["b"], d["a"], **kw)

    @staticmethod
    def createIIRBandpass(centerFreq, bandwidth, sampleRate, **kw):
        d = FilterDesign.butterworthBandpass(centerFreq, bandwidth, sampleRate)
        return IIRFilter(d["b"], d["a"], **kw)

    @staticmethod
    def createFIRLowpass(cutoffFreq, sampleRate, numTaps=51, **kw):
        return FIRFilter(FilterDesign.sincLowpass(cutoffFreq, sampleRate, numTaps), **kw)

    @staticmethod
    def createFIRHighpass(cutoffFreq, sampleRate, numTaps=51, **kw):
        return FIRFilter(FilterDesign.sincHighpass(cutoffFreq, sampleRate, numTaps), **kw)

    @staticmethod
    def createFIRBandpass(centerFreq, bandwidth, sampleRate, numTaps=51, **kw):
        return FIRFilter(FilterDesign.sincBandpass(centerFreq, bandwidth, sampleRate, numTaps), **kw)
