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
