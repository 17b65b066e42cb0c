"""ctypes binding of the CPU oracle (oracle/libfsk_oracle.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; nothing under webaudio_modem_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfsk_oracle.so")
MAX_PATTERN_BYTES = 16

PARITY = {"none": 0, "even": 1, "odd": 2}


class Config(C.Structure):
    _fields_ = [
        ("sampleRate", C.c_double), ("baudRate", C.c_double),
        ("markFrequency", C.c_double), ("spaceFrequency", C.c_double),
        ("preamblePattern", C.c_int32 * MAX_PATTERN_BYTES), ("preambleLen", C.c_int32),
        ("sfdPattern", C.c_int32 * MAX_PATTERN_BYTES), ("sfdLen", C.c_int32),
        ("startBits", C.c_int32), ("stopBits", C.c_int32), ("parity", C.c_int32),
        ("syncThreshold", C.c_double), ("agcEnabled", C.c_int32),
        ("preFilterBandwidth", C.c_double), ("adaptiveThreshold", C.c_int32),
    ]


class Status(C.Structure):
    _fields_ = [
        ("ready", C.c_int32), ("frameStarted", C.c_int32),
        ("globalSampleCounter", C.c_double), ("receivedBitsLength", C.c_double),
        ("byteBufferLength", C.c_double), ("demodulationCalls", C.c_double),
        ("syncDetections", C.c_double), ("silenceThreshold", C.c_double),
        ("totalSamplesProcessed", C.c_double), ("agcGain", C.c_double), ("eodCount", C.c_double),
    ]


class Trace(C.Structure):
    _fields_ = [
        ("cap", C.c_size_t), ("n", C.c_size_t),
        ("bit", C.POINTER(C.c_uint8)), ("amp", C.POINTER(C.c_double)),
        ("post_in", C.POINTER(C.c_double)), ("post_out", C.POINTER(C.c_double)),
        ("pre_cap", C.c_size_t), ("pre_n", C.c_size_t), ("pre_out", C.POINTER(C.c_float)),
    ]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it).  FSK_ORACLE_SANITIZE=1 selects the
    AddressSanitizer + UndefinedBehaviorSanitizer build (`make asan`); the process must then run with libasan preloaded
    (tests/test_oracle_sanitizers.py does: LD_PRELOAD=$(gcc -print-file-name=libasan.so))."""
    global _LIB_PATH
    src = os.path.join(_HERE, "fsk_oracle.c")
    target = "libfsk_oracle.so"
    if os.environ.get("FSK_ORACLE_SANITIZE") == "1":
        target = "libfsk_oracle_asan.so"
        _LIB_PATH = os.path.join(_HERE, target)
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "asan" if target.endswith("_asan.so") else target], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.fsko_create.restype = C.c_void_p
        L.fsko_configure.argtypes = [C.c_void_p, C.POINTER(Config)]
        L.fsko_destroy.argtypes = [C.c_void_p]
        L.fsko_demodulate.restype = C.c_long
        L.fsko_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_uint32)]
        L.fsko_modulate_length.restype = C.c_long
        L.fsko_modulate_length.argtypes = [C.c_void_p, C.c_size_t]
        L.fsko_modulate.restype = C.c_long
        L.fsko_modulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.fsko_reset.argtypes = [C.c_void_p]
        L.fsko_get_status.argtypes = [C.c_void_p, C.POINTER(Status)]
        L.fsko_set_trace.argtypes = [C.c_void_p, C.POINTER(Trace)]
        L.fsko_enable_quality.argtypes = [C.c_void_p, C.c_int]
        L.fsko_enable_quality.restype = None
        L.fsko_get_quality.argtypes = [C.c_void_p, C.POINTER(Quality)]
        L.fsko_get_quality.restype = None
        L.fsko_default_config.argtypes = [C.POINTER(Config)]
        for nm in ("lowpass", "highpass"):
            f = getattr(L, "fsko_butterworth_" + nm)
            f.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.fsko_butterworth_bandpass.argtypes = [C.c_double, C.c_double, C.c_double,
                                                C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.fsko_v8_sin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.fsko_v8_sin.restype = None
        L.fsko_sinc_lowpass.argtypes = [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.fsko_sinc_highpass.argtypes = [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.fsko_sinc_bandpass.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.fsko_iir_create.restype = C.c_void_p
        L.fsko_iir_create.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int]
        L.fsko_iir_destroy.argtypes = [C.c_void_p]
        L.fsko_iir_process.restype = C.c_double
        L.fsko_iir_process.argtypes = [C.c_void_p, C.c_double]
        L.fsko_iir_process_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.fsko_iir_reset.argtypes = [C.c_void_p]
        L.fsko_iir_coefficients.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.fsko_fir_create.restype = C.c_void_p
        L.fsko_fir_create.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.fsko_fir_destroy.argtypes = [C.c_void_p]
        L.fsko_fir_process.restype = C.c_double
        L.fsko_fir_process.argtypes = [C.c_void_p, C.c_double]
        L.fsko_fir_process_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.fsko_fir_reset.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class Quality(C.Structure):
    """fsko_quality (fsk_oracle.h): signal-quality estimates, an extension the oracle defines"""
    _fields_ = [(k, C.c_double) for k in ("snr", "ber", "eyeOpening", "phaseJitter", "frequencyOffset",
                                          "signalLevel", "noiseFloor", "frames", "bytes")]


def make_config(cfg=None):
    """dict with the reference's FSKConfig field names (partial: merged over the defaults)."""
    c = Config()
    lib().fsko_default_config(C.byref(c))
    for k, v in (cfg or {}).items():
        if k == "preamblePattern":
            c.preambleLen = len(v)
            for i, b in enumerate(v):
                c.preamblePattern[i] = int(b)
        elif k == "sfdPattern":
            c.sfdLen = len(v)
            for i, b in enumerate(v):
                c.sfdPattern[i] = int(b)
        elif k == "parity":
            c.parity = PARITY[v] if isinstance(v, str) else int(v)
        elif k in ("agcEnabled", "adaptiveThreshold"):
            setattr(c, k, 1 if v else 0)
        else:
            setattr(c, k, v)
    return c


class OracleCore:
    """Mirror of the reference FSKCore public surface, backed by the C oracle."""

    def __init__(self, cfg=None, configure=True):
        self._h = lib().fsko_create()
        self._trace = None
        self.eod_events = 0
        if configure:
            self.configure(cfg)

    def configure(self, cfg=None):
        c = make_config(cfg)
        if lib().fsko_configure(self._h, C.byref(c)) != 0:
            raise ValueError("oracle: bad config")
        self.cfg = c

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fsko_destroy(self._h)
            self._h = None

    def demodulate(self, samples, mutate=False):
        """Returns (bytes, eod_count). `samples` float32; copied unless mutate=True."""
        x = np.ascontiguousarray(samples, dtype=np.float32)
        if not mutate:
            x = x.copy()
        cap = max(16, x.size // 8 + 16)
        out = np.empty(cap, dtype=np.uint8)
        eod = C.c_uint32(0)
        n = lib().fsko_demodulate(self._h, x.ctypes.data, x.size, out.ctypes.data, cap, C.byref(eod))
        if n < 0:
            raise RuntimeError("FSK demodulator not configured")
        assert n <= cap
        self.eod_events += eod.value
        self.last_agc_out = x
        return out[:n].tobytes(), eod.value

    def modulate(self, data):
        d = np.frombuffer(bytes(data), dtype=np.uint8)
        n = lib().fsko_modulate_length(self._h, d.size)
        if n < 0:
            raise RuntimeError("FSK modulator not configured")
        out = np.empty(n, dtype=np.float32)
        r = lib().fsko_modulate(self._h, d.ctypes.data if d.size else None, d.size, out.ctypes.data, n)
        assert r == n
        return out

    def reset(self):
        lib().fsko_reset(self._h)

    def status(self):
        st = Status()
        lib().fsko_get_status(self._h, C.byref(st))
        return {k: getattr(st, k) for k, _ in Status._fields_}

    def enable_quality(self, on=True):
        lib().fsko_enable_quality(self._h, 1 if on else 0)

    def quality(self):
        q = Quality()
        lib().fsko_get_quality(self._h, C.byref(q))
        return {k: getattr(q, k) for k, _ in Quality._fields_}

    def enable_trace(self, cap, pre_cap):
        t = Trace()
        self._tr_arrays = dict(
            bit=np.zeros(cap, np.uint8), amp=np.zeros(cap, np.float64),
            post_in=np.zeros(cap, np.float64), post_out=np.zeros(cap, np.float64),
            pre_out=np.zeros(pre_cap, np.float32))
        t.cap, t.n, t.pre_cap, t.pre_n = cap, 0, pre_cap, 0
        t.bit = self._tr_arrays["bit"].ctypes.data_as(C.POINTER(C.c_uint8))
        t.amp = self._tr_arrays["amp"].ctypes.data_as(C.POINTER(C.c_double))
        t.post_in = self._tr_arrays["post_in"].ctypes.data_as(C.POINTER(C.c_double))
        t.post_out = self._tr_arrays["post_out"].ctypes.data_as(C.POINTER(C.c_double))
        t.pre_out = self._tr_arrays["pre_out"].ctypes.data_as(C.POINTER(C.c_float))
        self._trace = t
        lib().fsko_set_trace(self._h, C.byref(t))

    def trace(self):
        t, a = self._trace, self._tr_arrays
        return dict(bit=a["bit"][:t.n], amp=a["amp"][:t.n], post_in=a["post_in"][:t.n],
                    post_out=a["post_out"][:t.n], pre_out=a["pre_out"][:t.pre_n])


def _d3():
    return (C.c_double * 3)(), (C.c_double * 3)()


def butterworth_lowpass(fc, sr):
    b, a = _d3()
    lib().fsko_butterworth_lowpass(fc, sr, b, a)
    return list(b), list(a)


def butterworth_highpass(fc, sr):
    b, a = _d3()
    lib().fsko_butterworth_highpass(fc, sr, b, a)
    return list(b), list(a)


def butterworth_bandpass(fc, bw, sr):
    b, a = _d3()
    lib().fsko_butterworth_bandpass(fc, bw, sr, b, a)
    return list(b), list(a)


def sinc(kind, *args):
    nt = int(args[-1])
    buf = (C.c_double * (nt + 2))()
    fn = getattr(lib(), "fsko_sinc_" + kind)
    n = fn(*[float(x) for x in args[:-1]], nt, buf)
    return list(buf)[:n]


class IIR:
    def __init__(self, b, a):
        bb = (C.c_double * len(b))(*b)
        aa = (C.c_double * len(a))(*a)
        self._h = lib().fsko_iir_create(bb, len(b), aa, len(a))
        if not self._h:
            raise ValueError("IIRFilter constructor would throw")
        self.nb, self.na = len(b), len(a)

    def process(self, x):
        return lib().fsko_iir_process(self._h, float(x))

    def process_buffer(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        lib().fsko_iir_process_buffer(self._h, x.ctypes.data, y.ctypes.data, x.size)
        return y

    def reset(self):
        lib().fsko_iir_reset(self._h)

    def coefficients(self):
        b = (C.c_double * self.nb)()
        a = (C.c_double * self.na)()
        lib().fsko_iir_coefficients(self._h, b, a)
        return list(b), list(a)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fsko_iir_destroy(self._h)


class FIR:
    def __init__(self, taps):
        t = (C.c_double * len(taps))(*taps)
        self._h = lib().fsko_fir_create(t, len(taps))

    def process(self, x):
        return lib().fsko_fir_process(self._h, float(x))

    def process_buffer(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        lib().fsko_fir_process_buffer(self._h, x.ctypes.data, y.ctypes.data, x.size)
        return y

    def reset(self):
        lib().fsko_fir_reset(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fsko_fir_destroy(self._h)


def v8_sin(x):
    """Math.sin as V8 computes it (oracle/v8_sin.h) for an array of doubles."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().fsko_v8_sin(x.ctypes.data, y.ctypes.data, x.size)
    return y
