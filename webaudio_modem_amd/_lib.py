"""ctypes loader for libfskhip.so (the C ABI in include/fskhip.h and include/fskhip_next.h).

There is no fallback of any kind: if the HIP extension is missing this raises, and every
compute entry point fails with FSKHIP_E_NO_DEVICE when no GPU is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfskhip.so")   # (measurement tools load other builds by setting this before lib() runs)

MAX_PATTERN_BYTES = 16
OK = 0
E_INVALID, E_NOT_CONFIGURED, E_UNSUPPORTED, E_NO_DEVICE, E_HIP, E_NOMEM, E_OVERFLOW, E_HANDOFF = -1, -2, -3, -4, -5, -6, -7, -8
E_BUSY = -8
PROC_CLEAR_RX_ON_TX_COMPLETE, PROC_GRAPH = 1, 2
XM_NEED_MORE, XM_EOT, XM_TRUNCATED, XM_INVALID_SEQUENCE, XM_INVALID_CRC, XM_UNEXPECTED_SEQUENCE = 0, 1, 2, 3, 4, 5
PRECISION_F32, PRECISION_F64 = 0, 1
DEMOD_WRITEBACK_AGC = 1


class Config(C.Structure):
    """fskhip_config == FSKConfig (reference src/modems/fsk.ts:5-17)."""
    _fields_ = [
        ("sampleRate", C.c_double), ("baudRate", C.c_double),
        ("markFrequency", C.c_double), ("spaceFrequency", C.c_double),
        ("preamblePattern", C.c_int32 * MAX_PATTERN_BYTES), ("preambleLen", C.c_int32),
        ("sfdPattern", C.c_int32 * MAX_PATTERN_BYTES), ("sfdLen", C.c_int32),
        ("startBits", C.c_int32), ("stopBits", C.c_int32), ("parity", C.c_int32),
        ("syncThreshold", C.c_double), ("agcEnabled", C.c_int32),
        ("preFilterBandwidth", C.c_double), ("adaptiveThreshold", C.c_int32),
    ]


class SignalQuality(C.Structure):
    """fskhip_signal_quality (include/fskhip.h): opt-in estimates; the reference's getSignalQuality() is all zeros"""
    _fields_ = [(k, C.c_double) for k in ("snr", "ber", "eyeOpening", "phaseJitter", "frequencyOffset",
                                          "signalLevel", "noiseFloor", "frames", "bytes")]


class Status(C.Structure):
    """fskhip_status == getStatus() (fsk.ts:481-493) + agcGain + eodCount."""
    _fields_ = [
        ("ready", C.c_int32), ("frameStarted", C.c_int32),
        ("globalSampleCounter", C.c_double), ("receivedBitsLength", C.c_double),
        ("byteBufferLength", C.c_double), ("demodulationCalls", C.c_double),
        ("syncDetections", C.c_double), ("silenceThreshold", C.c_double),
        ("totalSamplesProcessed", C.c_double), ("agcGain", C.c_double), ("eodCount", C.c_double),
    ]


class XModemResult(C.Structure):
    """fskhip_xmodem_result (include/fskhip_next.h)."""
    _fields_ = [
        ("status", C.c_uint32), ("expected_after", C.c_uint32), ("packets", C.c_uint32), ("dropped", C.c_uint32),
        ("consumed", C.c_uint32), ("data_len", C.c_uint32), ("err_seq", C.c_int32), ("err_len", C.c_int32),
        ("crc_rx", C.c_int32), ("crc_calc", C.c_int32),
    ]


# every symbol include/*.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SYMBOLS = [
    ("fskhip_default_config", None, [C.POINTER(Config)]),
    ("fskhip_create", C.c_int, [C.POINTER(Config), C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(_P)]),
    ("fskhip_destroy", C.c_int, [_P]),
    ("fskhip_n_streams", C.c_uint32, [_P]),
    ("fskhip_max_bytes", C.c_size_t, [_P, C.c_size_t]),
    ("fskhip_last_kernel", C.c_char_p, [_P]),
    ("fskhip_blk_lanes", C.c_uint32, [_P]),
    ("fskhip_demodulate_host", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P, _P, C.c_uint32]),
    ("fskhip_demodulate_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P, _P, C.c_uint32, _P]),
    ("fskhip_modulated_length", C.c_size_t, [_P, C.c_size_t]),
    ("fskhip_modulate_host", C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    ("fskhip_modulate_device", C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_size_t, _P, _P]),
    ("fskhip_reset", C.c_int, [_P, C.c_int64]),
    ("fskhip_get_status", C.c_int, [_P, C.c_uint32, C.POINTER(Status)]),
    ("fskhip_get_faults", C.c_int, [_P, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("fskhip_synth_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint32,
                                      C.c_double, C.c_double, _P]),
    ("fskhip_synth_payload_byte", C.c_uint8, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("fskhip_synth_stream_params", None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_double)]),
    ("fskhip_add_awgn_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_double, C.c_uint64, _P]),
    ("fskhip_demod_supported", C.c_int, [_P]),
    ("fskhip_trace_enable", C.c_int, [_P, C.c_int64, C.c_size_t]),
    ("fskhip_trace_read_pre", C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("fskhip_trace_read", C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("fskhip_probe_read_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P]),
    ("fskhip_butterworth_lowpass", None, [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("fskhip_butterworth_highpass", None, [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("fskhip_butterworth_bandpass", None, [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]),
    ("fskhip_carry_over", C.c_int, [_P, _P]),
    ("fskhip_enable_signal_quality", C.c_int, [_P, C.c_int]),
    ("fskhip_get_signal_quality", C.c_int, [_P, C.c_uint32, C.POINTER(SignalQuality)]),
    ("fskhip_host_alloc", C.c_int, [C.c_size_t, C.POINTER(_P)]),
    ("fskhip_host_free", C.c_int, [_P]),
    ("fskhip_device_malloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    ("fskhip_device_free", C.c_int, [_P, _P]),
    ("fskhip_memcpy_h2d", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("fskhip_memcpy_d2h", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("fskhip_synchronize", C.c_int, [_P]),
    ("fskhip_set_option", C.c_int, [_P, C.c_char_p, C.c_char_p]),
    ("fskhip_debug_state", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("fskhip_clock_probe_begin", C.c_int, [_P, C.c_double]),
    ("fskhip_clock_probe_end", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("fskhip_timing_begin", C.c_int, [_P]),
    ("fskhip_timing_end", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]),
    ("fskhip_last_error", C.c_char_p, []),
    ("fskhip_abi_version", C.c_int, []),
    ("fskhip_device_count", C.c_int, []),
    # ---- include/fskhip_next.h ----
    ("fskhip_crc16_device", C.c_int, [_P, C.c_size_t, _P, C.c_uint32, _P, _P]),
    ("fskhip_crc16_host", C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_uint32, _P]),
    ("fskhip_xmodem_serialize_device", C.c_int, [_P, C.c_size_t, _P, _P, C.c_uint32, _P, C.c_size_t, _P, _P]),
    ("fskhip_xmodem_serialize_host", C.c_int, [C.c_int, _P, C.c_size_t, _P, _P, C.c_uint32, _P, C.c_size_t, _P]),
    ("fskhip_xmodem_scan_device", C.c_int, [_P, C.c_size_t, _P, _P, C.c_uint32, _P, C.c_size_t, _P, _P]),
    ("fskhip_xmodem_scan_host", C.c_int, [C.c_int, _P, C.c_size_t, _P, _P, C.c_uint32, _P, C.c_size_t, _P]),
    ("fskhip_processor_create", C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    ("fskhip_processor_destroy", C.c_int, [_P]),
    ("fskhip_processor_process_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_uint32, _P]),
    ("fskhip_processor_process_host", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_uint32]),
    ("fskhip_processor_modulate_host", C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    ("fskhip_processor_tx_state_host", C.c_int, [_P, _P, _P, _P, _P]),
    ("fskhip_processor_rx_drain_host", C.c_int, [_P, _P, C.c_size_t, _P]),
    ("fskhip_processor_rx_length_host", C.c_int, [_P, _P]),
    ("fskhip_processor_reset", C.c_int, [_P, C.c_int64]),
    ("fskhip_sinc_lowpass", C.c_int, [C.c_double, C.c_double, C.c_uint32, _P]),
    ("fskhip_sinc_highpass", C.c_int, [C.c_double, C.c_double, C.c_uint32, _P]),
    ("fskhip_sinc_bandpass", C.c_int, [C.c_double, C.c_double, C.c_double, C.c_uint32, _P]),
    ("fskhip_fir_create", C.c_int, [C.c_int, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
    ("fskhip_fir_destroy", C.c_int, [_P]),
    ("fskhip_fir_streams", C.c_uint32, [_P]),
    ("fskhip_iir_streams", C.c_uint32, [_P]),
    ("fskhip_fir_process_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P]),
    ("fskhip_fir_process_host", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t]),
    ("fskhip_fir_reset", C.c_int, [_P, C.c_int64]),
    ("fskhip_iir_create", C.c_int, [C.c_int, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
    ("fskhip_iir_destroy", C.c_int, [_P]),
    ("fskhip_iir_get_coefficients", C.c_int, [_P, _P, C.POINTER(C.c_uint32), _P, C.POINTER(C.c_uint32)]),
    ("fskhip_iir_process_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P]),
    ("fskhip_iir_process_host", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t]),
    ("fskhip_iir_process_f64_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P]),
    ("fskhip_iir_process_f64_host", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, C.c_size_t]),
    ("fskhip_iir_reset", C.c_int, [_P, C.c_int64]),
]
SYMBOL_NAMES = [s[0] for s in _SYMBOLS]

_lib = None


class FskHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fskhip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libfskhip.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libfskhip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C webaudio_modem_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise FskHipError(rc, lib().fskhip_last_error().decode("utf-8", "replace"))
    return rc
