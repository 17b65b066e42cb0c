"""Batch FSK engine: Python host side above the C ABI (include/fskhip.h).

`FSKEngine` holds S independent FSKCore-equivalent streams on one MI355X.  Names and argument
meaning follow the reference's FSKCore (src/modems/fsk.ts): configure-time `FSKConfig` dicts
with the reference's field names, `demodulate_data` / `modulate_data` == demodulateData /
modulateData applied to every stream, `reset`, `get_status`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, Status, FskHipError, PRECISION_F32, PRECISION_F64, DEMOD_WRITEBACK_AGC  # noqa: F401

PARITY = {"none": 0, "even": 1, "odd": 2}

# DEFAULT_FSK_CONFIG (fsk.ts:19-33)
DEFAULT_FSK_CONFIG = dict(
    sampleRate=48000, baudRate=1200, markFrequency=1650, spaceFrequency=1850,
    preamblePattern=[0x55, 0x55], sfdPattern=[0x7E], startBits=1, stopBits=1, parity="none",
    syncThreshold=0.85, agcEnabled=True, preFilterBandwidth=800, adaptiveThreshold=True)


def make_config(cfg=None):
    """Merge a (partial) FSKConfig dict over DEFAULT_FSK_CONFIG, like configure() does (fsk.ts:134)."""
    merged = dict(DEFAULT_FSK_CONFIG)
    merged.update(cfg or {})
    c = Config()
    for k, v in merged.items():
        if k == "preamblePattern":
            if len(v) > _lib.MAX_PATTERN_BYTES:
                raise ValueError("preamblePattern longer than %d bytes" % _lib.MAX_PATTERN_BYTES)
            c.preambleLen = len(v)
            for i, b in enumerate(v):
                c.preamblePattern[i] = int(b)
        elif k == "sfdPattern":
            if len(v) > _lib.MAX_PATTERN_BYTES:
                raise ValueError("sfdPattern longer than %d bytes" % _lib.MAX_PATTERN_BYTES)
            c.sfdLen = len(v)
            for i, b in enumerate(v):
                c.sfdPattern[i] = int(b)
        elif k == "parity":
            c.parity = PARITY[v] if isinstance(v, str) else int(v)
        elif k in ("agcEnabled", "adaptiveThreshold"):
            setattr(c, k, 1 if v else 0)
        elif hasattr(c, k):
            setattr(c, k, v)
        else:
            raise KeyError("unknown FSKConfig field %r" % k)
    return c, merged


def _status_dict(st):
    return {
        "ready": bool(st.ready), "frameStarted": bool(st.frameStarted),
        "globalSampleCounter": int(st.globalSampleCounter), "receivedBitsLength": int(st.receivedBitsLength),
        "byteBufferLength": int(st.byteBufferLength), "demodulationCalls": int(st.demodulationCalls),
        "syncDetections": int(st.syncDetections), "silenceThreshold": float(st.silenceThreshold),
        "totalSamplesProcessed": int(st.totalSamplesProcessed),
        "agcGain": float(st.agcGain), "eodCount": int(st.eodCount),
    }


class _PinnedBlock:
    """owner of one fskhip_host_alloc allocation (freed when the last array viewing it is collected)"""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _lib.check(_lib.lib().fskhip_host_alloc(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().fskhip_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """numpy array in page-locked host memory (fskhip_host_alloc): the buffer to hand to demodulate_data /
    modulate_data when the PCIe copies should overlap the kernels (include/fskhip.h, fskhip_demodulate_host)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    blk = _PinnedBlock(max(1, n * dt.itemsize))
    buf = (C.c_char * blk.nbytes).from_address(blk.ptr)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    _PINNED_OWNERS[id(buf)] = blk          # keep the block alive as long as the ctypes buffer is
    import weakref
    weakref.finalize(buf, _PINNED_OWNERS.pop, id(buf), None)
    return arr


_PINNED_OWNERS = {}


# Called with (n_streams, precision) by every new FSKEngine; returns a dict of fskhip_set_option() names -> values to apply on
# top of the `options` argument, or None.  The package sets nothing here and reads no environment variable; the test suite
# (tests/conftest.py) and the measurement tools (tools/envopts.py) install a hook that maps their FSKHIP_* variables.
option_hook = None


class FSKEngine:
    """S FSKCore instances on one GPU.

    configs: one FSKConfig dict (shared) or a list of S dicts that differ only in
    markFrequency / spaceFrequency / preFilterBandwidth (BASELINE config #4).
    options: tuning / test switches by name (include/fskhip.h, fskhip_set_option); none changes a result.
    """

    def __init__(self, n_streams, configs=None, device=0, precision=PRECISION_F32, options=None):
        L = _lib.lib()
        if isinstance(configs, (list, tuple)):
            if len(configs) != n_streams:
                raise ValueError("need one config per stream")
            made = [make_config(c) for c in configs]
            arr = (Config * n_streams)(*[m[0] for m in made])
            self.config = made[0][1]
            n_cfgs = n_streams
        else:
            c, self.config = make_config(configs)
            arr = (Config * 1)(c)
            n_cfgs = 1
        h = C.c_void_p()
        _lib.check(L.fskhip_create(arr, n_cfgs, n_streams, device, precision, C.byref(h)))
        self._h = h
        self._L = L
        self.n_streams = n_streams
        self.device = device
        self.precision = precision
        opts = dict(options or {})
        if option_hook is not None:
            opts.update(option_hook(n_streams, precision) or {})
        try:
            for k, v in opts.items():
                self.set_option(k, v)
        except Exception:
            self.close()
            raise

    def set_option(self, name, value):
        """fskhip_set_option (include/fskhip.h): before the first demodulate call"""
        _lib.check(self._L.fskhip_set_option(self._h, str(name).encode(), str(value).encode()))

    def close(self):
        if getattr(self, "_h", None):
            self._L.fskhip_destroy(self._h)
            self._h = None

    def carry_over_from(self, old):
        """what FSKCore.configure() leaves in place on a configured instance: silence threshold, debug counters"""
        _lib.check(self._L.fskhip_carry_over(self._h, old._h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- demodulateData (fsk.ts:190-222) -------------------------------------------------------
    def max_bytes(self, n_per_stream):
        """Upper bound on bytes one call can produce per stream (fskhip_max_bytes: size out_pitch with it)."""
        return int(self._L.fskhip_max_bytes(self._h, int(n_per_stream)))

    def last_kernel(self):
        """Name of the kernel the last demodulate_device call launched for its whole tiles (fskhip_last_kernel)."""
        return (self._L.fskhip_last_kernel(self._h) or b"").decode()

    def blk_lanes(self):
        """Streams per workgroup of the four-wave kernel for this engine (fskhip_blk_lanes): 64, 32, 16, 8, or 0."""
        return int(self._L.fskhip_blk_lanes(self._h))

    def demodulate_data(self, samples, writeback_agc=False, out_pitch=None):
        """samples: float32 [S, N] (host).  Returns (list of bytes per stream, eod counts ndarray).

        With writeback_agc=True `samples` is overwritten with the AGC-scaled samples, the side
        effect the reference has on its input buffer (fsk.ts:55, 201).
        """
        x = samples if (writeback_agc and isinstance(samples, np.ndarray) and samples.dtype == np.float32
                        and samples.flags.c_contiguous) else np.ascontiguousarray(samples, dtype=np.float32)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        S, N = x.shape
        if S != self.n_streams:
            raise ValueError("expected %d streams, got %d" % (self.n_streams, S))
        pitch = out_pitch or self.max_bytes(N)
        out = np.zeros((S, pitch), dtype=np.uint8)
        counts = np.zeros(S, dtype=np.uint32)
        eod = np.zeros(S, dtype=np.uint32)
        flags = DEMOD_WRITEBACK_AGC if writeback_agc else 0
        _lib.check(self._L.fskhip_demodulate_host(
            self._h, x.ctypes.data, N, x.strides[0] // 4 if N else max(N, 1), out.ctypes.data, pitch,
            counts.ctypes.data, eod.ctypes.data, flags))
        if writeback_agc and x is not samples:
            np.copyto(samples, x.reshape(samples.shape))
        return [out[s, :counts[s]].tobytes() for s in range(S)], eod

    def demodulate_device(self, d_samples, n_per_stream, pitch, d_out, out_pitch, d_counts, d_eod=None,
                          flags=0, stream=None):
        """Device-pointer form (ints): asynchronous on `stream` (a hipStream_t handle or None)."""
        _lib.check(self._L.fskhip_demodulate_device(self._h, d_samples, n_per_stream, pitch, d_out, out_pitch,
                                                    d_counts, d_eod, flags, stream))

    # ---- modulateData (fsk.ts:377-424) ---------------------------------------------------------
    def modulated_length(self, n_bytes):
        return int(self._L.fskhip_modulated_length(self._h, n_bytes))

    def modulate_data(self, payloads):
        """payloads: list of S bytes-like.  Returns list of float32 arrays (one signal per stream)."""
        if len(payloads) != self.n_streams:
            raise ValueError("need one payload per stream")
        S = self.n_streams
        lens = np.array([len(p) for p in payloads], dtype=np.uint32)
        ppitch = max(1, int(lens.max()))
        pay = np.zeros((S, ppitch), dtype=np.uint8)
        for s, p in enumerate(payloads):
            if len(p):
                pay[s, :len(p)] = np.frombuffer(bytes(p), dtype=np.uint8)
        opitch = max(4, self.modulated_length(int(lens.max())))
        out = np.zeros((S, opitch), dtype=np.float32)
        out_lens = np.zeros(S, dtype=np.uint32)
        _lib.check(self._L.fskhip_modulate_host(self._h, pay.ctypes.data, lens.ctypes.data, ppitch, out.ctypes.data,
                                                opitch, out_lens.ctypes.data))
        return [out[s, :out_lens[s]].copy() for s in range(S)]

    def modulate_device(self, d_payloads, d_lens, payload_pitch, d_out, out_pitch, d_out_lens, stream=None):
        _lib.check(self._L.fskhip_modulate_device(self._h, d_payloads, d_lens, payload_pitch, d_out, out_pitch,
                                                  d_out_lens, stream))

    # ---- reset / getStatus ---------------------------------------------------------------------
    def reset(self, stream=-1):
        _lib.check(self._L.fskhip_reset(self._h, stream))

    def get_status(self, stream=0):
        st = Status()
        _lib.check(self._L.fskhip_get_status(self._h, stream, C.byref(st)))
        return _status_dict(st)

    def faults(self):
        """uint8[S]: 1 = the stream's filter state is no longer finite (fskhip_get_faults): it absorbed a NaN / Inf sample -- the
        reference's instance is dead from there on too, and the engine does what it does -- or, fp32 engines only, a sample beyond
        their range (~1e19)."""
        out = np.zeros(self.n_streams, np.uint8)
        n = C.c_uint32(0)
        _lib.check(self._L.fskhip_get_faults(self._h, out.ctypes.data, C.byref(n)))
        assert int(out.sum()) == n.value
        return out

    def fault(self, stream=0):
        return bool(self.faults()[stream])

    # ---- opt-in signal-quality estimates (include/fskhip.h; the reference's getSignalQuality() returns zeros) -------
    def enable_signal_quality(self, on=True):
        _lib.check(self._L.fskhip_enable_signal_quality(self._h, 1 if on else 0))

    def get_signal_quality(self, stream=0):
        q = _lib.SignalQuality()
        _lib.check(self._L.fskhip_get_signal_quality(self._h, stream, C.byref(q)))
        return {k: getattr(q, k) for k, _ in _lib.SignalQuality._fields_}

    def demod_supported(self):
        return bool(self._L.fskhip_demod_supported(self._h))

    # ---- intermediate capture (parity tests) -----------------------------------------------------
    def trace_enable(self, stream, capacity):
        _lib.check(self._L.fskhip_trace_enable(self._h, stream, capacity))
        self._trace_cap = capacity

    def trace_read(self):
        cap = self._trace_cap
        amp = np.zeros(cap, np.float64)
        post = np.zeros(cap, np.float64)
        bit = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        _lib.check(self._L.fskhip_trace_read(self._h, amp.ctypes.data, post.ctypes.data, bit.ctypes.data, cap,
                                             C.byref(n)))
        pre = np.zeros(2 * cap, np.float64)
        npre = C.c_size_t(0)
        _lib.check(self._L.fskhip_trace_read_pre(self._h, pre.ctypes.data, 2 * cap, C.byref(npre)))
        return {"amp": amp[:n.value], "post_out": post[:n.value], "bit": bit[:n.value], "pre_out": pre[:min(npre.value, 2 * cap)]}

    # ---- measurement tooling -------------------------------------------------------------------
    def synth_device(self, d_out, n_per_stream, pitch, payload_len, seed, lead_max, amp_lo, amp_hi, stream=None):
        _lib.check(self._L.fskhip_synth_device(self._h, d_out, n_per_stream, pitch, payload_len, seed, lead_max,
                                               amp_lo, amp_hi, stream))

    def add_awgn_device(self, d_buf, n_per_stream, pitch, snr_db, seed, stream=None):
        _lib.check(self._L.fskhip_add_awgn_device(self._h, d_buf, n_per_stream, pitch, snr_db, seed, stream))

    def probe_read_device(self, d_buf, n_per_stream, pitch, stream=None):
        _lib.check(self._L.fskhip_probe_read_device(self._h, d_buf, n_per_stream, pitch, stream))

    def synth_payload(self, seed, stream, frame, payload_len):
        f = self._L.fskhip_synth_payload_byte
        return bytes(f(seed, stream, frame, i) for i in range(payload_len))

    def synth_stream_params(self, seed, stream, lead_max, amp_lo, amp_hi):
        lead, amp = C.c_uint32(), C.c_double()
        self._L.fskhip_synth_stream_params(seed, stream, lead_max, amp_lo, amp_hi, C.byref(lead), C.byref(amp))
        return lead.value, amp.value

    def device_malloc(self, nbytes):
        p = C.c_void_p()
        _lib.check(self._L.fskhip_device_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr):
        _lib.check(self._L.fskhip_device_free(self._h, ptr))

    def h2d(self, d_dst, arr):
        a = np.ascontiguousarray(arr)
        _lib.check(self._L.fskhip_memcpy_h2d(self._h, d_dst, a.ctypes.data, a.nbytes))

    def d2h(self, arr, d_src):
        assert arr.flags.c_contiguous
        _lib.check(self._L.fskhip_memcpy_d2h(self._h, arr.ctypes.data, d_src, arr.nbytes))

    def synchronize(self):
        _lib.check(self._L.fskhip_synchronize(self._h))

    def debug_state(self, stream):
        """(real words, integer words) of one stream's carried state, fsk_params.h order (diagnostics)"""
        r, i = (C.c_double * 128)(), (C.c_uint32 * 128)()
        nr, ni = C.c_uint32(), C.c_uint32()
        _lib.check(self._L.fskhip_debug_state(self._h, stream, r, 128, i, 128, C.byref(nr), C.byref(ni)))
        return list(r[:nr.value]), list(i[:ni.value])

    def clock_probe_begin(self, spin_ms):
        """start the shader-clock probe (include/fskhip.h); launch the work to observe behind it on other streams"""
        _lib.check(self._L.fskhip_clock_probe_begin(self._h, float(spin_ms)))

    def clock_probe_end(self):
        """-> (shader clock in GHz, milliseconds the probe covered)"""
        ghz, ms = C.c_double(), C.c_double()
        _lib.check(self._L.fskhip_clock_probe_end(self._h, C.byref(ghz), C.byref(ms)))
        return ghz.value, ms.value

    def timing_begin(self):
        _lib.check(self._L.fskhip_timing_begin(self._h))

    def timing_end(self):
        n, ms = C.c_uint32(), C.c_double()
        _lib.check(self._L.fskhip_timing_end(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value
