/*
 * fskhip.h -- C ABI of libfskhip.so: the MI355X (gfx950) batch FSK DSP engine.
 *
 * This is the drop-in boundary for ONE path of cho45/WebAudio-Modem: FSKCore.modulateData /
 * demodulateData (src/modems/fsk.ts:190-222, 377-424) and the dsp/filters.ts IIR chain they
 * call, batched over independent streams (one GPU lane per stream).  Plain pointers and sizes
 * only.  Every entry point cites the reference interface it replaces; the binding a
 * maintainer adds on the reference side (N-API) is shown in INTEGRATION.md.
 *
 * All functions return FSKHIP_OK (0) or a negative FSKHIP_E_* code; fskhip_last_error()
 * returns a thread-local message for the last failure.  There is NO CPU fallback: without a
 * usable HIP device fskhip_create() fails with FSKHIP_E_NO_DEVICE.
 */
#ifndef FSKHIP_H
#define FSKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: fskhip_max_bytes, fskhip_last_kernel.  3: fskhip_carry_over, fskhip_host_alloc / _free, the pipelined
 * fskhip_demodulate_host, fskhip_enable_signal_quality / fskhip_get_signal_quality.  4: fskhip_set_option (the library reads no
 * environment variable any more), fskhip_clock_probe_begin / _end, fskhip_debug_state.  5: fskhip_blk_lanes, option
 * "blk_lanes" (additions only).  6: kernel = seven-wave / auto-r04, options "stage_min_tiles" / "stage_y_slots" / "stage_roles"; the batched
 * IIRFilter of fskhip_next.h (additions only).  7: fskhip_get_faults (additions only).  8: FSKHIP_E_HANDOFF -- every hand-off wait of
 * the multi-wave kernels is bounded ("Hand-off waits" below; an addition: no healthy call returns it). */
#define FSKHIP_ABI_VERSION 8
#define FSKHIP_MAX_PATTERN_BYTES 16

enum {
  FSKHIP_OK = 0,
  FSKHIP_E_INVALID = -1,        /* bad argument */
  FSKHIP_E_NOT_CONFIGURED = -2, /* reference: throws 'FSK (de)modulator not configured' fsk.ts:191-193,378-380 */
  FSKHIP_E_UNSUPPORTED = -3,    /* configuration outside what the kernels implement (see fskhip_create) */
  FSKHIP_E_NO_DEVICE = -4,      /* no HIP device / HIP runtime failure at create */
  FSKHIP_E_HIP = -5,            /* HIP runtime error during a call */
  FSKHIP_E_NOMEM = -6,
  FSKHIP_E_OVERFLOW = -7,       /* an output slab was too small; counts still report the true size */
  FSKHIP_E_HANDOFF = -8         /* ABI 8: a wave of a multi-wave kernel waited for its neighbour beyond the bound and ended the launch
                                   early ("Hand-off waits" below); sticky, the engine is to be destroyed */
};

/* arithmetic the demodulator chain computes in */
enum {
  FSKHIP_PRECISION_F32 = 0, /* throughput path: fp32 VALU, hardware sin/cos/rcp/sqrt */
  FSKHIP_PRECISION_F64 = 1  /* parity path: fp64, op-for-op with the reference's double arithmetic */
};

/* FSKConfig (fsk.ts:5-17) + BaseModulatorConfig (core.ts:3-6), field for field. */
typedef struct fskhip_config {
  double sampleRate;
  double baudRate;
  double markFrequency;
  double spaceFrequency;
  int32_t preamblePattern[FSKHIP_MAX_PATTERN_BYTES];
  int32_t preambleLen;
  int32_t sfdPattern[FSKHIP_MAX_PATTERN_BYTES];
  int32_t sfdLen;
  int32_t startBits;
  int32_t stopBits;
  int32_t parity; /* 0 'none', 1 'even', 2 'odd' */
  double syncThreshold;
  int32_t agcEnabled;
  double preFilterBandwidth;
  int32_t adaptiveThreshold; /* accepted and ignored, as in the reference (fsk.ts:16,32: never read) */
} fskhip_config;

/* getStatus() (fsk.ts:481-493) for one stream, plus the AGC gain and the eod total. */
typedef struct fskhip_status {
  int32_t ready;
  int32_t frameStarted;
  double globalSampleCounter;
  double receivedBitsLength;
  double byteBufferLength;       /* always 0 between calls: demodulateData drains it (fsk.ts:210-214) */
  double demodulationCalls;
  double syncDetections;
  double silenceThreshold;
  double totalSamplesProcessed;
  double agcGain;                /* NaN when AGC is disabled */
  double eodCount;               /* 'eod' events since create (fsk.ts:289) */
} fskhip_status;

typedef struct fskhip_engine fskhip_engine;

/* DEFAULT_FSK_CONFIG (fsk.ts:19-33). */
void fskhip_default_config(fskhip_config *cfg);

/*
 * new FSKCore() + configure(cfg) for n_streams independent demodulator/modulator instances
 * (fsk.ts:133-157).  n_cfgs is 1 (all streams share cfgs[0]) or n_streams (per-stream
 * markFrequency / spaceFrequency / preFilterBandwidth; every other field must be equal across
 * streams -- BASELINE config #4).  `device` is the HIP device ordinal.  `precision` is
 * FSKHIP_PRECISION_*.
 * The modulator supports every configuration.  The demodulator reproduces the reference's
 * FRACTIONAL sync-ring capacities too (maxSyncBits*dsSPB*1.1 not an integer, fsk.ts:149: 44.1 kHz,
 * or 48 kHz with parity / two stop bits / longer preambles; the reference's RingBuffer freezes
 * after floor(capacity) pushes there, utils.ts:38-48, and so does this one).  Not implemented:
 * capacities whose fractional write index turns integral again within 2^40 pushes (x.5 and the
 * like), more than 63 preamble+SFD pattern bits, dsSPB too large for LDS: create succeeds,
 * fskhip_demod_supported() is 0 and fskhip_demodulate_* return FSKHIP_E_UNSUPPORTED with the
 * reason (loud, no fallback).
 * FSKHIP_E_UNSUPPORTED at create: per-stream configs that differ in a shared field.
 */
int fskhip_create(const fskhip_config *cfgs, uint32_t n_cfgs, uint32_t n_streams, int device,
                  int precision, fskhip_engine **out);
int fskhip_destroy(fskhip_engine *e);

/*
 * FSKCore.configure() on an already configured instance (fsk.ts:133-157) rebuilds everything and calls
 * resetState(), which leaves silence.threshold (fsk.ts:128, 321-326) and the debug counters
 * (fsk.ts:131) of the old life in place.  A host re-configures with fskhip_create + this + fskhip_destroy
 * of the old engine; the two engines must agree in stream count, precision and device.
 */
int fskhip_carry_over(fskhip_engine *dst, const fskhip_engine *src);

uint32_t fskhip_n_streams(const fskhip_engine *e);

/*
 * Upper bound on the bytes ONE demodulate call of n_per_stream samples can return per stream (a byte
 * takes bitsPerByte >= 8 bit times of samplesPerBit samples, fsk.ts:438-439): size out_pitch with it.
 */
size_t fskhip_max_bytes(const fskhip_engine *e, size_t n_per_stream);

/*
 * Name of the kernel the last fskhip_demodulate_device call launched for its whole 16-sample tiles
 * ("" before the first call): measurement harnesses report it instead of guessing the dispatch.
 */
const char *fskhip_last_kernel(const fskhip_engine *e);

/*
 * Streams per workgroup of the four-wave whole-tile kernel for this engine: 64 (a whole wave), or 32 / 16 / 8 when the
 * batch is too small to give every compute unit a 64-stream group ("narrow groups": more, narrower workgroups use the
 * idle CUs; results are identical).  0 if that kernel does not apply to the engine.  For measurement harnesses.
 */
uint32_t fskhip_blk_lanes(const fskhip_engine *e);

/*
 * demodulateData(samples) (fsk.ts:190-222) for every stream: `samples` is [n_streams][pitch]
 * float32 (stream-major, `pitch` in floats >= n_per_stream), all streams advance by
 * n_per_stream samples; per-stream state persists across calls like one FSKCore instance per
 * stream.  Decoded bytes of stream s go to out[s*out_pitch ...], their number to
 * out_counts[s] (the true count, even if it exceeds out_pitch -> FSKHIP_E_OVERFLOW), the number
 * of 'eod' events emitted during the call to eod_counts[s] (may be NULL).
 * flags: FSKHIP_DEMOD_WRITEBACK_AGC writes the AGC-scaled samples back into `samples`, which
 * the reference does as a side effect (fsk.ts:55,201).
 * The _host form takes host pointers (H2D/D2H inside, synchronous); a call of more than ~1.5 time
 * slabs (96 MB of samples each) is pipelined: the next slab crosses PCIe while the current one is
 * demodulated -- fully only if `samples` is page-locked memory (fskhip_host_alloc below, or the
 * caller's own pinned buffer).  The _device form takes device pointers, is asynchronous on
 * `hip_stream` (a hipStream_t, NULL = default stream) and moves nothing over PCIe.
 */
#define FSKHIP_DEMOD_WRITEBACK_AGC 1u
int fskhip_demodulate_host(fskhip_engine *e, float *samples, size_t n_per_stream, size_t pitch,
                           uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                           uint32_t *eod_counts, uint32_t flags);
int fskhip_demodulate_device(fskhip_engine *e, float *d_samples, size_t n_per_stream, size_t pitch,
                             uint8_t *d_out, size_t out_pitch, uint32_t *d_out_counts,
                             uint32_t *d_eod_counts, uint32_t flags, void *hip_stream);

/*
 * modulateData(bytes) (fsk.ts:377-424) for every stream.  Stream s modulates
 * payloads[s*payload_pitch .. + lens[s]) with ITS configuration into out[s*out_pitch ...];
 * out_lens[s] receives the signal length: totalBytes*bitsPerByte*spb + 2*spb + bitsPerByte*spb
 * (fsk.ts:391-394).  fskhip_modulated_length() gives that length for a payload size.
 * FSKHIP_PRECISION_F64 engines evaluate Math.sin with the operation sequence of the engine the reference runs on
 * (V8's fdlibm port), so the Float32Array is bit-identical at any length; FSKHIP_PRECISION_F32 engines use the
 * device library's sin() (about 1.8x faster; a sample can differ by one float ulp where two correct libms round the
 * double differently, about once in 1e9 samples).
 */
size_t fskhip_modulated_length(const fskhip_engine *e, size_t n_bytes);
int fskhip_modulate_host(fskhip_engine *e, const uint8_t *payloads, const uint32_t *lens,
                         size_t payload_pitch, float *out, size_t out_pitch, uint32_t *out_lens);
int fskhip_modulate_device(fskhip_engine *e, const uint8_t *d_payloads, const uint32_t *d_lens,
                           size_t payload_pitch, float *d_out, size_t out_pitch,
                           uint32_t *d_out_lens, void *hip_stream);

/* reset() (fsk.ts:464-469) for one stream, or all when stream < 0. */
int fskhip_reset(fskhip_engine *e, int64_t stream);
/* getStatus() (fsk.ts:481-493). Synchronises with outstanding work of the engine. */
int fskhip_get_status(fskhip_engine *e, uint32_t stream, fskhip_status *st);

/*
 * Streams whose filter state has left the finite range (ABI 7).  out[s] = 1 for such a stream (n_streams bytes, or NULL),
 * *n_faulty = how many (or NULL).  What puts a stream there, and what the engines then do (tests/test_gpu_hostile.py against
 * tests/golden/golden_hostile.npz, the REAL reference on the same samples):
 *   - a NaN or +-Inf sample.  The reference's pre-filter is never reset (fsk.ts:175-188, filters.ts:47-76), so its instance
 *     is dead from that sample on: every sliced bit is 0 (`NaN > 0`, fsk.ts:264), the amplitude is never below the silence
 *     threshold (`NaN < t`, fsk.ts:285) -- no further bytes, no further 'eod', until it is configured again.  BOTH precisions
 *     do exactly that, bit for bit: same bytes, same 'eod' count, same status as the reference; the flag tells the host why a
 *     stream went quiet.  Other streams of the batch are not affected.
 *   - FSKHIP_PRECISION_F32 only: a finite sample so large that the fp32 I/Q branch (which runs 2^60 times larger than the
 *     reference's values) overflows: |x * agcGain| beyond ~1e19, where the reference's doubles go on decoding.  From that call
 *     on the stream's output is NOT the reference's (it stays quiet, like a poisoned one) and this flag is how the host learns
 *     of it; FSKHIP_PRECISION_F64 engines follow the reference through the whole float range.
 * Subnormal samples need no flag: both precisions follow the reference down to the last subnormal bit -- with one exception, the
 * per-sample generic fp32 kernel (streams out of lock step, > 31 pattern bits, fractional ring capacities, "force_generic"),
 * whose I/Q branch is not scaled: a frame below ~1e-38 of full scale is a handful of subnormal bits there, and whether it
 * syncs on one (for the reference itself a marginal decision) can differ from the reference.
 */
int fskhip_get_faults(fskhip_engine *e, uint8_t *out, uint32_t *n_faulty);

/*
 * Synthetic multi-stream workload generator (measurement tooling): stream s
 * = lead_s zero samples, then back-to-back frames, each exactly what modulateData() returns for
 * a payload of payload_len bytes drawn from splitmix64(seed, s, frame, i), the whole stream
 * scaled by amp_s in [amp_lo, amp_hi]; lead_s = splitmix64(seed,s) mod (lead_max+1).
 * d_out is [n_streams][pitch] float32 on the device.  Payload byte (s, frame, i) and lead_s/amp_s
 * are reproducible on the host with fskhip_synth_payload_byte / fskhip_synth_stream_params.
 */
int fskhip_synth_device(fskhip_engine *e, float *d_out, size_t n_per_stream, size_t pitch,
                        uint32_t payload_len, uint64_t seed, uint32_t lead_max, double amp_lo,
                        double amp_hi, void *hip_stream);
uint8_t fskhip_synth_payload_byte(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t i);
void fskhip_synth_stream_params(uint64_t seed, uint32_t stream, uint32_t lead_max, double amp_lo,
                                double amp_hi, uint32_t *lead, double *amp);
/* Adds white Gaussian noise in place: sigma_s^2 = mean_square(stream s over n_per_stream) /
 * 10^(snr_db/10) (the reference tests' definition, tests/modems/fsk-demodulation.node.test.ts:
 * 1184-1205), counter-based generator keyed by (seed, stream, sample). */
int fskhip_add_awgn_device(fskhip_engine *e, float *d_buf, size_t n_per_stream, size_t pitch,
                           double snr_db, uint64_t seed, void *hip_stream);

/* 1 if the demodulator kernels implement this engine's configuration (see fskhip_create). */
int fskhip_demod_supported(const fskhip_engine *e);

/*
 * Intermediate capture for parity tests: records, for ONE stream, every decimated sample's I/Q
 * magnitude sqrt(avgI^2+avgQ^2) (fsk.ts:252), post-filter output (fsk.ts:261) and slicer bit
 * (fsk.ts:264) of the following demodulate calls, up to `capacity` samples.  stream < 0
 * disables.  fskhip_trace_read copies out what was captured (n = count).
 */
int fskhip_trace_enable(fskhip_engine *e, int64_t stream, size_t capacity);
int fskhip_trace_read(fskhip_engine *e, double *amp, double *post, uint8_t *bit, size_t cap, size_t *n);
/* ... and the pre-filter's output (fsk.ts:202: the Float32Array the band-pass returns), one value per INPUT sample of the traced
 * stream, up to 2 x `capacity` of them (n = count), in the reference's scale. */
int fskhip_trace_read_pre(fskhip_engine *e, double *pre, size_t cap, size_t *n);

/* Measurement tooling: streams d_buf with the demodulator's fast-path read pattern and nothing else, so
 * a FETCH_SIZE counter pass over it can be calibrated against the known n_streams*n*4 bytes. */
int fskhip_probe_read_device(fskhip_engine *e, const float *d_buf, size_t n_per_stream, size_t pitch,
                             void *hip_stream);

/* FilterDesign.butterworth* (filters.ts:180-234): the configure-time designs the engine uses. */
void fskhip_butterworth_lowpass(double cutoff, double sampleRate, double b[3], double a[3]);
void fskhip_butterworth_highpass(double cutoff, double sampleRate, double b[3], double a[3]);
void fskhip_butterworth_bandpass(double center, double bandwidth, double sampleRate, double b[3], double a[3]);

/*
 * Signal-quality ESTIMATES.  The reference's getSignalQuality() (fsk.ts:471-479,
 * core.ts:280-288) is a stub that returns zeros for the five fields of SignalQuality (core.ts:10-16), and the host
 * classes' getSignalQuality() keeps returning those zeros.  These two calls are an opt-in EXTENSION with its own
 * definition (there is no reference behaviour to match; oracle/fsk_oracle.h restates it for the tests), built from values
 * the demodulator has anyway, in its rare paths only:
 *   at a sync (fsk.ts:315-326)            signalLevel = the mean of syncAmplitudeBuffer the silence threshold is taken from
 *   at the first 'eod' after a sync       noiseFloor = mean of the newest floor(samplesForEOD) amplitudes (the silence that
 *                                         caused it); frames += 1
 *   at every completed byte (fsk.ts:367)  the stop bit's vote (ones of count slicer samples, fsk.ts:335-336):
 *                                         eye += |2 ones - count| / count, minority += min(ones, count - ones), votes +=
 *                                         count; and, if the data ended ...1 0, sums of f and f^2, f = the post-filter
 *                                         output at the decision instant
 *   at every good start bit (fsk.ts:352)  if the byte before ended ...0: sum of f at its decision instant (the mirror
 *                                         image of the case above: one isolated bit, then the flip)
 *   snr             20 log10(signalLevel / noiseFloor) dB, capped at 200 (digital silence has a floor of 0)
 *   ber             minority / votes: how often a slicer sample disagrees with the bit it was voted into
 *   eyeOpening      mean vote margin, 0 .. 1
 *   phaseJitter     standard deviation of f over those stop-bit instants (rad per decimated sample)
 *   frequencyOffset minus the mid-point of the two mean f, as a frequency f * (sampleRate / 2) / (2 pi) Hz: the post
 *                   filter's lag acts on both mirrored transitions alike, a carrier offset shifts both (indicative: the
 *                   decision instants sit one sample after the bit edge, which leaves a bias of ~15 % of the deviation)
 * fskhip_enable_signal_quality(e, 1) clears the accumulators of every stream and starts them, (e, 0) stops them; while
 * they run, fp32 engines demodulate on their sample-granular kernel (like traced engines), so this is a diagnostic.
 */
typedef struct fskhip_signal_quality {
  double snr, ber, eyeOpening, phaseJitter, frequencyOffset;   /* SignalQuality core.ts:10-16 */
  double signalLevel, noiseFloor, frames, bytes;               /* what they are made of */
} fskhip_signal_quality;
int fskhip_enable_signal_quality(fskhip_engine *e, int on);
int fskhip_get_signal_quality(fskhip_engine *e, uint32_t stream, fskhip_signal_quality *q);

/* Page-locked host memory for the _host entry points (hipHostMalloc / hipHostFree). */
int fskhip_host_alloc(size_t bytes, void **ptr);
int fskhip_host_free(void *ptr);

/* Raw device memory helpers for hosts without a HIP binding of their own (ctypes / N-API). */
int fskhip_device_malloc(fskhip_engine *e, size_t bytes, void **d_ptr);
int fskhip_device_free(fskhip_engine *e, void *d_ptr);
int fskhip_memcpy_h2d(fskhip_engine *e, void *d_dst, const void *src, size_t bytes);
int fskhip_memcpy_d2h(fskhip_engine *e, void *dst, const void *d_src, size_t bytes);
int fskhip_synchronize(fskhip_engine *e);

/* Timing of the dominant kernel with HIP events on the stream the launches went to:
 * fskhip_timing_begin() arms it, every fskhip_demodulate_device() after that is bracketed by
 * an event pair, fskhip_timing_end() synchronises and returns launches and total kernel ms. */
int fskhip_timing_begin(fskhip_engine *e);
int fskhip_timing_end(fskhip_engine *e, uint32_t *n_launches, double *total_ms);

/* Tuning and test switches, by name; call after fskhip_create (and after fskhip_carry_over, if any) and before the engine's
 * first demodulate call: FSKHIP_E_INVALID once THIS engine has demodulated (a flag of its own calls: the call counters that
 * fskhip_carry_over copies do not count), for unknown names and for values that are not numbers or out of range.  None
 * changes a result: every choice computes the same bytes (tests/test_gpu_parity.py runs the goldens through each kernel).
 * The library reads no environment variable.
 *   "kernel"         auto | auto-r04 | auto-r02 | seven-wave | four-wave | two-wave | one-wave   which whole-tile fp32 kernel.
 *                    auto (default): seven waves per group (demod_blk6_kernel) for uniform configurations in batches of up
 *                    to 64 x compute units streams (every workgroup a compute unit to itself) on calls of at least
 *                    "stage_min_tiles" tiles, four waves (demod_blk_kernel / _r / _rp) otherwise wherever they apply;
 *                    auto-r04: never seven ("six-wave" was this kernel's name before it had seven)
 *   "stage_min_tiles" n             calls with fewer whole tiles stay off the seven-wave kernel (default 8: one 128-sample quantum)
 *   "stage_y_slots"  14 .. 24, even  depth of the seven-wave kernel's y ring in half tiles (default and upper limit: what the LDS of a compute
 *                                   unit allows at this dsSPB; lower limit: 2 + the half tiles its iq wave may lead the frame wave by).  Values
 *                                   outside what the kernel can use are FSKHIP_E_INVALID, as for blk_y_slots (they used to be clamped silently)
 *   "stage_roles"    auto | seven digits, a permutation of 0..6: the part each of a workgroup's seven waves plays (measurements)
 *   "exact_waves"    auto | 1 | 2   the fp64 kernel on one wave per 64-stream group (auto) or on two (loads + AGC + pre-filter | the rest:
 *                                   exact by construction -- the hand-over is the reference's own Float32Array, fsk.ts:202 -- and bit-identical, but
 *                                   slower at this register budget; kept for measurements)
 *   "force_generic"  0 | 1          never a whole-tile kernel: the sample-serial kernel only
 *   "blk_y_slots"    6 .. 28        depth of the four-wave kernel's first ring (checked against the LDS it needs)
 *   "blk_min_tiles"  n              calls with fewer whole tiles stay off the four-wave kernel
 *   "blk_resets"     auto | 0 | 1   which four-wave kernel: the one whose block path takes 'eod' resets itself (1: pays on idle receiver
 *                                   banks, costs ~4 % where resets are rare) or the one that hands such blocks to its per-sample path
 *                                   (0); auto (default): by the previous call's share of tiles off the fast path
 *   "blk_lanes"      auto|64|32|16|8 streams per workgroup of the four-wave kernel (auto: the widest that gives every workgroup its own CU)
 *   "blk_resident"   n >= 1         treat the device as holding n workgroups at once (time-sliced launches on small batches)
 *   "slice_tiles"    n >= 1 | off   tiles per time slice of a persistent launch
 *   "host_slab"      n              samples per time slab of fskhip_demodulate_host's pipeline (0 = no pipeline) */
int fskhip_set_option(fskhip_engine *e, const char *name, const char *value);

/* Diagnostics (tools/diag_*.py): one stream's per-stream state words as the kernels carry them between launches, in
 * fsk_params.h's RF_* / IF_* order (an implementation detail, not a contract); n_real / n_int return the field counts. */
int fskhip_debug_state(fskhip_engine *e, uint32_t stream, double *real_out, uint32_t real_cap, uint32_t *int_out,
                       uint32_t int_cap, uint32_t *n_real, uint32_t *n_int);

/* The shader clock the device holds under load (measurement aid, no reference counterpart): begin launches a one-wave
 * kernel on a stream of its own that stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around spin_ms of
 * sleeping; end waits for it and returns delta(s_memtime) / delta(s_memrealtime) x 0.1 GHz and the milliseconds it really
 * covered.  Start it first and launch the work to be observed behind it, on other streams -- NOT around a timed region: the
 * probe's wave takes a slot that one workgroup of a full-device launch then has to wait for. */
int fskhip_clock_probe_begin(fskhip_engine *e, double spin_ms);
int fskhip_clock_probe_end(fskhip_engine *e, double *shader_ghz, double *covered_ms);

/*
 * Hand-off waits (a property of the multi-wave kernels, not an entry point).  The two-, four- and seven-wave demodulator kernels,
 * the exact path's two-wave cut and the wide modulator pass tiles from wave to wave through LDS rings guarded by counters; a wave
 * that finds its input not there yet polls the counter with s_sleep in between.  The producer is a wave of the same workgroup
 * and is always running (every part is played exactly once: the waves settle their parts among themselves at the start, the
 * seven-wave kernel checks its part map in every wave), so a wait ends within a few tiles' work.  ABI 8: the waits are BOUNDED
 * all the same (csrc/fsk_wait.h) -- a wave that has polled 2^22 times in ONE wait without getting on (>= 0.15 s of shader clock,
 * four orders of magnitude beyond the longest wait of a healthy launch; polls are counted, not timed, so a preempted queue does
 * not trip it) sets the engine's hand-off fault word and ends, the waves waiting on it run into their own bound, workgroups of a
 * persistent launch that wait on a time slice of such a group give up as soon as they see the word, and the launch FINISHES.  The
 * host reports FSKHIP_E_HANDOFF from fskhip_synchronize, fskhip_get_faults, the _host calls (all of which wait for the device
 * anyway) and from every later fskhip_demodulate_device (from the statistics copy that trails the launches: no extra wait); it
 * is sticky -- the streams of that workgroup stopped mid-call -- and the engine is to be destroyed.  A lost counter update is
 * thus an error code, not a hung GPU; tools/handoff_check.py shows it on a build whose bound is one poll
 * (profiles/r06_handoff_bound.txt), tests/test_gpu_parity.py that healthy launches of every such kernel leave the word clear.
 * The bound sits on the waits' slow paths only: measured cost nil (same file).
 */
const char *fskhip_last_error(void);
int fskhip_abi_version(void);
int fskhip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* FSKHIP_H */
