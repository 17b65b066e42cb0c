// fsk_params.h -- launch parameters and persistent per-stream state layout shared by the host
// API (fsk_api.hip) and the kernels (fsk_demod.hip / fsk_mod.hip).
//
// State is struct-of-arrays in HBM: field-major [field][stream] so that a wave (64 consecutive
// streams) loads/stores each field with one coalesced 256-B (f32/u32) or 512-B (f64) access.
#pragma once
#include <stdint.h>

namespace fsk {

// ---- real-valued per-stream state (f32 or f64 depending on engine precision) -----------------
// Names follow the reference members they hold (src/modems/fsk.ts:87-128, src/dsp/filters.ts:11-12).
#define FSK_REAL_FIELDS(X)                                                                  \
  X(agc_gain)   /* AGCProcessor.currentGain fsk.ts:40 */                                    \
  X(bp_x1) X(bp_x2) X(bp_y1) X(bp_y2)   /* preFilter history filters.ts:11-12 */            \
  X(li_x1) X(li_x2) X(li_y1) X(li_y2)   /* iqFilters.i */                                   \
  X(lq_x1) X(lq_x2) X(lq_y1) X(lq_y2)   /* iqFilters.q */                                   \
  X(po_x1) X(po_x2) X(po_y1) X(po_y2)   /* postFilter */                                    \
  X(acc_i) X(acc_q)                     /* downsample.{i,q}Accumulator fsk.ts:105-109 */    \
  X(last_phase)                         /* iqState.lastPhase fsk.ts:102 */                  \
  X(nco_phase)                          /* iqState.localOscPhase (f64 path only) */         \
  X(nco_c) X(nco_s)                     /* f64 path: cos / sin of nco_phase as the rotated phasor stands (round 6: carried across */ \
                                        /* calls, re-evaluated where the stream's ABSOLUTE sample count is a multiple of 32) */ \
  X(sil_thr)                            /* silence.threshold fsk.ts:128 */
// fp32 engines keep the I/Q low-pass in a FREE-RUNNING frame (fsk_pipe.hip): li_*, lq_*, last_phase are that frame's.
// After a resetState() the next kDirectPairs decimated samples come from a zero-started direct instance zd_* (zr_dph =
// how many it has produced); kZeroLagPairs decimated samples after the reset the free-running filters are zeroed, and
// what they then lack -- the direct instance's memory at that point -- is carried as a zero-input response (pair sums
// zq_*) from the direct instance's last two samples on.  (Not part of the generic kernel's Lane.)
#define FSK_REAL_FIELDS_PIPE(X)                                                             \
  X(zq_ai) X(zq_aq) X(zq_bi) X(zq_bq) X(zq_0i) X(zq_0q)                                     \
  X(zd_ix1) X(zd_ix2) X(zd_iy) X(zd_iv) X(zd_qx1) X(zd_qx2) X(zd_qy) X(zd_qv)

// ---- integer per-stream state ----------------------------------------------------------------
#define FSK_INT_FIELDS(X)                                                                   \
  X(nco_lo) X(nco_hi)   /* f32 path: NCO phase as a 64-bit fraction of a turn */            \
  X(ds_cnt)             /* downsample.counter fsk.ts:106 */                                 \
  X(gsc)                /* bitSync.globalSampleCounter fsk.ts:113 */                        \
  X(cad_ctr)            /* gsc % round(dsSPB/4), kept incrementally (fsk.ts:302) */         \
  X(sil_cnt)            /* silence.sampleCount */                                           \
  X(started)            /* frame.started */                                                 \
  X(bit_acc)            /* bitSync.bitAccumulator fsk.ts:113 */                             \
  X(bit_wait)           /* int32: nextBitSampleIndex - bitSampleCounter (fsk.ts:335); */    \
                        /* kBigWait while no frame is started */                            \
  X(bit_reload)         /* bit_wait right after the last bit decision: bitAccumCount at */  \
                        /* the next decision = bit_reload - bit_wait (fsk.ts:336) */        \
  X(byte_cur) X(bit_pos)                /* byteState fsk.ts:125 */                          \
  X(ring_len)           /* syncSamplesBuffer.length (utils.ts:10) */                        \
  X(poly_phase)         /* pushes into the sync ring mod dsSPB (polyphase register index) */\
  X(matched)            /* running value of the fsk.ts:304-312 match count */               \
  X(amp_pos) X(amp_len) /* syncAmplitudeBuffer write index / length */                      \
  X(sync_det)           /* debug.syncDetections */                                          \
  X(eod_total)          /* 'eod' events since create */
// Opt-in signal-quality estimates (fskhip_enable_signal_quality; include/fskhip.h has the definition): touched by the
// rare paths only (sync, first 'eod' after it, start / stop bit positions), read-modify-written in HBM.
#define FSK_REAL_FIELDS_QUALITY(X)                                                          \
  X(q_signal) X(q_floor) X(q_f_sum) X(q_f2_sum) X(q_f0_sum) X(q_eye_sum)
#define FSK_INT_FIELDS_QUALITY(X)                                                           \
  X(q_armed) X(q_prev_d0) X(q_frames) X(q_bytes) X(q_minor) X(q_votes) X(q_starts) X(q_ftrans)
#define FSK_INT_FIELDS_PIPE(X)                                                              \
  X(fr_lo) X(fr_hi)     /* fp32: NCO phase minus the free-running frame's phase (64-bit turns); */ \
                        /* changes only at resetState()                                     */      \
  X(zr_dph)             /* fp32: decimated samples since resetState(), saturating at kHandPairs; < kDirectPairs: the direct instance runs, */ \
                        /* from kDirectPairs on zq_* is valid, from kHandPairs on it may retire */

enum RealField {
#define X(n) RF_##n,
  FSK_REAL_FIELDS(X)
  FSK_REAL_FIELDS_PIPE(X)
  FSK_REAL_FIELDS_QUALITY(X)
#undef X
  RF_COUNT
};
enum IntField {
#define X(n) IF_##n,
  FSK_INT_FIELDS(X)
  FSK_INT_FIELDS_PIPE(X)
  FSK_INT_FIELDS_QUALITY(X)
#undef X
  IF_COUNT
};

// per-stream configure-time constants, [field][stream] doubles (converted in-kernel for f32)
enum CoefField {
  CF_bp_b0,   // butterworthBandpass b[0]   (b[1] = 0, b[2] = -b[0], filters.ts:230)
  CF_bp_a1,
  CF_bp_a2,
  CF_omega,   // 2*pi*centerFreq/sampleRate (fsk.ts:228)
  CF_mark_w,  // 2*pi*markFrequency/sampleRate  (modulator, fsk.ts:404)
  CF_space_w, // 2*pi*spaceFrequency/sampleRate
  CF_w1_re, CF_w1_im,  // e^{j*k*omega}, k = 1..3: NCO phasors of samples 1..3 of a 4-sample block
  CF_w2_re, CF_w2_im,  // relative to sample 0 (fast fp32 kernel)
  CF_w3_re, CF_w3_im,
  CF_COUNT
};

struct DemodParams {
  uint32_t n_streams;
  uint32_t d;             // downsampledSamplesPerBit (fsk.ts:442)
  uint32_t cadence;       // Math.round(dsSPB/4) (fsk.ts:299); 0 = the % never hits
  uint32_t n_bits;        // preambleSfdBits.length
  uint32_t sample_count;  // n_bits * d (fsk.ts:298)
  uint32_t ring_cap;      // syncSamplesBuffer capacity (integer, checked at create)
  uint32_t amp_cap;       // syncAmplitudeBuffer capacity = 8*d (fsk.ts:150)
  uint32_t matched_min;   // smallest integer `matched` with matched/total > syncThreshold in f64
  uint32_t eod_min;       // ceil(silence.samplesForEOD) (fsk.ts:148, 288)
  uint64_t pat_q;         // bit j (1..n_bits-1) = preambleSfdBits[n_bits - j] (fsk.ts:307)
  uint64_t pat_mask;      // bits 1..n_bits-1
  uint32_t wide;          // 1: polyphase registers are 64-bit (n_bits > 31, or frac)
  uint32_t frac;          // 1: the reference's ring capacity is fractional (see fsk_demod.hip)
  uint32_t ring_int;      // frac: typed-array length A = floor(capacity); pushes >= A store `undefined`
  uint32_t stop_pos;      // 9 or 10 (fsk.ts:348)
  uint32_t parity_on;
  uint32_t agc_on;
  double lp_b0, lp_b1, lp_b2, lp_a1, lp_a2;  // butterworthLowpass(baud, sr) (fsk.ts:458-461)
  double agc_attack, agc_release;            // fsk.ts:48-49
  // the same constants rounded once on the host for the fp32 kernels (a device-side (float) of the
  // f64 fields gets re-materialised by hipcc with a quarter-rate v_cvt_f32_f64 at every use)
  float f_lp_b0, f_lp_b0h, f_lp_a2, f_lp_delta, f_agc_att, f_agc_rel;
  // uni_cfg = 1: every stream shares one configuration, so what is otherwise a per-stream array entry is a
  // wave-uniform constant (SGPRs instead of VGPRs in the fast kernel)
  uint32_t nco_anchor;    // fp64 generic kernel: samples the engine had taken before this call, mod 32 (where the NCO phasor is re-evaluated)
  uint32_t uni_cfg;
  float u_bp_b0h;                  // pre-filter b0 with the low-pass gain b0/2 folded in
  float u_bp_na1, u_bp_na2;        // -a1, -a2
  float u_bp_c1y, u_bp_c2y;        // a1*a1 - a2, a1*a2: second row of the two-sample look-ahead form
  float u_w1_re, u_w1_im;          // e^{j omega}
  float u_w2_re, u_w2_im;          // e^{2 j omega}
  uint32_t u_inc2_lo, u_inc2_hi;   // 2 NCO steps (turns * 2^64)
  uint32_t u_inc16_lo, u_inc16_hi; // 16 NCO steps = one tile
  // fsk_pipe.hip (free-running front, ZIR-corrected back)
  uint32_t u_inc_lo, u_inc_hi;     // one NCO step (turns * 2^64), uniform configuration
  float z_c1, z_c2;                // zero-input response of the I/Q low-pass as pair sums: q[m+2] = c1 q[m+1] - c2 q[m]
  float z_ya, z_yb, z_va, z_vb;    // (y, v) of that response at an even sample from the next two pair sums (q[m], q[m+1])
  // opt-in signal-quality estimates
  uint32_t quality;                // 1: accumulate them
  uint32_t q_last_d0;              // lowest bit of the last pattern byte (the byte in front of a frame's first start bit)
  uint32_t q_eod_n;                // floor(silence.samplesForEOD): amplitudes averaged for the noise floor
};

struct DemodState {
  void *rs;           // Real   [RF_COUNT][n_streams]
  uint32_t *is;       // u32    [IF_COUNT][n_streams]
  void *poly;         // u32|u64 [n_blocks][d][64]  polyphase sync-bit registers
  void *poly_u;       // u64     [n_blocks][d][64]  frac only: 1 = that tap holds `undefined`
  float *amp_ring;    // f32    [amp_cap][n_streams] syncAmplitudeBuffer storage
  const double *coef; // double [CF_COUNT][n_streams]
  const uint64_t *nco_inc; // u64 [n_streams]: round(centerFreq/sampleRate * 2^64)
  // optional intermediate capture of ONE stream (parity tests: fsk.ts:252 amplitude, :264 bit)
  double *trace_amp;
  double *trace_post;     // post-filter output (fsk.ts:261)
  uint8_t *trace_bit;
  uint32_t *trace_n;      // running count of decimated samples captured
  uint32_t trace_cap;
  uint32_t trace_stream;  // 0xFFFFFFFF = off
  uint32_t *cu_ctr;       // u32 [2048]: workgroups started per compute unit (fsk_blk.hip spreads its waves' roles with it)
  float *blk_stash;       // f32 [7][n_streams][4] or null: fsk_blk.hip's block path with resets parks a lane's entry state here
                          // (written per such block, read back only when the block has to be redone sample by sample)
  uint32_t *blk_stat;     // u32 [4]: tiles fsk_blk.hip's back waves have processed, how many of them left the fast block loop, the hand-off fault word (fsk_wait.h: bit 0 a wait ran into its bound, bit 1 a bad part map), unused
                          // (running totals; the host picks the next call's kernel by their increments)
  uint32_t *blk_q;        // u32 [16 + groups * 127] or null: fsk_blk.hip's (group, time slice) queue for batches beyond one round
};

struct ModParams {
  uint32_t n_streams;
  uint32_t spb;           // samplesPerBit (fsk.ts:438)
  uint32_t bits_per_byte; // fsk.ts:439
  uint32_t start_bits, stop_bits, parity; // parity 0/1/2
  uint32_t n_pre;         // preamble + sfd bytes
  uint8_t pre[2 * 16];
  uint32_t *stat;         // the engine's DemodState::blk_stat (the hand-off fault word of modulate_wide_kernel's waits)
  uint32_t exact_sin;     // 1 (fp64 engines): Math.sin by V8's own operation sequence (fsk_fdlibm.h), bit-identical signal;
                          // 0 (fp32 engines): the device library's sin(), ~1.8x faster, may differ by one f32 ulp on ~1e-9 of samples
};

// FSKProcessor + ChunkedModulator per stream (fsk-processor.ts, chunked-modulator.ts), device resident.
// The pending signal is kept as the modulator's generator state (payload + phase + position), not as samples:
// a slice of n samples is produced on demand and is bit-identical to the same slice of modulateData()'s output
// because the reference's signal is itself one sequential f64 phase accumulation (fsk.ts:398-406).
struct ProcState {
  uint8_t *rx_buf;       // [stream][rx_cap] demodulatedBuffer storage (fsk-processor.ts:84)
  uint32_t *rx_w, *rx_r, *rx_len;  // writeIndex / readIndex / _length (utils.ts:7-9)
  uint32_t rx_cap;
  uint8_t *tx_payload;   // [stream][tx_payload_pitch] bytes of the pending modulation
  size_t tx_payload_pitch;
  double *tx_phase;      // FrameGen state at samplePosition
  uint32_t *tx_pos;      // samplePosition (chunked-modulator.ts:25)
  uint32_t *tx_len;      // pendingSignal.length, 0 = no signal
  uint32_t *tx_in_bit, *tx_bit_idx, *tx_cur_bit, *tx_n_payload;
  uint32_t *tx_pending;  // pendingModulation != null (fsk-processor.ts:64)
  uint32_t *tx_completed;
};

// fp32 free-running frame (fsk_pipe.hip): the front end zeroes a stream's I/Q low-pass kZeroLagPairs decimated samples
// after a resetState() -- far enough for the wave that owns the low-pass to learn of the reset in time, however far it runs
// ahead of the wave with the frame logic (at most six half tiles = 24 decimated samples in fsk_blk.hip, four half tiles in
// fsk_pipe.hip, kZeroLagPairs / 8 tiles in fsk_blk6.hip) -- and a direct instance covers those plus two more.
// Round 5: 24 -> 48.  The small-batch kernel (fsk_blk6.hip) has four stages on the ring this lag bounds; at 24 they shared
// three tiles and ran in turn rather than side by side.  What it costs the others is the longer span the frame logic works
// itself after a reset (74 decimated samples in the four-wave kernels -- kOwnPairs4 below --, 98 in the seven-wave kernel,
// instead of 50): config #3 +- 0, config #2 at 65 536 streams - 2 %, a batch whose streams' frames do not line up (a reset in
// some lane every few tiles) - 6 % (profiles/r05_lag.txt).  One value for every kernel: the fp32 results after a reset depend
// on it, and the kernels must stay interchangeable call by call.
#ifndef FSK_ZLAG
#define FSK_ZLAG 48
#endif
static constexpr uint32_t kZeroLagPairs = FSK_ZLAG;
static constexpr uint32_t kDirectPairs = kZeroLagPairs + 2;
// The zero-input response left by a reset is carried un-retired for another kHandLag decimated samples (zr_dph counts on
// to kHandPairs).  That fixed span is what lets the multi-wave kernels move the correction from the wave with the frame logic
// to the discriminator wave running up to kHandLag - 1 decimated samples AHEAD of it (23 in fsk_blk.hip, kHandLag / 8 tiles
// in fsk_blk6.hip): the values at the hand-over sample follow from the two start values by the recurrence alone, so they can
// be posted kHandLag samples early -- and the discriminator wave, whatever its lead, has not reached the hand-over sample
// when they are posted.
#ifndef FSK_HLAG
#define FSK_HLAG 48
#endif
static constexpr uint32_t kHandLag = FSK_HLAG;
static constexpr uint32_t kHandPairs = kDirectPairs + kHandLag;
// What every kernel shares is the span the correction stays un-retired (zr_dph < kHandPairs) -- not who applies it.  The
// four-wave kernel's discriminator wave is never more than 23 decimated samples ahead, so its back wave lets go of the
// correction after kOwnLag4 = 24 samples as in round 4 (its own span: kOwnPairs4 = 74 decimated samples, not 98) and the
// discriminator wave applies it WITHOUT the retirement test for the rest of the span; the seven-wave kernel's frame wave
// keeps it for all of kHandLag.  Sample for sample the same operations on the same values either way.
static constexpr uint32_t kOwnLag4 = 24;
static constexpr uint32_t kOwnPairs4 = kDirectPairs + kOwnLag4;
static_assert(kOwnLag4 <= kHandLag, "the back wave cannot keep the correction longer than it stays un-retired");
static constexpr uint32_t kBigWait = 0x40000000u;  // bit_wait while !started (12 h of decimated samples)
#ifndef FSK_TILE
#define FSK_TILE 32
#endif
static constexpr int kTile = FSK_TILE;      // samples per stream per LDS tile (FSK_TILE*4 B per row)
static constexpr int kChunks = kTile / 4;   // 16-B chunks per row per tile (8, 16 or 32)
static constexpr int kRowsPerLoad = 64 / kChunks;  // rows one wave-wide 16-B/lane load covers
static constexpr int kSlotStride = 65;      // 16-B slots per chunk column (64 lanes + 1 pad)

}  // namespace fsk
