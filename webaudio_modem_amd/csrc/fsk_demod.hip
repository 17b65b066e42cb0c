// fsk_demod.hip -- fused FSK demodulator kernel for gfx950 (MI355X).
//
// One lane per stream, one wave (64 streams) per workgroup.  For every input sample the lane
// runs the whole reference chain of FSKCore.demodulateData (src/modems/fsk.ts:190-375):
//   AGC (52-76) -> band-pass biquad (filters.ts:47-76) -> centre-frequency NCO I/Q mix (228-232)
//   -> two low-pass biquads (235-238) -> /2 boxcar (241-248) -> atan2 / magnitude (251-252)
//   -> wrapped phase difference (255-258) -> post low-pass (261) -> slicer (264)
//   -> silence/EOD (285-295) -> preamble+SFD correlation (297-328) -> majority-vote bit clock
//   (330-341) -> UART byte framing (346-375)
// with no intermediate ever leaving the CU.  HBM traffic is the 4 B/sample input stream, the
// f32 amplitude ring (the reference's syncAmplitudeBuffer, needed verbatim for the silence
// threshold at sync time: 2 B/sample of stores) and the per-stream state once per launch.
//
// Input rows are [stream][sample]; a wave loads a 64-row x 32-sample tile with coalesced
// 16-B/lane loads (8 lanes cover one 128-B row segment), parks it in LDS chunk-major with a
// one-slot pad so both the ds_write_b128 (8 lanes x 4 banks) and the per-lane ds_read_b128
// are bank-conflict free, and prefetches the next tile into registers while it computes.
//
// The kernel is VALU-bound, not HBM-bound, so the per-decimated-sample state
// machine is written as straight-line code plus three wave-uniform rare paths:
//  * sync correlator: NOT the reference's O(nBits*dsSPB) brute force.  The decision-bit history
//    is kept as dsSPB polyphase shift registers in LDS (register p holds the bits pushed at
//    times == p mod dsSPB, newest in bit 0), so the window slots' entering/leaving bits at each
//    step are two masked popcounts of ONE register; `matched` is carried incrementally and is at
//    every step exactly the count the reference's double loop would produce;
//  * bit clock: nextBitSampleIndex - bitSampleCounter is one down-counter (parked at kBigWait
//    while no frame is started) and the vote accumulator runs ungated -- every path that
//    (re)starts a frame zeroes it, like the reference does;
//  * ring positions that are equal for all streams of a launch (push slot, amplitude-ring slot,
//    pushes so far) live in SGPRs in the UNI kernels.
//
// Fractional ring capacities (FRAC kernels).  The reference sizes its sync ring
// maxSyncBits*dsSPB*1.1 (fsk.ts:149), which is often NOT an integer in f64 (44.1 kHz; 48 kHz with
// parity, two stop bits or a longer preamble: 65*20*1.1 = 1430.0000000000002).  Its RingBuffer
// (utils.ts:28-48) then works for the first A = floor(capacity) pushes after configure()/clear()
// and degenerates: the write index turns fractional, every later store is dropped and every
// later slot reads `undefined`, which equals nothing but the equally undefined
// preambleSfdBits[length] of window slot 0 (fsk.ts:306-307).  That is reproduced exactly: once a
// stream has pushed A samples, pushes mark their tap UNDEFINED in a second register set; slots
// 1.. count only defined matching taps and slot 0 counts undefined ones.  (The host verifies at
// create time that the fractional index sequence cannot become integral again within 2^40
// pushes; otherwise the configuration is refused.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fsk_params.h"
#include "fsk_dev.h"

#ifndef FSK_FAST_WAVES
#define FSK_FAST_WAVES 4
#endif
#ifndef FSK_F32_WAVES_PER_SIMD
#define FSK_F32_WAVES_PER_SIMD 2
#endif

namespace fsk {

// ------------------------------------------------------------------------------------------------
// per-precision arithmetic
// ------------------------------------------------------------------------------------------------
template <typename Real>
struct Consts;

template <>
struct Consts<float> {
  float lp_b0, lp_b0h, lp_a2, lp_delta, agc_att, agc_rel;
  float bp_b0, bp_a1, bp_a2;
  uint32_t inc_lo, inc_hi;
  __device__ void init(const DemodParams &P, const DemodState &S, uint32_t row) {
    // host-rounded constants (DemodParams::f_*).  delta = 1 + a1 + a2 (= b0+b1+b2 for the unity-DC-
    // gain Butterworth) is formed in f64 and only then rounded: rounding a1 ~ -1.94 itself to f32
    // would move the DC gain by ~1e-4 at 300 baud.  The I/Q low-passes run with b0/2: a power-of-two
    // scaling is exact in binary floating point, so their outputs are bit for bit half the
    // reference's and the /2 boxcar sum IS the average.
    lp_b0 = P.f_lp_b0; lp_b0h = P.f_lp_b0h; lp_a2 = P.f_lp_a2; lp_delta = P.f_lp_delta;
    agc_att = P.f_agc_att; agc_rel = P.f_agc_rel;
    // ... and that b0/2 is folded into the band-pass gain below (the pre-filter is linear, so its whole
    // output -- and its y history -- simply carries the factor), which removes one multiply per
    // sample from the I/Q low-passes
    size_t n = P.n_streams;
    bp_b0 = (float)(S.coef[(size_t)CF_bp_b0 * n + row] * (0.5 * P.lp_b0));
    bp_a1 = (float)S.coef[(size_t)CF_bp_a1 * n + row];
    bp_a2 = (float)S.coef[(size_t)CF_bp_a2 * n + row];
    uint64_t inc = S.nco_inc[row];
    inc_lo = (uint32_t)inc; inc_hi = (uint32_t)(inc >> 32);
  }
};

template <>
struct Consts<double> {
  double lp_b0, lp_b1, lp_b2, lp_a1, lp_a2, agc_att, agc_rel;
  double bp_b0, bp_a1, bp_a2, omega;
  __device__ void init(const DemodParams &P, const DemodState &S, uint32_t row) {
    lp_b0 = P.lp_b0; lp_b1 = P.lp_b1; lp_b2 = P.lp_b2; lp_a1 = P.lp_a1; lp_a2 = P.lp_a2;
    agc_att = P.agc_attack; agc_rel = P.agc_release;
    size_t n = P.n_streams;
    bp_b0 = S.coef[(size_t)CF_bp_b0 * n + row];
    bp_a1 = S.coef[(size_t)CF_bp_a1 * n + row];
    bp_a2 = S.coef[(size_t)CF_bp_a2 * n + row];
    omega = S.coef[(size_t)CF_omega * n + row];
  }
};

template <typename Real>
struct Lane {
#define X(n) Real n;
  FSK_REAL_FIELDS(X)
#undef X
#define X(n) uint32_t n;
  FSK_INT_FIELDS(X)
#undef X
  uint32_t thr_eff;  // matched_min while searching, 0xFFFFFFFF while a frame is started (not stored)
};

template <typename Real>
__device__ inline void load_lane(Lane<Real> &L, const DemodState &S, size_t n, uint32_t row) {
  const Real *rs = (const Real *)S.rs;
#define X(f) L.f = rs[(size_t)RF_##f * n + row];
  FSK_REAL_FIELDS(X)
#undef X
#define X(f) L.f = S.is[(size_t)IF_##f * n + row];
  FSK_INT_FIELDS(X)
#undef X
}
template <typename Real>
__device__ inline void store_lane(const Lane<Real> &L, const DemodState &S, size_t n, uint32_t row) {
  Real *rs = (Real *)S.rs;
#define X(f) rs[(size_t)RF_##f * n + row] = L.f;
  FSK_REAL_FIELDS(X)
#undef X
#define X(f) S.is[(size_t)IF_##f * n + row] = L.f;
  FSK_INT_FIELDS(X)
#undef X
}

// ---- fp64: op-for-op with the reference's double arithmetic (this TU is built with
// -ffp-contract=off, so every * and + below rounds separately, like JavaScript) ----------------

// IIRFilter.process (filters.ts:47-76), order 2: sum starts at 0 and runs b0,b1,b2,-a1,-a2.
__device__ inline double biquad64(double b0, double b1, double b2, double a1, double a2,
                                  double &x1, double &x2, double &y1, double &y2, double x) {
  double out = 0.0;
  out += b0 * x;
  out += b1 * x1;
  out += b2 * x2;
  out -= a1 * y1;
  out -= a2 * y2;
  x2 = x1; x1 = x;
  y2 = y1; y1 = out;
  return out;
}

// AGC + pre-filter for one input sample (fsk.ts:52-76, 202): returns the pre-filter's Float32Array
// output.  Neither stage is touched by resetState().
__device__ inline float pre_stage(Lane<double> &L, const Consts<double> &C, bool agc_on, float xin, float &agc_out) {
  float xs = xin;
  if (agc_on) {
    xs = (float)((double)xin * L.agc_gain);  // samples[i] *= gain : Float32Array store
    double level = fabs((double)xs);
    if (level > 0.5) {
      double target = 0.5 / level;
      L.agc_gain += (target - L.agc_gain) * C.agc_att;
    } else if (level > 0.0) {
      double target = 0.5 / level;
      L.agc_gain += (target - L.agc_gain) * C.agc_rel;
    }
    double g = L.agc_gain < 10.0 ? L.agc_gain : 10.0;
    L.agc_gain = g > 0.1 ? g : 0.1;
  }
  agc_out = xs;
  // preFilter.processBuffer: f64 state, f32 result (filters.ts:81-87)
  return (float)biquad64(C.bp_b0, 0.0, -C.bp_b0, C.bp_a1, C.bp_a2, L.bp_x1, L.bp_x2, L.bp_y1, L.bp_y2, (double)xs);
}

// NCO mix + I/Q low-pass for one pre-filtered sample (fsk.ts:228-238): the part resetState() zeroes.
__device__ inline void mix_lp(Lane<double> &L, const Consts<double> &C, float pre, double &fi, double &fq) {
  double s = (double)pre;
  double ci = s * cos(L.nco_phase);
  double cq = s * sin(L.nco_phase);
  L.nco_phase = fmod(L.nco_phase + C.omega, 2.0 * 3.14159265358979323846);
  fi = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.li_x1, L.li_x2, L.li_y1, L.li_y2, ci);
  fq = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.lq_x1, L.lq_x2, L.lq_y1, L.lq_y2, cq);
}

// decimated-rate discriminator (fsk.ts:245-264): returns the slicer bit
__device__ inline bool discriminate(Lane<double> &L, const Consts<double> &C, double sum_i, double sum_q,
                                    double &amp, double &post) {
  const double PI = 3.14159265358979323846;
  double avg_i = sum_i / 2.0;
  double avg_q = sum_q / 2.0;
  double phase = atan2(avg_q, avg_i);
  amp = sqrt(avg_i * avg_i + avg_q * avg_q);
  double dphi = phase - L.last_phase;
  if (dphi > PI) dphi -= 2.0 * PI;
  else if (dphi < -PI) dphi += 2.0 * PI;
  L.last_phase = phase;
  double f = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.po_x1, L.po_x2, L.po_y1, L.po_y2, dphi);
  post = f;
  return f > 0.0;
}

__device__ inline void nco_reset(Lane<double> &L) { L.nco_phase = 0.0; }

// ---- fp32: throughput path ---------------------------------------------------------------------
// Same chain, leaner forms: b1 = 0 / b2 = -b0 (band-pass) and b1 = 2*b0, b2 = b0 (low-pass) are
// folded, FMAs are explicit, 1/x is v_rcp_f32, sqrt is v_sqrt_f32, atan2 is a degree-15 odd
// polynomial, the NCO is a 64-bit turn accumulator feeding v_sin_f32 / v_cos_f32 (which take
// revolutions).

// (lp32 and atan2_amp_fast: fsk_dev.h)

__device__ inline float pre_stage(Lane<float> &L, const Consts<float> &C, bool agc_on, float xin, float &agc_out) {
  // AGC (fsk.ts:52-76), always executed: with AGC disabled the host sets both rates to 0 and the
  // gain to 1, which makes this block an exact no-op (x*1, g + t*0 = g)
  (void)agc_on;
  const float xs = xin * L.agc_gain;
  {
    const float level = __builtin_fabsf(xs);
    const float t = __builtin_fmaf(0.5f, __builtin_amdgcn_rcpf(level), -L.agc_gain);  // target - gain
    const float rate = level > 0.5f ? C.agc_att : C.agc_rel;
    float g = __builtin_fmaf(t, rate, L.agc_gain);
    g = level > 0.0f ? g : L.agc_gain;  // exact zero holds the gain (fsk.ts:67)
    L.agc_gain = __builtin_amdgcn_fmed3f(g, 0.1f, 10.0f);
  }
  agc_out = xs;
  // band-pass: y = b0*(x - x2) - a2*y2 - a1*y1 (the y1 term last: shortest recurrence)
  float y = C.bp_b0 * (xs - L.bp_x2);
  y = __builtin_fmaf(-C.bp_a2, L.bp_y2, y);
  y = __builtin_fmaf(-C.bp_a1, L.bp_y1, y);
  L.bp_x2 = L.bp_x1; L.bp_x1 = xs;
  L.bp_y2 = L.bp_y1; L.bp_y1 = y;
  return y;
}

__device__ inline void mix_lp(Lane<float> &L, const Consts<float> &C, float y, float &fi, float &fq) {
  // NCO: phase in turns = top 32 bits of the accumulator
  float turns = (float)L.nco_hi * 2.3283064365386963e-10f;  // 2^-32
  float c = __builtin_amdgcn_cosf(turns);
  float s = __builtin_amdgcn_sinf(turns);
  uint32_t lo = L.nco_lo + C.inc_lo;
  L.nco_hi = L.nco_hi + C.inc_hi + (lo < L.nco_lo ? 1u : 0u);
  L.nco_lo = lo;
  fi = lp32<false>(0.0f, C.lp_a2, C.lp_delta, L.li_x1, L.li_x2, L.li_y1, L.li_y2, y * c);
  fq = lp32<false>(0.0f, C.lp_a2, C.lp_delta, L.lq_x1, L.lq_x2, L.lq_y1, L.lq_y2, y * s);
}

__device__ inline bool discriminate(Lane<float> &L, const Consts<float> &C, float sum_i, float sum_q, float &amp,
                                    float &post) {
  const float PI = 3.14159265358979323846f;
  float avg_i = sum_i;  // sums of the half-scale low-pass outputs = the averages (see lp_b0h)
  float avg_q = sum_q;
  float phase = atan2_amp_fast(avg_q, avg_i, amp);
  float dphi = phase - L.last_phase;
  float wrap = dphi > PI ? -2.0f * PI : 0.0f;
  wrap = dphi < -PI ? 2.0f * PI : wrap;
  dphi += wrap;
  L.last_phase = phase;
  float f = lp32(C.lp_b0, C.lp_a2, C.lp_delta, L.po_x1, L.po_x2, L.po_y1, L.po_y2, dphi);
  post = f;
  return f > 0.0f;
}

__device__ inline void nco_reset(Lane<float> &L) { L.nco_lo = 0; L.nco_hi = 0; }

// ------------------------------------------------------------------------------------------------
// frame state machine (precision independent except for the amplitude compare)
// ------------------------------------------------------------------------------------------------

// resetState() fsk.ts:175-188.  Not touched: AGC gain, pre-filter, both rings (and therefore
// `matched`, the polyphase registers, ring positions), silence threshold.
template <typename Real>
__device__ inline void reset_state(Lane<Real> &L, uint32_t matched_min) {
  nco_reset(L);
  L.last_phase = (Real)0;
  L.gsc = 0; L.cad_ctr = 0;
  L.bit_acc = 0; L.bit_wait = kBigWait; L.bit_reload = 0;
  L.byte_cur = 0; L.bit_pos = 0;
  L.started = 0; L.thr_eff = matched_min;
  L.sil_cnt = 0;
  L.li_x1 = L.li_x2 = L.li_y1 = L.li_y2 = (Real)0;
  L.lq_x1 = L.lq_x2 = L.lq_y1 = L.lq_y2 = (Real)0;
  L.po_x1 = L.po_x2 = L.po_y1 = L.po_y2 = (Real)0;
  L.acc_i = (Real)0; L.acc_q = (Real)0;
  L.ds_cnt = 0;
}

struct OutCtx {
  uint8_t *out_row;    // this stream's byte slab
  uint32_t out_pitch;
  uint32_t out_cnt;    // bytes produced this call
  uint32_t eod_cnt;    // eod events this call
};

// Ring bookkeeping that is identical for all lanes of a UNI launch (kept in SGPRs there) and
// per-lane otherwise.
struct RingPos {
  uint32_t phase;    // push slot = pushes mod dsSPB
  uint32_t amp_pos;  // syncAmplitudeBuffer write index
  uint32_t k;        // pushes made so far in this launch
};

// processDownsampledBit (fsk.ts:278-344) for every lane with act set.  Must be called by the
// whole wave (rare paths are wave-uniform branches; the sync path reads the amplitude ring
// cooperatively).
template <typename Real, typename PolyT, bool FRAC>
__device__ inline bool downsampled_bit(Lane<Real> &L, const DemodParams &P, const DemodState &S, PolyT *poly,
                                       PolyT *poly_u, RingPos &R, uint32_t need, uint32_t ring_base,
                                       uint32_t amp_base, uint32_t lane, uint32_t row, bool valid, bool act, bool bitb,
                                       Real amp, OutCtx &O) {
  const PolyT qn = (PolyT)~P.pat_q, mask = (PolyT)P.pat_mask;
  const PolyT qn2 = (PolyT)(qn << 1), mask2 = (PolyT)(mask << 1);
  const uint32_t bit = bitb ? 1u : 0u;
  bool eod = false, cand = false, decide = false, did_reset = false;
  if (act) {
    // ---- syncSamplesBuffer.put(bit): polyphase register of this push slot, newest bit in bit 0
    const uint32_t idx = R.phase * 64u + lane;
    PolyT r, u = 0;
    if (FRAC) {
      const bool undef = ring_base + R.k >= P.ring_int;  // this store is dropped by the reference's ring
      u = (PolyT)(poly_u[idx] << 1) | (PolyT)(undef ? 1u : 0u);
      poly_u[idx] = u;
      r = (PolyT)(poly[idx] << 1) | (PolyT)(undef ? 0u : bit);
    } else {
      r = (PolyT)(poly[idx] << 1) | (PolyT)bit;
    }
    poly[idx] = r;
    // slots j = 1..n_bits-1 gain tap j and lose tap j+1: [tap_j == q_j] - [tap_{j+1} == q_j]
    L.matched += popc((PolyT)((r ^ qn) & ~u & mask));
    L.matched -= popc((PolyT)((r ^ qn2) & ~u & mask2));
    if (FRAC) {  // slot 0 compares against `undefined`: it counts undefined taps
      L.matched += (uint32_t)(u & 1u);
      L.matched -= (uint32_t)((u >> 1) & 1u);
    }
    // ---- syncAmplitudeBuffer.put(amp): Float32Array store
    if (valid) S.amp_ring[(size_t)R.amp_pos * P.n_streams + row] = (float)amp;
    R.phase = (R.phase + 1 == P.d) ? 0u : R.phase + 1;
    R.amp_pos = (R.amp_pos + 1 == P.amp_cap) ? 0u : R.amp_pos + 1;
    R.k++;

    L.gsc++;
    L.cad_ctr = (L.cad_ctr + 1 == P.cadence) ? 0u : L.cad_ctr + 1;
    // ---- silence detection (fsk.ts:285-295)
    L.sil_cnt = (amp < L.sil_thr) ? L.sil_cnt + 1 : 0u;
    eod = L.sil_cnt >= P.eod_min;
    // ---- bit clock, ungated (fsk.ts:331-335): every (re)start of a frame zeroes these
    L.bit_acc += bit;
    L.bit_wait -= 1u;
    decide = (int32_t)L.bit_wait <= 0;
    // ---- frame sync candidate (fsk.ts:297-315); thr_eff is 0xFFFFFFFF while started
    cand = (L.matched >= L.thr_eff) & (L.cad_ctr == 0) & (R.k >= need);
  }

  // ---- rare path 1: end of data (fsk.ts:288-291) ------------------------------------------------
  if (__ballot(eod)) {
    if (eod) {
      O.eod_cnt++;
      L.eod_total++;
      reset_state(L, P.matched_min);
      did_reset = true;
    }
  }
  // ---- rare path 2: frame sync found (fsk.ts:315-327) -------------------------------------------
  const bool sync_now = cand & !eod;
  uint64_t m = __ballot(sync_now);
  if (m) {
    if (sync_now) {
      L.started = 1; L.thr_eff = 0xFFFFFFFFu;
      L.byte_cur = 0; L.bit_pos = 0;
      L.bit_acc = 0; L.bit_wait = 0; L.bit_reload = 0;
      L.sync_det++;
    }
    // silence.threshold = mean(syncAmplitudeBuffer) * 0.1: the 64 lanes read the syncing stream's
    // ring column together and tree-reduce in f64
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's ring stores have reached L2
    const uint32_t pushes = amp_base + R.k;
    const uint32_t my_len = pushes < P.amp_cap ? pushes : P.amp_cap;
    while (m) {
      const int src = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      const uint32_t srow = __shfl(row, src, 64);
      const uint32_t slen = __shfl(my_len, src, 64);
      double part = 0.0;
      for (uint32_t i = lane; i < slen; i += 64) {
        const float *p = S.amp_ring + (size_t)i * P.n_streams + srow;
        part += (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L1 bypass
      }
      const double sum = wave_sum(part);
      if ((int)lane == src) L.sil_thr = (Real)((sum / (double)slen) * 0.1);
    }
  }
  // ---- rare path 3 (some lane nearly every step): bit decision + processByte (fsk.ts:335-375) ---
  const bool dec_now = decide & !eod & (L.started != 0);
  if (__ballot(dec_now)) {
    bool emit = false, bad_start = false;
    if (dec_now) {
      const uint32_t cnt = L.bit_reload - L.bit_wait;           // bitAccumCount
      const uint32_t b = (2u * L.bit_acc > cnt) ? 1u : 0u;      // fsk.ts:336
      L.bit_acc = 0;
      L.bit_wait += P.d;                                        // nextBitSampleIndex += dsSPB
      L.bit_reload = L.bit_wait;
      const uint32_t pos = L.bit_pos;
      // data bits MSB first: positions 1..8 land in bits 7..0; position 0 (start bit, must be 0 to
      // get here) and positions >= 9 land above bit 7 and are masked off when the byte is emitted
      L.byte_cur |= b << ((8u - pos) & 31u);
      const bool is_stop = pos == P.stop_pos;
      bad_start = (pos == 0) & (b != 0);
      emit = is_stop & (b != 0);
      L.bit_pos = is_stop ? 0u : pos + 1;
      if (is_stop & (b == 0)) {                                 // bad stop bit: fsk.ts:363-366
        L.started = 0; L.thr_eff = P.matched_min; L.bit_wait = kBigWait;
        L.bit_pos = pos;
      }
    }
    if (__ballot(bad_start)) {
      if (bad_start) { reset_state(L, P.matched_min); did_reset = true; }  // fsk.ts:352-355
    }
    if (__ballot(emit)) {
      if (emit) {
        if (valid && O.out_cnt < O.out_pitch) O.out_row[O.out_cnt] = (uint8_t)L.byte_cur;
        O.out_cnt++;
        L.byte_cur = 0;
      }
    }
  }
  // bit_wait ran down without a frame (12 h of decimated samples): park it again
  if (__ballot(decide & (L.started == 0))) {
    if (decide & (L.started == 0)) L.bit_wait = kBigWait;
  }
  return did_reset;
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
// fp32 kernels are capped at 128 VGPRs (4 waves/SIMD): the chain is VALU-bound and one wave per
// SIMD issues a VALU op only every 4 cycles (MI355X_MICROARCH.md), so occupancy is throughput.
template <typename Real, typename PolyT, bool FRAC, bool UNI, bool TRACE>
__global__ __launch_bounds__(64, (sizeof(Real) == 4 ? FSK_F32_WAVES_PER_SIMD : 1)) void demod_kernel(DemodParams P, DemodState S, float *__restrict__ samples,
                                                   size_t n, size_t pitch, int vec_ok, int writeback, int append,
                                                   uint8_t *__restrict__ out, size_t out_pitch,
                                                   uint32_t *__restrict__ out_counts,
                                                   uint32_t *__restrict__ eod_counts) {
  extern __shared__ float4 lds[];
  float4 *stage = lds;                                               // [kChunks][kSlotStride]
  PolyT *poly = (PolyT *)(lds + kChunks * kSlotStride);              // [d][64]
  PolyT *gpoly = (PolyT *)S.poly + (size_t)blockIdx.x * P.d * 64u;   // this wave's registers in HBM
  PolyT *poly_u = poly + (FRAC ? 64u * P.d : 0u);
  PolyT *gpoly_u = (PolyT *)S.poly_u + (size_t)blockIdx.x * P.d * 64u;

  const uint32_t lane = threadIdx.x;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const bool valid = stream < P.n_streams;
  const uint32_t row = valid ? stream : P.n_streams - 1;
  const size_t ns = P.n_streams;

  Lane<Real> L;
  load_lane(L, S, ns, row);
  L.thr_eff = L.started ? 0xFFFFFFFFu : P.matched_min;
  Consts<Real> C;
  C.init(P, S, row);
  for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
  if (FRAC)
    for (uint32_t p = 0; p < P.d; p++) poly_u[p * 64u + lane] = gpoly_u[p * 64u + lane];

  // ring positions: wave-uniform (SGPR) in UNI launches
  RingPos R;
  R.phase = UNI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.poly_phase) : L.poly_phase;
  R.amp_pos = UNI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.amp_pos) : L.amp_pos;
  R.k = 0;
  const uint32_t ring_base = L.ring_len, amp_base = L.amp_len;
  // the sync search needs ring length >= sample_count: true once k >= need
  const uint32_t need = ring_base >= P.sample_count ? 0u : P.sample_count - ring_base;

  OutCtx O;
  O.out_row = out + (size_t)row * out_pitch;
  O.out_pitch = (uint32_t)out_pitch;
  O.out_cnt = (append && valid) ? out_counts[stream] : 0;       // continue a preceding launch of the same call
  O.eod_cnt = (append && valid && eod_counts) ? eod_counts[stream] : 0;

  // tile prefetch: load i covers kRowsPerLoad rows; lane -> (row i*kRowsPerLoad + lane/kChunks,
  // chunk lane%kChunks), i.e. kChunks lanes sweep one row's contiguous tile segment.
  // Two separate code paths on purpose: when both lived in one loop hipcc split every 16-B load
  // into dwordx3 + dword with a vmcnt(0) in between, serialising 8 HBM round trips per tile.
  const uint32_t sub_row = lane / kChunks, chunk = lane % kChunks;
  const bool rows_full = (blockIdx.x + 1u) * 64u <= P.n_streams;
  const float *lane_src = samples + (size_t)(blockIdx.x * 64u + sub_row) * pitch + 4u * chunk;
  float4 pre[kChunks];
#pragma unroll
  for (int i = 0; i < kChunks; i++) pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // defined on every path (keeps it in VGPRs)
  auto load_tile_fast = [&](size_t t0) {
#pragma unroll
    for (int i = 0; i < kChunks; i++)
      pre[i] = *reinterpret_cast<const float4 *>(lane_src + (size_t)(kRowsPerLoad * i) * pitch + t0);
  };
  // ragged tiles (row tail of the batch, sample tail of the call, unaligned buffers) are staged
  // synchronously element by element, straight into LDS, so that the fast path above stays free of
  // shared code
  auto stage_tile_slow = [&](size_t t0) {
    for (int i = 0; i < kChunks; i++) {
      uint32_t r = blockIdx.x * 64u + (uint32_t)kRowsPerLoad * i + sub_row;
      r = r < P.n_streams ? r : P.n_streams - 1;
      const size_t c0 = t0 + 4u * chunk;
      const float *src = samples + (size_t)r * pitch + c0;
      float *dst = reinterpret_cast<float *>(&stage[chunk * kSlotStride + (uint32_t)kRowsPerLoad * i + sub_row]);
      for (int q = 0; q < 4; q++) dst[q] = (c0 + q < n) ? src[q] : 0.0f;
    }
  };
  auto tile_is_fast = [&](size_t t0) { return vec_ok && rows_full && t0 + kTile <= n; };

  const bool agc_on = P.agc_on != 0;
#ifdef FSK_ABLATE
  const int ablate = writeback >> 8;
  writeback &= 0xFF;
#endif

  auto trace_put = [&](Real amp, Real post, bool bit) {
    if (TRACE) {
      if (stream == S.trace_stream) {
        uint32_t kk = *S.trace_n;
        if (kk < S.trace_cap) {
          S.trace_amp[kk] = (double)amp;
          S.trace_post[kk] = (double)post;
          S.trace_bit[kk] = bit ? 1 : 0;
        }
        *S.trace_n = kk + 1;
      }
    }
  };

  // generic path: one sample at a time, per-lane decimator phase (fsk.ts:224-276 as written)
  auto step_generic = [&](float x, float &wbv) {
    float pre_y = pre_stage(L, C, agc_on, x, wbv);
    Real fi, fq;
    mix_lp(L, C, pre_y, fi, fq);
    L.acc_i += fi;
    L.acc_q += fq;
    L.ds_cnt++;
    const bool dec = L.ds_cnt >= 2;
    const bool any = UNI ? (bool)__builtin_amdgcn_readfirstlane((int)dec) : (__ballot(dec) != 0);
    if (any) {
      Real amp = (Real)0, post = (Real)0;
      bool bit = false;
      if (UNI || dec) {
        bit = discriminate(L, C, L.acc_i, L.acc_q, amp, post);
        L.acc_i = (Real)0; L.acc_q = (Real)0;
        L.ds_cnt = 0;
        trace_put(amp, post, bit);
      }
      downsampled_bit<Real, PolyT, FRAC>(L, P, S, poly, poly_u, R, need, ring_base, amp_base, lane, row, valid,
                                         UNI || dec, bit, amp, O);
    }
  };

  // Fast path (UNI launches whose decimator is at a pair boundary): four samples at a time.  The
  // four front ends and both discriminators form ONE branch-free block, so the independent
  // recurrences (AGC gain, band-pass, NCO, I/Q low-pass, post filter) of neighbouring samples can
  // overlap in the in-order pipeline.  Samples 2,3 are therefore computed BEFORE the frame state
  // machine has seen pair 0; if that step resets a stream (EOD or bad start bit, fsk.ts:288-291,
  // 352-355) the reset-sensitive part of samples 2,3 is recomputed for those lanes from the
  // zeroed state, which is exactly what the reference's sample-serial order produces.
  auto block4 = [&](const float (&x)[4], float (&wbv)[4]) {
    float pre_y[4];
    Real fi[4], fq[4];
#ifdef FSK_ABLATE
    if (ablate) {  // timing experiments only (tools/): skips stages, results are wrong
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (ablate & 8) { pre_y[k] = x[k]; wbv[k] = x[k]; } else pre_y[k] = pre_stage(L, C, agc_on, x[k], wbv[k]);
        if (ablate & 4) { fi[k] = (Real)pre_y[k]; fq[k] = (Real)pre_y[k] * (Real)0.5; } else mix_lp(L, C, pre_y[k], fi[k], fq[k]);
      }
      Real a0, p0, a1, p1;
      bool b0, b1;
      if (ablate & 2) {
        a0 = fi[0] + fi[1]; p0 = fq[0] + fq[1]; b0 = p0 > (Real)0;
        a1 = fi[2] + fi[3]; p1 = fq[2] + fq[3]; b1 = p1 > (Real)0;
      } else {
        b0 = discriminate(L, C, fi[0] + fi[1], fq[0] + fq[1], a0, p0);
        b1 = discriminate(L, C, fi[2] + fi[3], fq[2] + fq[3], a1, p1);
      }
      if (ablate & 1) {
        L.sil_thr += a0 + a1 + (Real)((b0 ? 1 : 0) + (b1 ? 2 : 0)) + p0 + p1;  // keep everything live
      } else {
        downsampled_bit<Real, PolyT, FRAC>(L, P, S, poly, poly_u, R, need, ring_base, amp_base, lane, row, valid, true, b0, a0, O);
        downsampled_bit<Real, PolyT, FRAC>(L, P, S, poly, poly_u, R, need, ring_base, amp_base, lane, row, valid, true, b1, a1, O);
      }
      return;
    }
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) {
      pre_y[k] = pre_stage(L, C, agc_on, x[k], wbv[k]);
      mix_lp(L, C, pre_y[k], fi[k], fq[k]);
    }
    Real amp0, post0, amp1, post1;
    const bool bit0 = discriminate(L, C, fi[0] + fi[1], fq[0] + fq[1], amp0, post0);
    bool bit1 = discriminate(L, C, fi[2] + fi[3], fq[2] + fq[3], amp1, post1);
    // one copy of the state machine, run twice (keeps the kernel and its live ranges small)
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
      const Real amp = p ? amp1 : amp0;
      const Real post = p ? post1 : post0;
      const bool bit = p ? bit1 : bit0;
      trace_put(amp, post, bit);
      const bool rst = downsampled_bit<Real, PolyT, FRAC>(L, P, S, poly, poly_u, R, need, ring_base, amp_base, lane,
                                                          row, valid, true, bit, amp, O);
      if (p == 0 && __ballot(rst)) {
        if (rst) {  // resetState() ran after pair 0: redo the reset-sensitive half of samples 2,3
          Real gi2, gq2, gi3, gq3;
          mix_lp(L, C, pre_y[2], gi2, gq2);
          mix_lp(L, C, pre_y[3], gi3, gq3);
          bit1 = discriminate(L, C, gi2 + gi3, gq2 + gq3, amp1, post1);
        }
      }
    }
  };

  const bool fast = UNI && (__builtin_amdgcn_readfirstlane((int)L.ds_cnt) == 0);
  bool cur_fast = n > 0 && tile_is_fast(0);
  if (cur_fast) load_tile_fast(0);
  for (size_t t0 = 0; t0 < n; t0 += kTile) {
    __syncthreads();  // single-wave workgroup: orders last tile's LDS reads before the overwrite
    if (cur_fast) {
#pragma unroll
      for (int i = 0; i < kChunks; i++) stage[chunk * kSlotStride + (uint32_t)kRowsPerLoad * i + sub_row] = pre[i];
    } else {
      stage_tile_slow(t0);
    }
    __syncthreads();
    cur_fast = (t0 + kTile < n) && tile_is_fast(t0 + kTile);
    if (cur_fast) load_tile_fast(t0 + kTile);

    const uint32_t tile_len = (uint32_t)((n - t0) < (size_t)kTile ? (n - t0) : (size_t)kTile);
    const uint32_t n_chunks = (tile_len + 3u) >> 2;
    for (uint32_t c = 0; c < n_chunks; c++) {
      float4 v4 = stage[c * kSlotStride + lane];
      float xv[4] = {v4.x, v4.y, v4.z, v4.w};
      float wb[4];
      const uint32_t lim = tile_len - 4u * c < 4u ? tile_len - 4u * c : 4u;
      if (fast && lim == 4u) {
        block4(xv, wb);
      } else {
        const float *xs = reinterpret_cast<const float *>(&stage[c * kSlotStride + lane]);
#pragma unroll 1
        for (uint32_t k = 0; k < lim; k++) {
          float w;
          step_generic(xs[k], w);
          if (writeback && valid) samples[(size_t)row * pitch + t0 + 4u * c + k] = w;
        }
        continue;
      }
      if (writeback && valid) {
        float *dst = samples + (size_t)row * pitch + t0 + 4u * c;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if ((uint32_t)k < lim) dst[k] = wb[k];
      }
    }
  }

  // ring bookkeeping back to per-stream state
  {
    const uint32_t rl = ring_base + R.k;
    L.ring_len = rl < P.ring_cap ? rl : P.ring_cap;
    const uint32_t al = amp_base + R.k;
    L.amp_len = al < P.amp_cap ? al : P.amp_cap;
    L.poly_phase = R.phase;
    L.amp_pos = R.amp_pos;
  }
  for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
  if (FRAC)
    for (uint32_t p = 0; p < P.d; p++) gpoly_u[p * 64u + lane] = poly_u[p * 64u + lane];
  if (valid) {
    store_lane(L, S, ns, row);
    out_counts[stream] = O.out_cnt;
    if (eod_counts) eod_counts[stream] = O.eod_cnt;
  }
}


// ================================================================================================
// Fast kernel: fp32, <= 31 pattern bits, integer ring capacity, every stream of the launch in lock
// step at a decimator pair boundary, whole 16-sample tiles.  Same arithmetic and the same state
// arrays as the generic fp32 kernel above (a call may run this kernel for its first n - n%16
// samples and the generic one for the rest); what differs is instruction economy:
//  * the four front ends of a block and both discriminators are one branch-free region;
//  * the NCO phasor of sample 0 comes from the exact 64-bit turn accumulator through v_cos/v_sin,
//    samples 1..3 by one complex multiply with per-stream constants e^{j k omega};
//  * ring positions / push counts live in SGPRs, the amplitude ring is written with a buffer store
//    whose row offset is an SGPR, globalSampleCounter is derived from the push count;
//  * polyphase registers of both decimated steps are fetched from LDS at block start.
// ================================================================================================

__device__ inline f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f2 bc2(float v) { return (f2){v, v}; }

// Registers of one stream.  Everything that only rare paths touch (eod / sync counters,
// globalSampleCounter, ring lengths) stays in the state arrays in HBM and is read-modify-written
// there: the kernel is capped at 128 VGPRs (4 waves per SIMD).
struct FastLane {
  float g, bx1, bx2, by1, by2;       // AGC gain, pre-filter history
  f2 lx1, lx2, ly, lv;               // I/Q low-pass (x = I, y = Q), velocity form, half scale
  float px1, px2, py, pv;            // post filter
  float last_phase, thr;
  uint32_t nco_lo, nco_hi, cad, sil, acc, wait, reload, byte_cur, bit_pos;
  uint32_t thr_eff;                  // matched_min while searching, kStarted while a frame is started
  uint32_t matched, out_cnt;
};
struct FastConst {                   // per-stream constants (VGPRs)
  float bp_b0, bp_a1, bp_a2;
  f2 w1;                             // e^{j omega}: NCO phasor of the second sample of a pair
  uint32_t inc2_lo, inc2_hi;         // two NCO steps
};
struct FastUni {                     // wave-uniform constants
  float lp_b0, lp_a2, lp_delta, agc_att, agc_rel;   // scalar uses (SGPR operands)
  f2 a2v, ndv;                       // (a2,a2), (-delta,-delta): packed-math operands, pinned in VGPRs
};
// resetState() fsk.ts:175-188.  globalSampleCounter = k + koff with koff kept in the gsc state word.
// UNI (demod_fast_kernel with wave-uniform constants): the NCO runs as a phasor recurrence that is re-seeded from
// the exact 64-bit accumulator at every tile top, and the accumulator only advances per tile (+16 steps); a reset
// after pair p of the tile (k counts pairs of this launch, tiles are 8 pairs) restarts the phasor at 1 and sets the
// accumulator so that the tile-end advance lands on the phase of the next tile's first sample: -(p+1) pair steps.
template <bool UNI>
__device__ inline void fast_reset(FastLane &F, const FastMem &M, uint32_t k, uint32_t matched_min, f2 &z,
                                  uint32_t inc2_lo, uint32_t inc2_hi) {
  if (UNI) {
    const uint64_t back = (uint64_t)(((k - 1u) & 7u) + 1u) * (((uint64_t)inc2_hi << 32) | inc2_lo);
    const uint64_t acc = 0ull - back;
    F.nco_lo = (uint32_t)acc; F.nco_hi = (uint32_t)(acc >> 32);
    z = (f2){1.0f, 0.0f};
  } else {
    F.nco_lo = 0; F.nco_hi = 0;
  }
  F.last_phase = 0.0f;
  ist_store(M, IF_gsc, 0u - k);
  F.cad = 0;
  F.acc = 0; F.wait = kBigWait; F.reload = 0;
  F.byte_cur = 0; F.bit_pos = 0;
  F.thr_eff = M.voff < 0xFFFFFFF0u ? matched_min : 0xFFFFFFFEu;  // lanes beyond the batch stay parked
  F.sil = 0;
  F.lx1 = bc2(0.f); F.lx2 = bc2(0.f); F.ly = bc2(0.f); F.lv = bc2(0.f);
  F.px1 = 0.f; F.px2 = 0.f; F.py = 0.f; F.pv = 0.f;
}

// I/Q low-pass step on the mixed sample m = (y*cos, y*sin); returns the (half-scale) outputs
__device__ inline f2 fast_lp2(FastLane &F, const FastUni &U, f2 m) {
  f2 t = fma2(bc2(2.0f), F.lx1, m) + F.lx2;
  f2 u = fma2(U.ndv, F.ly, t);       // the b0/2 gain rides on the pre-filter output
  F.lv = fma2(U.a2v, F.lv, u);
  F.ly = F.ly + F.lv;
  F.lx2 = F.lx1; F.lx1 = m;
  return F.ly;
}
__device__ inline f2 cmul(f2 z, f2 w) {  // z * w
  return fma2(bc2(z.y), (f2){-w.y, w.x}, bc2(z.x) * w);
}

// phase / amplitude / slicer of one decimated sample from the pair sums (fsk.ts:247-264)
__device__ inline bool fast_disc(FastLane &F, const FastUni &U, f2 sum, float &amp) {
  const float PI = 3.14159265358979323846f;
  const float phase = atan2_amp_fast(sum.y, sum.x, amp);
  // wrap into (-pi, pi] (fsk.ts:255-257): |dphi| <= 2 pi, so one rounded quotient does both branches
  float dphi = phase - F.last_phase;
  dphi = __builtin_fmaf(-2.0f * PI, __builtin_rintf(dphi * (0.5f / PI)), dphi);
  F.last_phase = phase;
  const float f = lp32(U.lp_b0, U.lp_a2, U.lp_delta, F.px1, F.px2, F.py, F.pv, dphi);
  return f > 0.0f;
}

// processDownsampledBit (fsk.ts:278-344); k = pushes of this launch including this one (SGPR),
// phase = push slot (SGPR), amp_soff = byte offset of the amplitude-ring row (SGPR).
template <bool UNI>
__device__ inline void fast_fsm(FastLane &F, const DemodParams &P, const DemodState &S, const FastMem &M,
                                uint32_t *poly, uint32_t lane, __amdgpu_buffer_rsrc_t amp_rsrc, uint8_t *out,
                                uint32_t out_pitch, uint32_t *eod_counts, bool bitb, float amp, uint32_t r_old,
                                uint32_t phase, uint32_t k, uint32_t amp_soff, f2 &z, uint32_t inc2_lo,
                                uint32_t inc2_hi) {
  const uint32_t qn = ~(uint32_t)P.pat_q, mask = (uint32_t)P.pat_mask;
  const uint32_t bit = bitb ? 1u : 0u;
  // syncSamplesBuffer.put(bit)
  const uint32_t r = (r_old << 1) | bit;
  poly[phase * 64u + lane] = r;
  F.matched += (uint32_t)__builtin_popcount((r ^ qn) & mask);
  F.matched -= (uint32_t)__builtin_popcount((r ^ (qn << 1)) & (mask << 1));
  // syncAmplitudeBuffer.put(amp)
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, amp), amp_rsrc, M.voff, amp_soff, 0);
  // globalSampleCounter % round(dsSPB/4)
  const uint32_t c1 = F.cad + 1;
  const bool hit = c1 == P.cadence;
  F.cad = hit ? 0u : c1;
  // silence run (fsk.ts:285-295)
  F.sil = (amp < F.thr) ? F.sil + 1 : 0u;
  const bool eod = F.sil >= P.eod_min;
  // bit clock, ungated
  F.acc += bit;
  F.wait -= 1u;
  const bool decide = (int32_t)F.wait <= 0;
  const bool cand = hit & (F.matched >= F.thr_eff);

  if (__builtin_amdgcn_ballot_w64(eod | cand)) {
    if (eod) {                                                   // fsk.ts:288-291
      ist_store(M, IF_eod_total, ist_load(M, IF_eod_total) + 1u);
      if (eod_counts && M.voff < 0xFFFFFFF0u) eod_counts[M.voff >> 2] += 1u;
      fast_reset<UNI>(F, M, k, P.matched_min, z, inc2_lo, inc2_hi);
    }
    // ring length >= preamble window? (fsk.ts:302); ring_len / amp_len in HBM hold the launch-start values
    bool sync_now = false;
    uint32_t slen = 0;
    if (cand & !eod) {
      const uint32_t ring_base = ist_load(M, IF_ring_len);
      // (lanes beyond the batch are parked at kernel entry and never get here; the row index below must be a real one)
      sync_now = (ring_base + k >= P.sample_count) & (M.voff < 0xFFFFFFF0u);
      const uint32_t pushes = ist_load(M, IF_amp_len) + k;
      slen = pushes < P.amp_cap ? pushes : P.amp_cap;
    }
    uint64_t m = __builtin_amdgcn_ballot_w64(sync_now);
    if (m) {
      if (sync_now) {                                            // fsk.ts:315-319
        F.thr_eff = kStarted;
        F.byte_cur = 0; F.bit_pos = 0;
        F.acc = 0; F.wait = 0; F.reload = 0;
        ist_store(M, IF_sync_det, ist_load(M, IF_sync_det) + 1u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // ring stores have reached L2
      while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t srow = (uint32_t)__builtin_amdgcn_readlane((int)M.voff, src) >> 2;
        const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)slen, src);
        double part = 0.0;
        for (uint32_t i = lane; i < sl; i += 64) {
          const float *p = S.amp_ring + (size_t)i * P.n_streams + srow;
          part += (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double sum = wave_sum(part);
        if ((int)lane == src) F.thr = (float)((sum / (double)sl) * 0.1);   // fsk.ts:321-326
      }
    }
  }

  const bool dec_now = decide & !eod;
  if (__builtin_amdgcn_ballot_w64(dec_now)) {
    bool bad_start = false, emit = false, bad_stop = false, stale = false;
    if (dec_now) {
      if (F.thr_eff == kStarted) {
        const uint32_t cnt = F.reload - F.wait;                  // bitAccumCount
        const uint32_t b = (2u * F.acc > cnt) ? 1u : 0u;         // fsk.ts:336
        F.acc = 0;
        F.wait += P.d;
        F.reload = F.wait;
        const uint32_t pos = F.bit_pos;
        F.byte_cur |= b << ((8u - pos) & 31u);                   // data bits MSB first (fsk.ts:358)
        const bool is_stop = pos == P.stop_pos;
        bad_start = (pos == 0) & (b != 0);
        emit = is_stop & (b != 0);
        bad_stop = is_stop & (b == 0);
        F.bit_pos = is_stop ? 0u : pos + 1;
      } else {
        stale = true;                                            // bit_wait ran down without a frame
      }
    }
    if (__builtin_amdgcn_ballot_w64(bad_start | bad_stop | stale)) {
      if (bad_start) fast_reset<UNI>(F, M, k, P.matched_min, z, inc2_lo, inc2_hi);   // fsk.ts:352-355
      if (bad_stop) { F.thr_eff = P.matched_min; F.wait = kBigWait; F.bit_pos = P.stop_pos; }  // fsk.ts:363-366
      if (stale) F.wait = kBigWait;
    }
    if (__builtin_amdgcn_ballot_w64(emit)) {
      if (emit) {                                                // fsk.ts:367-368
        if (M.voff < 0xFFFFFFF0u && F.out_cnt < out_pitch)
          out[(size_t)(M.voff >> 2) * out_pitch + F.out_cnt] = (uint8_t)F.byte_cur;
        F.out_cnt++;
        F.byte_cur = 0;
      }
    }
  }
}

// WB: also write the AGC-scaled samples back (fsk.ts:55); that variant keeps four more values live per
// chunk and is built for 3 waves/SIMD, the plain one for FSK_FAST_WAVES (4: 128 VGPRs).
// UNI: every stream shares one configuration (DemodParams::uni_cfg): pre-filter coefficients, NCO phasors and
// increments are wave-uniform constants in SGPRs instead of seven VGPRs per lane, which pays for
//   * the pre-filter evaluated two samples at a time in look-ahead form as packed math,
//     (y0, y1) = (u0, u1 - a1 u0) + (-a1, a1^2 - a2) y[-1] + (-a2, a1 a2) y[-2]        (5 instructions instead of 8),
//   * the NCO as a phasor recurrence z <- z e^{2j omega} per pair, re-seeded from the exact 64-bit turn accumulator
//     with v_cos/v_sin at every tile top (8 pairs: drift <= 1e-6), the accumulator advancing once per tile.
template <bool WB, bool UNI>
__global__ __launch_bounds__(64, (WB ? 3 : FSK_FAST_WAVES)) void demod_fast_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n, size_t pitch,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts) {
  extern __shared__ float4 lds[];
  v4f *stage = reinterpret_cast<v4f *>(lds);                 // [4 chunks][kSlotStride]
  uint32_t *poly = (uint32_t *)(lds + 4 * kSlotStride);      // [d][64]
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)blockIdx.x * P.d * 64u;

  const uint32_t lane = threadIdx.x;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const bool valid = stream < P.n_streams;
  const uint32_t row = valid ? stream : P.n_streams - 1;
  // Per-stream state goes through buffer instructions: descriptor in SGPRs, field offset as the
  // SGPR soffset, ONE VGPR offset (row*4).  With plain pointers hipcc keeps a 64-bit address pair
  // per field alive across the whole sample loop for the stores at the end (~90 VGPRs).
  const uint32_t fld = P.n_streams * 4u;  // bytes per state field
  const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.rs, 0, (int)(fld * RF_COUNT), 0x00020000);
  const __amdgpu_buffer_rsrc_t cf_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)S.coef, 0, (int)(2u * fld * CF_COUNT), 0x00020000);
  FastMem M;
  M.is_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.is, 0, (int)(fld * IF_COUNT), 0x00020000);
  M.fld = fld;
  M.voff = valid ? row * 4u : 0xFFFFFFF0u;  // stores of lanes beyond the batch are dropped by the bounds check
  const uint32_t row4 = row * 4u;
#define RLOAD(f) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, row4, (uint32_t)RF_##f * fld, 0))
#define ILOAD(f) __builtin_amdgcn_raw_buffer_load_b32(M.is_rsrc, row4, (uint32_t)IF_##f * fld, 0)
#define CLOAD(f) ((float)__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cf_rsrc, row4 * 2u, (uint32_t)(f) * fld * 2u, 0)))

  FastLane F;
  F.g = RLOAD(agc_gain);
  F.bx1 = RLOAD(bp_x1); F.bx2 = RLOAD(bp_x2); F.by1 = RLOAD(bp_y1); F.by2 = RLOAD(bp_y2);
  F.lx1 = (f2){RLOAD(li_x1), RLOAD(lq_x1)}; F.lx2 = (f2){RLOAD(li_x2), RLOAD(lq_x2)};
  F.ly = (f2){RLOAD(li_y1), RLOAD(lq_y1)}; F.lv = (f2){RLOAD(li_y2), RLOAD(lq_y2)};
  F.px1 = RLOAD(po_x1); F.px2 = RLOAD(po_x2); F.py = RLOAD(po_y1); F.pv = RLOAD(po_y2);
  F.last_phase = RLOAD(last_phase); F.thr = RLOAD(sil_thr);
  F.nco_lo = ILOAD(nco_lo); F.nco_hi = ILOAD(nco_hi);
  F.cad = ILOAD(cad_ctr); F.sil = ILOAD(sil_cnt); F.acc = ILOAD(bit_acc); F.wait = ILOAD(bit_wait);
  F.reload = ILOAD(bit_reload); F.byte_cur = ILOAD(byte_cur); F.bit_pos = ILOAD(bit_pos);
  F.matched = ILOAD(matched);
  F.thr_eff = ILOAD(started) ? kStarted : P.matched_min;
  F.out_cnt = 0;
  if (valid && eod_counts) eod_counts[stream] = 0;  // incremented in memory by the (rare) EOD path
  if (!valid) {
    // Lanes beyond the batch run on zeros with a copy of the last stream's state.  Park them: no sync candidate (a
    // threshold `matched` cannot reach), no silence run (nothing is below a negative threshold), no bit clock -- so
    // they never enter a rare path, where their out-of-range row index would be used as an address (the sync path's
    // amplitude-column read faulted on exactly that: tools/soak.py, S = 1 with a lowered syncThreshold).
    F.thr_eff = 0xFFFFFFFEu; F.thr = -1.0f; F.wait = kBigWait;
  }

  FastConst K;
  if (UNI) {
    K.bp_b0 = P.u_bp_b0h; K.bp_a1 = -P.u_bp_na1; K.bp_a2 = -P.u_bp_na2;
    K.w1 = (f2){P.u_w1_re, P.u_w1_im};
    K.inc2_lo = P.u_inc2_lo; K.inc2_hi = P.u_inc2_hi;
  } else {
    K.bp_b0 = (float)(__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cf_rsrc, row4 * 2u, (uint32_t)CF_bp_b0 * fld * 2u, 0)) * (0.5 * P.lp_b0));
    K.bp_a1 = CLOAD(CF_bp_a1); K.bp_a2 = CLOAD(CF_bp_a2);
    K.w1 = (f2){CLOAD(CF_w1_re), CLOAD(CF_w1_im)};
    const uint64_t inc = S.nco_inc[row];
    K.inc2_lo = (uint32_t)(inc << 1); K.inc2_hi = (uint32_t)((inc << 1) >> 32);
  }
  // UNI state views: pre-filter history as (older, newer) pairs, NCO phasor of the next pair's first sample
  f2 ubx = (f2){F.bx2, F.bx1}, uby = (f2){F.by2, F.by1};
  f2 z = (f2){1.0f, 0.0f};
  const f2 w2 = (f2){P.u_w2_re, P.u_w2_im};
  FastUni U;
  U.lp_b0 = P.f_lp_b0; U.lp_a2 = P.f_lp_a2; U.lp_delta = P.f_lp_delta;
  U.agc_att = P.f_agc_att; U.agc_rel = P.f_agc_rel;
  U.a2v = bc2(P.f_lp_a2); U.ndv = bc2(-P.f_lp_delta);
  // opaque VGPR pairs: otherwise hipcc parks these uniform values in scratch and reloads them (a
  // VMEM op, hence a vmcnt wait behind the tile prefetch) at the top of every tile
  asm volatile("" : "+v"(U.a2v), "+v"(U.ndv));

  for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];

  // wave-uniform ring bookkeeping (SGPRs)
  uint32_t phase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ILOAD(poly_phase));
  const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ILOAD(amp_pos));
  uint32_t k = 0;
  const uint32_t amp_row_bytes = P.n_streams * 4u;
  uint32_t amp_soff = amp_pos0 * amp_row_bytes;
  const uint32_t amp_wrap = P.amp_cap * amp_row_bytes;
  const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);

  // Tile prefetch: 4 loads of 16 rows x 64 B; lane -> (row 16*i + lane/4, chunk lane%4), through a per-wave
  // buffer descriptor over this wave's 64 rows (rows beyond the batch read as 0 via the bounds check on the
  // VGPR offset).  Inline asm on purpose: vmcnt counts loads AND stores in issue
  // order and hipcc cannot count the stores this loop issues conditionally, so with compiler-visible loads
  // it waits vmcnt(0) at the top of every tile -- i.e. for the amplitude-ring stores issued a few hundred
  // cycles earlier (~1-2 us each).  With asm loads the wait is ours: each tile issues at least 8 VMEM ops
  // after its prefetch (the unconditional ring stores), so vmcnt(8) retires exactly the loads; extra
  // conditional stores only make the wait more conservative, never unsafe.
  const uint32_t sub_row = lane >> 2, chunk = lane & 3;
  const uint32_t rows_here = P.n_streams - blockIdx.x * 64u < 64u ? P.n_streams - blockIdx.x * 64u : 64u;
  v4i in_rsrc;
  {
    const uint64_t base = reinterpret_cast<uint64_t>(samples + (size_t)blockIdx.x * 64u * pitch);
    in_rsrc.x = (int)(uint32_t)base;
    in_rsrc.y = (int)(uint32_t)(base >> 32);          // stride 0
    in_rsrc.z = (int)(uint32_t)(rows_here * pitch * 4u);
    in_rsrc.w = 0x00020000;
  }
  const uint32_t in_voff = (uint32_t)((sub_row * pitch + 4u * chunk) * 4u);
  const uint32_t in_row16 = (uint32_t)(16u * pitch * 4u);   // byte step between the four loads
  v4f pre0, pre1, pre2, pre3;
  // The 16-row step between the four loads rides in the VGPR offset, not in soffset: the bounds check that turns rows
  // beyond the batch into zeros covers voffset only -- soffset is added to the address unchecked, and with it a partial
  // wave read (and for a small batch faulted on) memory past the end of the buffer (found by tools/soak.py).
#define FSK_BLOAD4(dst, rows16, soff)                                                                      \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(in_voff + (rows16) * in_row16), \
               "s"(in_rsrc), "s"(soff) : "memory")
  {
    const uint32_t s0 = 0u;
    FSK_BLOAD4(pre0, 0u, s0); FSK_BLOAD4(pre1, 1u, s0); FSK_BLOAD4(pre2, 2u, s0); FSK_BLOAD4(pre3, 3u, s0);
  }
  // everything loaded so far (state, constants, first tile) is complete before the loop; the builtin form
  // also tells hipcc's own scoreboard, so it needs no vmcnt wait inside the loop
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre0), "+v"(pre1), "+v"(pre2), "+v"(pre3) : : "memory");
  const uint32_t st_slot = chunk * kSlotStride + sub_row;

  for (size_t t0 = 0; t0 < n; t0 += kFastTile) {
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(pre0), "+v"(pre1), "+v"(pre2), "+v"(pre3) : : "memory");
    stage[st_slot] = pre0;
    stage[st_slot + 16] = pre1;
    stage[st_slot + 32] = pre2;
    stage[st_slot + 48] = pre3;
    __syncthreads();
    {
      // next tile (the last iteration re-reads its own tile: always in bounds, never used)
      const uint32_t tn = (uint32_t)((t0 + kFastTile < n ? t0 + kFastTile : t0) * 4u);
      FSK_BLOAD4(pre0, 0u, tn); FSK_BLOAD4(pre1, 1u, tn); FSK_BLOAD4(pre2, 2u, tn); FSK_BLOAD4(pre3, 3u, tn);
    }

    if (UNI) {  // re-seed the phasor from the exact accumulator (first sample of this tile)
      const float turns = (float)F.nco_hi * 2.3283064365386963e-10f;
      z = (f2){__builtin_amdgcn_cosf(turns), __builtin_amdgcn_sinf(turns)};
    }

#pragma unroll 1
    for (uint32_t c = 0; c < 4; c++) {
      const v4f x4 = stage[c * kSlotStride + lane];
      // polyphase registers of this chunk's two decimated steps (d >= 2: distinct slots)
      const uint32_t ph0 = phase, ph1 = (phase + 1 == P.d) ? 0u : phase + 1;
      const uint32_t r0 = poly[ph0 * 64u + lane];
      const uint32_t r1 = poly[ph1 * 64u + lane];
      // two samples at a time: front ends of a decimator pair, discriminator, state machine.  (A 4-sample
      // region would save ~1.5 instructions/sample on the NCO but costs ~40 VGPRs and a speculative
      // recompute path; at 4 waves/SIMD the pair form is faster.)
      const float xin[4] = {x4.x, x4.y, x4.z, x4.w};
      float xs[4];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float y[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          // AGC (fsk.ts:52-76), branch-free; exact zero holds the gain
          const float xv = xin[2 * h + j] * F.g;
          xs[2 * h + j] = xv;
          const float level = __builtin_fabsf(xv);
          const float t = __builtin_fmaf(0.5f, __builtin_amdgcn_rcpf(level), -F.g);
          const float rate = level > 0.5f ? U.agc_att : U.agc_rel;
          float g = __builtin_fmaf(t, rate, F.g);
          g = level > 0.0f ? g : F.g;
          F.g = __builtin_amdgcn_fmed3f(g, 0.1f, 10.0f);
          if (!UNI) {
            // pre-filter (filters.ts:47-87): y = b0*(x - x2) - a2*y2 - a1*y1
            float v = K.bp_b0 * (xv - F.bx2);
            v = __builtin_fmaf(-K.bp_a2, F.by2, v);
            v = __builtin_fmaf(-K.bp_a1, F.by1, v);
            F.bx2 = F.bx1; F.bx1 = xv;
            F.by2 = F.by1; F.by1 = v;
            y[j] = v;
          }
        }
        f2 z0, z1;
        if (UNI) {
          // pre-filter, both samples of the pair at once (look-ahead form, see the kernel comment)
          const f2 xv2 = (f2){xs[2 * h], xs[2 * h + 1]};
          f2 u = bc2(P.u_bp_b0h) * (xv2 - ubx);
          u.y = __builtin_fmaf(P.u_bp_na1, u.x, u.y);
          f2 yy = fma2((f2){P.u_bp_na1, P.u_bp_c1y}, bc2(uby.y), u);
          yy = fma2((f2){P.u_bp_na2, P.u_bp_c2y}, bc2(uby.x), yy);
          ubx = xv2; uby = yy;
          y[0] = yy.x; y[1] = yy.y;
          // NCO: phasor recurrence
          z0 = z;
          z1 = cmul(z0, K.w1);
          z = cmul(z0, w2);
        } else {
          // NCO (fsk.ts:228-232): first sample from the exact 64-bit turn accumulator, second = first * e^{j omega}
          const float turns = (float)F.nco_hi * 2.3283064365386963e-10f;
          z0 = (f2){__builtin_amdgcn_cosf(turns), __builtin_amdgcn_sinf(turns)};
          z1 = cmul(z0, K.w1);
          const uint32_t lo = F.nco_lo + K.inc2_lo;
          F.nco_hi = F.nco_hi + K.inc2_hi + (lo < F.nco_lo ? 1u : 0u);
          F.nco_lo = lo;
        }
        // mix + I/Q low-pass + /2 boxcar (fsk.ts:229-248), discriminator (fsk.ts:251-264)
        const f2 o0 = fast_lp2(F, U, bc2(y[0]) * z0);
        const f2 o1 = fast_lp2(F, U, bc2(y[1]) * z1);
        float amp;
        const bool bit = fast_disc(F, U, o0 + o1, amp);
        k++;
        fast_fsm<UNI>(F, P, S, M, poly, lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, bit, amp, h ? r1 : r0,
                      h ? ph1 : ph0, k, amp_soff, z, K.inc2_lo, K.inc2_hi);
        amp_soff += amp_row_bytes; if (amp_soff == amp_wrap) amp_soff = 0;
      }
      phase = (ph1 + 1 == P.d) ? 0u : ph1 + 1;

      if (WB) {
        if (valid) {
          const v4f w4 = {xs[0], xs[1], xs[2], xs[3]};
          *reinterpret_cast<v4f *>(samples + (size_t)row * pitch + t0 + 4u * c) = w4;
        }
      }
    }
    if (UNI) {  // the accumulator moves one tile at a time (fast_reset accounts for that)
      const uint32_t lo = F.nco_lo + P.u_inc16_lo;
      F.nco_hi = F.nco_hi + P.u_inc16_hi + (lo < F.nco_lo ? 1u : 0u);
      F.nco_lo = lo;
    }
  }
  if (UNI) { F.bx2 = ubx.x; F.bx1 = ubx.y; F.by2 = uby.x; F.by1 = uby.y; }

  // The last iteration's prefetch is still in flight and its destination registers are dead to the compiler:
  // without this wait the epilogue reuses them (e.g. as the high half of a store address) and a late-landing
  // load overwrites them -- a wild global store.  Keep them allocated until the loads have landed.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre0), "+v"(pre1), "+v"(pre2), "+v"(pre3) : : "memory");

  for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
  {
#define RSTORE(f, v) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)(v)), rs_rsrc, M.voff, (uint32_t)RF_##f * fld, 0)
#define ISTORE(f, v) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(v), M.is_rsrc, M.voff, (uint32_t)IF_##f * fld, 0)
    RSTORE(agc_gain, F.g);
    RSTORE(bp_x1, F.bx1); RSTORE(bp_x2, F.bx2); RSTORE(bp_y1, F.by1); RSTORE(bp_y2, F.by2);
    RSTORE(li_x1, F.lx1.x); RSTORE(lq_x1, F.lx1.y); RSTORE(li_x2, F.lx2.x); RSTORE(lq_x2, F.lx2.y);
    RSTORE(li_y1, F.ly.x); RSTORE(lq_y1, F.ly.y); RSTORE(li_y2, F.lv.x); RSTORE(lq_y2, F.lv.y);
    RSTORE(po_x1, F.px1); RSTORE(po_x2, F.px2); RSTORE(po_y1, F.py); RSTORE(po_y2, F.pv);
    RSTORE(last_phase, F.last_phase); RSTORE(sil_thr, F.thr);
    ISTORE(nco_lo, F.nco_lo); ISTORE(nco_hi, F.nco_hi);
    ISTORE(cad_ctr, F.cad); ISTORE(sil_cnt, F.sil); ISTORE(bit_acc, F.acc); ISTORE(bit_wait, F.wait);
    ISTORE(bit_reload, F.reload); ISTORE(byte_cur, F.byte_cur); ISTORE(bit_pos, F.bit_pos);
    ISTORE(started, F.thr_eff == kStarted ? 1u : 0u); ISTORE(matched, F.matched);
    ISTORE(gsc, k + ILOAD(gsc));              // the gsc word held koff during the launch
    const uint32_t rl = ILOAD(ring_len) + k, al = ILOAD(amp_len) + k;
    ISTORE(ring_len, rl < P.ring_cap ? rl : P.ring_cap);
    ISTORE(amp_len, al < P.amp_cap ? al : P.amp_cap);
    ISTORE(poly_phase, phase);
    ISTORE(amp_pos, amp_soff / amp_row_bytes);
    if (valid) out_counts[stream] = F.out_cnt;
#undef RSTORE
#undef ISTORE
  }
#undef RLOAD
#undef ILOAD
#undef CLOAD
#undef FSK_BLOAD4
}


// ================================================================================================
// Split variant of the fast kernel for batches too small to give every SIMD more than one wave (one wave per
// 64 streams cannot hide its own dependency stalls): TWO waves share a 64-stream group as a two-stage pipeline
// over 16-sample tiles.  Wave 0 ("front") stages the raw tile, runs AGC + pre-filter and leaves the filtered
// samples in a double-buffered LDS tile; wave 1 ("back") runs NCO / mix / low-pass / discriminator / state machine
// one tile behind.  The cut is where the reference's resetState() stops reaching (fsk.ts:175-188 resets neither
// the AGC nor the pre-filter), so a reset in the back wave never invalidates anything the front wave has produced.
// One s_barrier per tile.  Same arithmetic, same state layout, interchangeable with demod_fast_kernel call by
// call.  (A more even cut -- NCO, mixer and I/Q low-pass in the front wave too, run speculatively and repaired
// after a reset -- was built, passed parity, and was slower: 269 vs 285 Gsamples/s at 65 536 streams; the lock
// step of the two waves costs more than the better balance gains.  Dealing the roles by CU arrival order read
// from HW_ID made no difference either, and running both pairs of a chunk as one straight-line block in the back wave, with a
// recompute of the second pair after a reset, was slower as well: 287 vs 307.)
// ================================================================================================
template <bool WB, bool UNI>
__global__ __launch_bounds__(128) void demod_split_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n, size_t pitch,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts) {
  extern __shared__ float4 lds[];
  v4f *stage = reinterpret_cast<v4f *>(lds);                                  // [4 chunks][kSlotStride]
  v4f *ybuf = reinterpret_cast<v4f *>(lds + 4 * kSlotStride);                 // [2 tiles][4 chunks][64 lanes]
  uint32_t *poly = (uint32_t *)(lds + 4 * kSlotStride + 2 * 4 * 64);          // [d][64]
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)blockIdx.x * P.d * 64u;

  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const bool valid = stream < P.n_streams;
  const uint32_t row = valid ? stream : P.n_streams - 1;
  const uint32_t fld = P.n_streams * 4u;
  const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.rs, 0, (int)(fld * RF_COUNT), 0x00020000);
  const __amdgpu_buffer_rsrc_t cf_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)S.coef, 0, (int)(2u * fld * CF_COUNT), 0x00020000);
  const uint32_t row4 = row * 4u;
  const uint32_t voff = valid ? row * 4u : 0xFFFFFFF0u;
  const size_t n_tiles = n / kFastTile;
#define RLOAD(f) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, row4, (uint32_t)RF_##f * fld, 0))
#define CLOAD(f) ((float)__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cf_rsrc, row4 * 2u, (uint32_t)(f) * fld * 2u, 0)))
#define RSTORE(f, v) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)(v)), rs_rsrc, voff, (uint32_t)RF_##f * fld, 0)
  // LDS hand-off: this wave's ds ops are done, then the workgroup barrier
#define TILE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  if (wave == 0) {
    // ------------------------------------------------------------------ front: AGC + pre-filter
    float g = RLOAD(agc_gain), bx1 = RLOAD(bp_x1), bx2 = RLOAD(bp_x2), by1 = RLOAD(bp_y1), by2 = RLOAD(bp_y2);
    const float bp_b0 = (float)(__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cf_rsrc, row4 * 2u, (uint32_t)CF_bp_b0 * fld * 2u, 0)) * (0.5 * P.lp_b0));
    const float bp_a1 = CLOAD(CF_bp_a1), bp_a2 = CLOAD(CF_bp_a2);
    const float agc_att = P.f_agc_att, agc_rel = P.f_agc_rel;
    // per-wave descriptor over this group's 64 rows; rows beyond the batch read as 0 through the bounds check
    const uint32_t sub_row = lane >> 2, chunk = lane & 3;
    const uint32_t rows_here = P.n_streams - blockIdx.x * 64u < 64u ? P.n_streams - blockIdx.x * 64u : 64u;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        samples + (size_t)blockIdx.x * 64u * pitch, 0, (int)(uint32_t)(rows_here * pitch * 4u), 0x00020000);
    const uint32_t in_voff = (uint32_t)((sub_row * pitch + 4u * chunk) * 4u);
    const uint32_t in_row16 = (uint32_t)(16u * pitch * 4u);
    const uint32_t st_slot = chunk * kSlotStride + sub_row;
    // compiler-visible loads: this wave has a quarter of the back wave's arithmetic per tile, so the vmcnt waits
    // hipcc places (conservative at the loop header) are hidden behind the barrier it would wait at anyway
    auto load_tile = [&](size_t t, v4f &a, v4f &b, v4f &c, v4f &d) {
      const uint32_t tn = (uint32_t)((t < n_tiles ? t : n_tiles - 1) * kFastTile * 4u);
      // (row step in the bounds-checked VGPR offset, see demod_fast_kernel)
      a = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, in_voff, tn, 0));
      b = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, in_voff + in_row16, tn, 0));
      c = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, in_voff + 2u * in_row16, tn, 0));
      d = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, in_voff + 3u * in_row16, tn, 0));
    };
    v4f a0, a1, a2, a3, b0, b1, b2, b3;
    load_tile(0, a0, a1, a2, a3);
    load_tile(1, b0, b1, b2, b3);
    for (size_t t = 0; t <= n_tiles; t++) {
      if (t < n_tiles) {
        stage[st_slot] = a0; stage[st_slot + 16] = a1; stage[st_slot + 32] = a2; stage[st_slot + 48] = a3;
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
        load_tile(t + 2, b0, b1, b2, b3);
        v4f *yb = ybuf + (t & 1) * 256u;
        v4f xnext = stage[lane];                          // written by this wave: a wave's ds ops are ordered
#pragma unroll 1
        for (uint32_t c = 0; c < 4; c++) {
          const v4f x4 = xnext;
          xnext = stage[(c < 3 ? c + 1 : 3u) * kSlotStride + lane];
          const float xin[4] = {x4.x, x4.y, x4.z, x4.w};
          float xs[4], y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // AGC (fsk.ts:52-76), branch-free; exact zero holds the gain
            const float xv = xin[j] * g;
            xs[j] = xv;
            const float level = __builtin_fabsf(xv);
            const float tt = __builtin_fmaf(0.5f, __builtin_amdgcn_rcpf(level), -g);
            const float rate = level > 0.5f ? agc_att : agc_rel;
            float gn = __builtin_fmaf(tt, rate, g);
            gn = level > 0.0f ? gn : g;
            g = __builtin_amdgcn_fmed3f(gn, 0.1f, 10.0f);
            // pre-filter (filters.ts:47-87): y = b0*(x - x2) - a2*y2 - a1*y1
            float v = bp_b0 * (xv - bx2);
            v = __builtin_fmaf(-bp_a2, by2, v);
            v = __builtin_fmaf(-bp_a1, by1, v);
            bx2 = bx1; bx1 = xv;
            by2 = by1; by1 = v;
            y[j] = v;
          }
          yb[c * 64u + lane] = (v4f){y[0], y[1], y[2], y[3]};
          if (WB) {
            if (valid) *reinterpret_cast<v4f *>(samples + (size_t)row * pitch + t * kFastTile + 4u * c) = (v4f){xs[0], xs[1], xs[2], xs[3]};
          }
        }
      }
      TILE_BARRIER();
    }
    RSTORE(agc_gain, g);
    RSTORE(bp_x1, bx1); RSTORE(bp_x2, bx2); RSTORE(bp_y1, by1); RSTORE(bp_y2, by2);
  } else {
    // ------------------------------------------------------------------ back: everything resetState() reaches
    FastMem M;
    M.is_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.is, 0, (int)(fld * IF_COUNT), 0x00020000);
    M.fld = fld;
    M.voff = voff;
#define ILOAD(f) __builtin_amdgcn_raw_buffer_load_b32(M.is_rsrc, row4, (uint32_t)IF_##f * fld, 0)
#define ISTORE(f, v) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(v), M.is_rsrc, M.voff, (uint32_t)IF_##f * fld, 0)
    FastLane F;
    F.g = 0.f; F.bx1 = 0.f; F.bx2 = 0.f; F.by1 = 0.f; F.by2 = 0.f;   // front-wave state, unused here
    F.lx1 = (f2){RLOAD(li_x1), RLOAD(lq_x1)}; F.lx2 = (f2){RLOAD(li_x2), RLOAD(lq_x2)};
    F.ly = (f2){RLOAD(li_y1), RLOAD(lq_y1)}; F.lv = (f2){RLOAD(li_y2), RLOAD(lq_y2)};
    F.px1 = RLOAD(po_x1); F.px2 = RLOAD(po_x2); F.py = RLOAD(po_y1); F.pv = RLOAD(po_y2);
    F.last_phase = RLOAD(last_phase); F.thr = RLOAD(sil_thr);
    F.nco_lo = ILOAD(nco_lo); F.nco_hi = ILOAD(nco_hi);
    F.cad = ILOAD(cad_ctr); F.sil = ILOAD(sil_cnt); F.acc = ILOAD(bit_acc); F.wait = ILOAD(bit_wait);
    F.reload = ILOAD(bit_reload); F.byte_cur = ILOAD(byte_cur); F.bit_pos = ILOAD(bit_pos);
    F.matched = ILOAD(matched);
    F.thr_eff = ILOAD(started) ? kStarted : P.matched_min;
    F.out_cnt = 0;
    if (valid && eod_counts) eod_counts[stream] = 0;
    if (!valid) { F.thr_eff = 0xFFFFFFFEu; F.thr = -1.0f; F.wait = kBigWait; }  // park lanes beyond the batch (see demod_fast_kernel)
    FastConst K;
    K.bp_b0 = 0.f; K.bp_a1 = 0.f; K.bp_a2 = 0.f;
    if (UNI) {
      K.w1 = (f2){P.u_w1_re, P.u_w1_im};
      K.inc2_lo = P.u_inc2_lo; K.inc2_hi = P.u_inc2_hi;
    } else {
      K.w1 = (f2){CLOAD(CF_w1_re), CLOAD(CF_w1_im)};
      const uint64_t inc = S.nco_inc[row];
      K.inc2_lo = (uint32_t)(inc << 1); K.inc2_hi = (uint32_t)((inc << 1) >> 32);
    }
    f2 z = (f2){1.0f, 0.0f};
    const f2 w2 = (f2){P.u_w2_re, P.u_w2_im};
    FastUni U;
    U.lp_b0 = P.f_lp_b0; U.lp_a2 = P.f_lp_a2; U.lp_delta = P.f_lp_delta;
    U.agc_att = P.f_agc_att; U.agc_rel = P.f_agc_rel;
    U.a2v = bc2(P.f_lp_a2); U.ndv = bc2(-P.f_lp_delta);
    for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
    uint32_t phase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ILOAD(poly_phase));
    const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ILOAD(amp_pos));
    uint32_t k = 0;
    const uint32_t amp_row_bytes = P.n_streams * 4u;
    uint32_t amp_soff = amp_pos0 * amp_row_bytes;
    const uint32_t amp_wrap = P.amp_cap * amp_row_bytes;
    const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);

    for (size_t t = 0; t <= n_tiles; t++) {
      if (t > 0) {
        const v4f *yb = ybuf + ((t - 1) & 1) * 256u;
        v4f ynext = yb[lane];
        if (UNI) {  // phasor recurrence as in demod_fast_kernel<.., true>: re-seed from the exact accumulator per tile
          const float turns = (float)F.nco_hi * 2.3283064365386963e-10f;
          z = (f2){__builtin_amdgcn_cosf(turns), __builtin_amdgcn_sinf(turns)};
        }
#pragma unroll 1
        for (uint32_t c = 0; c < 4; c++) {
          const v4f y4 = ynext;
          ynext = yb[(c < 3 ? c + 1 : 3u) * 64u + lane];   // next chunk's read in flight while this one computes
          const uint32_t ph0 = phase, ph1 = (phase + 1 == P.d) ? 0u : phase + 1;
          const uint32_t r0 = poly[ph0 * 64u + lane];
          const uint32_t r1 = poly[ph1 * 64u + lane];
          const float yin[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
          for (int h = 0; h < 2; h++) {
            f2 z0, z1;
            if (UNI) {
              z0 = z;
              z1 = cmul(z0, K.w1);
              z = cmul(z0, w2);
            } else {
              // NCO (fsk.ts:228-232): first sample from the exact 64-bit turn accumulator, second = first * e^{j omega}
              const float turns = (float)F.nco_hi * 2.3283064365386963e-10f;
              z0 = (f2){__builtin_amdgcn_cosf(turns), __builtin_amdgcn_sinf(turns)};
              z1 = cmul(z0, K.w1);
              const uint32_t lo = F.nco_lo + K.inc2_lo;
              F.nco_hi = F.nco_hi + K.inc2_hi + (lo < F.nco_lo ? 1u : 0u);
              F.nco_lo = lo;
            }
            const f2 o0 = fast_lp2(F, U, bc2(yin[2 * h]) * z0);
            const f2 o1 = fast_lp2(F, U, bc2(yin[2 * h + 1]) * z1);
            float amp;
            const bool bit = fast_disc(F, U, o0 + o1, amp);
            k++;
            fast_fsm<UNI>(F, P, S, M, poly, lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, bit, amp, h ? r1 : r0,
                          h ? ph1 : ph0, k, amp_soff, z, K.inc2_lo, K.inc2_hi);
            amp_soff += amp_row_bytes; if (amp_soff == amp_wrap) amp_soff = 0;
          }
          phase = (ph1 + 1 == P.d) ? 0u : ph1 + 1;
        }
        if (UNI) {  // the accumulator moves one tile at a time (fast_reset accounts for that)
          const uint32_t lo = F.nco_lo + P.u_inc16_lo;
          F.nco_hi = F.nco_hi + P.u_inc16_hi + (lo < F.nco_lo ? 1u : 0u);
          F.nco_lo = lo;
        }
      }
      TILE_BARRIER();
    }

    for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
    RSTORE(li_x1, F.lx1.x); RSTORE(lq_x1, F.lx1.y); RSTORE(li_x2, F.lx2.x); RSTORE(lq_x2, F.lx2.y);
    RSTORE(li_y1, F.ly.x); RSTORE(lq_y1, F.ly.y); RSTORE(li_y2, F.lv.x); RSTORE(lq_y2, F.lv.y);
    RSTORE(po_x1, F.px1); RSTORE(po_x2, F.px2); RSTORE(po_y1, F.py); RSTORE(po_y2, F.pv);
    RSTORE(last_phase, F.last_phase); RSTORE(sil_thr, F.thr);
    ISTORE(nco_lo, F.nco_lo); ISTORE(nco_hi, F.nco_hi);
    ISTORE(cad_ctr, F.cad); ISTORE(sil_cnt, F.sil); ISTORE(bit_acc, F.acc); ISTORE(bit_wait, F.wait);
    ISTORE(bit_reload, F.reload); ISTORE(byte_cur, F.byte_cur); ISTORE(bit_pos, F.bit_pos);
    ISTORE(started, F.thr_eff == kStarted ? 1u : 0u); ISTORE(matched, F.matched);
    ISTORE(gsc, k + ILOAD(gsc));
    const uint32_t rl = ILOAD(ring_len) + k, al = ILOAD(amp_len) + k;
    ISTORE(ring_len, rl < P.ring_cap ? rl : P.ring_cap);
    ISTORE(amp_len, al < P.amp_cap ? al : P.amp_cap);
    ISTORE(poly_phase, phase);
    ISTORE(amp_pos, amp_soff / amp_row_bytes);
    if (valid) out_counts[stream] = F.out_cnt;
#undef ILOAD
#undef ISTORE
  }
#undef RLOAD
#undef CLOAD
#undef RSTORE
#undef TILE_BARRIER
}

size_t demod_split_lds_bytes(const DemodParams &P) {
  return sizeof(float4) * (4 * kSlotStride + 2 * 4 * 64) + sizeof(uint32_t) * 64u * P.d;
}
hipError_t launch_demod_split(bool writeback, const DemodParams &P, const DemodState &S, float *samples, size_t n,
                              size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                              uint32_t *eod_counts, hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds = demod_split_lds_bytes(P);
#define FSK_LAUNCH_SPLIT(WBV, UNIV)                                                                          \
  hipLaunchKernelGGL((demod_split_kernel<WBV, UNIV>), dim3(blocks), dim3(128), lds, stream, P, S, samples, n, pitch, \
                     out, out_pitch, out_counts, eod_counts)
  const bool uni = P.uni_cfg != 0;
  if (writeback) { if (uni) FSK_LAUNCH_SPLIT(true, true); else FSK_LAUNCH_SPLIT(true, false); }
  else { if (uni) FSK_LAUNCH_SPLIT(false, true); else FSK_LAUNCH_SPLIT(false, false); }
#undef FSK_LAUNCH_SPLIT
  return hipGetLastError();
}

size_t demod_fast_lds_bytes(const DemodParams &P) { return sizeof(float4) * 4 * kSlotStride + sizeof(uint32_t) * 64u * P.d; }

// The fast kernel applies to fp32 engines with narrow integer-capacity rings whose streams are in lock
// step at a pair boundary; n must be a multiple of 16, the buffer 16-B aligned with pitch % 4 == 0.
bool demod_fast_applicable(int precision, bool uniform_even, const DemodParams &P, const DemodState &S,
                           const float *samples, size_t pitch) {
  return precision == 0 && uniform_even && !P.wide && !P.frac && P.d >= 2 && S.trace_stream == 0xFFFFFFFFu &&
         (pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(samples) & 15u) == 0) &&
         demod_fast_lds_bytes(P) <= 64 * 1024 && (uint64_t)P.amp_cap * P.n_streams * 4u < 0xFFFFFFF0ull &&
         (uint64_t)pitch * 4u * 64u < 0x7FFFFFF0ull;  // per-wave input descriptor and offsets fit 31 bits
}
hipError_t launch_demod_fast(bool writeback, const DemodParams &P, const DemodState &S, float *samples, size_t n,
                             size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                             uint32_t *eod_counts, hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds = demod_fast_lds_bytes(P);
#define FSK_LAUNCH_FAST(WBV, UNIV)                                                                          \
  hipLaunchKernelGGL((demod_fast_kernel<WBV, UNIV>), dim3(blocks), dim3(64), lds, stream, P, S, samples, n, pitch, \
                     out, out_pitch, out_counts, eod_counts)
  const bool uni = P.uni_cfg != 0;
  if (writeback) { if (uni) FSK_LAUNCH_FAST(true, true); else FSK_LAUNCH_FAST(true, false); }
  else { if (uni) FSK_LAUNCH_FAST(false, true); else FSK_LAUNCH_FAST(false, false); }
#undef FSK_LAUNCH_FAST
  return hipGetLastError();
}

size_t demod_lds_bytes(const DemodParams &P) {
  const size_t reg = (P.wide ? sizeof(uint64_t) : sizeof(uint32_t)) * 64u * P.d;
  return sizeof(float4) * kChunks * kSlotStride + reg * (P.frac ? 2u : 1u);
}

// kernel variants: Real x {u32, u64, u64+frac} x {uniform, per-lane decimator phase} x trace
#define FSK_FOR_RT(X, R, T, F)                                                                     \
  X(R, T, F, true, false) X(R, T, F, false, false) X(R, T, F, true, true) X(R, T, F, false, true)
#define FSK_FOR_ALL_VARIANTS(X)                                                                    \
  FSK_FOR_RT(X, float, uint32_t, false) FSK_FOR_RT(X, float, uint64_t, false)                      \
  FSK_FOR_RT(X, float, uint64_t, true) FSK_FOR_RT(X, double, uint32_t, false)                      \
  FSK_FOR_RT(X, double, uint64_t, false) FSK_FOR_RT(X, double, uint64_t, true)

// Host-side launcher (called from fsk_api.hip).  uniform_ds: every stream's downsample.counter
// and ring positions are equal (true unless single streams were reset at odd sample positions).
hipError_t launch_demod(int precision, bool uniform_ds, bool writeback, bool append, const DemodParams &P,
                        const DemodState &S, float *samples, size_t n, size_t pitch, uint8_t *out,
                        size_t out_pitch, uint32_t *out_counts, uint32_t *eod_counts,
                        hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds_bytes = demod_lds_bytes(P);
  const int vec_ok = (pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(samples) & 15u) == 0);
  int wb = writeback ? 1 : 0;
#ifdef FSK_ABLATE
  if (const char *a = getenv("FSK_ABLATE")) wb |= atoi(a) << 8;
#endif
  const bool f64 = precision != 0, wide = P.wide != 0, frac = P.frac != 0;
  const bool trace = S.trace_stream != 0xFFFFFFFFu;
  dim3 g(blocks), b(64);
#define FSK_LAUNCH(R, T, F, U, TR)                                                                 \
  if (f64 == (sizeof(R) == 8) && wide == (sizeof(T) == 8) && frac == F && uniform_ds == U &&       \
      trace == TR)                                                                                 \
    hipLaunchKernelGGL((demod_kernel<R, T, F, U, TR>), g, b, lds_bytes, stream, P, S, samples, n,  \
                       pitch, vec_ok, wb, append ? 1 : 0, out, out_pitch, out_counts, eod_counts);
  FSK_FOR_ALL_VARIANTS(FSK_LAUNCH)
#undef FSK_LAUNCH
  return hipGetLastError();
}

hipError_t set_demod_lds_limit(size_t lds_bytes) {
  hipError_t e = hipSuccess;
#define FSK_ATTR(R, T, F, U, TR)                                                                   \
  if (e == hipSuccess)                                                                             \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_kernel<R, T, F, U, TR>),         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  FSK_FOR_ALL_VARIANTS(FSK_ATTR)
#undef FSK_ATTR
  return e;
}

}  // namespace fsk
