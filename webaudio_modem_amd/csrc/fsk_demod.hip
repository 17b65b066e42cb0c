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
// threshold at sync time) and the per-stream state once per launch.
//
// Input rows are [stream][sample]; a wave loads a 64-row x 32-sample tile with coalesced
// 16-B/lane loads (8 lanes cover one 128-B row segment), parks it in LDS chunk-major with a
// one-slot pad so both the ds_write_b128 (8 lanes x 4 banks) and the per-lane ds_read_b128
// are bank-conflict free, and prefetches the next tile into registers while it computes.
//
// The sync correlator is NOT the reference's O(nBits*dsSPB) brute force: the decision-bit
// history is kept as dsSPB polyphase shift registers in LDS (register p holds the bits pushed
// at times == p mod dsSPB, newest in bit 0), so the 30 window slots' entering/leaving bits at
// each step are two masked popcounts of ONE register; `matched` is carried incrementally and is
// at every step exactly the count the reference's double loop would produce.
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

#include "fsk_params.h"

namespace fsk {

// ------------------------------------------------------------------------------------------------
// per-precision arithmetic
// ------------------------------------------------------------------------------------------------
template <typename Real>
struct Biquad {
  Real x1, x2, y1, y2;
};

template <typename Real>
struct Consts;

template <>
struct Consts<float> {
  float lp_b0, lp_a2, lp_delta, agc_att, agc_rel;
  float bp_b0, bp_a1, bp_a2;
  uint32_t inc_lo, inc_hi;
  __device__ void init(const DemodParams &P, const DemodState &S, uint32_t row) {
    // delta = 1 + a1 + a2 (= b0+b1+b2 for the unity-DC-gain Butterworth) is formed in f64 and only
    // then rounded: rounding a1 ~ -1.94 itself to f32 would move the DC gain by ~1e-4 at 300 baud
    lp_b0 = (float)P.lp_b0; lp_a2 = (float)P.lp_a2; lp_delta = (float)(1.0 + P.lp_a1 + P.lp_a2);
    agc_att = (float)P.agc_attack; agc_rel = (float)P.agc_release;
    size_t n = P.n_streams;
    bp_b0 = (float)S.coef[(size_t)CF_bp_b0 * n + row];
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

// AGC + pre-filter + mixer + I/Q low-pass for one input sample (fsk.ts:52-76, 202, 228-243).
__device__ inline void front(Lane<double> &L, const Consts<double> &C, bool agc_on, float xin, float &agc_out) {
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
  float pre = (float)biquad64(C.bp_b0, 0.0, -C.bp_b0, C.bp_a1, C.bp_a2, L.bp_x1, L.bp_x2, L.bp_y1, L.bp_y2, (double)xs);
  double s = (double)pre;
  double ci = s * cos(L.nco_phase);
  double cq = s * sin(L.nco_phase);
  L.nco_phase = fmod(L.nco_phase + C.omega, 2.0 * 3.14159265358979323846);
  double fi = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.li_x1, L.li_x2, L.li_y1, L.li_y2, ci);
  double fq = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.lq_x1, L.lq_x2, L.lq_y1, L.lq_y2, cq);
  L.acc_i += fi;
  L.acc_q += fq;
}

// decimated-rate discriminator (fsk.ts:245-264): returns the slicer bit, amplitude by reference
__device__ inline uint32_t discriminate(Lane<double> &L, const Consts<double> &C, double &amp, double &post) {
  const double PI = 3.14159265358979323846;
  double avg_i = L.acc_i / 2.0;
  double avg_q = L.acc_q / 2.0;
  double phase = atan2(avg_q, avg_i);
  amp = sqrt(avg_i * avg_i + avg_q * avg_q);
  double dphi = phase - L.last_phase;
  if (dphi > PI) dphi -= 2.0 * PI;
  else if (dphi < -PI) dphi += 2.0 * PI;
  L.last_phase = phase;
  double f = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.po_x1, L.po_x2, L.po_y1, L.po_y2, dphi);
  L.acc_i = 0.0; L.acc_q = 0.0;
  post = f;
  return f > 0.0 ? 1u : 0u;
}

__device__ inline void nco_reset(Lane<double> &L) { L.nco_phase = 0.0; }

// ---- fp32: throughput path ---------------------------------------------------------------------
// Same chain, leaner forms: b1 = 0 / b2 = -b0 (band-pass) and b1 = 2*b0, b2 = b0 (low-pass) are
// folded, FMAs are explicit, 1/x is v_rcp_f32, the NCO is a 64-bit turn accumulator feeding
// v_sin_f32 / v_cos_f32 (which take revolutions).

// Low-pass biquad in "velocity" form.  With v = y[n-1] - y[n-2] kept as state,
//   y[n] = y[n-1] + a2*v + (b0*(x + 2*x1 + x2) - delta*y[n-1]),   delta = 1 + a1 + a2
// is algebraically the reference's Direct Form I (filters.ts:47-76) but the poles sit close to
// z = 1 (cutoff = baud << fs), where DF-I in f32 amplifies both coefficient and state rounding by
// 1/|A(1)| ~ 700 (300 baud); here the rounding of y is fed back only through delta ~ 1.5e-3.
// State: y1 holds y[n-1], y2 holds v.
__device__ inline float lp32(float b0, float a2, float delta, float &x1, float &x2, float &y, float &v, float x) {
  float t = __builtin_fmaf(2.0f, x1, x) + x2;
  float u = __builtin_fmaf(-delta, y, b0 * t);
  v = __builtin_fmaf(a2, v, u);
  y = y + v;
  x2 = x1; x1 = x;
  return y;
}

__device__ inline void front(Lane<float> &L, const Consts<float> &C, bool agc_on, float xin, float &agc_out) {
  float xs = xin;
  if (agc_on) {
    xs = xin * L.agc_gain;
    float level = __builtin_fabsf(xs);
    float target = 0.5f * __builtin_amdgcn_rcpf(level);
    float rate = level > 0.5f ? C.agc_att : C.agc_rel;
    float g = __builtin_fmaf(target - L.agc_gain, rate, L.agc_gain);
    g = level > 0.0f ? g : L.agc_gain;  // exact zero holds the gain (fsk.ts:67)
    L.agc_gain = __builtin_fminf(__builtin_fmaxf(g, 0.1f), 10.0f);
  }
  agc_out = xs;
  // band-pass: y = b0*(x - x2) - a1*y1 - a2*y2
  float y = C.bp_b0 * (xs - L.bp_x2);
  y = __builtin_fmaf(-C.bp_a1, L.bp_y1, y);
  y = __builtin_fmaf(-C.bp_a2, L.bp_y2, y);
  L.bp_x2 = L.bp_x1; L.bp_x1 = xs;
  L.bp_y2 = L.bp_y1; L.bp_y1 = y;
  // NCO: phase in turns = top 32 bits of the accumulator
  float turns = (float)L.nco_hi * 2.3283064365386963e-10f;  // 2^-32
  float c = __builtin_amdgcn_cosf(turns);
  float s = __builtin_amdgcn_sinf(turns);
  uint32_t lo = L.nco_lo + C.inc_lo;
  L.nco_hi = L.nco_hi + C.inc_hi + (lo < L.nco_lo ? 1u : 0u);
  L.nco_lo = lo;
  float fi = lp32(C.lp_b0, C.lp_a2, C.lp_delta, L.li_x1, L.li_x2, L.li_y1, L.li_y2, y * c);
  float fq = lp32(C.lp_b0, C.lp_a2, C.lp_delta, L.lq_x1, L.lq_x2, L.lq_y1, L.lq_y2, y * s);
  L.acc_i += fi;
  L.acc_q += fq;
}

__device__ inline uint32_t discriminate(Lane<float> &L, const Consts<float> &C, float &amp, float &post) {
  const float PI = 3.14159265358979323846f;
  // + 0.0f canonicalises -0 to +0: the reference's averages are never -0 (sums start at +0),
  // and atan2(+0, -0) would be pi instead of 0 on an all-zero input
  float avg_i = L.acc_i * 0.5f + 0.0f;
  float avg_q = L.acc_q * 0.5f + 0.0f;
  float phase = atan2f(avg_q, avg_i);
  amp = __builtin_sqrtf(__builtin_fmaf(avg_i, avg_i, avg_q * avg_q));
  float dphi = phase - L.last_phase;
  if (dphi > PI) dphi -= 2.0f * PI;
  else if (dphi < -PI) dphi += 2.0f * PI;
  L.last_phase = phase;
  float f = lp32(C.lp_b0, C.lp_a2, C.lp_delta, L.po_x1, L.po_x2, L.po_y1, L.po_y2, dphi);
  L.acc_i = 0.0f; L.acc_q = 0.0f;
  post = f;
  return f > 0.0f ? 1u : 0u;
}

__device__ inline void nco_reset(Lane<float> &L) { L.nco_lo = 0; L.nco_hi = 0; }

// ------------------------------------------------------------------------------------------------
// frame state machine (precision independent except for the amplitude compare)
// ------------------------------------------------------------------------------------------------

// resetState() fsk.ts:175-188.  Not touched: AGC gain, pre-filter, both rings (and therefore
// `matched`, the polyphase registers, ring_len, amp_pos/len), silence threshold.
template <typename Real>
__device__ inline void reset_state(Lane<Real> &L) {
  nco_reset(L);
  L.last_phase = (Real)0;
  L.gsc = 0; L.cad_ctr = 0; L.bit_sample_ctr = 0; L.bit_acc = 0; L.bit_cnt = 0; L.next_bit_idx = 0;
  L.byte_cur = 0; L.bit_pos = 0;
  L.started = 0;
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

// processByte fsk.ts:346-375
template <typename Real>
__device__ inline void process_byte(Lane<Real> &L, const DemodParams &P, uint32_t bit, OutCtx &O, bool valid) {
  uint32_t pos = L.bit_pos;
  if (pos == 0) {
    if (bit != 0) { reset_state(L); return; }
  } else if (pos <= 8) {
    L.byte_cur |= bit << (8 - pos);
  } else if (P.parity_on && pos == 9) {
    // parity bit: not validated by the reference
  } else if (pos == P.stop_pos) {
    if (bit != 1) { L.started = 0; return; }
    if (valid && O.out_cnt < O.out_pitch) O.out_row[O.out_cnt] = (uint8_t)L.byte_cur;
    O.out_cnt++;
    L.byte_cur = 0;
    L.bit_pos = 0;
    return;
  } else {
    L.started = 0;
    return;
  }
  L.bit_pos = pos + 1;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// processDownsampledBit fsk.ts:278-344 for every lane with act set.  Must be called by the
// whole wave (it contains a wave-cooperative amplitude-ring read at sync time).
__device__ inline uint32_t popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
__device__ inline uint32_t popc(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }

template <typename Real, typename PolyT, bool FRAC>
__device__ inline void downsampled_bit(Lane<Real> &L, const DemodParams &P, const DemodState &S,
                                       PolyT *poly, PolyT *poly_u, uint32_t lane, uint32_t row, bool valid,
                                       bool act, uint32_t bit, Real amp, OutCtx &O) {
  const PolyT pat_q = (PolyT)P.pat_q, pat_mask = (PolyT)P.pat_mask;
  bool sync_now = false;
  if (act) {
    // syncSamplesBuffer.put(bit): polyphase register of this push slot, newest bit in bit 0
    uint32_t idx = L.poly_phase * 64u + lane;
    PolyT r, u = 0;
    if (FRAC) {
      const bool undef = L.ring_len >= P.ring_int;  // this store is dropped by the reference's ring
      u = (PolyT)(poly_u[idx] << 1) | (PolyT)(undef ? 1u : 0u);
      poly_u[idx] = u;
      r = (PolyT)(poly[idx] << 1) | (PolyT)(undef ? 0u : bit);
    } else {
      r = (PolyT)(poly[idx] << 1) | (PolyT)bit;
    }
    poly[idx] = r;
    L.poly_phase = (L.poly_phase + 1 == P.d) ? 0u : L.poly_phase + 1;
    // slots j = 1..n_bits-1 gain tap j and lose tap j+1 (see file header)
    L.matched += popc((PolyT)(~(r ^ pat_q) & ~u & pat_mask));
    L.matched -= popc((PolyT)(~((r >> 1) ^ pat_q) & ~(u >> 1) & pat_mask));
    if (FRAC) {  // slot 0 compares against `undefined`: it counts undefined taps
      L.matched += (uint32_t)(u & 1u);
      L.matched -= (uint32_t)((u >> 1) & 1u);
    }
    if (L.ring_len < P.ring_cap) L.ring_len++;
    // syncAmplitudeBuffer.put(amp): Float32Array store
    if (valid) S.amp_ring[(size_t)L.amp_pos * P.n_streams + row] = (float)amp;
    L.amp_pos = (L.amp_pos + 1 == P.amp_cap) ? 0u : L.amp_pos + 1;
    if (L.amp_len < P.amp_cap) L.amp_len++;

    L.gsc++;
    L.cad_ctr = (L.cad_ctr + 1 == P.cadence) ? 0u : L.cad_ctr + 1;
    bool eod = false;
    if (amp < L.sil_thr) {
      L.sil_cnt++;
      if (L.sil_cnt >= P.eod_min) {
        O.eod_cnt++;
        L.eod_total++;
        reset_state(L);
        eod = true;
      }
    } else {
      L.sil_cnt = 0;
    }
    if (!eod) {
      if (!L.started) {
        if (L.ring_len >= P.sample_count && P.cadence != 0 && L.cad_ctr == 0 && L.matched >= P.matched_min) {
          L.started = 1;
          L.byte_cur = 0; L.bit_pos = 0;
          L.bit_acc = 0; L.bit_cnt = 0; L.bit_sample_ctr = 0; L.next_bit_idx = 0;
          L.sync_det++;
          sync_now = true;
        }
      } else {
        L.bit_acc += bit;
        L.bit_cnt++;
        L.bit_sample_ctr++;
        if (L.bit_sample_ctr >= L.next_bit_idx) {
          uint32_t b = (2 * L.bit_acc > L.bit_cnt) ? 1u : 0u;
          L.bit_acc = 0; L.bit_cnt = 0;
          L.next_bit_idx += P.d;
          process_byte(L, P, b, O, valid);
        }
      }
    }
  }
  // silence.threshold = mean(syncAmplitudeBuffer) * 0.1 (fsk.ts:321-326), wave-cooperative:
  // the 64 lanes read the syncing stream's ring column together and tree-reduce in f64.
  uint64_t m = __ballot(sync_now);
  if (m) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's ring stores have reached L2
    while (m) {
      int src = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      uint32_t srow = __shfl(row, src, 64);
      uint32_t slen = __shfl(L.amp_len, src, 64);
      double part = 0.0;
      for (uint32_t i = lane; i < slen; i += 64) {
        const float *p = S.amp_ring + (size_t)i * P.n_streams + srow;
        part += (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L1 bypass
      }
      double sum = wave_sum(part);
      if ((int)lane == src) L.sil_thr = (Real)((sum / (double)slen) * 0.1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
template <typename Real, typename PolyT, bool FRAC, bool UNI>
__global__ __launch_bounds__(64) void demod_kernel(DemodParams P, DemodState S, float *__restrict__ samples,
                                                   size_t n, size_t pitch, int vec_ok, int writeback,
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
  Consts<Real> C;
  C.init(P, S, row);
  for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
  if (FRAC)
    for (uint32_t p = 0; p < P.d; p++) poly_u[p * 64u + lane] = gpoly_u[p * 64u + lane];
  const bool tracing = (S.trace_stream >> 6) == blockIdx.x;  // wave-uniform

  OutCtx O;
  O.out_row = out + (size_t)row * out_pitch;
  O.out_pitch = (uint32_t)out_pitch;
  O.out_cnt = 0;
  O.eod_cnt = 0;

  // tile prefetch: instruction i covers rows 8i..8i+7, lane -> (row 8i + lane/8, chunk lane%8)
  const uint32_t sub_row = lane >> 3, chunk = lane & 7;
  float4 pre[8];
  auto load_tile = [&](size_t t0) {
    const bool full = vec_ok && (t0 + kTile <= n);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t r = blockIdx.x * 64u + 8u * i + sub_row;
      r = r < P.n_streams ? r : P.n_streams - 1;
      const float *src = samples + (size_t)r * pitch + t0 + 4u * chunk;
      if (full) {
        pre[i] = *reinterpret_cast<const float4 *>(src);
      } else {
        size_t c0 = t0 + 4u * chunk;
        float4 v;
        v.x = c0 + 0 < n ? src[0] : 0.0f;
        v.y = c0 + 1 < n ? src[1] : 0.0f;
        v.z = c0 + 2 < n ? src[2] : 0.0f;
        v.w = c0 + 3 < n ? src[3] : 0.0f;
        pre[i] = v;
      }
    }
  };

  const bool agc_on = P.agc_on != 0;
  if (n > 0) load_tile(0);
  for (size_t t0 = 0; t0 < n; t0 += kTile) {
    __syncthreads();  // single-wave workgroup: orders last tile's LDS reads before the overwrite
#pragma unroll
    for (int i = 0; i < 8; i++) stage[chunk * kSlotStride + 8u * i + sub_row] = pre[i];
    __syncthreads();
    if (t0 + kTile < n) load_tile(t0 + kTile);

    const uint32_t tile_len = (uint32_t)((n - t0) < (size_t)kTile ? (n - t0) : (size_t)kTile);
    const uint32_t n_chunks = (tile_len + 3u) >> 2;
    for (uint32_t c = 0; c < n_chunks; c++) {
      float4 v4 = stage[c * kSlotStride + lane];
      float xv[4] = {v4.x, v4.y, v4.z, v4.w};
      float wb[4];
      const uint32_t lim = tile_len - 4u * c < 4u ? tile_len - 4u * c : 4u;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if ((uint32_t)k < lim) {
          front(L, C, agc_on, xv[k], wb[k]);
          L.ds_cnt++;
          bool dec = L.ds_cnt >= 2;
          bool any = UNI ? (bool)__builtin_amdgcn_readfirstlane((int)dec) : (__ballot(dec) != 0);
          if (any) {
            Real amp = (Real)0, post = (Real)0;
            uint32_t bit = 0;
            if (UNI || dec) {
              bit = discriminate(L, C, amp, post);
              L.ds_cnt = 0;
            }
            if (tracing && stream == S.trace_stream && (UNI || dec)) {
              uint32_t k = *S.trace_n;
              if (k < S.trace_cap) {
                S.trace_amp[k] = (double)amp;
                S.trace_post[k] = (double)post;
                S.trace_bit[k] = (uint8_t)bit;
              }
              *S.trace_n = k + 1;
            }
            downsampled_bit<Real, PolyT, FRAC>(L, P, S, poly, poly_u, lane, row, valid, UNI || dec, bit, amp, O);
          }
        }
      }
      if (writeback && valid) {
        float *dst = samples + (size_t)row * pitch + t0 + 4u * c;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if ((uint32_t)k < lim) dst[k] = wb[k];
      }
    }
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

size_t demod_lds_bytes(const DemodParams &P) {
  const size_t reg = (P.wide ? sizeof(uint64_t) : sizeof(uint32_t)) * 64u * P.d;
  return sizeof(float4) * kChunks * kSlotStride + reg * (P.frac ? 2u : 1u);
}

// kernel variants: Real x {u32, u64, u64+frac} x {uniform, per-lane decimator phase}
#define FSK_FOR_ALL_VARIANTS(X)                                                                    \
  X(float, uint32_t, false, true) X(float, uint32_t, false, false)                                 \
  X(float, uint64_t, false, true) X(float, uint64_t, false, false)                                 \
  X(float, uint64_t, true, true) X(float, uint64_t, true, false)                                   \
  X(double, uint32_t, false, true) X(double, uint32_t, false, false)                               \
  X(double, uint64_t, false, true) X(double, uint64_t, false, false)                               \
  X(double, uint64_t, true, true) X(double, uint64_t, true, false)

// Host-side launcher (called from fsk_api.hip).  uniform_ds: every stream's downsample.counter is
// equal (true unless single streams were reset at odd sample positions).
hipError_t launch_demod(int precision, bool uniform_ds, bool writeback, const DemodParams &P,
                        const DemodState &S, float *samples, size_t n, size_t pitch, uint8_t *out,
                        size_t out_pitch, uint32_t *out_counts, uint32_t *eod_counts,
                        hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds_bytes = demod_lds_bytes(P);
  const int vec_ok = (pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(samples) & 15u) == 0);
  const int wb = writeback ? 1 : 0;
  const bool f64 = precision != 0, wide = P.wide != 0, frac = P.frac != 0;
  dim3 g(blocks), b(64);
#define FSK_LAUNCH(R, T, F, U)                                                                     \
  if (f64 == (sizeof(R) == 8) && wide == (sizeof(T) == 8) && frac == F && uniform_ds == U)         \
    hipLaunchKernelGGL((demod_kernel<R, T, F, U>), g, b, lds_bytes, stream, P, S, samples, n,      \
                       pitch, vec_ok, wb, out, out_pitch, out_counts, eod_counts);
  FSK_FOR_ALL_VARIANTS(FSK_LAUNCH)
#undef FSK_LAUNCH
  return hipGetLastError();
}

hipError_t set_demod_lds_limit(size_t lds_bytes) {
  hipError_t e = hipSuccess;
#define FSK_ATTR(R, T, F, U)                                                                       \
  if (e == hipSuccess)                                                                             \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_kernel<R, T, F, U>),             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  FSK_FOR_ALL_VARIANTS(FSK_ATTR)
#undef FSK_ATTR
  return e;
}

}  // namespace fsk
