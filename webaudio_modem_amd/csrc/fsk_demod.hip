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
#include "fsk_f64math.h"
#include "fsk_wait.h"

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
  double cw, sw;        // e^{j omega}: one NCO step as a rotation (mix_lp)
  bool omega_small;     // omega < 2 pi: the reference's `% (2 pi)` is one conditional subtraction
  __device__ void init(const DemodParams &P, const DemodState &S, uint32_t row) {
    lp_b0 = P.lp_b0; lp_b1 = P.lp_b1; lp_b2 = P.lp_b2; lp_a1 = P.lp_a1; lp_a2 = P.lp_a2;
    agc_att = P.agc_attack; agc_rel = P.agc_release;
    size_t n = P.n_streams;
    bp_b0 = S.coef[(size_t)CF_bp_b0 * n + row];
    bp_a1 = S.coef[(size_t)CF_bp_a1 * n + row];
    bp_a2 = S.coef[(size_t)CF_bp_a2 * n + row];
    omega = S.coef[(size_t)CF_omega * n + row];
    cw = S.coef[(size_t)CF_w1_re * n + row]; sw = S.coef[(size_t)CF_w1_im * n + row];
    omega_small = omega >= 0.0 && omega < 6.283185307179586476925;
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

// ---- fp32 engines: state arrays <-> the reference's own state ---------------------------------------------------
// Between launches an fp32 engine keeps its I/Q low-pass in the free-running frame of fsk_pipe.hip (li_*/lq_* = the
// never-reset filters, fr_* = NCO phase minus that frame's phase, zq_*/zd_*/zr_dph = what the last resetState() took
// away, last_phase in that frame).  This kernel runs the reference's sample-serial order with real resets, so it
// converts: on load  state = e^{j theta} (free - correction), theta = 2 pi fr / 2^64; on store the inverse with the
// correction folded in (zq = 0).  f64 rotation; the whole-tile kernels themselves never convert.
struct PipeFrame { uint64_t free0; };

__device__ inline void rot2(double c, double s, double a, double b, float &ra, float &rb) {   // (a + jb)(c + js)
  ra = (float)(a * c - b * s);
  rb = (float)(b * c + a * s);
}
__device__ inline void pipe_to_actual(Lane<float> &L, const DemodParams &P, const DemodState &S, size_t n, uint32_t row,
                                      PipeFrame &fr, uint64_t inc) {
  const float *rs = (const float *)S.rs;
#define RQ(f) rs[(size_t)RF_##f * n + row]
#define IQ(f) S.is[(size_t)IF_##f * n + row]
  const uint64_t off = ((uint64_t)IQ(fr_hi) << 32) | IQ(fr_lo);
  const uint64_t acc = ((uint64_t)L.nco_hi << 32) | L.nco_lo;
  fr.free0 = acc - off;
  const double th = (double)off * 5.42101086242752217e-20 * 6.283185307179586476925;   // 2^-64 turns -> radians
  const double c = cos(th), sn = sin(th);
  double x1i, x1q, x2i, x2q, yi, yq, vi, vq;
  double ai = L.acc_i, aq = L.acc_q;               // an open decimator pair's first low-pass outputs (free-running frame)
  const bool open_pair = L.ds_cnt == 1u;           // the previous call ended between the two samples of a decimator pair
  const double a2 = P.lp_a2, delta = 1.0 + P.lp_a1 + P.lp_a2;
  if (IQ(zr_dph) < kDirectPairs) {   // the direct instance IS the state
    x1i = RQ(zd_ix1); x1q = RQ(zd_qx1); x2i = RQ(zd_ix2); x2q = RQ(zd_qx2);
    yi = RQ(zd_iy); yq = RQ(zd_qy); vi = RQ(zd_iv); vq = RQ(zd_qv);
    if (open_pair) {
      // ... but back_pair feeds it a pair's two samples when the pair closes: the open pair's first sample (pre-filter
      // output bp_y1, mixed with the free-running NCO one step back) is still missing, and the pair's partial sums must be
      // this instance's, not the free-running filters' (a stream handed to this kernel mid-pair within
      // kDirectPairs decimated samples of a resetState() got a wrong I/Q pair and a lasting low-pass state error)
      const double tp = (double)(fr.free0 - inc) * 5.42101086242752217e-20 * 6.283185307179586476925;
      const double y = (double)L.bp_y1, mi = y * cos(tp), mq = y * sin(tp);
      const double ti = 2.0 * x1i + mi + x2i, tq = 2.0 * x1q + mq + x2q;
      vi = a2 * vi + (ti - delta * yi); vq = a2 * vq + (tq - delta * yq);
      yi += vi; yq += vq;
      x2i = x1i; x1i = mi; x2q = x1q; x1q = mq;
      ai = yi; aq = yq;
    }
  } else {                 // free-running state minus the zero-input response (its x history is zero by now)
    const float qai = RQ(zq_ai), qaq = RQ(zq_aq), qbi = RQ(zq_bi), qbq = RQ(zq_bq);
    x1i = L.li_x1; x1q = L.lq_x1; x2i = L.li_x2; x2q = L.lq_x2;
    // (y, v) of the response at the even sample in front of the upcoming pair ...
    double zyi = __builtin_fmaf(P.z_yb, qbi, P.z_ya * qai), zyq = __builtin_fmaf(P.z_yb, qbq, P.z_ya * qaq);
    double zvi = __builtin_fmaf(P.z_vb, qbi, P.z_va * qai), zvq = __builtin_fmaf(P.z_vb, qbq, P.z_va * qaq);
    if (open_pair) {       // ... one homogeneous step later when that pair's first sample has been taken already; its
      zvi = a2 * zvi - delta * zyi; zvq = a2 * zvq - delta * zyq;   // output there also sits in the pair's partial sums
      zyi += zvi; zyq += zvq;
      ai -= zyi; aq -= zyq;
    }
    yi = L.li_y1 - zyi; yq = L.lq_y1 - zyq;
    vi = L.li_y2 - zvi; vq = L.lq_y2 - zvq;
  }
  // fsk_pipe.hip runs the branch from the pre-filter output on 2^60 times the reference's size (kIqScale there)
  const double un = 8.67361737988403547e-19;       // 2^-60
  L.bp_y1 *= 8.67361737988403547e-19f; L.bp_y2 *= 8.67361737988403547e-19f;
  rot2(c, sn, x1i * un, x1q * un, L.li_x1, L.lq_x1);
  rot2(c, sn, x2i * un, x2q * un, L.li_x2, L.lq_x2);
  rot2(c, sn, yi * un, yq * un, L.li_y1, L.lq_y1);
  rot2(c, sn, vi * un, vq * un, L.li_y2, L.lq_y2);
  rot2(c, sn, ai * un, aq * un, L.acc_i, L.acc_q);
  double r = (double)L.last_phase + th;            // th in [0, 2 pi)
  r = r > 3.14159265358979323846 ? r - 6.283185307179586476925 : r;
  L.last_phase = (float)r;
#undef RQ
#undef IQ
}
// n_done: samples this launch processed (every stream the same number)
__device__ inline void actual_to_pipe(Lane<float> &L, const DemodState &S, size_t n, uint32_t row, const PipeFrame &fr,
                                      uint64_t inc, size_t n_done) {
  float *rs = (float *)S.rs;
  const uint64_t acc = ((uint64_t)L.nco_hi << 32) | L.nco_lo;
  const uint64_t off = acc - (fr.free0 + inc * (uint64_t)n_done);
  const double th = (double)off * 5.42101086242752217e-20 * 6.283185307179586476925;
  const double c = cos(th), sn = -sin(th);         // rotate back by -theta
  const double up = 1152921504606846976.0;         // 2^60 (kIqScale of fsk_pipe.hip)
  const double x1i = up * L.li_x1, x1q = up * L.lq_x1, x2i = up * L.li_x2, x2q = up * L.lq_x2, yi = up * L.li_y1, yq = up * L.lq_y1,
               vi = up * L.li_y2, vq = up * L.lq_y2, ai = up * L.acc_i, aq = up * L.acc_q;
  L.bp_y1 *= 1152921504606846976.0f; L.bp_y2 *= 1152921504606846976.0f;
  rot2(c, sn, x1i, x1q, L.li_x1, L.lq_x1);
  rot2(c, sn, x2i, x2q, L.li_x2, L.lq_x2);
  rot2(c, sn, yi, yq, L.li_y1, L.lq_y1);
  rot2(c, sn, vi, vq, L.li_y2, L.lq_y2);
  rot2(c, sn, ai, aq, L.acc_i, L.acc_q);
  double r = (double)L.last_phase - th;
  r = r < -3.14159265358979323846 ? r + 6.283185307179586476925 : r;
  L.last_phase = (float)r;
  S.is[(size_t)IF_fr_lo * n + row] = (uint32_t)off;
  S.is[(size_t)IF_fr_hi * n + row] = (uint32_t)(off >> 32);
  S.is[(size_t)IF_zr_dph * n + row] = kHandPairs;
  const int zf[] = {RF_zq_ai, RF_zq_aq, RF_zq_bi, RF_zq_bq, RF_zq_0i, RF_zq_0q, RF_zd_ix1, RF_zd_ix2, RF_zd_iy, RF_zd_iv,
                    RF_zd_qx1, RF_zd_qx2, RF_zd_qy, RF_zd_qv};
  for (int f : zf) rs[(size_t)f * n + row] = 0.0f;
}
__device__ inline void pipe_to_actual(Lane<double> &, const DemodParams &, const DemodState &, size_t, uint32_t, PipeFrame &, uint64_t) {}
__device__ inline void actual_to_pipe(Lane<double> &, const DemodState &, size_t, uint32_t, const PipeFrame &, uint64_t, size_t) {}

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
  // Always executed: with the AGC disabled the host sets both rates to 0 and the gain stays 1, which makes this block an
  // exact no-op (x * 1; g + (t - g) * 0 = g; clamp(1) = 1) -- and one division and two selects instead of the reference's
  // two branches: the same operations on the same values in every case.  The divisor is made 1 where the level is 0 (the
  // result is discarded there) so that the compiler has no unused quotient to jump around: a branch instruction, taken or
  // not, costs a lone wave ~35 cycles, and this chain is the kernel's longest per sample.
  (void)agc_on;
  const float xs = (float)((double)xin * L.agc_gain);  // samples[i] *= gain : Float32Array store
  {
    const double level = fabs((double)xs);
    const double target = 0.5 / (level > 0.0 ? level : 1.0);
    const double rate = level > 0.5 ? C.agc_att : C.agc_rel;
    double gn = L.agc_gain + (target - L.agc_gain) * rate;
    asm volatile("" : "+v"(gn));   // (evaluated for every lane: without this hipcc sinks the whole update into an `if (level > 0)`)
    L.agc_gain = level > 0.0 ? gn : L.agc_gain;
    const double g = L.agc_gain < 10.0 ? L.agc_gain : 10.0;
    L.agc_gain = g > 0.1 ? g : 0.1;
  }
  agc_out = xs;
  // preFilter.processBuffer: f64 state, f32 result (filters.ts:81-87)
  // (b1 = 0: the reference's `out += 0 * x[n-1]` adds +-0 to a sum that is never -0 -- it starts from +0 -- and leaves it
  // unchanged for every finite x[n-1], so the term's multiply and add are not issued)
  {
    const double x = (double)xs;
    double out = 0.0;
    out += C.bp_b0 * x;
    out += -C.bp_b0 * L.bp_x2;
    out -= C.bp_a1 * L.bp_y1;
    out -= C.bp_a2 * L.bp_y2;
    L.bp_x2 = L.bp_x1; L.bp_x1 = x;
    L.bp_y2 = L.bp_y1; L.bp_y1 = out;
    return (float)out;
  }
}

// NCO mix + I/Q low-pass for one pre-filtered sample (fsk.ts:228-238): the part resetState() zeroes.
//
// The NCO.  The reference evaluates Math.cos / Math.sin of localOscPhase and advances it by `(phase + omega) % (2 pi)`
// every sample.  The phase itself is kept exactly as the reference has it (for 0 <= omega < 2 pi the `%` is one
// conditional subtraction, exact by Sterbenz' lemma, so nco_phase is bit for bit the reference's at every sample); its
// cosine and sine are carried as a phasor that is ROTATED by e^{j omega} each sample (four FMAs), re-evaluated from the
// exact phase by nco_refresh() wherever the stream's ABSOLUTE sample count (samples since the engine was created: the same
// for every stream) is a multiple of 32, and set to exactly (1, 0) by a reset.  The phasor is part of the stream's state
// (nco_c, nco_s: round 6), so its value at any sample is a function of the samples before it alone: cutting a stream into
// calls anywhere changes no fp64 intermediate, bit for bit (rounds 1-5 refreshed every 32 samples counted from the start of
// each call and did not store the phasor: intermediates could differ by ~1e-14 between two cuts).  Between two
// refreshes the phasor drifts from the true cos / sin of nco_phase by at most 32 x (the `+`'s rounding 4.4e-16 + the
// rotation's 2e-16) = 2e-14, against the 1e-12 the fp64 intermediates are held to; evaluating both functions afresh per
// sample (the device library's: ~150 instructions and a dozen branches) was 60 % of this kernel's time.
__device__ inline void nco_refresh(Lane<double> &L, const Consts<double> &C) {
  if (C.omega_small) sincos_0_2pi(L.nco_phase, L.nco_c, L.nco_s);
}
__device__ inline void nco_refresh(Lane<float> &, const Consts<float> &) {}
__device__ inline bool nco_rot_ok(const Consts<double> &C) { return __builtin_amdgcn_ballot_w64(!C.omega_small) == 0ull; }
__device__ inline bool nco_rot_ok(const Consts<float> &) { return true; }
template <bool ROT = false>
__device__ inline void mix_lp(Lane<double> &L, const Consts<double> &C, float pre, double &fi, double &fq) {
  double s = (double)pre;
  double ci, cq;
  if (ROT || C.omega_small) {   // (ROT: the caller has checked omega_small for the whole wave)
    ci = s * L.nco_c;
    cq = s * L.nco_s;
    const double t = L.nco_phase + C.omega;                       // in [0, 4 pi)
    L.nco_phase = t >= 2.0 * 3.14159265358979323846 ? t - 2.0 * 3.14159265358979323846 : t;   // == fmod(t, 2 pi)
    const double nc = __builtin_fma(L.nco_c, C.cw, -(L.nco_s * C.sw));
    const double ns = __builtin_fma(L.nco_s, C.cw, L.nco_c * C.sw);
    L.nco_c = nc; L.nco_s = ns;
  } else {                                                        // (a centre frequency above the sample rate)
    ci = s * cos(L.nco_phase);
    cq = s * sin(L.nco_phase);
    L.nco_phase = fmod(L.nco_phase + C.omega, 2.0 * 3.14159265358979323846);
  }
  fi = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.li_x1, L.li_x2, L.li_y1, L.li_y2, ci);
  fq = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.lq_x1, L.lq_x2, L.lq_y1, L.lq_y2, cq);
}

// decimated-rate discriminator (fsk.ts:245-264): returns the slicer bit
__device__ inline bool discriminate(Lane<double> &L, const Consts<double> &C, double sum_i, double sum_q,
                                    double &amp, double &post) {
  const double PI = 3.14159265358979323846;
  double avg_i = sum_i / 2.0;
  double avg_q = sum_q / 2.0;
  double phase = atan2_lean(avg_q, avg_i);     // (fsk_f64math.h: within 1.5 ulp of Math.atan2)
  amp = sqrt(avg_i * avg_i + avg_q * avg_q);
  double dphi = phase - L.last_phase;
  {
    // fsk.ts:255-257 as two selects (both candidates evaluated for every lane: the `if / else if` compiled to two exec-masked
    // branches per decimated sample, ~35 cycles each for a lone wave)
    double dm = dphi - 2.0 * PI, dp = dphi + 2.0 * PI;
    asm volatile("" : "+v"(dm), "+v"(dp));
    dphi = dphi > PI ? dm : (dphi < -PI ? dp : dphi);
  }
  L.last_phase = phase;
  double f = biquad64(C.lp_b0, C.lp_b1, C.lp_b2, C.lp_a1, C.lp_a2, L.po_x1, L.po_x2, L.po_y1, L.po_y2, dphi);
  post = f;
  return f > 0.0;
}

__device__ inline void nco_reset(Lane<double> &L) { L.nco_phase = 0.0; L.nco_c = 1.0; L.nco_s = 0.0; }

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

template <bool ROT = false>
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
                                       Real amp, OutCtx &O, Real post = (Real)0) {
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
    if (valid) S.amp_ring[amp_index(R.amp_pos, row, P.n_streams)] = (float)amp;
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

  // Control flow: TWO wave-uniform tests per decimated sample -- "something rare" (an 'eod', a sync candidate) and "a
  // decision with consequences" (a byte completes, a bad start bit, a parked clock) -- around straight-line code: with one
  // wave per SIMD (fp64 at BASELINE config #3's size) every branch instruction, taken or not, stalls the wave ~35 cycles
  // (profiles/r03_valu_probe_ctl.txt), and the five separate tests this function used to make were a third of the
  // fp64 kernel's time.
  const bool sync_now = cand & !eod;
  if (__ballot(eod | sync_now)) {
    // ---- rare path 1: end of data (fsk.ts:288-291) ----------------------------------------------
    if (__ballot(eod)) {
      if (P.quality) {   // opt-in estimates: the noise floor of the silence that caused the first 'eod' after a sync
        const uint32_t pushes = amp_base + R.k;
        quality_on_eod<Real>(P, S, lane, row, eod & valid, R.amp_pos ? R.amp_pos - 1u : P.amp_cap - 1u,
                             pushes < P.amp_cap ? pushes : P.amp_cap);
      }
      if (eod) {
        O.eod_cnt++;
        L.eod_total++;
        reset_state(L, P.matched_min);
        did_reset = true;
      }
    }
    // ---- rare path 2: frame sync found (fsk.ts:315-327) -----------------------------------------
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
          const float *p = S.amp_ring + amp_index(i, srow, P.n_streams);
          part += (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L1 bypass
        }
        const double sum = wave_sum(part);
        if ((int)lane == src) {
          L.sil_thr = (Real)((sum / (double)slen) * 0.1);
          if (P.quality) quality_on_sync<Real>(P, S, row, sum / (double)slen);
        }
      }
    }
  }
  // ---- bit decision + processByte (fsk.ts:335-375): some lane decides at nearly every step, so the decision itself is
  // masked arithmetic for all lanes; only its consequences branch
  const bool dec_now = decide & !eod & (L.started != 0);
  const uint32_t cnt = L.bit_reload - L.bit_wait;               // bitAccumCount
  const uint32_t b = (2u * L.bit_acc > cnt) ? 1u : 0u;          // fsk.ts:336
  const uint32_t q_cnt = cnt, q_ones = L.bit_acc;
  const uint32_t pos = L.bit_pos;
  const bool is_stop = pos == P.stop_pos;
  const bool bad_start = dec_now & (pos == 0) & (b != 0);
  const bool good_start = dec_now & (pos == 0) & (b == 0);
  const bool emit = dec_now & is_stop & (b != 0);
  const bool bad_stop = dec_now & is_stop & (b == 0);           // fsk.ts:363-366
  {
    const uint32_t wait_n = L.bit_wait + P.d;                   // nextBitSampleIndex += dsSPB
    L.bit_acc = dec_now ? 0u : L.bit_acc;
    L.bit_reload = dec_now ? wait_n : L.bit_reload;
    L.bit_wait = dec_now ? (bad_stop ? kBigWait : wait_n) : L.bit_wait;
    // data bits MSB first: positions 1..8 land in bits 7..0; position 0 (start bit, must be 0 to
    // get here) and positions >= 9 land above bit 7 and are masked off when the byte is emitted
    L.byte_cur |= dec_now ? (b << ((8u - pos) & 31u)) : 0u;
    L.bit_pos = dec_now ? ((is_stop & !bad_stop) ? 0u : (bad_stop ? pos : pos + 1u)) : pos;
    L.started = bad_stop ? 0u : L.started;
    L.thr_eff = bad_stop ? P.matched_min : L.thr_eff;
  }
  const bool park = decide & (L.started == 0) & !bad_start;     // bit_wait ran down without a frame (12 h of decimated samples)
  if (__ballot(bad_start | emit | park | (good_start & (P.quality != 0)))) {
    if (__ballot(bad_start)) {
      if (bad_start) { reset_state(L, P.matched_min); did_reset = true; }  // fsk.ts:352-355
    }
    if (P.quality) {
      if (good_start & valid) quality_on_start<Real>(P, S, row, post);
      if (emit & valid) quality_on_byte<Real>(P, S, row, (uint32_t)(uint8_t)L.byte_cur, q_ones, q_cnt, post);
    }
    if (emit) {
      if (valid && O.out_cnt < O.out_pitch) O.out_row[O.out_cnt] = (uint8_t)L.byte_cur;
      O.out_cnt++;
      L.byte_cur = 0;
    }
    // park the clock again
    if (decide & (L.started == 0)) L.bit_wait = kBigWait;
  }
  return did_reset;
}

// ------------------------------------------------------------------------------------------------
// eight decimated samples through the frame state machine at once (fp64 kernels, round 4)
// ------------------------------------------------------------------------------------------------
// processDownsampledBit x 8 + processByte (fsk.ts:278-375) for a block whose amplitudes, post-filter outputs and slicer
// bits (w: sample 1 in bit 7) are there already -- the integer part, with ONE exit test for everything that is not plain
// bit clocking: a possible 'eod' (bounded from above: the run a wholly silent block would end with), a sync candidate at
// any of the eight samples, a bad start or stop bit, a clock that ran down without a frame or stands at zero right after a
// sync.  Returns false, with nothing touched, if the block has to be redone in the per-sample order; otherwise commits:
// polyphase registers, amplitude ring, counters, at most ONE bit decision (they are dsSPB >= 8 decimated samples apart:
// the caller checks) at sample jd = bit_wait on entry, a completed byte.
template <typename Real, typename PolyT, bool TRACE, int NB>
__device__ inline bool block_fsm(Lane<Real> &L, const DemodParams &P, const DemodState &S, PolyT *poly, RingPos &R, OutCtx &O,
                                  uint32_t lane, uint32_t row, uint32_t stream, bool valid, const Real (&amp)[NB], const Real (&post)[NB],
                                  uint32_t w) {
  const PolyT qn = (PolyT)~P.pat_q, mask = (PolyT)P.pat_mask;
  PolyT reg[NB];
  uint32_t ph = R.phase;
  uint32_t matched = L.matched;
  bool rare = false;
  uint32_t last_loud = 0;                                     // 1..8, 0 = none in this block
#pragma unroll
  for (int j = 0; j < NB; j++) {
    const PolyT rold = poly[ph * 64u + lane];
    const PolyT r = (PolyT)(rold << 1) | (PolyT)((w >> (NB - 1 - j)) & 1u);   // syncSamplesBuffer.put(bit)
    reg[j] = r;
    matched += popc((PolyT)((r ^ qn) & mask));
    matched -= popc((PolyT)((rold ^ qn) & mask));
    rare |= matched >= L.thr_eff;                             // a sync candidate (whether on the search cadence: the per-sample path looks)
    last_loud = (amp[j] < L.sil_thr) ? last_loud : (uint32_t)(j + 1);    // fsk.ts:285
    ph = (ph + 1 == P.d) ? 0u : ph + 1;
  }
  // 'eod' (fsk.ts:288): no run inside the block is longer than the one a wholly silent block would end with
  rare |= L.sil_cnt + (uint32_t)NB >= P.eod_min;
  // ---- bit clock (fsk.ts:331-341): the decision, if one falls into this block, at sample jd
  const int32_t wait0 = (int32_t)L.bit_wait;
  const bool started = L.started != 0;
  const bool md = started & (wait0 <= NB);
  rare |= !started & (wait0 <= NB);                            // the clock ran down without a frame (parked again by the per-sample path)
  rare |= started & (wait0 < 1);                              // right after a sync (decided with the first sample; at dsSPB 8 a second decision would follow)
  const uint32_t jd = wait0 < 1 ? 1u : (uint32_t)wait0;
  const uint32_t hi = w >> (((uint32_t)NB - jd) & 31u);                 // bits of samples 1 .. jd
  const uint32_t ones = L.bit_acc + popc(hi & ((1u << NB) - 1u));
  const uint32_t cnt = L.bit_reload - (uint32_t)(wait0 - (int32_t)jd);   // bitAccumCount at the decision
  const uint32_t b = (2u * ones > cnt) ? 1u : 0u;             // fsk.ts:336
  const uint32_t pos = L.bit_pos;
  const bool is_stop = pos == P.stop_pos;
  rare |= md & (((pos == 0) & (b != 0)) | (is_stop & (b == 0)));   // bad start bit (fsk.ts:352-355) / bad stop bit (363-366)
  if (__ballot(rare)) return false;
  // ---- commit
  ph = R.phase;
  uint32_t apos[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    poly[ph * 64u + lane] = reg[j];
    ph = (ph + 1 == P.d) ? 0u : ph + 1;
    apos[j] = R.amp_pos;
    R.amp_pos = (R.amp_pos + 1 == P.amp_cap) ? 0u : R.amp_pos + 1;
    if (TRACE) {
      if (stream == S.trace_stream) {
        const uint32_t kk = *S.trace_n;
        if (kk < S.trace_cap) {
          S.trace_amp[kk] = (double)amp[j];
          S.trace_post[kk] = (double)post[j];
          S.trace_bit[kk] = (uint8_t)((w >> (NB - 1 - j)) & 1u);
        }
        *S.trace_n = kk + 1;
      }
    }
  }
  if (valid) {                                                // syncAmplitudeBuffer.put x 8
#pragma unroll
    for (int j = 0; j < NB; j++) S.amp_ring[amp_index(apos[j], row, P.n_streams)] = (float)amp[j];
  }
  R.phase = ph;
  R.k += NB;
  L.matched = matched;
  L.gsc += NB;
  L.cad_ctr = (L.cad_ctr + (uint32_t)NB) % P.cadence;
  L.sil_cnt = last_loud ? (uint32_t)NB - last_loud : L.sil_cnt + (uint32_t)NB;
  const uint32_t tot = popc(w & ((1u << NB) - 1u)), nhi = popc(hi & ((1u << NB) - 1u));
  const bool emit = md & is_stop;                             // (b == 1: a bad stop bit left through the exit above)
  L.bit_acc = md ? tot - nhi : L.bit_acc + tot;
  L.bit_reload = md ? (uint32_t)(wait0 - (int32_t)jd) + P.d : L.bit_reload;
  L.bit_wait = md ? (uint32_t)(wait0 - NB) + P.d : (uint32_t)(wait0 - NB);
  // data bits MSB first (see downsampled_bit)
  L.byte_cur |= md ? (b << ((8u - pos) & 31u)) : 0u;
  L.bit_pos = md ? (is_stop ? 0u : pos + 1u) : pos;
  if (__ballot(emit)) {
    if (emit) {                                               // fsk.ts:367-368
      if (valid && O.out_cnt < O.out_pitch) O.out_row[O.out_cnt] = (uint8_t)L.byte_cur;
      O.out_cnt++;
      L.byte_cur = 0;
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
// fp32 kernels are capped at 128 VGPRs (4 waves/SIMD): the chain is VALU-bound and one wave per
// SIMD issues a VALU op only every 4 cycles (MI355X_MICROARCH.md), so occupancy is throughput.
// SPLIT2 (round 6, the exact path at batches of one wave per SIMD): TWO waves per 64-stream group.  Wave 0 is the part of the chain
// that resetState() never touches (fsk.ts:175-188) -- tile loads, AGC with its correctly rounded division, pre-filter, the
// Float32Array store of fsk.ts:202 -- and hands the pre-filter's floats to wave 1 through a ring of kPreTiles tiles in LDS; wave 1 is
// everything from the NCO on, exactly the one-wave kernel's code with those floats for input.  Exact by construction: the hand-over
// is the reference's own Float32Array, and nothing behind it feeds back into what is in front of it.  What it buys: a double-
// precision instruction occupies the pipe for 4 cycles while a lone wave issues one instruction per ~5.7 -- two instruction
// streams per SIMD instead of one.
static constexpr uint32_t kPreTiles = 3;
#ifndef FSK_SPLIT2_BLOCK
#define FSK_SPLIT2_BLOCK 8      // decimated samples per block of the two-wave kernel's back wave (4 was tried: 736 instead of 864 bytes of spills per lane, and slower still -- 144 against 156 Gsamples/s)
#endif
template <typename Real, typename PolyT, bool FRAC, bool UNI, bool TRACE, bool SPLIT2 = false>
__global__ __launch_bounds__(SPLIT2 ? 128 : 64, (sizeof(Real) == 4 ? FSK_F32_WAVES_PER_SIMD : SPLIT2 ? 2 : 1)) void demod_kernel(DemodParams P, DemodState S, float *__restrict__ samples,
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
  // SPLIT2: [kPreTiles][kChunks][64] float4 of pre-filter outputs (lane = stream, four consecutive samples), then two counters
  float4 *pring = reinterpret_cast<float4 *>(reinterpret_cast<char *>(poly) + (((FRAC ? 2u : 1u) * sizeof(PolyT) * 64u * P.d + 15u) & ~(size_t)15u));
  uint32_t *pctr = reinterpret_cast<uint32_t *>(pring + kPreTiles * kChunks * 64);     // [0] tiles produced, [1] tiles consumed
  const uint32_t wave2 = SPLIT2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const bool valid = stream < P.n_streams;
  const uint32_t row = valid ? stream : P.n_streams - 1;
  const size_t ns = P.n_streams;

  Lane<Real> L;
  load_lane(L, S, ns, row);
  PipeFrame frame{0};
  if (n > 0) pipe_to_actual(L, P, S, ns, row, frame, S.nco_inc[row]);
  L.thr_eff = L.started ? 0xFFFFFFFFu : P.matched_min;
  Consts<Real> C;
  C.init(P, S, row);
  if (!SPLIT2 || wave2 == 1u) {
    for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
    if (FRAC)
      for (uint32_t p = 0; p < P.d; p++) poly_u[p * 64u + lane] = gpoly_u[p * 64u + lane];
  }
  if (SPLIT2) {
    if (threadIdx.x == 0) { pctr[0] = 0u; pctr[1] = 0u; }
    __syncthreads();
  }

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

  // (SPLIT2, wave 1: the input IS the pre-filter's output)
  auto pre_in = [&](float x, float &wbv) -> float {
    if (SPLIT2) { wbv = x; return x; }
    return pre_stage(L, C, agc_on, x, wbv);
  };
  // generic path: one sample at a time, per-lane decimator phase (fsk.ts:224-276 as written)
  auto step_generic = [&](float x, float &wbv) {
    float pre_y = pre_in(x, wbv);
    if (TRACE) { if (stream == S.trace_stream) trace_pre_put(S, (double)pre_y); }
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
                                         UNI || dec, bit, amp, O, post);
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
      pre_y[k] = pre_in(x[k], wbv[k]);
      mix_lp<true>(L, C, pre_y[k], fi[k], fq[k]);
    }
    if (TRACE) {
      if (stream == S.trace_stream) {
#pragma unroll
        for (int k = 0; k < 4; k++) trace_pre_put(S, (double)pre_y[k]);
      }
    }
    Real amp0, post0, amp1, post1;
    const bool bit0 = discriminate(L, C, fi[0] + fi[1], fq[0] + fq[1], amp0, post0);
    bool bit1 = discriminate(L, C, fi[2] + fi[3], fq[2] + fq[3], amp1, post1);
    // the state machine for both decimated samples.  fp32: one copy run twice (keeps the kernel and its live ranges small:
    // two waves per SIMD); fp64 (one wave per SIMD, registers to spare): two copies, no per-iteration selects
    auto fsm = [&](int p, Real amp, Real post, bool bit) {
      trace_put(amp, post, bit);
      const bool rst = downsampled_bit<Real, PolyT, FRAC>(L, P, S, poly, poly_u, R, need, ring_base, amp_base, lane,
                                                          row, valid, true, bit, amp, O, post);
      if (p == 0 && __ballot(rst)) {
        if (rst) {  // resetState() ran after pair 0: redo the reset-sensitive half of samples 2,3
          Real gi2, gq2, gi3, gq3;
          mix_lp<true>(L, C, pre_y[2], gi2, gq2);
          mix_lp<true>(L, C, pre_y[3], gi3, gq3);
          bit1 = discriminate(L, C, gi2 + gi3, gq2 + gq3, amp1, post1);
        }
      }
    };
    if (sizeof(Real) == 8) {
      fsm(0, amp0, post0, bit0);
      fsm(1, amp1, post1, bit1);
    } else {
#pragma unroll 1
      for (int p = 0; p < 2; p++) fsm(p, p ? amp1 : amp0, p ? post1 : post0, p ? bit1 : bit0);
    }
  };

  // fp64, round 4: SIXTEEN samples at a time with the frame state machine restated per block, as fsk_blk.hip does for the
  // fp32 path.  One wave per SIMD pays ~35 cycles for every branch instruction and exposes every LDS round trip, and the
  // per-sample state machine above makes two wave-uniform tests and an LDS read-modify-write per decimated sample.  Here the
  // sixteen front ends and eight discriminators run as one branch-free block (speculatively: nothing is committed), then the
  // eight decimated samples go through integer logic with ONE exit test for everything that is not plain bit clocking --
  // an 'eod' (bounded from above: the run a wholly silent block would end with), a sync candidate at any of the eight
  // samples, a bad start or stop bit, the clock running down without a frame.  A block that trips it is undone (the lane's
  // state is a copy) and goes through block4 four times: the per-sample order with real resets, unchanged.  The block path
  // itself restates fsk.ts:278-375 for a lane without such an event: at most ONE bit decision falls into a block (they
  // are dsSPB >= 8 decimated samples apart -- the caller checks), at sample jd = bit_wait on entry.
  // Returns false if the block has to be redone.
  // (round 6: NBD decimated samples per block -- eight, or four in the two-wave kernel, whose back wave has 256 registers)
  constexpr int NBD = SPLIT2 ? FSK_SPLIT2_BLOCK : 8;
  auto block16 = [&](const float (&x)[2 * NBD], float (&wbv)[2 * NBD]) -> bool {
    const Lane<Real> L0 = L;                                   // everything this block may touch (poly / amplitude ring / output: written at the end)
    Real amp[NBD], post[NBD];
    float pre16[2 * NBD];                                       // (traced engines: recorded once the block stands)
    uint32_t w = 0;                                             // slicer bits, sample 1 in bit NBD - 1
#pragma unroll
    for (int j = 0; j < NBD; j++) {
      Real fi0, fq0, fi1, fq1;
      const float y0 = pre_in(x[2 * j], wbv[2 * j]);
      mix_lp<true>(L, C, y0, fi0, fq0);
      const float y1 = pre_in(x[2 * j + 1], wbv[2 * j + 1]);
      mix_lp<true>(L, C, y1, fi1, fq1);
      pre16[2 * j] = y0; pre16[2 * j + 1] = y1;
      const bool bit = discriminate(L, C, fi0 + fi1, fq0 + fq1, amp[j], post[j]);
      w = (w << 1) | (bit ? 1u : 0u);
    }
    if (!block_fsm<Real, PolyT, TRACE, NBD>(L, P, S, poly, R, O, lane, row, stream, valid, amp, post, w)) { L = L0; return false; }
    if (TRACE) {
      if (stream == S.trace_stream) {
#pragma unroll
        for (int k = 0; k < 2 * NBD; k++) trace_pre_put(S, (double)pre16[k]);
      }
    }
    return true;
  };

  // (fp64: the four-sample block runs the NCO as a rotation, which needs omega < 2 pi in every lane -- any real configuration)
  const bool fast = UNI && (__builtin_amdgcn_readfirstlane((int)L.ds_cnt) == 0) && nco_rot_ok(C);
  // the sixteen-sample block: fp64 only (registers: one wave per SIMD), at most one bit decision per eight decimated samples, none
  // of the opt-in estimates (they hook the per-sample state machine)
  const bool fast16 = fast && sizeof(Real) == 8 && P.d >= 8 && P.cadence > 0 && !P.quality;
  FSK_WAIT_DECL
  auto ctr_peek = [&](const uint32_t *q) -> uint32_t {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)q) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  };
  auto ctr_post = [&](uint32_t *q, uint32_t v) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)q), "v"(v) : "memory");
  };
  if (SPLIT2 && wave2 == 0u) {
    // ---- wave 0: loads, AGC, pre-filter -> the ring (and the AGC write-back)
    bool cf = n > 0 && tile_is_fast(0);
    if (cf) load_tile_fast(0);
    uint32_t consumed = 0, ti = 0;
    for (size_t t0 = 0; t0 < n; t0 += kTile, ti++) {
      if (cf) {
#pragma unroll
        for (int i = 0; i < kChunks; i++) stage[chunk * kSlotStride + (uint32_t)kRowsPerLoad * i + sub_row] = pre[i];
      } else {
        stage_tile_slow(t0);
      }
      cf = (t0 + kTile < n) && tile_is_fast(t0 + kTile);
      if (cf) load_tile_fast(t0 + kTile);
      FSK_WAIT_BEGIN
      while (ti - consumed >= kPreTiles) {                    // ring full: wave 1 has not released the slot
        consumed = ctr_peek(&pctr[1]);
        if (ti - consumed >= kPreTiles) FSK_SPIN(1, S.blk_stat);
      }
      float4 *dstt = pring + (ti % kPreTiles) * kChunks * 64u;
      const uint32_t tile_len = (uint32_t)((n - t0) < (size_t)kTile ? (n - t0) : (size_t)kTile);
      const uint32_t n_chunks = (tile_len + 3u) >> 2;
      for (uint32_t c = 0; c < n_chunks; c++) {
        const float4 v4 = stage[c * kSlotStride + lane];
        const float xv[4] = {v4.x, v4.y, v4.z, v4.w};
        float y[4], wb[4];
        const uint32_t lim = tile_len - 4u * c < 4u ? tile_len - 4u * c : 4u;
#pragma unroll
        for (int k = 0; k < 4; k++) {                         // (a ragged last chunk: samples beyond the call's end are not there)
          y[k] = 0.f; wb[k] = 0.f;
          if ((uint32_t)k < lim) y[k] = pre_stage(L, C, agc_on, xv[k], wb[k]);
        }
        dstt[c * 64u + lane] = make_float4(y[0], y[1], y[2], y[3]);
        if (writeback && valid) {
          float *dst = samples + (size_t)row * pitch + t0 + 4u * c;
#pragma unroll
          for (int k = 0; k < 4; k++)
            if ((uint32_t)k < lim) dst[k] = wb[k];
        }
      }
      ctr_post(&pctr[0], ti + 1u);
    }
  } else {
  bool cur_fast = !SPLIT2 && n > 0 && tile_is_fast(0);
  if (cur_fast) load_tile_fast(0);
  uint32_t produced2 = 0, ti2 = 0;
  // fp64: the NCO phasor afresh from the exact phase (mix_lp) where the stream's absolute sample count is a multiple of 32 -- sample
  // nco_r0 of every tile of this call (tiles are 32 samples; P.nco_anchor = samples the engine had taken before the call, mod 32)
  static_assert(kTile == 32, "the NCO refresh period is a tile");
  const uint32_t nco_r0 = sizeof(Real) == 8 ? ((32u - (P.nco_anchor & 31u)) & 31u) : 0u;
  for (size_t t0 = 0; t0 < n; t0 += kTile) {
    if (nco_r0 == 0u) nco_refresh(L, C);
    const float4 *tin = stage;                              // this tile's input: [chunk][tstride] float4, lane = stream
    uint32_t tstride = kSlotStride;
    if (SPLIT2) {
      FSK_WAIT_BEGIN
      while (produced2 <= ti2) {                            // wave 0's tile
        produced2 = ctr_peek(&pctr[0]);
        if (produced2 <= ti2) FSK_SPIN(1, S.blk_stat);
      }
      tin = pring + (ti2 % kPreTiles) * kChunks * 64u;
      tstride = 64u;
    } else {
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
    }

    const uint32_t tile_len = (uint32_t)((n - t0) < (size_t)kTile ? (n - t0) : (size_t)kTile);
    const uint32_t n_chunks = (tile_len + 3u) >> 2;
    for (uint32_t c = 0; c < n_chunks; c++) {
      float4 v4 = tin[c * tstride + lane];
      float xv[4] = {v4.x, v4.y, v4.z, v4.w};
      float wb[4];
      const uint32_t lim = tile_len - 4u * c < 4u ? tile_len - 4u * c : 4u;
      // (a call that does not start on the refresh grid -- earlier calls of lengths that are no multiples of 32 -- has its refresh
      // point inside the tile: at the start of a block where it can be, else that block goes chunk by chunk / sample by sample)
      const uint32_t s0 = 4u * c;
      if (sizeof(Real) == 8 && nco_r0 != 0u && nco_r0 == s0) nco_refresh(L, C);
      constexpr uint32_t BS = 2u * (uint32_t)NBD, CB = BS / 4u;   // samples / four-sample chunks per block
      const bool r_in16 = sizeof(Real) == 8 && nco_r0 > s0 && nco_r0 < s0 + BS;
      const bool r_in4 = sizeof(Real) == 8 && nco_r0 > s0 && nco_r0 < s0 + 4u;
      if (sizeof(Real) == 8 && !FRAC && fast16 && (c & (CB - 1u)) == 0u && 4u * c + BS <= tile_len && !r_in16) {
        float x16[BS], wb16[BS];
#pragma unroll
        for (int q = 0; q < (int)CB; q++) {
          const float4 u4 = tin[(c + q) * tstride + lane];
          x16[4 * q] = u4.x; x16[4 * q + 1] = u4.y; x16[4 * q + 2] = u4.z; x16[4 * q + 3] = u4.w;
        }
        if (!block16(x16, wb16)) {                               // something rare in these samples: the per-sample order
#pragma unroll 1
          for (int q = 0; q < (int)CB; q++) {
            float xq[4] = {x16[4 * q], x16[4 * q + 1], x16[4 * q + 2], x16[4 * q + 3]}, wq[4];
            block4(xq, wq);
            wb16[4 * q] = wq[0]; wb16[4 * q + 1] = wq[1]; wb16[4 * q + 2] = wq[2]; wb16[4 * q + 3] = wq[3];
          }
        }
        if (writeback && valid && !SPLIT2) {
          float *dst = samples + (size_t)row * pitch + t0 + 4u * c;
#pragma unroll
          for (int k = 0; k < (int)BS; k++) dst[k] = wb16[k];
        }
        c += CB - 1u;
        continue;
      }
      if (fast && lim == 4u && !r_in4) {
        block4(xv, wb);
      } else {
        const float *xs = reinterpret_cast<const float *>(&tin[c * tstride + lane]);
#pragma unroll 1
        for (uint32_t k = 0; k < lim; k++) {
          float w;
          if (r_in4 && s0 + k == nco_r0) nco_refresh(L, C);
          step_generic(xs[k], w);
          if (writeback && valid && !SPLIT2) samples[(size_t)row * pitch + t0 + 4u * c + k] = w;
        }
        continue;
      }
      if (writeback && valid && !SPLIT2) {
        float *dst = samples + (size_t)row * pitch + t0 + 4u * c;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if ((uint32_t)k < lim) dst[k] = wb[k];
      }
    }
    if (SPLIT2) { ti2++; ctr_post(&pctr[1], ti2); }        // (the tile's reads are complete: ctr_post waits for them)
  }
  }   // (wave 1 / the one-wave kernel)

  // ring bookkeeping back to per-stream state
  {
    const uint32_t rl = ring_base + R.k;
    L.ring_len = rl < P.ring_cap ? rl : P.ring_cap;
    const uint32_t al = amp_base + R.k;
    L.amp_len = al < P.amp_cap ? al : P.amp_cap;
    L.poly_phase = R.phase;
    L.amp_pos = R.amp_pos;
  }
  if (SPLIT2 && wave2 == 0u) {
    // wave 0's share of the state: the AGC gain and the pre-filter's history -- written AFTER wave 1 has stored the lane (its copies
    // of these five are the launch's start values)
    __syncthreads();
    if (valid) {
      Real *rs = (Real *)S.rs;
      rs[(size_t)RF_agc_gain * ns + row] = L.agc_gain;
      rs[(size_t)RF_bp_x1 * ns + row] = L.bp_x1; rs[(size_t)RF_bp_x2 * ns + row] = L.bp_x2;
      rs[(size_t)RF_bp_y1 * ns + row] = L.bp_y1; rs[(size_t)RF_bp_y2 * ns + row] = L.bp_y2;
    }
    return;
  }
  for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
  if (FRAC)
    for (uint32_t p = 0; p < P.d; p++) gpoly_u[p * 64u + lane] = poly_u[p * 64u + lane];
  if (valid) {
    if (n > 0) actual_to_pipe(L, S, ns, row, frame, S.nco_inc[row], n);
    store_lane(L, S, ns, row);
    out_counts[stream] = O.out_cnt;
    if (eod_counts) eod_counts[stream] = O.eod_cnt;
  }
  if (SPLIT2) __syncthreads();
}




// (The round-1 whole-tile kernels that lived here -- demod_fast_kernel, demod_split_kernel -- are replaced by
// fsk_pipe.hip's demod_fused_kernel / demod_pipe_kernel.)

// The whole-tile kernels (fsk_pipe.hip) apply to fp32 engines with narrow integer-capacity rings whose streams are in lock
// step (any decimator parity, any alignment: fskhip_demodulate_device cuts a call into head / whole tiles / tail).
bool demod_fast_applicable(int precision, bool uniform, const DemodParams &P, const DemodState &S,
                           const float *samples, size_t pitch) {
  (void)samples;
  return precision == 0 && uniform && !P.wide && !P.frac && P.d >= 2 &&
         sizeof(float4) * (4 * kSlotStride + 16) + sizeof(uint32_t) * 64u * (P.d + 1u) <= 48 * 1024 && (uint64_t)P.amp_cap * P.n_streams * 4u < 0xFFFFFFF0ull &&
         (uint64_t)pitch * 4u * 64u < 0x7FFFFFF0ull;  // per-wave input descriptor and offsets fit 31 bits
}
size_t demod_lds_bytes(const DemodParams &P) {
  const size_t reg = (P.wide ? sizeof(uint64_t) : sizeof(uint32_t)) * 64u * P.d;
  return sizeof(float4) * kChunks * kSlotStride + reg * (P.frac ? 2u : 1u);
}
// ... of the two-wave exact kernel (SPLIT2): the pre-filter ring and its two counters behind the polyphase registers
size_t demod_split2_lds_bytes(const DemodParams &P) {
  return ((demod_lds_bytes(P) + 15u) & ~(size_t)15u) + sizeof(float4) * kPreTiles * kChunks * 64u + 16u;
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
                        hipStream_t stream, bool split2) {
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
  if (split2 && f64 && !wide && !frac && uniform_ds && !trace) {     // the exact path on two waves per group (SPLIT2)
    hipLaunchKernelGGL((demod_kernel<double, uint32_t, false, true, false, true>), g, dim3(128), demod_split2_lds_bytes(P), stream, P, S, samples, n,
                       pitch, vec_ok, wb, append ? 1 : 0, out, out_pitch, out_counts, eod_counts);
    return hipGetLastError();
  }
#define FSK_LAUNCH(R, T, F, U, TR)                                                                 \
  if (f64 == (sizeof(R) == 8) && wide == (sizeof(T) == 8) && frac == F && uniform_ds == U &&       \
      trace == TR)                                                                                 \
    hipLaunchKernelGGL((demod_kernel<R, T, F, U, TR>), g, b, lds_bytes, stream, P, S, samples, n,  \
                       pitch, vec_ok, wb, append ? 1 : 0, out, out_pitch, out_counts, eod_counts);
  FSK_FOR_ALL_VARIANTS(FSK_LAUNCH)
#undef FSK_LAUNCH
  return hipGetLastError();
}

hipError_t set_demod_split2_lds_limit(size_t lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_kernel<double, uint32_t, false, true, false, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
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
