// fsk_pipe_dev.h -- device code shared by the whole-tile fp32 demodulator kernels (fsk_pipe.hip: two / three / one wave per
// 64-stream group and the sample-granular kernel; fsk_blk.hip: the block-batched kernels): the free-running front
// (AGC, pre-filter, mixer, I/Q low-pass), the branch-free discriminator, the per-sample back (back_pair: ZIR repair,
// post filter, slicer, frame state machine), state load / store, LDS counters.  See fsk_pipe.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fsk_params.h"
#include "fsk_dev.h"
#include "fsk_wait.h"

namespace fsk {

// Stage ablation (-DFSK_ABLATE builds only: tools/build_variant.sh abl -DFSK_ABLATE, then tools/variants.py
// "label@abl:FSK_ABLATE=mask"): bit w of g_ablate set = wave w of a group skips its arithmetic and only moves data
// (loads, LDS hand-off, counters), so that the remaining waves set the pace; bit 3 in fsk_blk.hip = the back wave never
// takes its per-sample path.  Results are wrong by construction; only the kernel time is read.  Compiled out of the
// shipped library.
#ifdef FSK_ABLATE
static __device__ int g_ablate;
#define FSK_ABL_INIT const int abl_mask = __builtin_amdgcn_readfirstlane(g_ablate);
#define FSK_ABL(w) (abl_mask & (1 << (w)))
#else
#define FSK_ABL_INIT
#define FSK_ABL(w) 0
#endif

// Wave stamps (-DFSK_STAMP builds only, tools/variants.py --stamps): every wave of the two- and three-wave kernels adds up
// the cycles it spends in its hand-off wait loops and in its whole main loop (s_memtime), and writes both to g_stamp at
// the end: which wave paces the group and how much slack the others have.  Compiled out of the shipped library.
#ifdef FSK_STAMP
static __device__ unsigned long long g_stamp[8 * 2048 * 8];   // [wave][group][wait, total, HW_ID, four event counters, s_memrealtime at the loop's start]
#define FSK_STAMP_DECL unsigned long long st_wait = 0, st_t0 = 0, st_w0 = 0, st_r0 = 0; unsigned st_c0 = 0, st_c1 = 0, st_c2 = 0, st_c3 = 0;
#define FSK_STAMP_COUNT(i) st_c##i += 1u;
#define FSK_STAMP_BEGIN st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime();
#define FSK_STAMP_W0 st_w0 = __builtin_amdgcn_s_memtime();
#define FSK_STAMP_W1 st_wait += __builtin_amdgcn_s_memtime() - st_w0;
#define FSK_STAMP_END(w)                                                               \
  if (blockIdx.x < 2048 && lane == 0) {                                                \
    g_stamp[((w) * 2048 + blockIdx.x) * 8] = st_wait;                                  \
    g_stamp[((w) * 2048 + blockIdx.x) * 8 + 1] = __builtin_amdgcn_s_memtime() - st_t0; \
    g_stamp[((w) * 2048 + blockIdx.x) * 8 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)(__builtin_amdgcn_s_getreg(0x1814) & 7u) << 32); \
    g_stamp[((w) * 2048 + blockIdx.x) * 8 + 3] = st_c0; g_stamp[((w) * 2048 + blockIdx.x) * 8 + 4] = st_c1; \
    g_stamp[((w) * 2048 + blockIdx.x) * 8 + 5] = st_c2; g_stamp[((w) * 2048 + blockIdx.x) * 8 + 6] = st_c3; \
    g_stamp[((w) * 2048 + blockIdx.x) * 8 + 7] = st_r0;                                \
  }
#else
#define FSK_STAMP_DECL
#define FSK_STAMP_COUNT(i)
#define FSK_STAMP_BEGIN
#define FSK_STAMP_W0
#define FSK_STAMP_W1
#define FSK_STAMP_END(w)
#endif

#ifndef FSK_PIPE_SLOTS
#define FSK_PIPE_SLOTS 4
#endif
static constexpr uint32_t kPipeSlots = FSK_PIPE_SLOTS;   // half tiles (8 samples) in the LDS ring between the two waves
static constexpr uint32_t kStartedP = 0x7FFFFFFFu;  // thr_eff while a frame is started, in these kernels: matched - thr_eff stays negative
static constexpr uint32_t kSlotV4 = 6 * 64;
// Everything between the pre-filter and the discriminator is linear, so the whole I/Q branch runs 2^60 times larger than
// the reference's (folded into the pre-filter's b0; the magnitude is scaled back where it is formed).  The reference's
// doubles follow the ringing after a frame down to 1e-300; plain fp32 would lose it at 1e-38 -- and the (0, 0) guard of
// atan2_amp_fma would bend phases from 1e-35 on -- while the bits sliced from that ringing are what a lowered
// syncThreshold syncs on (tools/soak.py).  Scaled, the fp32 branch is exact in the same sense down to 1e-56.
static constexpr float kIqScale = 1152921504606846976.0f;          // 2^60
static constexpr float kIqUnscale = 8.67361737988403547e-19f;      // 2^-60        // v4f per ring slot and lane: y[8] | U[4 pairs x (I, Q)] | (phase, magnitude)[4 pairs]

// ---- front: everything before the decimator, free-running --------------------------------------------------------
struct FrontLane {
  float g, bx1, bx2, by1, by2;   // AGC gain; pre-filter history (outputs carry the low-pass gain b0/2)
  float ix1, ix2, iy, iv;        // I low-pass: x[n-1], x[n-2], y[n-1], velocity      (launch frame)
  float qx1, qx2, qy, qv;        // Q low-pass
};
struct FrontK {                   // all in VGPRs: an SGPR operand halves a vector instruction's rate (valu_probe)
  float att_m_rel, rel;          // AGC: attack - release, release
  float step_k, step_b;          // 2^40, -2^39: clamp(level*2^40 - 2^39) = [level > 0.5]
  float g_lo, g_hi;              // 0.1, 10
  float bp_b0, bp_na1, bp_na2;   // pre-filter: y = b0*(x - x2) - a2*y2 - a1*y1; b0 carries the low-pass gain b0/2 and kIqScale
  float lp_a2, lp_nd;            // low-pass a2, -(1 + a1 + a2)
  float tiny; uint32_t sgn;      // 2^-123, 0x80000000 (atan2_amp_fma)
};

// one input sample, first half: AGC and pre-filter (what resetState() never touches); returns the AGC'd sample (write-back)
// and the pre-filter output
__device__ inline void front_agc_bp(FrontLane &F, const FrontK &K, float xin, float &xs, float &y) {
  // AGC (fsk.ts:52-76); exact zero holds the gain.  The attack/release choice is arithmetic (a clamped fma is an
  // FMA-class instruction, compare + select are two half-rate ones): exact for every level, since
  // (level - 0.5) * 2^40 >= 2^16 for the smallest level above 0.5.
  const float xv = xin * F.g;
  xs = xv;
  const float level = __builtin_fabsf(xv);
  const float t = __builtin_fmaf(0.5f, __builtin_amdgcn_rcpf(level), -F.g);
  float st;
  asm("v_fma_f32 %0, |%1|, %2, %3 clamp" : "=v"(st) : "v"(xv), "v"(K.step_k), "v"(K.step_b));
  const float rate = __builtin_fmaf(st, K.att_m_rel, K.rel);
  float gn = __builtin_fmaf(t, rate, F.g);
  gn = level > 0.0f ? gn : F.g;
  F.g = __builtin_amdgcn_fmed3f(gn, K.g_lo, K.g_hi);
  // pre-filter (filters.ts:47-87), b1 = 0, b2 = -b0
  float v = K.bp_b0 * (xv - F.bx2);
  v = __builtin_fmaf(K.bp_na2, F.by2, v);
  v = __builtin_fmaf(K.bp_na1, F.by1, v);
  F.bx2 = F.bx1; F.bx1 = xv;
  F.by2 = F.by1; F.by1 = v;
  y = v;
}
// front_agc_bp as its two halves (the seven-wave kernel of fsk_blk6.hip and the five-wave kernel of fsk_blk.hip give them to different waves), instruction for instruction: the AGC (fsk.ts:52-76) ...
__device__ __forceinline__ float front_agc(FrontLane &F, const FrontK &K, float xin) {
  const float xv = xin * F.g;
  const float level = __builtin_fabsf(xv);
  const float t = __builtin_fmaf(0.5f, __builtin_amdgcn_rcpf(level), -F.g);
  float st;
  asm("v_fma_f32 %0, |%1|, %2, %3 clamp" : "=v"(st) : "v"(xv), "v"(K.step_k), "v"(K.step_b));
  const float rate = __builtin_fmaf(st, K.att_m_rel, K.rel);
  float gn = __builtin_fmaf(t, rate, F.g);
  gn = level > 0.0f ? gn : F.g;
  F.g = __builtin_amdgcn_fmed3f(gn, K.g_lo, K.g_hi);
  return xv;
}
// ... and the pre-filter (filters.ts:47-87), b1 = 0, b2 = -b0
__device__ __forceinline__ float front_bp(FrontLane &F, const FrontK &K, float xv) {
  float v = K.bp_b0 * (xv - F.bx2);
  v = __builtin_fmaf(K.bp_na2, F.by2, v);
  v = __builtin_fmaf(K.bp_na1, F.by1, v);
  F.bx2 = F.bx1; F.bx1 = xv;
  F.by2 = F.by1; F.by1 = v;
  return v;
}

// second half: mix with the free-running NCO + I/Q low-pass (fsk.ts:229-238), velocity form, gain already on y
__device__ inline void front_mix_lp(FrontLane &F, const FrontK &K, float v, float c, float s, float &oi, float &oq) {
  const float mi = v * c, mq = v * s;
  const float ti = __builtin_fmaf(2.0f, F.ix1, mi) + F.ix2;
  const float tq = __builtin_fmaf(2.0f, F.qx1, mq) + F.qx2;
  F.iv = __builtin_fmaf(K.lp_a2, F.iv, __builtin_fmaf(K.lp_nd, F.iy, ti));
  F.qv = __builtin_fmaf(K.lp_a2, F.qv, __builtin_fmaf(K.lp_nd, F.qy, tq));
  F.iy += F.iv; F.qy += F.qv;
  F.ix2 = F.ix1; F.ix1 = mi;
  F.qx2 = F.qx1; F.qx1 = mq;
  oi = F.iy; oq = F.qy;
}
// one input sample: returns the AGC'd sample (write-back), the pre-filter output and the I/Q low-pass outputs
__device__ inline void front_sample(FrontLane &F, const FrontK &K, float xin, float c, float s, float &xs, float &y,
                                    float &oi, float &oq) {
  front_agc_bp(F, K, xin, xs, y);
  front_mix_lp(F, K, y, c, s, oi, oq);
}

// resetState() reaches the free-running I/Q low-pass kZeroLagPairs decimated samples late (see the file comment)
__device__ inline void front_zero(FrontLane &F, bool hit) {
  if (hit) { F.ix1 = F.ix2 = F.iy = F.iv = 0.f; F.qx1 = F.qx2 = F.qy = F.qv = 0.f; }
}
// mailbox value at the start of a launch: the direct instance has produced dph decimated samples since the reset
__device__ inline uint32_t zmail_init(uint32_t dph) { return dph <= kZeroLagPairs ? kZeroLagPairs - dph : 0xFFFFFFFFu; }

// Discriminator front half (fsk.ts:251-252): phase and magnitude of one decimated I/Q pair sum, with FMA-class
// instructions only besides the two transcendentals (no min/max, compare or select: those issue at half rate).
//   atan(|y|/|x|) = pi/4 + atan(u),  u = (|y| - |x|) / (|y| + |x|) in [-1, 1]      (no octant swap)
//   |(x, y)| = (|x| + |y|) * sqrt((1 + u^2) / 2)                                     (cannot underflow: see fsk_dev.h)
//   quadrant: pi/2 + ((atan(u) - pi/4) XOR signbit(x)), then OR signbit(y)
// |x| carries +2^-123 so that (0, 0) gives u = -1 exactly (the reciprocal of a power of two is exact), without a guard
// instruction; the polynomial's leading coefficient is nudged so that atan(-1) evaluates to -fl(pi/4) bit for bit, and
// with fl(pi/2) = 2 fl(pi/4) the angle of (0, 0) is then exactly +-0 like Math.atan2(0, 0) (tests/test_gpu_parity.py
// checks that on the device).  Otherwise fsk_dev.h's odd minimax polynomial (|error| <= 1.6e-7 rad on [-1, 1]).
// tiny = 2^-123 and sgn = 0x80000000 arrive in VGPRs: as literals they would be hoisted into SGPRs (VOP3 cannot
// encode a literal), and an SGPR operand halves the instruction's rate.
__device__ inline float atan2_amp_fma(float y, float x, float &amp, float tiny, uint32_t sgn) {
  x = x + 0.0f;                                        // -0 counts as +0 (the reference's averages are never -0)
  const float axp = __builtin_fabsf(x) + tiny;
  const float ay = __builtin_fabsf(y);
  const float sm = ay + axp;
  const float u = (ay - axp) * __builtin_amdgcn_rcpf(sm);
  const float s = u * u;
  amp = sm * __builtin_amdgcn_sqrtf(__builtin_fmaf(s, 0.5f, 0.5f));
  float p = -4.3553458527e-03f;   // (nudged by 135 ulp: see above)
  p = __builtin_fmaf(p, s, 2.304014596e-02f);
  p = __builtin_fmaf(p, s, -5.777360382e-02f);
  p = __builtin_fmaf(p, s, 9.794235514e-02f);
  p = __builtin_fmaf(p, s, -1.397658244e-01f);
  p = __builtin_fmaf(p, s, 1.996270403e-01f);
  p = __builtin_fmaf(p, s, -3.333165903e-01f);
  const float r = __builtin_fmaf(u * s, p, u);         // atan(u)
  const float phi = r - 0.78539816339744831f;          // in [-pi/2, 0]
  const uint32_t sx = __builtin_bit_cast(uint32_t, x) & sgn;
  const float th = 1.57079632679489662f + __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, phi) ^ sx);
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, th) | (__builtin_bit_cast(uint32_t, y) & sgn));
}

// all ones iff v < 0.  Right shifts, add/sub and the bit ops are FMA-class instructions on gfx950 while v_cmp, v_cndmask,
// v_addc and the left shifts issue at half rate (profiles/r02_valu_probe*.txt), so per-sample flags are computed as sign
// bits.  The empty asm keeps hipcc from folding the mask back into a compare + select.
__device__ inline uint32_t neg_mask(uint32_t v) {
  uint32_t m = (uint32_t)((int32_t)v >> 31);
  asm("" : "+v"(m));
  return m;
}
// The slicer (fsk.ts:264: bitValue = filteredPhaseDiff > 0 ? 1 : 0) as a value whose SIGN BIT is the bit -- what the kernels shift
// into the polyphase registers (v_alignbit) without extracting it.  Round 6: 0 - clamp(4 f) instead of 0 - f.  clamp(4 f) is in
// (0, 1] exactly when f > 0 (denormals are kept: a subnormal f gives a subnormal product) and +0 for f <= 0, f = +-0 AND f = NaN
// (the kernels run with DX10_CLAMP = 1: a clamped result of NaN is 0) -- `NaN > 0` is false in the reference, whereas the sign of
// 0 - NaN is the NaN's own, flipped: a stream poisoned by a +NaN sample sliced ones where the reference slices zeros
// (tests/golden/golden_hostile.npz, h_*_qnan_mid).  One more full-rate instruction per decimated sample.
__device__ inline float slicer_nf(float f) {
  float r;
  asm("v_mul_f32 %0, 4.0, %1 clamp" : "=v"(r) : "v"(f));
  return 0.0f - r;
}
__device__ inline uint32_t sign_bit(uint32_t v) {
  uint32_t m = v >> 31;
  asm("" : "+v"(m));
  return m;
}

// ---- back: decimated rate ----------------------------------------------------------------------------------------
struct BackLane {
  float qai, qaq, qbi, qbq;      // ZIR pair sums of the upcoming two decimated samples
  float px1, px2, py, pv;        // post filter (velocity form)
  float last_phase, thr;         // lastPhase in the free-running frame; silence threshold
  float thf;                     // the reference's phase 0 seen from the free-running frame (changes at resetState() only)
  // the kDirectPairs decimated samples after a reset come from a zero-started filter instance (dph = how many so far)
  uint32_t dph;
  float dix1, dix2, diy, dvi, dqx1, dqx2, dqy, dqv, q0i, q0q;
  // frame state machine, as absolute push counts of this launch (k = pushes so far, wave-uniform):
  uint32_t matched, thr_eff;     // sync correlator count; matched_min while searching, kStartedP while a frame is started
  uint32_t rho;                  // globalSampleCounter % cadence == 0  <=>  k % cadence == rho
  uint32_t ls;                   // silence.sampleCount = k - ls
  uint32_t acc, T, tlast;        // bit vote; nextBitSampleIndex - bitSampleCounter = T - k; bitAccumCount = k - tlast
  uint32_t sreg;                 // byteState as a shift register under a sentinel bit: 1 = waiting for the start bit,
                                 // 1 s d7..d0 (bit 9 set) = all data bits in, the next decision is the stop (or parity) bit
  uint32_t out_cnt;
};
struct BackK {                    // VGPRs, like FrontK
  float c1, c2;                  // ZIR pair-sum recurrence
  float lp_b0, lp_a2, lp_nd;     // post filter: b0, a2, -(1 + a1 + a2)
  uint32_t qn, mask;             // ~pattern, window mask (bits 1 .. nBits-1)
  uint32_t d;                    // downsampledSamplesPerBit
  float tiny; uint32_t sgn;      // 2^-123, 0x80000000 (atan2_amp_fma)
  uint32_t edge_min;             // (1 << stop_pos) - 2: sreg - 2 >= edge_min (unsigned) <=> start or stop position
  uint32_t eod_m1;               // samplesForEOD - 1
  float zk;                      // -2^123: clamp(2 + zk * |w|) = [the I/Q pair sum is exactly (0, 0)] (its scaled magnitude is the 2^-123 guard)
  float unscale;                 // 2^-60
};
struct BackU {                   // wave-uniform context of one decimated sample
  uint32_t k;                    // pushes of this launch including this one
  uint32_t kv;                   // the same in a VGPR (operand of the per-lane selects and differences)
  uint32_t phase;                // polyphase slot of this push
  uint32_t amp_soff;             // amplitude ring: byte offset of the push slot (quad row + slot within the quad, fsk_dev.h)
  uint32_t direct;               // decimated samples for which some lane still runs the direct instance (after a reset)
  uint32_t zlive;                // some lane carries a non-zero correction (or runs the direct instance)
  uint32_t own_pairs, hand_lag;  // HAND kernels: zr_dph up to which the correction is this wave's own, and the lag of the hand-over (kHandPairs, kHandLag
                                 // in fsk_blk6.hip; kOwnPairs4, kOwnLag4 in fsk_blk.hip)
  uint32_t *zmail;               // LDS [64]: decimated-sample index of this launch before which the front zeroes the lane's filters
  uint32_t *cmail;               // LDS [6][64] or null (fsk_blk.hip): hand-over of the ZIR correction to the discriminator wave
                                 // ([0] from which sample, [1..4] its value kHandLag steps of the recurrence before that sample,
                                 // or there if [0] <= kHandLag), [5] the sample this wave's own span began at
  uint64_t free0;                // free-running frame: NCO phase (turns * 2^64) at the first sample of the launch
};

// resetState() fsk.ts:175-188 at the end of push k: the next input sample is n0 = 2k of this launch.
template <bool UNI, int COH = 0>
__device__ inline void back_reset(BackLane &B, const DemodParams &P, const FastMem &M, const BackU &X, uint64_t inc,
                                  uint32_t lane) {
  // the reference's NCO restarts at 0: from here on its phase is the free-running frame's minus that frame's phase at
  // n0, and its lastPhase = 0 is that phase in the free frame
  const uint64_t fr0 = X.free0 + inc * (uint64_t)(2u * X.k);
  const uint64_t off = 0ull - fr0;
  ist_store<COH>(M, IF_fr_lo, (uint32_t)off);
  ist_store<COH>(M, IF_fr_hi, (uint32_t)(off >> 32));
  {
    double r = (double)fr0 * 5.42101086242752217e-20 * 6.283185307179586476925;   // 2^-64 turns -> radians
    r = r > 3.14159265358979323846 ? r - 6.283185307179586476925 : r;
    B.last_phase = (float)r;
    B.thf = (float)r;
  }
  B.dph = 0;
  X.zmail[lane] = X.k + kZeroLagPairs;   // the next decimated sample is number X.k of this launch (0-based)
  if (X.cmail) {
    X.cmail[lane] = 0xFFFFFFFFu;               // a hand-over posted for an earlier reset is void
    X.cmail[320u + lane] = X.k;                // from this sample on the pair sums themselves are wanted, not their phase
  }
  B.dix1 = B.dix2 = B.diy = B.dvi = 0.f;
  B.dqx1 = B.dqx2 = B.dqy = B.dqv = 0.f;
  B.px1 = B.px2 = B.py = B.pv = 0.f;
  ist_store<COH>(M, IF_gsc, 0u - X.k);
  B.rho = X.k % P.cadence;
  B.ls = X.k;
  B.acc = 0; B.T = X.k + kBigWait; B.tlast = B.T;
  B.sreg = 1u;
  B.thr_eff = M.voff < 0xFFFFFFF0u ? P.matched_min : 0x7FFFFFFEu;  // lanes beyond the batch stay parked
}

// e^{j 2 pi acc / 2^64}: the hardware's sin/cos take turns
__device__ inline void nco_phasor(uint64_t acc, float &c, float &s) {
  const float turns = (float)(uint32_t)(acc >> 32) * 2.3283064365386963e-10f;  // 2^-32
  c = __builtin_amdgcn_cosf(turns);
  s = __builtin_amdgcn_sinf(turns);
}

// Discriminator tail of one decimated sample (fsk.ts:251-264): phase `ph` (free-running frame) and scaled magnitude `amp`
// of the corrected I/Q pair sum -> wrapped phase difference -> post filter.  Returns the post filter's output, leaves the
// reference's magnitude in `amp`.  One function for the per-sample path (back_pair) and the block path (fsk_blk.hip): the
// same instruction sequence, so which path a decimated sample takes cannot show in its value.
//   Math.atan2(0, 0) = 0 is a convention of the reference's frame: an exactly-zero I/Q pair sum (digital silence through
//   zero-started filters: lead-ins, long gaps) has the reference's phase 0, which in the free-running frame is thf --
//   otherwise a resetState() inside such silence would feed the post filter a spurious step of w*n0, and its decaying
//   response would put bits into the silence that the reference does not see (found by tools/soak.py).  Arithmetic
//   select: zf = 1 exactly when the scaled magnitude is the 2^-123 guard alone, 0 from twice that on.
__device__ inline float disc_post(BackLane &B, const BackK &K, float ph, float &amp) {
  {
    float zf;
    asm("v_fma_f32 %0, %1, %2, 2.0 clamp" : "=v"(zf) : "v"(amp), "v"(K.zk));
    ph = __builtin_fmaf(zf, B.thf, ph);                        // (ph is exactly +-0 there)
  }
  amp *= K.unscale;                                            // the reference's magnitude (0 for the guard alone)
  float dphi = ph - B.last_phase;
  {
    // wrap into (-pi, pi] (fsk.ts:255-257): |dphi| < 2 pi, so one rounded quotient does both branches; rounding to
    // nearest-even by adding and subtracting 1.5 * 2^23 (v_rndne_f32 is a half-rate instruction)
    float tq = __builtin_fmaf(dphi, 0.15915494309189535f, 12582912.0f);
    tq -= 12582912.0f;
    dphi = __builtin_fmaf(-6.283185307179586f, tq, dphi);
  }
  B.last_phase = ph;
  // post filter (fsk.ts:261), velocity form as fsk_dev.h's lp32
  const float tt = __builtin_fmaf(2.0f, B.px1, dphi) + B.px2;
  B.pv = __builtin_fmaf(K.lp_a2, B.pv, __builtin_fmaf(K.lp_nd, B.py, K.lp_b0 * tt));
  B.py += B.pv;
  B.px2 = B.px1; B.px1 = dphi;
  return B.py;
}

// The ZIR correction's share of one decimated sample, for a wave some lane of which carries one (X.zlive): w = U - q, q
// advances; lanes inside the span after a reset run the zero-started ("direct") instance instead and derive q's start
// values from it; phase / magnitude are re-evaluated where that changes them (HAND: for the lanes whose correction is this
// wave's own).  One function for the per-sample path (back_pair) and the block path's own-span blocks (fsk_blk.hip).
template <bool UNI, bool HAND>
__device__ inline void zir_step(BackLane &B, const BackK &K, BackU &X, float Ui, float Uq, const float *ypair, uint64_t inc,
                                uint32_t lane, float &ph, float &amp, const float *zph = nullptr) {
  const uint32_t dph0 = B.dph;
  float wi = Ui - B.qai, wq = Uq - B.qaq;
  {
    const float ni = __builtin_fmaf(K.c1, B.qbi, -(K.c2 * B.qai));
    const float nq = __builtin_fmaf(K.c1, B.qbq, -(K.c2 * B.qaq));
    B.qai = B.qbi; B.qaq = B.qbq; B.qbi = ni; B.qbq = nq;
  }
  // ---- rare: the decimated samples after a reset come from the zero-started instance; its last two (the first two
  // after the front has zeroed its filters) also yield q's start values
  if (__builtin_expect(X.direct != 0u, 0)) {
    X.direct--;
    if (B.dph < kDirectPairs) {
      const float y0 = ypair[0], y1 = ypair[1];
      const uint32_t n0 = 2u * (X.k - 1u);
      // the front's phasors of these two samples: evaluated the same way (nco_phasor), or -- zph, fsk_blk.hip with a uniform
      // configuration -- the very values wave 0 evaluated for the tile and left in LDS (c0, s0, c1, s1)
      float c0, s0, c1, s1;
      if (zph) { c0 = zph[0]; s0 = zph[1]; c1 = zph[2]; s1 = zph[3]; }
      else {
        nco_phasor(X.free0 + inc * (uint64_t)n0, c0, s0);
        nco_phasor(X.free0 + inc * (uint64_t)(n0 + 1u), c1, s1);
      }
      float di, dq;
      {
        const float mi = y0 * c0, mq = y0 * s0;
        const float ti = __builtin_fmaf(2.0f, B.dix1, mi) + B.dix2, tq = __builtin_fmaf(2.0f, B.dqx1, mq) + B.dqx2;
        B.dvi = __builtin_fmaf(K.lp_a2, B.dvi, __builtin_fmaf(K.lp_nd, B.diy, ti));
        B.dqv = __builtin_fmaf(K.lp_a2, B.dqv, __builtin_fmaf(K.lp_nd, B.dqy, tq));
        B.diy += B.dvi; B.dqy += B.dqv;
        B.dix2 = B.dix1; B.dix1 = mi; B.dqx2 = B.dqx1; B.dqx1 = mq;
        di = B.diy; dq = B.dqy;
      }
      {
        const float mi = y1 * c1, mq = y1 * s1;
        const float ti = __builtin_fmaf(2.0f, B.dix1, mi) + B.dix2, tq = __builtin_fmaf(2.0f, B.dqx1, mq) + B.dqx2;
        B.dvi = __builtin_fmaf(K.lp_a2, B.dvi, __builtin_fmaf(K.lp_nd, B.diy, ti));
        B.dqv = __builtin_fmaf(K.lp_a2, B.dqv, __builtin_fmaf(K.lp_nd, B.dqy, tq));
        B.diy += B.dvi; B.dqy += B.dqv;
        B.dix2 = B.dix1; B.dix1 = mi; B.dqx2 = B.dqx1; B.dqx1 = mq;
        di += B.diy; dq += B.dqy;
      }
      wi = di; wq = dq;
      if (B.dph == kZeroLagPairs) {
        B.q0i = Ui - di; B.q0q = Uq - dq;
      } else if (B.dph == kZeroLagPairs + 1u) {
        const float q1i = Ui - di, q1q = Uq - dq;
        B.qai = __builtin_fmaf(K.c1, q1i, -(K.c2 * B.q0i));
        B.qaq = __builtin_fmaf(K.c1, q1q, -(K.c2 * B.q0q));
        B.qbi = __builtin_fmaf(K.c1, B.qai, -(K.c2 * q1i));
        B.qbq = __builtin_fmaf(K.c1, B.qaq, -(K.c2 * q1q));
        if (HAND) {
          // the correction stays here for the next kHandLag decimated samples (numbers X.k .. X.k + kHandLag - 1 of this
          // launch) and then moves to the discriminator wave.  Its values at the hand-over sample follow from these by
          // kHandLag steps of the recurrence alone: posted are the start values, and the discriminator wave -- the one with
          // time to spare wherever resets are frequent -- runs the steps when it takes them (round 4; the back wave used to,
          // ~100 instructions of the wave that paces an idle receiver bank, per reset).  A hand-over posted inside a launch
          // is due at sample kHandLag + 1 of it at the earliest; one posted at a launch's start (fsk_blk.hip: the remaining
          // steps run there, once) at sample kHandLag at the latest -- which is how the taker tells them apart.
          X.cmail[64u + lane] = __builtin_bit_cast(uint32_t, B.qai); X.cmail[128u + lane] = __builtin_bit_cast(uint32_t, B.qaq);
          X.cmail[192u + lane] = __builtin_bit_cast(uint32_t, B.qbi); X.cmail[256u + lane] = __builtin_bit_cast(uint32_t, B.qbq);
          X.cmail[lane] = X.k + X.hand_lag;
        }
      }
      B.dph += 1u;
    }
  }
  if (dph0 >= kDirectPairs && dph0 < (HAND ? X.own_pairs : kHandPairs)) {   // the un-retired span after the direct instance (HAND: this wave's part of it)
    B.dph = dph0 + 1u;
    if (HAND && B.dph == X.own_pairs) { B.qai = 0.f; B.qaq = 0.f; B.qbi = 0.f; B.qbq = 0.f; }   // handed over
  }
  if (HAND) {
    // lanes whose correction (or direct instance) is this wave's own evaluate the discriminator here; the others keep
    // the discriminator wave's result, which already carries their correction
    const bool own = dph0 < X.own_pairs;
    if (__builtin_amdgcn_ballot_w64(own)) {
      float a2;
      const float p2 = atan2_amp_fma(wq, wi, a2, K.tiny, K.sgn);
      ph = own ? p2 : ph; amp = own ? a2 : amp;
    }
    X.zlive = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(B.dph < X.own_pairs) != 0));
  } else {
    const uint32_t changed = (__builtin_bit_cast(uint32_t, wi) ^ __builtin_bit_cast(uint32_t, Ui)) |
                             (__builtin_bit_cast(uint32_t, wq) ^ __builtin_bit_cast(uint32_t, Uq));
    if (__builtin_amdgcn_ballot_w64(changed != 0u)) ph = atan2_amp_fma(wq, wi, amp, K.tiny, K.sgn);
  }
  if (!HAND) {
    // a correction that has decayed below 2^-28 of the magnitude it corrects is retired to exactly zero (both decay at
    // the low-pass's own rate at least, so it stays negligible; a rule of the stream's own values only, so every kernel
    // and every chunking retires it at the same decimated sample)
    const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(B.qai), __builtin_fabsf(B.qaq)),
                                      __builtin_fmaxf(__builtin_fabsf(B.qbi), __builtin_fabsf(B.qbq)));
    const bool steady = dph0 >= kHandPairs;                       // (from the sample on that the discriminator wave would own it)
    const bool small = !(big > amp * 3.7252902984619141e-09f);   // (<=: a zero correction under the bare (0, 0) guard retires too)
    if (steady & small) { B.qai = 0.f; B.qaq = 0.f; B.qbi = 0.f; B.qbq = 0.f; }
    X.zlive = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(!steady | !small) != 0));
  }
}

// One decimated sample: ZIR correction, discriminator (fsk.ts:245-264), processDownsampledBit (fsk.ts:278-344),
// processByte (346-375).  ypair: LDS address of this pair's two pre-filter outputs (read by the direct instance only);
// pslot: where the polyphase sync-bit register of this push lives (r_old is its value).
// PA: the front wave has evaluated the discriminator's phase / magnitude on U already (ph_u, amp_u); otherwise this
// function does.  They stand unless the correction changed a bit of U in some lane, in which case the wave re-evaluates --
// same function, same inputs where nothing changed, so the result does not depend on which wave computed it.
// TRC: honour fskhip_trace_enable and fskhip_enable_signal_quality (the sample-granular kernel only; an engine with
// either switched on runs entirely on it).
// COH: cache policy of this function's state accesses and of the amplitude ring's store (kCohSc1 in fsk_blk.hip's
// time-sliced launches, whose next slice may run behind another L2; 0 otherwise).
// HAND (fsk_blk.hip's back wave): the correction belongs to this wave only while zr_dph < kHandPairs; after that the
// discriminator wave applies it and ph_u / amp_u already are the corrected pair sum's.
template <bool UNI, bool PA = false, bool TRC = false, bool HAND = false, int COH = 0>
__device__ inline void back_pair(BackLane &B, const BackK &K, const DemodParams &P, const DemodState &S, const FastMem &M,
                                 uint32_t *pslot, uint32_t lane, __amdgpu_buffer_rsrc_t amp_rsrc, uint8_t *out,
                                 uint32_t out_pitch, uint32_t *eod_counts, BackU &X, float Ui, float Uq,
                                 const float *ypair, uint32_t r_old, uint64_t inc, float ph_u = 0.f, float amp_u = 0.f,
                                 const float *zph = nullptr) {
  // ---- ZIR correction: w = U - q, q advances by its two-term recurrence.  Skipped (exactly: U - 0 = U) while no lane of
  // the wave carries a correction.
  // Phase and magnitude are evaluated on U first (by the front wave already, in the two-wave kernel) and stand unless the
  // correction changes a bit of U in some lane; so the common path has no else-branch for the register allocator to park
  // the correction's state copies in.
  float amp, ph;
  if (PA) { ph = ph_u; amp = amp_u; }
  else ph = atan2_amp_fma(Uq, Ui, amp, K.tiny, K.sgn);
  if (__builtin_expect(X.zlive != 0u, 0)) zir_step<UNI, HAND>(B, K, X, Ui, Uq, ypair, inc, lane, ph, amp, zph);
  // ---- discriminator (fsk.ts:251-264)
  const float f = disc_post(B, K, ph, amp);
  // slicer (fsk.ts:264): f > 0  <=>  sign bit of 0 - f  (f = +-0 gives +0, i.e. bit 0)
  const uint32_t bit = sign_bit(__builtin_bit_cast(uint32_t, slicer_nf(f)));
  if (TRC) {
    if (S.trace_stream != 0xFFFFFFFFu && M.voff == S.trace_stream * 4u) {
      const uint32_t kk = *S.trace_n;
      if (kk < S.trace_cap) {
        S.trace_amp[kk] = (double)amp; S.trace_post[kk] = (double)f; S.trace_bit[kk] = (uint8_t)bit;
      }
      *S.trace_n = kk + 1;
    }
  }

  // ---- processDownsampledBit (fsk.ts:278-344)
  const uint32_t qn = K.qn, mask = K.mask;
  const uint32_t r = r_old + r_old + bit;                      // syncSamplesBuffer.put(bit)
  *pslot = r;                                                  // (the polyphase register of this push slot)
  B.matched += (uint32_t)__builtin_popcount((r ^ qn) & mask);
  B.matched -= (uint32_t)__builtin_popcount((r_old ^ qn) & mask);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, amp), amp_rsrc, M.avoff, X.amp_soff, COH);  // syncAmplitudeBuffer.put
  {
    const uint32_t silent = neg_mask(__builtin_bit_cast(uint32_t, amp - B.thr));   // amp < threshold (fsk.ts:285)
    B.ls = (B.ls & silent) | (X.kv & ~silent);                 // silence run = k - ls (fsk.ts:285-295)
  }
  const uint32_t e1 = K.eod_m1 - (X.kv - B.ls);                // negative <=> silence.sampleCount >= samplesForEOD
  B.acc += bit;                                                // bit clock, ungated
  // nextBitSampleIndex reached (only frames that are started get here: T is parked otherwise), and no 'eod' in this step
  const uint32_t dm = ~(neg_mask(X.kv - B.T) | neg_mask(e1));
  // one test for both rare events: 'eod', or a sync candidate (matched >= threshold while searching; whether this step
  // is on the search cadence is only looked at inside)
  const uint32_t m1 = B.matched - B.thr_eff;                   // >= 0 (as int32) <=> matched >= thr_eff

  if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)(e1 | ~m1) < 0) != 0, 0)) {
    const bool eod = (int32_t)e1 < 0;
    const bool cand = ((int32_t)m1 >= 0) & (B.rho == X.k % P.cadence);   // globalSampleCounter % round(dsSPB/4) == 0
    if (__builtin_amdgcn_ballot_w64(eod)) {                    // fsk.ts:288-291
      if (TRC && P.quality) {   // opt-in estimates (sample-granular kernel only): the noise floor of the silence behind the first 'eod' after a sync
        const uint32_t pushes = ist_load<COH>(M, IF_amp_len) + X.k;
        quality_on_eod<float>(P, S, lane, M.voff >> 2, eod & (M.voff < 0xFFFFFFF0u), amp_pos_of(X.amp_soff, P.n_streams * 16u),
                              pushes < P.amp_cap ? pushes : P.amp_cap);
      }
      if (eod) {
        ist_add<COH>(M, IF_eod_total, 1u);
        if (eod_counts && M.voff < 0xFFFFFFF0u) __hip_atomic_fetch_add(&eod_counts[M.voff >> 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        back_reset<UNI, COH>(B, P, M, X, inc, lane);
      }
      X.direct = kDirectPairs; X.zlive = 1u;                   // (wave-uniform: set where the wave-uniform branch is)
    }
    // ring length >= preamble window? (fsk.ts:302); ring_len / amp_len in HBM hold the launch-start values
    bool sync_now = false;
    uint32_t slen = 0;
    if (cand & !eod) {
      const uint32_t ring_base = ist_load<COH>(M, IF_ring_len);
      sync_now = (ring_base + X.k >= P.sample_count) & (M.voff < 0xFFFFFFF0u);
      const uint32_t pushes = ist_load<COH>(M, IF_amp_len) + X.k;
      slen = pushes < P.amp_cap ? pushes : P.amp_cap;
    }
    uint64_t m = __builtin_amdgcn_ballot_w64(sync_now);
    if (m) {
      if (sync_now) {                                          // fsk.ts:315-319
        B.thr_eff = kStartedP;
        B.sreg = 1u;
        B.acc = 0; B.T = X.k; B.tlast = X.k;
        ist_add<COH>(M, IF_sync_det, 1u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's ring stores have reached L2
      while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t srow = (uint32_t)__builtin_amdgcn_readlane((int)M.voff, src) >> 2;
        const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)slen, src);
        double part = 0.0;
        for (uint32_t i = lane; i < sl; i += 64) {
          const float *p = S.amp_ring + amp_index(i, srow, P.n_streams);
          part += (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double sum = wave_sum(part);
        if ((int)lane == src) {
          B.thr = (float)((sum / (double)sl) * 0.1);           // fsk.ts:321-326
          if (TRC && P.quality) quality_on_sync<float>(P, S, M.voff >> 2, sum / (double)sl);
        }
      }
    }
  }

  // ---- bit decision (fsk.ts:335-341) + processByte (346-375).  Some lane decides at nearly every step, so the common
  // part -- vote, clock advance, shifting the bit in -- runs for all lanes as masked arithmetic (one mask, bit ops) instead
  // of a divergent block per step; only start and stop positions (two in ten decisions) branch.
  const uint32_t s0 = B.sreg;
  const uint32_t cnt0 = X.kv - B.tlast, ones0 = B.acc;         // the vote, for the opt-in estimates at start / stop positions
  const uint32_t b = sign_bit((X.kv - B.tlast) - B.acc - B.acc);   // 2 * bitAccumulator > bitAccumCount (fsk.ts:336)
  B.sreg = (s0 & ~dm) | ((s0 + s0 + b) & dm);
  B.acc &= ~dm;
  B.T += K.d & dm;                                             // nextBitSampleIndex += dsSPB
  B.tlast = (B.tlast & ~dm) | (X.kv & dm);
  const bool edge = ((s0 - 2u >= K.edge_min) ? dm : 0u) != 0u;
  if (__builtin_amdgcn_ballot_w64(edge)) {
    // flat on purpose: one masked region for the common case (a byte completes), one wave-uniform test for the two rare ones
    const bool at_stop = edge & (s0 != 1u);
    const bool good_stop = at_stop & (b != 0u);
    const bool bad_stop = at_stop & (b == 0u);
    const bool bad_start = edge & (s0 == 1u) & (b != 0u);            // fsk.ts:352-355
    if (TRC && P.quality) {
      if (edge & (s0 == 1u) & (b == 0u) & (M.voff < 0xFFFFFFF0u)) quality_on_start<float>(P, S, M.voff >> 2, f);
      if (good_stop & (M.voff < 0xFFFFFFF0u))
        quality_on_byte<float>(P, S, M.voff >> 2, (s0 >> (P.stop_pos - 9u)) & 0xFFu, ones0, cnt0, f);
    }
    if (good_stop) {                                                 // stop bit: fsk.ts:367-368
      if (M.voff < 0xFFFFFFF0u && B.out_cnt < out_pitch)
        out[(size_t)(M.voff >> 2) * out_pitch + B.out_cnt] = (uint8_t)(s0 >> (P.stop_pos - 9u));
      B.out_cnt++;
      B.sreg = 1u;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad_stop | bad_start) != 0, 0)) {
      if (bad_stop) {                                                // fsk.ts:363-366 -- started = false, byteState stays
        const uint32_t reload = B.T - B.tlast;
        B.T = X.k + kBigWait; B.tlast = B.T - reload;
        B.thr_eff = P.matched_min;
        B.sreg = s0;
      }
      if (__builtin_amdgcn_ballot_w64(bad_start)) {
        if (bad_start) back_reset<UNI, COH>(B, P, M, X, inc, lane);
        X.direct = kDirectPairs; X.zlive = 1u;
      }
    }
  }
}

// ---- state arrays <-> registers ------------------------------------------------------------------------------
#define PIPE_RLOAD(f) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, row4, (uint32_t)RF_##f * fld, COH))
#define PIPE_ILOAD(f) __builtin_amdgcn_raw_buffer_load_b32(M.is_rsrc, row4, (uint32_t)IF_##f * fld, COH)
#define PIPE_RSTORE(f, v) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)(v)), rs_rsrc, M.voff, (uint32_t)RF_##f * fld, COH)
#define PIPE_ISTORE(f, v) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(v), M.is_rsrc, M.voff, (uint32_t)IF_##f * fld, COH)
#define PIPE_CLOAD(f) (__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cf_rsrc, row4 * 2u, (uint32_t)(f) * fld * 2u, 0)))

struct PipeCtx {   // descriptors and offsets both halves use
  __amdgpu_buffer_rsrc_t rs_rsrc, cf_rsrc;
  FastMem M;
  uint32_t fld, row4;
  bool valid;
};
__device__ inline PipeCtx pipe_ctx(const DemodParams &P, const DemodState &S, uint32_t stream) {
  PipeCtx C;
  C.valid = stream < P.n_streams;
  const uint32_t row = C.valid ? stream : P.n_streams - 1;
  C.fld = P.n_streams * 4u;
  C.rs_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.rs, 0, (int)(C.fld * RF_COUNT), 0x00020000);
  C.cf_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)S.coef, 0, (int)(2u * C.fld * CF_COUNT), 0x00020000);
  C.M.is_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.is, 0, (int)(C.fld * IF_COUNT), 0x00020000);
  C.M.fld = C.fld;
  C.M.voff = C.valid ? row * 4u : 0xFFFFFFF0u;
  C.M.avoff = C.valid ? row * 16u : 0xFFFFFFF0u;
  C.row4 = row * 4u;
  return C;
}

// NCO phase of the free-running frame at the first sample of the launch = the stream's NCO phase minus its frame offset.
// Uniform configuration: every stream of the batch shares the frame, so this is a wave-uniform value (SGPRs).
template <bool UNI, int COH = 0>
__device__ inline uint64_t pipe_free0(const PipeCtx &C) {
  const FastMem &M = C.M;
  const uint32_t fld = C.fld, row4 = C.row4;
  const uint64_t acc = ((uint64_t)PIPE_ILOAD(nco_hi) << 32) | PIPE_ILOAD(nco_lo);
  const uint64_t off = ((uint64_t)PIPE_ILOAD(fr_hi) << 32) | PIPE_ILOAD(fr_lo);
  const uint64_t f = acc - off;
  if (!UNI) return f;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)f);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(f >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// front state (AGC, pre-filter, free-running I/Q low-pass)
template <bool UNI, int COH = 0>
__device__ inline void front_load(FrontLane &F, FrontK &K, const DemodParams &P, const DemodState &S, const PipeCtx &C) {
  const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc, cf_rsrc = C.cf_rsrc;
  const uint32_t fld = C.fld, row4 = C.row4;
  F.g = PIPE_RLOAD(agc_gain);
  F.bx1 = PIPE_RLOAD(bp_x1); F.bx2 = PIPE_RLOAD(bp_x2); F.by1 = PIPE_RLOAD(bp_y1); F.by2 = PIPE_RLOAD(bp_y2);
  F.ix1 = PIPE_RLOAD(li_x1); F.ix2 = PIPE_RLOAD(li_x2); F.iy = PIPE_RLOAD(li_y1); F.iv = PIPE_RLOAD(li_y2);
  F.qx1 = PIPE_RLOAD(lq_x1); F.qx2 = PIPE_RLOAD(lq_x2); F.qy = PIPE_RLOAD(lq_y1); F.qv = PIPE_RLOAD(lq_y2);
  K.att_m_rel = P.f_agc_att - P.f_agc_rel; K.rel = P.f_agc_rel;
  K.step_k = 1099511627776.0f; K.step_b = -549755813888.0f;
  K.g_lo = 0.1f; K.g_hi = 10.0f;
  if (UNI) {
    K.bp_b0 = P.u_bp_b0h * kIqScale; K.bp_na1 = P.u_bp_na1; K.bp_na2 = P.u_bp_na2;
  } else {
    K.bp_b0 = (float)(PIPE_CLOAD(CF_bp_b0) * (0.5 * P.lp_b0)) * kIqScale;
    K.bp_na1 = -(float)PIPE_CLOAD(CF_bp_a1); K.bp_na2 = -(float)PIPE_CLOAD(CF_bp_a2);
  }
  K.lp_a2 = P.f_lp_a2; K.lp_nd = -P.f_lp_delta;
  K.tiny = 0x1p-123f; K.sgn = 0x80000000u;
  asm volatile("" : "+v"(K.tiny), "+v"(K.sgn));
  asm volatile("" : "+v"(K.att_m_rel), "+v"(K.rel), "+v"(K.step_k), "+v"(K.step_b), "+v"(K.g_lo), "+v"(K.g_hi));
  asm volatile("" : "+v"(K.bp_b0), "+v"(K.bp_na1), "+v"(K.bp_na2), "+v"(K.lp_a2), "+v"(K.lp_nd));
}

// the back's constants as the parameters give them (scalars), and pinned into VGPRs: an SGPR operand halves a vector
// instruction's rate, which matters where four waves share a SIMD (the hot paths) and not where a wave runs alone or
// rarely (fsk_blk.hip's block path with resets and per-sample path use them unpinned and leave the registers to the data)
__device__ inline void back_consts(BackK &K, const DemodParams &P) {
  K.c1 = P.z_c1; K.c2 = P.z_c2;
  K.lp_b0 = P.f_lp_b0; K.lp_a2 = P.f_lp_a2; K.lp_nd = -P.f_lp_delta;
  K.qn = ~(uint32_t)P.pat_q; K.mask = (uint32_t)P.pat_mask; K.d = P.d;
  K.tiny = 0x1p-123f; K.sgn = 0x80000000u; K.edge_min = (1u << P.stop_pos) - 2u; K.eod_m1 = P.eod_min - 1u; K.zk = -0x1p123f; K.unscale = kIqUnscale;
}
__device__ inline void back_consts_pin(BackK &K) {
  asm volatile("" : "+v"(K.edge_min), "+v"(K.eod_m1), "+v"(K.zk), "+v"(K.unscale));
  asm volatile("" : "+v"(K.c1), "+v"(K.c2), "+v"(K.lp_b0), "+v"(K.lp_a2), "+v"(K.lp_nd), "+v"(K.qn), "+v"(K.mask), "+v"(K.d));
  asm volatile("" : "+v"(K.tiny), "+v"(K.sgn));
}

template <bool UNI, int COH = 0>
__device__ inline void back_load(BackLane &B, BackK &K, const DemodParams &P, const DemodState &S, const PipeCtx &C,
                                 uint32_t stream, uint32_t *out_counts, uint32_t *eod_counts, int append) {
  const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
  const FastMem &M = C.M;
  const uint32_t fld = C.fld, row4 = C.row4;
  B.qai = PIPE_RLOAD(zq_ai); B.qaq = PIPE_RLOAD(zq_aq); B.qbi = PIPE_RLOAD(zq_bi); B.qbq = PIPE_RLOAD(zq_bq);
  B.q0i = PIPE_RLOAD(zq_0i); B.q0q = PIPE_RLOAD(zq_0q);
  B.px1 = PIPE_RLOAD(po_x1); B.px2 = PIPE_RLOAD(po_x2); B.py = PIPE_RLOAD(po_y1); B.pv = PIPE_RLOAD(po_y2);
  B.last_phase = PIPE_RLOAD(last_phase);
  {
    const uint64_t off = ((uint64_t)PIPE_ILOAD(fr_hi) << 32) | PIPE_ILOAD(fr_lo);
    double r = (double)(0ull - off) * 5.42101086242752217e-20 * 6.283185307179586476925;   // as back_reset computes it
    r = r > 3.14159265358979323846 ? r - 6.283185307179586476925 : r;
    B.thf = (float)r;
  }
  B.thr = PIPE_RLOAD(sil_thr);
  B.dph = PIPE_ILOAD(zr_dph);
  B.dix1 = PIPE_RLOAD(zd_ix1); B.dix2 = PIPE_RLOAD(zd_ix2); B.diy = PIPE_RLOAD(zd_iy); B.dvi = PIPE_RLOAD(zd_iv);
  B.dqx1 = PIPE_RLOAD(zd_qx1); B.dqx2 = PIPE_RLOAD(zd_qx2); B.dqy = PIPE_RLOAD(zd_qy); B.dqv = PIPE_RLOAD(zd_qv);
  B.matched = PIPE_ILOAD(matched);
  B.thr_eff = PIPE_ILOAD(started) ? kStartedP : P.matched_min;
  {
    const uint32_t cc = PIPE_ILOAD(cad_ctr);
    B.rho = cc ? P.cadence - cc : 0u;
  }
  B.ls = 0u - PIPE_ILOAD(sil_cnt);
  B.acc = PIPE_ILOAD(bit_acc);
  B.T = PIPE_ILOAD(bit_wait);
  B.tlast = B.T - PIPE_ILOAD(bit_reload);
  {
    // byteState (fsk.ts:125) -> shift register: position p, bits received so far under a sentinel at bit p
    const uint32_t pos = PIPE_ILOAD(bit_pos), bc = PIPE_ILOAD(byte_cur);
    B.sreg = pos <= 9u ? (1u << pos) | ((bc & 0x1FFu) >> (9u - pos)) : (1u << 10) | ((bc & 0x1FFu) << 1) | (bc >> 31);
  }
  if (B.thr_eff != kStartedP) { B.T = kBigWait; B.tlast = B.T - PIPE_ILOAD(bit_reload); }  // (re)park: decisions imply a started frame
  // append: a preceding launch of the same call (head samples up to a pair / 16-byte boundary) has produced output already
  B.out_cnt = (append && C.valid) ? (COH ? __hip_atomic_load(&out_counts[stream], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : out_counts[stream]) : 0u;
  if (!append && C.valid && eod_counts) { if (COH) __hip_atomic_store(&eod_counts[stream], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else eod_counts[stream] = 0; }  // incremented in memory by the (rare) EOD path
  if (!C.valid) {
    // Lanes beyond the batch run on zeros with a copy of the last stream's state.  Park them: no sync candidate (a
    // threshold `matched` cannot reach), no silence run (nothing is below a negative threshold), no bit clock -- so
    // they never enter a rare path, where their out-of-range row index would be used as an address.
    B.thr_eff = 0x7FFFFFFEu; B.thr = -1.0f; B.T = kBigWait; B.tlast = B.T;
  }
  back_consts(K, P);
  back_consts_pin(K);
}

// everything back to the state arrays.  F: the front's final state, n: samples of the launch, k: decimated samples,
// kappa = k % cadence, free0: the free-running frame's NCO phase at the first sample of the launch.
template <bool UNI, int COH = 0>
__device__ inline void pipe_store(const FrontLane &F, bool store_front, const BackLane &B, const DemodParams &P,
                                  const PipeCtx &C, uint32_t stream, uint32_t *out_counts, size_t n, uint32_t k,
                                  uint32_t kappa, uint32_t phase, uint32_t amp_pos, uint64_t inc, uint64_t free0) {
  const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
  const FastMem &M = C.M;
  const uint32_t fld = C.fld, row4 = C.row4;
  if (store_front) {
    PIPE_RSTORE(agc_gain, F.g);
    PIPE_RSTORE(bp_x1, F.bx1); PIPE_RSTORE(bp_x2, F.bx2); PIPE_RSTORE(bp_y1, F.by1); PIPE_RSTORE(bp_y2, F.by2);
  }
  // free-running I/Q low-pass, the correction, lastPhase in the free frame; NCO phase = free phase + frame offset
  PIPE_RSTORE(li_x1, F.ix1); PIPE_RSTORE(li_x2, F.ix2); PIPE_RSTORE(li_y1, F.iy); PIPE_RSTORE(li_y2, F.iv);
  PIPE_RSTORE(lq_x1, F.qx1); PIPE_RSTORE(lq_x2, F.qx2); PIPE_RSTORE(lq_y1, F.qy); PIPE_RSTORE(lq_y2, F.qv);
  {
    // the correction's four values are dead while the direct instance runs (formed from its last two samples, first used at
    // zr_dph = kDirectPairs): zeros are stored, so that the state arrays do not depend on the kernel (the one- and two-wave
    // kernels' recurrence runs on over the span, the block kernels park it)
    const bool dead = B.dph < kDirectPairs;
    PIPE_RSTORE(zq_ai, dead ? 0.f : B.qai); PIPE_RSTORE(zq_aq, dead ? 0.f : B.qaq);
    PIPE_RSTORE(zq_bi, dead ? 0.f : B.qbi); PIPE_RSTORE(zq_bq, dead ? 0.f : B.qbq);
  }
  PIPE_RSTORE(zq_0i, B.q0i); PIPE_RSTORE(zq_0q, B.q0q);
  PIPE_ISTORE(zr_dph, B.dph);
  {
    // the direct instance's registers mean something only while it runs (zr_dph < kDirectPairs); after that zeros are
    // stored, so that the state arrays do not depend on which path took the samples in between (fsk_blk.hip's block path
    // with resets lets the instance run on for lanes that are past it)
    const bool live = B.dph < kDirectPairs;
    PIPE_RSTORE(zd_ix1, live ? B.dix1 : 0.f); PIPE_RSTORE(zd_ix2, live ? B.dix2 : 0.f);
    PIPE_RSTORE(zd_iy, live ? B.diy : 0.f); PIPE_RSTORE(zd_iv, live ? B.dvi : 0.f);
    PIPE_RSTORE(zd_qx1, live ? B.dqx1 : 0.f); PIPE_RSTORE(zd_qx2, live ? B.dqx2 : 0.f);
    PIPE_RSTORE(zd_qy, live ? B.dqy : 0.f); PIPE_RSTORE(zd_qv, live ? B.dqv : 0.f);
  }
  PIPE_RSTORE(last_phase, B.last_phase);
  {
    const uint64_t off = ((uint64_t)PIPE_ILOAD(fr_hi) << 32) | PIPE_ILOAD(fr_lo);
    const uint64_t acc = free0 + inc * (uint64_t)n + off;
    PIPE_ISTORE(nco_lo, (uint32_t)acc); PIPE_ISTORE(nco_hi, (uint32_t)(acc >> 32));
  }
  PIPE_RSTORE(po_x1, B.px1); PIPE_RSTORE(po_x2, B.px2); PIPE_RSTORE(po_y1, B.py); PIPE_RSTORE(po_y2, B.pv);
  PIPE_RSTORE(sil_thr, B.thr);
  {
    const uint32_t cc = kappa + P.cadence - B.rho;
    PIPE_ISTORE(cad_ctr, cc >= P.cadence ? cc - P.cadence : cc);
  }
  PIPE_ISTORE(sil_cnt, k - B.ls);
  PIPE_ISTORE(bit_acc, B.acc); PIPE_ISTORE(bit_wait, B.T - k); PIPE_ISTORE(bit_reload, B.T - B.tlast);
  {
    const uint32_t pos = 31u - (uint32_t)__builtin_clz(B.sreg | 1u);
    const uint32_t bc = pos <= 9u ? (B.sreg & ((1u << pos) - 1u)) << (9u - pos) : ((B.sreg >> 1) & 0x1FFu) | ((B.sreg & 1u) << 31);
    PIPE_ISTORE(byte_cur, bc); PIPE_ISTORE(bit_pos, pos);
  }
  PIPE_ISTORE(started, B.thr_eff == kStartedP ? 1u : 0u); PIPE_ISTORE(matched, B.matched);
  PIPE_ISTORE(gsc, k + PIPE_ILOAD(gsc));              // the gsc word held the offset during the launch
  const uint32_t rl = PIPE_ILOAD(ring_len) + k, al = PIPE_ILOAD(amp_len) + k;
  PIPE_ISTORE(ring_len, rl < P.ring_cap ? rl : P.ring_cap);
  PIPE_ISTORE(amp_len, al < P.amp_cap ? al : P.amp_cap);
  PIPE_ISTORE(poly_phase, phase);
  PIPE_ISTORE(amp_pos, amp_pos);
  if (C.valid) { if (COH) __hip_atomic_store(&out_counts[stream], B.out_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else out_counts[stream] = B.out_cnt; }
}

// wave-uniform LDS word, polled by the other wave of the workgroup
__device__ inline uint32_t lds_peek(const uint32_t *p) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// All four hand-off counters with ONE LDS read whose latency hides behind the caller's own work: issue it first in an
// iteration; the wait inside the iteration's lds_post covers it (LDS operations of a wave return in order), after which
// lds_peek4_get hands out the values.  They are up to one iteration old -- counters only grow, so stale is safe.
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__device__ inline void lds_peek4_begin(const uint32_t *ctr, v4u32 &c) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(c) : "v"((uint32_t)(uintptr_t)ctr) : "memory");
}
__device__ inline uint32_t lds_peek4_get(v4u32 &c, int i) {
  // (the wait is this helper's own: every caller has an lds_post -- which waits -- between begin and get, so it is
  // already satisfied and costs nothing; it makes the dependency structural instead of a property of the callers,
  // ADVICE r03.  tools/check_isa.py still proves from the ISA that nothing touches the registers before a wait.)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c));
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(i == 0 ? c.x : i == 1 ? c.y : i == 2 ? c.z : c.w));
}
__device__ inline void lds_post(uint32_t *p, uint32_t v) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}

}  // namespace fsk
