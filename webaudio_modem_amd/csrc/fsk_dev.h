// fsk_dev.h -- device helpers shared by the demodulator translation units (fsk_demod.hip: generic + r01 whole-tile
// kernels; fsk_pipe.hip: the free-running front / ZIR-corrected back kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fsk_params.h"

namespace fsk {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

static constexpr int kFastTile = 16;
// Cache policy of the accesses to what one time slice of a persistent launch (fsk_blk.hip) hands to the next -- possibly
// on another XCD, behind another L2: sc1 = coherent at device scope (stores write through, loads take no cached copy).
// Only those launches use it (template parameter COH of the state helpers, 0 everywhere else): between launches the
// kernel boundary does the job, and the write-through costs -- a 128-sample FSKProcessor quantum moves about as many
// bytes of state as of samples and ran 0.20 instead of 0.14 ms when every kernel carried it.
// INVARIANT (tests/test_abi.py greps for it): inside fsk_blk.hip every helper that touches handed-on state is instantiated
// with the kernel's COH, and a field added to the state must go through those helpers (PIPE_* / ist_* / back_*): an access
// without it would be stale on another XCD only now and then.  The hand-over itself: every storing wave's `s_waitcnt
// vmcnt(0)`, the workgroup's barrier, then the queue push (MI355X_MICROARCH.md, "valid forms": all stores and loads sc1).
static constexpr int kCohSc1 = 16;
static constexpr uint32_t kStarted = 0xFFFFFFFFu;  // thr_eff while a frame is started (matched_min is <= 0xFFFFFFFE)

__device__ inline uint32_t popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
__device__ inline uint32_t popc(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Low-pass biquad in "velocity" form.  With v = y[n-1] - y[n-2] kept as state,
//   y[n] = y[n-1] + a2*v + (b0*(x + 2*x1 + x2) - delta*y[n-1]),   delta = 1 + a1 + a2
// is algebraically the reference's Direct Form I (filters.ts:47-76) but the poles sit close to
// z = 1 (cutoff = baud << fs), where DF-I in f32 amplifies both coefficient and state rounding by
// 1/|A(1)| ~ 700 (300 baud); here the rounding of y is fed back only through delta ~ 1.5e-3.
// State: y holds y[n-1], v holds the velocity.
// GAIN = false is the variant whose input already carries the b0 gain (the I/Q filters).
template <bool GAIN = true>
__device__ inline float lp32(float b0, float a2, float delta, float &x1, float &x2, float &y, float &v, float x) {
  float t = __builtin_fmaf(2.0f, x1, x) + x2;
  float u = GAIN ? __builtin_fmaf(-delta, y, b0 * t) : __builtin_fmaf(-delta, y, t);
  v = __builtin_fmaf(a2, v, u);
  y = y + v;
  x2 = x1; x1 = x;
  return y;
}

// atan2 for the discriminator: |error| <= 1.5e-7 rad.  min/max ratio through v_rcp_f32, odd
// minimax polynomial on [0,1] (coefficients fitted for this file), quadrant by compares.  -0 counts
// as +0 for x (the reference's averages are never -0: its sums start at +0), atan2(0, 0) = 0.
// Also returns the magnitude sqrt(x^2 + y^2) as max * sqrt(1 + (min/max)^2) -- same instruction count as squaring,
// but it cannot underflow: in the exact-zero tail after a frame the I/Q averages decay through 1e-20 .. 1e-38, their
// squares are zero in fp32 long before the values are, and a (false) sync detected there would set the silence
// threshold to the mean of zeros where the reference's doubles still see the decaying amplitudes (found by
// tools/soak.py: eod counts diverged).  Below fp32's own range (~1e-38) the paths still differ; see DESIGN.md.
__device__ inline float atan2_amp_fast(float y, float x, float &amp) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
  const float a = mn * __builtin_amdgcn_rcpf(__builtin_fmaxf(mx, 1.0e-37f));
  const float s = a * a;
  amp = mx * __builtin_amdgcn_sqrtf(s + 1.0f);
  float p = -4.355408570e-03f;
  p = __builtin_fmaf(p, s, 2.304014596e-02f);
  p = __builtin_fmaf(p, s, -5.777360382e-02f);
  p = __builtin_fmaf(p, s, 9.794235514e-02f);
  p = __builtin_fmaf(p, s, -1.397658244e-01f);
  p = __builtin_fmaf(p, s, 1.996270403e-01f);
  p = __builtin_fmaf(p, s, -3.333165903e-01f);
  float r = __builtin_fmaf(a * s, p, a);
  r = ay > ax ? 1.57079632679489662f - r : r;
  r = x < 0.0f ? 3.14159265358979323846f - r : r;
  return __builtin_copysignf(r, y);
}

// How the whole-tile kernels' rare paths reach the per-stream integer state in HBM: one buffer descriptor (SGPRs), the
// field offset as the SGPR soffset, one VGPR lane offset (row*4, or out of range for lanes beyond the batch, whose
// stores the bounds check then drops).
struct FastMem {
  __amdgpu_buffer_rsrc_t is_rsrc;    // integer state [IF_COUNT][n_streams]
  uint32_t fld;                      // bytes per field
  uint32_t voff;                     // row*4, or 0xFFFFFFF0 for lanes beyond the batch
  uint32_t avoff;                    // row*16 (the stream's quad in an amplitude-ring row), or 0xFFFFFFF0
};

// syncAmplitudeBuffer storage (fsk.ts:150): [slot / 4][stream][slot % 4] floats -- four consecutive ring slots of a
// stream are one 16-byte quad, so that the block kernel stores a tile's eight amplitudes with two 16-byte stores per lane
// (a store instruction costs the issuing wave ~35 cycles whatever its width).  amp_cap = 8 dsSPB is a multiple of four
// for every integer dsSPB.
__host__ __device__ inline size_t amp_index(uint32_t pos, uint32_t row, uint32_t n_streams) {
  return ((size_t)(pos >> 2) * n_streams + row) * 4u + (pos & 3u);
}
// the same as a byte offset split into the scalar part (quad row + slot within the quad) the whole-tile kernels carry
__device__ inline uint32_t amp_soff_of(uint32_t pos, uint32_t quad_bytes) { return (pos >> 2) * quad_bytes + (pos & 3u) * 4u; }
__device__ inline uint32_t amp_pos_of(uint32_t soff, uint32_t quad_bytes) { return (soff / quad_bytes) * 4u + ((soff % quad_bytes) >> 2); }
__device__ inline void amp_advance(uint32_t &soff, uint32_t quad_bytes, uint32_t wrap) {
  soff += 4u;
  if ((soff & 12u) == 0u) { soff += quad_bytes - 16u; if (soff == wrap) soff = 0u; }
}
template <int COH = 0>
__device__ inline uint32_t ist_load(const FastMem &M, uint32_t field) {
  return __builtin_amdgcn_raw_buffer_load_b32(M.is_rsrc, M.voff, field * M.fld, COH);
}
template <int COH = 0>
__device__ inline void ist_store(const FastMem &M, uint32_t field, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, M.is_rsrc, M.voff, field * M.fld, COH);
}
// counter += v without waiting for the old value (a load + store pair stalls the issuing wave for a memory round trip:
// ~17 % of the back wave's time on an idle receiver bank, where some lane of a group fires an 'eod' every other decimated sample)
template <int COH = 0>
__device__ inline void ist_add(const FastMem &M, uint32_t field, uint32_t v) {
  (void)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)v, M.is_rsrc, M.voff, field * M.fld, COH);
}


// Intermediate capture of the pre-filter's output (fsk.ts:202: the Float32Array the band-pass returns), one value per INPUT
// sample of the traced stream, behind the post-filter trace in the same buffer: trace_post[cap .. 3 cap), counted by
// trace_n[1] (the kernels' argument block stays as it is).  Called by the traced stream's lane only.
__device__ inline void trace_pre_put(const DemodState &S, double v) {
  const uint32_t kk = S.trace_n[1];
  if (kk < 2u * S.trace_cap) S.trace_post[S.trace_cap + kk] = v;
  S.trace_n[1] = kk + 1u;
}

// ---- opt-in signal-quality estimates (definition: include/fskhip.h) -- rare paths only, state read-modify-written -----
template <typename Real>
__device__ inline Real &q_real(const DemodState &S, uint32_t n, int field, uint32_t row) { return ((Real *)S.rs)[(size_t)field * n + row]; }
__device__ inline uint32_t &q_int(const DemodState &S, uint32_t n, int field, uint32_t row) { return S.is[(size_t)field * n + row]; }

template <typename Real>
__device__ inline void quality_on_sync(const DemodParams &P, const DemodState &S, uint32_t row, double mean_amp) {
  q_real<Real>(S, P.n_streams, RF_q_signal, row) = (Real)mean_amp;
  q_int(S, P.n_streams, IF_q_armed, row) = 1u;
  q_int(S, P.n_streams, IF_q_prev_d0, row) = P.q_last_d0;
}
template <typename Real>
__device__ inline void quality_on_start(const DemodParams &P, const DemodState &S, uint32_t row, Real post) {
  if (q_int(S, P.n_streams, IF_q_prev_d0, row) == 0u) {
    q_real<Real>(S, P.n_streams, RF_q_f0_sum, row) += post;
    q_int(S, P.n_streams, IF_q_starts, row) += 1u;
  }
}
template <typename Real>
__device__ inline void quality_on_byte(const DemodParams &P, const DemodState &S, uint32_t row, uint32_t byte, uint32_t ones,
                                       uint32_t cnt, Real post) {
  const uint32_t n = P.n_streams, zeros = cnt - ones;
  const int32_t margin = (int32_t)(2u * ones) - (int32_t)cnt;
  q_real<Real>(S, n, RF_q_eye_sum, row) += (Real)(margin < 0 ? -margin : margin) / (Real)cnt;
  q_int(S, n, IF_q_minor, row) += ones < zeros ? ones : zeros;
  q_int(S, n, IF_q_votes, row) += cnt;
  if ((byte & 3u) == 2u) {
    q_real<Real>(S, n, RF_q_f_sum, row) += post;
    q_real<Real>(S, n, RF_q_f2_sum, row) += post * post;
    q_int(S, n, IF_q_ftrans, row) += 1u;
  }
  q_int(S, n, IF_q_prev_d0, row) = byte & 1u;
  q_int(S, n, IF_q_bytes, row) += 1u;
}
// First 'eod' after a sync: mean of the newest min(q_eod_n, len) amplitudes.  Whole wave; `hit` marks the lanes at an
// 'eod', newest = ring index of the amplitude just stored, len = entries in the ring.
template <typename Real>
__device__ inline void quality_on_eod(const DemodParams &P, const DemodState &S, uint32_t lane, uint32_t row, bool hit,
                                      uint32_t newest, uint32_t len) {
  const uint32_t n = P.n_streams;
  const bool armed = hit && q_int(S, n, IF_q_armed, row) != 0u;
  uint64_t m = __builtin_amdgcn_ballot_w64(armed);
  if (!m) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's ring stores have reached L2
  while (m) {
    const int src = __builtin_ctzll(m);
    m &= m - 1;
    const uint32_t srow = (uint32_t)__builtin_amdgcn_readlane((int)row, src);
    const uint32_t snew = (uint32_t)__builtin_amdgcn_readlane((int)newest, src);
    uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)len, src);
    cnt = cnt < P.q_eod_n ? cnt : P.q_eod_n;
    double part = 0.0;
    for (uint32_t i = lane; i < cnt; i += 64) {
      const uint32_t pos = snew >= i ? snew - i : snew + P.amp_cap - i;
      const float *p = S.amp_ring + amp_index(pos, srow, n);
      part += (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double sum = wave_sum(part);
    if ((int)lane == src) {
      q_real<Real>(S, n, RF_q_floor, row) = (Real)(cnt ? sum / (double)cnt : 0.0);
      q_int(S, n, IF_q_frames, row) += 1u;
      q_int(S, n, IF_q_armed, row) = 0u;
    }
  }
}

}  // namespace fsk
