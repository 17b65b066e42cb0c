// fsk_pipe.hip -- round-2 whole-tile fp32 demodulator kernels for gfx950 (MI355X): a FREE-RUNNING front end and a
// back end that repairs resetState() by linearity, so that the two can run as a decoupled wave pipeline.
//
// Why.  valu_probe (tools/, profiles/r02_valu_probe.txt): one wave issues a vector instruction every ~4 cycles, a SIMD
// retires one every ~2 once two or more waves interleave, packed fp32 costs twice a plain op and transcendentals four
// times.  At 65 536 streams (BASELINE config #3 on one GPU) one lane per stream gives each SIMD ONE wave, so the r01
// kernels ran at half the SIMD's rate whatever their instruction count.  Splitting a 64-stream group over two waves
// needs a cut with no feedback across it -- and the reference has one: processDownsampledBit -> resetState()
// (fsk.ts:175-188, 288-291, 352-355) zeroes the NCO phase and the I/Q low-pass state from inside the frame logic.
//
// How.  The I/Q branch is linear, so the front wave never resets anything:
//   front (per input sample):  AGC (fsk.ts:52-76) -> pre-filter (filters.ts:47-87) -> mix with the NCO of a frame that
//                              started at phase 0 when the launch did -> I/Q low-pass -> pair sums U[m]  (fsk.ts:228-248)
//   back (per decimated sample): w[m] = U[m] - q[m], discriminator, post filter, slicer (fsk.ts:251-264) and the whole
//                              frame state machine (fsk.ts:278-375).
// After a reset the back wave computes the next kDirectPairs = 18 decimated samples itself, with a second, zero-started
// filter instance fed with the same pre-filter outputs (W_direct).  kZeroLagPairs = 16 decimated samples after the
// reset -- the front wave is at most 16 ahead, and the lag is counted in the stream's own samples, so it does not
// depend on how the stream is cut into launches -- the front zeroes that stream's filters (the back posts the position
// in an LDS mailbox).  From there U lacks only the direct instance's memory at that point, a zero-input response q[m]
// of the size of the signal itself: as a pair sum, with Z[n] the response of  y[n] = -a1 y[n-1] - a2 y[n-2],  it obeys
// q[m+2] = (a1^2 - 2 a2) q[m+1] - a2^2 q[m],  and its first two values are U - W_direct of the direct instance's last
// two samples.  (Subtracting the response of the state AT the reset instead, without ever zeroing the front, is the
// same algebra but not the same arithmetic: where the input has dropped by more than ~1e-7 before the reset --
// the ringing after a frame followed by digital silence -- U - q cancels to rounding noise; tools/soak.py found that.)
// What the reference's restarted NCO changes on top is a constant rotation e^{-j w n0} of the I/Q plane, which the
// amplitude does not see and the phase DIFFERENCE only sees once: lastPhase = 0 in the reference's frame is w*n0 in the
// free-running one.  tools/zir_model.py checks the algebra against a sample-serial model.
//
// Because the front's frame is uniform over the batch when all streams share one configuration, its NCO is not per-lane
// arithmetic: sixteen lanes evaluate e^{j w n} for the sixteen samples of a tile (v_cos / v_sin of the exact 64-bit turn
// accumulator, as in round 1), park them in LDS, and every lane reads them back as broadcasts -- VGPR operands for
// the mixer (an operand from an SGPR would make every multiply a half-rate instruction, profiles/r02_valu_probe_summary.md).
//
// State.  fp32 engines keep this representation in HBM between launches (fsk_params.h: li_*/lq_*/last_phase are the
// free-running frame's, fr_* the frame offset, zq_*/zd_*/zr_dph the correction), so a stream cut into launches
// anywhere computes bit for bit what one launch computes; the generic kernel (fsk_demod.hip), which runs
// ragged tails and the uncommon configurations with real resets, converts on load and store (pipe_to_actual /
// actual_to_pipe in fsk_dev.h, f64 rotation).
//
// Kernels:
//   demod_pipe_kernel   two waves per 64-stream group: wave 0 front, wave 1 back, hand-off through an LDS ring of
//                       8-sample half tiles with producer/consumer counters (no barrier in the loop).  For batches
//                       that give a SIMD fewer than ~3 waves (BASELINE configs #2, #3, #5).
//   (demod_pipe3_kernel, three waves per group, was retired in round 4: fsk_blk.hip's four-wave kernel covers every
//                       batch size it was selected for.)
//   demod_fused_kernel  the same two halves called back to back by one wave, pair by pair through registers, for
//                       batches large enough to fill the SIMDs with one wave per group.
//   demod_tail_kernel   the same arithmetic one sample at a time: heads and tails of calls, traced engines, engines with
//                       the signal-quality estimates switched on.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fsk_params.h"
#include "fsk_dev.h"
#include "fsk_pipe_dev.h"

namespace fsk {
// these kernels hand their state on across kernel boundaries only: plain cache policy for the PIPE_* accesses (fsk_dev.h)
static constexpr int COH = 0;
}

namespace fsk {

// ================================================================================================================
// Two waves per 64-stream group.
// LDS: stage [4][65] v4f | ring [kPipeSlots][6][64] v4f | fin [2][64] v4f | zt [2][8] v4f | poly [d][64] u32 | counters | zmail [64] u32
// ================================================================================================================
template <bool WB, bool UNI>
__global__ __launch_bounds__(128) void demod_pipe_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n, size_t pitch, int append,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts) {
  FSK_ABL_INIT
  FSK_STAMP_DECL
  FSK_WAIT_DECL
  extern __shared__ float4 lds[];
  v4f *stage = reinterpret_cast<v4f *>(lds);
  v4f *ring = stage + 4 * kSlotStride;
  v4f *fin = ring + kPipeSlots * kSlotV4;
  v4f *zt = fin + 2 * 64;                                 // NCO phasors of two tiles: [2][16 samples] x (cos, sin)
  uint32_t *poly = reinterpret_cast<uint32_t *>(zt + 2 * 8);
  uint32_t *ctr = poly + 64u * P.d;                       // [0] tiles produced, [1] tiles consumed
  uint32_t *zmail = ctr + 4;                              // [64] back -> front: where to zero a lane's I/Q low-pass
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)blockIdx.x * P.d * 64u;

  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const PipeCtx C = pipe_ctx(P, S, stream);
  const size_t n_tiles = n / kFastTile;
  const uint64_t inc = UNI ? (((uint64_t)P.u_inc_hi << 32) | P.u_inc_lo) : S.nco_inc[C.row4 >> 2];
  const uint64_t free0 = pipe_free0<UNI>(C);

  if (threadIdx.x == 0) { ctr[0] = 0; ctr[1] = 0; }
  if (wave == 0) {
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    zmail[lane] = zmail_init(PIPE_ILOAD(zr_dph));
  }
  __syncthreads();

  if (wave == 0) {
    // ---------------------------------------------------------------------------------------------- front
    FrontLane F;
    FrontK K;
    front_load<UNI>(F, K, P, S, C);
    // per-lane NCO (per-stream tone pairs): phasor recurrence, re-seeded from the exact accumulator at every tile
    float wre = 1.f, wim = 0.f;
    if (!UNI) {
      const __amdgpu_buffer_rsrc_t cf_rsrc = C.cf_rsrc;
      const uint32_t fld = C.fld, row4 = C.row4;
      wre = (float)PIPE_CLOAD(CF_w1_re); wim = (float)PIPE_CLOAD(CF_w1_im);
    }
    const uint32_t sub_row = lane >> 2, chunk = lane & 3;
    const uint32_t rows_here = P.n_streams - blockIdx.x * 64u < 64u ? P.n_streams - blockIdx.x * 64u : 64u;
    v4i in_rsrc;
    {
      const uint64_t base = reinterpret_cast<uint64_t>(samples + (size_t)blockIdx.x * 64u * pitch);
      in_rsrc.x = (int)(uint32_t)base;
      in_rsrc.y = (int)(uint32_t)(base >> 32);
      in_rsrc.z = (int)(uint32_t)(rows_here * pitch * 4u);
      in_rsrc.w = 0x00020000;
    }
    const uint32_t in_voff = (uint32_t)((sub_row * pitch + 4u * chunk) * 4u);
    const uint32_t in_row16 = (uint32_t)(16u * pitch * 4u);
    const uint32_t st_slot = chunk * kSlotStride + sub_row;
    // Three tiles in flight, in three register sets used in turn (the loop is unrolled by three so that no set is ever
    // copied).  The loads are inline asm with hand-counted waits, as in demod_fused_kernel: vmcnt counts in issue order
    // and hipcc, which cannot see across the loop's back edge, would drain everything (vmcnt(0)) at the top of every
    // iteration.  When a set is staged, the two sets issued after it (8 loads) may still be in flight, plus this wave's
    // write-back stores of the two tiles in between (4 each) in the write-back variant.
    // (rows beyond the batch read as 0: the row step rides in the bounds-checked VGPR offset)
#define PIPE_BLOAD4(dst, rows16, soff)                                                                      \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(in_voff + (rows16) * in_row16), \
               "s"(in_rsrc), "s"(soff) : "memory")
    auto load_tile = [&](size_t t, v4f &a, v4f &b, v4f &c, v4f &d) {
      const uint32_t tn = (uint32_t)((t < n_tiles ? t : n_tiles - 1) * kFastTile * 4u);
      PIPE_BLOAD4(a, 0u, tn); PIPE_BLOAD4(b, 1u, tn); PIPE_BLOAD4(c, 2u, tn); PIPE_BLOAD4(d, 3u, tn);
    };
    v4f a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the state loads above are complete, the count starts clean
    load_tile(0, a0, a1, a2, a3);
    load_tile(1, b0, b1, b2, b3);
    load_tile(2, c0, c1, c2, c3);
    // the loop may receive the three sets in other registers than these loads were issued into: any such copy must see
    // landed data (inside the loop tools/check_isa.py proves that nothing touches a set between issue and its wait)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
    // NCO phasors (uniform configuration): lane j & 15 evaluates sample j of the tile from the exact accumulator, the
    // sixteen (cos, sin) pairs are parked in LDS and read back as broadcasts
    uint64_t zacc = free0 + inc * (uint64_t)(lane & 15u);     // this lane's sample of the current tile (UNI)
    uint64_t tacc = free0;                                     // first sample of the current tile (per-stream tones)
    const uint64_t inc16 = inc * 16u;
    uint32_t consumed = 0, slot_i = 0;
    auto do_tile = [&](uint32_t t, v4f &r0, v4f &r1, v4f &r2, v4f &r3) {
      if (WB) asm volatile("s_waitcnt vmcnt(16)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      stage[st_slot] = r0; stage[st_slot + 16] = r1; stage[st_slot + 32] = r2; stage[st_slot + 48] = r3;
      load_tile((size_t)t + 3, r0, r1, r2, r3);
      const v4f *ztile = zt + (t & 1u) * 8u;
      float zr = 1.f, zi = 0.f;
      if (UNI) {
        float pc, ps;
        nco_phasor(zacc, pc, ps);
        reinterpret_cast<f2 *>(zt + (t & 1u) * 8u)[lane & 15u] = (f2){pc, ps};
        zacc += inc16;
      } else {
        nco_phasor(tacc, zr, zi);
        tacc += inc16;
      }
#pragma unroll
      for (uint32_t hf = 0; hf < 2; hf++) {
        const uint32_t hidx = 2u * t + hf;                  // half tiles produced so far
        if (hidx - consumed >= kPipeSlots) {
          FSK_STAMP_W0 FSK_WAIT_BEGIN
          while (hidx - consumed >= kPipeSlots) {           // ring full: wait for the back wave
            consumed = lds_peek(&ctr[1]);
            if (hidx - consumed >= kPipeSlots) FSK_SPIN(1, S.blk_stat);
          }
          FSK_STAMP_W1
        }
        v4f *slot = ring + slot_i * kSlotV4;
        slot_i = slot_i + 1u == kPipeSlots ? 0u : slot_i + 1u;
        // a reset the back wave has seen: zero this lane's filters in front of decimated sample zj (rare; one compare per
        // half tile and two scalar branches per four samples otherwise)
        const uint32_t zj = zmail[lane];
        const uint64_t zh = __builtin_amdgcn_ballot_w64(zj - 4u * hidx < 4u);
#pragma unroll
        for (uint32_t cc = 0; cc < 2; cc++) {
          const uint32_t c = 2u * hf + cc;
          const v4f x4 = stage[c * kSlotStride + lane];     // written by this wave: a wave's ds ops are ordered
          const uint32_t pb = 4u * hidx + 2u * cc;
          float zc[4], zs[4];
          if (UNI) {
            const v4f z01 = ztile[c * 2u], z23 = ztile[c * 2u + 1u];   // same address in every lane: LDS broadcast
            zc[0] = z01.x; zs[0] = z01.y; zc[1] = z01.z; zs[1] = z01.w;
            zc[2] = z23.x; zs[2] = z23.y; zc[3] = z23.z; zs[3] = z23.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              zc[j] = zr; zs[j] = zi;
              const float nr = __builtin_fmaf(-zi, wim, zr * wre), ni = __builtin_fmaf(zi, wre, zr * wim);
              zr = nr; zi = ni;
            }
          }
          const float xin[4] = {x4.x, x4.y, x4.z, x4.w};
          float xs[4], y[4], oi[4], oq[4];
          if (FSK_ABL(0)) {
#pragma unroll
            for (int j = 0; j < 4; j++) { xs[j] = y[j] = oi[j] = oq[j] = xin[j] + zc[j]; }
          } else if (__builtin_expect(zh != 0ull, 0)) {   // (its own copy of the four samples: the common one carries no per-lane test)
            asm volatile("s_nop 0");
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (!(j & 1)) front_zero(F, zj == pb + (uint32_t)(j >> 1));
              front_sample(F, K, xin[j], zc[j], zs[j], xs[j], y[j], oi[j], oq[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) front_sample(F, K, xin[j], zc[j], zs[j], xs[j], y[j], oi[j], oq[j]);
          }
          const float u0i = oi[0] + oi[1], u0q = oq[0] + oq[1], u1i = oi[2] + oi[3], u1q = oq[2] + oq[3];
          // the discriminator's phase / magnitude of the uncorrected pair sums, speculatively (see back_pair)
          float am0 = u0i, am1 = u1i, p0 = u0q, p1 = u1q;
          if (!FSK_ABL(0)) {
            p0 = atan2_amp_fma(u0q, u0i, am0, K.tiny, K.sgn);
            p1 = atan2_amp_fma(u1q, u1i, am1, K.tiny, K.sgn);
          }
          slot[cc * 64u + lane] = (v4f){y[0], y[1], y[2], y[3]};
          slot[(2u + cc) * 64u + lane] = (v4f){u0i, u0q, u1i, u1q};
          slot[(4u + cc) * 64u + lane] = (v4f){p0, am0, p1, am1};
          if (WB) {
            if (C.valid)
              *reinterpret_cast<v4f *>(samples + (size_t)(C.row4 >> 2) * pitch + (size_t)t * kFastTile + 4u * c) = (v4f){xs[0], xs[1], xs[2], xs[3]};
          }
        }
        lds_post(&ctr[0], hidx + 1u);                       // this wave's ring writes are done (lgkmcnt(0) inside)
      }
    };
    const uint32_t nt = (uint32_t)n_tiles;
    FSK_STAMP_BEGIN
    for (uint32_t t = 0; t < nt; t += 3) {
      do_tile(t, a0, a1, a2, a3);
      if (t + 1 < nt) do_tile(t + 1, b0, b1, b2, b3);
      if (t + 2 < nt) do_tile(t + 2, c0, c1, c2, c3);
    }
    FSK_STAMP_END(0)
    // the last prefetches are still in flight and their registers are dead to the compiler: keep them until they land
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
#undef PIPE_BLOAD4
    // hand the final I/Q filter state to the back wave, which owns the epilogue
    fin[lane] = (v4f){F.ix1, F.ix2, F.iy, F.iv};
    fin[64u + lane] = (v4f){F.qx1, F.qx2, F.qy, F.qv};
    lds_post(&ctr[0], 2u * nt + 1u);
    {
      const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
      const FastMem &M = C.M;
      const uint32_t fld = C.fld;
      PIPE_RSTORE(agc_gain, F.g);
      PIPE_RSTORE(bp_x1, F.bx1); PIPE_RSTORE(bp_x2, F.bx2); PIPE_RSTORE(bp_y1, F.by1); PIPE_RSTORE(bp_y2, F.by2);
    }
  } else {
    // ---------------------------------------------------------------------------------------------- back
    BackLane B;
    BackK K;
    back_load<UNI>(B, K, P, S, C, stream, out_counts, eod_counts, append);
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
    BackU X;
    X.own_pairs = kHandPairs; X.hand_lag = kHandLag;   // (no hand-over in this kernel: HAND = false)
    X.k = 0; X.kv = 0; X.free0 = free0; X.zmail = zmail; X.cmail = nullptr;
    X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) ? kDirectPairs : 0u;
    X.zlive = __builtin_amdgcn_ballot_w64((B.dph < kHandPairs) | (B.qai != 0.f) | (B.qaq != 0.f) | (B.qbi != 0.f) | (B.qbq != 0.f)) ? 1u : 0u;
    asm volatile("" : "+v"(X.kv));
    X.phase = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(poly_phase));
    const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(amp_pos));
    const uint32_t amp_quad_bytes = P.n_streams * 16u;
    X.amp_soff = amp_soff_of(amp_pos0, amp_quad_bytes);
    const uint32_t amp_wrap = (P.amp_cap >> 2) * amp_quad_bytes;
    const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);
    uint32_t produced = 0, slot_i = 0;
    const uint32_t nh = 2u * (uint32_t)n_tiles;             // half tiles
    FSK_STAMP_BEGIN
#pragma unroll 2
    for (uint32_t t = 0; t < nh; t++) {
      if (produced <= t) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced <= t) {
          produced = lds_peek(&ctr[0]);
          if (produced <= t) FSK_SPIN(1, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      const v4f *slot = ring + slot_i * kSlotV4;
      slot_i = slot_i + 1u == kPipeSlots ? 0u : slot_i + 1u;
#pragma unroll
      for (uint32_t c = 0; c < 2; c++) {
        const v4f u4 = slot[(2u + c) * 64u + lane];
        const v4f pa = slot[(4u + c) * 64u + lane];
        const uint32_t ph0 = X.phase, ph1 = (X.phase + 1 == P.d) ? 0u : X.phase + 1;
        const uint32_t r0 = poly[ph0 * 64u + lane];
        const uint32_t r1 = poly[ph1 * 64u + lane];
        const float *yp = reinterpret_cast<const float *>(&slot[c * 64u + lane]);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          X.k++;
          X.kv += 1u;
          X.phase = h ? ph1 : ph0;
          if (!FSK_ABL(1))
            back_pair<UNI, true>(B, K, P, S, M, &poly[X.phase * 64u + lane], lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, X, h ? u4.z : u4.x,
                                 h ? u4.w : u4.y, yp + 2 * h, h ? r1 : r0, inc, h ? pa.z : pa.x, h ? pa.w : pa.y);
          amp_advance(X.amp_soff, amp_quad_bytes, amp_wrap);
        }
        X.phase = (ph1 + 1 == P.d) ? 0u : ph1 + 1;
      }
      lds_post(&ctr[1], t + 1u);                            // slot free (this wave's reads of it are complete)
    }
    FSK_STAMP_END(1)
    FSK_WAIT_BEGIN
    while (produced <= nh) {
      produced = lds_peek(&ctr[0]);
      if (produced <= nh) FSK_SPIN(1, S.blk_stat);
    }
    FrontLane F;
    {
      const v4f fi = fin[lane], fq = fin[64u + lane];
      F.ix1 = fi.x; F.ix2 = fi.y; F.iy = fi.z; F.iv = fi.w;
      F.qx1 = fq.x; F.qx2 = fq.y; F.qy = fq.z; F.qv = fq.w;
      F.g = F.bx1 = F.bx2 = F.by1 = F.by2 = 0.f;
    }
    for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
    pipe_store<UNI>(F, false, B, P, C, stream, out_counts, n, X.k, X.k % P.cadence, X.phase, amp_pos_of(X.amp_soff, amp_quad_bytes), inc, free0);
  }
}

// ================================================================================================================
// One wave per 64-stream group: the same two halves, pair by pair through registers.
// LDS: stage [4][65] v4f | zt [2][8] v4f | poly [d][64] u32 | zmail [64] u32
// ================================================================================================================
template <bool WB, bool UNI>
__global__ __launch_bounds__(64, 3) void demod_fused_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n, size_t pitch, int append,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts) {
  extern __shared__ float4 lds[];
  v4f *stage = reinterpret_cast<v4f *>(lds);
  v4f *zt = stage + 4 * kSlotStride;
  uint32_t *poly = reinterpret_cast<uint32_t *>(zt + 2 * 8);
  uint32_t *zmail = poly + 64u * P.d;
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)blockIdx.x * P.d * 64u;
  const uint32_t lane = threadIdx.x;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const PipeCtx C = pipe_ctx(P, S, stream);
  const FastMem &M = C.M;
  const uint32_t fld = C.fld, row4 = C.row4;
  const uint64_t inc = UNI ? (((uint64_t)P.u_inc_hi << 32) | P.u_inc_lo) : S.nco_inc[C.row4 >> 2];
  const uint64_t free0 = pipe_free0<UNI>(C);

  FrontLane F;
  FrontK FK;
  front_load<UNI>(F, FK, P, S, C);
  BackLane B;
  BackK BK;
  back_load<UNI>(B, BK, P, S, C, stream, out_counts, eod_counts, append);
  float wre = 1.f, wim = 0.f;
  if (!UNI) {
    const __amdgpu_buffer_rsrc_t cf_rsrc = C.cf_rsrc;
    wre = (float)PIPE_CLOAD(CF_w1_re); wim = (float)PIPE_CLOAD(CF_w1_im);
  }
  for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
  BackU X;
  X.own_pairs = kHandPairs; X.hand_lag = kHandLag;   // (no hand-over in this kernel: HAND = false)
  X.k = 0; X.kv = 0; X.free0 = free0; X.zmail = zmail; X.cmail = nullptr;
  zmail[lane] = zmail_init(B.dph);
  X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) ? kDirectPairs : 0u;
  X.zlive = __builtin_amdgcn_ballot_w64((B.dph < kHandPairs) | (B.qai != 0.f) | (B.qaq != 0.f) | (B.qbi != 0.f) | (B.qbq != 0.f)) ? 1u : 0u;
  asm volatile("" : "+v"(X.kv));
  X.phase = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(poly_phase));
  const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(amp_pos));
  const uint32_t amp_quad_bytes = P.n_streams * 16u;
  X.amp_soff = amp_soff_of(amp_pos0, amp_quad_bytes);
  const uint32_t amp_wrap = (P.amp_cap >> 2) * amp_quad_bytes;
  const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);

  // tile prefetch as in the r01 kernel: inline-asm loads with a hand-counted vmcnt (each tile issues at least 8
  // VMEM operations after its prefetch -- the unconditional ring stores -- so vmcnt(8) retires exactly the loads)
  const uint32_t sub_row = lane >> 2, chunk = lane & 3;
  const uint32_t rows_here = P.n_streams - blockIdx.x * 64u < 64u ? P.n_streams - blockIdx.x * 64u : 64u;
  v4i in_rsrc;
  {
    const uint64_t base = reinterpret_cast<uint64_t>(samples + (size_t)blockIdx.x * 64u * pitch);
    in_rsrc.x = (int)(uint32_t)base;
    in_rsrc.y = (int)(uint32_t)(base >> 32);
    in_rsrc.z = (int)(uint32_t)(rows_here * pitch * 4u);
    in_rsrc.w = 0x00020000;
  }
  const uint32_t in_voff = (uint32_t)((sub_row * pitch + 4u * chunk) * 4u);
  const uint32_t in_row16 = (uint32_t)(16u * pitch * 4u);
  v4f pre0, pre1, pre2, pre3;
#define FSK_BLOAD4(dst, rows16, soff)                                                                      \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(in_voff + (rows16) * in_row16), \
               "s"(in_rsrc), "s"(soff) : "memory")
  {
    const uint32_t s0 = 0u;
    FSK_BLOAD4(pre0, 0u, s0); FSK_BLOAD4(pre1, 1u, s0); FSK_BLOAD4(pre2, 2u, s0); FSK_BLOAD4(pre3, 3u, s0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): state, constants and the first tile are complete
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre0), "+v"(pre1), "+v"(pre2), "+v"(pre3) : : "memory");
  const uint32_t st_slot = chunk * kSlotStride + sub_row;
  // NCO phasors: see demod_pipe_kernel
  const uint64_t zoff = inc * (uint64_t)(lane & 15u);

  for (size_t t0 = 0; t0 < n; t0 += kFastTile) {
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(pre0), "+v"(pre1), "+v"(pre2), "+v"(pre3) : : "memory");
    stage[st_slot] = pre0;
    stage[st_slot + 16] = pre1;
    stage[st_slot + 32] = pre2;
    stage[st_slot + 48] = pre3;
    const v4f *ztile = zt + ((t0 >> 4) & 1) * 8u;
    float zr = 1.f, zi = 0.f;
    if (UNI) {
      float pc, ps;
      nco_phasor(free0 + inc * (uint64_t)t0 + zoff, pc, ps);
      reinterpret_cast<f2 *>(zt + ((t0 >> 4) & 1) * 8u)[lane & 15u] = (f2){pc, ps};
    } else {
      nco_phasor(free0 + inc * (uint64_t)t0, zr, zi);
    }
    // a reset some lane has seen: zero its filters in front of decimated sample zj (see demod_pipe_kernel; posted at
    // least kZeroLagPairs = 16 decimated samples ahead, so one look per tile is early enough)
    const uint32_t zj = zmail[lane];
#ifdef FSK_EXP_NOZ
    const uint32_t zh = 0;
#else
    const uint32_t zh = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(zj - X.k < 8u) != 0));
#endif
    __syncthreads();
    {
      const uint32_t tn = (uint32_t)((t0 + kFastTile < n ? t0 + kFastTile : t0) * 4u);
      FSK_BLOAD4(pre0, 0u, tn); FSK_BLOAD4(pre1, 1u, tn); FSK_BLOAD4(pre2, 2u, tn); FSK_BLOAD4(pre3, 3u, tn);
    }
#pragma unroll 1
    for (uint32_t c = 0; c < 4; c++) {
      const v4f x4 = stage[c * kSlotStride + lane];
      const uint32_t ph0 = X.phase, ph1 = (X.phase + 1 == P.d) ? 0u : X.phase + 1;
      const uint32_t r0 = poly[ph0 * 64u + lane];
      const uint32_t r1 = poly[ph1 * 64u + lane];
      float zc[4], zs[4];
      if (UNI) {
        const v4f z01 = ztile[c * 2u], z23 = ztile[c * 2u + 1u];
        zc[0] = z01.x; zs[0] = z01.y; zc[1] = z01.z; zs[1] = z01.w;
        zc[2] = z23.x; zs[2] = z23.y; zc[3] = z23.z; zs[3] = z23.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          zc[j] = zr; zs[j] = zi;
          const float nr = __builtin_fmaf(-zi, wim, zr * wre), ni = __builtin_fmaf(zi, wre, zr * wim);
          zr = nr; zi = ni;
        }
      }
      const float xin[4] = {x4.x, x4.y, x4.z, x4.w};
      float xs[4];
      const uint32_t pb = X.k;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float y0, y1, oi0, oq0, oi1, oq1;
        if (__builtin_expect(zh != 0u, 0)) { asm volatile("s_nop 0"); front_zero(F, zj == pb + (uint32_t)h); }
        front_sample(F, FK, xin[2 * h], zc[2 * h], zs[2 * h], xs[2 * h], y0, oi0, oq0);
        front_sample(F, FK, xin[2 * h + 1], zc[2 * h + 1], zs[2 * h + 1], xs[2 * h + 1], y1, oi1, oq1);
        X.k++;
        X.kv += 1u;
        X.phase = h ? ph1 : ph0;
        const float ypr[2] = {y0, y1};
        back_pair<UNI>(B, BK, P, S, M, &poly[X.phase * 64u + lane], lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, X, oi0 + oi1, oq0 + oq1,
                       ypr, h ? r1 : r0, inc);
        amp_advance(X.amp_soff, amp_quad_bytes, amp_wrap);
      }
      X.phase = (ph1 + 1 == P.d) ? 0u : ph1 + 1;
      if (WB) {
        if (C.valid)
          *reinterpret_cast<v4f *>(samples + (size_t)(C.row4 >> 2) * pitch + t0 + 4u * c) = (v4f){xs[0], xs[1], xs[2], xs[3]};
      }
    }
  }
  // the last prefetch is still in flight and its registers are dead to the compiler: keep them until it has landed
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre0), "+v"(pre1), "+v"(pre2), "+v"(pre3) : : "memory");
#undef FSK_BLOAD4
  for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
  pipe_store<UNI>(F, true, B, P, C, stream, out_counts, n, X.k, X.k % P.cadence, X.phase, amp_pos_of(X.amp_soff, amp_quad_bytes), inc, free0);
}

// ================================================================================================================
// The same arithmetic one sample at a time, for what is not whole 16-sample tiles of a lock-step batch: the odd sample
// that completes a decimator pair left open by the previous call, the samples up to the next 16-byte boundary, the
// tail of a call, buffers without 16-byte alignment.  One wave per 64-stream group, strided loads -- slow, but a
// stream cut into calls of ANY lengths then computes bit for bit what one call computes (the reference is a streaming
// state machine: fsk-demodulation.node.test.ts:363-398, 668-753), with the generic kernel left to the fp64 path, wide
// or fractional rings, traces and batches that are not in lock step.
// parity0: downsample.counter at the first sample (fsk.ts:106); the open pair's first low-pass outputs are acc_i / acc_q.
// LDS: poly [d][64] u32 | zmail [64] u32
// ================================================================================================================
template <bool WB, bool UNI>
__global__ __launch_bounds__(64, 2) void demod_tail_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n, size_t pitch, int parity0, int append,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts) {
  extern __shared__ float4 lds[];
  uint32_t *poly = reinterpret_cast<uint32_t *>(lds);
  uint32_t *zmail = poly + 64u * P.d;
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)blockIdx.x * P.d * 64u;
  const uint32_t lane = threadIdx.x;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const PipeCtx C = pipe_ctx(P, S, stream);
  const FastMem &M = C.M;
  const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
  const uint32_t fld = C.fld, row4 = C.row4;
  const uint64_t inc = UNI ? (((uint64_t)P.u_inc_hi << 32) | P.u_inc_lo) : S.nco_inc[C.row4 >> 2];
  const uint64_t free0 = pipe_free0<UNI>(C);

  FrontLane F;
  FrontK FK;
  front_load<UNI>(F, FK, P, S, C);
  BackLane B;
  BackK BK;
  back_load<UNI>(B, BK, P, S, C, stream, out_counts, eod_counts, append);
  for (uint32_t p = 0; p < P.d; p++) poly[p * 64u + lane] = gpoly[p * 64u + lane];
  BackU X;
  X.own_pairs = kHandPairs; X.hand_lag = kHandLag;   // (no hand-over in this kernel: HAND = false)
  X.k = 0; X.kv = 0; X.zmail = zmail; X.cmail = nullptr;
  zmail[lane] = zmail_init(B.dph);
  X.free0 = free0 - (parity0 ? inc : 0ull);             // decimated sample 0 of this launch starts one sample early then
  X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) ? kDirectPairs : 0u;
  X.zlive = __builtin_amdgcn_ballot_w64((B.dph < kHandPairs) | (B.qai != 0.f) | (B.qaq != 0.f) | (B.qbi != 0.f) | (B.qbq != 0.f)) ? 1u : 0u;
  asm volatile("" : "+v"(X.kv));
  X.phase = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(poly_phase));
  const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(amp_pos));
  const uint32_t amp_quad_bytes = P.n_streams * 16u;
  X.amp_soff = amp_soff_of(amp_pos0, amp_quad_bytes);
  const uint32_t amp_wrap = (P.amp_cap >> 2) * amp_quad_bytes;
  const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);
  float acc_i = parity0 ? PIPE_RLOAD(acc_i) : 0.f, acc_q = parity0 ? PIPE_RLOAD(acc_q) : 0.f;
  float *xrow = samples + (size_t)(C.row4 >> 2) * pitch;

  uint32_t par = (uint32_t)parity0;
  for (size_t t = 0; t < n; t++) {
    const float xin = C.valid ? xrow[t] : 0.f;
    float zc, zs, xs, y, oi, oq;
    nco_phasor(free0 + inc * (uint64_t)t, zc, zs);
    if (par == 0) front_zero(F, zmail[lane] == X.k);     // first sample of decimated sample number X.k
    front_sample(F, FK, xin, zc, zs, xs, y, oi, oq);
    if (S.trace_stream != 0xFFFFFFFFu && C.M.voff == S.trace_stream * 4u) trace_pre_put(S, (double)y / (0.5 * P.lp_b0 * 1152921504606846976.0));   // (back to the reference's scale HERE -- this kernel carries the value times the low-pass gain b0 / 2 and 2^60 -- so that the buffer holds one scale whichever kernels a traced engine's calls went through: ADVICE r05)
    if (WB) { if (C.valid) xrow[t] = xs; }
    if (par == 0) {
      acc_i = oi; acc_q = oq;
      par = 1;
    } else {
      par = 0;
      X.k++;
      X.kv += 1u;
      const uint32_t r_old = poly[X.phase * 64u + lane];
      const float ypr[2] = {F.by2, F.by1};               // the pair's two pre-filter outputs
      back_pair<UNI, false, true>(B, BK, P, S, M, &poly[X.phase * 64u + lane], lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, X, acc_i + oi,
                                  acc_q + oq, ypr, r_old, inc);
      amp_advance(X.amp_soff, amp_quad_bytes, amp_wrap);
      X.phase = (X.phase + 1 == P.d) ? 0u : X.phase + 1;
      acc_i = 0.f; acc_q = 0.f;
    }
  }
  for (uint32_t p = 0; p < P.d; p++) gpoly[p * 64u + lane] = poly[p * 64u + lane];
  pipe_store<UNI>(F, true, B, P, C, stream, out_counts, n, X.k, X.k % P.cadence, X.phase, amp_pos_of(X.amp_soff, amp_quad_bytes), inc, free0);
  PIPE_RSTORE(acc_i, acc_i); PIPE_RSTORE(acc_q, acc_q);
  PIPE_ISTORE(ds_cnt, par);
}

// ---- host side ---------------------------------------------------------------------------------------------------
size_t demod_pipe_lds_bytes(const DemodParams &P) {
  return sizeof(float4) * (4 * kSlotStride + kPipeSlots * kSlotV4 + 2 * 64 + 2 * 8) + sizeof(uint32_t) * (64u * P.d + 4u + 64u);
}
size_t demod_fused_lds_bytes(const DemodParams &P) { return sizeof(float4) * (4 * kSlotStride + 2 * 8) + sizeof(uint32_t) * 64u * (P.d + 1u); }

hipError_t set_pipe_lds_limit(size_t pipe_bytes) {
  hipError_t e = hipSuccess;
#define FSK_ATTR(WBV, UNIV)                                                                                      \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_pipe_kernel<WBV, UNIV>),                       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe_bytes);
  FSK_ATTR(false, false) FSK_ATTR(false, true) FSK_ATTR(true, false) FSK_ATTR(true, true)
#undef FSK_ATTR
  return e;
}

#ifdef FSK_ABLATE
static void set_ablate() {
  const char *a = getenv("FSK_ABLATE");
  const int v = a ? atoi(a) : 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &v, sizeof(v));
}
#else
static inline void set_ablate() {}
#endif

hipError_t launch_demod_pipe(bool writeback, bool append, const DemodParams &P, const DemodState &S, float *samples, size_t n,
                             size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                             uint32_t *eod_counts, hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds = demod_pipe_lds_bytes(P);
  set_ablate();
#define FSK_LAUNCH_PIPE(WBV, UNIV)                                                                          \
  hipLaunchKernelGGL((demod_pipe_kernel<WBV, UNIV>), dim3(blocks), dim3(128), lds, stream, P, S, samples, n, pitch, \
                     append ? 1 : 0, out, out_pitch, out_counts, eod_counts)
  const bool uni = P.uni_cfg != 0;
  if (writeback) { if (uni) FSK_LAUNCH_PIPE(true, true); else FSK_LAUNCH_PIPE(true, false); }
  else { if (uni) FSK_LAUNCH_PIPE(false, true); else FSK_LAUNCH_PIPE(false, false); }
#undef FSK_LAUNCH_PIPE
  return hipGetLastError();
}

hipError_t launch_demod_fused(bool writeback, bool append, const DemodParams &P, const DemodState &S, float *samples, size_t n,
                              size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                              uint32_t *eod_counts, hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds = demod_fused_lds_bytes(P);
#define FSK_LAUNCH_FUSED(WBV, UNIV)                                                                          \
  hipLaunchKernelGGL((demod_fused_kernel<WBV, UNIV>), dim3(blocks), dim3(64), lds, stream, P, S, samples, n, pitch, \
                     append ? 1 : 0, out, out_pitch, out_counts, eod_counts)
  const bool uni = P.uni_cfg != 0;
  if (writeback) { if (uni) FSK_LAUNCH_FUSED(true, true); else FSK_LAUNCH_FUSED(true, false); }
  else { if (uni) FSK_LAUNCH_FUSED(false, true); else FSK_LAUNCH_FUSED(false, false); }
#undef FSK_LAUNCH_FUSED
  return hipGetLastError();
}

hipError_t launch_demod_tail(bool writeback, bool append, int parity0, const DemodParams &P, const DemodState &S,
                             float *samples, size_t n, size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                             uint32_t *eod_counts, hipStream_t stream) {
  const uint32_t blocks = (P.n_streams + 63u) / 64u;
  const size_t lds = sizeof(uint32_t) * 64u * (P.d + 1u);
#define FSK_LAUNCH_TAIL(WBV, UNIV)                                                                          \
  hipLaunchKernelGGL((demod_tail_kernel<WBV, UNIV>), dim3(blocks), dim3(64), lds, stream, P, S, samples, n, pitch, \
                     parity0, append ? 1 : 0, out, out_pitch, out_counts, eod_counts)
  const bool uni = P.uni_cfg != 0;
  if (writeback) { if (uni) FSK_LAUNCH_TAIL(true, true); else FSK_LAUNCH_TAIL(true, false); }
  else { if (uni) FSK_LAUNCH_TAIL(false, true); else FSK_LAUNCH_TAIL(false, false); }
#undef FSK_LAUNCH_TAIL
  return hipGetLastError();
}

}  // namespace fsk

#ifdef FSK_STAMP
extern "C" int fskdbg_read_stamps(unsigned long long *out, size_t count) {   // diagnostic builds only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fsk::g_stamp), count * sizeof(unsigned long long));
}
#endif
