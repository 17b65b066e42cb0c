// fsk_blk.hip -- round-3 whole-tile fp32 demodulator for gfx950 (MI355X): FOUR waves per 64-stream group with a
// BLOCK-BATCHED back wave.
//
// Why.  Round 2 priced the kernels by their vector instruction count.  Wave stamps (s_memtime around each wave's main loop
// and its hand-off waits, profiles/r03_wave_stamps.txt) show what that misses: the back wave of demod_pipe_kernel is busy
// 267-288 cycles per input sample even ALONE on its SIMD, for ~28 vector instructions per sample.  The probe rows of
// profiles/r03_valu_probe_ctl.txt say why: besides its 4-cycle issue slot, every branch (taken or not), every LDS or
// global memory instruction and every scalar compare-and-branch costs the issuing wave another 25-40 cycles -- and the
// per-sample back executed three branches, an LDS write, a global store and two LDS reads per decimated sample.  Other
// waves fill those cycles, so the SIMD does not care; the group does, because its pipeline runs at the pace of its
// slowest wave.  (Removing the back wave's arithmetic piece by piece confirmed it: profiles/r03_back_cuts.txt.)
//
// What.  The chain is cut where nothing but a reset feeds back, into four waves of about equal length:
//   0  tile loads, AGC, pre-filter, the tile's NCO phasors           -> y ring   (what resetState() never touches)
//   1  mixer + free-running I/Q low-pass + pair sums U               -> x ring   (zeroes a lane's filters kZeroLagPairs after a reset)
//   2  ZIR correction (once it is this wave's) + discriminator       -> x ring, IN PLACE (phase, magnitude) over U
//   3  post filter, slicer, sync correlator, frame state machine
// every one of them a whole tile (sixteen input = eight decimated samples) per step, its LDS inputs read up front, one
// branch per tile besides the loop's; the back wave (3) restates fsk.ts:278-375 per block:
//   * per decimated sample only what is a recurrence: discriminator tail + post filter + slicer (disc_post, shared with
//     the per-sample path), the sync correlator's running count, the silence run's last loud sample;
//   * once per block: the bit clock.  A lane decides at most one bit per block (decisions are dsSPB >= 8 decimated
//     samples apart), at sample jd = nextBitSampleIndex - k0 of the block, so the vote, the byte shift register and the
//     start / stop-bit classification are evaluated once from the block's eight slicer bits (popcounts of an 8-bit word);
//   * ONE exit test per block for everything rare: 'eod' (bounded from above by the silence run at the block's end), a
//     sync candidate (matched >= threshold at any of the eight samples), a bad start or stop bit.  A block that trips it
//     -- and every block while a lane is inside this wave's own span after a reset, or while the amplitude ring is off
//     its quad grid -- is redone sample by sample by back_pair (the round-2 code, unchanged arithmetic) from the block's
//     entry state; the block path commits nothing before the test.  3.5 % of the blocks of BASELINE config #3's signal.
//   * the polyphase sync registers live lane-major in LDS (four consecutive phases = one ds_read_b128 / ds_write_b128),
//     rotated so that every block starts at a multiple of four;
//   * the amplitudes of a block are stored as two 16-byte quads at its end (fsk_dev.h: the ring's layout); completed
//     bytes go to a four-byte queue per lane that is flushed every sixteen blocks (at most two bytes per lane can
//     complete in between: a byte takes 10 x dsSPB >= 80 decimated samples).
// Who owns the ZIR correction.  After a reset the back wave runs the direct instance (kDirectPairs samples) and then keeps
// the correction for kOwnLag4 = 24 more samples (fsk_params.h); its values at the hand-over sample follow from the
// recurrence alone, so it posts them (cmail) to wave 2 -- which may be up to 23 decimated samples ahead -- when the direct
// instance ends.  From the hand-over on wave 2 subtracts the correction before the discriminator -- un-retired until zr_dph
// has reached kHandPairs (the span every kernel keeps it un-retired for; wave 2 counts from the hand-over sample), then
// retiring it by the common rule -- and the back wave's block path needs no discriminator of its own.  While a lane is inside the back wave's
// span, wave 2 leaves its pair sums in the x ring instead of (phase, magnitude) (cmail[5] = where the span began).
// Bytes, counters and state are those of the per-sample kernels by construction: same float instruction sequence per
// decimated sample, integer logic restated exactly (tests/test_gpu_parity.py runs every golden through this kernel).
//
// Which wave plays which part follows the SIMD it landed on, rotated by the workgroups the CU has started, so that every
// SIMD hosts one wave of each part (see the kernel).
// Batches beyond one round of resident workgroups run persistently over (group, time slice) items: BlkSched below.
// LDS: stage [4][65] v4f (the final-state hand-over fin [3][64] over it) | yring [y_slots][2][64] v4f, y_slots = 6 .. 28 by
//      what the batch leaves (demod_blk_plan) | xring [6][2][64] v4f | zt [4, 8 or 16][8] v4f | poly [64][PS] u32 |
//      counters [8] | zmail [64] u32 | cmail [6][64] u32   (not a byte more: 256 B more per workgroup cost config #3 2 %,
//      profiles/r04_block_resets.txt section 11)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fsk_params.h"
#include "fsk_dev.h"
#include "fsk_pipe_dev.h"
#include "fsk_blk_dev.h"
#include "fsk_blk_sched.h"

namespace fsk {

// ---- The block path that takes resets (round 4) --------------------------------------------------------------------
// A block that holds an 'eod' (resetState(), fsk.ts:175-188, 288-291) or a lane inside this wave's own span after one
// used to be redone by back_pair, sample by sample: ~276 instructions and ~30 branches per decimated sample from a wave
// that, on an idle receiver bank (every stream fires an 'eod' each samplesForEOD decimated samples on its own schedule:
// 3.7 resets per tile and group), is the only busy one of its group.  This is the same arithmetic as ONE straight-line
// sequence per decimated sample for all lanes, every condition a mask:
//   * the correction's recurrence, the zero-started ("direct") instance on the tile's pre-filter outputs and the NCO
//     phasors wave 0 left in LDS, the formation of the correction's start values at zr_dph 24 / 25 (zir_step, op for op),
//     and the lanes' own discriminator where their span is this wave's (zr_dph < kHandPairs);
//   * disc_post, slicer, correlator, silence run as in blk_fast; 'eod' is tested exactly, per sample, and resetState()
//     applied under its mask: post filter, direct instance, zr_dph, lastPhase = the free-running frame's phase at that
//     sample (thf8: eight values per tile, evaluated by one lane each), silence run, search threshold;
//   * what a reset writes to memory and to the other waves' mailboxes is NOT done here: the sample of the reset (E.jr) and
//     of the start values' formation (E.jc, with the values) are returned, and the caller performs them once per tile,
//     before it releases the tile's ring slots (the other waves cannot have passed the points the mailboxes name: the
//     lag constants' static_asserts);
//   * the bit clock once per block (blk_clock), for lanes without a reset in it.
// Everything else -- a sync candidate, a bad start / stop bit, a second reset or a reset plus a bit decision of one lane in
// one block -- sets the returned word's sign bit: the caller puts the entry state back and redoes the block sample by sample.
// The entry state of a block that blk_medium is about to work on IN PLACE (a copy in registers would not fit beside it:
// the kernel's 128-VGPR budget), as seven 16-byte stores per lane to a buffer of the engine's: fire and forget -- read back
// (med_unstash) only if the block turns out to need the per-sample path.  Everything blk_medium writes.
__device__ inline void med_stash(const BackLane &B, __amdgpu_buffer_rsrc_t rs, uint32_t avoff, uint32_t fld16) {
#define FSK_ST4(i, a, b, c, d) \
  __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, c), __builtin_bit_cast(uint32_t, d)}, rs, avoff, (i) * fld16, 0)
  FSK_ST4(0u, B.qai, B.qaq, B.qbi, B.qbq);
  FSK_ST4(1u, B.px1, B.px2, B.py, B.pv);
  FSK_ST4(2u, B.last_phase, B.thf, B.dph, B.matched);
  FSK_ST4(3u, B.dix1, B.dix2, B.diy, B.dvi);
  FSK_ST4(4u, B.dqx1, B.dqx2, B.dqy, B.dqv);
  FSK_ST4(5u, B.q0i, B.q0q, B.thr_eff, B.ls);
  FSK_ST4(6u, B.acc, B.T, B.tlast, B.sreg);
#undef FSK_ST4
}
__device__ inline void med_unstash(BackLane &B, __amdgpu_buffer_rsrc_t rs, uint32_t avoff, uint32_t fld16) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const v4u r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, 0u, kCohSc1);
  const v4u r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, fld16, kCohSc1);
  const v4u r2 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, 2u * fld16, kCohSc1);
  const v4u r3 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, 3u * fld16, kCohSc1);
  const v4u r4 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, 4u * fld16, kCohSc1);
  const v4u r5 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, 5u * fld16, kCohSc1);
  const v4u r6 = __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, 6u * fld16, kCohSc1);
  // lanes beyond the batch have no row (their stores were dropped, their loads return zeros): they keep what they have --
  // parked lanes whose filters run on zeros, wherever those stand
  const bool mine = avoff < 0xFFFFFFF0u;
  auto f = [mine](uint32_t x, float old) { return mine ? __builtin_bit_cast(float, x) : old; };
  auto u = [mine](uint32_t x, uint32_t old) { return mine ? x : old; };
  B.qai = f(r0.x, B.qai); B.qaq = f(r0.y, B.qaq); B.qbi = f(r0.z, B.qbi); B.qbq = f(r0.w, B.qbq);
  B.px1 = f(r1.x, B.px1); B.px2 = f(r1.y, B.px2); B.py = f(r1.z, B.py); B.pv = f(r1.w, B.pv);
  B.last_phase = f(r2.x, B.last_phase); B.thf = f(r2.y, B.thf); B.dph = u(r2.z, B.dph); B.matched = u(r2.w, B.matched);
  B.dix1 = f(r3.x, B.dix1); B.dix2 = f(r3.y, B.dix2); B.diy = f(r3.z, B.diy); B.dvi = f(r3.w, B.dvi);
  B.dqx1 = f(r4.x, B.dqx1); B.dqx2 = f(r4.y, B.dqx2); B.dqy = f(r4.z, B.dqy); B.dqv = f(r4.w, B.dqv);
  B.q0i = f(r5.x, B.q0i); B.q0q = f(r5.y, B.q0q); B.thr_eff = u(r5.z, B.thr_eff); B.ls = u(r5.w, B.ls);
  B.acc = u(r6.x, B.acc); B.T = u(r6.y, B.T); B.tlast = u(r6.z, B.tlast); B.sreg = u(r6.w, B.sreg);
}

// (works on Bn IN PLACE: the caller has parked the entry state with med_stash)
// UNI = false (round 5: per-stream tone pairs, BASELINE config #4's kind): every lane has its own NCO, so the direct instance's
// phasors are evaluated here per lane (nco_phasor of the lane's free-running accumulator: what zir_step does without a phasor
// tile) and so is lastPhase after a reset (back_reset's expression); k0 = decimated samples of the launch before the tile.
template <bool UNI>
__device__ inline uint32_t blk_medium(BackLane &Bn, const BackK &K, const BlkK &Q, uint32_t matched_min, uint32_t kv0,
                                      const v4f *slot, const v4f *slot2, const v4f *ys0, const v4f *ys1, uint32_t lane,
                                      const v4f *ztile, const float (&thf8)[kBlk], const uint32_t *prow, uint32_t pidx, uint32_t pidx2,
                                      float (&am)[kBlk], uint32_t &bq, uint32_t &nq, MedEv &E, uint32_t &w_out,
                                      uint64_t free0 = 0, uint64_t inc = 0, uint32_t k0 = 0) {
  uint32_t w = 0, hard = 0;
  uint32_t matched = Bn.matched, thr_cur = Bn.thr_eff, ls = Bn.ls;
  E.jr = 0; E.jc = 0; E.cai = E.caq = E.cbi = E.cbq = 0.f;
  // two decimated samples' inputs at a time: ring entries (pair sums where the lane's span is this wave's, else phase and
  // magnitude), the four pre-filter outputs behind them, the two polyphase registers (not kept: the caller re-forms the new
  // ones from the old ones and the block's slicer bits), the phasors.  Read one pair AHEAD of the arithmetic (the wave runs
  // alone: an LDS round trip it waits for is ~120 cycles of nothing); the fences keep the compiler from reading all eight
  // samples' inputs up front -- 40 registers it does not have.
  v4f pe_n = slot[lane], ye_n = ys0[lane], zz0_n = {}, zz1_n = {};
  if (UNI) { zz0_n = ztile[0]; zz1_n = ztile[1]; }
  uint2 rp_n = *reinterpret_cast<const uint2 *>(prow + pidx);
#pragma unroll
  for (int jj = 0; jj < kBlk / 2; jj++) {
    const v4f pe = pe_n, ye2 = ye_n, zz0 = zz0_n, zz1 = zz1_n;
    const uint2 rp2 = rp_n;
    asm volatile("" ::: "memory");
    if (jj + 1 < kBlk / 2) {
      const int jn = jj + 1;
      pe_n = (jn < 2 ? slot : slot2)[(jn & 1) * 64 + (int)lane];
      ye_n = (jn < 2 ? ys0 : ys1)[(jn & 1) * 64 + (int)lane];
      rp_n = *reinterpret_cast<const uint2 *>(prow + (jn < 2 ? pidx : pidx2) + 2 * (jn & 1));
      if (UNI) { zz0_n = ztile[2 * jn]; zz1_n = ztile[2 * jn + 1]; }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; h++) {
    const int j = 2 * jj + h;
    const float e0 = h ? pe.z : pe.x, e1 = h ? pe.w : pe.y;
    const float y0 = h ? ye2.z : ye2.x, y1 = h ? ye2.w : ye2.y;
    const uint32_t kvj = kv0 + (uint32_t)(j + 1);
    v4f z = h ? zz1 : zz0;                                                   // (c0, s0, c1, s1) of the pair's two input samples
    float thf_j = thf8[j];
    if (!UNI) {
      const uint64_t n0 = (uint64_t)(2u * (k0 + (uint32_t)j));
      float c0, s0, c1, s1;
      nco_phasor(free0 + inc * n0, c0, s0);
      nco_phasor(free0 + inc * (n0 + 1ull), c1, s1);
      z = (v4f){c0, s0, c1, s1};
      const uint64_t fr0 = free0 + inc * (uint64_t)(2u * (k0 + (uint32_t)j + 1u));
      double r = (double)fr0 * 5.42101086242752217e-20 * 6.283185307179586476925;   // 2^-64 turns -> radians (back_reset)
      r = r > 3.14159265358979323846 ? r - 6.283185307179586476925 : r;
      thf_j = (float)r;
    }
    med_sample(Bn, K, j, e0, e1, y0, y1, z, thf_j, h ? rp2.y : rp2.x, kvj, matched_min, kOwnPairs4, matched, thr_cur, ls, w, hard, E, am[j]);
    }
  }
  hard |= med_finish(Bn, K, Q, kv0, matched, thr_cur, ls, w, bq, nq, E);
  w_out = w;
  return hard;
}

// Time-sliced priority.  With equal priorities a SIMD issues from its OLDEST wave first: the four groups sharing a CU then
// finish one after the other (185 .. 284 cycles per sample, profiles/r03_blk4_stamps.txt), the last one largely alone
// and at a single group's latency-bound pace.  Rotating four priority levels over the CU's workgroups every 64 half
// tiles gives each group every level a quarter of the time.  (s_setprio takes an immediate.)
#ifndef FSK_BLK_WGS
#define FSK_BLK_WGS 4
#endif
#ifndef FSK_BLK5_WPE
#define FSK_BLK5_WPE 5           // waves per SIMD the five-wave kernel is compiled for (96 VGPRs)
#endif
#ifndef FSK_BLK_LDS_PAD
#define FSK_BLK_LDS_PAD 0      // (measurement builds: unused bytes at the end of the workgroup's LDS -- where the fourth workgroup of a CU stops fitting)
#endif
#ifndef FSK_BLK_PRIO
#define FSK_BLK_PRIO 1
#endif
// the back wave's lean block (blk_fast<true>, fsk_blk_dev.h) while every stream of its group is inside a frame
#ifndef FSK_BLK_LEAN
#define FSK_BLK_LEAN 0
#endif
// s_sleep argument (x 64 cycles) of each wave's hand-off poll
#ifndef FSK_BLK_SLEEP_A
#define FSK_BLK_SLEEP_A 1
#endif
#ifndef FSK_BLK_SLEEP_B
#define FSK_BLK_SLEEP_B 1
#endif
#ifndef FSK_BLK_SLEEP_C
#define FSK_BLK_SLEEP_C 1
#endif
#ifndef FSK_BLK_SLEEP_D
#define FSK_BLK_SLEEP_D 1
#endif
#ifndef FSK_BLK_PRIO_PERIOD
#define FSK_BLK_PRIO_PERIOD 64
#endif
// demod_blk_kernel_r / _rp: fixed priority levels by part (1, see BYROLE below) or the plain kernel's rotation (0: measurement builds)
#ifndef FSK_BLK_R_BYROLE
#define FSK_BLK_R_BYROLE 1
#endif
// BYROLE (demod_blk_kernel_r): fixed levels by part instead, the back wave highest -- in the calls that kernel is picked for
// the back wave IS the group's time and the other three mostly wait; measured on the idle bank 291 -> 312 Gsamples/s
// (and the rotation, which evens out four equally loaded groups of a CU, has nothing to even out there).
template <bool BYROLE = false>
__device__ inline void blk_prio(uint32_t hidx, uint32_t wgj, uint32_t role = 0) {
  if (BYROLE) {
    if (hidx == 0u) {
      switch (role) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
    return;
  }
#if FSK_BLK_PRIO == 1
  if ((hidx & (FSK_BLK_PRIO_PERIOD - 1u)) == 0u) {
    switch (((hidx / FSK_BLK_PRIO_PERIOD) + wgj) & 3u) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
  }
#elif FSK_BLK_PRIO == 2      // (measurement builds) by part: the back wave first
  if (hidx == 0u) {
    switch (role) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
  }
#elif FSK_BLK_PRIO == 3      // (measurement builds) the back wave always highest, the others rotate over three levels
  if (role == 3u) { if (hidx == 0u) __builtin_amdgcn_s_setprio(3); }
  else if ((hidx & (FSK_BLK_PRIO_PERIOD - 1u)) == 0u) {
    switch (((hidx / FSK_BLK_PRIO_PERIOD) + wgj) % 3u) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(2); break;
    }
  }
#endif
}

// Time slices.  One workgroup per group fills the chip only in whole rounds of (workgroups resident at once) = 4 per CU:
// at 81 920 streams the second round runs a quarter full and the launch takes as long as 131 072 streams do.  Above one
// round the launch is therefore PERSISTENT: as many workgroups as fit, each taking (group, time slice) items from one
// queue in device memory -- slice 0 of every group first, in group order, then slices in the order their predecessors
// finished (a group's slices are strictly sequential: its state travels through the state arrays exactly as it does
// between two launches; its chain of slices is as long as the group's whole time, so every group must start early).  A
// workgroup that completes slice s of a group pushes slice s + 1 behind the queue's tail; a popper whose entry has not
// been pushed yet waits for it (the pusher is running and never waits, so this cannot lock up).
// Coherence.  A group's next slice may run on another XCD, behind another L2.  Everything a slice hands on -- state
// arrays, polyphase registers, amplitude ring, output counts -- is therefore written and read with device-scope cache
// policy (sc1: kCohSc1, fsk_dev.h; stores write through, loads take no cached copy) and the push waits for the stores'
// completion (vmcnt); no cache-wide write-back or invalidate is needed.  (An agent-scope release / acquire fence pair per
// slice was measured first: buffer_wbl2 with the amplitude ring's dirty lines in L2 cost ~70 us per slice; queues
// private to an XCD avoid it too but balance worse: profiles/r03_slices.txt.)
struct BlkSched {
  uint32_t *q;                   // [0] head, [1] tail, [16 + j] pushed entries: 1 << 31 | slice << 20 | group
  uint32_t groups;               // 64-stream groups
  uint32_t nslices;              // per group
  uint32_t slice_tiles;          // tiles per slice (the last one may be shorter)
  uint32_t total;                // groups * nslices
  uint32_t y_slots;              // half tiles in the y ring (>= kBlkSlots): how far wave 0 may run ahead of the back wave
  uint32_t zt_tiles;             // tiles of NCO phasors in flight (wave 0 -> wave 1), a power of two > y_slots / 2
  uint32_t lanes;                // streams per workgroup: 64, or 32 / 16 / 8 for batches that would otherwise leave CUs idle
                                 // (round 4, "narrow groups": see launch_demod_blk)
  uint32_t medium;               // 1: blocks with an 'eod' or a lane inside the back wave's own span take blk_medium; 0: the
                                 // per-sample path, as in round 3; 2 (tests): blk_medium, then the entry state back and the per-sample path
};

// MED: the back wave's block path takes resets (blk_medium).  A second kernel rather than a switch inside one: the
// straight-line path needs nearly the whole 128-register budget, and merely compiled in beside the other two it costs
// them ~4 % where resets are rare (constants pushed out of registers, spills around it: profiles/r04_block_resets.txt).
// The host launches demod_blk_kernel_r when the previous call's share of tiles off the fast path says it pays.
// NW = 5 (round 6, demod_blk5_kernel): the front wave's two halves on a wave each -- part 4: tile loads, transposition, AGC, IN PLACE into
// one of two staging tiles (and the AGC write-back); part 0: pre-filter + the tile's NCO phasors from there -> y ring.  Five
// instruction streams per SIMD at four workgroups per CU (every part's per-tile loop fits 96 VGPRs as it stands); the AGC's
// ten-instruction dependent chain -- 32 of the front wave's 44.8 class-priced cycles per sample -- no longer shares a wave with
// the loads and the pre-filter.  Same instruction sequence per sample (front_agc + front_bp are front_agc_bp's two halves).
template <bool WB, bool UNI, bool SL, bool MED, int NW = 4>
__device__ __forceinline__ void demod_blk_body(
    const DemodParams &P, const DemodState &S, float *__restrict__ samples, size_t n_call, size_t pitch, int append_call,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts, const BlkSched &Z) {
  FSK_ABL_INIT
  FSK_STAMP_DECL
  FSK_WAIT_DECL
#ifdef FSK_SCRATCH_PAD       // (measurement builds: private memory nobody touches -- does a larger scratch frame cost resident waves?)
  volatile uint32_t scratch_pad[FSK_SCRATCH_PAD / 4];
  if (n_call == 0xDEADBEEFull) { for (int i = 0; i < FSK_SCRATCH_PAD / 4; i++) scratch_pad[i] = (uint32_t)i; out_counts[0] = scratch_pad[threadIdx.x & 15]; }
#endif
  constexpr int COH = SL ? kCohSc1 : 0;           // cache policy of what a time slice hands to the next (fsk_dev.h)
  extern __shared__ float4 lds[];
  const uint32_t PS = blk_poly_stride(P.d);
  const uint32_t NY = Z.y_slots;
  v4f *stage = reinterpret_cast<v4f *>(lds);
  v4f *fin = stage;                                       // [0..1] wave 1's final I/Q low-pass state, [2] wave 2's final correction
                                                          // (over the staging tile: wave 0 is done with it when they are written)
  v4f *yring = stage + (NW == 5 ? 8 : 4) * kSlotStride;   // wave 0 -> wave 1 (and the back wave after a reset): pre-filter outputs
                                                          // (NW = 5: two staging tiles, part 4 -> part 0)
  v4f *ring = yring + NY * 2 * 64;                        // wave 1: pair sums U -> wave 2: (phase, magnitude) IN PLACE -> wave 3
  v4f *zt = ring + kBlkSlots * kBlkSlotV4;
  const uint32_t ZTM = Z.zt_tiles - 1u;
  uint32_t *poly = reinterpret_cast<uint32_t *>(zt + Z.zt_tiles * 8);   // [lane][PS], index 0 = the phase of the launch's first push
  uint32_t *ctr = poly + 64u * PS;                        // produced by wave 0, 1, 2 | consumed by wave 3 | [4] CU arrival | [5] item
                                                          // NW = 5: [6] tiles part 4 has staged, [7] staging tiles part 0 has released
  uint32_t *zmail = ctr + 8;                              // back -> wave 1: where to zero a lane's I/Q low-pass
  uint32_t *cmail = zmail + 64;                           // back -> wave 2: [0] from which decimated sample, [1..4] the correction kHandLag
                                                          // steps of its recurrence before that (or there, if [0] <= kHandLag), [5] since
                                                          // which sample the back wave wants a lane's pair sums kept
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t lane = threadIdx.x & 63u;

  if (threadIdx.x == 0) {
    // Which wave plays which part.  A workgroup's four waves sit on the CU's four SIMDs (one each, in cyclic order from a
    // varying start); four workgroups share a CU at BASELINE config #3's size.  If the part followed the wave index, some
    // SIMDs would host three or four back waves and their groups would set the kernel's time (measured: 198 .. 295
    // cycles per sample over the groups of one launch).  So the part follows the SIMD, rotated by how many workgroups
    // this CU has started: every SIMD then hosts one wave of each part.
    const uint32_t hw = __builtin_amdgcn_s_getreg(0xF804);        // HW_REG_HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
    const uint32_t xcc = __builtin_amdgcn_s_getreg(0x1814) & 7u;  // HW_REG_XCC_ID
    const uint32_t key = (xcc << 8) | ((hw >> 8) & 0xFFu);
    ctr[4] = __hip_atomic_fetch_add(&S.cu_ctr[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const uint32_t simd = (__builtin_amdgcn_s_getreg(0xF804) >> 4) & 3u;
  // The part a wave WANTS follows its SIMD; the part it GETS is settled among the four waves through LDS, so that every
  // part is played exactly once whatever the placement (were two waves of a workgroup ever placed on one
  // SIMD -- co-resident kernels, CU masking, another firmware -- a part would be missing and the others would wait for
  // its counter forever).  Every wave posts its wish, then all four evaluate the same rule on the same four words: in wave
  // order, a wave takes its wish if no earlier wave has it, else the lowest part still free.  One LDS write and one
  // barrier per kernel; with the usual one-wave-per-SIMD placement every wave gets its wish.
  if (lane == 0) ctr[8 + wave] = NW == 5 ? simd : (simd + ctr[4]) & 3u;      // (ctr[8..12]: the first words of zmail, initialised after this)
  __syncthreads();
  uint32_t role = 0;
  if (NW == 5) {
    // Five parts on four SIMDs: a workgroup's five waves sit on the SIMDs in cyclic order, so one SIMD -- qd -- hosts two of
    // them, and with four workgroups per CU (config #3's size) every SIMD is some workgroup's qd.  Rows and columns 0..3 of the
    // cyclic 5 x 5 Latin square: the first wave on SIMD q plays part (qd + q) mod 5, the second wave on qd plays (qd + 4) mod 5 --
    // every part once per workgroup, and where the four workgroups of a CU have different qd (the usual placement) every SIMD
    // hosts one wave of each part.  Any other placement: the part is the wave index (every part still played exactly once).
    uint32_t sd[5], packed = 0;
    for (uint32_t w = 0; w < 5u; w++) {
      sd[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[8 + w]) & 3u;
      packed += 1u << (8u * sd[w]);
    }
    uint32_t qd = 4u;
    for (uint32_t q = 0; q < 4u; q++)
      if (packed == 0x01010101u + (1u << (8u * q))) qd = q;
    role = wave;
    if (qd < 4u) {
      bool first = true;
      for (uint32_t w = 0; w < wave; w++) first = first && sd[w] != simd;
      role = first ? (qd + simd) % 5u : (qd + 4u) % 5u;
    }
  } else {
    uint32_t taken = 0;
    for (uint32_t w = 0; w <= wave; w++) {
      uint32_t want = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[8 + w]) & 3u;
      if (taken & (1u << want)) want = (uint32_t)__builtin_ctz(~taken & 15u);
      taken |= 1u << want;
      role = want;
    }
  }
  const uint32_t wgj = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ctr[4] & 3u));
  __syncthreads();                                          // (zmail is written below)

  uint32_t push_e = 0;             // the entry to put behind the queue's tail: the slice after the one just completed
#pragma unroll 1
  for (;;) {
  // ---- this workgroup's next item: group grp, tiles [t_begin, t_begin + n_tiles) of the call
  uint32_t grp = blockIdx.x, t_begin = 0, slice = 0;
  if (SL) {
    // (push and pop in ONE divergent region inside the iteration: split across the loop's back edge, the compiler ran
    // lane 0's part after the other lanes' barrier)
    if (wave == 0 && lane == 0) {
      if (push_e != 0u) {
        const uint32_t j = __hip_atomic_fetch_add(&Z.q[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&Z.q[16u + j], push_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const uint32_t i = __hip_atomic_fetch_add(&Z.q[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t e = 0xFFFFFFFFu;
      if (i < Z.groups) e = i;
      else if (i < Z.total) {
        const uint32_t *slot = &Z.q[16u + (i - Z.groups)];
        // (bounded like the hand-off waits, fsk_pipe_dev.h: the slice before this one belongs to a workgroup that took its item
        // earlier and is running; if any workgroup of the launch has given up -- the fault word -- so does this one)
        uint32_t polls = 0;
        while ((e = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
          __builtin_amdgcn_s_sleep(8);
          const bool dead = S.blk_stat && __hip_atomic_load(&S.blk_stat[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
          if (dead || ++polls > (uint32_t)(FSK_SPIN_CAP)) {
            if (S.blk_stat) __hip_atomic_fetch_or(&S.blk_stat[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            e = 0xFFFFFFFFu;
            break;
          }
        }
        if (e != 0xFFFFFFFFu) e &= 0x7FFFFFFFu;
      }
      ctr[5] = e;
    }
    __syncthreads();
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[5]);
    if (e == 0xFFFFFFFFu) return;
    grp = e & 0xFFFFFu; slice = e >> 20;
    t_begin = slice * Z.slice_tiles;
  }
  const size_t n_tiles = SL ? (size_t)((uint32_t)(n_call / kFastTile) - t_begin < Z.slice_tiles ? (uint32_t)(n_call / kFastTile) - t_begin : Z.slice_tiles)
                            : n_call / kFastTile;
  const size_t n = n_tiles * kFastTile;
  const int append = (SL && slice != 0u) ? 1 : append_call;
  // Narrow groups: the workgroup owns streams [grp * W, grp * W + W); lanes W .. 63 are handled exactly as the lanes
  // beyond the batch's last stream always were (parked on zeros, every store dropped: pipe_ctx, back_load).  The state
  // arrays, the amplitude ring and the outputs are indexed by stream; only the polyphase registers are blocked by 64.
  const uint32_t W = Z.lanes, s0 = grp * W;
  const bool mine = lane < W;
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)(s0 >> 6) * P.d * 64u + (s0 & 63u);
  const uint32_t stream = mine ? s0 + lane : 0xFFFFFFFFu;
  const PipeCtx C = pipe_ctx(P, S, stream);
  const uint32_t nh = 2u * (uint32_t)n_tiles;               // half tiles = blocks
  const uint64_t inc = UNI ? (((uint64_t)P.u_inc_hi << 32) | P.u_inc_lo) : S.nco_inc[C.row4 >> 2];
  const uint64_t free0 = pipe_free0<UNI, COH>(C);

  if (threadIdx.x == 0) { ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; if (NW == 5) { ctr[6] = 0; ctr[7] = 0; } }
  if (wave == 1) {
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    zmail[lane] = zmail_init(PIPE_ILOAD(zr_dph));
  }
  if (wave == 3) {
    // a correction still on the back wave's part of its un-retired span when the launch starts: the back wave keeps it until
    // zr_dph reaches kOwnPairs4, the discriminator wave takes it from decimated sample kOwnPairs4 - zr_dph on (fsk_params.h)
    const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    const uint32_t dph = PIPE_ILOAD(zr_dph);
    uint32_t kq = 0xFFFFFFFFu;
    float ai = 0.f, aq = 0.f, bi = 0.f, bq = 0.f;
    if (dph >= kDirectPairs && dph < kOwnPairs4) {            // (the steps that are left run here, once per launch: kq <= kOwnLag4
                                                              // tells the taker that the values are final -- see zir_step)
      ai = PIPE_RLOAD(zq_ai); aq = PIPE_RLOAD(zq_aq); bi = PIPE_RLOAD(zq_bi); bq = PIPE_RLOAD(zq_bq);
      const float c1 = P.z_c1, c2 = P.z_c2;
      for (uint32_t g = dph; g < kOwnPairs4; g++) {
        const float ni = __builtin_fmaf(c1, bi, -(c2 * ai)), nq = __builtin_fmaf(c1, bq, -(c2 * aq));
        ai = bi; aq = bq; bi = ni; bq = nq;
      }
      kq = kOwnPairs4 - dph;
    }
    cmail[64u + lane] = __builtin_bit_cast(uint32_t, ai); cmail[128u + lane] = __builtin_bit_cast(uint32_t, aq);
    cmail[192u + lane] = __builtin_bit_cast(uint32_t, bi); cmail[256u + lane] = __builtin_bit_cast(uint32_t, bq);
    cmail[lane] = kq;
    cmail[320u + lane] = 0u - dph;                          // (the decimated sample of this launch at which the stream was last reset, as a signed number)
  }
  __syncthreads();

  if (NW == 5 && role == 4) {
    // ------------------------------------------------------------------------------ NW = 5, part 4: loads, AGC (in place in a staging tile)
    FrontLane F;
    FrontK K;
    front_load<UNI, COH>(F, K, P, S, C);
    const uint32_t sub_row = lane >> 2, chunk = lane & 3;
    const uint32_t rows_here = P.n_streams - s0 < W ? P.n_streams - s0 : W;
    v4i in_rsrc;
    {
      const uint64_t base = reinterpret_cast<uint64_t>(samples + (size_t)s0 * pitch);
      in_rsrc.x = (int)(uint32_t)base;
      in_rsrc.y = (int)(uint32_t)(base >> 32);
      in_rsrc.z = (int)(uint32_t)(rows_here * pitch * 4u);
      in_rsrc.w = 0x00020000;
    }
    const uint32_t in_voff = (uint32_t)((sub_row * pitch + 4u * chunk) * 4u);
    const uint32_t in_row16 = (uint32_t)(16u * pitch * 4u);
    const uint32_t st_slot = chunk * kSlotStride + sub_row;
#define BLK_BLOAD4(dst, rows16, soff)                                                                       \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(in_voff + (rows16) * in_row16), \
               "s"(in_rsrc), "s"(soff) : "memory")
    auto load_tile = [&](size_t t, v4f &a, v4f &b, v4f &c, v4f &d) {
      const uint32_t tn = (uint32_t)((t_begin + (t < n_tiles ? t : n_tiles - 1)) * kFastTile * 4u);
      BLK_BLOAD4(a, 0u, tn); BLK_BLOAD4(b, 1u, tn); BLK_BLOAD4(c, 2u, tn); BLK_BLOAD4(d, 3u, tn);
    };
    v4f a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the state loads above are complete, the count starts clean
    load_tile(0, a0, a1, a2, a3);
    load_tile(1, b0, b1, b2, b3);
    load_tile(2, c0, c1, c2, c3);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
    uint32_t released = 0;                                  // staging tiles part 0 is done with
    auto do_tile = [&](uint32_t t, v4f &r0, v4f &r1, v4f &r2, v4f &r3) {
      if (WB) asm volatile("s_waitcnt vmcnt(16)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      if (t >= released + 2u) {                             // both staging tiles in use: part 0 has not released tile t - 2
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (t >= released + 2u) {
          released = lds_peek(&ctr[7]);
          if (t >= released + 2u) FSK_SPIN(FSK_BLK_SLEEP_A, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      v4f *stg = stage + (t & 1u) * 4u * kSlotStride;
      stg[st_slot] = r0; stg[st_slot + 16] = r1; stg[st_slot + 32] = r2; stg[st_slot + 48] = r3;
      load_tile((size_t)t + 3, r0, r1, r2, r3);
      v4u32 cw;
      lds_peek4_begin(ctr + 4, cw);
      v4f x4[4];
#pragma unroll
      for (uint32_t c = 0; c < 4; c++) x4[c] = stg[c * kSlotStride + lane];
#pragma unroll
      for (uint32_t c = 0; c < 4; c++) {
        const float xin[4] = {x4[c].x, x4[c].y, x4[c].z, x4[c].w};
        float xs[4];
#pragma unroll
        for (int j = 0; j < 4; j++) xs[j] = FSK_ABL(4) ? xin[j] : front_agc(F, K, xin[j]);
        stg[c * kSlotStride + lane] = (v4f){xs[0], xs[1], xs[2], xs[3]};
        if (WB) {
          if (C.valid)
            *reinterpret_cast<v4f *>(samples + (size_t)(C.row4 >> 2) * pitch + (size_t)(t_begin + t) * kFastTile + 4u * c) = (v4f){xs[0], xs[1], xs[2], xs[3]};
        }
      }
      lds_post(&ctr[6], t + 1u);
      released = lds_peek4_get(cw, 3);
    };
    const uint32_t nt = (uint32_t)n_tiles;
    FSK_STAMP_BEGIN
    for (uint32_t t0 = 0; t0 < nt; t0 += 96u) {             // (a multiple of three tiles and of 64 half tiles)
      blk_prio<(MED && FSK_BLK_R_BYROLE)>(2u * t0, wgj, 0u);
      const uint32_t te = t0 + 96u < nt ? t0 + 96u : nt;
      for (uint32_t t = t0; t < te; t += 3) {
        do_tile(t, a0, a1, a2, a3);
        if (t + 1 < te) do_tile(t + 1, b0, b1, b2, b3);
        if (t + 2 < te) do_tile(t + 2, c0, c1, c2, c3);
      }
    }
    FSK_STAMP_END(4)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
#undef BLK_BLOAD4
    {
      const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
      const FastMem &M = C.M;
      const uint32_t fld = C.fld;
      PIPE_RSTORE(agc_gain, F.g);
    }
  } else if (NW == 5 && role == 0) {
    // ------------------------------------------------------------------------------ NW = 5, part 0: pre-filter, phasors
    FrontLane F;
    FrontK K;
    front_load<UNI, COH>(F, K, P, S, C);
    uint32_t consumed = 0, slot_i = 0, staged = 0;
    uint64_t zacc = free0 + inc * (uint64_t)(lane & 15u);
    const uint64_t inc16 = inc * 16u;
    const uint32_t nt = (uint32_t)n_tiles;
    FSK_STAMP_BEGIN
    for (uint32_t t = 0; t < nt; t++) {
      if ((t & 31u) == 0u) blk_prio<(MED && FSK_BLK_R_BYROLE)>(2u * t, wgj, 0u);
      const uint32_t hidx = 2u * t;
      v4u32 cv, cw;
      lds_peek4_begin(ctr, cv);                             // (read now, looked at after the tile: see lds_peek4_begin)
      lds_peek4_begin(ctr + 4, cw);
      if (staged <= t || hidx + 1u - consumed >= NY) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (staged <= t) {                               // part 4's tile
          staged = lds_peek(&ctr[6]);
          if (staged <= t) FSK_SPIN(FSK_BLK_SLEEP_A, S.blk_stat);
        }
        while (hidx + 1u - consumed >= NY) {                // ring full: the back wave (which may still need the slots'
          consumed = lds_peek(&ctr[3]);                     // pre-filter outputs after a reset) has not released them
          if (hidx + 1u - consumed >= NY) FSK_SPIN(FSK_BLK_SLEEP_A, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      const v4f *stg = stage + (t & 1u) * 4u * kSlotStride;
      v4f x4[4];
#pragma unroll
      for (uint32_t c = 0; c < 4; c++) x4[c] = stg[c * kSlotStride + lane];
      lds_post(&ctr[7], t + 1u);                            // (waits for the four reads: the staging tile is part 4's again)
      if (UNI) {
        float pc, ps;
        nco_phasor(zacc, pc, ps);
        reinterpret_cast<f2 *>(zt + (t & ZTM) * 8u)[lane & 15u] = (f2){pc, ps};
        zacc += inc16;
      }
#pragma unroll
      for (uint32_t hf = 0; hf < 2; hf++) {
        v4f *slot = yring + slot_i * 2u * 64u;
        slot_i = slot_i + 1u == NY ? 0u : slot_i + 1u;
#pragma unroll
        for (uint32_t cc = 0; cc < 2; cc++) {
          const uint32_t c = 2u * hf + cc;
          const float xv[4] = {x4[c].x, x4[c].y, x4[c].z, x4[c].w};
          float y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) y[j] = FSK_ABL(0) ? xv[j] : front_bp(F, K, xv[j]);
          slot[cc * 64u + lane] = (v4f){y[0], y[1], y[2], y[3]};
        }
      }
      lds_post(&ctr[0], hidx + 2u);
      consumed = lds_peek4_get(cv, 3);
      staged = lds_peek4_get(cw, 2);
    }
    FSK_STAMP_END(0)
    {
      const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
      const FastMem &M = C.M;
      const uint32_t fld = C.fld;
      PIPE_RSTORE(bp_x1, F.bx1); PIPE_RSTORE(bp_x2, F.bx2); PIPE_RSTORE(bp_y1, F.by1); PIPE_RSTORE(bp_y2, F.by2);
    }
  } else if (role == 0) {
    // ------------------------------------------------------------------------------ loads, AGC, pre-filter
    FrontLane F;
    FrontK K;
    front_load<UNI, COH>(F, K, P, S, C);
    const uint32_t sub_row = lane >> 2, chunk = lane & 3;
    const uint32_t rows_here = P.n_streams - s0 < W ? P.n_streams - s0 : W;   // (rows beyond: the range check returns zeros)
    v4i in_rsrc;
    {
      const uint64_t base = reinterpret_cast<uint64_t>(samples + (size_t)s0 * pitch);
      in_rsrc.x = (int)(uint32_t)base;
      in_rsrc.y = (int)(uint32_t)(base >> 32);
      in_rsrc.z = (int)(uint32_t)(rows_here * pitch * 4u);
      in_rsrc.w = 0x00020000;
    }
    const uint32_t in_voff = (uint32_t)((sub_row * pitch + 4u * chunk) * 4u);
    const uint32_t in_row16 = (uint32_t)(16u * pitch * 4u);
    const uint32_t st_slot = chunk * kSlotStride + sub_row;
    // tile prefetch exactly as in demod_pipe_kernel (three register sets, hand-counted waits)
#define BLK_BLOAD4(dst, rows16, soff)                                                                       \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(in_voff + (rows16) * in_row16), \
               "s"(in_rsrc), "s"(soff) : "memory")
    auto load_tile = [&](size_t t, v4f &a, v4f &b, v4f &c, v4f &d) {
      const uint32_t tn = (uint32_t)((t_begin + (t < n_tiles ? t : n_tiles - 1)) * kFastTile * 4u);
      BLK_BLOAD4(a, 0u, tn); BLK_BLOAD4(b, 1u, tn); BLK_BLOAD4(c, 2u, tn); BLK_BLOAD4(d, 3u, tn);
    };
    v4f a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the state loads above are complete, the count starts clean
    load_tile(0, a0, a1, a2, a3);
    load_tile(1, b0, b1, b2, b3);
    load_tile(2, c0, c1, c2, c3);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
    uint32_t consumed = 0, slot_i = 0;
    // NCO phasors of the free-running frame (uniform configuration): lane j & 15 evaluates sample j of the tile from the
    // exact accumulator (v_cos / v_sin take turns), the sixteen (cos, sin) pairs go to the tile's zt slot and wave 1 reads
    // them back as broadcasts.  (They are this wave's work because it has the slack: profiles/r03_blk4_stamps.txt.)
    uint64_t zacc = free0 + inc * (uint64_t)(lane & 15u);
    const uint64_t inc16 = inc * 16u;
    auto do_tile = [&](uint32_t t, v4f &r0, v4f &r1, v4f &r2, v4f &r3) {
      if (WB) asm volatile("s_waitcnt vmcnt(16)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      stage[st_slot] = r0; stage[st_slot + 16] = r1; stage[st_slot + 32] = r2; stage[st_slot + 48] = r3;
      load_tile((size_t)t + 3, r0, r1, r2, r3);
      const uint32_t hidx = 2u * t;
      v4u32 cv;
      lds_peek4_begin(ctr, cv);                             // (read now, looked at after the tile: see lds_peek4_begin)
      if (hidx + 1u - consumed >= NY) {                     // both of the tile's slots must be free
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (hidx + 1u - consumed >= NY) {                // ring full: the back wave (which may still need the slots'
          consumed = lds_peek(&ctr[3]);                     // pre-filter outputs after a reset) has not released them
          if (hidx + 1u - consumed >= NY) FSK_SPIN(FSK_BLK_SLEEP_A, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      if (UNI) {
        float pc, ps;
        nco_phasor(zacc, pc, ps);
        reinterpret_cast<f2 *>(zt + (t & ZTM) * 8u)[lane & 15u] = (f2){pc, ps};
        zacc += inc16;
      }
#pragma unroll
      for (uint32_t hf = 0; hf < 2; hf++) {
        v4f *slot = yring + slot_i * 2u * 64u;
        slot_i = slot_i + 1u == NY ? 0u : slot_i + 1u;
#pragma unroll
        for (uint32_t cc = 0; cc < 2; cc++) {
          const uint32_t c = 2u * hf + cc;
          const v4f x4 = stage[c * kSlotStride + lane];
          const float xin[4] = {x4.x, x4.y, x4.z, x4.w};
          float xs[4], y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { if (FSK_ABL(0)) xs[j] = y[j] = xin[j]; else front_agc_bp(F, K, xin[j], xs[j], y[j]); }
          slot[cc * 64u + lane] = (v4f){y[0], y[1], y[2], y[3]};
          if (WB) {
            if (C.valid)
              *reinterpret_cast<v4f *>(samples + (size_t)(C.row4 >> 2) * pitch + (size_t)(t_begin + t) * kFastTile + 4u * c) = (v4f){xs[0], xs[1], xs[2], xs[3]};
          }
        }
      }
      lds_post(&ctr[0], hidx + 2u);                         // (per tile: the other waves work in whole tiles too)
      consumed = lds_peek4_get(cv, 3);
    };
    const uint32_t nt = (uint32_t)n_tiles;
    FSK_STAMP_BEGIN
    for (uint32_t t0 = 0; t0 < nt; t0 += 96u) {             // (a multiple of three tiles and of 64 half tiles)
      blk_prio<(MED && FSK_BLK_R_BYROLE)>(2u * t0, wgj, 0u);
      const uint32_t te = t0 + 96u < nt ? t0 + 96u : nt;
      for (uint32_t t = t0; t < te; t += 3) {
        do_tile(t, a0, a1, a2, a3);
        if (t + 1 < te) do_tile(t + 1, b0, b1, b2, b3);
        if (t + 2 < te) do_tile(t + 2, c0, c1, c2, c3);
      }
    }
    FSK_STAMP_END(0)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
#undef BLK_BLOAD4
    {
      const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
      const FastMem &M = C.M;
      const uint32_t fld = C.fld;
      PIPE_RSTORE(agc_gain, F.g);
      PIPE_RSTORE(bp_x1, F.bx1); PIPE_RSTORE(bp_x2, F.bx2); PIPE_RSTORE(bp_y1, F.by1); PIPE_RSTORE(bp_y2, F.by2);
    }
  } else if (role == 1) {
    // ------------------------------------------------------------------------------ mixer, I/Q low-pass, pair sums
    FrontLane F;
    FrontK K;
    front_load<UNI, COH>(F, K, P, S, C);
    float wre = 1.f, wim = 0.f;
    if (!UNI) {
      const __amdgpu_buffer_rsrc_t cf_rsrc = C.cf_rsrc;
      const uint32_t fld = C.fld, row4 = C.row4;
      wre = (float)PIPE_CLOAD(CF_w1_re); wim = (float)PIPE_CLOAD(CF_w1_im);
    }
    uint64_t tacc = free0;
    const uint64_t inc16 = inc * 16u;
    uint32_t consumed = 0, produced = 0, slot_i = 0, yslot_i = 0;
    float zr = 1.f, zi = 0.f;
    FSK_STAMP_BEGIN
    uint32_t hidx = 0;                                        // half tiles done; this wave works a tile (two of them) at a time
    while (hidx < nh) {
      if ((hidx & 63u) == 0u) blk_prio<(MED && FSK_BLK_R_BYROLE)>(hidx, wgj, 1u);
      if (produced < hidx + 2u || hidx + 2u - consumed > kBlkSlots) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced < hidx + 2u) {                        // wave 0's tile
          produced = lds_peek(&ctr[0]);
          if (produced < hidx + 2u) FSK_SPIN(FSK_BLK_SLEEP_B, S.blk_stat);
        }
        while (hidx + 2u - consumed > kBlkSlots) {           // ring full: wait for the back wave
          consumed = lds_peek(&ctr[3]);
          if (hidx + 2u - consumed > kBlkSlots) FSK_SPIN(FSK_BLK_SLEEP_B, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      // as far as the rings allow, up to the next priority point: the back edge is the loop's only branch besides the
      // (rare) zeroing of a lane's filters
      uint32_t lim0 = (hidx | 63u) + 1u;
      lim0 = lim0 < nh ? lim0 : nh;
      uint32_t lim = lim0 < produced ? lim0 : produced;
      lim = lim < consumed + kBlkSlots ? lim : consumed + kBlkSlots;
      do {
        v4u32 cv;
        lds_peek4_begin(ctr, cv);
        const v4f *ztile = zt + ((hidx >> 1) & ZTM) * 8u;
        if (!UNI) {                                           // per-stream tones: the tile's first phasor from the exact accumulator
          nco_phasor(tacc, zr, zi);
          tacc += inc16;
        }
        const uint32_t zj = zmail[lane];
        // the tile's inputs first (twelve LDS reads in flight together), then straight-line arithmetic: ONE branch per tile
        // (a lane's filters to be zeroed inside it: rare), none per quad
        v4f y4[4], zz[8];
        {
          const v4f *ys0 = yring + yslot_i * 2u * 64u;
          yslot_i = yslot_i + 1u == NY ? 0u : yslot_i + 1u;
          const v4f *ys1 = yring + yslot_i * 2u * 64u;
          yslot_i = yslot_i + 1u == NY ? 0u : yslot_i + 1u;
          y4[0] = ys0[lane]; y4[1] = ys0[64u + lane]; y4[2] = ys1[lane]; y4[3] = ys1[64u + lane];
          if (UNI) {
#pragma unroll
            for (int i = 0; i < 8; i++) zz[i] = ztile[i];
          }
        }
        v4f *slot = ring + slot_i * kBlkSlotV4;
        slot_i = slot_i + 1u == kBlkSlots ? 0u : slot_i + 1u;
        v4f *slot2 = ring + slot_i * kBlkSlotV4;
        slot_i = slot_i + 1u == kBlkSlots ? 0u : slot_i + 1u;
        v4f u4[4];
        auto quad = [&](const uint32_t c, const bool zeroing) {
          float zc[4], zs[4];
          if (UNI) {
            zc[0] = zz[2 * c].x; zs[0] = zz[2 * c].y; zc[1] = zz[2 * c].z; zs[1] = zz[2 * c].w;
            zc[2] = zz[2 * c + 1].x; zs[2] = zz[2 * c + 1].y; zc[3] = zz[2 * c + 1].z; zs[3] = zz[2 * c + 1].w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              zc[j] = zr; zs[j] = zi;
              const float nr = __builtin_fmaf(-zi, wim, zr * wre), ni = __builtin_fmaf(zi, wre, zr * wim);
              zr = nr; zi = ni;
            }
          }
          const float y[4] = {y4[c].x, y4[c].y, y4[c].z, y4[c].w};
          const uint32_t pb = 4u * hidx + 2u * c;
          float oi[4], oq[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (FSK_ABL(1)) { oi[j] = oq[j] = y[j] + zc[j]; continue; }
            if (zeroing && !(j & 1)) front_zero(F, zj == pb + (uint32_t)(j >> 1));
            front_mix_lp(F, K, y[j], zc[j], zs[j], oi[j], oq[j]);
          }
          u4[c] = (v4f){oi[0] + oi[1], oq[0] + oq[1], oi[2] + oi[3], oq[2] + oq[3]};   // U (I, Q) x 2
        };
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(zj - 4u * hidx < 8u) != 0ull, 0)) {
          asm volatile("s_nop 0");
          quad(0, true); quad(1, true); quad(2, true); quad(3, true);
        } else {
          quad(0, false); quad(1, false); quad(2, false); quad(3, false);
        }
        slot[lane] = u4[0]; slot[64u + lane] = u4[1]; slot2[lane] = u4[2]; slot2[64u + lane] = u4[3];
        hidx += 2u;
        lds_post(&ctr[1], hidx);                              // for the discriminator wave
        produced = lds_peek4_get(cv, 0); consumed = lds_peek4_get(cv, 3);
        lim = lim0 < produced ? lim0 : produced;
        lim = lim < consumed + kBlkSlots ? lim : consumed + kBlkSlots;
      } while (hidx + 2u <= lim);
    }
    FSK_STAMP_END(1)
    fin[lane] = (v4f){F.ix1, F.ix2, F.iy, F.iv};
    fin[64u + lane] = (v4f){F.qx1, F.qx2, F.qy, F.qv};
    lds_post(&ctr[1], nh + 1u);
  } else if (role == 2) {
    // ------------------------------------------------------------------------------ ZIR correction + discriminator
    const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    QLane Qz = {0.f, 0.f, 0.f, 0.f};
    const uint32_t dph_s = PIPE_ILOAD(zr_dph);
    if (dph_s >= kOwnPairs4) {                                // this wave's from the first sample on
      Qz.ai = PIPE_RLOAD(zq_ai); Qz.aq = PIPE_RLOAD(zq_aq); Qz.bi = PIPE_RLOAD(zq_bi); Qz.bq = PIPE_RLOAD(zq_bq);
    }
    // the decimated sample of this launch (signed) at which the correction became this wave's: zr_dph was kOwnPairs4 there, and
    // the correction retires no earlier than kHandLag - kOwnLag4 samples later, where zr_dph reaches kHandPairs (fsk_params.h)
    int32_t kq_last = (int32_t)kOwnPairs4 - (int32_t)dph_s;
    float c1 = P.z_c1, c2 = P.z_c2, tiny = 0x1p-123f, rel = 3.7252902984619141e-09f;
    uint32_t sgn = 0x80000000u;
    asm volatile("" : "+v"(c1), "+v"(c2), "+v"(tiny), "+v"(rel), "+v"(sgn));
    uint64_t qlive = __builtin_amdgcn_ballot_w64((Qz.ai != 0.f) | (Qz.aq != 0.f) | (Qz.bi != 0.f) | (Qz.bq != 0.f));   // (raw masks: no bool round trips)
    uint32_t produced = 0, slot_i = 0;
    FSK_STAMP_BEGIN
    uint32_t hidx = 0;                                        // half tiles done; a tile (two of them, eight decimated samples) at a time
    while (hidx < nh) {
      if ((hidx & 63u) == 0u) blk_prio<(MED && FSK_BLK_R_BYROLE)>(hidx, wgj, 2u);
      if (produced < hidx + 2u) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced < hidx + 2u) {
          produced = lds_peek(&ctr[1]);
          if (produced < hidx + 2u) FSK_SPIN(FSK_BLK_SLEEP_C, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      uint32_t lim0 = (hidx | 63u) + 1u;
      lim0 = lim0 < nh ? lim0 : nh;
      uint32_t lim = lim0 < produced ? lim0 : produced;
      do {
        v4u32 cv;
        lds_peek4_begin(ctr, cv);
        v4f *slot = ring + slot_i * kBlkSlotV4;
        slot_i = slot_i + 1u == kBlkSlots ? 0u : slot_i + 1u;
        v4f *slot2 = ring + slot_i * kBlkSlotV4;
        slot_i = slot_i + 1u == kBlkSlots ? 0u : slot_i + 1u;
        // (the two mailbox words first, fenced: LDS answers in order, and the one test below that needs them then waits for
        // them alone while the tile's ring entries are still on their way -- the compiler, left alone, put them last)
        const uint32_t kq = cmail[lane];
        const uint32_t ow = 4u * hidx - cmail[320u + lane];     // decimated samples since the back wave's own span began
        asm volatile("" ::: "memory");
        const v4f ua = slot[lane], ub = slot[64u + lane], uc = slot2[lane], ud = slot2[64u + lane];
        const float ui[8] = {ua.x, ua.z, ub.x, ub.z, uc.x, uc.z, ud.x, ud.z}, uq[8] = {ua.y, ua.w, ub.y, ub.w, uc.y, uc.w, ud.y, ud.w};
        float ph[8], am[8];
        // one test for everything that is not the plain discriminator: a hand-over due in this tile, a lane inside the
        // back wave's own span, a live correction
        if (__builtin_expect((__builtin_amdgcn_ballot_w64((kq - 4u * hidx < 8u) | (ow < kOwnPairs4)) | qlive) != 0ull, 0)) {
          // a hand-over due in this tile: the posted start values, advanced by the posted number of steps of the
          // recurrence (all of the tile's lanes at once)
          QLane H = {0.f, 0.f, 0.f, 0.f};
          if (kq - 4u * hidx < 8u) {
            H.ai = __builtin_bit_cast(float, cmail[64u + lane]); H.aq = __builtin_bit_cast(float, cmail[128u + lane]);
            H.bi = __builtin_bit_cast(float, cmail[192u + lane]); H.bq = __builtin_bit_cast(float, cmail[256u + lane]);
            const uint32_t steps = kq > kOwnLag4 ? kOwnLag4 : 0u;   // (posted inside this launch: kOwnLag4 steps before its sample)
            for (uint32_t g = 0; g < steps; g++) {
              const float ni = __builtin_fmaf(c1, H.bi, -(c2 * H.ai)), nq = __builtin_fmaf(c1, H.bq, -(c2 * H.aq));
              H.ai = H.bi; H.aq = H.bq; H.bi = ni; H.bq = nq;
            }
          }
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (kq == 4u * hidx + (uint32_t)j) {              // the back wave's correction becomes this wave's here
              Qz = H;
              kq_last = (int32_t)kq;
            }
            const float wi = ui[j] - Qz.ai, wq = uq[j] - Qz.aq;
            {
              const float ni = __builtin_fmaf(c1, Qz.bi, -(c2 * Qz.ai)), nq = __builtin_fmaf(c1, Qz.bq, -(c2 * Qz.aq));
              Qz.ai = Qz.bi; Qz.aq = Qz.bq; Qz.bi = ni; Qz.bq = nq;
            }
            ph[j] = atan2_amp_fma(wq, wi, am[j], tiny, sgn);
            const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(Qz.ai), __builtin_fabsf(Qz.aq)),
                                              __builtin_fmaxf(__builtin_fabsf(Qz.bi), __builtin_fabsf(Qz.bq)));
            const bool steady = (int32_t)(4u * hidx + (uint32_t)j) - kq_last >= (int32_t)(kHandLag - kOwnLag4);   // zr_dph >= kHandPairs (zir_step's `steady`)
            if (steady & !(big > am[j] * rel)) { Qz.ai = 0.f; Qz.aq = 0.f; Qz.bi = 0.f; Qz.bq = 0.f; }
            // lanes inside the back wave's own span (direct instance, then its own correction): it evaluates their
            // discriminator itself and needs the pair sums for that, so they stay
            const bool own = ow + (uint32_t)j < kOwnPairs4;
            ph[j] = own ? ui[j] : ph[j]; am[j] = own ? uq[j] : am[j];
          }
          qlive = __builtin_amdgcn_ballot_w64((Qz.ai != 0.f) | (Qz.aq != 0.f) | (Qz.bi != 0.f) | (Qz.bq != 0.f));
        } else if (FSK_ABL(2)) {
#pragma unroll
          for (int j = 0; j < 8; j++) { ph[j] = ui[j]; am[j] = uq[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) ph[j] = atan2_amp_fma(uq[j], ui[j], am[j], tiny, sgn);
        }
        slot[lane] = (v4f){ph[0], am[0], ph[1], am[1]};       // in place: wave 1 will not touch the slots before the back wave frees them
        slot[64u + lane] = (v4f){ph[2], am[2], ph[3], am[3]};
        slot2[lane] = (v4f){ph[4], am[4], ph[5], am[5]};
        slot2[64u + lane] = (v4f){ph[6], am[6], ph[7], am[7]};
        hidx += 2u;
        lds_post(&ctr[2], hidx);
        produced = lds_peek4_get(cv, 1);
        lim = lim0 < produced ? lim0 : produced;
      } while (hidx + 2u <= lim);
    }
    FSK_STAMP_END(2)
    fin[128u + lane] = (v4f){Qz.ai, Qz.aq, Qz.bi, Qz.bq};
    lds_post(&ctr[2], nh + 1u);
  } else {
    // ---------------------------------------------------------------------------------------------- block back
    BackLane B;
    BackK Kp;
    back_load<UNI, COH>(B, Kp, P, S, C, stream, out_counts, eod_counts, append);
    BackK Ks;                                                 // the constants as scalars, for the paths that are not the fast block loop
    back_consts(Ks, P);
    if (B.dph >= kOwnPairs4) { B.qai = 0.f; B.qaq = 0.f; B.qbi = 0.f; B.qbq = 0.f; B.dph = kOwnPairs4; }   // the discriminator wave's (zr_dph is re-formed at the end)
    BlkK Qs;
    Qs.stop_m1 = (1u << P.stop_pos) - 1u; Qs.sh9 = P.stop_pos - 9u; Qs.ff = 0xFFu;
    BlkK Qp = Qs;
    if (!MED) asm volatile("" : "+v"(Qp.stop_m1), "+v"(Qp.sh9), "+v"(Qp.ff));
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    const uint32_t phase0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(poly_phase));
    {
      uint32_t ph = phase0;
      for (uint32_t i = 0; i < P.d; i++) {                    // rotate: LDS index 0 = the register of the first push
        uint32_t r = 0u;
        if (mine) r = COH ? __hip_atomic_load(&gpoly[ph * 64u + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gpoly[ph * 64u + lane];
        poly[lane * PS + i] = r;
        ph = ph + 1u == P.d ? 0u : ph + 1u;
      }
    }
    BackU X;
    X.own_pairs = kOwnPairs4; X.hand_lag = kOwnLag4;
    X.k = 0; X.kv = 0; X.free0 = free0; X.zmail = zmail; X.cmail = cmail; X.phase = 0;
    X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) ? kDirectPairs : 0u;
    X.zlive = __builtin_amdgcn_ballot_w64(B.dph < kOwnPairs4) ? 1u : 0u;
    asm volatile("" : "+v"(X.kv));
    const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(amp_pos));
    const uint32_t amp_quad_bytes = P.n_streams * 16u;
    X.amp_soff = amp_soff_of(amp_pos0, amp_quad_bytes);
    const uint32_t amp_wrap = (P.amp_cap >> 2) * amp_quad_bytes;
    // the block path stores a tile's amplitudes as two whole quads: the launch must start on a quad boundary (it does
    // unless earlier calls had odd lengths; the host then launches the per-sample kernels, fsk_api.hip) -- otherwise every
    // block takes the per-sample path
    const bool amp_misaligned = (amp_pos0 & 3u) != 0u;
    const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);
    const __amdgpu_buffer_rsrc_t stash_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.blk_stash, 0, (int)(7u * amp_quad_bytes), 0x00020000);
    uint32_t produced = 0, slot_i = 0;                        // (x-ring slot of half tile t; even wherever a block starts)
    uint32_t pidx = 0;                                        // LDS index of the block's first polyphase register
    uint32_t bq = 0, nq = 0;                                  // completed bytes not yet stored (newest in the low byte)
    uint32_t *prow = poly + lane * PS;
    uint32_t yb = 0, yb_t = 0;                                // y-ring slot of half tile yb_t (kept while consecutive tiles need it)
    uint32_t rare_tiles = 0;                                  // tiles of this item that blk_medium took, or would be given
#if FSK_BLK_LEAN
    bool matched_stale = false;                               // lean blocks have run: B.matched is to be re-formed from the polyphase registers
    auto matched_now = [&]() {
      uint32_t m = 0;
      for (uint32_t i = 0; i < P.d; i += 4u) {
        const uint4 r = *reinterpret_cast<const uint4 *>(prow + i);
        m += (uint32_t)__builtin_popcount((r.x ^ Ks.qn) & Ks.mask) + (uint32_t)__builtin_popcount((r.y ^ Ks.qn) & Ks.mask) +
             (uint32_t)__builtin_popcount((r.z ^ Ks.qn) & Ks.mask) + (uint32_t)__builtin_popcount((r.w ^ Ks.qn) & Ks.mask);
      }
      B.matched = m;
      matched_stale = false;
    };
#endif
    FSK_STAMP_BEGIN
    uint32_t t = 0;                                           // half tiles consumed (a block = a tile = two of them)
    while (t < nh) {
      blk_prio<(MED && FSK_BLK_R_BYROLE)>(t & ~15u, wgj, 3u);                            // (t advances in steps of two; the outer loop sees every multiple of 16)
      if (produced < t + 2u) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced < t + 2u) {
          produced = lds_peek(&ctr[2]);
          if (produced < t + 2u) FSK_SPIN(FSK_BLK_SLEEP_D, S.blk_stat);
        }
        FSK_STAMP_W1
      }
      // every tile that is there, up to the next flush point
      uint32_t lim0 = (t | (kFlushBlocks - 1u)) + 1u;
      lim0 = lim0 < nh ? lim0 : nh;
      uint32_t lim = lim0 < (produced & ~1u) ? lim0 : (produced & ~1u);
      // ---- the block loop: straight-line code with two branches, its back edge and the exit on anything rare (a branch
      // costs a wave ~35 cycles whether taken or not)
      // (not while a lane is inside this wave's own span after a reset, nor with the amplitude ring off its quad grid: those
      // tiles go sample by sample.  Running the own-span lanes' zir_step ahead of the block path was tried: with frames
      // that end about together, 81 % of such tiles also hold another lane's 'eod' or false start and fail the block's
      // rare test after paying for it -- 8 192 streams 118.7 -> 111.3 Gsamples/s, profiles/r03_own_span_blocks.txt.)
      bool rare_exit = (X.zlive != 0u || amp_misaligned) && !FSK_ABL(3);
      uint32_t hardw = 0;                                     // (looked at after the loop: its exit stays one branch)
      // the fast loop's constants live in VGPRs for the loop only (re-made from the scalars at every entry: the registers are
      // the data's in the other two paths)
      BackK Kf = Ks;
      if (MED && !rare_exit) back_consts_pin(Kf);             // (the bit clock's three constants stay scalars in this kernel: the
                                                              // time-sliced instantiation's fast loop was one register short)
      const BackK &K = MED ? Kf : Kp;                         // (without blk_medium: pinned once, for the fast loop and the per-sample path)
      const BlkK &Q = MED ? Qs : Qp;
#if FSK_BLK_LEAN
      // every stream of the wave inside a frame (lanes without a stream do not count): the lean block, `matched` re-formed later
      if (!MED && !rare_exit && __builtin_amdgcn_ballot_w64(C.valid & (B.thr_eff != kStartedP)) == 0ull) {
        matched_stale = true;
      for (;;) {
          v4u32 cv;
          lds_peek4_begin(ctr, cv);
          const uint32_t slot_j = slot_i + 1u;                  // (kBlkSlots is even and blocks start on even slots)
          const uint32_t pidx2 = pidx + 4u >= P.d ? 0u : pidx + 4u;
          const v4f *slot = ring + slot_i * kBlkSlotV4, *slot2 = ring + slot_j * kBlkSlotV4;
          const v4f pa[4] = {slot[lane], slot[64u + lane], slot2[lane], slot2[64u + lane]};
          const uint4 rpa = *reinterpret_cast<const uint4 *>(prow + pidx), rpb = *reinterpret_cast<const uint4 *>(prow + pidx2);
          BackLane Bn = B;
          uint32_t rp[kBlk] = {rpa.x, rpa.y, rpa.z, rpa.w, rpb.x, rpb.y, rpb.z, rpb.w};
          float am[kBlk];
          uint32_t bqn = bq, nqn = nq;
          const uint32_t rare = blk_fast<true>(Bn, K, Q, X.kv, pa, rp, am, bqn, nqn, hardw);
          FSK_STAMP_COUNT(0)                                    // blocks
          if (__builtin_expect((__builtin_amdgcn_ballot_w64((int32_t)rare < 0) != 0ull) & !FSK_ABL(3), 0)) { rare_exit = true; break; }
          B = Bn; bq = bqn; nq = nqn;
          *reinterpret_cast<uint4 *>(prow + pidx) = make_uint4(rp[0], rp[1], rp[2], rp[3]);
          *reinterpret_cast<uint4 *>(prow + pidx2) = make_uint4(rp[4], rp[5], rp[6], rp[7]);
          {                                                      // syncAmplitudeBuffer.put x 8 = two quads
            uint32_t q2 = X.amp_soff + amp_quad_bytes; q2 = q2 == amp_wrap ? 0u : q2;
            __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, am[0]), __builtin_bit_cast(uint32_t, am[1]),
                                                          __builtin_bit_cast(uint32_t, am[2]), __builtin_bit_cast(uint32_t, am[3])},
                                                   amp_rsrc, M.avoff, X.amp_soff, COH);
            __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, am[4]), __builtin_bit_cast(uint32_t, am[5]),
                                                          __builtin_bit_cast(uint32_t, am[6]), __builtin_bit_cast(uint32_t, am[7])},
                                                   amp_rsrc, M.avoff, q2, COH);
            X.amp_soff = q2 + amp_quad_bytes; X.amp_soff = X.amp_soff == amp_wrap ? 0u : X.amp_soff;
          }
          X.k += (uint32_t)kBlk; X.kv += (uint32_t)kBlk;
          slot_i = slot_i + 2u == kBlkSlots ? 0u : slot_i + 2u;
          pidx = pidx2 + 4u >= P.d ? 0u : pidx2 + 4u;
          t += 2u;
          lds_post(&ctr[3], t);                                 // slots free (this wave's reads of them are complete)
          produced = lds_peek4_get(cv, 2);
          lim = lim0 < (produced & ~1u) ? lim0 : (produced & ~1u);
          if (!(t < lim)) break;
        }
      } else
#endif
      if (!rare_exit) for (;;) {
        v4u32 cv;
        lds_peek4_begin(ctr, cv);
        const uint32_t slot_j = slot_i + 1u;                  // (kBlkSlots is even and blocks start on even slots)
        const uint32_t pidx2 = pidx + 4u >= P.d ? 0u : pidx + 4u;
        const v4f *slot = ring + slot_i * kBlkSlotV4, *slot2 = ring + slot_j * kBlkSlotV4;
        const v4f pa[4] = {slot[lane], slot[64u + lane], slot2[lane], slot2[64u + lane]};
        const uint4 rpa = *reinterpret_cast<const uint4 *>(prow + pidx), rpb = *reinterpret_cast<const uint4 *>(prow + pidx2);
        BackLane Bn = B;
        uint32_t rp[kBlk] = {rpa.x, rpa.y, rpa.z, rpa.w, rpb.x, rpb.y, rpb.z, rpb.w};
        float am[kBlk];
        uint32_t bqn = bq, nqn = nq;
        const uint32_t rare = blk_fast(Bn, K, Q, X.kv, pa, rp, am, bqn, nqn, hardw);
        FSK_STAMP_COUNT(0)                                    // blocks
        if (__builtin_expect((__builtin_amdgcn_ballot_w64((int32_t)rare < 0) != 0ull) & !FSK_ABL(3), 0)) { rare_exit = true; break; }
        B = Bn; bq = bqn; nq = nqn;
        *reinterpret_cast<uint4 *>(prow + pidx) = make_uint4(rp[0], rp[1], rp[2], rp[3]);
        *reinterpret_cast<uint4 *>(prow + pidx2) = make_uint4(rp[4], rp[5], rp[6], rp[7]);
        {                                                      // syncAmplitudeBuffer.put x 8 = two quads
          uint32_t q2 = X.amp_soff + amp_quad_bytes; q2 = q2 == amp_wrap ? 0u : q2;
          __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, am[0]), __builtin_bit_cast(uint32_t, am[1]),
                                                        __builtin_bit_cast(uint32_t, am[2]), __builtin_bit_cast(uint32_t, am[3])},
                                                 amp_rsrc, M.avoff, X.amp_soff, COH);
          __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, am[4]), __builtin_bit_cast(uint32_t, am[5]),
                                                        __builtin_bit_cast(uint32_t, am[6]), __builtin_bit_cast(uint32_t, am[7])},
                                                 amp_rsrc, M.avoff, q2, COH);
          X.amp_soff = q2 + amp_quad_bytes; X.amp_soff = X.amp_soff == amp_wrap ? 0u : X.amp_soff;
        }
        X.k += (uint32_t)kBlk; X.kv += (uint32_t)kBlk;
        slot_i = slot_i + 2u == kBlkSlots ? 0u : slot_i + 2u;
        pidx = pidx2 + 4u >= P.d ? 0u : pidx2 + 4u;
        t += 2u;
        lds_post(&ctr[3], t);                                 // slots free (this wave's reads of them are complete)
        produced = lds_peek4_get(cv, 2);
        lim = lim0 < (produced & ~1u) ? lim0 : (produced & ~1u);
        if (!(t < lim)) break;
      }
      // a sync candidate, a bad start / stop bit, an amplitude ring off its quad grid: nothing but the per-sample path will do
      const bool hard_exit = MED && (amp_misaligned || __builtin_amdgcn_ballot_w64((int32_t)hardw < 0) != 0ull);
      // (a tile blk_medium would be given: counted are those inside a lane's own span -- six for every reset -- because telling an
      // 'eod' from a sync candidate at the fast loop's exit would keep blk_fast's second flag word alive in this kernel: 3.6 % at
      // 8 192 streams)
      if (!MED && rare_exit && X.zlive != 0u) rare_tiles++;
      if (MED && rare_exit && !hard_exit && Z.medium != 0u) {
        // an 'eod' in the tile at t, or a lane inside this wave's own span after one: the block path that takes resets
        // (blk_medium), in place, the tile's entry state parked in the engine's stash
        const uint32_t slot_j = slot_i + 1u;
        const uint32_t pidx2 = pidx + 4u >= P.d ? 0u : pidx + 4u;
        const v4f *slot = ring + slot_i * kBlkSlotV4, *slot2 = ring + slot_j * kBlkSlotV4;
        if (yb_t != t) { yb = t % NY; yb_t = t; }
        const uint32_t yb2 = yb + 1u == NY ? 0u : yb + 1u;
        const v4f *ys0 = yring + yb * 2u * 64u, *ys1 = yring + yb2 * 2u * 64u;
        const v4f *ztile = zt + ((t >> 1) & ZTM) * 8u;
        // lastPhase after a resetState() at the end of sample j of this tile: the free-running frame's phase there
        // (back_reset's expression; lane j evaluates it, the wave reads it back as a scalar)
        float thf8[kBlk] = {};
        if (UNI) {
          const uint64_t fr0 = X.free0 + inc * (uint64_t)(2u * (X.k + (lane & 7u) + 1u));
          double r = (double)fr0 * 5.42101086242752217e-20 * 6.283185307179586476925;
          r = r > 3.14159265358979323846 ? r - 6.283185307179586476925 : r;
          const float thfv = (float)r;
#pragma unroll
          for (int j = 0; j < kBlk; j++) thf8[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thfv), j));
        }
        med_stash(B, stash_rsrc, M.avoff, amp_quad_bytes);
        float am[kBlk];
        uint32_t bqn = bq, nqn = nq, wbits;
        MedEv E;
        const uint32_t hard = blk_medium<UNI>(B, Ks, Qs, P.matched_min, X.kv, slot, slot2, ys0, ys1, lane, ztile, thf8, prow, pidx, pidx2,
                                              am, bqn, nqn, E, wbits, X.free0, inc, X.k);
        FSK_STAMP_COUNT(2)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)hard < 0) != 0ull || Z.medium == 2u, 0)) {
          med_unstash(B, stash_rsrc, M.avoff, amp_quad_bytes);   // a sync candidate or a bad bit after all: from the entry state, sample by sample
        } else {
          bq = bqn; nq = nqn;
          {
            // syncSamplesBuffer.put x 8: old register << 1 | the sample's slicer bit (sample 1 is bit 7 of wbits)
            const uint4 oa = *reinterpret_cast<const uint4 *>(prow + pidx), ob = *reinterpret_cast<const uint4 *>(prow + pidx2);
            *reinterpret_cast<uint4 *>(prow + pidx) = make_uint4((oa.x << 1) | ((wbits >> 7) & 1u), (oa.y << 1) | ((wbits >> 6) & 1u),
                                                                 (oa.z << 1) | ((wbits >> 5) & 1u), (oa.w << 1) | ((wbits >> 4) & 1u));
            *reinterpret_cast<uint4 *>(prow + pidx2) = make_uint4((ob.x << 1) | ((wbits >> 3) & 1u), (ob.y << 1) | ((wbits >> 2) & 1u),
                                                                  (ob.z << 1) | ((wbits >> 1) & 1u), (ob.w << 1) | (wbits & 1u));
          }
          {
            uint32_t q2 = X.amp_soff + amp_quad_bytes; q2 = q2 == amp_wrap ? 0u : q2;
            __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, am[0]), __builtin_bit_cast(uint32_t, am[1]),
                                                          __builtin_bit_cast(uint32_t, am[2]), __builtin_bit_cast(uint32_t, am[3])},
                                                   amp_rsrc, M.avoff, X.amp_soff, COH);
            __builtin_amdgcn_raw_buffer_store_b128((v4u){__builtin_bit_cast(uint32_t, am[4]), __builtin_bit_cast(uint32_t, am[5]),
                                                          __builtin_bit_cast(uint32_t, am[6]), __builtin_bit_cast(uint32_t, am[7])},
                                                   amp_rsrc, M.avoff, q2, COH);
            X.amp_soff = q2 + amp_quad_bytes; X.amp_soff = X.amp_soff == amp_wrap ? 0u : X.amp_soff;
          }
          const uint32_t k0 = X.k;
          X.k += (uint32_t)kBlk; X.kv += (uint32_t)kBlk;
          slot_i = slot_i + 2u == kBlkSlots ? 0u : slot_i + 2u;
          pidx = pidx2 + 4u >= P.d ? 0u : pidx2 + 4u;
          t += 2u;
          yb = yb2 + 1u == NY ? 0u : yb2 + 1u; yb_t = t;
          // what the tile's resets owe memory and the other waves' mailboxes, once, before the tile's slots are released
          if (__builtin_amdgcn_ballot_w64((E.jr | E.jc) != 0u)) {
            if (E.jc != 0u) {                                  // (zir_step: the correction's start values, for the discriminator wave)
              cmail[64u + lane] = __builtin_bit_cast(uint32_t, E.cai); cmail[128u + lane] = __builtin_bit_cast(uint32_t, E.caq);
              cmail[192u + lane] = __builtin_bit_cast(uint32_t, E.cbi); cmail[256u + lane] = __builtin_bit_cast(uint32_t, E.cbq);
              cmail[lane] = k0 + E.jc + kOwnLag4;
            }
            if (E.jr != 0u) {                                  // (back_pair's 'eod' + back_reset)
              const uint32_t kr = k0 + E.jr;
              ist_add<COH>(M, IF_eod_total, 1u);
              if (eod_counts && M.voff < 0xFFFFFFF0u) __hip_atomic_fetch_add(&eod_counts[M.voff >> 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const uint64_t off = 0ull - (X.free0 + inc * (uint64_t)(2u * kr));
              ist_store<COH>(M, IF_fr_lo, (uint32_t)off);
              ist_store<COH>(M, IF_fr_hi, (uint32_t)(off >> 32));
              ist_store<COH>(M, IF_gsc, 0u - kr);
              zmail[lane] = kr + kZeroLagPairs;
              cmail[lane] = 0xFFFFFFFFu;
              cmail[320u + lane] = kr;
              B.rho = kr % P.cadence;
            }
          }
          X.zlive = __builtin_amdgcn_ballot_w64(B.dph < kOwnPairs4) != 0ull ? 1u : 0u;
          X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) != 0ull ? kDirectPairs : 0u;
          lds_post(&ctr[3], t);
          rare_exit = false;
          rare_tiles++;                                         // (a tile blk_medium took)
        }
      }
      if (rare_exit) {
        // something rare in the tile at t: sample by sample from its entry state (the round-2 path, unchanged)
        if (X.zlive != 0u) { FSK_STAMP_COUNT(1) } else { FSK_STAMP_COUNT(3) }
#if FSK_BLK_LEAN
        if (matched_stale) matched_now();
#endif
        blk_flush(B, bq, nq, M, out, (uint32_t)out_pitch);
        // (round 4: a half tile's inputs -- ring entries, pre-filter outputs, polyphase registers, the NCO phasors wave 0 left
        // in LDS -- are read up front and its four decimated samples run unrolled on registers: the per-sample order and
        // arithmetic are back_pair's as before, but the wave no longer waits out four LDS round trips and two loop branches
        // per decimated sample, nor re-evaluates the phasors of the direct instance with v_cos / v_sin)
        const v4f *ztile = zt + ((t >> 1) & ZTM) * 8u;
#pragma unroll 1
        for (uint32_t hh = 0; hh < 2; hh++) {
          const v4f *slot = ring + slot_i * kBlkSlotV4;
          const v4f *yslot = yring + (t % NY) * 2u * 64u;    // (wave 0 wrote half tile t of this launch there)
          const v4f ua = slot[lane], ub = slot[64u + lane];  // pair sums where this wave's own span covers the lane, else (phase, magnitude)
          const v4f ya = yslot[lane], yb = yslot[64u + lane];
          const uint4 rq = *reinterpret_cast<const uint4 *>(prow + pidx);
          v4f zq[4] = {};
          if (UNI) {
#pragma unroll
            for (int i = 0; i < 4; i++) zq[i] = ztile[4u * hh + (uint32_t)i];
          }
          const float uv[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
          const float yv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
          const uint32_t ro[4] = {rq.x, rq.y, rq.z, rq.w};
          uint32_t rn[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            X.k++;
            X.kv += 1u;
            const float zph[4] = {zq[j].x, zq[j].y, zq[j].z, zq[j].w};
            back_pair<UNI, true, false, true, COH>(B, MED ? Ks : Kp, P, S, M, &rn[j], lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, X,
                                              uv[2 * j], uv[2 * j + 1], &yv[2 * j], ro[j], inc, uv[2 * j], uv[2 * j + 1], UNI ? zph : nullptr);
            amp_advance(X.amp_soff, amp_quad_bytes, amp_wrap);
          }
          *reinterpret_cast<uint4 *>(prow + pidx) = make_uint4(rn[0], rn[1], rn[2], rn[3]);
          slot_i = slot_i + 1u == kBlkSlots ? 0u : slot_i + 1u;
          pidx = pidx + 4u >= P.d ? 0u : pidx + 4u;
          t++;
          lds_post(&ctr[3], t);
        }
      }
      if ((t & (kFlushBlocks - 1u)) == 0u) blk_flush(B, bq, nq, M, out, (uint32_t)out_pitch);
    }
    blk_flush(B, bq, nq, M, out, (uint32_t)out_pitch);
#if FSK_BLK_LEAN
    if (matched_stale) matched_now();
#endif
    FSK_STAMP_END(3)
    // what the host picks the next call's kernel by: a sample -- every 64th group reports (two atomics on two words from every
    // workgroup of a launch serialise at one L2 channel: +16 us on a 40-us call of 65 536 streams x 128 samples)
    if (lane == 0 && S.blk_stat && (grp & 63u) == 0u) {
      __hip_atomic_fetch_add(&S.blk_stat[0], (uint32_t)n_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&S.blk_stat[1], rare_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    FSK_WAIT_BEGIN
    while (produced <= nh) {
      produced = lds_peek(&ctr[2]);
      if (produced <= nh) FSK_SPIN(1, S.blk_stat);
    }
    while (lds_peek(&ctr[1]) <= nh) FSK_SPIN(1, S.blk_stat);
    FrontLane F;
    {
      const v4f fi = fin[lane], fq = fin[64u + lane];
      F.ix1 = fi.x; F.ix2 = fi.y; F.iy = fi.z; F.iv = fi.w;
      F.qx1 = fq.x; F.qx2 = fq.y; F.qy = fq.z; F.qv = fq.w;
      F.g = F.bx1 = F.bx2 = F.by1 = F.by2 = 0.f;
    }
    if (B.dph >= kOwnPairs4) {                                // the correction as the discriminator wave left it
      const v4f qz = fin[128u + lane];
      B.qai = qz.x; B.qaq = qz.y; B.qbi = qz.z; B.qbq = qz.w;
      // ... unless the hand-over sample is the FIRST of the next launch (or time slice): this wave has let go of the
      // correction (zr_dph reached kHandPairs with the launch's last sample) and the discriminator wave has not taken it
      // yet -- it is still in the mailbox.  (Round 4: found by the idle-bank test, where streams reset every 140
      // decimated samples; a launch boundary there used to drop the un-retired correction and an 'eod' moved.)
      if (cmail[lane] == X.k) {
        B.qai = __builtin_bit_cast(float, cmail[64u + lane]); B.qaq = __builtin_bit_cast(float, cmail[128u + lane]);
        B.qbi = __builtin_bit_cast(float, cmail[192u + lane]); B.qbq = __builtin_bit_cast(float, cmail[256u + lane]);
        const uint32_t steps = X.k > kOwnLag4 ? kOwnLag4 : 0u;   // (what the discriminator wave would have run on taking them)
        for (uint32_t g = 0; g < steps; g++) {
          const float ni = __builtin_fmaf(Ks.c1, B.qbi, -(Ks.c2 * B.qai)), nq = __builtin_fmaf(Ks.c1, B.qbq, -(Ks.c2 * B.qaq));
          B.qai = B.qbi; B.qaq = B.qbq; B.qbi = ni; B.qbq = nq;
        }
      }
      // zr_dph as every kernel understands it: decimated samples since the reset, saturating at kHandPairs (this wave stopped
      // counting at kOwnPairs4; the mailbox has the sample of this launch the stream was last reset at, as a signed number)
      const uint32_t since = X.k - cmail[320u + lane];
      B.dph = since < kHandPairs ? since : kHandPairs;
    }
    {
      uint32_t ph = phase0;
      for (uint32_t i = 0; i < P.d; i++) {
        if (mine) {
          if (COH) __hip_atomic_store(&gpoly[ph * 64u + lane], poly[lane * PS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else gpoly[ph * 64u + lane] = poly[lane * PS + i];
        }
        ph = ph + 1u == P.d ? 0u : ph + 1u;
      }
    }
    const uint32_t phase_end = (phase0 + X.k) % P.d;
    pipe_store<UNI, COH>(F, false, B, P, C, stream, out_counts, n, X.k, X.k % P.cadence, phase_end, amp_pos_of(X.amp_soff, amp_quad_bytes), inc, free0);
  }
  if (!SL) return;
  // ---- the group's next slice goes behind the queue's tail once this one's state is in memory (pushed at the loop's top)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  push_e = slice + 1u < Z.nslices ? 0x80000000u | ((slice + 1u) << 20) | grp : 0u;
  }
}

template <bool WB, bool UNI, bool SL>
__global__ __launch_bounds__(256, FSK_BLK_WGS) void demod_blk_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n_call, size_t pitch, int append_call,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts, BlkSched Z) {
  demod_blk_body<WB, UNI, SL, false>(P, S, samples, n_call, pitch, append_call, out, out_pitch, out_counts, eod_counts, Z);
}
// ... with the block path that takes resets
template <bool WB, bool SL>
__global__ __launch_bounds__(256, 4) void demod_blk_kernel_r(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n_call, size_t pitch, int append_call,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts, BlkSched Z) {
  demod_blk_body<WB, true, SL, true>(P, S, samples, n_call, pitch, append_call, out, out_pitch, out_counts, eod_counts, Z);
}
// ... and for per-stream tone pairs (round 5)
template <bool WB, bool SL>
__global__ __launch_bounds__(256, 4) void demod_blk_kernel_rp(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n_call, size_t pitch, int append_call,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts, BlkSched Z) {
  demod_blk_body<WB, false, SL, true>(P, S, samples, n_call, pitch, append_call, out, out_pitch, out_counts, eod_counts, Z);
}

// ... and with five waves per group (round 6): the front wave's two halves on a wave each (see demod_blk_body, NW = 5).
// MEASUREMENT BUILDS ONLY (-DFSK_BLK_FIVE, tools/build_variant.sh): bit-exact with the four-wave kernel but 16-19 % slower at
// equal residency, and only three of its 320-thread workgroups are resident per CU at 96 VGPRs (profiles/r06_five_wave.txt);
// the shipped library does not contain it and refuses kernel = five-wave.
#ifdef FSK_BLK_FIVE
template <bool WB, bool UNI, bool SL>
__global__ __launch_bounds__(320, FSK_BLK5_WPE) void demod_blk5_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n_call, size_t pitch, int append_call,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts, BlkSched Z) {
  demod_blk_body<WB, UNI, SL, false, 5>(P, S, samples, n_call, pitch, append_call, out, out_pitch, out_counts, eod_counts, Z);
}
#endif
bool demod_blk5_built() {
#ifdef FSK_BLK_FIVE
  return true;
#else
  return false;
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------
size_t demod_blk_lds_bytes(const DemodParams &P, uint32_t y_slots, uint32_t waves) {
  return sizeof(float4) * ((waves == 5u ? 8 : 4) * kSlotStride + y_slots * 2 * 64 + kBlkSlots * kBlkSlotV4 + blk_zt_tiles(y_slots) * 8) +
         sizeof(uint32_t) * (64u * blk_poly_stride(P.d) + 8u + 64u + 6u * 64u) + FSK_BLK_LDS_PAD;
}
size_t demod_blk_lds_bytes(const DemodParams &P, uint32_t y_slots) { return demod_blk_lds_bytes(P, y_slots, 4u); }
size_t demod_blk_lds_bytes(const DemodParams &P) { return demod_blk_lds_bytes(P, kBlkSlots); }
// the block path needs whole blocks of polyphase registers (dsSPB a multiple of 4) and at most one bit decision per block
// (in practice dsSPB = 20, 40, 80: the whole-tile kernels only see configurations whose sync-ring capacity, (bits + 32) *
// dsSPB * 1.1 in doubles, is an integer -- fsk_api.hip)
bool demod_blk_applicable(const DemodParams &P) { return P.d >= 8u && (P.d & 3u) == 0u && !P.wide && !P.frac; }

hipError_t set_blk_lds_limit(const DemodParams &P) {
  hipError_t e = hipSuccess;
  if (demod_blk_lds_bytes(P) > 160 * 1024) return hipSuccess;
  size_t bytes = demod_blk_lds_bytes(P, kBlkYMax, 5u);
  bytes = bytes > 160 * 1024 ? 160 * 1024 : bytes;
#define FSK_ATTR(WBV, UNIV, SLV)                                                                                 \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_blk_kernel<WBV, UNIV, SLV>),                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  FSK_ATTR(false, false, false) FSK_ATTR(false, true, false) FSK_ATTR(true, false, false) FSK_ATTR(true, true, false)
  FSK_ATTR(false, false, true) FSK_ATTR(false, true, true) FSK_ATTR(true, false, true) FSK_ATTR(true, true, true)
#undef FSK_ATTR
#ifdef FSK_BLK_FIVE
#define FSK_ATTR(WBV, UNIV, SLV)                                                                                 \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_blk5_kernel<WBV, UNIV, SLV>),                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  FSK_ATTR(false, false, false) FSK_ATTR(false, true, false) FSK_ATTR(true, false, false) FSK_ATTR(true, true, false)
  FSK_ATTR(false, false, true) FSK_ATTR(false, true, true) FSK_ATTR(true, false, true) FSK_ATTR(true, true, true)
#undef FSK_ATTR
#endif
#define FSK_ATTR(WBV, SLV)                                                                                       \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_blk_kernel_r<WBV, SLV>),                      \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  FSK_ATTR(false, false) FSK_ATTR(false, true) FSK_ATTR(true, false) FSK_ATTR(true, true)
#undef FSK_ATTR
#define FSK_ATTR(WBV, SLV)                                                                                       \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_blk_kernel_rp<WBV, SLV>),                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  FSK_ATTR(false, false) FSK_ATTR(false, true) FSK_ATTR(true, false) FSK_ATTR(true, true)
#undef FSK_ATTR
  return e;
}

// How to launch a batch of `groups` on `device`: the y ring as deep as the LDS allows while every CU still holds its
// share of the groups (at most four workgroups: the register file's limit), and how many workgroups the device then
// holds at once (one round; larger batches are run persistently, in time slices).  Zeros if it cannot tell.
void demod_blk_plan(const DemodParams &P, uint32_t groups, int device, uint32_t *y_slots, uint32_t *resident_wgs, uint32_t waves) {
  *y_slots = kBlkSlots; *resident_wgs = 0;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return;
  auto held = [&](uint32_t y) {
    int per_cu = 0;
#ifdef FSK_BLK_FIVE
    const hipError_t oe = waves == 5u
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&demod_blk5_kernel<false, true, true>), 320,
                                                       demod_blk_lds_bytes(P, y, 5u))
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&demod_blk_kernel<false, true, true>), 256,
                                                       demod_blk_lds_bytes(P, y));
#else
    if (waves == 5u) return 0u;
    const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&demod_blk_kernel<false, true, true>), 256,
                                                                       demod_blk_lds_bytes(P, y));
#endif
    if (oe != hipSuccess) return 0u;
    return per_cu > 0 ? (uint32_t)per_cu : 0u;
  };
  const uint32_t most = held(kBlkSlots);                       // (four)
  if (!most) return;
  uint32_t want = (groups + (uint32_t)cus - 1u) / (uint32_t)cus;
  want = want < 1u ? 1u : want > most ? most : want;
  uint32_t y = kBlkSlots;
  // (the occupancy query, and 512-byte granules, let a third workgroup "fit" beside 2 x 54 112 bytes; the hardware did not
  // place it and 49 152 streams ran in two rounds.  1 280 = 160 KB / 128 and 2 048 both explain what was measured.)
  auto fits = [&](uint32_t yy) {
    const size_t b = demod_blk_lds_bytes(P, yy, waves);
    return (size_t)want * ((b + 1279u) / 1280u * 1280u) <= 160u * 1024u && (size_t)want * ((b + 2047u) & ~(size_t)2047u) <= 160u * 1024u;
  };
  while (y + 2u <= kBlkYMax && fits(y + 2u) && held(y + 2u) >= want) y += 2u;
  *y_slots = y;
  *resident_wgs = held(y) * (uint32_t)cus;
}

#ifdef FSK_ABLATE
static void set_ablate_blk() {
  const char *a = getenv("FSK_ABLATE");
  const int v = a ? atoi(a) : 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &v, sizeof(v));
}
#else
static inline void set_ablate_blk() {}
#endif

// Time slices (see BlkSched; how a call is cut: fsk_blk_sched.h)
size_t demod_blk_queue_words(uint32_t groups) { return 16u + (size_t)groups * (kBlkMaxSlices - 1u); }

// Slices per group for a call of n samples (1 = one workgroup per group, not persistent) and their length in tiles: the
// arithmetic is fsk_blk_sched.h's.
uint32_t demod_blk_slices(const DemodParams &P, const DemodState &S, size_t n, uint32_t resident_wgs, uint32_t slice_tiles,
                          uint32_t *slice_tiles_out) {
  return blk_slice_count((P.n_streams + 63u) / 64u, (uint32_t)(n / kFastTile), resident_wgs, slice_tiles, S.blk_q != nullptr,
                         slice_tiles_out);
}

// Narrow groups (round 4).  A workgroup's pace is its slowest wave's instruction stream and does not depend on how many
// of the 64 lanes carry a stream; a batch of fewer than 64 x (compute units) streams leaves whole CUs idle while the
// groups it does have run at that pace (4 096 streams: 64 workgroups on 256 CUs).  Such a batch is cut into groups of
// 32, 16 or 8 streams instead -- the widest that still gives every workgroup a CU of its own -- and the idle lanes are
// the price of using the idle CUs.  Above 64 x CUs streams the groups are whole waves as before.
uint32_t demod_blk_lanes(uint32_t n_streams, int device) {
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 64u;
  for (uint32_t w = 8u; w < 64u; w *= 2u)
    if ((n_streams + w - 1u) / w <= (uint32_t)cus) return w;
  return 64u;
}

hipError_t launch_demod_blk(bool writeback, bool append, const DemodParams &P, const DemodState &S, float *samples, size_t n,
                             size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts,
                             uint32_t *eod_counts, hipStream_t stream, uint32_t resident_wgs, uint32_t slice_tiles, uint32_t y_slots,
                             uint32_t lanes, uint32_t medium, bool *sliced_out, uint32_t waves) {
  lanes = (lanes == 8u || lanes == 16u || lanes == 32u) ? lanes : 64u;
  if (medium != 0u || !demod_blk5_built()) waves = 4u;        // (the kernels whose block path takes resets have four waves)
  const uint32_t blocks = (P.n_streams + lanes - 1u) / lanes;
  y_slots = y_slots < kBlkSlots ? kBlkSlots : y_slots > kBlkYMax ? kBlkYMax : y_slots;
  const size_t lds = demod_blk_lds_bytes(P, y_slots, waves);
  set_ablate_blk();
  BlkSched Z = {nullptr, blocks, 1u, 0u, 0u, y_slots, blk_zt_tiles(y_slots), lanes, medium};
  uint32_t st = 0;
  // (time slices are for batches beyond one round of whole-wave groups; narrow groups are never sliced -- the queue is
  // sized for 64-stream groups)
  const uint32_t ns = lanes == 64u ? demod_blk_slices(P, S, n, resident_wgs, slice_tiles, &st) : 1u;
  const bool sliced = ns >= 2u;
  if (sliced) {
    Z.q = S.blk_q; Z.nslices = ns; Z.slice_tiles = st; Z.total = blocks * ns;
    const hipError_t e = hipMemsetAsync(S.blk_q, 0, sizeof(uint32_t) * (16u + (size_t)blocks * (ns - 1u)), stream);
    if (e != hipSuccess) return e;
  }
  if (sliced_out) *sliced_out = sliced;
  const uint32_t grid = sliced ? resident_wgs : blocks;
#define FSK_LAUNCH_BLK(WBV, UNIV, SLV)                                                                          \
  hipLaunchKernelGGL((demod_blk_kernel<WBV, UNIV, SLV>), dim3(grid), dim3(256), lds, stream, P, S, samples, n, pitch, \
                     append ? 1 : 0, out, out_pitch, out_counts, eod_counts, Z)
  const bool uni = P.uni_cfg != 0;
#define FSK_LAUNCH_BLKR(WBV, SLV)                                                                               \
  hipLaunchKernelGGL((demod_blk_kernel_r<WBV, SLV>), dim3(grid), dim3(256), lds, stream, P, S, samples, n, pitch, \
                     append ? 1 : 0, out, out_pitch, out_counts, eod_counts, Z)
  if (uni && medium != 0u) {
    if (sliced) { if (writeback) FSK_LAUNCH_BLKR(true, true); else FSK_LAUNCH_BLKR(false, true); }
    else { if (writeback) FSK_LAUNCH_BLKR(true, false); else FSK_LAUNCH_BLKR(false, false); }
    return hipGetLastError();
  }
#undef FSK_LAUNCH_BLKR
#define FSK_LAUNCH_BLKRP(WBV, SLV)                                                                              \
  hipLaunchKernelGGL((demod_blk_kernel_rp<WBV, SLV>), dim3(grid), dim3(256), lds, stream, P, S, samples, n, pitch, \
                     append ? 1 : 0, out, out_pitch, out_counts, eod_counts, Z)
  if (!uni && medium != 0u) {
    if (sliced) { if (writeback) FSK_LAUNCH_BLKRP(true, true); else FSK_LAUNCH_BLKRP(false, true); }
    else { if (writeback) FSK_LAUNCH_BLKRP(true, false); else FSK_LAUNCH_BLKRP(false, false); }
    return hipGetLastError();
  }
#undef FSK_LAUNCH_BLKRP
#ifdef FSK_BLK_FIVE
#define FSK_LAUNCH_BLK5(WBV, UNIV, SLV)                                                                         \
  hipLaunchKernelGGL((demod_blk5_kernel<WBV, UNIV, SLV>), dim3(grid), dim3(320), lds, stream, P, S, samples, n, pitch, \
                     append ? 1 : 0, out, out_pitch, out_counts, eod_counts, Z)
  if (waves == 5u) {
    if (sliced) {
      if (writeback) { if (uni) FSK_LAUNCH_BLK5(true, true, true); else FSK_LAUNCH_BLK5(true, false, true); }
      else { if (uni) FSK_LAUNCH_BLK5(false, true, true); else FSK_LAUNCH_BLK5(false, false, true); }
    } else {
      if (writeback) { if (uni) FSK_LAUNCH_BLK5(true, true, false); else FSK_LAUNCH_BLK5(true, false, false); }
      else { if (uni) FSK_LAUNCH_BLK5(false, true, false); else FSK_LAUNCH_BLK5(false, false, false); }
    }
    return hipGetLastError();
  }
#undef FSK_LAUNCH_BLK5
#endif
  if (sliced) {
    if (writeback) { if (uni) FSK_LAUNCH_BLK(true, true, true); else FSK_LAUNCH_BLK(true, false, true); }
    else { if (uni) FSK_LAUNCH_BLK(false, true, true); else FSK_LAUNCH_BLK(false, false, true); }
  } else {
    if (writeback) { if (uni) FSK_LAUNCH_BLK(true, true, false); else FSK_LAUNCH_BLK(true, false, false); }
    else { if (uni) FSK_LAUNCH_BLK(false, true, false); else FSK_LAUNCH_BLK(false, false, false); }
  }
#undef FSK_LAUNCH_BLK
  return hipGetLastError();
}

}  // namespace fsk

#ifdef FSK_STAMP
extern "C" int fskdbg_read_stamps_blk(unsigned long long *out, size_t count) {   // diagnostic builds only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fsk::g_stamp), count * sizeof(unsigned long long));
}
#endif
