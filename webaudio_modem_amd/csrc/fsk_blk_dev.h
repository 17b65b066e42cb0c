// fsk_blk_dev.h -- device code shared by the block-batched whole-tile kernels: the four-wave kernels of fsk_blk.hip and the
// seven-wave small-batch kernel of fsk_blk6.hip.  Ring geometry, the bit clock of one block (blk_clock), the fast path of one
// block (blk_fast), the byte queue flush, the discriminator wave's correction lane.  See fsk_blk.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fsk_params.h"
#include "fsk_dev.h"
#include "fsk_pipe_dev.h"

namespace fsk {

#ifndef FSK_BLK_SLOTS
#define FSK_BLK_SLOTS 6
#endif
static constexpr uint32_t kBlkSlots = FSK_BLK_SLOTS;   // half tiles in the rings
static_assert((kBlkSlots & 1u) == 0u, "blocks are two half tiles on an even slot");
static_assert(4u * kBlkSlots <= kZeroLagPairs, "the wave that owns the I/Q low-pass must learn of a reset before it has passed the zeroing point");
static_assert(4u * kBlkSlots <= kOwnLag4, "the discriminator wave (up to 4 * kBlkSlots - 1 samples beyond the start of the back wave's tile) must not have reached the hand-over sample when it is posted");
static constexpr uint32_t kBlkSlotV4 = 2 * 64;     // v4f per x-ring slot: four pair sums (I, Q) -- in place -> four (phase, magnitude)
static constexpr uint32_t kFlushBlocks = 16;       // byte queues are flushed every this many blocks
// The y ring (wave 0 -> wave 1) may be deeper than the x ring.  With six slots everywhere the four waves hold exactly
// three tiles between them and each tile goes round the closed chain "back wave frees a slot -> wave 0 -> 1 -> 2 -> back
// wave": the group advances a tile per (t0 + t1 + t2 + t3) / 3, not per max(t) (measured busy cycles per sample at 8 192
// streams 116 + 132 + 121 + 148 = 517 -> 172; the loop ran at 182).  Only wave 1 .. 3's distance is tied to the lag
// constants (static_asserts above); wave 0 has no such tie, so its ring takes whatever LDS the launch has to spare.
static constexpr uint32_t kBlkYMax = 28;
// (tiles of phasors in flight: wave 0 runs at most y_slots - 2 half tiles ahead of the tile the back wave reads, so a power of two
// above y_slots / 2 holds them; 4 at the six slots a full device gets -- which also keeps the workgroup's LDS 670 bytes under the
// 29-granule mark above which config #3 loses 2 %, profiles/r04_block_resets.txt section 11-12)
__host__ __device__ inline uint32_t blk_zt_tiles(uint32_t y_slots) { return y_slots <= 6u ? 4u : y_slots <= 12u ? 8u : 16u; }

// lane stride of the polyphase registers in LDS: >= d, = 4 mod 8, so that the ds_read_b128 of 16 lanes at consecutive
// strides touches 64 different banks (20 for 1200 baud, 84 for 300 baud)
__host__ __device__ inline uint32_t blk_poly_stride(uint32_t d) {
  uint32_t s = (d + 3u) & ~3u;
  return (s & 4u) ? s : s + 4u;
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct BlkK {                    // block-level constants, VGPRs
  uint32_t stop_m1;              // 2^stop_pos - 1: sreg > stop_m1 <=> all data bits in
  uint32_t sh9;                  // stop_pos - 9: byte = (sreg >> sh9) & 0xFF
  uint32_t ff;                   // 0xFF
};

// Bit clock of one block (fsk.ts:335-341) and processByte (346-375), evaluated once from the block's eight slicer bits w
// (sample 1 in bit kBlk - 1): a lane decides at most one bit per block, at sample jd = nextBitSampleIndex - k0.  Shared by
// the fast block path and the one that takes resets.  Returns the bad start / stop bit flags (sign bit = rare), md = all
// ones in the lanes that decide a bit in this block.
static constexpr int kBlk = 8;                     // decimated samples per block
__device__ inline uint32_t blk_clock(BackLane &Bn, const BackK &K, const BlkK &Q, uint32_t kv0, uint32_t w, uint32_t &bq,
                                     uint32_t &nq, uint32_t &md_out) {
  uint32_t jd = Bn.T - kv0;                        // 1..kBlk in this block; 0 right after a sync (nextBitSampleIndex = k)
  jd -= neg_mask(jd - 1u);                         // 0 -> 1
  const uint32_t md = neg_mask(jd - (uint32_t)(kBlk + 1));   // all ones <=> a decision falls into this block
  const uint32_t hi = w >> (((uint32_t)kBlk - jd) & 31u);    // the slicer bits of samples 1..jd (garbage without a decision: masked)
  const uint32_t nhi = (uint32_t)__builtin_popcount(hi);
  const uint32_t ones = nhi + Bn.acc;
  const uint32_t tot = (uint32_t)__builtin_popcount(w);
  const uint32_t kvd = kv0 + jd;
  const uint32_t b = sign_bit((kvd - Bn.tlast) - ones - ones);               // 2 * bitAccumulator > bitAccumCount
  const uint32_t s0 = Bn.sreg;
  const uint32_t s1 = s0 + s0 + b;
  // processByte (fsk.ts:346-375) at the decision
  const uint32_t m_start = neg_mask(s0 - 2u);                                // waiting for the start bit
  const uint32_t m_stop = neg_mask(Q.stop_m1 - s0);                          // all data bits in: stop (or parity) position
  const uint32_t bm = 0u - b;
  const uint32_t good = md & m_stop & bm;                                    // a byte completes
  Bn.acc = tot + (Bn.acc & ~md) - (nhi & md);
  Bn.T += K.d & md;
  Bn.tlast = (Bn.tlast & ~md) | (kvd & md);
  Bn.sreg = (s0 & ~md) | (s1 & md & ~good) | (1u & good);
  const uint32_t byte = (s0 >> Q.sh9) & Q.ff;
  bq = (bq & ~good) | (((bq << 8) | byte) & good);
  nq -= good;
  md_out = md;
  return md & ((m_stop & ~bm) | (m_start & bm));                             // bad stop bit / bad start bit
}

// The fast path of one block (a tile: eight decimated samples).  Works on copies (Bn, rp, bq, nq): the caller commits
// them only if the returned flag word has its sign bit clear in every lane.  kv0 = pushes before the block.  hard_out:
// the same without the 'eod' bound -- a sync candidate or a bad start / stop bit, which only the per-sample path takes.
// LEAN (round 5): every stream of the wave is inside a frame (thr_eff = kStartedP: no sync search, fsk.ts:297), so the
// correlator's running count cannot matter before a rare path is taken: it is not carried (seven of the vector instructions
// per decimated sample) and the caller re-forms it from the polyphase registers when it next needs it.
template <bool LEAN = false>
__device__ inline uint32_t blk_fast(BackLane &Bn, const BackK &K, const BlkK &Q, uint32_t kv0, const v4f (&pa)[4],
                                    uint32_t (&rp)[kBlk], float (&am)[kBlk], uint32_t &bq, uint32_t &nq, uint32_t &hard_out) {
  const float phs[kBlk] = {pa[0].x, pa[0].z, pa[1].x, pa[1].z, pa[2].x, pa[2].z, pa[3].x, pa[3].z};
  am[0] = pa[0].y; am[1] = pa[0].w; am[2] = pa[1].y; am[3] = pa[1].w;
  am[4] = pa[2].y; am[5] = pa[2].w; am[6] = pa[3].y; am[7] = pa[3].w;
  uint32_t w = 0, hard = 0;
  // inside the block the correlator's count is carried as its distance to the threshold and the last loud sample as its
  // index relative to the block (1 .. kBlk, or <= 0 for "before it"): the per-sample test is then one operation and the
  // sample index an inline constant
  uint32_t dm = Bn.matched - Bn.thr_eff;
  const uint32_t lsr0 = Bn.ls - kv0;
  uint32_t lsr = lsr0;
#pragma unroll
  for (int j = 0; j < kBlk; j++) {
    const float f = disc_post(Bn, K, phs[j], am[j]);                        // fsk.ts:251-261
    // slicer (fsk.ts:264): the bit is the sign of 0 - f (f = +-0 gives +0, bit 0); it is shifted into the registers
    // straight from there (v_alignbit: {hi, lo} >> 31 = hi << 1 | sign of lo) without being extracted first
    const uint32_t nf = __builtin_bit_cast(uint32_t, slicer_nf(f));
    const uint32_t rold = rp[j];
    const uint32_t r = __builtin_amdgcn_alignbit(rold, nf, 31);              // syncSamplesBuffer.put(bit)
    rp[j] = r;
    if (!LEAN) {
      dm += (uint32_t)__builtin_popcount((r ^ K.qn) & K.mask);
      dm -= (uint32_t)__builtin_popcount((rold ^ K.qn) & K.mask);
      hard |= ~dm;                                                           // sign set <=> matched >= thr_eff (sync candidate)
    }
    const uint32_t silent = neg_mask(__builtin_bit_cast(uint32_t, am[j] - Bn.thr));   // fsk.ts:285
    lsr = (lsr & silent) | ((uint32_t)(j + 1) & ~silent);
    w = __builtin_amdgcn_alignbit(w, nf, 31);                                // sample 1 ends up in bit kBlk - 1
  }
  if (!LEAN) Bn.matched = dm + Bn.thr_eff;
  Bn.ls = lsr + kv0;
  // 'eod' (fsk.ts:288): no silence run inside the block is longer than the one a wholly silent block would end with
  // (eod_m1 - ((kv0 + kBlk) - ls at entry))
  const uint32_t soft = K.eod_m1 - (uint32_t)kBlk + lsr0;
  // ---- bit clock, once per block (fsk.ts:335-341)
  uint32_t md;
  hard |= blk_clock(Bn, K, Q, kv0, w, bq, nq, md);
  hard_out = hard;
  return hard | soft;
}

// completed bytes of the fast path -> out (oldest first); B.out_cnt counts them as the per-sample path does
__device__ inline void blk_flush(BackLane &B, uint32_t &bq, uint32_t &nq, const FastMem &M, uint8_t *out, uint32_t out_pitch) {
  while (__builtin_amdgcn_ballot_w64(nq != 0u)) {
    if (nq != 0u) {
      nq -= 1u;
      const uint32_t byte = (bq >> (8u * nq)) & 0xFFu;
      if (M.voff < 0xFFFFFFF0u && B.out_cnt < out_pitch) out[(size_t)(M.voff >> 2) * out_pitch + B.out_cnt] = (uint8_t)byte;
      B.out_cnt++;
    }
  }
}

// The discriminator wave's share of the ZIR correction (see back_pair for the arithmetic it restates op for op):
// w = U - q, q advances by its recurrence and retires to exactly zero once below 2^-28 of the magnitude it corrects.
struct QLane { float ai, aq, bi, bq; };


// ---- the block path that takes resets: its per-sample arithmetic, shared by fsk_blk.hip (blk_medium) and fsk_blk6.hip ------
struct MedEv {
  uint32_t jr;                   // 1..8: resetState() ran at the end of this sample of the block; 0 = no reset
  uint32_t jc;                   // 1..8: the correction's start values were formed at this sample; 0 = not in this block
  float cai, caq, cbi, cbq;      // those values (zq_a, zq_b right after their formation)
};
__device__ inline uint32_t bsel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
__device__ inline float bself(uint32_t m, float a, float b) {
  return __builtin_bit_cast(float, bsel(m, __builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b)));
}
__device__ inline float bzero(uint32_t keep, float a) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) & keep); }
__device__ inline uint32_t eq_mask(uint32_t v, uint32_t c) { return neg_mask((v ^ c) - 1u); }   // all ones <=> v == c (small values)


// One decimated sample of the block path that takes resets (see blk_medium in fsk_blk.hip for what it restates): e0 / e1 the
// x ring entry (pair sum where the lane's span is the back wave's, else phase and magnitude), y0 / y1 the pair's pre-filter
// outputs, z = (c0, s0, c1, s1) their NCO phasors, thf_j lastPhase after a reset at this sample, rold the sample's polyphase
// register.  ONE function for both kernels: the same instruction sequence per decimated sample whichever runs it.
__device__ __forceinline__ void med_sample(BackLane &Bn, const BackK &K, const int j, const float e0, const float e1, const float y0, const float y1,
                                           const v4f z, const float thf_j, const uint32_t rold, const uint32_t kvj, const uint32_t matched_min, const uint32_t own_pairs,
                                           uint32_t &matched, uint32_t &thr_cur, uint32_t &ls, uint32_t &w, uint32_t &hard, MedEv &E, float &am_j) {
    // ---- zir_step<UNI, HAND = true>, flat
    const uint32_t dph0 = Bn.dph;
    float wi = e0 - Bn.qai, wq = e1 - Bn.qaq;
    {
      const float ni = __builtin_fmaf(K.c1, Bn.qbi, -(K.c2 * Bn.qai));
      const float nq2 = __builtin_fmaf(K.c1, Bn.qbq, -(K.c2 * Bn.qaq));
      Bn.qai = Bn.qbi; Bn.qaq = Bn.qbq; Bn.qbi = ni; Bn.qbq = nq2;
    }
    float di, dq;
    {
      const float mi = y0 * z.x, mq = y0 * z.y;
      const float ti = __builtin_fmaf(2.0f, Bn.dix1, mi) + Bn.dix2, tq = __builtin_fmaf(2.0f, Bn.dqx1, mq) + Bn.dqx2;
      Bn.dvi = __builtin_fmaf(K.lp_a2, Bn.dvi, __builtin_fmaf(K.lp_nd, Bn.diy, ti));
      Bn.dqv = __builtin_fmaf(K.lp_a2, Bn.dqv, __builtin_fmaf(K.lp_nd, Bn.dqy, tq));
      Bn.diy += Bn.dvi; Bn.dqy += Bn.dqv;
      Bn.dix2 = Bn.dix1; Bn.dix1 = mi; Bn.dqx2 = Bn.dqx1; Bn.dqx1 = mq;
      di = Bn.diy; dq = Bn.dqy;
    }
    {
      const float mi = y1 * z.z, mq = y1 * z.w;
      const float ti = __builtin_fmaf(2.0f, Bn.dix1, mi) + Bn.dix2, tq = __builtin_fmaf(2.0f, Bn.dqx1, mq) + Bn.dqx2;
      Bn.dvi = __builtin_fmaf(K.lp_a2, Bn.dvi, __builtin_fmaf(K.lp_nd, Bn.diy, ti));
      Bn.dqv = __builtin_fmaf(K.lp_a2, Bn.dqv, __builtin_fmaf(K.lp_nd, Bn.dqy, tq));
      Bn.diy += Bn.dvi; Bn.dqy += Bn.dqv;
      Bn.dix2 = Bn.dix1; Bn.dix1 = mi; Bn.dqx2 = Bn.dqx1; Bn.dqx1 = mq;
      di += Bn.diy; dq += Bn.dqy;
    }
    // (lanes past the direct instance run it on as well: its state is not theirs any more -- pipe_store writes zeros for it
    // once zr_dph has reached kDirectPairs, whichever path the samples took)
    const uint32_t m_dir = neg_mask(dph0 - kDirectPairs);                    // zr_dph < kDirectPairs: the direct instance's output
    wi = bself(m_dir, di, wi); wq = bself(m_dir, dq, wq);
    {
      const float qi = e0 - di, qq = e1 - dq;                                // U - d: the free-running filters' zero-input response
      const uint32_t m24 = eq_mask(dph0, kZeroLagPairs), m25 = eq_mask(dph0, kZeroLagPairs + 1u);
      const float nai = __builtin_fmaf(K.c1, qi, -(K.c2 * Bn.q0i)), naq = __builtin_fmaf(K.c1, qq, -(K.c2 * Bn.q0q));
      const float nbi = __builtin_fmaf(K.c1, nai, -(K.c2 * qi)), nbq = __builtin_fmaf(K.c1, naq, -(K.c2 * qq));
      Bn.q0i = bself(m24, qi, Bn.q0i); Bn.q0q = bself(m24, qq, Bn.q0q);
      Bn.qai = bself(m25, nai, Bn.qai); Bn.qaq = bself(m25, naq, Bn.qaq);
      Bn.qbi = bself(m25, nbi, Bn.qbi); Bn.qbq = bself(m25, nbq, Bn.qbq);
      E.cai = bself(m25, nai, E.cai); E.caq = bself(m25, naq, E.caq);
      E.cbi = bself(m25, nbi, E.cbi); E.cbq = bself(m25, nbq, E.cbq);
      E.jc = bsel(m25, (uint32_t)(j + 1), E.jc);
      // (materialised here: left to itself the compiler spills the eight samples' candidates and forms these in the rare
      // branch that posts them, behind sixteen scratch round trips)
      asm volatile("" : "+v"(E.cai), "+v"(E.caq), "+v"(E.cbi), "+v"(E.cbq), "+v"(E.jc));
    }
    const uint32_t m_own = neg_mask(dph0 - own_pairs);                       // the span is this wave's (own_pairs: kHandPairs, or kOwnPairs4 in fsk_blk.hip)
    Bn.dph = dph0 - m_own;                                                   // + 1, saturating at own_pairs
    {
      const uint32_t keep = ~eq_mask(dph0, own_pairs - 1u);                 // handed over: the discriminator wave's from here on
      Bn.qai = bzero(keep, Bn.qai); Bn.qaq = bzero(keep, Bn.qaq); Bn.qbi = bzero(keep, Bn.qbi); Bn.qbq = bzero(keep, Bn.qbq);
    }
    float a2;
    const float p2 = atan2_amp_fma(wq, wi, a2, K.tiny, K.sgn);
    const float ph = bself(m_own, p2, e0);
    am_j = bself(m_own, a2, e1);
    // ---- discriminator tail, slicer, correlator, silence run (as blk_fast)
    const float f = disc_post(Bn, K, ph, am_j);
    const uint32_t nf = __builtin_bit_cast(uint32_t, slicer_nf(f));
    const uint32_t r = __builtin_amdgcn_alignbit(rold, nf, 31);
    matched += (uint32_t)__builtin_popcount((r ^ K.qn) & K.mask);
    matched -= (uint32_t)__builtin_popcount((rold ^ K.qn) & K.mask);
    hard |= ~(matched - thr_cur);                                            // sync candidate
    const uint32_t silent = neg_mask(__builtin_bit_cast(uint32_t, am_j - Bn.thr));
    ls = (ls & silent) | (kvj & ~silent);
    w = __builtin_amdgcn_alignbit(w, nf, 31);
    // ---- 'eod' (fsk.ts:288) -> resetState() at the end of this sample (back_reset, masked)
    const uint32_t me = neg_mask(K.eod_m1 - (kvj - ls));
    hard |= me & (0u - E.jr);                                                // a second reset in the block
    E.jr = bsel(me, (uint32_t)(j + 1), E.jr);
    asm volatile("" : "+v"(E.jr), "+v"(hard));
    Bn.last_phase = bself(me, thf_j, Bn.last_phase);
    Bn.thf = bself(me, thf_j, Bn.thf);
    const uint32_t keep = ~me;
    Bn.dph &= keep;
    Bn.dix1 = bzero(keep, Bn.dix1); Bn.dix2 = bzero(keep, Bn.dix2); Bn.diy = bzero(keep, Bn.diy); Bn.dvi = bzero(keep, Bn.dvi);
    Bn.dqx1 = bzero(keep, Bn.dqx1); Bn.dqx2 = bzero(keep, Bn.dqx2); Bn.dqy = bzero(keep, Bn.dqy); Bn.dqv = bzero(keep, Bn.dqv);
    Bn.px1 = bzero(keep, Bn.px1); Bn.px2 = bzero(keep, Bn.px2); Bn.py = bzero(keep, Bn.py); Bn.pv = bzero(keep, Bn.pv);
    ls = bsel(me, kvj, ls);
    thr_cur = bsel(me, matched_min, thr_cur);
}
// ... and what follows the block's eight samples: the bit clock, once, and what resetState() leaves of it
__device__ __forceinline__ uint32_t med_finish(BackLane &Bn, const BackK &K, const BlkK &Q, const uint32_t kv0, const uint32_t matched,
                                               const uint32_t thr_cur, const uint32_t ls, const uint32_t w, uint32_t &bq, uint32_t &nq, const MedEv &E) {
  uint32_t hard = 0;
  Bn.matched = matched;
  Bn.ls = ls;
  // ---- bit clock, once per block, from the entry state; a lane that was reset in this block may not also decide a bit
  uint32_t md;
  hard |= blk_clock(Bn, K, Q, kv0, w, bq, nq, md);
  const uint32_t mjr = neg_mask(0u - E.jr);
  hard |= md & mjr;
  {
    // what resetState() leaves of the bit clock: the vote holds the bits sliced after the reset, no decision pending
    const uint32_t after = w & ((1u << (((uint32_t)kBlk - E.jr) & 31u)) - 1u);
    const uint32_t park = kv0 + E.jr + kBigWait;
    Bn.acc = bsel(mjr, (uint32_t)__builtin_popcount(after), Bn.acc);
    Bn.T = bsel(mjr, park, Bn.T);
    Bn.tlast = bsel(mjr, park, Bn.tlast);
    Bn.sreg = bsel(mjr, 1u, Bn.sreg);
  }
  Bn.thr_eff = thr_cur;
  return hard;
}

}  // namespace fsk
