// fsk_blk6.hip -- round-5 SMALL-BATCH whole-tile fp32 demodulator for gfx950 (MI355X): SEVEN waves per stream group
// (the file and the kernel keep the name of the first, six-wave cut).
//
// Why.  At <= 64 x (compute units) streams every workgroup of the four-wave kernel (fsk_blk.hip) has a compute unit to itself:
// one wave per SIMD, and a lone wave issues one instruction per ~4.9 cycles whatever it is (profiles/r02_valu_probe*.txt), so
// the launch takes as long as the longest of the four instruction streams -- the back wave's 141 busy cycles per input sample
// against 105 / 104 / 94 for the others (profiles/r03_blk_final_stamps.txt) -- while three quarters of every SIMD's issue slots
// and, with narrow groups, half to seven eighths of the lanes sit idle (BASELINE configs #2, #3-as-sharded
// and #5-as-sharded run at 2-7 % of the roofline; eight GPUs would buy config #3 2.0x).  There a workgroup may have the whole
// CU's registers and LDS, so this kernel cuts the same chain into SEVEN waves, and uses the idle lanes of a narrow group where
// a stage is not a recurrence (parts, as the role map numbers them):
//   0 loader   tile loads (three register sets in flight) -> a ring of staging tiles; the tile's sixteen NCO phasors
//   1 agc      the AGC (a recurrence of ten dependent instructions per sample), IN PLACE in the staging tile
//   2 bp       the pre-filter (with the AGC: what resetState() never touches)                         -> y ring
//   3 iq       mixer + free-running I/Q low-pass + pair sums U.  Groups of <= 32 streams: the I chain in lanes 0..31 and
//              the Q chain in lanes 32..63 of the SAME wave (two independent recurrences: half the instructions)  -> x ring
//   6 disc     ZIR correction (once it is this wave's) + branch-free atan2 / magnitude, x ring IN PLACE.  Stateless while no
//              correction is live, so a group of W streams spreads a tile's eight decimated samples over 64 / W lanes per
//              stream (W = 32: two lanes x four samples ... W = 8: eight lanes x one sample)
//   4 post     discriminator tail + post filter + slicer (disc_post, the very function every other kernel calls), running
//              AHEAD of the frame logic on the assumption that no resetState() intervenes         -> f ring (+ its entry
//              state per tile)
//   5 frame    sync correlator, silence run, bit clock, byte assembly, 'eod' -- blk_fast without its disc_post -- and every
//              rare path (blk_medium's sequence for own-span tiles, the per-sample back_pair), which is where resets come from
// Four stages work on the x ring -- iq, disc, post, frame -- and the reset feedback bounds how far they may spread: the iq
// wave learns of a resetState() kZeroLagPairs decimated samples late, the discriminator wave is handed the correction
// kHandLag ahead of its sample (fsk_params.h), so they may lead the frame wave by kZeroLagPairs / 8 and kHandLag / 8 tiles
// and no more.  At 24 / 24 (round 4's constants) that was three tiles for four stages: the ring turned once per SUM of the
// stage times and the hand-offs / 3 (x 1.04 - 1.22 over the four-wave kernel, profiles/r05_six_wave.txt).  Round 5 raised both
// to 48 -- six tiles -- which is what makes the stages overlap (profiles/r05_lag.txt: x 1.40 at 8 192 streams, x 1.54 at 2 048;
// every kernel of the library takes the same constants, so their results stay one another's bit for bit).
// Speculate and rewind (DESIGN.md section 9 of round 4, VERDICT r04 #1c).  resetState() (fsk.ts:175-188) zeroes the post filter
// and lastPhase from inside the frame logic, i.e. the frame wave feeds back into the post wave.  The post wave therefore
// records the state it ENTERED every tile with; a tile whose block test trips in the frame wave (a sync candidate, a bad
// start / stop bit, a possible 'eod') -- and every tile while a lane is inside the own span after a reset -- is redone by the
// frame wave from that entry state, exactly as the four-wave kernel's back wave redoes it; when it returns to its block loop
// it posts the state it ended with and the tile to resume at (a generation number makes the post wave's stale output
// recognisable), the post wave drops what it ran ahead and restarts there.  Its float sequence per decimated sample is
// disc_post's, its inputs are the x ring's, so the values do not depend on who computed them or how often: bytes, counters
// and carried state are the four-wave kernel's bit for bit (tests/test_gpu_parity.py runs every golden through this kernel,
// tests/test_gpu_fullsize.py compares state words).
// Uniform configurations only (per-stream tone pairs stay on the four-wave kernel); never time-sliced (a batch this small is
// one round of workgroups by definition).
// LDS (one workgroup per CU; XT = kZeroLagPairs / 8 tiles): stage [4 tiles][4][65] v4f | yring [y_slots][2][64] v4f | xring [XT][4][64]
//      v4f (I or phase 0..3, 4..7; Q or magnitude 0..3, 4..7) | fring [XT][4][64] v4f (-f 0..3, 4..7; magnitude 0..3, 4..7) | trash
//      [4][64] v4f | hist [XT][2][64] v4f | rmail [2][64] v4f | fin [5][64] v4f | zt [tiles][8] v4f (cos[16], sin[16]) | poly [64][PS]
//      u32 | counters [16] | zmail [64] | cmail [6][64]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "fsk_params.h"
#include "fsk_dev.h"
#include "fsk_pipe_dev.h"
#include "fsk_blk_dev.h"

namespace fsk {

static constexpr uint32_t kB6Stage = 4;                    // staging tiles between the loader and the AGC wave
static constexpr uint32_t kB6XT = kZeroLagPairs / 8;       // tiles in the x ring, the f ring and the entry-state history: as far as the wave that
                                                           // owns the I/Q low-pass may run ahead of the frame wave (a reset reaches it kZeroLagPairs late)
static constexpr uint32_t kB6DT = kHandLag / 8;            // ... and the discriminator wave (the correction is handed to it kHandLag ahead of its sample)
static constexpr uint32_t kB6Waves = 7;
static constexpr uint32_t kB6YMin = 2 * kB6XT + 2;         // half tiles of y ring at least: the frame wave's tile, those the iq wave is ahead, one being written
static_assert(kZeroLagPairs % 8 == 0 && kHandLag % 8 == 0 && kB6DT <= kB6XT, "whole tiles");
static constexpr uint32_t kB6TileV4 = 4 * 64;              // v4f per x / f ring tile
static constexpr uint32_t kB6YMax = 24;
// counters (LDS words, half tiles): quad 0 = [0] loader, [1] AGC done in place, [2] y ring produced, [3] frame consumed;
//                       quad 1 = [4] x ring holds (phase, magnitude), [5] post produced | generation << 24, [6] rewind: tile |
//                                generation << 24, [7] frame done;  quad 2 = [8] pair sums in the x ring
enum { C6_LD = 0, C6_AGC = 1, C6_Y = 2, C6_CONS = 3, C6_X = 4, C6_P4 = 5, C6_RW = 6, C6_DONE = 7, C6_IQ = 8 };

struct Blk6Z {
  uint32_t y_slots;              // half tiles in the y ring
  uint32_t zt_tiles;             // tiles of NCO phasors in flight (a power of two > kB6Stage + y_slots / 2)
  uint32_t rolemap;              // part of wave w = (rolemap >> 3 w) & 7
};
__host__ __device__ inline uint32_t blk6_zt_tiles(uint32_t y_slots) {
  uint32_t n = 8u;
  while (n < kB6Stage + y_slots / 2u + 2u) n *= 2u;
  return n;
}

// blk_fast (fsk_blk_dev.h) without its disc_post: P4 has evaluated the discriminator tail, the post filter and the slicer's
// operand; fa[0..1] = 0 - f of the tile's eight decimated samples, fa[2..3] = their magnitudes (the reference's scale).
// LEAN: every stream of the wave is inside a frame (thr_eff = kStartedP: no sync search, fsk.ts:297), so the correlator's running
// count cannot matter before a rare path is taken -- it is not carried (seven of the twelve vector instructions per decimated
// sample) and the caller re-forms it from the polyphase registers when it next needs it (matched = the sum over the registers
// of their masked match counts, by construction of the incremental update).
template <bool LEAN>
__device__ inline uint32_t blk6_fast(BackLane &Bn, const BackK &K, const BlkK &Q, uint32_t kv0, const v4f (&fa)[4],
                                     uint32_t (&rp)[kBlk], float (&am)[kBlk], uint32_t &bq, uint32_t &nq, uint32_t &hard_out) {
  // (floats first: __builtin_bit_cast applied to a vector ELEMENT expression read element 0 for every component -- hipcc 7.2)
  const float nff[kBlk] = {fa[0].x, fa[0].y, fa[0].z, fa[0].w, fa[1].x, fa[1].y, fa[1].z, fa[1].w};
  uint32_t nfv[kBlk];
#pragma unroll
  for (int j = 0; j < kBlk; j++) nfv[j] = __builtin_bit_cast(uint32_t, nff[j]);
  am[0] = fa[2].x; am[1] = fa[2].y; am[2] = fa[2].z; am[3] = fa[2].w;
  am[4] = fa[3].x; am[5] = fa[3].y; am[6] = fa[3].z; am[7] = fa[3].w;
  uint32_t w = 0, hard = 0;
  uint32_t dm = Bn.matched - Bn.thr_eff;
  const uint32_t lsr0 = Bn.ls - kv0;
  uint32_t lsr = lsr0;
#pragma unroll
  for (int j = 0; j < kBlk; j++) {
    const uint32_t nf = nfv[j];
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
    w = __builtin_amdgcn_alignbit(w, nf, 31);
  }
  if (!LEAN) Bn.matched = dm + Bn.thr_eff;
  Bn.ls = lsr + kv0;
  const uint32_t soft = K.eod_m1 - (uint32_t)kBlk + lsr0;                    // 'eod' bound, as blk_fast
  uint32_t md;
  hard |= blk_clock(Bn, K, Q, kv0, w, bq, nq, md);
  hard_out = hard;                                                           // a sync candidate or a bad start / stop bit: only the per-sample path takes those
  return hard | soft;
}

// The block path that takes resets (blk_medium, fsk_blk.hip), for the frame wave's own-span tiles: the same per-sample
// function (med_sample, fsk_blk_dev.h) on this kernel's ring layouts, the tile's inputs read up front (the wave has 256
// registers here) and the entry state kept as a register copy by the caller instead of a stash in memory.  xa = the x ring
// tile (I or phase 0..3, 4..7; Q or magnitude 0..3, 4..7), ya = its four y ring entries, zc / zs = the tile's cos / sin rows,
// ro = the eight polyphase registers.
// UNI = false (round 6: per-stream tone pairs, BASELINE config #4's kind): every lane has its own NCO, so the direct instance's
// phasors and lastPhase after a reset are evaluated here per lane, as fsk_blk.hip's blk_medium<false> does (nco_phasor of the
// lane's free-running accumulator; back_reset's expression); k0 = decimated samples of the launch before the tile.
template <bool UNI>
__device__ inline uint32_t blk6_medium(BackLane &Bn, const BackK &K, const BlkK &Q, uint32_t matched_min, uint32_t kv0, const v4f (&xa)[4],
                                       const v4f (&ya)[4], const v4f (&zc)[4], const v4f (&zs)[4], const uint32_t (&ro)[kBlk],
                                       const float (&thf8)[kBlk], float (&am)[kBlk], uint32_t &bq, uint32_t &nq, MedEv &E, uint32_t &w_out,
                                       uint64_t free0 = 0, uint64_t inc = 0, uint32_t k0 = 0) {
  uint32_t w = 0, hard = 0;
  uint32_t matched = Bn.matched, thr_cur = Bn.thr_eff, ls = Bn.ls;
  E.jr = 0; E.jc = 0; E.cai = E.caq = E.cbi = E.cbq = 0.f;
  const float e0v[kBlk] = {xa[0].x, xa[0].y, xa[0].z, xa[0].w, xa[1].x, xa[1].y, xa[1].z, xa[1].w};
  const float e1v[kBlk] = {xa[2].x, xa[2].y, xa[2].z, xa[2].w, xa[3].x, xa[3].y, xa[3].z, xa[3].w};
  const float yv[2 * kBlk] = {ya[0].x, ya[0].y, ya[0].z, ya[0].w, ya[1].x, ya[1].y, ya[1].z, ya[1].w,
                              ya[2].x, ya[2].y, ya[2].z, ya[2].w, ya[3].x, ya[3].y, ya[3].z, ya[3].w};
  const float cv[2 * kBlk] = {zc[0].x, zc[0].y, zc[0].z, zc[0].w, zc[1].x, zc[1].y, zc[1].z, zc[1].w,
                              zc[2].x, zc[2].y, zc[2].z, zc[2].w, zc[3].x, zc[3].y, zc[3].z, zc[3].w};
  const float sv[2 * kBlk] = {zs[0].x, zs[0].y, zs[0].z, zs[0].w, zs[1].x, zs[1].y, zs[1].z, zs[1].w,
                              zs[2].x, zs[2].y, zs[2].z, zs[2].w, zs[3].x, zs[3].y, zs[3].z, zs[3].w};
#pragma unroll
  for (int j = 0; j < kBlk; j++) {
    v4f z = (v4f){cv[2 * j], sv[2 * j], cv[2 * j + 1], sv[2 * j + 1]};
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
    med_sample(Bn, K, j, e0v[j], e1v[j], yv[2 * j], yv[2 * j + 1], z, thf_j, ro[j], kv0 + (uint32_t)(j + 1), matched_min, kHandPairs, matched, thr_cur, ls,
               w, hard, E, am[j]);
  }
  hard |= med_finish(Bn, K, Q, kv0, matched, thr_cur, ls, w, bq, nq, E);
  w_out = w;
  return hard;
}

// one branch of the free-running I/Q low-pass (front_mix_lp's instruction sequence for one of its two chains)
struct LpLane { float x1, x2, y, v; };
__device__ __forceinline__ float lp_step(LpLane &L, float a2, float nd, float v, float z) {
  const float m = v * z;
  const float t = __builtin_fmaf(2.0f, L.x1, m) + L.x2;
  L.v = __builtin_fmaf(a2, L.v, __builtin_fmaf(nd, L.y, t));
  L.y += L.v;
  L.x2 = L.x1; L.x1 = m;
  return L.y;
}

// (front_agc / front_bp -- front_agc_bp as its two halves -- live in fsk_pipe_dev.h since round 6: fsk_blk.hip's five-wave kernel uses them too)

// Every hand-off wait of this kernel is a bounded poll (FSK_SPIN, fsk_pipe_dev.h): a wave that gets nowhere in FSK_SPIN_CAP polls
// of one wait flags the engine's hand-off fault word and ends, the waves waiting on it follow, the host reports FSKHIP_E_INTERNAL.
#define B6_SPIN(arg) FSK_SPIN(arg, S.blk_stat)
#ifndef FSK_B6_SLEEP
#define FSK_B6_SLEEP 1
#endif
#ifndef FSK_B6_LEAN
#define FSK_B6_LEAN 1
#endif
// the frame wave's own-span tiles on the block path that takes resets (blk6_medium) instead of sample by sample
#ifndef FSK_B6_MEDIUM
#define FSK_B6_MEDIUM 1
#endif
// Hand-off counters.  FSK_B6_POSTWAIT = 1: a wave waits for its own LDS writes (lgkmcnt(0)) before it writes the counter that
// publishes them, as fsk_blk.hip does; 0: it does not -- the LDS executes a wave's instructions in order, so the counter's write
// is performed after the data's, and a reader that has seen the counter reads after both.  The wait costs the producer ~100
// cycles per hand-off; with only three tiles of lead allowed between the iq wave and the frame wave (kZeroLagPairs) every hand-off's
// latency is on the ring's critical path.
#ifndef FSK_B6_POSTWAIT
#define FSK_B6_POSTWAIT 1
#endif
__device__ inline void b6_post(uint32_t *p, uint32_t v) {
#if FSK_B6_POSTWAIT
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
#else
  asm volatile("ds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
#endif
}
// poll period of the three waves on the x ring (x 64 cycles; the waves before it keep FSK_B6_SLEEP)
#ifndef FSK_B6_SLEEP_RING
#define FSK_B6_SLEEP_RING 1
#endif

// UNI = false (round 6): per-stream tone pairs.  What changes: the pre-filter wave takes its coefficients per stream
// (front_load<false>), the iq wave rotates a per-lane phasor -- the tile's first from the exact accumulator, the other fifteen by
// e^{j w} (fsk_blk.hip's wave 1, op for op; in a narrow group both halves of the wave rotate the same phasor and use its real or
// imaginary part) -- the loader's phasor table is not written, and the frame wave's block path with resets evaluates the direct
// instance's phasors and lastPhase per lane (blk6_medium<false>).  Bytes, counters and state words are demod_blk_kernel<., false, .>'s.
template <bool WB, int LW, bool UNI = true>
__global__ __launch_bounds__(64 * kB6Waves, 1) void demod_blk6_kernel(
    DemodParams P, DemodState S, float *__restrict__ samples, size_t n_call, size_t pitch, int append,
    uint8_t *__restrict__ out, size_t out_pitch, uint32_t *__restrict__ out_counts,
    uint32_t *__restrict__ eod_counts, Blk6Z Z) {
  FSK_STAMP_DECL
  constexpr bool SPLIT = LW <= 32;                         // P2: I chain in lanes 0..31, Q chain in lanes 32..63
  constexpr int PARTS = 64 / LW;                           // P3: lanes per stream
  constexpr int SP = kBlk / PARTS;                         // ... decimated samples per lane and tile
  extern __shared__ float4 lds[];
  const uint32_t PS = blk_poly_stride(P.d);
  const uint32_t NY = Z.y_slots;
  v4f *stage = reinterpret_cast<v4f *>(lds);
  v4f *yring = stage + kB6Stage * 4 * kSlotStride;
  v4f *xring = yring + NY * 2 * 64;
  v4f *fring = xring + kB6XT * kB6TileV4;
  v4f *trash = fring + kB6XT * kB6TileV4;                  // where the lanes without a stream write instead (below)
  v4f *hist = trash + kB6TileV4;                           // [tile % XT][0] (px1, px2, py, pv), [1].x lastPhase: P4's state on entering the tile
  v4f *rmail = hist + kB6XT * 2 * 64;                      // P5 -> P4: the state to resume with ([1] = lastPhase, thf)
  v4f *fin = rmail + 2 * 64;                               // final states: [0] I, [1] Q low-pass, [2] correction, [3] post filter, [4].x lastPhase
  v4f *zt = fin + 5 * 64;
  const uint32_t ZTM = Z.zt_tiles - 1u;
  uint32_t *poly = reinterpret_cast<uint32_t *>(zt + Z.zt_tiles * 8);
  uint32_t *ctr = poly + 64u * PS;
  uint32_t *zmail = ctr + 16;
  uint32_t *cmail = zmail + 64;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t role = (Z.rolemap >> (3u * wave)) & 7u;
  FSK_WAIT_DECL
  {
    // every part played exactly once (the host validates "stage_roles" and builds the default map; a map that is no permutation of
    // 0..6 would leave some counter without a writer and six waves waiting on it for good): checked here once per launch by every
    // wave alike -- scalar work -- and a bad map ends the launch at once, flagged in the statistics word
    uint32_t seen = 0;
    for (uint32_t w = 0; w < kB6Waves; w++) seen |= 1u << ((Z.rolemap >> (3u * w)) & 7u);
    if (seen != 0x7Fu) {
      if (threadIdx.x == 0 && S.blk_stat) __hip_atomic_fetch_or(&S.blk_stat[2], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }

  const size_t n_tiles = n_call / kFastTile;
  const size_t n = n_tiles * kFastTile;
  const uint32_t nt = (uint32_t)n_tiles;
  const uint32_t nh = 2u * nt;                             // half tiles (the counters' unit, as in fsk_blk.hip)
  constexpr uint32_t W = (uint32_t)LW;
  const uint32_t grp = blockIdx.x, s0 = grp * W;
  // which stream a lane works for: P2 (split) lanes 32.. mirror lanes 0.., P3 lanes l + W * part
  const uint32_t l = role == 6u ? lane % W : (SPLIT && role == 3u) ? (lane & 31u) : lane;
  const bool mine = l < W;
  const uint32_t stream = mine ? s0 + l : 0xFFFFFFFFu;
  const PipeCtx C = pipe_ctx(P, S, stream);
  const uint64_t inc = UNI ? (((uint64_t)P.u_inc_hi << 32) | P.u_inc_lo) : S.nco_inc[C.row4 >> 2];
  const uint64_t free0 = pipe_free0<UNI, 0>(C);
  uint32_t *gpoly = (uint32_t *)S.poly + (size_t)(s0 >> 6) * P.d * 64u + (s0 & 63u);
  constexpr int COH = 0;

  if (threadIdx.x < 16) ctr[threadIdx.x] = 0;
  // Lanes without a stream (beyond a narrow group's W streams, beyond the batch's last stream).  The x and f rings start as
  // zeros and such lanes write to `trash` instead, so what P4 and P5 read for them is exactly zero for good: their slicer bits
  // are 0, their polyphase registers and correlator count never move, and they never take their wave off its block path
  // (fsk_blk.hip parks them on zero input rows; here a narrow group has up to 56 of them and they must not ring).
  for (uint32_t i = threadIdx.x; i < 2u * kB6XT * kB6TileV4; i += 64u * kB6Waves) xring[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  const bool live = C.valid;                                // (a lane's WRITES to the x and f rings go to the tile if live, else to `trash`)
  if (role == 5u) {
    const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    const uint32_t dph = PIPE_ILOAD(zr_dph);
    zmail[lane] = zmail_init(dph);
    // a correction still on its un-retired span when the launch starts (fsk_blk.hip: the same mailboxes)
    uint32_t kq = 0xFFFFFFFFu;
    float ai = 0.f, aq = 0.f, bi = 0.f, bq = 0.f;
    if (dph >= kDirectPairs && dph < kHandPairs) {
      ai = PIPE_RLOAD(zq_ai); aq = PIPE_RLOAD(zq_aq); bi = PIPE_RLOAD(zq_bi); bq = PIPE_RLOAD(zq_bq);
      const float c1 = P.z_c1, c2 = P.z_c2;
      for (uint32_t g = dph; g < kHandPairs; g++) {
        const float ni = __builtin_fmaf(c1, bi, -(c2 * ai)), nq = __builtin_fmaf(c1, bq, -(c2 * aq));
        ai = bi; aq = bq; bi = ni; bq = nq;
      }
      kq = kHandPairs - dph;
    }
    cmail[64u + lane] = __builtin_bit_cast(uint32_t, ai); cmail[128u + lane] = __builtin_bit_cast(uint32_t, aq);
    cmail[192u + lane] = __builtin_bit_cast(uint32_t, bi); cmail[256u + lane] = __builtin_bit_cast(uint32_t, bq);
    cmail[lane] = kq;
    cmail[320u + lane] = 0u - dph;
  }
  __syncthreads();

  if (role == 0u) {
    // ---------------------------------------------------------------------------------------------- loader
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
#define B6_BLOAD4(dst, rows16, soff)                                                                        \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(in_voff + (rows16) * in_row16), \
               "s"(in_rsrc), "s"(soff) : "memory")
    auto load_tile = [&](size_t t, v4f &a, v4f &b, v4f &c, v4f &d) {
      const uint32_t tn = (uint32_t)((t < n_tiles ? t : n_tiles - 1) * kFastTile * 4u);
      B6_BLOAD4(a, 0u, tn); B6_BLOAD4(b, 1u, tn); B6_BLOAD4(c, 2u, tn); B6_BLOAD4(d, 3u, tn);
    };
    v4f a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the state loads above are complete, the count starts clean
    load_tile(0, a0, a1, a2, a3);
    load_tile(1, b0, b1, b2, b3);
    load_tile(2, c0, c1, c2, c3);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
    uint32_t used = 0;                                      // half tiles the pre-filter wave has taken out of the staging ring
    uint32_t sidx = 0;
    uint64_t zacc = free0 + inc * (uint64_t)(lane & 15u);
    const uint64_t inc16 = inc * 16u;
    auto do_tile = [&](uint32_t t, v4f &r0, v4f &r1, v4f &r2, v4f &r3) {
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
      if (2u * t + 2u - used > 2u * kB6Stage) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (2u * t + 2u - used > 2u * kB6Stage) {
          used = lds_peek(&ctr[C6_Y]);
          if (2u * t + 2u - used > 2u * kB6Stage) B6_SPIN(FSK_B6_SLEEP);
        }
        FSK_STAMP_W1
      }
      v4f *st = stage + sidx * 4u * kSlotStride;
      sidx = sidx + 1u == kB6Stage ? 0u : sidx + 1u;
      st[st_slot] = r0; st[st_slot + 16] = r1; st[st_slot + 32] = r2; st[st_slot + 48] = r3;
      load_tile((size_t)t + 3, r0, r1, r2, r3);
      v4u32 cv;
      lds_peek4_begin(ctr, cv);
      if (UNI) {
        float pc, ps;
        nco_phasor(zacc, pc, ps);
        float *ztf = reinterpret_cast<float *>(zt + (t & ZTM) * 8u);
        ztf[lane & 15u] = pc; ztf[16u + (lane & 15u)] = ps;
        zacc += inc16;
      }
      b6_post(&ctr[C6_LD], 2u * t + 2u);
      used = lds_peek4_get(cv, C6_Y);
    };
    FSK_STAMP_BEGIN
    for (uint32_t t = 0; t < nt; t += 3) {
      do_tile(t, a0, a1, a2, a3);
      if (t + 1 < nt) do_tile(t + 1, b0, b1, b2, b3);
      if (t + 2 < nt) do_tile(t + 2, c0, c1, c2, c3);
    }
    FSK_STAMP_END(0)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
#undef B6_BLOAD4
  } else if (role == 1u || role == 2u) {
    // ---------------------------------------------------------------------------------------------- AGC, pre-filter
    // The AGC wave (a recurrence of ten dependent instructions per sample, the longest chain left) writes the scaled samples
    // back IN PLACE into the staging tile, the pre-filter wave takes them from there.  The two halves are front_agc_bp's
    // instruction sequences (fsk_pipe_dev.h).
    FrontLane F;
    FrontK K;
    front_load<UNI, 0>(F, K, P, S, C);
    // (compile-time variants: a run-time choice inside the sample loop compiled to a branch per sample)
    auto front_stage = [&](auto agc_tag, auto bp_tag) {
    constexpr bool do_agc = decltype(agc_tag)::value, do_bp = decltype(bp_tag)::value;
    constexpr uint32_t c_in = do_agc ? C6_LD : C6_AGC;
    uint32_t produced = 0, consumed = 0, slot_i = 0, sidx = 0;
    FSK_STAMP_BEGIN
    for (uint32_t t = 0; t < nt; t++) {
      const uint32_t hidx = 2u * t;
      if (produced < hidx + 2u || (do_bp && hidx + 1u - consumed >= NY)) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced < hidx + 2u) {
          produced = lds_peek(&ctr[c_in]);
          if (produced < hidx + 2u) B6_SPIN(FSK_B6_SLEEP);
        }
        while (do_bp && hidx + 1u - consumed >= NY) {         // y ring full: the frame wave (which may still need the slots'
          consumed = lds_peek(&ctr[C6_CONS]);                 // pre-filter outputs after a reset) has not released them
          if (hidx + 1u - consumed >= NY) B6_SPIN(FSK_B6_SLEEP);
        }
        FSK_STAMP_W1
      }
      v4u32 cv;
      lds_peek4_begin(ctr, cv);
      v4f *st = stage + sidx * 4u * kSlotStride;
      sidx = sidx + 1u == kB6Stage ? 0u : sidx + 1u;
      v4f x4[4];
#pragma unroll
      for (uint32_t c = 0; c < 4; c++) x4[c] = st[c * kSlotStride + lane];
#pragma unroll
      for (uint32_t hf = 0; hf < 2; hf++) {
        v4f *slot = yring + slot_i * 2u * 64u;
        if (do_bp) slot_i = slot_i + 1u == NY ? 0u : slot_i + 1u;
#pragma unroll
        for (uint32_t cc = 0; cc < 2; cc++) {
          const uint32_t c = 2u * hf + cc;
          const float xin[4] = {x4[c].x, x4[c].y, x4[c].z, x4[c].w};
          float xs[4], y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            xs[j] = do_agc ? front_agc(F, K, xin[j]) : xin[j];
            if (do_bp) y[j] = front_bp(F, K, xs[j]);
          }
          if (do_bp) slot[cc * 64u + lane] = (v4f){y[0], y[1], y[2], y[3]};
          else st[c * kSlotStride + lane] = (v4f){xs[0], xs[1], xs[2], xs[3]};
          if (WB && do_agc) {
            if (C.valid)
              *reinterpret_cast<v4f *>(samples + (size_t)(C.row4 >> 2) * pitch + (size_t)t * kFastTile + 4u * c) = (v4f){xs[0], xs[1], xs[2], xs[3]};
          }
        }
      }
      b6_post(&ctr[do_bp ? C6_Y : C6_AGC], hidx + 2u);       // (the pre-filter's post also frees this tile of the staging ring)
      produced = lds_peek4_get(cv, do_agc ? C6_LD : C6_AGC); consumed = lds_peek4_get(cv, C6_CONS);
    }
    FSK_STAMP_END(do_agc ? 1 : 2)
    {
      const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
      const FastMem &M = C.M;
      const uint32_t fld = C.fld;
      if (do_agc) PIPE_RSTORE(agc_gain, F.g);
      if (do_bp) { PIPE_RSTORE(bp_x1, F.bx1); PIPE_RSTORE(bp_x2, F.bx2); PIPE_RSTORE(bp_y1, F.by1); PIPE_RSTORE(bp_y2, F.by2); }
    }
    };
    if (role == 1u) front_stage(std::true_type(), std::false_type());
    else front_stage(std::false_type(), std::true_type());
  } else if (role == 3u) {
    // ---------------------------------------------------------------------------------------------- mixer, I/Q low-pass, pair sums
    const bool upper = SPLIT && lane >= 32u;
    LpLane LI, LQ;                                           // SPLIT: LI is this lane's only chain (I in lanes 0..31, Q in 32..63)
    {
      const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
      const uint32_t fld = C.fld, row4 = C.row4;
      const float ix1 = PIPE_RLOAD(li_x1), ix2 = PIPE_RLOAD(li_x2), iy = PIPE_RLOAD(li_y1), iv = PIPE_RLOAD(li_y2);
      const float qx1 = PIPE_RLOAD(lq_x1), qx2 = PIPE_RLOAD(lq_x2), qy = PIPE_RLOAD(lq_y1), qv = PIPE_RLOAD(lq_y2);
      if (SPLIT) { LI.x1 = upper ? qx1 : ix1; LI.x2 = upper ? qx2 : ix2; LI.y = upper ? qy : iy; LI.v = upper ? qv : iv; LQ = LI; }
      else { LI.x1 = ix1; LI.x2 = ix2; LI.y = iy; LI.v = iv; LQ.x1 = qx1; LQ.x2 = qx2; LQ.y = qy; LQ.v = qv; }
    }
    float lp_a2 = P.f_lp_a2, lp_nd = -P.f_lp_delta;
    asm volatile("" : "+v"(lp_a2), "+v"(lp_nd));
    float wre = 1.f, wim = 0.f, zr = 1.f, zi = 0.f;          // UNI = false: e^{j w} of this lane's stream, and the running phasor
    uint64_t tacc = free0;
    const uint64_t inc16p = inc * 16u;
    if (!UNI) {
      const __amdgpu_buffer_rsrc_t cf_rsrc = C.cf_rsrc;
      const uint32_t fld = C.fld, row4 = C.row4;
      wre = (float)PIPE_CLOAD(CF_w1_re); wim = (float)PIPE_CLOAD(CF_w1_im);
    }
    uint32_t consumed = 0, produced = 0, yslot_i = 0, xt_i = 0;
    const uint32_t zoff = upper ? 4u : 0u;                   // v4f offset of this lane's phasor row in a zt tile (cos | sin)
    const uint32_t xoff = upper ? 2u * 64u : 0u;             // ... and of its rows in an x ring tile
    FSK_STAMP_BEGIN
    for (uint32_t t = 0; t < nt; t++) {
      const uint32_t hidx = 2u * t;
      if (produced < hidx + 2u || hidx + 2u - consumed > 2u * kB6XT) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced < hidx + 2u) {
          produced = lds_peek(&ctr[C6_Y]);
          if (produced < hidx + 2u) B6_SPIN(FSK_B6_SLEEP_RING);
        }
        while (hidx + 2u - consumed > 2u * kB6XT) {          // x ring full: wait for the frame wave
          consumed = lds_peek(&ctr[C6_CONS]);
          if (hidx + 2u - consumed > 2u * kB6XT) B6_SPIN(FSK_B6_SLEEP_RING);
        }
        FSK_STAMP_W1
      }
      v4u32 cv;
      lds_peek4_begin(ctr, cv);
      const v4f *ztile = zt + (t & ZTM) * 8u;
      const uint32_t zj = zmail[l];
      v4f y4[4], zz[8];
      if (!UNI) {                                             // the tile's first phasor from the exact accumulator
        nco_phasor(tacc, zr, zi);
        tacc += inc16p;
      }
      {
        const v4f *ys0 = yring + yslot_i * 2u * 64u;
        yslot_i = yslot_i + 1u == NY ? 0u : yslot_i + 1u;
        const v4f *ys1 = yring + yslot_i * 2u * 64u;
        yslot_i = yslot_i + 1u == NY ? 0u : yslot_i + 1u;
        y4[0] = ys0[l]; y4[1] = ys0[64u + l]; y4[2] = ys1[l]; y4[3] = ys1[64u + l];
        if (!UNI) {
#pragma unroll
          for (int i = 0; i < 8; i++) zz[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        } else if (SPLIT) {
#pragma unroll
          for (int i = 0; i < 4; i++) zz[i] = ztile[zoff + (uint32_t)i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) zz[i] = ztile[i];
        }
      }
      v4f *xt = xring + xt_i * kB6TileV4;
      xt_i = xt_i + 1u == kB6XT ? 0u : xt_i + 1u;
      float si[8], sq[8];                                    // pair sums U of the tile's eight decimated samples
      auto quad = [&](const uint32_t c, const bool zeroing) {
        const float y[4] = {y4[c].x, y4[c].y, y4[c].z, y4[c].w};
        float za[4] = {zz[c].x, zz[c].y, zz[c].z, zz[c].w};                             // SPLIT: this lane's row; else cos
        float zb[4] = {zz[(c + 4u) & 7u].x, zz[(c + 4u) & 7u].y, zz[(c + 4u) & 7u].z, zz[(c + 4u) & 7u].w};   // sin (not SPLIT)
        if (!UNI) {
#pragma unroll
          for (int j = 0; j < 4; j++) {                       // (fsk_blk.hip wave 1's rotation, op for op)
            za[j] = (SPLIT && upper) ? zi : zr; zb[j] = zi;
            const float nr = __builtin_fmaf(-zi, wim, zr * wre), ni = __builtin_fmaf(zi, wre, zr * wim);
            zr = nr; zi = ni;
          }
        }
        const uint32_t pb = 8u * t + 2u * c;
        float oi[4], oq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (zeroing && !(j & 1)) {
            if (zj == pb + (uint32_t)(j >> 1)) {
              LI.x1 = LI.x2 = LI.y = LI.v = 0.f;
              if (!SPLIT) LQ.x1 = LQ.x2 = LQ.y = LQ.v = 0.f;
            }
          }
          oi[j] = lp_step(LI, lp_a2, lp_nd, y[j], za[j]);
          if (!SPLIT) oq[j] = lp_step(LQ, lp_a2, lp_nd, y[j], zb[j]);
        }
        si[2 * c] = oi[0] + oi[1]; si[2 * c + 1] = oi[2] + oi[3];
        if (!SPLIT) { sq[2 * c] = oq[0] + oq[1]; sq[2 * c + 1] = oq[2] + oq[3]; }
      };
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(zj - 8u * t < 8u) != 0ull, 0)) {
        asm volatile("s_nop 0");
        quad(0, true); quad(1, true); quad(2, true); quad(3, true);
      } else {
        quad(0, false); quad(1, false); quad(2, false); quad(3, false);
      }
      v4f *xw = live ? xt : trash;
      xw[xoff + l] = (v4f){si[0], si[1], si[2], si[3]};
      xw[xoff + 64u + l] = (v4f){si[4], si[5], si[6], si[7]};
      if (!SPLIT) {
        xw[128u + l] = (v4f){sq[0], sq[1], sq[2], sq[3]};
        xw[192u + l] = (v4f){sq[4], sq[5], sq[6], sq[7]};
      }
      b6_post(&ctr[C6_IQ], hidx + 2u);
      produced = lds_peek4_get(cv, C6_Y); consumed = lds_peek4_get(cv, C6_CONS);
    }
    FSK_STAMP_END(3)
    if (SPLIT) fin[(upper ? 64u : 0u) + l] = (v4f){LI.x1, LI.x2, LI.y, LI.v};
    else { fin[l] = (v4f){LI.x1, LI.x2, LI.y, LI.v}; fin[64u + l] = (v4f){LQ.x1, LQ.x2, LQ.y, LQ.v}; }
    b6_post(&ctr[C6_IQ], nh + 1u);
  } else if (role == 6u) {
    // ---------------------------------------------------------------------------------------------- ZIR correction + discriminator
    const __amdgpu_buffer_rsrc_t rs_rsrc = C.rs_rsrc;
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    const uint32_t part = lane / W;
    const bool first = part == 0u;                           // the lane that carries the stream's correction
    QLane Qz = {0.f, 0.f, 0.f, 0.f};
    if (first && PIPE_ILOAD(zr_dph) >= kHandPairs) {        // this wave's from the first sample on
      Qz.ai = PIPE_RLOAD(zq_ai); Qz.aq = PIPE_RLOAD(zq_aq); Qz.bi = PIPE_RLOAD(zq_bi); Qz.bq = PIPE_RLOAD(zq_bq);
    }
    float c1 = P.z_c1, c2 = P.z_c2, tiny = 0x1p-123f, rel = 3.7252902984619141e-09f;
    uint32_t sgn = 0x80000000u;
    asm volatile("" : "+v"(c1), "+v"(c2), "+v"(tiny), "+v"(rel), "+v"(sgn));
    uint64_t qlive = __builtin_amdgcn_ballot_w64(first & ((Qz.ai != 0.f) | (Qz.aq != 0.f) | (Qz.bi != 0.f) | (Qz.bq != 0.f)));
    uint32_t produced = 0, consumed = 0, xt_i = 0;
    uint32_t *ctr2 = ctr + 8;
    FSK_STAMP_BEGIN
    for (uint32_t t = 0; t < nt; t++) {
      const uint32_t hidx = 2u * t;
      if (produced < hidx + 2u || hidx + 2u - consumed > 2u * kB6DT) {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        while (produced < hidx + 2u) {
          produced = lds_peek(&ctr[C6_IQ]);
          if (produced < hidx + 2u) B6_SPIN(FSK_B6_SLEEP_RING);
        }
        while (hidx + 2u - consumed > 2u * kB6DT) {          // a correction posted now is due kHandLag ahead: no further than that
          consumed = lds_peek(&ctr[C6_CONS]);
          if (hidx + 2u - consumed > 2u * kB6DT) B6_SPIN(FSK_B6_SLEEP_RING);
        }
        FSK_STAMP_W1
      }
      v4u32 cv, cv0;
      lds_peek4_begin(ctr2, cv);
      lds_peek4_begin(ctr, cv0);
      v4f *xt = xring + xt_i * kB6TileV4;
      xt_i = xt_i + 1u == kB6XT ? 0u : xt_i + 1u;
      const uint32_t kq = cmail[l];
      const uint32_t ow = 4u * hidx - cmail[320u + l];        // decimated samples since the frame wave's own span began
      asm volatile("" ::: "memory");
      if (__builtin_expect((__builtin_amdgcn_ballot_w64((kq - 4u * hidx < 8u) | (ow < kHandPairs)) | qlive) != 0ull, 0)) {
        // a hand-over due in this tile, a lane inside the frame wave's own span, a live correction: the stream's first lane
        // takes the whole tile, in order (fsk_blk.hip's discriminator wave, op for op)
        if (first) {
          const v4f ua = xt[l], ub = xt[64u + l], uc = xt[128u + l], ud = xt[192u + l];
          const float ui[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w}, uq[8] = {uc.x, uc.y, uc.z, uc.w, ud.x, ud.y, ud.z, ud.w};
          float ph[8], am[8];
          QLane H = {0.f, 0.f, 0.f, 0.f};
          if (kq - 4u * hidx < 8u) {
            H.ai = __builtin_bit_cast(float, cmail[64u + l]); H.aq = __builtin_bit_cast(float, cmail[128u + l]);
            H.bi = __builtin_bit_cast(float, cmail[192u + l]); H.bq = __builtin_bit_cast(float, cmail[256u + l]);
            const uint32_t steps = kq > kHandLag ? kHandLag : 0u;   // (posted inside this launch: kHandLag steps before its sample)
            for (uint32_t g = 0; g < steps; g++) {
              const float ni = __builtin_fmaf(c1, H.bi, -(c2 * H.ai)), nq = __builtin_fmaf(c1, H.bq, -(c2 * H.aq));
              H.ai = H.bi; H.aq = H.bq; H.bi = ni; H.bq = nq;
            }
          }
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (kq == 4u * hidx + (uint32_t)j) Qz = H;        // the frame wave's correction becomes this wave's here
            const float wi = ui[j] - Qz.ai, wq = uq[j] - Qz.aq;
            {
              const float ni = __builtin_fmaf(c1, Qz.bi, -(c2 * Qz.ai)), nq = __builtin_fmaf(c1, Qz.bq, -(c2 * Qz.aq));
              Qz.ai = Qz.bi; Qz.aq = Qz.bq; Qz.bi = ni; Qz.bq = nq;
            }
            ph[j] = atan2_amp_fma(wq, wi, am[j], tiny, sgn);
            const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(Qz.ai), __builtin_fabsf(Qz.aq)),
                                              __builtin_fmaxf(__builtin_fabsf(Qz.bi), __builtin_fabsf(Qz.bq)));
            if (!(big > am[j] * rel)) { Qz.ai = 0.f; Qz.aq = 0.f; Qz.bi = 0.f; Qz.bq = 0.f; }
            const bool own = ow + (uint32_t)j < kHandPairs;   // the frame wave evaluates these itself and needs the pair sums
            ph[j] = own ? ui[j] : ph[j]; am[j] = own ? uq[j] : am[j];
          }
          v4f *xw = live ? xt : trash;
          xw[l] = (v4f){ph[0], ph[1], ph[2], ph[3]}; xw[64u + l] = (v4f){ph[4], ph[5], ph[6], ph[7]};
          xw[128u + l] = (v4f){am[0], am[1], am[2], am[3]}; xw[192u + l] = (v4f){am[4], am[5], am[6], am[7]};
        }
        qlive = __builtin_amdgcn_ballot_w64(first & ((Qz.ai != 0.f) | (Qz.aq != 0.f) | (Qz.bi != 0.f) | (Qz.bq != 0.f)));
      } else {
        // the plain discriminator is stateless: SP decimated samples per lane, PARTS lanes per stream
        float *xf = reinterpret_cast<float *>(xt);
        const uint32_t fi = ((part * (uint32_t)SP) >> 2) * 256u + l * 4u + ((part * (uint32_t)SP) & 3u);   // float index of this lane's first I value
        float ui[SP], uq[SP], ph[SP], am[SP];
        if (SP == 8) {
          const v4f ua = xt[l], ub = xt[64u + l], uc = xt[128u + l], ud = xt[192u + l];
          const float ti[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w}, tq[8] = {uc.x, uc.y, uc.z, uc.w, ud.x, ud.y, ud.z, ud.w};
#pragma unroll
          for (int j = 0; j < SP; j++) { ui[j] = ti[j & 7]; uq[j] = tq[j & 7]; }
        } else if (SP == 4) {
          const v4f ua = *reinterpret_cast<const v4f *>(xf + fi), uc = *reinterpret_cast<const v4f *>(xf + fi + 512u);
          const float ti[4] = {ua.x, ua.y, ua.z, ua.w}, tq[4] = {uc.x, uc.y, uc.z, uc.w};
#pragma unroll
          for (int j = 0; j < SP; j++) { ui[j] = ti[j & 3]; uq[j] = tq[j & 3]; }
        } else if (SP == 2) {
          const f2 ua = *reinterpret_cast<const f2 *>(xf + fi), uc = *reinterpret_cast<const f2 *>(xf + fi + 512u);
          ui[0] = ua.x; ui[SP - 1] = ua.y; uq[0] = uc.x; uq[SP - 1] = uc.y;
        } else {
          ui[0] = xf[fi]; uq[0] = xf[fi + 512u];
        }
#pragma unroll
        for (int j = 0; j < SP; j++) ph[j] = atan2_amp_fma(uq[j], ui[j], am[j], tiny, sgn);
        float *xfw = live ? xf : reinterpret_cast<float *>(trash);
        if (SP == 8) {
          v4f *xw = live ? xt : trash;
          xw[l] = (v4f){ph[0], ph[1 % SP], ph[2 % SP], ph[3 % SP]}; xw[64u + l] = (v4f){ph[4 % SP], ph[5 % SP], ph[6 % SP], ph[7 % SP]};
          xw[128u + l] = (v4f){am[0], am[1 % SP], am[2 % SP], am[3 % SP]}; xw[192u + l] = (v4f){am[4 % SP], am[5 % SP], am[6 % SP], am[7 % SP]};
        } else if (SP == 4) {
          *reinterpret_cast<v4f *>(xfw + fi) = (v4f){ph[0], ph[1 % SP], ph[2 % SP], ph[3 % SP]};
          *reinterpret_cast<v4f *>(xfw + fi + 512u) = (v4f){am[0], am[1 % SP], am[2 % SP], am[3 % SP]};
        } else if (SP == 2) {
          *reinterpret_cast<f2 *>(xfw + fi) = (f2){ph[0], ph[SP - 1]};
          *reinterpret_cast<f2 *>(xfw + fi + 512u) = (f2){am[0], am[SP - 1]};
        } else {
          xfw[fi] = ph[0]; xfw[fi + 512u] = am[0];
        }
      }
      b6_post(&ctr[C6_X], hidx + 2u);
      produced = lds_peek4_get(cv, C6_IQ - 8); consumed = lds_peek4_get(cv0, C6_CONS);
    }
    FSK_STAMP_END(6)
    if (first) fin[128u + l] = (v4f){Qz.ai, Qz.aq, Qz.bi, Qz.bq};
    b6_post(&ctr[C6_X], nh + 1u);
  } else if (role == 4u) {
    // ---------------------------------------------------------------------------------------------- discriminator tail, post filter, slicer -- ahead of the frame logic
    BackLane B;
    BackK K;
    back_load<UNI, 0>(B, K, P, S, C, 0xFFFFFFFFu, nullptr, nullptr, 0);   // (the post filter, lastPhase and thf are what this wave uses of it)
    if (!C.valid) { B.px1 = B.px2 = B.py = B.pv = 0.f; B.last_phase = 0.f; B.thf = 0.f; }   // (zeros in, zeros out: see `trash`)
    uint32_t gen = 0, t = 0, xt_i = 0;
    uint32_t rw_cur = 0;                                      // the rewind word this wave has adopted (tile | generation << 24)
    uint32_t *ctr1 = ctr + 4;
    // One tile: `cur` holds its x ring entries (phase 0..3, 4..7, magnitude 0..3, 4..7); the NEXT tile's are read into `nxt`
    // while this one is worked -- whether they were there yet is known from the counters read just before them (LDS answers a
    // wave in order).  One branch per tile: the loop's.  Returns false when the next tile cannot follow at once.
    auto tile = [&](const v4f (&cur)[4], v4f (&nxt)[4]) -> bool {
      v4u32 cv;
      lds_peek4_begin(ctr1, cv);
      v4f *ft = live ? fring + xt_i * kB6TileV4 : trash;
      v4f *ht = hist + xt_i * 2u * 64u;
      xt_i = xt_i + 1u == kB6XT ? 0u : xt_i + 1u;
      {
        const v4f *xn = xring + xt_i * kB6TileV4;
        nxt[0] = xn[lane]; nxt[1] = xn[64u + lane]; nxt[2] = xn[128u + lane]; nxt[3] = xn[192u + lane];
      }
      ht[lane] = (v4f){B.px1, B.px2, B.py, B.pv};
      ht[64u + lane] = (v4f){B.last_phase, 0.f, 0.f, 0.f};
      const float phs[kBlk] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
      float am[kBlk] = {cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w};
      float nf[kBlk];
#pragma unroll
      for (int j = 0; j < kBlk; j++) {
        const float f = disc_post(B, K, phs[j], am[j]);                        // fsk.ts:251-261
        nf[j] = slicer_nf(f);                                                   // slicer (fsk.ts:264): the bit is this value's sign
      }
      ft[lane] = (v4f){nf[0], nf[1], nf[2], nf[3]}; ft[64u + lane] = (v4f){nf[4], nf[5], nf[6], nf[7]};
      ft[128u + lane] = (v4f){am[0], am[1], am[2], am[3]}; ft[192u + lane] = (v4f){am[4], am[5], am[6], am[7]};
      t++;
      b6_post(&ctr[C6_P4], (gen << 24) | (2u * t));
      const uint32_t produced = lds_peek4_get(cv, 0), rw = lds_peek4_get(cv, 2), done = lds_peek4_get(cv, 3);
      return (rw == rw_cur) & (done == 0u) & (t < nt) & (produced >= 2u * t + 2u);
    };
    FSK_STAMP_BEGIN
    for (;;) {
      // ---- until tile t can be worked, the frame wave rewinds this one, or the launch is over
      bool over = false;
      {
        FSK_STAMP_W0 FSK_WAIT_BEGIN
        for (;;) {
          v4u32 cv;
          lds_peek4_begin(ctr1, cv);
          const uint32_t produced = lds_peek4_get(cv, 0), rw = lds_peek4_get(cv, 2), done = lds_peek4_get(cv, 3);
          if (done != 0u) { over = true; break; }
          if (rw != rw_cur) {
            // the frame wave took some tiles sample by sample and has posted what it ended with: drop what ran ahead, resume there
            rw_cur = rw; gen = rw >> 24; t = rw & 0xFFFFFFu;
            const v4f a = rmail[lane], b = rmail[64u + lane];
            B.px1 = a.x; B.px2 = a.y; B.py = a.z; B.pv = a.w; B.last_phase = b.x; B.thf = b.y;
            continue;
          }
          if (t < nt && produced >= 2u * t + 2u) break;
          B6_SPIN(FSK_B6_SLEEP_RING);
        }
        FSK_STAMP_W1
      }
      if (over) break;
      xt_i = t % kB6XT;
      v4f ta[4], tb[4];
      {
        const v4f *xt = xring + xt_i * kB6TileV4;
        ta[0] = xt[lane]; ta[1] = xt[64u + lane]; ta[2] = xt[128u + lane]; ta[3] = xt[192u + lane];
      }
      for (;;) {
        if (!tile(ta, tb)) break;
        if (!tile(tb, ta)) break;
      }
      if (t == nt) {
        // the launch's last tile is done (in this generation): leave the final state where the frame wave looks for it
        fin[192u + lane] = (v4f){B.px1, B.px2, B.py, B.pv}; fin[256u + lane] = (v4f){B.last_phase, 0.f, 0.f, 0.f};
        b6_post(&ctr[C6_P4], (gen << 24) | (nh + 1u));
        t = nt + 1u;                                          // (nothing more to do but wait for `done` or a rewind)
      }
    }
    FSK_STAMP_END(4)
  } else if (role == 5u) {
    // ---------------------------------------------------------------------------------------------- frame logic (and every rare path)
    BackLane B;
    BackK K;
    back_load<UNI, 0>(B, K, P, S, C, stream, out_counts, eod_counts, append);
    if (B.dph >= kHandPairs) { B.qai = 0.f; B.qaq = 0.f; B.qbi = 0.f; B.qbq = 0.f; }   // the discriminator wave's
    if (!C.valid) { B.px1 = B.px2 = B.py = B.pv = 0.f; B.last_phase = 0.f; B.thf = 0.f; }   // (see `trash`)
    BlkK Q;
    Q.stop_m1 = (1u << P.stop_pos) - 1u; Q.sh9 = P.stop_pos - 9u; Q.ff = 0xFFu;
    asm volatile("" : "+v"(Q.stop_m1), "+v"(Q.sh9), "+v"(Q.ff));
    const FastMem &M = C.M;
    const uint32_t fld = C.fld, row4 = C.row4;
    const uint32_t phase0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(poly_phase));
    {
      uint32_t ph = phase0;
      for (uint32_t i = 0; i < P.d; i++) {                    // rotate: LDS index 0 = the register of the first push
        uint32_t r = 0u;
        if (mine) r = gpoly[ph * 64u + lane];
        poly[lane * PS + i] = r;
        ph = ph + 1u == P.d ? 0u : ph + 1u;
      }
    }
    BackU X;
    X.own_pairs = kHandPairs; X.hand_lag = kHandLag;          // (this wave keeps the correction for the whole un-retired span)
    X.k = 0; X.kv = 0; X.free0 = free0; X.zmail = zmail; X.cmail = cmail; X.phase = 0;
    X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) ? kDirectPairs : 0u;
    X.zlive = __builtin_amdgcn_ballot_w64(B.dph < kHandPairs) ? 1u : 0u;
    asm volatile("" : "+v"(X.kv));
    const uint32_t amp_pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)PIPE_ILOAD(amp_pos));
    const uint32_t amp_quad_bytes = P.n_streams * 16u;
    X.amp_soff = amp_soff_of(amp_pos0, amp_quad_bytes);
    const uint32_t amp_wrap = (P.amp_cap >> 2) * amp_quad_bytes;
    const bool amp_misaligned = (amp_pos0 & 3u) != 0u;
    const __amdgpu_buffer_rsrc_t amp_rsrc = __builtin_amdgcn_make_buffer_rsrc(S.amp_ring, 0, (int)amp_wrap, 0x00020000);
    uint32_t slot_t = 0;                                      // x / f ring tile of half tile t
    uint32_t pidx = 0;
    uint32_t bq = 0, nq = 0;
    uint32_t *prow = poly + lane * PS;
    uint32_t rare_tiles = 0;
    uint32_t gen = 0;
    bool own_post = false;                                    // the post filter / lastPhase in B are this wave's (a per-sample run is in progress)
    bool matched_stale = false;                               // lean blocks have run: B.matched is to be re-formed from the polyphase registers
    auto matched_now = [&]() {
      uint32_t m = 0;
      for (uint32_t i = 0; i < P.d; i += 4u) {
        const uint4 r = *reinterpret_cast<const uint4 *>(prow + i);
        m += (uint32_t)__builtin_popcount((r.x ^ K.qn) & K.mask) + (uint32_t)__builtin_popcount((r.y ^ K.qn) & K.mask) +
             (uint32_t)__builtin_popcount((r.z ^ K.qn) & K.mask) + (uint32_t)__builtin_popcount((r.w ^ K.qn) & K.mask);
      }
      B.matched = m;
      matched_stale = false;
    };
    uint32_t *ctr1 = ctr + 4;
    FSK_STAMP_BEGIN
    uint32_t t = 0;                                           // half tiles consumed (a block = a tile = two of them)
    while (t < nh) {
      bool rare_exit = X.zlive != 0u || amp_misaligned;
      uint32_t hardw = 0;                                     // the block test's hard flags of the tile that left the block loop
      if (!rare_exit) {
        if (own_post) {
          // back on the block path: P4 resumes at this tile with the state the per-sample run ended with
          gen = (gen + 1u) & 0xFFu;
          rmail[lane] = (v4f){B.px1, B.px2, B.py, B.pv};
          rmail[64u + lane] = (v4f){B.last_phase, B.thf, 0.f, 0.f};
          b6_post(&ctr[C6_RW], (gen << 24) | (t >> 1));
          own_post = false;
        }
        uint32_t pw = 0;
        {
          FSK_STAMP_W0 FSK_WAIT_BEGIN
          for (;;) {
            pw = lds_peek(&ctr[C6_P4]);
            if ((pw >> 24) == gen && (pw & 0xFFFFFFu) >= t + 2u) break;
            B6_SPIN(FSK_B6_SLEEP_RING);
          }
          FSK_STAMP_W1
        }
        uint32_t produced = pw & 0xFFFFFFu;
        uint32_t lim0 = (t | (kFlushBlocks - 1u)) + 1u;
        lim0 = lim0 < nh ? lim0 : nh;
        uint32_t lim = lim0 < (produced & ~1u) ? lim0 : (produced & ~1u);
        // One block: `cur` holds its f ring entries; the NEXT tile's are read into `nxt` meanwhile -- whether they were there
        // yet is known from the counters read just before them.  Returns 0 to go on with the next tile at once, 1 to stop (the
        // next tile is not there yet, or a flush point), 2 if this block has to be redone sample by sample (nothing committed).
        auto ftile = [&](auto lean, const v4f (&cur)[4], v4f (&nxt)[4]) -> int {
          v4u32 cv;
          lds_peek4_begin(ctr1, cv);
          const uint32_t pidx2 = pidx + 4u >= P.d ? 0u : pidx + 4u;
          const uint4 rpa = *reinterpret_cast<const uint4 *>(prow + pidx), rpb = *reinterpret_cast<const uint4 *>(prow + pidx2);
          {
            const uint32_t sn = slot_t + 1u == kB6XT ? 0u : slot_t + 1u;
            const v4f *fn = fring + sn * kB6TileV4;
            nxt[0] = fn[lane]; nxt[1] = fn[64u + lane]; nxt[2] = fn[128u + lane]; nxt[3] = fn[192u + lane];
          }
          BackLane Bn = B;
          uint32_t rp[kBlk] = {rpa.x, rpa.y, rpa.z, rpa.w, rpb.x, rpb.y, rpb.z, rpb.w};
          float am[kBlk];
          uint32_t bqn = bq, nqn = nq;
          const uint32_t rare = blk6_fast<decltype(lean)::value>(Bn, K, Q, X.kv, cur, rp, am, bqn, nqn, hardw);
          FSK_STAMP_COUNT(0)
          if (__builtin_expect(__builtin_amdgcn_ballot_w64((int32_t)rare < 0) != 0ull, 0)) return 2;
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
          slot_t = slot_t + 1u == kB6XT ? 0u : slot_t + 1u;
          pidx = pidx2 + 4u >= P.d ? 0u : pidx2 + 4u;
          t += 2u;
          b6_post(&ctr[C6_CONS], t);                           // slots free (this wave's reads of them are complete)
          const uint32_t pn = lds_peek4_get(cv, 1);
          if ((pn >> 24) == gen) produced = pn & 0xFFFFFFu;
          lim = lim0 < (produced & ~1u) ? lim0 : (produced & ~1u);
          if (t < lim) return 0;
          if (t >= lim0) return 1;
          // The next tile was not there when this one began -- the counters above are a tile old, and P4 works one tile ahead of
          // this wave, so that is the usual case: look again (and wait here while it still is not), then go on with entries read
          // afresh.  Leaving the loop instead cost ~500 cycles per tile (the outer loop's polls, two exposed LDS round trips).
          {
            FSK_STAMP_W0 FSK_WAIT_BEGIN
            for (;;) {
              const uint32_t pw2 = lds_peek(&ctr[C6_P4]);
              if ((pw2 >> 24) == gen && (pw2 & 0xFFFFFFu) >= t + 2u) { produced = pw2 & 0xFFFFFFu; break; }
              B6_SPIN(FSK_B6_SLEEP_RING);
            }
            FSK_STAMP_W1
          }
          lim = lim0 < (produced & ~1u) ? lim0 : (produced & ~1u);
          return 3;
        };
        auto fload = [&](v4f (&dst)[4]) {
          const v4f *ft = fring + slot_t * kB6TileV4;
          dst[0] = ft[lane]; dst[1] = ft[64u + lane]; dst[2] = ft[128u + lane]; dst[3] = ft[192u + lane];
        };
        v4f fa[4], fb[4];
        fload(fa);
        // every stream of the wave inside a frame (lanes without a stream do not count): the lean block, matched re-formed later
        if (FSK_B6_LEAN && __builtin_amdgcn_ballot_w64(C.valid & (B.thr_eff != kStartedP)) == 0ull) {
          matched_stale = true;
          for (;;) {
            int r = ftile(std::true_type(), fa, fb);
            if (r == 3) { fload(fb); r = 0; }
            if (r == 0) { r = ftile(std::true_type(), fb, fa); if (r == 3) { fload(fa); r = 0; } }
            if (r != 0) { rare_exit = r == 2; break; }
          }
        } else {
          for (;;) {
            int r = ftile(std::false_type(), fa, fb);
            if (r == 3) { fload(fb); r = 0; }
            if (r == 0) { r = ftile(std::false_type(), fb, fa); if (r == 3) { fload(fa); r = 0; } }
            if (r != 0) { rare_exit = r == 2; break; }
          }
        }
      }
      if (rare_exit) {
        // something rare in the tile at t (or a lane inside the own span after a reset): sample by sample from the tile's
        // entry state, as fsk_blk.hip's back wave does it -- the post filter and lastPhase being P4's on entering this tile
        if (!own_post) {
          if (t != 0u) {
            FSK_WAIT_BEGIN
            for (;;) {                                         // (P4 has produced this tile: its entry state is in the history)
              const uint32_t pw = lds_peek(&ctr[C6_P4]);
              if ((pw >> 24) == gen && (pw & 0xFFFFFFu) >= t + 2u) break;
              B6_SPIN(FSK_B6_SLEEP_RING);
            }
            const v4f *ht = hist + slot_t * 2u * 64u;
            const v4f a = ht[lane], b = ht[64u + lane];
            B.px1 = a.x; B.px2 = a.y; B.py = a.z; B.pv = a.w; B.last_phase = b.x;
          }                                                    // (t = 0: the launch's own start state, loaded above)
          own_post = true;
        }
        {
          uint32_t p3 = 0;
          FSK_STAMP_W0 FSK_WAIT_BEGIN
          while ((p3 = lds_peek(&ctr[C6_X])) < t + 2u) B6_SPIN(FSK_B6_SLEEP_RING);
          FSK_STAMP_W1
        }
        if (X.zlive != 0u) { rare_tiles++; FSK_STAMP_COUNT(1) } else { FSK_STAMP_COUNT(3) }
        if (matched_stale) matched_now();
        // ---- an 'eod' in the tile, or a lane inside the own span after one -- and no sync candidate or bad start / stop bit the
        // block test has seen: the block path that takes resets (blk_medium's sequence, straight-line: a rare tile sample by
        // sample is ~2 200 instructions and 240 branches, 13 fast tiles' time, and nothing upstream can run ahead meanwhile)
        bool medium_done = false;
        if (FSK_B6_MEDIUM && !amp_misaligned && __builtin_amdgcn_ballot_w64((int32_t)hardw < 0) == 0ull) {
          const v4f *xt = xring + slot_t * kB6TileV4;
          const v4f *ztile = zt + ((t >> 1) & ZTM) * 8u;
          const v4f *ys0 = yring + (t % NY) * 2u * 64u, *ys1 = yring + ((t + 1u) % NY) * 2u * 64u;
          const uint32_t pidx2 = pidx + 4u >= P.d ? 0u : pidx + 4u;
          const v4f xa[4] = {xt[lane], xt[64u + lane], xt[128u + lane], xt[192u + lane]};
          const v4f ya[4] = {ys0[lane], ys0[64u + lane], ys1[lane], ys1[64u + lane]};
          const v4f zc[4] = {ztile[0], ztile[1], ztile[2], ztile[3]}, zs[4] = {ztile[4], ztile[5], ztile[6], ztile[7]};
          const uint4 oa = *reinterpret_cast<const uint4 *>(prow + pidx), ob = *reinterpret_cast<const uint4 *>(prow + pidx2);
          const uint32_t ro[kBlk] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
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
          BackLane Bn = B;
          float am[kBlk];
          uint32_t bqn = bq, nqn = nq, wbits;
          MedEv E;
          const uint32_t hard = blk6_medium<UNI>(Bn, K, Q, P.matched_min, X.kv, xa, ya, zc, zs, ro, thf8, am, bqn, nqn, E, wbits, X.free0, inc, X.k);
          FSK_STAMP_COUNT(2)
          if (__builtin_amdgcn_ballot_w64((int32_t)hard < 0) == 0ull) {
            B = Bn; bq = bqn; nq = nqn;
            // syncSamplesBuffer.put x 8: old register << 1 | the sample's slicer bit (sample 1 is bit 7 of wbits)
            *reinterpret_cast<uint4 *>(prow + pidx) = make_uint4((oa.x << 1) | ((wbits >> 7) & 1u), (oa.y << 1) | ((wbits >> 6) & 1u),
                                                                 (oa.z << 1) | ((wbits >> 5) & 1u), (oa.w << 1) | ((wbits >> 4) & 1u));
            *reinterpret_cast<uint4 *>(prow + pidx2) = make_uint4((ob.x << 1) | ((wbits >> 3) & 1u), (ob.y << 1) | ((wbits >> 2) & 1u),
                                                                  (ob.z << 1) | ((wbits >> 1) & 1u), (ob.w << 1) | (wbits & 1u));
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
            pidx = pidx2 + 4u >= P.d ? 0u : pidx2 + 4u;
            t += 2u;
            // what the tile's resets owe memory and the other waves' mailboxes, once, before the tile's slots are released
            if (__builtin_amdgcn_ballot_w64((E.jr | E.jc) != 0u)) {
              if (E.jc != 0u) {                                // (zir_step: the correction's start values, for the discriminator wave)
                cmail[64u + lane] = __builtin_bit_cast(uint32_t, E.cai); cmail[128u + lane] = __builtin_bit_cast(uint32_t, E.caq);
                cmail[192u + lane] = __builtin_bit_cast(uint32_t, E.cbi); cmail[256u + lane] = __builtin_bit_cast(uint32_t, E.cbq);
                cmail[lane] = k0 + E.jc + kHandLag;
              }
              if (E.jr != 0u) {                                // (back_pair's 'eod' + back_reset)
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
            X.zlive = __builtin_amdgcn_ballot_w64(B.dph < kHandPairs) != 0ull ? 1u : 0u;
            X.direct = __builtin_amdgcn_ballot_w64(B.dph < kDirectPairs) != 0ull ? kDirectPairs : 0u;
            slot_t = slot_t + 1u == kB6XT ? 0u : slot_t + 1u;
            b6_post(&ctr[C6_CONS], t);
            medium_done = true;
          }
        }
        if (!medium_done) {
        blk_flush(B, bq, nq, M, out, (uint32_t)out_pitch);
        const v4f *ztile = zt + ((t >> 1) & ZTM) * 8u;
        const float *ztf = reinterpret_cast<const float *>(ztile);
        const v4f *xt = xring + slot_t * kB6TileV4;
#pragma unroll 1
        for (uint32_t hh = 0; hh < 2; hh++) {
          const v4f *yslot = yring + (t % NY) * 2u * 64u;    // (P1 wrote half tile t of this launch there)
          const v4f ua = xt[hh * 64u + lane], ub = xt[128u + hh * 64u + lane];   // pair sums where the own span covers the lane, else phase | magnitude
          const v4f ya = yslot[lane], yb = yslot[64u + lane];
          const uint4 rq = *reinterpret_cast<const uint4 *>(prow + pidx);
          const v4f zc0 = ztile[2u * hh], zc1 = ztile[2u * hh + 1u], zs0 = ztile[4u + 2u * hh], zs1 = ztile[5u + 2u * hh];
          const float zc[8] = {zc0.x, zc0.y, zc0.z, zc0.w, zc1.x, zc1.y, zc1.z, zc1.w};
          const float zs[8] = {zs0.x, zs0.y, zs0.z, zs0.w, zs1.x, zs1.y, zs1.z, zs1.w};
          const float u0[4] = {ua.x, ua.y, ua.z, ua.w}, u1[4] = {ub.x, ub.y, ub.z, ub.w};
          const float yv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
          const uint32_t ro[4] = {rq.x, rq.y, rq.z, rq.w};
          uint32_t rn[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            X.k++;
            X.kv += 1u;
            const float zph[4] = {zc[2 * j], zs[2 * j], zc[2 * j + 1], zs[2 * j + 1]};
            back_pair<UNI, true, false, true, COH>(B, K, P, S, M, &rn[j], lane, amp_rsrc, out, (uint32_t)out_pitch, eod_counts, X,
                                                   u0[j], u1[j], &yv[2 * j], ro[j], inc, u0[j], u1[j], UNI ? zph : nullptr);
            amp_advance(X.amp_soff, amp_quad_bytes, amp_wrap);
          }
          *reinterpret_cast<uint4 *>(prow + pidx) = make_uint4(rn[0], rn[1], rn[2], rn[3]);
          pidx = pidx + 4u >= P.d ? 0u : pidx + 4u;
          t++;
        }
        (void)ztf;
        slot_t = slot_t + 1u == kB6XT ? 0u : slot_t + 1u;
        b6_post(&ctr[C6_CONS], t);
        }
      }
      if ((t & (kFlushBlocks - 1u)) == 0u) blk_flush(B, bq, nq, M, out, (uint32_t)out_pitch);
    }
    blk_flush(B, bq, nq, M, out, (uint32_t)out_pitch);
    if (matched_stale) matched_now();
    FSK_STAMP_END(5)
    if (lane == 0 && S.blk_stat && (grp & 63u) == 0u) {
      __hip_atomic_fetch_add(&S.blk_stat[0], (uint32_t)n_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&S.blk_stat[1], rare_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the other waves' final states
    FSK_WAIT_BEGIN
    while (lds_peek(&ctr[C6_X]) <= nh) B6_SPIN(1);
    while (lds_peek(&ctr[C6_IQ]) <= nh) B6_SPIN(1);
    if (!own_post) {
      // (P4 has run the last tile in this generation -- this wave consumed it -- and leaves its state behind it)
      for (;;) {
        const uint32_t pw = lds_peek(&ctr[C6_P4]);
        if ((pw >> 24) == gen && (pw & 0xFFFFFFu) > nh) break;
        B6_SPIN(1);
      }
      const v4f a = fin[192u + lane], b = fin[256u + lane];
      B.px1 = a.x; B.px2 = a.y; B.py = a.z; B.pv = a.w; B.last_phase = b.x;
    }
    b6_post(&ctr[C6_DONE], 1u);
    FrontLane F;
    {
      const v4f fi = fin[lane], fq = fin[64u + lane];
      F.ix1 = fi.x; F.ix2 = fi.y; F.iy = fi.z; F.iv = fi.w;
      F.qx1 = fq.x; F.qx2 = fq.y; F.qy = fq.z; F.qv = fq.w;
      F.g = F.bx1 = F.bx2 = F.by1 = F.by2 = 0.f;
    }
    if (B.dph >= kHandPairs) {                                // the correction as the discriminator wave left it
      const v4f qz = fin[128u + lane];
      B.qai = qz.x; B.qaq = qz.y; B.qbi = qz.z; B.qbq = qz.w;
      // ... unless the hand-over sample is the FIRST of the next launch (fsk_blk.hip, round 4): it is still in the mailbox
      if (cmail[lane] == X.k) {
        B.qai = __builtin_bit_cast(float, cmail[64u + lane]); B.qaq = __builtin_bit_cast(float, cmail[128u + lane]);
        B.qbi = __builtin_bit_cast(float, cmail[192u + lane]); B.qbq = __builtin_bit_cast(float, cmail[256u + lane]);
        const uint32_t steps = X.k > kHandLag ? kHandLag : 0u;
        for (uint32_t g = 0; g < steps; g++) {
          const float ni = __builtin_fmaf(K.c1, B.qbi, -(K.c2 * B.qai)), nq2 = __builtin_fmaf(K.c1, B.qbq, -(K.c2 * B.qaq));
          B.qai = B.qbi; B.qaq = B.qbq; B.qbi = ni; B.qbq = nq2;
        }
      }
    }
    {
      uint32_t ph = phase0;
      for (uint32_t i = 0; i < P.d; i++) {
        if (mine) gpoly[ph * 64u + lane] = poly[lane * PS + i];
        ph = ph + 1u == P.d ? 0u : ph + 1u;
      }
    }
    const uint32_t phase_end = (phase0 + X.k) % P.d;
    pipe_store<UNI, 0>(F, false, B, P, C, stream, out_counts, n, X.k, X.k % P.cadence, phase_end, amp_pos_of(X.amp_soff, amp_quad_bytes), inc, free0);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
size_t demod_blk6_lds_bytes(const DemodParams &P, uint32_t y_slots) {
  return sizeof(float4) * (kB6Stage * 4 * kSlotStride + y_slots * 2 * 64 + (2 * kB6XT + 1) * kB6TileV4 + kB6XT * 2 * 64 + 2 * 64 + 5 * 64 +
                           blk6_zt_tiles(y_slots) * 8) +
         sizeof(uint32_t) * (64u * blk_poly_stride(P.d) + 16u + 64u + 6u * 64u);
}
// the y ring as deep as the LDS of a compute unit this workgroup has to itself allows
uint32_t demod_blk6_min_y_slots() { return kB6YMin; }
uint32_t demod_blk6_y_slots(const DemodParams &P) {
  uint32_t y = kB6YMax;
  while (y > kB6YMin && demod_blk6_lds_bytes(P, y) > 150u * 1024u) y -= 2u;
  return y;
}
bool demod_blk6_applicable(const DemodParams &P) {
  return P.d >= 8u && (P.d & 3u) == 0u && !P.wide && !P.frac && demod_blk6_lds_bytes(P, kB6YMin) <= 150u * 1024u;   // (round 6: per-stream tone pairs too)
}
// half-tile counters carry a generation in their top byte
size_t demod_blk6_max_samples() { return ((size_t)1 << 23) * 16u - 16u; }

hipError_t set_blk6_lds_limit(const DemodParams &P) {
  hipError_t e = hipSuccess;
  const size_t bytes = demod_blk6_lds_bytes(P, demod_blk6_y_slots(P));
#define FSK_ATTR(WBV, LWV)                                                                                       \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_blk6_kernel<WBV, LWV, true>),                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                            \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&demod_blk6_kernel<WBV, LWV, false>),                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  FSK_ATTR(false, 64) FSK_ATTR(false, 32) FSK_ATTR(false, 16) FSK_ATTR(false, 8)
  FSK_ATTR(true, 64) FSK_ATTR(true, 32) FSK_ATTR(true, 16) FSK_ATTR(true, 8)
#undef FSK_ATTR
  return e;
}

// default part of each wave (parts: 0 loader, 1 AGC, 2 pre-filter, 3 iq, 4 post, 5 frame, 6 disc).  Waves w and w + 4 of a
// workgroup share a SIMD (its seven waves go round the CU's four): the frame logic -- the longest instruction stream and the one
// every rare path runs on -- has a SIMD to itself, loader + post, AGC + pre-filter and iq + disc share.
uint32_t demod_blk6_default_rolemap(uint32_t lanes, bool uniform) {
  (void)lanes;
  // (per-stream tone pairs: the iq wave rotates a phasor per lane -- 83 instead of 66 busy cycles per sample -- and takes the
  // pre-filter for a neighbour instead of the post wave: loader + AGC, iq + pre-filter, post + disc share; 89 -> 94 Gsamples/s at
  // config #4's per-GPU share, profiles/r06_seven_wave_per_stream.txt)
  const uint32_t part_u[kB6Waves] = {0u, 1u, 3u, 5u, 6u, 2u, 4u}, part_p[kB6Waves] = {0u, 3u, 4u, 5u, 1u, 2u, 6u};
  const uint32_t *part = uniform ? part_u : part_p;
  uint32_t m = 0;
  for (uint32_t w = 0; w < kB6Waves; w++) m |= part[w] << (3u * w);
  return m;
}

hipError_t launch_demod_blk6(bool writeback, bool append, const DemodParams &P, const DemodState &S, float *samples, size_t n,
                              size_t pitch, uint8_t *out, size_t out_pitch, uint32_t *out_counts, uint32_t *eod_counts,
                              hipStream_t stream, uint32_t lanes, uint32_t y_slots, uint32_t rolemap) {
  lanes = (lanes == 8u || lanes == 16u || lanes == 32u) ? lanes : 64u;
  const uint32_t blocks = (P.n_streams + lanes - 1u) / lanes;
  const uint32_t ymax = demod_blk6_y_slots(P);
  y_slots = y_slots < kB6YMin ? kB6YMin : y_slots > ymax ? ymax : y_slots;
  y_slots &= ~1u;
  const size_t lds = demod_blk6_lds_bytes(P, y_slots);
  Blk6Z Z = {y_slots, blk6_zt_tiles(y_slots), rolemap ? rolemap : demod_blk6_default_rolemap(lanes, P.uni_cfg != 0u)};
#define FSK_LAUNCH_B6(WBV, LWV)                                                                                      \
  do {                                                                                                               \
    if (P.uni_cfg != 0u)                                                                                             \
      hipLaunchKernelGGL((demod_blk6_kernel<WBV, LWV, true>), dim3(blocks), dim3(64 * kB6Waves), lds, stream, P, S, samples, n, pitch, \
                         append ? 1 : 0, out, out_pitch, out_counts, eod_counts, Z);                                \
    else                                                                                                             \
      hipLaunchKernelGGL((demod_blk6_kernel<WBV, LWV, false>), dim3(blocks), dim3(64 * kB6Waves), lds, stream, P, S, samples, n, pitch, \
                         append ? 1 : 0, out, out_pitch, out_counts, eod_counts, Z);                                \
  } while (0)
  if (writeback) {
    if (lanes == 64u) FSK_LAUNCH_B6(true, 64); else if (lanes == 32u) FSK_LAUNCH_B6(true, 32);
    else if (lanes == 16u) FSK_LAUNCH_B6(true, 16); else FSK_LAUNCH_B6(true, 8);
  } else {
    if (lanes == 64u) FSK_LAUNCH_B6(false, 64); else if (lanes == 32u) FSK_LAUNCH_B6(false, 32);
    else if (lanes == 16u) FSK_LAUNCH_B6(false, 16); else FSK_LAUNCH_B6(false, 8);
  }
#undef FSK_LAUNCH_B6
  return hipGetLastError();
}

}  // namespace fsk

#ifdef FSK_STAMP
extern "C" int fskdbg_read_stamps_blk6(unsigned long long *out, size_t count) {   // diagnostic builds only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fsk::g_stamp), count * sizeof(unsigned long long));
}
#endif
