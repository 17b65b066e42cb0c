// fsk_blk_sched.h -- how a call of demod_blk_kernel is cut into time slices (fsk_blk.hip, "Time slices"): plain integer
// arithmetic, no device code, so that the CPU suite can compile and test it by itself (tests/test_blk_sched_cpu.py).
#pragma once
#include <stdint.h>

namespace fsk {

static constexpr uint32_t kBlkSliceTiles = 768;    // default slice: 12 288 samples
static constexpr uint32_t kBlkMaxSlices = 128;     // per group (the queue's capacity is groups * (kBlkMaxSlices - 1) entries)
static constexpr uint32_t kBlkMinSliceTiles = 96;  // the packing search below never goes under this

// Slices per group for a call of n_tiles whole tiles over `groups` 64-stream groups on a device that holds `resident`
// workgroups at once, and the slice length in tiles (*slice_tiles_out; 0 when not sliced).  1 = one workgroup per group,
// not persistent: batches within one round, calls too short for two slices, slicing switched off (slice_tiles =
// 0xFFFFFFFF) or no queue.  slice_tiles = 0 starts from kBlkSliceTiles (or what keeps a group within kBlkMaxSlices) and
// then picks the slice count that packs best: items are handed out whole, so groups * ns items on `resident` workgroups
// take about ceil(groups * ns / resident) slice times -- 1 088 groups in 8 slices need 9 of them (1.125 rounds) where 16
// slices need 17 (1.0625) -- and every slice change costs ~17 us (1.1 % of a 768-tile slice at config #3's rate).
inline uint32_t blk_slice_count(uint32_t groups, uint32_t n_tiles, uint32_t resident, uint32_t slice_tiles, bool have_queue,
                                uint32_t *slice_tiles_out) {
  if (slice_tiles_out) *slice_tiles_out = 0;
  if (!(have_queue && resident && groups > resident && groups < (1u << 20) && slice_tiles != 0xFFFFFFFFu)) return 1u;
  uint32_t st = slice_tiles ? slice_tiles : kBlkSliceTiles;
  const uint32_t st_min = (n_tiles + kBlkMaxSlices - 1u) / kBlkMaxSlices;
  st = st < st_min ? st_min : st;
  uint32_t ns = (n_tiles + st - 1u) / st;
  if (ns < 2u) return 1u;
  if (!slice_tiles) {
    double best = 1e30;
    uint32_t best_ns = ns;
    for (uint32_t c = ns; c <= 4u * ns && c <= kBlkMaxSlices && (n_tiles + c - 1u) / c >= kBlkMinSliceTiles; c++) {
      const double rounds = (double)(((uint64_t)groups * c + resident - 1u) / resident) / (double)c;
      const double cost = rounds * (1.0 + 0.011 * (double)c / (double)ns);
      if (cost < best - 1e-9) { best = cost; best_ns = c; }
    }
    ns = best_ns;
    st = (n_tiles + ns - 1u) / ns;
    ns = (n_tiles + st - 1u) / st;
  }
  if (slice_tiles_out) *slice_tiles_out = st;
  return ns;
}

}  // namespace fsk
