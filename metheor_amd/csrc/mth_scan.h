// mth_scan.h -- shared compaction helpers: per-block counts of set flags, single-block exclusive
// scan of those counts (+ running totals of a result table), used by the row emitters.
#pragma once
#include "mth_common.h"

namespace mth {

constexpr int SCAN_PER = 8;   // flags per thread in the blockcount / emit kernels (256 threads -> 2048 per block)

// blk[b] = number of entries with bit0 set among flags[b*2048 .. ), entries >= *n ignored
__global__ void k_flags_blockcount(const uint32_t *flags, const unsigned long long *n, uint32_t *blk);
// blk <- exclusive scan(blk); *base = *total; *total += sum; batch_rows[batch_idx] = sum
// (n_ptr, optional: the device-side number of entries the counts cover -- blocks beyond it are skipped; their blk entries are not
// rewritten, and the emitters do not read them: their blocks return before looking at blk when they hold no entry)
__global__ void k_block_scan(uint32_t *blk, uint32_t nblk, unsigned long long *total, unsigned long long *base,
                             uint32_t *batch_rows, uint32_t batch_idx, const unsigned long long *n_ptr = nullptr);

}  // namespace mth
