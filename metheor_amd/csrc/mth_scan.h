// mth_scan.h -- shared compaction helpers: per-block counts of set flags, single-block exclusive
// scan of those counts (+ running totals of a result table), used by the row emitters.
#pragma once
#include "mth_common.h"

namespace mth {

constexpr uint32_t SCAN_GRID_MAX = 4096;   // workgroups of the blockcount / emit kernels (they stride over the blocks that exist)
constexpr int SCAN_PER = 8;   // flags per thread in the blockcount / emit kernels (256 threads -> 2048 per block)

// blk[b] = number of entries with bit0 set among flags[b*2048 .. ), entries >= *n ignored
__global__ void k_flags_blockcount(const uint32_t *flags, const unsigned long long *n, uint32_t *blk);
// blk <- exclusive scan(blk); *base = *total; *total += sum; batch_rows[batch_idx] = sum
// (n_ptr, optional: the device-side number of entries the counts cover -- blocks beyond it are skipped; their blk entries are not
// rewritten, and the emitters do not read them: their blocks return before looking at blk when they hold no entry)
__global__ void k_block_scan(uint32_t *blk, uint32_t nblk, unsigned long long *total, unsigned long long *base,
                             uint32_t *batch_rows, uint32_t batch_idx, const unsigned long long *n_ptr = nullptr);

#ifdef __HIPCC__
// The emitters' compaction of one block's 256 x SCAN_PER entries, wave-cooperative: wave w takes the block's entries [w * 512, w * 512 + 512)
// in SCAN_PER steps of 64 CONSECUTIVE entries -- coalesced loads, and the kept entries of a step go to consecutive rows -- instead of
// SCAN_PER consecutive entries per lane (every load and store a 32-byte stride across the wave).  put(entry, row) for every entry
// with bit 0 of its flag set; rows ascend with the entries.  Block b's first row = row_base + blk[b].  All 256 threads must call.
template <class Put>
__device__ __forceinline__ void emit_block(const uint32_t *__restrict__ flags, unsigned long long n, unsigned long long row_base,
                                           const uint32_t *__restrict__ blk, Put put) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    __shared__ uint32_t ws[5];
    // the grid is sized from the host's upper bound of n (often hundreds of times larger): a fixed number of workgroups strides over
    // the blocks that exist instead of launching one mostly-empty workgroup per possible block (SCAN_GRID_MAX)
    const unsigned long long nb = (n + 256ull * SCAN_PER - 1ull) / (256ull * SCAN_PER);
    for (unsigned long long b = blockIdx.x; b < nb; b += gridDim.x) {
        const unsigned long long e0 = b * (256 * SCAN_PER) + (unsigned long long)wave * (64 * SCAN_PER) + lane;
        unsigned long long bal[SCAN_PER];
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) {
            const unsigned long long e = e0 + 64ull * k;
            bal[k] = __ballot(e < n && (flags[e] & 1u));
            tot += (uint32_t)__builtin_popcountll(bal[k]);
        }
        if (lane == 0) ws[wave + 1] = tot;
        __syncthreads();
        if (threadIdx.x == 0) { ws[0] = 0; for (int w = 1; w <= 4; ++w) ws[w] += ws[w - 1]; }
        __syncthreads();
        unsigned long long o = row_base + blk[b] + ws[wave];
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) {
            if ((bal[k] >> lane) & 1ull) put(e0 + 64ull * k, o + (unsigned long long)__builtin_popcountll(bal[k] & below));
            o += (unsigned long long)__builtin_popcountll(bal[k]);
        }
        __syncthreads();                                        // ws is rewritten by the next trip
    }
}
#endif

}  // namespace mth
