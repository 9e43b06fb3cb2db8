// mth_scan.hip -- see mth_scan.h
#include "mth_scan.h"

namespace mth {

__global__ __launch_bounds__(256) void k_flags_blockcount(const uint32_t *__restrict__ flags,
                                                          const unsigned long long *__restrict__ n_ptr,
                                                          uint32_t *__restrict__ blk) {
    const unsigned long long n = *n_ptr;
    __shared__ uint32_t ws[4];
    const unsigned long long nb = (n + 256ull * SCAN_PER - 1ull) / (256ull * SCAN_PER);
    for (unsigned long long b = blockIdx.x; b < nb; b += gridDim.x) {      // (see emit_block: the grid comes from an upper bound of n)
        // the block's 2048 entries in SCAN_PER sweeps of 256 consecutive ones: coalesced
        const unsigned long long s0 = b * (256 * SCAN_PER) + threadIdx.x;
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) m += (s0 + 256ull * k < n) ? (flags[s0 + 256ull * k] & 1u) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) blk[b] = ws[0] + ws[1] + ws[2] + ws[3];
        __syncthreads();
    }
}

// single block: exclusive scan of the per-block row counts; appends the batch to the totals
__global__ __launch_bounds__(1024) void k_block_scan(uint32_t *__restrict__ blk, uint32_t nblk,
                                                       unsigned long long *__restrict__ total,
                                                       unsigned long long *__restrict__ base,
                                                       uint32_t *__restrict__ batch_rows, uint32_t batch_idx,
                                                       const unsigned long long *__restrict__ n_ptr) {
    // n_ptr: the number of flagged entries the counts were taken over (device side); blocks beyond it counted nothing, and the
    // host's nblk comes from an upper bound that can be hundreds of times larger (every call of a contig against its rows)
    if (n_ptr) nblk = (uint32_t)min((unsigned long long)nblk, (*n_ptr + 256ull * SCAN_PER - 1ull) / (256ull * SCAN_PER));
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblk; b0 += 1024) {
        const uint32_t i = b0 + tid;
        const uint32_t v = i < nblk ? blk[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave + 1] = incl;
        __syncthreads();
        if (tid == 0) {
            wsum[0] = running;
            for (int w = 1; w <= 16; ++w) wsum[w] += wsum[w - 1];
        }
        __syncthreads();
        if (i < nblk) blk[i] = wsum[wave] + incl - v;
        __syncthreads();
        if (tid == 0) running = wsum[16];
        __syncthreads();
    }
    if (tid == 0) {
        *base = *total;
        *total += running;
        batch_rows[batch_idx] = running;
    }
}


}  // namespace mth
