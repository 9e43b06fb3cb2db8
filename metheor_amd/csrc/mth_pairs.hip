// mth_pairs.hip -- LPMD per-pair table (`metheor lpmd --pairs`): lpmd.rs:70-87 (add_pair_concordance),
// 89-122 (print_pair_statistics), pairs produced by readutil.rs:166-224.
//
// For every read with mapq >= min_qual and every pair (i<j) of its CpGs with
// min_distance <= relpos_j - relpos_i <= max_distance: key (abspos_i, abspos_j), +1 concordant or
// discordant.  Rows sorted by key; per-pair lpmd = n_d as f32 / (n_c as f32 + n_d as f32) (lpmd.rs:111).
//
// Device: global open-addressing table (key = pos1 << 32 | pos2, two u32 counters), sized from an
// exact counting pre-pass; a pair is owned by the batch whose region contains pos1 (halo reads
// included), so region / contig sharding needs no merge.  The reference sorts at print time; rows are
// sorted by key on the host in mth_lpmd_pairs_fetch (optional secondary output, not a hot kernel).
#include <algorithm>
#include <numeric>

#include "mth_ctx.h"
#include "mth_scan.h"

namespace mth {

constexpr unsigned long long PKEY_EMPTY = ~0ull;

__device__ __forceinline__ unsigned long long phash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

struct PairArgs {
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const void     *cpg_rel;
    unsigned long long *keys;
    uint32_t *cnt;                 // 2 per slot: concordant, discordant
    unsigned long long *n_updates; // counting pass
    unsigned long long mask;
    int32_t region_beg, region_end, min_dist, max_dist;
    uint32_t n_reads;
    uint8_t min_qual;
    unsigned long long *overflow;   // set when an insert ran out of probes
};

// COUNT: only count the updates (table sizing); otherwise insert them
template <typename RelT, bool COUNT>
__global__ __launch_bounds__(256) void k_pairs(const PairArgs a) {
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    unsigned long long mine = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.n_reads; i += gridDim.x * 256) {
        if (a.read_mapq[i] < a.min_qual) continue;                         // lpmd.rs:177
        const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
        for (uint32_t k = o0 + 1; k < o1; ++k) {
            const int32_t rk = (int32_t)rel[k];
            const uint32_t wk = a.cpg_pos[k];
            for (uint32_t j = k; j-- > o0;) {
                const int32_t dist = rk - (int32_t)rel[j];
                if (dist > a.max_dist) break;                              // readutil.rs:184
                if (dist < a.min_dist) continue;                           // readutil.rs:196
                const uint32_t wj = a.cpg_pos[j];
                const int32_t p1 = (int32_t)(wj & 0x7fffffffu);
                if (p1 < a.region_beg || p1 >= a.region_end) continue;     // owned by the region of pos1
                if (COUNT) { mine += 1; continue; }
                const unsigned long long key = ((unsigned long long)(uint32_t)p1 << 32) | (wk & 0x7fffffffu);
                unsigned long long h = phash(key) & a.mask;
                // table sized for the distinct pairs one expects, not for the updates: bounded probe + redo flag (as mth_quartet.hip)
                uint32_t probes = 0;
                bool placed = false;
                while (probes++ < 1024u) {
                    const unsigned long long cur = atomicCAS(&a.keys[h], PKEY_EMPTY, key);
                    if (cur == PKEY_EMPTY || cur == key) { placed = true; break; }
                    h = (h + 1) & a.mask;
                }
                if (placed) atomicAdd(&a.cnt[h * 2 + (((wj ^ wk) >> 31) ? 1 : 0)], 1u);  // lpmd.rs:79-86
                else *a.overflow = 1ull;
            }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
        __shared__ unsigned long long ws[4];
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.n_updates, ws[0] + ws[1] + ws[2] + ws[3]);
    }
}

__global__ __launch_bounds__(256) void k_pairs_blockcount(const unsigned long long *__restrict__ keys,
                                                          unsigned long long n_slots, uint32_t *__restrict__ blk) {
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) m += (s0 + k < n_slots && keys[s0 + k] != PKEY_EMPTY) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_pairs_emit(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ cnt,
                                                    unsigned long long n_slots, const uint32_t *__restrict__ blk,
                                                    const unsigned long long *__restrict__ base,
                                                    unsigned long long *__restrict__ out_key, uint32_t *__restrict__ out_cnt) {
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    uint32_t m = 0;
    unsigned long long kk[SCAN_PER];
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) { kk[k] = s0 + k < n_slots ? keys[s0 + k] : PKEY_EMPTY; m += kk[k] != PKEY_EMPTY ? 1u : 0u; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    __shared__ uint32_t ws[5];
    if (lane == 63) ws[wave + 1] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { ws[0] = 0; for (int w = 1; w <= 4; ++w) ws[w] += ws[w - 1]; }
    __syncthreads();
    unsigned long long o = *base + blk[blockIdx.x] + ws[wave] + incl - m;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        if (kk[k] == PKEY_EMPTY) continue;
        out_key[o] = kk[k];
        out_cnt[2 * o] = cnt[(s0 + k) * 2];
        out_cnt[2 * o + 1] = cnt[(s0 + k) * 2 + 1];
        ++o;
    }
}

}  // namespace mth

using namespace mth;

extern "C" {

int mth_lpmd_pairs_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_lpmd_pairs_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    mth_batch_t d;
    int rc = stage_batch(ctx, *batch, d);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    if (!ctx->p_state.p) {
        MTH_HIP(ctx, ctx->p_state.reserve(4 * sizeof(unsigned long long), s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->p_state.p, 0, 4 * sizeof(unsigned long long), s));
    }
    unsigned long long *ps = ctx->p_state.as<unsigned long long>();   // [0] updates of the batch [1] total rows [2] base
    MTH_HIP(ctx, hipMemsetAsync(ps, 0, sizeof(unsigned long long), s));
    PairArgs a;
    a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
    a.cpg_rel = d.cpg_rel ? (const void *)d.cpg_rel : (const void *)d.cpg_rel16;
    a.keys = nullptr; a.cnt = nullptr; a.n_updates = ps; a.mask = 0; a.overflow = ps + 3;
    a.region_beg = d.region_beg; a.region_end = d.region_end; a.min_dist = params->min_distance; a.max_dist = params->max_distance;
    a.n_reads = d.n_reads; a.min_qual = params->min_qual;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)d.n_reads + 255) / 256 + 1, 8192);
    const bool r8 = d.cpg_rel != nullptr;
    {
        LaunchTimer lt(ctx, K_PAIRS);
        if (r8) hipLaunchKernelGGL((k_pairs<uint8_t, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_pairs<uint16_t, true>), dim3(grid), dim3(256), 0, s, a);
    }
    unsigned long long bound = 0;
    MTH_HIP(ctx, hipMemcpyAsync(&bound, ps, sizeof bound, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipStreamSynchronize(s));                            // exact table sizing: one sync per batch
    // `bound` counts pair UPDATES; at depth D there are ~D per distinct pair, and the table is cleared and scanned once per
    // batch: start at bound / 2 slots and redo 4x larger if an insert ran out of probes (cannot happen at 2 x bound)
    unsigned long long n_slots = 1024;
    while (n_slots < bound / 2) n_slots <<= 1;
    if (const char *e = getenv("MTH_PAIRS_SLOTS_MIN")) { const unsigned long long k = strtoull(e, nullptr, 10); if (k >= 16) { n_slots = 16; while (n_slots < k) n_slots <<= 1; } }   // tests: force the retry
    for (;;) {
        MTH_HIP(ctx, ctx->p_keys.reserve(n_slots * 8, s));
        MTH_HIP(ctx, ctx->p_cnt.reserve(n_slots * 8, s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->p_keys.p, 0xFF, n_slots * 8, s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->p_cnt.p, 0, n_slots * 8, s));
        MTH_HIP(ctx, hipMemsetAsync(ps + 3, 0, sizeof(unsigned long long), s));
        a.keys = ctx->p_keys.as<unsigned long long>(); a.cnt = ctx->p_cnt.as<uint32_t>(); a.mask = n_slots - 1; a.overflow = ps + 3;
        {
            LaunchTimer lt(ctx, K_PAIRS);
            if (r8) hipLaunchKernelGGL((k_pairs<uint8_t, false>), dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_pairs<uint16_t, false>), dim3(grid), dim3(256), 0, s, a);
        }
        unsigned long long ovf = 0;
        MTH_HIP(ctx, hipMemcpyAsync(&ovf, ps + 3, sizeof ovf, hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));
        if (!ovf) break;
        n_slots <<= 2;
    }
    const uint64_t need = ctx->p_rows_bound + bound;
    if (need > ctx->p_cap) {
        const uint64_t ncap = need + need / 4 + 1024, used = ctx->p_rows_bound;
        MTH_HIP(ctx, ctx->p_out_key.reserve(ncap * 8, s, true, used * 8));
        MTH_HIP(ctx, ctx->p_out_cnt.reserve(ncap * 8, s, true, used * 8));
        ctx->p_cap = ncap;
    }
    ctx->p_rows_bound = need;
    const uint32_t nblk = (uint32_t)((n_slots + 256 * SCAN_PER - 1) / (256 * SCAN_PER));
    MTH_HIP(ctx, ctx->w_blk.reserve((size_t)nblk * 4, s));
    const size_t nb = ctx->p_batches.size();
    MTH_HIP(ctx, ctx->p_batch_rows.reserve((nb + 1) * 4, s, true, nb * 4));
    hipLaunchKernelGGL(k_pairs_blockcount, dim3(nblk), dim3(256), 0, s, ctx->p_keys.as<unsigned long long>(), n_slots,
                       ctx->w_blk.as<uint32_t>());
    hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->w_blk.as<uint32_t>(), nblk, ps + 1, ps + 2,
                       ctx->p_batch_rows.as<uint32_t>(), (uint32_t)nb);
    hipLaunchKernelGGL(k_pairs_emit, dim3(nblk), dim3(256), 0, s, ctx->p_keys.as<unsigned long long>(), ctx->p_cnt.as<uint32_t>(),
                       n_slots, ctx->w_blk.as<uint32_t>(), ps + 2, ctx->p_out_key.as<unsigned long long>(),
                       ctx->p_out_cnt.as<uint32_t>());
    MTH_HIP(ctx, hipGetLastError());
    ctx->p_batches.push_back(BatchMeta{batch->tid});
    return MTH_OK;
}

// rows sorted by ((tid,pos1),(tid,pos2)) given batches were submitted in (tid, region) order
int mth_lpmd_pairs_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos1, int32_t *pos2, float *lpmd,
                         uint32_t *n_concordant, uint32_t *n_discordant) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    unsigned long long ps[3] = {0, 0, 0};
    if (ctx->p_state.p) MTH_HIP(ctx, hipMemcpy(ps, ctx->p_state.p, sizeof ps, hipMemcpyDeviceToHost));
    const uint64_t n = ps[1];
    if (n_rows) *n_rows = n;
    if (n == 0 || (!tid && !pos1 && !pos2 && !lpmd && !n_concordant && !n_discordant)) return MTH_OK;
    std::vector<unsigned long long> key(n);
    std::vector<uint32_t> cnt(2 * n), rows(ctx->p_batches.size());
    MTH_HIP(ctx, hipMemcpy(key.data(), ctx->p_out_key.p, n * 8, hipMemcpyDeviceToHost));
    MTH_HIP(ctx, hipMemcpy(cnt.data(), ctx->p_out_cnt.p, n * 8, hipMemcpyDeviceToHost));
    if (!rows.empty()) MTH_HIP(ctx, hipMemcpy(rows.data(), ctx->p_batch_rows.p, rows.size() * 4, hipMemcpyDeviceToHost));
    // sort within runs of batches that share a tid (lpmd.rs:94: pairs.sort()); batches arrive tid-ordered
    std::vector<uint64_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    uint64_t o = 0;
    for (size_t b = 0; b < rows.size();) {
        size_t e = b;
        uint64_t cntrows = 0;
        while (e < rows.size() && ctx->p_batches[e].tid == ctx->p_batches[b].tid) { cntrows += rows[e]; ++e; }
        std::sort(order.begin() + o, order.begin() + o + cntrows, [&](uint64_t x, uint64_t y) { return key[x] < key[y]; });
        for (uint64_t r = 0; r < cntrows; ++r) if (tid) tid[o + r] = ctx->p_batches[b].tid;
        o += cntrows;
        b = e;
    }
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t x = order[r];
        const uint32_t c = cnt[2 * x], dd = cnt[2 * x + 1];
        if (pos1) pos1[r] = (int32_t)(key[x] >> 32);
        if (pos2) pos2[r] = (int32_t)(key[x] & 0x7fffffffull);
        if (n_concordant) n_concordant[r] = c;
        if (n_discordant) n_discordant[r] = dd;
        if (lpmd) lpmd[r] = (float)dd / ((float)c + (float)dd);           // lpmd.rs:111
    }
    return MTH_OK;
}

}  // extern "C"
