// mth_pdr_lpmd.hip -- fused PDR + LPMD pass for gfx950 (CDNA4, wave64).
//
// What it computes (reference: src/pdr.rs:119-212, src/lpmd.rs:154-202, src/readutil.rs:134-145,
// 166-224): per CpG site the number of concordant / discordant reads covering it, and genome-wide
// concordant / discordant CpG-pair counts inside a query-distance window.
//
// How (MI355X-first, nothing like the reference's hash maps):
//   k_build_index   one thread per read: a linear index "first read starting at or after q*256 bp"
//                   (reads are coordinate sorted) + sortedness / span validation.
//   k_pdr_lpmd_tile one 256-thread workgroup per 4096-bp tile of the contig.  The tile's site
//                   accumulators are DENSE in LDS (2 x u32 per reference position, 32 KiB), so the
//                   scatter is an LDS atomic and HBM only sees the streamed SoA.  Reads that can
//                   touch the tile are found through the index (halo reads are re-read by the
//                   neighbour tile).  LPMD pair counts are reduced per wave with DPP shuffles and
//                   stored as per-tile partials (no same-address global atomics).  The tile's
//                   non-empty sites are compacted with a block scan into a per-tile scratch slice.
//   k_tile_scan     one workgroup: exclusive scan of per-tile site counts, LPMD partial reduce.
//   k_gather        packs the scratch slices into the final sorted SoA and computes the f32 PDR.
//
// Roofline: integer streaming + LDS atomics, HBM-bound by design (no MFMA: there is no
// contraction here).  Algorithmic bytes: 16 B/read + 5 B/CpG call in, 12 B/site out.
#include "mth_ctx.h"

namespace mth {

struct TileArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const void     *cpg_rel;
    const uint32_t *idx;
    const DevState *st;
    uint32_t *tile_cnt;
    uint32_t *tile_lpmd;   // 4 x u32 per tile
    SiteRec  *scratch;     // TILE_W rows per tile
    int32_t region_beg, region_end, idx_base, max_span;
    uint32_t n_reads;
    uint32_t min_cov;      // max(pdr_min_depth, 1)
    uint32_t min_cpgs;
    int32_t  min_dist, max_dist;
    uint8_t  pdr_min_qual, lpmd_min_qual, want_pdr, want_lpmd;
};

// ---------------------------------------------------------------------------------------------
// idx[q] = first read i with read_start[i] >= idx_base + q*IDX_Q   (q = 0..nq)
__global__ __launch_bounds__(BLOCK) void k_build_index(const int32_t *__restrict__ read_start,
                                                       const int32_t *__restrict__ read_end,
                                                       uint32_t n_reads, int32_t idx_base,
                                                       uint32_t nq, int32_t max_span,
                                                       uint32_t *__restrict__ idx,
                                                       DevState *__restrict__ st) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i > n_reads) return;
    auto bucket = [&](int32_t s) -> int64_t {  // floor((s-base)/Q), -1 below the base
        const int64_t d = (int64_t)s - idx_base;
        return d < 0 ? -1 : (d >> IDX_QSHIFT);
    };
    int64_t g_prev = -1, g_cur;
    uint32_t err = 0;
    if (i < n_reads) {
        const int32_t s = read_start[i];
        g_cur = bucket(s);
        if (i > 0) {
            const int32_t sp = read_start[i - 1];
            g_prev = bucket(sp);
            if (s < sp) err |= ERRB_UNSORTED;
        }
        if (s >= 0 && (int64_t)read_end[i] - s + 1 > max_span) err |= ERRB_SPAN;
    } else {  // sentinel thread closes the index
        g_cur = nq;
        if (n_reads > 0) g_prev = bucket(read_start[n_reads - 1]);
    }
    if (g_cur > (int64_t)nq) g_cur = nq;
    for (int64_t q = g_prev + 1; q <= g_cur; ++q) idx[q] = i;
    if (err) atomicOr(&st->err, err);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

template <typename RelT>
__global__ __launch_bounds__(BLOCK) void k_pdr_lpmd_tile(const TileArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t cnt[2 * TILE_W];  // [0,W): concordant, [W,2W): discordant
    __shared__ uint32_t red[4][BLOCK / 64];
    __shared__ uint32_t wave_off[BLOCK / 64 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t t = blockIdx.x;
    const int32_t T0 = a.region_beg + (int32_t)(t * TILE_W);
    const int32_t T1 = min(T0 + TILE_W, a.region_end);
    const uint32_t Wt = (uint32_t)(T1 - T0);

    // a batch that failed validation in k_build_index has no usable index: emit nothing
    if (a.st->err != 0) {
        if (tid == 0) a.tile_cnt[t] = 0;
        if (tid < 4) a.tile_lpmd[t * 4 + tid] = 0;
        return;
    }
    for (int i = tid; i < 2 * TILE_W / 4; i += BLOCK)
        reinterpret_cast<uint4 *>(cnt)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // candidate reads: start in [T0 - max_span + 1, T0 + TILE_W]  (a call sits in [start-1, end])
    const uint32_t lo = a.idx[(uint32_t)(T0 - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT];
    const uint32_t hi = min(a.idx[((uint32_t)(T0 + TILE_W - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);

    uint32_t lp_c = 0, lp_d = 0, n_read = 0, n_valid = 0;
    for (uint32_t i = lo + tid; i < hi; i += BLOCK) {
        const int32_t s = a.read_start[i];
        const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
        const uint32_t n = o1 - o0;
        const uint8_t mq = a.read_mapq[i];
        const bool owned = (s >= T0) && (s < T1);
        // lpmd.rs:176-179
        const bool lp_ok = a.want_lpmd && owned && (mq >= a.lpmd_min_qual);
        if (a.want_lpmd && owned) { n_read += 1; n_valid += lp_ok ? 1u : 0u; }
        // pdr.rs:147-157
        const bool pdr_ok = a.want_pdr && (n >= a.min_cpgs) && (mq >= a.pdr_min_qual) && (n > 0);
        if (!(lp_ok || pdr_ok) || n == 0) continue;

        // pass 1: read concordance (readutil.rs:134-145) + windowed pair counts (166-224)
        const uint32_t first = a.cpg_pos[o0] >> 31;
        uint32_t disc = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t mk = a.cpg_pos[o0 + k] >> 31;
            disc |= (mk ^ first);
            if (lp_ok) {
                const int32_t rk = (int32_t)rel[o0 + k];
                for (uint32_t j = k; j-- > 0;) {
                    const int32_t dist = rk - (int32_t)rel[o0 + j];
                    if (dist > a.max_dist) break;          // readutil.rs:184 (anchors evicted)
                    if (dist < a.min_dist) continue;       // readutil.rs:196
                    if ((a.cpg_pos[o0 + j] >> 31) == mk) lp_c += 1; else lp_d += 1;
                }
            }
        }
        // pass 2: scatter +1 to the tile's sites (pdr.rs:180-191)
        if (pdr_ok) {
            uint32_t *base = cnt + (disc ? TILE_W : 0);
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t p = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)T0;
                if (p < Wt) atomicAdd(base + p, 1u);
            }
        }
    }

    // per-tile LPMD partials (wave DPP reduce -> LDS -> one plain store per tile)
    if (a.want_lpmd) {
        const uint32_t r0 = wave_sum(lp_c), r1 = wave_sum(lp_d), r2 = wave_sum(n_read), r3 = wave_sum(n_valid);
        if (lane == 0) { red[0][wave] = r0; red[1][wave] = r1; red[2][wave] = r2; red[3][wave] = r3; }
    }
    __syncthreads();
    if (a.want_lpmd && tid < 4) {
        uint32_t s = 0;
        for (int w = 0; w < BLOCK / 64; ++w) s += red[tid][w];
        a.tile_lpmd[t * 4 + tid] = s;
    }
    if (!a.want_pdr) { if (tid == 0) a.tile_cnt[t] = 0; return; }

    // compaction: thread owns 16 consecutive positions; emit sites with coverage >= min_cov
    constexpr int PER = TILE_W / BLOCK;  // 16
    uint32_t c[PER], d[PER];
#pragma unroll
    for (int v = 0; v < PER / 4; ++v) {
        const uint4 x = reinterpret_cast<const uint4 *>(cnt)[tid * (PER / 4) + v];
        const uint4 y = reinterpret_cast<const uint4 *>(cnt + TILE_W)[tid * (PER / 4) + v];
        c[4 * v] = x.x; c[4 * v + 1] = x.y; c[4 * v + 2] = x.z; c[4 * v + 3] = x.w;
        d[4 * v] = y.x; d[4 * v + 1] = y.y; d[4 * v + 2] = y.z; d[4 * v + 3] = y.w;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int v = 0; v < PER; ++v) mine += (c[v] + d[v] >= a.min_cov) ? 1u : 0u;
    // block exclusive scan of `mine`
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_off[wave + 1] = incl;
    __syncthreads();
    if (tid == 0) {
        wave_off[0] = 0;
        for (int w = 1; w <= BLOCK / 64; ++w) wave_off[w] += wave_off[w - 1];
        a.tile_cnt[t] = wave_off[BLOCK / 64];
    }
    __syncthreads();
    uint32_t o = wave_off[wave] + incl - mine;
    SiteRec *__restrict__ out = a.scratch + (size_t)t * TILE_W;
#pragma unroll
    for (int v = 0; v < PER; ++v) {
        if (c[v] + d[v] >= a.min_cov) {
            SiteRec r; r.pos = T0 + tid * PER + v; r.n_conc = c[v]; r.n_disc = d[v]; r.pad = 0;
            out[o++] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// single workgroup: tile_base = exclusive scan(tile_cnt); LPMD partials -> DevState
__global__ __launch_bounds__(1024) void k_tile_scan(const uint32_t *__restrict__ tile_cnt,
                                                    const uint32_t *__restrict__ tile_lpmd,
                                                    uint32_t ntiles, uint32_t *__restrict__ tile_base,
                                                    uint32_t *__restrict__ batch_cnt, int want_lpmd,
                                                    DevState *__restrict__ st) {
    __shared__ uint32_t wsum[16 + 1];
    __shared__ unsigned long long lsum[4][16];
    __shared__ uint32_t running_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) running_s = 0;
    unsigned long long acc[4] = {0, 0, 0, 0};
    __syncthreads();
    for (uint32_t b = 0; b < ntiles; b += 1024) {
        const uint32_t i = b + tid;
        const uint32_t v = i < ntiles ? tile_cnt[i] : 0u;
        if (want_lpmd && i < ntiles) {
            const uint4 l = reinterpret_cast<const uint4 *>(tile_lpmd)[i];
            acc[0] += l.x; acc[1] += l.y; acc[2] += l.z; acc[3] += l.w;
        }
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave + 1] = incl;
        __syncthreads();
        if (tid == 0) {
            wsum[0] = running_s;
            for (int w = 1; w <= 16; ++w) wsum[w] += wsum[w - 1];
        }
        __syncthreads();
        if (i < ntiles) tile_base[i] = wsum[wave] + incl - v;
        __syncthreads();
        if (tid == 0) running_s = wsum[16];
        __syncthreads();
    }
    if (want_lpmd) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if (lane == 0) lsum[k][wave] = x;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = running_s;
        st->cur_base = st->n_sites;
        st->n_sites += total;
        batch_cnt[st->n_batches] = total;
        st->n_batches += 1;
    }
    if (want_lpmd && tid < 4) {
        unsigned long long s = 0;
        for (int w = 0; w < 16; ++w) s += lsum[tid][w];
        st->lpmd[tid] += (long long)s;
    }
}

// ---------------------------------------------------------------------------------------------
// pack per-tile scratch slices into the final SoA; pdr.rs:47-49 f32 expression
__global__ __launch_bounds__(64) void k_gather(const SiteRec *__restrict__ scratch,
                                               const uint32_t *__restrict__ tile_cnt,
                                               const uint32_t *__restrict__ tile_base,
                                               const DevState *__restrict__ st,
                                               int32_t *__restrict__ out_pos, float *__restrict__ out_pdr,
                                               uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    const uint32_t t = blockIdx.x;
    const uint32_t n = tile_cnt[t];
    const uint64_t base = st->cur_base + tile_base[t];
    const SiteRec *__restrict__ src = scratch + (size_t)t * TILE_W;
    for (uint32_t j = threadIdx.x; j < n; j += 64) {
        const SiteRec r = src[j];
        out_pos[base + j] = r.pos;
        out_nc[base + j] = r.n_conc;
        out_nd[base + j] = r.n_disc;
        out_pdr[base + j] = (float)r.n_disc / ((float)r.n_conc + (float)r.n_disc);
    }
}

// ---------------------------------------------------------------------------------------------
int launch_pdr_lpmd(mth_ctx *ctx, const mth_batch_t &b, const mth_pdr_lpmd_params_t &p) {
    hipStream_t s = ctx->stream;
    const int64_t region_len = (int64_t)b.region_end - b.region_beg;
    const uint32_t ntiles = (uint32_t)((region_len + TILE_W - 1) / TILE_W);
    if (ntiles == 0) return MTH_OK;
    // index origin: a whole number of quanta below the region so that halo reads are indexed
    const int32_t ext = ((b.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    const int32_t idx_base = b.region_beg - ext;
    const uint32_t nq = (uint32_t)(((int64_t)ntiles * TILE_W + ext) >> IDX_QSHIFT) + 2;

    MTH_HIP(ctx, ctx->idx.reserve((size_t)(nq + 1) * 4, s));
    MTH_HIP(ctx, ctx->tile_cnt.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_base.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_lpmd.reserve((size_t)ntiles * 16, s));
    if (p.want_pdr) MTH_HIP(ctx, ctx->scratch.reserve((size_t)ntiles * TILE_W * sizeof(SiteRec), s));

    {
        LaunchTimer lt(ctx, K_INDEX);
        const uint32_t nb = (b.n_reads + 1 + BLOCK - 1) / BLOCK;
        hipLaunchKernelGGL(k_build_index, dim3(nb), dim3(BLOCK), 0, s, b.read_start, b.read_end,
                           b.n_reads, idx_base, nq, b.max_span, ctx->idx.as<uint32_t>(), ctx->d_state);
    }
    TileArgs a;
    a.read_start = b.read_start; a.read_mapq = b.read_mapq; a.cpg_off = b.cpg_off; a.cpg_pos = b.cpg_pos;
    a.cpg_rel = b.cpg_rel ? (const void *)b.cpg_rel : (const void *)b.cpg_rel16;
    a.idx = ctx->idx.as<uint32_t>(); a.st = ctx->d_state;
    a.tile_cnt = ctx->tile_cnt.as<uint32_t>(); a.tile_lpmd = ctx->tile_lpmd.as<uint32_t>();
    a.scratch = ctx->scratch.as<SiteRec>();
    a.region_beg = b.region_beg; a.region_end = b.region_end; a.idx_base = idx_base; a.max_span = b.max_span;
    a.n_reads = b.n_reads;
    a.min_cov = p.pdr_min_depth > 1 ? p.pdr_min_depth : 1;
    a.min_cpgs = p.pdr_min_cpgs;
    a.min_dist = p.lpmd_min_distance; a.max_dist = p.lpmd_max_distance;
    a.pdr_min_qual = p.pdr_min_qual; a.lpmd_min_qual = p.lpmd_min_qual;
    a.want_pdr = p.want_pdr; a.want_lpmd = p.want_lpmd;
    {
        LaunchTimer lt(ctx, K_TILE);
        if (b.cpg_rel)
            hipLaunchKernelGGL(k_pdr_lpmd_tile<uint8_t>, dim3(ntiles), dim3(BLOCK), 0, s, a);
        else
            hipLaunchKernelGGL(k_pdr_lpmd_tile<uint16_t>, dim3(ntiles), dim3(BLOCK), 0, s, a);
    }
    {
        LaunchTimer lt(ctx, K_SCAN);
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, ctx->tile_cnt.as<uint32_t>(),
                           ctx->tile_lpmd.as<uint32_t>(), ntiles, ctx->tile_base.as<uint32_t>(),
                           ctx->batch_cnt.as<uint32_t>(), (int)p.want_lpmd, ctx->d_state);
    }
    if (p.want_pdr) {
        LaunchTimer lt(ctx, K_GATHER);
        hipLaunchKernelGGL(k_gather, dim3(ntiles), dim3(64), 0, s, ctx->scratch.as<SiteRec>(),
                           ctx->tile_cnt.as<uint32_t>(), ctx->tile_base.as<uint32_t>(), ctx->d_state,
                           ctx->out_pos.as<int32_t>(), ctx->out_pdr.as<float>(), ctx->out_nc.as<uint32_t>(),
                           ctx->out_nd.as<uint32_t>());
    }
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

}  // namespace mth
