// mth_pdr_lpmd.hip -- fused PDR + LPMD pass for gfx950 (CDNA4, wave64).
//
// What it computes (reference: src/pdr.rs:119-212, src/lpmd.rs:154-202, src/readutil.rs:134-145,
// 166-224): per CpG site the number of concordant / discordant reads covering it, and genome-wide
// concordant / discordant CpG-pair counts inside a query-distance window.
//
// How (MI355X-first, nothing like the reference's hash maps):
//   k_build_index   one thread per read: a linear index "first read starting at or after q*256 bp"
//                   (reads are coordinate sorted) + sortedness validation.
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
    uint32_t n_reads, n_cpgs;
    uint32_t min_cov;      // max(pdr_min_depth, 1)
    uint32_t min_cpgs;
    int32_t  min_dist, max_dist;
    uint8_t  pdr_min_qual, lpmd_min_qual, want_pdr, want_lpmd;
};

// ---------------------------------------------------------------------------------------------
// idx[q] = first read i with read_start[i] >= idx_base + q*IDX_Q   (q = 0..nq)
// One thread handles 4 consecutive reads (one 16-byte load + the element before them); the thread
// whose group contains index n_reads also plays the sentinel that closes the index.
__global__ __launch_bounds__(BLOCK) void k_build_index(const int32_t *__restrict__ read_start,
                                                       uint32_t n_reads, int32_t idx_base,
                                                       uint32_t nq, int aligned16,
                                                       uint32_t *__restrict__ idx,
                                                       DevState *__restrict__ st) {
    const uint32_t i0 = (blockIdx.x * BLOCK + threadIdx.x) * 4u;
    if (i0 > n_reads) return;
    auto bucket = [&](int32_t s) -> int64_t {  // floor((s-base)/Q), -1 below the base
        const int64_t d = (int64_t)s - idx_base;
        return d < 0 ? -1 : (d >> IDX_QSHIFT);
    };
    int32_t sv[4];
    if (aligned16 && i0 + 4 <= n_reads) {
        const int4 x = *reinterpret_cast<const int4 *>(read_start + i0);
        sv[0] = x.x; sv[1] = x.y; sv[2] = x.z; sv[3] = x.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[k] = (i0 + k < n_reads) ? read_start[i0 + k] : 0;
    }
    int64_t g_prev = -1;
    int32_t s_prev = 0;
    bool have_prev = false;
    if (i0 > 0) { s_prev = read_start[i0 - 1]; g_prev = bucket(s_prev); have_prev = true; }
    uint32_t err = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = i0 + k;
        if (i > n_reads) break;
        int64_t g_cur;
        if (i < n_reads) {
            const int32_t s = sv[k];
            g_cur = bucket(s);
            if (have_prev && s < s_prev) err |= ERRB_UNSORTED;
            s_prev = s; have_prev = true;
        } else {
            g_cur = nq;   // sentinel closes the index
        }
        if (g_cur > (int64_t)nq) g_cur = nq;
        for (int64_t q = g_prev + 1; q <= g_cur; ++q) idx[q] = i;
        if (g_cur > g_prev) g_prev = g_cur;
    }
    if (err) atomicOr(&st->err, err);
}

// ---------------------------------------------------------------------------------------------
// DPP controls (gfx9 family): no LDS traffic, one VALU per step
#define MTH_DPP(v, ctrl, rmask, bctl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, (bctl)))
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {   // wave-uniform result
    v += MTH_DPP(v, 0xb1 /*quad_perm [1,0,3,2]*/, 0xf, true);
    v += MTH_DPP(v, 0x4e /*quad_perm [2,3,0,1]*/, 0xf, true);
    v += MTH_DPP(v, 0x141 /*row_half_mirror*/, 0xf, true);
    v += MTH_DPP(v, 0x140 /*row_mirror*/, 0xf, true);          // every lane: sum of its row of 16
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += MTH_DPP(v, 0x111 /*row_shr:1*/, 0xf, true);
    v += MTH_DPP(v, 0x112 /*row_shr:2*/, 0xf, true);
    v += MTH_DPP(v, 0x114 /*row_shr:4*/, 0xf, true);
    v += MTH_DPP(v, 0x118 /*row_shr:8*/, 0xf, true);
    v += MTH_DPP(v, 0x142 /*row_bcast:15*/, 0xa, false);
    v += MTH_DPP(v, 0x143 /*row_bcast:31*/, 0xc, false);
    return v;
}

template <int W, int B>
__device__ __forceinline__ void tile_epilogue(const TileArgs &a, const uint32_t t, const int32_t T0,
                                              uint32_t *cnt, uint32_t (*red)[B / 64], uint32_t *wave_off,
                                              uint32_t lp_c, uint32_t lp_d, uint32_t n_read,
                                              uint32_t n_valid, uint32_t bad) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (bad) atomicOr(const_cast<uint32_t *>(&a.st->err), (uint32_t)ERRB_SPAN);

    // per-tile LPMD partials (wave DPP reduce -> LDS -> one plain store per tile)
    if (a.want_lpmd) {
        const uint32_t r0 = wave_sum(lp_c), r1 = wave_sum(lp_d), r2 = wave_sum(n_read), r3 = wave_sum(n_valid);
        if (lane == 0) { red[0][wave] = r0; red[1][wave] = r1; red[2][wave] = r2; red[3][wave] = r3; }
    }
    __syncthreads();
    if (a.want_lpmd && tid < 4) {
        uint32_t s = 0;
        for (int w = 0; w < B / 64; ++w) s += red[tid][w];
        a.tile_lpmd[t * 4 + tid] = s;
    }
    if (!a.want_pdr) { if (tid == 0) a.tile_cnt[t] = 0; return; }

    // compaction: thread owns PER consecutive positions; emit sites with coverage >= min_cov
    constexpr int PER = W / B;
    static_assert(PER % 4 == 0, "uint4 LDS reads");
    uint32_t c[PER], d[PER];
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
        const uint4 x = reinterpret_cast<const uint4 *>(cnt)[tid * (PER / 4) + q];
        const uint4 y = reinterpret_cast<const uint4 *>(cnt + W)[tid * (PER / 4) + q];
        c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
        d[4 * q] = y.x; d[4 * q + 1] = y.y; d[4 * q + 2] = y.z; d[4 * q + 3] = y.w;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) mine += (c[q] + d[q] >= a.min_cov) ? 1u : 0u;
    const uint32_t incl = wave_scan_incl(mine);
    if (lane == 63) wave_off[wave + 1] = incl;
    __syncthreads();
    if (tid == 0) {
        wave_off[0] = 0;
        for (int w = 1; w <= B / 64; ++w) wave_off[w] += wave_off[w - 1];
        a.tile_cnt[t] = wave_off[B / 64];
    }
    __syncthreads();
    uint32_t o = wave_off[wave] + incl - mine;
    SiteRec *__restrict__ out = a.scratch + (size_t)t * W;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (c[q] + d[q] >= a.min_cov) {
            SiteRec rr; rr.pos = T0 + tid * PER + q; rr.n_conc = c[q]; rr.n_disc = d[q]; rr.pad = 0;
            out[o++] = rr;
        }
    }
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
// NB relative positions of a read with one load (global memory takes unaligned vector loads)
template <typename RelT, int NB>
__device__ __forceinline__ void load_rel(const RelT *__restrict__ rp, int32_t (&r)[NB]) {
    if constexpr (sizeof(RelT) == 1) {
#pragma unroll
        for (int k8 = 0; k8 < NB / 8; ++k8) {
            const u32x2_a1 x = *reinterpret_cast<const u32x2_a1 *>(rp + 8 * k8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { r[8 * k8 + k] = (int32_t)((x.x >> (8 * k)) & 0xffu); r[8 * k8 + 4 + k] = (int32_t)((x.y >> (8 * k)) & 0xffu); }
        }
    } else {
#pragma unroll
        for (int k8 = 0; k8 < NB / 8; ++k8) {
            const u32x4_a2 x = *reinterpret_cast<const u32x4_a2 *>(rp + 8 * k8);
            r[8 * k8] = (int32_t)(x.x & 0xffffu); r[8 * k8 + 1] = (int32_t)(x.x >> 16);
            r[8 * k8 + 2] = (int32_t)(x.y & 0xffffu); r[8 * k8 + 3] = (int32_t)(x.y >> 16);
            r[8 * k8 + 4] = (int32_t)(x.z & 0xffffu); r[8 * k8 + 5] = (int32_t)(x.z >> 16);
            r[8 * k8 + 6] = (int32_t)(x.w & 0xffffu); r[8 * k8 + 7] = (int32_t)(x.w >> 16);
        }
    }
}

// Tile kernel.  W = reference positions per tile, B = threads per workgroup, NB = CpG calls of a
// read held in registers (reads with more calls take the memory loop for the tail).
//
// Latency structure (what v1 got wrong: one dependent HBM round trip per call): per tile the
// dependent chain is  idx -> read fields -> ALL calls of the read (NB independent loads in
// flight) -> LDS atomics.  Blocks are mapped to tiles XCD-aware so the halo reads of neighbouring
// tiles are served by the same L2.
template <int W, int B, int NB, typename RelT>
__global__ __launch_bounds__(B) void k_pdr_lpmd_tile(const TileArgs a, const uint32_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t cnt[2 * W + B];  // [0,W): concordant, [W,2W): discordant, then one trash word per thread
    __shared__ uint32_t red[4][B / 64];
    __shared__ uint32_t wave_off[B / 64 + 1];

    const int tid = threadIdx.x;
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    const int32_t T0 = a.region_beg + (int32_t)(t * W);
    const int32_t T1 = min(T0 + W, a.region_end);
    const uint32_t Wt = (uint32_t)(T1 - T0);

    // candidate reads: start in [T0 - max_span + 1, T0 + W]  (a call sits in [start-1, end]).
    // Both bounds are clamped to n_reads: a batch that failed validation in k_build_index (stale or
    // partial index) then only ever touches in-bounds reads, and its rows are discarded because the
    // getters report the error.  (An explicit load of the error flag here cost every tile a dependent
    // round trip before its first useful load.)
    const uint32_t lo = min(a.idx[(uint32_t)(T0 - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
    const uint32_t hi = min(a.idx[((uint32_t)(T0 + W - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    for (int i = tid; i < 2 * W / 4; i += B)
        reinterpret_cast<uint4 *>(cnt)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    uint32_t lp_c = 0, lp_d = 0, n_read = 0, n_valid = 0, bad = 0;
    for (uint32_t i = lo + tid; i < hi; i += B) {
        const int32_t s = a.read_start[i];
        const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
        const uint8_t mq = a.read_mapq[i];
        const uint32_t n = o1 - o0;
        const bool owned = (s >= T0) && (s < T1);
        // lpmd.rs:176-179
        const bool lp_ok = a.want_lpmd && owned && (mq >= a.lpmd_min_qual);
        if (a.want_lpmd && owned) { n_read += 1; n_valid += lp_ok ? 1u : 0u; }
        // pdr.rs:147-157
        const bool pdr_ok = a.want_pdr && (n >= a.min_cpgs) && (mq >= a.pdr_min_qual) && (n > 0);
        if (!(lp_ok || pdr_ok) || n == 0) continue;

        // All calls of the read in flight at once: two 16-byte loads from a per-read base (dword alignment
        // is all global_load_dwordx4 needs) and one 8/16-byte load of the relative positions.  Slots k >= n
        // read the NEXT reads' calls and are neutralised below; only the batch's last few reads could run
        // past the end of the arrays, and those take the clamped form (wave-uniform choice).
        uint32_t v[NB];
        int32_t r[NB];
        const uint32_t *__restrict__ cp = a.cpg_pos + o0;
        const RelT *__restrict__ rp = rel + o0;
        // (distances between live calls are < 2^16, so capping max_distance keeps dead-slot differences outside)
        const int32_t maxd = min(a.max_dist, 1 << 20);
        const bool any_lp = maxd >= a.min_dist && __any(lp_ok && n > 1);   // min > max: no pair can qualify (and the range trick below would wrap)
        if (!__any(o0 + (uint32_t)NB > a.n_cpgs)) {
#pragma unroll
            for (int k4 = 0; k4 < NB / 4; ++k4) {
                const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(cp + 4 * k4);
                v[4 * k4] = x.x; v[4 * k4 + 1] = x.y; v[4 * k4 + 2] = x.z; v[4 * k4 + 3] = x.w;
            }
            if (any_lp) load_rel<RelT, NB>(rp, r);
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) v[k] = cp[min((uint32_t)k, n - 1)];
            if (any_lp) {
#pragma unroll
                for (int k = 0; k < NB; ++k) r[k] = (int32_t)rp[min((uint32_t)k, n - 1)];
            }
        }
        // Every instruction type issues from the same few waves here (profiles/r01_tile_variants.md: the
        // kernel is issue-bound), and predicates that are AND-ed / OR-ed per slot become s_and_b64 /
        // s_or_b64 / saveexec chains on the scalar unit.  So the liveness of a slot (k < n) is used ONCE,
        // to neutralise dead slots, and everything after is plain integer arithmetic:
        //   dead call word  = far position (never inside a tile) with the first call's state (concordant)
        //   dead rel        = (k+1) << 24 (any difference involving it exceeds every max_distance)
        // Span check (every call in [start-1, start+max_span-1] -- this is what makes the halo complete,
        // checked on the calls themselves instead of trusting read_end): max over the live slots.
        const uint32_t sm1 = (uint32_t)(s - 1);
        const uint32_t dead_w = 0x7fffffffu | (v[0] & 0x80000000u);
        const uint32_t n_lp = (lp_ok && n <= (uint32_t)NB) ? n : 0u;   // pairs evaluated from the registers
        uint32_t acc = 0, xmax = (v[0] & 0x7fffffffu) - sm1;
#pragma unroll
        for (int k = 1; k < NB; ++k) {
            const bool live = (uint32_t)k < n;
            const uint32_t x = (v[k] & 0x7fffffffu) - sm1;
            xmax = max(xmax, live ? x : 0u);
            v[k] = live ? v[k] : dead_w;
            acc |= v[k] ^ v[0];
        }
        if (any_lp) {
#pragma unroll
            for (int k = 0; k < NB; ++k) r[k] = ((uint32_t)k < n_lp) ? r[k] : (int32_t)((k + 1) << 24);
        }
        bad |= (xmax > (uint32_t)a.max_span) ? 1u : 0u;
        uint32_t disc = acc >> 31;
        if (n > (uint32_t)NB) {
            const uint32_t first = v[0] >> 31;
            for (uint32_t k = NB; k < n; ++k) {
                const uint32_t x = a.cpg_pos[o0 + k];
                disc |= (x >> 31) ^ first;
                bad |= ((x & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
            }
        }
        // windowed pair counts (readutil.rs:166-224): pairs (j<k) with min <= rel_k - rel_j <= max.
        // The calls are sorted by relpos, so the distance at call-index gap g+1 is >= the distance at gap g:
        // walk the pair matrix by diagonals g = 1, 2, .. and stop once NO lane of the wave has a pair
        // within max_distance on the current diagonal (wave-uniform break).
        if (any_lp) {
            const uint32_t span_ok = (uint32_t)(maxd - a.min_dist);
            uint32_t lp_n = 0, lp_dd = 0;
#pragma unroll
            for (int g = 1; g < NB; ++g) {
                int32_t dmin = 0x7fffffff;
#pragma unroll
                for (int k = g; k < NB; ++k) {
                    const int32_t dist = r[k] - r[k - g];
                    dmin = min(dmin, dist);
                    const bool in = (uint32_t)(dist - a.min_dist) <= span_ok;      // min <= dist <= max (min <= max)
                    lp_n += in ? 1u : 0u;
                    lp_dd += (in ? (v[k] ^ v[k - g]) : 0u) >> 31;
                }
                if (!__any(dmin <= maxd)) break;
            }
            lp_c += lp_n - lp_dd;
            lp_d += lp_dd;
        }
        if (lp_ok && n > (uint32_t)NB) {   // a read with more than NB calls: memory loop (divergent, rare)
            for (uint32_t k = 1; k < n; ++k) {
                const int32_t rk = (int32_t)rel[o0 + k];
                const uint32_t mk = a.cpg_pos[o0 + k] >> 31;
                for (uint32_t j = k; j-- > 0;) {
                    const int32_t dist = rk - (int32_t)rel[o0 + j];
                    if (dist > a.max_dist) break;          // readutil.rs:184 (anchors evicted)
                    if (dist < a.min_dist) continue;       // readutil.rs:196
                    if ((a.cpg_pos[o0 + j] >> 31) == mk) lp_c += 1; else lp_d += 1;
                }
            }
        }
        // scatter +1 to the tile's sites (pdr.rs:180-191), branch-free: a slot that is dead, outside the tile
        // or belongs to a read PDR skips adds into the thread's own trash word instead (no exec juggling)
        {
            const uint32_t wt = pdr_ok ? Wt : 0u;
            const uint32_t dw = disc ? (uint32_t)W : 0u;
            const uint32_t trash = (uint32_t)(2 * W + tid) - dw;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const uint32_t pk = (v[k] & 0x7fffffffu) - (uint32_t)T0;
                atomicAdd(cnt + ((pk < wt ? pk : trash) + dw), 1u);
            }
            if (pdr_ok) {
                for (uint32_t k = NB; k < n; ++k) {
                    const uint32_t pk = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)T0;
                    if (pk < Wt) atomicAdd(cnt + dw + pk, 1u);
                }
            }
        }
    }
    tile_epilogue<W, B>(a, t, T0, cnt, red, wave_off, lp_c, lp_d, n_read, n_valid, bad);
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative tile kernel -- EXPERIMENTAL, NOT the default (select with MTH_TILE_VARIANT=4..6).
// History: its first form (v3) gave every call its read with a 6-step cross-lane max-scan and measured
// SLOWER than the lane = read kernel above (tile kernel 0.301 vs 0.237 ms, profiles/r01_tile_variants.md).
// This form (v4) keeps the structure but finds segment heads with one ballot + count-leading-zeros.
// Idea: the lane = read kernel pads every read to NB call slots and NB*(NB-1)/2 pair slots (rocprof:
// ~1650 VALU wave-instructions per wave and tile, ~3 of 8 slots useful).  Here a wave takes a chunk of
// up to 64 consecutive reads (lane = read for the 9 B/read record) and then walks the chunk's CONTIGUOUS
// call range with lane = call: coalesced loads, no padding.
//   read of a call   : each read lane drops (lane+1 | flags) at its first call in a per-wave LDS u16
//                      table; a round loads the 64 entries, ballot(entry != 0) is the head mask, the
//                      nearest head at or before lane t is 63 - clz(mask & low_bits(t)), its entry comes
//                      with one bpermute; "distance to head" makes same-read tests a compare
//   read concordance : a read is discordant iff two ADJACENT calls differ (readutil.rs:134-145) ->
//                      one lane shift + a (rare) LDS atomic-or on the read's flag word
//   LPMD pairs       : call k looks back over calls k-1, k-2, .. of the same read while the query
//                      distance stays <= max (readutil.rs:166-224) -- lane shifts, work proportional
//                      to the pairs that exist; rounds overlap by OV lanes so no carry is needed;
//                      look-backs deeper than OV finish with a per-lane memory loop
//   PDR scatter      : second walk over the calls held in registers, LDS atomic add on the dense
//                      tile counters
constexpr int WC_MAXC = 1024;  // calls per chunk covered by the per-wave head table
constexpr int WC_OV = 8;       // context lanes per round (look-back reach without touching memory)
constexpr int WC_RQ = 8;       // rounds whose calls stay in registers for the scatter walk

template <int W, int NW, typename RelT, bool WANT_PDR, bool WANT_LPMD>
__global__ __launch_bounds__(NW * 64) void k_pdr_lpmd_tile_wc(const TileArgs a, const uint32_t ntiles) {
    constexpr int B = NW * 64;
    __shared__ __attribute__((aligned(16))) uint32_t cnt[2 * W];
    __shared__ uint32_t red[4][B / 64];
    __shared__ uint32_t wave_off[B / 64 + 1];
    __shared__ __attribute__((aligned(16))) uint16_t head_s[NW][WC_MAXC];  // 0 = not a first call, else (read lane + 1)
    __shared__ uint32_t rinfo_s[NW][64];                                    // bit0 pdr_ok, bit1 lp_ok, bit2 discordant

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const int32_t T0 = a.region_beg + (int32_t)(t * W);
    const int32_t T1 = min(T0 + W, a.region_end);
    const uint32_t Wt = (uint32_t)(T1 - T0);
    uint16_t *head = head_s[wave];
    uint32_t *rinfo = rinfo_s[wave];

    for (int i = tid; i < 2 * W / 4; i += B)
        reinterpret_cast<uint4 *>(cnt)[i] = make_uint4(0, 0, 0, 0);
    // candidate reads: start in [T0 - max_span + 1, T0 + W].  idx is clamped so that a batch that
    // failed validation (stale index) only ever produces in-bounds reads; its rows are discarded
    // because the error flag is reported by the getters.
    const uint32_t lo = min(a.idx[(uint32_t)(T0 - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
    const uint32_t hi = min(a.idx[((uint32_t)(T0 + W - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    __syncthreads();

    uint32_t lp_c = 0, lp_d = 0, n_read = 0, n_valid = 0, bad = 0;
    const uint32_t span = hi > lo ? hi - lo : 0;
    const uint32_t per = (span + NW - 1) / NW;
    const uint32_t r_end = min(lo + (wave + 1) * per, hi);
    for (uint32_t i0 = lo + wave * per; i0 < r_end;) {
        // ---- read round: lane <-> read i0 + lane ------------------------------------------
        const uint32_t i = i0 + lane;
        const bool inb = i < r_end;
        const uint32_t off0 = a.cpg_off[inb ? i : i0], off1 = a.cpg_off[(inb ? i : i0) + 1];
        const int32_t s = a.read_start[inb ? i : i0];
        const uint32_t mq = a.read_mapq[inb ? i : i0];
        const uint32_t cbeg = __builtin_amdgcn_readfirstlane(off0);
        // reads of the chunk: the longest prefix whose calls fit the head table
        const unsigned long long fits = __ballot(inb && (off1 - cbeg <= (uint32_t)WC_MAXC));
        const int nr = fits == ~0ull ? 64 : __builtin_ctzll(~fits);
        if (nr == 0) {
            // one read with more than WC_MAXC calls (only possible with 16-bit relpos): the wave
            // walks its calls from memory, lane-strided
            const uint32_t e1 = __builtin_amdgcn_readfirstlane(off1);
            const int32_t s0 = __builtin_amdgcn_readfirstlane(s);
            const uint32_t mq0 = __builtin_amdgcn_readfirstlane(mq);
            const uint32_t n = e1 - cbeg;
            const bool owned = (s0 >= T0) && (s0 < T1);
            const bool lp_ok = WANT_LPMD && owned && (mq0 >= a.lpmd_min_qual);
            const bool pdr_ok = WANT_PDR && (n >= a.min_cpgs) && (mq0 >= a.pdr_min_qual);
            if (WANT_LPMD && owned && lane == 0) { n_read += 1; n_valid += lp_ok ? 1u : 0u; }
            if (lp_ok || pdr_ok) {
                uint32_t dsc = 0;
                for (uint32_t c = cbeg + lane; c < e1; c += 64) {
                    const uint32_t x = a.cpg_pos[c];
                    bad |= ((x & 0x7fffffffu) - (uint32_t)(s0 - 1) > (uint32_t)a.max_span) ? 1u : 0u;
                    if (c > cbeg) dsc |= (x ^ a.cpg_pos[c - 1]) >> 31;
                    if (lp_ok) {
                        const int32_t rk = (int32_t)rel[c];
                        for (uint32_t j = c; j-- > cbeg;) {
                            const int32_t dist = rk - (int32_t)rel[j];
                            if (dist > a.max_dist) break;
                            if (dist < a.min_dist) continue;
                            if ((a.cpg_pos[j] >> 31) == (x >> 31)) lp_c += 1; else lp_d += 1;
                        }
                    }
                }
                const bool disc = __ballot(dsc != 0) != 0ull;
                if (pdr_ok) {
                    uint32_t *base = cnt + (disc ? W : 0);
                    for (uint32_t c = cbeg + lane; c < e1; c += 64) {
                        const uint32_t p = (a.cpg_pos[c] & 0x7fffffffu) - (uint32_t)T0;
                        if (p < Wt) atomicAdd(base + p, 1u);
                    }
                }
            }
            i0 += 1;
            continue;
        }
        const bool act = lane < nr;
        const uint32_t cend = __builtin_amdgcn_readlane(off1, nr - 1);
        const uint32_t n = off1 - off0;
        const bool owned = act && (s >= T0) && (s < T1);
        const bool lp_ok = WANT_LPMD && owned && (mq >= a.lpmd_min_qual);                      // lpmd.rs:176-179
        const bool pdr_ok = WANT_PDR && act && (n >= a.min_cpgs) && (mq >= a.pdr_min_qual) && (n > 0);  // pdr.rs:147-157
        if (WANT_LPMD && owned) { n_read += 1; n_valid += lp_ok ? 1u : 0u; }
        if (!__any((lp_ok || pdr_ok) && n > 0)) { i0 += nr; continue; }   // nothing to do (halo chunk)

        // ---- publish the chunk's read table ------------------------------------------------
        reinterpret_cast<uint4 *>(head)[lane] = make_uint4(0, 0, 0, 0);            // 2 x 64 x 16 B = WC_MAXC u16
        reinterpret_cast<uint4 *>(head)[lane + 64] = make_uint4(0, 0, 0, 0);
        rinfo[lane] = (pdr_ok ? 1u : 0u) | (lp_ok ? 2u : 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (act && n > 0) head[off0 - cbeg] = (uint16_t)(lane + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- call rounds: lane <-> call, rounds overlap by WC_OV context lanes -----------------
        // one round = 64 consecutive calls starting at `base`; returns the call word and its read id
        uint32_t carry = 0;   // read id (lane+1) of the call just before the round's lane 0
        auto call_round = [&](const uint32_t base, uint32_t &v_out, uint32_t &rid_out) {
            const uint32_t c = base + lane;
            const bool valid = (int32_t)(c - cbeg) >= 0 && c < cend;
            const bool isnew = valid && lane >= WC_OV;    // lanes < WC_OV only give context
            const uint32_t v = valid ? a.cpg_pos[c] : 0u;
            const uint32_t rl = (valid && WANT_LPMD) ? (uint32_t)rel[c] : 0u;
            const uint32_t h = valid ? (uint32_t)head[c - cbeg] : 0u;
            // segment heads: one ballot, then count-leading-zeros on the heads at or before this lane
            const unsigned long long hm = __ballot(h != 0u);
            const unsigned long long below = hm & ((2ull << lane) - 1ull);
            const int hp = below ? 63 - __builtin_clzll(below) : 0;
            const uint32_t hv = __shfl(h, hp, 64);
            const uint32_t rid = valid ? (below ? hv : carry) : 0u;
            const uint32_t dh = below ? (uint32_t)(lane - hp) : 255u;       // lanes since the head (255: head before the round)
            carry = __builtin_amdgcn_readlane(rid, 63 - WC_OV);   // read of the call before the next round's lane 0
            const int rlane = rid ? (int)rid - 1 : 0;
            const uint32_t m = v >> 31;
            // every call must lie in [start-1, start-1+max_span] (halo completeness)
            const int32_t sR = __shfl(s, rlane, 64);
            bad |= (valid && ((v & 0x7fffffffu) - (uint32_t)(sR - 1) > (uint32_t)a.max_span)) ? 1u : 0u;
            const uint32_t info = valid ? rinfo[rlane] : 0u;
            const uint32_t pk = (m << 16) | (rl & 0xffffu);
            // adjacent calls of one read that differ make the read discordant
            const uint32_t p1 = __shfl_up(pk, 1, 64);
            if (WANT_PDR) {
                if (isnew && dh >= 1u && ((p1 >> 16) != m) && (info & 1u))
                    atomicOr(&rinfo[rlane], 4u);
            }
            if (WANT_LPMD) {
                const bool lpc = isnew && (info & 2u);
                bool deeper = false;
#pragma unroll
                for (int d = 1; d <= WC_OV; ++d) {
                    const uint32_t pd = d == 1 ? p1 : __shfl_up(pk, d, 64);
                    const int32_t dist = (int32_t)rl - (int32_t)(pd & 0xffffu);
                    const bool on = lpc && dh >= (uint32_t)d && dist <= a.max_dist;   // same read; readutil.rs:184
                    if (!__any(on)) break;
                    const bool in = on && dist >= a.min_dist;                         // readutil.rs:196
                    const bool sm = (pd >> 16) == m;
                    lp_c += (in && sm) ? 1u : 0u;
                    lp_d += (in && !sm) ? 1u : 0u;
                    if (d == WC_OV) deeper = on;
                }
                // look-backs that outran the register window (dense CpGs or a wide --max-distance)
                if (__any(deeper)) {
                    const uint32_t o0r = __shfl(off0, rlane, 64);
                    if (deeper) {
                        for (uint32_t j = c - WC_OV; j-- > o0r;) {
                            const int32_t dist = (int32_t)rl - (int32_t)rel[j];
                            if (dist > a.max_dist) break;
                            if (dist < a.min_dist) continue;
                            if ((a.cpg_pos[j] >> 31) == m) lp_c += 1; else lp_d += 1;
                        }
                    }
                }
            }
            v_out = isnew ? v : 0u;
            rid_out = isnew ? rid : 0u;
        };
        uint32_t vq[WC_RQ], rq[WC_RQ];   // statically indexed: stay in VGPRs
        uint32_t base = cbeg - WC_OV;
        bool more = true;                 // wave-uniform
#pragma unroll
        for (int k = 0; k < WC_RQ; ++k) {
            vq[k] = 0u; rq[k] = 0u;
            if (more) {
                call_round(base, vq[k], rq[k]);
                more = base + 64 < cend;
                base += 64 - WC_OV;
            }
        }
        const uint32_t tail_base = base;
        const uint32_t tail_carry = carry;
        while (more) {                    // chunks with more than WC_RQ rounds: nothing kept
            uint32_t v_, r_;
            call_round(base, v_, r_);
            more = base + 64 < cend;
            base += 64 - WC_OV;
        }
        // ---- scatter walk: +1 on the tile's dense counters (pdr.rs:180-191) ----------------------
        if (WANT_PDR) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < WC_RQ; ++k) {
                const uint32_t rid = rq[k];
                if (rid) {
                    const uint32_t info = rinfo[rid - 1];
                    const uint32_t p = (vq[k] & 0x7fffffffu) - (uint32_t)T0;
                    if ((info & 1u) && p < Wt) atomicAdd(cnt + ((info & 4u) ? W : 0) + p, 1u);
                }
            }
            // rounds beyond the register window: re-derive the read of each call from the table
            if (tail_base + WC_OV < cend) {
                uint32_t cr = tail_carry;
                for (uint32_t b2 = tail_base;; b2 += 64 - WC_OV) {
                    const uint32_t c = b2 + lane;
                    const bool valid = (int32_t)(c - cbeg) >= 0 && c < cend;
                    const uint32_t h = valid ? (uint32_t)head[c - cbeg] : 0u;
                    const unsigned long long hm = __ballot(h != 0u);
                    const unsigned long long below = hm & ((2ull << lane) - 1ull);
                    const int hp = below ? 63 - __builtin_clzll(below) : 0;
                    const uint32_t hv = __shfl(h, hp, 64);
                    const uint32_t rid = valid ? (below ? hv : cr) : 0u;
                    cr = __builtin_amdgcn_readlane(rid, 63 - WC_OV);
                    if (valid && lane >= WC_OV) {
                        const uint32_t info = rinfo[rid - 1];
                        const uint32_t p = (a.cpg_pos[c] & 0x7fffffffu) - (uint32_t)T0;
                        if ((info & 1u) && p < Wt) atomicAdd(cnt + ((info & 4u) ? W : 0) + p, 1u);
                    }
                    if (b2 + 64 >= cend) break;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        i0 += nr;
    }
    tile_epilogue<W, B>(a, t, T0, cnt, red, wave_off, lp_c, lp_d, n_read, n_valid, bad);
}

// ---------------------------------------------------------------------------------------------
// single workgroup: tile_base = exclusive scan(tile_cnt); LPMD partials -> DevState.
// Chunks of 4096 tiles: each thread takes 4 consecutive tiles with one 16-byte load (tile_cnt is
// hipMalloc-aligned), one block scan per chunk.  (History: 1024-tile chunks measured 0.0216 ms for
// 14 312 tiles; a variant where each thread walked its own contiguous segment serially was slower.)
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
    for (uint32_t b = 0; b < ntiles; b += 4096) {
        const uint32_t i = b + 4u * tid;
        uint32_t v[4] = {0, 0, 0, 0};
        if (i + 4 <= ntiles) {
            const uint4 x = *reinterpret_cast<const uint4 *>(tile_cnt + i);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (i + k < ntiles) ? tile_cnt[i + k] : 0u;
        }
        if (want_lpmd) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i + k < ntiles) {
                    const uint4 l = reinterpret_cast<const uint4 *>(tile_lpmd)[i + k];
                    acc[0] += l.x; acc[1] += l.y; acc[2] += l.z; acc[3] += l.w;
                }
            }
        }
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = mine;
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
        uint32_t run = wsum[wave] + incl - mine;
        if (i + 4 <= ntiles) {
            uint4 o;
            o.x = run; o.y = run + v[0]; o.z = o.y + v[1]; o.w = o.z + v[2];
            *reinterpret_cast<uint4 *>(tile_base + i) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (i + k < ntiles) tile_base[i + k] = run; run += v[k]; }
        }
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
                                               const DevState *__restrict__ st, uint32_t tile_w,
                                               int32_t *__restrict__ out_pos, float *__restrict__ out_pdr,
                                               uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    const uint32_t t = blockIdx.x;
    const uint32_t n = tile_cnt[t];
    const uint64_t base = st->cur_base + tile_base[t];
    const SiteRec *__restrict__ src = scratch + (size_t)t * tile_w;
    for (uint32_t j = threadIdx.x; j < n; j += 64) {
        const SiteRec r = src[j];
        out_pos[base + j] = r.pos;
        out_nc[base + j] = r.n_conc;
        out_nd[base + j] = r.n_disc;
        out_pdr[base + j] = (float)r.n_disc / ((float)r.n_conc + (float)r.n_disc);
    }
}

// ---------------------------------------------------------------------------------------------
template <int W, int B, typename RelT>
static void launch_tile(const TileArgs &a, uint32_t ntiles, hipStream_t s) {
    const uint32_t grid = ((ntiles + 7) / 8) * 8;   // whole rows of 8 XCDs (remap in the kernel)
    hipLaunchKernelGGL((k_pdr_lpmd_tile<W, B, 8, RelT>), dim3(grid), dim3(B), 0, s, a, ntiles);
}

template <int W, int NW, typename RelT>
static void launch_tile_wc(const TileArgs &a, uint32_t ntiles, hipStream_t s) {
    if (a.want_pdr && a.want_lpmd)
        hipLaunchKernelGGL((k_pdr_lpmd_tile_wc<W, NW, RelT, true, true>), dim3(ntiles), dim3(NW * 64), 0, s, a, ntiles);
    else if (a.want_pdr)
        hipLaunchKernelGGL((k_pdr_lpmd_tile_wc<W, NW, RelT, true, false>), dim3(ntiles), dim3(NW * 64), 0, s, a, ntiles);
    else
        hipLaunchKernelGGL((k_pdr_lpmd_tile_wc<W, NW, RelT, false, true>), dim3(ntiles), dim3(NW * 64), 0, s, a, ntiles);
}

int launch_pdr_lpmd(mth_ctx *ctx, const mth_batch_t &b, const mth_pdr_lpmd_params_t &p, const TileSink *sink) {
    hipStream_t s = ctx->stream;
    // where the compacted rows and their counters go: the PDR result columns by default, or a
    // caller-supplied sink (site discovery for the site-walk measures)
    DevState *cst = sink ? sink->st : ctx->d_state;
    uint32_t *bcnt = sink ? sink->batch_cnt : ctx->batch_cnt.as<uint32_t>();
    int32_t *o_pos = sink ? sink->pos : ctx->out_pos.as<int32_t>();
    float *o_pdr = sink ? sink->pdr : ctx->out_pdr.as<float>();
    uint32_t *o_nc = sink ? sink->nc : ctx->out_nc.as<uint32_t>();
    uint32_t *o_nd = sink ? sink->nd : ctx->out_nd.as<uint32_t>();
    const int variant = ctx->tile_variant;
    const int tile_w = (variant == 0 || variant == 4 || variant == 5) ? 4096 : (variant == 3 ? 1024 : 2048);
    const int64_t region_len = (int64_t)b.region_end - b.region_beg;
    const uint32_t ntiles = (uint32_t)((region_len + tile_w - 1) / tile_w);
    if (ntiles == 0) return MTH_OK;
    // index origin: a whole number of quanta below the region so that halo reads are indexed
    const int32_t ext = ((b.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    const int32_t idx_base = b.region_beg - ext;
    const uint32_t nq = (uint32_t)(((int64_t)ntiles * tile_w + ext) >> IDX_QSHIFT) + 2;

    MTH_HIP(ctx, ctx->idx.reserve((size_t)(nq + 1) * 4, s));
    MTH_HIP(ctx, ctx->tile_cnt.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_base.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_lpmd.reserve((size_t)ntiles * 16, s));
    if (p.want_pdr) MTH_HIP(ctx, ctx->scratch.reserve((size_t)ntiles * tile_w * sizeof(SiteRec), s));

    {
        LaunchTimer lt(ctx, K_INDEX);
        const uint32_t nb = (b.n_reads / 4 + 1 + BLOCK - 1) / BLOCK;
        hipLaunchKernelGGL(k_build_index, dim3(nb), dim3(BLOCK), 0, s, b.read_start,
                           b.n_reads, idx_base, nq, (int)((reinterpret_cast<uintptr_t>(b.read_start) & 15u) == 0),
                           ctx->idx.as<uint32_t>(), ctx->d_state);
    }
    TileArgs a;
    a.read_start = b.read_start; a.read_mapq = b.read_mapq; a.cpg_off = b.cpg_off; a.cpg_pos = b.cpg_pos;
    a.cpg_rel = b.cpg_rel ? (const void *)b.cpg_rel : (const void *)b.cpg_rel16;
    a.idx = ctx->idx.as<uint32_t>(); a.st = ctx->d_state;
    a.tile_cnt = ctx->tile_cnt.as<uint32_t>(); a.tile_lpmd = ctx->tile_lpmd.as<uint32_t>();
    a.scratch = ctx->scratch.as<SiteRec>();
    a.region_beg = b.region_beg; a.region_end = b.region_end; a.idx_base = idx_base; a.max_span = b.max_span;
    a.n_reads = b.n_reads; a.n_cpgs = b.n_cpgs;
    a.min_cov = p.pdr_min_depth > 1 ? p.pdr_min_depth : 1;
    a.min_cpgs = p.pdr_min_cpgs;
    a.min_dist = p.lpmd_min_distance; a.max_dist = p.lpmd_max_distance;
    a.pdr_min_qual = p.pdr_min_qual; a.lpmd_min_qual = p.lpmd_min_qual;
    a.want_pdr = p.want_pdr; a.want_lpmd = p.want_lpmd;
    {
        LaunchTimer lt(ctx, K_TILE);
        const bool r8 = b.cpg_rel != nullptr;
        switch (variant) {
            case 0: r8 ? launch_tile<4096, 256, uint8_t>(a, ntiles, s) : launch_tile<4096, 256, uint16_t>(a, ntiles, s); break;
            case 2: r8 ? launch_tile<2048, 256, uint8_t>(a, ntiles, s) : launch_tile<2048, 256, uint16_t>(a, ntiles, s); break;
            case 3: r8 ? launch_tile<1024, 256, uint8_t>(a, ntiles, s) : launch_tile<1024, 256, uint16_t>(a, ntiles, s); break;
            case 4: r8 ? launch_tile_wc<4096, 4, uint8_t>(a, ntiles, s) : launch_tile_wc<4096, 4, uint16_t>(a, ntiles, s); break;
            case 5: r8 ? launch_tile_wc<4096, 8, uint8_t>(a, ntiles, s) : launch_tile_wc<4096, 8, uint16_t>(a, ntiles, s); break;
            case 6: r8 ? launch_tile_wc<2048, 4, uint8_t>(a, ntiles, s) : launch_tile_wc<2048, 4, uint16_t>(a, ntiles, s); break;
            default: r8 ? launch_tile<2048, 512, uint8_t>(a, ntiles, s) : launch_tile<2048, 512, uint16_t>(a, ntiles, s); break;
        }
    }
    {
        LaunchTimer lt(ctx, K_SCAN);
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, ctx->tile_cnt.as<uint32_t>(),
                           ctx->tile_lpmd.as<uint32_t>(), ntiles, ctx->tile_base.as<uint32_t>(),
                           bcnt, (int)p.want_lpmd, cst);
    }
    if (p.want_pdr) {
        LaunchTimer lt(ctx, K_GATHER);
        hipLaunchKernelGGL(k_gather, dim3(ntiles), dim3(64), 0, s, ctx->scratch.as<SiteRec>(),
                           ctx->tile_cnt.as<uint32_t>(), ctx->tile_base.as<uint32_t>(), cst,
                           (uint32_t)tile_w, o_pos, o_pdr, o_nc, o_nd);
    }
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

}  // namespace mth
