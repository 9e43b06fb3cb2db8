// mth_pdr_lpmd.hip -- fused PDR + LPMD pass for gfx950 (CDNA4, wave64).
//
// What it computes (reference: src/pdr.rs:119-212, src/lpmd.rs:154-202, src/readutil.rs:134-145,
// 166-224): per CpG site the number of concordant / discordant reads covering it, and genome-wide
// concordant / discordant CpG-pair counts inside a query-distance window.
//
// How (MI355X-first, nothing like the reference's hash maps):
//   k_build_index   one thread per read: a linear index "first read starting at or after q*32 bp"
//                   (reads are coordinate sorted) + sortedness validation.
//   k_pdr_lpmd_tile one 256-thread workgroup per 4096-bp tile of the contig.  The tile's site
//                   accumulators are DENSE in LDS (one 32-bit word per reference position holding both
//                   16-bit counts, 16 KiB; tiles with > 65535 candidate reads run two half-tile passes with
//                   32-bit counters), so the scatter is an LDS atomic and HBM only sees the streamed SoA.
//                   Reads that can touch the tile are found through the index (halo reads are re-read by
//                   the neighbour tile).  LPMD pair counts are reduced per wave with DPP and added to
//                   per-256-tile bucket sums (one atomic per counter and tile).  The tile's qualifying
//                   sites are compacted (bit mask + DPP scan) into a per-tile scratch slice.
//   k_gather        one wave per tile: its output base = rows of the 256-tile buckets before it (summed by the
//                   tile kernel with one atomic per tile) + rows of its bucket's earlier tiles; packs the scratch
//                   slice into the final sorted SoA and computes the f32 PDR.  The last tile's wave commits the
//                   batch totals (rows, LPMD counters) to DevState.  (A separate single-workgroup scan kernel
//                   used to sit here: 15 us of latency per batch.)
//
// Roofline: integer streaming + LDS atomics, HBM-bound by design (no MFMA: there is no
// contraction here).  Algorithmic bytes: 16 B/read + 5 B/CpG call in, 12 B/site out.
#include "mth_ctx.h"
#include "mth_tile_dev.h"

#ifndef MTH_TILE_PF
#define MTH_TILE_PF 1
#endif

namespace mth {


// ---------------------------------------------------------------------------------------------
// idx[q] = first read i with read_start[i] >= idx_base + q*IDX_Q   (q = 0..nq)
// One thread handles 4 consecutive reads (one 16-byte load + the element before them); the thread
// whose group contains index n_reads also plays the sentinel that closes the index.
constexpr int IDX_GROUPS = 1;   // groups of 4 reads per thread; 4 (all loads hoisted) measured slower: 0.0177 against 0.0157 ms on config 2.  Under the batch
                                // pipeline 2 looked 1.4 % better in one interleaved A/B (tools/ab_lib.sh) -- and three copies of ONE build differed by up to 3 % in the next: not adopted
// NIDX = 1: the fine index (QSHIFT = IDX_QSHIFT, 32-bp quanta) every tile / site kernel can look any position up in.
// NIDX = 2 (the dense PDR + LPMD tile kernel's own, round 4): that kernel asks two questions per 4096-bp tile only -- the first
// read starting at or after T0 - max_span + 1 and the first one starting after T0 + W -- so two indices with ONE entry per tile
// (quantum = tile width, origins region_beg - max_span + 1 and region_beg + 1) answer them exactly: 2 x 14 312 entries instead of
// 1.83 M on config 2, no store loop for four reads in five, and the tile's candidate range loses the up to 2 x 31 bp of reads the
// 32-bp rounding handed it.
template <int NIDX>
__global__ __launch_bounds__(BLOCK) void k_build_index(const int32_t *__restrict__ read_start,
                                                       uint32_t n_reads, int32_t idx_base, int32_t idx_base2, int qshift,
                                                       uint32_t nq, int aligned16,
                                                       uint32_t *__restrict__ idx, uint32_t *__restrict__ idx2,
                                                       DevState *__restrict__ st, DevState *__restrict__ cst,
                                                       unsigned long long *__restrict__ bucket_sums, uint32_t n_bucket_words,
                                                       const uint32_t *__restrict__ cpg_off, uint32_t n_cpgs) {
    // first kernel of a batch: its rows go after everything emitted so far, and the bucket sums start at zero
    // (nothing else runs between the previous batch's last kernel and this one on the stream)
    const uint32_t gtid = blockIdx.x * BLOCK + threadIdx.x;
    if (gtid == 0 && cst) cst->cur_base = cst->n_sites;      // (pipelined batches: the base travels along the chain of gathers instead)
    for (uint32_t w = gtid; w < n_bucket_words; w += gridDim.x * BLOCK) bucket_sums[w] = 0ull;
    // safe_hi: the largest read index r with cpg_off[r] + 8 <= n_cpgs, searched among the batch's last 256 indices by the
    // first wave (0 if it is not there: every tile then takes the clamped loads).  The tile kernel used to load
    // cpg_off[hi] for this decision: a dependent round trip before its first useful load.
    if (cpg_off && gtid < 64) {
        uint32_t best = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t back = gtid * 4u + k;
            if (back <= n_reads) {
                const uint32_t r = n_reads - back;
                if ((uint64_t)cpg_off[r] + 8u <= (uint64_t)n_cpgs) best = max(best, r);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 64));
        if (gtid == 0) st->safe_hi = best;
    }
    auto bucket = [&](int32_t s, int32_t base) -> int32_t {  // min(floor((s-base)/Q), nq), -1 below the base
        const int64_t d = (int64_t)s - base;
        return d < 0 ? -1 : (int32_t)min(d >> qshift, (int64_t)nq);
    };
    // IDX_GROUPS groups of 4 reads per thread, BLOCK groups apart (coalesced); all their loads are requested before the
    // first is used
    int32_t sv[IDX_GROUPS][4], sp[IDX_GROUPS];
    uint32_t gi[IDX_GROUPS];
#pragma unroll
    for (int u = 0; u < IDX_GROUPS; ++u) {
        const uint32_t i0 = ((blockIdx.x * IDX_GROUPS + u) * BLOCK + threadIdx.x) * 4u;
        gi[u] = i0;
        sp[u] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[u][k] = 0;
        if (i0 <= n_reads) {
            if (aligned16 && i0 + 4 <= n_reads) {
                const int4 x = *reinterpret_cast<const int4 *>(read_start + i0);
                sv[u][0] = x.x; sv[u][1] = x.y; sv[u][2] = x.z; sv[u][3] = x.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[u][k] = (i0 + k < n_reads) ? read_start[i0 + k] : 0;
            }
        }
        // (taking the element before the group from the neighbouring lane -- one load instruction per lane less -- was measured
        // slower: 0.0171 against 0.0163 ms; the load hits the line its neighbour fetches and waits for nothing extra)
        if (i0 > 0 && i0 <= n_reads) sp[u] = read_start[i0 - 1];
    }
    // ---- the fine index (NIDX = 1), usual case: every thread's four reads open at most FAST_SPAN entries between them.  Entry q of
    // (g_prev, g_last] is the first of the four reads whose quantum reaches q: i0 + the number of reads whose quantum lies below q --
    // one loop over the thread's entries, three compares each, 32-bit quantum arithmetic (the values of a valid batch are >= idx_base;
    // anything below it counts as quantum -1, as in the general form).  A wave where some thread spans more (assembly gaps), and
    // NIDX = 2, take the general form below.  (Round 5: the general form alone was 85 % vector-unit bound at WGBS depth -- 0.31 ms
    // per pass on config 3, paid by every measure's pass.)
    if (NIDX == 1 && IDX_GROUPS == 1) {
        constexpr int FAST_SPAN = 16;
        const uint32_t i0 = gi[0];
        const bool gact = i0 <= n_reads;
        auto q32 = [&](const int32_t sx) -> int32_t { return sx < idx_base ? -1 : (int32_t)min(((uint32_t)sx - (uint32_t)idx_base) >> qshift, nq); };
        int32_t g[4];
        uint32_t errf = 0;
        int32_t s_prev = sp[0];
        bool have_prev = gact && i0 > 0;
        int32_t g_prev = have_prev ? q32(s_prev) : -1, g_run = g_prev;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + (uint32_t)k;
            if (gact && i < n_reads) {
                const int32_t sx = sv[0][k];
                if (have_prev && sx < s_prev) errf |= ERRB_UNSORTED;
                s_prev = sx; have_prev = true;
                g_run = max(g_run, q32(sx));                     // (an unsorted batch is an error; the running maximum keeps the stores in range)
            } else if (gact) g_run = (int32_t)nq;               // the sentinel (i == n_reads) and what lies beyond it close the index
            g[k] = g_run;
        }
        const int32_t span = gact ? g[3] - g_prev : 0;
        if (!__any(span > FAST_SPAN)) {
            for (int32_t q = g_prev + 1; q <= g[3]; ++q)
                idx[q] = i0 + (g[0] < q ? 1u : 0u) + (g[1] < q ? 1u : 0u) + (g[2] < q ? 1u : 0u);
            if (errf) atomicOr(&st->err, errf);
            return;
        }
    }
    // The fine quantum is 32 bp (it was 256: the candidates of a tile or a site then carried up to 362 bp of reads that cannot
    // touch it, 8 % of a 4096-bp tile's loop iterations), so a read usually opens an entry or two; a stretch without reads
    // (assembly gaps: megabases) is filled by the whole wave, 64 entries per step, not by the one lane that found it.
    const int lane = threadIdx.x & 63;
    uint32_t err = 0;
#pragma unroll
    for (int u = 0; u < IDX_GROUPS; ++u) {
        const uint32_t i0 = gi[u];
        const bool gact = i0 <= n_reads;
        if (NIDX == 2) {
            // sortedness once, on the values alone (the per-family loops below only run where a boundary is crossed)
            int32_t sprev = sp[u];
            bool hp = gact && i0 > 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (gact && i0 + k < n_reads) { if (hp && sv[u][k] < sprev) err |= ERRB_UNSORTED; sprev = sv[u][k]; hp = true; }
            }
        }
#pragma unroll
        for (int w = 0; w < NIDX; ++w) {
            uint32_t *__restrict__ out = w ? idx2 : idx;
            const int32_t base = w ? idx_base2 : idx_base;
            int32_t g_prev = -1, s_prev = 0;
            bool have_prev = false;
            if (gact && i0 > 0) { s_prev = sp[u]; g_prev = bucket(s_prev, base); have_prev = true; }
            if (NIDX == 2) {
                // tile-granular families: the reads of a wave (256 consecutive ones) cross a boundary of the family in about one
                // wave out of three on config 2 -- the others are done after two bucket computations per lane
                int32_t g_last = g_prev;
                if (gact) {
                    const uint32_t last = min(i0 + 3u, n_reads);            // the group's last index (n_reads: the sentinel)
                    g_last = last == n_reads ? (int32_t)nq : bucket(sv[u][last - i0], base);
                }
                if (!__any(g_last > g_prev)) continue;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = i0 + k;
                const bool act = gact && i <= n_reads;
                int32_t g_cur = g_prev;
                if (act) {
                    if (i < n_reads) {
                        const int32_t s = sv[u][k];
                        g_cur = bucket(s, base);
                        if (NIDX == 1 && have_prev && s < s_prev) err |= ERRB_UNSORTED;
                        s_prev = s; have_prev = true;
                    } else {
                        g_cur = (int32_t)nq;   // sentinel closes the index
                    }
                }
                const bool big = act && g_cur - g_prev > 32;
                if (act && !big) for (int32_t q = g_prev + 1; q <= g_cur; ++q) out[q] = i;
                unsigned long long m = __ballot(big);
                while (m) {
                    const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                    m &= m - 1;
                    const int32_t gp = __builtin_amdgcn_readlane(g_prev, l), gc = __builtin_amdgcn_readlane(g_cur, l);
                    const uint32_t ii = (uint32_t)__builtin_amdgcn_readlane((int)i, l);
                    for (int32_t q = gp + 1 + lane; q <= gc; q += 64) out[q] = ii;
                }
                if (g_cur > g_prev) g_prev = g_cur;
            }
        }
    }
    if (err) atomicOr(&st->err, err);
}

// ---------------------------------------------------------------------------------------------
// LPMD per-tile partials: wave DPP reduce -> LDS -> one atomic per counter into the tile's bucket (256 tiles share
// an address; the per-read counts never touch global memory)
// A wave reduction is ~50 issue cycles (4 DPP adds, 4 readlanes) and a tile is only ~3 reads per lane: the four sums are taken
// as two where their sizes allow it.  SMALL (tiles with <= 65535 candidate reads): a wave's read counts fit 16 bits, so n_read
// and n_valid share a word; the pair counts share one when no lane of the wave has more than 1023 of either (the usual case:
// 64 x 1023 < 2^16), decided by one ballot.
template <int B, bool SMALL>
__device__ __forceinline__ void tile_lpmd_partials(const TileArgs &a, const uint32_t t, uint32_t (*red)[B / 64],
                                                   uint32_t lp_c, uint32_t lp_d, uint32_t n_read, uint32_t n_valid) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t r0, r1, r2, r3;
    if (SMALL) {
        const uint32_t rv = wave_sum(n_read | (n_valid << 16));
        r2 = rv & 0xffffu; r3 = rv >> 16;
        if (!__any((lp_c | lp_d) > 1023u)) {
            const uint32_t cd = wave_sum(lp_c | (lp_d << 16));
            r0 = cd & 0xffffu; r1 = cd >> 16;
        } else { r0 = wave_sum(lp_c); r1 = wave_sum(lp_d); }
    } else { r0 = wave_sum(lp_c); r1 = wave_sum(lp_d); r2 = wave_sum(n_read); r3 = wave_sum(n_valid); }
    if (lane == 0) { red[0][wave] = r0; red[1][wave] = r1; red[2][wave] = r2; red[3][wave] = r3; }
}
// second half, after the workgroup barrier that follows (the one the compaction needs anyway)
template <int B>
__device__ __forceinline__ void tile_lpmd_commit(const TileArgs &a, const uint32_t t, uint32_t (*red)[B / 64]) {
    const int tid = threadIdx.x;
    if (tid < 4) {
        uint32_t s = 0;
        for (int w = 0; w < B / 64; ++w) s += red[tid][w];
        if (s) atomicAdd(a.bucket + a.nbk + (size_t)(t >> TILE_BUCKET_SHIFT) * 4 + tid, (unsigned long long)s);
    }
}

// Compaction of one pass: NPOS = W (packed) or W/2 (wide) positions starting at reference position P0; the
// thread owns PER consecutive positions and emits those with coverage >= min_cov to scratch[out_base..].
// Sites are sparse (a few % of the positions): the thread keeps a bit mask of its qualifying positions and then
// loops over the set bits only, re-reading the counters from LDS (PER predicated store blocks cost PER exec
// save/restore pairs per wave whether or not anything qualifies).  Returns the pass's row count.
template <int W, int B, bool WIDE>
__device__ __forceinline__ uint32_t tile_compact(const TileArgs &a, const uint32_t t, const int32_t P0, const uint32_t Wp,
                                                 const uint32_t *cnt, uint32_t *wave_off, const uint32_t out_base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NPOS = WIDE ? W / 2 : W;
    constexpr int PER = NPOS / B;
    static_assert(PER % 4 == 0 && PER <= 32, "uint4 LDS reads, 32-bit mask");
    uint32_t qual = 0;
    if (WIDE) {
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const uint4 x = reinterpret_cast<const uint4 *>(cnt)[tid * (PER / 4) + q];
            const uint4 y = reinterpret_cast<const uint4 *>(cnt + W / 2)[tid * (PER / 4) + q];
            const uint4 cov = make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
            qual |= (cov.x >= a.min_cov ? 1u : 0u) << (4 * q);
            qual |= (cov.y >= a.min_cov ? 1u : 0u) << (4 * q + 1);
            qual |= (cov.z >= a.min_cov ? 1u : 0u) << (4 * q + 2);
            qual |= (cov.w >= a.min_cov ? 1u : 0u) << (4 * q + 3);
        }
    } else {
        // packed word = coverage | discordant << 16.  Per position: coverage - min_cov (the sign says "below"), shifted into the
        // mask from the right by one funnel shift, last position first -- 3 instructions against the compare / select / or chain's
        // 5 (half-rate, profiles/r02_ubench_valu.md) of the form that kept concordant | discordant and had to add the halves first.
        uint32_t below = 0;
#ifdef MTH_TILE_ROT
        // A/B switch, not the default (round 5, built after three rounds of "counted, not built"; bit-identical, 145 tests; same box, three
        // runs each: kernel 0.0826 / 0.0852 / 0.0844 ms against 0.0829 / 0.0838 / 0.0840 without, two batches in turn 0.0938 / 0.0927 / 0.0922
        // against 0.0918 -- the LDS pipe is not what the kernel waits for): a thread's four 16-byte reads in an order rotated by (lane >> 2) & 3, so
        // that the four lanes of a 16-lane group that share a 16-byte slot (16 words apart: same banks) read different slots in every
        // instruction; the 16-bit mask comes out rotated by 4 rot bits and is turned back by one shift of its doubled copy.
        static_assert(PER == 16 || WIDE, "the rotation is written for 16 positions per thread");
        const uint32_t rot = ((uint32_t)lane >> 2) & 3u;
#pragma unroll
        for (int q = PER / 4 - 1; q >= 0; --q) {
            const uint4 x = reinterpret_cast<const uint4 *>(cnt)[tid * (PER / 4) + (((uint32_t)q + rot) & 3u)];
            below = __builtin_amdgcn_alignbit(below, (x.w & 0xffffu) - a.min_cov, 31);
            below = __builtin_amdgcn_alignbit(below, (x.z & 0xffffu) - a.min_cov, 31);
            below = __builtin_amdgcn_alignbit(below, (x.y & 0xffffu) - a.min_cov, 31);
            below = __builtin_amdgcn_alignbit(below, (x.x & 0xffffu) - a.min_cov, 31);
        }
        // bits 4 s .. 4 s + 3 of `below` belong to chunk (s + rot) & 3: rotate left by 4 rot within the 16 bits
        {
            const uint32_t dbl = below | (below << 16);
            below = (dbl >> (16u - 4u * rot)) & 0xffffu;
        }
#else
#pragma unroll
        for (int q = PER / 4 - 1; q >= 0; --q) {
            const uint4 x = reinterpret_cast<const uint4 *>(cnt)[tid * (PER / 4) + q];
            below = __builtin_amdgcn_alignbit(below, (x.w & 0xffffu) - a.min_cov, 31);
            below = __builtin_amdgcn_alignbit(below, (x.z & 0xffffu) - a.min_cov, 31);
            below = __builtin_amdgcn_alignbit(below, (x.y & 0xffffu) - a.min_cov, 31);
            below = __builtin_amdgcn_alignbit(below, (x.x & 0xffffu) - a.min_cov, 31);
        }
#endif
        qual = ~below & (PER == 32 ? 0xffffffffu : (1u << PER) - 1u);
    }
    {   // only the pass's positions [0, Wp) exist (the last tile of a region is short)
        const int32_t left = (int32_t)Wp - tid * PER;
        qual &= left >= PER ? 0xffffffffu : (left > 0 ? (1u << left) - 1u : 0u);
    }
    const uint32_t mine = __builtin_popcount(qual);
    const uint32_t incl = wave_scan_incl(mine);
    static_assert(B == 256, "four wave totals in one 16-byte LDS word");
    if (lane == 63) wave_off[wave] = incl;
    __syncthreads();
    // every thread adds up the totals of the waves before its own (a single thread doing the prefix cost a second barrier)
    const uint4 wt = *reinterpret_cast<const uint4 *>(wave_off);
    const uint32_t before = (wave > 0 ? wt.x : 0u) + (wave > 1 ? wt.y : 0u) + (wave > 2 ? wt.z : 0u);
    const uint32_t total = wt.x + wt.y + wt.z + wt.w;
    uint32_t o = out_base + before + incl - mine;
    SiteRec *__restrict__ out = a.scratch + (size_t)t * W;
    while (qual) {
        const uint32_t idx = (uint32_t)tid * PER + (uint32_t)__builtin_ctz(qual);
        qual &= qual - 1;
        SiteRec rr; rr.pos = P0 + (int32_t)idx; rr.pad = 0;
        if (WIDE) { rr.n_conc = cnt[idx]; rr.n_disc = cnt[W / 2 + idx]; }
        else { const uint32_t x = cnt[idx]; rr.n_disc = x >> 16; rr.n_conc = (x & 0xffffu) - rr.n_disc; }
        out[o++] = rr;
    }
    return total;
}

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

// One pass of a tile over its candidate reads [lo, hi): LDS counters for the reference positions
// [P0, P0 + Wp), then compaction.  do_lp: also the LPMD pair counts and read totals of the reads the tile owns
// (first pass only).  Returns the number of rows the pass appended at scratch[out_base..].
// MG > 0 (batches with max_span <= MG): the counter array has MG margin words on either side of the tile's W, so every call of a
// read that can touch the tile has a word of its own and the scatter address needs no clamp (see the scatter below).
#ifdef MTH_TILE_TRACE
#define g_tk4 (*mth_tk4p)
#define g_tk5 (*mth_tk5p)
__device__ unsigned long long mth_tk_dummy[2];
#endif
template <int W, int B, int NB, typename RelT, bool WIDE, bool CLAMP, int MG>
__device__ __forceinline__ uint32_t tile_pass(const TileArgs &a, const uint32_t t, const int32_t T0, const int32_t T1,
                                              const int32_t P0, const uint32_t Wp, const uint32_t lo, const uint32_t hi,
                                              const bool do_lp, const uint32_t out_base, uint32_t *cnt_raw,
                                              uint32_t (*red)[B / 64], uint32_t *wave_off, SlotTabs &tabs
#ifdef MTH_TILE_TRACE
                                              , unsigned long long *mth_tk4p = &mth_tk_dummy[0], unsigned long long *mth_tk5p = &mth_tk_dummy[1]
#endif
                                              ) {
    const int tid = threadIdx.x;
    uint32_t *const cnt = cnt_raw + MG;                         // the tile's first position
    constexpr bool MARGIN = MG > 0 && !WIDE;
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    constexpr bool PACKED = sizeof(RelT) == 1 && NB == 8;      // the table / packed-field forms below (8-bit relpos)
    // (the caller has cleared the counters and built the slot tables)

    uint32_t lp_c = 0, lp_d = 0, n_read = 0, n_valid = 0, bad = 0;
    uint32_t i = lo + tid;
    int32_t s = 0;
    uint32_t o0 = 0, n = 0, mq = 0;
    // PF (the plain-load instantiation with 8-bit relpos): ONE round trip per iteration.  Only the read's two call offsets are
    // fetched ahead (two registers); its calls, relative positions, start and mapq are then requested together -- the calls
    // unconditionally: every read of such a tile has its 8-slot window inside the arrays (safe_hi) -- and the NEXT read's
    // offsets right behind them, so that they travel during this read's arithmetic.  (The version that fetched the whole next
    // record ahead cost the registers of the 8th wave; profiles/r02_tile_latency.md.)
    constexpr bool PF = MTH_TILE_PF && !CLAMP && sizeof(RelT) == 1 && NB == 8;
    uint32_t o1 = 0;
    if (PF) { if (i < hi) { o0 = a.cpg_off[i]; o1 = a.cpg_off[i + 1]; } }
    else if (i < hi) { s = a.read_start[i]; o0 = a.cpg_off[i]; n = a.cpg_off[i + 1] - o0; mq = a.read_mapq[i]; }
    while (i < hi) {
        const uint32_t inext = i + B;
        const bool more = inext < hi;
        uint32_t v[NB];
        int32_t r[NB];
        uint32_t rraw0 = 0, rraw1 = 0;                         // PACKED, !CLAMP: the 8 relpos bytes as loaded
        uint32_t o0n = 0, o1n = 0;
        if constexpr (PF) {
            n = o1 - o0;
            const uint32_t *__restrict__ cp = a.cpg_pos + o0;
#pragma unroll
            for (int k4 = 0; k4 < NB / 4; ++k4) {
                const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(cp + 4 * k4);
                v[4 * k4] = x.x; v[4 * k4 + 1] = x.y; v[4 * k4 + 2] = x.z; v[4 * k4 + 3] = x.w;
            }
            if (do_lp) { const u32x2_a1 x = *reinterpret_cast<const u32x2_a1 *>(reinterpret_cast<const RelT *>(a.cpg_rel) + o0); rraw0 = x.x; rraw1 = x.y; }
            s = a.read_start[i]; mq = a.read_mapq[i];
            if (more) { o0n = a.cpg_off[inext]; o1n = a.cpg_off[inext + 1]; }
        }
        const bool owned = (s >= T0) && (s < T1);
        // lpmd.rs:176-179
        const bool lp_ok = do_lp && owned && (mq >= a.lpmd_min_qual);
        if (do_lp && owned) { n_read += 1; n_valid += lp_ok ? 1u : 0u; }
        // pdr.rs:147-157
        const bool pdr_ok = a.want_pdr && (n >= a.min_cpgs) && (mq >= a.pdr_min_qual) && (n > 0);
        const bool work = (lp_ok || pdr_ok) && n != 0;
        // (distances between live calls are < 2^16, so capping max_distance keeps dead-slot differences outside)
        const int32_t maxd = PACKED ? min(a.max_dist, 255) : min(a.max_dist, 1 << 20);   // 8-bit relpos: no distance beyond 255
        const int32_t mind = max(a.min_dist, 0);
        const bool any_lp = maxd >= a.min_dist && maxd >= 0 && __any(work && lp_ok && n > 1);   // min > max: no pair can qualify (and the range trick below would wrap)
        // All calls of the read in flight at once: two 16-byte loads from a per-read base (dword alignment
        // is all global_load_dwordx4 needs) and one 8/16-byte load of the relative positions.  Slots k >= n
        // read the NEXT reads' calls and are neutralised below; only the batch's last few reads could run
        // past the end of the arrays: a tile that holds them runs the CLAMP instantiation (per-tile choice,
        // so neither instantiation merges two load paths inside the loop).
        if (work) {
            const uint32_t *__restrict__ cp = a.cpg_pos + o0;
            const RelT *__restrict__ rp = rel + o0;
            if (PF) {
                (void)cp; (void)rp;                    // requested at the top of the iteration
            } else if (!CLAMP) {
#pragma unroll
                for (int k4 = 0; k4 < NB / 4; ++k4) {
                    const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(cp + 4 * k4);
                    v[4 * k4] = x.x; v[4 * k4 + 1] = x.y; v[4 * k4 + 2] = x.z; v[4 * k4 + 3] = x.w;
                }
                if (any_lp) {
                    if constexpr (PACKED) { const u32x2_a1 x = *reinterpret_cast<const u32x2_a1 *>(rp); rraw0 = x.x; rraw1 = x.y; }
                    else load_rel<RelT, NB>(rp, r);
                }
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
        //   dead call word  = a position 2^28 bp past the tile with the first call's state (concordant)
        //   dead rel        = (k+1) << 24 (any difference involving it exceeds every max_distance)
        // Span check (every call in [start-1, start+max_span-1] -- this is what makes the halo complete,
        // checked on the calls themselves instead of trusting read_end): max over the live slots.
        const uint32_t sm1 = (uint32_t)(s - 1);
        // dead call word: a position 2^28 bp past the tile (clamped to the trash word by the scatter), or -- with margins -- the
        // lane's own word at the start of the low margin; the first call's state either way (concordant)
        const uint32_t dead_w = ((MARGIN ? (uint32_t)(P0 - MG + (tid & 63)) : (uint32_t)T0 + (1u << 28)) & 0x7fffffffu) | (v[0] & 0x80000000u);
        const uint32_t n_lp = lp_ok ? min(n, (uint32_t)NB) : 0u;       // calls whose pairs are evaluated from the registers
        uint32_t acc = 0, xmax = (v[0] & 0x7fffffffu) - sm1;
        if constexpr (PACKED) {
            // masks of the live slots from the table row of n (rows 8.. are all-live): and / sub / and / bfi / xor per slot
            const uint32_t nrow = min(n, 8u);
            const uint4 ma = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[0], mb = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[1];
            const uint32_t mk[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
            uint32_t xs[8];
            xs[0] = xmax;
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                // v_bitop3_b32 (gfx950: any three-input bit function in one instruction; the compiler expanded the select
                // into not / and / and / or for six of the seven slots).  The state bit rides through the subtraction (no
                // borrow for a call at or after start - 1; a call before it leaves a huge value either way) and is masked
                // together with the liveness.
                xs[k] = __builtin_amdgcn_bitop3_b32(v[k] - sm1, mk[k], 0x7fffffffu, 0x80);       // a & b & c
                v[k] = __builtin_amdgcn_bitop3_b32(v[k], dead_w, mk[k], 0xe4);                    // c ? a : b
                acc = __builtin_amdgcn_bitop3_b32(acc, v[k], v[0], 0xf6);                         // a | (b ^ c)
            }
            xmax = max(max(max(xs[0], xs[1]), max(xs[2], xs[3])), max(max(xs[4], xs[5]), max(xs[6], xs[7])));
        } else {
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
        }
        uint32_t bad_it = (xmax > (uint32_t)a.max_span) ? 1u : 0u;      // this read's span violations
        uint32_t disc = acc >> 31;
        const bool any_long = __any(n > (uint32_t)NB);   // wave-uniform: the three tails below are rare
        if (any_long && n > (uint32_t)NB) {
            const uint32_t first = v[0] >> 31;
            for (uint32_t k = NB; k < n; ++k) {
                const uint32_t x = a.cpg_pos[o0 + k];
                disc |= (x >> 31) ^ first;
                bad_it |= ((x & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
            }
        }
        bad |= bad_it;
        // windowed pair counts (readutil.rs:166-224): pairs (j<k) with min <= rel_k - rel_j <= max.
        // The calls are sorted by relpos, so the distance at call-index gap g+1 is >= the distance at gap g:
        // walk the pair matrix by diagonals g = 1, 2, .. and stop once NO lane of the wave has a pair
        // within max_distance on the current diagonal (wave-uniform break).
        if constexpr (PACKED) {
            if (any_lp) {
                // packed call states (bit 15 of each field; the other bits of the top bytes are ignored by the masks below)
                uint32_t SQ[4], SO[4], Q[4], O[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) SQ[e] = __builtin_amdgcn_perm(v[2 * e + 1], v[2 * e], 0x070c030cu);
#pragma unroll
                for (int e = 0; e < 3; ++e) SO[e] = __builtin_amdgcn_perm(v[2 * e + 2], v[2 * e + 1], 0x070c030cu);
                SO[3] = __builtin_amdgcn_perm(0u, v[7], 0x070c030cu);
                if (!CLAMP) {
                    Q[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c010c00u); Q[1] = __builtin_amdgcn_perm(0u, rraw0, 0x0c030c02u);
                    Q[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c010c00u); Q[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c030c02u);
                    O[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c020c01u); O[1] = __builtin_amdgcn_perm(rraw1, rraw0, 0x0c040c03u);
                    O[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c020c01u); O[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c0c0c03u);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Q[e] = (uint32_t)r[2 * e] | ((uint32_t)r[2 * e + 1] << 16);
#pragma unroll
                    for (int e = 0; e < 3; ++e) O[e] = (uint32_t)r[2 * e + 1] | ((uint32_t)r[2 * e + 2] << 16);
                    O[3] = (uint32_t)r[7];
                }
                {
                    const uint4 da = reinterpret_cast<const uint4 *>(&tabs.dtab[n_lp][0])[0], db = reinterpret_cast<const uint4 *>(&tabs.dtab[n_lp][0])[1];
                    Q[0] += da.x; Q[1] += da.y; Q[2] += da.z; Q[3] += da.w; O[0] += db.x; O[1] += db.y; O[2] += db.z; O[3] += db.w;
                }
                const uint32_t KA = (0x8000u - (uint32_t)mind) * 0x10001u, KB = (0x8000u + (uint32_t)maxd) * 0x10001u;
                uint32_t accIN = 0, accDD = 0;
#pragma unroll
                for (int g = 1; g < 8; ++g) {
                    uint32_t orB = 0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int li = (g & 1) ? (g - 1) / 2 + m : g / 2 + m;      // index of the later operand in O (g odd) / Q (g even)
                        if (li > 3) break;
                        const uint32_t later = (g & 1) ? O[li] : Q[li], sl = (g & 1) ? SO[li] : SQ[li];
                        const uint32_t D = later - Q[m];
                        const uint32_t Bw = KB - D;
                        const uint32_t IN = __builtin_amdgcn_bitop3_b32(D + KA, Bw, 0x80008000u, 0x80);   // min <= distance <= max (readutil.rs:184, 196)
                        const uint32_t DD = IN & (sl ^ SQ[m]);
                        accIN += __builtin_popcount(IN);        // v_bcnt_u32_b32 adds its second operand: one instruction per count
                        accDD += __builtin_popcount(DD);
                        orB |= Bw;
                    }
                    if (!__any((orB & 0x80008000u) != 0u)) break;      // no lane has a pair within max_distance on this diagonal
                }
                lp_c += accIN - accDD;
                lp_d += accDD;
            }
        } else if (any_lp) {
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
        // a read with more than NB calls: the pairs among its first NB calls were counted above from the registers; pairs
        // whose LATER call is the (NB+1)-th or beyond come from memory (divergent, rare).  [This loop used to redo ALL
        // pairs of such a read from memory: 0.24 % of config 2's reads cost 14 % of the kernel, tools/tile_tail_probe.py]
        if (any_long && lp_ok && n > (uint32_t)NB) {
            for (uint32_t k = NB; k < n; ++k) {
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
        // scatter +1 to the pass's sites (pdr.rs:180-191), branch-free: a slot that is dead, outside the pass's
        // positions or belongs to a read PDR skips adds into a trash word instead (no exec juggling).
        // Packed: one word per position, coverage (concordant + discordant reads) in the low half, discordant in the high half; the
        // address is formed in byte units modulo 2^32 -- (word << 2) drops the state bit, candidates lie within
        // W + max_span of the tile, dead words 2^28 bp away and PDR-skipped reads get a base shifted by 2^28 bp --
        // and clamped with one min to the thread's trash word (3 VALU per slot).  Positions of the tile past
        // the region end may collect adds that way; the compaction masks them.
        // Wide: concordant at [0, W/2), discordant at [W/2, W); explicit range test.
        // With margins (MG): a read that passes the span check and starts in [P0 - MG + 1, P0 + W] has all its calls in
        // [P0 - MG, P0 + W + MG), a word each; dead slots point at the lane's word in the low margin; reads outside that
        // start range (the index hands out whole 32-bp quanta) cannot call a position of the tile and skip the scatter
        // together with the reads PDR skips and the span violators: one branch per read instead of a clamp per slot
        // (1 VALU per slot).  Margin words are never read.
        if (MARGIN) {
            if (pdr_ok && !bad_it && (uint32_t)s - (uint32_t)(P0 - MG + 1) <= (uint32_t)(W + MG - 1)) {      // unsigned: starts near 2^31 do not overflow
                const uint32_t one = disc ? 0x10001u : 1u;           // coverage in the low half, discordant reads in the high half
                const uint32_t base4 = (uint32_t)(P0 - MG) << 2;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt_raw) + ((v[k] << 2) - base4)), one);
                if (any_long) {
                    for (uint32_t k = NB; k < n; ++k) {
                        const uint32_t pk = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)P0;
                        if (pk < Wp) atomicAdd(cnt + pk, one);
                    }
                }
            }
        } else if (!WIDE) {
            const uint32_t one = disc ? 0x10001u : 1u;           // coverage in the low half, discordant reads in the high half
            const uint32_t base4 = ((uint32_t)P0 << 2) - (pdr_ok ? 0u : (1u << 30));
            const uint32_t trash4 = (uint32_t)(W + (tid & 63)) << 2;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const uint32_t a4 = min((v[k] << 2) - base4, trash4);
                atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt) + a4), one);
            }
            if (any_long && pdr_ok) {
                for (uint32_t k = NB; k < n; ++k) {
                    const uint32_t pk = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)P0;
                    if (pk < Wp) atomicAdd(cnt + pk, one);
                }
            }
        } else {
            const uint32_t wt = pdr_ok ? Wp : 0u;
            const uint32_t dw = disc ? (uint32_t)(W / 2) : 0u;
            const uint32_t trash = (uint32_t)(W + (tid & 63)) - dw;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const uint32_t pk = (v[k] & 0x7fffffffu) - (uint32_t)P0;
                atomicAdd(cnt + ((pk < wt ? pk : trash) + dw), 1u);
            }
            if (any_long && pdr_ok) {
                for (uint32_t k = NB; k < n; ++k) {
                    const uint32_t pk = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)P0;
                    if (pk < Wp) atomicAdd(cnt + dw + pk, 1u);
                }
            }
        }
        }   // work
        // the thread's next read.  (Requesting it together with the calls above and parking it in LDS -- one round
        // trip per iteration instead of two -- was built and measured: no change, profiles/r02_tile_latency.md.)
        if (PF) { o0 = o0n; o1 = o1n; }
        else if (more) { s = a.read_start[inext]; o0 = a.cpg_off[inext]; n = a.cpg_off[inext + 1] - o0; mq = a.read_mapq[inext]; }
        i = inext;
    }
#ifdef MTH_TILE_TRACE
    g_tk4 = __builtin_readcyclecounter();
#endif
    if (bad) atomicOr(const_cast<uint32_t *>(&a.st->err), (uint32_t)ERRB_SPAN);
    if (do_lp) tile_lpmd_partials<B, !WIDE>(a, t, red, lp_c, lp_d, n_read, n_valid);
    __syncthreads();
#ifdef MTH_TILE_TRACE
    g_tk5 = __builtin_readcyclecounter();
#endif
    if (do_lp) tile_lpmd_commit<B>(a, t, red);
    if (!a.want_pdr) return 0u;
    return tile_compact<W, B, WIDE>(a, t, P0, Wp, cnt, wave_off, out_base);
}

// Tile kernel.  W = reference positions per tile, B = threads per workgroup, NB = CpG calls of a
// read held in registers (reads with more calls take the memory loop for the tail).
//
// Latency structure (what v1 got wrong: one dependent HBM round trip per call): per tile the
// dependent chain is  idx -> read fields -> ALL calls of the read (NB independent loads in
// flight) -> LDS atomics.  Blocks are mapped to tiles XCD-aware so the halo reads of neighbouring
// tiles are served by the same L2.
//
// LDS: one 32-bit word per reference position (16 KiB per tile -> 8 waves per SIMD; with two words the LDS
// capped residency at 4 and the issue-bound kernel ran 27 % slower, profiles/r01_tile_variants.md).  A position
// is called at most once per candidate read, so while the tile has <= 65535 candidates both counts fit 16 bits
// (packed).  Heavier tiles (deep amplicons) take two passes over their reads with 32-bit counters, each
// covering half of the tile's positions -- same LDS footprint, no extra launch, exact.
#ifndef MTH_TILE_OCC
#define MTH_TILE_OCC 8      // workgroups per CU the kernel is compiled for (7 / 6 / 5: 72 / 80 / 96 VGPRs -- A/B in round 5)
#endif
template <int W, int B, int NB, typename RelT, int MG>
__global__ __launch_bounds__(B, MTH_TILE_OCC) void k_pdr_lpmd_tile(const TileArgs a, const uint32_t ntiles) {
    static_assert(MG == 0 || (MG >= 64 && MG % 4 == 0), "the wide passes keep their trash words in the high margin");
    __shared__ __attribute__((aligned(16))) uint32_t cnt[MG ? MG + W + MG : W + 64];   // [margin,] counters, then margin / one trash word per lane
    __shared__ uint32_t red[4][B / 64];
    __shared__ __attribute__((aligned(16))) uint32_t wave_off[B / 64];
    __shared__ __attribute__((aligned(16))) SlotTabs tabs;

    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
#ifdef MTH_TILE_TRACE
    unsigned long long tk[8];
    tk[0] = __builtin_readcyclecounter();
#define MTH_TK(k) do { tk[k] = __builtin_readcyclecounter(); } while (0)
#else
#define MTH_TK(k) do {} while (0)
#endif
    const int32_t T0 = a.region_beg + (int32_t)(t * W);
    // (T0 + W can exceed INT32_MAX on a contig of ~2^31 bp: bounds in 64-bit / unsigned arithmetic)
    const int32_t T1 = (int32_t)min((int64_t)T0 + W, (int64_t)a.region_end);

    // candidate reads: start in [T0 - max_span + 1, T0 + W]  (a call sits in [start-1, end]).
    // Both bounds are clamped to n_reads: a batch that failed validation in k_build_index (stale or
    // partial index) then only ever touches in-bounds reads, and its rows are discarded because the
    // getters report the error.  (An explicit load of the error flag here cost every tile a dependent
    // round trip before its first useful load.)
    // The three words the tile needs first are requested together; the counters are cleared while they travel.
    // (idx2: the kernel's own tile-granular index, exact bounds; otherwise the fine index every kernel can use)
    const uint32_t lo_raw = a.idx2 ? a.idx[t] : a.idx[((uint32_t)T0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT];
    const uint32_t hi_raw = a.idx2 ? a.idx2[t + 1] : a.idx[(((uint32_t)T0 + (uint32_t)W - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1];
    // only a tile that holds the batch's last reads can have a read whose NB-slot window runs past the call arrays:
    // k_build_index left the last read index that is safe for every tile ending at or before it
    const uint32_t safe_hi = a.st->safe_hi;
    auto clear = [&]() {
        for (int i = threadIdx.x; i < W / 4; i += B) reinterpret_cast<uint4 *>(cnt + MG)[i] = make_uint4(0, 0, 0, 0);
    };
    clear();
    slot_tabs_init(tabs, threadIdx.x);
    MTH_TK(1);
    __syncthreads();
    MTH_TK(2);
    const uint32_t lo = min(lo_raw, a.n_reads), hi = min(hi_raw, a.n_reads);
    MTH_TK(3);
    uint32_t rows;
    if (hi - lo <= 65535u) {
        static_assert(NB == 8, "safe_hi is computed for 8 call slots");
        if (hi <= safe_hi)
            rows = tile_pass<W, B, NB, RelT, false, false, MG>(a, t, T0, T1, T0, (uint32_t)(T1 - T0), lo, hi, a.want_lpmd != 0, 0u, cnt, red, wave_off, tabs
#ifdef MTH_TILE_TRACE
                                                               , &tk[4], &tk[5]
#endif
                                                               );
        else
            rows = tile_pass<W, B, NB, RelT, false, true, MG>(a, t, T0, T1, T0, (uint32_t)(T1 - T0), lo, hi, a.want_lpmd != 0, 0u, cnt, red, wave_off, tabs);
    } else {
        const int32_t Tm = (int32_t)min((int64_t)T0 + W / 2, (int64_t)T1);
        rows = tile_pass<W, B, NB, RelT, true, true, MG>(a, t, T0, T1, T0, (uint32_t)(Tm - T0), lo, hi, a.want_lpmd != 0, 0u, cnt, red, wave_off, tabs);
        __syncthreads();
        clear();
        __syncthreads();
        rows += tile_pass<W, B, NB, RelT, true, true, MG>(a, t, T0, T1, Tm, (uint32_t)(T1 - Tm), lo, hi, false, rows, cnt, red, wave_off, tabs);
    }
    if (threadIdx.x == 0) {
        a.tile_cnt[t] = rows;
        if (rows) atomicAdd(a.bucket + (t >> TILE_BUCKET_SHIFT), (unsigned long long)rows);
    }
#ifdef MTH_TILE_TRACE
    MTH_TK(7);
    if (threadIdx.x == 0 && a.trace) { tk[6] = hi - lo; for (int k = 0; k < 8; ++k) a.trace[8 * (size_t)t + k] = tk[k]; }
#endif
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the dense tile kernel (round 5): k_pdr_lpmd_runs + k_gather_runs, two launches per batch, no read index,
// no global atomic.  (A first persistent form drew tiles from per-XCD ticket counters and kept the per-tile row atomics: a
// device-scope atomic on one address costs ~100 ns on this part and they serialise -- 14 311 tickets over 8 addresses added
// 0.16 ms, 57 k row atomics over 56 addresses 0.14 ms; profiles/r05_persistent.md.)
//
// The grid is the chip's resident slots (8 workgroups on each of 256 CUs).  Workgroup w owns a RUN of consecutive tiles: the reads are
// cut into G equal pieces at r_w = w n / G and the run is [tile of start[r_w], tile of start[r_w+1]) -- runs are balanced by reads,
// not by positions, found with two loads and no index.  Inside a run everything a tile needs first comes from the tile before it:
//   * candidate reads of a tile are a contiguous range of the sorted reads.  Its first one is known from the previous tile (the waves
//     note, while they stream, where the reads that start before the next tile's lower bound end: one ds_max each); its end is
//     found by streaming: wave k takes the chunks lo + 256 j + 64 k of 64 reads until one holds no read starting at or before the
//     tile's last position + 1.  A read's start and call offsets are fetched one iteration ahead, so that test never waits;
//   * the run's first lower bound is a wave-cooperative 64-ary search next to r_w (two round trips when the window fits);
//   * sortedness is validated on the workgroup's own piece of read_start in the preamble (every adjacent pair, whatever the data);
//   * the compaction is wave-private: a wave owns 1024 consecutive positions and its own quarter of the tile's scratch slice
//     (tile_cnt holds four counts per tile), so it needs no cross-wave prefix and no barrier, and it zeroes the counters behind it:
//     two workgroup barriers per tile (after the read loop, after the compaction) instead of three;
//   * LPMD counters and the span-error bit stay in registers across the run; one set of plain stores per workgroup at the end (the
//     gather's last workgroup adds them up), the run's row total likewise -- k_gather_runs takes its row base from the totals of the
//     runs before it (G <= 2048 words) instead of bucket sums kept by atomics;
//   * calls and relative positions are fetched through buffer descriptors: the hardware bounds check replaces the CLAMP
//     instantiation for the batch's last reads (an 8-slot window that runs past the arrays reads zeros into dead slots);
//   * a stretch of tiles without reads is skipped in one step (their counts written as zeros);
//   * a tile with more candidate reads than a 16-bit counter can count (deep amplicons) is taken in passes of 255 iterations, the
//     counters added into the tile's scratch slice (used as dense 32-bit rows) in between.
// 8-bit relative positions only (reads of <= 255 bases); launch_pdr_lpmd keeps the one-tile-per-workgroup form for the rest.
constexpr uint32_t PT_PASS_ITERS = 255u;   // iterations of 256 reads per pass over 16-bit counters (255 x 256 <= 65 535)

struct RunArgs {
    uint32_t *run_tile0;             // [G + 1] first tile of each run (written by the tile kernel)
    uint32_t *run_rows;              // [G] rows of each run
    unsigned long long *run_lpmd;    // [G][4] LPMD partial sums of each run
    DevState *cst;                   // the sink's / job's counters: cur_base is set here for unpipelined batches (nullptr: pipelined)
    uint32_t hint_stride;            // probe spacing of the first search round (~ reads per tile / 48)
    unsigned long long *trace;       // -DMTH_RUNS_TRACE builds: 8 words per workgroup (tools/runs_trace.py)
};
#ifdef MTH_RUNS_TRACE
#define RT_NOW() __builtin_readcyclecounter()
#define RT_ADD(acc, t0) do { const unsigned long long n__ = RT_NOW(); acc += n__ - t0; t0 = n__; } while (0)
#else
#define RT_NOW() 0ull
#define RT_ADD(acc, t0) do {} while (0)
#endif

__device__ __forceinline__ unsigned long long wave_sum64(uint32_t v) {
    unsigned long long x = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// first index i in [0, n] with a[i] >= bound (n if none), for sorted a; every lane of the wave calls it with the same arguments.
// Round 0 probes 64 elements `stride` apart ending at `hint` (the caller expects the answer just below it); then 64-ary rounds.
__device__ __forceinline__ uint32_t wave_lower_bound(const int32_t *__restrict__ a, uint32_t n, int32_t bound, uint32_t hint, uint32_t stride) {
    const uint32_t lane = threadIdx.x & 63u;
    if (n == 0) return 0;
    uint32_t lo, hi;
    {
        const uint32_t top = min(hint, n - 1u);
        const uint32_t back = (63u - lane) * stride;
        const bool in = back <= top;
        const uint32_t q = in ? top - back : 0u;
        const int32_t v = a[q];
        const unsigned long long below = __ballot(in && v < bound);      // a prefix of the probes that exist
        const unsigned long long exist = __ballot(in);
        const uint32_t first = (uint32_t)__builtin_ctzll(exist);         // lowest existing probe (lane 63 always exists)
        const uint32_t c = (uint32_t)__builtin_popcountll(below);
        if (c == 0) { lo = 0; hi = top - (63u - first) * stride; }        // a[lowest probe] >= bound
        else {
            const uint32_t lastb = first + c - 1u;                        // lane of the last probe below the bound
            lo = top - (63u - lastb) * stride + 1u;
            hi = lastb == 63u ? n : top - (63u - lastb - 1u) * stride;
        }
    }
    while (hi > lo) {                                                     // answer in [lo, hi]; a[hi] >= bound or hi == n
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t q = lo + (lane + 1u) * step - 1u;
        const bool in = q < hi;
        const int32_t v = in ? a[q] : 0;
        const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(in && v < bound));
        const uint32_t qc = lo + (c + 1u) * step - 1u;
        lo += c * step;
        if (qc < hi) hi = qc;
    }
    return lo;
}

#ifndef MTH_RUNS_OCC
#define MTH_RUNS_OCC 8
#endif
template <int W, int MG>
__global__ __launch_bounds__(256, MTH_RUNS_OCC) void k_pdr_lpmd_runs(const TileArgs a, const RunArgs ra, const uint32_t ntiles) {
    constexpr int B = 256, NB = 8;
    constexpr bool MARGIN = MG > 0;
    static_assert(W == 4096, "a wave owns 1024 positions: 16 per lane");
    __shared__ __attribute__((aligned(16))) uint32_t cnt_raw[MG ? MG + W + MG : W + 64];
    __shared__ __attribute__((aligned(16))) SlotTabs tabs;
    __shared__ uint32_t s_lon[2], s_more[2], s_rows[4];
    __shared__ unsigned long long s_lp[4][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *const cnt = cnt_raw + MG;
    const uint32_t nr = a.n_reads, G = gridDim.x;
    // which run: workgroup b runs on XCD b % 8 (observed; speed only).  MTH_RUNS_MAP 1: each XCD takes a contiguous eighth of the runs;
    // 2: and the workgroups that share a CU (dispatched 32 apart within the XCD) take neighbouring runs
#ifndef MTH_RUNS_MAP
#define MTH_RUNS_MAP 0
#endif
    uint32_t w = blockIdx.x;
    if (MTH_RUNS_MAP && (G & 255u) == 0) {
        const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3, cpx = G >> 3, k = cpx >> 5;
        w = MTH_RUNS_MAP == 1 ? x * cpx + j : x * cpx + (j & 31u) * k + (j >> 5);
    }
    [[maybe_unused]] const unsigned long long rt_beg = RT_NOW();
    [[maybe_unused]] unsigned long long rt_loop = 0, rt_b1 = 0, rt_comp = 0, rt_b2 = 0, rt_pre = 0, rt_t = rt_beg, rt_ntile = 0, rt_wait = 0, rt_iters = 0;
    for (int i = tid; i < W / 4; i += B) reinterpret_cast<uint4 *>(cnt)[i] = make_uint4(0, 0, 0, 0);
    slot_tabs_init(tabs, tid);
    if (tid < 2) { s_lon[tid] = 0; s_more[tid] = 0; }
    if (tid < 4) s_rows[tid] = 0;
    if (tid < 16) s_lp[tid >> 2][tid & 3] = 0ull;
    if (w == 0 && tid == 0 && ra.cst) ra.cst->cur_base = ra.cst->n_sites;      // (pipelined batches: the base travels along the chain of gathers)

    // the run: tiles [tb, te)
    const uint32_t r0 = (uint32_t)(((unsigned long long)w * nr) / G), r1 = (uint32_t)(((unsigned long long)(w + 1u) * nr) / G);
    auto tile_of = [&](uint32_t r) -> uint32_t {
        const int64_t d = (int64_t)a.read_start[r] - a.region_beg;
        return d <= 0 ? 0u : (uint32_t)min((int64_t)ntiles, d / W);
    };
    const uint32_t tb = __builtin_amdgcn_readfirstlane((w == 0 || nr == 0) ? 0u : tile_of(r0));
    const uint32_t te = __builtin_amdgcn_readfirstlane((w + 1u == G || nr == 0) ? ntiles : max(tb, tile_of(r1)));
    // sortedness of the piece: the pairs (i - 1, i) for i in (r0, r1] (workgroup 0's piece starts at the pair (0, 1)).  Four reads and
    // the one before them per thread and step; the first eight steps' loads are all requested before the first is looked at
    {
        uint32_t err = 0;
        const uint32_t vend = min(r1 + 1u, nr);                      // one past the last i of the piece
        auto check4 = [&](const u32x4_a4 x, int32_t pv, uint32_t i0) {
            const int32_t e0 = (int32_t)x.x, e1 = (int32_t)x.y, e2 = (int32_t)x.z, e3 = (int32_t)x.w;
            if (i0 < vend) err |= e0 < pv ? 1u : 0u;
            if (i0 + 1u < vend) err |= e1 < e0 ? 1u : 0u;
            if (i0 + 2u < vend) err |= e2 < e1 ? 1u : 0u;
            if (i0 + 3u < vend) err |= e3 < e2 ? 1u : 0u;
        };
        constexpr int VSTEPS = 8;
        u32x4_a4 vx[VSTEPS]; int32_t vp[VSTEPS];
#pragma unroll
        for (int j = 0; j < VSTEPS; ++j) {
            const uint32_t i0 = r0 + 1u + 4u * ((uint32_t)tid + 256u * (uint32_t)j);
            vx[j] = u32x4_a4{0, 0, 0, 0}; vp[j] = 0;
            if (i0 < vend) {
                vp[j] = a.read_start[i0 - 1u];
                if (i0 + 4u <= nr) vx[j] = *reinterpret_cast<const u32x4_a4 *>(a.read_start + i0);
                else { vx[j].x = (uint32_t)a.read_start[i0]; vx[j].y = i0 + 1u < nr ? (uint32_t)a.read_start[i0 + 1u] : 0u; vx[j].z = i0 + 2u < nr ? (uint32_t)a.read_start[i0 + 2u] : 0u; vx[j].w = 0u; }
            }
        }
#pragma unroll
        for (int j = 0; j < VSTEPS; ++j) check4(vx[j], vp[j], r0 + 1u + 4u * ((uint32_t)tid + 256u * (uint32_t)j));
        for (uint32_t i = r0 + 1u + 1024u * VSTEPS + (uint32_t)tid; i < vend; i += B) err |= a.read_start[i] < a.read_start[i - 1u] ? 1u : 0u;   // (pieces of more than 8192 reads)
        if (err) atomicOr(const_cast<uint32_t *>(&a.st->err), (uint32_t)ERRB_UNSORTED);
    }
    if (tid == 0) { ra.run_tile0[w] = tb; if (w + 1u == G) ra.run_tile0[G] = ntiles; }

    const __amdgpu_buffer_rsrc_t rs_pos = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.cpg_pos), 0, a.n_cpgs * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rel = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.cpg_rel), 0, a.n_cpgs, 0x00020000);
    const bool do_lp = a.want_lpmd != 0;
    const int32_t maxd = min(a.max_dist, 255);
    const int32_t mind = max(a.min_dist, 0);
    const bool lp_range = maxd >= a.min_dist && maxd >= 0;
    const uint32_t KA = (0x8000u - (uint32_t)mind) * 0x10001u, KB = (0x8000u + (uint32_t)maxd) * 0x10001u;

    // per-thread LPMD counters across the run.  nrv: reads owned (bits 0-14), of them with mapq >= min_qual (bits 15-29), a span
    // violation seen (bit 31).  Moved to the wave's 64-bit sums in LDS when a lane nears its field's top.
    uint32_t lp_c = 0, lp_d = 0, nrv = 0;
    auto flush_lpmd = [&]() {
        const unsigned long long f0 = wave_sum64(lp_c), f1 = wave_sum64(lp_d), f2 = wave_sum64(nrv & 0x7fffu), f3 = wave_sum64((nrv >> 15) & 0x7fffu);
        if (lane == 0) { s_lp[wave][0] += f0; s_lp[wave][1] += f1; s_lp[wave][2] += f2; s_lp[wave][3] += f3; }      // the wave's own row
        lp_c = lp_d = 0; nrv &= 0x80000000u;
    };

    uint32_t t = tb;
    uint32_t lo = 0;
    if (t < te) {
        const int32_t T0 = a.region_beg + (int32_t)(t * W);
        lo = __builtin_amdgcn_readfirstlane(wave_lower_bound(a.read_start, nr, (int32_t)max((int64_t)T0 - a.max_span + 1, (int64_t)INT32_MIN), r0, ra.hint_stride));
    }
    __syncthreads();
    uint32_t par = 0;
    // the thread's read of the coming iteration: index, start, call offsets (INT32_MAX start: no such read)
    uint32_t i = 0, o0 = 0, o1 = 0;
    int32_t s = INT32_MAX;
    auto fetch = [&](uint32_t idx) {
        i = idx; s = INT32_MAX; o0 = 0; o1 = 0;
        if (idx < nr) { s = a.read_start[idx]; o0 = a.cpg_off[idx]; o1 = a.cpg_off[idx + 1u]; }
    };
    int32_t s_first = INT32_MAX;                              // start of read lo (uniform)
    if (t < te) { fetch(lo + (uint32_t)tid); s_first = __builtin_amdgcn_readfirstlane(lo < nr ? a.read_start[lo] : INT32_MAX); }
    uint32_t lon_run = 0;                                     // deep tiles: where the next tile's candidates begin, over the passes so far

    RT_ADD(rt_pre, rt_t);
    while (t < te) {
        const int32_t T0 = a.region_beg + (int32_t)(t * W);
        const int64_t T0W = (int64_t)T0 + W;
        const int32_t T1 = (int32_t)min(T0W, (int64_t)a.region_end);
        const uint32_t Wp = (uint32_t)(T1 - T0);
        const int32_t P0 = T0;
        if ((int64_t)s_first > T0W) {
            // no read can touch this tile, nor the ones before the tile that holds read lo's start - 1: zero counts, same lo
            uint32_t tnext = te;
            if (lo < nr) tnext = (uint32_t)min((int64_t)te, max((int64_t)t + 1, ((int64_t)s_first - a.region_beg + W - 1) / W - 1));
            if (a.want_pdr) for (uint32_t q = 4u * t + (uint32_t)tid; q < 4u * tnext; q += B) a.tile_cnt[q] = 0u;
            t = tnext;
            continue;
        }
        const int32_t cmax = (int32_t)min(T0W, (int64_t)INT32_MAX);          // candidates start at or before T0 + W
        const int32_t nb = (int32_t)max(T0W - a.max_span + 1, (int64_t)INT32_MIN);   // the next tile's lower bound
        bool first = true, last;
        do {
            uint32_t below_end = 0;                                            // wave: one past the last read seen that starts before nb
            uint32_t it = 0;
            bool cand = s <= cmax;
            while (it < PT_PASS_ITERS && __any(cand)) {
                uint32_t v[NB];
                const uint32_t n = o1 - o0;
#ifdef MTH_RUNS_GLOBAL_LOADS      // experiment: plain loads (may run past the arrays for the batch's last reads)
                {
                    const u32x4_a4 x0 = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0);
                    const u32x4_a4 x1 = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0 + 4);
                    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                }
                uint32_t rraw0 = 0, rraw1 = 0;
                if (do_lp) { const u32x2_a1 x = *reinterpret_cast<const u32x2_a1 *>(reinterpret_cast<const uint8_t *>(a.cpg_rel) + o0); rraw0 = x.x; rraw1 = x.y; }
#else
                {
                    const u32x4_a4 x0 = __builtin_amdgcn_raw_buffer_load_b128(rs_pos, o0 * 4u, 0, 0);
                    const u32x4_a4 x1 = __builtin_amdgcn_raw_buffer_load_b128(rs_pos, o0 * 4u + 16u, 0, 0);
                    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                }
                uint32_t rraw0 = 0, rraw1 = 0;
                if (do_lp) { const u32x2_a1 x = __builtin_amdgcn_raw_buffer_load_b64(rs_rel, o0, 0, 0); rraw0 = x.x; rraw1 = x.y; }
#endif
                const uint32_t mq = cand ? a.read_mapq[i] : 0u;
                // the thread's next read (start and offsets): requested now, tested at the bottom
                const uint32_t inx = i + B;
                int32_t sn = INT32_MAX; uint32_t o0n = 0, o1n = 0;
                if (inx < nr) { sn = a.read_start[inx]; o0n = a.cpg_off[inx]; o1n = a.cpg_off[inx + 1u]; }
#ifdef MTH_RUNS_TRACE
                { const unsigned long long w0 = RT_NOW(); __builtin_amdgcn_s_waitcnt(0); rt_wait += RT_NOW() - w0; rt_iters += 1; }
#endif
                {
                    const unsigned long long bm = __ballot(cand && s < nb);
                    if (bm) below_end = max(below_end, i - (uint32_t)lane + 64u - (uint32_t)__builtin_clzll(bm));
                }
                const bool owned = cand && (s >= T0) && (s < T1);
                const bool lp_ok = do_lp && owned && (mq >= a.lpmd_min_qual);                 // lpmd.rs:176-179
                if (do_lp && owned) nrv += lp_ok ? 0x8001u : 1u;
                const bool pdr_ok = cand && a.want_pdr && (n >= a.min_cpgs) && (mq >= a.pdr_min_qual) && (n > 0);   // pdr.rs:147-157
                const bool work = (lp_ok || pdr_ok) && n != 0;
                const bool any_lp = lp_range && __any(work && lp_ok && n > 1);
                if (work) {
                    const uint32_t sm1 = (uint32_t)(s - 1);
                    const uint32_t dead_w = ((MARGIN ? (uint32_t)(P0 - MG + lane) : (uint32_t)T0 + (1u << 28)) & 0x7fffffffu) | (v[0] & 0x80000000u);
                    const uint32_t n_lp = lp_ok ? min(n, (uint32_t)NB) : 0u;
                    uint32_t acc = 0, xmax;
                    {
                        const uint32_t nrow = min(n, 8u);
                        const uint4 ma = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[0], mb = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[1];
                        const uint32_t mk[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
                        uint32_t xs[8];
                        xs[0] = (v[0] & 0x7fffffffu) - sm1;
#pragma unroll
                        for (int k = 1; k < 8; ++k) {
                            xs[k] = __builtin_amdgcn_bitop3_b32(v[k] - sm1, mk[k], 0x7fffffffu, 0x80);       // a & b & c
                            v[k] = __builtin_amdgcn_bitop3_b32(v[k], dead_w, mk[k], 0xe4);                    // c ? a : b
                            acc = __builtin_amdgcn_bitop3_b32(acc, v[k], v[0], 0xf6);                         // a | (b ^ c)
                        }
                        xmax = max(max(max(xs[0], xs[1]), max(xs[2], xs[3])), max(max(xs[4], xs[5]), max(xs[6], xs[7])));
                    }
                    uint32_t bad_it = (xmax > (uint32_t)a.max_span) ? 1u : 0u;
                    uint32_t disc = acc >> 31;
                    const bool any_long = __any(n > (uint32_t)NB);
                    if (any_long && n > (uint32_t)NB) {
                        const uint32_t firstst = v[0] >> 31;
                        for (uint32_t k = NB; k < n; ++k) {
                            const uint32_t x = a.cpg_pos[o0 + k];
                            disc |= (x >> 31) ^ firstst;
                            bad_it |= ((x & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
                        }
                    }
                    nrv |= bad_it << 31;
                    if (any_lp) {       // windowed pair counts, two pairs per instruction (see tile_pass)
                        if (__any(o0 + 8u > a.n_cpgs)) {      // the batch's last few reads: the bounds check drops whole dwords of a window that crosses the end
                            if (o0 + 8u > a.n_cpgs) {
                                const uint8_t *__restrict__ rel = reinterpret_cast<const uint8_t *>(a.cpg_rel);
                                rraw0 = rraw1 = 0;
#pragma unroll 1
                                for (uint32_t k = 0; k < 8u && o0 + k < a.n_cpgs; ++k) {
                                    const uint32_t bte = rel[o0 + k];
                                    if (k < 4) rraw0 |= bte << (8 * k); else rraw1 |= bte << (8 * (k - 4));
                                }
                            }
                        }
                        uint32_t SQ[4], SO[4], Q[4], O[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) SQ[e] = __builtin_amdgcn_perm(v[2 * e + 1], v[2 * e], 0x070c030cu);
#pragma unroll
                        for (int e = 0; e < 3; ++e) SO[e] = __builtin_amdgcn_perm(v[2 * e + 2], v[2 * e + 1], 0x070c030cu);
                        SO[3] = __builtin_amdgcn_perm(0u, v[7], 0x070c030cu);
                        Q[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c010c00u); Q[1] = __builtin_amdgcn_perm(0u, rraw0, 0x0c030c02u);
                        Q[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c010c00u); Q[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c030c02u);
                        O[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c020c01u); O[1] = __builtin_amdgcn_perm(rraw1, rraw0, 0x0c040c03u);
                        O[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c020c01u); O[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c0c0c03u);
                        {
                            const uint4 da = reinterpret_cast<const uint4 *>(&tabs.dtab[n_lp][0])[0], db = reinterpret_cast<const uint4 *>(&tabs.dtab[n_lp][0])[1];
                            Q[0] += da.x; Q[1] += da.y; Q[2] += da.z; Q[3] += da.w; O[0] += db.x; O[1] += db.y; O[2] += db.z; O[3] += db.w;
                        }
                        uint32_t accIN = 0, accDD = 0;
#pragma unroll
                        for (int g = 1; g < 8; ++g) {
                            uint32_t orB = 0;
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                const int li = (g & 1) ? (g - 1) / 2 + m : g / 2 + m;
                                if (li > 3) break;
                                const uint32_t later = (g & 1) ? O[li] : Q[li], sl = (g & 1) ? SO[li] : SQ[li];
                                const uint32_t D = later - Q[m];
                                const uint32_t Bw = KB - D;
                                const uint32_t IN = __builtin_amdgcn_bitop3_b32(D + KA, Bw, 0x80008000u, 0x80);   // readutil.rs:184, 196
                                const uint32_t DD = IN & (sl ^ SQ[m]);
                                accIN += __builtin_popcount(IN);
                                accDD += __builtin_popcount(DD);
                                orB |= Bw;
                            }
                            if (!__any((orB & 0x80008000u) != 0u)) break;
                        }
                        lp_c += accIN - accDD;
                        lp_d += accDD;
                    }
                    if (any_long && lp_ok && n > (uint32_t)NB) {      // pairs whose later call is the 9th or beyond: from memory (rare)
                        const uint8_t *__restrict__ rel = reinterpret_cast<const uint8_t *>(a.cpg_rel);
                        for (uint32_t k = NB; k < n; ++k) {
                            const int32_t rk = (int32_t)rel[o0 + k];
                            const uint32_t mk = a.cpg_pos[o0 + k] >> 31;
                            for (uint32_t j = k; j-- > 0;) {
                                const int32_t dist = rk - (int32_t)rel[o0 + j];
                                if (dist > a.max_dist) break;          // readutil.rs:184
                                if (dist < a.min_dist) continue;       // readutil.rs:196
                                if ((a.cpg_pos[o0 + j] >> 31) == mk) lp_c += 1; else lp_d += 1;
                            }
                        }
                    }
                    // scatter (pdr.rs:180-191); see tile_pass for the two address forms
                    if (MARGIN) {
                        if (pdr_ok && !bad_it && (uint32_t)s - (uint32_t)(P0 - MG + 1) <= (uint32_t)(W + MG - 1)) {
                            const uint32_t one = disc ? 0x10001u : 1u;
                            const uint32_t base4 = (uint32_t)(P0 - MG) << 2;
#pragma unroll
                            for (int k = 0; k < NB; ++k)
                                atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt_raw) + ((v[k] << 2) - base4)), one);
                            if (any_long) {
                                for (uint32_t k = NB; k < n; ++k) {
                                    const uint32_t pk = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)P0;
                                    if (pk < Wp) atomicAdd(cnt + pk, one);
                                }
                            }
                        }
                    } else {
                        const uint32_t one = disc ? 0x10001u : 1u;
                        const uint32_t base4 = ((uint32_t)P0 << 2) - (pdr_ok ? 0u : (1u << 30));
                        const uint32_t trash4 = (uint32_t)(W + lane) << 2;
#pragma unroll
                        for (int k = 0; k < NB; ++k) {
                            const uint32_t a4 = min((v[k] << 2) - base4, trash4);
                            atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt) + a4), one);
                        }
                        if (any_long && pdr_ok) {
                            for (uint32_t k = NB; k < n; ++k) {
                                const uint32_t pk = (a.cpg_pos[o0 + k] & 0x7fffffffu) - (uint32_t)P0;
                                if (pk < Wp) atomicAdd(cnt + pk, one);
                            }
                        }
                    }
                }   // work
                i = inx; s = sn; o0 = o0n; o1 = o1n;
                cand = s <= cmax;
                ++it;
            }
            // the wave is through with this pass: where the next tile's candidates begin as far as it has seen, and whether it stopped
            // at the pass limit with candidates left
            const bool more_w = __any(cand);
            if (lane == 0) {
                if (below_end) atomicMax(&s_lon[par], below_end);
                if (more_w) s_more[par] = 1u;
            }
            if (__any((lp_c | lp_d) >> 24 | (nrv & 0x20004000u))) flush_lpmd();
            RT_ADD(rt_loop, rt_t);
            __syncthreads();
            RT_ADD(rt_b1, rt_t);
            last = __builtin_amdgcn_readfirstlane(s_more[par]) == 0u;
            lon_run = max(lon_run, (uint32_t)__builtin_amdgcn_readfirstlane(s_lon[par]));
            const uint32_t lo_next = max(lo, lon_run);
            int32_t s_first_v = INT32_MAX;
            if (last) {
                // the thread's first read of the next tile (and the start of its first candidate): they travel under the compaction
                fetch(lo_next + (uint32_t)tid);
                if (lo_next < nr) s_first_v = a.read_start[lo_next];
            }
            if (a.want_pdr) {
                SiteRec *__restrict__ slice = a.scratch + (size_t)t * W;
                if (first && last) {
                    // wave-private compaction of the wave's 1024 positions, counters zeroed behind
                    uint32_t below = 0;
#pragma unroll
                    for (int q = 3; q >= 0; --q) {
                        const uint4 x = reinterpret_cast<const uint4 *>(cnt)[tid * 4 + q];
                        below = __builtin_amdgcn_alignbit(below, (x.w & 0xffffu) - a.min_cov, 31);
                        below = __builtin_amdgcn_alignbit(below, (x.z & 0xffffu) - a.min_cov, 31);
                        below = __builtin_amdgcn_alignbit(below, (x.y & 0xffffu) - a.min_cov, 31);
                        below = __builtin_amdgcn_alignbit(below, (x.x & 0xffffu) - a.min_cov, 31);
                    }
                    uint32_t qual = ~below & 0xffffu;
                    {
                        const int32_t left = (int32_t)Wp - tid * 16;
                        qual &= left >= 16 ? 0xffffu : (left > 0 ? (1u << left) - 1u : 0u);
                    }
                    const uint32_t mine = __builtin_popcount(qual);
                    const uint32_t incl = wave_scan_incl(mine);
                    uint32_t o = incl - mine;
                    SiteRec *__restrict__ out = slice + (size_t)wave * (W / 4);
                    while (qual) {
                        const uint32_t idx = (uint32_t)tid * 16u + (uint32_t)__builtin_ctz(qual);
                        qual &= qual - 1;
                        const uint32_t x = cnt[idx];
                        SiteRec rr; rr.pos = P0 + (int32_t)idx; rr.pad = 0;
                        rr.n_disc = x >> 16; rr.n_conc = (x & 0xffffu) - rr.n_disc;
                        out[o++] = rr;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) reinterpret_cast<uint4 *>(cnt)[tid * 4 + q] = make_uint4(0, 0, 0, 0);
                    if (lane == 63) { a.tile_cnt[4u * t + (uint32_t)wave] = incl; s_rows[wave] += incl; }
                } else {
                    // deep tile: the pass's counters into the slice's dense rows (row p: coverage in n_conc, discordant reads in n_disc;
                    // the thread owns rows tid*16 ..), and after the last pass the wave compacts its 1024 rows in place, 64 per round
                    // (a round's rows are read before any of them is overwritten, and rows land at or before where they came from)
#pragma unroll 1
                    for (uint32_t k = 0; k < 16; ++k) {
                        const uint32_t p = (uint32_t)tid * 16u + k;
                        const uint32_t x = cnt[p];
                        cnt[p] = 0;
                        SiteRec rr = slice[p];
                        if (first) { rr.n_conc = 0; rr.n_disc = 0; }
                        rr.n_conc += x & 0xffffu; rr.n_disc += x >> 16;
                        slice[p] = rr;
                    }
                    if (last) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        uint32_t nout = 0;
#pragma unroll 1
                        for (uint32_t r = 0; r < 16; ++r) {
                            const uint32_t p = (uint32_t)wave * 1024u + r * 64u + (uint32_t)lane;
                            SiteRec rr = slice[p];
                            const bool q = p < Wp && rr.n_conc >= a.min_cov;
                            const unsigned long long m = __ballot(q);
                            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            if (q) {
                                rr.pos = P0 + (int32_t)p; rr.pad = 0;
                                rr.n_conc -= rr.n_disc;
                                slice[(size_t)wave * 1024u + nout + before] = rr;
                            }
                            nout += (uint32_t)__builtin_popcountll(m);
                        }
                        if (lane == 63) { a.tile_cnt[4u * t + (uint32_t)wave] = nout; s_rows[wave] += nout; }
                    }
                }
            }
            if (tid == 0) { s_lon[par ^ 1u] = 0u; s_more[par ^ 1u] = 0u; }      // the words of the next pass / tile
            RT_ADD(rt_comp, rt_t);
            __syncthreads();
            RT_ADD(rt_b2, rt_t);
            rt_ntile += 1;
            first = false;
            par ^= 1u;
            if (last) { lo = lo_next; lon_run = 0; s_first = __builtin_amdgcn_readfirstlane(s_first_v); }
        } while (!last);
        ++t;
    }
    // the run's totals: plain stores, summed by the gather
    if (nrv >> 31) atomicOr(const_cast<uint32_t *>(&a.st->err), (uint32_t)ERRB_SPAN);
    flush_lpmd();
    __syncthreads();
#ifdef MTH_RUNS_TRACE
    if (ra.trace && lane == 0) {
        unsigned long long *tr = ra.trace + ((size_t)w * 4u + (uint32_t)wave) * 8u;
        tr[0] = rt_wait; tr[1] = RT_NOW() - rt_beg; tr[2] = rt_pre; tr[3] = rt_loop; tr[4] = rt_b1; tr[5] = rt_comp; tr[6] = rt_b2; tr[7] = rt_ntile | (rt_iters << 32);
    }
#endif
    if (tid == 0) ra.run_rows[w] = s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3];
    if (tid < 4) ra.run_lpmd[(size_t)w * 4u + tid] = s_lp[0][tid] + s_lp[1][tid] + s_lp[2][tid] + s_lp[3][tid];
}

// Gather of the run form: one 256-thread workgroup per run.  Row base = the batch's base + the rows of the runs before it (one
// word per run, <= 2048); the run's slots (four per tile) are scanned 1024 at a time and their rows copied as one flat range.  The
// last run's workgroup commits the batch (rows, LPMD counters summed over the runs) as k_gather does.
__global__ __launch_bounds__(256) void k_gather_runs(const SiteRec *__restrict__ scratch, const uint32_t *__restrict__ slot_cnt,
                                                     const uint32_t *__restrict__ run_tile0, const uint32_t *__restrict__ run_rows,
                                                     const unsigned long long *__restrict__ run_lpmd, uint32_t G, int fin_only, int want_lpmd,
                                                     DevState *__restrict__ st, const DevState *__restrict__ base_st, DevState *__restrict__ next_st,
                                                     DevState *__restrict__ lane_st, int reset_first, uint32_t *__restrict__ batch_cnt, uint32_t slot_w,
                                                     int32_t *__restrict__ out_pos, float *__restrict__ out_pdr, uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    __shared__ uint32_t s_off[1025];
    __shared__ uint32_t s_part[4], s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t w = fin_only ? G - 1u : blockIdx.x;
    const uint64_t cur = reset_first ? 0ull : base_st->cur_base;
    uint32_t part = 0;
    if (!fin_only) for (uint32_t r = tid; r < w; r += 256u) part += run_rows[r];          // rows of one batch fit 32 bits
    const uint32_t wsum = wave_sum(part);
    if (lane == 0) s_part[wave] = wsum;
    __syncthreads();
    const uint32_t before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    uint32_t done = 0;                                                                      // rows of the run copied so far
    if (!fin_only) {
        const uint32_t e_beg = run_tile0[w] * 4u, e_end = run_tile0[w + 1u] * 4u;
        const uint32_t mine_total = run_rows[w];
        for (uint32_t e0 = e_beg; e0 < e_end && done < mine_total; e0 += 1024u) {
            // four slots (one tile) per thread
            uint4 c = make_uint4(0, 0, 0, 0);
            if (e0 + 4u * tid < e_end) c = *reinterpret_cast<const uint4 *>(slot_cnt + e0 + 4u * tid);
            const uint32_t mine = c.x + c.y + c.z + c.w;
            const uint32_t incl = wave_scan_incl(mine);
            __syncthreads();                                                                // (the previous chunk's s_off / s_w are no longer read)
            if (lane == 63) s_w[wave] = incl;
            __syncthreads();
            const uint32_t wbefore = (wave > 0 ? s_w[0] : 0u) + (wave > 1 ? s_w[1] : 0u) + (wave > 2 ? s_w[2] : 0u);
            const uint32_t x0 = wbefore + incl - mine;
            s_off[4u * tid] = x0; s_off[4u * tid + 1u] = x0 + c.x; s_off[4u * tid + 2u] = x0 + c.x + c.y; s_off[4u * tid + 3u] = x0 + c.x + c.y + c.z;
            if (tid == 255u) s_off[1024] = x0 + mine;
            __syncthreads();
            const uint32_t total = s_off[1024];
            const uint64_t base = cur + before + done;
            for (uint32_t r = tid; r < total; r += 256u) {
                uint32_t q = 0;                                                             // the largest q with s_off[q] <= r
#pragma unroll
                for (uint32_t step = 512; step > 0; step >>= 1) q += (s_off[q + step] <= r) ? step : 0u;
                const SiteRec rec = scratch[(size_t)(e0 + q) * slot_w + (r - s_off[q])];
                out_pos[base + r] = rec.pos;
                out_nc[base + r] = rec.n_conc;
                out_nd[base + r] = rec.n_disc;
                out_pdr[base + r] = (float)rec.n_disc / ((float)rec.n_conc + (float)rec.n_disc);   // pdr.rs:47-49
            }
            done += total;
        }
    }
    if (w + 1u != G) return;
    // the batch's last workgroup commits it (see k_gather)
    const uint32_t batch_total = fin_only ? 0u : before + run_rows[w];
    if (tid == 0) {
        const uint32_t nb = reset_first ? 0u : st->n_batches;
        st->n_sites = cur + batch_total;
        batch_cnt[nb] = batch_total;
        st->n_batches = nb + 1;
        if (next_st) next_st->cur_base = cur + batch_total;
        uint32_t e = reset_first ? 0u : st->err;
        if (lane_st) { e |= lane_st->err; lane_st->err = 0; }
        if (lane_st || reset_first) st->err = e;
    }
    if (want_lpmd || reset_first) {
        __shared__ unsigned long long s_l[4][4];
        unsigned long long x[4] = {0, 0, 0, 0};
        if (want_lpmd) for (uint32_t r = tid; r < G; r += 256u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] += run_lpmd[(size_t)r * 4u + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x[k] += __shfl_down(x[k], o, 64);
            if (lane == 0) s_l[wave][k] = x[k];
        }
        __syncthreads();
        if (tid < 4) st->lpmd[tid] = (reset_first ? 0ll : st->lpmd[tid]) + (long long)(s_l[0][tid] + s_l[1][tid] + s_l[2][tid] + s_l[3][tid]);
    }
}

// ---------------------------------------------------------------------------------------------
// One wave per tile.  base = cur_base + rows of the buckets before the tile's bucket + rows of the bucket's
// earlier tiles (a few coalesced loads per lane, two wave reductions); then scratch -> final sorted SoA with
// the reference's f32 PDR (pdr.rs:47-49).  The wave of the batch's last tile also commits the batch to
// DevState: n_sites, the batch's row count, and the LPMD counters summed over the buckets.  Nobody in this
// launch reads what it writes (cur_base is set by the next batch's k_build_index).
// fin_only (LPMD-only passes): a single wave that only commits.
// ---------------------------------------------------------------------------------------------
constexpr int GATHER_WAVES = 4;   // tiles per workgroup (one wave each)
__global__ __launch_bounds__(64 * GATHER_WAVES) void k_gather(const SiteRec *__restrict__ scratch,
                                               const uint32_t *__restrict__ tile_cnt,
                                               const unsigned long long *__restrict__ bucket, uint32_t nbk,
                                               uint32_t ntiles, int fin_only, int want_lpmd,
                                               DevState *__restrict__ st, const DevState *__restrict__ base_st,
                                               DevState *__restrict__ next_st, DevState *__restrict__ lane_st, int reset_first,
                                               uint32_t *__restrict__ batch_cnt,
                                               uint32_t tile_w,
                                               int32_t *__restrict__ out_pos, float *__restrict__ out_pdr,
                                               uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    const uint32_t t = fin_only ? ntiles - 1 : blockIdx.x * GATHER_WAVES + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t bk = t >> TILE_BUCKET_SHIFT;
    // Everything the wave needs first is requested before anything is used: its own row count, the batch's base, the bucket
    // sums before its bucket, the (up to 255 = 4 x 64) earlier tiles of its bucket and -- speculatively, scratch holds tile_w
    // >= 64 rows per tile whether or not the tile wrote them -- its first 64 rows.  (The earlier-tiles loop used to wait for
    // each of its four loads in turn: five dependent round trips per wave.)
    static_assert(TILE_BUCKET_SHIFT == 8, "a bucket's earlier tiles are four loads per lane");
    const SiteRec *__restrict__ src = scratch + (size_t)t * tile_w;
    SiteRec r0; r0.pos = 0; r0.n_conc = 0; r0.n_disc = 0; r0.pad = 0;
    if (!fin_only) r0 = src[lane];
    const uint32_t n = tile_cnt[t];
    const uint64_t cur = reset_first ? 0ull : base_st->cur_base;
    const uint32_t u0 = (bk << TILE_BUCKET_SHIFT) + lane;
    uint32_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = (u0 + 64u * k < t) ? tile_cnt[u0 + 64u * k] : 0u;
    uint32_t part = 0;                                  // rows of one batch fit 32 bits (<= region positions)
    for (uint32_t b = lane; b < bk; b += 64) part += (uint32_t)bucket[b];
    part += (x[0] + x[1]) + (x[2] + x[3]);
    const uint32_t before = wave_sum(part);
    const uint64_t base = cur + before;
    if (!fin_only) {
        for (uint32_t j = lane; j < n; j += 64) {
            const SiteRec r = j < 64 ? r0 : src[j];
            out_pos[base + j] = r.pos;
            out_nc[base + j] = r.n_conc;
            out_nd[base + j] = r.n_disc;
            out_pdr[base + j] = (float)r.n_disc / ((float)r.n_conc + (float)r.n_disc);
        }
    }
    if (t != ntiles - 1) return;
    const uint32_t total = before + n;
    // Pipelined batches (base_st = lane_st = the lane's block, next_st = the other lane's): this gather runs strictly after the one
    // before it, so it may read and write the job state; the next batch's gather finds its row base in ITS lane's block (other
    // waves of this launch may still be reading base_st->cur_base: nobody in a launch reads what the launch writes).  The batch's
    // own error bits (index: unsorted, tile: span) were collected in the lane's block, because the job's may be cleared by a
    // mth_reset that sits in the chain between two batches; reset_first: this batch is the first after such a reset.
    if (lane == 0) {
        const uint32_t nb = reset_first ? 0u : st->n_batches;
        st->n_sites = cur + total;
        batch_cnt[nb] = total;
        st->n_batches = nb + 1;
        if (next_st) next_st->cur_base = cur + total;
        uint32_t e = reset_first ? 0u : st->err;
        if (lane_st) { e |= lane_st->err; lane_st->err = 0; }
        if (lane_st || reset_first) st->err = e;
    }
    if (want_lpmd || reset_first) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = 0;
            if (want_lpmd) for (uint32_t b = lane; b < nbk; b += 64) x += bucket[nbk + (size_t)b * 4 + k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if (lane == 0) st->lpmd[k] = (reset_first ? 0ll : st->lpmd[k]) + (long long)x;
        }
    }
}

// k_gather, second form (round 4): one 256-thread workgroup per 64 consecutive tiles.  The per-tile form above starts a wave per tile
// (14 311 on config 2), and every wave loads its bucket's earlier counts again (up to 1 KiB) to find its base; beside a tile kernel
// those waves queue for slots and the gather takes 20 us instead of 11.  Here the 64 counts are scanned once, the rows of the 64
// tiles are ONE flat range (row r -> tile by a binary search in the 65 offsets in LDS) copied 256 at a time, every load independent.
constexpr int GATHER2_TILES = 64;
__global__ __launch_bounds__(256) void k_gather2(const SiteRec *__restrict__ scratch, const uint32_t *__restrict__ tile_cnt,
                                                 const unsigned long long *__restrict__ bucket, uint32_t nbk, uint32_t ntiles, int fin_only, int want_lpmd,
                                                 DevState *__restrict__ st, const DevState *__restrict__ base_st, DevState *__restrict__ next_st,
                                                 DevState *__restrict__ lane_st, int reset_first, uint32_t *__restrict__ batch_cnt, uint32_t tile_w,
                                                 int32_t *__restrict__ out_pos, float *__restrict__ out_pdr, uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    static_assert(TILE_BUCKET_SHIFT == 8 && GATHER2_TILES == 64, "a bucket's earlier tiles are at most 192: one per thread of waves 1..3");
    __shared__ uint32_t s_off[GATHER2_TILES + 1];
    __shared__ uint32_t s_part[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t t0 = fin_only ? ntiles : blockIdx.x * GATHER2_TILES;
    const uint32_t nt = fin_only ? 0u : min((uint32_t)GATHER2_TILES, ntiles - t0);
    const uint32_t bk = t0 >> TILE_BUCKET_SHIFT;
    const uint64_t cur = reset_first ? 0ull : base_st->cur_base;
    uint32_t part = 0, mine = 0;
    if (!fin_only) {
        if (wave == 0) mine = lane < nt ? tile_cnt[t0 + lane] : 0u;
        else { const uint32_t u = (bk << TILE_BUCKET_SHIFT) + (tid - 64u); if (u < t0) part = tile_cnt[u]; }
        for (uint32_t b = tid; b < bk; b += 256u) part += (uint32_t)bucket[b];            // rows of one batch fit 32 bits
    }
    const uint32_t incl = wave_scan_incl(mine);
    if (wave == 0) { s_off[lane + 1] = incl; if (lane == 0) s_off[0] = 0u; }
    const uint32_t wsum = wave_sum(part);
    if (lane == 0) s_part[wave] = wsum;
    __syncthreads();
    const uint32_t before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const uint32_t total = s_off[nt];
    const uint64_t base = cur + before;
    for (uint32_t r = tid; r < total; r += 256u) {
        uint32_t q = 0;                                                                   // the largest q with s_off[q] <= r
#pragma unroll
        for (uint32_t step = GATHER2_TILES / 2; step > 0; step >>= 1) q += (s_off[q + step] <= r) ? step : 0u;
        const SiteRec rec = scratch[(size_t)(t0 + q) * tile_w + (r - s_off[q])];
        out_pos[base + r] = rec.pos;
        out_nc[base + r] = rec.n_conc;
        out_nd[base + r] = rec.n_disc;
        out_pdr[base + r] = (float)rec.n_disc / ((float)rec.n_conc + (float)rec.n_disc);   // pdr.rs:47-49
    }
    if (t0 + nt != ntiles) return;
    // the batch's last workgroup commits it (see k_gather)
    const uint32_t batch_total = before + total;
    if (tid == 0) {
        const uint32_t nb = reset_first ? 0u : st->n_batches;
        st->n_sites = cur + batch_total;
        batch_cnt[nb] = batch_total;
        st->n_batches = nb + 1;
        if (next_st) next_st->cur_base = cur + batch_total;
        uint32_t e = reset_first ? 0u : st->err;
        if (lane_st) { e |= lane_st->err; lane_st->err = 0; }
        if (lane_st || reset_first) st->err = e;
    }
    if (wave == 0 && (want_lpmd || reset_first)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = 0;
            if (want_lpmd) for (uint32_t b = lane; b < nbk; b += 64) x += bucket[nbk + (size_t)b * 4 + k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if (lane == 0) st->lpmd[k] = (reset_first ? 0ll : st->lpmd[k]) + (long long)x;
        }
    }
}

// first pipelined batch since ctx->stream was last joined: its lane's row base is the job's row count as that stream leaves it
__global__ void k_pipe_seed(const DevState *__restrict__ st, DevState *__restrict__ lane_st) {
    lane_st->cur_base = st->n_sites;
    lane_st->err = 0;
}

// ---------------------------------------------------------------------------------------------
constexpr int DENSE_TILE_SHIFT = 12;   // the dense kernel's 4096-bp tiles
constexpr int TILE_MARGIN = 256;   // counter margins on either side of a tile for batches with max_span <= 256
template <int W, int B, typename RelT>
static void launch_tile(const TileArgs &a, uint32_t ntiles, hipStream_t s) {
    const uint32_t grid = ((ntiles + 7) / 8) * 8;   // whole rows of 8 XCDs (remap in the kernel)
    static const bool no_margin = getenv("MTH_TILE_NO_MARGIN") != nullptr;   // A/B switch
    if (a.max_span <= TILE_MARGIN && !no_margin) hipLaunchKernelGGL((k_pdr_lpmd_tile<W, B, 8, RelT, TILE_MARGIN>), dim3(grid), dim3(B), 0, s, a, ntiles);
    else hipLaunchKernelGGL((k_pdr_lpmd_tile<W, B, 8, RelT, 0>), dim3(grid), dim3(B), 0, s, a, ntiles);
}

// run form: the chip's resident slots (8 workgroups per CU), or fewer for a batch with fewer tiles
static uint32_t runs_grid(uint32_t ntiles) {
    static const uint32_t slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char *e = getenv("MTH_RUNS_WGS_PER_CU");
        const int per = e && atoi(e) > 0 ? atoi(e) : MTH_RUNS_OCC;
        return (uint32_t)std::max(cus, 1) * (uint32_t)per;
    }();
    return std::max(1u, std::min(slots, ntiles));
}
static void launch_runs(const TileArgs &a, const RunArgs &ra, uint32_t ntiles, uint32_t G, hipStream_t s) {
    static const bool no_margin = getenv("MTH_TILE_NO_MARGIN") != nullptr;   // A/B switch
    if (a.max_span <= TILE_MARGIN && !no_margin) hipLaunchKernelGGL((k_pdr_lpmd_runs<(1 << DENSE_TILE_SHIFT), TILE_MARGIN>), dim3(G), dim3(256), 0, s, a, ra, ntiles);
    else hipLaunchKernelGGL((k_pdr_lpmd_runs<(1 << DENSE_TILE_SHIFT), 0>), dim3(G), dim3(256), 0, s, a, ra, ntiles);
}

// the linear read index alone (for kernels that find a tile's candidate reads without running the PDR/LPMD pass):
// idx lives in ctx->idx; same origin and quantum as launch_pdr_lpmd builds
// first kernel of a PDR + LPMD batch whose index exists already (prepared batches): what k_build_index does beside the index -- the row
// base, the bucket sums, safe_hi and the batch's own findings (unsorted) into the lane's block
__global__ __launch_bounds__(256) void k_batch_begin(DevState *__restrict__ lane_st, DevState *__restrict__ cst, const DevState *__restrict__ prep_st,
                                                     unsigned long long *__restrict__ bucket_sums, uint32_t n_bucket_words) {
    if (threadIdx.x == 0) {
        if (cst) cst->cur_base = cst->n_sites;
        lane_st->safe_hi = prep_st->safe_hi;
        if (prep_st->err) atomicOr(&lane_st->err, prep_st->err);
    }
    for (uint32_t w = threadIdx.x; w < n_bucket_words; w += 256) bucket_sums[w] = 0ull;
}
__global__ void k_batch_err(DevState *__restrict__ st, const DevState *__restrict__ prep_st) { if (prep_st->err) atomicOr(&st->err, prep_st->err); }

int build_fine_index(mth_ctx *ctx, const mth_batch_t &b, int32_t idx_base, uint32_t nq, uint32_t *idx, DevState *st) {
    hipStream_t s = ctx->stream;
    LaunchTimer lt(ctx, K_INDEX);
    const uint32_t nb = (b.n_reads / 4 + 1 + BLOCK * IDX_GROUPS - 1) / (BLOCK * IDX_GROUPS);
    hipLaunchKernelGGL(k_build_index<1>, dim3(nb), dim3(BLOCK), 0, s, b.read_start, b.n_reads, idx_base, 0, (int)IDX_QSHIFT, nq,
                       (int)((reinterpret_cast<uintptr_t>(b.read_start) & 15u) == 0), idx, (uint32_t *)nullptr, st,
                       (DevState *)nullptr, (unsigned long long *)nullptr, 0u, b.cpg_off, b.n_cpgs);
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

int build_read_index(mth_ctx *ctx, const mth_batch_t &b, int tile_w, int32_t &idx_base, uint32_t &ntiles) {
    hipStream_t s = ctx->stream;
    const int64_t region_len = (int64_t)b.region_end - b.region_beg;
    ntiles = (uint32_t)((region_len + tile_w - 1) / tile_w);
    const int32_t ext = ((b.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    idx_base = b.region_beg - ext;
    const uint32_t nq = (uint32_t)(((int64_t)ntiles * tile_w + ext) >> IDX_QSHIFT) + 2;
    ctx->cur_idx = nullptr;
    if (Prepared *pr = ctx->cur_prep) {
        // a prepared batch: its index was built once (same origin and quantum; it covers every tile width up to 65536)
        if (pr->idx_base == idx_base && nq <= pr->nq) {
            ctx->cur_idx = pr->idx.as<uint32_t>();
            hipLaunchKernelGGL(k_batch_err, dim3(1), dim3(1), 0, s, ctx->d_state, (const DevState *)pr->st);      // an unsorted batch is this measure's error too
            return MTH_OK;
        }
    }
    MTH_HIP(ctx, ctx->idx.reserve((size_t)(nq + 1) * 4, s));
    LaunchTimer lt(ctx, K_INDEX);
    const uint32_t nb = (b.n_reads / 4 + 1 + BLOCK * IDX_GROUPS - 1) / (BLOCK * IDX_GROUPS);
    hipLaunchKernelGGL(k_build_index<1>, dim3(nb), dim3(BLOCK), 0, s, b.read_start, b.n_reads, idx_base, 0, (int)IDX_QSHIFT, nq,
                       (int)((reinterpret_cast<uintptr_t>(b.read_start) & 15u) == 0), ctx->idx.as<uint32_t>(), (uint32_t *)nullptr, ctx->d_state,
                       ctx->d_state, (unsigned long long *)nullptr, 0u, (const uint32_t *)nullptr, 0u);   // cur_base is rewritten by the next PDR batch's own index build
    return MTH_OK;
}

// ---------------------------------------------------------------------------------------------
// Pipelined batches.  A job is a sequence of batches (contigs, regions).  Within one batch the three kernels depend on each
// other, but batch k+1's index build and tile kernel need nothing of batch k: only the gathers form a chain (row bases, per-batch
// counts and the LPMD totals are job state).  Consecutive device-resident batches therefore alternate between two LANES -- a
// stream, a set of work buffers (index, per-tile counts, bucket sums, scratch rows) and a 64-byte block for the batch's own
// error bits / safe_hi / row base each -- and each gather waits for the event behind the previous one:
//     lane 0:  index 0 | tile 0 ........ | gather 0 | index 2 | tile 2 ........ |        gather 2
//     lane 1:            index 1 | tile 1 ........ |  gather 1 | index 3 | tile 3 ...
// so a tile kernel's drain (a third of a 10 M-read launch is filling and draining the chip, profiles/HISTORY.md section 6) is covered by the
// next batch's fill, and the small index / gather kernels run beside a tile kernel instead of between two.  ctx->stream is
// ordered behind the lanes by mth::enter() at every other entry point; the lanes wait for ctx->stream's work as of the call
// (the caller's producers of the batch arrays).
static int pipe_begin(mth_ctx *ctx, int &li) {
    if (!ctx->pipe_in) {
        MTH_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe_in, hipEventDisableTiming));
        for (PdrLane &l : ctx->lane) {
            MTH_HIP(ctx, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
            MTH_HIP(ctx, hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
            MTH_HIP(ctx, hipMalloc((void **)&l.st, sizeof(DevState)));
            MTH_HIP(ctx, hipMemsetAsync(l.st, 0, sizeof(DevState), l.stream));
        }
    }
    li = ctx->pipe_next;
    ctx->pipe_next ^= 1;
    PdrLane &L = ctx->lane[li];
    // the caller's producers of the batch arrays, and whatever a join put on ctx->stream: the lane waits for them -- unless that
    // stream is idle, which one query tells (a cross-stream wait costs the lane's queue a barrier packet)
    if (!ctx->pipe_active || hipStreamQuery(ctx->stream) != hipSuccess) {
        MTH_HIP(ctx, hipEventRecord(ctx->pipe_in, ctx->stream));
        MTH_HIP(ctx, hipStreamWaitEvent(L.stream, ctx->pipe_in, 0));
    }
    if (!ctx->pipe_active) {
        hipLaunchKernelGGL(k_pipe_seed, dim3(1), dim3(1), 0, L.stream, ctx->d_state, L.st);
        ctx->pipe_active = true;
        ctx->pipe_tail = -1;
    }
    L.used = true;
    return MTH_OK;
}

int launch_pdr_lpmd(mth_ctx *ctx, const mth_batch_t &b, const mth_pdr_lpmd_params_t &p, const TileSink *sink, bool pipelined) {
    if (sink) pipelined = false;
    int li = 0;
    if (pipelined) { const int rc = pipe_begin(ctx, li); if (rc) return rc; }
    PdrLane *L = pipelined ? &ctx->lane[li] : nullptr;
    hipStream_t s = L ? L->stream : ctx->stream;
    // lane 0 (and every unpipelined batch) works in the context's own buffers, lane 1 in its second set
    DevBuf &b_idx = li ? L->idx : ctx->idx, &b_tile_cnt = li ? L->tile_cnt : ctx->tile_cnt;
    DevBuf &b_bucket = li ? L->tile_bucket : ctx->tile_bucket, &b_scratch = li ? L->scratch : ctx->scratch;
    DevState *lane_st = L ? L->st : ctx->d_state;        // error bits, safe_hi of the batch in flight
    // where the compacted rows and their counters go: the PDR result columns by default, or a
    // caller-supplied sink (site discovery for the site-walk measures)
    DevState *cst = sink ? sink->st : ctx->d_state;
    uint32_t *bcnt = sink ? sink->batch_cnt : ctx->batch_cnt.as<uint32_t>();
    int32_t *o_pos = sink ? sink->pos : ctx->out_pos.as<int32_t>();
    float *o_pdr = sink ? sink->pdr : ctx->out_pdr.as<float>();
    uint32_t *o_nc = sink ? sink->nc : ctx->out_nc.as<uint32_t>();
    uint32_t *o_nd = sink ? sink->nd : ctx->out_nd.as<uint32_t>();
    const int64_t region_len = (int64_t)b.region_end - b.region_beg;
    // Two forms of the tile kernel.  Dense batches: one LDS counter per position, 4096-bp tiles (8192-bp tiles, 4 workgroups per CU:
    // config 3 0.1655 -> 0.1956 ms, config 2 0.0858 -> 0.1069; profiles/r03_stream_kernel.md).  Sparse batches (WGBS depth: few calls
    // per read and few reads per 4096 bp, so that a tile's fixed chain of round trips is what the kernel waits for): hashed sites,
    // 16384- to 65536-bp tiles (mth_pdr_wide.hip).  MTH_PDR_WIDE=0 / 14 / 15 / 16 overrides the choice (tests, A/B).
    int wide_shift = 0;
    if (b.n_reads && region_len > 0) {
        const double sites_per_bp = (double)b.n_cpgs / (double)b.n_reads / (double)std::max(b.max_span, 1);
        const double reads_per_bp = (double)b.n_reads / (double)region_len;
        // (1024 slots: up to ~0.6 of them for the sites a tile can expect; a denser stretch is redone in halves)
        // A hashed insert (compare-and-swap + adds) costs more than the dense form's one LDS add: the wide form only pays while few
        // calls are inserted.  Expected insertions per read = E[n; n >= min_cpgs] = lambda P(N >= min_cpgs - 1) for Poisson calls per
        // read: 0.22 under the CLI defaults at config-3 density (0.117 against 0.170 ms), 1.37 for FDRP's site discovery (min_cpgs 1:
        // 0.202 against 0.153 ms -- that pass keeps the dense form).
        double ins_per_read = 0.0;
        if (p.want_pdr) {
            const double lam = (double)b.n_cpgs / (double)b.n_reads;
            const int kmin = (int)std::min<uint32_t>(std::max<uint32_t>(p.pdr_min_cpgs, 1u), 64u) - 1;     // P(N >= kmin)
            double term = std::exp(-lam), cdf = 0.0;
            for (int k = 0; k < kmin; ++k) { cdf += term; term *= lam / (double)(k + 1); }
            ins_per_read = lam * std::max(0.0, 1.0 - cdf);
        }
        // Depth: measured on a chr1-sized contig at density 0.0091 -- 16 M reads (10x) dense 0.170 / wide 0.113 ms, 24 M 0.201 / 0.160,
        // 32 M 0.238 / 0.196, 48 M (29x) 0.295 / 0.277, each with a gather of 0.027 / 0.008 on top; 65536-bp tiles only lead at 10x.
        if (sites_per_bp <= 0.012 && reads_per_bp * 4096.0 <= 900.0 && ins_per_read <= 0.6)
        {
            wide_shift = sites_per_bp * 65536.0 <= 0.6 * 1024 && reads_per_bp * 65536.0 <= 4500.0 ? 16 : sites_per_bp * 32768.0 <= 0.6 * 1024 ? 15 : 14;
            // ... and enough tiles to fill the chip twice over (7 workgroups on each of 256 CUs): a short contig takes narrower tiles
            // (config 3's 24 contigs: 2.33 -> 2.20 ms; the same rule made the quartet / pairs kernels slower -- 2.36 -> 2.62, 2.72 -> 2.95 ms --
            // whose persistent workgroups gain more from the wider tile than they lose to a half-filled last round)
            while (wide_shift > 14 && (region_len >> wide_shift) < 2 * 7 * 256) --wide_shift;
        }
    }
    if (const char *e = getenv("MTH_PDR_WIDE")) { const int k = atoi(e); wide_shift = k >= 14 && k <= 16 ? k : 0; }
    const int tile_w = wide_shift ? 1 << wide_shift : 4096;
    // Wide form: the tile need not be as wide as its slice.  A chr1-sized contig is 2.12 rounds of 65536-position tiles over the chip's
    // 1 792 resident workgroups (7 on each of 256 CUs) and ends on a nearly empty third round; tiles of region / (3 x 1 792) positions
    // make three full rounds of smaller tiles.  Only where the last round would be less than half full, and never below
    // half the slice (the chain of round trips per tile is what the wide form exists to pay less often).  MTH_PDR_WIDE_W forces a width.
    uint32_t tile_w_rt = 0;
    if (wide_shift) {
        const double slots = 7.0 * 256.0, rounds = (double)region_len / (double)tile_w / slots;
        const double frac = rounds - std::floor(rounds);
        if (rounds > 1.0 && rounds < 8.0 && frac > 0.02 && frac < 0.5) {
            const uint64_t w = (uint64_t)std::ceil((double)region_len / (std::ceil(rounds) * slots));
            const uint32_t wr = (uint32_t)((w + 63) & ~63ull);
            if (wr >= (uint32_t)tile_w / 2 && wr < (uint32_t)tile_w) tile_w_rt = wr;
        }
        if (const char *e = getenv("MTH_PDR_WIDE_W")) { const int k = atoi(e); tile_w_rt = (k >= 1024 && k < tile_w) ? (uint32_t)k & ~63u : 0u; }
    }
    const uint32_t tile_step = tile_w_rt ? tile_w_rt : (uint32_t)tile_w;
    const uint32_t ntiles = (uint32_t)((region_len + tile_step - 1) / tile_step);
    if (ntiles == 0) return MTH_OK;
    // index origin: a whole number of quanta below the region so that halo reads are indexed
    const int32_t ext = ((b.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    const int32_t idx_base = b.region_beg - ext;
    const uint32_t nq = (uint32_t)(((int64_t)ntiles * tile_step + ext) >> IDX_QSHIFT) + 2;

    // The dense tile kernel's own index: one entry per tile in each of two families (k_build_index<2>).  Site discovery for the walk
    // measures (sink) leaves the fine index behind for them; the wide form looks up arbitrary stretch bounds.  MTH_COARSE_INDEX=0: A/B.
    static const bool coarse_off = getenv("MTH_COARSE_INDEX") && atoi(getenv("MTH_COARSE_INDEX")) == 0;
    // a prepared batch brings its fine index: no index kernel in this call (k_batch_begin does what else that kernel did)
    Prepared *prep = ctx->cur_prep;
    if (prep && !(prep->idx_base == idx_base && nq <= prep->nq)) prep = nullptr;
    const bool coarse = !sink && wide_shift == 0 && !coarse_off && !prep;
    const uint32_t coarse_stride = (ntiles + 2u + 3u) & ~3u;
    if (!prep) MTH_HIP(ctx, b_idx.reserve(coarse ? (size_t)coarse_stride * 2 * 4 : (size_t)(nq + 1) * 4, s));
    const uint32_t *fine_idx = prep ? prep->idx.as<uint32_t>() : nullptr;
    ctx->cur_idx = fine_idx;                      // (the walks that follow a discovery pass read the index this call used)
    // the run form of the dense kernel (k_pdr_lpmd_runs; persistent workgroups, no index, no atomics): opt-in with MTH_TILE_RUNS=1 --
    // parity-green on every PDR / LPMD case, measured SLOWER than the one-tile-per-workgroup form on config 2 (0.127 against 0.084 ms;
    // profiles/r05_persistent.md).  8-bit relative positions, call offsets that fit a buffer descriptor's 32-bit byte offset.
    const char *runs_env = getenv("MTH_TILE_RUNS");          // (read per call: the tests switch it per case)
    const bool runs = wide_shift == 0 && b.cpg_rel != nullptr && b.n_cpgs < (1u << 30) && runs_env && atoi(runs_env) == 1;
    const uint32_t G = runs ? runs_grid(ntiles) : 0u;
    MTH_HIP(ctx, b_tile_cnt.reserve((size_t)ntiles * 4 * (runs ? 4 : 1), s));
    const uint32_t nbk = (ntiles + (1u << TILE_BUCKET_SHIFT) - 1) >> TILE_BUCKET_SHIFT;
    const uint32_t n_bucket_words = nbk * 5u;      // rows, 4 LPMD sums per bucket
    // run form: [G + 1] first tiles, [G] rows (32-bit words), then [G][4] LPMD sums
    const size_t run_words = ((size_t)G + 2u) / 2u + ((size_t)G + 1u) / 2u;
    MTH_HIP(ctx, b_bucket.reserve(((size_t)n_bucket_words + run_words + (size_t)G * 4u) * sizeof(unsigned long long), s));
    if (p.want_pdr) MTH_HIP(ctx, b_scratch.reserve((size_t)ntiles * (size_t)tile_w * sizeof(SiteRec), s));

    if (prep) {
        if (runs) hipLaunchKernelGGL(k_batch_err, dim3(1), dim3(1), 0, s, lane_st, (const DevState *)prep->st);
        else hipLaunchKernelGGL(k_batch_begin, dim3(1), dim3(256), 0, s, lane_st, L ? (DevState *)nullptr : cst, (const DevState *)prep->st,
                                b_bucket.as<unsigned long long>(), n_bucket_words);
    } else if (!runs || sink) {      // (the run form needs no index; site discovery leaves the fine one behind for the walks)
        LaunchTimer lt(ctx, K_INDEX);
        const uint32_t nb = (b.n_reads / 4 + 1 + BLOCK * IDX_GROUPS - 1) / (BLOCK * IDX_GROUPS);
        const int al16 = (int)((reinterpret_cast<uintptr_t>(b.read_start) & 15u) == 0);
        if (coarse)
            hipLaunchKernelGGL(k_build_index<2>, dim3(nb), dim3(BLOCK), 0, s, b.read_start, b.n_reads, b.region_beg - b.max_span + 1, b.region_beg + 1,
                               DENSE_TILE_SHIFT, ntiles + 1u, al16, b_idx.as<uint32_t>(), b_idx.as<uint32_t>() + coarse_stride, lane_st,
                               L ? (DevState *)nullptr : cst, b_bucket.as<unsigned long long>(), n_bucket_words, b.cpg_off, b.n_cpgs);
        else
            hipLaunchKernelGGL(k_build_index<1>, dim3(nb), dim3(BLOCK), 0, s, b.read_start, b.n_reads, idx_base, 0, (int)IDX_QSHIFT, nq, al16,
                               b_idx.as<uint32_t>(), (uint32_t *)nullptr, lane_st, L ? (DevState *)nullptr : cst,
                               b_bucket.as<unsigned long long>(), n_bucket_words, b.cpg_off, b.n_cpgs);
    }
    TileArgs a;
    a.read_start = b.read_start; a.read_mapq = b.read_mapq; a.cpg_off = b.cpg_off; a.cpg_pos = b.cpg_pos;
    a.cpg_rel = b.cpg_rel ? (const void *)b.cpg_rel : (const void *)b.cpg_rel16;
    a.idx = fine_idx ? fine_idx : b_idx.as<uint32_t>(); a.idx2 = coarse ? b_idx.as<uint32_t>() + coarse_stride : nullptr; a.st = lane_st;
    a.tile_cnt = b_tile_cnt.as<uint32_t>(); a.bucket = b_bucket.as<unsigned long long>(); a.nbk = nbk;
    a.scratch = b_scratch.as<SiteRec>();
    a.region_beg = b.region_beg; a.region_end = b.region_end; a.idx_base = idx_base; a.max_span = b.max_span;
    a.n_reads = b.n_reads; a.n_cpgs = b.n_cpgs;
    a.min_cov = p.pdr_min_depth > 1 ? p.pdr_min_depth : 1;
    a.min_cpgs = p.pdr_min_cpgs;
    a.min_dist = p.lpmd_min_distance; a.max_dist = p.lpmd_max_distance;
    a.pdr_min_qual = p.pdr_min_qual; a.lpmd_min_qual = p.lpmd_min_qual;
    a.want_pdr = p.want_pdr; a.want_lpmd = p.want_lpmd;
    a.tile_w_rt = tile_w_rt;
    RunArgs ra;
    ra.run_tile0 = reinterpret_cast<uint32_t *>(b_bucket.as<unsigned long long>() + n_bucket_words);
    ra.run_rows = ra.run_tile0 + (((size_t)G + 2u) / 2u) * 2u;
    ra.run_lpmd = b_bucket.as<unsigned long long>() + n_bucket_words + run_words;
    ra.cst = L ? (DevState *)nullptr : cst;
    ra.trace = nullptr;
#ifdef MTH_RUNS_TRACE
    static unsigned long long *d_rtrace = nullptr;
    if (!d_rtrace) (void)hipMalloc((void **)&d_rtrace, 8 * 8 * 4 * 4096);
    ra.trace = G <= 4096 ? d_rtrace : nullptr;
#endif
    ra.hint_stride = (uint32_t)std::max(1.0, std::ceil((double)b.n_reads / (double)ntiles * (1.0 + (double)b.max_span / (double)tile_w) * 1.3 / 63.0));
#ifdef MTH_TILE_TRACE
    static unsigned long long *d_ttrace = nullptr;
    if (!d_ttrace) (void)hipMalloc((void **)&d_ttrace, 8 * 8 * 262144);
    a.trace = ntiles <= 262144 ? d_ttrace : nullptr;
#endif
    {
        LaunchTimer lt(ctx, wide_shift ? K_WIDE : K_TILE);
        const bool r8 = b.cpg_rel != nullptr;
        if (wide_shift) launch_tile_wide(a, ntiles, wide_shift, r8, s);
        else if (runs) launch_runs(a, ra, ntiles, G, s);
        else if (r8) launch_tile<(1 << DENSE_TILE_SHIFT), 256, uint8_t>(a, ntiles, s); else launch_tile<(1 << DENSE_TILE_SHIFT), 256, uint16_t>(a, ntiles, s);
    }
    {
        LaunchTimer lt(ctx, K_GATHER);
        int reset_first = 0;
        if (L) {
            // the chain: behind the previous batch's gather (the other lane's; this lane's own earlier work precedes in stream order)
            if (ctx->pipe_tail >= 0 && ctx->pipe_tail != li) MTH_HIP(ctx, hipStreamWaitEvent(s, ctx->lane[ctx->pipe_tail].done, 0));
            reset_first = ctx->reset_pending ? 1 : 0;
            ctx->reset_pending = false;
        }
        static const bool gather1 = getenv("MTH_GATHER") && atoi(getenv("MTH_GATHER")) == 1;       // A/B: the per-tile form
        if (runs)
            hipLaunchKernelGGL(k_gather_runs, dim3(p.want_pdr ? G : 1u), dim3(256), 0, s, b_scratch.as<SiteRec>(),
                               b_tile_cnt.as<uint32_t>(), ra.run_tile0, ra.run_rows, ra.run_lpmd, G,
                               p.want_pdr ? 0 : 1, (int)p.want_lpmd, cst, L ? (const DevState *)L->st : (const DevState *)cst,
                               L ? ctx->lane[li ^ 1].st : (DevState *)nullptr, L ? L->st : (DevState *)nullptr, reset_first,
                               bcnt, (uint32_t)tile_w / 4u, o_pos, o_pdr, o_nc, o_nd);
        else if (gather1)
            hipLaunchKernelGGL(k_gather, dim3(p.want_pdr ? (ntiles + GATHER_WAVES - 1) / GATHER_WAVES : 1u), dim3(p.want_pdr ? 64 * GATHER_WAVES : 64), 0, s, b_scratch.as<SiteRec>(),
                               b_tile_cnt.as<uint32_t>(), b_bucket.as<unsigned long long>(), nbk, ntiles,
                               p.want_pdr ? 0 : 1, (int)p.want_lpmd, cst, L ? (const DevState *)L->st : (const DevState *)cst,
                               L ? ctx->lane[li ^ 1].st : (DevState *)nullptr, L ? L->st : (DevState *)nullptr, reset_first,
                               bcnt, (uint32_t)tile_w, o_pos, o_pdr, o_nc, o_nd);
        else
            hipLaunchKernelGGL(k_gather2, dim3(p.want_pdr ? (ntiles + GATHER2_TILES - 1) / GATHER2_TILES : 1u), dim3(256), 0, s, b_scratch.as<SiteRec>(),
                               b_tile_cnt.as<uint32_t>(), b_bucket.as<unsigned long long>(), nbk, ntiles,
                               p.want_pdr ? 0 : 1, (int)p.want_lpmd, cst, L ? (const DevState *)L->st : (const DevState *)cst,
                               L ? ctx->lane[li ^ 1].st : (DevState *)nullptr, L ? L->st : (DevState *)nullptr, reset_first,
                               bcnt, (uint32_t)tile_w, o_pos, o_pdr, o_nc, o_nd);
        if (L) { MTH_HIP(ctx, hipEventRecord(L->done, s)); ctx->pipe_tail = li; }
    }
#ifdef MTH_RUNS_TRACE
    if (getenv("MTH_RUNS_TRACE_OUT") && runs && G <= 4096) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> tt(8 * 4 * (size_t)G);
        (void)hipMemcpy(tt.data(), d_rtrace, tt.size() * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("MTH_RUNS_TRACE_OUT"), "wb");
        if (f) { fwrite(tt.data(), 8, tt.size(), f); fclose(f); }
    }
#endif
#ifdef MTH_TILE_TRACE
    if (getenv("MTH_TILE_TRACE_OUT") && ntiles <= 262144) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> tt(8 * (size_t)ntiles);
        (void)hipMemcpy(tt.data(), d_ttrace, tt.size() * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("MTH_TILE_TRACE_OUT"), "wb");
        if (f) { fwrite(tt.data(), 8, tt.size(), f); fclose(f); }
    }
#endif
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

}  // namespace mth
