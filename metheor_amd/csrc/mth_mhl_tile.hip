// mth_mhl_tile.hip -- MHL (mhl.rs:135-208, 43-73; readutil.rs:147-164) as ONE tile pass over the reads (gfx950).
//
// compute_mhl (mhl.rs:43-73) only needs, per site, two small histograms over the reads that cover it:
//   hn[n]  = covering reads with n CpGs                       ->  D[l] = sum_n hn[n] * max(0, n - l + 1)    (mhl.rs:53-58)
//   hm[m]  = maximal methylated runs of length m in them       ->  S[l] = sum_m hm[m] * max(0, m - l + 1)    (mhl.rs:36-41)
// (D is the reference's f32 running sum of integers: exact and order-free below 2^24, which 65535 reads x 16 stay under.)
// Both are sums over reads, so a site whose covering reads form ONE segment of the reference's stream (mhl.rs:162-173: a read
// whose first CpG lies beyond c flushes c; a later contribution re-opens it) needs no walk at all: every read adds 1 to
// hn[n] and its run lengths to hm[] at each of its calls -- the PDR tile kernel with three LDS atomics per call instead of one.
// The tile keeps its sites in an LDS hash table (key = position; 16 + 16 sixteen-bit bins per slot), so there is no site
// discovery pass either: one pass over the reads replaces the PDR-style discovery, the per-site walk and its staging.
//
// Exactness.  Site c can have more than one segment only if some read k with >= 1 CpG has start_k - 1 <= c < first_cpg(k)
// while a LATER read still calls c (a later contributor j has start_j <= c + 1, a flusher between two contributors starts at or
// before that too).  Two cases:
//   c >= start_k:      the tile marks [start_k, first_cpg(k)) in a position bitmap F; a site under a mark is handed on.  Reads
//                      that call every CpG they overlap (the usual case) mark stretches without sites.
//   c == start_k - 1:  forward reads starting one base past a CpG do this all the time, but only a reverse read with the SAME
//                      start can call c after them.  Each slot keeps the largest index of a contributor that calls its own
//                      start - 1; at the end the few reads with that start and a smaller index are looked at.
// Sites handed on (flag 4), sites covered by a read with more than 16 CpGs and the sites of tiles with more than 8191 candidate
// reads (the 16-bit bins) take k_mhl_walk_big's exact per-site walk (mth_sites.hip) -- conservative hand-ons cost time only.
// A sub-range of a tile with more distinct sites than slots (CpG islands) is redone in halves (down to 256 positions = 256 slots).
//
// Output: per tile, rows sorted by position in a scratch slice (site, mhl, coverage, flag); k_mhl_tile_gather packs the slices
// into the candidate-site arrays the rest of the MHL pipeline (k_mhl_walk_big / _huge, k_mhl_emit) already works on.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "mth_ctx.h"
#include "mth_scan.h"
#include "mth_tile_dev.h"

namespace mth {

struct MhlRec { int32_t pos; float val; uint32_t cov, flags; };
static_assert(sizeof(MhlRec) == sizeof(SiteRec), "the PDR scratch buffer is reused");

struct MhlTileArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    int32_t region_beg, region_end, idx_base, max_span;
    uint32_t n_reads, ntiles, n_cpgs, min_depth, min_cpgs;
    uint8_t min_qual;
    uint8_t force_sub;            // tests: start every tile with 256-position sub-ranges
    uint8_t force_hand_on;        // tests: every site goes to the exact walk
    uint8_t dbg;                  // timing experiments (MTH_MHL_DBG): 1 = no flusher marks, 2 = no contributions, 4 = no row values, 8 = no same-start check
    MhlRec *scratch;              // W rows per tile (the PDR pipeline's 16 bytes per position)
    uint32_t *tile_cnt;           // rows of the tile
    unsigned long long *bucket;   // rows per 256 tiles
    DevState *st;
    unsigned long long *trace;    // -DMTH_MT_TRACE builds: cycles per phase, summed over the sub-ranges (thread 0 of each workgroup)
};

#ifndef MTH_MT_U
#define MTH_MT_U 2
#endif
constexpr int MT_S = 256, MT_B = 256, MT_U = MTH_MT_U, MT_NC = 8, MT_LCAP = 16;
// slot h: tkey[h] = position; taux[h] bit 31 = "a read with > 16 CpGs calls it", low bits = largest (index - lo + 1) of a contributor
// calling its own start - 1; thist[17 h ..]: words 0..7 hn, 8..15 hm (bin b in half (b - 1) & 1 of word (b - 1) >> 1); the 17th word
// is padding: with 16 the slots' words fall on 4 of the 64 LDS banks (measured: contributions 0.17 -> 0.22 ms)
constexpr int MT_HW = 17;
constexpr int MT_Q = 1024;         // contributing reads a sub-range may queue (more: the sub-range is halved)
constexpr uint32_t MT_EMPTY = 0xffffffffu;
constexpr uint32_t MT_HEAVY = 8191;   // candidate reads a sub-range may have with 16-bit bins (a read adds up to 8 to one hm bin)

// A contributing read of a `heavy` stretch: a slot per called position, a 32-bit count in the slot's first histogram word.  Out of
// line: the stretch is rare and its registers would otherwise be live across the whole kernel (21 spilled VGPRs when it was inline).
__device__ __noinline__ bool mhl_count_only(const uint32_t *__restrict__ cpg_pos, const uint32_t o0, const uint32_t o1, const uint32_t sm1,
                                            const uint32_t max_span, const uint32_t P0, const uint32_t Wp, uint32_t *tkey, uint32_t *thist,
                                            uint32_t &bad) {
    bool over = false;
    for (uint32_t k = o0; k < o1; ++k) {
        const uint32_t pw = cpg_pos[k] & 0x7fffffffu, d = pw - P0;
        bad |= (pw - sm1 > max_span) ? 1u : 0u;
        if (d >= Wp) continue;
        uint32_t h = (d >> 1) & (MT_S - 1), probes = 0;
        bool placed = false;
        while (probes++ < (uint32_t)MT_S) {
            const uint32_t cur = atomicCAS(&tkey[h], MT_EMPTY, pw);
            if (cur == MT_EMPTY || cur == pw) { atomicAdd(&thist[h * MT_HW], 1u); placed = true; break; }
            h = (h + 1) & (MT_S - 1);
        }
        over = over || !placed;
    }
    return over;
}

template <int MT_SHIFT, bool ROWCHK>
#ifndef MTH_MT_OCC
#define MTH_MT_OCC 6
#endif
__global__ __launch_bounds__(MT_B, MTH_MT_OCC) void k_mhl_tile(const MhlTileArgs a) {
    constexpr int W = 1 << MT_SHIFT;
    __shared__ uint32_t tkey[MT_S], taux[MT_S];
    __shared__ uint32_t thist[MT_S * MT_HW];
#ifdef MTH_MT_KEEPF
    __shared__ uint32_t F[W / 32];
#else
    __shared__ uint32_t F[ROWCHK ? 1 : W / 32];
#endif
    // the contributor queue (phases 1 and 2) and the sort arrays of the row phase share their LDS
    __shared__ uint32_t q_or_sort[MT_Q > 2 * MT_B ? MT_Q : 2 * MT_B];
    __shared__ uint32_t bcnt[MT_B];
    uint32_t *const rq = q_or_sort, *const bbase = q_or_sort;
    int32_t *const skey = reinterpret_cast<int32_t *>(q_or_sort + MT_B);
    __shared__ uint32_t ws[MT_B / 64 + 1];
    __shared__ uint32_t s_over, s_qn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (a.ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= a.ntiles) return;
    const int32_t T0 = a.region_beg + (int32_t)(t * W);
    const int32_t T1 = (int32_t)min((int64_t)T0 + W, (int64_t)a.region_end);
    MhlRec *__restrict__ out = a.scratch + (size_t)t * W;
    uint32_t rows_out = 0;
    int sub_shift = a.force_sub ? 8 : MT_SHIFT;            // log2 of the sub-range width (block-uniform)
#ifdef MTH_MT_TRACE
    unsigned long long tk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    tk[8] = __builtin_readcyclecounter();
#define MT_TK(k) do { tk[k] = __builtin_readcyclecounter(); } while (0)
#else
#define MT_TK(k) do {} while (0)
#endif
    uint32_t bad = 0;
    bool heavy_redo = false;                               // block-uniform
    for (int64_t P0l = T0; P0l < T1;) {
        const int32_t P0 = (int32_t)P0l;
        const int32_t P1 = (int32_t)min(P0l + (1ll << sub_shift), (int64_t)T1);
        const uint32_t Wp = (uint32_t)(P1 - P0);
        // candidate reads: start in [P0 - max_span + 1, P1]  (a call sits in [start - 1, start - 1 + max_span])
        const uint32_t lo = min(a.idx[((uint32_t)P0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[(((uint32_t)P1 - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        // more candidate reads than the 16-bit bins can count: halve the stretch; at 256 positions (thousands-fold depth) the sites
        // are only counted and all handed on (`heavy`); the same when the contributor queue overflows at 256 positions
        if (hi - lo > MT_HEAVY && sub_shift > 8) { --sub_shift; continue; }
        MT_TK(0);
        const bool heavy = hi - lo > MT_HEAVY || heavy_redo;
        const uint32_t o_lo = lo < hi ? a.cpg_off[lo] : 0u;      // queue entries carry call offsets relative to the sub-range's first
        static_assert(MT_S == MT_B, "one slot per thread");
        tkey[tid] = MT_EMPTY; taux[tid] = 0u;
        for (int i = tid; i < MT_S * MT_HW; i += MT_B) thist[i] = 0u;
        if constexpr (!ROWCHK) for (int i = tid; i < W / 32; i += MT_B) F[i] = 0u;
        bcnt[tid] = 0u;
        if (tid == 0) { s_over = 0u; s_qn = 0u; }
        __syncthreads();
        MT_TK(1);
        if (lo < hi) {
            // Phase 1, every candidate read (offsets one round ahead, then start, mapq and the FIRST call in one wait): the flusher's
            // mark, and the read's index into the queue if it contributes.  Contributors are a minority (reads with >= min_cpgs CpGs:
            // a third of config 2's reads, a twentieth at WGBS density) while the contribution code below is the expensive part of
            // the kernel: run over all candidates it executed with 5-30 % of the lanes live (0.17 of 0.25 ms on config 2).
            uint32_t o0s[MT_U], o1s[MT_U];
#pragma unroll
            for (int u = 0; u < MT_U; ++u) {
                const uint32_t ii = min(lo + (uint32_t)u * MT_B + tid, hi - 1);
                o0s[u] = a.cpg_off[ii]; o1s[u] = a.cpg_off[ii + 1];
            }
            for (uint32_t b0 = lo; b0 < hi; b0 += MT_B * MT_U) {
                int32_t st[MT_U];
                uint32_t mq[MT_U], o0n[MT_U], o1n[MT_U], fw[MT_U];
#pragma unroll
                for (int u = 0; u < MT_U; ++u) {
                    const uint32_t i = b0 + (uint32_t)u * MT_B + tid, ii = min(i, hi - 1);
                    const bool has = i < hi && o1s[u] != o0s[u];                 // a read without a CpG neither flushes nor contributes (mhl.rs:162)
                    // (ROWCHK: no flusher marks here -- neither the start nor the dependent first-call load is needed per read)
                    if constexpr (ROWCHK) { fw[u] = 0u; st[u] = 0; (void)has; }
                    else { fw[u] = has ? a.cpg_pos[o0s[u]] : 0u; st[u] = a.read_start[ii]; }
                    mq[u] = a.read_mapq[ii];
                    const uint32_t in = min(i + (uint32_t)MT_U * MT_B, hi - 1);
                    o0n[u] = a.cpg_off[in]; o1n[u] = a.cpg_off[in + 1];
                }
#pragma unroll
                for (int u = 0; u < MT_U; ++u) {
                    const uint32_t i = b0 + (uint32_t)u * MT_B + tid;
                    const uint32_t n = i < hi ? o1s[u] - o0s[u] : 0u;
                    if (!ROWCHK && n && !(a.dbg & 1)) {
                        // candidate ranges rely on every call lying in [start - 1, start - 1 + max_span] (rule of the PDR tile kernel):
                        // the first call here, a contributor's other calls in phase 2
                        const uint32_t first = fw[u] & 0x7fffffffu;
                        bad |= (first - ((uint32_t)st[u] - 1u) > (uint32_t)a.max_span) ? 1u : 0u;
                        // a flusher: the stretch [start, first CpG) of the sub-range is marked
                        const int64_t f0 = max((int64_t)st[u], (int64_t)P0) - P0, f1 = min((int64_t)first, (int64_t)P1) - P0;
                        if (f0 < f1) {
                            const uint32_t b_lo = (uint32_t)f0, b_hi = (uint32_t)f1 - 1u;       // inclusive bit range
                            for (uint32_t w = b_lo >> 5; w <= (b_hi >> 5); ++w) {
                                uint32_t m = 0xffffffffu;
                                if (w == (b_lo >> 5)) m &= 0xffffffffu << (b_lo & 31u);
                                if (w == (b_hi >> 5)) m &= 0xffffffffu >> (31u - (b_hi & 31u));
                                atomicOr(&F[w], m);
                            }
                        }
                    }
                    const bool contrib = n != 0 && mq[u] >= a.min_qual && n >= a.min_cpgs && !(a.dbg & 2);      // mhl.rs:176, 181
                    if (heavy) {                                                   // (rare: slots made and counted straight from memory)
                        const uint32_t sm1h = ROWCHK ? (uint32_t)a.read_start[min(i, hi - 1)] - 1u : (uint32_t)st[u] - 1u;
                        if (contrib && mhl_count_only(a.cpg_pos, o0s[u], o1s[u], sm1h, (uint32_t)a.max_span, (uint32_t)P0, Wp, tkey, thist, bad))
                            s_over = 1u;
                        continue;
                    }
                    const unsigned long long bal = __ballot(contrib);
                    if (bal) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_qn, (uint32_t)__builtin_popcountll(bal));
                        base = __builtin_amdgcn_readfirstlane(base);
                        const uint32_t at = base + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                        // entry: read (13 bits: <= 8191 candidates) and call offset (19 bits; more -- long reads -- goes the heavy way)
                        if (contrib) {
                            const uint32_t dr = i - lo, doff = o0s[u] - o_lo;
                            if (at < (uint32_t)MT_Q && doff < (1u << 19)) rq[at] = dr | (doff << 13);
                            else s_over = 1u;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < MT_U; ++u) { o0s[u] = o0n[u]; o1s[u] = o1n[u]; }
            }
        }
        MT_TK(2);
        __syncthreads();
        MT_TK(3);
        // Phase 2, the queued contributors with every lane live: the read's calls (two 16-byte loads; its first MT_NC), its runs,
        // then per call the slot (found or made) and the histogram increments.
        const uint32_t qn = heavy ? 0u : min(s_qn, (uint32_t)MT_Q);
        if (!s_over) {
            // `runs`: the read's maximal methylated runs, length - 1 in four bits each, n_runs of them (readutil.rs:147-164)
            auto contribute = [&](const uint32_t word, const uint32_t n, const uint32_t runs, const uint32_t n_runs, const uint32_t max_runs,
                                  const uint32_t sm1, const uint32_t rel_idx) {
                const uint32_t p = word & 0x7fffffffu, d = p - (uint32_t)P0;
                if (d >= Wp) return;
                // CpG sites lie at least two positions apart: (d >> 1) spreads a dense stretch over consecutive slots (a multiplicative
                // hash costs a quarter-rate multiply per call)
                uint32_t h = (d >> 1) & (MT_S - 1), probes = 0;
                bool placed = false;
                while (probes++ < (uint32_t)MT_S) {
                    const uint32_t cur = atomicCAS(&tkey[h], MT_EMPTY, p);
                    if (cur == MT_EMPTY || cur == p) { placed = true; break; }
                    h = (h + 1) & (MT_S - 1);
                }
                if (!placed) { s_over = 1u; return; }
                uint32_t *hist = &thist[h * MT_HW];
                if (n > (uint32_t)MT_LCAP) { atomicOr(&taux[h], 0x80000000u); atomicAdd(&hist[0], 1u); return; }   // (counted for the min_depth test)
                if (p == sm1) atomicMax(&taux[h], rel_idx);
                atomicAdd(&hist[(n - 1u) >> 1], ((n - 1u) & 1u) ? 0x10000u : 1u);
                for (uint32_t r = 0; r < max_runs; ++r) {                          // max_runs: the wave's largest n_runs
                    const uint32_t m1 = (runs >> (4u * r)) & 15u;
                    if (r < n_runs) atomicAdd(&hist[8u + (m1 >> 1)], (m1 & 1u) ? 0x10000u : 1u);
                }
            };
            // (NEGATIVE, twice: giving the lanes of an instruction entries a hundred bp or more apart -- 64 stretches of the queue, one per
            // lane -- so that their LDS atomics do not meet on one address changed nothing, here (0.211 against 0.212 ms) or in the
            // one-phase form.  The cycle trace (-DMTH_MT_TRACE) puts a config-2 tile at 37 % phase 1, 32 % phase 2, 19 % rows, the rest
            // barriers: every phase is a couple of dependent round trips with four waves to hide them.)
            // (NEGATIVE, round 5, config-3 density, same box, three runs each -- profiles/r05_mhl_tile.md: a read's calls spread over 8 lanes
            // when the queue is short (one wave walked eight compare-and-swap chains while three waited): phase 2 9.7 k -> 8.4 k of a
            // tile's 32.6 k cycles, the kernel 0.168 -> 0.178 ms (every lane of a read re-derives its runs); the flusher bitmap in 64-bit
            // words (half the atomics per stretch): + 0.005 ms; four reads per thread and trip: + 0.003 ms.  With 6 / 4 / 3 workgroups per
            // CU the kernel takes 0.187 / 0.208 / 0.258 ms: at six it is bound by what it issues, not by the round trips.)
            for (uint32_t j0 = 0; j0 < qn; j0 += MT_B) {
                const uint32_t j = j0 + tid;
                const bool act = j < qn;
                const uint32_t e = act ? rq[j] : 0u;
                const uint32_t i = lo + (e & 8191u);
                const uint32_t o0 = o_lo + (e >> 13), o1 = a.cpg_off[i + 1];      // (the calls are requested at once)
                const uint32_t n = act ? o1 - o0 : 0u, nl = min(n, (uint32_t)MT_NC);
                uint32_t vv[MT_NC];
                static_assert(MT_NC == 8, "two 16-byte loads per read");
                if (__all(!act || (unsigned long long)o0 + MT_NC <= (unsigned long long)a.n_cpgs)) {
                    if (act) {
                        const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0), y = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0 + 4);
                        vv[0] = x.x; vv[1] = x.y; vv[2] = x.z; vv[3] = x.w; vv[4] = y.x; vv[5] = y.y; vv[6] = y.z; vv[7] = y.w;
                    }
                } else if (act) {
#pragma unroll
                    for (int k = 0; k < MT_NC; ++k) vv[k] = a.cpg_pos[o0 + min((uint32_t)k, n - 1)];
                }
                const uint32_t sm1 = (uint32_t)a.read_start[i] - 1u;
                uint32_t xmax = 0;
#pragma unroll
                for (int k = 0; k < MT_NC; ++k) xmax = max(xmax, (uint32_t)k < nl ? (vv[k] & 0x7fffffffu) - sm1 : 0u);
                bad |= (xmax > (uint32_t)a.max_span) ? 1u : 0u;
                uint32_t runs = 0, n_runs = 0;
                const bool longer = n > (uint32_t)MT_NC;
                if (act && n <= (uint32_t)MT_LCAP) {
                    // methylation states of the read's calls, then its maximal runs
                    uint32_t mb = 0;
#pragma unroll
                    for (int k = 0; k < MT_NC; ++k) mb |= ((uint32_t)k < nl ? vv[k] >> 31 : 0u) << k;
                    if (longer)
                        for (uint32_t k = MT_NC; k < n; ++k) mb |= (a.cpg_pos[o0 + k] >> 31) << k;
                    uint32_t x = mb;
                    while (x) {                                                     // at most 8 runs in 16 calls
                        x >>= __builtin_ctz(x);
                        const uint32_t m = (uint32_t)__builtin_ctz(~x);           // 1..16 (x < 2^16)
                        x >>= m;
                        runs |= (m - 1u) << (4u * n_runs);
                        ++n_runs;
                    }
                }
                uint32_t max_runs = n_runs;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) max_runs = max(max_runs, (uint32_t)__shfl_xor((int)max_runs, o, 64));
                max_runs = __builtin_amdgcn_readfirstlane(max_runs);
                const uint32_t rel_idx = i - lo + 1u;
                // (NEGATIVE, round 6, rows-checked form, same box twice: the first probes of all eight calls issued together before the
                // increments -- one LDS round trip instead of eight -- 0.093 / 0.093 ms against 0.093 / 0.094; three or four reads per
                // thread and trip in phase 1 (-DMTH_MT_U): 0.090-0.096.  profiles/r06_mhl_rowcheck.md)
#pragma unroll
                for (int k = 0; k < MT_NC; ++k) {
                    if (!__any((uint32_t)k < nl)) break;                            // wave-uniform
                    if ((uint32_t)k < nl) contribute(vv[k], n, runs, n_runs, max_runs, sm1, rel_idx);
                }
                if (__any(longer) && longer)
                    for (uint32_t k = o0 + MT_NC; k < o1; ++k) {
                        const uint32_t w = a.cpg_pos[k];
                        bad |= ((w & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
                        contribute(w, n, runs, n_runs, max_runs, sm1, rel_idx);
                    }
            }
        }
        MT_TK(4);
        __syncthreads();
        MT_TK(5);
        const uint32_t over = s_over;
        __syncthreads();                                    // (s_over is cleared at the top of the next trip; the queue is done with)
        if (over && sub_shift > 8) { --sub_shift; continue; }      // too many distinct sites or contributors: the same stretch again in halves
        if (over && !heavy) { heavy_redo = true; continue; }       // 256 positions and still more contributors than the queue holds
        if (over) bad |= 2u;                                // cannot happen: 256 positions, 256 slots
        heavy_redo = false;
        // rows: the slots whose coverage reaches min_depth (no segment holds more reads than that), sorted by position.
        // Bucket sort: thread = slot = bucket; a bucket is Wsub / 256 positions.
        const uint32_t key = tkey[tid];
        uint32_t cov = 0;
        uint32_t hn[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) hn[w] = 0;
        if (key != MT_EMPTY) {
#pragma unroll
            for (int w = 0; w < 8; ++w) hn[w] = thist[tid * MT_HW + w];
            const bool counted = heavy || (taux[tid] >> 31);
            if (counted) cov = 0xffffffffu;                 // decided below
            else {
#pragma unroll
                for (int w = 0; w < 8; ++w) cov += (hn[w] & 0xffffu) + (hn[w] >> 16);
            }
        }
        const uint32_t aux = key != MT_EMPTY ? taux[tid] : 0u;
        bool hand_on = key != MT_EMPTY && (heavy || (aux >> 31) || a.force_hand_on);
        if (key != MT_EMPTY && cov == 0xffffffffu) {
            // heavy: hist word 0 is a 32-bit count of all contributors; "> 16 CpGs": word 0 holds hn[1] | hn[2] << 16 plus one per long read --
            // an upper bound of the coverage is all the row test needs (the walk decides)
            cov = 0;
            if (heavy) cov = hn[0];
            else {
#pragma unroll
                for (int w = 1; w < 8; ++w) cov += (hn[w] & 0xffffu) + (hn[w] >> 16);
                cov += hn[0];                               // >= hn[1] + hn[2] + long reads
            }
        }
        const bool row = key != MT_EMPTY && cov >= a.min_depth;
        const uint32_t bk = row ? (key - (uint32_t)P0) >> (sub_shift - 8) : 0u;
        uint32_t pib = 0;
        if (row) pib = atomicAdd(&bcnt[bk], 1u);
        __syncthreads();
        const uint32_t m_b = bcnt[tid];
        const uint32_t incl = wave_scan_incl(m_b);
        if (lane == 63) ws[wave + 1] = incl;
        __syncthreads();
        if (tid == 0) { ws[0] = 0; for (int w = 1; w <= MT_B / 64; ++w) ws[w] += ws[w - 1]; }
        __syncthreads();
        const uint32_t n_rows = ws[MT_B / 64];
        bbase[tid] = ws[wave] + incl - m_b;
        __syncthreads();
        if (row) skey[bbase[bk] + pib] = (int32_t)key;
        __syncthreads();
        if (row) {
            const uint32_t b0 = bbase[bk], b1 = b0 + bcnt[bk];
            uint32_t r = b0;
            for (uint32_t i = b0; i < b1; ++i) r += skey[i] < (int32_t)key ? 1u : 0u;
            const int32_t c = (int32_t)key;
            // handed on: under a flusher's mark, or a reverse read with start == c + 1 calls c after a read of that start whose
            // first CpG lies beyond c
            if (!ROWCHK && !hand_on) hand_on = (F[(key - (uint32_t)P0) >> 5] >> ((key - (uint32_t)P0) & 31u)) & 1u;
            const uint32_t self = aux & 0x7fffffffu;
            if (!hand_on && self && !(a.dbg & 8)) {
                const uint32_t K = lo + self - 1u;          // the last contributor that calls its own start - 1 here
                uint32_t r2 = min(a.idx[((uint32_t)c + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
                for (; r2 < K; ++r2) {
                    const int32_t s = a.read_start[r2];
                    if (s > c + 1) break;
                    if (s != c + 1) continue;
                    const uint32_t q0 = a.cpg_off[r2], q1 = a.cpg_off[r2 + 1];
                    if (q1 != q0 && (int32_t)(a.cpg_pos[q0] & 0x7fffffffu) > c) { hand_on = true; break; }
                }
            }
            MhlRec rec;
            rec.pos = c; rec.cov = cov; rec.val = 0.0f; rec.flags = 4u;
            if (!hand_on && !(a.dbg & 4)) {
                // compute_mhl (mhl.rs:43-73) from the histograms: suffix sums twice give S[l], D[l]; same operations and order as
                // mhl_walk_site's finalize()
                uint32_t hm[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) hm[w] = thist[tid * MT_HW + 8 + w];
                uint32_t S[MT_LCAP], D[MT_LCAP];
                uint32_t maxn = 0;
#pragma unroll
                for (int l = 0; l < MT_LCAP; ++l) {
                    S[l] = (l & 1) ? hm[l >> 1] >> 16 : hm[l >> 1] & 0xffffu;
                    D[l] = (l & 1) ? hn[l >> 1] >> 16 : hn[l >> 1] & 0xffffu;
                    if (D[l]) maxn = (uint32_t)l + 1u;
                }
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                    for (int l = MT_LCAP - 2; l >= 0; --l) { S[l] += S[l + 1]; D[l] += D[l + 1]; }
                float l_sum = 0.0f;
                for (uint32_t l = 1; l < maxn + 1; ++l) l_sum = l_sum + (float)l;
                float mhl = 0.0f;
#pragma unroll
                for (int l = 1; l <= MT_LCAP; ++l)
                    if (S[l - 1] > 0) { const float tq = ((float)l * (float)S[l - 1]) / (float)D[l - 1]; mhl = mhl + tq; }
                rec.val = mhl / l_sum;
                rec.flags = 1u;
            }
            out[rows_out + r] = rec;
        }
        rows_out += n_rows;
        P0l = P1;
        MT_TK(6);
        __syncthreads();                                    // the table is cleared by the next trip
        MT_TK(7);
#ifdef MTH_MT_TRACE
        // (plain stores per tile: atomics on eight shared addresses serialise at ~100 ns each and quadrupled the traced kernel's time)
        if (tid == 0 && a.trace && P0 == T0) { for (int k = 1; k < 8; ++k) a.trace[(size_t)t * 8 + k] = tk[k] - tk[k - 1]; a.trace[(size_t)t * 8] = 1ull | ((tk[0] - tk[8]) << 8); }
#endif
    }
    if (bad & 1u) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
    if (bad & 2u) atomicOr(&a.st->err, (uint32_t)ERRB_CAPACITY);
    if (tid == 0) {
        a.tile_cnt[t] = rows_out;
        if (rows_out) atomicAdd(a.bucket + (t >> TILE_BUCKET_SHIFT), (unsigned long long)rows_out);
    }
}

// One wave per tile: the tile's first row = rows of the buckets before its bucket + rows of the bucket's earlier tiles; its rows
// go to the candidate-site arrays.  The wave of the last tile leaves the total in sites_st->n_sites.
constexpr int MG_WAVES = 4;
__global__ __launch_bounds__(64 * MG_WAVES) void k_mhl_tile_gather(const MhlRec *__restrict__ scratch, const uint32_t *__restrict__ tile_cnt,
                                                                   const unsigned long long *__restrict__ bucket, const uint32_t ntiles,
                                                                   const uint32_t rows_per_tile, DevState *__restrict__ sites_st,
                                                                   int32_t *__restrict__ site_pos, float *__restrict__ val,
                                                                   uint32_t *__restrict__ cov, uint32_t *__restrict__ flags,
                                                                   uint32_t *__restrict__ hand_list) {
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * MG_WAVES + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const uint32_t bk = t >> TILE_BUCKET_SHIFT, t_first = bk << TILE_BUCKET_SHIFT;
    unsigned long long before = 0;
    for (uint32_t b = lane; b < bk; b += 64) before += bucket[b];
    uint32_t in_bucket = 0;
    for (uint32_t q = t_first + lane; q < t; q += 64) in_bucket += tile_cnt[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o, 64);
        in_bucket += __shfl_xor(in_bucket, o, 64);
    }
    const unsigned long long base = before + in_bucket;
    const uint32_t n = tile_cnt[t];
    const MhlRec *__restrict__ src = scratch + (size_t)t * rows_per_tile;
    for (uint32_t i = lane; i < n; i += 64) {
        const MhlRec r = src[i];
        site_pos[base + i] = r.pos; val[base + i] = r.val; cov[base + i] = r.cov; flags[base + i] = r.flags;
        // the handed-on sites, listed for k_mhl_walk_wave (any order; the count lives in the sink state's first spare counter)
        if (r.flags == 4u) hand_list[atomicAdd(reinterpret_cast<unsigned long long *>(&sites_st->lpmd[0]), 1ull)] = (uint32_t)(base + i);
    }
    if (t == ntiles - 1 && lane == 0) sites_st->n_sites = base + n;
}


// ---- the flush rule for the ROWS only (k_mhl_tile<.., ROWCHK = true>) ------------------------------------------------------
// With sparse calls (WGBS: 1.35 CpGs a read, one read in twenty contributes, ~3 rows per 16 384-bp tile) the tile kernel's
// per-read flusher marks -- the read's start, a DEPENDENT load of its first call, the bitmap atomics -- were 29 % of the kernel for
// a test that only the rows need.  Here 16 lanes take one finished row c and look at the reads that can be its flushers: >= 1
// CpG, start <= c < first CpG (mhl.rs:162-173), so start in [c - max_span + 2, c].  The same criterion as the bitmap's, read by
// read; a row that has one is handed on (flag 4, listed for k_mhl_walk_wave) exactly as before.
struct MhlRowChkArgs {
    const int32_t  *read_start;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    DevState *sites_st;               // n_sites; lpmd[0] = entries of hand_list
    const int32_t *site_pos;
    uint32_t *flags, *hand_list;
    DevState *st;
    int32_t idx_base, max_span;
    uint32_t n_reads;
};

__global__ __launch_bounds__(256) void k_mhl_rowcheck(const MhlRowChkArgs a) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const uint32_t wv = (blockIdx.x * 256u + threadIdx.x) >> 6, nwv = (gridDim.x * 256u) >> 6;
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    uint32_t bad = 0;
    for (uint32_t j0 = wv * 4u; j0 < n_sites; j0 += nwv * 4u) {               // wave-uniform
        const uint32_t j = j0 + ((uint32_t)lane >> 4);
        const bool live = j < n_sites && a.flags[j] == 1u;
        const int32_t c = live ? a.site_pos[j] : 0;
        uint32_t rlo = 0, rhi = 0;
        if (live) {
            rlo = min(a.idx[((uint32_t)c - (uint32_t)a.max_span + 2u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
            rhi = min(a.idx[(((uint32_t)c - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        }
        bool fl = false;
        for (uint32_t i = rlo + (uint32_t)sub; __any(i < rhi); i += 16u) {
            if (i < rhi) {
                const int32_t s = a.read_start[i];
                const uint32_t q0 = a.cpg_off[i], q1 = a.cpg_off[i + 1];
                if (q1 != q0 && s <= c) {
                    const uint32_t first = a.cpg_pos[q0] & 0x7fffffffu;
                    bad |= (first - ((uint32_t)s - 1u) > (uint32_t)a.max_span) ? 1u : 0u;
                    fl = fl || (int32_t)first > c;
                }
            }
        }
        const unsigned long long b = __ballot(fl);
        if (live && sub == 0 && ((b >> (lane & 48)) & 0xffffull)) {
            a.flags[j] = 4u;
            a.hand_list[atomicAdd(reinterpret_cast<unsigned long long *>(&a.sites_st->lpmd[0]), 1ull)] = j;
        }
    }
    if (bad) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
}


// ---- the handed-on sites: one WAVE per site --------------------------------------------------------------------------------
// k_mhl_walk_big gives a site to one lane, which walks its ~40 candidate reads through a chain of dependent loads: ~75 us for the
// longest chain however few sites there are (config 2: 1 750 of 726 028 rows).  Here lane = candidate read: the reads' fields and calls
// arrive in two round trips for the whole site, every lane decides alone what its read is -- a flusher (>= 1 CpG, the first one
// beyond c: mhl.rs:162-173), a contributor (mapq, n_cpgs, calls c: mhl.rs:176-192) or neither -- and the reference's sequential loop
// becomes bit arithmetic on two ballots: the contributors between two flushers are one segment; the LAST segment with >=
// min_depth reads is the site's row (mhl.rs:163-171, 201-205).  Per-lane byte histograms (CpG count, run lengths) of the open
// segment are summed over the wave when a segment with enough reads closes.  Sites with more than 512 candidates, or with a
// covering read of more than 16 CpGs, keep flag 4 and take k_mhl_walk_big as before.
struct MhlWaveArgs {
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    const DevState *sites_st;
    const uint32_t *hand_list;
    const int32_t  *site_pos;
    float *val; uint32_t *cov, *flags;
    int32_t idx_base, max_span;
    uint32_t n_reads, n_cpgs, min_depth, min_cpgs;
    uint8_t min_qual;
};
constexpr uint32_t MWV_MAX_CAND = 512;

__global__ __launch_bounds__(256) void k_mhl_walk_wave(const MhlWaveArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t wv = (blockIdx.x * 256u + threadIdx.x) >> 6, nwv = (gridDim.x * 256u) >> 6;
    const uint32_t n_hand = (uint32_t)a.sites_st->lpmd[0];
    for (uint32_t e = wv; e < n_hand; e += nwv) {
        const uint32_t j = a.hand_list[e];
        const int32_t c = a.site_pos[j];
        const uint32_t lo = min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        if (hi - lo > MWV_MAX_CAND) continue;                                   // wave-uniform: left to k_mhl_walk_big
        // the open segment: per-lane byte histograms (bins 1..16 of the CpG count / of the run lengths), reads in it
        unsigned long long hn_lo = 0, hn_hi = 0, hm_lo = 0, hm_hi = 0;
        uint32_t open_cov = 0;
        bool have = false, defer = false;
        float res = 0.0f;
        uint32_t res_cov = 0;
        auto close_segment = [&]() {                                            // wave-uniform call; compute_mhl (mhl.rs:43-73)
            uint32_t hn[8], hm[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const unsigned long long sn = w < 4 ? hn_lo >> (16 * w) : hn_hi >> (16 * (w - 4));
                const unsigned long long sm = w < 4 ? hm_lo >> (16 * w) : hm_hi >> (16 * (w - 4));
                hn[w] = wave_sum(((uint32_t)sn & 0xffu) | (((uint32_t)sn & 0xff00u) << 8));
                hm[w] = wave_sum(((uint32_t)sm & 0xffu) | (((uint32_t)sm & 0xff00u) << 8));
            }
            uint32_t S[MT_LCAP], D[MT_LCAP], maxn = 0;
#pragma unroll
            for (int l = 0; l < MT_LCAP; ++l) {
                S[l] = (l & 1) ? hm[l >> 1] >> 16 : hm[l >> 1] & 0xffffu;
                D[l] = (l & 1) ? hn[l >> 1] >> 16 : hn[l >> 1] & 0xffffu;
                if (D[l]) maxn = (uint32_t)l + 1u;
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int l = MT_LCAP - 2; l >= 0; --l) { S[l] += S[l + 1]; D[l] += D[l + 1]; }
            float l_sum = 0.0f;
            for (uint32_t l = 1; l < maxn + 1; ++l) l_sum = l_sum + (float)l;
            float mhl = 0.0f;
#pragma unroll
            for (int l = 1; l <= MT_LCAP; ++l)
                if (S[l - 1] > 0) { const float tq = ((float)l * (float)S[l - 1]) / (float)D[l - 1]; mhl = mhl + tq; }
            res = mhl / l_sum; res_cov = open_cov; have = true;
        };
        for (uint32_t b0 = lo; b0 < hi; b0 += 64) {
            const uint32_t i = b0 + (uint32_t)lane;
            const bool in = i < hi;
            const uint32_t ii = in ? i : hi - 1;
            const uint32_t o0 = a.cpg_off[ii], o1 = a.cpg_off[ii + 1], mq = a.read_mapq[ii];
            const uint32_t n = in ? o1 - o0 : 0u, nl = min(n, (uint32_t)MT_NC);
            uint32_t vv[MT_NC];
            if (__all(n == 0 || (unsigned long long)o0 + MT_NC <= (unsigned long long)a.n_cpgs)) {
                if (n) {
                    const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0), y = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0 + 4);
                    vv[0] = x.x; vv[1] = x.y; vv[2] = x.z; vv[3] = x.w; vv[4] = y.x; vv[5] = y.y; vv[6] = y.z; vv[7] = y.w;
                }
            } else if (n) {
#pragma unroll
                for (int k = 0; k < MT_NC; ++k) vv[k] = a.cpg_pos[o0 + min((uint32_t)k, n - 1)];
            }
            const bool flusher = n != 0 && c < (int32_t)(vv[0] & 0x7fffffffu);                  // mhl.rs:163 (strict '<', before the filters)
            bool contrib = false;
            uint32_t mb = 0;
            if (n != 0 && !flusher && mq >= a.min_qual && n >= a.min_cpgs) {                    // mhl.rs:176, 181
#pragma unroll
                for (int k = 0; k < MT_NC; ++k) {
                    const bool live = (uint32_t)k < nl;
                    contrib = contrib || (live && (int32_t)(vv[k] & 0x7fffffffu) == c);
                    mb |= (live ? vv[k] >> 31 : 0u) << k;
                }
                if (n > (uint32_t)MT_NC && n <= (uint32_t)MT_LCAP)
                    for (uint32_t k = MT_NC; k < n; ++k) {
                        const uint32_t w = a.cpg_pos[o0 + k];
                        contrib = contrib || (int32_t)(w & 0x7fffffffu) == c;
                        mb |= (w >> 31) << k;
                    }
                if (n > (uint32_t)MT_LCAP) {                                                    // does it call c at all ?
                    for (uint32_t k = MT_NC; k < n && !contrib; ++k) contrib = (int32_t)(a.cpg_pos[o0 + k] & 0x7fffffffu) == c;
                    if (contrib) defer = true;
                }
            }
            if (__any(defer)) { defer = true; break; }
            // this read's own histogram entries
            unsigned long long my_n_lo = 0, my_n_hi = 0, my_m_lo = 0, my_m_hi = 0;
            if (contrib) {
                if (n <= 8u) my_n_lo = 1ull << (8u * (n - 1u)); else my_n_hi = 1ull << (8u * (n - 9u));
                uint32_t x = mb;
                while (x) {                                                                     // readutil.rs:147-164
                    x >>= __builtin_ctz(x);
                    const uint32_t m = (uint32_t)__builtin_ctz(~x);
                    x >>= m;
                    if (m <= 8u) my_m_lo += 1ull << (8u * (m - 1u)); else my_m_hi += 1ull << (8u * (m - 9u));
                }
            }
            unsigned long long Fm = __ballot(flusher), Cm = __ballot(contrib);
            // segments of this chunk, in lane order: the contributors below each flusher join the open segment, which then closes
            unsigned long long done = 0;                                                        // lanes already accounted for
            while (true) {
                const unsigned long long upto = Fm ? ((Fm & (~Fm + 1ull)) - 1ull) : ~0ull;      // lanes below the next flusher (all, if none)
                const unsigned long long grp = Cm & upto & ~done;
                if ((grp >> lane) & 1ull) { hn_lo += my_n_lo; hn_hi += my_n_hi; hm_lo += my_m_lo; hm_hi += my_m_hi; }
                open_cov += (uint32_t)__builtin_popcountll(grp);
                if (!Fm) break;
                if (open_cov > 0) {                                                             // the flush (mhl.rs:163-171)
                    if (open_cov >= a.min_depth) close_segment();
                    hn_lo = hn_hi = hm_lo = hm_hi = 0; open_cov = 0;
                }
                done = upto | (Fm & (~Fm + 1ull));
                Fm &= Fm - 1ull;
            }
        }
        if (defer) continue;                                                                    // a read with > 16 CpGs covers c: k_mhl_walk_big
        if (open_cov > 0 && open_cov >= a.min_depth) close_segment();                          // mhl.rs:201-205
        if (lane == 0) { a.val[j] = res; a.cov[j] = res_cov; a.flags[j] = have ? 1u : 0u; }
    }
}

// The tile pass of one batch: candidate-site arrays (ctx->s_pos, w_val, w_cov, w_flags; count in d_state2->n_sites) filled with
// the finished rows (flag 1) and the sites left to k_mhl_walk_big (flag 4); the read index is left for that walk.
int launch_mhl_tile(mth_ctx *ctx, const mth_batch_t &d, const mth_mhl_params_t &p, uint64_t &bound) {
    hipStream_t s = ctx->stream;
    const int64_t region_len = (int64_t)d.region_end - d.region_beg;
    bound = (uint64_t)std::min<int64_t>((int64_t)d.n_cpgs, std::max<int64_t>(region_len, 0));
    if (!ctx->d_state2) MTH_HIP(ctx, hipMalloc((void **)&ctx->d_state2, sizeof(DevState)));
    MTH_HIP(ctx, hipMemsetAsync(ctx->d_state2, 0, sizeof(DevState), s));
    MTH_HIP(ctx, ctx->s_pos.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->w_val.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->w_cov.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->w_flags.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->w_aux.reserve((bound + 1) * 4, s));
    if (d.n_reads == 0 || region_len <= 0) return MTH_OK;
    // Tile width: up to 0.65 x 256 slots' worth of sites per tile at the batch's call density (a denser stretch is redone in halves)
    // and few enough candidate reads for the 16-bit bins.  A tile is a chain of ~6 dependent round trips and ~10 barriers whatever it
    // holds, so the widest tile that fits wins: config-3 density 0.319 / 0.220 / 0.144 ms at 4096 / 8192 / 16384 bp, config 2 0.209 /
    // 0.204 at 4096 / 8192 (16384 overflows the slots there: 1.27 ms)
    // (round 6: the one-wave-per-tile form that pays for FDRP was built for MHL too -- parity-green and 30 % slower than this kernel, whose
    // tile is eleven times larger: profiles/r06_mhl_wtile.md, tools/experiments/mth_mhl_wtile.hip)
    int shift = 12, W = 0;
    int32_t idx_base = 0;
    uint32_t ntiles = 0;
    MhlTileArgs a;
    a.trace = nullptr;
    // the flush rule per read (position bitmap in the tile kernel) or per finished row (k_mhl_rowcheck): rows are few where calls are
    // sparse -- at <= 2 CpGs a read one read in twenty reaches min_cpgs = 4 -- and there the per-read marks (start, a dependent first-call
    // load, bitmap atomics) were 29 % of the tile kernel for a test only the rows need (chr1-sized contig at config-3 density: 0.170 ->
    // 0.130 ms + 0.009 for the row kernel); config 2 (2.94 CpGs a read, 100 rows a tile) keeps the marks
    bool rowchk = (double)d.n_cpgs <= 1.8 * (double)d.n_reads;       // (density sweep, profiles/r06_mhl_rowcheck.md: even at 2.0)
    if (const char *e = getenv("MTH_MHL_ROWCHK")) rowchk = atoi(e) != 0;                               // tests / tuning
    {
        const double cpr = (double)d.n_cpgs / (double)d.n_reads;
        const double sites_per_bp = cpr / (double)std::max(d.max_span, 1);
        const double reads_per_bp = (double)d.n_reads / (double)region_len;
        while (shift < 14 && sites_per_bp * (double)(2 << shift) <= 0.65 * MT_S &&
               reads_per_bp * (double)((2 << shift) + d.max_span + 2 * IDX_Q) <= 0.75 * MT_HEAVY)
            ++shift;
        // Without the marks a tile's cost is its contributors', and a slot is taken only by a site that a CONTRIBUTOR calls: 32 768-position
        // tiles where those sites fit (config 3: MHL pass 2.21 -> 1.56 ms; 65 536 overflows the slots on a chr1-sized contig: 0.092 ->
        // 0.227 ms) and the region still gives three rounds of tiles.  The share of such sites: a covering read has min_cpgs - 1 other
        // CpGs with probability pc (Poisson at the batch's calls per read); overlapping readers are far from independent -- sqrt(depth)
        // of them counted (measured at 10 x: 0.35 of the sites).  An underestimate costs time only (the stretch is redone in halves).
        if (rowchk && shift == 14) {
            double pc = 1.0, term = std::exp(-cpr);
            for (uint32_t k = 0; k + 1 < p.min_cpgs && k < 64; ++k) { pc -= term; term *= cpr / (double)(k + 1); }
            pc = std::min(1.0, std::max(pc, 0.0));
            const double depth = reads_per_bp * (double)std::max(d.max_span, 1);
            const double share = 1.0 - std::pow(1.0 - pc, std::max(1.0, std::sqrt(depth)));
            if (sites_per_bp * share * 32768.0 <= 0.5 * MT_S && reads_per_bp * (double)(32768 + d.max_span + 2 * IDX_Q) <= 0.75 * MT_HEAVY &&
                region_len >= 3ll * 1536 * 32768)
                shift = 15;
        }
    }
    if (const char *e = getenv("MTH_MHL_TILE_SHIFT")) shift = std::min(16, std::max(12, atoi(e)));   // tests / tuning
    W = 1 << shift;
    int rc = build_read_index(ctx, d, W, idx_base, ntiles);
    if (rc) return rc;
    const uint32_t nbk = (ntiles + (1u << TILE_BUCKET_SHIFT) - 1) >> TILE_BUCKET_SHIFT;
    MTH_HIP(ctx, ctx->tile_cnt.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_bucket.reserve((size_t)nbk * 5 * sizeof(unsigned long long), s));
    MTH_HIP(ctx, hipMemsetAsync(ctx->tile_bucket.p, 0, (size_t)nbk * sizeof(unsigned long long), s));
    MTH_HIP(ctx, ctx->scratch.reserve((size_t)ntiles * W * sizeof(MhlRec), s));
    a.read_start = d.read_start; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos; a.idx = idx_ptr(ctx);
    a.region_beg = d.region_beg; a.region_end = d.region_end; a.idx_base = idx_base; a.max_span = d.max_span;
    a.n_reads = d.n_reads; a.ntiles = ntiles; a.n_cpgs = (uint32_t)d.n_cpgs; a.min_depth = p.min_depth; a.min_cpgs = p.min_cpgs;
    a.min_qual = p.min_qual;
    a.force_sub = getenv("MTH_MHL_FORCE_SUB") ? 1 : 0; a.force_hand_on = getenv("MTH_MHL_FORCE_HAND_ON") ? 1 : 0;
    a.dbg = getenv("MTH_MHL_DBG") ? (uint8_t)atoi(getenv("MTH_MHL_DBG")) : 0;
    a.scratch = reinterpret_cast<MhlRec *>(ctx->scratch.p); a.tile_cnt = ctx->tile_cnt.as<uint32_t>();
    a.bucket = ctx->tile_bucket.as<unsigned long long>(); a.st = ctx->d_state;
    a.trace = nullptr;
#ifdef MTH_MT_TRACE
    static unsigned long long *d_trace = nullptr;
    static size_t d_trace_n = 0;
    if (d_trace_n < (size_t)ntiles * 8) { if (d_trace) (void)hipFree(d_trace); d_trace_n = (size_t)ntiles * 8; MTH_HIP(ctx, hipMalloc((void **)&d_trace, d_trace_n * 8)); }
    MTH_HIP(ctx, hipMemsetAsync(d_trace, 0, (size_t)ntiles * 64, s));
    a.trace = d_trace;
#endif
    const uint32_t grid = ((ntiles + 7) / 8) * 8;
    {
        LaunchTimer lt(ctx, K_MHLTILE);
#define MTH_MT_LAUNCH(SH) do { if (rowchk) hipLaunchKernelGGL((k_mhl_tile<SH, true>), dim3(grid), dim3(MT_B), 0, s, a); \
                                else hipLaunchKernelGGL((k_mhl_tile<SH, false>), dim3(grid), dim3(MT_B), 0, s, a); } while (0)
        if (shift == 12) MTH_MT_LAUNCH(12);
        else if (shift == 13) MTH_MT_LAUNCH(13);
        else if (shift == 14) MTH_MT_LAUNCH(14);
        else if (shift == 15) MTH_MT_LAUNCH(15);
        else MTH_MT_LAUNCH(16);
#undef MTH_MT_LAUNCH
    }
    {
        LaunchTimer lt(ctx, K_GATHER);
        hipLaunchKernelGGL(k_mhl_tile_gather, dim3((ntiles + MG_WAVES - 1) / MG_WAVES), dim3(64 * MG_WAVES), 0, s,
                           reinterpret_cast<const MhlRec *>(ctx->scratch.p), ctx->tile_cnt.as<uint32_t>(),
                           ctx->tile_bucket.as<unsigned long long>(), ntiles, (uint32_t)W, ctx->d_state2, ctx->s_pos.as<int32_t>(),
                           ctx->w_val.as<float>(), ctx->w_cov.as<uint32_t>(), ctx->w_flags.as<uint32_t>(), ctx->w_aux.as<uint32_t>());
    }
    if (rowchk) {
        LaunchTimer lt(ctx, K_MHLROWCHK);
        MhlRowChkArgs rc;
        rc.read_start = d.read_start; rc.cpg_off = d.cpg_off; rc.cpg_pos = d.cpg_pos; rc.idx = idx_ptr(ctx);
        rc.sites_st = ctx->d_state2; rc.site_pos = ctx->s_pos.as<int32_t>(); rc.flags = ctx->w_flags.as<uint32_t>();
        rc.hand_list = ctx->w_aux.as<uint32_t>(); rc.st = ctx->d_state; rc.idx_base = idx_base; rc.max_span = d.max_span; rc.n_reads = d.n_reads;
        hipLaunchKernelGGL(k_mhl_rowcheck, dim3(2048), dim3(256), 0, s, rc);
    }
    if (!getenv("MTH_MHL_NO_WAVE_WALK")) {
        LaunchTimer lt(ctx, K_MHLWALK);
        MhlWaveArgs w;
        w.read_mapq = d.read_mapq; w.cpg_off = d.cpg_off; w.cpg_pos = d.cpg_pos; w.idx = idx_ptr(ctx);
        w.sites_st = ctx->d_state2; w.hand_list = ctx->w_aux.as<uint32_t>(); w.site_pos = ctx->s_pos.as<int32_t>();
        w.val = ctx->w_val.as<float>(); w.cov = ctx->w_cov.as<uint32_t>(); w.flags = ctx->w_flags.as<uint32_t>();
        w.idx_base = idx_base; w.max_span = d.max_span; w.n_reads = d.n_reads; w.n_cpgs = (uint32_t)d.n_cpgs;
        w.min_depth = p.min_depth; w.min_cpgs = p.min_cpgs; w.min_qual = p.min_qual;
        hipLaunchKernelGGL(k_mhl_walk_wave, dim3(1024), dim3(256), 0, s, w);
    }
#ifdef MTH_MT_TRACE
    {
        std::vector<unsigned long long> hv((size_t)ntiles * 8);
        MTH_HIP(ctx, hipStreamSynchronize(s));
        MTH_HIP(ctx, hipMemcpy(hv.data(), a.trace, hv.size() * 8, hipMemcpyDeviceToHost));
        double h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, idxw = 0;
        for (size_t q = 0; q < ntiles; ++q) { for (int k = 1; k < 8; ++k) h[k] += (double)hv[q * 8 + k]; h[0] += (double)(hv[q * 8] & 1ull); idxw += (double)(hv[q * 8] >> 8); }
        fprintf(stderr, "[mhl tile trace] tiles %.0f; cycles of a tile's first sub-range: index look-up %.0f  clear %.0f  phase1 %.0f  barrier %.0f  phase2 %.0f  barrier %.0f  rows %.0f  barrier %.0f\n", h[0],
                idxw / h[0], h[1] / h[0], h[2] / h[0], h[3] / h[0], h[4] / h[0], h[5] / h[0], h[6] / h[0], h[7] / h[0]);
    }
#endif
    MTH_HIP(ctx, hipGetLastError());
    if (getenv("MTH_MHL_DEBUG")) {       // how many sites the tile pass finished / handed on (tuning aid; synchronises)
        DevState st;
        MTH_HIP(ctx, hipStreamSynchronize(s));
        MTH_HIP(ctx, hipMemcpy(&st, ctx->d_state2, sizeof st, hipMemcpyDeviceToHost));
        std::vector<uint32_t> f((size_t)st.n_sites);
        if (!f.empty()) MTH_HIP(ctx, hipMemcpy(f.data(), ctx->w_flags.p, f.size() * 4, hipMemcpyDeviceToHost));
        size_t n4 = 0;
        for (uint32_t x : f) n4 += x == 4u;
        fprintf(stderr, "[mhl tile] W=%d tiles=%u rows=%llu handed_on=%zu\n", W, ntiles, (unsigned long long)st.n_sites, n4);
    }
    return MTH_OK;
}

}  // namespace mth
