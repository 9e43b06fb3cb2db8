// mth_pdr_wide.hip -- the fused PDR + LPMD pass (pdr.rs:119-212, lpmd.rs:154-202, readutil.rs:166-224) for SPARSE batches:
// hashed sites and wide tiles (gfx950).
//
// k_pdr_lpmd_tile (mth_pdr_lpmd.hip) keeps one LDS counter per reference position, which fixes its tile at 4096 bp (16 KiB, 8
// workgroups per CU).  At WGBS depth such a tile holds ~280 candidate reads -- one loop trip per thread -- and the kernel's time is
// the tile's fixed chain (index look-up -> offsets -> calls -> LDS atomics -> barrier -> compaction of 4096 positions -> row count)
// times 60 781 tiles over 2 048 resident workgroups: 0.170 ms for 16 M reads, 0.29 of the HBM roofline (profiles/r03_tile_pmc_wgbs.md;
// removing instructions from the chain changed nothing, 8192-bp tiles halve the residency).  The quartet, pairs and MHL tile kernels
// showed the way out this round: their LDS tables do not grow with the tile, so they run 16384-32768-bp tiles and pay the chain 4-8
// times less often (k_quartet_tile 0.160 -> 0.113 ms at this depth from the tile width alone).  This kernel does the same for PDR +
// LPMD: the sites a tile's PDR-passing reads call live in a 1024-slot LDS hash table (position, coverage, discordant reads: 12 KiB
// whatever the tile width), rows are the slots with coverage >= min_depth, bucket-sorted by position.
//
// Per tile (16384, 32768 or 65536 bp, one 256-thread workgroup), in stretches of 3072 candidate reads:
//   phase 1   every candidate (offsets one round ahead; start, mapq): LPMD's read totals (lpmd.rs:176-179) for the reads the tile
//             owns, and the reads that have work -- LPMD pairs (>= 2 calls, mapq) or PDR (>= min_cpgs calls, mapq; pdr.rs:147-157)
//             -- into a queue of 16-bit read numbers.  At WGBS depth that is 40 % of the reads.
//   phase 2   the queue with every lane live: the read's first 8 calls as two 16-byte loads + its relative positions, the
//             concordance state, the windowed pair counts two pairs per instruction (the tile kernel's packed form) and, for a
//             PDR-passing read, one compare-and-swap + one or two LDS adds per call.
//   rows      slots with coverage >= min_depth, bucket sort by position, straight into the tile's scratch slice.
// A stretch with more distinct sites than slots is redone in halves (its LPMD sums are only committed when the stretch is done).
// Same outputs as k_pdr_lpmd_tile (scratch slices, tile_cnt, bucket sums): k_gather and every caller (PDR result columns, site
// discovery for FDRP) are unchanged.  launch_pdr_lpmd picks the form per batch from its call density.
#include <type_traits>

#include "mth_ctx.h"
#include "mth_tile_dev.h"

namespace mth {

typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));

// PW_QCAP / MTH_PW_OCC (tools/ab_measure.sh-style A/B on config-3 density, 32768- / 65536-bp tiles, ms): 2048 / 8 waves 0.121 / 0.115,
// 3072 / 8 (3 spilled VGPRs) 0.121 / 0.115, 3072 / 7 (72 VGPRs, no spill) 0.114 / 0.111, 4096 / 6 0.117 / 0.115, 6144 / 6 0.117 / 0.110
#ifndef MTH_PW_QCAP
#define MTH_PW_QCAP 3072
#endif
constexpr int PW_S = 1024, PW_B = 256, PW_U = 2, PW_NB = 8, PW_QCAP = MTH_PW_QCAP;
constexpr uint32_t PW_EMPTY = 0xffffffffu;

template <int SHIFT, typename RelT>
#ifndef MTH_PW_OCC
#define MTH_PW_OCC 7
#endif
__global__ __launch_bounds__(PW_B, MTH_PW_OCC) void k_pdr_lpmd_wide(const TileArgs a, const uint32_t ntiles) {
    constexpr int W = 1 << SHIFT;
    constexpr bool PACKED = sizeof(RelT) == 1;                 // 8-bit relpos: the packed pair form
    __shared__ uint32_t tkey[PW_S], tcov[PW_S], tdisc[PW_S];   // the site table; in the row phase: keys / counters in bucket order
    __shared__ uint32_t bcnt[PW_B];
    // the work queue of the read phases shares its LDS with the row phase's bucket bases
    __shared__ uint32_t q_or_sort[PW_QCAP / 2];
    static_assert(PW_QCAP / 2 >= PW_B, "bbase fits under the queue");
    uint16_t *const rq = reinterpret_cast<uint16_t *>(q_or_sort);
    uint32_t *const bbase = q_or_sort;
    __shared__ uint32_t red[4][PW_B / 64], ws[PW_B / 64 + 1];
    __shared__ __attribute__((aligned(16))) SlotTabs tabs;
    __shared__ uint32_t s_over, s_qn, s_qh;          // s_qn / s_qh: queued reads of <= 4 calls (front of the queue) / of more (back)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    // (a tile may be narrower than its scratch slice: the host picks the width that fills whole rounds of resident workgroups)
    const uint32_t Wt = a.tile_w_rt ? a.tile_w_rt : (uint32_t)W;
    const int32_t T0 = a.region_beg + (int32_t)(t * Wt);
    const int32_t T1 = (int32_t)min((int64_t)T0 + Wt, (int64_t)a.region_end);
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    SiteRec *__restrict__ out = a.scratch + (size_t)t * W;
    slot_tabs_init(tabs, tid);
    // (distances between live calls are < 2^16, so capping max_distance keeps dead-slot differences outside)
    const int32_t maxd = PACKED ? min(a.max_dist, 255) : min(a.max_dist, 1 << 20);   // 8-bit relpos: no distance beyond 255
    const int32_t mind = max(a.min_dist, 0);
    const bool lp_possible = a.want_lpmd && maxd >= a.min_dist && maxd >= 0;          // min > max: no pair can qualify
    uint32_t lp_c = 0, lp_d = 0, n_read = 0, n_valid = 0;      // the thread's LPMD sums over the finished stretches
    uint32_t rows_out = 0, bad = 0;
    int sub_shift = SHIFT;                                     // log2 of the stretch of positions worked on (block-uniform)
    for (int64_t P0l = T0; P0l < T1;) {
        const int32_t P0 = (int32_t)P0l;
        const int32_t P1 = (int32_t)min(P0l + (1ll << sub_shift), (int64_t)T1);
        const uint32_t Wp = (uint32_t)(P1 - P0);
        // candidate reads: start in [P0 - max_span + 1, P1]  (a call sits in [start - 1, start - 1 + max_span])
        const uint32_t lo = min(a.idx[((uint32_t)P0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[(((uint32_t)P1 - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        for (int i = tid; i < PW_S; i += PW_B) { tkey[i] = PW_EMPTY; tcov[i] = 0u; tdisc[i] = 0u; }
        bcnt[tid] = 0u;
        if (tid == 0) { s_over = 0u; s_qn = 0u; s_qh = 0u; }
        __syncthreads();
        uint32_t a_c = 0, a_d = 0, a_r = 0, a_v = 0;           // this attempt's LPMD sums
        for (uint32_t c0 = lo; c0 < hi; c0 += PW_QCAP) {
            const uint32_t c1 = min(c0 + (uint32_t)PW_QCAP, hi);
            if (c0 != lo) {
                __syncthreads();                               // the previous stretch's queue is done with
                if (tid == 0) { s_qn = 0u; s_qh = 0u; }
                __syncthreads();
            }
            // ---- phase 1
            uint32_t o0s[PW_U], o1s[PW_U];
#pragma unroll
            for (int u = 0; u < PW_U; ++u) {
                const uint32_t ii = min(c0 + (uint32_t)u * PW_B + tid, c1 - 1);
                o0s[u] = a.cpg_off[ii]; o1s[u] = a.cpg_off[ii + 1];
            }
            for (uint32_t b0 = c0; b0 < c1; b0 += PW_B * PW_U) {
                int32_t st[PW_U];
                uint32_t mq[PW_U], o0n[PW_U], o1n[PW_U];
#pragma unroll
                for (int u = 0; u < PW_U; ++u) {
                    const uint32_t i = b0 + (uint32_t)u * PW_B + tid, ii = min(i, c1 - 1);
                    st[u] = a.read_start[ii]; mq[u] = a.read_mapq[ii];
                    const uint32_t in = min(i + (uint32_t)PW_U * PW_B, c1 - 1);
                    o0n[u] = a.cpg_off[in]; o1n[u] = a.cpg_off[in + 1];
                }
#pragma unroll
                for (int u = 0; u < PW_U; ++u) {
                    const uint32_t i = b0 + (uint32_t)u * PW_B + tid;
                    const bool in = i < c1;
                    const uint32_t n = in ? o1s[u] - o0s[u] : 0u;
                    const bool owned = in && st[u] >= P0 && st[u] < P1;
                    // lpmd.rs:176-179
                    const bool lp_ok = a.want_lpmd && owned && mq[u] >= a.lpmd_min_qual;
                    if (a.want_lpmd && owned) { a_r += 1; a_v += lp_ok ? 1u : 0u; }
                    // pdr.rs:147-157
                    const bool pdr_ok = a.want_pdr && n >= a.min_cpgs && mq[u] >= a.pdr_min_qual && n > 0;
                    const bool work = (lp_ok && lp_possible && n > 1) || pdr_ok;
                    // the queue is filled from both ends: reads of <= 4 calls from the front, the others from the back (a stretch holds at
                    // most PW_QCAP candidates, so the two never meet) -- phase 2 then runs whole waves of either kind
                    const unsigned long long bal = __ballot(work);
                    if (bal) {
                        const bool light = n <= 4u;
                        const unsigned long long bl = __ballot(work && light), bh = bal & ~bl;
                        uint32_t base_l = 0, base_h = 0;
                        if (lane == 0) {
                            if (bl) base_l = atomicAdd(&s_qn, (uint32_t)__builtin_popcountll(bl));
                            if (bh) base_h = atomicAdd(&s_qh, (uint32_t)__builtin_popcountll(bh));
                        }
                        base_l = __builtin_amdgcn_readfirstlane(base_l); base_h = __builtin_amdgcn_readfirstlane(base_h);
                        const unsigned long long below = (1ull << lane) - 1ull;
                        if (work) {
                            const uint32_t at = light ? base_l + (uint32_t)__builtin_popcountll(bl & below)
                                                      : (uint32_t)PW_QCAP - 1u - (base_h + (uint32_t)__builtin_popcountll(bh & below));
                            rq[at] = (uint16_t)(i - c0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < PW_U; ++u) { o0s[u] = o0n[u]; o1s[u] = o1n[u]; }
            }
            __syncthreads();
            // ---- phase 2: first the reads of <= 4 calls (front of the queue) with four call slots -- one 16-byte load of calls, half the
            // slot arithmetic, three pair diagonals --, then the others with eight
            auto phase2 = [&](auto nbc, auto ec, const uint32_t qn, const bool back) {
            constexpr int NB = decltype(nbc)::value, E = decltype(ec)::value;          // call slots; queue entries per thread and trip
            static_assert(PW_NB == 8 && (NB == 4 || NB == 8), "one or two 16-byte loads per read");
            // everything of a queued read after its loads
            auto compute = [&](const bool act, const uint32_t i, const uint32_t o0, const uint32_t n, uint32_t (&v)[NB], int32_t (&r)[NB],
                               const uint32_t rraw0, const uint32_t rraw1, const int32_t s, const uint32_t mq) {
                (void)i; (void)rraw1;
                const bool owned = act && s >= P0 && s < P1;
                const bool lp_ok = lp_possible && owned && mq >= a.lpmd_min_qual && n > 1;
                const bool pdr_ok = act && a.want_pdr && n >= a.min_cpgs && mq >= a.pdr_min_qual;
                const uint32_t sm1 = (uint32_t)(s - 1);
                // slot liveness, span check, concordance state (pdr.rs:37-45 via readutil.rs:226-251): the tile kernel's table form
                const uint32_t nrow = min(n, (uint32_t)NB);
                uint32_t mk[NB];
                {
                    const uint4 ma = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[0];
                    mk[0] = ma.x; mk[1] = ma.y; mk[2] = ma.z; mk[3] = ma.w;
                    if constexpr (NB == 8) { const uint4 mb = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[1]; mk[4] = mb.x; mk[5] = mb.y; mk[6] = mb.z; mk[7] = mb.w; }
                }
                uint32_t acc = 0, xmax = act ? (v[0] & 0x7fffffffu) - sm1 : 0u;
#pragma unroll
                for (int k = 1; k < NB; ++k) {
                    xmax = max(xmax, __builtin_amdgcn_bitop3_b32(v[k] - sm1, mk[k], 0x7fffffffu, 0x80));   // a & b & c
                    v[k] = __builtin_amdgcn_bitop3_b32(v[k], v[0], mk[k], 0xe4);                             // live ? own word : the first call's
                    acc |= v[k] ^ v[0];
                }
                uint32_t bad_it = (xmax > (uint32_t)a.max_span) ? 1u : 0u;
                uint32_t disc = acc >> 31;
                const bool any_long = NB == 8 && __any(n > (uint32_t)PW_NB);
                if (any_long && n > (uint32_t)PW_NB) {
                    const uint32_t first = v[0] >> 31;
                    for (uint32_t k = PW_NB; k < n; ++k) {
                        const uint32_t x = a.cpg_pos[o0 + k];
                        disc |= (x >> 31) ^ first;
                        bad_it |= ((x & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
                    }
                }
                bad |= act ? bad_it : 0u;
                // windowed pair counts (readutil.rs:166-224): pairs (j < k) with min <= rel_k - rel_j <= max, diagonal by diagonal
                if (__any(lp_ok)) {
                    const uint32_t n_lp = lp_ok ? min(n, (uint32_t)NB) : 0u;
                    if constexpr (PACKED) {
                        constexpr int H = NB / 2;                                  // packed registers per operand family
                        uint32_t SQ[H], SO[H], Q[H], O[H];
#pragma unroll
                        for (int e = 0; e < H; ++e) SQ[e] = __builtin_amdgcn_perm(v[2 * e + 1], v[2 * e], 0x070c030cu);
#pragma unroll
                        for (int e = 0; e < H - 1; ++e) SO[e] = __builtin_amdgcn_perm(v[2 * e + 2], v[2 * e + 1], 0x070c030cu);
                        SO[H - 1] = __builtin_amdgcn_perm(0u, v[NB - 1], 0x070c030cu);
                        Q[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c010c00u); Q[1] = __builtin_amdgcn_perm(0u, rraw0, 0x0c030c02u);
                        O[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c020c01u);
                        if constexpr (NB == 8) {
                            Q[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c010c00u); Q[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c030c02u);
                            O[1] = __builtin_amdgcn_perm(rraw1, rraw0, 0x0c040c03u);
                            O[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c020c01u); O[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c0c0c03u);
                        } else {
                            O[1] = __builtin_amdgcn_perm(0u, rraw0, 0x0c0c0c03u);      // (slot 3, slot 4: dead by the table below)
                        }
                        {
                            const uint4 da = reinterpret_cast<const uint4 *>(&tabs.dtab[n_lp][0])[0], db = reinterpret_cast<const uint4 *>(&tabs.dtab[n_lp][0])[1];
                            Q[0] += da.x; Q[1] += da.y; O[0] += db.x; O[1] += db.y;
                            if constexpr (NB == 8) { Q[2] += da.z; Q[3] += da.w; O[2] += db.z; O[3] += db.w; }
                        }
                        const uint32_t KA = (0x8000u - (uint32_t)mind) * 0x10001u, KB = (0x8000u + (uint32_t)maxd) * 0x10001u;
                        uint32_t accIN = 0, accDD = 0;
#pragma unroll
                        for (int g = 1; g < NB; ++g) {
                            uint32_t orB = 0;
#pragma unroll
                            for (int m = 0; m < H; ++m) {
                                const int li = (g & 1) ? (g - 1) / 2 + m : g / 2 + m;      // index of the later operand in O (g odd) / Q (g even)
                                if (li > H - 1) break;
                                const uint32_t later = (g & 1) ? O[li] : Q[li], sl = (g & 1) ? SO[li] : SQ[li];
                                const uint32_t D = later - Q[m];
                                const uint32_t Bw = KB - D;
                                const uint32_t IN = __builtin_amdgcn_bitop3_b32(D + KA, Bw, 0x80008000u, 0x80);   // min <= distance <= max (readutil.rs:184, 196)
                                const uint32_t DD = IN & (sl ^ SQ[m]);
                                accIN += __builtin_popcount(IN);
                                accDD += __builtin_popcount(DD);
                                orB |= Bw;
                            }
                            if (!__any((orB & 0x80008000u) != 0u)) break;      // no lane has a pair within max_distance on this diagonal
                        }
                        a_c += accIN - accDD;
                        a_d += accDD;
                    } else {
#pragma unroll
                        for (int k = 0; k < NB; ++k) r[k] = ((uint32_t)k < n_lp) ? r[k] : (int32_t)((k + 1) << 24);
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
                        a_c += lp_n - lp_dd;
                        a_d += lp_dd;
                    }
                    // a read with more than 8 calls: the pairs whose LATER call is the 9th or beyond, from memory (divergent, rare)
                    if (any_long && lp_ok && n > (uint32_t)PW_NB) {
                        for (uint32_t k = PW_NB; k < n; ++k) {
                            const int32_t rk = (int32_t)rel[o0 + k];
                            const uint32_t mkk = a.cpg_pos[o0 + k] >> 31;
                            for (uint32_t jj = k; jj-- > 0;) {
                                const int32_t dist = rk - (int32_t)rel[o0 + jj];
                                if (dist > a.max_dist) break;          // readutil.rs:184 (anchors evicted)
                                if (dist < a.min_dist) continue;       // readutil.rs:196
                                if ((a.cpg_pos[o0 + jj] >> 31) == mkk) a_c += 1; else a_d += 1;
                            }
                        }
                    }
                }
                // PDR (pdr.rs:180-191): +1 coverage, +1 discordant for a discordant read, at each of the read's calls the stretch holds
                if (__any(pdr_ok && !bad_it)) {
                    auto insert = [&](const uint32_t word) {
                        const uint32_t p = word & 0x7fffffffu, d = p - (uint32_t)P0;
                        if (d >= Wp) return;
                        // CpG sites lie at least two positions apart: (d >> 1) spreads a dense stretch over consecutive slots
                        uint32_t h = (d >> 1) & (PW_S - 1), probes = 0;
                        while (probes++ < (uint32_t)PW_S) {
                            const uint32_t cur = atomicCAS(&tkey[h], PW_EMPTY, p);
                            if (cur == PW_EMPTY || cur == p) { atomicAdd(&tcov[h], 1u); if (disc) atomicAdd(&tdisc[h], 1u); return; }
                            h = (h + 1) & (PW_S - 1);
                        }
                        s_over = 1u;
                    };
                    const bool go = pdr_ok && !bad_it;
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        if (!__any(go && (uint32_t)k < n)) break;            // wave-uniform
                        if (go && (uint32_t)k < n) insert(v[k]);
                    }
                    if (any_long && go && n > (uint32_t)PW_NB)
                        for (uint32_t k = PW_NB; k < n; ++k) insert(a.cpg_pos[o0 + k]);
                }
            };
            // E entries per thread: their offsets, then ALL their loads, then the work -- twice the bytes in flight per lane at the register
            // cost of one 8-slot entry when the entries hold four slots
            for (uint32_t j0 = 0; j0 < qn; j0 += PW_B * E) {
                bool act[E];
                uint32_t ii[E], o0[E], n[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t j = j0 + (uint32_t)e * PW_B + tid;
                    act[e] = j < qn;
                    ii[e] = c0 + (act[e] ? (uint32_t)rq[back ? (uint32_t)PW_QCAP - 1u - j : j] : 0u);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    o0[e] = a.cpg_off[ii[e]];
                    const uint32_t o1 = a.cpg_off[ii[e] + 1];
                    n[e] = act[e] ? o1 - o0[e] : 0u;
                }
                uint32_t v[E][NB];
                int32_t r[E][NB];
                uint32_t rraw0[E], rraw1[E];
                int32_t s[E];
                uint32_t mq[E];
                bool inb = true;
#pragma unroll
                for (int e = 0; e < E; ++e) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) { v[e][k] = 0u; r[e][k] = 0; }
                    rraw0[e] = 0; rraw1[e] = 0;
                    inb = inb && (!act[e] || (unsigned long long)o0[e] + NB <= (unsigned long long)a.n_cpgs);
                }
                if (__all(inb)) {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (act[e]) {
                            const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0[e]);
                            v[e][0] = x.x; v[e][1] = x.y; v[e][2] = x.z; v[e][3] = x.w;
                            if constexpr (NB == 8) {
                                const u32x4_a4 y = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0[e] + 4);
                                v[e][4] = y.x; v[e][5] = y.y; v[e][6] = y.z; v[e][7] = y.w;
                                if constexpr (PACKED) { const u32x2_a1 z = *reinterpret_cast<const u32x2_a1 *>(rel + o0[e]); rraw0[e] = z.x; rraw1[e] = z.y; }
                                else {
                                    const u32x4_a2 z = *reinterpret_cast<const u32x4_a2 *>(rel + o0[e]);
                                    r[e][0] = (int32_t)(z.x & 0xffffu); r[e][1] = (int32_t)(z.x >> 16); r[e][2] = (int32_t)(z.y & 0xffffu); r[e][3] = (int32_t)(z.y >> 16);
                                    r[e][4] = (int32_t)(z.z & 0xffffu); r[e][5] = (int32_t)(z.z >> 16); r[e][6] = (int32_t)(z.w & 0xffffu); r[e][7] = (int32_t)(z.w >> 16);
                                }
                            } else {
                                if constexpr (PACKED) rraw0[e] = *reinterpret_cast<const u32_a1 *>(rel + o0[e]);
                                else {
                                    const u32x2_a2 z = *reinterpret_cast<const u32x2_a2 *>(rel + o0[e]);
                                    r[e][0] = (int32_t)(z.x & 0xffffu); r[e][1] = (int32_t)(z.x >> 16); r[e][2] = (int32_t)(z.y & 0xffffu); r[e][3] = (int32_t)(z.y >> 16);
                                }
                            }
                        }
                } else {                                        // the batch's last reads: a window of NB calls would leave the arrays
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (act[e]) {
#pragma unroll
                            for (int k = 0; k < NB; ++k) {
                                const uint32_t kk = o0[e] + min((uint32_t)k, n[e] - 1);
                                v[e][k] = a.cpg_pos[kk];
                                const uint32_t rv = (uint32_t)rel[kk];
                                if constexpr (PACKED) { if (k < 4) rraw0[e] |= rv << (8 * k); else rraw1[e] |= rv << (8 * (k - 4)); }
                                else r[e][k] = (int32_t)rv;
                            }
                        }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) { s[e] = a.read_start[ii[e]]; mq[e] = a.read_mapq[ii[e]]; }
#pragma unroll
                for (int e = 0; e < E; ++e) compute(act[e], ii[e], o0[e], n[e], v[e], r[e], rraw0[e], rraw1[e], s[e], mq[e]);
            }
            };
#ifndef MTH_PW_E4
#define MTH_PW_E4 2
#endif
            phase2(std::integral_constant<int, 4>{}, std::integral_constant<int, MTH_PW_E4>{}, s_qn, false);
            phase2(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{}, s_qh, true);
        }
        __syncthreads();
        const uint32_t over = s_over;
        __syncthreads();                                    // (s_over is cleared at the top of the next trip; the queue is done with)
        if (over && sub_shift > 8) { --sub_shift; continue; }      // more distinct sites than slots: the same stretch again in halves
        if (over) bad |= 2u;                                // cannot happen: 256 positions, 1024 slots
        lp_c += a_c; lp_d += a_d; n_read += a_r; n_valid += a_v;
        // rows: slots with coverage >= min_depth, sorted by position.  Bucket sort on the position (256 buckets per stretch): every
        // thread holds its slots in registers, so the table is rebuilt in place in bucket order; a key's final rank = start of its
        // bucket + the keys of that bucket below it (a few).
        constexpr int PER = PW_S / PW_B;
        uint32_t kk[PER], kc[PER], kd[PER], pib[PER];
        const int bshift = sub_shift - 8;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            kk[k] = tkey[tid * PER + k]; kc[k] = tcov[tid * PER + k]; kd[k] = tdisc[tid * PER + k];
            pib[k] = 0;
            if (kk[k] != PW_EMPTY && kc[k] < a.min_cov) kk[k] = PW_EMPTY;
            if (kk[k] != PW_EMPTY) pib[k] = atomicAdd(&bcnt[(kk[k] - (uint32_t)P0) >> bshift], 1u);
        }
        __syncthreads();                                    // every slot is in registers now: the table can be overwritten
        const uint32_t m_b = bcnt[tid];
        const uint32_t incl = wave_scan_incl(m_b);
        if (lane == 63) ws[wave + 1] = incl;
        __syncthreads();
        if (tid == 0) { ws[0] = 0; for (int w = 1; w <= PW_B / 64; ++w) ws[w] += ws[w - 1]; }
        __syncthreads();
        const uint32_t n_rows = ws[PW_B / 64];
        bbase[tid] = ws[wave] + incl - m_b;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (kk[k] != PW_EMPTY) {
                const uint32_t dst = bbase[(kk[k] - (uint32_t)P0) >> bshift] + pib[k];
                tkey[dst] = kk[k]; tcov[dst] = kc[k]; tdisc[dst] = kd[k];
            }
        __syncthreads();
        for (uint32_t j = tid; j < n_rows; j += PW_B) {
            const uint32_t key = tkey[j];
            const uint32_t bk = (key - (uint32_t)P0) >> bshift, b0 = bbase[bk], b1 = b0 + bcnt[bk];
            uint32_t rnk = b0;
            for (uint32_t i = b0; i < b1; ++i) rnk += tkey[i] < key ? 1u : 0u;
            SiteRec rr;
            rr.pos = (int32_t)key; rr.n_disc = tdisc[j]; rr.n_conc = tcov[j] - rr.n_disc; rr.pad = 0;
            out[rows_out + rnk] = rr;
        }
        rows_out += n_rows;
        P0l = P1;
        __syncthreads();                                    // the table is cleared by the next trip
    }
    if (bad & 1u) atomicOr(const_cast<uint32_t *>(&a.st->err), (uint32_t)ERRB_SPAN);
    if (bad & 2u) atomicOr(const_cast<uint32_t *>(&a.st->err), (uint32_t)ERRB_CAPACITY);
    // LPMD partials: wave sums -> LDS -> one atomic per counter into the tile's bucket
    if (a.want_lpmd) {
        const uint32_t r0 = wave_sum(lp_c), r1 = wave_sum(lp_d), r2 = wave_sum(n_read), r3 = wave_sum(n_valid);
        if (lane == 0) { red[0][wave] = r0; red[1][wave] = r1; red[2][wave] = r2; red[3][wave] = r3; }
        __syncthreads();
        if (tid < 4) {
            uint32_t sum = 0;
            for (int w = 0; w < PW_B / 64; ++w) sum += red[tid][w];
            if (sum) atomicAdd(a.bucket + a.nbk + (size_t)(t >> TILE_BUCKET_SHIFT) * 4 + tid, (unsigned long long)sum);
        }
    }
    if (tid == 0) {
        a.tile_cnt[t] = rows_out;
        if (rows_out) atomicAdd(a.bucket + (t >> TILE_BUCKET_SHIFT), (unsigned long long)rows_out);
    }
}

void launch_tile_wide(const TileArgs &a, uint32_t ntiles, int shift, bool rel8, hipStream_t s) {
    const uint32_t grid = ((ntiles + 7) / 8) * 8;   // whole rows of 8 XCDs (remap in the kernel)
    if (shift == 14) {
        if (rel8) hipLaunchKernelGGL((k_pdr_lpmd_wide<14, uint8_t>), dim3(grid), dim3(PW_B), 0, s, a, ntiles);
        else hipLaunchKernelGGL((k_pdr_lpmd_wide<14, uint16_t>), dim3(grid), dim3(PW_B), 0, s, a, ntiles);
    } else if (shift == 16) {
        if (rel8) hipLaunchKernelGGL((k_pdr_lpmd_wide<16, uint8_t>), dim3(grid), dim3(PW_B), 0, s, a, ntiles);
        else hipLaunchKernelGGL((k_pdr_lpmd_wide<16, uint16_t>), dim3(grid), dim3(PW_B), 0, s, a, ntiles);
    } else {
        if (rel8) hipLaunchKernelGGL((k_pdr_lpmd_wide<15, uint8_t>), dim3(grid), dim3(PW_B), 0, s, a, ntiles);
        else hipLaunchKernelGGL((k_pdr_lpmd_wide<15, uint16_t>), dim3(grid), dim3(PW_B), 0, s, a, ntiles);
    }
}

}  // namespace mth
