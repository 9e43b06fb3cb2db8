// mth_fdrp.hip -- FDRP and qFDRP (fdrp.rs:176-246, 51-145; qfdrp.rs:188-258, 109-157) on gfx950.
//
// Reference, per CpG site c: keep up to max_depth covering reads (reservoir sampling beyond that) as
// 403-slot byte arrays centred on c (bit0 base covered, bit1 CpG call, bit2 methylated; a read that does
// not fit +-201 bp is dropped), flush with strict '<' by reads passing mapq with >= 1 CpG, and at
// flush evaluate all C(n,2) read pairs: skip if fewer than min_overlap bases are covered by both;
// FDRP counts pairs with >= 1 position that both reads cover AND call with different states, qFDRP
// adds (#such positions) / (#positions both reads call) in lexicographic (i,j) order; both divide by
// (n*(n-1)) as f32 / 2.0 -- skipped pairs stay in the denominator.
//
// Device: ONE WAVE per site (sites come from the tile pipeline's site discovery, mth_sites.hip).
//  walk      the candidate reads (linear read index, file order) are inspected 64 at a time, one per lane:
//            mapq / n_cpgs filters, "calls c" and "first call > c" (the flush test) become ballots.  The common
//            chunk -- hits, then optionally reads starting past c+1 -- is handled without a serial loop: each
//            hit lane writes its row {cpg_off, n, start, end, FD_NB packed calls} into the wave's LDS slot
//            array at slot = arrival order.  Anything else (reservoir replacement, a flush followed by
//            re-opening reads, spans > 200 bp where add_read can drop a read) runs the reference's
//            per-read state machine over the ballots, still without touching memory.
//  finalize  lane = slot loads its row.  The 403-byte arrays are never materialised.  Compact path: every
//            position a stored read calls is a discovered site, so with the 64 sites around c covering
//            +-200 bp a read is three 64-bit masks (calls, covered calls, methylated covered calls) and a
//            pair is a few and/xor/popcounts, evaluated pair-parallel (lane k = k-th pair in (i,j) order).
//            Otherwise (denser window, region-edge site, spans > 200 bp): i-uniform loop, read i's calls
//            matched against every slot's call registers by an integer min-chain.
//  qFDRP sum accumulated in the reference's lexicographic (i,j) order so the f32 rounding matches: a DPP
//            wave_shr:1 chain x[l] = x[l-1] + term[l] (one VALU per pair, no scalar work).
// Both measures come out of one walk.  History and counters: profiles/r01_fdrp_pmc.md.
//
// Reservoir branch (depth > max_depth): the reference draws from an OS-seeded RNG (fdrp.rs:90), so
// there is nothing to be bit-equal to; device and oracle share the counter-based sample_j below.
#include <cmath>
#include <type_traits>

#include "mth_ctx.h"
#include "mth_scan.h"

namespace mth {

constexpr int FD_WIN = 201;   // MAX_READ_LEN, fdrp.rs:10
// Stored reads of a site (template parameter SLOTS of the walk): 64 -- one lane each -- in the main pass.  With max_depth > 64 a
// site that holds more than 64 reads at once is flagged (flags = 2) and redone by a second pass with 256 slots, whose
// finalize works pair-parallel from the LDS rows (a sorted merge of the two reads' calls per pair) instead of lane = slot.
// With max_depth > 256 a site that holds more than 256 reads at once is flagged again (flags = 3) and redone by a third pass
// whose rows live in HBM scratch, max_depth of them per wave (SLOTS = 0: capacity at run time); same pair-parallel finalize.
// max_depth above FD_DEPTH_MAX is refused (MTH_ERR_CAPACITY, loud): the pair index arithmetic is 32-bit.
constexpr int FD_SLOTS_DEEP = 256;
constexpr uint32_t FD_DEPTH_MAX = 16384;
// FD_NB (template parameter of the walk): calls of a stored read held in the slot's registers (8 or 16)

struct FdrpArgs {
    const int32_t  *read_start, *read_end;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const uint32_t *idx;
    const DevState *sites_st;
    const int32_t  *site_pos;
    const uint32_t *site_nc, *site_nd;   // discovery's per-site read counts (reads passing mapq that call the site)
    const int32_t *grp_tab;              // a contig group's batch: n voff then n tids (else nullptr); the draw below is keyed by the real site
    uint32_t grp_n;
    DevState *st;
    float    *fdrp, *qfdrp;     // per candidate site
    uint32_t *nreads, *flags;
    unsigned long long seed;
    int32_t idx_base, max_span, tid, min_overlap;
    int32_t region_beg, region_end;   // sites are discovered for [region_beg, region_end) only
    uint32_t n_reads, n_cpgs, min_depth, max_depth;
    uint8_t min_qual;
    uint32_t *rows_scratch;           // SLOTS = 0: slots_cap rows of (4 + FD_NB) words per wave of the launch
    const uint16_t *pair_tab;         // SLOTS = 64: (i | j << 8) of the k-th pair of n reads at [n (n-1) (n-2) / 6 + k], n <= 64
    uint32_t slots_cap;
    uint32_t only_flag;               // SLOTS = 64: 0 = every site, else only the sites k_fdrp_walk4 handed back
    unsigned long long *redo_mask;    // per block of 64 consecutive sites: the sites k_fdrp_walk4 / k_fdrp_tile handed back (written for every block)
    // k_fdrp_tile / k_fdrp_chain: the non-zero qFDRP terms of a site as byte codes, in the reference's (i, j) order
    uint8_t  *terms;                  // term lists (budget bytes), claimed per tile through *cursor
    unsigned long long *cursor;
    unsigned long long budget;
    unsigned long long *site_off;     // per site: first byte of its list
    uint32_t *site_nz, *site_disc;    // per site: listed terms, discordant pairs
    uint32_t *redo_list, *redo_cnt;   // the sites k_fdrp_tile / k_fdrp_wtile handed back, one after the other (k_fdrp_walk takes them wave by wave)
    uint32_t wide_rows;               // tests: k_fdrp_tile keeps every site's rows in the eight-word form (METHEOR_FDRP_TILE_WIDE_ROWS=1)
    uint32_t no_compact;              // the site list holds only the sites that can produce a row (k_fdrp_wtile): no window of ALL sites around c
};

// the oracle's orc_sample_j: splitmix64 over (seed, tid, pos, total) -> 1..=total
__device__ __forceinline__ int32_t sample_j(unsigned long long seed, int32_t tid, int32_t pos, int32_t total) {
    unsigned long long z = seed ^ (((unsigned long long)(uint32_t)tid << 32) | (uint32_t)pos);
    z += 0x9e3779b97f4a7c15ULL * (unsigned long long)(uint32_t)total;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z = z ^ (z >> 31);
    return (int32_t)(z % (unsigned long long)(uint32_t)total) + 1;
}

// the reservoir draw of site c of the batch: on the site's own contig and position when the batch is a contig group
__device__ __forceinline__ int32_t sample_site(const int32_t *__restrict__ grp_tab, uint32_t grp_n, unsigned long long seed, int32_t tid, int32_t c, int32_t total) {
    if (grp_tab) {
        uint32_t lo = 0, hi = grp_n;                              // last contig whose offset is <= c
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (grp_tab[mid] <= c) lo = mid; else hi = mid; }
        tid = grp_tab[grp_n + lo];
        c -= grp_tab[lo];
    }
    return sample_j(seed, tid, c, total);
}

// PMC of the first version (profiles/r01_fdrp_pmc.md): the per-CU scalar unit was ~81 % busy -- wave-uniform
// loops compiled as divergent ones, '&&' / '|=' over comparisons became s_and/s_or_b64 chains.  Hence:
// wave-uniform values are made explicitly scalar with readfirstlane, per-lane logic is integer arithmetic
// or selects, and skipped pairs add +0.0 (x + 0.0 == x exactly) instead of branching.
constexpr uint32_t FD_NOPOS = 0xffffffffu;   // "no call" in a slot's call registers (never equals a 31-bit position)

__device__ __forceinline__ uint32_t sgpr(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int32_t sgpr(int32_t x) { return (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)x); }

template <int FD_NB, int SLOTS>
__global__ __launch_bounds__(256, (FD_NB == 8 && SLOTS == 64) ? 8 : 1) void k_fdrp_walk(const FdrpArgs a) {
    const int FD_SLOTS = SLOTS ? SLOTS : (int)a.slots_cap;
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = sgpr((uint32_t)((blockIdx.x * 256 + threadIdx.x) >> 6)), n_waves = (gridDim.x * 256) >> 6;
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    // per wave: window position (p - (c - FD_WIN)) -> index of that CpG in the wave's 64-site window
    __shared__ uint8_t s_bit[4][2 * FD_WIN + 1 + 13];
    uint8_t *const bit_of = s_bit[threadIdx.x >> 6];
    __shared__ float s_terms[4][64];           // per wave: a round's non-zero qFDRP terms, packed
    float *const s_term = s_terms[threadIdx.x >> 6];
    // per wave: the stored reads of the open segment, one row per slot: {cpg_off, n_calls, start, end, FD_NB packed calls}
    constexpr int ROW = 4 + FD_NB;
    __shared__ __attribute__((aligned(16))) uint32_t s_rows[4][(SLOTS ? SLOTS : 1) * ROW];
    uint32_t *const rows = SLOTS ? s_rows[threadIdx.x >> 6] : a.rows_scratch + (size_t)wave_id * a.slots_cap * ROW;
    // ham / ncpg for ncpg <= 16, computed here by the same f32 division the pair loop would do (17 instructions per round otherwise)
    constexpr int FD_QN = 17;
    __shared__ float s_quot[SLOTS == 64 ? FD_QN * FD_QN : 1];
    if (SLOTS == 64) {
        for (int t = threadIdx.x; t < FD_QN * FD_QN; t += 256) s_quot[t] = (float)(t / FD_QN) / (float)(t % FD_QN);
        __syncthreads();
    }
    // A site's segments hold at most the reads that call it and pass mapq -- the count the discovery pass left beside the
    // position -- so a site whose count is below min_depth cannot produce a row (fdrp.rs:239-243) and is not walked: at
    // WGBS depths (config 3: 9.7x against -d 10) that is more than half of the sites.  The count of the wave's NEXT site is
    // requested one site ahead, so a run of skipped sites is not a run of exposed round trips.
    // (after k_fdrp_walk4 -- only_flag set -- the pass walks the set bits of that kernel's per-block hand-back masks instead: one
    // word per 64 sites to look at, not one per site)
    const bool from_list = SLOTS == 64 && a.only_flag != 0u && a.redo_list != nullptr;
    const bool listed = SLOTS == 64 && a.only_flag != 0u && !from_list;
    const uint32_t n_list = from_list ? sgpr(*a.redo_cnt) : 0u;
    uint32_t cov_j = 0;
    if (SLOTS == 64 && !listed && !from_list && wave_id < n_sites) cov_j = a.site_nc[wave_id] + a.site_nd[wave_id];
    uint32_t l_blk = wave_id, l_cur = 0;
    unsigned long long l_m = 0;
    // (A site pipeline -- the wave holding this site's position and index entries and the next site's position as scalars,
    // requesting the next site's index entries and the position two sites ahead at the top of a site -- was built twice: on dense
    // data it changes nothing (1.4856 vs 1.4861 ms; 1.3456 vs 1.3383), at WGBS depth it is slower (0.880 -> 0.930 ms on a
    // 16 M-read chr1-sized contig): a site's time is instruction issue, not this chain's latency.)
    for (uint32_t it = wave_id;; it += n_waves) {
        uint32_t j;
        if (listed) {
            bool end = false;
            while (l_m == 0ull) {
                if ((uint64_t)l_blk * 64u >= n_sites) { end = true; break; }
                const unsigned long long w = a.redo_mask[l_blk];
                l_m = ((unsigned long long)sgpr((uint32_t)(w >> 32)) << 32) | sgpr((uint32_t)w);
                l_cur = l_blk; l_blk += n_waves;
            }
            if (end) break;
            j = l_cur * 64u + (uint32_t)__builtin_ctzll(l_m);
            l_m &= l_m - 1ull;
        } else if (from_list) {
            // (k_fdrp_tile's hand-backs come in runs -- a whole tile of a dense stretch -- and cost up to 100 us each on the
            // call-by-call path: taken from a list, a run is spread over as many waves as it has sites)
            if (it >= n_list) break;
            j = sgpr(a.redo_list[it]);
        } else {
            j = it;
            if (j >= n_sites) break;
        }
        if (SLOTS == 64 && !listed && !from_list) {
            const uint32_t cov = sgpr(cov_j);
            const uint32_t jn = j + n_waves;
            if (jn < n_sites) cov_j = a.site_nc[jn] + a.site_nd[jn];
            if (cov < a.min_depth) {
                if (lane == 0) { a.fdrp[j] = 0.0f; a.qfdrp[j] = 0.0f; a.nreads[j] = 0u; a.flags[j] = 0u; }
                continue;
            }
        }
        const int32_t c = sgpr(a.site_pos[j]);
        // the 64-site window of the compact finalize depends on j alone: requested here, a full walk before it is used
        const uint32_t j0w = (j >= 32u) ? min(j - 32u, n_sites > 64u ? n_sites - 64u : 0u) : 0u;
        const int32_t spw = (j0w + (uint32_t)lane < n_sites) ? a.site_pos[j0w + lane] : 0x7fffffff;
        const uint32_t lo = sgpr(min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads));
        const uint32_t hi = sgpr(min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads));
        // wave-uniform segment state
        int32_t total = 0, sampled = 0;
        bool entry = false, have = false, deep = false;
        float res_f = 0.0f, res_q = 0.0f;
        uint32_t res_n = 0;
        const bool win_check = a.max_span > 200;   // a stored read calls c and spans <= 200 bp: all its calls are inside +-201
        if (SLOTS == FD_SLOTS_DEEP && a.flags[j] != 2u) continue;   // second pass: only the sites the 64-slot pass could not hold
        if (SLOTS == 0 && a.flags[j] != 3u) continue;               // third pass: only the sites the 256-slot pass could not hold

        auto finalize = [&]() {   // compute_fdrp / compute_qfdrp over slots 0..sampled-1
            const int nS = sampled;
            uint32_t disc = 0;     // per lane j: discordant pairs (i, j)
            float q = 0.0f;
            // lane = slot: the slot's row; calls are packed words (position | state << 31), FD_NOPOS = none
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t r_o0 = 0, r_n = 0, vw[FD_NB];
            int32_t r_s = 0, r_e = 0;
            {
                const uint32_t *r = rows + (lane < nS ? lane : 0) * ROW;
                r_o0 = r[0]; r_n = r[1]; r_s = (int32_t)r[2]; r_e = (int32_t)r[3];
#pragma unroll
                for (int k = 0; k < FD_NB; ++k) vw[k] = r[4 + k];
            }
            const bool any_long = __any(lane < nS && r_n > (uint32_t)FD_NB);   // uniform: some stored read has calls beyond its registers
            // Compact path.  Every position a stored read calls is one of the discovered sites, so when the
            // 64 sites around c cover +-200 bp each call maps to a bit: a stored read becomes three 64-bit
            // masks (calls, covered calls, methylated covered calls) and a pair costs a few and/xor/popcounts
            // instead of an FD_NB-way compare per call.  Denser windows and spans > 200 bp keep the
            // call-by-call path below (same results).
            unsigned long long mC = 0, mA = 0, mM = 0;
            bool compact = false;
            // (halo reads of a region slice call positions outside the region; those are not in the site list)
            if (!win_check && !a.no_compact && (int64_t)c - 200 >= a.region_beg && (int64_t)c + 200 < a.region_end) {
                const uint32_t j0 = j0w;
                const int32_t sp = spw;
                const int32_t sp_lo = __builtin_amdgcn_readlane(sp, 0), sp_hi = __builtin_amdgcn_readlane(sp, 63);
                compact = (j0 == 0u || sp_lo < c - 200) && (int64_t)sp_hi > (int64_t)c + 200;       // absent sites read as +inf
                if (compact) {
                    const uint32_t rel = (uint32_t)(sp - (c - FD_WIN));
                    if (rel <= 2u * FD_WIN) bit_of[rel] = (uint8_t)lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // mC: the positions the read calls; mM: those it calls methylated.  A read calls positions in
                    // [start - 1, end] only, so the one call that can lie outside the covered bases is its first, at start - 1
                    // (mA below).  Registers past the last call repeat it (see the walk): the same bit again.
                    auto add_call = [&](const uint32_t w) {
                        const uint32_t rel_p = (w & 0x7fffffffu) - (uint32_t)(c - FD_WIN);
                        const unsigned long long b = 1ull << bit_of[rel_p];
                        mC |= b;
                        mM |= b & (unsigned long long)((long long)(int32_t)w >> 31);
                    };
                    if (lane < nS) {
#pragma unroll
                        for (int k = 0; k < FD_NB; ++k) add_call(vw[k]);
                        if (any_long)
                            for (uint32_t t = FD_NB; t < r_n; ++t) add_call(a.cpg_pos[r_o0 + t]);
                        const uint32_t p0 = vw[0] & 0x7fffffffu;
                        const unsigned long long b0 = 1ull << bit_of[p0 - (uint32_t)(c - FD_WIN)];
                        mA = ((int32_t)p0 >= r_s) ? mC : mC & ~b0;
                        mM &= mA;
                    }
                }
            }
#define MTH_FD_DPP x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)) + term;
#define MTH_FD_DPP8 MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP
            if (compact) {
                // Pair-parallel: the slots' {start, end, masks} go back to the LDS rows, lane k of a round takes
                // the k-th pair of the reference's lexicographic (i,j) order (all 64 lanes busy instead of the
                // j > i lanes of one i), and the round's 64 terms are chained in that order by DPP adds:
                // x[l] = x[l-1] + term[l], lane 0 seeded with the running sum -- the reference's rounding.
                if (lane < nS) {
                    uint32_t *r = rows + lane * ROW;
                    r[0] = (uint32_t)r_s; r[1] = (uint32_t)r_e;
                    r[2] = (uint32_t)mC; r[3] = (uint32_t)(mC >> 32); r[4] = (uint32_t)mA; r[5] = (uint32_t)(mA >> 32);
                    r[6] = (uint32_t)mM; r[7] = (uint32_t)(mM >> 32);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int P = (nS * (nS - 1)) >> 1;
                const bool lut_ok = !__any(lane < nS && r_n >= (uint32_t)FD_QN);   // wave-uniform: every ncpg of the site is in the table
                // k -> (i, j) from a table (the closed form -- a square root, three 32-bit multiplies, two corrections -- was more
                // than half of a round's vector instructions); the next round's entry is requested a round ahead
                const uint16_t *const tab = a.pair_tab + (uint32_t)(nS * (nS - 1) * (nS - 2)) / 6u;
                uint32_t ent_next = tab[min(lane, P - 1)];
                for (int k0 = 0; k0 < P; k0 += 64) {
                    const uint32_t ent = ent_next;
                    if (k0 + 64 < P) ent_next = tab[min(k0 + 64 + lane, P - 1)];
                    const int pi = (int)(ent & 0xffu), pj = (int)(ent >> 8);
                    const uint32_t *ri = rows + pi * ROW, *rj = rows + pj * ROW;
                    const int32_t si = (int32_t)ri[0], ei = (int32_t)ri[1], sj = (int32_t)rj[0], ej = (int32_t)rj[1];
                    const int32_t ov = min(ei, ej) - max(si, sj) + 1;        // get_num_overlap_bases, fdrp.rs:97-107
                    const bool pair_ok = (k0 + lane < P) & (max(ov, 0) >= a.min_overlap);   // fdrp.rs:134
                    const uint32_t ncpg = __builtin_popcount(ri[2] & rj[2]) + __builtin_popcount(ri[3] & rj[3]);   // qfdrp.rs:109-119
                    const uint32_t ham = __builtin_popcount(ri[4] & rj[4] & (ri[6] ^ rj[6])) +
                                         __builtin_popcount(ri[5] & rj[5] & (ri[7] ^ rj[7]));                      // fdrp.rs:114-115
                    disc += (pair_ok && ham != 0u) ? 1u : 0u;                // fdrp.rs:138-140
                    const float quot = lut_ok ? s_quot[ham * FD_QN + ncpg] : (float)ham / (float)ncpg;
                    const float term0 = pair_ok ? quot : 0.0f;               // qfdrp.rs:152; +0.0 for skipped pairs
                    // x + 0.0 == x, so only the non-zero terms (NaN included: 0 / 0 when two reads share no CpG) have to be
                    // chained, in their order: they are packed into the low lanes through LDS first.  (The chain is one dependent
                    // VALU per term and was ~30 % of a VALU-bound kernel; most pairs of a site agree and contribute +0.0.)
                    const unsigned long long nz = __ballot(term0 != 0.0f);
                    const int m_nz = __popcll(nz);
                    if (m_nz == 0) continue;                                 // wave-uniform
                    if (term0 != 0.0f) s_term[__builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u))] = term0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const float term = lane < m_nz ? s_term[lane] : 0.0f;
                    float x = (lane == 0) ? q + term : term;
                    const int steps = m_nz - 1;
                    const int up8 = (steps + 7) & ~7;                        // lanes past the last term hold +0.0: sliding is exact
                    const bool slide = up8 <= 56;
                    const int blocks = slide ? up8 >> 3 : 7;
                    for (int b8 = 0; b8 < blocks; ++b8) { MTH_FD_DPP8 }
                    if (!slide) { MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP MTH_FD_DPP }
                    q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), slide ? up8 : 63));
                    __builtin_amdgcn_wave_barrier();                         // s_term is rewritten by the next round
                }
            }
            for (int i = 0; !compact && i + 1 < nS; ++i) {
                const uint32_t bn = __builtin_amdgcn_readlane(r_n, i);
                const int32_t bs = __builtin_amdgcn_readlane(r_s, i), be = __builtin_amdgcn_readlane(r_e, i);
                const int32_t mx = max(bs, r_s);
                const int32_t ov = min(be, r_e) - mx + 1;                    // get_num_overlap_bases, fdrp.rs:97-107
                const bool pair_ok = (lane > i) & (lane < nS) & (max(ov, 0) >= a.min_overlap);   // fdrp.rs:134 (an empty overlap counts 0 bases)
                uint32_t ham = 0, ncpg = 0;
                {
                // One call pw of read i against every slot.  Integer arithmetic only -- comparisons whose
                // results are OR-ed together compile to s_or_b64 chains on the (per-CU) scalar unit.
                auto match = [&](const uint32_t pw) {
                    const uint32_t p = pw & 0x7fffffffu;
                    if (win_check && (uint32_t)((int32_t)p - (c - FD_WIN)) > 2u * FD_WIN) return;   // outside the 403-slot array (uniform)
                    // key = rotl(word ^ p, 1): 0 = the slot calls p unmethylated, 1 = methylated, >= 2 = other position
                    uint32_t mn = 0xffffffffu;
#pragma unroll
                    for (int t = 0; t < FD_NB; ++t) {
                        const uint32_t x = vw[t] ^ p;
                        mn = min(mn, __builtin_amdgcn_alignbit(x, x, 31));
                    }
                    if (any_long) {                                           // rare: more than FD_NB calls in a stored read
                        for (uint32_t t = FD_NB; t < r_n; ++t) {
                            const uint32_t x = a.cpg_pos[r_o0 + t] ^ p;
                            mn = min(mn, __builtin_amdgcn_alignbit(x, x, 31));
                        }
                    }
                    const uint32_t fnd = 1u - min(mn >> 1, 1u);
                    ncpg += fnd;                                              // get_num_overlap_cpgs, qfdrp.rs:109-119 (bit1 & bit1)
                    // hamming / is_discordant: both cover p (bit0), both call it, states differ (fdrp.rs:114-115).
                    // A read calls only positions in [start-1, end], so for a position both call,
                    // "both cover it" is p >= max(start_i, start_j).
                    const uint32_t cov = (uint32_t)(((int32_t)p - mx) >> 31) + 1u;
                    ham += fnd & cov & (mn ^ (pw >> 31));
                };
                // read i's first FD_NB calls come from lane i's registers (no memory in the pair loop);
                // nested so a read with bn calls costs bn+1 uniform branches
                const uint32_t bn_eff = bn;
                [&]() {
#define MTH_FD_STEP(K) if ((K) >= FD_NB || (uint32_t)(K) >= bn_eff) return; match(__builtin_amdgcn_readlane(vw[(K) < FD_NB ? (K) : 0], i));
                    MTH_FD_STEP(0) MTH_FD_STEP(1) MTH_FD_STEP(2) MTH_FD_STEP(3) MTH_FD_STEP(4) MTH_FD_STEP(5) MTH_FD_STEP(6) MTH_FD_STEP(7)
                    MTH_FD_STEP(8) MTH_FD_STEP(9) MTH_FD_STEP(10) MTH_FD_STEP(11) MTH_FD_STEP(12) MTH_FD_STEP(13) MTH_FD_STEP(14) MTH_FD_STEP(15)
#undef MTH_FD_STEP
                }();
                if (bn_eff > (uint32_t)FD_NB) {                               // beyond the registers: wave-uniform scalar loads
                    const uint32_t bo0 = __builtin_amdgcn_readlane(r_o0, i);
                    for (uint32_t k = FD_NB; k < bn_eff; ++k) match(a.cpg_pos[bo0 + k]);
                }
                }
                disc += (pair_ok && ham != 0u) ? 1u : 0u;                     // fdrp.rs:138-140
                // qfdrp.rs:152 in the reference's lexicographic (i,j) order, so the f32 rounding matches: the
                // running sum sits in lane i, the pair terms in lanes i+1..nS-1 (+0.0 for skipped pairs leaves
                // an f32 sum unchanged); each DPP step computes x[j] = x[j-1] + term[j], so after nS-1-i steps
                // lane nS-1 holds ((q + t[i+1]) + t[i+2]) + ... -- one VALU and no scalar work per element.
                const float term = pair_ok ? (float)ham / (float)ncpg : 0.0f;
                float x = (lane == i) ? q : term;
                // Blocks of 8 steps; lanes >= nS hold +0.0, so overshooting just slides the finished sum to a
                // higher lane (while one exists).
                const int steps = nS - 1 - i;
                const int up8 = (steps + 7) & ~7;
                const bool slide = i + up8 <= 63;
                const int blocks = slide ? up8 >> 3 : steps >> 3;
                for (int b8 = 0; b8 < blocks; ++b8) { MTH_FD_DPP8 }
                if (!slide) for (int st = blocks << 3; st < steps; ++st) { MTH_FD_DPP }
                q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), slide ? i + up8 : nS - 1));
            }
#undef MTH_FD_DPP8
#undef MTH_FD_DPP
            uint32_t n_disc = disc;                                           // wave sum of the per-lane counts
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) n_disc += __shfl_xor(n_disc, o, 64);
            n_disc = sgpr(n_disc);
            // (num_reads * (num_reads - 1)) as f32 / 2.0 in usize arithmetic (fdrp.rs:143)
            const unsigned long long prod = (unsigned long long)(long long)nS * (unsigned long long)((long long)nS - 1);
            const float den = (float)prod / 2.0f;
            res_f = (float)n_disc / den;
            res_q = q / den;
            res_n = (uint32_t)nS;
            have = true;
            __builtin_amdgcn_wave_barrier();                                  // LDS reads done before the next segment's / site's writes
        };

        // SLOTS > 64: lane = pair.  Rows stay {cpg_off, n_calls, start, end, FD_NB calls}; the k-th pair of the reference's
        // lexicographic (i, j) order goes to lane k of a round, walks the two sorted call lists once (calls beyond the row's
        // registers come from memory), and the round's 64 terms are chained in that order by DPP adds as in the compact path.
        auto finalize_deep = [&]() {
            const int nS = sampled;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t disc = 0;
            float q = 0.0f;
            const int P = (nS * (nS - 1)) >> 1;
            const int twoN = 2 * nS;
            const float bq = (float)(twoN - 1);
            auto call_of = [&](const uint32_t *r, uint32_t t) { return t < (uint32_t)FD_NB ? r[4 + t] : a.cpg_pos[r[0] + t]; };
#define MTH_FD_DPP x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)) + term;
            for (int k0 = 0; k0 < P; k0 += 64) {
                const int k = min(k0 + lane, P - 1);
                // f32 estimate of the row, then exact integer steps: beyond ~12 000 stored reads the estimate can be two rows off near the
                // end of the list (one conditional step each way was not enough: nS = 12 000, pair 71 993 996)
                int pi = (int)((bq - __builtin_sqrtf(fmaxf(bq * bq - 8.0f * (float)k, 0.0f))) * 0.5f);
                pi = max(0, min(pi, nS - 2));
                int off = (pi * (twoN - pi - 1)) >> 1;
                while (k < off) { pi -= 1; off = (pi * (twoN - pi - 1)) >> 1; }
                while (pi + 2 < nS && k >= (((pi + 1) * (twoN - pi - 2)) >> 1)) { pi += 1; off = (pi * (twoN - pi - 1)) >> 1; }
                const int pj = k - off + pi + 1;
                const uint32_t *ri = rows + pi * ROW, *rj = rows + pj * ROW;
                const int32_t si = (int32_t)ri[2], ei = (int32_t)ri[3], sj = (int32_t)rj[2], ej = (int32_t)rj[3];
                const int32_t mx = max(si, sj);
                const int32_t ov = min(ei, ej) - mx + 1;                     // get_num_overlap_bases, fdrp.rs:97-107
                const bool pair_ok = (k0 + lane < P) & (max(ov, 0) >= a.min_overlap);   // fdrp.rs:134
                uint32_t ham = 0, ncpg = 0;
                const uint32_t ni = ri[1], nj = rj[1];
                uint32_t ti = 0, tj = 0;
                while (ti < ni && tj < nj) {
                    const uint32_t wi = call_of(ri, ti), wj = call_of(rj, tj);
                    const uint32_t pa = wi & 0x7fffffffu, pb = wj & 0x7fffffffu;
                    if (pa == pb) {
                        // (positions outside the reference's 403-slot array around c are not compared, as in the 64-slot path)
                        if (!(win_check && (uint32_t)((int32_t)pa - (c - FD_WIN)) > 2u * FD_WIN)) {
                            ncpg += 1;                                        // get_num_overlap_cpgs, qfdrp.rs:109-119
                            ham += ((int32_t)pa >= mx && ((wi ^ wj) >> 31)) ? 1u : 0u;   // fdrp.rs:114-115
                        }
                        ++ti; ++tj;
                    } else if (pa < pb) ++ti; else ++tj;
                }
                disc += (pair_ok && ham != 0u) ? 1u : 0u;                    // fdrp.rs:138-140
                const float term = pair_ok ? (float)ham / (float)ncpg : 0.0f;   // qfdrp.rs:152; +0.0 for skipped pairs
                float x = (lane == 0) ? q + term : term;
                const int steps = min(64, P - k0) - 1;
                for (int st = 0; st < steps; ++st) { MTH_FD_DPP }
                q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), steps));
            }
#undef MTH_FD_DPP
            uint32_t n_disc = disc;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) n_disc += __shfl_xor(n_disc, o, 64);
            n_disc = sgpr(n_disc);
            const unsigned long long prod = (unsigned long long)(long long)nS * (unsigned long long)((long long)nS - 1);
            const float den = (float)prod / 2.0f;                            // fdrp.rs:143
            res_f = (float)n_disc / den;
            res_q = q / den;
            res_n = (uint32_t)nS;
            have = true;
            __builtin_amdgcn_wave_barrier();
        };
// (not a wrapper lambda: one more call level and the compiler stops inlining finalize -- its by-reference captures then live in
// scratch memory and the walk runs 3.5x slower)
#define MTH_FD_FINISH() do { if constexpr (SLOTS != 64) finalize_deep(); else finalize(); } while (0)

        // Candidates are inspected 64 at a time, one per lane (their field and call loads are issued together:
        // the one-candidate-per-iteration scalar walk was latency-bound, ~3 dependent loads x ~35 candidates
        // per site); the ordered part below runs over ballots and touches no memory.
        for (uint32_t base = lo; base < hi; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool valid = i < hi;
            const uint32_t ii = valid ? i : lo;                              // (lo < hi here) every lane loads: no dependent round trip
            // (all five loads unconditional and issued together: written as `valid ? load : 0` each became its own exec-masked
            // block with a wait inside, three round trips per chunk instead of two)
            const uint32_t o0 = a.cpg_off[ii], o1 = a.cpg_off[ii + 1];
            const int32_t cs_raw = a.read_start[ii], ce_raw = a.read_end[ii];
            const uint32_t mq = a.read_mapq[ii];
            const uint32_t n = o1 - o0;
            const bool pass = valid & (mq >= (uint32_t)a.min_qual) & (n > 0u);   // fdrp.rs:205, 208
            const int32_t cs = pass ? cs_raw : 0, ce = pass ? ce_raw : 0;
            uint32_t cw[FD_NB];
            bool hit = false;                                                // does the candidate call c ?
            // The first FD_NB calls of a candidate, four per load (a read's calls are consecutive words; the quads past its
            // fourth call are fetched only when some candidate of the chunk has that many).  Registers past the read's last call
            // repeat its first call: a repeated call changes nothing below (same position, same state -- the same bit in the
            // compact masks, the same key in the min-chain).  Lanes without a candidate load valid words that `pass` discards.
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int q = 0; q < FD_NB / 4; ++q) {
                const bool want = q == 0 || __any(pass && n > (uint32_t)(4 * q));      // wave-uniform
                if (want) {
                    // (the address is clamped so that the quad lies inside the array; the few candidates at the batch's end whose
                    // quad would cross it are re-read word by word afterwards -- as an if / else the two arms shared registers and
                    // the compiler waited for the quad right behind its load)
                    const uint32_t oq = o0 + (uint32_t)(4 * q);
                    const uint32_t oq_safe = a.n_cpgs >= 4u ? min(oq, a.n_cpgs - 4u) : 0u;
                    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + oq_safe);
                    cw[4 * q] = v.x; cw[4 * q + 1] = v.y; cw[4 * q + 2] = v.z; cw[4 * q + 3] = v.w;
                    if (__builtin_expect(__any(oq != oq_safe || a.n_cpgs < 4u), 0)) {
                        if (oq != oq_safe || a.n_cpgs < 4u) {
#pragma unroll
                            for (int k = 4 * q; k < 4 * q + 4; ++k) cw[k] = (o0 + (uint32_t)k < a.n_cpgs) ? a.cpg_pos[o0 + k] : 0u;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 4 * q; k < 4 * q + 4; ++k) cw[k] = 0u;
                }
            }
#pragma unroll
            for (int k = 1; k < FD_NB; ++k) cw[k] = ((uint32_t)k < n) ? cw[k] : cw[0];
            {   // "some register holds position c" as an integer min-chain (a chain of compares becomes scalar and-masks per call)
                uint32_t mn = 0xffffffffu;
#pragma unroll
                for (int k = 0; k < FD_NB; ++k) mn = min(mn, (cw[k] ^ (uint32_t)c) & 0x7fffffffu);
                hit = mn == 0u;
            }
            if (pass && n > (uint32_t)FD_NB)
                for (uint32_t k = FD_NB; k < n; ++k) hit = hit || (a.cpg_pos[o0 + k] & 0x7fffffffu) == (uint32_t)c;
            const unsigned long long m_hit = __ballot(pass && hit);
            const unsigned long long m_flush = __ballot(pass && c < (int32_t)(cw[0] & 0x7fffffffu));   // c < first call, fdrp.rs:212
            auto put_row = [&](const int slot) {                              // executed by the candidate's lane
                uint32_t *r = rows + slot * ROW;
                r[0] = o0; r[1] = n; r[2] = (uint32_t)cs; r[3] = (uint32_t)ce;
#pragma unroll
                for (int k = 0; k < FD_NB; ++k) r[4 + k] = cw[k];
            };
            if ((m_hit | m_flush) == 0ull) continue;
            // Common shape of a chunk: the reads calling c, then (optionally) reads that start past c + 1 and
            // so flush the site for good.  With spans <= 200 bp add_read drops nothing, and below max_depth
            // slot = arrival order, so every hit lane stores itself (slot = total + hits in lower lanes).
            const int n_hit = __popcll(m_hit);
            const int first_flush = m_flush ? __builtin_ctzll(m_flush) : 64;
            const bool fast = !win_check && total + n_hit <= (int32_t)a.max_depth &&
                              (m_hit == 0ull || 63 - __builtin_clzll(m_hit) < first_flush) &&
                              (m_flush == 0ull || (int32_t)__builtin_amdgcn_readlane(cs, first_flush & 63) > c + 1);
            if (fast) {
                if (m_hit) {
                    entry = true;                                             // fdrp.rs:226-228, 81-85
                    const int pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_hit, 0u));
                    if (pass && hit && total + pre < FD_SLOTS) put_row(total + pre);
                    total += n_hit; sampled += n_hit;
                    deep = deep || total > FD_SLOTS;                          // only reachable with max_depth > 64
                }
                // fdrp.rs:212-223.  Reads are sorted by start: none from here on can call c, and further flushes find no entry --
                // the flush is the one after the loop (same condition; one inlined copy of finalize less)
                if (m_flush) break;
                continue;
            }
            unsigned long long ev = m_hit | m_flush;                          // a read calling c has first <= c: never both
            while (ev) {                                                      // stream order; wave-uniform
                const int l = __builtin_ctzll(ev);
                ev &= ev - 1;
                if ((m_flush >> l) & 1ull) {                                  // fdrp.rs:212-223
                    if (entry) {
                        if ((uint32_t)sampled >= a.min_depth && !deep) MTH_FD_FINISH();
                        entry = false; total = 0; sampled = 0;
                    }
                    continue;
                }
                entry = true;                                                 // entry().or_insert(...), fdrp.rs:226-228
                const int32_t s = __builtin_amdgcn_readlane(cs, l), e = __builtin_amdgcn_readlane(ce, l);
                if (FD_WIN + (s - c) < 0) continue;                           // add_read, fdrp.rs:58-63
                if (FD_WIN + (e - c) > 2 * FD_WIN) continue;
                int slot;
                if (total < (int32_t)a.max_depth) {                           // fdrp.rs:81-85
                    slot = total; total += 1; sampled += 1;
                } else {                                                       // fdrp.rs:87-94 (reservoir)
                    total += 1;
                    const int32_t jr = sample_site(a.grp_tab, a.grp_n, a.seed, a.tid, c, total);
                    if (jr > (int32_t)a.max_depth) continue;
                    slot = jr - 1;
                }
                if (slot >= FD_SLOTS) { deep = true; continue; }               // only reachable with max_depth > 64
                if (lane == l) put_row(slot);
            }
        }
        if (entry && (uint32_t)sampled >= a.min_depth && !deep) MTH_FD_FINISH();  // fdrp.rs:239-243
        // more reads stored at once than this pass has slots: the next pass redoes the site (flags 2 -> 256 slots, 3 -> rows in HBM);
        // the last pass has max_depth slots, which a site cannot exceed (fdrp.rs:81-85)
        if (deep && (SLOTS == 0 || (SLOTS == FD_SLOTS_DEEP && a.max_depth <= (uint32_t)FD_SLOTS_DEEP)) && lane == 0) atomicOr(&a.st->err, (uint32_t)ERRB_CAPACITY);
        if (lane == 0) {
            a.fdrp[j] = res_f; a.qfdrp[j] = res_q; a.nreads[j] = res_n;
            a.flags[j] = deep ? (SLOTS == 64 ? 2u : 3u) : (have ? 1u : 0u);
        }
    }
}

#undef MTH_FD_FINISH

// ---------------------------------------------------------------------------------------------
// WGBS depth (config 3: ~10x, a dozen candidate reads per site): the wave-per-site walk above fills a fifth of its lanes
// and pays its whole chain of dependent round trips -- site, index entries, candidate fields, calls -- per site.  Here a wave
// takes 64 / GL consecutive sites (GL = 16 or 32 lanes each): lane = candidate read of its site, the hit lanes turn their
// calls into the three 64-bit masks of the compact finalize themselves and store them at slot = arrival order, then lane k
// of a site takes the site's k-th, (k + GL)-th ... pair (pair index from an LDS copy of the pair table: no load in the
// round loop).  The qFDRP terms of a round are chained in the reference's (i, j) order inside each site's lanes (DPP
// wave_shr:1: x[l] = x[l-1] + term[l], a skipped pair adds +0.0; the lane read after s steps depends on the s lanes below
// it only, all of its own site).
// Only the common shape is handled: one chunk (<= GL candidates), calls in registers (<= 8 per candidate), the "hits, then
// reads past c + 1" order, at most max_depth hits, spans <= 200 bp (host side), the 64 sites around c covering +-200 bp.
// Anything else is handed back (flags = 4) and done by k_fdrp_walk<8, 64> with only_flag = 4 -- same results either way.
constexpr uint32_t FD_REDO = 4u;
// WIN: sites of the window around c that the call masks index, 64 or 32.  At WGBS density the +-200 bp a stored read can reach hold ~4
// sites: 32 are plenty (a denser stretch fails `compact` and is handed back, as with 64), and the masks, the rows and the popcounts of a
// pair round are single words (round 5).
template <int GL, int WIN>      // GL: lanes (= candidate reads, stored reads, pairs per round) of a site: 16 or 32
__global__ __launch_bounds__(256, 8) void k_fdrp_walk4(const FdrpArgs a) {
    static_assert(WIN == 64 || WIN == 32, "mask words");
    typedef typename std::conditional<WIN == 64, unsigned long long, uint32_t>::type mask_t;
    constexpr int NG = 64 / GL, WS = WIN / GL;      // sites per wave; sites of the window per lane
    constexpr uint32_t GMASK = GL == 32 ? 0xffffffffu : (1u << GL) - 1u;
    constexpr int NTAB = (GL + 1) * GL * (GL - 1) / 6;   // pairs of n = 2..GL stored reads, one list after the other
    const int lane = threadIdx.x & 63, gl = lane & (GL - 1), row0 = lane & ~(GL - 1), g = lane / GL, wave = threadIdx.x >> 6;
    const uint32_t wave_id = sgpr((uint32_t)((blockIdx.x * 256 + threadIdx.x) >> 6)), n_waves = (gridDim.x * 256) >> 6;
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    __shared__ uint8_t s_bit[4][NG][2 * FD_WIN + 1 + 13];
    __shared__ __attribute__((aligned(16))) uint32_t s_rows[4][NG][GL * 8];   // per slot: start, end, mC, mA, mM
    __shared__ uint16_t s_tab[NTAB + 1];
    for (int t = threadIdx.x; t < NTAB; t += 256) s_tab[t] = a.pair_tab[t];
    if (threadIdx.x == 0) s_tab[NTAB] = 0;
    // ham / ncpg of a pair round from a table: a candidate holds at most 8 calls here (more: handed back), so ncpg <= 8; the entries
    // are made by the same f32 division the round would do (12 instructions per round otherwise, a fifth of its vector work)
    constexpr int W4_QN = 9;
    __shared__ float s_quot4[W4_QN * W4_QN];
    for (int t = threadIdx.x; t < W4_QN * W4_QN; t += 256) s_quot4[t] = (float)(t / W4_QN) / (float)(t % W4_QN);
    __syncthreads();
    uint8_t *const bit_of = s_bit[wave][g];
    uint32_t *const rows = s_rows[wave][g];
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    __shared__ uint32_t s_list[4][64];
    uint32_t *const list = s_list[wave];
    // A wave takes 64 consecutive sites at a time, lane = site: the sites whose count is below min_depth cannot produce a row
    // (fdrp.rs:239-243; more than half of them at config-3 depth) and get their empty result here, the others are packed into
    // a list and walked NG at a time -- every site of a step is one that needs the walk.
    for (uint32_t blk = wave_id; (uint64_t)blk * 64u < n_sites; blk += n_waves) {
        uint32_t n_act;
        {
            const uint32_t jl = blk * 64u + (uint32_t)lane;
            const bool in = jl < n_sites;
            const uint32_t cov_l = in ? a.site_nc[jl] + a.site_nd[jl] : 0u;
            const bool active = in && cov_l >= a.min_depth;
            if (in && !active) { a.fdrp[jl] = 0.0f; a.qfdrp[jl] = 0.0f; a.nreads[jl] = 0u; a.flags[jl] = 0u; }
            // The list in order of the sites' read counts (16 classes, a ballot each): the NG sites of a step run max-over-sites pair
            // rounds, ceil(n (n - 1) / 2 / GL) each -- 2 at 8 stored reads, 6 at 14 -- so sites of a kind go together (round 5).
            const uint32_t cls = active ? min(cov_l, 15u) : 16u;
            uint32_t at = 0, base_l = 0;
#pragma unroll
            for (uint32_t b = 0; b < 16u; ++b) {
                const unsigned long long m = __ballot(cls == b);
                if (cls == b) at = base_l + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                base_l += (uint32_t)__popcll(m);
            }
            n_act = base_l;
            __builtin_amdgcn_wave_barrier();                                     // the previous block's list reads are done
            if (active) list[at] = jl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    unsigned long long redo_bits = 0;
    for (uint32_t t0 = 0; t0 < n_act; t0 += (uint32_t)NG) {
        const bool jv = t0 + (uint32_t)g < n_act;
        const uint32_t j = list[jv ? t0 + (uint32_t)g : t0];
        const uint32_t jj = j;
        const int32_t c = a.site_pos[jj];
        const bool act = jv;                                                // site-uniform (as everything named per site below)
        bool redo = false;
        // the WIN sites around c, WS per lane
        const uint32_t j0w = (jj >= (uint32_t)(WIN / 2)) ? min(jj - (uint32_t)(WIN / 2), n_sites > (uint32_t)WIN ? n_sites - (uint32_t)WIN : 0u) : 0u;
        int32_t sp[WS];
#pragma unroll
        for (int t = 0; t < WS; ++t) { const uint32_t si = j0w + (uint32_t)(WS * gl + t); sp[t] = si < n_sites ? a.site_pos[si] : 0x7fffffff; }
        const uint32_t lo = min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        redo = redo || hi - lo > 2u * (uint32_t)GL || a.n_cpgs < 8u;
        {
            const int32_t sp_first = __shfl(sp[0], row0, 64), sp_last = __shfl(sp[WS - 1], row0 | (GL - 1), 64);
            const bool compact = (int64_t)c - 200 >= a.region_beg && (int64_t)c + 200 < a.region_end &&
                                 (j0w == 0u || sp_first < c - 200) && (int64_t)sp_last > (int64_t)c + 200;   // absent sites read as +inf
            redo = redo || !compact;
        }
#pragma unroll
        for (int t = 0; t < WS; ++t) {
            const uint32_t rel = (uint32_t)(sp[t] - (c - FD_WIN));
            if (rel <= 2u * FD_WIN) bit_of[rel] = (uint8_t)(WS * gl + t);
        }
        // Candidates.  The index hands out whole 32-bp quanta: up to 2 GL reads, two per lane, of which those starting before
        // c - max_span + 1 (they end before c: neither a call at c nor a first call past it) and those starting past c + 1 (a read
        // calls [start - 1, end]: no call at c; they could only flush after every hit, which changes nothing here) are inert.
        // Reads are sorted by start, so the inert ones are a prefix and a suffix; the rest -- GL at most, else the site is handed
        // back -- moves to lanes 0.. of the site by two lane permutes per field.
        const uint32_t n_idx = hi - lo;
        const bool vA = act && !redo && (uint32_t)gl < n_idx, vB = act && !redo && (uint32_t)(GL + gl) < n_idx;
        const uint32_t iA = vA ? lo + (uint32_t)gl : 0u, iB = vB ? lo + (uint32_t)(GL + gl) : 0u;
        const uint32_t oA0 = a.cpg_off[iA], oA1 = a.cpg_off[iA + 1], oB0 = a.cpg_off[iB], oB1 = a.cpg_off[iB + 1];
        const int32_t sA = a.read_start[iA], sB = a.read_start[iB], eA = a.read_end[iA], eB = a.read_end[iB];
        const uint32_t mqA = a.read_mapq[iA], mqB = a.read_mapq[iB];
        const int32_t s_min = c - a.max_span + 1;
        auto site_count = [&](const bool p) { return (uint32_t)__builtin_popcount((uint32_t)(__ballot(p) >> row0) & GMASK); };
        const uint32_t below = site_count(vA && sA < s_min) + site_count(vB && sB < s_min);
        const uint32_t above = site_count(vA && sA > c + 1) + site_count(vB && sB > c + 1);
        const uint32_t n_c = (act && !redo) ? n_idx - below - above : 0u;
        redo = redo || n_c > (uint32_t)GL;
        const uint32_t kk = below + (uint32_t)gl;
        const bool fromB = kk >= (uint32_t)GL;
        const int src = row0 | (int)(kk & (uint32_t)(GL - 1));
        auto pick = [&](const uint32_t xa, const uint32_t xb) { const uint32_t ya = __shfl(xa, src, 64), yb = __shfl(xb, src, 64); return fromB ? yb : ya; };
        const bool valid = act && !redo && (uint32_t)gl < n_c;
        const uint32_t o0 = pick(oA0, oB0), o1 = pick(oA1, oB1);
        const int32_t cs_raw = (int32_t)pick((uint32_t)sA, (uint32_t)sB), ce_raw = (int32_t)pick((uint32_t)eA, (uint32_t)eB);
        const uint32_t mq = pick(mqA, mqB);
        const uint32_t n = o1 - o0;
        const bool pass = valid & (mq >= (uint32_t)a.min_qual) & (n > 0u);  // fdrp.rs:205, 208
        const int32_t cs = pass ? cs_raw : 0, ce = pass ? ce_raw : 0;
        uint32_t cw[8];
        {
            // (a candidate whose 8-word window would cross the end of the call array, or with more than 8 calls: handed back)
            const bool edge = pass && ((uint64_t)o0 + 8u > (uint64_t)a.n_cpgs || n > 8u);
            const uint32_t ob = a.n_cpgs >= 8u ? min(o0, a.n_cpgs - 8u) : 0u;
            u32x4_a4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
            if (a.n_cpgs >= 8u) {                                               // wave-uniform
                v0 = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + ob);
                if (__any(pass && n > 4u)) v1 = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + ob + 4u);
            }
            cw[0] = v0.x; cw[1] = v0.y; cw[2] = v0.z; cw[3] = v0.w; cw[4] = v1.x; cw[5] = v1.y; cw[6] = v1.z; cw[7] = v1.w;
            const unsigned long long m_edge = __ballot(edge);
            redo = redo || ((uint32_t)(m_edge >> row0) & GMASK) != 0u;
        }
#pragma unroll
        for (int k = 1; k < 8; ++k) cw[k] = ((uint32_t)k < n) ? cw[k] : cw[0];
        bool hit;
        {
            uint32_t mn = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < 8; ++k) mn = min(mn, (cw[k] ^ (uint32_t)c) & 0x7fffffffu);
            hit = mn == 0u;
        }
        const uint32_t mh = (uint32_t)(__ballot(pass && hit) >> row0) & GMASK;
        const uint32_t mf = (uint32_t)(__ballot(pass && c < (int32_t)(cw[0] & 0x7fffffffu)) >> row0) & GMASK;   // c < first call, fdrp.rs:212
        const int n_hit = __builtin_popcount(mh);
        const int first_flush = mf ? __builtin_ctz(mf) : GL;
        // (the wave-per-site walk also wants the first flusher to start past c + 1, because it stops looking at that read; here the
        // site's lanes hold EVERY read that starts at or before c + 1, so "no hit after the first flusher" is checked, not inferred:
        // the segment the flusher closes is the one evaluated, nothing re-opens it -- fdrp.rs:212-223)
        const bool fast = (uint32_t)n_hit <= a.max_depth && (mh == 0u || 31 - __builtin_clz(mh) < first_flush);
        redo = redo || !fast;
        // fdrp.rs:239-243: the open segment is evaluated when it holds >= min_depth reads (and exists at all)
        const bool fin = act && !redo && mh != 0u && (uint32_t)n_hit >= a.min_depth;
        const int nS = fin ? n_hit : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (fin && pass && hit) {
            mask_t mC = 0, mM = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t rel_p = (cw[k] & 0x7fffffffu) - (uint32_t)(c - FD_WIN);
                const mask_t b = (mask_t)1 << bit_of[rel_p];
                mC |= b;
                mM |= b & (mask_t)((long long)(int32_t)cw[k] >> 31);
            }
            const uint32_t p0 = cw[0] & 0x7fffffffu;
            const mask_t b0 = (mask_t)1 << bit_of[p0 - (uint32_t)(c - FD_WIN)];
            const mask_t mA = ((int32_t)p0 >= cs) ? mC : mC & ~b0;   // the one call that can lie outside the covered bases: start - 1
            mM &= mA;
            uint32_t *r = rows + 8 * __builtin_popcount(mh & ((1u << gl) - 1u));   // slot = arrival order
            r[0] = (uint32_t)cs; r[1] = (uint32_t)ce;
            if (WIN == 64) {
                r[2] = (uint32_t)mC; r[3] = (uint32_t)((unsigned long long)mC >> 32); r[4] = (uint32_t)mA; r[5] = (uint32_t)((unsigned long long)mA >> 32);
                r[6] = (uint32_t)mM; r[7] = (uint32_t)((unsigned long long)mM >> 32);
            } else { r[2] = (uint32_t)mC; r[3] = (uint32_t)mA; r[4] = (uint32_t)mM; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // pairs: lane k of the site takes pairs k, k + GL, ... of the reference's (i, j) order (fdrp.rs:129-141)
        const int P = (nS * (nS - 1)) >> 1;
        const int rg = (P + GL - 1) / GL;
        int rounds = 0;
#pragma unroll
        for (int w = 0; w < NG; ++w) rounds = max(rounds, __builtin_amdgcn_readlane(rg, w * GL));
        const uint16_t *const tab = s_tab + (uint32_t)(nS > 2 ? nS * (nS - 1) * (nS - 2) : 0) / 6u;
        uint32_t disc = 0;
        float q = 0.0f;
        unsigned long long inexact = 0;                                          // lanes that have held a non-dyadic term this step (wave-uniform value)
        for (int r = 0; r < rounds; ++r) {
            const int k = GL * r + gl;
            const uint32_t ent = tab[P ? min(k, P - 1) : 0];                     // (requested one round ahead: measured, no change)
            const int pi = (int)(ent & 0xffu) & (GL - 1), pj = (int)(ent >> 8) & (GL - 1);
            const uint32_t *ri = rows + pi * 8, *rj = rows + pj * 8;
            const int32_t si = (int32_t)ri[0], ei = (int32_t)ri[1], sj = (int32_t)rj[0], ej = (int32_t)rj[1];
            const int32_t ov = min(ei, ej) - max(si, sj) + 1;                  // get_num_overlap_bases, fdrp.rs:97-107
            const bool pair_ok = (k < P) & (max(ov, 0) >= a.min_overlap);        // fdrp.rs:134
            uint32_t ncpg, ham;
            if (WIN == 64) {
                ncpg = __builtin_popcount(ri[2] & rj[2]) + __builtin_popcount(ri[3] & rj[3]);   // qfdrp.rs:109-119
                ham = __builtin_popcount(ri[4] & rj[4] & (ri[6] ^ rj[6])) +
                      __builtin_popcount(ri[5] & rj[5] & (ri[7] ^ rj[7]));                      // fdrp.rs:114-115
            } else {
                ncpg = __builtin_popcount(ri[2] & rj[2]);
                ham = __builtin_popcount(ri[3] & rj[3] & (ri[4] ^ rj[4]));
            }
            disc += (pair_ok && ham != 0u) ? 1u : 0u;                            // fdrp.rs:138-140
            const float term = pair_ok ? s_quot4[min(ham, 8u) * W4_QN + min(ncpg, 8u)] : 0.0f;   // qfdrp.rs:152 ((float)ham / (float)ncpg); +0.0 for skipped pairs
            const unsigned long long nz = __ballot(term != 0.0f);
            if (nz == 0ull) continue;                                            // wave-uniform: x + 0.0 == x
            // The ordered f32 sum (qfdrp.rs:152) needs its order only once a term is not a dyadic fraction.  ham / ncpg with ncpg a
            // power of two is exact, <= 1 and a multiple of 1/64; at most 496 of them per site add up exactly in 24 bits whatever
            // the order -- so while every term of every site of the wave has been such a one (most pairs at WGBS depth share one
            // or two calls), the round is a tree sum over the site's lanes; from the first other term on, the chain.
            inexact |= __ballot(term != 0.0f && (ncpg & (ncpg - 1u)) != 0u);
            if (inexact == 0ull) {
                float t = term;
                if (GL == 32) t += __shfl_xor(t, 16, 64);
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x128 /*row_ror:8*/, 0xf, 0xf, true));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x124 /*row_ror:4*/, 0xf, 0xf, true));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4e /*quad_perm [2,3,0,1]*/, 0xf, 0xf, true));
                t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xb1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, true));
                q += t;
                continue;
            }
            // chain length: the highest lane of any site that holds a non-zero term (later lanes add +0.0: exact)
            int steps = 0;
#pragma unroll
            for (int w = 0; w < NG; ++w) { const uint32_t m = (uint32_t)(nz >> (GL * w)) & GMASK; steps = max(steps, m ? 31 - __builtin_clz(m) : 0); }
            float x = (gl == 0) ? q + term : term;
            for (int st = 0; st < steps; ++st)
                x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)) + term;
            q = __shfl(x, row0 | steps, 64);
        }
        uint32_t n_disc = disc;
#pragma unroll
        for (int o = GL / 2; o > 0; o >>= 1) n_disc += __shfl_xor(n_disc, o, 64);  // within the site's lanes
        if (gl == 0 && jv) {
            float res_f = 0.0f, res_q = 0.0f;
            uint32_t res_n = 0u, fl = 0u;
            if (fin) {
                // (num_reads * (num_reads - 1)) as f32 / 2.0 in usize arithmetic (fdrp.rs:143)
                const unsigned long long prod = (unsigned long long)(long long)nS * (unsigned long long)((long long)nS - 1);
                const float den = (float)prod / 2.0f;
                res_f = (float)n_disc / den; res_q = q / den; res_n = (uint32_t)nS; fl = 1u;
            } else if (act && redo) fl = FD_REDO;
            a.fdrp[j] = res_f; a.qfdrp[j] = res_q; a.nreads[j] = res_n; a.flags[j] = fl;
        }
        {   // the step's handed-back sites, as bits of the block's mask (wave-uniform)
            const uint32_t rel = (act && redo) ? (j - blk * 64u) + 1u : 0u;
#pragma unroll
            for (int w = 0; w < NG; ++w) { const uint32_t r1 = __builtin_amdgcn_readlane(rel, w * GL); if (r1) redo_bits |= 1ull << (r1 - 1u); }
        }
        __builtin_amdgcn_wave_barrier();                                         // LDS reads done before the next sites' writes
    }
        if (lane == 0) a.redo_mask[blk] = redo_bits;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the read x read form.  What the wave-per-site walk repeats at every site -- inspecting the ~60-120
// candidate reads, turning the stored ones into three 64-bit masks -- depends on the READ, not on the site: in the reference a
// stored read is never cut by the +-201 window, it is kept whole or dropped (fdrp.rs:51-63), so overlap bases, shared calls and
// Hamming count of a read pair are the same at every site the two share.  And the ordered f32 sum (qfdrp.rs:152) is a chain
// of dependent adds that uses one lane of 64 when a wave owns one site.  On config 4 (50x hotspots, -D 64) the old kernel's 6.55 ms
// were walk + set-up 2.6, pair rounds 1.4, chain 2.5 (ablation, profiles/r04_fdrp_tile.md).
//   k_fdrp_tile   one workgroup per TILE of FT_CORE consecutive sites.  Phase 1, once per tile: every candidate read (one per
//                 thread) is loaded, filtered (mapq, >= 1 call: fdrp.rs:205-210) and turned into {start, end, first call, calls,
//                 covered calls, methylated covered calls} over the tile's window of <= 64 sites (the core plus the sites within
//                 max_span on either side: all a core site's readers can call), kept in LDS.  Phase 2, per core site (a wave
//                 each, in turn): the stored reads are the passing reads whose call mask has the site's bit, in file order --
//                 ballots over the LDS table, with the flush rule (a passing read whose first call lies past the site, strict:
//                 fdrp.rs:212) checked on the same ballots; their rows go to the wave's slot array and the pair rounds run as in
//                 the walk's compact finalize (lane k = k-th pair of the (i, j) order), except that a round does not chain its
//                 terms: it appends the non-zero ones, as one byte each (ncpg (ncpg + 1) / 2 + ham, ncpg <= 21), to the site's
//                 list in HBM -- claimed per tile with one atomic from a buffer whose size bounds them exactly.
//   k_fdrp_chain  one THREAD per site: the chain over the site's list (byte -> quotient from a 253-entry LDS table filled by the
//                 same f32 divisions), 64 sites per wave instead of one; then fdrp.rs:143 / qfdrp.rs:155.
// Only the common shape: spans <= 200 bp (host side), a window of <= 64 sites within FT_SPAN bp, <= FT_RMAX candidates, every
// call of a stored read on a window site, one segment per site (no hit after a flusher), <= min(max_depth, 64) stored reads,
// <= 21 shared calls per pair.  Anything else is handed back through redo_mask to k_fdrp_walk (only_flag), bit-identical either way.
constexpr int FT_CORE = 32, FT_RMAX = 384, FT_SPAN = 2048, FT_LIST = 128;
constexpr uint32_t FD_CHAIN = 8u;                 // flags: the site's terms are listed, k_fdrp_chain finishes it
constexpr uint32_t FT_NCPG_MAX = 21u;             // 21 * 22 / 2 + 21 = 252: the codes fit a byte

template <int FD_NB>
__global__ __launch_bounds__(256, 7) void k_fdrp_tile(const FdrpArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    // the tile's table, one row per candidate read: start; span | first call - (start - 1) | window bit of an uncovered first call;
    // calls; methylated calls (both over the tile's window of <= 64 sites)
    __shared__ uint8_t s_bit[FT_SPAN];
    __shared__ int32_t s_start[FT_RMAX];
    __shared__ uint32_t s_pack[FT_RMAX];
    __shared__ unsigned long long s_mC[FT_RMAX], s_mM[FT_RMAX];
    __shared__ __attribute__((aligned(16))) uint32_t s_rows[4][64 * 8];     // per wave: the stored reads of the site in hand
    __shared__ uint16_t s_list[4][FT_LIST];                                    // per wave: the site's readers in arrival order
    __shared__ uint8_t s_draw[4][FT_LIST];                                     // per wave: reservoir draws of the arrivals past max_depth
    __shared__ uint32_t s_cnt[FT_CORE];
    __shared__ unsigned long long s_off[FT_CORE];
    __shared__ uint32_t s_hdr[8];                 // 0: window first site, 1: window sites, 2: fallback, 3: lo, 4: hi
    __shared__ uint32_t s_redo[4];
    uint32_t *const rows = s_rows[wave];
    uint16_t *const list = s_list[wave];
    uint8_t *const draw = s_draw[wave];
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t n_tiles = (n_sites + FT_CORE - 1) / FT_CORE;
    const uint32_t dcap = min(a.max_depth, 64u);  // stored reads a site of this kernel can hold (fdrp.rs:81-85)
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      const uint32_t t_beg = tile * FT_CORE, t_end = min(t_beg + (uint32_t)FT_CORE, n_sites);
      // a tile is taken as ONE core of 32 sites -- or, where the window of such a core does not fit (a CpG-dense stretch), as cores of
      // 16 / 8 sites: handing a dense stretch to the wave-per-site walk costs up to a millisecond PER SITE there (call-by-call path)
      uint32_t cs = FT_CORE;
      for (uint32_t ja = t_beg, jb = 0; ja < t_end; ja = jb) {
        jb = min(ja + cs, t_end);                 // core sites [ja, jb)
        const uint32_t ncore = jb - ja;
        __syncthreads();                          // the previous core's LDS is no longer read
        // ---- the window: the core and the sites within max_span of it (sorted positions: a count on either side) ----
        if (wave == 0) {
            const int32_t p_first = a.site_pos[ja], p_last = a.site_pos[jb - 1];
            const bool lower = lane < 32;
            const int64_t js = lower ? (int64_t)ja - 1 - lane : (int64_t)jb + (lane - 32);
            const bool in = js >= 0 && js < (int64_t)n_sites;
            const int32_t p = in ? a.site_pos[in ? js : 0] : 0;
            const bool q = in && (lower ? (int64_t)p >= (int64_t)p_first - a.max_span : (int64_t)p <= (int64_t)p_last + a.max_span);
            const unsigned long long m = __ballot(q);
            const uint32_t L = (uint32_t)__builtin_popcount((uint32_t)m), U = (uint32_t)__builtin_popcount((uint32_t)(m >> 32));
            const uint32_t W = L + ncore + U;
            const int32_t w_first = a.site_pos[ja - L], w_last = a.site_pos[jb - 1 + U];
            // (32 qualifying sites on a side: there may be more -- not a window this kernel holds)
            bool fb = L >= 32u || U >= 32u;
            bool too_wide = W > 64u || (int64_t)w_last - w_first >= FT_SPAN;
            // (halo reads of a region slice call positions outside the region, which are not in the site list: phase 1 notices)
            const uint32_t lo = min(a.idx[(uint32_t)(p_first - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
            const uint32_t hi = min(a.idx[((uint32_t)(p_last + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
            too_wide = too_wide || hi - lo > (uint32_t)FT_RMAX;
            fb = fb || a.n_cpgs < 4u || (too_wide && ncore <= 8u);
            if (lane == 0) { s_hdr[0] = ja - L; s_hdr[1] = W; s_hdr[2] = fb ? 1u : (too_wide ? 2u : 0u); s_hdr[3] = lo; s_hdr[4] = hi; }
        }
        if (tid < FT_CORE) s_cnt[tid] = 0u;
        __syncthreads();
        // (workgroup-uniform values are made scalar explicitly: loaded from LDS they count as divergent, and every loop they bound
        // would be compiled with exec-mask bookkeeping per trip)
        const uint32_t jw = sgpr(s_hdr[0]), W = sgpr(s_hdr[1]), lo = sgpr(s_hdr[3]), hi = sgpr(s_hdr[4]);
        const uint32_t R = hi - lo;
        if (sgpr(s_hdr[2]) == 2u) { cs = cs > 16u ? 16u : 8u; jb = ja; continue; }            // the same sites again, as smaller cores
        bool fallback = sgpr(s_hdr[2]) != 0u;     // workgroup-uniform
        const uint32_t cbit = ja - jw;            // window bit of the first core site
        const int32_t w_base = sgpr(a.site_pos[jw]);
        if (!fallback) {
            const int32_t w_span = sgpr(a.site_pos[jw + W - 1]) - w_base + 1;
            for (int p = tid; p < ((w_span + 3) & ~3); p += 256) s_bit[p] = 0xffu;
            __syncthreads();
            if (tid < (int)W) s_bit[a.site_pos[jw + tid] - w_base] = (uint8_t)tid;
            __syncthreads();
            // ---- phase 1: one candidate read per thread -> its row of the tile's table ----
            bool bad = false;                     // a read that stores itself at a core site has a call without a window bit
            for (uint32_t r = tid; r < ((R + 63u) & ~63u); r += 256) {
                const bool valid = r < R;
                const uint32_t i = lo + (valid ? r : 0u);
                const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
                const int32_t rs = a.read_start[i], re = a.read_end[i];
                const uint32_t mq = a.read_mapq[i];
                const uint32_t n = o1 - o0;
                const bool pass = valid & (mq >= (uint32_t)a.min_qual) & (n > 0u);            // fdrp.rs:205, 208
                unsigned long long mC = 0, mM = 0;
                uint32_t miss = 0, w0 = 0;
                auto bit_at = [&](const uint32_t pos) {
                    const uint32_t rel = pos - (uint32_t)w_base;
                    return rel < (uint32_t)w_span ? (uint32_t)s_bit[rel] : 0xffu;
                };
                auto add_call = [&](const uint32_t w, const bool live) {
                    const uint32_t b = bit_at(w & 0x7fffffffu);
                    miss |= live ? b >> 7 : 0u;                                                // 0xff: not a site of the window
                    const unsigned long long bb = live ? 1ull << (b & 63u) : 0ull;
                    mC |= bb;
                    mM |= bb & (unsigned long long)((long long)(int32_t)w >> 31);
                };
                // the calls four per load, a quad at a time (the address is clamped so that the quad lies inside the array; the few
                // reads at the batch's end whose quad would cross it are re-read word by word)
#pragma unroll
                for (int q = 0; q < FD_NB / 4; ++q) {
                    if (q != 0 && !__any(pass && n > (uint32_t)(4 * q))) break;                 // wave-uniform
                    const uint32_t oq = o0 + (uint32_t)(4 * q);
                    const uint32_t oq_safe = min(oq, a.n_cpgs - 4u);
                    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + oq_safe);
                    uint32_t cw[4] = {v.x, v.y, v.z, v.w};
                    if (__builtin_expect(__any(oq != oq_safe), 0)) {
                        if (oq != oq_safe) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) cw[k] = (oq + (uint32_t)k < a.n_cpgs) ? a.cpg_pos[oq + k] : 0u;
                        }
                    }
                    if (q == 0) w0 = cw[0];
#pragma unroll
                    for (int k = 0; k < 4; ++k) add_call(cw[k], pass && (uint32_t)(4 * q + k) < n);
                }
                if (pass) for (uint32_t t = FD_NB; t < n; ++t) add_call(a.cpg_pos[o0 + t], true);
                const uint32_t p0 = w0 & 0x7fffffffu;
                // the one call that can lie outside the covered bases is the first, at start - 1 (a reverse read's)
                const uint32_t abit = (pass && (int32_t)p0 < rs) ? bit_at(p0) & 63u : 0xffu;
                if (abit != 0xffu) mM &= ~(1ull << abit);
                // does the read store itself at a core site?  (those are the reads whose masks must be complete)
                const unsigned long long core = miss ? 0ull : (mC >> cbit) & (ncore == 64u ? ~0ull : (1ull << ncore) - 1ull);
                if (pass && miss) {
                    // it calls a position outside the window: harmless if none of its calls is a core site -- decided on the calls
                    bool hits_core = false;
                    for (uint32_t t = 0; t < n; ++t) { const uint32_t b = bit_at(a.cpg_pos[o0 + t] & 0x7fffffffu); hits_core |= b != 0xffu && b >= cbit && b < cbit + ncore; }
                    bad |= hits_core;
                    mC = 0; mM = 0;
                }
                if (valid) {
                    s_start[r] = rs;
                    // first call - (start - 1) in 0 .. span; 0xff: not a flusher (the read does not pass)
                    s_pack[r] = (uint32_t)(re - rs) | ((pass ? (uint32_t)((int32_t)p0 - (rs - 1)) : 0xffu) << 8) | (abit << 16);
                    s_mC[r] = mC; s_mM[r] = mM;
                    unsigned long long cm = core;
                    while (cm) { const int b = __builtin_ctzll(cm); cm &= cm - 1ull; atomicAdd(&s_cnt[b], 1u); }
                }
            }
            if (__syncthreads_or(bad ? 1 : 0)) fallback = true;
        }
        // ---- list space: C(n, 2) bytes (a multiple of 64) per core site that will be evaluated, one claim per tile ----
        if (!fallback) {
            if (wave == 0) {
                const uint32_t n = lane < (int)ncore ? s_cnt[lane] : 0u;
                const uint32_t ns = min(n, dcap);
                const bool ev = lane < (int)ncore && ns >= max(a.min_depth, 1u) && n <= (uint32_t)FT_LIST;
                const uint32_t cap = ev ? ((ns * (ns - 1u) / 2u + 63u) & ~63u) : 0u;      // a round writes 64 bytes
                uint32_t incl = cap;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
                const uint32_t total = __shfl(incl, 63, 64);
                unsigned long long base = 0;
                if (lane == 0 && total) base = atomicAdd(a.cursor, (unsigned long long)total);
                base = __shfl(base, 0, 64);
                if (lane < (int)ncore) s_off[lane] = base + incl - cap;
                if (lane == 0) s_hdr[2] = (base + total > a.budget) ? 1u : 0u;
            }
            __syncthreads();
            if (sgpr(s_hdr[2])) fallback = true;
        }
        if (fallback) {
            if (tid < (int)ncore) a.flags[ja + tid] = FD_REDO;
            if (wave == 0) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.redo_cnt, ncore);
                base = __shfl(base, 0, 64);
                if (lane < (int)ncore) a.redo_list[base + lane] = ja + lane;
            }
            continue;
        }
        // ---- phase 2: a wave per core site ----
        uint32_t redo_bits = 0;
        // a site's readers start in [c - max_span + 1, c + 1]: a sub-range of the tile's candidates, from the same index; the
        // next site's two entries are requested a site ahead
        auto range_of = [&](const uint32_t kk, uint32_t &r_lo, uint32_t &r_hi) {
            const int32_t cc = a.site_pos[ja + min(kk, ncore - 1u)];
            r_lo = a.idx[(uint32_t)(cc - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT];
            r_hi = a.idx[((uint32_t)(cc + 1 - a.idx_base) >> IDX_QSHIFT) + 1];
        };
        uint32_t nx_lo, nx_hi;
        range_of((uint32_t)wave, nx_lo, nx_hi);
        for (uint32_t k = (uint32_t)wave; k < ncore; k += 4u) {
            const uint32_t j = ja + k;
            const uint32_t c_lo = max(sgpr(min(nx_lo, a.n_reads)), lo) - lo, c_hi = min(sgpr(min(nx_hi, a.n_reads)), hi) - lo;   // rows of the tile's table
            range_of(k + 4u, nx_lo, nx_hi);
            const int32_t c = sgpr(a.site_pos[j]);
            const uint32_t b = cbit + k;
            const uint32_t n_all = sgpr(s_cnt[k]);
            __builtin_amdgcn_wave_barrier();      // the previous site's LDS reads are done
            // fdrp.rs:239-243: a row needs get_num_reads() = the STORED reads (at most max_depth of the arrivals) >= min_depth
            if (min(n_all, a.max_depth) < max(a.min_depth, 1u)) {
                if (lane == 0) { a.fdrp[j] = 0.0f; a.qfdrp[j] = 0.0f; a.nreads[j] = 0u; a.flags[j] = 0u; }
                continue;
            }
            bool redo = n_all > (uint32_t)FT_LIST;
            // The site's readers in file order (its arrivals, fdrp.rs:226-231) and the flush rule on the same ballots: a passing read
            // whose first call lies past the site closes the open segment (fdrp.rs:212-223), a reader behind it opens a new one; the
            // LAST segment whose stored reads reach min_depth is the site's result (BTreeMap::insert overwrites, fdrp.rs:216 / 241).
            uint32_t n_arr = 0, seg0 = 0, best0 = 0, best1 = 0;
            bool open = false;
            auto close_seg = [&]() {
                if (min(n_arr - seg0, a.max_depth) >= max(a.min_depth, 1u)) { best0 = seg0; best1 = n_arr; }
                open = false;
            };
            for (uint32_t r0 = c_lo; r0 < c_hi && !redo; r0 += 64u) {
                const uint32_t r = r0 + (uint32_t)lane;
                const bool v = r < c_hi;
                const unsigned long long mc = v ? s_mC[r] : 0ull;
                const uint32_t pk = v ? s_pack[r] : 0xff00u;
                const bool hit = (mc >> b) & 1ull;
                const uint32_t fo = (pk >> 8) & 0xffu;
                const bool fl = fo != 0xffu && s_start[v ? r : 0u] - 1 + (int32_t)fo > c;       // c < first call, fdrp.rs:212
                const unsigned long long mh = __ballot(hit), mf = __ballot(fl);
                if (!mh) { if (open && mf) close_seg(); continue; }
                // the flushers that can matter: all of them while a segment is open, else those above the chunk's first reader
                const unsigned long long f2 = mf & (open ? ~0ull : ~((1ull << __builtin_ctzll(mh)) - 1ull));
                const int ff = f2 ? __builtin_ctzll(f2) : 64;
                if (ff == 64 || ff == 63 || (mh >> (ff + 1)) == 0ull) {
                    // the chunk's readers join ONE segment; a flusher behind the last of them closes it
                    if (!open) { seg0 = n_arr; open = true; }
                    if (hit) list[n_arr + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u))] = (uint16_t)r;
                    n_arr += (uint32_t)__popcll(mh);
                    if (f2) close_seg();
                } else {
                    // readers on both sides of a flusher: the chunk's events one by one, in file order (wave-uniform loop)
                    unsigned long long ev = mh | mf;
                    while (ev) {
                        const int l = __builtin_ctzll(ev);
                        ev &= ev - 1ull;
                        if ((mf >> l) & 1ull) { if (open) close_seg(); continue; }
                        if (!open) { seg0 = n_arr; open = true; }
                        if (lane == l) list[n_arr] = (uint16_t)r;
                        n_arr += 1u;
                    }
                }
            }
            if (open) close_seg();                                                              // the final flush, fdrp.rs:239-243
            const uint16_t *const lst = list + best0;
            const uint32_t n_sel = best1 - best0;                                               // arrivals of the chosen segment (0: no row)
            if (!redo && n_sel == 0u) {
                if (lane == 0) { a.fdrp[j] = 0.0f; a.qfdrp[j] = 0.0f; a.nreads[j] = 0u; a.flags[j] = 0u; }
                continue;
            }
            // fdrp.rs:81-94: the first max_depth arrivals fill the slots; arrival t beyond (total = t + 1) replaces slot j - 1 when
            // its draw j is <= max_depth.  Slot s ends up with the LAST such arrival, or with arrival s.
            const uint32_t nS = min(n_sel, dcap);
            if (!redo && a.max_depth > 64u && n_sel > 64u) redo = true;                         // more than this kernel's 64 slots are in use
            if (!redo && n_sel > nS) {
                for (uint32_t t = nS + (uint32_t)lane; t < n_sel; t += 64u) {
                    const int32_t jr = sample_site(a.grp_tab, a.grp_n, a.seed, a.tid, c, (int32_t)t + 1);
                    draw[t] = (uint8_t)(jr <= (int32_t)nS ? jr : 0);                            // (max_depth <= 64 here: nS = max_depth)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            bool wide_row = false;                                                              // a stored read with more than 21 calls on window sites
            // The stored reads' rows in one of two forms (round 6).  NARROW, when every stored read's calls lie within 16 consecutive
            // window sites (any two readers of the site then lie within 31 of them: bit = window site mod 32 is unambiguous): four words
            // -- start | end << 16 relative to c - 4096, and the three masks folded to 32 bits -- one 16-byte LDS read per operand and
            // round instead of two, half the mask arithmetic.  The round's four 16-byte reads were this kernel's other limit beside the
            // vector unit (0.94 busy; LDS 51 % with 41 % bank conflicts).  WIDE (64-bit masks, eight words) otherwise.
            bool span_wide = false;
            int32_t row_s = 0, row_e = 0;
            unsigned long long row_c = 0, row_a = 0, row_m = 0;
            if (!redo && (uint32_t)lane < nS) {
                uint32_t src = (uint32_t)lane;
                for (uint32_t t = nS; t < n_sel; ++t) src = draw[t] == (uint8_t)(lane + 1) ? t : src;
                const uint32_t r = lst[src];
                const uint32_t pk = s_pack[r];
                const int32_t rs = s_start[r];
                const unsigned long long mc = s_mC[r], mm = s_mM[r];
                wide_row = (uint32_t)__popcll(mc) > FT_NCPG_MAX;
                const uint32_t ab = pk >> 16;
                const unsigned long long ma = ab == 0xffu ? mc : mc & ~(1ull << ab);
                span_wide = mc == 0ull || 63u - (uint32_t)__builtin_clzll(mc) - (uint32_t)__builtin_ctzll(mc) > 15u;
                row_s = rs; row_e = rs + (int32_t)(pk & 0xffu); row_c = mc; row_a = ma; row_m = mm;
            }
            const bool narrow = !a.wide_rows && !__any(span_wide);                              // wave-uniform
            if (!redo && (uint32_t)lane < nS) {
                if (narrow) {
                    const uint32_t base = (uint32_t)c - 4096u;                                  // (spans <= 200 bp here: both offsets fit 16 bits)
                    *reinterpret_cast<uint4 *>(rows + (uint32_t)lane * 4u) =
                        make_uint4((((uint32_t)row_s - base) & 0xffffu) | (((uint32_t)row_e - base) << 16), (uint32_t)row_c | (uint32_t)(row_c >> 32),
                                   (uint32_t)row_a | (uint32_t)(row_a >> 32), (uint32_t)row_m | (uint32_t)(row_m >> 32));
                } else {
                    uint32_t *rw = rows + (uint32_t)lane * 8u;
                    rw[0] = (uint32_t)row_s; rw[1] = (uint32_t)row_e;
                    rw[2] = (uint32_t)row_c; rw[3] = (uint32_t)(row_c >> 32); rw[4] = (uint32_t)row_a; rw[5] = (uint32_t)(row_a >> 32);
                    rw[6] = (uint32_t)row_m; rw[7] = (uint32_t)(row_m >> 32);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t disc = 0;
            bool wide = false;
            float q_wide = 0.0f;
            const bool maybe_wide = __any(wide_row);                                            // else no pair shares more than 21 calls
            const unsigned long long off_v = s_off[k];
            const unsigned long long off = ((unsigned long long)sgpr((uint32_t)(off_v >> 32)) << 32) | sgpr((uint32_t)off_v);
            if (!redo) {
                const int P = (int)(nS * (nS - 1u)) >> 1;
                const uint16_t *const tab = a.pair_tab + (nS * (nS - 1u) * (nS - 2u)) / 6u;
                uint8_t *const tp = a.terms + off;
                uint32_t ent_next = tab[P ? min(lane, P - 1) : 0];
                // (round 6: the kernel runs at 0.94 of the vector unit, so the round was counted instruction by instruction -- hipcc -S:
                // 48 -> 40.  max(ov, 0) >= min_overlap is ov - 1 >= min_overlap - 1 when min_overlap >= 1 and true otherwise; the
                // discordant pairs are counted on the scalar unit from a ballot; "more than 21 shared calls" can only happen when a
                // stored read holds more than 21 calls -- known before the rounds, and such a site (CpG every 7 bp) takes the chained rounds
                // below straight away; ncpg (ncpg + 1) as one multiply-add; 32-bit pair indices.)
                const bool mo_any = a.min_overlap <= 0;
                const int32_t mo_m1 = a.min_overlap - 1;
                uint32_t disc_s = 0;                                                            // wave-uniform
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                for (uint32_t k0 = 0; narrow && k0 < (uint32_t)P; k0 += 64u) {                  // (narrow rows: never a pair with > 21 shared calls)
                    const uint32_t ent = ent_next;
                    if (k0 + 64u < (uint32_t)P) ent_next = tab[min(k0 + 64u + (uint32_t)lane, (uint32_t)P - 1u)];
                    typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));
                    const u32x4_a16 *const rows4 = reinterpret_cast<const u32x4_a16 *>(rows);
                    const u32x4_a16 ri = rows4[ent & 0xffu], rj = rows4[ent >> 8];
                    // both halves at once: the larger start in the low half of one, the smaller end in the high half of the other
                    const u16x2 lo2 = __builtin_elementwise_max(__builtin_bit_cast(u16x2, ri.x), __builtin_bit_cast(u16x2, rj.x));
                    const u16x2 hi2 = __builtin_elementwise_min(__builtin_bit_cast(u16x2, ri.x), __builtin_bit_cast(u16x2, rj.x));
                    const int32_t ov_m1 = (int32_t)hi2.y - (int32_t)lo2.x;                      // get_num_overlap_bases - 1, fdrp.rs:97-107
                    const bool pair_ok = (k0 + (uint32_t)lane < (uint32_t)P) & (mo_any | (ov_m1 >= mo_m1));   // fdrp.rs:134 (no short circuit: one 16-byte read per row)
                    const uint32_t ncpg = (uint32_t)__builtin_popcount(ri.y & rj.y);            // qfdrp.rs:109-119
                    const uint32_t ham = (uint32_t)__builtin_popcount(ri.z & rj.z & (ri.w ^ rj.w));   // fdrp.rs:114-115
                    // fdrp.rs:138-140: the discordant pairs are exactly the pairs with a non-zero term; only those are listed (x + 0.0 == x),
                    // packed in the pairs' own order
                    const bool dsc = pair_ok && ham != 0u;
                    const unsigned long long nzm = __builtin_amdgcn_ballot_w64(dsc);
                    uint32_t tri2;
                    asm("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(tri2) : "v"(ncpg));
                    if (dsc) tp[disc_s + __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u))] = (uint8_t)((tri2 >> 1) + ham);
                    disc_s += (uint32_t)__popcll(nzm);
                }
                const uint32_t Pu = (maybe_wide || narrow) ? 0u : (uint32_t)P;                  // (a wide site: no listed terms, the chained rounds below)
                for (uint32_t k0 = 0; k0 < Pu; k0 += 64u) {
                    const uint32_t ent = ent_next;
                    if (k0 + 64u < Pu) ent_next = tab[min(k0 + 64u + (uint32_t)lane, Pu - 1u)];
                    const uint32_t pi = ent & 0xffu, pj = ent >> 8;
                    const uint32_t *ri = rows + pi * 8u, *rj = rows + pj * 8u;
                    const int32_t si = (int32_t)ri[0], ei = (int32_t)ri[1], sj = (int32_t)rj[0], ej = (int32_t)rj[1];
                    const int32_t ov_m1 = min(ei, ej) - max(si, sj);                            // get_num_overlap_bases - 1, fdrp.rs:97-107
                    const bool pair_ok = (k0 + (uint32_t)lane < Pu) && (mo_any || ov_m1 >= mo_m1);   // fdrp.rs:134
                    const uint32_t ncpg = __builtin_popcount(ri[2] & rj[2]) + __builtin_popcount(ri[3] & rj[3]);   // qfdrp.rs:109-119
                    const uint32_t ham = __builtin_popcount(ri[4] & rj[4] & (ri[6] ^ rj[6])) +
                                         __builtin_popcount(ri[5] & rj[5] & (ri[7] ^ rj[7]));                      // fdrp.rs:114-115
                    const bool dsc = pair_ok && ham != 0u;                                       // fdrp.rs:138-140
                    const unsigned long long nzm = __builtin_amdgcn_ballot_w64(dsc);
                    uint32_t tri2;                                                               // ncpg (ncpg + 1) in one instruction
                    asm("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(tri2) : "v"(ncpg));
                    if (dsc) tp[disc_s + __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u))] = (uint8_t)((tri2 >> 1) + ham);
                    disc_s += (uint32_t)__popcll(nzm);
                }
                // the list's last 64-byte line is filled up with code 1 = 0 / 1 = +0.0: k_fdrp_chain adds whole lines
                if (!maybe_wide && (uint32_t)lane < ((64u - (disc_s & 63u)) & 63u)) tp[disc_s + (uint32_t)lane] = (uint8_t)1;
                disc = lane == 0 ? disc_s : 0u;                                                  // (summed over the wave below)
                // A pair that shares more than 21 calls: its code did not fit a byte and the list is void.  The site is CpG-dense -- handed to
                // the wave-per-site walk it takes the call-by-call path with calls beyond the registers, up to a millisecond for ONE site --
                // so its rounds are run again HERE with the ordered sum chained in the wave (one DPP add per pair, as the walk's compact
                // finalize does it): a few thousand such sites per batch, and the chain kernel sees none of them.
                wide = maybe_wide;
                if (wide) {
                    disc = 0;
                    ent_next = tab[P ? min(lane, P - 1) : 0];
                    for (int k0 = 0; k0 < P; k0 += 64) {
                        const uint32_t ent = ent_next;
                        if (k0 + 64 < P) ent_next = tab[min(k0 + 64 + lane, P - 1)];
                        const int pi = (int)(ent & 0xffu), pj = (int)(ent >> 8);
                        const uint32_t *ri = rows + pi * 8, *rj = rows + pj * 8;
                        const int32_t si = (int32_t)ri[0], ei = (int32_t)ri[1], sj = (int32_t)rj[0], ej = (int32_t)rj[1];
                        const int32_t ov = min(ei, ej) - max(si, sj) + 1;
                        const bool pair_ok = (k0 + lane < P) & (max(ov, 0) >= a.min_overlap);
                        const uint32_t ncpg = __builtin_popcount(ri[2] & rj[2]) + __builtin_popcount(ri[3] & rj[3]);
                        const uint32_t ham = __builtin_popcount(ri[4] & rj[4] & (ri[6] ^ rj[6])) + __builtin_popcount(ri[5] & rj[5] & (ri[7] ^ rj[7]));
                        disc += (pair_ok && ham != 0u) ? 1u : 0u;
                        const float term = pair_ok ? (float)ham / (float)ncpg : 0.0f;            // qfdrp.rs:152; +0.0 for skipped pairs
                        float x = (lane == 0) ? q_wide + term : term;
                        for (int stp = 0; stp < 63; ++stp)                                       // x[l] = x[l - 1] + term[l]: the reference's order
                            x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)) + term;
                        q_wide = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
                    }
                }
            }
            if (redo) {
                redo_bits |= 1u << k;
                if (lane == 0) a.flags[j] = FD_REDO;
            } else {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) disc += __shfl_xor(disc, o, 64);
                if (lane == 0) {
                    a.nreads[j] = nS;
                    if (wide) {
                        const unsigned long long prod = (unsigned long long)(long long)nS * (unsigned long long)((long long)nS - 1);
                        const float den = (float)prod / 2.0f;                                   // fdrp.rs:143
                        a.fdrp[j] = (float)disc / den; a.qfdrp[j] = q_wide / den; a.flags[j] = 1u;
                    } else { a.site_off[j] = off; a.site_nz[j] = disc; a.site_disc[j] = disc; a.flags[j] = FD_CHAIN; }
                }
            }
        }
        // the tile's handed-back sites, appended to the list k_fdrp_walk takes them from
        if (lane == 0) s_redo[wave] = redo_bits;
        __syncthreads();
        if (wave == 0) {
            const uint32_t m = s_redo[0] | s_redo[1] | s_redo[2] | s_redo[3];
            if (m) {                                                            // workgroup-uniform
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.redo_cnt, (uint32_t)__builtin_popcount(m));
                base = __shfl(base, 0, 64);
                if (lane < 32 && ((m >> lane) & 1u)) a.redo_list[base + (uint32_t)__builtin_popcount(m & ((1u << lane) - 1u))] = ja + (uint32_t)lane;
            }
        }
      }
    }
}

// one thread per site: the ordered f32 sum over the site's listed terms (qfdrp.rs:152), then fdrp.rs:143 / qfdrp.rs:155
__global__ __launch_bounds__(256) void k_fdrp_chain(const FdrpArgs a) {
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    if (blockIdx.x * 256u >= n_sites) return;              // (the host only has an upper bound of the site count)
    __shared__ float s_q[256];
    {   // code = ncpg (ncpg + 1) / 2 + ham, ham <= ncpg <= 21: the quotient by the division the pair loop would do (0 / 0 = NaN at code 0)
        uint32_t ncpg = 0;
        while ((ncpg + 1u) * (ncpg + 2u) / 2u <= (uint32_t)threadIdx.x) ++ncpg;
        s_q[threadIdx.x] = (float)((uint32_t)threadIdx.x - ncpg * (ncpg + 1u) / 2u) / (float)ncpg;
    }
    __syncthreads();
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n_sites; j += gridDim.x * 256u) {
    if (a.flags[j] != FD_CHAIN) continue;
    const uint8_t *__restrict__ t = a.terms + a.site_off[j];
    const uint32_t nz = a.site_nz[j];
    float q = 0.0f;
    {
    // A list starts on a 64-byte boundary and owns a multiple of 64 bytes: a lane takes a whole cache line per trip (four 16-byte
    // loads; 16 bytes per trip had every line fetched four times over, the L1 does not hold 64 lanes' lines), the next line is
    // requested before this one is used.  The codes past the list's end inside its last line are what the tile kernel's last
    // round wrote for the lanes without a pair: code 1 = 0 / 1 = +0.0, and x + 0.0 == x -- whole lines are added.
    const uint4 *__restrict__ t4 = reinterpret_cast<const uint4 *>(t);
    uint4 v[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (nz) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = t4[u];
    }
    for (uint32_t i = 0; i < nz; i += 64u) {
        uint32_t w[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) { w[4 * u] = v[u].x; w[4 * u + 1] = v[u].y; w[4 * u + 2] = v[u].z; w[4 * u + 3] = v[u].w; }
        if (i + 64u < nz) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t4[(i >> 4) + 4u + (uint32_t)u];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float f[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) f[k] = s_q[(w[4 * g + (k >> 2)] >> (8 * (k & 3))) & 0xffu];
#pragma unroll
            for (int k = 0; k < 16; ++k) q += f[k];
        }
    }
    }
    const uint32_t nS = a.nreads[j];
    // (num_reads * (num_reads - 1)) as f32 / 2.0 in usize arithmetic (fdrp.rs:143)
    const unsigned long long prod = (unsigned long long)(long long)nS * (unsigned long long)((long long)nS - 1);
    const float den = (float)prod / 2.0f;
    a.fdrp[j] = (float)a.site_disc[j] / den;
    a.qfdrp[j] = q / den;
    a.flags[j] = 1u;
    }
}

__global__ __launch_bounds__(256) void k_fdrp_emit(const uint32_t *__restrict__ flags, const int32_t *__restrict__ site_pos,
                                                   const float *__restrict__ f, const float *__restrict__ q,
                                                   const uint32_t *__restrict__ nr, const DevState *__restrict__ sites_st,
                                                   const uint32_t *__restrict__ blk, const unsigned long long *__restrict__ base,
                                                   int32_t *__restrict__ out_pos, float *__restrict__ out_f,
                                                   float *__restrict__ out_q, uint32_t *__restrict__ out_n) {
    emit_block(flags, sites_st->n_sites, *base, blk, [&](unsigned long long e, unsigned long long o) {
        out_pos[o] = site_pos[e]; out_f[o] = f[e]; out_q[o] = q[e]; out_n[o] = nr[e];
    });
}

// fdrp.rs:51-76: add_read() writes new_read[MAX_READ_LEN + (cpg - site)] for every call of the read once the read's own span has
// passed its two window tests (start - site >= -201, end - site <= 201).  The calls of a read lie in [start - 1, end] (start - 1: a
// reverse-strand call on the read's first base, readutil.rs:332-340), so the index leaves 0..=402 in exactly one case: a call at
// start - 1, the site at start + 201, end <= start + 402 -- index -1, an out-of-bounds panic.  One thread per read; any such read
// (passing mapq, as fdrp.rs:205 skips the others) raises ERRB_FDRPPANIC and the measure fails as the reference does.
__global__ __launch_bounds__(256) void k_fdrp_guard(const int32_t *__restrict__ read_start, const int32_t *__restrict__ read_end,
                                                    const uint8_t *__restrict__ read_mapq, const uint32_t *__restrict__ cpg_off,
                                                    const uint32_t *__restrict__ cpg_pos, uint32_t n_reads, uint32_t min_qual, uint32_t *err) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_reads) return;
    const int32_t s = read_start[i], e = read_end[i];
    if (e - s < 202 || e - s > 402 || read_mapq[i] < min_qual) return;
    const uint32_t o0 = cpg_off[i], o1 = cpg_off[i + 1];
    if (o1 - o0 < 2u || (cpg_pos[o0] & 0x7fffffffu) != ((uint32_t)(s - 1) & 0x7fffffffu)) return;
    for (uint32_t k = o0 + 1; k < o1; ++k)
        if ((cpg_pos[k] & 0x7fffffffu) == (uint32_t)(s + 201)) { atomicOr(err, (uint32_t)ERRB_FDRPPANIC); return; }
}

}  // namespace mth

using namespace mth;

extern "C" {

int mth_fdrp_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_fdrp_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    // (a group handle that is not defined is refused by stage_batch, before anything of the batch is recorded: ADVICE r04)
    mth_batch_t d;
    int rc = stage_batch(ctx, *batch, d);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    uint64_t bound = 0;
    // WGBS depth (about a dozen candidate reads per site -- reads starting in [c - max_span + 1, c + 1]), sparse calls, spans <= 200 bp:
    // ONE tile pass finds the sites and computes them (k_fdrp_wtile, round 6); the general walk only redoes what it hands back.
    // METHEOR_FDRP_WTILE=0 / 1 forces the choice (A/B, tests).
    const bool dense = d.n_reads && ((double)d.n_cpgs / (double)d.n_reads) > 6.0;
    const double cand = (double)d.n_reads * ((double)d.max_span + 2.0) / std::max<double>(1.0, (double)d.region_end - (double)d.region_beg);
    // (round 6: a dense form of the tile pass -- 64-bit masks, sites up to 64 reads, term lists in HBM -- was built for the depths of config 4:
    // parity-green, 4.2 ms against this file's k_fdrp_tile + k_fdrp_chain 3.4 ms: profiles/r06_fdrp_dense.md, tools/experiments/)
    // How deep: a site holds up to 64 readers there, but readers beyond max_depth are the reservoir's (handed back), so the depth must
    // stay 4 sigma below max_depth (CLI default 40: ~22 candidate reads a site).  Measured on a chr1-sized contig at density 0.0091,
    // pass ms, this form against rounds 2-5's choice: 16 M reads (10 x) 0.45 / 0.615, 24 M 0.85 / 1.48, 32 M (20 x) 1.26 / 2.91 (the
    // read x read form), 48 M (29 x) 3.06 / 3.45 (hand-backs 0.7), 64 M (39 x) 10.5 / 4.8.  Sparse calls only (<= 2 a read):
    // config 2 -- 25.6 x, 3 calls a read -- 2.1 / 1.35.
    const bool sparse_calls = d.n_reads && (double)d.n_cpgs / (double)d.n_reads <= 2.0;
    bool wtile = sparse_calls && d.max_span <= 200 && cand <= 24.0 && cand + 4.0 * std::sqrt(cand) <= (double)params->max_depth &&
                 !getenv("METHEOR_FDRP_WALK4") && !getenv("METHEOR_FDRP_TILE");
    if (const char *e = getenv("METHEOR_FDRP_WTILE")) wtile = d.max_span <= 200 && atoi(e) != 0;
    if (wtile) {
        // the candidate-site arrays the tile pass fills itself (no discovery pass)
        const uint64_t region_len = (uint64_t)((int64_t)d.region_end - d.region_beg);
        bound = d.n_cpgs < region_len ? d.n_cpgs : region_len;
        if (!ctx->d_state2) MTH_HIP(ctx, hipMalloc((void **)&ctx->d_state2, sizeof(DevState)));
        MTH_HIP(ctx, hipMemsetAsync(ctx->d_state2, 0, sizeof(DevState), s));
        MTH_HIP(ctx, ctx->s_pos.reserve((bound + 1) * 4, s));
    } else if ((rc = discover_sites(ctx, d, 0, params->min_qual, bound))) return rc;   // sites = positions called by reads passing mapq with >= 1 CpG (fdrp.rs:205-231)
    ctx->f_batches.push_back(BatchMeta{batch->tid});
    if (bound == 0) bound = 1;
    MTH_HIP(ctx, ctx->w_val.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_cov.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_aux.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_flags.reserve(bound * 4, s));
    if (!ctx->f_state.p) {
        MTH_HIP(ctx, ctx->f_state.reserve(4 * sizeof(unsigned long long), s));      // rows, base, the tile form's list cursor
        MTH_HIP(ctx, hipMemsetAsync(ctx->f_state.p, 0, 4 * sizeof(unsigned long long), s));
    }
    const uint64_t need = ctx->f_rows_bound + bound;
    if (need > ctx->f_cap) {
        const uint64_t ncap = need + need / 4 + 1024, used = ctx->f_rows_bound;
        MTH_HIP(ctx, ctx->f_pos.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->f_val.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->f_qval.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->f_n.reserve(ncap * 4, s, true, used * 4));
        ctx->f_cap = ncap;
    }
    ctx->f_rows_bound = need;
    const size_t nb = ctx->f_batches.size() - 1;
    MTH_HIP(ctx, ctx->f_batch_rows.reserve((nb + 1) * 4, s, true, nb * 4));

    // the reference's one panic on this path needs a read of >= 203 reference bases (see k_fdrp_guard): short-read batches skip the pass
    if (d.max_span >= 203 && d.n_reads)
        hipLaunchKernelGGL(k_fdrp_guard, dim3((uint32_t)((d.n_reads + 255) / 256)), dim3(256), 0, s, d.read_start, d.read_end, d.read_mapq, d.cpg_off,
                           d.cpg_pos, (uint32_t)d.n_reads, (uint32_t)params->min_qual, &ctx->d_state->err);
    const int32_t ext = ((d.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    FdrpArgs a;
    a.read_start = d.read_start; a.read_end = d.read_end; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
    a.idx = idx_ptr(ctx); a.sites_st = ctx->d_state2; a.site_pos = ctx->s_pos.as<int32_t>(); a.st = ctx->d_state;
    a.site_nc = ctx->s_nc.as<uint32_t>(); a.site_nd = ctx->s_nd.as<uint32_t>();
    a.fdrp = ctx->w_val.as<float>(); a.qfdrp = reinterpret_cast<float *>(ctx->w_aux.p); a.nreads = ctx->w_cov.as<uint32_t>();
    a.flags = ctx->w_flags.as<uint32_t>();
    a.seed = params->seed; a.idx_base = d.region_beg - ext; a.max_span = d.max_span; a.tid = d.tid;
    a.grp_tab = group_table(ctx, batch->tid, &a.grp_n);
    if (batch->tid <= -2 && !a.grp_tab) return fail(ctx, MTH_ERR_INVALID, "batch.tid is not a defined contig group's handle");
    a.min_overlap = params->min_overlap; a.n_reads = d.n_reads; a.n_cpgs = d.n_cpgs; a.region_beg = d.region_beg; a.region_end = d.region_end;
    a.min_depth = (uint32_t)std::min<uint64_t>(params->min_depth, 0xffffffffull); a.max_depth = params->max_depth;
    a.min_qual = params->min_qual;
    a.rows_scratch = nullptr; a.slots_cap = 0;
    a.redo_list = nullptr; a.redo_cnt = nullptr; a.no_compact = 0u;
    a.wide_rows = getenv("METHEOR_FDRP_TILE_WIDE_ROWS") ? 1u : 0u;
    a.terms = nullptr; a.cursor = nullptr; a.budget = 0; a.site_off = nullptr; a.site_nz = nullptr; a.site_disc = nullptr;
    if (!ctx->f_pairtab.p) {
        // the pairs of n stored reads in the reference's (i, j) loop order (fdrp.rs:129-141), n = 2..64, one after the other
        std::vector<uint16_t> t;
        t.reserve(43680 + 64);
        for (int n = 0; n <= 64; ++n)
            for (int i = 0; i + 1 < n; ++i)
                for (int j = i + 1; j < n; ++j) t.push_back((uint16_t)(i | (j << 8)));
        t.resize(t.size() + 64, 0);
        MTH_HIP(ctx, ctx->f_pairtab.reserve(t.size() * 2, s));
        MTH_HIP(ctx, hipMemcpyAsync(ctx->f_pairtab.p, t.data(), t.size() * 2, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));            // t is a local
    }
    a.pair_tab = ctx->f_pairtab.as<uint16_t>();
    if (params->max_depth > FD_DEPTH_MAX) return fail(ctx, MTH_ERR_CAPACITY, "FDRP / qFDRP max_depth above 16384 (the pair index of one site is 32-bit arithmetic)");
    const uint32_t grid = (uint32_t)std::min<uint64_t>((bound + 3) / 4, 16384);   // 4 waves (sites) per block
    {
        // (each kernel of the pass under its own timer id: bench.py's kernels_ms decomposes -- VERDICT r04 item 2)
        // dense CpGs (hotspots, RRBS): 16 call registers per stored read keep the per-call match out of
        // the memory loop; sparse WGBS keeps 8 (half the compares per call)
        // WGBS depth: when a site has about a dozen reads that can call it (reads starting in [c - max_span + 1, c + 1]), four
        // sites share a wave (k_fdrp_walk4<16>) and the general walk only redoes what that kernel handed back.  Measured on a
        // chr1-sized contig (profiles/r02_fdrp_walk4.md): ~10 such reads per site 0.887 -> 0.568 ms, 12 1.29 -> 0.94, 14.6 1.58 -> 1.39,
        // 17 1.79 -> 1.83 (the switch is at 16), ~19.5 per site 1.95 -> 2.17
        // (16 lanes) / 2.13 (32 lanes) -- deeper sites are bound by their pair rounds and keep the wave-per-site walk.
        // METHEOR_FDRP_WALK4=0 / 16 / 32 (1 = 16) forces the choice (A/B, tests).
        int walk4 = (!wtile && !dense && d.max_span <= 200 && cand <= 16.0) ? 16 : 0;                        // lanes per site
        if (const char *e = getenv("METHEOR_FDRP_WALK4")) { const int k = atoi(e); walk4 = d.max_span <= 200 ? (k == 1 ? 16 : (k == 16 || k == 32 ? k : 0)) : 0; }
        a.only_flag = 0u; a.redo_mask = nullptr;
        // The read x read form (k_fdrp_tile + k_fdrp_chain) for everything the four-sites-per-wave kernel does not take:
        // METHEOR_FDRP_TILE=0 / 1 forces the choice (A/B, tests).
        bool tile = d.max_span <= 200 && !walk4 && !wtile;
        if (const char *e = getenv("METHEOR_FDRP_TILE")) { tile = d.max_span <= 200 && atoi(e) != 0 && !wtile; if (tile) walk4 = 0; }
        if (wtile) {
            walk4 = 0;
            MTH_HIP(ctx, ctx->f_redo.reserve((bound + 2) * 4, s));
            unsigned long long *cur = ctx->f_state.as<unsigned long long>() + 2;
            MTH_HIP(ctx, hipMemsetAsync(cur, 0, 16, s));
            a.redo_list = ctx->f_redo.as<uint32_t>(); a.redo_cnt = reinterpret_cast<uint32_t *>(cur + 1);
            if ((rc = launch_fdrp_wtile(ctx, d, *params, a.pair_tab, a.redo_list, a.redo_cnt))) return rc;
            a.idx = idx_ptr(ctx);
            a.only_flag = FD_REDO; a.no_compact = 1u;
            if (getenv("METHEOR_FDRP_DEBUG")) {          // how much the tile pass handed back (synchronises: debugging only)
                DevState st;
                uint32_t redo = 0;
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(&st, ctx->d_state2, sizeof st, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&redo, a.redo_cnt, 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "[fdrp wtile] candidate rows %llu handed back %u\n", (unsigned long long)st.n_sites, redo);
            }
        }
        const uint64_t dcap = std::min<uint64_t>(std::max<uint32_t>(params->max_depth, 1u), 64u);
        // sum over sites of C(n, 2) <= (dcap - 1) / 2 x the sum of n <= (dcap - 1) / 2 x the batch's calls; + the 64-byte rounding
        const uint64_t budget = (uint64_t)d.n_cpgs * (dcap - 1) / 2 + 64 * bound + 64;
        if (tile && budget > ctx->f_terms.cap) {
            // a contig group can hold 2^32 calls: ~80 GB of term codes at -D 40.  The list is a bound, not a need -- when it would take
            // more than half of what the device has free, the batch keeps the walk kernels (same rows)
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || budget > fr / 2) tile = false;
        }
        if (tile) {
            MTH_HIP(ctx, ctx->f_terms.reserve(budget, s));
            MTH_HIP(ctx, ctx->f_soff.reserve(bound * 8, s));
            MTH_HIP(ctx, ctx->f_snz.reserve(bound * 4, s));
            MTH_HIP(ctx, ctx->f_sdisc.reserve(bound * 4, s));
            MTH_HIP(ctx, ctx->f_redo.reserve((bound + 2) * 4, s));
            unsigned long long *cur = ctx->f_state.as<unsigned long long>() + 2;
            MTH_HIP(ctx, hipMemsetAsync(cur, 0, 16, s));
            a.redo_list = ctx->f_redo.as<uint32_t>(); a.redo_cnt = reinterpret_cast<uint32_t *>(cur + 1);
            a.terms = ctx->f_terms.as<uint8_t>(); a.cursor = cur; a.budget = budget;
            a.site_off = ctx->f_soff.as<unsigned long long>(); a.site_nz = ctx->f_snz.as<uint32_t>(); a.site_disc = ctx->f_sdisc.as<uint32_t>();
            const uint32_t gridt = (uint32_t)std::min<uint64_t>((bound + FT_CORE - 1) / FT_CORE, 8192);
            {
                LaunchTimer lt(ctx, K_FDRPTILE);
                if (dense) hipLaunchKernelGGL(k_fdrp_tile<16>, dim3(gridt), dim3(256), 0, s, a);
                else hipLaunchKernelGGL(k_fdrp_tile<8>, dim3(gridt), dim3(256), 0, s, a);
            }
            {
                LaunchTimer lt(ctx, K_FDRPCHAIN);
                hipLaunchKernelGGL(k_fdrp_chain, dim3((uint32_t)std::min<uint64_t>((bound + 255) / 256, 8192)), dim3(256), 0, s, a);
            }
            a.only_flag = FD_REDO;
            if (getenv("METHEOR_FDRP_DEBUG")) {          // how much the tile form handed back (synchronises: debugging only)
                DevState st;
                unsigned long long cursor = 0;
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(&st, ctx->d_state2, sizeof st, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&cursor, cur, 8, hipMemcpyDeviceToHost);
                unsigned long long redo = 0;
                { uint32_t rc = 0; (void)hipMemcpy(&rc, a.redo_cnt, 4, hipMemcpyDeviceToHost); redo = rc; }
                fprintf(stderr, "[fdrp tile] sites %llu handed back %llu, list bytes claimed %llu of %llu\n", (unsigned long long)st.n_sites, redo, cursor, (unsigned long long)budget);
            }
        }
        if (walk4) {
            MTH_HIP(ctx, ctx->f_redo.reserve((bound / 64 + 2) * 8, s));
            a.redo_mask = ctx->f_redo.as<unsigned long long>();
            const uint32_t grid4 = (uint32_t)std::min<uint64_t>((bound + 255) / 256, 16384);   // 4 waves x 64 sites per block and step
            LaunchTimer lt(ctx, K_FDRPWALK4);
            // 32-site windows where 32 sites reach well beyond +- 200 bp (at the batch's site density; METHEOR_FDRP_WIN=64 / 32 forces it)
            const double sites_per_bp = d.n_reads ? (double)d.n_cpgs / (double)d.n_reads / (double)std::max(d.max_span, 1) : 0.0;   // (the site count itself is on the device)
            int win = sites_per_bp * 402.0 <= 12.0 ? 32 : 64;
            if (const char *e = getenv("METHEOR_FDRP_WIN")) win = atoi(e) == 32 ? 32 : 64;
            if (walk4 == 16 && win == 32) hipLaunchKernelGGL((k_fdrp_walk4<16, 32>), dim3(grid4), dim3(256), 0, s, a);
            else if (walk4 == 16) hipLaunchKernelGGL((k_fdrp_walk4<16, 64>), dim3(grid4), dim3(256), 0, s, a);
            else if (win == 32) hipLaunchKernelGGL((k_fdrp_walk4<32, 32>), dim3(grid4), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_fdrp_walk4<32, 64>), dim3(grid4), dim3(256), 0, s, a);
            a.only_flag = FD_REDO;
        }
        LaunchTimer lt(ctx, K_FDRPWALK);
        // (the tile pass's hand-backs are a few per thousand sites, taken from a list: a small grid -- 16 384 workgroups that find nothing took 40 us)
        const uint32_t grid1 = wtile ? std::min(grid, std::max(1024u, (uint32_t)std::min<uint64_t>(bound / 4096, 16384))) : grid;
        if (dense) hipLaunchKernelGGL((k_fdrp_walk<16, 64>), dim3(grid1), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_fdrp_walk<8, 64>), dim3(grid1), dim3(256), 0, s, a);
        a.only_flag = 0u;
        // max_depth > 64: the sites that held more than 64 reads at once were flagged, not computed: 256-slot pass over them
        if (params->max_depth > 64) hipLaunchKernelGGL((k_fdrp_walk<8, FD_SLOTS_DEEP>), dim3(grid), dim3(256), 0, s, a);
        // max_depth > 256: what the 256-slot pass flagged again, with max_depth rows per wave in HBM scratch (as many waves as
        // ~256 MB of rows allow; these sites are few and their pair count, not the wave count, is what takes the time)
        if (params->max_depth > (uint32_t)FD_SLOTS_DEEP) {
            const size_t row_bytes = (size_t)params->max_depth * (4 + 8) * 4;
            const uint32_t waves3 = (uint32_t)std::max<size_t>(64, std::min<size_t>(((size_t)256 << 20) / row_bytes, (size_t)grid * 4)) & ~3u;
            MTH_HIP(ctx, ctx->f_rows.reserve((size_t)waves3 * row_bytes, s));
            a.rows_scratch = ctx->f_rows.as<uint32_t>(); a.slots_cap = params->max_depth;
            hipLaunchKernelGGL((k_fdrp_walk<8, 0>), dim3(waves3 / 4), dim3(256), 0, s, a);
        }
    }
    unsigned long long *fs = ctx->f_state.as<unsigned long long>();
    const uint32_t nblk = (uint32_t)((bound + 256 * SCAN_PER - 1) / (256 * SCAN_PER));
    MTH_HIP(ctx, ctx->w_blk.reserve((size_t)nblk * 4, s));
    {
        LaunchTimer lt(ctx, K_FDRPEMIT);
        hipLaunchKernelGGL(k_flags_blockcount, dim3(std::min(nblk, SCAN_GRID_MAX)), dim3(256), 0, s, ctx->w_flags.as<uint32_t>(),
                           (const unsigned long long *)&ctx->d_state2->n_sites, ctx->w_blk.as<uint32_t>());
        hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->w_blk.as<uint32_t>(), nblk, fs, fs + 1,
                           ctx->f_batch_rows.as<uint32_t>(), (uint32_t)nb, (const unsigned long long *)&ctx->d_state2->n_sites);
        hipLaunchKernelGGL(k_fdrp_emit, dim3(std::min(nblk, SCAN_GRID_MAX)), dim3(256), 0, s, ctx->w_flags.as<uint32_t>(), ctx->s_pos.as<int32_t>(),
                           ctx->w_val.as<float>(), reinterpret_cast<const float *>(ctx->w_aux.p), ctx->w_cov.as<uint32_t>(),
                           ctx->d_state2, ctx->w_blk.as<uint32_t>(), fs + 1, ctx->f_pos.as<int32_t>(), ctx->f_val.as<float>(),
                           ctx->f_qval.as<float>(), ctx->f_n.as<uint32_t>());
    }
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

int mth_fdrp_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos, float *fdrp, float *qfdrp,
                   uint32_t *n_reads) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    unsigned long long fs[2] = {0, 0};
    if (ctx->f_state.p) MTH_HIP(ctx, hipMemcpy(fs, ctx->f_state.p, sizeof fs, hipMemcpyDeviceToHost));
    const uint64_t n = fs[0];
    if (n_rows) *n_rows = n;
    if (n == 0) return MTH_OK;
    if (pos) MTH_HIP(ctx, hipMemcpy(pos, ctx->f_pos.p, n * 4, hipMemcpyDeviceToHost));
    if (fdrp) MTH_HIP(ctx, hipMemcpy(fdrp, ctx->f_val.p, n * 4, hipMemcpyDeviceToHost));
    if (qfdrp) MTH_HIP(ctx, hipMemcpy(qfdrp, ctx->f_qval.p, n * 4, hipMemcpyDeviceToHost));
    if (n_reads) MTH_HIP(ctx, hipMemcpy(n_reads, ctx->f_n.p, n * 4, hipMemcpyDeviceToHost));
    const bool grouped = (tid || pos) && has_group_batch(ctx, 2);          // rows of contig groups: back under their own contig
    std::vector<int32_t> tmp;
    if (grouped && !(tid && pos)) {
        tmp.resize(n);
        if (!pos) MTH_HIP(ctx, hipMemcpy(tmp.data(), ctx->f_pos.p, n * 4, hipMemcpyDeviceToHost));
    }
    int32_t *tid_w = tid ? tid : (grouped ? tmp.data() : nullptr), *pos_w = pos ? pos : (grouped ? tmp.data() : nullptr);
    if (tid_w) {
        std::vector<uint32_t> rows(ctx->f_batches.size());
        if (!rows.empty()) MTH_HIP(ctx, hipMemcpy(rows.data(), ctx->f_batch_rows.p, rows.size() * 4, hipMemcpyDeviceToHost));
        uint64_t o = 0;
        for (size_t b = 0; b < rows.size(); ++b)
            for (uint32_t j = 0; j < rows[b]; ++j) tid_w[o++] = ctx->f_batches[b].tid;
    }
    if (grouped) return ungroup_rows(ctx, n, tid_w, pos_w, 1, 1, nullptr);
    return MTH_OK;
}

}  // extern "C"
