// mth_mhl_wtile.hip -- MHL (mhl.rs:135-208, 43-73; readutil.rs:147-164) at WGBS depth: ONE WAVE per tile (round 6).
//
// Same computation as k_mhl_tile (mth_mhl_tile.hip: per site two small histograms over the reads that cover it -- hn[n] reads with n
// CpGs, hm[m] maximal methylated runs of length m -- and S[l], D[l] by suffix sums; a site whose covering reads may form several
// segments of the reference's stream is handed on to the exact walks), in the form k_fdrp_wtile found for this depth class: a tile of
// ~1500 positions, ~110 candidate reads, one wave, no workgroup barrier; the reads one per lane, then their CALLS one per lane
// (coalesced; a read holds 1.4 calls at WGBS density and one read in twenty contributes at all -- mhl.rs:176, 181: mapq, >= min_cpgs
// CpGs), the sites from a position bitmap of the contributors' calls (rank = slot: no hash table, rows come out sorted), histograms
// by LDS atomics.  k_mhl_tile on config 3: a 256-thread workgroup on 16 384 positions, ~10 barriers, 2.16 ms at 0.22 of the HBM
// roofline, bound by what it issues at six workgroups per CU (profiles/r05_mhl_tile.md).
//
// Exactness (as k_mhl_tile): site c can have more than one segment only if some read k with >= 1 CpG has start_k - 1 <= c < first_cpg(k)
// while a LATER read still calls c.  c >= start_k: the stretch [start_k, first_cpg(k)) is marked in a bitmap F, a site under a mark is
// handed on.  c == start_k - 1: a bitmap G2 of start - 1 of every read that does not call it, a bitmap G1 of the positions a contributor
// calls at its own start - 1; a site in both is handed on (the order of the two reads is not looked at: conservative, ~1 site in 10^4).
// Sites covered by a read with more than 16 CpGs are handed on as well.  Rows: per tile, sorted by position, in a scratch slice;
// k_mhl_wtile_gather packs the slices into the candidate-site arrays the rest of the MHL pipeline works on.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mth_ctx.h"
#include "mth_scan.h"
#include "mth_tile_dev.h"
#include "mth_wave_tile.h"

namespace mth {

struct MwRec { int32_t pos; float val; uint32_t cov, flags; };
static_assert(sizeof(MwRec) == sizeof(SiteRec), "the PDR scratch buffer is reused");

struct MwArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    int32_t region_beg, region_end, idx_base, max_span;
    uint32_t n_reads, n_cpgs, ntiles, min_depth, min_cpgs;
    uint32_t tile_w;              // positions per tile (<= MW_WMAX)
    uint8_t min_qual;
    uint8_t force_sub;            // tests: start every tile with 128-position stretches
    uint8_t force_hand_on;        // tests: every site goes to the exact walks
    MwRec *scratch;               // rows_per_tile rows per tile
    uint32_t rows_per_tile;
    uint32_t *tile_cnt;
    unsigned long long *bucket;   // rows per 256 tiles
    DevState *st;
};

constexpr int MW_WMAX = 2048;                           // one bitmap word per lane
constexpr int MW_U = 2, MW_RCAP = 64 * MW_U;            // candidate reads of a stretch
constexpr int MW_V = 4, MW_CCAP = 64 * MW_V;            // the calls of the contributors among them (<= 255: the owner scan packs a lane + 1 in eight bits)
constexpr int MW_SC = 32;                               // sites (positions a contributor calls) of a stretch: one per lane, one histogram each
constexpr int MW_HW = 17;                               // histogram words per site: 8 hn, 8 hm (two 16-bit bins a word), one of padding (bank spread)
constexpr int MW_LCAP = 16;                             // CpGs of a contributing read (more: its sites are handed on)

__global__ __launch_bounds__(64, 8) void k_mhl_wtile(const MwArgs a) {
    constexpr int U = MW_U;
    __shared__ uint2 s_bp[64];                                // {site bits, sites in the words before} over the stretch's positions
    __shared__ uint32_t s_F[64], s_G1[64], s_G2[64];          // flusher marks; "a contributor calls its own start - 1 here"; "start - 1 of a read that does not call it"
    __shared__ uint32_t s_hist[MW_SC * MW_HW];
    __shared__ uint32_t s_sflag[MW_SC];                       // a read with > 16 CpGs calls the site
    __shared__ uint32_t s_rs[MW_RCAP];                        // per read: start - (P0 - max_span - 1) | contributes << 30 | > 16 CpGs << 31
    __shared__ uint32_t s_ro0[MW_RCAP];                       // per read: offset of its first call
    __shared__ uint32_t s_rw[MW_RCAP * 2];                    // per contributing read: methylation bits by call index -> its runs (length - 1, 4 bits each); n | runs << 8
    __shared__ int32_t s_cpos[MW_SC];
    __shared__ __attribute__((aligned(4))) uint8_t s_owner[MW_CCAP];   // call slot -> read slot + 1 where a read's calls begin, else 0
    const int lane = threadIdx.x;
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (a.ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= a.ntiles) return;
    const int32_t T0 = a.region_beg + (int32_t)(t * a.tile_w);
    const int32_t T1 = (int32_t)min((int64_t)T0 + a.tile_w, (int64_t)a.region_end);
    MwRec *__restrict__ out = a.scratch + (size_t)t * a.rows_per_tile;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t mincp = max(a.min_cpgs, 1u);
    uint32_t rows_out = 0, bad = 0;
    uint32_t sub_w = a.force_sub ? 128u : a.tile_w;                     // stretch width (wave-uniform): halved when a stretch does not fit
    for (int64_t P0l = T0; P0l < T1;) {
        const int32_t P0 = (int32_t)P0l;
        const int32_t P1 = (int32_t)min(P0l + (int64_t)sub_w, (int64_t)T1);
        const uint32_t Wp = (uint32_t)(P1 - P0);
        // candidate reads: start in [P0 - max_span + 1, P1]  (a call sits in [start - 1, start - 1 + max_span])
        const uint32_t lo = __builtin_amdgcn_readfirstlane(min(a.idx[((uint32_t)P0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads));
        const uint32_t hi = __builtin_amdgcn_readfirstlane(min(a.idx[(((uint32_t)P1 - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads));
        const uint32_t R = hi - lo;
        // more candidate reads than the arrays hold: the stretch in halves; at 64 positions (hundreds-fold depth) the sites are only
        // found and all handed on
        if (R > (uint32_t)MW_RCAP && sub_w > 64u) { sub_w = max((sub_w >> 1) & ~31u, 64u); continue; }
        uint32_t o0[U], o1[U], mq[U];
        int32_t rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                                               // every chunk's fields are requested at once
            const uint32_t r = (uint32_t)(u * 64 + lane);
            const uint32_t i = min(lo + r, hi ? hi - 1u : 0u);
            o0[u] = a.cpg_off[i]; o1[u] = a.cpg_off[i + 1];
            rs[u] = a.read_start[i]; mq[u] = a.read_mapq[i];
        }
        bool heavy = R > (uint32_t)MW_RCAP;
        const uint32_t wb0 = (uint32_t)P0 - (uint32_t)a.max_span - 1u;              // origin of the reads' start offsets
        FW_SYNC();                                                                  // the previous stretch's LDS is done with
        s_bp[lane] = make_uint2(0u, 0u);
        s_F[lane] = 0u; s_G1[lane] = 0u; s_G2[lane] = 0u;
        for (int i = lane; i < MW_SC * MW_HW; i += 64) s_hist[i] = 0u;
        if (lane < MW_SC) s_sflag[lane] = 0u;
        reinterpret_cast<uint32_t *>(s_owner)[lane] = 0u;
        FW_SYNC();
        // a stretch of positions [f0, f1) of the core, marked in a bitmap
        auto mark_range = [&](uint32_t *bm, const int64_t f0l, const int64_t f1l) {
            const int64_t f0 = max(f0l, (int64_t)P0) - P0, f1 = min(f1l, (int64_t)P1) - P0;
            if (f0 < f1) {
                const uint32_t b_lo = (uint32_t)f0, b_hi = (uint32_t)f1 - 1u;       // inclusive bit range
                for (uint32_t w = b_lo >> 5; w <= (b_hi >> 5); ++w) {
                    uint32_t m = 0xffffffffu;
                    if (w == (b_lo >> 5)) m &= 0xffffffffu << (b_lo & 31u);
                    if (w == (b_hi >> 5)) m &= 0xffffffffu >> (31u - (b_hi & 31u));
                    atomicOr(&bm[w], m);
                }
            }
        };
        uint32_t ncall[U], rflag[U];                                                 // rflag: 1 contributes (mapq, min_cpgs <= n <= 16), 2 the same with n > 16
        uint32_t C_c = 0;                                                            // calls of the stretch's contributing reads (wave-uniform)
        if (!heavy) {
            // ---- A1: the candidate reads, one per lane: the FIRST call of each (every read with a CpG is a flusher, mhl.rs:162-173, before
            // the filters): the stretch [start, first CpG) is marked, start - 1 if the read does not call it; the contributors (mhl.rs:176,
            // 181: one read in twenty at WGBS density) get a run of call lanes each ----
            uint32_t fw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) fw[u] = a.cpg_pos[min(o0[u], a.n_cpgs ? a.n_cpgs - 1u : 0u)];   // (a.cpg_pos holds >= 1 word, mth_api stage_batch)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = (uint32_t)(u * 64 + lane);
                ncall[u] = 0u; rflag[u] = 0u;
                if ((uint32_t)(u * 64) >= R) continue;                              // wave-uniform
                const uint32_t n = r < R ? o1[u] - o0[u] : 0u;
                // (the index hands out whole 32-bp quanta: reads that start outside [P0 - max_span + 1, P1] can neither call nor flush a
                // position of the stretch)
                const bool inr = (uint32_t)rs[u] - ((uint32_t)P0 - (uint32_t)a.max_span + 1u) <= Wp + (uint32_t)a.max_span - 1u;
                const bool passf = inr && n >= mincp && mq[u] >= (uint32_t)a.min_qual;   // mhl.rs:176, 181
                rflag[u] = passf ? (n > (uint32_t)MW_LCAP ? 2u : 1u) : 0u;
                ncall[u] = n;
                if (inr && n) {
                    const uint32_t first = fw[u] & 0x7fffffffu;
                    // candidate ranges rely on every call lying in [start - 1, start - 1 + max_span]: the first call here, a contributor's others in A2
                    bad |= (first - ((uint32_t)rs[u] - 1u) > (uint32_t)a.max_span) ? 1u : 0u;
                    mark_range(s_F, (int64_t)rs[u], (int64_t)first);
                    if ((int64_t)first > (int64_t)rs[u] - 1) mark_range(s_G2, (int64_t)rs[u] - 1, (int64_t)rs[u]);
                    else if (passf) mark_range(s_G1, (int64_t)rs[u] - 1, (int64_t)rs[u]);   // a contributor calls its own start - 1
                }
                // call lanes of the contributors: base = calls of the contributors before (chunks, then lanes)
                const uint32_t nc = passf ? n : 0u;
                const uint32_t incl = wave_scan_incl(nc);
                const uint32_t cb = C_c + incl - nc;
                C_c += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
                s_rs[r] = (((uint32_t)rs[u] - wb0) & 0x3fffffffu) | (rflag[u] << 30);
                s_ro0[r] = o0[u];
                s_rw[r * 2] = 0u; s_rw[r * 2 + 1] = 0u;
                if (nc && cb < (uint32_t)MW_CCAP) s_owner[cb] = (uint8_t)(r + 1u);
            }
            if (C_c > (uint32_t)MW_CCAP) {
                if (sub_w > 64u) { sub_w = max((sub_w >> 1) & ~31u, 64u); continue; }
                heavy = true;
                FW_SYNC();
                s_F[lane] = 0u;                                                      // (the count-only pass below does not need the marks: every site is handed on)
            }
        }
        if (heavy) {
            // count-only: every contributor's calls are sites; all of them are handed on
            for (uint32_t r = (uint32_t)lane; r < R; r += 64) {
                const uint32_t i = lo + r;
                const uint32_t q0 = a.cpg_off[i], q1 = a.cpg_off[i + 1];
                if (q1 == q0) continue;
                const int32_t st = a.read_start[i];
                const bool passf = q1 - q0 >= mincp && a.read_mapq[i] >= a.min_qual;
                if (passf) for (uint32_t k = q0; k < q1; ++k) {
                    const uint32_t pw = a.cpg_pos[k] & 0x7fffffffu, d = pw - (uint32_t)P0;
                    bad |= (pw - ((uint32_t)st - 1u) > (uint32_t)a.max_span) ? 1u : 0u;
                    if (d < Wp) atomicOr(&s_bp[d >> 5].x, 1u << (d & 31u));
                }
            }
        }
        FW_SYNC();
        // ---- A2: the contributors' calls, one per lane: the read (latest owner mark at or before the lane) and the call's index in it; the
        // calls are sites and set the read's methylation bits ----
        uint32_t cwv[MW_V], cown[MW_V];                                              // the call word; read slot + 1 | contributes << 8 | > 16 CpGs << 9
        {
            uint32_t own_carry = 0;
#pragma unroll
            for (int v = 0; v < MW_V; ++v) {
                cwv[v] = 0u; cown[v] = 0u;
                if (heavy || (uint32_t)(v * 64) >= C_c) continue;                    // wave-uniform
                const uint32_t c = (uint32_t)(v * 64 + lane);
                const bool valid = c < C_c;
                const uint32_t own = valid ? (uint32_t)s_owner[c] : 0u;
                // (call lane of the read's first call + 1) << 8 | read slot + 1: both ascend along the lanes -- one maximum scan carries them
                const uint32_t hv = max(fw_wave_scan_max_incl(own ? ((c + 1u) << 8) | own : 0u), own_carry);
                own_carry = (uint32_t)__builtin_amdgcn_readlane(hv, 63);
                if (!valid) continue;
                const uint32_t r1 = hv & 0xffu, idx = c + 1u - (hv >> 8);
                const uint32_t rsw = s_rs[r1 - 1u];                                 // (every call lane has an owner: r1 >= 1)
                const uint32_t w = a.cpg_pos[s_ro0[r1 - 1u] + idx];
                const uint32_t pw = w & 0x7fffffffu;
                const uint32_t st_m1 = (rsw & 0x3fffffffu) + wb0 - 1u;              // the read's start - 1
                cwv[v] = w;
                cown[v] = r1 | ((rsw >> 30) << 8);
                bad |= (pw - st_m1 > (uint32_t)a.max_span) ? 1u : 0u;
                const uint32_t d = pw - (uint32_t)P0;
                if (d < Wp) atomicOr(&s_bp[d >> 5].x, 1u << (d & 31u));
                if ((rsw >> 30) == 1u && (w >> 31)) atomicOr(&s_rw[(r1 - 1u) * 2], 1u << idx);   // (idx < 16)
            }
        }
        FW_SYNC();
        // ---- B: the sites' slots = their ranks ----
        const uint32_t wb = s_bp[lane].x;
        const uint32_t wcnt = (uint32_t)__builtin_popcount(wb);
        uint32_t pre = wave_scan_incl(wcnt) - wcnt;
        s_bp[lane].y = pre;
        const uint32_t nsites = (uint32_t)__builtin_amdgcn_readlane(pre + wcnt, 63);
        if (nsites > (uint32_t)MW_SC) { sub_w = max((min(sub_w, Wp) >> 1) & ~15u, 32u); continue; }   // (32 positions hold <= 16 sites)
        {
            uint32_t bits = wb;
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
                s_cpos[pre] = P0 + (int32_t)((uint32_t)lane * 32u + b);
                ++pre;
            }
        }
        FW_SYNC();
        auto slot_of = [&](const uint32_t d) {                                       // sites of the stretch below offset d
            const uint2 e = s_bp[d >> 5];
            return e.y + (uint32_t)__builtin_popcount(e.x & ((1u << (d & 31u)) - 1u));
        };
        if (!heavy) {
            // ---- C0: per contributing read: its maximal methylated runs, length - 1 in four bits each (readutil.rs:147-164) ----
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((uint32_t)(u * 64) >= R) continue;                              // wave-uniform
                const uint32_t r = (uint32_t)(u * 64 + lane);
                uint32_t runs = 0, n_runs = 0;
                if (rflag[u] == 1u) {
                    uint32_t x = s_rw[r * 2];
                    while (x) {                                                     // at most 8 runs in 16 calls
                        x >>= __builtin_ctz(x);
                        const uint32_t m = (uint32_t)__builtin_ctz(~x);           // 1..16 (x < 2^16)
                        x >>= m;
                        runs |= (m - 1u) << (4u * n_runs);
                        ++n_runs;
                    }
                    s_rw[r * 2] = runs; s_rw[r * 2 + 1] = ncall[u] | (n_runs << 8);
                }
            }
            FW_SYNC();
            // ---- C1: per call of a contributor: the site's histogram increments ----
#pragma unroll
            for (int v = 0; v < MW_V; ++v) {
                if ((uint32_t)(v * 64) >= C_c) continue;                             // wave-uniform
                const uint32_t kind = cown[v] >> 8;
                const uint32_t d = (cwv[v] & 0x7fffffffu) - (uint32_t)P0;
                const bool live = kind != 0u && d < Wp;
                const uint32_t h = live ? slot_of(d) : 0u;
                uint32_t *hist = &s_hist[h * MW_HW];
                uint32_t runs = 0, nn = 1, n_runs = 0;
                if (live && kind == 1u) { const uint32_t r = (cown[v] & 0xffu) - 1u; runs = s_rw[r * 2]; const uint32_t x = s_rw[r * 2 + 1]; nn = x & 0xffu; n_runs = x >> 8; }
                if (live && kind == 2u) { atomicOr(&s_sflag[h], 1u); atomicAdd(&hist[0], 1u); }   // (counted for the min_depth test)
                if (live && kind == 1u) atomicAdd(&hist[(nn - 1u) >> 1], ((nn - 1u) & 1u) ? 0x10000u : 1u);
                const uint32_t max_runs = fw_wave_max(n_runs);
                for (uint32_t rr = 0; rr < max_runs; ++rr) {
                    const uint32_t m1 = (runs >> (4u * rr)) & 15u;
                    if (rr < n_runs) atomicAdd(&hist[8u + (m1 >> 1)], (m1 & 1u) ? 0x10000u : 1u);
                }
            }
            FW_SYNC();
        }
        // ---- D: one lane per site: coverage, hand-on tests, compute_mhl (mhl.rs:43-73) ----
        const bool is_site = (uint32_t)lane < nsites;
        uint32_t hn[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) hn[w] = is_site ? s_hist[lane * MW_HW + w] : 0u;
        const bool longr = is_site && s_sflag[lane & (MW_SC - 1)] != 0u;
        uint32_t cov = 0;
        if (heavy) cov = is_site ? 0xffffffffu : 0u;                                 // (unknown: the walk decides)
        else if (longr) {
            // word 0 holds hn[1] | hn[2] << 16 plus one per long read: an upper bound of the coverage is all the row test needs
#pragma unroll
            for (int w = 1; w < 8; ++w) cov += (hn[w] & 0xffffu) + (hn[w] >> 16);
            cov += hn[0];
        } else {
#pragma unroll
            for (int w = 0; w < 8; ++w) cov += (hn[w] & 0xffffu) + (hn[w] >> 16);
        }
        const bool row = is_site && cov >= a.min_depth;
        MwRec rec;
        rec.pos = 0; rec.val = 0.0f; rec.cov = cov; rec.flags = 4u;
        if (row) {
            const int32_t c = s_cpos[lane & (MW_SC - 1)];
            rec.pos = c;
            const uint32_t d = (uint32_t)(c - P0);
            bool hand_on = heavy || longr || a.force_hand_on;
            hand_on = hand_on || ((s_F[d >> 5] >> (d & 31u)) & 1u) || (((s_G1[d >> 5] & s_G2[d >> 5]) >> (d & 31u)) & 1u);
            if (!hand_on) {
                // suffix sums twice give S[l], D[l]; same operations and order as mhl_walk_site's finalize() / k_mhl_tile
                uint32_t hm[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) hm[w] = s_hist[lane * MW_HW + 8 + w];
                uint32_t S[MW_LCAP], D[MW_LCAP];
                uint32_t maxn = 0;
#pragma unroll
                for (int l = 0; l < MW_LCAP; ++l) {
                    S[l] = (l & 1) ? hm[l >> 1] >> 16 : hm[l >> 1] & 0xffffu;
                    D[l] = (l & 1) ? hn[l >> 1] >> 16 : hn[l >> 1] & 0xffffu;
                    if (D[l]) maxn = (uint32_t)l + 1u;
                }
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                    for (int l = MW_LCAP - 2; l >= 0; --l) { S[l] += S[l + 1]; D[l] += D[l + 1]; }
                float l_sum = 0.0f;
                for (uint32_t l = 1; l < maxn + 1; ++l) l_sum = l_sum + (float)l;
                float mhl = 0.0f;
#pragma unroll
                for (int l = 1; l <= MW_LCAP; ++l)
                    if (S[l - 1] > 0) { const float tq = ((float)l * (float)S[l - 1]) / (float)D[l - 1]; mhl = mhl + tq; }
                rec.val = mhl / l_sum;
                rec.flags = 1u;
            }
        }
        const unsigned long long em = fw_ballot(row);
        if (row && rows_out + (uint32_t)__popcll(em & lt_mask) < a.rows_per_tile) out[rows_out + (uint32_t)__popcll(em & lt_mask)] = rec;
        rows_out += (uint32_t)__popcll(em);
        P0l = P1;
    }
    if (bad & 1u) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
    if (lane == 0) {
        a.tile_cnt[t] = rows_out;
        if (rows_out) atomicAdd(a.bucket + (t >> TILE_BUCKET_SHIFT), (unsigned long long)rows_out);
    }
}

// Eight tiles per wave, eight lanes each (as k_fdrp_wtile_gather): the tiles' rows to the candidate-site arrays, the handed-on sites (flag 4)
// also to the list k_mhl_walk_wave takes them from (its count lives in the sink state's first spare counter).
constexpr int MWG_WAVES = 4;
__global__ __launch_bounds__(64 * MWG_WAVES) void k_mhl_wtile_gather(const MwRec *__restrict__ scratch, const uint32_t *__restrict__ tile_cnt,
                                                                     const unsigned long long *__restrict__ bucket_pre, const uint32_t ntiles,
                                                                     const uint32_t rows_per_tile, DevState *__restrict__ sites_st,
                                                                     int32_t *__restrict__ site_pos, float *__restrict__ val,
                                                                     uint32_t *__restrict__ cov, uint32_t *__restrict__ flags,
                                                                     uint32_t *__restrict__ hand_list) {
    const int lane = threadIdx.x & 63, sub = lane >> 3, l8 = lane & 7;
    const uint32_t tw = (blockIdx.x * MWG_WAVES + (threadIdx.x >> 6)) * 8u;         // the wave's first tile
    if (tw >= ntiles) return;
    const uint32_t t_first = (tw >> TILE_BUCKET_SHIFT) << TILE_BUCKET_SHIFT;
    uint32_t in_bucket = 0;
    for (uint32_t q = t_first + lane; q < tw; q += 64) in_bucket += tile_cnt[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) in_bucket += __shfl_xor(in_bucket, o, 64);
    const uint32_t t = tw + (uint32_t)sub;
    const uint32_t n = t < ntiles ? tile_cnt[t] : 0u;
    uint32_t before = 0, n_max = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint32_t nk = __shfl(n, k * 8, 64); if (k < sub) before += nk; n_max = max(n_max, nk); }
    const unsigned long long base = bucket_pre[tw >> TILE_BUCKET_SHIFT] + in_bucket + before;
    const MwRec *__restrict__ src = scratch + (size_t)min(t, ntiles - 1u) * rows_per_tile;
    for (uint32_t i0 = 0; i0 < n_max; i0 += 8) {
        const uint32_t i = i0 + (uint32_t)l8;
        const bool in = i < n;
        MwRec r;
        r.pos = 0; r.val = 0.0f; r.cov = 0u; r.flags = 0u;
        if (in) r = src[i];
        if (in) { site_pos[base + i] = r.pos; val[base + i] = r.val; cov[base + i] = r.cov; flags[base + i] = r.flags; }
        const unsigned long long hb = __ballot(in && r.flags == 4u);
        if (hb) {                                                                    // wave-uniform
            unsigned long long at = 0;
            if (lane == 0) at = atomicAdd(reinterpret_cast<unsigned long long *>(&sites_st->lpmd[0]), (unsigned long long)__popcll(hb));
            at = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(at >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)at);
            if (in && r.flags == 4u) hand_list[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u))] = (uint32_t)(base + i);
        }
    }
    if (t == ntiles - 1 && l8 == 0) sites_st->n_sites = base + n;
}

// The wave-per-tile pass of one batch (called by launch_mhl_tile for sparse batches): candidate-site arrays and hand list as
// launch_mhl_tile leaves them; idx_base_out = origin of the read index (the hand-on walk's).
int launch_mhl_wtile(mth_ctx *ctx, const mth_batch_t &d, const mth_mhl_params_t &p, int32_t &idx_base_out) {
    hipStream_t s = ctx->stream;
    const int64_t region_len = (int64_t)d.region_end - d.region_beg;
    // Tile width: the widest whose candidate reads (start in [P0 - max_span + 1, P1]) fill the stretch's two 64-read chunks without spilling
    // over too often (mean + 2 sigma <= 128: an overfull tile is redone in halves), at most MW_WMAX
    int W;
    {
        const double rpb = (double)d.n_reads / (double)region_len;
        const double want = 128.0 - 2.0 * std::sqrt(128.0);
        W = (int)(want / std::max(rpb, 1e-9)) - d.max_span;
        W = std::max(256, std::min(MW_WMAX, W)) & ~63;
    }
    if (const char *e = getenv("MTH_MHL_WTILE_W")) W = std::min(MW_WMAX, std::max(64, atoi(e))) & ~63;   // tests / tuning
    uint32_t ntiles = 0;
    int rc = build_read_index(ctx, d, W, idx_base_out, ntiles);
    if (rc) return rc;
    const uint32_t nbk = (ntiles + (1u << TILE_BUCKET_SHIFT) - 1) >> TILE_BUCKET_SHIFT;
    const uint32_t rows_per_tile = (uint32_t)W / 2u;                                 // CpG sites lie at least two positions apart
    MTH_HIP(ctx, ctx->tile_cnt.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_bucket.reserve((size_t)nbk * 5 * sizeof(unsigned long long), s));
    MTH_HIP(ctx, hipMemsetAsync(ctx->tile_bucket.p, 0, (size_t)nbk * sizeof(unsigned long long), s));
    MTH_HIP(ctx, ctx->scratch.reserve((size_t)ntiles * rows_per_tile * sizeof(MwRec), s));
    MwArgs a;
    a.read_start = d.read_start; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos; a.idx = idx_ptr(ctx);
    a.region_beg = d.region_beg; a.region_end = d.region_end; a.idx_base = idx_base_out; a.max_span = d.max_span;
    a.n_reads = d.n_reads; a.n_cpgs = (uint32_t)d.n_cpgs; a.ntiles = ntiles; a.min_depth = p.min_depth; a.min_cpgs = p.min_cpgs;
    a.tile_w = (uint32_t)W; a.min_qual = p.min_qual;
    a.force_sub = getenv("MTH_MHL_FORCE_SUB") ? 1 : 0; a.force_hand_on = getenv("MTH_MHL_FORCE_HAND_ON") ? 1 : 0;
    a.scratch = reinterpret_cast<MwRec *>(ctx->scratch.p); a.rows_per_tile = rows_per_tile;
    a.tile_cnt = ctx->tile_cnt.as<uint32_t>(); a.bucket = ctx->tile_bucket.as<unsigned long long>(); a.st = ctx->d_state;
    const uint32_t grid = ((ntiles + 7) / 8) * 8;
    {
        LaunchTimer lt(ctx, K_MHLTILE);
        hipLaunchKernelGGL(k_mhl_wtile, dim3(grid), dim3(64), 0, s, a);
    }
    {
        LaunchTimer lt(ctx, K_GATHER);
        unsigned long long *bucket_pre = ctx->tile_bucket.as<unsigned long long>() + nbk;
        hipLaunchKernelGGL(k_fw_bucket_scan, dim3(1), dim3(1024), 0, s, ctx->tile_bucket.as<unsigned long long>(), bucket_pre, nbk);
        hipLaunchKernelGGL(k_mhl_wtile_gather, dim3((ntiles + MWG_WAVES * 8 - 1) / (MWG_WAVES * 8)), dim3(64 * MWG_WAVES), 0, s,
                           reinterpret_cast<const MwRec *>(ctx->scratch.p), ctx->tile_cnt.as<uint32_t>(), bucket_pre, ntiles, rows_per_tile,
                           ctx->d_state2, ctx->s_pos.as<int32_t>(), ctx->w_val.as<float>(), ctx->w_cov.as<uint32_t>(), ctx->w_flags.as<uint32_t>(),
                           ctx->w_aux.as<uint32_t>());
    }
    return MTH_OK;
}

}  // namespace mth
