// mth_fdrp_wtile.hip -- FDRP + qFDRP (fdrp.rs:176-246, 51-145; qfdrp.rs:188-258, 109-157) at WGBS depth as ONE tile pass (round 6).
//
// Rounds 2-5 ran this depth class (about a dozen candidate reads per site) as four launches: a full PDR-style site discovery
// (k_pdr_lpmd_tile into a sink + its gather), k_fdrp_walk4 (four sites per wave, every site re-loading its candidate reads from
// HBM and re-deriving their call masks), k_fdrp_walk for what that handed back, and the emit.  Here ONE WAVE owns a tile of
// ~1 500 reference positions (the width follows the batch's read and call density: launch_fdrp_wtile) and keeps everything the
// tile's sites need in 5 KB of LDS -- no workgroup barrier anywhere, 32 waves per CU:
//   A1 every candidate read once (one per lane, two 64-read chunks): fields, the pass test (mapq, >= 1 CpG: fdrp.rs:205-210),
//      its row's start | end, an owner mark at its first call's slot.
//   A2 every call once (one per lane, coalesced): its read = the latest owner mark at or before it (a wave max-scan); the calls
//      of the passing reads set bits in a position bitmap of the tile + max_span on either side -- the tile's SITES.
//   B  prefix popcount over the bitmap: a position's RANK among the window's sites; the core sites' positions.
//   C1 per call: its rank -> the read's three 32-bit masks (calls, covered calls, methylated covered calls; bit = rank mod 32: a
//      stored read spans <= 16 ranks here -- else its sites are handed back -- so two reads that share a site never alias) and
//      the site's reader count, all by LDS atomics.
//   C2 per read: the finished row; the flush rule (fdrp.rs:212-223) as an exclusive prefix maximum of the passing reads'
//      first-call ranks: a reader that has a passing read with a first call beyond the site before it may sit behind a flush
//      -> the site is handed back.
//   D1 per site that can produce a row (>= min_depth readers): its readers IN FILE ORDER as a list of row offsets (ballots per
//      chunk), then every pair of them, lane = pair in the reference's loop order (fdrp.rs:128-141; index -> (i, j) from a
//      table, the next site's entries requested ahead): overlap bases, shared calls, Hamming count from the two rows; the
//      non-zero terms as one byte each (ncpg (ncpg + 1) / 2 + ham), packed in that order.
//   D2 one lane per site: the ordered f32 sum over its terms (qfdrp.rs:152; quotients from a table filled by the same
//      division), the discordant count, fdrp.rs:143 / qfdrp.rs:155.
// Rows go to the tile's scratch slice, sorted by position (ranks ascend); k_fdrp_wtile_gather packs the slices into the
// candidate-site arrays the rest of the FDRP pipeline works on (k_fdrp_walk for the handed-back sites -- flag 4 -- and the
// emit).  No discovery launch, no per-site global loads, no hand-back of merely deep sites (a site holds up to 64 reads here).
//
// Only the common shape is computed here; everything else is handed back and stays bit-identical (k_fdrp_walk, call-by-call path):
// a reader with > 16 calls or spanning > 16 window sites, more readers than min(max_depth, 64) (reservoir: fdrp.rs:87-94), a
// possible flush between two readers, a site whose terms do not fit the term array.  Spans > 200 bp never come here (host side;
// add_read's window test, fdrp.rs:58-63).  A stretch with more candidate reads / calls / sites than the LDS arrays hold is redone
// in halves (down to 192 positions; beyond that every site of the stretch is handed back).
// How it got here, step by step with counters: profiles/r06_wtile_pmc.md.
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "mth_ctx.h"
#include "mth_scan.h"
#include "mth_tile_dev.h"
#include "mth_wave_tile.h"

namespace mth {

struct FdRec { int32_t pos; float f, q; uint32_t nf; };   // nf: stored reads | flag << 24 (1: a row; 4: handed back)
static_assert(sizeof(FdRec) == sizeof(SiteRec), "the PDR scratch buffer is reused");

struct FwArgs {
    const int32_t  *read_start, *read_end;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    int32_t region_beg, region_end, idx_base, max_span, min_overlap;
    uint32_t n_reads, n_cpgs, ntiles, min_depth, max_depth;
    uint32_t tile_w;              // positions per tile (a multiple of 64, <= FW_WMAX)
    uint8_t min_qual;
    uint8_t force_sub;            // tests: start every tile with 192-position stretches
    uint8_t force_heavy;          // tests: every stretch takes the count-only path (all sites handed back)
    FdRec *scratch;               // rows_per_tile rows per tile
    uint32_t rows_per_tile;
    uint32_t *tile_cnt;
    unsigned long long *bucket;   // rows per 256 tiles
    DevState *st;
    const uint16_t *pair_tab;     // (i | j << 8) of the k-th pair of n reads, reference loop order, at [n (n - 1) (n - 2) / 6 + k], n <= 64
    const float *quot;            // [ham * 9 + ncpg] = (float)ham / (float)ncpg, the division done once on the device (k_fw_quot)
    unsigned long long *trace;    // -DMTH_FW_TRACE builds: cycles per phase of each tile's first stretch (lane 0)
};

#ifndef MTH_FW_U
#define MTH_FW_U 2
#endif
constexpr int FW_U = MTH_FW_U;                          // 64-read chunks of a stretch
constexpr int FW_WPT = FW_U > 2 ? 2 : 1;                // bitmap words per lane
constexpr int FW_WMAX = FW_WPT * 2048 - 448;            // the tile + 200 on either side fit the bitmap (1600 / 3648 positions)
constexpr int FW_RCAP = 64 * FW_U;                      // candidate reads of a stretch (more: the stretch is halved)
constexpr int FW_V = FW_U > 2 ? 6 : 4;                  // 64-call chunks of a stretch
constexpr int FW_CCAP = 64 * FW_V;                      // calls of a stretch's candidate reads (more: the stretch is halved)
constexpr int FW_SC = 32;                               // core sites of a stretch (one per lane; a denser stretch is halved)
constexpr int FW_LCAP = 64;                             // stored reads of a site here (more: handed back)
constexpr int FW_NZCAP = 1024;                           // non-zero qFDRP terms of a stretch's sites (more: the site is handed back)
constexpr int FW_QC = 17 * 18 / 2;                      // codes ncpg (ncpg + 1) / 2 + ham, ham <= ncpg <= 16 (a reader's calls span <= 16 window sites)
constexpr uint32_t FW_PASS = 1u, FW_BAD = 2u;           // per-read flags

__global__ void k_fw_quot(float *q) {   // ham / ncpg by the same f32 division the reference does (qfdrp.rs:152); ncpg = 0 is never looked up
    const uint32_t t = threadIdx.x;
    uint32_t ncpg = 0;
    while ((ncpg + 1u) * (ncpg + 2u) / 2u <= t) ++ncpg;
    if (t < (uint32_t)FW_QC) q[t] = ncpg ? (float)(t - ncpg * (ncpg + 1u) / 2u) / (float)ncpg : 0.0f;
}

// every pair of the n readers in s_list, lane = pair in the reference's loop order (fdrp.rs:128-141): the discordant pairs (returned) and,
// appended to s_nz at nzbase in that order, their terms ham / ncpg (qfdrp.rs:152) -- the pairs with a zero term add +0.0 and are not kept
__device__ __forceinline__ uint32_t fw_rounds(const uint16_t *s_list, const uint32_t *s_row, uint8_t *s_nz, const uint16_t *__restrict__ pair_tab,
                                              const int lane, const uint32_t n, uint32_t ent_next, uint32_t &nzbase, const bool mo_any, const int32_t mo_m1) {
    const uint32_t P = n * (n - 1u) / 2u;
    const uint16_t *const tab = pair_tab + (n * (n - 1u) * (n - 2u)) / 6u;              // (n <= 1: no pair, the table is not read)
    uint32_t disc = 0;
    for (uint32_t k0p = 0; k0p < P; k0p += 64u) {
        const uint32_t ent = ent_next;
        if (k0p + 64u < P) ent_next = tab[min(k0p + 64u + (uint32_t)lane, P - 1u)];
        const uint32_t li = s_list[ent & 0xffu], lj = s_list[ent >> 8];
        const uint4 ri = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(s_row) + li), rj = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(s_row) + lj);
        // get_num_overlap_bases (fdrp.rs:97-107) = max(min(end) - max(start) + 1, 0) >= min_overlap (fdrp.rs:134)
        const int32_t ovl = (int32_t)min(ri.x >> 16, rj.x >> 16) - (int32_t)max(ri.x & 0xffffu, rj.x & 0xffffu);
        const bool pair_ok = (k0p + (uint32_t)lane < P) && (mo_any || ovl >= mo_m1);
        const uint32_t hm = ri.z & rj.z & (ri.w ^ rj.w);                             // both cover and call, states differ: fdrp.rs:114-115
        const bool dsc = pair_ok && hm != 0u;                                        // fdrp.rs:138-140; exactly the pairs with a non-zero term
        const unsigned long long nzb = fw_ballot(dsc);
        if (nzb == 0ull) continue;                                                   // wave-uniform
        // the term ham / ncpg (qfdrp.rs:109-131, 152) as one byte, ncpg (ncpg + 1) / 2 + ham: D2 looks the quotient up
        const uint32_t ham = (uint32_t)__builtin_popcount(hm), ncpg = (uint32_t)__builtin_popcount(ri.y & rj.y);
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzb, nzbase));
        if (dsc && slot < (uint32_t)FW_NZCAP) s_nz[slot] = (uint8_t)(ncpg * (ncpg + 1u) / 2u + ham);
        const uint32_t c = (uint32_t)__popcll(nzb);
        disc += c; nzbase += c;
    }
    return disc;
}

// ONE WAVE per tile (~1500 positions, ~110 candidate reads at WGBS depth): no workgroup barrier anywhere, every phase's round
// trips are covered by the other waves of the SIMD (the workgroup form -- 256 threads on 4096 positions, a dozen barriers,
// single-wave phases -- took 36 k cycles a tile with five tiles in flight per CU: 0.88 ms on a chr1-sized contig against the
// four-launch path's 0.615).  Because one wave sees the reads in file order, a site's reader list is built in ONE ordered pass
// (slot = readers so far + ballot prefix), and the non-zero terms of a site's pairs are packed in the reference's loop order
// as they are made -- D2 adds ~20 terms a site, not ~66 codes.  The kernel is bound by vector-instruction issue (a wave64
// instruction occupies its SIMD for four cycles): what counts is instructions per read and per pair round.
__global__ __launch_bounds__(64, 8) void k_fdrp_wtile(const FwArgs a) {
    constexpr int U = FW_U;
    // 5 KB of LDS per wave: 32 waves per CU (the kernel is bound by each wave's own chain of round trips: what counts is waves in flight)
    __shared__ uint2 s_bp[64 * FW_WPT];                       // {site bits, sites in the words before}; lane l owns words FW_WPT l ..
    __shared__ __attribute__((aligned(16))) uint32_t s_row[FW_RCAP * 4];   // {start | end << 16, mC, first-call rank -> mA, mM}
    __shared__ uint16_t s_list[FW_LCAP];                      // D1: the readers of the site in hand (byte offsets of their rows, file order)
    __shared__ float s_quot[FW_QC];                           // [ncpg (ncpg + 1) / 2 + ham] = ham / ncpg (a division is ten vector instructions)
    __shared__ __attribute__((aligned(4))) uint8_t s_nz[FW_NZCAP];   // D: the sites' non-zero terms as codes, each site's in the reference's loop order
    uint8_t *const s_owner = s_nz;                            // A: call slot -> read slot + 1 where a read's calls begin, else 0
    __shared__ int32_t s_cpos[FW_SC];
    __shared__ uint32_t s_sflag[FW_SC];                       // per core site: passing reads that call it | bit 31: one of them spans > 16 window sites
    static_assert(FW_NZCAP >= FW_CCAP, "the owner marks share the term array");
    const int lane = threadIdx.x;
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (a.ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= a.ntiles) return;
    for (int q = lane; q < FW_QC; q += 64) s_quot[q] = a.quot[q];
    const int32_t T0 = a.region_beg + (int32_t)(t * a.tile_w);
    const int32_t T1 = (int32_t)min((int64_t)T0 + a.tile_w, (int64_t)a.region_end);
    FdRec *__restrict__ out = a.scratch + (size_t)t * a.rows_per_tile;
    const uint32_t cap = min(a.max_depth, (uint32_t)FW_LCAP);          // stored reads a site may hold here (fdrp.rs:81-85)
    const uint32_t mind = max(a.min_depth, 1u);                         // fdrp.rs:239-243: an entry exists and holds >= min_depth reads
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t rows_out = 0, bad = 0;
    uint32_t sub_w = a.force_sub ? 192u : a.tile_w;                     // stretch width (wave-uniform): halved when a stretch does not fit
    bool heavy_redo = false;                                            // wave-uniform
#ifdef MTH_FW_TRACE
    unsigned long long tk[12];
    for (int k = 0; k < 12; ++k) tk[k] = 0;
    tk[11] = __builtin_readcyclecounter();
#define FW_TK(k) do { tk[k] = __builtin_readcyclecounter(); } while (0)
#else
#define FW_TK(k) do {} while (0)
#endif
    for (int64_t P0l = T0; P0l < T1;) {
        const int32_t P0 = (int32_t)P0l;
        const int32_t P1 = (int32_t)min(P0l + (int64_t)sub_w, (int64_t)T1);
        const uint32_t Wp = (uint32_t)(P1 - P0);
        // candidate reads: start in [P0 - max_span + 1, P1] (a reader of c calls c in [start - 1, end]; a flusher between two
        // readers starts between them)
        const uint32_t lo = __builtin_amdgcn_readfirstlane(min(a.idx[((uint32_t)P0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads));
        const uint32_t hi = __builtin_amdgcn_readfirstlane(min(a.idx[(((uint32_t)P1 - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads));
        const uint32_t R = hi - lo;
        if (R > (uint32_t)FW_RCAP && sub_w > 192u) { sub_w = max((sub_w >> 1) & ~31u, 192u); continue; }
        const bool heavy = R > (uint32_t)FW_RCAP || heavy_redo || a.force_heavy;
        const uint32_t wbase = (uint32_t)P0 - (uint32_t)a.max_span - 1u;            // window offset 0
        const uint32_t wbits = Wp + 2u * (uint32_t)a.max_span + 1u;
        const uint32_t c_lo = (uint32_t)a.max_span + 1u, c_hi = c_lo + Wp;          // window offsets of the core [P0, P1)
        FW_SYNC();                                                                  // the previous stretch's LDS is done with
#pragma unroll
        for (int q = 0; q < FW_WPT; ++q) s_bp[lane * FW_WPT + q] = make_uint2(0u, 0u);
        if (lane < FW_SC) s_sflag[lane] = 0u;
        FW_SYNC();
        FW_TK(0);
        // ---- A1: the candidate reads, one per lane: fields, the pass test, the row's start | end, the owner mark of its first call ----
        // (the calls themselves are taken one per LANE in A2 / C1: a read holds 1.4 calls at WGBS density, eight call slots per read
        // -- the first form of this kernel -- spent most of their instructions on empty slots)
        uint32_t fl[U], ncall[U];
        uint32_t o0[U], o1[U], mq[U];
        int32_t rs[U], re[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                                               // every chunk's fields are requested at once, with the call range's end
            const uint32_t r = (uint32_t)(u * 64 + lane);
            const uint32_t i = min(lo + r, hi ? hi - 1u : 0u);
            o0[u] = a.cpg_off[i]; o1[u] = a.cpg_off[i + 1];
            rs[u] = a.read_start[i]; re[u] = a.read_end[i]; mq[u] = a.read_mapq[i];
        }
        const uint32_t olast = __builtin_amdgcn_readfirstlane(a.cpg_off[hi]);
        const uint32_t ofirst = R ? __builtin_amdgcn_readfirstlane(o0[0]) : olast;  // (lane 0 of chunk 0 is read lo)
        const uint32_t C_n = heavy ? 0u : olast - ofirst;                            // calls of the stretch's candidate reads
        if (C_n > (uint32_t)FW_CCAP) {
            if (sub_w > 192u) { sub_w = max((sub_w >> 1) & ~31u, 192u); continue; }
            heavy_redo = true; continue;
        }
        auto mark = [&](const uint32_t w, const uint32_t srel_m1) -> uint32_t {      // a passing read's call: the bitmap bit; window offset
            const uint32_t rel = (w & 0x7fffffffu) - wbase;
            bad |= (rel - srel_m1 > (uint32_t)a.max_span) ? 1u : 0u;                 // every call lies in [start - 1, start - 1 + max_span]
            if (rel < wbits) atomicOr(&s_bp[rel >> 5].x, 1u << (rel & 31u));
            return rel;
        };
        if (!heavy) {
            for (int i = lane; i < FW_CCAP / 4; i += 64) reinterpret_cast<uint32_t *>(s_owner)[i] = 0u;
            FW_SYNC();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = (uint32_t)(u * 64 + lane);
                fl[u] = 0u; ncall[u] = 0u;
                if ((uint32_t)(u * 64) >= R) continue;                              // wave-uniform
                const uint32_t n = r < R ? o1[u] - o0[u] : 0u;
                const bool inr = r < R && (uint32_t)rs[u] - ((uint32_t)P0 - (uint32_t)a.max_span + 1u) <= Wp + (uint32_t)a.max_span - 1u;   // start in [P0 - max_span + 1, P1]
                const bool pass = inr && mq[u] >= (uint32_t)a.min_qual && n > 0u;   // fdrp.rs:205, 208
                if (pass) bad |= ((uint32_t)(re[u] - rs[u]) >= (uint32_t)a.max_span) ? 1u : 0u;   // max_span >= end - start + 1 (include/metheor_hip.h)
                // start | end << 16 as window offsets (a passing read's start offset is >= 2: the word is non-zero exactly for them)
                const uint32_t se = pass ? ((((uint32_t)rs[u] - wbase) & 0xffffu) | ((((uint32_t)re[u] - wbase) & 0xffffu) << 16)) : 0u;
                *reinterpret_cast<uint4 *>(&s_row[r * 4]) = make_uint4(se, 0u, 0u, 0u);
                if (n) s_owner[o0[u] - ofirst] = (uint8_t)(r + 1u);                 // the read's first call (offsets < C_n <= FW_CCAP)
                fl[u] = pass ? FW_PASS : 0u;
                ncall[u] = n;
            }
        } else {
            // count-only: more candidate reads / calls than the arrays hold -- the sites are found, every one of them is handed back
            for (uint32_t r = (uint32_t)lane; r < R; r += 64) {
                const uint32_t i = lo + r;
                const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
                const int32_t rs = a.read_start[i];
                const bool pass = (int64_t)rs >= (int64_t)P0 - a.max_span + 1 && rs <= P1 && a.read_mapq[i] >= a.min_qual;
                if (pass) for (uint32_t k = o0; k < o1; ++k) mark(a.cpg_pos[k], (uint32_t)rs - wbase - 1u);
            }
        }
        FW_SYNC();
        // ---- A2: the calls, one per lane (coalesced): its read (the latest owner mark at or before it), the bitmap bit ----
        uint32_t cwv[FW_V], cown[FW_V];                                              // the call word; read slot + 1 | head << 8 | passing << 9
        {
            uint32_t own_carry = 0;
#pragma unroll
            for (int v = 0; v < FW_V; ++v) {
                cwv[v] = 0u; cown[v] = 0u;
                if ((uint32_t)(v * 64) >= C_n) continue;                            // wave-uniform
                const uint32_t c = (uint32_t)(v * 64 + lane);
                const bool valid = c < C_n;
                const uint32_t w = a.cpg_pos[ofirst + (valid ? c : 0u)];
                const uint32_t own = valid ? (uint32_t)s_owner[c] : 0u;
                const uint32_t r1 = max(fw_wave_scan_max_incl(own), own_carry);
                own_carry = (uint32_t)__builtin_amdgcn_readlane(r1, 63);
                const uint32_t se = valid ? s_row[(r1 - 1u) * 4u] : 0u;             // (every call has an owner: r1 >= 1)
                cwv[v] = w;
                cown[v] = valid ? (r1 | (own ? 1u << 8 : 0u) | (se ? 1u << 9 : 0u)) : 0u;
                if (se) mark(w, (se & 0xffffu) - 1u);
            }
        }
        FW_TK(1);
        FW_SYNC();
        // ---- B: ranks ----
        uint32_t wb[FW_WPT], wcnt = 0;
#pragma unroll
        for (int q = 0; q < FW_WPT; ++q) { wb[q] = s_bp[lane * FW_WPT + q].x; wcnt += (uint32_t)__builtin_popcount(wb[q]); }
        uint32_t pre = wave_scan_incl(wcnt) - wcnt;
        const uint32_t pre0 = pre;
#pragma unroll
        for (int q = 0; q < FW_WPT; ++q) { s_bp[lane * FW_WPT + q].y = pre; pre += (uint32_t)__builtin_popcount(wb[q]); }
        pre = pre0;
        FW_SYNC();
        auto rank_of = [&](const uint32_t rel) {                                     // sites of the window below offset rel
            const uint2 e = s_bp[rel >> 5];
            return e.y + (uint32_t)__builtin_popcount(e.x & ((1u << (rel & 31u)) - 1u));
        };
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(rank_of(c_lo)), k1 = __builtin_amdgcn_readfirstlane(rank_of(c_hi));
        const uint32_t ncore = k1 - k0;
        if (ncore > (uint32_t)FW_SC) { sub_w = max((min(sub_w, Wp) >> 1) & ~15u, 48u); continue; }   // (48 positions hold <= 24 sites)
        // core site positions, rank order
#pragma unroll
        for (int q = 0; q < FW_WPT; ++q) {
            uint32_t bits = wb[q];
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
                const uint32_t rel = (uint32_t)(lane * FW_WPT + q) * 32u + b;
                if (rel >= c_lo && rel < c_hi) s_cpos[pre - k0] = (int32_t)(wbase + rel);
                ++pre;
            }
        }
        FW_TK(2);
        if (heavy) {
            FW_SYNC();
            if ((uint32_t)lane < ncore) {
                FdRec rec; rec.pos = s_cpos[lane]; rec.f = 0.0f; rec.q = 0.0f; rec.nf = 4u << 24;
                if (rows_out + lane < a.rows_per_tile) out[rows_out + lane] = rec;
            }
            rows_out += ncore;
            heavy_redo = false;
            P0l = P1;
            continue;
        }
        // ---- C1: per call: its rank -> the read's masks (bit = rank mod 32), first-call rank, highest rank ----
#pragma unroll
        for (int v = 0; v < FW_V; ++v) {
            if ((uint32_t)(v * 64) >= C_n) continue;                                // wave-uniform
            if ((cown[v] >> 9) & 1u) {
                const uint32_t r = (cown[v] & 0xffu) - 1u;
                const uint32_t rel = (cwv[v] & 0x7fffffffu) - wbase;
                const uint32_t rk = rank_of(min(rel, wbits - 1u));
                const uint32_t bit = 1u << (rk & 31u);
                atomicOr(&s_row[r * 4 + 1], bit);
                if (rk - k0 < ncore) atomicAdd(&s_sflag[rk - k0], 1u);                // a core site's passing readers (fdrp.rs:226-231)
                if (cwv[v] >> 31) atomicOr(&s_row[r * 4 + 3], bit);
                // the read's first call: its rank; the one call that can lie outside the covered bases, at start - 1 (readutil.rs:332-340)
                if ((cown[v] >> 8) & 1u) s_row[r * 4 + 2] = rk | ((rel < (s_row[r * 4] & 0xffffu)) ? 0x80000000u : 0u);
            }
        }
        FW_SYNC();
        // ---- C2: per read: the finished row, "its sites are handed back", the flush rule's prefix maximum ----
        uint32_t mCr[U], r0r[U], pmr[U];                                             // kept for D1: call mask (0: not a reader of any list), first-call rank, ...
        uint32_t carry = 0;                                                          // first-call rank + 1, maximum over the chunks before
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mCr[u] = 0u; r0r[u] = 0u; pmr[u] = 0u;
            if ((uint32_t)(u * 64) >= R) continue;                                  // wave-uniform
            const uint32_t r = (uint32_t)(u * 64 + lane);
            uint32_t mC = 0, r0 = 0;
            if (fl[u] & FW_PASS) {
                const uint4 row = *reinterpret_cast<const uint4 *>(&s_row[r * 4]);
                mC = row.y; r0 = row.z & 0x7fffffffu;
                // its highest rank from the mask (bit = rank mod 32): exact while its ranks span < 32 -- and if they do not, two of its
                // calls may share a bit: fewer bits than calls
                const uint32_t span = 31u - (uint32_t)__builtin_clz(__builtin_amdgcn_alignbit(mC, mC, r0 & 31u) | 1u);
                if (span > 15u || (uint32_t)__builtin_popcount(mC) != ncall[u]) fl[u] |= FW_BAD;
                const uint32_t mA = (row.z >> 31) ? mC & ~(1u << (r0 & 31u)) : mC;
                const uint32_t mM = row.w & mA;
                if (fl[u] & FW_BAD) {
                    // (> 16 window sites between its first and last call): every core site it calls is handed back
                    const uint32_t q0 = a.cpg_off[lo + r];
                    for (uint32_t k = q0; k < q0 + ncall[u]; ++k) {
                        const uint32_t rel = (a.cpg_pos[k] & 0x7fffffffu) - wbase;
                        if (rel < wbits) { const uint32_t q = rank_of(rel) - k0; if (q < ncore) atomicOr(&s_sflag[q], 0x80000000u); }
                    }
                    mC = 0u;                                                         // not a reader of any list
                }
                *reinterpret_cast<uint4 *>(&s_row[r * 4]) = make_uint4(row.x, mC, mA, mM);
            }
            // first-call rank + 1 of the passing reads before this one, maximum (file order)
            const uint32_t fc = (fl[u] & FW_PASS) ? r0 + 1u : 0u;
            const uint32_t incl = fw_wave_scan_max_incl(fc);
            pmr[u] = max(MTH_DPP(incl, 0x138 /*wave_shr:1*/, 0xf, true), carry);
            carry = max(carry, (uint32_t)__builtin_amdgcn_readlane(incl, 63));
            mCr[u] = mC; r0r[u] = r0;
        }
        FW_TK(3);
        FW_SYNC();
        // ---- D1: per site that can produce a row (wave-uniform loop): its readers in file order, then every pair of them, lane = pair
        // in the reference's loop order ----
        const uint32_t sf = s_sflag[lane & (FW_SC - 1)];
        const uint32_t n_s = sf & 0x7fffffffu;                                       // lane = core site: reads that call it and pass (C1)
        const bool in_core = (uint32_t)lane < ncore;
        // handed back: a reader spanning > 16 window sites; reservoir (fdrp.rs:87-94) / more reads than this kernel's list holds
        bool redo = in_core && ((sf >> 31) || (n_s > cap && n_s >= mind));
        const bool act = in_core && n_s >= mind && n_s <= cap && !redo;
        uint32_t disc_vec = 0, nzoff_vec = 0, nzcnt_vec = 0, n_seg = 0;              // n_seg: stored reads when the readers form several segments
        bool no_row = false;
        const bool mo_any = a.min_overlap <= 0;                                      // every pair overlaps enough
        const int32_t mo_m1 = a.min_overlap - 1;
        uint32_t nzbase = 0;                                                         // wave-uniform
        unsigned long long todo = fw_ballot(act);
        // the k-th pair of n reads in the reference's loop order comes from a table (a.pair_tab; the closed form is a square root and two
        // corrections per round); a site's first entries are requested while the site before it is in hand
        auto first_ent = [&](const unsigned long long m) -> uint32_t {
            if (!m) return 0u;
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane(n_s, (uint32_t)__builtin_ctzll(m));
            const uint32_t P = n * (n - 1u) / 2u;
            return P ? a.pair_tab[(n * (n - 1u) * (n - 2u)) / 6u + min((uint32_t)lane, P - 1u)] : 0u;
        };
        uint32_t ent_site = first_ent(todo);
        while (todo) {
            const uint32_t q = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t ent_first = ent_site;
            ent_site = first_ent(todo);
            const uint32_t rq = k0 + q;
            FW_SYNC();                                                               // the site before this one is done with the list
            // the site's readers in file order: slot = readers in the chunks before + in the lanes below
            uint32_t n = 0;
            bool flush = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((uint32_t)(u * 64) >= R) continue;                              // wave-uniform
                const bool has = rq - r0r[u] < 16u && ((mCr[u] >> (rq & 31u)) & 1u);
                const unsigned long long b = fw_ballot(has);
                // a passing read before one of them whose first call lies beyond the site (fdrp.rs:212): the readers may form two segments
                flush = flush || fw_ballot(has && pmr[u] > rq + 1u) != 0ull;
                if (has) s_list[min(__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, n)), (uint32_t)FW_LCAP - 1u)] = (uint16_t)((u * 64 + lane) * 16);
                n += (uint32_t)__popcll(b);
            }
#ifndef MTH_FW_FLUSH_EXACT
            // (the segments can be told apart here -- -DMTH_FW_FLUSH_EXACT, parity-green: the kernel then takes 8 % longer for EVERY site, whether
            // the branch's code sits here, in a loop of its own behind this one, or out of line (15 %); a site per thousand is not worth it)
            if (flush) { if ((uint32_t)lane == q) redo = true; continue; }
#else
            static_assert(FW_U == 2, "the exact flush branch is written for two chunks of reads");
            if (flush) {
                // (rare: a few sites per thousand)  The flush rule, exactly (fdrp.rs:212-223): a passing read whose first call lies beyond the
                // site closes the open segment; two readers share a segment iff no such read lies between them.  F = flushers before a
                // reader (file order; a read's first-call rank is 0 unless it passes); the LAST segment that holds >= min_depth readers is
                // the site's row (the map's insert overwrites).  Everything is recomputed from the per-read registers the loop holds anyway:
                // keeping two more alive across the round loop cost 10 % for every site, a call out of line 15 %.
                auto seg_of = [&](const int u, const uint32_t before) {              // F + 1 of the chunk's readers, 0 for the other lanes
                    const bool has = rq - r0r[u] < 16u && ((mCr[u] >> (rq & 31u)) & 1u);
                    const unsigned long long fm = fw_ballot(r0r[u] > rq);
                    return has ? __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, before)) + 1u : 0u;
                };
                const uint32_t f0 = (uint32_t)__popcll(fw_ballot(r0r[0] > rq));       // flushers of the first chunk
                uint32_t top = max(fw_wave_max(seg_of(0, 0u)), fw_wave_max(seg_of(1, f0)));
                n = 0;
                while (top) {                                                        // top - 1: the segment looked at (wave-uniform)
                    const uint32_t sg0 = seg_of(0, 0u), sg1 = seg_of(1, f0);
                    const uint32_t c = (uint32_t)__popcll(fw_ballot(sg0 == top)) + (uint32_t)__popcll(fw_ballot(sg1 == top));
                    if (c >= mind) { n = c; break; }
                    top = max(fw_wave_max(sg0 < top ? sg0 : 0u), fw_wave_max(sg1 < top ? sg1 : 0u));
                }
                if (n == 0u) { if ((uint32_t)lane == q) no_row = true; continue; }   // no segment reaches min_depth (fdrp.rs:239-243)
                FW_SYNC();
                const bool h0 = seg_of(0, 0u) == top, h1 = seg_of(1, f0) == top;
                const unsigned long long b0 = fw_ballot(h0), b1 = fw_ballot(h1);
                if (h0) s_list[min(__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)), (uint32_t)FW_LCAP - 1u)] = (uint16_t)(lane * 16);
                if (h1) s_list[min(__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, (uint32_t)__popcll(b0))), (uint32_t)FW_LCAP - 1u)] = (uint16_t)((64 + lane) * 16);
                if ((uint32_t)lane == q) n_seg = n;
            }
#endif
            FW_SYNC();
            const uint32_t nz0 = nzbase;
            // (the table entries were requested for n_s readers: another n after a flush)
            const uint32_t Pn = n * (n - 1u) / 2u;
            const uint32_t ent0 = !flush ? ent_first : (Pn ? (uint32_t)a.pair_tab[(n * (n - 1u) * (n - 2u)) / 6u + min((uint32_t)lane, Pn - 1u)] : 0u);
            const uint32_t disc = fw_rounds(s_list, s_row, s_nz, a.pair_tab, lane, n, ent0, nzbase, mo_any, mo_m1);
            if (nzbase > (uint32_t)FW_NZCAP) {                                       // the term array is full: this site goes to the walk
                nzbase = nz0;
                if ((uint32_t)lane == q) redo = true;
            } else if ((uint32_t)lane == q) { disc_vec = disc; nzoff_vec = nz0; nzcnt_vec = nzbase - nz0; }
        }
        FW_TK(4);
        FW_SYNC();
        // ---- D2: one lane per site ----
        const bool emit = (redo || act) && !no_row;
        FdRec rec;
        rec.pos = 0; rec.f = 0.0f; rec.q = 0.0f; rec.nf = 4u << 24;
        if (emit) {
            rec.pos = s_cpos[lane & (FW_SC - 1)];
            if (!redo) {
                float q = 0.0f;
                // qfdrp.rs:152, the reference's order (x + 0.0 == x: the zero terms are not kept); four codes, then their four quotients, per trip
                uint32_t i = 0;
                for (; i + 4u <= nzcnt_vec; i += 4u) {
                    const uint32_t c0 = s_nz[nzoff_vec + i], c1 = s_nz[nzoff_vec + i + 1u], c2 = s_nz[nzoff_vec + i + 2u], c3 = s_nz[nzoff_vec + i + 3u];
                    const float t0 = s_quot[c0], t1 = s_quot[c1], t2 = s_quot[c2], t3 = s_quot[c3];
                    q += t0; q += t1; q += t2; q += t3;
                }
                for (; i < nzcnt_vec; ++i) q += s_quot[s_nz[nzoff_vec + i]];
                // (num_reads * (num_reads - 1)) as f32 / 2.0 in usize arithmetic (fdrp.rs:143)
                const uint32_t nr = n_seg ? n_seg : n_s;
                const unsigned long long prod = (unsigned long long)nr * (unsigned long long)(nr - 1u);
                const float den = (float)prod / 2.0f;
                rec.f = (float)disc_vec / den; rec.q = q / den; rec.nf = nr | (1u << 24);
            }
        }
        const unsigned long long em = fw_ballot(emit);
        if (emit && rows_out + (uint32_t)__popcll(em & lt_mask) < a.rows_per_tile) out[rows_out + (uint32_t)__popcll(em & lt_mask)] = rec;
        rows_out += (uint32_t)__popcll(em);
        FW_TK(5);
#ifdef MTH_FW_TRACE
        if (a.trace && P0 == T0) {   // why sites were handed back: a reader spanning > 16 window sites; more reads than the list holds; a possible flush; terms
            const unsigned long long m1 = fw_ballot(in_core && (sf >> 31)), m2 = fw_ballot(in_core && !(sf >> 31) && n_s > cap && n_s >= mind), m3 = fw_ballot(redo) & ~m1 & ~m2;
            unsigned long long nb = 0;
            for (int u = 0; u < U; ++u) nb += (unsigned long long)__popcll(fw_ballot((fl[u] & FW_BAD) != 0u));
            if (lane == 0) { a.trace[(size_t)t * 12 + 7] = (unsigned long long)__popcll(m1); a.trace[(size_t)t * 12 + 8] = (unsigned long long)__popcll(m2); a.trace[(size_t)t * 12 + 9] = (unsigned long long)__popcll(m3); a.trace[(size_t)t * 12 + 10] = nb; }
        }
        if (lane == 0 && a.trace && P0 == T0) { a.trace[(size_t)t * 12] = 1ull; a.trace[(size_t)t * 12 + 1] = tk[0] - tk[11]; for (int k = 1; k < 6; ++k) a.trace[(size_t)t * 12 + 1 + k] = tk[k] - tk[k - 1]; }
#endif
        P0l = P1;
    }
    if (bad & 1u) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
    if (lane == 0) {
        a.tile_cnt[t] = rows_out;
        if (rows_out) atomicAdd(a.bucket + (t >> TILE_BUCKET_SHIFT), (unsigned long long)rows_out);
    }
}

// exclusive scan of the buckets' row counts (one workgroup; a few thousand buckets)
__global__ __launch_bounds__(1024) void k_fw_bucket_scan(const unsigned long long *__restrict__ bucket, unsigned long long *__restrict__ bucket_pre, const uint32_t nbk) {
    __shared__ unsigned long long part[1024];
    const uint32_t per = (nbk + 1023u) / 1024u, b0 = threadIdx.x * per, b1 = min(b0 + per, nbk);
    unsigned long long sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += bucket[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long acc = 0; for (int i = 0; i < 1024; ++i) { const unsigned long long v = part[i]; part[i] = acc; acc += v; } }
    __syncthreads();
    unsigned long long acc = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) { bucket_pre[b] = acc; acc += bucket[b]; }
}

// Eight tiles per wave, eight lanes each: a tile's first row = rows of the buckets before its bucket + rows of the bucket's
// earlier tiles; its rows go to the candidate-site arrays, the handed-back ones (flag 4) also to the list k_fdrp_walk takes
// them from.  The lanes of the last tile leave the total in sites_st->n_sites.
constexpr int FG_WAVES = 4;
__global__ __launch_bounds__(64 * FG_WAVES) void k_fdrp_wtile_gather(const FdRec *__restrict__ scratch, const uint32_t *__restrict__ tile_cnt,
                                                                     const unsigned long long *__restrict__ bucket_pre, const uint32_t ntiles,
                                                                     const uint32_t rows_per_tile, DevState *__restrict__ sites_st,
                                                                     int32_t *__restrict__ site_pos, float *__restrict__ fdrp, float *__restrict__ qfdrp,
                                                                     uint32_t *__restrict__ nreads, uint32_t *__restrict__ flags,
                                                                     uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_cnt) {
    const int lane = threadIdx.x & 63, sub = lane >> 3, l8 = lane & 7;
    const uint32_t tw = (blockIdx.x * FG_WAVES + (threadIdx.x >> 6)) * 8u;          // the wave's first tile
    if (tw >= ntiles) return;
    // rows of the bucket's tiles before the wave's first tile (tw is a multiple of 8: the wave's tiles share a bucket)
    const uint32_t t_first = (tw >> TILE_BUCKET_SHIFT) << TILE_BUCKET_SHIFT;
    uint32_t in_bucket = 0;
    for (uint32_t q = t_first + lane; q < tw; q += 64) in_bucket += tile_cnt[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) in_bucket += __shfl_xor(in_bucket, o, 64);
    const uint32_t t = tw + (uint32_t)sub;
    const uint32_t n = t < ntiles ? tile_cnt[t] : 0u;
    uint32_t before = 0;                                                             // rows of the wave's tiles before this one
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint32_t nk = __shfl(n, k * 8, 64); if (k < sub) before += nk; }
    const unsigned long long base = bucket_pre[tw >> TILE_BUCKET_SHIFT] + in_bucket + before;
    const uint32_t n_max = max(max(max(__shfl(n, 0, 64), __shfl(n, 8, 64)), max(__shfl(n, 16, 64), __shfl(n, 24, 64))),
                               max(max(__shfl(n, 32, 64), __shfl(n, 40, 64)), max(__shfl(n, 48, 64), __shfl(n, 56, 64))));
    const FdRec *__restrict__ src = scratch + (size_t)min(t, ntiles - 1u) * rows_per_tile;
    for (uint32_t i0 = 0; i0 < n_max; i0 += 8) {
        const uint32_t i = i0 + (uint32_t)l8;
        const bool in = i < n;
        FdRec r;
        r.pos = 0; r.f = 0.0f; r.q = 0.0f; r.nf = 0u;
        if (in) r = src[i];
        const uint32_t fg = r.nf >> 24;
        if (in) { site_pos[base + i] = r.pos; fdrp[base + i] = r.f; qfdrp[base + i] = r.q; nreads[base + i] = r.nf & 0xffffffu; flags[base + i] = fg; }
        const unsigned long long hb = __ballot(in && fg == 4u);
        if (hb) {                                                                    // wave-uniform
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(redo_cnt, (uint32_t)__popcll(hb));
            at = __builtin_amdgcn_readfirstlane(at);
            if (in && fg == 4u) redo_list[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u))] = (uint32_t)(base + i);
        }
    }
    if (t == ntiles - 1 && l8 == 0) sites_st->n_sites = base + n;
}

// The tile pass of one batch: candidate-site arrays (ctx->s_pos, w_val = fdrp, w_aux = qfdrp, w_cov = stored reads, w_flags;
// count in d_state2->n_sites) filled with the finished rows (flag 1) and the handed-back sites (flag 4, listed in redo_list /
// *redo_cnt -- cleared by the caller); the fine read index is left for the walk.
int launch_fdrp_wtile(mth_ctx *ctx, const mth_batch_t &d, const mth_fdrp_params_t &p, const uint16_t *pair_tab, uint32_t *redo_list, uint32_t *redo_cnt) {
    hipStream_t s = ctx->stream;
    const int64_t region_len = (int64_t)d.region_end - d.region_beg;
    if (d.n_reads == 0 || region_len <= 0) return MTH_OK;
    // Tile width: the widest whose candidate reads (start in [P0 - max_span + 1, P1]) fill the stretch's two 64-read chunks without
    // spilling over too often (mean + 2 sigma <= 128: an overfull tile is redone in halves), at most FW_WMAX.
    int W;
    {
        const double rpb = (double)d.n_reads / (double)region_len;
        const double want = (double)FW_RCAP - 2.0 * std::sqrt((double)FW_RCAP);     // ~105 of 128 reads
        W = (int)(want / std::max(rpb, 1e-9)) - d.max_span;
        // ... and their calls the stretch's call lanes (denser calls: a narrower tile rather than a tile redone in halves)
        const double cpr = (double)d.n_cpgs / (double)d.n_reads, wantc = (double)FW_CCAP - 2.0 * std::sqrt((double)FW_CCAP);
        W = std::min(W, (int)(wantc / std::max(rpb * cpr, 1e-9)) - d.max_span);
        W = std::max(256, std::min(FW_WMAX, W)) & ~63;
    }
    if (const char *e = getenv("METHEOR_FDRP_WTILE_W")) W = std::min(FW_WMAX, std::max(64, atoi(e))) & ~63;   // tests / tuning
    int32_t idx_base = 0;
    uint32_t ntiles = 0;
    int rc = build_read_index(ctx, d, W, idx_base, ntiles);
    if (rc) return rc;
    const uint32_t nbk = (ntiles + (1u << TILE_BUCKET_SHIFT) - 1) >> TILE_BUCKET_SHIFT;
    const uint32_t rows_per_tile = (uint32_t)W / 2u;                                 // CpG sites lie at least two positions apart
    MTH_HIP(ctx, ctx->tile_cnt.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->tile_bucket.reserve((size_t)nbk * 5 * sizeof(unsigned long long), s));
    MTH_HIP(ctx, hipMemsetAsync(ctx->tile_bucket.p, 0, (size_t)nbk * sizeof(unsigned long long), s));
    MTH_HIP(ctx, ctx->scratch.reserve((size_t)ntiles * rows_per_tile * sizeof(FdRec), s));
    if (!ctx->f_quot.p) {
        MTH_HIP(ctx, ctx->f_quot.reserve(256 * sizeof(float), s));
        hipLaunchKernelGGL(k_fw_quot, dim3(1), dim3(192), 0, s, ctx->f_quot.as<float>());
    }
    FwArgs a;
    a.read_start = d.read_start; a.read_end = d.read_end; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
    a.idx = idx_ptr(ctx);
    a.region_beg = d.region_beg; a.region_end = d.region_end; a.idx_base = idx_base; a.max_span = d.max_span; a.min_overlap = p.min_overlap;
    a.n_reads = d.n_reads; a.n_cpgs = (uint32_t)d.n_cpgs; a.ntiles = ntiles; a.tile_w = (uint32_t)W;
    a.min_depth = (uint32_t)std::min<uint64_t>(p.min_depth, 0xffffffffull); a.max_depth = p.max_depth; a.min_qual = p.min_qual;
    a.force_sub = getenv("METHEOR_FDRP_WTILE_SUB") ? 1 : 0; a.force_heavy = getenv("METHEOR_FDRP_WTILE_HEAVY") ? 1 : 0;
    a.scratch = reinterpret_cast<FdRec *>(ctx->scratch.p); a.rows_per_tile = rows_per_tile;
    a.tile_cnt = ctx->tile_cnt.as<uint32_t>(); a.bucket = ctx->tile_bucket.as<unsigned long long>(); a.st = ctx->d_state; a.pair_tab = pair_tab;
    a.quot = ctx->f_quot.as<float>();
    const uint32_t grid = ((ntiles + 7) / 8) * 8;
    a.trace = nullptr;
#ifdef MTH_FW_TRACE
    static unsigned long long *d_trace = nullptr;
    static size_t d_trace_n = 0;
    if (d_trace_n < (size_t)ntiles * 12) { if (d_trace) (void)hipFree(d_trace); d_trace_n = (size_t)ntiles * 12; MTH_HIP(ctx, hipMalloc((void **)&d_trace, d_trace_n * 8)); }
    MTH_HIP(ctx, hipMemsetAsync(d_trace, 0, (size_t)ntiles * 96, s));
    a.trace = d_trace;
#endif
    {
        LaunchTimer lt(ctx, K_FDRPWTILE);
        hipLaunchKernelGGL(k_fdrp_wtile, dim3(grid), dim3(64), 0, s, a);
    }
    {
        LaunchTimer lt(ctx, K_GATHER);
        unsigned long long *bucket_pre = ctx->tile_bucket.as<unsigned long long>() + nbk;
        hipLaunchKernelGGL(k_fw_bucket_scan, dim3(1), dim3(1024), 0, s, ctx->tile_bucket.as<unsigned long long>(), bucket_pre, nbk);
        hipLaunchKernelGGL(k_fdrp_wtile_gather, dim3((ntiles + FG_WAVES * 8 - 1) / (FG_WAVES * 8)), dim3(64 * FG_WAVES), 0, s,
                           reinterpret_cast<const FdRec *>(ctx->scratch.p), ctx->tile_cnt.as<uint32_t>(), bucket_pre,
                           ntiles, rows_per_tile, ctx->d_state2, ctx->s_pos.as<int32_t>(), ctx->w_val.as<float>(),
                           reinterpret_cast<float *>(ctx->w_aux.p), ctx->w_cov.as<uint32_t>(), ctx->w_flags.as<uint32_t>(), redo_list, redo_cnt);
    }
#ifdef MTH_FW_TRACE
    {
        std::vector<unsigned long long> hv((size_t)ntiles * 12);
        MTH_HIP(ctx, hipStreamSynchronize(s));
        MTH_HIP(ctx, hipMemcpy(hv.data(), a.trace, hv.size() * 8, hipMemcpyDeviceToHost));
        double h[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t q = 0; q < ntiles; ++q) for (int k = 0; k < 12; ++k) h[k] += (double)hv[q * 12 + k];
        fprintf(stderr, "[fdrp wtile trace] W %d tiles %.0f; cycles of a tile's first stretch: head %.0f  A %.0f  B %.0f  C %.0f  D1 %.0f  D2+rows %.0f\n",
                W, h[0], h[1] / h[0], h[2] / h[0], h[3] / h[0], h[4] / h[0], h[5] / h[0], h[6] / h[0]);
        fprintf(stderr, "[fdrp wtile trace] handed back (first stretches): wide reader %.0f  deep %.0f  flush / terms %.0f;  wide readers %.0f\n", h[7], h[8], h[9], h[10]);
    }
#endif
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

}  // namespace mth
