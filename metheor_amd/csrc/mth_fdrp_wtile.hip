// mth_fdrp_wtile.hip -- FDRP + qFDRP (fdrp.rs:176-246, 51-145; qfdrp.rs:188-258, 109-157) at WGBS depth as ONE tile pass (round 6).
//
// Rounds 2-5 ran this depth class (about a dozen candidate reads per site) as four launches: a full PDR-style site discovery
// (k_pdr_lpmd_tile into a sink + its gather), k_fdrp_walk4 (four sites per wave, every site re-loading its candidate reads from
// HBM and re-deriving their call masks), k_fdrp_walk for what that handed back, and the emit.  Here one workgroup owns a tile of
// 4096 reference positions and keeps everything the tile's sites need in LDS:
//   A  every candidate read once (one per thread): fields, <= 8 calls; the calls of the reads that pass (mapq, >= 1 CpG:
//      fdrp.rs:205-210) set bits in a position bitmap of the tile + max_span on either side -- the tile's SITES -- and are kept
//      as 16-bit window offsets.
//   B  prefix popcount over the bitmap: a position's RANK among the window's sites.
//   C  per read: its calls as ranks -> three 32-bit masks (calls, covered calls, methylated covered calls) with bit = rank mod 32
//      (a stored read spans <= 16 ranks here -- else its sites are handed back -- so two reads that share a site never alias);
//      per site: its readers IN FILE ORDER as a list of read slots (ballots per 64-read chunk: count, prefix over chunks, place);
//      the flush rule (fdrp.rs:212-223) as an exclusive prefix maximum of the passing reads' first-call ranks: a reader that
//      has a passing read with a first call beyond the site before it may sit behind a flush -> the site is handed back.
//   D1 every pair of every site's readers, flat over the workgroup's threads (lane = pair, all lanes busy whatever the sites'
//      depths): overlap bases, shared calls, Hamming count from the two reads' rows -> one byte (ham * 9 + ncpg, 0 = skipped
//      pair) at the pair's LEXICOGRAPHIC index in the site's code list (fdrp.rs:128-141 loop order).
//   D2 one thread per site: the ordered f32 sum over the list (qfdrp.rs:152; quotients from a table filled by the same
//      division), the discordant count, fdrp.rs:143 / qfdrp.rs:155.
// Rows go to the tile's scratch slice, sorted by position (ranks ascend); k_fdrp_wtile_gather packs the slices into the
// candidate-site arrays the rest of the FDRP pipeline works on (k_fdrp_walk for the handed-back sites -- flag 4 -- and the
// emit).  No discovery launch, no per-site global loads, no hand-back of merely deep sites (a site holds up to 64 reads here).
//
// Only the common shape is computed here; everything else is handed back and stays bit-identical (k_fdrp_walk, call-by-call path):
// a reader with > 8 calls or spanning > 16 window sites, more readers than min(max_depth, 64) (reservoir: fdrp.rs:87-94), a
// possible flush between two readers.  Spans > 200 bp never come here (host side; add_read's window test, fdrp.rs:58-63).
// A stretch with more candidate reads / sites / readers / pairs than the LDS arrays hold is redone in halves (down to 256
// positions; beyond that every site of the stretch is handed back).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mth_ctx.h"
#include "mth_scan.h"
#include "mth_tile_dev.h"

namespace mth {

struct FdRec { int32_t pos; float f, q; uint32_t nf; };   // nf: stored reads | flag << 24 (1: a row; 4: handed back)
static_assert(sizeof(FdRec) == sizeof(SiteRec), "the PDR scratch buffer is reused");

struct FwArgs {
    const int32_t  *read_start, *read_end;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    int32_t region_beg, region_end, idx_base, max_span, min_overlap;
    uint32_t n_reads, n_cpgs, ntiles, min_depth, max_depth;
    uint8_t min_qual;
    uint8_t force_sub;            // tests: start every tile with 256-position sub-ranges
    uint8_t force_heavy;          // tests: every stretch takes the count-only path (all sites handed back)
    FdRec *scratch;               // rows_per_tile rows per tile
    uint32_t rows_per_tile;
    uint32_t *tile_cnt;
    unsigned long long *bucket;   // rows per 256 tiles
    DevState *st;
    unsigned long long *trace;    // -DMTH_FW_TRACE builds: cycles per phase of each tile's first stretch (thread 0)
};

constexpr int FW_B = 256;
constexpr int FW_SC = 256;                              // core sites of a stretch (one per thread)
constexpr int FW_LPOOL = 2048;                          // reader-list entries of a stretch
constexpr int FW_TCAP = 8192;                           // pair codes of a stretch
constexpr int FW_NMAX = 64;                             // stored reads of a site
constexpr int FW_CT = FW_NMAX * (FW_NMAX - 1) / 2;      // pairs of 64 reads
constexpr int FW_QN = 9;                                // ncpg <= 8 (a reader holds <= 8 calls)
constexpr uint32_t FW_PASS = 1u, FW_BAD = 2u;           // per-read flags

__device__ __forceinline__ uint32_t fw_wave_max(uint32_t v) {   // wave-uniform result
    v = max(v, MTH_DPP(v, 0xb1 /*quad_perm [1,0,3,2]*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x4e /*quad_perm [2,3,0,1]*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x141 /*row_half_mirror*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x140 /*row_mirror*/, 0xf, true));
    // (readlane returns int: the maxima are taken as unsigned)
    return max(max((uint32_t)__builtin_amdgcn_readlane(v, 0), (uint32_t)__builtin_amdgcn_readlane(v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane(v, 32), (uint32_t)__builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ uint32_t fw_wave_scan_max_incl(uint32_t v) {   // values >= 0; lanes outside a row read 0
    v = max(v, MTH_DPP(v, 0x111 /*row_shr:1*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x112 /*row_shr:2*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x114 /*row_shr:4*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x118 /*row_shr:8*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x142 /*row_bcast:15*/, 0xa, false));
    v = max(v, MTH_DPP(v, 0x143 /*row_bcast:31*/, 0xc, false));
    return v;
}
// exclusive scan over the workgroup's 256 threads (two barriers); total = sum over all threads
__device__ __forceinline__ uint32_t fw_block_scan_excl(const uint32_t v, uint32_t *ws, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl(v);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    const uint32_t w0 = ws[0], w1 = ws[1], w2 = ws[2], w3 = ws[3];
    __syncthreads();
    total = w0 + w1 + w2 + w3;
    const uint32_t base = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
    return base + incl - v;
}

template <int SHIFT, int RCAP>
__global__ __launch_bounds__(FW_B, 4) void k_fdrp_wtile(const FwArgs a) {
    constexpr int W = 1 << SHIFT;
    constexpr int WW = (W + 2 * 200 + 1 + 31) / 32 + 1;       // bitmap words: the tile + max_span (<= 200) on either side
    constexpr int WPT = (WW + FW_B - 1) / FW_B;               // bitmap words per thread
    constexpr int U = RCAP / FW_B;                            // candidate reads per thread
    constexpr int NCH = RCAP / 64;                            // 64-read chunks
    static_assert(RCAP % FW_B == 0 && W + 402 < 32768, "16-bit window offsets with the state in bit 15");
    __shared__ uint2 s_bp[WW];                                // {site bits, sites in the words before}
    __shared__ __attribute__((aligned(16))) uint32_t s_row[RCAP * 4];   // A -> C: 8 calls as 16-bit words; C -> D: {start | end << 16, mC, mA, mM}
    __shared__ uint8_t s_cnt[NCH * FW_SC];                    // readers of a core site in a chunk -> readers in the chunks before
    __shared__ uint16_t s_pool[FW_LPOOL];                     // the sites' reader lists (read slots, file order)
    __shared__ __attribute__((aligned(4))) uint8_t s_code[FW_TCAP];
    __shared__ uint16_t s_ctab[FW_CT];                        // m = j (j - 1) / 2 + i  ->  i | j << 8
    __shared__ int32_t s_cpos[FW_SC];
    __shared__ uint32_t s_sinfo[FW_SC];                       // readers (low 16) | evaluated here << 16 | handed back << 17
    __shared__ uint32_t s_off[FW_SC];                         // list offset | code offset << 16
    __shared__ uint16_t s_hint[FW_TCAP / 64 + 1];             // the site that holds code offset 64 b
    __shared__ float s_quot[FW_QN * FW_QN];
    __shared__ uint32_t s_cmax[NCH], s_ccar[NCH];             // a chunk's maximum first-call rank + 1; the maximum carried into it
    __shared__ uint32_t ws[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of tiles
    const uint32_t per_xcd = (a.ntiles + 7) / 8;
    const uint32_t t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= a.ntiles) return;
    for (int m = tid; m < FW_CT; m += FW_B) {
        int j = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)m)) * 0.5f);
        while (j * (j - 1) / 2 > m) --j;
        while ((j + 1) * j / 2 <= m) ++j;
        s_ctab[m] = (uint16_t)((m - j * (j - 1) / 2) | (j << 8));
    }
    // ham / ncpg by the same f32 division the reference does (qfdrp.rs:152); code 0 (a skipped pair) and ham = 0 add +0.0
    if (tid < FW_QN * FW_QN) s_quot[tid] = tid < FW_QN ? 0.0f : (float)(tid / FW_QN) / (float)(tid % FW_QN);
    const int32_t T0 = a.region_beg + (int32_t)(t * W);
    const int32_t T1 = (int32_t)min((int64_t)T0 + W, (int64_t)a.region_end);
    FdRec *__restrict__ out = a.scratch + (size_t)t * a.rows_per_tile;
    const uint32_t cap = min(a.max_depth, (uint32_t)FW_NMAX);          // stored reads a site may hold here (fdrp.rs:81-85)
    const uint32_t mind = max(a.min_depth, 1u);                         // fdrp.rs:239-243: an entry exists and holds >= min_depth reads
    uint32_t rows_out = 0, bad = 0;
    int sub_shift = a.force_sub ? 8 : SHIFT;
#ifdef MTH_FW_TRACE
    unsigned long long tk[12];
    for (int k = 0; k < 12; ++k) tk[k] = 0;
    tk[11] = __builtin_readcyclecounter();
#define FW_TK(k) do { tk[k] = __builtin_readcyclecounter(); } while (0)
#else
#define FW_TK(k) do {} while (0)
#endif
    bool heavy_redo = false;                                            // block-uniform
    for (int64_t P0l = T0; P0l < T1;) {
        const int32_t P0 = (int32_t)P0l;
        const int32_t P1 = (int32_t)min(P0l + (1ll << sub_shift), (int64_t)T1);
        const uint32_t Wp = (uint32_t)(P1 - P0);
        // candidate reads: start in [P0 - max_span + 1, P1] (a reader of c calls c in [start - 1, end]; a flusher between two
        // readers starts between them)
        const uint32_t lo = min(a.idx[((uint32_t)P0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[(((uint32_t)P1 - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        const uint32_t R = hi - lo;
        if (R > (uint32_t)RCAP && sub_shift > 8) { --sub_shift; continue; }
        const bool heavy = R > (uint32_t)RCAP || heavy_redo || a.force_heavy;
        const uint32_t wbase = (uint32_t)P0 - (uint32_t)a.max_span - 1u;            // window offset 0
        const uint32_t wbits = Wp + 2u * (uint32_t)a.max_span + 1u;
        const uint32_t c_lo = (uint32_t)a.max_span + 1u, c_hi = c_lo + Wp;          // window offsets of the core [P0, P1)
        __syncthreads();                                                            // the previous stretch's LDS is done with
        for (int w = tid; w < WW; w += FW_B) s_bp[w] = make_uint2(0u, 0u);
        for (int i = tid; i < NCH * FW_SC / 4; i += FW_B) reinterpret_cast<uint32_t *>(s_cnt)[i] = 0u;
        s_sinfo[tid] = 0u;
        __syncthreads();
        FW_TK(0);
        // ---- A: the candidate reads ----
        uint32_t se[U], fl[U];
        auto mark = [&](const uint32_t w, const int32_t rs) -> uint32_t {           // a passing read's call: the bitmap bit; window offset
            const uint32_t p = w & 0x7fffffffu, rel = p - wbase;
            bad |= (p - ((uint32_t)rs - 1u) > (uint32_t)a.max_span) ? 1u : 0u;      // every call lies in [start - 1, start - 1 + max_span]
            if (rel < wbits) atomicOr(&s_bp[rel >> 5].x, 1u << (rel & 31u));
            return rel;
        };
        if (!heavy) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = (uint32_t)(u * FW_B + tid);
                se[u] = 0u; fl[u] = 0u;
                if ((uint32_t)(u * FW_B + wave * 64) >= R) continue;                // wave-uniform
                const bool valid = r < R;
                const uint32_t i = lo + (valid ? r : 0u);
                const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
                const int32_t rs = a.read_start[i], re = a.read_end[i];
                const uint32_t mq = a.read_mapq[i];
                const uint32_t n = o1 - o0;
                const bool inr = valid && (int64_t)rs >= (int64_t)P0 - a.max_span + 1 && rs <= P1;
                const bool pass = inr && mq >= (uint32_t)a.min_qual && n > 0u;      // fdrp.rs:205, 208
                uint32_t cw[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) cw[k] = 0u;
                if (__all(!pass || (unsigned long long)o0 + 8u <= (unsigned long long)a.n_cpgs)) {
                    if (pass) {
                        const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0);
                        cw[0] = x.x; cw[1] = x.y; cw[2] = x.z; cw[3] = x.w;
                        if (n > 4u) {
                            const u32x4_a4 y = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0 + 4);
                            cw[4] = y.x; cw[5] = y.y; cw[6] = y.z; cw[7] = y.w;
                        }
                    }
                } else if (pass) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) cw[k] = a.cpg_pos[o0 + min((uint32_t)k, n - 1u)];
                }
                if (!pass) continue;
                bad |= ((uint32_t)(re - rs) >= (uint32_t)a.max_span) ? 1u : 0u;     // max_span >= end - start + 1 (include/metheor_hip.h)
                uint32_t c16[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w = (uint32_t)k < n ? cw[k] : cw[0];            // slots past the last call repeat the first (same bit)
                    const uint32_t rel = (uint32_t)k < n ? mark(w, rs) : (cw[0] & 0x7fffffffu) - wbase;
                    c16[k] = (rel & 0x7fffu) | ((w >> 31) << 15);
                }
                uint32_t f = FW_PASS;
                if (n > 8u) {                                                       // its sites are handed back; their bits are still needed
                    f |= FW_BAD;
                    for (uint32_t k = 8; k < n; ++k) mark(a.cpg_pos[o0 + k], rs);
                }
                fl[u] = f;
                se[u] = (((uint32_t)rs - wbase) & 0xffffu) | ((((uint32_t)re - wbase) & 0xffffu) << 16);
                *reinterpret_cast<uint4 *>(&s_row[r * 4]) = make_uint4(c16[0] | (c16[1] << 16), c16[2] | (c16[3] << 16), c16[4] | (c16[5] << 16), c16[6] | (c16[7] << 16));
            }
        } else {
            // count-only: more candidate reads (or, at 256 positions, more readers / pairs) than the arrays hold -- the sites are
            // found, every one of them is handed back
            for (uint32_t r = (uint32_t)tid; r < R; r += FW_B) {
                const uint32_t i = lo + r;
                const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
                const int32_t rs = a.read_start[i];
                const bool pass = (int64_t)rs >= (int64_t)P0 - a.max_span + 1 && rs <= P1 && a.read_mapq[i] >= a.min_qual;
                if (pass) for (uint32_t k = o0; k < o1; ++k) mark(a.cpg_pos[k], rs);
            }
        }
        FW_TK(1);
        __syncthreads();
        FW_TK(2);
        // ---- B: ranks ----
        uint32_t wsum = 0, wb[WPT];
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int w = tid * WPT + q;
            wb[q] = w < WW ? s_bp[w].x : 0u;
            wsum += (uint32_t)__builtin_popcount(wb[q]);
        }
        uint32_t n_win;
        uint32_t pre = fw_block_scan_excl(wsum, ws, n_win);
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int w = tid * WPT + q;
            if (w < WW) s_bp[w].y = pre;
            pre += (uint32_t)__builtin_popcount(wb[q]);
        }
        __syncthreads();
        auto rank_of = [&](const uint32_t rel) {                                     // sites of the window below offset rel
            const uint2 e = s_bp[rel >> 5];
            return e.y + (uint32_t)__builtin_popcount(e.x & ((1u << (rel & 31u)) - 1u));
        };
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(rank_of(c_lo)), k1 = __builtin_amdgcn_readfirstlane(rank_of(c_hi));
        const uint32_t ncore = k1 - k0;
        if (ncore > (uint32_t)FW_SC) {                                               // (256 positions hold <= 128 sites)
            if (sub_shift > 8) { --sub_shift; continue; }
            bad |= 2u;
        }
        // core site positions, rank order
        pre = s_bp[min(tid * WPT, WW - 1)].y;
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int w = tid * WPT + q;
            uint32_t bits = wb[q];
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
                const uint32_t rel = (uint32_t)w * 32u + b;
                if (rel >= c_lo && rel < c_hi && pre - k0 < (uint32_t)FW_SC) s_cpos[pre - k0] = (int32_t)(wbase + rel);
                ++pre;
            }
        }
        if (heavy) {
            __syncthreads();
            for (uint32_t k = (uint32_t)tid; k < min(ncore, (uint32_t)FW_SC); k += FW_B) {
                FdRec rec; rec.pos = s_cpos[k]; rec.f = 0.0f; rec.q = 0.0f; rec.nf = 4u << 24;
                if (rows_out + k < a.rows_per_tile) out[rows_out + k] = rec;
            }
            rows_out += min(ncore, (uint32_t)FW_SC);
            heavy_redo = false;
            P0l = P1;
            continue;
        }
        FW_TK(3);
        // ---- C1: ranks of the reads' calls, masks, readers per chunk and site, prefix maximum of the first-call ranks ----
        uint32_t mC[U], r0[U], pmx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = (uint32_t)(u * FW_B + tid);
            mC[u] = 0u; r0[u] = 0u; pmx[u] = 0u;
            if ((uint32_t)(u * FW_B + wave * 64) >= R) continue;                    // wave-uniform
            const uint32_t chunk = (uint32_t)(u * (FW_B / 64) + wave);
            uint32_t rmax = 0;
            if (fl[u] & FW_PASS) {
                const uint4 cv = *reinterpret_cast<const uint4 *>(&s_row[r * 4]);
                const uint32_t c16[8] = {cv.x & 0xffffu, cv.x >> 16, cv.y & 0xffffu, cv.y >> 16, cv.z & 0xffffu, cv.z >> 16, cv.w & 0xffffu, cv.w >> 16};
                uint32_t mM = 0, rk[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    rk[k] = rank_of(c16[k] & 0x7fffu);
                    rmax = max(rmax, rk[k]);
                    mC[u] |= 1u << (rk[k] & 31u);
                    mM |= (c16[k] >> 15) << (rk[k] & 31u);
                }
                r0[u] = rk[0];                                                       // calls ascend: the first is the lowest
                if (rmax - r0[u] > 15u) fl[u] |= FW_BAD;
                // the one call that can lie outside the covered bases is the first, at start - 1 (readutil.rs:332-340)
                const uint32_t mA = ((c16[0] & 0x7fffu) >= (se[u] & 0xffffu)) ? mC[u] : mC[u] & ~(1u << (r0[u] & 31u));
                mM &= mA;
                if (fl[u] & FW_BAD) {
                    // (> 8 calls, or > 16 window sites between its first and last call): every core site it calls is handed back
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (rk[k] - k0 < ncore) atomicOr(&s_sinfo[rk[k] - k0], 1u << 17);
                    const uint32_t i = lo + r;
                    const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
                    for (uint32_t k = o0 + 8u; k < o1; ++k) {
                        const uint32_t rel = (a.cpg_pos[k] & 0x7fffffffu) - wbase;
                        if (rel < wbits) { const uint32_t q = rank_of(rel) - k0; if (q < ncore) atomicOr(&s_sinfo[q], 1u << 17); }
                    }
                    mC[u] = 0u;                                                      // not a reader of any list
                }
                *reinterpret_cast<uint4 *>(&s_row[r * 4]) = make_uint4(se[u], mC[u], mA, mM);
            }
            const bool ok = (fl[u] & (FW_PASS | FW_BAD)) == FW_PASS;
            // exclusive prefix maximum (file order) of the passing reads' first-call ranks + 1
            const uint32_t fc = (fl[u] & FW_PASS) ? r0[u] + 1u : 0u;
            const uint32_t incl = fw_wave_scan_max_incl(fc);
            pmx[u] = MTH_DPP(incl, 0x138 /*wave_shr:1*/, 0xf, true);
            if (lane == 63) s_cmax[chunk] = incl;
            // readers of each core site in this chunk
            const uint32_t s_lo = max(~fw_wave_max(ok ? ~r0[u] : 0u), k0), s_hi = min(fw_wave_max(ok ? rmax + 1u : 0u), k1);
#ifdef MTH_FW_DEBUG
            if (tid == 0 && t == 0) printf("[wtile C1] u %d fl %u r0 %u rmax %u mC %x s_lo %u s_hi %u ok %d\n", u, fl[u], r0[u], rmax, mC[u], s_lo, s_hi, (int)ok);
#endif
            for (uint32_t s = s_lo; s < s_hi; ++s) {
                const bool has = ok && s - r0[u] < 16u && ((mC[u] >> (s & 31u)) & 1u);
                const unsigned long long b = __ballot(has);
                if (lane == 0) s_cnt[chunk * FW_SC + (s - k0)] = (uint8_t)__popcll(b);
            }
        }
        FW_TK(4);
        __syncthreads();
        FW_TK(5);
        // ---- C2: per site: readers, list and code offsets ----
        const uint32_t nch = (R + 63u) >> 6;
        uint32_t n_s = 0;
        if ((uint32_t)tid < ncore) {
            for (uint32_t ch = 0; ch < nch; ++ch) {
                const uint32_t c = s_cnt[ch * FW_SC + tid];
                s_cnt[ch * FW_SC + tid] = (uint8_t)min(n_s, 255u);
                n_s += c;
            }
        }
        if ((uint32_t)tid < nch) {                                                   // the chunks' carried-in maximum
            uint32_t m = 0;
            for (uint32_t ch = 0; ch < (uint32_t)tid; ++ch) m = max(m, s_cmax[ch]);
            s_ccar[tid] = m;
        }
        const bool over = n_s > cap;                                                 // reservoir / more than this kernel's slots: handed back
        const bool act = n_s >= mind && !over;
        const uint32_t P = act ? n_s * (n_s - 1u) / 2u : 0u, P4 = (P + 3u) & ~3u;
        uint32_t tot_l, tot_c;
        const uint32_t poff = fw_block_scan_excl(act ? n_s : 0u, ws, tot_l);
        const uint32_t coff = fw_block_scan_excl(P4, ws, tot_c);
        if (tot_l > (uint32_t)FW_LPOOL || tot_c > (uint32_t)FW_TCAP) {              // block-uniform
            if (sub_shift > 8) { --sub_shift; continue; }
            heavy_redo = true; continue;
        }
        if ((uint32_t)tid < ncore) {
            atomicOr(&s_sinfo[tid], n_s | (act ? 1u << 16 : 0u) | ((over && n_s >= mind) ? 1u << 17 : 0u));
            s_off[tid] = poff | (coff << 16);
            // the site that holds each 64th code offset
            if (P4) for (uint32_t b = (coff + 63u) >> 6; (b << 6) < coff + P4; ++b) s_hint[b] = (uint16_t)tid;
        }
        __syncthreads();
        FW_TK(6);
        // ---- C3: the readers into their lists (file order); the flush rule ----
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((uint32_t)(u * FW_B + wave * 64) >= R) continue;                    // wave-uniform
            const uint32_t r = (uint32_t)(u * FW_B + tid);
            const uint32_t chunk = (uint32_t)(u * (FW_B / 64) + wave);
            const bool ok = (fl[u] & (FW_PASS | FW_BAD)) == FW_PASS;
            const uint32_t pm = max(pmx[u], s_ccar[chunk]);           // first-call rank + 1 of the passing reads before this one
            uint32_t rmax = 0;
            if (ok) rmax = r0[u] + (31u - (uint32_t)__builtin_clz(__builtin_amdgcn_alignbit(mC[u], mC[u], r0[u] & 31u)));
            const uint32_t s_lo = max(~fw_wave_max(ok ? ~r0[u] : 0u), k0), s_hi = min(fw_wave_max(ok ? rmax + 1u : 0u), k1);
            for (uint32_t s = s_lo; s < s_hi; ++s) {
                const bool has = ok && s - r0[u] < 16u && ((mC[u] >> (s & 31u)) & 1u);
                const unsigned long long b = __ballot(has);
                if (!has) continue;
                const uint32_t q = s - k0;
                // a passing read before this reader whose first call lies beyond the site (fdrp.rs:212): the readers may form two
                // segments -- the walk decides
                if (pm > s + 1u) atomicOr(&s_sinfo[q], 1u << 17);
                const uint32_t info = s_sinfo[q];
                const uint32_t slot = (uint32_t)s_cnt[chunk * FW_SC + q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if ((info >> 16) & 1u) s_pool[(s_off[q] & 0xffffu) + slot] = (uint16_t)r;
            }
        }
        FW_TK(7);
        __syncthreads();
        // ---- D1: every pair of every evaluated site, lane = pair ----
        for (uint32_t f = (uint32_t)tid; f < tot_c; f += FW_B) {
            uint32_t q = s_hint[f >> 6];
            uint32_t off = s_off[q], info = s_sinfo[q];
            auto end_of = [&]() { const uint32_t n = info & 0xffffu; return ((info >> 16) & 1u) ? (off >> 16) + ((n * (n - 1u) / 2u + 3u) & ~3u) : (off >> 16); };
            while (f >= end_of()) { ++q; off = s_off[q]; info = s_sinfo[q]; }
            const uint32_t n = info & 0xffffu, m = f - (off >> 16), Pq = n * (n - 1u) / 2u;
            if (m >= Pq) { s_code[f] = 0; continue; }                                // the list's padding: +0.0
            const uint32_t ent = s_ctab[m];
            const uint32_t pi = ent & 0xffu, pj = ent >> 8;
            const uint32_t li = s_pool[(off & 0xffffu) + pi], lj = s_pool[(off & 0xffffu) + pj];
            const uint4 ri = *reinterpret_cast<const uint4 *>(&s_row[li * 4]), rj = *reinterpret_cast<const uint4 *>(&s_row[lj * 4]);
            const int32_t si = (int32_t)(ri.x & 0xffffu), ei = (int32_t)(ri.x >> 16), sj = (int32_t)(rj.x & 0xffffu), ej = (int32_t)(rj.x >> 16);
            const int32_t ov = min(ei, ej) - max(si, sj) + 1;                        // get_num_overlap_bases, fdrp.rs:97-107
            const bool pair_ok = max(ov, 0) >= a.min_overlap;                        // fdrp.rs:134
            const uint32_t ncpg = (uint32_t)__builtin_popcount(ri.y & rj.y);         // qfdrp.rs:109-119
            const uint32_t ham = (uint32_t)__builtin_popcount(ri.z & rj.z & (ri.w ^ rj.w));   // fdrp.rs:114-115
            // the pair's place in the reference's loop order (fdrp.rs:128-141): i (2 n - i - 1) / 2 + (j - i - 1)
            const uint32_t kl = pi * (2u * n - pi - 1u) / 2u + (pj - pi - 1u);
            s_code[(off >> 16) + kl] = (uint8_t)(pair_ok ? ham * FW_QN + ncpg : 0u);
        }
        FW_TK(8);
        __syncthreads();
        FW_TK(9);
        // ---- D2: one thread per site ----
        const uint32_t info = (uint32_t)tid < ncore ? s_sinfo[tid] : 0u;
        const bool redo = (info >> 17) & 1u;
        const bool emit = (uint32_t)tid < ncore && (redo || ((info >> 16) & 1u));
        FdRec rec;
        rec.pos = 0; rec.f = 0.0f; rec.q = 0.0f; rec.nf = 4u << 24;
        if (emit) {
            rec.pos = s_cpos[tid];
            if (!redo) {
                const uint32_t nq = P4 >> 2;
                const uint32_t *cw = reinterpret_cast<const uint32_t *>(s_code + coff);
                float q = 0.0f;
                uint32_t disc = 0;
                for (uint32_t i = 0; i < nq; ++i) {
                    const uint32_t w = cw[i];
                    const uint32_t c0 = w & 0xffu, c1 = (w >> 8) & 0xffu, c2 = (w >> 16) & 0xffu, c3 = w >> 24;
                    const float t0 = s_quot[c0], t1 = s_quot[c1], t2 = s_quot[c2], t3 = s_quot[c3];
                    q += t0; q += t1; q += t2; q += t3;                               // qfdrp.rs:152, the reference's order
                    disc += (c0 >= (uint32_t)FW_QN) + (c1 >= (uint32_t)FW_QN) + (c2 >= (uint32_t)FW_QN) + (c3 >= (uint32_t)FW_QN);   // fdrp.rs:138-140
                }
                // (num_reads * (num_reads - 1)) as f32 / 2.0 in usize arithmetic (fdrp.rs:143)
                const unsigned long long prod = (unsigned long long)n_s * (unsigned long long)(n_s - 1u);
                const float den = (float)prod / 2.0f;
                rec.f = (float)disc / den; rec.q = q / den; rec.nf = n_s | (1u << 24);
            }
        }
        uint32_t n_rows;
        const uint32_t ro = fw_block_scan_excl(emit ? 1u : 0u, ws, n_rows);
#ifdef MTH_FW_DEBUG
        if (tid == 0 && t == 0) printf("[wtile] P0 %d P1 %d lo %u hi %u n_win %u k0 %u k1 %u tot_l %u tot_c %u rows %u n_s %u act %d info %x\n", P0, P1, lo, hi, n_win, k0, k1, tot_l, tot_c, n_rows, n_s, (int)act, info);
#endif
        if (emit && rows_out + ro < a.rows_per_tile) out[rows_out + ro] = rec;
        rows_out += n_rows;
        FW_TK(10);
#ifdef MTH_FW_TRACE
        if (tid == 0 && a.trace && P0 == T0) { a.trace[(size_t)t * 12] = 1ull; a.trace[(size_t)t * 12 + 1] = tk[0] - tk[11]; for (int k = 1; k < 11; ++k) a.trace[(size_t)t * 12 + 1 + k] = tk[k] - tk[k - 1]; }
#endif
        P0l = P1;
    }
    if (bad & 1u) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
    if (bad & 2u) atomicOr(&a.st->err, (uint32_t)ERRB_CAPACITY);
    if (tid == 0) {
        a.tile_cnt[t] = rows_out;
        if (rows_out) atomicAdd(a.bucket + (t >> TILE_BUCKET_SHIFT), (unsigned long long)rows_out);
    }
}

// One wave per tile: the tile's first row = rows of the buckets before its bucket + rows of the bucket's earlier tiles; its rows go
// to the candidate-site arrays, the handed-back ones (flag 4) also to the list k_fdrp_walk takes them from.  The wave of the
// last tile leaves the total in sites_st->n_sites.
constexpr int FG_WAVES = 4;
__global__ __launch_bounds__(64 * FG_WAVES) void k_fdrp_wtile_gather(const FdRec *__restrict__ scratch, const uint32_t *__restrict__ tile_cnt,
                                                                     const unsigned long long *__restrict__ bucket, const uint32_t ntiles,
                                                                     const uint32_t rows_per_tile, DevState *__restrict__ sites_st,
                                                                     int32_t *__restrict__ site_pos, float *__restrict__ fdrp, float *__restrict__ qfdrp,
                                                                     uint32_t *__restrict__ nreads, uint32_t *__restrict__ flags,
                                                                     uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_cnt) {
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * FG_WAVES + (threadIdx.x >> 6);
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
    const FdRec *__restrict__ src = scratch + (size_t)t * rows_per_tile;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
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
    if (t == ntiles - 1 && lane == 0) sites_st->n_sites = base + n;
}

// The tile pass of one batch: candidate-site arrays (ctx->s_pos, w_val = fdrp, w_aux = qfdrp, w_cov = stored reads, w_flags;
// count in d_state2->n_sites) filled with the finished rows (flag 1) and the handed-back sites (flag 4, listed in redo_list /
// *redo_cnt -- cleared by the caller); the fine read index is left for the walk.
int launch_fdrp_wtile(mth_ctx *ctx, const mth_batch_t &d, const mth_fdrp_params_t &p, uint32_t *redo_list, uint32_t *redo_cnt) {
    hipStream_t s = ctx->stream;
    const int64_t region_len = (int64_t)d.region_end - d.region_beg;
    if (d.n_reads == 0 || region_len <= 0) return MTH_OK;
    int shift = 12;
    if (const char *e = getenv("METHEOR_FDRP_WTILE_SHIFT")) shift = std::min(13, std::max(12, atoi(e)));   // tests / tuning
    const int W = 1 << shift;
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
    FwArgs a;
    a.read_start = d.read_start; a.read_end = d.read_end; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
    a.idx = idx_ptr(ctx);
    a.region_beg = d.region_beg; a.region_end = d.region_end; a.idx_base = idx_base; a.max_span = d.max_span; a.min_overlap = p.min_overlap;
    a.n_reads = d.n_reads; a.n_cpgs = (uint32_t)d.n_cpgs; a.ntiles = ntiles;
    a.min_depth = (uint32_t)std::min<uint64_t>(p.min_depth, 0xffffffffull); a.max_depth = p.max_depth; a.min_qual = p.min_qual;
    a.force_sub = getenv("METHEOR_FDRP_WTILE_SUB") ? 1 : 0; a.force_heavy = getenv("METHEOR_FDRP_WTILE_HEAVY") ? 1 : 0;
    a.scratch = reinterpret_cast<FdRec *>(ctx->scratch.p); a.rows_per_tile = rows_per_tile;
    a.tile_cnt = ctx->tile_cnt.as<uint32_t>(); a.bucket = ctx->tile_bucket.as<unsigned long long>(); a.st = ctx->d_state;
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
        if (shift == 12) hipLaunchKernelGGL((k_fdrp_wtile<12, 512>), dim3(grid), dim3(FW_B), 0, s, a);
        else hipLaunchKernelGGL((k_fdrp_wtile<13, 1024>), dim3(grid), dim3(FW_B), 0, s, a);
    }
    {
        LaunchTimer lt(ctx, K_GATHER);
        hipLaunchKernelGGL(k_fdrp_wtile_gather, dim3((ntiles + FG_WAVES - 1) / FG_WAVES), dim3(64 * FG_WAVES), 0, s,
                           reinterpret_cast<const FdRec *>(ctx->scratch.p), ctx->tile_cnt.as<uint32_t>(), ctx->tile_bucket.as<unsigned long long>(),
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
        fprintf(stderr, "[fdrp wtile trace] tiles %.0f; cycles of a tile's first stretch: head %.0f  A %.0f  bar %.0f  B %.0f  C1 %.0f  bar %.0f  C2 %.0f  C3 %.0f  D1 %.0f  bar %.0f  D2+rows %.0f\n",
                h[0], h[1] / h[0], h[2] / h[0], h[3] / h[0], h[4] / h[0], h[5] / h[0], h[6] / h[0], h[7] / h[0], h[8] / h[0], h[9] / h[0], h[10] / h[0], h[11] / h[0]);
    }
#endif
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

}  // namespace mth
