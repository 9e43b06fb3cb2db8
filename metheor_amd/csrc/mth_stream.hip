// mth_stream.hip -- fused PDR + LPMD as ONE STREAM over the coordinate-sorted reads (gfx950, CDNA4, wave64).
//
// Same results as the tile pipeline of mth_pdr_lpmd.hip (reference: src/pdr.rs:119-212, src/lpmd.rs:154-202,
// src/readutil.rs:134-145, 166-224), different shape.  The tile kernel cuts the CONTIG into 4096-bp tiles: every tile looks
// its candidate reads up in an index (a kernel of its own), re-reads the halo of its neighbour, clears and compacts 4096
// counters whether the tile holds 700 reads (config 2) or 265 (WGBS depth, config 3), and meets at three workgroup barriers.
// Here the READS are cut instead: wave w takes reads [w*C, (w+1)*C) in file order, 64 per step, and owns the reference
// positions [start[w*C] - 1, start[(w+1)*C] - 1).  Its site counters live in a wave-private RING of R words in LDS
// (word = position mod R; coverage | discordant << 16 as in the tile kernel): because the reads are sorted, every position
// below the current step's first start - 1 is final, so each step first FLUSHES that stretch (compaction of the words with
// coverage >= min_depth into the wave's scratch slice, words zeroed behind it) and then scatters its 64 reads in front of it.
//   * no read index (k_build_index is not launched), no per-tile look-up chain: the next step's offsets are requested a step ahead
//   * no workgroup barrier after the slot tables are built: a wave never waits for another wave
//   * halo = the few reads before w*C that can still reach the wave's first position (found by a backward scan), not 4 % of every tile
//   * work per position happens once per 512 flushed positions of ONE wave, next to that wave's own reads
//   * LPMD partials: one wave reduction per wave, not per tile
// A wave whose reads (halo included) could push a 16-bit counter past 65535 runs the same code with 32-bit counters over half
// the ring (WIDE).  Rows go to a position-indexed scratch slice per wave; k_gather_stream packs the slices into the sorted
// result columns (as k_gather does for tiles) and commits the batch.
//
// Used for batches with 8-bit relative positions and max_span <= 256 (launch_pdr_lpmd decides); everything else keeps the
// tile pipeline.  Roofline: HBM (16 B/read + 5 B/call in, 12 B/site out); integer / bit work only, no MFMA.
#include "mth_ctx.h"
#include "mth_tile_dev.h"

namespace mth {

constexpr int ST_WAVES = 4;          // waves per workgroup; they share the slot tables and nothing else

struct StreamArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const uint8_t  *cpg_rel;
    DevState *st;                    // error bits
    DevState *cst;                   // the sink's counters (cur_base is set here, the gather commits the rest)
    uint32_t *slice_cnt;             // rows per wave
    uint32_t *slice_base;            // first scratch row of the wave's slice (= its first owned position - region_beg)
    unsigned long long *bucket;      // per 256-wave bucket: [nbk] rows, then [nbk][4] LPMD partial sums (zero on entry)
    uint32_t nbk;
    SiteRec *scratch;                // one row per position of the region
    int32_t region_beg, region_end, max_span;
    uint32_t n_reads, n_cpgs;
    uint32_t C, nwaves;              // reads per wave, waves
    uint32_t min_cov, min_cpgs;
    int32_t  min_dist, max_dist;
    uint8_t  pdr_min_qual, lpmd_min_qual, want_pdr, want_lpmd;
#ifdef MTH_STREAM_TRACE
    unsigned long long *trace;       // experiment build: per-wave begin / end ticks and placement (tools/stream_trace.py)
#endif
};

// ---------------------------------------------------------------------------------------------
// Flush: emit the rows of positions [F, F + len) (F and len multiples of PER, len <= positions in the ring) that lie in
// [F + lo_rel, F + hi_rel) and have coverage >= min_cov, in position order, to out[rows ...]; ZERO: leave the words zero.
// Lane l takes PER consecutive positions per round of 64 * PER.
template <int RSH, int PER, bool WIDE, bool ZERO>
__device__ __forceinline__ uint32_t stream_flush(uint32_t *__restrict__ ring, const uint32_t F, const uint32_t len,
                                                 const uint32_t lo_rel, const uint32_t hi_rel, const uint32_t min_cov,
                                                 SiteRec *__restrict__ out, uint32_t rows) {
    constexpr uint32_t R = 1u << RSH;
    constexpr uint32_t RP = WIDE ? R / 2 : R;
    static_assert(PER % 4 == 0 && PER <= 32, "uint4 LDS reads, 32-bit mask");
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t off = 0; off < len; off += 64u * PER) {
        const uint32_t rel0 = off + lane * PER;
        const bool mine_valid = rel0 < len;
        uint32_t qual = 0;
        if (mine_valid) {
            if (!WIDE) {
                // packed word = coverage | discordant << 16: coverage - min_cov funnel-shifted into the mask, last position first
                uint32_t below = 0;
#pragma unroll
                for (int q = PER / 4 - 1; q >= 0; --q) {
                    const uint32_t wi = (F + rel0 + 4u * q) & (RP - 1u);
                    const uint4 x = *reinterpret_cast<const uint4 *>(ring + wi);
                    below = __builtin_amdgcn_alignbit(below, (x.w & 0xffffu) - min_cov, 31);
                    below = __builtin_amdgcn_alignbit(below, (x.z & 0xffffu) - min_cov, 31);
                    below = __builtin_amdgcn_alignbit(below, (x.y & 0xffffu) - min_cov, 31);
                    below = __builtin_amdgcn_alignbit(below, (x.x & 0xffffu) - min_cov, 31);
                }
                qual = ~below & (PER == 32 ? 0xffffffffu : (1u << PER) - 1u);
            } else {
#pragma unroll
                for (int q = 0; q < PER / 4; ++q) {
                    const uint32_t wi = (F + rel0 + 4u * q) & (RP - 1u);
                    const uint4 x = *reinterpret_cast<const uint4 *>(ring + wi);
                    const uint4 y = *reinterpret_cast<const uint4 *>(ring + RP + wi);
                    qual |= (x.x + y.x >= min_cov ? 1u : 0u) << (4 * q);
                    qual |= (x.y + y.y >= min_cov ? 1u : 0u) << (4 * q + 1);
                    qual |= (x.z + y.z >= min_cov ? 1u : 0u) << (4 * q + 2);
                    qual |= (x.w + y.w >= min_cov ? 1u : 0u) << (4 * q + 3);
                }
            }
            // only positions the wave owns: lo_rel <= rel0 + j < hi_rel
            const int32_t left_hi = (int32_t)hi_rel - (int32_t)rel0;
            qual &= left_hi >= 32 ? 0xffffffffu : (left_hi > 0 ? (1u << left_hi) - 1u : 0u);
            const int32_t left_lo = (int32_t)lo_rel - (int32_t)rel0;
            qual &= left_lo <= 0 ? 0xffffffffu : (left_lo >= 32 ? 0u : ~((1u << left_lo) - 1u));
        }
        const uint32_t mine = __builtin_popcount(qual);
        const uint32_t incl = wave_scan_incl(mine);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        uint32_t o = rows + incl - mine;
        while (qual) {
            const uint32_t prel = rel0 + (uint32_t)__builtin_ctz(qual);
            qual &= qual - 1;
            const uint32_t wi = (F + prel) & (RP - 1u);
            SiteRec rr; rr.pos = (int32_t)(F + prel); rr.pad = 0;
            if (WIDE) { rr.n_conc = ring[wi]; rr.n_disc = ring[RP + wi]; }
            else { const uint32_t x = ring[wi]; rr.n_disc = x >> 16; rr.n_conc = (x & 0xffffu) - rr.n_disc; }
            out[o++] = rr;
        }
        if (ZERO && mine_valid) {
#pragma unroll
            for (int q = 0; q < PER / 4; ++q) {
                const uint32_t wi = (F + rel0 + 4u * q) & (RP - 1u);
                *reinterpret_cast<uint4 *>(ring + wi) = make_uint4(0, 0, 0, 0);
                if (WIDE) *reinterpret_cast<uint4 *>(ring + RP + wi) = make_uint4(0, 0, 0, 0);
            }
        }
        rows += total;
    }
    return rows;
}

// ---------------------------------------------------------------------------------------------
// One wave's whole range.  WIDE: 32-bit counters, concordant at ring[0, R/2), discordant at ring[R/2, R).
// b0 / e: the wave's own reads; h <= b0: first halo read; emit_lo / emit_hi: the positions it owns (clipped to the region).
template <int RSH, int PER, bool WIDE>
__device__ __forceinline__ void stream_wave(const StreamArgs &a, const uint32_t w, uint32_t *__restrict__ ring, const uint32_t wvoff,
                                            uint32_t *__restrict__ ring_base, const SlotTabs &tabs,
                                            const uint32_t h, const uint32_t b0, const uint32_t e,
                                            const int64_t emit_lo, const int64_t emit_hi, const bool owns, int32_t s_prev) {
    constexpr uint32_t R = 1u << RSH;
    constexpr uint32_t RP = WIDE ? R / 2 : R;
    constexpr uint32_t RM4 = (RP - 1u) << 2;
    constexpr int NB = 8;
    const uint32_t lane = threadIdx.x & 63;
    const bool do_pdr = a.want_pdr && owns;
    const bool do_lp = a.want_lpmd != 0;
    SiteRec *__restrict__ out = a.scratch + (owns ? (size_t)(emit_lo - a.region_beg) : 0);

    uint32_t i = h, rows = 0;
    uint32_t lp_c = 0, lp_d = 0, n_rv = 0, bad = 0, uns = 0;
    uint32_t F = ((uint32_t)(a.read_start[h] - 1)) & ~(uint32_t)(PER - 1);     // whole lane strides of the flush
    uint32_t o0 = 0, o1 = 0;
    { const uint32_t r = i + lane; if (r < e) { o0 = a.cpg_off[r]; o1 = a.cpg_off[r + 1]; } }
    // (distances between live calls are < 2^16, so capping max_distance keeps dead-slot differences outside)
    const int32_t maxd = min(a.max_dist, 255);       // 8-bit relpos: no distance beyond 255
    const int32_t mind = max(a.min_dist, 0);
    const bool lp_possible = do_lp && maxd >= a.min_dist && maxd >= 0;      // min > max: no pair can qualify
    auto window = [&](uint32_t &lo_rel, uint32_t &hi_rel, const uint32_t len) {
        const int64_t Fs = (int64_t)(int32_t)F;
        const int64_t lo = emit_lo - Fs, hi = emit_hi - Fs;
        lo_rel = (uint32_t)(lo < 0 ? 0 : (lo > (int64_t)len ? (int64_t)len : lo));
        hi_rel = (uint32_t)(hi < 0 ? 0 : (hi > (int64_t)len ? (int64_t)len : hi));
    };

    while (i < e) {
        const uint32_t r = i + lane;
        const bool act = r < e;
        const uint32_t nact = min(e - i, 64u);
        // the 8-slot window of every read of the step lies inside the call arrays unless the step holds the batch's last reads
        const uint32_t o_last = (uint32_t)__builtin_amdgcn_readlane((int)o0, (int)(nact - 1u));
        const bool safe = (uint64_t)o_last + NB <= (uint64_t)a.n_cpgs;
        uint32_t v[NB];
        uint32_t rraw0 = 0, rraw1 = 0;
        int32_t s = 0x7fffffff;
        uint32_t mq = 0;
        const uint32_t n = o1 - o0;
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = 0;
        if (safe) {
            if (act) {
                const uint32_t *__restrict__ cp = a.cpg_pos + o0;
#pragma unroll
                for (int k4 = 0; k4 < NB / 4; ++k4) {
                    const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(cp + 4 * k4);
                    v[4 * k4] = x.x; v[4 * k4 + 1] = x.y; v[4 * k4 + 2] = x.z; v[4 * k4 + 3] = x.w;
                }
                if (do_lp) { const u32x2_a1 x = *reinterpret_cast<const u32x2_a1 *>(a.cpg_rel + o0); rraw0 = x.x; rraw1 = x.y; }
                s = a.read_start[r]; mq = a.read_mapq[r];
            }
        } else if (act) {
            s = a.read_start[r]; mq = a.read_mapq[r];
            if (n) {
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const uint32_t ci = o0 + min((uint32_t)k, n - 1u);
                    v[k] = a.cpg_pos[ci];
                    const uint32_t rb = a.cpg_rel[ci];
                    if (k < 4) rraw0 |= rb << (8 * k); else rraw1 |= rb << (8 * (k - 4));
                }
            }
        }
        // the next step's offsets travel during this step's arithmetic
        uint32_t o0n = 0, o1n = 0;
        { const uint32_t rn = r + 64u; if (rn < e) { o0n = a.cpg_off[rn]; o1n = a.cpg_off[rn + 1]; } }

        const int32_t s_first = __builtin_amdgcn_readfirstlane(s);
        uint32_t np = nact;
        if (do_pdr) {
            // everything below the step's first start - 1 is final (sorted reads; a call lies at or after its read's start - 1)
            const uint32_t newF = ((uint32_t)(s_first - 1)) & ~(uint32_t)(PER - 1);
            const int32_t adv = (int32_t)(newF - F);
            if (adv > 0) {
                const uint32_t len = min((uint32_t)adv, RP);
                uint32_t lo_rel, hi_rel;
                window(lo_rel, hi_rel, len);
                rows = stream_flush<RSH, PER, WIDE, true>(ring, F, len, lo_rel, hi_rel, a.min_cov, out, rows);
            }
            F = newF;       // (adv < 0: unsorted input, flagged below; the ring then holds garbage nobody will report)
            // the ring holds [F, F + RP): a read takes part in this step if its calls, [s - 1, s - 1 + max_span], fit
            const bool fits = act && (uint32_t)(s - (int32_t)F) <= RP - (uint32_t)a.max_span;
            const unsigned long long pm = __ballot(fits);
            np = pm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~pm);
        }
        const bool on = lane < np;
        {   // sortedness (ERRB_UNSORTED): every read against its predecessor in file order
            const int32_t sp = __builtin_amdgcn_update_dpp(s_prev, s, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
            if (on && s < sp) uns = 1;
            s_prev = __builtin_amdgcn_readlane(s, (int)(np - 1u));
        }

        const bool owned = on && r >= b0 && s >= a.region_beg && s < a.region_end;
        // lpmd.rs:176-179
        const bool lp_ok = do_lp && owned && (mq >= a.lpmd_min_qual);
        if (do_lp && owned) n_rv += lp_ok ? 0x10001u : 1u;
        // pdr.rs:147-157
        const bool pdr_ok = on && do_pdr && (n >= a.min_cpgs) && (mq >= a.pdr_min_qual) && (n > 0);
        const bool work = (lp_ok || pdr_ok) && n != 0;
        const bool any_lp = lp_possible && __any(work && lp_ok && n > 1);
        if (work) {
            const uint32_t sm1 = (uint32_t)(s - 1);
            // dead call word: a lane-private position just behind the flush pointer (its word receives +0), with the first call's state
            const uint32_t dead_w = ((F - 1u - lane) & 0x7fffffffu) | (v[0] & 0x80000000u);
            const uint32_t n_lp = lp_ok ? min(n, (uint32_t)NB) : 0u;
            uint32_t acc = 0, xmax;
            // masks of the live slots from the table row of n (rows 8.. are all-live)
            const uint32_t nrow = min(n, 8u);
            const uint4 ma = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[0], mb = reinterpret_cast<const uint4 *>(&tabs.mtab[nrow][0])[1];
            const uint32_t mk[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
            {
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
            uint32_t bad_it = (xmax > (uint32_t)a.max_span) ? 1u : 0u;      // this read's span violations
            uint32_t disc = acc >> 31;
            const bool any_long = __any(n > (uint32_t)NB);   // wave-uniform: the tails below are rare
            if (any_long && n > (uint32_t)NB) {
                const uint32_t first = v[0] >> 31;
                for (uint32_t k = NB; k < n; ++k) {
                    const uint32_t x = a.cpg_pos[o0 + k];
                    disc |= (x >> 31) ^ first;
                    bad_it |= ((x & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
                }
            }
            bad |= bad_it;
            // windowed pair counts (readutil.rs:166-224), two pairs per instruction in 16-bit fields: see mth_tile_dev.h
            if (any_lp) {
                uint32_t SQ[4], SO[4], Q[4], O[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) SQ[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x070c030cu);
#pragma unroll
                for (int q = 0; q < 3; ++q) SO[q] = __builtin_amdgcn_perm(v[2 * q + 2], v[2 * q + 1], 0x070c030cu);
                SO[3] = __builtin_amdgcn_perm(0u, v[7], 0x070c030cu);
                Q[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c010c00u); Q[1] = __builtin_amdgcn_perm(0u, rraw0, 0x0c030c02u);
                Q[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c010c00u); Q[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c030c02u);
                O[0] = __builtin_amdgcn_perm(0u, rraw0, 0x0c020c01u); O[1] = __builtin_amdgcn_perm(rraw1, rraw0, 0x0c040c03u);
                O[2] = __builtin_amdgcn_perm(0u, rraw1, 0x0c020c01u); O[3] = __builtin_amdgcn_perm(0u, rraw1, 0x0c0c0c03u);
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
                        accIN += __builtin_popcount(IN);
                        accDD += __builtin_popcount(DD);
                        orB |= Bw;
                    }
                    if (!__any((orB & 0x80008000u) != 0u)) break;      // no lane has a pair within max_distance on this diagonal
                }
                lp_c += accIN - accDD;
                lp_d += accDD;
            }
            // a read with more than NB calls: pairs whose LATER call is the (NB+1)-th or beyond come from memory (divergent, rare)
            if (any_long && lp_ok && n > (uint32_t)NB) {
                for (uint32_t k = NB; k < n; ++k) {
                    const int32_t rk = (int32_t)a.cpg_rel[o0 + k];
                    const uint32_t mkk = a.cpg_pos[o0 + k] >> 31;
                    for (uint32_t j = k; j-- > 0;) {
                        const int32_t dist = rk - (int32_t)a.cpg_rel[o0 + j];
                        if (dist > a.max_dist) break;          // readutil.rs:184 (anchors evicted)
                        if (dist < a.min_dist) continue;       // readutil.rs:196
                        if ((a.cpg_pos[o0 + j] >> 31) == mkk) lp_c += 1; else lp_d += 1;
                    }
                }
            }
            // scatter +1 to the read's sites (pdr.rs:180-191): every call of a read that passed the span check and the
            // participation test has a ring word of its own; dead slots add 0 to the lane's word behind the flush pointer
            if (pdr_ok && !bad_it) {
                if (!WIDE) {
                    const uint32_t one = disc ? 0x10001u : 1u;           // coverage in the low half, discordant reads in the high half
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const uint32_t a4 = __builtin_amdgcn_bitop3_b32(v[k] << 2, RM4, wvoff, 0xea);      // (a & b) | c
                        atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ring_base) + a4), k == 0 ? one : (one & mk[k]));
                    }
                    if (any_long) {
                        for (uint32_t k = NB; k < n; ++k) atomicAdd(ring + (a.cpg_pos[o0 + k] & (RP - 1u)), one);
                    }
                } else {
                    const uint32_t wd = wvoff | (disc ? RP * 4u : 0u);   // concordant reads at [0, RP), discordant at [RP, 2 RP)
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const uint32_t a4 = __builtin_amdgcn_bitop3_b32(v[k] << 2, RM4, wd, 0xea);
                        atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ring_base) + a4), k == 0 ? 1u : (1u & mk[k]));
                    }
                    if (any_long) {
                        for (uint32_t k = NB; k < n; ++k) atomicAdd(ring + (disc ? RP : 0u) + (a.cpg_pos[o0 + k] & (RP - 1u)), 1u);
                    }
                }
            }
        }   // work
        i += np;
        if (np == 64u) { o0 = o0n; o1 = o1n; }
        else { const uint32_t r2 = i + lane; o0 = 0; o1 = 0; if (r2 < e) { o0 = a.cpg_off[r2]; o1 = a.cpg_off[r2 + 1]; } }
    }
    if (do_pdr) {   // what is still in the ring
        uint32_t lo_rel, hi_rel;
        window(lo_rel, hi_rel, RP);
        rows = stream_flush<RSH, PER, WIDE, false>(ring, F, RP, lo_rel, hi_rel, a.min_cov, out, rows);
    }
    if (uns | bad) atomicOr(&a.st->err, (uns ? (uint32_t)ERRB_UNSORTED : 0u) | (bad ? (uint32_t)ERRB_SPAN : 0u));
    // LPMD partials: one reduction per wave (a wave holds <= 32768 reads: the read counts share a word)
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (do_lp) {
        const uint32_t rv = wave_sum(n_rv);
        r2 = rv & 0xffffu; r3 = rv >> 16;
        r0 = wave_sum(lp_c); r1 = wave_sum(lp_d);
    }
    if (lane == 0) {
        unsigned long long *bk = a.bucket + a.nbk + (size_t)(w >> TILE_BUCKET_SHIFT) * 4;
        if (r0) atomicAdd(bk + 0, (unsigned long long)r0);
        if (r1) atomicAdd(bk + 1, (unsigned long long)r1);
        if (r2) atomicAdd(bk + 2, (unsigned long long)r2);
        if (r3) atomicAdd(bk + 3, (unsigned long long)r3);
        a.slice_cnt[w] = rows;
        a.slice_base[w] = owns ? (uint32_t)(emit_lo - a.region_beg) : 0u;
        if (rows) atomicAdd(a.bucket + (w >> TILE_BUCKET_SHIFT), (unsigned long long)rows);
    }
}

template <int RSH, int PER>
__global__ __launch_bounds__(64 * ST_WAVES, 8) void k_pdr_lpmd_stream(const StreamArgs a) {
    constexpr uint32_t R = 1u << RSH;
    __shared__ __attribute__((aligned(16))) uint32_t ring_all[ST_WAVES][R];
    __shared__ __attribute__((aligned(16))) SlotTabs tabs;
#ifdef MTH_ST_PADLDS
    __shared__ uint32_t pad_lds[MTH_ST_PADLDS / 4];      // experiment build: fewer workgroups per CU
    if (a.n_reads == 0xffffffffu) pad_lds[threadIdx.x] = 1;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    slot_tabs_init(tabs, tid);
    uint32_t *ring = ring_all[wv];
    for (uint32_t k = lane; k < R / 4; k += 64) reinterpret_cast<uint4 *>(ring)[k] = make_uint4(0, 0, 0, 0);
    __syncthreads();     // the only workgroup barrier: the slot tables
    // block b runs on XCD b % 8 (observed; speed only): give each XCD a contiguous run of waves (neighbours share their halo in L2)
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const uint32_t w = blk * ST_WAVES + wv;
    if (w >= a.nwaves) return;
#ifdef MTH_STREAM_TRACE
    const unsigned long long t_beg = __builtin_readcyclecounter();
#endif
    if (w == 0 && lane == 0) a.cst->cur_base = a.cst->n_sites;   // this batch's rows go after everything emitted so far (read by the gather)

    const uint32_t b0 = w * a.C, e = min(b0 + a.C, a.n_reads);
    // owned positions: [start[b0] - 1, start[e] - 1), open at the batch's ends, clipped to the region
    const int32_t S_w = a.read_start[b0];
    int64_t emit_lo = a.region_beg, emit_hi = a.region_end;
    if (w > 0) emit_lo = max(emit_lo, (int64_t)S_w - 1);
    if (e < a.n_reads) emit_hi = min(emit_hi, (int64_t)a.read_start[e] - 1);
    const bool owns = a.want_pdr && emit_hi > emit_lo;
    // halo: the reads before b0 that can call an owned position, start >= start[b0] - max_span (backward scan, 64 at a time)
    uint32_t h = b0;
    if (owns && b0 > 0) {
        const int64_t thr = (int64_t)S_w - a.max_span;
        for (;;) {
            const int64_t idx = (int64_t)h - 1 - lane;
            const bool ok = idx >= 0 && (int64_t)a.read_start[idx >= 0 ? idx : 0] >= thr;
            const unsigned long long m = __ballot(ok);
            const uint32_t c = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
            h -= c;
            if (c < 64u || h == 0u) break;
        }
    }
    const int32_t s_prev = h > 0 ? a.read_start[h - 1] : (int32_t)0x80000000;
    const uint32_t wvoff = wv * (R * 4u);
    // a position is called at most once per read: 16-bit counters are exact while the wave sees <= 65535 reads
    if (e - h <= 65535u) stream_wave<RSH, PER, false>(a, w, ring, wvoff, &ring_all[0][0], tabs, h, b0, e, emit_lo, emit_hi, owns, s_prev);
    else stream_wave<RSH, PER, true>(a, w, ring, wvoff, &ring_all[0][0], tabs, h, b0, e, emit_lo, emit_hi, owns, s_prev);
#ifdef MTH_STREAM_TRACE
    if (lane == 0 && a.trace) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[4 * (size_t)w + 0] = t_beg;
        a.trace[4 * (size_t)w + 1] = __builtin_readcyclecounter();
        a.trace[4 * (size_t)w + 2] = ((unsigned long long)xcc << 32) | hwid;
        a.trace[4 * (size_t)w + 3] = ((unsigned long long)(b0 - h) << 32) | (e - b0);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// One wave per slice (= wave of the stream kernel): k_gather of mth_pdr_lpmd.hip with the slices' own scratch bases.  The
// wave of the last slice commits the batch to DevState and zeroes the bucket sums the NEXT batch will use (the two sets
// alternate: nobody else touches the other set between two batches of a stream).
constexpr int SGATHER_WAVES = 4;
__global__ __launch_bounds__(64 * SGATHER_WAVES) void k_gather_stream(const SiteRec *__restrict__ scratch,
                                               const uint32_t *__restrict__ slice_cnt, const uint32_t *__restrict__ slice_base,
                                               const unsigned long long *__restrict__ bucket, uint32_t nbk,
                                               uint32_t nslices, int fin_only, int want_lpmd,
                                               DevState *__restrict__ st, uint32_t *__restrict__ batch_cnt,
                                               unsigned long long *__restrict__ zero_words, uint32_t n_zero,
                                               int32_t *__restrict__ out_pos, float *__restrict__ out_pdr,
                                               uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    const uint32_t t = fin_only ? nslices - 1 : blockIdx.x * SGATHER_WAVES + (threadIdx.x >> 6);
    if (t >= nslices) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t bk = t >> TILE_BUCKET_SHIFT;
    static_assert(TILE_BUCKET_SHIFT == 8, "a bucket's earlier slices are four loads per lane");
    const uint32_t n = slice_cnt[t];
    const SiteRec *__restrict__ src = scratch + slice_base[t];
    const uint64_t cur = st->cur_base;
    const uint32_t u0 = (bk << TILE_BUCKET_SHIFT) + lane;
    uint32_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = (u0 + 64u * k < t) ? slice_cnt[u0 + 64u * k] : 0u;
    uint32_t part = 0;                                  // rows of one batch fit 32 bits (<= region positions)
    for (uint32_t b = lane; b < bk; b += 64) part += (uint32_t)bucket[b];
    part += (x[0] + x[1]) + (x[2] + x[3]);
    const uint32_t before = wave_sum(part);
    const uint64_t base = cur + before;
    if (!fin_only) {
        for (uint32_t j = lane; j < n; j += 64) {
            const SiteRec r = src[j];
            out_pos[base + j] = r.pos;
            out_nc[base + j] = r.n_conc;
            out_nd[base + j] = r.n_disc;
            out_pdr[base + j] = (float)r.n_disc / ((float)r.n_conc + (float)r.n_disc);      // pdr.rs:47-49
        }
    }
    if (t != nslices - 1) return;
    const uint32_t total = before + n;
    if (lane == 0) {
        st->n_sites = cur + total;
        batch_cnt[st->n_batches] = total;
        st->n_batches += 1;
    }
    if (want_lpmd) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long y = 0;
            for (uint32_t b = lane; b < nbk; b += 64) y += bucket[nbk + (size_t)b * 4 + k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) y += __shfl_down(y, o, 64);
            if (lane == 0) st->lpmd[k] += (long long)y;
        }
    }
    for (uint32_t k = lane; k < n_zero; k += 64) zero_words[k] = 0ull;
}

// ---------------------------------------------------------------------------------------------
bool stream_eligible(const mth_batch_t &b) {
    // A/B switches: MTH_STREAM=1 takes the streaming kernel, MTH_NO_STREAM=1 keeps the tile pipeline (the default while the
    // streaming kernel is still the slower of the two on config 2)
    const bool off = getenv("MTH_NO_STREAM") != nullptr || getenv("MTH_STREAM") == nullptr;      // read per batch: the tests run both forms in one process
    return !off && b.cpg_rel != nullptr && b.max_span >= 1 && b.max_span <= 256 && b.n_reads > 0;
}

int launch_pdr_lpmd_stream(mth_ctx *ctx, const mth_batch_t &b, const mth_pdr_lpmd_params_t &p, const TileSink *sink) {
    hipStream_t s = ctx->stream;
    DevState *cst = sink ? sink->st : ctx->d_state;
    uint32_t *bcnt = sink ? sink->batch_cnt : ctx->batch_cnt.as<uint32_t>();
    int32_t *o_pos = sink ? sink->pos : ctx->out_pos.as<int32_t>();
    float *o_pdr = sink ? sink->pdr : ctx->out_pdr.as<float>();
    uint32_t *o_nc = sink ? sink->nc : ctx->out_nc.as<uint32_t>();
    uint32_t *o_nd = sink ? sink->nd : ctx->out_nd.as<uint32_t>();
    const int64_t region_len = (int64_t)b.region_end - b.region_beg;
    // reads per wave: one round of resident waves (256 CUs x 32) when the batch is large enough, whole steps of 64
    static const uint32_t c_env = getenv("MTH_STREAM_C") ? (uint32_t)atoi(getenv("MTH_STREAM_C")) : 0u;
    uint32_t C = (uint32_t)((((uint64_t)b.n_reads + 8191u) / 8192u + 63u) / 64u * 64u);
    C = std::min(std::max(C, 256u), 32768u);
    if (c_env) C = std::min(std::max(c_env / 64u * 64u, 64u), 32768u);
    const uint32_t nwaves = (b.n_reads + C - 1) / C;
    const uint32_t nbk = (nwaves + (1u << TILE_BUCKET_SHIFT) - 1) >> TILE_BUCKET_SHIFT;

    MTH_HIP(ctx, ctx->tile_cnt.reserve((size_t)nwaves * 4, s));
    MTH_HIP(ctx, ctx->slice_base.reserve((size_t)nwaves * 4, s));
    // two sets of bucket sums, used alternately; a fresh (or regrown) buffer is cleared as a whole
    const size_t set_words = 32 * 5;       // room for 8192 waves per set; larger batches regrow
    const size_t need_words = std::max(set_words, (size_t)nbk * 5);
    if (ctx->sbucket.cap < need_words * 2 * sizeof(unsigned long long) || ctx->sbucket_words != need_words) {
        MTH_HIP(ctx, ctx->sbucket.reserve(need_words * 2 * sizeof(unsigned long long), s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->sbucket.p, 0, need_words * 2 * sizeof(unsigned long long), s));
        ctx->sbucket_words = need_words;
        ctx->sbucket_set = 0;
    }
    unsigned long long *bucket = ctx->sbucket.as<unsigned long long>() + (size_t)ctx->sbucket_set * need_words;
    unsigned long long *other = ctx->sbucket.as<unsigned long long>() + (size_t)(ctx->sbucket_set ^ 1) * need_words;
    ctx->sbucket_set ^= 1;
    if (p.want_pdr) MTH_HIP(ctx, ctx->scratch.reserve(((size_t)region_len + 64) * sizeof(SiteRec), s));

    StreamArgs a;
    a.read_start = b.read_start; a.read_mapq = b.read_mapq; a.cpg_off = b.cpg_off; a.cpg_pos = b.cpg_pos; a.cpg_rel = b.cpg_rel;
    a.st = ctx->d_state; a.cst = cst;
    a.slice_cnt = ctx->tile_cnt.as<uint32_t>(); a.slice_base = ctx->slice_base.as<uint32_t>();
    a.bucket = bucket; a.nbk = nbk; a.scratch = ctx->scratch.as<SiteRec>();
    a.region_beg = b.region_beg; a.region_end = b.region_end; a.max_span = b.max_span;
    a.n_reads = b.n_reads; a.n_cpgs = b.n_cpgs; a.C = C; a.nwaves = nwaves;
    a.min_cov = p.pdr_min_depth > 1 ? p.pdr_min_depth : 1;
    a.min_cpgs = p.pdr_min_cpgs;
    a.min_dist = p.lpmd_min_distance; a.max_dist = p.lpmd_max_distance;
    a.pdr_min_qual = p.pdr_min_qual; a.lpmd_min_qual = p.lpmd_min_qual;
    a.want_pdr = p.want_pdr; a.want_lpmd = p.want_lpmd;
#ifdef MTH_STREAM_TRACE
    static unsigned long long *d_trace = nullptr;
    if (!d_trace) (void)hipMalloc((void **)&d_trace, 4 * 8 * 65536);
    a.trace = d_trace;
#endif
    {
        LaunchTimer lt(ctx, K_STREAM);
        const uint32_t nblk = ((nwaves + ST_WAVES - 1) / ST_WAVES + 7) / 8 * 8;   // whole rows of 8 XCDs (remap in the kernel)
        hipLaunchKernelGGL((k_pdr_lpmd_stream<10, 8>), dim3(nblk), dim3(64 * ST_WAVES), 0, s, a);
    }
    {
        LaunchTimer lt(ctx, K_GATHER);
        hipLaunchKernelGGL(k_gather_stream, dim3(p.want_pdr ? (nwaves + SGATHER_WAVES - 1) / SGATHER_WAVES : 1u),
                           dim3(p.want_pdr ? 64 * SGATHER_WAVES : 64), 0, s, ctx->scratch.as<SiteRec>(), ctx->tile_cnt.as<uint32_t>(),
                           ctx->slice_base.as<uint32_t>(), bucket, nbk, nwaves, p.want_pdr ? 0 : 1, (int)p.want_lpmd, cst, bcnt,
                           other, (uint32_t)need_words, o_pos, o_pdr, o_nc, o_nd);
    }
    MTH_HIP(ctx, hipGetLastError());
#ifdef MTH_STREAM_TRACE
    if (getenv("MTH_STREAM_TRACE_OUT") && nwaves <= 65536) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> t(4 * (size_t)nwaves);
        (void)hipMemcpy(t.data(), d_trace, t.size() * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("MTH_STREAM_TRACE_OUT"), "wb");
        if (f) { fwrite(t.data(), 8, t.size(), f); fclose(f); }
    }
#endif
    if (sink) {
        // site discovery for the site walks (MHL, FDRP / qFDRP, exact PDR): they find a site's candidate reads through the linear
        // read index, which the tile pipeline used to leave behind
        int32_t idx_base; uint32_t ntiles;
        return build_read_index(ctx, b, 4096, idx_base, ntiles);
    }
    return MTH_OK;
}

}  // namespace mth
