// mth_fileorder.hip -- PDR, MHL, FDRP and qFDRP of a decoded stream that is NOT coordinate-sorted, on gfx950.
//
// The reference iterates the records in whatever order the file has and never checks it (pdr.rs:139, mhl.rs:155, fdrp.rs:197,
// qfdrp.rs:209).  These four measures keep a map of open sites that every record flushes -- `retain`: a site whose key lies more
// than a margin before the record's FIRST CpG is finalised and removed (pdr.rs:160-177: margin 150, only records that pass the
// filters flush; mhl.rs:162-173: margin 0, every record with a CpG flushes, before the filters; fdrp.rs:212-223: margin 0,
// passing records) -- and a later record that calls a removed site opens a NEW entry whose result overwrites the earlier one if it
// reaches min_depth (BTreeMap::insert).  So a site's result is that of the LAST SEGMENT with enough reads, a segment being a
// maximal run of the site's contributions, in file order, with no flusher beyond the site between two of them.  On a sorted file
// the batches' tile kernels and site walks compute exactly that from the locality sorting gives; on any other order the stream
// itself has to be replayed, which is what this file does -- for the whole file at once, all contigs, keyed (tid, pos):
//   1  per record t (file order): its first-CpG key if it flushes (F[t], 0 otherwise), its contribution count;
//   2  one (site key, t) pair per call of a contributing record, in file order, then a STABLE radix sort by site key (rocPRIM
//      through hipCUB, as in mth_sort.hip): every site's contributions, file order kept;
//   3  a range-maximum structure over F (block maxima of 32 records + a sparse table over the blocks): "is there a flusher
//      beyond key k between records t1 and t2" is one query;
//   4  one thread per site: walks the site's contributions, cuts segments where the query says so, keeps the last one that
//      qualifies (pdr.rs:163 / mhl.rs:165 coverage >= min_depth; fdrp.rs:216 stored reads >= min_depth);
//   5  the chosen segment evaluated: PDR in the same thread (counts), MHL and FDRP / qFDRP by one wave per site with the
//      reference's f32 expressions in the reference's order (mhl.rs:43-73 ascending l as the oracle fixes it; fdrp.rs:124-145,
//      qfdrp.rs:137-157 pairs in lexicographic order, reservoir with the counter-based draw of mth_fdrp.hip);
//   6  rows compacted in key order = the BTreeMap's.
// Nothing here is tuned: an unsorted Bismark file gives these measures mostly one-read segments, and nobody waits for them.
// Limits (loud, MTH_ERR_CAPACITY): < 2^31 calls of contributing records in one file; MHL reads with > 1024 CpGs or segment
// denominators >= 2^24; FDRP / qFDRP --max-depth > 16384 (as on the sorted path; up to 256 stored reads a site's slots live in LDS, beyond
// that in a per-wave row of HBM scratch).
#include <hipcub/hipcub.hpp>

#include "mth_ctx.h"

namespace mth {

constexpr int FO_BLK = 32;          // records per block of the range-maximum structure
constexpr int FO_MHL_MAXN = 1024;   // CpGs of a read the MHL evaluation holds
constexpr int FO_FDRP_SLOTS = 256;  // stored reads of a site the FDRP evaluation holds in LDS (more: rows_deep)
constexpr uint32_t FO_FDRP_DEPTH_MAX = 16384;   // the pair index of one site is 32-bit arithmetic (mth_fdrp.hip: FD_DEPTH_MAX)
constexpr int FO_WIN = 201;         // MAX_READ_LEN, fdrp.rs:10

struct FoArgs {
    const int32_t *tid, *start, *end;
    const uint8_t *mapq;
    const unsigned long long *off;
    const uint32_t *pos;
    uint32_t n_reads;
    int measure;                     // MTH_FO_PDR / MHL / FDRP
    uint32_t min_depth, min_cpgs, max_depth;
    int32_t min_overlap;
    uint32_t min_qual;
    unsigned long long seed;
    unsigned long long margin;       // 150 (pdr.rs:162) or 0
    // per record
    unsigned long long *F;           // first-CpG key + 1 of a flushing record, else 0
    uint32_t *cnt;                   // calls of a contributing record, else 0
    uint8_t *disc;                   // PDR: the read is discordant (readutil.rs:134-145)
    const unsigned long long *coff;  // exclusive scan of cnt
    // contributions
    unsigned long long *ckey;        // site key + 1
    uint32_t *cval;                  // record index
    uint32_t n_contrib;
    // range maximum
    const unsigned long long *bm;    // level 0: block maxima; level j at bm + j * n_blk
    uint32_t n_blk, n_lvl;
    // per contribution that starts a site: the chosen segment [sel_a, sel_b) (sel_b = 0: none), then the row
    uint32_t *sel_a, *sel_b;
    uint32_t *rowflag;
    uint32_t *rows_deep;             // FDRP with max_depth > FO_FDRP_SLOTS: max_depth slots per wave of the launch
    float *v0, *v1;
    uint32_t *c0, *c1;
    DevState *st;
};

__device__ __forceinline__ unsigned long long fo_key(int32_t tid, uint32_t pos) { return (((unsigned long long)(uint32_t)tid << 32) | pos) + 1ull; }

// the oracle's orc_sample_j (see mth_fdrp.hip)
__device__ __forceinline__ int32_t fo_sample_j(unsigned long long seed, int32_t tid, int32_t pos, int32_t total) {
    unsigned long long z = seed ^ (((unsigned long long)(uint32_t)tid << 32) | (uint32_t)pos);
    z += 0x9e3779b97f4a7c15ULL * (unsigned long long)(uint32_t)total;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z = z ^ (z >> 31);
    return (int32_t)(z % (unsigned long long)(uint32_t)total) + 1;
}

// ---- 1: per-record attributes --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_attrs(const FoArgs a) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_reads) return;
    const unsigned long long o0 = a.off[t], o1 = a.off[t + 1];
    const uint32_t n = (uint32_t)(o1 - o0);
    const uint32_t mq = a.mapq[t];
    bool flush, contrib;
    if (a.measure == MTH_FO_PDR) {             // pdr.rs:147-157: the filters come first, only a passing read flushes
        contrib = n >= a.min_cpgs && mq >= a.min_qual && n > 0u;
        flush = contrib;
    } else if (a.measure == MTH_FO_MHL) {      // mhl.rs:162 before 176-183
        flush = n > 0u;
        contrib = mq >= a.min_qual && n >= a.min_cpgs && n > 0u;
    } else {                                   // fdrp.rs:205-210
        contrib = mq >= a.min_qual && n > 0u;
        flush = contrib;
        // fdrp.rs:70-72: the one read on which the reference's window index is -1 (k_fdrp_guard in mth_fdrp.hip states the case)
        const int32_t s = a.start[t], e = a.end[t];
        if (contrib && n >= 2u && e - s >= 202 && e - s <= 402 && (a.pos[o0] & 0x7fffffffu) == ((uint32_t)(s - 1) & 0x7fffffffu))
            for (unsigned long long k = o0 + 1; k < o1; ++k)
                if ((a.pos[k] & 0x7fffffffu) == (uint32_t)(s + 201)) { atomicOr(&a.st->err, (uint32_t)ERRB_FDRPPANIC); break; }
    }
    uint8_t d = 0;
    unsigned long long f = 0;
    // a record without a contig (tid -1) that carries CpG calls: in the reference's i32 order (readutil.rs:290-314) it sorts BEFORE every
    // contig and flushes nothing, and its rows would reach the writer's tid2name(-1); the key below packs the tid as unsigned, which
    // would make it the largest.  Refused: the run fails with MTH_ERR_RANGE instead of a silently different flush order.
    if (n && a.tid[t] < 0 && (flush || contrib)) atomicOr(&a.st->err, (uint32_t)ERRB_RANGE);
    if (n) {
        const uint32_t w0 = a.pos[o0];
        if (flush) f = fo_key(a.tid[t], w0 & 0x7fffffffu);
        if (a.measure == MTH_FO_PDR && contrib)
            for (unsigned long long k = o0 + 1; k < o1; ++k) d |= (uint8_t)((a.pos[k] ^ w0) >> 31);
    }
    a.F[t] = f;
    a.cnt[t] = contrib ? n : 0u;
    a.disc[t] = d;
}

// ---- 2: contributions in file order -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_fill(const FoArgs a) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_reads) return;
    const uint32_t n = a.cnt[t];
    if (!n) return;
    const unsigned long long o0 = a.off[t], c0 = a.coff[t];
    const int32_t tid = a.tid[t];
    for (uint32_t k = 0; k < n; ++k) { a.ckey[c0 + k] = fo_key(tid, a.pos[o0 + k] & 0x7fffffffu); a.cval[c0 + k] = t; }
}

// ---- 3: range maximum over F ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_blockmax(const unsigned long long *__restrict__ F, uint32_t n, unsigned long long *__restrict__ bm, uint32_t n_blk) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blk) return;
    unsigned long long m = 0;
    for (uint32_t k = 0; k < (uint32_t)FO_BLK; ++k) { const uint32_t t = b * FO_BLK + k; if (t < n) m = max(m, F[t]); }
    bm[b] = m;
}
__global__ __launch_bounds__(256) void k_fo_level(const unsigned long long *__restrict__ prev, unsigned long long *__restrict__ cur, uint32_t n_blk, uint32_t half) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blk) return;
    cur[b] = max(prev[b], b + half < n_blk ? prev[b + half] : 0ull);
}
// max F[lo .. hi], lo <= hi
__device__ unsigned long long fo_rmq(const FoArgs &a, uint32_t lo, uint32_t hi) {
    unsigned long long m = 0;
    const uint32_t bl = lo / FO_BLK, bh = hi / FO_BLK;
    if (bl == bh) { for (uint32_t t = lo; t <= hi; ++t) m = max(m, a.F[t]); return m; }
    for (uint32_t t = lo; t < (bl + 1) * FO_BLK; ++t) m = max(m, a.F[t]);
    for (uint32_t t = bh * FO_BLK; t <= hi; ++t) m = max(m, a.F[t]);
    if (bl + 1 <= bh - 1) {
        const uint32_t x = bl + 1, len = bh - 1 - x + 1;
        const uint32_t j = 31u - (uint32_t)__builtin_clz(len);
        m = max(m, max(a.bm[(size_t)j * a.n_blk + x], a.bm[(size_t)j * a.n_blk + (bh - 1) - (1u << j) + 1]));
    }
    return m;
}

// ---- 4: one thread per site: segments, the last qualifying one; PDR evaluated here --------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_sites(const FoArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_contrib) return;
    a.rowflag[i] = 0u; a.sel_a[i] = 0u; a.sel_b[i] = 0u;
    const unsigned long long key = a.ckey[i];
    if (i > 0 && a.ckey[i - 1] == key) return;            // not the first contribution of its site
    const int32_t c = (int32_t)(uint32_t)(key - 1ull);
    // stored reads of [x, y) for FDRP: the arrivals add_read does not drop (fdrp.rs:55-63), at most max_depth of them (81-85)
    auto qualifies = [&](uint32_t x, uint32_t y) {
        if (a.measure != MTH_FO_FDRP) return y - x >= a.min_depth;
        uint32_t v = 0;
        for (uint32_t r = x; r < y; ++r) {
            const uint32_t t = a.cval[r];
            v += (FO_WIN + (a.start[t] - c) >= 0 && FO_WIN + (a.end[t] - c) <= 2 * FO_WIN) ? 1u : 0u;
        }
        return min(v, a.max_depth) >= a.min_depth;
    };
    uint32_t seg = i, best_a = 0, best_b = 0, prev_t = a.cval[i], j = i + 1;
    for (; j < a.n_contrib && a.ckey[j] == key; ++j) {
        const uint32_t t = a.cval[j];
        // a flusher strictly between the two contributions whose first CpG lies beyond the site (+ margin) closes the segment
        if (t - prev_t > 1u && fo_rmq(a, prev_t + 1u, t - 1u) > key + a.margin) {
            if (qualifies(seg, j)) { best_a = seg; best_b = j; }
            seg = j;
        }
        prev_t = t;
    }
    if (qualifies(seg, j)) { best_a = seg; best_b = j; }       // the final flush (pdr.rs:199-210)
    if (best_b == 0u) return;
    a.sel_a[i] = best_a; a.sel_b[i] = best_b;
    if (a.measure == MTH_FO_PDR) {
        uint32_t nd = 0;
        for (uint32_t r = best_a; r < best_b; ++r) nd += a.disc[a.cval[r]];
        const uint32_t nc = best_b - best_a - nd;
        a.v0[i] = (float)nd / ((float)nc + (float)nd);         // pdr.rs:47-49
        a.c0[i] = nc; a.c1[i] = nd;
        a.rowflag[i] = 1u;
    }
}

// ---- 5a: MHL of the chosen segments, one wave per site ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_mhl(const FoArgs a) {
    __shared__ uint32_t s_hn[4][FO_MHL_MAXN + 2], s_S[4][FO_MHL_MAXN + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hn = s_hn[wave], *S = s_S[wave];
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint32_t i0 = (blockIdx.x * 4u + (uint32_t)wave) * 64u; i0 < a.n_contrib; i0 += n_waves * 64u) {
        const uint32_t il = i0 + (uint32_t)lane;
        unsigned long long todo = __ballot(il < a.n_contrib && a.sel_b[il] != 0u);
        while (todo) {
            const uint32_t i = i0 + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t x = a.sel_a[i], y = a.sel_b[i];
            for (int l = lane; l < FO_MHL_MAXN + 2; l += 64) { hn[l] = 0u; S[l] = 0u; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t maxn = 0;
            bool bad = false;
            for (uint32_t r = x + (uint32_t)lane; r < y; r += 64u) {
                const uint32_t t = a.cval[r];
                const unsigned long long o0 = a.off[t], o1 = a.off[t + 1];
                const uint32_t n = (uint32_t)(o1 - o0);
                if (n > (uint32_t)FO_MHL_MAXN) { bad = true; continue; }
                maxn = max(maxn, n);
                atomicAdd(&hn[n], 1u);                                   // add_num_cpgs, mhl.rs:75-80
                uint32_t run = 0;                                        // get_stretch_info, readutil.rs:147-164: a run of m adds m - l + 1 at every l <= m
                for (unsigned long long k = o0; k <= o1; ++k) {
                    const bool meth = k < o1 && (a.pos[k] >> 31);
                    if (meth) { run += 1; continue; }
                    for (uint32_t l = 1; l <= run; ++l) atomicAdd(&S[l], run - l + 1u);
                    run = 0;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) maxn = max(maxn, (uint32_t)__shfl_xor((int)maxn, o, 64));
            if (__any(bad)) { if (lane == 0) atomicOr(&a.st->err, (uint32_t)ERRB_CAPACITY); continue; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0) {
                // denominators (mhl.rs:53-58: f32 adds of integers, exact below 2^24): D[l] = sum over reads with n >= l of (n - l + 1)
                unsigned long long cnt_ge = 0, sum_ge = 0;
                bool big = false;
                for (uint32_t l = maxn; l >= 1u; --l) {
                    cnt_ge += hn[l]; sum_ge += (unsigned long long)hn[l] * l;
                    const unsigned long long D = sum_ge - (unsigned long long)(l - 1u) * cnt_ge;
                    big |= D >= (1ull << 24);
                    hn[l] = (uint32_t)D;
                }
                if (big) atomicOr(&a.st->err, (uint32_t)ERRB_CAPACITY);
                float mhl = 0.0f, l_sum = 0.0f;
                for (uint32_t l = 1; l <= maxn; ++l) l_sum += (float)l;                           // mhl.rs:46-48
                for (uint32_t l = 1; l <= maxn; ++l)                                              // mhl.rs:50-69, ascending l (the oracle's order)
                    if (S[l] > 0u) mhl += ((float)l * (float)S[l]) / (float)hn[l];
                mhl /= l_sum;                                                                     // mhl.rs:71
                a.v0[i] = mhl; a.c0[i] = y - x; a.rowflag[i] = 1u;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- 5b: FDRP / qFDRP of the chosen segments, one wave per site -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_fdrp(const FoArgs a) {
    __shared__ uint32_t s_rows[4][FO_FDRP_SLOTS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_waves = gridDim.x * 4u;
    uint32_t *rows = a.rows_deep ? a.rows_deep + (size_t)(blockIdx.x * 4u + (uint32_t)wave) * a.max_depth : s_rows[wave];
    for (uint32_t i0 = (blockIdx.x * 4u + (uint32_t)wave) * 64u; i0 < a.n_contrib; i0 += n_waves * 64u) {
        const uint32_t il = i0 + (uint32_t)lane;
        unsigned long long todo = __ballot(il < a.n_contrib && a.sel_b[il] != 0u);
        while (todo) {
            const uint32_t i = i0 + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t x = a.sel_a[i], y = a.sel_b[i];
            const unsigned long long key = a.ckey[i] - 1ull;
            const int32_t c = (int32_t)(uint32_t)key, tid = (int32_t)(key >> 32);
            // add_read over the segment's arrivals (fdrp.rs:51-95), wave-uniform
            int32_t total = 0, sampled = 0;
            for (uint32_t r = x; r < y; ++r) {
                const uint32_t t = a.cval[r];
                const int32_t s = a.start[t], e = a.end[t];
                if (FO_WIN + (s - c) < 0) continue;                                               // fdrp.rs:58
                if (FO_WIN + (e - c) > 2 * FO_WIN) continue;                                      // fdrp.rs:61
                int slot;
                if (total < (int32_t)a.max_depth) { slot = total; total += 1; sampled += 1; }     // fdrp.rs:81-85
                else {                                                                            // fdrp.rs:87-94
                    total += 1;
                    const int32_t jr = fo_sample_j(a.seed, tid, c, total);
                    if (jr > (int32_t)a.max_depth) continue;
                    slot = jr - 1;
                }
                if (lane == 0) rows[slot] = t;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nS = sampled;
            const int P = (nS * (nS - 1)) >> 1;
            const int twoN = 2 * nS;
            const float bq = (float)(twoN - 1);
            uint32_t disc = 0;
            float q = 0.0f;
            for (int k0 = 0; k0 < P; k0 += 64) {
                const int k = min(k0 + lane, P - 1);
                // k -> (i, j), lexicographic (fdrp.rs:128): an f32 estimate of the row, then exact integer steps (beyond ~12 000 stored reads
                // the estimate can be two rows off near the end of the list: one conditional step each way is not enough)
                int pi = (int)((bq - __builtin_sqrtf(fmaxf(bq * bq - 8.0f * (float)k, 0.0f))) * 0.5f);
                pi = max(0, min(pi, nS - 2));
                int offp = (pi * (twoN - pi - 1)) >> 1;
                while (k < offp) { pi -= 1; offp = (pi * (twoN - pi - 1)) >> 1; }
                while (pi + 2 < nS && k >= (((pi + 1) * (twoN - pi - 2)) >> 1)) { pi += 1; offp = (pi * (twoN - pi - 1)) >> 1; }
                const int pj = k - offp + pi + 1;
                const uint32_t ti = rows[pi], tj = rows[pj];
                const int32_t si = a.start[ti], ei = a.end[ti], sj = a.start[tj], ej = a.end[tj];
                const int32_t mx = max(si, sj);
                const int32_t ov = min(ei, ej) - mx + 1;                                          // get_num_overlap_bases, fdrp.rs:97-107
                const bool pair_ok = (k0 + lane < P) & (max(ov, 0) >= a.min_overlap);             // fdrp.rs:134
                uint32_t ham = 0, ncpg = 0;
                unsigned long long ai = a.off[ti], aj = a.off[tj];
                const unsigned long long bi = a.off[ti + 1], bj = a.off[tj + 1];
                while (ai < bi && aj < bj) {
                    const uint32_t wi = a.pos[ai], wj = a.pos[aj];
                    const uint32_t pa = wi & 0x7fffffffu, pb = wj & 0x7fffffffu;
                    if (pa == pb) {
                        // a call outside the 403-slot array around c is not in it (the reference would index out of bounds: mth_fdrp.hip)
                        if ((uint32_t)((int32_t)pa - (c - FO_WIN)) <= 2u * FO_WIN) {
                            ncpg += 1;                                                            // qfdrp.rs:109-119
                            ham += ((int32_t)pa >= mx && ((wi ^ wj) >> 31)) ? 1u : 0u;            // fdrp.rs:114-115
                        }
                        ++ai; ++aj;
                    } else if (pa < pb) ++ai; else ++aj;
                }
                disc += (pair_ok && ham != 0u) ? 1u : 0u;                                         // fdrp.rs:138-140
                const float term = pair_ok ? (float)ham / (float)ncpg : 0.0f;                     // qfdrp.rs:152; +0.0 for skipped pairs
                float xsum = (lane == 0) ? q + term : term;
                const int steps = min(64, P - k0) - 1;
                for (int stp = 0; stp < steps; ++stp)
                    xsum = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xsum), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)) + term;
                q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xsum), steps));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) disc += __shfl_xor(disc, o, 64);
            if (lane == 0) {
                const unsigned long long prod = (unsigned long long)(long long)nS * (unsigned long long)((long long)nS - 1);
                const float den = (float)prod / 2.0f;                                             // fdrp.rs:143
                a.v0[i] = (float)disc / den; a.v1[i] = q / den; a.c0[i] = (uint32_t)nS; a.rowflag[i] = 1u;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ __launch_bounds__(256) void k_fo_narrow(const unsigned long long *__restrict__ r64, uint32_t *__restrict__ r32, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) r32[i] = (uint32_t)r64[i];
}

// ---- 6: rows in key order ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fo_emit(const FoArgs a, const uint32_t *__restrict__ rank, int32_t *__restrict__ o_tid, int32_t *__restrict__ o_pos,
                                                 float *__restrict__ o_v0, float *__restrict__ o_v1, uint32_t *__restrict__ o_c0, uint32_t *__restrict__ o_c1) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_contrib || !a.rowflag[i]) return;
    const uint32_t o = rank[i];
    const unsigned long long key = a.ckey[i] - 1ull;
    o_tid[o] = (int32_t)(key >> 32); o_pos[o] = (int32_t)(uint32_t)key;
    o_v0[o] = a.v0[i]; o_v1[o] = a.v1[i]; o_c0[o] = a.c0[i]; o_c1[o] = a.c1[i];
}

}  // namespace mth

using namespace mth;

extern "C" {

int mth_fileorder_run(mth_ctx_t *ctx, const mth_fileorder_params_t *p) {
    if (!ctx || !p || p->measure < MTH_FO_PDR || p->measure > MTH_FO_FDRP) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    ctx->fo_rows = 0;
    const uint64_t R = ctx->dec_reads;
    if (R == 0) return MTH_OK;
    if (R >= (1ull << 31)) return fail(ctx, MTH_ERR_CAPACITY, "file-order measures: more than 2^31 records");
    if (p->measure == MTH_FO_FDRP && p->max_depth > FO_FDRP_DEPTH_MAX)
        return fail(ctx, MTH_ERR_CAPACITY, "FDRP / qFDRP max_depth above 16384 (the pair index of one site is 32-bit arithmetic)");
    const uint32_t n = (uint32_t)R, nb = (n + 255) / 256;
    DevBuf F, cnt, disc, coff, ckey, ckey2, cval, cval2, tmp, bm, rank, deep;
    auto drop = [&]() { for (DevBuf *b : {&F, &cnt, &disc, &coff, &ckey, &ckey2, &cval, &cval2, &tmp, &bm, &rank, &deep}) b->release(); };
#define FO_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { drop(); return fail(ctx, MTH_ERR_HIP, #call, e__); } } while (0)
    FO_HIP(F.reserve((size_t)n * 8 + 8, s)); FO_HIP(cnt.reserve((size_t)n * 4 + 4, s)); FO_HIP(disc.reserve((size_t)n + 4, s));
    FO_HIP(coff.reserve(((size_t)n + 1) * 8, s));
    FoArgs a;
    memset(&a, 0, sizeof a);
    a.tid = ctx->dec_tid.as<int32_t>(); a.start = ctx->dec_start.as<int32_t>(); a.end = ctx->dec_end.as<int32_t>();
    a.mapq = ctx->dec_mapq.as<uint8_t>(); a.off = ctx->dec_off.as<unsigned long long>(); a.pos = ctx->dec_pos.as<uint32_t>();
    a.n_reads = n; a.measure = p->measure; a.min_depth = p->min_depth; a.min_cpgs = p->min_cpgs; a.max_depth = p->max_depth;
    a.min_overlap = p->min_overlap; a.min_qual = p->min_qual; a.seed = p->seed;
    a.margin = p->measure == MTH_FO_PDR ? (unsigned long long)PDR_FLUSH_MARGIN : 0ull;
    a.F = F.as<unsigned long long>(); a.cnt = cnt.as<uint32_t>(); a.disc = disc.as<uint8_t>(); a.st = ctx->d_state;
    hipLaunchKernelGGL(k_fo_attrs, dim3(nb), dim3(256), 0, s, a);
    unsigned long long total = 0;
    {
        const int rc = scan_u32_to_u64(ctx, cnt.as<uint32_t>(), n, 0ull, coff.as<unsigned long long>(), &total);
        if (rc) { drop(); return rc; }
    }
    if (total == 0) { drop(); return MTH_OK; }
    if (total >= (1ull << 31)) { drop(); return fail(ctx, MTH_ERR_CAPACITY, "file-order measures: more than 2^31 CpG calls of contributing records"); }
    const uint32_t M = (uint32_t)total, mb = (M + 255) / 256;
    a.coff = coff.as<unsigned long long>(); a.n_contrib = M;
    FO_HIP(ckey.reserve((size_t)M * 8, s)); FO_HIP(ckey2.reserve((size_t)M * 8, s)); FO_HIP(cval.reserve((size_t)M * 4, s)); FO_HIP(cval2.reserve((size_t)M * 4, s));
    a.ckey = ckey.as<unsigned long long>(); a.cval = cval.as<uint32_t>();
    hipLaunchKernelGGL(k_fo_fill, dim3(nb), dim3(256), 0, s, a);
    size_t tmp_bytes = 0;
    FO_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ckey.as<unsigned long long>(), ckey2.as<unsigned long long>(), cval.as<uint32_t>(), cval2.as<uint32_t>(), (int)M, 0, 64, s));
    FO_HIP(tmp.reserve(tmp_bytes + 16, s));
    FO_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, ckey.as<unsigned long long>(), ckey2.as<unsigned long long>(), cval.as<uint32_t>(), cval2.as<uint32_t>(), (int)M, 0, 64, s));
    a.ckey = ckey2.as<unsigned long long>(); a.cval = cval2.as<uint32_t>();
    // range maximum over F: block maxima, then the sparse table's levels
    const uint32_t n_blk = (n + FO_BLK - 1) / FO_BLK;
    uint32_t n_lvl = 1;
    while ((1u << n_lvl) <= n_blk) ++n_lvl;
    FO_HIP(bm.reserve((size_t)n_blk * n_lvl * 8, s));
    hipLaunchKernelGGL(k_fo_blockmax, dim3((n_blk + 255) / 256), dim3(256), 0, s, F.as<unsigned long long>(), n, bm.as<unsigned long long>(), n_blk);
    for (uint32_t j = 1; j < n_lvl; ++j)
        hipLaunchKernelGGL(k_fo_level, dim3((n_blk + 255) / 256), dim3(256), 0, s, bm.as<unsigned long long>() + (size_t)(j - 1) * n_blk,
                           bm.as<unsigned long long>() + (size_t)j * n_blk, n_blk, 1u << (j - 1));
    a.bm = bm.as<unsigned long long>(); a.n_blk = n_blk; a.n_lvl = n_lvl;
    // per-contribution work arrays (reused through the context's buffers of the site walks)
    FO_HIP(ctx->fo_sel.reserve((size_t)M * 8, s)); FO_HIP(ctx->fo_flag.reserve((size_t)M * 4, s)); FO_HIP(ctx->fo_tmp.reserve((size_t)M * 16, s));
    a.sel_a = ctx->fo_sel.as<uint32_t>(); a.sel_b = a.sel_a + M; a.rowflag = ctx->fo_flag.as<uint32_t>();
    a.v0 = ctx->fo_tmp.as<float>(); a.v1 = a.v0 + M; a.c0 = reinterpret_cast<uint32_t *>(a.v1 + M); a.c1 = a.c0 + M;
    FO_HIP(hipMemsetAsync(ctx->fo_tmp.p, 0, (size_t)M * 16, s));
    hipLaunchKernelGGL(k_fo_sites, dim3(mb), dim3(256), 0, s, a);
    if (p->measure == MTH_FO_MHL) hipLaunchKernelGGL(k_fo_mhl, dim3(std::min<uint32_t>((M + 255) / 256, 2048u)), dim3(256), 0, s, a);
    if (p->measure == MTH_FO_FDRP) {
        uint32_t grid = std::min<uint32_t>((M + 255) / 256, 2048u);
        a.rows_deep = nullptr;
        if (p->max_depth > (uint32_t)FO_FDRP_SLOTS) {
            // a row of max_depth record numbers per wave in HBM; as many waves as ~256 MB of rows allow (such sites are bound by their
            // pair count, not by the number of waves)
            grid = std::max<uint32_t>(1u, std::min<uint32_t>(grid, (uint32_t)(((size_t)256 << 20) / ((size_t)p->max_depth * 4 * 4))));
            FO_HIP(deep.reserve((size_t)grid * 4 * p->max_depth * 4, s));
            a.rows_deep = deep.as<uint32_t>();
        }
        hipLaunchKernelGGL(k_fo_fdrp, dim3(grid), dim3(256), 0, s, a);
    }
    // rank of every row = exclusive scan of the flags
    FO_HIP(rank.reserve(((size_t)M + 1) * 8, s));
    unsigned long long rows = 0;
    {
        const int rc = scan_u32_to_u64(ctx, a.rowflag, M, 0ull, rank.as<unsigned long long>(), &rows);
        if (rc) { drop(); return rc; }
    }
    FO_HIP(ctx->fo_out.reserve((size_t)rows * 24 + 64, s));
    int32_t *o_tid = ctx->fo_out.as<int32_t>(), *o_pos = o_tid + rows;
    float *o_v0 = reinterpret_cast<float *>(o_pos + rows), *o_v1 = o_v0 + rows;
    uint32_t *o_c0 = reinterpret_cast<uint32_t *>(o_v1 + rows), *o_c1 = o_c0 + rows;
    // (the 64-bit ranks are narrowed on the fly: rows < 2^31)
    {
        DevBuf r32;
        FO_HIP(r32.reserve((size_t)M * 4 + 4, s));
        hipLaunchKernelGGL(k_fo_narrow, dim3(mb), dim3(256), 0, s, rank.as<unsigned long long>(), r32.as<uint32_t>(), M);
        hipLaunchKernelGGL(k_fo_emit, dim3(mb), dim3(256), 0, s, a, r32.as<uint32_t>(), o_tid, o_pos, o_v0, o_v1, o_c0, o_c1);
        const hipError_t e = hipStreamSynchronize(s);
        r32.release();
        if (e != hipSuccess) { drop(); return fail(ctx, MTH_ERR_HIP, "file-order emit", e); }
    }
#undef FO_HIP
    drop();
    ctx->fo_rows = rows;
    return sync_and_check(ctx);
}

int mth_fileorder_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos, float *v0, float *v1, uint32_t *c0, uint32_t *c1) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    const uint64_t n = ctx->fo_rows;
    if (n_rows) *n_rows = n;
    if (n == 0) return MTH_OK;
    const uint8_t *base = static_cast<const uint8_t *>(ctx->fo_out.p);
    void *dst[6] = {tid, pos, v0, v1, c0, c1};
    for (int k = 0; k < 6; ++k)
        if (dst[k]) MTH_HIP(ctx, hipMemcpy(dst[k], base + (size_t)k * n * 4, n * 4, hipMemcpyDeviceToHost));
    return MTH_OK;
}

}  // extern "C"
