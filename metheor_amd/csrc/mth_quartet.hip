// mth_quartet.hip -- ME / PM: per-quartet 16-bin epiallele histograms on gfx950.
//
// Reference: readutil.rs:97-132 (get_cpg_quartets_and_patterns: every window of 4 consecutive CpGs of
// a read, pattern = 8*m0 + 4*m1 + 2*m2 + m3), me.rs:90-132 / pm.rs:85-128 (global
// HashMap<Quartet,[u32;16]> over reads with mapq >= min_qual, never flushed), me.rs:42-55
// (me = -0.25 * sum_{c>0} p*log2(p)), pm.rs:42-51 (pm = 1 - sum p*p), p = c as f32 / total as f32,
// bins visited 0..15 in order; min_depth is applied when rows are written (me.rs:82).
//
// Device design.  A quartet is owned by the position of its first CpG, so -- as for the per-site counters of PDR -- the
// tile (8192 / 16384 / 32768 bp, by the batch's call density) that contains p1 sees ALL updates of its quartets: k_quartet_tile keeps a 512-slot hash table in LDS
// (64-bit key, sixteen 16-bit bins packed in 8 words), fed by the tile's candidate reads (linear read index), and emits
// its rows straight from LDS (one global atomic per tile to claim the output range).  The first version sent every
// update to a global table: 6 M CAS + 6 M adds through L2 per 10 M reads were what it spent its time on.  Tiles that do
// not fit (> 65535 candidate reads: 16-bit bins; more distinct quartets than the table holds) are flagged and take
// that global path, restricted to their quartets: a global open-addressing table (64-bit key CAS, then one L2 atomic add on the
// bin) sized from an exact counting pre-pass.  Key = p1 (31 bits) | three position deltas (11 bits
// each): consecutive CpGs of one read further than 2047 bp apart are refused (MTH_ERR_CAPACITY).
// A quartet is owned by the batch whose region contains p1 (same rule as sites), so region / contig
// sharding needs no exchange.  Each tile sorts its rows in LDS and the fetch walks the tiles in order, so rows come out
// sorted by (tid, p1..p4); rows of tiles that took the global path follow in table order (the reference's order is
// HashMap-random, compare as sets).
#include <algorithm>
#include <cmath>

#include "mth_ctx.h"
#include "mth_scan.h"
#include "mth_tile_dev.h"

namespace mth {

constexpr unsigned long long QKEY_EMPTY = ~0ull;
// tile kernel: reference positions per tile, LDS table slots, threads.  Measured on S-chr19-10M (profiles/r01_quartet_tile.md):
// wider tiles re-read fewer halo reads and clear LDS less often, a smaller table lets more tiles share a CU (25 KB each).
constexpr int QT_NC = 8;   // calls of a read held in registers
constexpr int QT_QCAP = 2048;   // candidate reads per fill of the contributor queue
constexpr int QT_OCC = 6;  // waves per SIMD (LDS: 23 KiB per workgroup = 6 per CU)
constexpr int QT_S = 512, QT_B = 256, QT_U = 2, QT_CHUNK = 512, QT_GRID = 8192;   // (the tile width is a template parameter: 8192 / 16384 / 32768, chosen per batch)
constexpr int Q_STATE_WORDS = 8;
constexpr int Q_QUEUE_MAX = 4096;       // queued batches between two resolves (their snapshots: 256 KB)

__device__ __forceinline__ unsigned long long qhash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// slot of a key in the tile kernel's LDS table: two 32-bit multiplies (the 64-bit mixer above costs ~30 VALU)
__device__ __forceinline__ uint32_t qslot(unsigned long long key) {
    uint32_t h = (uint32_t)(key >> 33) * 0x9E3779B1u ^ (uint32_t)key * 0x85EBCA6Bu;
    h ^= h >> 15;
    return h & (QT_S - 1);
}

// global path: upper bound of the number of (read, window) updates: sum over passing reads of max(0, n - 3)
__global__ __launch_bounds__(256) void k_quartet_bound(const uint32_t *__restrict__ cpg_off,
                                                       const uint8_t *__restrict__ mapq, uint32_t n_reads,
                                                       uint8_t min_qual, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_reads; i += gridDim.x * 256) {
        const uint32_t n = cpg_off[i + 1] - cpg_off[i];
        if (n >= 4 && mapq[i] >= min_qual) acc += n - 3;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ unsigned long long ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

constexpr uint32_t QPROBE_MAX = 1024;
__global__ __launch_bounds__(256) void k_quartet_insert(const uint32_t *__restrict__ cpg_off,
                                                        const uint32_t *__restrict__ cpg_pos,
                                                        const uint8_t *__restrict__ mapq, uint32_t n_reads,
                                                        uint8_t min_qual, int32_t region_beg, int32_t region_end,
                                                        unsigned long long *__restrict__ keys,
                                                        uint32_t *__restrict__ hist, unsigned long long mask,
                                                        unsigned long long *__restrict__ overflow, DevState *__restrict__ st,
                                                        const uint32_t *__restrict__ tile_flag /* nullptr: every quartet */,
                                                        int tile_shift /* log2 of the tile width the flags were made with */,
                                                        uint4 *__restrict__ wide_pos, uint32_t *__restrict__ wide_pat,
                                                        unsigned long long wide_cap, unsigned long long *__restrict__ wide_n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_reads) return;
    const uint32_t o0 = cpg_off[i], o1 = cpg_off[i + 1];
    if (o1 - o0 < 4 || mapq[i] < min_qual) return;          // readutil.rs:101, me.rs:115
    uint32_t a = cpg_pos[o0], b = cpg_pos[o0 + 1], c = cpg_pos[o0 + 2];
    for (uint32_t k = o0 + 3; k < o1; ++k) {                // readutil.rs:105-129
        const uint32_t d = cpg_pos[k];
        const int32_t p1 = (int32_t)(a & 0x7fffffffu);
        if (p1 >= region_beg && p1 < region_end && (!tile_flag || tile_flag[(uint32_t)(p1 - region_beg) >> tile_shift])) {
            const uint32_t d2 = (b & 0x7fffffffu) - (a & 0x7fffffffu), d3 = (c & 0x7fffffffu) - (b & 0x7fffffffu),
                           d4 = (d & 0x7fffffffu) - (c & 0x7fffffffu);
            const unsigned long long key = ((unsigned long long)(uint32_t)p1 << 33) | ((unsigned long long)d2 << 22) |
                                           ((unsigned long long)d3 << 11) | (unsigned long long)d4;
            const uint32_t pat = ((a >> 31) << 3) | ((b >> 31) << 2) | ((c >> 31) << 1) | (d >> 31);
            if (d2 - 1u >= 0x7fffffffu || d3 - 1u >= 0x7fffffffu || d4 - 1u >= 0x7fffffffu) {
                atomicOr(&st->err, (uint32_t)ERRB_SPAN);                 // calls of a read not in ascending position order
            } else if (d2 >= 2048u || d3 >= 2048u || d4 >= 2048u || key == QKEY_EMPTY) {
                // CpGs of the window >= 2048 bp apart do not fit the packed key: the instance goes to the wide list
                // (aggregated by k_quartet_wide_insert under a 128-bit key); counted even when the list is full (the host redoes)
                const unsigned long long w = atomicAdd(wide_n, 1ull);
                if (w < wide_cap) {
                    wide_pos[w] = make_uint4((uint32_t)p1, b & 0x7fffffffu, c & 0x7fffffffu, d & 0x7fffffffu);
                    wide_pat[w] = pat;
                }
            } else {
                unsigned long long h = qhash(key) & mask;
                // The table is sized for the DISTINCT quartets one expects (half the instances), not for the instances:
                // a bounded probe, and a flag that makes the host redo the pass with a larger table, keep that safe.
                uint32_t probes = 0;
                bool placed = false;
                while (probes++ < QPROBE_MAX) {
                    const unsigned long long cur = atomicCAS(&keys[h], QKEY_EMPTY, key);
                    if (cur == QKEY_EMPTY || cur == key) { placed = true; break; }
                    h = (h + 1) & mask;
                }
                if (placed) atomicAdd(&hist[h * 16 + pat], 1u);         // me.rs:121-125
                else *overflow = 1ull;
            }
        }
        a = b; b = c; c = d;
    }
}

// Wide quartets (some pair of consecutive CpGs >= 2048 bp apart: reference skips, deletions, long reads) under a 128-bit
// key: k0 = p1 << 32 | p2 claims the slot with a CAS, k1 = p3 << 32 | p4 is published right after by the claimer.  A lane that
// finds its k0 in a slot waits for the slot's k1 to appear and compares; the claimer's store comes before every wait in
// program order (same iteration, straight-line), so lanes of one wave cannot wait on each other's unpublished slot.
__global__ __launch_bounds__(256) void k_quartet_wide_insert(const uint4 *__restrict__ wide_pos, const uint32_t *__restrict__ wide_pat,
                                                             unsigned long long n, unsigned long long *__restrict__ k0,
                                                             unsigned long long *__restrict__ k1, uint32_t *__restrict__ hist,
                                                             unsigned long long mask, unsigned long long *__restrict__ overflow) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 p = wide_pos[i];
    const unsigned long long key0 = ((unsigned long long)p.x << 32) | p.y, key1 = ((unsigned long long)p.z << 32) | p.w;
    unsigned long long h = qhash(key0 ^ qhash(key1)) & mask;
    for (uint32_t probes = 0; probes < QPROBE_MAX; ++probes) {
        const unsigned long long cur = atomicCAS(&k0[h], QKEY_EMPTY, key0);
        if (cur == QKEY_EMPTY) __hip_atomic_store(&k1[h], key1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == QKEY_EMPTY || cur == key0) {
            unsigned long long v;
            do { v = __hip_atomic_load(&k1[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (v == QKEY_EMPTY);
            if (v == key1) { atomicAdd(&hist[h * 16 + wide_pat[i]], 1u); return; }
        }
        h = (h + 1) & mask;
    }
    *overflow = 1ull;
}

// rows per block of 256*8 slots
constexpr int QE_PER = 8;
__global__ __launch_bounds__(256) void k_quartet_blockcount(const unsigned long long *__restrict__ keys,
                                                            unsigned long long n_slots, uint32_t *__restrict__ blk) {
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * QE_PER;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < QE_PER; ++k) m += (s0 + k < n_slots && keys[s0 + k] != QKEY_EMPTY) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// me.rs:42-55 and pm.rs:42-51 with the reference's operation order.  Plain operators, and the whole
// engine is compiled with -ffp-contract=off: with hipcc's default (contract=fast) `pm - p*p` became an
// FMA -- HIP's __fmul_rn/__fsub_rn header functions did not prevent it -- and 2.8 % of PM values were
// one ulp off the reference expression (measured; tools/pm_probe.py).
__device__ __forceinline__ void quartet_values(const uint32_t *c, float &me, float &pm, uint32_t &total) {
#pragma clang fp contract(off)
    total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += c[k];
    const float tf = (float)total;
    me = 0.0f;
    pm = 1.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float p = (float)c[k] / tf;
        if (c[k] > 0) {
            const float t = p * log2f(p);
            me = me + t;
        }
        const float sq = p * p;
        pm = pm - sq;
    }
    me = me * -0.25f;
}

__global__ __launch_bounds__(256) void k_quartet_emit(const unsigned long long *__restrict__ keys,
                                                      const unsigned long long *__restrict__ keys1 /* wide table: p3 << 32 | p4; else nullptr */,
                                                      const uint32_t *__restrict__ hist, unsigned long long n_slots,
                                                      const uint32_t *__restrict__ blk,
                                                      const unsigned long long *__restrict__ q_base,
                                                      int32_t *__restrict__ out_pos, uint32_t *__restrict__ out_cnt,
                                                      float *__restrict__ out_me, float *__restrict__ out_pm,
                                                      uint32_t *__restrict__ out_depth) {
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * QE_PER;
    uint32_t m = 0;
    unsigned long long kk[QE_PER];
#pragma unroll
    for (int k = 0; k < QE_PER; ++k) {
        kk[k] = s0 + k < n_slots ? keys[s0 + k] : QKEY_EMPTY;
        m += kk[k] != QKEY_EMPTY ? 1u : 0u;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    __shared__ uint32_t ws[5];
    if (lane == 63) ws[wave + 1] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { ws[0] = 0; for (int w = 1; w <= 4; ++w) ws[w] += ws[w - 1]; }
    __syncthreads();
    unsigned long long o = *q_base + blk[blockIdx.x] + ws[wave] + incl - m;
#pragma unroll
    for (int k = 0; k < QE_PER; ++k) {
        if (kk[k] == QKEY_EMPTY) continue;
        const unsigned long long key = kk[k];
        int32_t p1, p2, p3, p4;
        if (keys1) {
            const unsigned long long key1 = keys1[s0 + k];
            p1 = (int32_t)(key >> 32); p2 = (int32_t)(uint32_t)key; p3 = (int32_t)(key1 >> 32); p4 = (int32_t)(uint32_t)key1;
        } else {
            p1 = (int32_t)(key >> 33);
            p2 = p1 + (int32_t)((key >> 22) & 2047u); p3 = p2 + (int32_t)((key >> 11) & 2047u); p4 = p3 + (int32_t)(key & 2047u);
        }
        reinterpret_cast<int4 *>(out_pos)[o] = make_int4(p1, p2, p3, p4);
        uint32_t c[16];
        const uint4 *src = reinterpret_cast<const uint4 *>(hist + (s0 + k) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = src[q];
            c[4 * q] = x.x; c[4 * q + 1] = x.y; c[4 * q + 2] = x.z; c[4 * q + 3] = x.w;
            reinterpret_cast<uint4 *>(out_cnt + o * 16)[q] = x;
        }
        float me, pm;
        uint32_t total;
        quartet_values(c, me, pm, total);
        out_me[o] = me; out_pm[o] = pm; out_depth[o] = total;
        ++o;
    }
}


// ---- tile kernel ---------------------------------------------------------------------------------------------------
struct QTileArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    int32_t region_beg, region_end, idx_base, max_span;
    uint32_t n_reads, ntiles, n_cpgs;
    uint8_t min_qual, force_heavy;            // force_heavy: tests send every tile down the global path
    unsigned long long *row_total;            // rows claimed so far (all batches, gaps included)
    unsigned long long row_cap;               // rows the output holds; a range beyond it is claimed but not written ...
    unsigned long long *unfit;                // ... and reported here (the host redoes the batch with the exact size)
    unsigned long long *n_heavy;              // tiles left to the global path
    uint32_t *tile_flag;                      // per tile of the batch: 1 = left to the global path
    unsigned long long *tile_row0;            // per tile: first row of its range ...
    uint32_t *tile_rows;                      // ... and its length (the fetch walks tiles in order: rows come out sorted)
    int32_t *out_pos; uint32_t *out_cnt; float *out_me, *out_pm; uint32_t *out_depth;
    DevState *st;
};
template <int QT_W>
__global__ __launch_bounds__(QT_B, QT_OCC) void k_quartet_tile(const QTileArgs a) {
    __shared__ unsigned long long keys[QT_S];  // the hash table; later its keys in bucket order
    __shared__ uint32_t bcnt[QT_B];            // bucket sort on p1: keys per bucket
    // the contributor queue of the read phases (16-bit read numbers) shares its LDS with two arrays of the row phase: the first
    // rank of each bucket, and the slot each sorted key came from
    __shared__ uint32_t q_or_sort[QT_QCAP / 2];
    static_assert(QT_QCAP / 2 >= QT_B + QT_S / 2, "bbase and sslot fit under the queue");
    uint16_t *const rq = reinterpret_cast<uint16_t *>(q_or_sort);
    uint32_t *const bbase = q_or_sort;
    uint16_t *const sslot = reinterpret_cast<uint16_t *>(q_or_sort + QT_B);
    __shared__ uint32_t s_qn;
    __shared__ uint32_t bins[QT_S * 8];        // bin 2w in the low half of word w, bin 2w+1 in the high half
    __shared__ uint32_t s_heavy, ws[QT_B / 64 + 1];
    __shared__ unsigned long long s_row0;
    __shared__ unsigned long long s_chunk_pos, s_chunk_end;   // rows claimed from the global counter, handed out tile by tile
    const int tid = threadIdx.x;
    if (tid == 0) { s_chunk_pos = 0; s_chunk_end = 0; }
    // Persistent workgroups: each takes tiles t, t + grid, ... and claims output rows for several tiles at once.  (One claim
    // per tile was the kernel's floor on sparse WGBS: same-address returning atomics serialise at ~12 ns each, and a
    // human genome is 378 k tiles.)  The unused tail of a chunk stays a gap; the fetch walks per-tile ranges.
    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    __syncthreads();                                    // the previous tile's rows have left LDS
    const int32_t T0 = a.region_beg + (int32_t)(t * QT_W);
    const int32_t T1 = (int32_t)min((int64_t)T0 + QT_W, (int64_t)a.region_end);
    // candidate reads: a read with a CpG at p1 >= T0 starts after T0 - max_span and not after T1 - 1
    const uint32_t lo = min(a.idx[((uint32_t)T0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
    const uint32_t hi = min(a.idx[(((uint32_t)T0 + (uint32_t)QT_W - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
    if (lo >= hi) {                                             // nothing starts here: no LDS work at all
        if (tid == 0) { a.tile_flag[t] = 0u; a.tile_rows[t] = 0u; a.tile_row0[t] = 0ull; }
        continue;
    }
    for (int i = tid; i < QT_S; i += QT_B) keys[i] = QKEY_EMPTY;
    bcnt[tid] = 0;
    for (int i = tid; i < QT_S * 8; i += QT_B) bins[i] = 0;
    if (tid == 0) { s_heavy = (hi - lo > 65535u || a.force_heavy) ? 1u : 0u; s_qn = 0u; }      // a bin counts at most one update per candidate read
    __syncthreads();
    if (!s_heavy) {
        uint32_t bad = 0;
        // one window of four consecutive calls (readutil.rs:105-129) into the tile's table; the quartet belongs to the tile of p1
        auto window = [&](const uint32_t x, const uint32_t y, const uint32_t z, const uint32_t w) {
            const int32_t p1 = (int32_t)(x & 0x7fffffffu);
            if (p1 < T0 || p1 >= T1) return;
            const uint32_t d2 = (y & 0x7fffffffu) - (x & 0x7fffffffu), d3 = (z & 0x7fffffffu) - (y & 0x7fffffffu),
                           d4 = (w & 0x7fffffffu) - (z & 0x7fffffffu);
            const unsigned long long key = ((unsigned long long)(uint32_t)p1 << 33) | ((unsigned long long)d2 << 22) |
                                           ((unsigned long long)d3 << 11) | (unsigned long long)d4;
            if (d2 - 1u >= 2047u || d3 - 1u >= 2047u || d4 - 1u >= 2047u || key == QKEY_EMPTY) {
                s_heavy = 1u;                                        // CpGs >= 2048 bp apart (or out of order): the global path sorts it out
                return;
            }
            const uint32_t pat = ((x >> 31) << 3) | ((y >> 31) << 2) | ((z >> 31) << 1) | (w >> 31);
            uint32_t h = qslot(key), probes = 0;
            bool placed = false;
            while (probes++ < (uint32_t)QT_S) {
                const unsigned long long cur = atomicCAS(&keys[h], QKEY_EMPTY, key);
                if (cur == QKEY_EMPTY || cur == key) { placed = true; break; }
                h = (h + 1) & (QT_S - 1);
            }
            if (placed) atomicAdd(&bins[h * 8 + (pat >> 1)], (pat & 1u) ? 0x10000u : 1u);      // me.rs:121-125
            else s_heavy = 1u;                                   // more distinct quartets than slots
        };
        // Two phases per stretch of QT_QCAP candidate reads.  Phase 1, every candidate (offsets one round ahead, mapq): the reads with
        // >= 4 CpGs that pass mapq (readutil.rs:101, me.rs:115) -- a third of config 2's reads, a twentieth at WGBS density -- are
        // queued.  Phase 2, the queue with every lane live: the read's first QT_NC calls as two 16-byte loads from a per-read base
        // (the PDR tile kernel's form; a wave that holds one of the batch's last reads takes clamped single loads instead), its
        // windows from registers; only a read with more than QT_NC calls goes back to memory, one dependent load per further window.
        // (One phase over all candidates ran the window code with 5-30 % of the lanes live; measured gain of the split: config 2
        // 0.124 -> 0.121 ms, config-3 density 0.160 -> 0.158 -- the tile's fixed chain, not the windows, is what this kernel waits for.)
        for (uint32_t c0 = lo; c0 < hi; c0 += QT_QCAP) {
            const uint32_t c1 = min(c0 + (uint32_t)QT_QCAP, hi);
            if (c0 != lo) {
                __syncthreads();                            // the previous stretch's queue is done with
                if (tid == 0) s_qn = 0u;
                __syncthreads();
            }
            uint32_t o0s[QT_U], o1s[QT_U];
#pragma unroll
            for (int u = 0; u < QT_U; ++u) {
                const uint32_t ii = min(c0 + (uint32_t)u * QT_B + tid, c1 - 1);
                o0s[u] = a.cpg_off[ii]; o1s[u] = a.cpg_off[ii + 1];
            }
            for (uint32_t b0 = c0; b0 < c1; b0 += QT_B * QT_U) {
                uint32_t mq[QT_U], o0n[QT_U], o1n[QT_U];
#pragma unroll
                for (int u = 0; u < QT_U; ++u) {
                    const uint32_t i = b0 + (uint32_t)u * QT_B + tid;
                    mq[u] = a.read_mapq[min(i, c1 - 1)];
                    const uint32_t in = min(i + (uint32_t)QT_U * QT_B, c1 - 1);
                    o0n[u] = a.cpg_off[in]; o1n[u] = a.cpg_off[in + 1];
                }
#pragma unroll
                for (int u = 0; u < QT_U; ++u) {
                    const uint32_t i = b0 + (uint32_t)u * QT_B + tid;
                    const bool contrib = i < c1 && o1s[u] - o0s[u] >= 4 && mq[u] >= a.min_qual;
                    const unsigned long long bal = __ballot(contrib);
                    if (bal) {
                        uint32_t base = 0;
                        if ((tid & 63) == 0) base = atomicAdd(&s_qn, (uint32_t)__builtin_popcountll(bal));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (contrib) rq[base + (uint32_t)__builtin_popcountll(bal & ((1ull << (tid & 63)) - 1ull))] = (uint16_t)(i - c0);
                    }
                }
#pragma unroll
                for (int u = 0; u < QT_U; ++u) { o0s[u] = o0n[u]; o1s[u] = o1n[u]; }
            }
            __syncthreads();
            const uint32_t qn = s_qn;
            for (uint32_t j0 = 0; j0 < qn; j0 += QT_B) {
                const uint32_t j = j0 + tid;
                const bool act = j < qn;
                const uint32_t i = c0 + (act ? (uint32_t)rq[j] : 0u);
                const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
                const uint32_t n_calls = act ? o1 - o0 : 0u;
                uint32_t vv[QT_NC];
                static_assert(QT_NC == 8, "two 16-byte loads per read");
                if (__all(!act || (unsigned long long)o0 + QT_NC <= (unsigned long long)a.n_cpgs)) {
                    if (act) {
                        const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0), y = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0 + 4);
                        vv[0] = x.x; vv[1] = x.y; vv[2] = x.z; vv[3] = x.w; vv[4] = y.x; vv[5] = y.y; vv[6] = y.z; vv[7] = y.w;
                    }
                } else if (act) {
#pragma unroll
                    for (int k = 0; k < QT_NC; ++k) vv[k] = a.cpg_pos[o0 + min((uint32_t)k, n_calls - 1)];
                }
                // candidate ranges rely on every call lying in [start - 1, start - 1 + max_span] (a reverse read's first call
                // may sit one base before its start, readutil.rs:338; rule of the PDR tile kernel).  Unsigned: a call further
                // left is caught too; the windows' deltas are checked to be 1..2047 below, so the calls in between are ordered.
                const uint32_t sm1 = (uint32_t)a.read_start[i] - 1u;
                uint32_t xmax = 0;
#pragma unroll
                for (int k = 0; k < QT_NC; ++k) xmax = max(xmax, (uint32_t)k < n_calls ? (vv[k] & 0x7fffffffu) - sm1 : 0u);
                bad |= (xmax > (uint32_t)a.max_span) ? 1u : 0u;
#pragma unroll
                for (int k = 3; k < QT_NC; ++k) {
                    if (!__any((uint32_t)k < n_calls)) break;               // wave-uniform
                    if ((uint32_t)k < n_calls) window(vv[k - 3], vv[k - 2], vv[k - 1], vv[k]);
                }
                if (__any(n_calls > (uint32_t)QT_NC) && n_calls > (uint32_t)QT_NC) {
                    uint32_t x = vv[QT_NC - 3], y = vv[QT_NC - 2], z = vv[QT_NC - 1];
                    for (uint32_t k = o0 + QT_NC; k < o1; ++k) {
                        const uint32_t w = a.cpg_pos[k];
                        bad |= ((w & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
                        window(x, y, z, w);
                        x = y; y = z; z = w;
                    }
                }
            }
        }
        if (bad) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
    }
    __syncthreads();
    if (s_heavy) {                                      // block-uniform: the whole tile goes to the global path
        if (tid == 0) { a.tile_flag[t] = 1u; a.tile_rows[t] = 0u; a.tile_row0[t] = 0ull; atomicAdd(a.n_heavy, 1ull); }
        continue;
    }
    // Rows go out sorted by key = (p1, d2, d3, d4) = (p1, p2, p3, p4).  Bucket sort on p1: QT_B buckets of QT_W / QT_B
    // positions, a few keys each.  Every thread holds its slots in registers, so the table is rebuilt in place in bucket
    // order; a key's final rank = start of its bucket + the keys of that bucket below it.  (Before: an all-pairs rank
    // sort / bitonic network over the tile's keys -- most of the kernel's time on sparse WGBS.)
    static_assert(QT_S % QT_B == 0 && QT_B == 256, "each thread owns QT_S / QT_B slots and one bucket");
    constexpr int PER = QT_S / QT_B;
    constexpr int BSHIFT = __builtin_ctz((unsigned)QT_W) - 8;
    unsigned long long kk[PER];
    uint32_t pib[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        kk[k] = keys[tid * PER + k];
        pib[k] = 0;
        if (kk[k] != QKEY_EMPTY) pib[k] = atomicAdd(&bcnt[((uint32_t)(kk[k] >> 33) - (uint32_t)T0) >> BSHIFT], 1u);
    }
    __syncthreads();                                    // every slot is in registers now: the table can be overwritten
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t m = bcnt[tid];
    uint32_t incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) ws[wave + 1] = incl;
    __syncthreads();
    if (tid == 0) {
        ws[0] = 0;
        for (int w = 1; w <= QT_B / 64; ++w) ws[w] += ws[w - 1];
        const uint32_t n_all = ws[QT_B / 64];
        a.tile_flag[t] = 0u; a.tile_rows[t] = n_all;
        if (n_all && s_chunk_pos + n_all > s_chunk_end) {       // next chunk (what is left of the old one stays unused)
            // enough for this workgroup's remaining tiles if they are like this one, at most QT_CHUNK rows: the gaps stay
            // small next to the rows (a single-tile workgroup claims exactly its rows)
            const unsigned long long left = (unsigned long long)((a.ntiles - 1u - t) / gridDim.x + 1u);
            const unsigned long long claim = max((unsigned long long)n_all, min((unsigned long long)n_all * left, (unsigned long long)QT_CHUNK));
            s_chunk_pos = atomicAdd(a.row_total, claim);
            s_chunk_end = s_chunk_pos + claim;
            if (s_chunk_end > a.row_cap) atomicAdd(a.unfit, 1ull);
        }
        s_row0 = s_chunk_pos;
        s_chunk_pos += n_all;
        a.tile_row0[t] = s_row0;
    }
    __syncthreads();
    const uint32_t n = ws[QT_B / 64];
    if (n == 0 || s_row0 + n > a.row_cap) continue;      // block-uniform
    bbase[tid] = ws[wave] + incl - m;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (kk[k] == QKEY_EMPTY) continue;
        const uint32_t dst = bbase[((uint32_t)(kk[k] >> 33) - (uint32_t)T0) >> BSHIFT] + pib[k];
        keys[dst] = kk[k];
        sslot[dst] = (uint16_t)(tid * PER + k);
    }
    __syncthreads();
    for (uint32_t j = tid; j < n; j += QT_B) {
        const unsigned long long key = keys[j];
        const uint32_t bk = ((uint32_t)(key >> 33) - (uint32_t)T0) >> BSHIFT, b0 = bbase[bk], b1 = b0 + bcnt[bk];
        uint32_t r = b0;
        for (uint32_t i = b0; i < b1; ++i) r += keys[i] < key ? 1u : 0u;
        const uint32_t h = sslot[j];
        const unsigned long long o = s_row0 + r;
        const int32_t p1 = (int32_t)(key >> 33);
        const int32_t p2 = p1 + (int32_t)((key >> 22) & 2047u), p3 = p2 + (int32_t)((key >> 11) & 2047u),
                      p4 = p3 + (int32_t)(key & 2047u);
        reinterpret_cast<int4 *>(a.out_pos)[o] = make_int4(p1, p2, p3, p4);
        uint32_t c[16];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const uint32_t v = bins[h * 8 + w];
            c[2 * w] = v & 0xffffu; c[2 * w + 1] = v >> 16;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            reinterpret_cast<uint4 *>(a.out_cnt + o * 16)[q] = make_uint4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
        float me, pm;
        uint32_t total;
        quartet_values(c, me, pm, total);
        a.out_me[o] = me; a.out_pm[o] = pm; a.out_depth[o] = total;
    }
    }
}

// (re)start of a batch: back to the row count before it
__global__ void k_quartet_rewind(unsigned long long *qs, unsigned long long rows_before) { qs[1] = rows_before; qs[5] = 0; qs[6] = 0; }

// a queued batch's state words as its tile kernel left them ([1] rows so far, [5] tiles for the global path and [6] tiles that did not
// fit, both since the run of queued batches began): read back by quartet_resolve, not by the call that queued the batch
__global__ void k_quartet_snap(const unsigned long long *__restrict__ qs, unsigned long long *__restrict__ snap) {
    if (threadIdx.x < Q_STATE_WORDS) snap[threadIdx.x] = qs[threadIdx.x];
}

}  // namespace mth

using namespace mth;

// One batch.  queued = false: the call ends knowing the batch's rows (one host sync; redone with the exact size if the rows did not fit,
// the global path for tiles the LDS table could not hold).  queued = true (device-resident batches after the first of a job): tile
// kernel and a snapshot of the state words only -- whether everything fitted is looked at by quartet_resolve, at the next call that
// needs the rows, and anything else than "all fitted, no tile for the global path" replays the batches from the first such one on
// through the synchronous form (their arrays are still there: include/metheor_hip.h, device-resident batches stay untouched until
// the next synchronising call).
static int quartet_batch(mth_ctx *ctx, const mth_batch_t &d, const mth_quartet_params_t *params, int32_t batch_tid, bool queued) {
    int rc = MTH_OK;
    hipStream_t s = ctx->stream;
    if (!ctx->q_state.p) {
        MTH_HIP(ctx, ctx->q_state.reserve(Q_STATE_WORDS * sizeof(unsigned long long), s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->q_state.p, 0, Q_STATE_WORDS * sizeof(unsigned long long), s));
    }
    // [0] updates (counting pass of the global path) [1] total rows [2] first row of the global path's rows
    // [3] its overflow flag [5] tiles left to the global path [6] tiles whose rows did not fit the output
    unsigned long long *qs = ctx->q_state.as<unsigned long long>();
    const int64_t region_len = (int64_t)d.region_end - d.region_beg;
    // Tile width: wide tiles pay the per-tile costs (index look-up, LDS clear, barriers, halo reads) less often, but the
    // table holds QT_S quartets and a 16-bit bin 65535 candidate reads.  About one quartet starts per CpG site; sites per
    // bp ~ calls per read / read length (max_span stands in for the length).  Tiles that overflow anyway take the global path.
    int tile_shift = 13;
    if (d.n_reads && region_len > 0) {
        const double sites_per_bp = (double)d.n_cpgs / (double)d.n_reads / (double)std::max(d.max_span, 1);
        const double reads_per_bp = (double)d.n_reads / (double)region_len;
        // a site starts a quartet only if three more sites follow within a read's span: Poisson tail P(>= 3 | sites_per_bp x span).
        // (Without it config 2 stayed at 8192 bp and config-3 density at 16384: 0.119 against 0.100 ms at 16384 and 0.159 against
        // 0.112 ms at 32768 -- a tile's fixed chain of round trips and barriers is what these kernels wait for.)
        const double lam = sites_per_bp * (double)std::max(d.max_span, 1);
        const double p3 = std::max(0.05, 1.0 - std::exp(-lam) * (1.0 + lam + 0.5 * lam * lam));
        while (tile_shift < 16 && sites_per_bp * p3 * (double)(2 << tile_shift) <= 0.4 * QT_S &&
               reads_per_bp * (double)((2 << tile_shift) + d.max_span + 2 * IDX_Q) <= 30000.0)
            ++tile_shift;
    }
    if (const char *e = getenv("MTH_QUARTET_TILE_SHIFT")) tile_shift = std::min(16, std::max(13, atoi(e)));   // tests / tuning
    const int QT_W = 1 << tile_shift;
    const uint32_t ntiles = (d.n_reads && region_len > 0) ? (uint32_t)((region_len + QT_W - 1) / QT_W) : 0u;
    const uint64_t tiles_before = ctx->q_meta.empty() ? 0 : ctx->q_meta.back().tile_end;
    // queued: the exact count is on the device only; q_rows_est bounds it from above (every queued batch so far within its estimate --
    // if one was not, the resolve replays from there and none of this batch's rows survive anyway)
    const uint64_t rows_before = queued && !ctx->q_pending.empty() ? ctx->q_rows_est : ctx->q_rows;
    mth_ctx::TileBatch meta{batch_tid, 0, rows_before, tiles_before + ntiles};
    if (!ntiles && queued && !ctx->q_pending.empty()) queued = false, rc = quartet_resolve(ctx);       // (rare: an empty batch inside a run)
    if (rc) return rc;
    if (!ntiles) { meta.heavy0 = ctx->q_rows; ctx->q_meta.push_back(meta); return MTH_OK; }
    auto grow_rows = [&](uint64_t cap, uint64_t used) -> hipError_t {      // keeps the rows of earlier batches
        hipError_t e;
        if ((e = ctx->q_pos.reserve(cap * 16, s, true, used * 16)) != hipSuccess) return e;
        if ((e = ctx->q_cnt.reserve(cap * 64, s, true, used * 64)) != hipSuccess) return e;
        if ((e = ctx->q_me.reserve(cap * 4, s, true, used * 4)) != hipSuccess) return e;
        if ((e = ctx->q_pm.reserve(cap * 4, s, true, used * 4)) != hipSuccess) return e;
        if ((e = ctx->q_depth.reserve(cap * 4, s, true, used * 4)) != hipSuccess) return e;
        ctx->q_cap = cap;
        return hipSuccess;
    };
    int32_t idx_base = 0;
    uint32_t nt = 0;
    rc = build_read_index(ctx, d, QT_W, idx_base, nt);
    if (rc) return rc;
    MTH_HIP(ctx, ctx->q_tflag.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->q_tile_row0.reserve((tiles_before + ntiles) * 8, s, true, tiles_before * 8));
    MTH_HIP(ctx, ctx->q_tile_rows.reserve((tiles_before + ntiles) * 4, s, true, tiles_before * 4));
    // output size: rows per CpG call of the batches so far (first batch: a guess); the kernel reports the exact need --
    // there is no counting pre-pass
    uint64_t want = rows_before + (uint64_t)((double)d.n_cpgs * ctx->q_rows_per_cpg * 1.25) + 4096;
    if (const char *e = getenv("MTH_QUARTET_ROWS_MIN")) want = rows_before + strtoull(e, nullptr, 10);   // tests: force the redo
    unsigned long long *st = ctx->h_words;     // pinned: the read-back does not go through a staging copy
    for (int attempt = 0;; ++attempt) {
        if (want > ctx->q_cap) MTH_HIP(ctx, grow_rows(want + (queued ? want / 4 : 0), std::min<uint64_t>(rows_before, ctx->q_cap)));
        // a run of queued batches continues from the device's own row count; its first batch (and every synchronous one) starts from
        // the host's, which is exact then
        if (!queued || ctx->q_pending.empty()) hipLaunchKernelGGL(k_quartet_rewind, dim3(1), dim3(1), 0, s, qs, (unsigned long long)rows_before);
        QTileArgs a;
        a.read_start = d.read_start; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
        a.idx = idx_ptr(ctx);
        a.region_beg = d.region_beg; a.region_end = d.region_end; a.idx_base = idx_base; a.max_span = d.max_span;
        a.n_reads = d.n_reads; a.ntiles = ntiles; a.n_cpgs = (uint32_t)d.n_cpgs; a.min_qual = params->min_qual;
        a.force_heavy = getenv("MTH_QUARTET_FORCE_GLOBAL") ? 1 : 0;
        // a queued batch must stay within its ESTIMATE, not just within the buffer: the next queued batch takes the estimate as the
        // rows in use and keeps only that many when it grows the buffer (rows between the estimate and the device's count
        // were lost without a flag); beyond the estimate the batch is unfit and quartet_resolve replays it
        a.row_total = qs + 1; a.row_cap = queued ? std::min<uint64_t>(ctx->q_cap, want) : ctx->q_cap; a.unfit = qs + 6; a.n_heavy = qs + 5;
        a.tile_flag = ctx->q_tflag.as<uint32_t>();
        a.tile_row0 = ctx->q_tile_row0.as<unsigned long long>() + tiles_before;
        a.tile_rows = ctx->q_tile_rows.as<uint32_t>() + tiles_before;
        a.out_pos = ctx->q_pos.as<int32_t>(); a.out_cnt = ctx->q_cnt.as<uint32_t>(); a.out_me = ctx->q_me.as<float>();
        a.out_pm = ctx->q_pm.as<float>(); a.out_depth = ctx->q_depth.as<uint32_t>(); a.st = ctx->d_state;
        {
            LaunchTimer lt(ctx, K_QTILE);
            if (tile_shift == 13) hipLaunchKernelGGL(k_quartet_tile<8192>, dim3(std::min<uint32_t>(ntiles, QT_GRID)), dim3(QT_B), 0, s, a);
            else if (tile_shift == 14) hipLaunchKernelGGL(k_quartet_tile<16384>, dim3(std::min<uint32_t>(ntiles, QT_GRID)), dim3(QT_B), 0, s, a);
            else if (tile_shift == 15) hipLaunchKernelGGL(k_quartet_tile<32768>, dim3(std::min<uint32_t>(ntiles, QT_GRID)), dim3(QT_B), 0, s, a);
            else hipLaunchKernelGGL(k_quartet_tile<65536>, dim3(std::min<uint32_t>(ntiles, QT_GRID)), dim3(QT_B), 0, s, a);
        }
        if (queued) {
            const size_t k = ctx->q_pending.size();
            MTH_HIP(ctx, ctx->q_snap.reserve((size_t)Q_QUEUE_MAX * Q_STATE_WORDS * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k_quartet_snap, dim3(1), dim3(64), 0, s, (const unsigned long long *)qs, ctx->q_snap.as<unsigned long long>() + k * Q_STATE_WORDS);
            MTH_HIP(ctx, hipGetLastError());
            ctx->q_pending.push_back(mth_ctx::QueuedBatch{d, *params, batch_tid, d.n_cpgs});
            ctx->q_rows_est = want;
            ctx->q_meta.push_back(meta);                   // rows / heavy0: quartet_resolve
            return MTH_OK;
        }
        MTH_HIP(ctx, hipMemcpyAsync(st, qs, Q_STATE_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));            // one sync per batch: rows, flagged tiles, fit
        if (!st[6]) break;
        if (attempt) return fail(ctx, MTH_ERR_STATE, "quartets: rows did not fit an exactly sized output");
        want = st[1];                                     // every tile claimed its range: this is the exact size
    }
    uint64_t total = st[1];
    meta.heavy0 = total;
    if (st[5]) {
        // The tiles the LDS table could not hold: the global table, for their quartets only.  Its size: `bound` counts
        // quartet INSTANCES of the whole batch; bound / 2 slots to start with, redone 4x larger if an insert ran out of
        // probes -- at 2 x bound slots that cannot happen.
        MTH_HIP(ctx, hipMemsetAsync(qs, 0, sizeof(unsigned long long), s));
        {
            LaunchTimer lt(ctx, K_QBOUND);
            hipLaunchKernelGGL(k_quartet_bound, dim3(1024), dim3(256), 0, s, d.cpg_off, d.read_mapq, d.n_reads,
                               params->min_qual, qs);
        }
        unsigned long long bound = 0;
        MTH_HIP(ctx, hipMemcpyAsync(&bound, qs, sizeof bound, hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));
        unsigned long long n_slots = 1024;
        while (n_slots < bound / 2) n_slots <<= 1;
        unsigned long long wide_cap = std::max<unsigned long long>(ctx->q_wpat.cap / 4, 65536), wide_n = 0;
        if (const char *e = getenv("MTH_QUARTET_SLOTS_MIN")) { const unsigned long long k = strtoull(e, nullptr, 10); if (k >= 16) { n_slots = 16; while (n_slots < k) n_slots <<= 1; } }   // tests: force the retry
        for (;;) {
            MTH_HIP(ctx, ctx->q_keys.reserve(n_slots * 8, s));
            MTH_HIP(ctx, ctx->q_hist.reserve(n_slots * 64, s));
            MTH_HIP(ctx, hipMemsetAsync(ctx->q_keys.p, 0xFF, n_slots * 8, s));
            MTH_HIP(ctx, hipMemsetAsync(ctx->q_hist.p, 0, n_slots * 64, s));
            MTH_HIP(ctx, hipMemsetAsync(qs + 3, 0, sizeof(unsigned long long), s));
            MTH_HIP(ctx, hipMemsetAsync(qs + 7, 0, sizeof(unsigned long long), s));
            MTH_HIP(ctx, ctx->q_wpos.reserve(wide_cap * 16, s));
            MTH_HIP(ctx, ctx->q_wpat.reserve(wide_cap * 4, s));
            {
                LaunchTimer lt(ctx, K_QINSERT);
                hipLaunchKernelGGL(k_quartet_insert, dim3((d.n_reads + 255) / 256), dim3(256), 0, s, d.cpg_off, d.cpg_pos,
                                   d.read_mapq, d.n_reads, params->min_qual, d.region_beg, d.region_end,
                                   ctx->q_keys.as<unsigned long long>(), ctx->q_hist.as<uint32_t>(), n_slots - 1, qs + 3,
                                   ctx->d_state, (const uint32_t *)ctx->q_tflag.as<uint32_t>(), tile_shift,
                                   ctx->q_wpos.as<uint4>(), ctx->q_wpat.as<uint32_t>(), wide_cap, qs + 7);
            }
            unsigned long long ovf = 0;
            MTH_HIP(ctx, hipMemcpyAsync(&ovf, qs + 3, sizeof ovf, hipMemcpyDeviceToHost, s));
            MTH_HIP(ctx, hipMemcpyAsync(&wide_n, qs + 7, sizeof wide_n, hipMemcpyDeviceToHost, s));
            MTH_HIP(ctx, hipStreamSynchronize(s));
            if (wide_n > wide_cap) { wide_cap = wide_n; continue; }        // the wide list was too short: same table, again
            if (!ovf) break;
            n_slots <<= 2;
        }
        if (total + bound > ctx->q_cap) MTH_HIP(ctx, grow_rows(total + bound, total));     // distinct quartets <= instances
        const uint32_t nblk = (uint32_t)((n_slots + 256 * QE_PER - 1) / (256 * QE_PER));
        MTH_HIP(ctx, ctx->q_blk.reserve((size_t)nblk * 4, s));
        MTH_HIP(ctx, ctx->q_batch_rows.reserve(4, s));
        {
            LaunchTimer lt(ctx, K_QEMIT);
            hipLaunchKernelGGL(k_quartet_blockcount, dim3(nblk), dim3(256), 0, s, ctx->q_keys.as<unsigned long long>(),
                               n_slots, ctx->q_blk.as<uint32_t>());
            hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->q_blk.as<uint32_t>(), nblk, qs + 1, qs + 2,
                               ctx->q_batch_rows.as<uint32_t>(), 0u);
            hipLaunchKernelGGL(k_quartet_emit, dim3(nblk), dim3(256), 0, s, ctx->q_keys.as<unsigned long long>(),
                               (const unsigned long long *)nullptr,
                               ctx->q_hist.as<uint32_t>(), n_slots, ctx->q_blk.as<uint32_t>(), qs + 2,
                               ctx->q_pos.as<int32_t>(), ctx->q_cnt.as<uint32_t>(), ctx->q_me.as<float>(),
                               ctx->q_pm.as<float>(), ctx->q_depth.as<uint32_t>());
        }
        unsigned long long t2 = 0;
        MTH_HIP(ctx, hipMemcpyAsync(&t2, qs + 1, sizeof t2, hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));
        total = t2;
        if (wide_n) {
            // the wide instances under their 128-bit keys: same emit, rows appended after the packed-key rows of the batch
            unsigned long long w_slots = 1024;
            while (w_slots < 2 * wide_n) w_slots <<= 1;
            for (;;) {
                MTH_HIP(ctx, ctx->q_wk0.reserve(w_slots * 8, s));
                MTH_HIP(ctx, ctx->q_wk1.reserve(w_slots * 8, s));
                MTH_HIP(ctx, ctx->q_hist.reserve(w_slots * 64, s));
                MTH_HIP(ctx, hipMemsetAsync(ctx->q_wk0.p, 0xFF, w_slots * 8, s));
                MTH_HIP(ctx, hipMemsetAsync(ctx->q_wk1.p, 0xFF, w_slots * 8, s));
                MTH_HIP(ctx, hipMemsetAsync(ctx->q_hist.p, 0, w_slots * 64, s));
                MTH_HIP(ctx, hipMemsetAsync(qs + 3, 0, sizeof(unsigned long long), s));
                hipLaunchKernelGGL(k_quartet_wide_insert, dim3((uint32_t)((wide_n + 255) / 256)), dim3(256), 0, s, ctx->q_wpos.as<uint4>(),
                                   ctx->q_wpat.as<uint32_t>(), wide_n, ctx->q_wk0.as<unsigned long long>(), ctx->q_wk1.as<unsigned long long>(),
                                   ctx->q_hist.as<uint32_t>(), w_slots - 1, qs + 3);
                unsigned long long ovf = 0;
                MTH_HIP(ctx, hipMemcpyAsync(&ovf, qs + 3, sizeof ovf, hipMemcpyDeviceToHost, s));
                MTH_HIP(ctx, hipStreamSynchronize(s));
                if (!ovf) break;
                w_slots <<= 2;
            }
            if (total + wide_n > ctx->q_cap) MTH_HIP(ctx, grow_rows(total + wide_n, total));
            const uint32_t wblk = (uint32_t)((w_slots + 256 * QE_PER - 1) / (256 * QE_PER));
            MTH_HIP(ctx, ctx->q_blk.reserve((size_t)wblk * 4, s));
            hipLaunchKernelGGL(k_quartet_blockcount, dim3(wblk), dim3(256), 0, s, ctx->q_wk0.as<unsigned long long>(), w_slots, ctx->q_blk.as<uint32_t>());
            hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->q_blk.as<uint32_t>(), wblk, qs + 1, qs + 2,
                               ctx->q_batch_rows.as<uint32_t>(), 0u);
            hipLaunchKernelGGL(k_quartet_emit, dim3(wblk), dim3(256), 0, s, ctx->q_wk0.as<unsigned long long>(),
                               (const unsigned long long *)ctx->q_wk1.as<unsigned long long>(),
                               ctx->q_hist.as<uint32_t>(), w_slots, ctx->q_blk.as<uint32_t>(), qs + 2,
                               ctx->q_pos.as<int32_t>(), ctx->q_cnt.as<uint32_t>(), ctx->q_me.as<float>(),
                               ctx->q_pm.as<float>(), ctx->q_depth.as<uint32_t>());
            MTH_HIP(ctx, hipMemcpyAsync(&t2, qs + 1, sizeof t2, hipMemcpyDeviceToHost, s));
            MTH_HIP(ctx, hipStreamSynchronize(s));
            total = t2;
        }
    }
    MTH_HIP(ctx, hipGetLastError());
    meta.rows = total - rows_before;
    ctx->q_rows = total;
    if (d.n_cpgs) { ctx->q_rows_per_cpg = std::max(ctx->q_rows_per_cpg * 0.5, (double)meta.rows / (double)d.n_cpgs); ctx->q_learned = true; }
    ctx->q_meta.push_back(meta);
    return MTH_OK;
}

namespace mth {

// The queued batches' rows: one read-back of their snapshots.  All fitted and no tile was left to the global path: the metas get
// their row counts.  Otherwise the batches from the first one that says so are replayed synchronously, in order.
int quartet_resolve(mth_ctx *ctx) {
    if (ctx->q_pending.empty()) return MTH_OK;
    // (a replay below rebuilds its batch's read index in the context's own buffer: whatever prepared batch the latest entry point
    // worked on is not this one's)
    ctx->cur_prep = nullptr; ctx->cur_idx = nullptr;
    MTH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<mth_ctx::QueuedBatch> pend;
    pend.swap(ctx->q_pending);
    const size_t n = pend.size(), base = ctx->q_meta.size() - n;
    std::vector<unsigned long long> snap(n * Q_STATE_WORDS);
    MTH_HIP(ctx, hipMemcpyAsync(snap.data(), ctx->q_snap.p, snap.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t good = 0;
    uint64_t rows = 0, cpgs = 0;
    for (; good < n; ++good) {
        const unsigned long long *w = snap.data() + good * Q_STATE_WORDS;
        if (w[5] || w[6]) break;
        mth_ctx::TileBatch &m = ctx->q_meta[base + good];
        m.rows = w[1] - ctx->q_rows;
        m.heavy0 = w[1];
        rows += m.rows; cpgs += pend[good].n_cpgs;
        ctx->q_rows = w[1];
    }
    if (cpgs) ctx->q_rows_per_cpg = std::max(ctx->q_rows_per_cpg * 0.5, (double)rows / (double)cpgs);
    if (getenv("MTH_QUARTET_DEBUG")) fprintf(stderr, "[quartet] queued batches %zu, replayed %zu\n", n, n - good);     // tests
    if (good == n) return MTH_OK;
    ctx->q_meta.resize(base + good);
    for (size_t k = good; k < n; ++k) {
        const int rc = quartet_batch(ctx, pend[k].d, &pend[k].params, pend[k].tid, false);
        if (rc) return rc;
    }
    return MTH_OK;
}

}  // namespace mth

extern "C" {

int mth_quartet_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_quartet_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    ctx->q_epoch += 1;
    // Queued (no host sync in the call): a device-resident batch once a batch of this context has taught the output sizing, unless the
    // launches are being timed.  MTH_QUARTET_QUEUE=0 switches it off (A/B).  Every other entry point settles the queue (mth::enter).
    static const bool queue_off = getenv("MTH_QUARTET_QUEUE") && atoi(getenv("MTH_QUARTET_QUEUE")) == 0;
    const bool queued = (batch->mem == MTH_MEM_DEVICE || batch->mem == MTH_MEM_PREPARED) && ctx->q_learned && !ctx->timing && !queue_off && ctx->q_pending.size() < (size_t)Q_QUEUE_MAX;
    mth_batch_t d;
    ctx->tile_queue_hold = queued;
    int rc = stage_batch(ctx, *batch, d);
    ctx->tile_queue_hold = false;
    if (rc) return rc;
    if (!queued && (rc = quartet_resolve(ctx))) return rc;
    return quartet_batch(ctx, d, params, batch->tid, queued);
}

// rows of all batches with depth >= min_depth (me.rs:82 / pm.rs:77); any pointer may be NULL.
// Call with every pointer NULL to get the count.
int mth_quartet_fetch(mth_ctx_t *ctx, uint32_t min_depth, uint64_t *n_rows, int32_t *tid, int32_t *pos4,
                      uint32_t *counts16, float *me, float *pm) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);          // resolves the queued batches first
    if (rc) return rc;
    // Rows live in [0, q_rows) with gaps (the unused tails of the tile kernel's chunks).  Row order: per batch, sorted by
    // (p1..p4): the tiles in position order give that for free (each tile's rows are sorted in LDS); a batch with rows from
    // the global path is sorted here.  The reference's order is HashMap-random.
    const uint64_t total = ctx->q_rows;
    std::vector<int32_t> hp;                // pos1..pos4 of every device row
    std::vector<uint64_t> &order = ctx->q_order;            // device row of every output row that passes the depth filter
    std::vector<int32_t> &order_tid = ctx->q_order_tid;
    const bool cached = ctx->q_order_epoch == ctx->q_epoch && ctx->q_order_min_depth == min_depth;
    if (!cached) {
        std::vector<uint32_t> depth(total);
        if (total) MTH_HIP(ctx, hipMemcpy(depth.data(), ctx->q_depth.p, total * 4, hipMemcpyDeviceToHost));
        const uint64_t n_tiles = ctx->q_meta.empty() ? 0 : ctx->q_meta.back().tile_end;
        std::vector<unsigned long long> trow0(n_tiles);
        std::vector<uint32_t> trows(n_tiles);
        if (n_tiles) {
            MTH_HIP(ctx, hipMemcpy(trow0.data(), ctx->q_tile_row0.p, n_tiles * 8, hipMemcpyDeviceToHost));
            MTH_HIP(ctx, hipMemcpy(trows.data(), ctx->q_tile_rows.p, n_tiles * 4, hipMemcpyDeviceToHost));
        }
        order.clear(); order_tid.clear();
        bool any_heavy = false;
        {
            uint64_t batch_end = 0;
            for (const auto &mb : ctx->q_meta) { batch_end += mb.rows; any_heavy |= mb.heavy0 < batch_end; }
        }
        if (total && (pos4 || any_heavy)) {
            hp.resize(total * 4);
            MTH_HIP(ctx, hipMemcpy(hp.data(), ctx->q_pos.p, total * 16, hipMemcpyDeviceToHost));
        }
        {
            uint64_t batch_end = 0;
            for (size_t b = 0; b < ctx->q_meta.size(); ++b) {
                const auto &mb = ctx->q_meta[b];
                batch_end += mb.rows;
                const size_t first = order.size();
                auto put = [&](uint64_t i) { if (depth[i] >= min_depth) { order.push_back(i); order_tid.push_back(mb.tid); } };
                for (uint64_t t = b ? ctx->q_meta[b - 1].tile_end : 0; t < mb.tile_end; ++t)
                    for (uint32_t j = 0; j < trows[t]; ++j) put(trow0[t] + j);
                const size_t sorted_end = order.size();
                for (uint64_t i = mb.heavy0; i < batch_end; ++i) put(i);
                if (order.size() > sorted_end)      // rows of the global path came in table order: put the batch in (pos1..pos4) order
                    std::sort(order.begin() + (ptrdiff_t)first, order.end(), [&](uint64_t x, uint64_t y) {
                        return std::lexicographical_compare(hp.begin() + (ptrdiff_t)(x * 4), hp.begin() + (ptrdiff_t)(x * 4 + 4),
                                                            hp.begin() + (ptrdiff_t)(y * 4), hp.begin() + (ptrdiff_t)(y * 4 + 4));
                    });
            }
        }
        ctx->q_order_epoch = ctx->q_epoch; ctx->q_order_min_depth = min_depth;
    }
    const uint64_t n = order.size();
    if (n_rows) *n_rows = n;
    if (!tid && !pos4 && !counts16 && !me && !pm) return MTH_OK;
    std::vector<uint32_t> hc(counts16 ? total * 16 : 0);
    std::vector<float> hme(me ? total : 0), hpm(pm ? total : 0);
    // rows of contig groups go back under their own contig: tid and positions are both needed for that, whichever was asked for
    const bool grouped = (tid || pos4) && has_group_batch(ctx, 3);
    std::vector<int32_t> tmp_tid, tmp_pos;
    if (grouped && !tid) { tmp_tid.resize(n); tid = tmp_tid.data(); }
    if (grouped && !pos4) { tmp_pos.resize(n * 4); pos4 = tmp_pos.data(); }
    if (total) {
        if (pos4 && hp.empty()) { hp.resize(total * 4); MTH_HIP(ctx, hipMemcpy(hp.data(), ctx->q_pos.p, total * 16, hipMemcpyDeviceToHost)); }
        if (counts16) MTH_HIP(ctx, hipMemcpy(hc.data(), ctx->q_cnt.p, total * 64, hipMemcpyDeviceToHost));
        if (me) MTH_HIP(ctx, hipMemcpy(hme.data(), ctx->q_me.p, total * 4, hipMemcpyDeviceToHost));
        if (pm) MTH_HIP(ctx, hipMemcpy(hpm.data(), ctx->q_pm.p, total * 4, hipMemcpyDeviceToHost));
    }
    for (uint64_t o = 0; o < n; ++o) {
        const uint64_t i = order[o];
        if (tid) tid[o] = order_tid[o];
        if (pos4) memcpy(pos4 + o * 4, hp.data() + i * 4, 16);
        if (counts16) memcpy(counts16 + o * 16, hc.data() + i * 16, 64);
        if (me) me[o] = hme[i];
        if (pm) pm[o] = hpm[i];
    }
    if (grouped) return ungroup_rows(ctx, n, tid, pos4, 4, 4, nullptr);
    return MTH_OK;
}

}  // extern "C"
