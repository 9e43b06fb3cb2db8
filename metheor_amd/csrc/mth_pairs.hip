// mth_pairs.hip -- LPMD per-pair table (`metheor lpmd --pairs`): lpmd.rs:70-87 (add_pair_concordance),
// 89-122 (print_pair_statistics), pairs produced by readutil.rs:166-224.
//
// For every read with mapq >= min_qual and every pair (i<j) of its CpGs with
// min_distance <= relpos_j - relpos_i <= max_distance: key (abspos_i, abspos_j), +1 concordant or
// discordant.  Rows sorted by key; per-pair lpmd = n_d as f32 / (n_c as f32 + n_d as f32) (lpmd.rs:111).
//
// Device: a pair is owned by the position of its first CpG, so the tile (8192 / 16384 / 32768 bp, by the batch's call
// density) holding pos1 sees every update of its
// pairs (the ownership argument of the PDR tile kernel and of mth_quartet.hip).  k_pairs_tile: one workgroup per tile,
// candidate reads from the linear read index, a 2048-slot table in LDS -- one 64-bit word per slot: key = (pos1 - tile
// start) << (32 - log2 W) | (pos2 - pos1) in the high half, the two 16-bit counters in the low half, so sorting the words sorts the
// pairs -- bucket-sorted in place in LDS, rows written straight to the output (one global atomic per tile claims the
// range).  The fetch walks the tiles in order: rows come out sorted as lpmd.rs:94 wants them, with no sort.
// The row buffer is sized from the rows-per-CpG of earlier batches; a batch that does not fit is redone once with the
// exact size (the kernel reports it) -- there is no counting pre-pass.
// Tiles the LDS table cannot hold (> 65535 candidate reads: 16-bit counters; > 2048 distinct pairs; pos2 - pos1 >= 2^(32 - log2 W))
// are flagged and take the first version's path for their pairs only: a global open-addressing table (key = pos1 << 32 |
// pos2, two u32 counters) sized from a counting pre-pass; their rows follow the batch's sorted rows and the fetch then
// sorts that contig's rows on the host.  A pair is owned by the batch whose region contains pos1 (halo reads
// included), so region / contig sharding needs no merge.
#include <algorithm>
#include <numeric>

#include "mth_ctx.h"
#include "mth_scan.h"
#include "mth_tile_dev.h"

namespace mth {

constexpr unsigned long long PKEY_EMPTY = ~0ull;

__device__ __forceinline__ unsigned long long phash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

struct PairArgs {
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const void     *cpg_rel;
    unsigned long long *keys;
    uint32_t *cnt;                 // 2 per slot: concordant, discordant
    unsigned long long *n_updates; // counting pass
    unsigned long long mask;
    int32_t region_beg, region_end, min_dist, max_dist;
    uint32_t n_reads;
    uint8_t min_qual;
    unsigned long long *overflow;   // set when an insert ran out of probes
    const uint32_t *tile_flag;      // nullptr: every pair; else only pairs whose pos1 lies in a flagged tile
    int tile_shift;                 // log2 of the tile width the flags were made with
};

// PT_U / PT_OCC: 2 reads per thread and round at 6 waves per SIMD (80 VGPRs; 26 KiB of LDS allow 6 workgroups per CU) measured best
// (tools/ab_measure.sh pairs: 1 / 6 +5 %, 2 / 5 +2 %, 3 / 5 +17 %, 4 / 4 +21 %; profiles/r03_pairs_tile.md)
constexpr int PT_OCC = 6;
constexpr int PT_S = 2048, PT_B = 256, PT_U = 2, PT_CHUNK = 1024, PT_GRID = 8192;   // tile kernel: LDS slots, threads, reads per thread and round (the tile
                                                   // width is a template parameter: 8192 / 16384 / 32768, chosen per batch)
constexpr int P_STATE_WORDS = 8;
constexpr int P_QUEUE_MAX = 4096;       // queued batches between two resolves
constexpr int PT_NC = 8;                                 // calls of a read parked in LDS for the pair loops

// COUNT: only count the updates (table sizing); otherwise insert them
template <typename RelT, bool COUNT>
__global__ __launch_bounds__(256) void k_pairs(const PairArgs a) {
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    unsigned long long mine = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.n_reads; i += gridDim.x * 256) {
        if (a.read_mapq[i] < a.min_qual) continue;                         // lpmd.rs:177
        const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
        for (uint32_t k = o0 + 1; k < o1; ++k) {
            const int32_t rk = (int32_t)rel[k];
            const uint32_t wk = a.cpg_pos[k];
            for (uint32_t j = k; j-- > o0;) {
                const int32_t dist = rk - (int32_t)rel[j];
                if (dist > a.max_dist) break;                              // readutil.rs:184
                if (dist < a.min_dist) continue;                           // readutil.rs:196
                const uint32_t wj = a.cpg_pos[j];
                const int32_t p1 = (int32_t)(wj & 0x7fffffffu);
                if (p1 < a.region_beg || p1 >= a.region_end) continue;     // owned by the region of pos1
                if (a.tile_flag && !a.tile_flag[(uint32_t)(p1 - a.region_beg) >> a.tile_shift]) continue;
                if (COUNT) { mine += 1; continue; }
                const unsigned long long key = ((unsigned long long)(uint32_t)p1 << 32) | (wk & 0x7fffffffu);
                unsigned long long h = phash(key) & a.mask;
                // table sized for the distinct pairs one expects, not for the updates: bounded probe + redo flag (as mth_quartet.hip)
                uint32_t probes = 0;
                bool placed = false;
                while (probes++ < 1024u) {
                    const unsigned long long cur = atomicCAS(&a.keys[h], PKEY_EMPTY, key);
                    if (cur == PKEY_EMPTY || cur == key) { placed = true; break; }
                    h = (h + 1) & a.mask;
                }
                if (placed) atomicAdd(&a.cnt[h * 2 + (((wj ^ wk) >> 31) ? 1 : 0)], 1u);  // lpmd.rs:79-86
                else *a.overflow = 1ull;
            }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
        __shared__ unsigned long long ws[4];
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.n_updates, ws[0] + ws[1] + ws[2] + ws[3]);
    }
}

__global__ __launch_bounds__(256) void k_pairs_blockcount(const unsigned long long *__restrict__ keys,
                                                          unsigned long long n_slots, uint32_t *__restrict__ blk) {
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) m += (s0 + k < n_slots && keys[s0 + k] != PKEY_EMPTY) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_pairs_emit(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ cnt,
                                                    unsigned long long n_slots, const uint32_t *__restrict__ blk,
                                                    const unsigned long long *__restrict__ base,
                                                    unsigned long long *__restrict__ out_key, uint32_t *__restrict__ out_cnt) {
    const unsigned long long s0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    uint32_t m = 0;
    unsigned long long kk[SCAN_PER];
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) { kk[k] = s0 + k < n_slots ? keys[s0 + k] : PKEY_EMPTY; m += kk[k] != PKEY_EMPTY ? 1u : 0u; }
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
    unsigned long long o = *base + blk[blockIdx.x] + ws[wave] + incl - m;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) {
        if (kk[k] == PKEY_EMPTY) continue;
        out_key[o] = kk[k];
        out_cnt[2 * o] = cnt[(s0 + k) * 2];
        out_cnt[2 * o + 1] = cnt[(s0 + k) * 2 + 1];
        ++o;
    }
}


// ---- tile kernel ---------------------------------------------------------------------------------------------------
struct PTileArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off, *cpg_pos, *idx;
    const void     *cpg_rel;
    int32_t region_beg, region_end, idx_base, max_span, min_dist, max_dist;
    uint32_t n_reads, ntiles, n_cpgs;
    uint8_t min_qual, force_heavy;            // force_heavy: tests send every tile down the global path
    unsigned long long *row_total;            // rows claimed so far (all batches, gaps included)
    unsigned long long row_cap;               // rows the output holds; a range beyond it is claimed but not written ...
    unsigned long long *unfit;                // ... and reported here (the host redoes the batch with the exact size)
    unsigned long long *n_heavy;              // tiles left to the global path
    uint32_t *tile_flag;                      // per tile of the batch: 1 = left to the global path
    unsigned long long *tile_row0;            // per tile: first row of its range ...
    uint32_t *tile_rows;                      // ... and its length
    unsigned long long *out_key; uint32_t *out_cnt;
    DevState *st;
};
template <typename RelT, int PT_SHIFT>
__global__ __launch_bounds__(PT_B, PT_OCC) void k_pairs_tile(const PTileArgs a) {
    constexpr int PT_W = 1 << PT_SHIFT;
    constexpr int DBITS = 32 - PT_SHIFT;              // key = (pos1 - tile start) << DBITS | (pos2 - pos1)
    // slot h: tab[2h] = counters (concordant | discordant << 16), tab[2h+1] = key; as a 64-bit word key is the high half
    __shared__ unsigned long long tab64[PT_S];
    __shared__ uint32_t s_heavy, ws[PT_B / 64 + 1];
    __shared__ uint32_t bcnt[PT_B], bbase[PT_B];        // bucket sort: words per bucket, first rank of the bucket
    __shared__ uint32_t s_cw[PT_NC][PT_B];              // per thread (column): the first calls of the read it is working on
    __shared__ unsigned long long s_row0;
    uint32_t *tab = reinterpret_cast<uint32_t *>(tab64);
    const RelT *__restrict__ rel = reinterpret_cast<const RelT *>(a.cpg_rel);
    __shared__ unsigned long long s_chunk_pos, s_chunk_end;   // rows claimed from the global counter, handed out tile by tile
    const int tid = threadIdx.x;
    if (tid == 0) { s_chunk_pos = 0; s_chunk_end = 0; }
    // Persistent workgroups: each takes tiles t, t + grid, ... and claims output rows for several tiles at once.  (One claim
    // per tile was the kernel's floor on sparse WGBS: same-address returning atomics serialise at ~12 ns each, and a
    // human genome is 378 k tiles.)  The unused tail of a chunk stays a gap; the fetch walks per-tile ranges.
    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    __syncthreads();                                    // the previous tile's rows have left LDS
    const int32_t T0 = a.region_beg + (int32_t)(t * PT_W);
    const int32_t T1 = (int32_t)min((int64_t)T0 + PT_W, (int64_t)a.region_end);
    const uint32_t lo = min(a.idx[((uint32_t)T0 - (uint32_t)a.max_span + 1u - (uint32_t)a.idx_base) >> IDX_QSHIFT], a.n_reads);
    const uint32_t hi = min(a.idx[(((uint32_t)T0 + (uint32_t)PT_W - (uint32_t)a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
    if (lo >= hi) {
        if (tid == 0) { a.tile_flag[t] = 0u; a.tile_rows[t] = 0u; a.tile_row0[t] = 0ull; }
        continue;
    }
    for (int i = tid; i < PT_S; i += PT_B) tab64[i] = 0xffffffff00000000ull;
    bcnt[tid] = 0;
    if (tid == 0) s_heavy = (hi - lo > 65535u || a.force_heavy) ? 1u : 0u;   // a counter takes at most one update per read
    __syncthreads();
    if (!s_heavy) {
        uint32_t bad = 0;
        // (distances between live calls are < 2^16; dead slots sit (k + 1) << 24 away, beyond every capped max_distance)
        const int32_t maxd = min(a.max_dist, 1 << 20), mind = max(a.min_dist, 0);
        // one pair of a read into the tile's table (lpmd.rs:70-87); the pair belongs to the tile of its first position
        auto insert = [&](const uint32_t wj, const uint32_t wk) {
            const int32_t p1 = (int32_t)(wj & 0x7fffffffu);
            if (p1 < T0 || p1 >= T1) return;
            const uint32_t delta = (wk & 0x7fffffffu) - (uint32_t)p1;
            if (delta >= (1u << DBITS) - 1u) { s_heavy = 1u; return; }      // does not fit the 32-bit key (also: calls out of order)
            const uint32_t key = ((uint32_t)(p1 - T0) << DBITS) | delta;
            uint32_t h = (key * 0x9E3779B1u) >> (32 - 11), probes = 0;
            static_assert(PT_S == 1 << 11, "slot hash takes the top 11 bits");
            bool placed = false;
            while (probes++ < (uint32_t)PT_S) {
                const uint32_t cur = atomicCAS(&tab[2 * h + 1], 0xffffffffu, key);
                if (cur == 0xffffffffu || cur == key) { placed = true; break; }
                h = (h + 1) & (PT_S - 1);
            }
            if (placed) atomicAdd(&tab[2 * h], ((wj ^ wk) >> 31) ? 0x10000u : 1u);    // lpmd.rs:79-86
            else s_heavy = 1u;                                                          // more distinct pairs than slots
        };
        // PT_U reads per thread and round: their offsets, then ALL their calls (the first PT_NC of each: two 16-byte loads of the
        // call words and one of the relative positions, from a per-read base -- the PDR tile kernel's form) are requested before
        // any of them is used.  (Round 2's form fetched 2 x PT_NC single words per read behind a first / last look-up -- four times
        // the PDR kernel's load instructions -- and walked the pairs in two nested loops over an LDS column: a chain of
        // dependent LDS look-ups with divergent trip counts, 62 % of the wave cycles waiting; profiles/r03_mhl_sparse.md.)
        // Two round trips per round would be offsets -> calls; the NEXT round's offsets are requested right behind this round's
        // calls (the PDR tile kernel's PF form), start and mapq travel with the calls, so a round waits once.
        uint32_t o0s[PT_U], o1s[PT_U];
#pragma unroll
        for (int u = 0; u < PT_U; ++u) {
            const uint32_t ii = min(lo + (uint32_t)u * PT_B + tid, hi - 1);
            o0s[u] = a.cpg_off[ii]; o1s[u] = a.cpg_off[ii + 1];
        }
        for (uint32_t b0 = lo; b0 < hi; b0 += PT_B * PT_U) {
          int32_t st[PT_U];
          uint32_t mq[PT_U], o0n[PT_U], o1n[PT_U];
          bool ok[PT_U];
          bool inside = true;
          uint32_t vv[PT_U][PT_NC], rw[PT_U][4];
#pragma unroll
          for (int u = 0; u < PT_U; ++u) {
              const uint32_t i = b0 + (uint32_t)u * PT_B + tid;
              ok[u] = i < hi && o1s[u] - o0s[u] >= 2;
              inside = inside && (!ok[u] || (unsigned long long)o0s[u] + PT_NC <= (unsigned long long)a.n_cpgs);
          }
          static_assert(PT_NC == 8, "two 16-byte loads per read");
          if (__all(inside)) {                     // wave-uniform: only the batch's last few reads have a window that leaves the arrays
#pragma unroll
              for (int u = 0; u < PT_U; ++u) {
                  if (!ok[u]) continue;
                  const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0s[u]), y = *reinterpret_cast<const u32x4_a4 *>(a.cpg_pos + o0s[u] + 4);
                  vv[u][0] = x.x; vv[u][1] = x.y; vv[u][2] = x.z; vv[u][3] = x.w; vv[u][4] = y.x; vv[u][5] = y.y; vv[u][6] = y.z; vv[u][7] = y.w;
                  if constexpr (sizeof(RelT) == 1) { const u32x2_a1 z = *reinterpret_cast<const u32x2_a1 *>(rel + o0s[u]); rw[u][0] = z.x; rw[u][1] = z.y; }
                  else { const u32x4_a2 z = *reinterpret_cast<const u32x4_a2 *>(rel + o0s[u]); rw[u][0] = z.x; rw[u][1] = z.y; rw[u][2] = z.z; rw[u][3] = z.w; }
              }
          } else {
#pragma unroll
              for (int u = 0; u < PT_U; ++u) {
                  if (!ok[u]) continue;
                  const uint32_t nc = o1s[u] - o0s[u];
                  rw[u][0] = rw[u][1] = rw[u][2] = rw[u][3] = 0;
#pragma unroll
                  for (int k = 0; k < PT_NC; ++k) {
                      const uint32_t kk = o0s[u] + min((uint32_t)k, nc - 1);
                      vv[u][k] = a.cpg_pos[kk];
                      if constexpr (sizeof(RelT) == 1) rw[u][k >> 2] |= (uint32_t)rel[kk] << (8 * (k & 3));
                      else rw[u][k >> 1] |= (uint32_t)rel[kk] << (16 * (k & 1));
                  }
              }
          }
#pragma unroll
          for (int u = 0; u < PT_U; ++u) {
              const uint32_t ii = min(b0 + (uint32_t)u * PT_B + tid, hi - 1);
              st[u] = a.read_start[ii]; mq[u] = a.read_mapq[ii];
              const uint32_t in = min(b0 + (uint32_t)(PT_U + u) * PT_B + tid, hi - 1);
              o0n[u] = a.cpg_off[in]; o1n[u] = a.cpg_off[in + 1];
          }
#pragma unroll
          for (int u = 0; u < PT_U; ++u) ok[u] = ok[u] && mq[u] >= a.min_qual;                 // lpmd.rs:177
#pragma unroll
          for (int u = 0; u < PT_U; ++u) {
            if (!__any(ok[u])) continue;
            const uint32_t o0 = o0s[u], o1 = o1s[u];
            const uint32_t n_calls = ok[u] ? o1 - o0 : 0u, nl = min(n_calls, (uint32_t)PT_NC);
            // candidate ranges rely on every call lying in [start - 1, start - 1 + max_span] (rule of the PDR tile kernel)
            const uint32_t sm1 = (uint32_t)st[u] - 1u;
            int32_t r[PT_NC];
            uint32_t xmax = 0;
#pragma unroll
            for (int k = 0; k < PT_NC; ++k) {
                const bool live = (uint32_t)k < nl;
                const uint32_t raw = sizeof(RelT) == 1 ? (rw[u][k >> 2] >> (8 * (k & 3))) & 0xffu : (rw[u][k >> 1] >> (16 * (k & 1))) & 0xffffu;
                r[k] = live ? (int32_t)raw : (int32_t)((k + 1) << 24);
                xmax = max(xmax, live ? (vv[u][k] & 0x7fffffffu) - sm1 : 0u);
                if (ok[u]) s_cw[k][tid] = vv[u][k];                    // the thread's own LDS column: the pairs found below are looked up by slot number
            }
            bad |= (xmax > (uint32_t)a.max_span) ? 1u : 0u;
            // Pairs (j, j + g) of the first PT_NC calls with min <= rel[j + g] - rel[j] <= max (readutil.rs:166-224), diagonal by
            // diagonal: the calls are sorted, so once no lane has a distance <= max on a diagonal the later ones have none either.
            // One mask byte per diagonal (bit (7 - g) - j for the pair starting at slot j).
            uint32_t mlo = 0, mhi = 0;
#pragma unroll
            for (int g = 1; g < PT_NC; ++g) {
                uint32_t notin = 0, within = 0;
#pragma unroll
                for (int j = 0; j + g < PT_NC; ++j) {
                    const int32_t dist = r[j + g] - r[j];
                    const uint32_t below = (uint32_t)(dist - mind), above = (uint32_t)(maxd - dist);     // sign bit: outside
                    notin = __builtin_amdgcn_alignbit(notin, below | above, 31);
                    within |= ~above;
                }
                const uint32_t dm = ~notin & ((1u << (PT_NC - g)) - 1u);
                if (g <= 4) mlo |= dm << (8 * (g - 1)); else mhi |= dm << (8 * (g - 5));
                if (!__any((int32_t)within < 0)) break;
            }
            while (__any((mlo | mhi) != 0u)) {
                if (mlo | mhi) {
                    uint32_t bit;
                    if (mlo) { bit = (uint32_t)__builtin_ctz(mlo); mlo &= mlo - 1; }
                    else { bit = 32u + (uint32_t)__builtin_ctz(mhi); mhi &= mhi - 1; }
                    const uint32_t g = (bit >> 3) + 1u, j = (7u - g) - (bit & 7u);
                    insert(s_cw[j][tid], s_cw[j + g][tid]);
                }
            }
            // a read with more than PT_NC calls: the pairs whose LATER call is the (PT_NC + 1)-th or beyond, from memory (rare)
            if (__any(n_calls > (uint32_t)PT_NC) && n_calls > (uint32_t)PT_NC) {
                for (uint32_t k = o0 + PT_NC; k < o1; ++k) {
                    const int32_t rk = (int32_t)rel[k];
                    const uint32_t wk = a.cpg_pos[k];
                    bad |= ((wk & 0x7fffffffu) - sm1 > (uint32_t)a.max_span) ? 1u : 0u;
                    for (uint32_t j = k; j-- > o0;) {
                        const int32_t dist = rk - (int32_t)rel[j];
                        if (dist > a.max_dist) break;                              // readutil.rs:184
                        if (dist < a.min_dist) continue;                           // readutil.rs:196
                        insert(a.cpg_pos[j], wk);
                    }
                }
            }
          }
#pragma unroll
          for (int u = 0; u < PT_U; ++u) { o0s[u] = o0n[u]; o1s[u] = o1n[u]; }
        }
        if (bad) atomicOr(&a.st->err, (uint32_t)ERRB_SPAN);
    }
    __syncthreads();
    if (s_heavy) {                                      // block-uniform: the whole tile goes to the global path
        if (tid == 0) { a.tile_flag[t] = 1u; a.tile_rows[t] = 0u; a.tile_row0[t] = 0ull; atomicAdd(a.n_heavy, 1ull); }
        continue;
    }
    // Rows go out sorted by key.  Bucket sort on the key's top 8 bits (= 256 position ranges of the tile): every thread
    // holds its PER slots in registers, so the table is rebuilt in place in bucket order; a word's final rank = start of
    // its bucket + the words of that bucket below it (a few).
    static_assert(PT_S % PT_B == 0 && PT_B == 256, "each thread owns PT_S / PT_B slots and one bucket");
    constexpr int PER = PT_S / PT_B;
    unsigned long long kk[PER];
    uint32_t pib[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        kk[k] = tab64[tid * PER + k];
        pib[k] = 0;
        if ((kk[k] >> 32) != 0xffffffffull) pib[k] = atomicAdd(&bcnt[(uint32_t)(kk[k] >> 56)], 1u);
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
        for (int w = 1; w <= PT_B / 64; ++w) ws[w] += ws[w - 1];
        const uint32_t n_all = ws[PT_B / 64];
        a.tile_flag[t] = 0u; a.tile_rows[t] = n_all;
        if (n_all && s_chunk_pos + n_all > s_chunk_end) {       // next chunk (what is left of the old one stays unused)
            // enough for this workgroup's remaining tiles if they are like this one, at most PT_CHUNK rows: the gaps stay
            // small next to the rows (a single-tile workgroup claims exactly its rows)
            const unsigned long long left = (unsigned long long)((a.ntiles - 1u - t) / gridDim.x + 1u);
            const unsigned long long claim = max((unsigned long long)n_all, min((unsigned long long)n_all * left, (unsigned long long)PT_CHUNK));
            s_chunk_pos = atomicAdd(a.row_total, claim);
            s_chunk_end = s_chunk_pos + claim;
            if (s_chunk_end > a.row_cap) atomicAdd(a.unfit, 1ull);
        }
        s_row0 = s_chunk_pos;
        s_chunk_pos += n_all;
        a.tile_row0[t] = s_row0;
    }
    __syncthreads();
    const uint32_t n = ws[PT_B / 64];
    if (n == 0 || s_row0 + n > a.row_cap) continue;       // block-uniform
    bbase[tid] = ws[wave] + incl - m;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if ((kk[k] >> 32) != 0xffffffffull) tab64[bbase[(uint32_t)(kk[k] >> 56)] + pib[k]] = kk[k];
    __syncthreads();
    for (uint32_t j = tid; j < n; j += PT_B) {
        const unsigned long long w = tab64[j];
        const uint32_t bk = (uint32_t)(w >> 56), b0 = bbase[bk], b1 = b0 + bcnt[bk];
        uint32_t r = b0;
        for (uint32_t i = b0; i < b1; ++i) r += tab64[i] < w ? 1u : 0u;     // keys are distinct: the high halves decide
        const unsigned long long o = s_row0 + r;
        const uint32_t key = (uint32_t)(w >> 32), c = (uint32_t)w;
        const uint32_t p1 = (uint32_t)T0 + (key >> DBITS), p2 = p1 + (key & ((1u << DBITS) - 1u));
        a.out_key[o] = ((unsigned long long)p1 << 32) | p2;
        reinterpret_cast<uint2 *>(a.out_cnt)[o] = make_uint2(c & 0xffffu, c >> 16);
    }
    }
}

// restart of a batch whose rows did not fit: back to the row count before it
__global__ void k_pairs_rewind(unsigned long long *ps, unsigned long long rows_before) { ps[1] = rows_before; ps[5] = 0; ps[6] = 0; }

// a queued batch's state words as its tile kernel left them (mth_quartet.hip: k_quartet_snap)
__global__ void k_pairs_snap(const unsigned long long *__restrict__ ps, unsigned long long *__restrict__ snap) {
    if (threadIdx.x < P_STATE_WORDS) snap[threadIdx.x] = ps[threadIdx.x];
}

}  // namespace mth

using namespace mth;

// One batch, synchronous or queued: the scheme of mth_quartet.hip's quartet_batch / quartet_resolve.
static int pairs_batch(mth_ctx *ctx, const mth_batch_t &d, const mth_lpmd_pairs_params_t *params, int32_t batch_tid, bool queued) {
    int rc = MTH_OK;
    hipStream_t s = ctx->stream;
    if (!ctx->p_state.p) {
        MTH_HIP(ctx, ctx->p_state.reserve(P_STATE_WORDS * sizeof(unsigned long long), s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->p_state.p, 0, P_STATE_WORDS * sizeof(unsigned long long), s));
    }
    // [0] updates (counting pass of the global path) [1] total rows [2] first row of the global path's rows
    // [3] its overflow flag [5] tiles left to the global path [6] tiles whose rows did not fit the output
    unsigned long long *ps = ctx->p_state.as<unsigned long long>();
    const int64_t region_len = (int64_t)d.region_end - d.region_beg;
    // Tile width (see mth_quartet.hip): distinct pairs per tile ~ sites per tile x sites per pair window
    int tile_shift = 13;
    if (d.n_reads && region_len > 0) {
        const double sites_per_bp = (double)d.n_cpgs / (double)d.n_reads / (double)std::max(d.max_span, 1);
        const double reads_per_bp = (double)d.n_reads / (double)region_len;
        const double window = std::max(1.0, (double)std::min<int64_t>((int64_t)params->max_distance - std::max(params->min_distance, 0) + 1, d.max_span));
        while (tile_shift < 16 && sites_per_bp * (double)(2 << tile_shift) * std::max(1.0, sites_per_bp * window) <= 0.3 * PT_S &&
               reads_per_bp * (double)((2 << tile_shift) + d.max_span + 2 * IDX_Q) <= 30000.0)
            ++tile_shift;
    }
    if (const char *e = getenv("MTH_PAIRS_TILE_SHIFT")) tile_shift = std::min(16, std::max(13, atoi(e)));   // tests / tuning
    const int PT_W = 1 << tile_shift;
    const uint32_t ntiles = (d.n_reads && region_len > 0) ? (uint32_t)((region_len + PT_W - 1) / PT_W) : 0u;
    const uint64_t tiles_before = ctx->p_meta.empty() ? 0 : ctx->p_meta.back().tile_end;
    const uint64_t rows_before = queued && !ctx->p_pending.empty() ? ctx->p_rows_est : ctx->p_rows;
    mth_ctx::TileBatch meta{batch_tid, 0, rows_before, tiles_before + ntiles};
    if (!ntiles && queued && !ctx->p_pending.empty()) queued = false, rc = pairs_resolve(ctx);
    if (rc) return rc;
    if (!ntiles) { meta.heavy0 = ctx->p_rows; ctx->p_meta.push_back(meta); return MTH_OK; }
    const bool r8 = d.cpg_rel != nullptr;
    int32_t idx_base = 0;
    uint32_t nt = 0;
    rc = build_read_index(ctx, d, PT_W, idx_base, nt);
    if (rc) return rc;
    MTH_HIP(ctx, ctx->p_tflag.reserve((size_t)ntiles * 4, s));
    MTH_HIP(ctx, ctx->p_tile_row0.reserve((tiles_before + ntiles) * 8, s, true, tiles_before * 8));
    MTH_HIP(ctx, ctx->p_tile_rows.reserve((tiles_before + ntiles) * 4, s, true, tiles_before * 4));
    // output size: rows per CpG call of the batches so far (first batch: a guess); the kernel reports the exact need
    uint64_t want = rows_before + (uint64_t)((double)d.n_cpgs * ctx->p_rows_per_cpg * 1.25) + 4096;
    if (const char *e = getenv("MTH_PAIRS_ROWS_MIN")) want = rows_before + strtoull(e, nullptr, 10);   // tests: force the redo
    unsigned long long *st = ctx->h_words;     // pinned: the read-back does not go through a staging copy
    for (int attempt = 0;; ++attempt) {
        if (want > ctx->p_cap) {
            const uint64_t cap = want + (queued ? want / 4 : 0), used = std::min<uint64_t>(rows_before, ctx->p_cap);
            MTH_HIP(ctx, ctx->p_out_key.reserve(cap * 8, s, true, used * 8));
            MTH_HIP(ctx, ctx->p_out_cnt.reserve(cap * 8, s, true, used * 8));
            ctx->p_cap = cap;
        }
        if (!queued || ctx->p_pending.empty()) hipLaunchKernelGGL(k_pairs_rewind, dim3(1), dim3(1), 0, s, ps, (unsigned long long)rows_before);
        PTileArgs a;
        a.read_start = d.read_start; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
        a.idx = idx_ptr(ctx); a.cpg_rel = r8 ? (const void *)d.cpg_rel : (const void *)d.cpg_rel16;
        a.region_beg = d.region_beg; a.region_end = d.region_end; a.idx_base = idx_base; a.max_span = d.max_span;
        a.min_dist = params->min_distance; a.max_dist = params->max_distance; a.n_reads = d.n_reads; a.ntiles = ntiles; a.n_cpgs = (uint32_t)d.n_cpgs;
        a.min_qual = params->min_qual; a.force_heavy = getenv("MTH_PAIRS_FORCE_GLOBAL") ? 1 : 0;
        // (a queued batch must stay within its estimate: see quartet_batch)
        a.row_total = ps + 1; a.row_cap = queued ? std::min<uint64_t>(ctx->p_cap, want) : ctx->p_cap; a.unfit = ps + 6; a.n_heavy = ps + 5;
        a.tile_flag = ctx->p_tflag.as<uint32_t>();
        a.tile_row0 = ctx->p_tile_row0.as<unsigned long long>() + tiles_before;
        a.tile_rows = ctx->p_tile_rows.as<uint32_t>() + tiles_before;
        a.out_key = ctx->p_out_key.as<unsigned long long>(); a.out_cnt = ctx->p_out_cnt.as<uint32_t>(); a.st = ctx->d_state;
        {
            LaunchTimer lt(ctx, K_PAIRSTILE);
            if (tile_shift == 13) {
                if (r8) hipLaunchKernelGGL((k_pairs_tile<uint8_t, 13>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
                else hipLaunchKernelGGL((k_pairs_tile<uint16_t, 13>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
            } else if (tile_shift == 14) {
                if (r8) hipLaunchKernelGGL((k_pairs_tile<uint8_t, 14>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
                else hipLaunchKernelGGL((k_pairs_tile<uint16_t, 14>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
            } else if (tile_shift == 15) {
                if (r8) hipLaunchKernelGGL((k_pairs_tile<uint8_t, 15>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
                else hipLaunchKernelGGL((k_pairs_tile<uint16_t, 15>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
            } else {
                if (r8) hipLaunchKernelGGL((k_pairs_tile<uint8_t, 16>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
                else hipLaunchKernelGGL((k_pairs_tile<uint16_t, 16>), dim3(std::min<uint32_t>(ntiles, PT_GRID)), dim3(PT_B), 0, s, a);
            }
        }
        if (queued) {
            const size_t k = ctx->p_pending.size();
            MTH_HIP(ctx, ctx->p_snap.reserve((size_t)P_QUEUE_MAX * P_STATE_WORDS * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k_pairs_snap, dim3(1), dim3(64), 0, s, (const unsigned long long *)ps, ctx->p_snap.as<unsigned long long>() + k * P_STATE_WORDS);
            MTH_HIP(ctx, hipGetLastError());
            ctx->p_pending.push_back(mth_ctx::QueuedPairs{d, *params, batch_tid, d.n_cpgs});
            ctx->p_rows_est = want;
            ctx->p_meta.push_back(meta);                   // rows / heavy0: pairs_resolve
            return MTH_OK;
        }
        MTH_HIP(ctx, hipMemcpyAsync(st, ps, P_STATE_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));            // one sync per batch: rows, flagged tiles, fit
        if (!st[6]) break;
        if (attempt) return fail(ctx, MTH_ERR_STATE, "pairs: rows did not fit an exactly sized output");
        want = st[1];                                     // every tile claimed its range: this is the exact size
    }
    uint64_t total = st[1];
    meta.heavy0 = total;
    if (st[5]) {
        // The tiles the LDS table could not hold: the global table, for their pairs only.
        PairArgs g;
        g.read_mapq = d.read_mapq; g.cpg_off = d.cpg_off; g.cpg_pos = d.cpg_pos;
        g.cpg_rel = r8 ? (const void *)d.cpg_rel : (const void *)d.cpg_rel16;
        g.keys = nullptr; g.cnt = nullptr; g.n_updates = ps; g.mask = 0; g.overflow = ps + 3;
        g.region_beg = d.region_beg; g.region_end = d.region_end; g.min_dist = params->min_distance; g.max_dist = params->max_distance;
        g.n_reads = d.n_reads; g.min_qual = params->min_qual; g.tile_flag = ctx->p_tflag.as<uint32_t>(); g.tile_shift = tile_shift;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)d.n_reads + 255) / 256 + 1, 8192);
        MTH_HIP(ctx, hipMemsetAsync(ps, 0, sizeof(unsigned long long), s));
        {
            LaunchTimer lt(ctx, K_PAIRS);
            if (r8) hipLaunchKernelGGL((k_pairs<uint8_t, true>), dim3(grid), dim3(256), 0, s, g);
            else hipLaunchKernelGGL((k_pairs<uint16_t, true>), dim3(grid), dim3(256), 0, s, g);
        }
        unsigned long long bound = 0;
        MTH_HIP(ctx, hipMemcpyAsync(&bound, ps, sizeof bound, hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));
        // `bound` counts pair UPDATES; at depth D there are ~D per distinct pair, and the table is cleared and scanned once per
        // batch: start at bound / 2 slots and redo 4x larger if an insert ran out of probes (cannot happen at 2 x bound)
        unsigned long long n_slots = 1024;
        while (n_slots < bound / 2) n_slots <<= 1;
        if (const char *e = getenv("MTH_PAIRS_SLOTS_MIN")) { const unsigned long long k = strtoull(e, nullptr, 10); if (k >= 16) { n_slots = 16; while (n_slots < k) n_slots <<= 1; } }   // tests: force the retry
        for (;;) {
            MTH_HIP(ctx, ctx->p_keys.reserve(n_slots * 8, s));
            MTH_HIP(ctx, ctx->p_cnt.reserve(n_slots * 8, s));
            MTH_HIP(ctx, hipMemsetAsync(ctx->p_keys.p, 0xFF, n_slots * 8, s));
            MTH_HIP(ctx, hipMemsetAsync(ctx->p_cnt.p, 0, n_slots * 8, s));
            MTH_HIP(ctx, hipMemsetAsync(ps + 3, 0, sizeof(unsigned long long), s));
            g.keys = ctx->p_keys.as<unsigned long long>(); g.cnt = ctx->p_cnt.as<uint32_t>(); g.mask = n_slots - 1;
            {
                LaunchTimer lt(ctx, K_PAIRS);
                if (r8) hipLaunchKernelGGL((k_pairs<uint8_t, false>), dim3(grid), dim3(256), 0, s, g);
                else hipLaunchKernelGGL((k_pairs<uint16_t, false>), dim3(grid), dim3(256), 0, s, g);
            }
            unsigned long long ovf = 0;
            MTH_HIP(ctx, hipMemcpyAsync(&ovf, ps + 3, sizeof ovf, hipMemcpyDeviceToHost, s));
            MTH_HIP(ctx, hipStreamSynchronize(s));
            if (!ovf) break;
            n_slots <<= 2;
        }
        const uint64_t need = total + bound;              // distinct pairs <= updates
        if (need > ctx->p_cap) {
            MTH_HIP(ctx, ctx->p_out_key.reserve(need * 8, s, true, total * 8));
            MTH_HIP(ctx, ctx->p_out_cnt.reserve(need * 8, s, true, total * 8));
            ctx->p_cap = need;
        }
        const uint32_t nblk = (uint32_t)((n_slots + 256 * SCAN_PER - 1) / (256 * SCAN_PER));
        MTH_HIP(ctx, ctx->w_blk.reserve((size_t)nblk * 4, s));
        MTH_HIP(ctx, ctx->p_batch_rows.reserve(4, s));
        hipLaunchKernelGGL(k_pairs_blockcount, dim3(nblk), dim3(256), 0, s, ctx->p_keys.as<unsigned long long>(), n_slots,
                           ctx->w_blk.as<uint32_t>());
        hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->w_blk.as<uint32_t>(), nblk, ps + 1, ps + 2,
                           ctx->p_batch_rows.as<uint32_t>(), 0u);
        hipLaunchKernelGGL(k_pairs_emit, dim3(nblk), dim3(256), 0, s, ctx->p_keys.as<unsigned long long>(), ctx->p_cnt.as<uint32_t>(),
                           n_slots, ctx->w_blk.as<uint32_t>(), ps + 2, ctx->p_out_key.as<unsigned long long>(),
                           ctx->p_out_cnt.as<uint32_t>());
        unsigned long long t2 = 0;
        MTH_HIP(ctx, hipMemcpyAsync(&t2, ps + 1, sizeof t2, hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));
        total = t2;
    }
    MTH_HIP(ctx, hipGetLastError());
    meta.rows = total - rows_before;
    ctx->p_rows = total;
    if (d.n_cpgs) { ctx->p_rows_per_cpg = std::max(ctx->p_rows_per_cpg * 0.5, (double)meta.rows / (double)d.n_cpgs); ctx->p_learned = true; }
    ctx->p_meta.push_back(meta);
    return MTH_OK;
}

namespace mth {

int pairs_resolve(mth_ctx *ctx) {
    if (ctx->p_pending.empty()) return MTH_OK;
    // (a replay below rebuilds its batch's read index in the context's own buffer: whatever prepared batch the latest entry point
    // worked on is not this one's)
    ctx->cur_prep = nullptr; ctx->cur_idx = nullptr;
    MTH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<mth_ctx::QueuedPairs> pend;
    pend.swap(ctx->p_pending);
    const size_t n = pend.size(), base = ctx->p_meta.size() - n;
    std::vector<unsigned long long> snap(n * P_STATE_WORDS);
    MTH_HIP(ctx, hipMemcpyAsync(snap.data(), ctx->p_snap.p, snap.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t good = 0;
    uint64_t rows = 0, cpgs = 0;
    for (; good < n; ++good) {
        const unsigned long long *w = snap.data() + good * P_STATE_WORDS;
        if (w[5] || w[6]) break;
        mth_ctx::TileBatch &m = ctx->p_meta[base + good];
        m.rows = w[1] - ctx->p_rows;
        m.heavy0 = w[1];
        rows += m.rows; cpgs += pend[good].n_cpgs;
        ctx->p_rows = w[1];
    }
    if (cpgs) ctx->p_rows_per_cpg = std::max(ctx->p_rows_per_cpg * 0.5, (double)rows / (double)cpgs);
    if (getenv("MTH_PAIRS_DEBUG")) fprintf(stderr, "[pairs] queued batches %zu, replayed %zu\n", n, n - good);     // tests
    if (good == n) return MTH_OK;
    ctx->p_meta.resize(base + good);
    for (size_t k = good; k < n; ++k) {
        const int rc = pairs_batch(ctx, pend[k].d, &pend[k].params, pend[k].tid, false);
        if (rc) return rc;
    }
    return MTH_OK;
}

}  // namespace mth

extern "C" {

int mth_lpmd_pairs_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_lpmd_pairs_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    static const bool queue_off = getenv("MTH_PAIRS_QUEUE") && atoi(getenv("MTH_PAIRS_QUEUE")) == 0;      // A/B: one sync per batch
    const bool queued = (batch->mem == MTH_MEM_DEVICE || batch->mem == MTH_MEM_PREPARED) && ctx->p_learned && !ctx->timing && !queue_off && ctx->p_pending.size() < (size_t)P_QUEUE_MAX;
    mth_batch_t d;
    ctx->tile_queue_hold = queued;
    int rc = stage_batch(ctx, *batch, d);
    ctx->tile_queue_hold = false;
    if (rc) return rc;
    if (!queued && (rc = pairs_resolve(ctx))) return rc;
    return pairs_batch(ctx, d, params, batch->tid, queued);
}

// rows sorted by ((tid,pos1),(tid,pos2)) given batches were submitted in (tid, region) order
int mth_lpmd_pairs_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos1, int32_t *pos2, float *lpmd,
                         uint32_t *n_concordant, uint32_t *n_discordant) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    // Rows live in [0, p_rows) with gaps (the unused tails of the tile kernel's chunks); the per-tile ranges say where.
    const uint64_t extent = ctx->p_rows;
    const uint64_t n_tiles = ctx->p_meta.empty() ? 0 : ctx->p_meta.back().tile_end;
    std::vector<unsigned long long> trow0(n_tiles);
    std::vector<uint32_t> trows(n_tiles);
    if (n_tiles) {
        MTH_HIP(ctx, hipMemcpy(trow0.data(), ctx->p_tile_row0.p, n_tiles * 8, hipMemcpyDeviceToHost));
        MTH_HIP(ctx, hipMemcpy(trows.data(), ctx->p_tile_rows.p, n_tiles * 4, hipMemcpyDeviceToHost));
    }
    // lpmd.rs:94 (pairs.sort()): the tiles in position order give sorted rows; only a run of batches of one contig that
    // holds rows of the global path is sorted here (batches arrive tid-ordered)
    std::vector<uint64_t> order;
    std::vector<std::pair<size_t, size_t>> unsorted_runs;
    std::vector<std::pair<size_t, int32_t>> run_tid;       // (first output row of the run, tid)
    {
        uint64_t batch_end = 0;
        const size_t nb = ctx->p_meta.size();
        for (size_t b = 0; b < nb;) {
            size_t e = b;
            const size_t run0 = order.size();
            bool unsorted = false;
            while (e < nb && ctx->p_meta[e].tid == ctx->p_meta[b].tid) {
                const auto &mb = ctx->p_meta[e];
                batch_end += mb.rows;
                for (uint64_t t = e ? ctx->p_meta[e - 1].tile_end : 0; t < mb.tile_end; ++t)
                    for (uint32_t j = 0; j < trows[t]; ++j) order.push_back(trow0[t] + j);
                for (uint64_t i = mb.heavy0; i < batch_end; ++i) { order.push_back(i); unsorted = true; }
                ++e;
            }
            if (unsorted) unsorted_runs.emplace_back(run0, order.size());
            run_tid.emplace_back(run0, ctx->p_meta[b].tid);
            b = e;
        }
    }
    const uint64_t n = order.size();
    if (n_rows) *n_rows = n;
    if (n == 0 || (!tid && !pos1 && !pos2 && !lpmd && !n_concordant && !n_discordant)) return MTH_OK;
    const bool grouped = (tid || pos1 || pos2) && has_group_batch(ctx, 4);      // rows of contig groups: back under their own contig
    std::vector<int32_t> tmp_tid, tmp_p1;
    if (grouped && !tid) { tmp_tid.resize(n); tid = tmp_tid.data(); }
    if (grouped && !pos1) { tmp_p1.resize(n); pos1 = tmp_p1.data(); }
    std::vector<unsigned long long> key(extent);
    std::vector<uint32_t> cnt(2 * extent);
    MTH_HIP(ctx, hipMemcpy(key.data(), ctx->p_out_key.p, extent * 8, hipMemcpyDeviceToHost));
    MTH_HIP(ctx, hipMemcpy(cnt.data(), ctx->p_out_cnt.p, extent * 8, hipMemcpyDeviceToHost));
    for (const auto &r : unsorted_runs)
        std::sort(order.begin() + r.first, order.begin() + r.second, [&](uint64_t x, uint64_t y) { return key[x] < key[y]; });
    if (tid)
        for (size_t k = 0; k < run_tid.size(); ++k) {
            const size_t r1 = k + 1 < run_tid.size() ? run_tid[k + 1].first : n;
            for (size_t r = run_tid[k].first; r < r1; ++r) tid[r] = run_tid[k].second;
        }
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t x = order[r];
        const uint32_t c = cnt[2 * x], dd = cnt[2 * x + 1];
        if (pos1) pos1[r] = (int32_t)(key[x] >> 32);
        if (pos2) pos2[r] = (int32_t)(key[x] & 0x7fffffffull);
        if (n_concordant) n_concordant[r] = c;
        if (n_discordant) n_discordant[r] = dd;
        if (lpmd) lpmd[r] = (float)dd / ((float)c + (float)dd);           // lpmd.rs:111
    }
    if (grouped) return ungroup_rows(ctx, n, tid, pos1, 1, 1, pos2);
    return MTH_OK;
}

}  // extern "C"
