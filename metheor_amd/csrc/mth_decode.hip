// mth_decode.hip -- BAM record + XM decode on the device (the step before the hot path).
//
// Replaces, per record, BismarkRead::new + get_cpgs (readutil.rs:24-53, 323-345) as the host decoder
// (csrc/host/parallel_decode.cpp) restates them: start/end = first/last reference position covered by an
// M/=/X CIGAR run; a CpG call for every query offset q inside such a run whose XM character is z/Z, at
// abspos (flags in {0,99,147}, readutil.rs:332) or abspos-1, methylated iff 'Z'; insertions and soft clips
// advance the query only, deletions and skips the reference only.
//
// Input: the INFLATED record stream of a BAM file (what BGZF decompression yields after the header) plus the
// byte offset of every record -- the walk over the block_size fields is inherently sequential and stays on the
// host, where it overlaps the inflate.  One thread per record, two passes (count -> scan -> fill) because the
// number of calls per read is only known after the XM scan; the outputs are the SoA arrays of mth_batch_t,
// resident in HBM, ready for the measures without ever existing on the host.
// Roofline: HBM/L2-bound byte parsing (each thread streams its own ~350-byte record; lanes touch different
// cache lines, so the texture path, not the ALUs, is the limit); no MFMA.
#include <algorithm>

#include "mth_ctx.h"

namespace mth {

struct DecArgs {
    const uint8_t *raw;
    const uint64_t *off;          // n_rec + 1 byte offsets of the records (each starts with its block_size)
    uint32_t n_rec;
    int32_t *tid, *start, *end;
    uint8_t *mapq, *fwd;
    uint32_t *ncpg;               // pass 1 out
    uint2 *xm_loc;                // pass 1 out / pass 2 in: {offset of the XM string from the record core, its length}
    const unsigned long long *cpg_off;   // pass 2 in (exclusive scan of ncpg, n_rec + 1; global call indices)
    uint32_t *cpg_pos;
    uint16_t *cpg_rel;
    uint32_t *err;                // DevState.err
    uint32_t *notes;              // DevState.pad_: non-fatal findings (bit 0: a CIGAR P operation), read back with the error bits
    const unsigned long long *filt;   // --cpg-set: sorted keys tid << 32 | pos, or nullptr (no filter)
    uint64_t n_filt;
    uint32_t xm_min_mapq;         // a record WITHOUT XM:Z is an error only if its mapq >= this (lpmd.rs:176-181 filters on mapq first)
};

typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

// aux fields until XM:Z (SAM spec 4.2.4); false = malformed or absent
__device__ __forceinline__ bool dev_find_xm(const uint8_t *aux, uint32_t len, const uint8_t *&xm, uint32_t &xm_len) {
    uint32_t o = 0;
    while (o + 3 <= len) {
        const uint8_t t0 = aux[o], t1 = aux[o + 1], ty = aux[o + 2];
        o += 3;
        switch (ty) {
            case 'A': case 'c': case 'C': o += 1; break;
            case 's': case 'S': o += 2; break;
            case 'i': case 'I': case 'f': o += 4; break;
            case 'Z': case 'H': {
                const uint32_t b = o;
                // NUL search sixteen, then four bytes at a time (global memory takes unaligned loads), bytewise at the very end
                for (; o + 16 <= len; o += 16) {
                    const u32x4_a1 w4 = *reinterpret_cast<const u32x4_a1 *>(aux + o);
                    const uint32_t z = (((w4.x - 0x01010101u) & ~w4.x) | ((w4.y - 0x01010101u) & ~w4.y) | ((w4.z - 0x01010101u) & ~w4.z) |
                                        ((w4.w - 0x01010101u) & ~w4.w)) & 0x80808080u;
                    if (z) break;                                   // a NUL somewhere in these 16 bytes: the 4-byte loop below finds it
                }
                for (;;) {
                    if (o >= len) return false;
                    if (o + 4 > len) { if (aux[o] == 0) break; ++o; continue; }
                    const uint32_t w = ld_u32(aux + o);
                    const uint32_t z = (w - 0x01010101u) & ~w & 0x80808080u;
                    if (z) { o += (uint32_t)(__builtin_ctz(z) >> 3); break; }
                    o += 4;
                }
                if (o >= len) return false;
                if (ty == 'Z' && t0 == 'X' && t1 == 'M') { xm = aux + b; xm_len = o - b; return true; }
                o += 1;
                break;
            }
            case 'B': {
                if (o + 5 > len) return false;
                const uint8_t sub = aux[o];
                const uint64_t cnt = ld_u32(aux + o + 1);
                uint64_t w;
                switch (sub) {
                    case 'c': case 'C': w = 1; break;
                    case 's': case 'S': w = 2; break;
                    case 'i': case 'I': case 'f': w = 4; break;
                    default: return false;
                }
                // 64-bit: a crafted count must not wrap the cursor back into the block (the thread would never leave)
                const uint64_t nx = (uint64_t)o + 5u + cnt * w;
                if (nx > len) return false;
                o = (uint32_t)nx;
                break;
            }
            default: return false;
        }
    }
    return false;
}

// filter_isin (readutil.rs:87-95): is (tid, pos) in the sorted key array ?
__device__ __forceinline__ bool in_cpg_set(const unsigned long long *__restrict__ keys, uint64_t n, unsigned long long key) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && keys[lo] == key;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_decode(const DecArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_rec) return;
    const uint64_t o0 = a.off[i], o1 = a.off[i + 1];
    const uint8_t *p = a.raw + o0 + 4;                       // past block_size
    const uint32_t len = (uint32_t)(o1 - o0 - 4);
    bool bad = o1 < o0 + 4 + 32 || ld_u32(a.raw + o0) != len;
    uint32_t n = 0;
    int32_t tid = -1, first = -1, last = -1;
    uint8_t mapq = 0, fwd = 0;
    if (!bad) {
        tid = (int32_t)ld_u32(p);
        const int32_t pos = (int32_t)ld_u32(p + 4);
        const uint32_t l_read_name = p[8], n_cigar = ld_u16(p + 12), l_seq = ld_u32(p + 16);
        mapq = p[9];
        const uint32_t flag = ld_u16(p + 14);
        const uint64_t o_cigar = 32ull + l_read_name;
        const uint64_t o_aux = o_cigar + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
        const uint8_t *xm = nullptr;
        uint32_t xm_len = 0;
        bool have_xm = false;
        if (o_aux > len) {
            bad = true;
        } else if (FILL) {                                       // found by the count pass
            const uint2 loc = a.xm_loc[i];
            xm = p + loc.x; xm_len = loc.y; have_xm = loc.x != 0u;
        } else {
            have_xm = dev_find_xm(p + o_aux, (uint32_t)(len - o_aux), xm, xm_len);
            a.xm_loc[i] = have_xm ? make_uint2((uint32_t)(xm - p), xm_len) : make_uint2(0u, 0u);
            // readutil.rs:46: the reference panics without XM -- in every measure but lpmd for every record, in lpmd only
            // for records that pass its mapq filter (lpmd.rs:176-181 skips the others before BismarkRead::new)
            if (!have_xm && (uint32_t)mapq >= a.xm_min_mapq) atomicOr(a.err, (uint32_t)ERRB_NOXM);
        }
        if (!bad && !have_xm) { xm = p; xm_len = 0; have_xm = true; }   // (tolerated, or reported above) no calls; start / end from the CIGAR
        if (!bad && have_xm) {
            const bool forward = flag == 0u || flag == 99u || flag == 147u;   // readutil.rs:332
            fwd = forward ? 1 : 0;
            int64_t r = pos;
            uint32_t q = 0;
            unsigned long long w = FILL ? a.cpg_off[i] : 0ull;
            const uint8_t *cg = p + o_cigar;
            auto call = [&](const uint32_t qq, const int64_t rr, const uint8_t ch) {
                const int32_t ap = forward ? (int32_t)rr : (int32_t)(rr - 1);
                // --cpg-set: calls outside the set are dropped, relpos of the kept ones unchanged (readutil.rs:87-95)
                if (a.filt && !in_cpg_set(a.filt, a.n_filt, ((unsigned long long)(uint32_t)tid << 32) | (uint32_t)ap)) return;
                if (FILL) {
                    a.cpg_pos[w] = ((uint32_t)ap & 0x7fffffffu) | (ch == 'Z' ? 0x80000000u : 0u);
                    a.cpg_rel[w] = (uint16_t)qq;
                    ++w;
                }
                ++n;
            };
            for (uint32_t c = 0; c < n_cigar; ++c) {
                const uint32_t cw = ld_u32(cg + 4 * c), op = cw & 15u, ln = cw >> 4;
                if (op == 0 || op == 7 || op == 8) {                          // M = X: query and reference advance
                    if (ln) { if (first < 0) first = (int32_t)r; last = (int32_t)(r + ln - 1); }
                    const uint32_t qe = min(q + ln, xm_len);                   // (XM shorter than the query: nothing beyond it)
                    uint32_t qq = q;
                    // sixteen XM characters per load (the lanes of a wave read 64 different cache lines and nothing stays in L1:
                    // every load is an L2 round trip); a window without z / Z (72 % of them at 2 % CpG density) is skipped at once
                    for (; qq + 16 <= qe; qq += 16) {
                        const u32x4_a1 x4 = *reinterpret_cast<const u32x4_a1 *>(xm + qq);
                        uint32_t any = 0;
                        const uint32_t xw[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const uint32_t y = (xw[j] | 0x20202020u) ^ 0x7a7a7a7au; any |= (y - 0x01010101u) & ~y & 0x80808080u; }
                        if (any == 0u) continue;
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const uint8_t ch = (uint8_t)(xw[k >> 2] >> (8 * (k & 3)));
                            if (ch == 'z' || ch == 'Z') call(qq + k, r + (qq + k - q), ch);
                        }
                    }
                    // four XM characters per load; a word without z / Z (most of them) is skipped at once
                    for (; qq + 4 <= qe; qq += 4) {
                        const uint32_t x = ld_u32(xm + qq) | 0x20202020u;      // 'Z' -> 'z'
                        const uint32_t y = x ^ 0x7a7a7a7au;                    // a zero byte where the character is z / Z
                        if (((y - 0x01010101u) & ~y & 0x80808080u) == 0u) continue;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint8_t ch = xm[qq + k];
                            if (ch == 'z' || ch == 'Z') call(qq + k, r + (qq + k - q), ch);
                        }
                    }
                    for (; qq < qe; ++qq) {
                        const uint8_t ch = xm[qq];
                        if (ch == 'z' || ch == 'Z') call(qq, r + (qq - q), ch);
                    }
                    q += ln; r += ln;
                } else if (op == 1 || op == 4) {                              // I, S: query only
                    q += ln;
                } else if (op == 2 || op == 3) {                              // D, N: reference only
                    r += ln;
                } else if (op == 6 && !FILL) {
                    // P (padding): no query base, no reference base here -- rust-htslib's aligned-pairs iterator is believed to panic on
                    // it (readutil.rs:28 reference_positions_full; the crate is not under /root/reference): noted, the CLI says so
                    atomicOr(a.notes, 1u);
                }
            }
        }
    }
    if (bad) atomicOr(a.err, (uint32_t)ERRB_FORMAT);
    if (!FILL) {
        a.ncpg[i] = n;
        a.tid[i] = tid; a.start[i] = first; a.end[i] = last; a.mapq[i] = mapq; a.fwd[i] = fwd;
    }
}

// cpg_off = exclusive scan of ncpg (64-bit): per-block sums, a single-block scan of those, then each block adds
constexpr int DS_PER = 8;   // records per thread
__global__ __launch_bounds__(256) void k_dec_blocksum(const uint32_t *__restrict__ n, uint32_t n_rec, unsigned long long *__restrict__ blk) {
    __shared__ unsigned long long ws[4];
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * DS_PER;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < DS_PER; ++k) s += (i0 + k < n_rec) ? n[i0 + k] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void k_dec_blockscan(unsigned long long *__restrict__ blk, uint32_t nblk, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long ws[17];
    __shared__ unsigned long long run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (uint32_t b = 0; b < nblk; b += 1024) {
        const unsigned long long v = (b + tid < nblk) ? blk[b + tid] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) ws[wave + 1] = incl;
        __syncthreads();
        if (tid == 0) { ws[0] = run; for (int w = 1; w <= 16; ++w) ws[w] += ws[w - 1]; }
        __syncthreads();
        if (b + tid < nblk) blk[b + tid] = ws[wave] + incl - v;
        __syncthreads();
        if (tid == 0) run = ws[16];
        __syncthreads();
    }
    if (tid == 0) *total = run;
}
__global__ __launch_bounds__(256) void k_dec_offsets(const uint32_t *__restrict__ n, uint32_t n_rec, const unsigned long long *__restrict__ blk,
                                                     unsigned long long base, unsigned long long *__restrict__ off) {
    __shared__ unsigned long long ws[5];
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * DS_PER;
    uint32_t v[DS_PER];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < DS_PER; ++k) { v[k] = (i0 + k < n_rec) ? n[i0 + k] : 0u; s += v[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) ws[wave + 1] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { ws[0] = base + blk[blockIdx.x]; for (int w = 1; w <= 4; ++w) ws[w] += ws[w - 1]; }
    __syncthreads();
    unsigned long long run = ws[wave] + incl - s;
#pragma unroll
    for (int k = 0; k < DS_PER; ++k) {
        if (i0 + k < n_rec) off[i0 + k] = run;
        run += v[k];
    }
    if (i0 < n_rec && i0 + DS_PER >= n_rec) off[n_rec] = run;     // the thread holding the last record closes the array
}

// ---- contig groups (include/metheor_hip.h): extents per run of the decoded stream, then the in-place shift ----------------------------
__device__ __forceinline__ uint32_t grp_run_of(const unsigned long long *__restrict__ run_beg, uint32_t n_runs, unsigned long long i) {
    uint32_t lo = 0, hi = n_runs;                             // last run whose first read is <= i
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (run_beg[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// ext[k] = last covered position + 1 over run k's reads; st[0] = widest read, st[1] = calls at position -1 (a reverse read at 0:
// not representable once shifted) + reads without an aligned base
__global__ __launch_bounds__(256) void k_grp_extent(const int32_t *__restrict__ start, const int32_t *__restrict__ end,
                                                    const unsigned long long *__restrict__ off, const uint32_t *__restrict__ pos,
                                                    unsigned long long r0, unsigned long long n, const unsigned long long *__restrict__ run_beg,
                                                    uint32_t n_runs, uint32_t *__restrict__ ext, uint32_t *__restrict__ st) {
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * 256) {
        const unsigned long long i = r0 + t;
        const int32_t s = start[i], e = end[i];
        if (s < 0 || e < s) { atomicAdd(&st[1], 1u); continue; }
        const uint32_t k = grp_run_of(run_beg, n_runs, i);
        // (sorted reads: a wave's ends are close to each other -- the maximum rarely moves, so most atomics are skipped)
        if ((uint32_t)e + 1u > ext[k]) atomicMax(&ext[k], (uint32_t)e + 1u);
        if ((uint32_t)(e - s + 1) > st[0]) atomicMax(&st[0], (uint32_t)(e - s + 1));
        if (s == 0)
            for (unsigned long long c = off[i]; c < off[i + 1]; ++c)
                if ((pos[c] & 0x7fffffffu) == 0x7fffffffu) atomicAdd(&st[1], 1u);
    }
}

// out[k] = off[at[k]]: the call offsets at the runs' boundaries in one read-back (a BAM can have thousands of contigs)
__global__ __launch_bounds__(256) void k_grp_pick(const unsigned long long *__restrict__ off, const unsigned long long *__restrict__ at, uint32_t n,
                                                  unsigned long long *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = off[at[k]];
}

__global__ __launch_bounds__(256) void k_grp_shift(int32_t *__restrict__ start, int32_t *__restrict__ end, const unsigned long long *__restrict__ off,
                                                   uint32_t *__restrict__ pos, unsigned long long r0, unsigned long long n,
                                                   const unsigned long long *__restrict__ run_beg, const int32_t *__restrict__ run_voff, uint32_t n_runs) {
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * 256) {
        const unsigned long long i = r0 + t;
        const int32_t v = run_voff[grp_run_of(run_beg, n_runs, i)];
        if (!v) continue;
        start[i] += v; end[i] += v;
        for (unsigned long long c = off[i]; c < off[i + 1]; ++c) { const uint32_t w = pos[c]; pos[c] = ((w & 0x7fffffffu) + (uint32_t)v) | (w & 0x80000000u); }
    }
}

// a contiguous read range of ONE contig as a device-resident batch: 32-bit offsets rebased to the range's first call,
// and the widest read (max_span) reduced on the device
__global__ __launch_bounds__(256) void k_dec_rebase(const unsigned long long *__restrict__ off, uint64_t r0, uint32_t n_reads,
                                                    const int32_t *__restrict__ start, const int32_t *__restrict__ end,
                                                    uint32_t *__restrict__ off32, uint32_t *__restrict__ max_span) {
    // grid-stride, one pair of atomics per WORKGROUP: with one per wave the 156 k same-address atomicMax of a 10 M-read
    // contig serialised into 3.5 ms (profiles/r02_e2e.md)
    __shared__ uint32_t s_sp[4], s_me[4];
    const unsigned long long c0 = off[r0];
    uint32_t sp = 0, me = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i <= n_reads; i += (uint64_t)gridDim.x * 256) {
        off32[i] = (uint32_t)(off[r0 + i] - c0);
        if (i < n_reads) {
            const int32_t s = start[r0 + i], e = end[r0 + i];
            sp = max(sp, (s >= 0 && e >= s) ? (uint32_t)(e - s + 1) : 0u);
            me = max(me, e >= 0 ? (uint32_t)e + 1u : 0u);       // max_span[1]: last covered position + 1 over the reads
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sp = max(sp, (uint32_t)__shfl_down(sp, o, 64)); me = max(me, (uint32_t)__shfl_down(me, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_sp[threadIdx.x >> 6] = sp; s_me[threadIdx.x >> 6] = me; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sp = max(max(s_sp[0], s_sp[1]), max(s_sp[2], s_sp[3]));
        me = max(max(s_me[0], s_me[1]), max(s_me[2], s_me[3]));
        if (sp) atomicMax(max_span, sp);
        if (me) atomicMax(max_span + 1, me);
    }
}

// contig runs of the decoded stream: every i where tid changes opens a run (appended through an atomic counter, sorted by
// the host -- there are only as many runs as contigs); flags: bit0 a read without aligned base (start < 0), bit1 a record
// without contig (tid < 0), bit2 a read that starts before its predecessor on the same contig (input not coordinate-sorted)
__global__ __launch_bounds__(256) void k_dec_contigs(const int32_t *__restrict__ tid, const int32_t *__restrict__ start, uint64_t n,
                                                     uint32_t cap, uint32_t *__restrict__ count, uint64_t *__restrict__ run_beg,
                                                     int32_t *__restrict__ run_tid, uint32_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t t = tid[i];
    uint32_t f = (start[i] < 0 ? 1u : 0u) | (t < 0 ? 2u : 0u);
    // bit2: not coordinate-sorted inside a contig's run (a read without an aligned base has start -1 wherever it stands: bit0 reports it,
    // it is not taken for disorder)
    if (i > 0 && tid[i - 1] == t && start[i] >= 0 && start[i - 1] >= 0 && start[i] < start[i - 1]) f |= 4u;
    if (f) atomicOr(flags, f);
    if (i == 0 || tid[i - 1] != t) {
        const uint32_t k = atomicAdd(count, 1u);
        if (k < cap) { run_beg[k] = i; run_tid[k] = t; }
    }
}

}  // namespace mth

using namespace mth;

extern "C" {

}  // extern "C"

namespace mth {

// off[0..count] = base + exclusive scan of n[0..count) (64-bit); *total_host = sum.  Synchronises the stream and
// surfaces pending device errors.
int scan_u32_to_u64(mth_ctx *ctx, const uint32_t *n, uint32_t count, unsigned long long base, unsigned long long *off,
                    unsigned long long *total_host) {
    hipStream_t s = ctx->stream;
    const uint32_t nblk = (uint32_t)(((size_t)count + 256 * DS_PER - 1) / (256 * DS_PER));
    MTH_HIP(ctx, ctx->dec_blk.reserve(((size_t)nblk + 2) * 8, s));
    unsigned long long *d_total = ctx->dec_blk.as<unsigned long long>() + nblk;
    *total_host = 0;
    if (count == 0) { MTH_HIP(ctx, hipMemcpyAsync(off, &base, 8, hipMemcpyHostToDevice, s)); return sync_and_check(ctx); }
    hipLaunchKernelGGL(k_dec_blocksum, dim3(nblk), dim3(256), 0, s, n, count, ctx->dec_blk.as<unsigned long long>());
    hipLaunchKernelGGL(k_dec_blockscan, dim3(1), dim3(1024), 0, s, ctx->dec_blk.as<unsigned long long>(), nblk, d_total);
    hipLaunchKernelGGL(k_dec_offsets, dim3(nblk), dim3(256), 0, s, n, count, ctx->dec_blk.as<unsigned long long>(), base, off);
    MTH_HIP(ctx, hipMemcpyAsync(total_host, d_total, 8, hipMemcpyDeviceToHost, s));
    return sync_and_check(ctx);
}

// the decode proper: d_raw / d_off are device-resident
int decode_core(mth_ctx *ctx, const uint8_t *d_raw, const uint64_t *d_off, uint64_t n_rec, int append, mth_decoded_t *out) {
    hipStream_t s = ctx->stream;
    if (append && ctx->dec_grouped) return fail(ctx, MTH_ERR_STATE, "records appended to a decoded stream whose positions mth_decoded_group has shifted");
    if (!append) { ctx->dec_reads = 0; ctx->dec_cpgs = 0; ctx->dec_grouped = false; }
    // the SoA grows geometrically when windows are appended (a reallocation copies what is already decoded)
    const size_t R0 = (size_t)ctx->dec_reads, C0 = (size_t)ctx->dec_cpgs, nr = (size_t)n_rec, R1 = R0 + nr;
    auto grow = [&](DevBuf &b, size_t need, size_t used) -> hipError_t {
        if (need <= b.cap) return hipSuccess;
        return b.reserve(std::max(need, b.cap + b.cap / 2), s, used > 0, used);
    };
    MTH_HIP(ctx, grow(ctx->dec_tid, R1 * 4 + 4, R0 * 4));
    MTH_HIP(ctx, grow(ctx->dec_start, R1 * 4 + 4, R0 * 4));
    MTH_HIP(ctx, grow(ctx->dec_end, R1 * 4 + 4, R0 * 4));
    MTH_HIP(ctx, grow(ctx->dec_mapq, R1 + 4, R0));
    MTH_HIP(ctx, grow(ctx->dec_fwd, R1 + 4, R0));
    MTH_HIP(ctx, grow(ctx->dec_off, (R1 + 1) * 8, R0 ? (R0 + 1) * 8 : 0));
    MTH_HIP(ctx, ctx->dec_n.reserve(nr * 4 + 4, s));
    MTH_HIP(ctx, ctx->dec_xm.reserve(nr * 8 + 8, s));
    DecArgs a{};
    a.raw = d_raw; a.off = d_off; a.n_rec = (uint32_t)n_rec;
    a.tid = ctx->dec_tid.as<int32_t>() + R0; a.start = ctx->dec_start.as<int32_t>() + R0; a.end = ctx->dec_end.as<int32_t>() + R0;
    a.mapq = ctx->dec_mapq.as<uint8_t>() + R0; a.fwd = ctx->dec_fwd.as<uint8_t>() + R0; a.ncpg = ctx->dec_n.as<uint32_t>(); a.xm_loc = ctx->dec_xm.as<uint2>();
    a.err = &ctx->d_state->err; a.notes = &ctx->d_state->pad_;
    a.xm_min_mapq = ctx->dec_xm_min_mapq;
    a.filt = ctx->dec_filter_on ? ctx->dec_filter.as<unsigned long long>() : nullptr; a.n_filt = ctx->dec_filter_n;
    if (ctx->dec_filter_on && ctx->dec_filter_n == 0) a.filt = reinterpret_cast<const unsigned long long *>(ctx->d_state);   // empty set: drops every call
    unsigned long long total = 0;
    if (n_rec) {
        {
            LaunchTimer lt(ctx, K_DECODE);
            hipLaunchKernelGGL((k_decode<false>), dim3((uint32_t)((nr + 255) / 256)), dim3(256), 0, s, a);
        }
        int rc = scan_u32_to_u64(ctx, a.ncpg, a.n_rec, (unsigned long long)C0, ctx->dec_off.as<unsigned long long>() + R0, &total);
        if (rc) return rc;                     // also surfaces malformed records / a record without XM
    } else if (R0 == 0) {
        MTH_HIP(ctx, hipMemsetAsync(ctx->dec_off.p, 0, 8, s));
    }
    const size_t C1 = C0 + (size_t)total;
    MTH_HIP(ctx, grow(ctx->dec_pos, C1 * 4 + 4, C0 * 4));
    MTH_HIP(ctx, grow(ctx->dec_rel, C1 * 2 + 4, C0 * 2));
    if (n_rec && total) {
        a.cpg_off = ctx->dec_off.as<unsigned long long>() + R0;
        a.cpg_pos = ctx->dec_pos.as<uint32_t>(); a.cpg_rel = ctx->dec_rel.as<uint16_t>();
        LaunchTimer lt(ctx, K_DECODE);
        hipLaunchKernelGGL((k_decode<true>), dim3((uint32_t)((nr + 255) / 256)), dim3(256), 0, s, a);
    }
    MTH_HIP(ctx, hipGetLastError());
    ctx->dec_reads = R1; ctx->dec_cpgs = C1;
    out->n_reads = R1; out->n_cpgs = C1;
    out->tid = ctx->dec_tid.as<int32_t>(); out->start = ctx->dec_start.as<int32_t>(); out->end = ctx->dec_end.as<int32_t>();
    out->mapq = ctx->dec_mapq.as<uint8_t>(); out->fwd = ctx->dec_fwd.as<uint8_t>();
    out->cpg_off = ctx->dec_off.as<uint64_t>(); out->cpg_pos = ctx->dec_pos.as<uint32_t>(); out->cpg_rel = ctx->dec_rel.as<uint16_t>();
    return MTH_OK;
}

}  // namespace mth

extern "C" {

int mth_decode_records(mth_ctx_t *ctx, const void *raw, uint64_t n_bytes, const uint64_t *rec_off, uint64_t n_rec, int mem,
                       int append, mth_decoded_t *out) {
    if (!ctx || !out || (n_rec && (!raw || !rec_off))) return MTH_ERR_INVALID;
    if (n_rec >= (1ull << 32) - 16) return fail(ctx, MTH_ERR_CAPACITY, "more than 2^32 records in one decode call: split the stream");
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const uint8_t *d_raw = (const uint8_t *)raw;
    const uint64_t *d_off = rec_off;
    if (mem == MTH_MEM_HOST) {
        MTH_HIP(ctx, ctx->dec_raw.reserve(n_bytes + 16, s));
        MTH_HIP(ctx, ctx->dec_recoff.reserve((n_rec + 1) * 8, s));
        if (n_bytes) MTH_HIP(ctx, hipMemcpyAsync(ctx->dec_raw.p, raw, n_bytes, hipMemcpyHostToDevice, s));
        if (n_rec) MTH_HIP(ctx, hipMemcpyAsync(ctx->dec_recoff.p, rec_off, (n_rec + 1) * 8, hipMemcpyHostToDevice, s));
        d_raw = ctx->dec_raw.as<uint8_t>(); d_off = ctx->dec_recoff.as<uint64_t>();
    } else if (mem != MTH_MEM_DEVICE) {
        return fail(ctx, MTH_ERR_INVALID, "mem");
    }
    const int rc = decode_core(ctx, d_raw, d_off, n_rec, append, out);
    if (rc) return rc;
    if (mem == MTH_MEM_HOST) MTH_HIP(ctx, hipStreamSynchronize(s));   // the caller may reuse its buffers (and ours is restaged next call)
    return MTH_OK;
}

int mth_decode_set_cpg_filter(mth_ctx_t *ctx, const uint64_t *keys_sorted, uint64_t n_keys, int enabled) {
    if (!ctx || (enabled && n_keys && !keys_sorted)) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    ctx->dec_filter_on = enabled != 0;
    ctx->dec_filter_n = enabled ? n_keys : 0;
    if (enabled && n_keys) {
        for (uint64_t i = 1; i < n_keys; ++i)
            if (keys_sorted[i - 1] >= keys_sorted[i]) return fail(ctx, MTH_ERR_INVALID, "cpg-set keys must be strictly ascending");
        MTH_HIP(ctx, ctx->dec_filter.reserve((size_t)n_keys * 8, ctx->stream));
        MTH_HIP(ctx, hipMemcpyAsync(ctx->dec_filter.p, keys_sorted, (size_t)n_keys * 8, hipMemcpyHostToDevice, ctx->stream));
        MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MTH_OK;
}

int mth_decode_reserve(mth_ctx_t *ctx, uint64_t n_reads, uint64_t n_cpgs) {
    if (!ctx) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const size_t R0 = (size_t)ctx->dec_reads, C0 = (size_t)ctx->dec_cpgs, R = (size_t)n_reads, C = (size_t)n_cpgs;
    MTH_HIP(ctx, ctx->dec_tid.reserve(R * 4 + 4, s, R0 > 0, R0 * 4));
    MTH_HIP(ctx, ctx->dec_start.reserve(R * 4 + 4, s, R0 > 0, R0 * 4));
    MTH_HIP(ctx, ctx->dec_end.reserve(R * 4 + 4, s, R0 > 0, R0 * 4));
    MTH_HIP(ctx, ctx->dec_mapq.reserve(R + 4, s, R0 > 0, R0));
    MTH_HIP(ctx, ctx->dec_fwd.reserve(R + 4, s, R0 > 0, R0));
    MTH_HIP(ctx, ctx->dec_off.reserve((R + 1) * 8, s, R0 > 0, R0 ? (R0 + 1) * 8 : 0));
    MTH_HIP(ctx, ctx->dec_pos.reserve(C * 4 + 4, s, C0 > 0, C0 * 4));
    MTH_HIP(ctx, ctx->dec_rel.reserve(C * 2 + 4, s, C0 > 0, C0 * 2));
    return MTH_OK;
}

int mth_decode_set_xm_min_mapq(mth_ctx_t *ctx, uint32_t min_mapq) {
    if (!ctx) return MTH_ERR_INVALID;
    ctx->dec_xm_min_mapq = min_mapq;
    return MTH_OK;
}

int mth_decoded_contigs(mth_ctx_t *ctx, uint32_t cap, int32_t *tids, uint64_t *read_beg, uint64_t *read_end, uint32_t *n_runs,
                        uint32_t *flags) {
    if (!ctx || !n_runs || !flags || (cap && (!tids || !read_beg || !read_end))) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const uint64_t n = ctx->dec_reads;
    *n_runs = 0; *flags = 0;
    if (n == 0) return MTH_OK;
    MTH_HIP(ctx, ctx->dec_runs.reserve((size_t)cap * 12 + 64, s));
    uint8_t *base = static_cast<uint8_t *>(ctx->dec_runs.p);
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(base);                 // [0] count, [1] flags
    uint64_t *d_beg = reinterpret_cast<uint64_t *>(base + 16);
    int32_t *d_tid = reinterpret_cast<int32_t *>(base + 16 + (size_t)cap * 8);
    MTH_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, s));
    hipLaunchKernelGGL(k_dec_contigs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, ctx->dec_tid.as<int32_t>(),
                       ctx->dec_start.as<int32_t>(), n, cap, d_cnt, d_beg, d_tid, d_cnt + 1);
    uint32_t hc[2] = {0, 0};
    MTH_HIP(ctx, hipMemcpyAsync(hc, d_cnt, 8, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipStreamSynchronize(s));
    *n_runs = hc[0]; *flags = hc[1];
    ctx->dec_contig_flags = hc[1];
    const uint32_t k = std::min(hc[0], cap);
    if (k) {
        std::vector<uint64_t> hb(k);
        std::vector<int32_t> ht(k);
        MTH_HIP(ctx, hipMemcpy(hb.data(), d_beg, (size_t)k * 8, hipMemcpyDeviceToHost));
        MTH_HIP(ctx, hipMemcpy(ht.data(), d_tid, (size_t)k * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> ord(k);
        for (uint32_t i = 0; i < k; ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return hb[x] < hb[y]; });
        for (uint32_t i = 0; i < k; ++i) {
            tids[i] = ht[ord[i]]; read_beg[i] = hb[ord[i]];
            read_end[i] = (i + 1 < k) ? hb[ord[i + 1]] : (hc[0] <= cap ? n : hb[ord[i]]);
        }
    }
    return MTH_OK;
}

int mth_decoded_fetch(mth_ctx_t *ctx, int32_t *tid, int32_t *start, int32_t *end, uint8_t *mapq, uint8_t *fwd,
                      uint64_t *cpg_off, uint32_t *cpg_pos, uint16_t *cpg_rel) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    const size_t nr = (size_t)ctx->dec_reads, nc = (size_t)ctx->dec_cpgs;
    if (tid && nr) MTH_HIP(ctx, hipMemcpy(tid, ctx->dec_tid.p, nr * 4, hipMemcpyDeviceToHost));
    if (start && nr) MTH_HIP(ctx, hipMemcpy(start, ctx->dec_start.p, nr * 4, hipMemcpyDeviceToHost));
    if (end && nr) MTH_HIP(ctx, hipMemcpy(end, ctx->dec_end.p, nr * 4, hipMemcpyDeviceToHost));
    if (mapq && nr) MTH_HIP(ctx, hipMemcpy(mapq, ctx->dec_mapq.p, nr, hipMemcpyDeviceToHost));
    if (fwd && nr) MTH_HIP(ctx, hipMemcpy(fwd, ctx->dec_fwd.p, nr, hipMemcpyDeviceToHost));
    if (cpg_off) MTH_HIP(ctx, hipMemcpy(cpg_off, ctx->dec_off.p, (nr + 1) * 8, hipMemcpyDeviceToHost));
    if (cpg_pos && nc) MTH_HIP(ctx, hipMemcpy(cpg_pos, ctx->dec_pos.p, nc * 4, hipMemcpyDeviceToHost));
    if (cpg_rel && nc) MTH_HIP(ctx, hipMemcpy(cpg_rel, ctx->dec_rel.p, nc * 2, hipMemcpyDeviceToHost));
    return MTH_OK;
}

int mth_decoded_group(mth_ctx_t *ctx, uint32_t n_contigs, const int32_t *tids, const uint64_t *read_beg, const uint64_t *read_end,
                      uint32_t *n_groups, uint32_t *first_contig, int32_t *batch_tid) {
    if (!ctx || !n_groups || (n_contigs && (!tids || !read_beg || !read_end || !first_contig || !batch_tid))) return MTH_ERR_INVALID;
    *n_groups = 0;
    if (n_contigs < 2 || ctx->dec_grouped || ctx->dec_contig_flags) return MTH_OK;
    for (uint32_t k = 0; k < n_contigs; ++k) {
        if (tids[k] < 0 || read_end[k] < read_beg[k] || read_end[k] > ctx->dec_reads) return MTH_ERR_INVALID;
        if (k && (tids[k] <= tids[k - 1] || read_beg[k] != read_end[k - 1])) return MTH_OK;        // not one ascending, gap-free sequence of runs
    }
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const uint64_t r0 = read_beg[0], n = read_end[n_contigs - 1] - r0;
    if (n == 0) return MTH_OK;
    // device: the runs' first reads, their extents, two status words, later their offsets
    MTH_HIP(ctx, ctx->dec_runs.reserve(((size_t)n_contigs + 1) * 16 + (size_t)n_contigs * 8 + 64, s));
    uint8_t *base = static_cast<uint8_t *>(ctx->dec_runs.p);
    unsigned long long *d_beg = reinterpret_cast<unsigned long long *>(base);          // n_contigs + 1 entries: the runs' first reads, then the end
    unsigned long long *d_coff = d_beg + n_contigs + 1;                                // the call offsets there
    uint32_t *d_ext = reinterpret_cast<uint32_t *>(d_coff + n_contigs + 1);
    int32_t *d_voff = reinterpret_cast<int32_t *>(d_ext + n_contigs);
    uint32_t *d_st = reinterpret_cast<uint32_t *>(d_voff + n_contigs);
    std::vector<unsigned long long> hb(read_beg, read_beg + n_contigs);
    hb.push_back(read_end[n_contigs - 1]);
    MTH_HIP(ctx, hipMemcpyAsync(d_beg, hb.data(), ((size_t)n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
    MTH_HIP(ctx, hipMemsetAsync(d_ext, 0, (size_t)n_contigs * 8 + 16, s));
    hipLaunchKernelGGL(k_grp_pick, dim3((n_contigs + 1 + 255) / 256), dim3(256), 0, s, ctx->dec_off.as<unsigned long long>(), d_beg, n_contigs + 1, d_coff);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_grp_extent, dim3(grid), dim3(256), 0, s, ctx->dec_start.as<int32_t>(), ctx->dec_end.as<int32_t>(),
                       ctx->dec_off.as<unsigned long long>(), ctx->dec_pos.as<uint32_t>(), (unsigned long long)r0, (unsigned long long)n,
                       d_beg, n_contigs, d_ext, d_st);
    std::vector<uint32_t> ext(n_contigs);
    uint32_t st[2] = {0, 0};
    std::vector<unsigned long long> coff((size_t)n_contigs + 1);
    MTH_HIP(ctx, hipMemcpyAsync(ext.data(), d_ext, (size_t)n_contigs * 4, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipMemcpyAsync(st, d_st, 8, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipMemcpyAsync(coff.data(), d_coff, ((size_t)n_contigs + 1) * 8, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipStreamSynchronize(s));
    if (st[1]) return MTH_OK;                                  // a call at -1 / a read without an aligned base: leave the stream as it is
    // The gap after a contig: wider than anything a measure looks across -- a read's span, PDR's flush margin (pdr.rs:162: 150), the
    // FDRP window (fdrp.rs:10: 201) and the index quanta -- and a multiple of the dense tile width.
    const int64_t gap = (int64_t)st[0] + 1024;
    // ... and no wider than the work buffers can be: the passes keep per-position scratch rows (16 B a position for the PDR / MHL tile
    // passes and site discovery, a few 4-byte columns on top) -- 64 B a position of what is free now, so that a GPU shared with other
    // work falls back towards one batch per contig instead of failing an allocation
    int64_t vmax = ((int64_t)1 << 31) - ((int64_t)1 << 22);
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) vmax = std::min<int64_t>(vmax, (int64_t)(fr / 64));
        if (const char *e = getenv("MTH_GROUP_MAX_POSITIONS")) vmax = std::min<int64_t>(vmax, std::max<long long>(atoll(e), 4096));      // tests
    }
    std::vector<int32_t> voff(n_contigs, 0);
    std::vector<uint32_t> first;
    int64_t vlen = 0;
    uint64_t g_reads = 0, g_calls = 0;
    for (uint32_t k = 0; k < n_contigs; ++k) {
        const int64_t e = (((int64_t)ext[k] + gap + 4095) / 4096) * 4096;
        const uint64_t kr = read_end[k] - read_beg[k], kc = coff[k + 1] - coff[k];
        if (first.empty() || vlen + e > vmax || g_reads + kr >= (1ull << 32) - 1 || g_calls + kc >= (1ull << 32)) { first.push_back(k); vlen = 0; g_reads = g_calls = 0; }
        voff[k] = (int32_t)vlen;
        vlen += e; g_reads += kr; g_calls += kc;
    }
    if (first.size() == n_contigs) return MTH_OK;              // nothing to merge
    MTH_HIP(ctx, hipMemcpyAsync(d_voff, voff.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_grp_shift, dim3(grid), dim3(256), 0, s, ctx->dec_start.as<int32_t>(), ctx->dec_end.as<int32_t>(),
                       ctx->dec_off.as<unsigned long long>(), ctx->dec_pos.as<uint32_t>(), (unsigned long long)r0, (unsigned long long)n,
                       d_beg, (const int32_t *)d_voff, n_contigs);
    MTH_HIP(ctx, hipGetLastError());
    MTH_HIP(ctx, hipStreamSynchronize(s));                      // voff / hb are locals
    ctx->dec_grouped = true;
    first.push_back(n_contigs);
    for (size_t g = 0; g + 1 < first.size(); ++g) {
        const uint32_t k0 = first[g], k1 = first[g + 1];
        first_contig[g] = k0;
        if (k1 - k0 == 1) { batch_tid[g] = tids[k0]; continue; }
        std::vector<int64_t> vo(voff.begin() + k0, voff.begin() + k1);
        const int rc = mth_group_define(ctx, k1 - k0, tids + k0, vo.data(), &batch_tid[g]);
        if (rc) return rc;
    }
    first_contig[first.size() - 1] = n_contigs;
    *n_groups = (uint32_t)first.size() - 1;
    return MTH_OK;
}

int mth_decoded_batch(mth_ctx_t *ctx, uint64_t read_beg, uint64_t read_end, int32_t tid, int32_t region_beg, int32_t region_end,
                      mth_batch_t *batch) {
    if (!ctx || !batch || read_end < read_beg || read_end > ctx->dec_reads) return MTH_ERR_INVALID;
    const uint64_t n = read_end - read_beg;
    if (n >= (1ull << 32) - 1) return fail(ctx, MTH_ERR_CAPACITY, "batch of more than 2^32 reads");
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    MTH_HIP(ctx, ctx->dec_off32.reserve((size_t)(n + 1) * 4 + 16, s));
    uint32_t *d_span = ctx->dec_off32.as<uint32_t>() + n + 1;
    MTH_HIP(ctx, hipMemsetAsync(d_span, 0, 8, s));
    hipLaunchKernelGGL(k_dec_rebase, dim3((uint32_t)std::min<uint64_t>((n + 1 + 255) / 256, 2048)), dim3(256), 0, s, ctx->dec_off.as<unsigned long long>(),
                       read_beg, (uint32_t)n, ctx->dec_start.as<int32_t>(), ctx->dec_end.as<int32_t>(),
                       ctx->dec_off32.as<uint32_t>(), d_span);
    uint32_t span_end[2] = {0, 0};
    unsigned long long c01[2] = {0, 0};
    MTH_HIP(ctx, hipMemcpyAsync(span_end, d_span, 8, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipMemcpyAsync(&c01[0], ctx->dec_off.as<unsigned long long>() + read_beg, 8, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipMemcpyAsync(&c01[1], ctx->dec_off.as<unsigned long long>() + read_end, 8, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipStreamSynchronize(s));
    if (c01[1] - c01[0] >= (1ull << 32)) return fail(ctx, MTH_ERR_CAPACITY, "batch of more than 2^32 CpG calls: split the contig into regions");
    mth_batch_t b{};
    const uint32_t span = span_end[0];
    // region_end < 0: to the end of the DATA -- one past the last position a call of these reads can have (a reverse
    // read reports start - 1 <= end), whatever the header's LN says: the reference emits every site it sees
    if (region_end < 0) region_end = (int32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)span_end[1] + 1u, (uint64_t)std::max(region_beg, 0)), (uint64_t)INT32_MAX);
    b.tid = tid; b.region_beg = region_beg; b.region_end = region_end; b.max_span = (int32_t)span;
    b.n_reads = (uint32_t)n; b.n_cpgs = (uint32_t)(c01[1] - c01[0]); b.mem = MTH_MEM_DEVICE;
    b.read_start = ctx->dec_start.as<int32_t>() + read_beg; b.read_end = ctx->dec_end.as<int32_t>() + read_beg;
    b.read_mapq = ctx->dec_mapq.as<uint8_t>() + read_beg; b.read_fwd = ctx->dec_fwd.as<uint8_t>() + read_beg;
    b.cpg_off = ctx->dec_off32.as<uint32_t>();
    b.cpg_pos = ctx->dec_pos.as<uint32_t>() + c01[0];
    b.cpg_rel = nullptr; b.cpg_rel16 = ctx->dec_rel.as<uint16_t>() + c01[0];
    *batch = b;
    return MTH_OK;
}

}  // extern "C"
