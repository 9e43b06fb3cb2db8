// mth_sites.hip -- "site walk": measures whose reference implementation flushes per-site state while
// streaming.  Built here: MHL (mhl.rs:135-208, 43-73; readutil.rs:147-164).
//
// Exact stream semantics without a stream: all reads that can contribute to site c -- and every
// read that sits BETWEEN two contributions in file order -- have start in [c-max_span+1, c+1], a
// contiguous index range of the coordinate-sorted batch.  One thread per site walks that range in
// file order (out of LDS: k_mhl_walk_lds stages a block's candidates once) and applies the reference's loop literally:
//   flusher k (any read with >= 1 CpG, mhl.rs:162-173) with c < first_cpg(k)  -> the open segment is
//       finalised (kept only if coverage >= min_depth; a later qualifying segment overwrites it) and
//       removed; a later contribution re-opens the site
//   contributor (mapq >= min_qual, n_cpgs >= min_cpgs, has a call at c)       -> coverage, n_r, stretch counts
//   end of range                                                               -> final flush (mhl.rs:201-205)
// f32 work follows compute_mhl(): denom accumulated in f32 in read order, terms summed over
// ascending l (the reference iterates a HashMap: its own order is random), no FMA contraction.
//
// Sites come from the tile pipeline run as a site-discovery pass (positions called by >= 1
// contributor), redirected into a private sink.
#include "mth_ctx.h"
#include "mth_scan.h"

namespace mth {

struct WalkArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const uint32_t *idx;
    const DevState *sites_st;   // n_sites of the discovery pass
    const int32_t  *site_pos;
    const uint32_t *site_nc, *site_nd;   // discovery's per-site counts: contributors (reads passing the filters that call the site)
    DevState *st;               // main state (error bits)
    float    *val;              // per candidate site
    uint32_t *cov;
    uint32_t *flags;            // bit0 keep (some segment reached min_depth), bit1 needs the big variant, 8 = needs the HBM-scratch variant
    uint32_t *huge_any;         // set by k_mhl_walk_big when it hands a site to k_mhl_walk_huge
    int32_t idx_base, max_span;
    uint32_t n_reads, min_depth, min_cpgs;
    uint8_t min_qual;
};

// The fast variant again, with the gathers taken out.  One thread per site is efficient in VALU terms (every
// lane busy) but each candidate costs ~5 dependent gathers from three arrays, while neighbouring sites share
// almost all their candidates.  A block takes 256 consecutive sites and stages ALL its candidate reads in LDS with
// coalesced loads -- 16-bit call offsets, mapq, and the calls as 16-bit words (15-bit position relative to the
// block, state in bit 15) -- then every thread walks its own range [lo, hi) out of LDS.  A block whose candidates
// do not fit (too many reads or calls, or positions spread over >= 2^15 bp) runs the same walk on global memory.
constexpr int MW_RC = 3072;    // reads staged per block
constexpr int MW_CC = 10240;   // call words staged per block
constexpr int MW_MAX_PARTS = 2; // a group is staged whole or as two runs of waves; deeper data walks global memory (measured:
                                // four runs of 64 sites are slower than the global walk at 100x depth)

struct MhlGlobalSrc {
    const uint32_t *cpg_off, *cpg_pos;
    const uint8_t *mapq;
    uint32_t k_end;     // one past the last call that may be read
    __device__ __forceinline__ uint32_t off(uint32_t i) const { return cpg_off[i]; }
    __device__ __forceinline__ uint32_t mq(uint32_t i) const { return mapq[i]; }
    __device__ __forceinline__ int32_t pos(uint32_t k) const { return (int32_t)(cpg_pos[k] & 0x7fffffffu); }
    __device__ __forceinline__ uint32_t meth(uint32_t k) const { return cpg_pos[k] >> 31; }
};
struct MhlLdsSrc {
    const uint16_t *ofs, *call;
    const uint8_t *mapq;
    uint32_t r0;
    int32_t base;
    uint32_t k_end;     // one past the last staged call
    __device__ __forceinline__ uint32_t off(uint32_t i) const { return ofs[i - r0]; }
    __device__ __forceinline__ uint32_t mq(uint32_t i) const { return mapq[i - r0]; }
    __device__ __forceinline__ int32_t pos(uint32_t k) const { return base + (int32_t)(call[k] & 0x7fffu); }
    __device__ __forceinline__ uint32_t meth(uint32_t k) const { return call[k] >> 15; }
};

// The per-site walk (file order over the site's candidate reads [lo, hi)), read lengths up to LCAP = 16 CpGs.
// Both accumulations are "add the vector v_m[l] = max(0, m-l+1), l = 1..16": S (mhl.rs:36-41) once per maximal
// methylated run of length m (the run contributes m-l+1 windows of length l), D (mhl.rs:53-58) once per covering
// read with m = n_r.  Unrolled over 16 register slots that is 32-80 VALU per event, and under SIMT the whole wave
// pays for every event of any lane.  Here v_m comes from a 17-row LDS table packed as 16-bit pairs: two 16-byte
// LDS reads + 8 packed adds per event.  The packed sums spill into 32-bit accumulators every 2048 covering reads
// (a field grows by at most 16 per read), so the counts are exact at any depth; D is the reference's f32 running
// sum of integers, exact (and order-free) below 2^24 -- deeper sites go to the sequential variant.
constexpr int MHL_SPILL = 2048;
// SPILL: keep 32-bit accumulators behind the packed 16-bit ones (exact at any depth; 32 more registers).  Without them a
// segment that reaches MHL_SPILL - 1 covering reads is handed to the sequential variant (flags = 2).
template <typename Src, bool SPILL>
__device__ __forceinline__ void mhl_walk_site(const WalkArgs &a, const Src &src, const uint4 *__restrict__ vtab,
                                              const uint32_t j, const int32_t c, const uint32_t lo, const uint32_t hi) {
    constexpr int LCAP = 16;
    uint32_t Sp[LCAP / 2], Dp[LCAP / 2];     // packed pairs: low half l = 2w+1, high half l = 2w+2
    uint32_t S32[LCAP], D32[LCAP];
#pragma unroll
    for (int w = 0; w < LCAP / 2; ++w) { Sp[w] = 0; Dp[w] = 0; }
#pragma unroll
    for (int l = 0; l < LCAP; ++l) { S32[l] = 0; D32[l] = 0; }
    uint32_t seg_cov = 0, maxn = 0, res_cov = 0;
    float res = 0.0f;
    bool have = false, overflow = false;
    auto add_vec = [&](uint32_t (&acc)[LCAP / 2], const uint32_t m) {
        const uint4 x = vtab[2 * m], y = vtab[2 * m + 1];
        acc[0] += x.x; acc[1] += x.y; acc[2] += x.z; acc[3] += x.w;
        acc[4] += y.x; acc[5] += y.y; acc[6] += y.z; acc[7] += y.w;
    };
    auto spill = [&]() {
#pragma unroll
        for (int w = 0; w < LCAP / 2; ++w) {
            S32[2 * w] += Sp[w] & 0xffffu; S32[2 * w + 1] += Sp[w] >> 16; Sp[w] = 0;
            D32[2 * w] += Dp[w] & 0xffffu; D32[2 * w + 1] += Dp[w] >> 16; Dp[w] = 0;
        }
    };
    auto finalize = [&]() {   // compute_mhl, mhl.rs:43-73
        spill();
        float l_sum = 0.0f;
        for (uint32_t l = 1; l < maxn + 1; ++l) l_sum = l_sum + (float)l;
        float mhl = 0.0f;
#pragma unroll
        for (int l = 1; l <= LCAP; ++l)
            if (S32[l - 1] > 0) { const float t = ((float)l * (float)S32[l - 1]) / (float)D32[l - 1]; mhl = mhl + t; }
        return mhl / l_sum;
    };
    // Only the LAST segment that reaches min_depth is reported, and the lanes of a wave close their segments at different
    // reads: evaluating compute_mhl at every close made the whole wave run its ~300 instructions a dozen times per 64
    // sites.  A close now only keeps the segment's packed counters (16 registers); the f32 evaluation happens once, after
    // the walk, with all lanes together.  (A segment deep enough to have spilled its 16-bit counters is evaluated on the spot.)
    uint32_t qS[LCAP / 2], qD[LCAP / 2], q_maxn = 0;
    bool pending = false;
#pragma unroll
    for (int w = 0; w < LCAP / 2; ++w) { qS[w] = 0; qD[w] = 0; }
    auto close_segment = [&]() {
        res_cov = seg_cov; have = true;
        if (!SPILL || seg_cov < (uint32_t)MHL_SPILL) {
#pragma unroll
            for (int w = 0; w < LCAP / 2; ++w) { qS[w] = Sp[w]; qD[w] = Dp[w]; }
            q_maxn = maxn; pending = true;
        } else { res = finalize(); pending = false; }
    };
    auto finalize_packed = [&]() {   // compute_mhl (mhl.rs:43-73) on the kept counters: same operations, same order as finalize()
        float l_sum = 0.0f;
        for (uint32_t l = 1; l < q_maxn + 1; ++l) l_sum = l_sum + (float)l;
        float mhl = 0.0f;
#pragma unroll
        for (int l = 1; l <= LCAP; ++l) {
            const uint32_t S = (l & 1) ? (qS[(l - 1) >> 1] & 0xffffu) : (qS[(l - 1) >> 1] >> 16);
            const uint32_t D = (l & 1) ? (qD[(l - 1) >> 1] & 0xffffu) : (qD[(l - 1) >> 1] >> 16);
            if (S > 0) { const float t = ((float)l * (float)S) / (float)D; mhl = mhl + t; }
        }
        return mhl / l_sum;
    };
    // Software pipeline: a read's call offsets, mapq and FIRST FOUR CALLS (82 % of WGBS reads have no more) are requested
    // one iteration ahead, so the flush test, the hit test and the run lengths of the common read work from registers --
    // the walk is a chain of dependent look-ups otherwise (4 waves per SIMD, 62 % of their cycles waiting; PMC).  The four
    // slots are read past the read's last call when it has fewer (clamped to the staged / batch range): every use checks t < n.
    const uint32_t k_last = src.k_end - 1u;
    uint32_t o_cur = 0, o_nxt = 0, mq_nxt = 0;
    int32_t p4n[4] = {0, 0, 0, 0};
    uint32_t m4n[4] = {0, 0, 0, 0};
    if (lo < hi) {
        o_cur = src.off(lo); o_nxt = src.off(lo + 1); mq_nxt = src.mq(lo);
#pragma unroll
        for (int t = 0; t < 4; ++t) { const uint32_t kk = min(o_cur + (uint32_t)t, k_last); p4n[t] = src.pos(kk); m4n[t] = src.meth(kk); }
    }
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t o0 = o_cur, n = o_nxt - o_cur, mq_i = mq_nxt;
        int32_t p4[4];
        uint32_t m4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { p4[t] = p4n[t]; m4[t] = m4n[t]; }
        o_cur = o_nxt;
        if (i + 1 < hi) {
            o_nxt = src.off(i + 2); mq_nxt = src.mq(i + 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) { const uint32_t kk = min(o_cur + (uint32_t)t, k_last); p4n[t] = src.pos(kk); m4n[t] = src.meth(kk); }
        }
        if (n == 0) continue;                                          // no first CpG: neither flushes nor contributes
        if (c < p4[0] && seg_cov > 0) {                                // mhl.rs:163-171 (strict '<', before the filters)
            if (seg_cov >= a.min_depth) close_segment();
            seg_cov = 0; maxn = 0;
#pragma unroll
            for (int w = 0; w < LCAP / 2; ++w) { Sp[w] = 0; Dp[w] = 0; }
            if constexpr (SPILL) {
#pragma unroll
                for (int l = 0; l < LCAP; ++l) { S32[l] = 0; D32[l] = 0; }
            }
        }
        if (mq_i < a.min_qual) continue;                               // mhl.rs:176
        if (n < a.min_cpgs) continue;                                  // mhl.rs:181
        bool hit = false, decided = false;                             // does the read call c ?
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool live = !decided && (uint32_t)t < n;
            hit = hit || (live && p4[t] == c);
            decided = decided || (live && p4[t] >= c);
        }
        if (!decided)
            for (uint32_t k = o0 + 4; k < o0 + n; ++k) {
                const int32_t p = src.pos(k);
                if (p == c) { hit = true; break; }
                if (p > c) break;
            }
        if (!hit) continue;
        if (n > (uint32_t)LCAP) { overflow = true; continue; }         // deferred to the sequential variant / refused
        seg_cov += 1;                                                   // add_num_cpgs, mhl.rs:75-80
        if (seg_cov >= (1u << 20)) overflow = true;                     // D would leave the exact f32 range
        maxn = max(maxn, n);
        add_vec(Dp, n);
        uint32_t cur = 0;                                               // get_stretch_info, readutil.rs:147-164
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((uint32_t)t < n) {
                if (m4[t]) { cur += 1; }
                else if (cur) { add_vec(Sp, cur); cur = 0; }
            }
        }
        for (uint32_t k = o0 + 4; k < o0 + n; ++k) {
            if (src.meth(k)) { cur += 1; }
            else if (cur) { add_vec(Sp, cur); cur = 0; }
        }
        if (cur) add_vec(Sp, cur);
        if constexpr (SPILL) { if ((seg_cov & (MHL_SPILL - 1)) == 0) spill(); }
        else if (seg_cov >= (uint32_t)MHL_SPILL - 1u) overflow = true;   // the 16-bit counters are full: sequential variant
    }
    if (seg_cov > 0 && seg_cov >= a.min_depth) close_segment();       // mhl.rs:201-205
    if (pending) res = finalize_packed();
    if (overflow) { a.flags[j] = 2u; return; }
    a.val[j] = res;
    a.cov[j] = res_cov;
    a.flags[j] = have ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_mhl_walk_lds(const WalkArgs a) {
    __shared__ uint16_t s_call[MW_CC];
    __shared__ uint16_t s_ofs[MW_RC + 2];     // call offset of a staged read relative to the block's first call
    __shared__ uint8_t s_mq[MW_RC];
    __shared__ uint16_t s_st[MW_RC];          // read start relative to the block's base
    __shared__ uint32_t s_wv[4][8];           // per wave: lo, hi, first / last site position, has sites, call offsets at lo / hi
    __shared__ __attribute__((aligned(16))) uint32_t s_vtab[17 * 8];   // row m: max(0, m-l+1) for l = 1..16 as 16-bit pairs
    // A site whose contributors (all of them, over the whole batch: the discovery pass counted them) number fewer than
    // min_depth cannot produce a row: no segment holds more than that (mhl.rs:163-171, 201-205).  At WGBS depths that is most
    // sites (config 3: 601 k rows from several million sites).  The group still stages the candidates of its 256
    // consecutive sites -- filtering the site list instead makes the groups sparse and the staging miss (measured: config 2
    // 0.56 -> 0.75 ms) -- but only the sites that can reach min_depth are walked, packed into as few waves as they need:
    // wave w lists its walkable sites at s_l*[w * 64 ..], and thread k takes the k-th of the run.
    __shared__ uint32_t s_lj[256], s_llo[256], s_lhi[256];
    __shared__ int32_t s_lc[256];
    __shared__ uint32_t s_wcnt[4];
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    const int tid = threadIdx.x;
    if (tid < 17 * 8) {
        const int m = tid >> 3, w = tid & 7;
        s_vtab[tid] = (uint32_t)max(0, m - 2 * w) | ((uint32_t)max(0, m - 2 * w - 1) << 16);
    }
    for (uint32_t g0 = blockIdx.x * 256u; g0 < n_sites; g0 += gridDim.x * 256u) {
        const uint32_t j = g0 + (uint32_t)tid;
        const bool valid = j < n_sites;
        int32_t c = 0;
        uint32_t lo = 0, hi = 0;
        bool active = false;
        if (valid) {
            c = a.site_pos[j];
            active = a.site_nc[j] + a.site_nd[j] >= a.min_depth;
            lo = min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
            hi = min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        }
        // Sites ascend, so do lo and hi: the candidates of a run of sites are [lo of its first site, hi of its last site).
        // The group's 256 sites are staged together when their reads fit; denser or deeper stretches are staged as 2 or 4
        // runs of whole waves, one after the other (a group that misses the staging walks global memory in
        // k_mhl_walk_big -- a few such groups per batch used to be a third of this kernel's time).
        const int wv = tid >> 6;
        if ((tid & 63) == 0) {
            s_wv[wv][0] = valid ? lo : 0u; s_wv[wv][2] = (uint32_t)c; s_wv[wv][4] = valid ? 1u : 0u;
            s_wv[wv][5] = valid ? a.cpg_off[lo] : 0u;
        }
        if (valid && (j + 1 == n_sites || (tid & 63) == 63)) { s_wv[wv][1] = hi; s_wv[wv][3] = (uint32_t)c; s_wv[wv][6] = a.cpg_off[hi]; }
        __syncthreads();
        int nw = 0;                                     // waves of the group that hold sites (a prefix)
        while (nw < 4 && s_wv[nw][4]) ++nw;
        nw = __builtin_amdgcn_readfirstlane(nw);
        int parts = 0;                                  // 1, 2 or 4 runs; 0: does not fit
        for (int pz = 1; pz <= MW_MAX_PARTS && !parts; pz <<= 1) {
            const int per = 4 / pz;
            bool fits = true;
            for (int q = 0; q * per < nw; ++q) {
                const int w0 = q * per, w1 = min(w0 + per, nw) - 1;
                const uint32_t rlo = s_wv[w0][0], rhi = s_wv[w1][1];
                const long long sp = (long long)(int32_t)s_wv[w1][3] - (long long)(int32_t)s_wv[w0][2] + 2LL * a.max_span + 2LL * (IDX_Q + 1) + 1;
                fits = fits && rhi - rlo <= (uint32_t)MW_RC && s_wv[w1][6] - s_wv[w0][5] <= (uint32_t)MW_CC && sp < 32768;
            }
            if (fits) parts = pz;
        }
        parts = __builtin_amdgcn_readfirstlane(parts);
        if (!parts) {
            if (valid) a.flags[j] = active ? 4u : 0u;   // deep data: k_mhl_walk_big walks these sites from global memory
        } else {
            const int per = 4 / parts;
            for (int q = 0; q * per < nw; ++q) {
                const int w0 = q * per, w1 = min(w0 + per, nw) - 1;
                const uint32_t blo = __builtin_amdgcn_readfirstlane(s_wv[w0][0]), bhi = __builtin_amdgcn_readfirstlane(s_wv[w1][1]);
                const int32_t c_first = (int32_t)__builtin_amdgcn_readfirstlane(s_wv[w0][2]);
                // the index hands out whole IDX_Q-bp quanta: a staged read starts in [c_first - max_span - (IDX_Q - 1), c_last + IDX_Q + 1]
                // and calls positions in [start-1, start+max_span-1] (the tile pipeline that discovered the sites has
                // checked that on every call)
                const int32_t base = c_first - a.max_span - (IDX_Q + 1);
                const uint32_t c0 = __builtin_amdgcn_readfirstlane(s_wv[w0][5]);
                const uint32_t ncall = __builtin_amdgcn_readfirstlane(s_wv[w1][6]) - c0;
                if (q) __syncthreads();                 // the previous run's walks are done with the buffers
                {   // the run's walkable sites, listed per wave
                    const bool in_run = valid && wv >= w0 && wv <= w1;
                    const bool mine = in_run && active;
                    const unsigned long long bal = __ballot(mine);
                    const int ln = tid & 63;
                    if (mine) {
                        const uint32_t e = (uint32_t)wv * 64u + (uint32_t)__builtin_popcountll(bal & ((1ull << ln) - 1ull));
                        s_lj[e] = j; s_lc[e] = c; s_llo[e] = lo; s_lhi[e] = hi;
                    }
                    if (ln == 0) s_wcnt[wv] = (uint32_t)__builtin_popcountll(bal);
                    if (in_run && !active) a.flags[j] = 0u;
                }
                // Staging, eight elements per thread requested before the first is stored: written as plain loops these were one
                // load -> wait -> LDS store per trip (ISA checked), ~23 dependent round trips per group -- a quarter of the kernel.
                {
                    const uint32_t nr = bhi - blo;
                    for (uint32_t k0 = tid; k0 <= nr; k0 += 256 * 8) {
                        uint32_t x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = (k0 + 256u * u <= nr) ? a.cpg_off[blo + k0 + 256u * u] : 0u;
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (k0 + 256u * u <= nr) s_ofs[k0 + 256u * u] = (uint16_t)(x[u] - c0);
                    }
                    for (uint32_t k0 = tid; k0 < nr; k0 += 256 * 4) {
                        uint32_t q[4]; int32_t st4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool in = k0 + 256u * u < nr;
                            q[u] = in ? (uint32_t)a.read_mapq[blo + k0 + 256u * u] : 0u;
                            st4[u] = in ? a.read_start[blo + k0 + 256u * u] : 0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (k0 + 256u * u < nr) {
                                s_mq[k0 + 256u * u] = (uint8_t)q[u];
                                s_st[k0 + 256u * u] = (uint16_t)((uint32_t)(st4[u] - base) & 0x7fffu);
                            }
                    }
                    for (uint32_t w0 = tid; w0 < ncall; w0 += 256 * 8) {
                        uint32_t x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = (w0 + 256u * u < ncall) ? a.cpg_pos[c0 + w0 + 256u * u] : 0u;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (w0 + 256u * u < ncall)
                                s_call[w0 + 256u * u] = (uint16_t)((((x[u] & 0x7fffffffu) - (uint32_t)base) & 0x7fffu) | ((x[u] >> 31) << 15));
                    }
                }
                __syncthreads();
                uint32_t e = 0xffffffffu;                // this thread's entry of the run's list, if any
                {
                    uint32_t k = (uint32_t)tid;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t cw = s_wcnt[w];
                        if (e == 0xffffffffu) { if (k < cw) e = (uint32_t)w * 64u + k; else k -= cw; }
                    }
                }
                if (e != 0xffffffffu) {
                    const uint32_t j = s_lj[e], lo = s_llo[e], hi = s_lhi[e];
                    const int32_t c = s_lc[e];
                    // The index hands out whole quanta (32 bp; 256 bp and ~110 candidates when this was written); only reads starting in
                    // [c - max_span + 1, c + 1] matter (~35): an earlier read cannot call c and comes before every
                    // contributor (its flush finds nothing open), and the first later read can only flush what the
                    // end-of-range flush below finalises identically.  Two binary searches over the staged starts.
                    const uint32_t s_lo = (uint32_t)(c - a.max_span + 1 - base), s_hi = (uint32_t)(c + 1 - base);
                    uint32_t x0 = lo - blo, x1 = hi - blo;
                    while (x0 < x1) { const uint32_t m = (x0 + x1) >> 1; if (s_st[m] < s_lo) x0 = m + 1; else x1 = m; }
                    const uint32_t e_lo = x0;
                    x1 = hi - blo;
                    while (x0 < x1) { const uint32_t m = (x0 + x1) >> 1; if (s_st[m] <= s_hi) x0 = m + 1; else x1 = m; }
                    mhl_walk_site<MhlLdsSrc, false>(a, MhlLdsSrc{s_ofs, s_call, s_mq, blo, base, max(ncall, 1u)},
                                                    reinterpret_cast<const uint4 *>(s_vtab), j, c, blo + e_lo, blo + x0);
                }
            }
        }
        __syncthreads();   // the LDS buffers and s_wv are rewritten by the next group
    }
}

// The sequential per-site walk (mhl.rs:155-205 literally) over caller-provided histograms of lcap entries: S[l-1] = sum over
// the covering reads of count_l (mhl.rs:36-41), D[l-1] = sum over the covering reads with n_r >= l of (n_r - l + 1) as f32
// (mhl.rs:53-58).  Returns true if a covering read has more than lcap CpGs (the caller redoes the site with larger arrays).
__device__ __forceinline__ bool mhl_seq_walk(const WalkArgs &a, const int32_t c, const uint32_t lo, const uint32_t hi, uint32_t *S, float *D,
                                             const uint32_t lcap, float &res, uint32_t &res_cov, bool &have) {
    // (only the first maxn entries of a segment's histograms are ever touched: they are zeroed when a longer read arrives)
    uint32_t seg_cov = 0, maxn = 0;
    bool overflow = false;
    auto finalize = [&]() {   // compute_mhl, mhl.rs:43-73
        float l_sum = 0.0f;
        for (uint32_t l = 1; l < maxn + 1; ++l) l_sum = l_sum + (float)l;
        float mhl = 0.0f;
        for (uint32_t l = 1; l <= min(maxn, lcap); ++l)
            if (S[l - 1] > 0) { const float t = ((float)l * (float)S[l - 1]) / D[l - 1]; mhl = mhl + t; }
        return mhl / l_sum;
    };
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
        const uint32_t n = o1 - o0;
        if (n == 0) continue;                                          // no first CpG: neither flushes nor contributes
        const int32_t first = (int32_t)(a.cpg_pos[o0] & 0x7fffffffu);
        if (c < first && seg_cov > 0) {                                // mhl.rs:163-171 (strict '<', before the filters)
            if (seg_cov >= a.min_depth) { res = finalize(); res_cov = seg_cov; have = true; }
            seg_cov = 0; maxn = 0;
        }
        if (a.read_mapq[i] < a.min_qual) continue;                     // mhl.rs:176
        if (n < a.min_cpgs) continue;                                  // mhl.rs:181
        bool hit = false;                                              // does the read call c ?
        for (uint32_t k = o0; k < o1; ++k) {
            const int32_t p = (int32_t)(a.cpg_pos[k] & 0x7fffffffu);
            if (p == c) { hit = true; break; }
            if (p > c) break;
        }
        if (!hit) continue;
        if (n > lcap) { overflow = true; continue; }         // refused below
        seg_cov += 1;                                                   // add_num_cpgs, mhl.rs:75-80
        for (uint32_t l = maxn; l < n; ++l) { S[l] = 0; D[l] = 0.0f; }
        maxn = max(maxn, n);
        for (uint32_t l = 1; l <= n; ++l) D[l - 1] = D[l - 1] + (float)(n - l + 1);
        uint32_t cur = 0;                                               // get_stretch_info, readutil.rs:147-164
        for (uint32_t k = o0; k < o1; ++k) {
            if (a.cpg_pos[k] >> 31) {
                cur += 1;
                for (uint32_t l = 1; l <= cur; ++l) S[l - 1] += 1u;
            } else {
                cur = 0;
            }
        }
    }
    if (seg_cov > 0 && seg_cov >= a.min_depth) { res = finalize(); res_cov = seg_cov; have = true; }   // mhl.rs:201-205
    return overflow;
}

// What k_mhl_walk_lds left: (flags == 4) the sites of groups whose candidate reads did not fit the LDS staging -- deep data --
// take the same per-site walk straight from global memory, with the 32-bit spill accumulators (exact at any depth); (flags == 2)
// a covering read with more than 16 CpGs, or a segment too deep for the 16-bit counters / for the f32 denominators to stay
// exact integers: the sequential variant, one thread per site, dynamically indexed arrays (scratch) -- rare sites only.
template <int LCAP>
__global__ __launch_bounds__(256) void k_mhl_walk_big(const WalkArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_vtab[17 * 8];   // as in k_mhl_walk_lds
    if (threadIdx.x < 17 * 8) {
        const int m = threadIdx.x >> 3, w = threadIdx.x & 7;
        s_vtab[threadIdx.x] = (uint32_t)max(0, m - 2 * w) | ((uint32_t)max(0, m - 2 * w - 1) << 16);
    }
    __syncthreads();
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n_sites; j += gridDim.x * 256) {
        if (!(a.flags[j] & 6u)) continue;                  // only what the fast kernel deferred
        const int32_t c = a.site_pos[j];
        const uint32_t lo = min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        if (a.flags[j] & 4u) {
            mhl_walk_site<MhlGlobalSrc, true>(a, MhlGlobalSrc{a.cpg_off, a.cpg_pos, a.read_mapq, max(a.cpg_off[hi], 1u)},
                                              reinterpret_cast<const uint4 *>(s_vtab), j, c, lo, hi);
            if (!(a.flags[j] & 2u)) continue;              // done unless that walk deferred the site in turn
        }
        uint32_t S[LCAP];      // S[l-1] = sum over covering reads of count_l   (mhl.rs:36-41)
        float D[LCAP];         // D[l-1] = sum over covering reads with n_r >= l of (n_r-l+1) as f32   (mhl.rs:53-58)
        float res = 0.0f;
        uint32_t res_cov = 0;
        bool have = false;
        const bool overflow = mhl_seq_walk(a, c, lo, hi, S, D, (uint32_t)LCAP, res, res_cov, have);
        if (overflow) {                                                     // a read with > LCAP CpGs covers the site:
            a.flags[j] = 8u;                                                // k_mhl_walk_huge redoes it with arrays in HBM
            atomicOr(a.huge_any, 1u);
            continue;
        }
        a.val[j] = res;
        a.cov[j] = res_cov;
        a.flags[j] = have ? 1u : 0u;
    }
}

// Sites with a covering read of more than 512 CpGs (long reads): the same walk with the two histograms in an HBM scratch
// slice per thread (lcap entries each; the reference has no limit, mhl.rs:185-192).  The launch is unconditional and ends
// at once unless k_mhl_walk_big flagged a site.  A read beyond lcap CpGs is still refused (ERRB_CAPACITY).
__global__ __launch_bounds__(64) void k_mhl_walk_huge(const WalkArgs a, uint32_t *__restrict__ S_all, float *__restrict__ D_all, const uint32_t lcap) {
    if (*a.huge_any == 0u) return;
    const uint32_t gtid = blockIdx.x * 64 + threadIdx.x, nthreads = gridDim.x * 64;
    uint32_t *S = S_all + (size_t)gtid * lcap;
    float *D = D_all + (size_t)gtid * lcap;
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    for (uint32_t j = gtid; j < n_sites; j += nthreads) {
        if (a.flags[j] != 8u) continue;
        const int32_t c = a.site_pos[j];
        const uint32_t lo = min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        float res = 0.0f;
        uint32_t res_cov = 0;
        bool have = false;
        if (mhl_seq_walk(a, c, lo, hi, S, D, lcap, res, res_cov, have)) {
            atomicOr(&a.st->err, (uint32_t)ERRB_CAPACITY);
            a.flags[j] = 0u;
            continue;
        }
        a.val[j] = res;
        a.cov[j] = res_cov;
        a.flags[j] = have ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_mhl_emit(const uint32_t *__restrict__ flags, const int32_t *__restrict__ site_pos,
                                                  const float *__restrict__ val, const uint32_t *__restrict__ cov,
                                                  const DevState *__restrict__ sites_st, const uint32_t *__restrict__ blk,
                                                  const unsigned long long *__restrict__ base,
                                                  int32_t *__restrict__ out_pos, float *__restrict__ out_val,
                                                  uint32_t *__restrict__ out_cov) {
    emit_block(flags, sites_st->n_sites, *base, blk, [&](unsigned long long e, unsigned long long o) {
        out_pos[o] = site_pos[e]; out_val[o] = val[e]; out_cov[o] = cov[e];
    });
}

// ---- PDR with the exact flush / re-open semantics (pdr.rs:139-210) ------------------------------
// Used when a read spans more than the 150-bp flush margin, where plain per-site counting (the
// fused tile kernel) is no longer equivalent to the stream.  Only PASSING reads (n_cpgs >= min_cpgs,
// mapq >= min_qual, >= 1 CpG) flush, and a site c is flushed when c + 150 < first_cpg (pdr.rs:162).
struct PdrWalkArgs {
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const uint32_t *idx;
    const DevState *sites_st;
    const int32_t  *site_pos;
    float    *pdr;
    uint32_t *nc, *nd, *flags;
    int32_t idx_base, max_span;
    uint32_t n_reads, min_depth, min_cpgs;
    uint8_t min_qual;
};

__global__ __launch_bounds__(256) void k_pdr_walk(const PdrWalkArgs a) {
    const uint32_t n_sites = (uint32_t)a.sites_st->n_sites;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n_sites; j += gridDim.x * 256) {
        const int32_t c = a.site_pos[j];
        const uint32_t lo = min(a.idx[(uint32_t)(c - a.max_span + 1 - a.idx_base) >> IDX_QSHIFT], a.n_reads);
        const uint32_t hi = min(a.idx[((uint32_t)(c + 1 - a.idx_base) >> IDX_QSHIFT) + 1], a.n_reads);
        uint32_t sc = 0, sd = 0, rc = 0, rd = 0;
        bool have = false;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t o0 = a.cpg_off[i], o1 = a.cpg_off[i + 1];
            const uint32_t n = o1 - o0;
            if (n < a.min_cpgs || n == 0) continue;                       // pdr.rs:147, 155
            if (a.read_mapq[i] < a.min_qual) continue;                    // pdr.rs:150
            const uint32_t w0 = a.cpg_pos[o0];
            const int32_t first = (int32_t)(w0 & 0x7fffffffu);
            if (c + PDR_FLUSH_MARGIN < first && (sc + sd) > 0) {          // pdr.rs:160-177 is_before(first,150)
                if (sc + sd >= a.min_depth) { rc = sc; rd = sd; have = true; }
                sc = 0; sd = 0;
            }
            bool hit = false;
            uint32_t disc = 0;                                            // readutil.rs:134-145
            for (uint32_t k = o0; k < o1; ++k) {
                const uint32_t w = a.cpg_pos[k];
                hit |= (int32_t)(w & 0x7fffffffu) == c;
                disc |= (w ^ w0) >> 31;
            }
            if (!hit) continue;
            if (disc) sd += 1; else sc += 1;                               // pdr.rs:180-191
        }
        if ((sc + sd) > 0 && sc + sd >= a.min_depth) { rc = sc; rd = sd; have = true; }   // pdr.rs:199-210
        a.nc[j] = rc; a.nd[j] = rd;
        a.pdr[j] = (float)rd / ((float)rc + (float)rd);                   // pdr.rs:47-49
        a.flags[j] = have ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_pdr_walk_emit(const uint32_t *__restrict__ flags, const int32_t *__restrict__ site_pos,
                                                       const float *__restrict__ pdr, const uint32_t *__restrict__ nc,
                                                       const uint32_t *__restrict__ nd, const DevState *__restrict__ sites_st,
                                                       const uint32_t *__restrict__ blk, DevState *__restrict__ st,
                                                       int32_t *__restrict__ out_pos, float *__restrict__ out_pdr,
                                                       uint32_t *__restrict__ out_nc, uint32_t *__restrict__ out_nd) {
    emit_block(flags, sites_st->n_sites, st->cur_base, blk, [&](unsigned long long e, unsigned long long o) {
        out_pos[o] = site_pos[e]; out_pdr[o] = pdr[e]; out_nc[o] = nc[e]; out_nd[o] = nd[e];
    });
}

__global__ void k_bump_batches(DevState *st) { st->n_batches += 1; }

// run the tile pipeline as a site-discovery pass: positions called by >= 1 read that passes
// (mapq >= min_qual, n_cpgs >= max(min_cpgs,1)) -> ctx->s_pos (sorted), count in ctx->d_state2
// min_cov > 1: only sites that at least min_cov such reads call.  A segment of the walks below holds a subset of those reads,
// so a site below the measure's min_depth cannot produce a row and need not be walked (at WGBS depths that is most sites).
int discover_sites(mth_ctx *ctx, const mth_batch_t &d, uint32_t min_cpgs, uint8_t min_qual, uint64_t &bound, uint32_t min_cov) {
    hipStream_t s = ctx->stream;
    const uint64_t region_len = (uint64_t)((int64_t)d.region_end - d.region_beg);
    bound = d.n_cpgs < region_len ? d.n_cpgs : region_len;
    if (!ctx->d_state2) MTH_HIP(ctx, hipMalloc((void **)&ctx->d_state2, sizeof(DevState)));
    MTH_HIP(ctx, hipMemsetAsync(ctx->d_state2, 0, sizeof(DevState), s));
    MTH_HIP(ctx, ctx->s_pos.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->s_pdr.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->s_nc.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->s_nd.reserve((bound + 1) * 4, s));
    MTH_HIP(ctx, ctx->s_batch_cnt.reserve(16, s));
    mth_pdr_lpmd_params_t p;
    memset(&p, 0, sizeof p);
    p.pdr_min_depth = min_cov; p.pdr_min_cpgs = min_cpgs; p.pdr_min_qual = min_qual; p.want_pdr = 1;
    TileSink sink{ctx->d_state2, ctx->s_pos.as<int32_t>(), ctx->s_pdr.as<float>(), ctx->s_nc.as<uint32_t>(),
                  ctx->s_nd.as<uint32_t>(), ctx->s_batch_cnt.as<uint32_t>()};
    return launch_pdr_lpmd(ctx, d, p, &sink);
}

// exact PDR of one batch appended to the ctx's PDR result columns (capacity reserved by the caller)
int launch_pdr_exact(mth_ctx *ctx, const mth_batch_t &d, const mth_pdr_lpmd_params_t &p) {
    hipStream_t s = ctx->stream;
    uint64_t bound = 0;
    int rc = discover_sites(ctx, d, p.pdr_min_cpgs, p.pdr_min_qual, bound, p.pdr_min_depth);
    if (rc) return rc;
    if (bound == 0) bound = 1;
    MTH_HIP(ctx, ctx->w_val.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_cov.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_aux.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_flags.reserve(bound * 4, s));
    const int32_t ext = ((d.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    PdrWalkArgs a;
    a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos; a.idx = idx_ptr(ctx);
    a.sites_st = ctx->d_state2; a.site_pos = ctx->s_pos.as<int32_t>();
    a.pdr = ctx->w_val.as<float>(); a.nc = ctx->w_cov.as<uint32_t>(); a.nd = ctx->w_aux.as<uint32_t>();
    a.flags = ctx->w_flags.as<uint32_t>();
    a.idx_base = d.region_beg - ext; a.max_span = d.max_span; a.n_reads = d.n_reads;
    a.min_depth = p.pdr_min_depth; a.min_cpgs = p.pdr_min_cpgs; a.min_qual = p.pdr_min_qual;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((bound + 255) / 256, 8192);
    const uint32_t nblk = (uint32_t)((bound + 256 * SCAN_PER - 1) / (256 * SCAN_PER));
    MTH_HIP(ctx, ctx->w_blk.reserve((size_t)nblk * 4, s));
    {
        LaunchTimer lt(ctx, K_PDRWALK);
        hipLaunchKernelGGL(k_pdr_walk, dim3(grid), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_flags_blockcount, dim3(std::min(nblk, SCAN_GRID_MAX)), dim3(256), 0, s, ctx->w_flags.as<uint32_t>(),
                           (const unsigned long long *)&ctx->d_state2->n_sites, ctx->w_blk.as<uint32_t>());
        // totals live in the main DevState: n_sites is the running total, cur_base the batch's base
        hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->w_blk.as<uint32_t>(), nblk,
                           (unsigned long long *)&ctx->d_state->n_sites, (unsigned long long *)&ctx->d_state->cur_base,
                           ctx->batch_cnt.as<uint32_t>(), (uint32_t)ctx->batches.size(), (const unsigned long long *)&ctx->d_state2->n_sites);
        hipLaunchKernelGGL(k_pdr_walk_emit, dim3(std::min(nblk, SCAN_GRID_MAX)), dim3(256), 0, s, ctx->w_flags.as<uint32_t>(), ctx->s_pos.as<int32_t>(),
                           ctx->w_val.as<float>(), ctx->w_cov.as<uint32_t>(), ctx->w_aux.as<uint32_t>(), ctx->d_state2,
                           ctx->w_blk.as<uint32_t>(), ctx->d_state, ctx->out_pos.as<int32_t>(), ctx->out_pdr.as<float>(),
                           ctx->out_nc.as<uint32_t>(), ctx->out_nd.as<uint32_t>());
        hipLaunchKernelGGL(k_bump_batches, dim3(1), dim3(1), 0, s, ctx->d_state);
    }
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

}  // namespace mth

using namespace mth;

extern "C" {

int mth_mhl_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_mhl_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    mth_batch_t d;
    int rc = stage_batch(ctx, *batch, d);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    uint64_t bound = 0;
    // Default: one tile pass computes every site whose covering reads form a single segment and leaves the others to
    // k_mhl_walk_big (mth_mhl_tile.hip).  MTH_MHL_WALK=1: round 2's form -- PDR-style site discovery, then the per-site walk
    // out of LDS (k_mhl_walk_lds) -- kept for A/B and as the tests' second implementation.
    const bool walk_form = getenv("MTH_MHL_WALK") != nullptr;
    if (walk_form) {
        // (the site list is NOT filtered by min_depth here: the walk skips such sites itself and keeps its staging dense)
        if ((rc = discover_sites(ctx, d, params->min_cpgs, params->min_qual, bound))) return rc;
    } else if ((rc = launch_mhl_tile(ctx, d, *params, bound))) return rc;
    if (bound == 0) { ctx->m_batches.push_back(BatchMeta{batch->tid}); bound = 1; }
    else ctx->m_batches.push_back(BatchMeta{batch->tid});
    // per-candidate-site work arrays
    MTH_HIP(ctx, ctx->w_val.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_cov.reserve(bound * 4, s));
    MTH_HIP(ctx, ctx->w_flags.reserve(bound * 4, s));
    // result rows (appended over batches)
    if (!ctx->m_state.p) {
        MTH_HIP(ctx, ctx->m_state.reserve(4 * sizeof(unsigned long long), s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->m_state.p, 0, 4 * sizeof(unsigned long long), s));
    }
    const uint64_t need = ctx->m_rows_bound + bound;
    if (need > ctx->m_cap) {
        const uint64_t ncap = need + need / 4 + 1024, used = ctx->m_rows_bound;
        MTH_HIP(ctx, ctx->m_pos.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->m_val.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->m_cov.reserve(ncap * 4, s, true, used * 4));
        ctx->m_cap = ncap;
    }
    ctx->m_rows_bound = need;
    const size_t nb = ctx->m_batches.size() - 1;
    MTH_HIP(ctx, ctx->m_batch_rows.reserve((nb + 1) * 4, s, true, nb * 4));

    // the walk uses the index the discovery pass just built (same batch, same idx_base arithmetic)
    const int32_t ext = ((d.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    WalkArgs a;
    a.read_start = d.read_start; a.read_mapq = d.read_mapq; a.cpg_off = d.cpg_off; a.cpg_pos = d.cpg_pos;
    a.idx = idx_ptr(ctx); a.sites_st = ctx->d_state2; a.site_pos = ctx->s_pos.as<int32_t>();
    a.site_nc = ctx->s_nc.as<uint32_t>(); a.site_nd = ctx->s_nd.as<uint32_t>();
    a.st = ctx->d_state; a.val = ctx->w_val.as<float>(); a.cov = ctx->w_cov.as<uint32_t>(); a.flags = ctx->w_flags.as<uint32_t>();
    a.idx_base = d.region_beg - ext; a.max_span = d.max_span; a.n_reads = d.n_reads;
    a.min_depth = params->min_depth; a.min_cpgs = params->min_cpgs; a.min_qual = params->min_qual;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((bound + 255) / 256, 8192);
    // [0] = "some site needs the HBM-scratch walk" (set by k_mhl_walk_big), cleared per batch; the scratch follows
    MTH_HIP(ctx, ctx->w_huge.reserve((size_t)2048 * 16384 * 8, s));
    // (d_state2 was cleared as a whole by discover_sites a moment ago and nothing of the tile pipeline writes this word: no
    // second fill kernel for it)
    a.huge_any = &ctx->d_state2->pad_;
    if (walk_form) {
        LaunchTimer lt(ctx, K_MHLWALK);
        hipLaunchKernelGGL(k_mhl_walk_lds, dim3(grid), dim3(256), 0, s, a);
    }
    {
        LaunchTimer lt(ctx, K_MHLWALKBIG);
        hipLaunchKernelGGL((k_mhl_walk_big<512>), dim3(grid), dim3(256), 0, s, a);
        // long reads (> 512 CpGs): 2048 threads, 16384 histogram entries each in 256 MB of HBM scratch
        constexpr uint32_t HUGE_THREADS = 2048, HUGE_LCAP = 16384;
        hipLaunchKernelGGL(k_mhl_walk_huge, dim3(HUGE_THREADS / 64), dim3(64), 0, s, a, ctx->w_huge.as<uint32_t>(),
                           reinterpret_cast<float *>(ctx->w_huge.as<uint32_t>() + (size_t)HUGE_THREADS * HUGE_LCAP), HUGE_LCAP);
    }
    unsigned long long *ms = ctx->m_state.as<unsigned long long>();   // [0] total rows [1] base of the batch
    const uint32_t nblk = (uint32_t)((bound + 256 * SCAN_PER - 1) / (256 * SCAN_PER));
    MTH_HIP(ctx, ctx->w_blk.reserve((size_t)nblk * 4, s));
    {
        LaunchTimer lt(ctx, K_MHLEMIT);
        hipLaunchKernelGGL(k_flags_blockcount, dim3(std::min(nblk, SCAN_GRID_MAX)), dim3(256), 0, s, ctx->w_flags.as<uint32_t>(),
                           (const unsigned long long *)&ctx->d_state2->n_sites, ctx->w_blk.as<uint32_t>());
        hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->w_blk.as<uint32_t>(), nblk, ms, ms + 1,
                           ctx->m_batch_rows.as<uint32_t>(), (uint32_t)nb, (const unsigned long long *)&ctx->d_state2->n_sites);
        hipLaunchKernelGGL(k_mhl_emit, dim3(std::min(nblk, SCAN_GRID_MAX)), dim3(256), 0, s, ctx->w_flags.as<uint32_t>(), ctx->s_pos.as<int32_t>(),
                           ctx->w_val.as<float>(), ctx->w_cov.as<uint32_t>(), ctx->d_state2, ctx->w_blk.as<uint32_t>(),
                           ms + 1, ctx->m_pos.as<int32_t>(), ctx->m_val.as<float>(), ctx->m_cov.as<uint32_t>());
    }
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

int mth_mhl_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos, float *mhl, uint32_t *coverage) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    unsigned long long ms[2] = {0, 0};
    if (ctx->m_state.p) MTH_HIP(ctx, hipMemcpy(ms, ctx->m_state.p, sizeof ms, hipMemcpyDeviceToHost));
    const uint64_t n = ms[0];
    if (n_rows) *n_rows = n;
    if (n == 0) return MTH_OK;
    if (pos) MTH_HIP(ctx, hipMemcpy(pos, ctx->m_pos.p, n * 4, hipMemcpyDeviceToHost));
    if (mhl) MTH_HIP(ctx, hipMemcpy(mhl, ctx->m_val.p, n * 4, hipMemcpyDeviceToHost));
    if (coverage) MTH_HIP(ctx, hipMemcpy(coverage, ctx->m_cov.p, n * 4, hipMemcpyDeviceToHost));
    const bool grouped = (tid || pos) && has_group_batch(ctx, 1);          // rows of contig groups: back under their own contig
    std::vector<int32_t> tmp;
    if (grouped && !(tid && pos)) {
        tmp.resize(n);
        if (!pos) MTH_HIP(ctx, hipMemcpy(tmp.data(), ctx->m_pos.p, n * 4, hipMemcpyDeviceToHost));
    }
    int32_t *tid_w = tid ? tid : (grouped ? tmp.data() : nullptr), *pos_w = pos ? pos : (grouped ? tmp.data() : nullptr);
    if (tid_w) {
        std::vector<uint32_t> rows(ctx->m_batches.size());
        if (!rows.empty()) MTH_HIP(ctx, hipMemcpy(rows.data(), ctx->m_batch_rows.p, rows.size() * 4, hipMemcpyDeviceToHost));
        uint64_t o = 0;
        for (size_t b = 0; b < rows.size(); ++b)
            for (uint32_t j = 0; j < rows[b]; ++j) tid_w[o++] = ctx->m_batches[b].tid;
    }
    if (grouped) return ungroup_rows(ctx, n, tid_w, pos_w, 1, 1, nullptr);
    return MTH_OK;
}

}  // extern "C"
