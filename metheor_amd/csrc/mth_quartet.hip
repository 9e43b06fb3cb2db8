// mth_quartet.hip -- ME / PM: per-quartet 16-bin epiallele histograms on gfx950.
//
// Reference: readutil.rs:97-132 (get_cpg_quartets_and_patterns: every window of 4 consecutive CpGs of
// a read, pattern = 8*m0 + 4*m1 + 2*m2 + m3), me.rs:90-132 / pm.rs:85-128 (global
// HashMap<Quartet,[u32;16]> over reads with mapq >= min_qual, never flushed), me.rs:42-55
// (me = -0.25 * sum_{c>0} p*log2(p)), pm.rs:42-51 (pm = 1 - sum p*p), p = c as f32 / total as f32,
// bins visited 0..15 in order; min_depth is applied when rows are written (me.rs:82).
//
// Device design: at WGBS CpG density a 150-bp read yields ~0.35 quartets, so the update stream is
// small and sparse: a global open-addressing table (64-bit key CAS, then one L2 atomic add on the
// bin) sized from an exact counting pre-pass.  Key = p1 (31 bits) | three position deltas (11 bits
// each): consecutive CpGs of one read further than 2047 bp apart are refused (MTH_ERR_CAPACITY).
// A quartet is owned by the batch whose region contains p1 (same rule as sites), so region / contig
// sharding needs no exchange.  Rows are emitted in slot order (deterministic; the reference's order
// is HashMap-random, compare as sets).
#include "mth_ctx.h"
#include "mth_scan.h"

namespace mth {

constexpr unsigned long long QKEY_EMPTY = ~0ull;

__device__ __forceinline__ unsigned long long qhash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// upper bound of the number of (read, window) updates: sum over passing reads of max(0, n - 3)
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
                                                        unsigned long long *__restrict__ overflow, DevState *__restrict__ st) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_reads) return;
    const uint32_t o0 = cpg_off[i], o1 = cpg_off[i + 1];
    if (o1 - o0 < 4 || mapq[i] < min_qual) return;          // readutil.rs:101, me.rs:115
    uint32_t a = cpg_pos[o0], b = cpg_pos[o0 + 1], c = cpg_pos[o0 + 2];
    for (uint32_t k = o0 + 3; k < o1; ++k) {                // readutil.rs:105-129
        const uint32_t d = cpg_pos[k];
        const int32_t p1 = (int32_t)(a & 0x7fffffffu);
        if (p1 >= region_beg && p1 < region_end) {
            const uint32_t d2 = (b & 0x7fffffffu) - (a & 0x7fffffffu), d3 = (c & 0x7fffffffu) - (b & 0x7fffffffu),
                           d4 = (d & 0x7fffffffu) - (c & 0x7fffffffu);
            const unsigned long long key = ((unsigned long long)(uint32_t)p1 << 33) | ((unsigned long long)d2 << 22) |
                                           ((unsigned long long)d3 << 11) | (unsigned long long)d4;
            if (d2 - 1u >= 2047u || d3 - 1u >= 2047u || d4 - 1u >= 2047u || key == QKEY_EMPTY) {
                atomicOr(&st->err, (uint32_t)ERRB_CAPACITY);
            } else {
                const uint32_t pat = ((a >> 31) << 3) | ((b >> 31) << 2) | ((c >> 31) << 1) | (d >> 31);
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
        const int32_t p1 = (int32_t)(key >> 33);
        const int32_t p2 = p1 + (int32_t)((key >> 22) & 2047u), p3 = p2 + (int32_t)((key >> 11) & 2047u),
                      p4 = p3 + (int32_t)(key & 2047u);
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

}  // namespace mth

using namespace mth;

extern "C" {

int mth_quartet_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_quartet_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    mth_batch_t d;
    int rc = stage_batch(ctx, *batch, d);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    if (!ctx->q_state.p) {
        MTH_HIP(ctx, ctx->q_state.reserve(4 * sizeof(unsigned long long), s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->q_state.p, 0, 4 * sizeof(unsigned long long), s));
    }
    unsigned long long *qs = ctx->q_state.as<unsigned long long>();   // [0] bound [1] total rows [2] base of the batch
    MTH_HIP(ctx, hipMemsetAsync(qs, 0, sizeof(unsigned long long), s));
    if (d.n_reads) {
        LaunchTimer lt(ctx, K_QBOUND);
        hipLaunchKernelGGL(k_quartet_bound, dim3(1024), dim3(256), 0, s, d.cpg_off, d.read_mapq, d.n_reads,
                           params->min_qual, qs);
    }
    unsigned long long bound = 0;
    MTH_HIP(ctx, hipMemcpyAsync(&bound, qs, sizeof bound, hipMemcpyDeviceToHost, s));
    MTH_HIP(ctx, hipStreamSynchronize(s));                            // table is sized exactly: one sync per batch
    // Table size: `bound` counts quartet INSTANCES; at sequencing depth D there are ~D instances per distinct quartet, and
    // the table is cleared and scanned once per batch (72 B per slot), so it starts at bound / 2 slots (load <= 50 % as
    // soon as D >= 4) and is redone 4x larger if an insert ran out of probes -- at 2 x bound slots that cannot happen.
    unsigned long long n_slots = 1024;
    while (n_slots < bound / 2) n_slots <<= 1;
    if (const char *e = getenv("MTH_QUARTET_SLOTS_MIN")) { const unsigned long long k = strtoull(e, nullptr, 10); if (k >= 16) { n_slots = 16; while (n_slots < k) n_slots <<= 1; } }   // tests: force the retry
    for (;;) {
        MTH_HIP(ctx, ctx->q_keys.reserve(n_slots * 8, s));
        MTH_HIP(ctx, ctx->q_hist.reserve(n_slots * 64, s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->q_keys.p, 0xFF, n_slots * 8, s));
        MTH_HIP(ctx, hipMemsetAsync(ctx->q_hist.p, 0, n_slots * 64, s));
        MTH_HIP(ctx, hipMemsetAsync(qs + 3, 0, sizeof(unsigned long long), s));
        if (!d.n_reads) break;
        {
            LaunchTimer lt(ctx, K_QINSERT);
            hipLaunchKernelGGL(k_quartet_insert, dim3((d.n_reads + 255) / 256), dim3(256), 0, s, d.cpg_off, d.cpg_pos,
                               d.read_mapq, d.n_reads, params->min_qual, d.region_beg, d.region_end,
                               ctx->q_keys.as<unsigned long long>(), ctx->q_hist.as<uint32_t>(), n_slots - 1, qs + 3, ctx->d_state);
        }
        unsigned long long ovf = 0;
        MTH_HIP(ctx, hipMemcpyAsync(&ovf, qs + 3, sizeof ovf, hipMemcpyDeviceToHost, s));
        MTH_HIP(ctx, hipStreamSynchronize(s));
        if (!ovf) break;
        n_slots <<= 2;
    }
    // distinct quartets <= bound: grow the row buffers (keeping earlier batches) before emitting
    const uint64_t need = ctx->q_rows_bound + bound;
    if (need > ctx->q_cap) {
        uint64_t ncap = need + need / 4 + 1024;
        const uint64_t used = ctx->q_rows_bound;   // upper bound of rows in use: copy that many
        MTH_HIP(ctx, ctx->q_pos.reserve(ncap * 16, s, true, used * 16));
        MTH_HIP(ctx, ctx->q_cnt.reserve(ncap * 64, s, true, used * 64));
        MTH_HIP(ctx, ctx->q_me.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->q_pm.reserve(ncap * 4, s, true, used * 4));
        MTH_HIP(ctx, ctx->q_depth.reserve(ncap * 4, s, true, used * 4));
        ctx->q_cap = ncap;
    }
    ctx->q_rows_bound = need;
    const uint32_t nblk = (uint32_t)((n_slots + 256 * QE_PER - 1) / (256 * QE_PER));
    MTH_HIP(ctx, ctx->q_blk.reserve((size_t)nblk * 4, s));
    MTH_HIP(ctx, ctx->q_batch_rows.reserve((ctx->q_batches.size() + 1) * 4, s, true, ctx->q_batches.size() * 4));
    {
        LaunchTimer lt(ctx, K_QEMIT);
        hipLaunchKernelGGL(k_quartet_blockcount, dim3(nblk), dim3(256), 0, s, ctx->q_keys.as<unsigned long long>(),
                           n_slots, ctx->q_blk.as<uint32_t>());
        hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, s, ctx->q_blk.as<uint32_t>(), nblk, qs + 1, qs + 2,
                           ctx->q_batch_rows.as<uint32_t>(), (uint32_t)ctx->q_batches.size());
        hipLaunchKernelGGL(k_quartet_emit, dim3(nblk), dim3(256), 0, s, ctx->q_keys.as<unsigned long long>(),
                           ctx->q_hist.as<uint32_t>(), n_slots, ctx->q_blk.as<uint32_t>(), qs + 2,
                           ctx->q_pos.as<int32_t>(), ctx->q_cnt.as<uint32_t>(), ctx->q_me.as<float>(),
                           ctx->q_pm.as<float>(), ctx->q_depth.as<uint32_t>());
    }
    MTH_HIP(ctx, hipGetLastError());
    ctx->q_batches.push_back(BatchMeta{batch->tid});
    return MTH_OK;
}

// rows of all batches with depth >= min_depth (me.rs:82 / pm.rs:77); any pointer may be NULL.
// Call with every pointer NULL to get the count.
int mth_quartet_fetch(mth_ctx_t *ctx, uint32_t min_depth, uint64_t *n_rows, int32_t *tid, int32_t *pos4,
                      uint32_t *counts16, float *me, float *pm) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    unsigned long long qs[3] = {0, 0, 0};
    if (ctx->q_state.p) MTH_HIP(ctx, hipMemcpy(qs, ctx->q_state.p, sizeof qs, hipMemcpyDeviceToHost));
    const uint64_t total = qs[1];
    std::vector<uint32_t> depth(total), rows(ctx->q_batches.size());
    if (total) MTH_HIP(ctx, hipMemcpy(depth.data(), ctx->q_depth.p, total * 4, hipMemcpyDeviceToHost));
    if (!rows.empty()) MTH_HIP(ctx, hipMemcpy(rows.data(), ctx->q_batch_rows.p, rows.size() * 4, hipMemcpyDeviceToHost));
    uint64_t n = 0;
    for (uint64_t i = 0; i < total; ++i) n += depth[i] >= min_depth ? 1 : 0;
    if (n_rows) *n_rows = n;
    if (!tid && !pos4 && !counts16 && !me && !pm) return MTH_OK;
    std::vector<int32_t> hp(pos4 ? total * 4 : 0);
    std::vector<uint32_t> hc(counts16 ? total * 16 : 0);
    std::vector<float> hme(me ? total : 0), hpm(pm ? total : 0);
    if (total) {
        if (pos4) MTH_HIP(ctx, hipMemcpy(hp.data(), ctx->q_pos.p, total * 16, hipMemcpyDeviceToHost));
        if (counts16) MTH_HIP(ctx, hipMemcpy(hc.data(), ctx->q_cnt.p, total * 64, hipMemcpyDeviceToHost));
        if (me) MTH_HIP(ctx, hipMemcpy(hme.data(), ctx->q_me.p, total * 4, hipMemcpyDeviceToHost));
        if (pm) MTH_HIP(ctx, hipMemcpy(hpm.data(), ctx->q_pm.p, total * 4, hipMemcpyDeviceToHost));
    }
    uint64_t o = 0, i = 0;
    for (size_t b = 0; b < rows.size(); ++b) {
        for (uint32_t j = 0; j < rows[b]; ++j, ++i) {
            if (depth[i] < min_depth) continue;
            if (tid) tid[o] = ctx->q_batches[b].tid;
            if (pos4) memcpy(pos4 + o * 4, hp.data() + i * 4, 16);
            if (counts16) memcpy(counts16 + o * 16, hc.data() + i * 16, 64);
            if (me) me[o] = hme[i];
            if (pm) pm[o] = hpm[i];
            ++o;
        }
    }
    return MTH_OK;
}

}  // extern "C"
