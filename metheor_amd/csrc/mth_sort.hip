// mth_sort.hip -- the decoded stream re-ordered by (tid, start) on the device (gfx950).
//
// LPMD, ME and PM do not depend on the order of the records (lpmd.rs:175-200, me.rs:106-125, pm.rs:101-121 iterate the file into
// maps that are only read at the end), and the reference accepts any BAM.  The tile kernels need a contig's reads sorted by start,
// so for those measures an input that is not coordinate-sorted is sorted HERE, after the device decode and before batching:
// 64-bit keys tid << 32 | start with the read index as payload (rocPRIM's radix sort through hipCUB: a library primitive, stable),
// the per-read columns gathered through the permutation, the call counts scanned into new CSR offsets, every read's calls copied to
// their new place.  The flush-order-dependent measures (PDR, MHL, FDRP, qFDRP) never come here: the CLI refuses unsorted input
// for them and says why.
#include <hipcub/hipcub.hpp>

#include "mth_ctx.h"

namespace mth {

__global__ __launch_bounds__(256) void k_sort_keys(const int32_t *__restrict__ tid, const int32_t *__restrict__ start, uint32_t n,
                                                   unsigned long long *__restrict__ keys, uint32_t *__restrict__ idx) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((unsigned long long)(uint32_t)tid[i] << 32) | (uint32_t)start[i];      // (tid, start >= 0: checked by the caller)
    idx[i] = i;
}

__global__ __launch_bounds__(256) void k_sort_gather_reads(const uint32_t *__restrict__ perm, uint32_t n,
                                                           const int32_t *__restrict__ tid, const int32_t *__restrict__ start,
                                                           const int32_t *__restrict__ end, const uint8_t *__restrict__ mapq,
                                                           const uint8_t *__restrict__ fwd, const unsigned long long *__restrict__ off,
                                                           int32_t *__restrict__ o_tid, int32_t *__restrict__ o_start, int32_t *__restrict__ o_end,
                                                           uint8_t *__restrict__ o_mapq, uint8_t *__restrict__ o_fwd, uint32_t *__restrict__ o_n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    o_tid[i] = tid[p]; o_start[i] = start[p]; o_end[i] = end[p]; o_mapq[i] = mapq[p]; o_fwd[i] = fwd[p];
    o_n[i] = (uint32_t)(off[p + 1] - off[p]);
}

__global__ __launch_bounds__(256) void k_sort_gather_calls(const uint32_t *__restrict__ perm, uint32_t n,
                                                           const unsigned long long *__restrict__ old_off, const unsigned long long *__restrict__ new_off,
                                                           const uint32_t *__restrict__ pos, const uint16_t *__restrict__ rel,
                                                           uint32_t *__restrict__ o_pos, uint16_t *__restrict__ o_rel) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    const unsigned long long s0 = old_off[p], s1 = old_off[p + 1], d0 = new_off[i];
    for (unsigned long long k = s0; k < s1; ++k) { o_pos[d0 + (k - s0)] = pos[k]; o_rel[d0 + (k - s0)] = rel[k]; }
}

}  // namespace mth

using namespace mth;

extern "C" int mth_decoded_sort(mth_ctx_t *ctx) {
    if (!ctx) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const uint64_t R = ctx->dec_reads, Cn = ctx->dec_cpgs;
    // keys are tid << 32 | start as unsigned numbers: a record without a contig (tid < 0) or without an aligned base (start < 0) would
    // sort last, silently, into batches the tile kernels do not expect -- the caller must have looked (mth_decoded_contigs) and
    // take the host path for such files
    if (ctx->dec_contig_flags & 3u) return fail(ctx, MTH_ERR_STATE, "mth_decoded_sort: the decoded stream holds records without a contig or without an aligned base (mth_decoded_contigs flags bit 0 / 1)");
    if (R < 2) return MTH_OK;
    if (R >= (1ull << 31)) return fail(ctx, MTH_ERR_CAPACITY, "more than 2^31 records in one sort");
    const uint32_t n = (uint32_t)R;
    const uint32_t nb = (n + 255) / 256;
    DevBuf keys_in, keys_out, idx_in, perm, tmp, n_tid, n_start, n_end, n_mapq, n_fwd, n_cnt, n_off, n_pos, n_rel;
    auto drop = [&]() { for (DevBuf *b : {&keys_in, &keys_out, &idx_in, &perm, &tmp, &n_tid, &n_start, &n_end, &n_mapq, &n_fwd, &n_cnt, &n_off, &n_pos, &n_rel}) b->release(); };
#define SORT_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { drop(); return fail(ctx, MTH_ERR_HIP, #call, e__); } } while (0)
    SORT_HIP(keys_in.reserve((size_t)n * 8, s)); SORT_HIP(keys_out.reserve((size_t)n * 8, s));
    SORT_HIP(idx_in.reserve((size_t)n * 4, s)); SORT_HIP(perm.reserve((size_t)n * 4, s));
    hipLaunchKernelGGL(k_sort_keys, dim3(nb), dim3(256), 0, s, ctx->dec_tid.as<int32_t>(), ctx->dec_start.as<int32_t>(), n,
                       keys_in.as<unsigned long long>(), idx_in.as<uint32_t>());
    size_t tmp_bytes = 0;
    SORT_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                                idx_in.as<uint32_t>(), perm.as<uint32_t>(), (int)n, 0, 64, s));
    SORT_HIP(tmp.reserve(tmp_bytes + 16, s));
    SORT_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                                idx_in.as<uint32_t>(), perm.as<uint32_t>(), (int)n, 0, 64, s));
    SORT_HIP(n_tid.reserve((size_t)n * 4 + 4, s)); SORT_HIP(n_start.reserve((size_t)n * 4 + 4, s)); SORT_HIP(n_end.reserve((size_t)n * 4 + 4, s));
    SORT_HIP(n_mapq.reserve((size_t)n + 4, s)); SORT_HIP(n_fwd.reserve((size_t)n + 4, s)); SORT_HIP(n_cnt.reserve((size_t)n * 4 + 4, s));
    SORT_HIP(n_off.reserve(((size_t)n + 1) * 8, s));
    SORT_HIP(n_pos.reserve((size_t)Cn * 4 + 4, s)); SORT_HIP(n_rel.reserve((size_t)Cn * 2 + 4, s));
    hipLaunchKernelGGL(k_sort_gather_reads, dim3(nb), dim3(256), 0, s, perm.as<uint32_t>(), n, ctx->dec_tid.as<int32_t>(),
                       ctx->dec_start.as<int32_t>(), ctx->dec_end.as<int32_t>(), ctx->dec_mapq.as<uint8_t>(), ctx->dec_fwd.as<uint8_t>(),
                       ctx->dec_off.as<unsigned long long>(), n_tid.as<int32_t>(), n_start.as<int32_t>(), n_end.as<int32_t>(),
                       n_mapq.as<uint8_t>(), n_fwd.as<uint8_t>(), n_cnt.as<uint32_t>());
    unsigned long long total = 0;
    const int rc = scan_u32_to_u64(ctx, n_cnt.as<uint32_t>(), n, 0ull, n_off.as<unsigned long long>(), &total);
    if (rc) { drop(); return rc; }
    if (total != Cn) { drop(); return fail(ctx, MTH_ERR_STATE, "sort: the call counts do not add up"); }
    hipLaunchKernelGGL(k_sort_gather_calls, dim3(nb), dim3(256), 0, s, perm.as<uint32_t>(), n, ctx->dec_off.as<unsigned long long>(),
                       n_off.as<unsigned long long>(), ctx->dec_pos.as<uint32_t>(), ctx->dec_rel.as<uint16_t>(), n_pos.as<uint32_t>(),
                       n_rel.as<uint16_t>());
    SORT_HIP(hipGetLastError());
    SORT_HIP(hipStreamSynchronize(s));
#undef SORT_HIP
    std::swap(ctx->dec_tid, n_tid); std::swap(ctx->dec_start, n_start); std::swap(ctx->dec_end, n_end);
    std::swap(ctx->dec_mapq, n_mapq); std::swap(ctx->dec_fwd, n_fwd); std::swap(ctx->dec_off, n_off);
    std::swap(ctx->dec_pos, n_pos); std::swap(ctx->dec_rel, n_rel);
    drop();
    return MTH_OK;
}
