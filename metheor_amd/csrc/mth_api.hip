// mth_api.hip -- C ABI (include/metheor_hip.h) over the gfx950 kernels: context, buffers,
// staging of host batches, result getters, per-kernel event timing.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mth_ctx.h"

namespace mth {

static const char *kKernelNames[K_NUM] = {"k_build_index", "k_pdr_lpmd_tile", "k_gather",
                                          "k_quartet_bound", "k_quartet_tile", "k_quartet_insert", "k_quartet_emit",
                                          "k_mhl_walk", "k_mhl_walk_big", "k_mhl_emit", "k_pdr_walk",
                                          "k_fdrp_walk", "k_fdrp_emit", "k_pairs", "k_pairs_tile", "k_decode", "k_inflate", "k_crc32", "k_mhl_tile", "k_pdr_lpmd_wide",
                                          "k_fdrp_tile", "k_fdrp_chain", "k_fdrp_walk4", "k_fdrp_wtile", "k_mhl_rowcheck"};

__global__ void k_lpmd_add2(DevState *st, long long n_read, long long n_valid) { st->lpmd[2] += n_read; st->lpmd[3] += n_valid; }

int fail(mth_ctx *ctx, int status, const char *what, hipError_t e) {
    if (ctx) {
        ctx->last_error = what ? what : "";
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return status;
}

hipError_t DevBuf::reserve(size_t bytes, hipStream_t s, bool keep, size_t keep_bytes) {
    if (bytes <= cap) return hipSuccess;
    size_t ncap = cap ? cap : 256;
    while (ncap < bytes) ncap += ncap / 2 + 256;
    void *np = nullptr;
    // an older launch may still use the old allocation: drain the stream before replacing it
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    e = hipMalloc(&np, ncap);
    if (e != hipSuccess) return e;
    if (keep && p && keep_bytes) {
        e = hipMemcpy(np, p, keep_bytes, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(np); return e; }
    }
    if (p) (void)hipFree(p);
    p = np; cap = ncap;
    return hipSuccess;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
}

LaunchTimer::LaunchTimer(mth_ctx *c, int k) : ctx(c), kernel(k) {
    if (!ctx->timing) return;
    auto get = [&]() {
        hipEvent_t ev = nullptr;
        if (!ctx->event_pool.empty()) { ev = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else (void)hipEventCreate(&ev);
        return ev;
    };
    beg = get(); end = get();
    (void)hipEventRecord(beg, ctx->stream);
}
LaunchTimer::~LaunchTimer() {
    if (!beg) return;
    (void)hipEventRecord(end, ctx->stream);
    ctx->timed.push_back(TimedLaunch{kernel, beg, end});
}

// a mth_reset that arrived inside a pipelined run and has not been folded into a gather: do it now, on ctx->stream
static int flush_reset(mth_ctx *ctx) {
    if (!ctx->reset_pending) return MTH_OK;
    ctx->reset_pending = false;
    MTH_HIP(ctx, hipMemsetAsync(ctx->d_state, 0, sizeof(DevState), ctx->stream));
    return MTH_OK;
}

int pipe_join(mth_ctx *ctx) {
    if (ctx->pipe_active) {
        for (PdrLane &l : ctx->lane)
            if (l.used) { MTH_HIP(ctx, hipStreamWaitEvent(ctx->stream, l.done, 0)); l.used = false; }
        ctx->pipe_active = false;
        ctx->pipe_tail = -1;
    }
    ctx->pdr_streak = 0;
    return flush_reset(ctx);
}

int enter(mth_ctx *ctx) {
    MTH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = pipe_join(ctx);
    // ME / PM / pairs batches queued without a host sync (mth_quartet.hip, mth_pairs.hip): every entry point but the call that queues
    // the next one settles them first -- the caller may be about to rewrite what a replay would read (mth_decoded_batch rebuilds the
    // offsets array the previous decoded batch points into), and "the next synchronising call" of the header's contract is any of them
    if (!rc && !ctx->tile_queue_hold) {
        // (the resolves clear the "prepared batch of the call in progress" for their replays: the call that is entering keeps its own --
        // a growth sync behind stage_batch must not make it drop its prepared index and rebuild one: ADVICE r05)
        Prepared *const prep = ctx->cur_prep;
        const uint32_t *const idx = ctx->cur_idx;
        if (!ctx->q_pending.empty()) rc = quartet_resolve(ctx);
        if (!rc && !ctx->p_pending.empty()) rc = pairs_resolve(ctx);
        ctx->cur_prep = prep; ctx->cur_idx = prep ? idx : nullptr;
    }
    return rc;
}

bool has_group_batch(const mth_ctx *ctx, int which) {
    if (ctx->groups.empty()) return false;
    auto any = [](const auto &v) { for (const auto &m : v) if (m.tid <= -2) return true; return false; };
    switch (which) {
        case 0: return any(ctx->batches);
        case 1: return any(ctx->m_batches);
        case 2: return any(ctx->f_batches);
        case 3: return any(ctx->q_meta);
        default: return any(ctx->p_meta);
    }
}

const int32_t *group_table(const mth_ctx *ctx, int32_t tid, uint32_t *n) {
    *n = 0;
    if (tid > -2 || (size_t)(-2 - (int64_t)tid) >= ctx->groups.size()) return nullptr;
    const mth_ctx::ContigGroup &g = ctx->groups[(size_t)(-2 - (int64_t)tid)];
    *n = (uint32_t)g.tids.size();
    return g.d_tab.as<int32_t>();
}

int ungroup_rows(mth_ctx *ctx, uint64_t n, int32_t *tid, int32_t *pos_a, int stride, int cols, int32_t *pos_b) {
    if (ctx->groups.empty() || !tid || !pos_a) return MTH_OK;
    for (uint64_t i = 0; i < n; ++i) {
        const int32_t h = tid[i];
        if (h > -2) continue;
        const size_t gi = (size_t)(-2 - (int64_t)h);
        if (gi >= ctx->groups.size()) return fail(ctx, MTH_ERR_STATE, "a result row carries the handle of a contig group that is not defined (any more)");
        const mth_ctx::ContigGroup &g = ctx->groups[gi];
        const int64_t v = pos_a[i * (uint64_t)stride];
        const size_t k = (size_t)(std::upper_bound(g.voff.begin(), g.voff.end(), v) - g.voff.begin());
        if (k == 0) return fail(ctx, MTH_ERR_STATE, "a grouped batch's row lies before the group's first contig");
        const int32_t off = (int32_t)g.voff[k - 1];
        tid[i] = g.tids[k - 1];
        for (int c = 0; c < cols; ++c) pos_a[i * (uint64_t)stride + c] -= off;
        if (pos_b) pos_b[i] -= off;
    }
    return MTH_OK;
}

int sync_and_check(mth_ctx *ctx) {
    MTH_ENTER(ctx);   // every entry point passes through here or stage_batch: the calling thread may be new
    MTH_HIP(ctx, hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(DevState), hipMemcpyDeviceToHost, ctx->stream));
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t e = ctx->h_state->err;
    ctx->notes |= ctx->h_state->pad_;          // non-fatal findings of the device decode (sticky on the host: mth_reset clears the device word)
    if (e & ERRB_UNSORTED) return fail(ctx, MTH_ERR_UNSORTED, "reads of a batch are not sorted by start position");
    if (e & ERRB_SPAN) return fail(ctx, MTH_ERR_SPAN, "a read spans more reference bases than batch.max_span");
    if (e & ERRB_RANGE) return fail(ctx, MTH_ERR_RANGE, "CpG position outside the declared range");
    if (e & ERRB_CAPACITY) return fail(ctx, MTH_ERR_CAPACITY, "capacity exceeded (a read with more than 16384 CpGs in MHL, or FDRP max_depth above 16384)");
    if (e & ERRB_CRC) return fail(ctx, MTH_ERR_FORMAT, "corrupt BGZF block (CRC32 mismatch)");
    if (e & ERRB_FORMAT) return fail(ctx, MTH_ERR_FORMAT, "corrupt BGZF block or malformed BAM record (DEFLATE / ISIZE / block_size / field lengths inconsistent)");
    if (e & ERRB_TAGPANIC) return fail(ctx, MTH_ERR_FORMAT, "tag: a record the reference cannot tag either (unplaced or outside its contig / the FASTA, a base without a complement, or a C whose context ends in a deletion): determine_xm_tag_string panics there");
    if (e & ERRB_FDRPPANIC) return fail(ctx, MTH_ERR_FORMAT, "fdrp / qfdrp: a reverse-strand read calls a CpG at its start - 1 and another one 202 bp further while spanning at most 403 bp: the reference's window index is -1 there and it panics (fdrp.rs:70-72, index out of bounds)");
    if (e & ERRB_NOXM) return fail(ctx, MTH_ERR_FORMAT, "a record has no XM:Z tag (the reference panics: Error reading XM tag)");
    if (e & ERRB_UNALIGNED) return fail(ctx, MTH_ERR_UNALIGNED, "a BAM record straddles two BGZF blocks: the per-block device walk does not apply (use the host walk)");
    return MTH_OK;
}

// copy one array of a host batch into its staging buffer
static int stage(mth_ctx *ctx, DevBuf &buf, const void *src, size_t bytes, const void **dst) {
    if (!src || bytes == 0) { *dst = nullptr; return MTH_OK; }
    MTH_HIP(ctx, buf.reserve(bytes, ctx->stream));
    MTH_HIP(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dst = buf.p;
    return MTH_OK;
}

Prepared *prepared_lookup(const mth_ctx *ctx, const void *handle) {
    Prepared *pr = const_cast<Prepared *>(reinterpret_cast<const Prepared *>(handle));
    return (pr && ctx->prepared.count(pr)) ? pr : nullptr;
}

void prepared_free(mth_ctx *ctx, Prepared *pr) {
    ctx->prepared.erase(pr);
    if (ctx->cur_prep == pr) { ctx->cur_prep = nullptr; ctx->cur_idx = nullptr; }
    for (DevBuf &x : pr->own) x.release();
    pr->idx.release();
    if (pr->st) (void)hipFree(pr->st);
    pr->magic = 0;
    delete pr;
}

int stage_batch(mth_ctx *ctx, const mth_batch_t &b, mth_batch_t &d, bool join) {
    if (b.region_end < b.region_beg || b.max_span < 0) return fail(ctx, MTH_ERR_INVALID, "bad region / max_span");
    // a contig group's handle must be defined -- checked here, before any accumulate entry point has recorded anything of the batch
    if (b.tid <= -2 && (size_t)(-2 - (int64_t)b.tid) >= ctx->groups.size()) return fail(ctx, MTH_ERR_INVALID, "batch.tid is not a defined contig group's handle");
    ctx->cur_prep = nullptr; ctx->cur_idx = nullptr;
    if (b.mem == MTH_MEM_PREPARED) {
        // a batch made by mth_batch_prepare: device-resident, its read index built -- the handle rides in read_fwd
        // (validated by look-up in the context's registry: a released handle, a copy used after release, another context's)
        Prepared *pr = prepared_lookup(ctx, b.read_fwd);
        if (!pr) return fail(ctx, MTH_ERR_INVALID, "not a batch prepared by this context (mth_batch_prepare), or released");
        // the entry points size their outputs from the caller's struct and run the kernels on the prepared one: they must agree
        if (b.n_reads != pr->dev.n_reads || b.n_cpgs != pr->dev.n_cpgs || b.region_beg != pr->dev.region_beg || b.region_end != pr->dev.region_end ||
            b.max_span != pr->dev.max_span)
            return fail(ctx, MTH_ERR_INVALID, "a prepared batch's n_reads / n_cpgs / region / max_span were changed after mth_batch_prepare");
        if (join) MTH_ENTER(ctx); else MTH_HIP(ctx, hipSetDevice(ctx->device));
        d = pr->dev;
        d.tid = b.tid;                                   // (the caller may submit it under a contig group's handle)
        ctx->cur_prep = pr;
        return MTH_OK;
    }
    if (b.n_reads && (!b.read_start || !b.read_end || !b.read_mapq || !b.cpg_off))
        return fail(ctx, MTH_ERR_INVALID, "batch arrays missing");
    if (b.n_cpgs && (!b.cpg_pos || (!b.cpg_rel == !b.cpg_rel16)))
        return fail(ctx, MTH_ERR_INVALID, "exactly one of cpg_rel / cpg_rel16 must be given");
    if (join || b.mem == MTH_MEM_HOST) MTH_ENTER(ctx);      // (host batches: the single set of staging buffers is reused -> always joined)
    else MTH_HIP(ctx, hipSetDevice(ctx->device));
    d = b;
    if (b.mem == MTH_MEM_HOST) {
        int rc;
        const size_t nr = b.n_reads, nc = b.n_cpgs;
        if ((rc = stage(ctx, ctx->st_start, b.read_start, nr * 4, (const void **)&d.read_start))) return rc;
        if ((rc = stage(ctx, ctx->st_end, b.read_end, nr * 4, (const void **)&d.read_end))) return rc;
        if ((rc = stage(ctx, ctx->st_mapq, b.read_mapq, nr, (const void **)&d.read_mapq))) return rc;
        if ((rc = stage(ctx, ctx->st_off, b.cpg_off, (nr + 1) * 4, (const void **)&d.cpg_off))) return rc;
        if ((rc = stage(ctx, ctx->st_pos, b.cpg_pos, nc * 4, (const void **)&d.cpg_pos))) return rc;
        if (b.cpg_rel) { if ((rc = stage(ctx, ctx->st_rel, b.cpg_rel, nc, (const void **)&d.cpg_rel))) return rc; }
        else { if ((rc = stage(ctx, ctx->st_rel, b.cpg_rel16, nc * 2, (const void **)&d.cpg_rel16))) return rc; }
        d.read_fwd = nullptr;
        d.mem = MTH_MEM_DEVICE;
    } else if (b.mem != MTH_MEM_DEVICE) {
        return fail(ctx, MTH_ERR_INVALID, "batch.mem");
    }
    if (b.n_cpgs == 0 && !d.cpg_rel && !d.cpg_rel16) d.cpg_rel = reinterpret_cast<const uint8_t *>(ctx->d_state);
    return MTH_OK;
}

}  // namespace mth

using namespace mth;

extern "C" {

int mth_abi_version(void) { return MTH_ABI_VERSION; }

const char *mth_strerror(int s) {
    switch (s) {
        case MTH_OK: return "ok";
        case MTH_ERR_INVALID: return "invalid argument";
        case MTH_ERR_HIP: return "HIP runtime error";
        case MTH_ERR_NO_DEVICE: return "no gfx950 (MI355X) device available; there is no CPU fallback";
        case MTH_ERR_UNSORTED: return "batch reads are not coordinate sorted";
        case MTH_ERR_SPAN: return "read span exceeds batch.max_span";
        case MTH_ERR_REOPEN: return "input needs flush re-open semantics not implemented on this path";
        case MTH_ERR_RANGE: return "CpG position out of declared range";
        case MTH_ERR_CAPACITY: return "capacity exceeded (a read with more than 16384 CpGs in MHL, or FDRP max_depth above 16384)";
        case MTH_ERR_STATE: return "call order violated";
        case MTH_ERR_FORMAT: return "malformed BAM record or record without XM:Z";
        case MTH_ERR_UNALIGNED: return "BAM records straddle BGZF blocks";
        case MTH_ERR_RCCL: return "RCCL unavailable or an RCCL call failed";
        default: return "unknown status";
    }
}

const char *mth_last_error(const mth_ctx_t *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int mth_ctx_create(int device_id, mth_ctx_t **out) {
    if (!out) return MTH_ERR_INVALID;
    *out = nullptr;
    // METHEOR_TIMING=1: what the 70-140 ms of start-up are made of (stderr)
    const bool tm = getenv("METHEOR_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    auto lap = [&](const char *what) {
        if (!tm) return;
        const auto t1 = now();
        fprintf(stderr, "[metheor timing]     ctx/%-22s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    };
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return MTH_ERR_NO_DEVICE;
    lap("hipGetDeviceCount");
    // (hipDeviceGetAttribute-free: the architecture name is all that is needed, but only the properties call gives it)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return MTH_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MTH_ERR_NO_DEVICE;  // kernels are built for gfx950 only
    lap("hipGetDeviceProperties");
    auto *ctx = new mth_ctx;
    ctx->device = device_id;
    bool ok = hipSetDevice(device_id) == hipSuccess;
    lap("hipSetDevice");
    ok = ok && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess;
    lap("hipStreamCreate");
    ok = ok && hipMalloc((void **)&ctx->d_state, sizeof(DevState)) == hipSuccess;
    lap("hipMalloc");
    ok = ok && hipHostMalloc((void **)&ctx->h_state, sizeof(DevState) + 16 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess;
    lap("hipHostMalloc");
    if (!ok) {
        mth_ctx_destroy(ctx);
        return MTH_ERR_HIP;
    }
    ctx->h_words = reinterpret_cast<unsigned long long *>(ctx->h_state + 1);      // one pinned allocation for both
    ctx->stream = ctx->own_stream;
    if (hipMemsetAsync(ctx->d_state, 0, sizeof(DevState), ctx->stream) != hipSuccess) {
        mth_ctx_destroy(ctx);
        return MTH_ERR_HIP;
    }
    lap("first memset");
    *out = ctx;
    return MTH_OK;
}

void mth_ctx_destroy(mth_ctx_t *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stage_thread.joinable()) ctx->stage_thread.join();
    for (PdrLane &l : ctx->lane) if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    while (!ctx->prepared.empty()) prepared_free(ctx, *ctx->prepared.begin());   // prepared batches the caller did not release
    rccl_release(ctx);
    for (PdrLane &l : ctx->lane) {
        for (DevBuf *b : {&l.idx, &l.tile_cnt, &l.tile_bucket, &l.scratch}) b->release();
        if (l.st) (void)hipFree(l.st);
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    if (ctx->pipe_in) (void)hipEventDestroy(ctx->pipe_in);
    for (DevBuf *b : {&ctx->st_start, &ctx->st_end, &ctx->st_mapq, &ctx->st_fwd, &ctx->st_off, &ctx->st_pos,
                      &ctx->st_rel, &ctx->idx, &ctx->tile_cnt, &ctx->tile_bucket, &ctx->scratch, &ctx->dec_raw, &ctx->dec_recoff, &ctx->dec_tid, &ctx->dec_start, &ctx->dec_end,
                      &ctx->dec_mapq, &ctx->dec_fwd, &ctx->dec_n, &ctx->dec_off, &ctx->dec_pos, &ctx->dec_rel, &ctx->dec_blk, &ctx->dec_off32, &ctx->dec_runs, &ctx->dec_xm, &ctx->dec_filter,
                      &ctx->inf_file, &ctx->inf_file2, &ctx->inf_tab, &ctx->inf_raw, &ctx->inf_cnt, &ctx->inf_base, &ctx->inf_recoff, &ctx->crc_mat,
                      &ctx->tag_genome, &ctx->tag_goff, &ctx->tag_ncol, &ctx->tag_coloff, &ctx->tag_xmlen, &ctx->tag_cols, &ctx->tag_xm,
                      &ctx->batch_cnt, &ctx->out_pos, &ctx->out_pdr, &ctx->out_nc, &ctx->out_nd, &ctx->q_state, &ctx->q_keys,
                      &ctx->q_hist, &ctx->q_blk, &ctx->q_batch_rows, &ctx->q_tflag, &ctx->q_tile_row0, &ctx->q_tile_rows, &ctx->q_wpos, &ctx->q_wpat, &ctx->q_wk0, &ctx->q_wk1, &ctx->q_snap, &ctx->p_snap, &ctx->q_pos, &ctx->q_cnt, &ctx->q_me, &ctx->q_pm, &ctx->q_depth,
                      &ctx->s_pos, &ctx->s_pdr, &ctx->s_nc, &ctx->s_nd, &ctx->s_batch_cnt, &ctx->w_val, &ctx->w_cov, &ctx->w_aux, &ctx->w_flags,
                      &ctx->w_blk, &ctx->w_huge, &ctx->m_state, &ctx->m_pos, &ctx->m_val, &ctx->m_cov, &ctx->m_batch_rows,
                      &ctx->f_state, &ctx->f_pos, &ctx->f_val, &ctx->f_qval, &ctx->f_n, &ctx->f_batch_rows, &ctx->f_rows, &ctx->f_pairtab, &ctx->fo_sel, &ctx->fo_flag, &ctx->fo_tmp, &ctx->fo_out, &ctx->f_redo, &ctx->f_terms, &ctx->f_soff, &ctx->f_snz, &ctx->f_sdisc, &ctx->f_quot,
                      &ctx->p_state, &ctx->p_keys, &ctx->p_cnt, &ctx->p_out_key, &ctx->p_out_cnt, &ctx->p_batch_rows, &ctx->p_tflag, &ctx->p_tile_row0, &ctx->p_tile_rows})
        b->release();
    for (auto &g : ctx->groups) g.d_tab.release();
    for (auto &t : ctx->timed) { (void)hipEventDestroy(t.beg); (void)hipEventDestroy(t.end); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->d_state) (void)hipFree(ctx->d_state);
    if (ctx->d_state2) (void)hipFree(ctx->d_state2);
    if (ctx->h_state) (void)hipHostFree(ctx->h_state);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->staged_ev) (void)hipEventDestroy(ctx->staged_ev);
    if (ctx->piece_stream) (void)hipStreamDestroy(ctx->piece_stream);
    for (hipEvent_t e : ctx->piece_ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int mth_ctx_set_stream(mth_ctx_t *ctx, void *hip_stream) {
    if (!ctx) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MTH_OK;
}

uint32_t mth_notes(const mth_ctx_t *ctx) { return ctx ? ctx->notes : 0u; }

int mth_batch_prepare(mth_ctx_t *ctx, const mth_batch_t *batch, mth_batch_t *prepared) {
    if (!ctx || !batch || !prepared) return MTH_ERR_INVALID;
    if (batch->mem == MTH_MEM_PREPARED) return mth::fail(ctx, MTH_ERR_INVALID, "the batch is a prepared one already");
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const mth_batch_t &b = *batch;
    if (b.region_end < b.region_beg || b.max_span < 0) return mth::fail(ctx, MTH_ERR_INVALID, "bad region / max_span");
    if (b.n_reads && (!b.read_start || !b.read_end || !b.read_mapq || !b.cpg_off)) return mth::fail(ctx, MTH_ERR_INVALID, "batch arrays missing");
    if (b.n_cpgs && (!b.cpg_pos || (!b.cpg_rel == !b.cpg_rel16))) return mth::fail(ctx, MTH_ERR_INVALID, "exactly one of cpg_rel / cpg_rel16 must be given");
    if (b.mem != MTH_MEM_HOST && b.mem != MTH_MEM_DEVICE) return mth::fail(ctx, MTH_ERR_INVALID, "batch.mem");
    std::unique_ptr<mth::Prepared> pr(new mth::Prepared());
    auto drop = [&]() { for (mth::DevBuf &x : pr->own) x.release(); pr->idx.release(); if (pr->st) (void)hipFree(pr->st); };
#define PREP_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { drop(); return mth::fail(ctx, MTH_ERR_HIP, #call, e__); } } while (0)
    pr->owner = ctx;
    pr->dev = b;
    pr->dev.read_fwd = nullptr;
    if (b.mem == MTH_MEM_HOST) {
        const size_t nr = b.n_reads, nc = b.n_cpgs;
        auto up = [&](mth::DevBuf &buf, const void *src, size_t bytes, const void **dst) -> hipError_t {
            *dst = nullptr;
            if (!src || !bytes) return hipSuccess;
            hipError_t e = buf.reserve(bytes, s);
            if (e != hipSuccess) return e;
            *dst = buf.p;
            return hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, s);
        };
        PREP_HIP(up(pr->own[0], b.read_start, nr * 4, (const void **)&pr->dev.read_start));
        PREP_HIP(up(pr->own[1], b.read_end, nr * 4, (const void **)&pr->dev.read_end));
        PREP_HIP(up(pr->own[2], b.read_mapq, nr, (const void **)&pr->dev.read_mapq));
        PREP_HIP(up(pr->own[3], b.cpg_off, (nr + 1) * 4, (const void **)&pr->dev.cpg_off));
        PREP_HIP(up(pr->own[4], b.cpg_pos, nc * 4, (const void **)&pr->dev.cpg_pos));
        if (b.cpg_rel) PREP_HIP(up(pr->own[5], b.cpg_rel, nc, (const void **)&pr->dev.cpg_rel));
        else PREP_HIP(up(pr->own[5], b.cpg_rel16, nc * 2, (const void **)&pr->dev.cpg_rel16));
        PREP_HIP(hipStreamSynchronize(s));                // the caller's host arrays may go away after this call
    }
    pr->dev.mem = MTH_MEM_DEVICE;
    if (b.n_cpgs == 0 && !pr->dev.cpg_rel && !pr->dev.cpg_rel16) pr->dev.cpg_rel = reinterpret_cast<const uint8_t *>(ctx->d_state);
    PREP_HIP(hipMalloc((void **)&pr->st, sizeof(mth::DevState)));
    PREP_HIP(hipMemsetAsync(pr->st, 0, sizeof(mth::DevState), s));
    mth::fine_index_extent(pr->dev, pr->idx_base, pr->nq);
    PREP_HIP(pr->idx.reserve((size_t)(pr->nq + 1) * 4, s));
    const int rc = mth::build_fine_index(ctx, pr->dev, pr->idx_base, pr->nq, pr->idx.as<uint32_t>(), pr->st);
    if (rc) { drop(); return rc; }
#undef PREP_HIP
    *prepared = pr->dev;
    prepared->mem = MTH_MEM_PREPARED;
    mth::Prepared *raw = pr.release();
    ctx->prepared.insert(raw);
    prepared->read_fwd = reinterpret_cast<const uint8_t *>(raw);
    return MTH_OK;
}

int mth_batch_release(mth_ctx_t *ctx, mth_batch_t *prepared) {
    if (!ctx || !prepared || prepared->mem != MTH_MEM_PREPARED) return MTH_ERR_INVALID;
    mth::Prepared *pr = mth::prepared_lookup(ctx, prepared->read_fwd);
    if (!pr) return mth::fail(ctx, MTH_ERR_INVALID, "not a batch prepared by this context, or released already");
    MTH_ENTER(ctx);
    (void)hipStreamSynchronize(ctx->stream);             // nothing queued may still read the index or the owned arrays
    mth::prepared_free(ctx, pr);
    prepared->read_fwd = nullptr;                        // still marked MTH_MEM_PREPARED, without a handle: every entry point refuses it
    return MTH_OK;
}

int mth_ctx_sync(mth_ctx_t *ctx) {
    if (!ctx) return MTH_ERR_INVALID;
    return sync_and_check(ctx);
}

int mth_group_define(mth_ctx_t *ctx, uint32_t n_contigs, const int32_t *tids, const int64_t *voff, int32_t *handle) {
    if (!ctx || !n_contigs || !tids || !voff || !handle) return MTH_ERR_INVALID;
    for (uint32_t k = 0; k < n_contigs; ++k) {
        if (tids[k] < 0) return fail(ctx, MTH_ERR_INVALID, "contig group: a tid below 0");
        if (voff[k] < 0 || voff[k] >= INT32_MAX || (k && voff[k] <= voff[k - 1])) return fail(ctx, MTH_ERR_INVALID, "contig group: offsets must ascend within [0, 2^31 - 1)");
    }
    if (ctx->groups.size() >= (size_t)(1u << 30)) return fail(ctx, MTH_ERR_CAPACITY, "too many contig groups");
    MTH_ENTER(ctx);
    mth_ctx::ContigGroup g;
    g.tids.assign(tids, tids + n_contigs);
    g.voff.assign(voff, voff + n_contigs);
    std::vector<int32_t> tab(2 * (size_t)n_contigs);
    for (uint32_t k = 0; k < n_contigs; ++k) { tab[k] = (int32_t)voff[k]; tab[n_contigs + k] = tids[k]; }
    MTH_HIP(ctx, g.d_tab.reserve(tab.size() * 4, ctx->stream));
    MTH_HIP(ctx, hipMemcpyAsync(g.d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));      // tab is a local
    *handle = -2 - (int32_t)ctx->groups.size();
    ctx->groups.push_back(std::move(g));
    return MTH_OK;
}

int mth_group_clear(mth_ctx_t *ctx) {
    if (!ctx) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));      // a kernel in flight may be reading a table
    for (auto &g : ctx->groups) g.d_tab.release();
    ctx->groups.clear();
    return MTH_OK;
}

int mth_reset(mth_ctx_t *ctx) {
    if (!ctx) return MTH_ERR_INVALID;
    MTH_HIP(ctx, hipSetDevice(ctx->device));
    // Inside a pipelined run of PDR + LPMD batches the job state (DevState: row count, batch counts, LPMD totals, error bits) is
    // owned by the chain of gathers: the reset is folded into the next batch's gather (or done on ctx->stream by the next join)
    // instead of draining the lanes here.  The other measures' states are only ever touched from ctx->stream.
    if (ctx->pipe_active) ctx->reset_pending = true;
    else MTH_HIP(ctx, hipMemsetAsync(ctx->d_state, 0, sizeof(DevState), ctx->stream));
    ctx->batches.clear();
    ctx->out_bound = 0;
    ctx->lpmd_reduced = false;
    if (ctx->q_state.p) MTH_HIP(ctx, hipMemsetAsync(ctx->q_state.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    ctx->q_meta.clear();
    ctx->q_pending.clear();                // queued batches' rows are dropped with the rest; their kernels precede the memset in stream order
    ctx->q_rows = 0;
    ctx->q_epoch += 1;
    if (ctx->m_state.p) MTH_HIP(ctx, hipMemsetAsync(ctx->m_state.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    ctx->m_batches.clear();
    ctx->m_rows_bound = 0;
    if (ctx->f_state.p) MTH_HIP(ctx, hipMemsetAsync(ctx->f_state.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    ctx->f_batches.clear();
    ctx->f_rows_bound = 0;
    if (ctx->p_state.p) MTH_HIP(ctx, hipMemsetAsync(ctx->p_state.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    ctx->p_meta.clear();
    ctx->p_pending.clear();
    ctx->p_rows = 0;
    return MTH_OK;
}

int mth_pdr_lpmd_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_pdr_lpmd_params_t *params) {
    if (!ctx || !batch || !params) return MTH_ERR_INVALID;
    const mth_batch_t &b = *batch;
    if (!params->want_pdr && !params->want_lpmd) return fail(ctx, MTH_ERR_INVALID, "nothing requested");
    // pdr.rs:160-177: with reads no longer than the 150-bp flush margin a coordinate-sorted input can
    // never re-open a flushed site, and the stream result equals plain per-site counting (the fused
    // tile kernel).  Longer spans take the exact site walk (mth_sites.hip) for the PDR half.
    const bool pdr_exact = params->want_pdr && b.max_span > PDR_FLUSH_MARGIN;
    if (params->want_lpmd) ctx->lpmd_reduced = false;
    // Pipelined batches (mth_pdr_lpmd.hip): device-resident batches that take the fused tile pass, from the second of a run of such
    // calls on (mth_reset does not break a run; any other entry point does).  Not while kernels are being timed one by one.
    if (ctx->pipe_mode < 0) { const char *e = getenv("MTH_PIPELINE"); ctx->pipe_mode = (e && atoi(e) == 0) ? 0 : 1; }
    // (not for a batch of more than 2^29 positions -- a contig group: its kernels fill the chip for a millisecond, the boundary the
    // pipeline hides is nothing beside that, and the second lane's scratch rows would be another 16 B a position)
    const bool eligible = ctx->pipe_mode == 1 && (b.mem == MTH_MEM_DEVICE || b.mem == MTH_MEM_PREPARED) && !pdr_exact && !ctx->timing &&
                          (int64_t)b.region_end - (int64_t)b.region_beg <= ((int64_t)1 << 29);
    mth_batch_t d;
    {
        const int rcs = stage_batch(ctx, b, d, !eligible);
        if (rcs) return rcs;
    }
    // an empty region owns no site and no read: nothing is launched, so nothing may be recorded either (the per-batch
    // row counts on the device are written by the batch's last kernel; an entry without one stayed uninitialised)
    if (b.region_end == b.region_beg) return MTH_OK;

    // result capacity: at most one row per call and per owned position
    if (params->want_pdr) {
        const uint64_t region_len = (uint64_t)((int64_t)b.region_end - b.region_beg);
        const uint64_t add = b.n_cpgs < region_len ? b.n_cpgs : region_len;
        if (ctx->out_bound + add > ctx->out_cap) {
            // learn how many rows are really in use before growing
            int rc = sync_and_check(ctx);
            if (rc) return rc;
            const uint64_t used = ctx->h_state->n_sites;
            uint64_t ncap = used + add;
            ncap += ncap / 4 + 1024;
            MTH_HIP(ctx, ctx->out_pos.reserve(ncap * 4, ctx->stream, true, used * 4));
            MTH_HIP(ctx, ctx->out_pdr.reserve(ncap * 4, ctx->stream, true, used * 4));
            MTH_HIP(ctx, ctx->out_nc.reserve(ncap * 4, ctx->stream, true, used * 4));
            MTH_HIP(ctx, ctx->out_nd.reserve(ncap * 4, ctx->stream, true, used * 4));
            ctx->out_cap = ctx->out_pos.cap / 4;
            ctx->out_bound = used;
        }
        ctx->out_bound += add;
    }
    if ((ctx->batches.size() + 2) * 4 > ctx->batch_cnt.cap) {
        MTH_ENTER(ctx);       // the gathers in flight write this array: drain them before it moves
        MTH_HIP(ctx, ctx->batch_cnt.reserve((ctx->batches.size() + 2) * 4 + 4096, ctx->stream, true, ctx->batches.size() * 4));
    }
    int rc;
    if (!pdr_exact) {
        const bool pipelined = eligible && ctx->pdr_streak >= 1;       // (a join above -- growth -- restarts the run)
        if ((rc = launch_pdr_lpmd(ctx, d, *params, nullptr, pipelined))) return rc;
        ctx->batches.push_back(BatchMeta{b.tid});
        if (eligible) ctx->pdr_streak += 1;
        return MTH_OK;
    }
    // (every pass appends one (possibly empty) entry to the per-batch row counts: two entries were reserved above)
    if (params->want_lpmd) {
        mth_pdr_lpmd_params_t lp = *params;
        lp.want_pdr = 0;
        if ((rc = launch_pdr_lpmd(ctx, d, lp))) return rc;
        ctx->batches.push_back(BatchMeta{b.tid});
    }
    if ((rc = launch_pdr_exact(ctx, d, *params))) return rc;
    ctx->batches.push_back(BatchMeta{b.tid});
    return MTH_OK;
}

int mth_pdr_count(mth_ctx_t *ctx, uint64_t *n_sites) {
    if (!ctx || !n_sites) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    *n_sites = ctx->h_state->n_sites;
    return MTH_OK;
}

int mth_pdr_fetch(mth_ctx_t *ctx, int32_t *tid, int32_t *pos, float *pdr, uint32_t *nc, uint32_t *nd) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    const uint64_t n = ctx->h_state->n_sites;
    if (n == 0) return MTH_OK;
    // four copies in flight, one wait (into mth_result_buffer_alloc memory they run at the link's rate; pageable destinations
    // go through the runtime's bounce buffers as before)
    if (pos) MTH_HIP(ctx, hipMemcpyAsync(pos, ctx->out_pos.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pdr) MTH_HIP(ctx, hipMemcpyAsync(pdr, ctx->out_pdr.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (nc) MTH_HIP(ctx, hipMemcpyAsync(nc, ctx->out_nc.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (nd) MTH_HIP(ctx, hipMemcpyAsync(nd, ctx->out_nd.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // rows of contig groups come back under their own contig: both columns are needed for that, whichever the caller asked for
    const bool grouped = (tid || pos) && has_group_batch(ctx, 0);
    std::vector<int32_t> tmp;
    if (grouped && !(tid && pos)) {
        tmp.resize(n);
        if (!pos) MTH_HIP(ctx, hipMemcpy(tmp.data(), ctx->out_pos.p, n * 4, hipMemcpyDeviceToHost));
    }
    int32_t *tid_w = tid ? tid : (grouped ? tmp.data() : nullptr), *pos_w = pos ? pos : (grouped ? tmp.data() : nullptr);
    if (tid_w) {
        std::vector<uint32_t> cnt(ctx->batches.size());
        if (!cnt.empty()) MTH_HIP(ctx, hipMemcpy(cnt.data(), ctx->batch_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
        uint64_t o = 0;
        for (size_t b = 0; b < cnt.size(); ++b)
            for (uint32_t j = 0; j < cnt[b] && o < n; ++j) tid_w[o++] = ctx->batches[b].tid;
        if (o != n) return fail(ctx, MTH_ERR_STATE, "per-batch row counts do not add up to the row count");
    }
    if (grouped) return ungroup_rows(ctx, n, tid_w, pos_w, 1, 1, nullptr);
    return MTH_OK;
}

int mth_result_buffer_alloc(mth_ctx_t *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return MTH_ERR_INVALID;
    *out = nullptr;
    MTH_ENTER(ctx);
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(ctx, MTH_ERR_HIP, "hipHostMalloc (result buffer)", e); }
    return MTH_OK;
}

int mth_result_buffer_free(mth_ctx_t *ctx, void *p) {
    if (!ctx) return MTH_ERR_INVALID;
    if (p) MTH_HIP(ctx, hipHostFree(p));
    return MTH_OK;
}

int mth_pdr_device_view(mth_ctx_t *ctx, uint64_t *n_sites, const int32_t **pos, const float **pdr,
                        const uint32_t **nc, const uint32_t **nd) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    if (n_sites) *n_sites = ctx->h_state->n_sites;
    if (pos) *pos = ctx->out_pos.as<int32_t>();
    if (pdr) *pdr = ctx->out_pdr.as<float>();
    if (nc) *nc = ctx->out_nc.as<uint32_t>();
    if (nd) *nd = ctx->out_nd.as<uint32_t>();
    return MTH_OK;
}

float mth_lpmd_from_counts(int64_t n_conc, int64_t n_disc) {
    // lpmd.rs:11-12: i32 counters (wrapping in a release build); lpmd.rs:51-55
    const int32_t wc = (int32_t)(uint32_t)(uint64_t)n_conc, wd = (int32_t)(uint32_t)(uint64_t)n_disc;
    const int32_t ws = (int32_t)((uint32_t)wc + (uint32_t)wd);
    return (float)wd / (float)ws;
}

int mth_lpmd_global(mth_ctx_t *ctx, int64_t out[4], float *lpmd) {
    if (!ctx) return MTH_ERR_INVALID;
    int rc = sync_and_check(ctx);
    if (rc) return rc;
    long long v[4];
    for (int k = 0; k < 4; ++k) v[k] = ctx->h_state->lpmd[k];
    if (ctx->lpmd_reduced && ctx->red_slot >= 0) {      // all-reduced on the side stream (mth_allreduce_lpmd_rank): totals are in the ring slot
        MTH_HIP(ctx, hipStreamSynchronize(ctx->red_stream));
        MTH_HIP(ctx, hipMemcpy(v, ctx->red_buf + 4 * ctx->red_slot, sizeof v, hipMemcpyDeviceToHost));
    }
    if (out) for (int k = 0; k < 4; ++k) out[k] = v[k];
    if (lpmd) *lpmd = mth_lpmd_from_counts(v[0], v[1]);
    return MTH_OK;
}

// lpmd.rs:176-179 counts EVERY record of the file in n_read, and in n_valid_read when its mapq passes -- also the records that never
// enter a batch (no contig, no aligned base: they have no CpG and no position).  The caller that dropped them hands their counts over.
int mth_lpmd_add_unbatched(mth_ctx_t *ctx, uint64_t n_read, uint64_t n_valid_read) {
    if (!ctx || n_valid_read > n_read) return MTH_ERR_INVALID;
    if (n_read == 0) return MTH_OK;
    MTH_ENTER(ctx);
    hipLaunchKernelGGL(k_lpmd_add2, dim3(1), dim3(1), 0, ctx->stream, ctx->d_state, (long long)n_read, (long long)n_valid_read);
    MTH_HIP(ctx, hipGetLastError());
    return MTH_OK;
}

int mth_lpmd_export_device(mth_ctx_t *ctx, int64_t *dst) {
    if (!ctx || !dst) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    MTH_HIP(ctx, hipMemcpyAsync(dst, ctx->d_state->lpmd, 4 * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    return MTH_OK;
}

int mth_timing_enable(mth_ctx_t *ctx, int on) {
    if (!ctx) return MTH_ERR_INVALID;
    ctx->timing = on != 0;
    return MTH_OK;
}

int mth_timing_reset(mth_ctx_t *ctx) {
    if (!ctx) return MTH_ERR_INVALID;
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &t : ctx->timed) { ctx->event_pool.push_back(t.beg); ctx->event_pool.push_back(t.end); }
    ctx->timed.clear();
    return MTH_OK;
}

int mth_timing_num_kernels(void) { return K_NUM; }
const char *mth_timing_kernel_name(int i) { return (i >= 0 && i < K_NUM) ? kKernelNames[i] : ""; }

int mth_timing_get(mth_ctx_t *ctx, const char *kernel, double *avg_ms, uint64_t *launches) {
    if (!ctx || !kernel) return MTH_ERR_INVALID;
    int id = -1;
    for (int i = 0; i < K_NUM; ++i) if (strcmp(kernel, kKernelNames[i]) == 0) id = i;
    if (id < 0) return fail(ctx, MTH_ERR_INVALID, "unknown kernel name");
    MTH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0; uint64_t n = 0;
    for (auto &t : ctx->timed) {
        if (t.kernel != id) continue;
        float ms = 0;
        MTH_HIP(ctx, hipEventElapsedTime(&ms, t.beg, t.end));
        total += ms; n += 1;
    }
    if (avg_ms) *avg_ms = n ? total / (double)n : 0.0;
    if (launches) *launches = n;
    return MTH_OK;
}

}  // extern "C"
