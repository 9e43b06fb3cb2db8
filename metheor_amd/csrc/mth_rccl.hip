// mth_rccl.hip -- the one exchange step of the path (lpmd.rs:11-12, 51-55): the four exact LPMD
// counters {n_concordant, n_discordant, n_read, n_valid_read} summed over the GPUs of a node with ONE RCCL
// all-reduce (ncclInt64 x 4, ncclSum) over xGMI.  Everything else of a sharded run is region-owned and needs no
// exchange.  Two host shapes:
//   * one process, one context per GPU (the `metheor --gpus N` executable): mth_allreduce_lpmd(ctxs, n);
//   * one process per GPU (torch.distributed / MPI style launchers): mth_rccl_unique_id on rank 0, the host ships
//     the 128 bytes to the other ranks, mth_rccl_init_rank everywhere, mth_allreduce_lpmd_rank everywhere.
// Afterwards mth_lpmd_global of every context returns the node-wide counters (the single-process form reduces in
// place on DevState.lpmd; the rank form on a side stream into a ring slot, so the context's stream never waits).
//
// librccl is loaded on first use (dlopen): a single-GPU run never pays for it (it is a large library, the CLI's
// whole run is ~0.3 s), and a process that already carries an RCCL (PyTorch) keeps using that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "mth_ctx.h"

static_assert(sizeof(ncclUniqueId) == MTH_RCCL_ID_BYTES, "mth_rccl_unique_id hands out an ncclUniqueId");

namespace mth {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

static Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why = std::string("cannot load librccl: ") + dlerror(); return; }
#define MTH_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f)); if (!r.f) { r.why = "librccl lacks nccl" #f; r.lib = nullptr; return; }
        MTH_SYM(GetUniqueId) MTH_SYM(CommInitRank) MTH_SYM(CommInitAll) MTH_SYM(CommDestroy) MTH_SYM(AllReduce)
        MTH_SYM(GroupStart) MTH_SYM(GroupEnd) MTH_SYM(GetErrorString)
#undef MTH_SYM
    });
    return r.lib ? &r : nullptr;
}

static int rccl_fail(mth_ctx *ctx, const char *what, ncclResult_t e) {
    Rccl *r = rccl();
    std::string m = what;
    if (r && e != ncclSuccess) { m += ": "; m += r->GetErrorString(e); }
    if (ctx) ctx->last_error = m;
    return MTH_ERR_RCCL;
}

__global__ void k_add4(long long *__restrict__ dst, const long long *__restrict__ src) {
    if (threadIdx.x < 4) dst[threadIdx.x] += src[threadIdx.x];
}

// communicators of the single-process form, one set per distinct device list (created once, kept for the process)
struct CommSet { std::vector<int> devs; std::vector<ncclComm_t> comms; };
static std::vector<CommSet> g_sets;
static std::mutex g_sets_mu;

}  // namespace mth

using namespace mth;

extern "C" {

int mth_device_count(int *n) {
    if (!n) return MTH_ERR_INVALID;
    int k = 0;
    if (hipGetDeviceCount(&k) != hipSuccess) k = 0;
    *n = k;
    return MTH_OK;
}

int mth_rccl_unique_id(void *id128) {
    if (!id128) return MTH_ERR_INVALID;
    Rccl *r = rccl();
    if (!r) return MTH_ERR_RCCL;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return MTH_ERR_RCCL;
    memcpy(id128, &id, sizeof id);
    return MTH_OK;
}

int mth_rccl_init_rank(mth_ctx_t *ctx, const void *id128, int rank, int world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return MTH_ERR_INVALID;
    Rccl *r = rccl();
    if (!r) return rccl_fail(ctx, "RCCL is not available", ncclSuccess);
    if (ctx->rccl_comm) return fail(ctx, MTH_ERR_STATE, "this context already has a communicator");
    MTH_ENTER(ctx);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, world, id, rank);
    if (e != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", e);
    ctx->rccl_comm = c; ctx->rccl_rank = rank; ctx->rccl_world = world;
    return MTH_OK;
}

// The ctx stream never waits for the collective: the counters are copied into a ring slot on the ctx stream, the
// all-reduce runs on a side stream ordered after that copy, and mth_lpmd_global reads the slot after draining the side
// stream.  A slot is reused RED_RING calls later (ordered by its done-event), so a host that issues one reduce per batch
// loop iteration overlaps it with the next iterations' kernels (32 bytes: pure latency, ~10-30 us over xGMI).
int mth_allreduce_lpmd_rank(mth_ctx_t *ctx) {
    if (!ctx) return MTH_ERR_INVALID;
    if (!ctx->rccl_comm) return fail(ctx, MTH_ERR_STATE, "mth_rccl_init_rank first");
    if (ctx->lpmd_reduced) return fail(ctx, MTH_ERR_STATE, "the LPMD counters of this context are already all-reduced");
    Rccl *r = rccl();
    MTH_ENTER(ctx);
    if (!ctx->red_stream) {
        MTH_HIP(ctx, hipStreamCreateWithFlags(&ctx->red_stream, hipStreamNonBlocking));
        MTH_HIP(ctx, hipMalloc((void **)&ctx->red_buf, mth_ctx::RED_RING * 4 * sizeof(long long)));
        for (int k = 0; k < mth_ctx::RED_RING; ++k) {
            MTH_HIP(ctx, hipEventCreateWithFlags(&ctx->red_ready[k], hipEventDisableTiming));
            MTH_HIP(ctx, hipEventCreateWithFlags(&ctx->red_done[k], hipEventDisableTiming));
        }
    }
    const int k = (int)(ctx->red_head++ % mth_ctx::RED_RING);
    long long *buf = ctx->red_buf + 4 * k;
    if (ctx->red_head > (uint64_t)mth_ctx::RED_RING) MTH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->red_done[k], 0));   // the slot's previous collective
    MTH_HIP(ctx, hipMemcpyAsync(buf, ctx->d_state->lpmd, 4 * sizeof(long long), hipMemcpyDeviceToDevice, ctx->stream));
    MTH_HIP(ctx, hipEventRecord(ctx->red_ready[k], ctx->stream));
    MTH_HIP(ctx, hipStreamWaitEvent(ctx->red_stream, ctx->red_ready[k], 0));
    const ncclResult_t e = r->AllReduce(buf, buf, 4, ncclInt64, ncclSum, (ncclComm_t)ctx->rccl_comm, ctx->red_stream);
    if (e != ncclSuccess) return rccl_fail(ctx, "ncclAllReduce", e);
    MTH_HIP(ctx, hipEventRecord(ctx->red_done[k], ctx->red_stream));
    ctx->lpmd_reduced = true;
    ctx->red_slot = k;
    return MTH_OK;
}

int mth_allreduce_lpmd(mth_ctx_t **ctxs, int n) {
    if (!ctxs || n < 1) return MTH_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return MTH_ERR_INVALID;
        for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return fail(ctxs[i], MTH_ERR_INVALID, "a context appears twice");
        if (ctxs[i]->lpmd_reduced) return fail(ctxs[i], MTH_ERR_STATE, "the LPMD counters of this context are already all-reduced");
    }
    for (int i = 0; i < n; ++i) { const int rc = enter(ctxs[i]); if (rc) return rc; }     // joins each context's PDR + LPMD pipeline
    if (n == 1) { ctxs[0]->lpmd_reduced = true; ctxs[0]->red_slot = -1; return MTH_OK; }
    // contexts that share a GPU (several shards per device) are summed on that GPU into the first of them; RCCL
    // then runs between one context per distinct GPU (it refuses two ranks on one device); the result is copied back
    std::vector<int> leader((size_t)n);
    std::vector<int> leaders;
    for (int i = 0; i < n; ++i) {
        leader[(size_t)i] = i;
        for (int j = 0; j < i; ++j) if (ctxs[j]->device == ctxs[i]->device) { leader[(size_t)i] = leader[(size_t)j]; break; }
        if (leader[(size_t)i] == i) leaders.push_back(i);
    }
    auto lp = [&](int i) { return reinterpret_cast<long long *>(ctxs[i]->d_state->lpmd); };
    std::vector<hipEvent_t> evs;
    auto cleanup = [&]() { for (hipEvent_t e : evs) (void)hipEventDestroy(e); };
    for (int i = 0; i < n; ++i) {
        const int L = leader[(size_t)i];
        if (L == i) continue;
        mth_ctx *c = ctxs[i], *l = ctxs[L];
        hipEvent_t ev = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { cleanup(); return fail(c, MTH_ERR_HIP, "event"); }
        evs.push_back(ev);
        if (hipEventRecord(ev, c->stream) != hipSuccess || hipStreamWaitEvent(l->stream, ev, 0) != hipSuccess) { cleanup(); return fail(c, MTH_ERR_HIP, "event"); }
        hipLaunchKernelGGL(k_add4, dim3(1), dim3(64), 0, l->stream, lp(L), lp(i));
    }
    if (leaders.size() > 1) {
        Rccl *r = rccl();
        if (!r) { cleanup(); return rccl_fail(ctxs[0], "RCCL is not available", ncclSuccess); }
        std::vector<int> devs;
        for (int i : leaders) devs.push_back(ctxs[i]->device);
        // the communicators of this device list, copied out while the lock is held: g_sets may grow (and move) under another
        // thread's call with another list as soon as it is released
        std::vector<ncclComm_t> comms;
        {
            std::lock_guard<std::mutex> g(g_sets_mu);
            for (CommSet &s : g_sets) if (s.devs == devs) comms = s.comms;
            if (comms.empty()) {
                CommSet s;
                s.devs = devs;
                s.comms.assign(devs.size(), (ncclComm_t) nullptr);
                const ncclResult_t e = r->CommInitAll(s.comms.data(), (int)devs.size(), devs.data());
                if (e != ncclSuccess) {
                    for (ncclComm_t c : s.comms) if (c) (void)r->CommDestroy(c);      // whatever was created before the failure
                    cleanup();
                    return rccl_fail(ctxs[0], "ncclCommInitAll", e);
                }
                comms = s.comms;
                g_sets.push_back(std::move(s));
            }
        }
        ncclResult_t e = r->GroupStart();
        for (size_t k = 0; k < leaders.size() && e == ncclSuccess; ++k) {
            mth_ctx *c = ctxs[leaders[k]];
            if (hipSetDevice(c->device) != hipSuccess) { e = ncclUnhandledCudaError; break; }
            e = r->AllReduce(lp(leaders[k]), lp(leaders[k]), 4, ncclInt64, ncclSum, comms[k], c->stream);
        }
        const ncclResult_t e2 = r->GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) { cleanup(); return rccl_fail(ctxs[0], "ncclAllReduce", e != ncclSuccess ? e : e2); }
    }
    for (int i = 0; i < n; ++i) {
        const int L = leader[(size_t)i];
        if (L == i) continue;
        mth_ctx *c = ctxs[i], *l = ctxs[L];
        hipEvent_t ev = nullptr;
        if (hipSetDevice(l->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { cleanup(); return fail(c, MTH_ERR_HIP, "event"); }
        evs.push_back(ev);
        if (hipEventRecord(ev, l->stream) != hipSuccess || hipStreamWaitEvent(c->stream, ev, 0) != hipSuccess ||
            hipMemcpyAsync(lp(i), lp(L), 4 * sizeof(long long), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { cleanup(); return fail(c, MTH_ERR_HIP, "copy back"); }
    }
    // the events must outlive the work that waits on them: drain before destroying (32 bytes were moved; this is the
    // end of a run's device work anyway)
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) { cleanup(); return fail(ctxs[i], MTH_ERR_HIP, "sync"); }
        ctxs[i]->lpmd_reduced = true;
        ctxs[i]->red_slot = -1;       // in place: DevState.lpmd holds the totals
    }
    cleanup();
    return MTH_OK;
}

}  // extern "C"

namespace mth {
void rccl_release(mth_ctx *ctx) {
    if (ctx->red_stream) {
        (void)hipStreamSynchronize(ctx->red_stream);
        for (int k = 0; k < mth_ctx::RED_RING; ++k) { (void)hipEventDestroy(ctx->red_ready[k]); (void)hipEventDestroy(ctx->red_done[k]); }
        (void)hipFree(ctx->red_buf);
        (void)hipStreamDestroy(ctx->red_stream);
        ctx->red_stream = nullptr; ctx->red_buf = nullptr;
    }
    if (!ctx->rccl_comm) return;
    if (Rccl *r = rccl()) (void)r->CommDestroy((ncclComm_t)ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
}
}  // namespace mth
