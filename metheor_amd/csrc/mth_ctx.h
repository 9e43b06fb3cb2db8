// mth_ctx.h -- host-side context of the engine (private).
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "mth_common.h"

namespace mth {

// grow-only device buffer
struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes, hipStream_t s, bool keep = false, size_t keep_bytes = 0);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct TimedLaunch {
    int kernel;
    hipEvent_t beg, end;
};

struct BatchMeta {
    int32_t tid;
};

// One lane of the PDR + LPMD batch pipeline (mth_pdr_lpmd.hip, "Pipelined batches"): a stream of its own, the per-batch work
// buffers (lane 0 borrows the context's own idx / tile_cnt / tile_bucket / scratch, lane 1 has a second set) and a small
// device block for what belongs to the batch in flight on the lane rather than to the job: error bits, safe_hi, the row base.
struct PdrLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;          // recorded behind the lane's latest k_gather
    DevBuf idx, tile_cnt, tile_bucket, scratch;      // lane 1 only
    DevState *st = nullptr;
    bool used = false;                  // has work that ctx->stream has not been made to wait for
};

}  // namespace mth

struct mth_ctx;
namespace mth { int quartet_resolve(mth_ctx *ctx); int pairs_resolve(mth_ctx *ctx); }

namespace mth {
// a prepared batch (include/metheor_hip.h, "prepared batches"): the device-resident batch, the buffers it owns when it was made
// from a host batch, its fine read index, and a device block with what k_build_index found (error bits, safe_hi)
struct Prepared {
    static constexpr uint64_t MAGIC = 0x6d74685f70726570ull;
    uint64_t magic = MAGIC;
    mth_ctx *owner = nullptr;
    mth_batch_t dev{};
    DevBuf own[7];
    DevBuf idx;
    int32_t idx_base = 0;
    uint32_t nq = 0;
    DevState *st = nullptr;
};
}  // namespace mth

struct mth_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    // the prepared batches this context owns (mth_batch_prepare): a handle is valid iff it is in here -- never dereferenced to find out;
    // mth_ctx_destroy frees what the caller did not release
    std::unordered_set<mth::Prepared *> prepared;
    mth::Prepared *cur_prep = nullptr;   // the batch of the entry point in progress is a prepared one (set by stage_batch, every call)
    const uint32_t *cur_idx = nullptr;   // the fine read index the call's kernels use: the prepared batch's, or ctx->idx after a build
    uint32_t notes = 0;                 // mth_notes(): non-fatal findings so far (MTH_NOTE_*)

    mth::DevState *d_state = nullptr;   // device
    mth::DevState *h_state = nullptr;   // pinned host mirror
    unsigned long long *h_words = nullptr;   // 16 pinned words: per-batch read-backs of the tile-kernel measures

    // staging for MTH_MEM_HOST batches
    mth::DevBuf st_start, st_end, st_mapq, st_fwd, st_off, st_pos, st_rel;
    // per-batch work buffers
    mth::DevBuf idx, tile_cnt, tile_bucket, scratch, batch_cnt;
    // PDR + LPMD batch pipeline: consecutive device-resident batches alternate between two lanes so that batch k+1's index build
    // and the head of its tile kernel run beside batch k's tail and gather; only the gathers form a chain (each waits for the one
    // before it: row bases, batch counts and the LPMD totals live in DevState).  Any other entry point joins the lanes back into
    // ctx->stream first (mth::enter).
    mth::PdrLane lane[2];
    hipEvent_t pipe_in = nullptr;       // recorded on ctx->stream at each pipelined call: the lane waits for the caller's producers
    int pipe_mode = -1;                 // -1: not decided (MTH_PIPELINE env), 0: off, 1: on
    int pipe_next = 0;                  // lane of the next pipelined batch
    int pipe_tail = -1;                 // lane that holds the latest gather (-1: none since the last join)
    bool pipe_active = false;           // lanes hold work ctx->stream has not been ordered behind
    unsigned pdr_streak = 0;            // consecutive eligible mth_pdr_lpmd_accumulate calls (mth_reset does not break the run)
    bool reset_pending = false;         // mth_reset inside a pipelined run: folded into the next batch's gather
    // device-side BAM record decode (mth_decode.hip): staged input, decoded SoA, scan scratch, one batch's 32-bit offsets
    mth::DevBuf dec_raw, dec_recoff, dec_tid, dec_start, dec_end, dec_mapq, dec_fwd, dec_n, dec_off, dec_pos, dec_rel, dec_blk, dec_off32, dec_runs, dec_xm, dec_filter;
    bool dec_filter_on = false;
    uint32_t dec_xm_min_mapq = 0;       // records without XM:Z are an error from this mapq up (0: always)
    uint64_t dec_filter_n = 0;
    uint64_t dec_reads = 0, dec_cpgs = 0;
    uint32_t dec_contig_flags = 0;      // flags of the latest mth_decoded_contigs over the decoded stream (bit0 / bit1: records a batch cannot hold)
    // device-side BGZF inflate + per-block record walk (mth_inflate.hip)
    mth::DevBuf inf_file, inf_tab, inf_raw, inf_cnt, inf_base, inf_recoff, crc_mat;
    // mth_bgzf_stage: the NEXT chunk's file bytes, copied on a side stream while the current chunk is being inflated
    mth::DevBuf inf_file2;
    hipStream_t copy_stream = nullptr;
    hipEvent_t staged_ev = nullptr;
    hipStream_t piece_stream = nullptr;  // an in-line chunk's file bytes travel here in pieces, each piece's blocks inflate as it lands
    std::vector<hipEvent_t> piece_ev;
    const void *staged_src = nullptr;
    uint64_t staged_bytes = 0;
    const void *next_src = nullptr;      // announced by mth_bgzf_stage, copied by a helper thread the next inflate call starts
    uint64_t next_bytes = 0;
    std::thread stage_thread;
    int stage_rc = 0;
    // `tag` (mth_tag.hip): the genome (contigs back to back), per-contig offsets + header lengths, per-record work arrays,
    // and the host copies mth_tag_records hands out
    mth::DevBuf tag_genome, tag_goff, tag_ncol, tag_coloff, tag_xmlen, tag_cols, tag_xm;
    int32_t tag_n_refs = -1;
    std::vector<uint64_t> tag_h_off;
    std::vector<uint32_t> tag_h_len;
    std::vector<uint8_t> tag_h_xm;
    // contig groups (include/metheor_hip.h, "contig groups"): handle -2 - k is groups[k]
    struct ContigGroup { std::vector<int32_t> tids; std::vector<int64_t> voff; mth::DevBuf d_tab; };   // d_tab: n voff (int32) then n tids
    std::vector<ContigGroup> groups;
    bool dec_grouped = false;                // mth_decoded_group has shifted the decoded stream's positions
    // results (PDR columns)
    mth::DevBuf out_pos, out_pdr, out_nc, out_nd;
    uint64_t out_cap = 0;        // rows
    uint64_t out_bound = 0;      // host-known upper bound of rows in use
    std::vector<mth::BatchMeta> batches;
    size_t batch_cnt_cap = 0;

    // ME / PM (mth_quartet.hip): hash table of the batch in flight + appended result rows
    mth::DevBuf q_state, q_keys, q_hist, q_blk, q_batch_rows, q_tflag, q_tile_row0, q_tile_rows, q_wpos, q_wpat, q_wk0, q_wk1;
    // a batch of the tile-kernel measures (quartets, pairs): heavy0 = first row of the global path's rows, tile_end = tiles of
    // all batches up to and including it
    struct TileBatch { int32_t tid; uint64_t rows, heavy0, tile_end; };
    std::vector<TileBatch> q_meta;
    double q_rows_per_cpg = 0.15;          // output sizing of the next batch
    mth::DevBuf q_pos, q_cnt, q_me, q_pm, q_depth;
    uint64_t q_cap = 0, q_rows = 0;        // row capacity, rows in use (exact as of the last synchronous batch / quartet_resolve)
    // batches queued without a host sync (mth_quartet.hip, quartet_batch): their device-side batch, a snapshot of the state words each
    struct QueuedBatch { mth_batch_t d; mth_quartet_params_t params; int32_t tid; uint64_t n_cpgs; };
    std::vector<QueuedBatch> q_pending;
    mth::DevBuf q_snap;
    uint64_t q_rows_est = 0;               // upper bound of the rows in use while batches are queued
    bool q_learned = false;                // a synchronous batch has set q_rows_per_cpg
    bool tile_queue_hold = false;          // set by the accumulate call that is queueing a batch: enter() leaves the queues alone
    // the row order worked out by a count-only mth_quartet_fetch, kept for the fetch that follows it (same min_depth,
    // nothing accumulated in between: q_epoch)
    uint64_t q_epoch = 0, q_order_epoch = ~0ull;
    uint32_t q_order_min_depth = 0;
    std::vector<uint64_t> q_order;
    std::vector<int32_t> q_order_tid;

    // site-walk measures (mth_sites.hip): discovery sink, per-candidate work arrays, MHL result rows
    mth::DevState *d_state2 = nullptr;
    mth::DevBuf s_pos, s_pdr, s_nc, s_nd, s_batch_cnt;
    mth::DevBuf w_val, w_cov, w_aux, w_flags, w_blk, w_huge;
    mth::DevBuf m_state, m_pos, m_val, m_cov, m_batch_rows;
    uint64_t m_cap = 0, m_rows_bound = 0;
    std::vector<mth::BatchMeta> m_batches;

    // FDRP / qFDRP result rows (mth_fdrp.hip)
    mth::DevBuf f_state, f_pos, f_val, f_qval, f_n, f_batch_rows, f_rows, f_pairtab, f_redo, f_terms, f_soff, f_snz, f_sdisc, f_quot;
    uint64_t f_cap = 0, f_rows_bound = 0;
    std::vector<mth::BatchMeta> f_batches;

    // the flush-based measures of a stream that is not coordinate-sorted (mth_fileorder.hip): work arrays, result rows
    mth::DevBuf fo_sel, fo_flag, fo_tmp, fo_out;
    uint64_t fo_rows = 0;

    // LPMD per-pair table (mth_pairs.hip)
    mth::DevBuf p_state, p_keys, p_cnt, p_out_key, p_out_cnt, p_batch_rows, p_tflag, p_tile_row0, p_tile_rows;
    uint64_t p_cap = 0, p_rows = 0;        // row capacity of p_out_*, rows in use (known exactly: one sync per batch)
    double p_rows_per_cpg = 0.1;           // output sizing of the next batch
    std::vector<TileBatch> p_meta;
    struct QueuedPairs { mth_batch_t d; mth_lpmd_pairs_params_t params; int32_t tid; uint64_t n_cpgs; };   // as q_pending (mth_pairs.hip)
    std::vector<QueuedPairs> p_pending;
    mth::DevBuf p_snap;
    uint64_t p_rows_est = 0;
    bool p_learned = false;

    // multi-GPU exchange step (mth_rccl.hip): communicator of the one-process-per-GPU form; lpmd_reduced = DevState.lpmd
    // already holds the all-reduced totals (cleared by the next batch that adds to them)
    void *rccl_comm = nullptr;
    int rccl_rank = 0, rccl_world = 1;
    bool lpmd_reduced = false;
    // the rank form reduces out of place on a side stream: ring of 4-counter slots, red_slot = slot of the latest reduce
    // (-1: the totals are in DevState.lpmd itself)
    static constexpr int RED_RING = 4;
    hipStream_t red_stream = nullptr;
    long long *red_buf = nullptr;
    hipEvent_t red_ready[RED_RING] = {}, red_done[RED_RING] = {};
    uint64_t red_head = 0;
    int red_slot = -1;

    bool timing = false;
    std::vector<mth::TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
};

namespace mth {

int  fail(mth_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess);
// first thing every entry point does: make the context's device current for the calling thread (it may be new) and order
// ctx->stream behind whatever the PDR + LPMD pipeline lanes still hold
int  enter(mth_ctx *ctx);
// rows of grouped batches back to (tid, position): tid[i] <= -2 is a group handle, pos_a[i * stride .. + cols) and pos_b[i] (optional)
// its virtual positions (mth_api.hip)
int  ungroup_rows(mth_ctx *ctx, uint64_t n, int32_t *tid, int32_t *pos_a, int stride, int cols, int32_t *pos_b);
bool has_group_batch(const mth_ctx *ctx, int which);      // any grouped batch among the accumulated ones of measure `which` (0 pdr 1 mhl 2 fdrp 3 quartet 4 pairs)
// device table of a group handle (n voff as int32, then n tids), nullptr / 0 for a plain tid
const int32_t *group_table(const mth_ctx *ctx, int32_t tid, uint32_t *n);
int  pipe_join(mth_ctx *ctx);
// the context's prepared batch behind a caller's handle, nullptr if it is not one of them (released, another context's, made up)
Prepared *prepared_lookup(const mth_ctx *ctx, const void *handle);
void prepared_free(mth_ctx *ctx, Prepared *pr);           // buffers + the object itself; the registry entry goes with it
#define MTH_ENTER(ctx)                                                  \
    do {                                                                \
        const int rc__ = mth::enter(ctx);                               \
        if (rc__) return rc__;                                          \
    } while (0)
#define MTH_HIP(ctx, call)                                              \
    do {                                                                \
        hipError_t e__ = (call);                                        \
        if (e__ != hipSuccess) return mth::fail(ctx, MTH_ERR_HIP, #call, e__); \
    } while (0)

// time-bracketed kernel launch helper
struct LaunchTimer {
    mth_ctx *ctx;
    int kernel;
    hipEvent_t beg = nullptr, end = nullptr;
    LaunchTimer(mth_ctx *c, int k);
    ~LaunchTimer();
};

void rccl_release(mth_ctx *ctx);    // mth_rccl.hip: destroy the context's communicator, if any
int sync_and_check(mth_ctx *ctx);   // stream sync + read DevState + map error bits
// validate a caller batch and make it device-resident (MTH_MEM_HOST arrays go through the staging buffers)
int stage_batch(mth_ctx *ctx, const mth_batch_t &b, mth_batch_t &dev, bool join = true);
// mth_decode.hip: 64-bit exclusive scan of a u32 array (synchronises), and the record decode over device-resident input
int scan_u32_to_u64(mth_ctx *ctx, const uint32_t *n, uint32_t count, unsigned long long base, unsigned long long *off,
                    unsigned long long *total_host);
int decode_core(mth_ctx *ctx, const uint8_t *d_raw, const uint64_t *d_off, uint64_t n_rec, int append, mth_decoded_t *out);
// implemented in mth_sites.hip: PDR with exact flush / re-open semantics (spans > 150 bp)
int launch_pdr_exact(mth_ctx *ctx, const mth_batch_t &dev_batch, const mth_pdr_lpmd_params_t &p);
// site discovery (tile pipeline into the private sink): positions called by >= 1 read with
// mapq >= min_qual and n_cpgs >= max(min_cpgs,1); bound = host-known upper bound of the site count
int discover_sites(mth_ctx *ctx, const mth_batch_t &dev_batch, uint32_t min_cpgs, uint8_t min_qual, uint64_t &bound,
                   uint32_t min_cov = 0);

// implemented in mth_pdr_lpmd.hip.  sink == nullptr: rows go to the ctx's PDR result columns.
struct TileSink {
    DevState *st;          // counters (n_sites / cur_base / n_batches / lpmd) of this sink
    int32_t *pos;
    float *pdr;
    uint32_t *nc, *nd;
    uint32_t *batch_cnt;
};
int build_read_index(mth_ctx *ctx, const mth_batch_t &dev_batch, int tile_w, int32_t &idx_base, uint32_t &ntiles);
// the fine index of a batch into a caller-owned buffer; errors (unsorted) and safe_hi land in *st (mth_batch_prepare)
int build_fine_index(mth_ctx *ctx, const mth_batch_t &dev_batch, int32_t idx_base, uint32_t nq, uint32_t *idx, DevState *st);
// the index the kernels of the call in progress look reads up in (after build_read_index / launch_pdr_lpmd)
inline const uint32_t *idx_ptr(const mth_ctx *ctx) { return ctx->cur_idx ? ctx->cur_idx : ctx->idx.as<uint32_t>(); }
// extent of the fine index of a batch: origin and entries, for any tile width up to 65536
inline void fine_index_extent(const mth_batch_t &b, int32_t &idx_base, uint32_t &nq) {
    const int64_t region_len = (int64_t)b.region_end - b.region_beg;
    const int32_t ext = ((b.max_span + 2 + IDX_Q - 1) / IDX_Q) * IDX_Q;
    idx_base = b.region_beg - ext;
    const int64_t padded = ((region_len + 65535) / 65536) * 65536;
    nq = (uint32_t)((padded + ext) >> IDX_QSHIFT) + 2;
}
// MHL as one tile pass (mth_mhl_tile.hip): candidate-site arrays filled with finished rows and the sites left to the exact walk
int launch_mhl_tile(mth_ctx *ctx, const mth_batch_t &dev_batch, const mth_mhl_params_t &p, uint64_t &bound);
// FDRP + qFDRP at WGBS depth as one tile pass (mth_fdrp_wtile.hip): candidate-site arrays filled with finished rows and the sites left
// to k_fdrp_walk (listed in redo_list / *redo_cnt)
int launch_fdrp_wtile(mth_ctx *ctx, const mth_batch_t &dev_batch, const mth_fdrp_params_t &p, const uint16_t *pair_tab, uint32_t *redo_list, uint32_t *redo_cnt);
int launch_pdr_lpmd(mth_ctx *ctx, const mth_batch_t &dev_batch, const mth_pdr_lpmd_params_t &p,
                    const TileSink *sink = nullptr, bool pipelined = false);

}  // namespace mth
