/*
 * metheor_hip.h -- C ABI of the MI355X (gfx950) methylation-heterogeneity engine.
 *
 * This is the drop-in boundary for the per-read CpG-pattern hot path of dohlee/metheor
 * (v0.1.9).  The reference has no FFI of its own: the seam is cut inside each measure's
 * `compute_helper()` AFTER `BismarkRead::new` (host: BAM iteration + XM decode) and BEFORE the
 * hash-map accumulation.  A host (Rust via `extern "C"`, C++, ctypes) decodes records into the
 * flat SoA batch below and calls one `mth_*_accumulate` per batch; the per-measure result
 * getters return exactly what the corresponding `compute_helper()` returns.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every entry point returns an
 * `int` status (MTH_OK == 0, negative on error, never throws); one `mth_ctx_t` per GPU; a ctx
 * is not thread-safe; work is enqueued on the ctx's HIP stream and the getters synchronise.
 * There is NO CPU fallback: without a gfx950 device `mth_ctx_create` fails.
 *
 * Reference interfaces replaced (paths under the reference repo):
 *   src/pdr.rs:119-125    pdr::compute_helper(input,min_depth,min_cpgs,min_qual,cpg_set)
 *                         -> BTreeMap<CpGPosition,(f32,u32,u32)>      => mth_pdr_lpmd_accumulate + mth_pdr_fetch
 *   src/lpmd.rs:154-160   lpmd::compute_helper(input,min_distance,max_distance,min_qual,cpg_set)
 *                         -> LPMDResult                               => mth_pdr_lpmd_accumulate + mth_lpmd_global (+ mth_lpmd_pairs_accumulate/_fetch)
 *   src/mhl.rs:135-141    mhl::compute_helper(...) -> BTreeMap<CpGPosition,f32>        => mth_mhl_accumulate + mth_mhl_fetch (built)
 *   src/me.rs:90-94       me::compute_helper(input,min_qual,cpg_set) -> HashMap<Quartet,QuartetStat>
 *   src/pm.rs:85-89       pm::compute_helper(...)                                      => mth_quartet_accumulate + mth_quartet_fetch (built)
 *   src/fdrp.rs:176-183   fdrp::compute_helper(input,min_qual,min_depth,max_depth,min_overlap,cpg_set)
 *   src/qfdrp.rs:188-195  qfdrp::compute_helper(...) -> BTreeMap<CpGPosition,f32>      => mth_fdrp_accumulate + mth_fdrp_fetch (built)
 *   src/readutil.rs:15-21 BismarkRead {start_pos,end_pos,cpgs:Vec<CpG{relpos,abspos,methylated}>}
 *                                                                                      => mth_batch_t (SoA)
 */
#ifndef METHEOR_HIP_H
#define METHEOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTH_ABI_VERSION 1

typedef struct mth_ctx mth_ctx_t;

enum {
    MTH_OK = 0,
    MTH_ERR_INVALID = -1,   /* bad argument */
    MTH_ERR_HIP = -2,       /* a HIP runtime call failed (mth_last_error has the text) */
    MTH_ERR_NO_DEVICE = -3, /* no gfx950 device: there is no CPU fallback */
    MTH_ERR_UNSORTED = -4,  /* reads of a batch are not sorted by start */
    MTH_ERR_SPAN = -5,      /* a read spans more than batch.max_span */
    MTH_ERR_REOPEN = -6,    /* reserved (was: flush re-open semantics not implemented; they are now) */
    MTH_ERR_RANGE = -7,     /* a CpG of an owned site lies outside what the batch declared */
    MTH_ERR_CAPACITY = -8,  /* an on-chip capacity was exceeded */
    MTH_ERR_STATE = -9,     /* call order violated */
    MTH_ERR_FORMAT = -10,   /* device decode: corrupt BGZF block, malformed BAM record, or a record without XM:Z */
    MTH_ERR_UNALIGNED = -11,/* mth_bgzf_decode: a record straddles two BGZF blocks (decode via the host walk instead) */
    MTH_ERR_RCCL = -12      /* librccl could not be loaded, or an RCCL call failed (mth_last_error has the text) */
};

enum { MTH_MEM_HOST = 0, MTH_MEM_DEVICE = 1, MTH_MEM_PREPARED = 2 /* only in batches made by mth_batch_prepare */ };

/*
 * One batch = the reads of ONE contig that can touch the genomic region [region_beg, region_end),
 * sorted by `read_start` (coordinate-sorted BAM order).  Sites inside the region and reads whose
 * start lies inside it are OWNED by the batch; reads starting outside are halo (they contribute
 * to owned sites only).  A whole contig is region [0, contig_length).
 *
 * Per read  (readutil.rs:15-21; pdr.rs:150 mapq):
 *   read_start/read_end  first / last reference position covered (readutil.rs:25-33), -1 if none
 *   read_mapq            Record::mapq()
 *   read_fwd             1 iff flags in {0,99,147} (readutil.rs:332) -- informational; cpg_pos
 *                        already has the strand rule applied
 *   cpg_off[n_reads+1]   CSR offsets into the per-call arrays
 * Per CpG call (readutil.rs:247-251), in query order:
 *   cpg_pos              abspos (31 bits) | methylated << 31
 *   cpg_rel              relpos = query offset of the call (u8; use cpg_rel16 when reads > 255 bp)
 */
typedef struct {
    int32_t  tid;
    int32_t  region_beg, region_end;
    int32_t  max_span;   /* >= max(read_end - read_start + 1) over the batch */
    uint32_t n_reads;
    uint32_t n_cpgs;
    int32_t  mem;        /* MTH_MEM_HOST or MTH_MEM_DEVICE: where ALL the arrays below live */
    const int32_t  *read_start;
    const int32_t  *read_end;
    const uint8_t  *read_mapq;
    const uint8_t  *read_fwd;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const uint8_t  *cpg_rel;    /* exactly one of cpg_rel / cpg_rel16 is non-NULL */
    const uint16_t *cpg_rel16;
} mth_batch_t;

/* pdr.rs:82-89 + lpmd.rs:125-133 arguments (clap defaults: lib.rs:36-46, 206-214) */
typedef struct {
    uint32_t pdr_min_depth;  /* -d 10 */
    uint32_t pdr_min_cpgs;   /* -p 4  */
    uint8_t  pdr_min_qual;   /* -q 10 */
    uint8_t  lpmd_min_qual;  /* -q 10 */
    uint8_t  want_pdr, want_lpmd;
    int32_t  lpmd_min_distance; /* -m 2  */
    int32_t  lpmd_max_distance; /* -M 16 */
} mth_pdr_lpmd_params_t;

/* ---- prepared batches: one read index per batch, shared by the measures ----------------------------------
 * Every measure's entry point builds the linear read index of its batch ("first read starting at or after q x 32 bp", plus the
 * sortedness check) before its kernels can find a tile's or a site's candidate reads -- once per call: PDR + LPMD, ME / PM, MHL,
 * FDRP / qFDRP and the pairs table over the same batch build it five times (each replaces one pass of the reference's
 * compute_helper over the file: pdr.rs:119, me.rs:90, mhl.rs:135, fdrp.rs:176, lpmd.rs:154).  mth_batch_prepare makes the batch
 * device-resident ONCE (a host batch is copied into buffers the prepared batch owns; a device batch is borrowed) and builds the index
 * ONCE; `*prepared` is then a batch (mem = MTH_MEM_PREPARED, its arrays are the device arrays, read_fwd carries the handle) that
 * every mth_*_accumulate entry point takes in place of the original and gives the same rows for.  Contract: the arrays of the batch
 * given to mth_batch_prepare (device batches) stay untouched until mth_batch_release; release only after a synchronising call has
 * returned for every measure that used it.  An unsorted batch is reported by the first synchronising call after a measure used it. */
int  mth_batch_prepare(mth_ctx_t *ctx, const mth_batch_t *batch, mth_batch_t *prepared);
int  mth_batch_release(mth_ctx_t *ctx, mth_batch_t *prepared);

/* ---- context ------------------------------------------------------------------------- */
int  mth_abi_version(void);
int  mth_ctx_create(int device_id, mth_ctx_t **out);
void mth_ctx_destroy(mth_ctx_t *ctx);
/* enqueue on a caller-owned hipStream_t (e.g. torch's current stream); NULL = ctx's own stream */
int  mth_ctx_set_stream(mth_ctx_t *ctx, void *hip_stream);
int  mth_ctx_sync(mth_ctx_t *ctx);
const char *mth_strerror(int status);
const char *mth_last_error(const mth_ctx_t *ctx);
/* Non-fatal findings of the device-side record decode so far (sticky; as of the latest synchronising call).
 * MTH_NOTE_CIGAR_PAD: a record's CIGAR holds a P (padding) operation.  readutil.rs:28,326 take the read's positions from
 * rust-htslib's reference_positions_full(), whose aligned-pairs iterator is believed to panic on Cigar::Pad (the crate is not part
 * of the reference tree and cannot be checked here); this engine treats P as "no query base, no reference base". */
#define MTH_NOTE_CIGAR_PAD 1u
uint32_t mth_notes(const mth_ctx_t *ctx);
/* forget all accumulated results (a new input file) */
int  mth_reset(mth_ctx_t *ctx);

/* ---- PDR + LPMD, fused single pass (pdr.rs:119-212, lpmd.rs:154-202) --------------------
 * Asynchronous: kernels are enqueued on the ctx stream; data errors (UNSORTED/SPAN/REOPEN/
 * RANGE) are detected on the device and reported by the next synchronising call below.
 * Pipelining (round 4): from the second of a run of consecutive calls with device-resident batches on (mth_reset does not break a
 * run), the engine runs the batches on two internal streams, ordered behind the work already enqueued on the ctx stream at the time
 * of the call; the ctx stream is ordered behind them again by every other entry point of this header (getters, mth_ctx_sync, the
 * other measures, mth_ctx_set_stream, mth_ctx_destroy).  Consequence for a caller that enqueues its OWN work on the stream it gave
 * to mth_ctx_set_stream: the arrays of a device-resident batch must not be overwritten or freed in stream order right behind the
 * call -- only after the next entry point that joins (as they always had to outlive the asynchronous kernels).  MTH_PIPELINE=0 in
 * the environment keeps everything on the ctx stream. */
int  mth_pdr_lpmd_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch,
                             const mth_pdr_lpmd_params_t *params);
/* number of emitted sites so far (synchronises) */
int  mth_pdr_count(mth_ctx_t *ctx, uint64_t *n_sites);
/* copy out the BTreeMap<CpGPosition,(f32,u32,u32)> rows, sorted by (tid,pos) given batches were
 * submitted in that order; any pointer may be NULL; buffers hold >= mth_pdr_count entries */
int  mth_pdr_fetch(mth_ctx_t *ctx, int32_t *tid, int32_t *pos, float *pdr,
                   uint32_t *n_concordant, uint32_t *n_discordant);
/* page-locked host memory for the *_fetch destinations: device-to-host copies into it run at the link's rate (pageable
 * destinations work too, through the runtime's bounce buffers -- ~4x slower on a 700 k-row table).  Freed by
 * mth_result_buffer_free or with the context's process. */
int  mth_result_buffer_alloc(mth_ctx_t *ctx, size_t bytes, void **out);
int  mth_result_buffer_free(mth_ctx_t *ctx, void *p);
/* device pointers of the same columns (valid until the next accumulate/reset) */
int  mth_pdr_device_view(mth_ctx_t *ctx, uint64_t *n_sites, const int32_t **pos,
                         const float **pdr, const uint32_t **n_concordant,
                         const uint32_t **n_discordant);
/* LPMDResult: out[4] = {n_concordant, n_discordant, n_read, n_valid_read} (exact, int64);
 * *lpmd = compute_lpmd() with the reference's wrapping i32 counters (lpmd.rs:11-12,51-55) */
int  mth_lpmd_global(mth_ctx_t *ctx, int64_t out[4], float *lpmd);
/* lpmd.rs:176-179: records that never enter a batch (no contig, no aligned base) still count in n_read, and in n_valid_read when
 * their mapq passes; the caller that left them out of its batches adds their counts here (queued on the context's stream).
 * lpmd.rs:181 builds a BismarkRead for every record whose mapq passes, placed or not: a passing record without XM:Z is the
 * reference's panic -- both decoders refuse such a record before it can be counted here (mth_decode_set_xm_min_mapq) */
int  mth_lpmd_add_unbatched(mth_ctx_t *ctx, uint64_t n_read, uint64_t n_valid_read);
/* multi-GPU: enqueue (on the ctx stream) a copy of the 4 exact int64 counters into a DEVICE buffer
 * the caller owns, ready for an RCCL all-reduce(sum) over ranks; no host synchronisation */
int  mth_lpmd_export_device(mth_ctx_t *ctx, int64_t *dst_device4);
/* compute_lpmd() (lpmd.rs:51-55, wrapping-i32 semantics) from summed integer counters, e.g. the
 * all-reduced ones */
float mth_lpmd_from_counts(int64_t n_concordant, int64_t n_discordant);

/* ---- multi-GPU: the path's one exchange step ------------------------------------------------
 * A sharded run gives every GPU a genomic region; per-site / per-quartet / per-pair rows are owned by region and
 * never exchanged.  Only LPMDResult's four counters (lpmd.rs:11-12: n_concordant, n_discordant, n_read,
 * n_valid_read, accumulated per read at lpmd.rs:176-200 and divided at lpmd.rs:51-55) are genome-wide: they are
 * summed with ONE RCCL all-reduce (ncclInt64 x 4, ncclSum) over xGMI, ordered after the work already enqueued on each
 * context's stream; afterwards mth_lpmd_global() of EVERY context returns the node-wide LPMDResult.  Call it once,
 * after the last mth_pdr_lpmd_accumulate (a second call without new batches is MTH_ERR_STATE: it would add the
 * totals again).  librccl.so.1 is loaded on first use; a single-GPU run never touches it. */
int  mth_device_count(int *n_devices);
/* one process, one context per GPU: collective over ctxs[0..n) called from ONE host thread.  Contexts that share a
 * device are summed on that device and RCCL runs between the distinct devices. */
int  mth_allreduce_lpmd(mth_ctx_t **ctxs, int n);
/* one process per GPU: rank 0 makes the id, the host ships its 128 bytes to every rank (MPI_Bcast,
 * torch.distributed.broadcast, a file), every rank joins, every rank calls the reduce. */
#define MTH_RCCL_ID_BYTES 128
int  mth_rccl_unique_id(void *id128);
int  mth_rccl_init_rank(mth_ctx_t *ctx, const void *id128, int rank, int world);
/* asynchronous: the counters are snapshotted on the context's stream and reduced on a side stream (the context's
 * stream does not wait for the collective; mth_lpmd_global does) */
int  mth_allreduce_lpmd_rank(mth_ctx_t *ctx);

/* ---- LPMD per-pair table (`lpmd --pairs`; lpmd.rs:70-122) -------------------------------------
 * Separate pass over the same batches (the global counters above do not need it). */
typedef struct {
    int32_t min_distance;  /* -m 2  */
    int32_t max_distance;  /* -M 16 */
    uint8_t min_qual;      /* -q 10 */
} mth_lpmd_pairs_params_t;
int  mth_lpmd_pairs_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_lpmd_pairs_params_t *params);
/* rows of print_pair_statistics, sorted by ((tid,cpg1),(tid,cpg2)); lpmd = n_d as f32/(n_c as f32+n_d as f32)
 * (lpmd.rs:111); *n_rows always set, arrays may be NULL */
int  mth_lpmd_pairs_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos1, int32_t *pos2, float *lpmd,
                          uint32_t *n_concordant, uint32_t *n_discordant);

/* ---- contig groups: many contigs in one batch -------------------------------------------------------------------------------------
 * A batch is a coordinate interval of ONE contig (mth_batch_t.tid).  A job over many contigs -- 24 for a human genome, thousands with
 * alt / decoy contigs -- pays every pass's fixed costs (index build, kernel boundaries, a half-filled last wave of workgroups) once
 * per contig.  A GROUP lays several contigs out in one virtual coordinate space: contig k's positions are shifted by voff[k], the gaps
 * between contigs are wider than anything a measure looks across (a read's span, PDR's 150-bp flush margin, FDRP's 201-bp window), and
 * the group is accumulated as ONE batch whose `tid` is the group's handle (<= -2).  No measure looks at a tid: the reference's
 * per-contig behaviour -- a record on a later contig flushes every earlier site (is_before(), readutil.rs:304-310) -- is what a larger
 * virtual position does too.  Every fetch maps a grouped batch's rows back: (handle, virtual position) -> (tids[k], position - voff[k])
 * with k the last contig whose voff is <= the (first) position of the row; row order stays the (tid, pos) order when tids ascend.
 * The reservoir draw of FDRP / qFDRP (keyed by tid and position) is made on the mapped-back site.
 * voff: ascending, voff[0] >= 0, voff[k + 1] >= voff[k] + contig k's extent + max_span + 152 + 202 (checked only for order), all
 * virtual positions below 2^31 - 1.  mth_group_clear forgets all groups (results fetched afterwards would keep their handles).
 * mth_pdr_device_view hands out the virtual positions unmapped. */
int  mth_group_define(mth_ctx_t *ctx, uint32_t n_contigs, const int32_t *tids, const int64_t *voff, int32_t *handle);
int  mth_group_clear(mth_ctx_t *ctx);

/* ---- ME / PM: per-quartet 16-bin epiallele histograms (me.rs:90-132, pm.rs:85-128) ------------
 * One accumulate serves both measures (they share the histogram).  Windows whose consecutive CpGs are >= 2048 bp
 * apart (reference skips, long reads) are aggregated under a 128-bit key on a side path; the calls of a read must be in
 * ascending position order (MTH_ERR_SPAN otherwise).
 * Device-resident batches after a context's first are QUEUED: the call returns without a host sync, and the next call on the
 * context that is not another mth_quartet_accumulate of a device-resident batch (mth_ctx_sync, any fetch, any other accumulate,
 * decode or mth_decoded_batch) reads back whether every queued batch fitted the output it was given -- one that did not, or that
 * has tiles for the global-table path, is replayed there together with the batches behind it.  As for every device-resident
 * batch, the arrays must stay untouched until then.  MTH_QUARTET_QUEUE=0 restores one sync per batch.  mth_lpmd_pairs_accumulate
 * queues the same way (MTH_PAIRS_QUEUE=0). */
typedef struct {
    uint8_t min_qual;   /* -q 10 (lib.rs:66-68, 90-92) */
} mth_quartet_params_t;
int  mth_quartet_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_quartet_params_t *params);
/* HashMap<Quartet,QuartetStat> rows with read depth >= min_depth (the write-time filter of me.rs:82 /
 * pm.rs:77; pass 0 for compute_helper's full map).  *n_rows is always set; the arrays may be NULL
 * (count only) or hold >= *n_rows entries: pos4[n*4] = pos1..pos4, counts16[n*16] = pattern
 * histogram (pattern = 8*m1+4*m2+2*m3+m4), me = compute_me() (me.rs:42-55), pm = compute_pm()
 * (pm.rs:42-51).  Rows come sorted by (batch, pos1..pos4) -- deterministic; the reference's order is HashMap-random.  Like the other measures, batches must be coordinate-sorted (MTH_ERR_UNSORTED) and no CpG call
 * of a read may lie outside [start - 1, start - 1 + max_span] (MTH_ERR_SPAN). */
int  mth_quartet_fetch(mth_ctx_t *ctx, uint32_t min_depth, uint64_t *n_rows, int32_t *tid, int32_t *pos4,
                       uint32_t *counts16, float *me, float *pm);

/* ---- MHL: methylation haplotype load per site (mhl.rs:135-208, 43-73) -------------------------
 * Exact stream semantics of the reference, including its flush / re-open of sites:
 * flush on strict '<' by any read with >= 1 CpG, before the mapq / min_cpgs filters.  Reads with
 * more than 16384 CpGs covering a site are refused (MTH_ERR_CAPACITY); reads with 513..16384 take a walk whose
 * histograms live in 256 MB of HBM scratch.  MTH_ERR_SPAN (a call outside [start - 1, start - 1 + max_span]) is
 * raised for the reads the pass evaluates: every contributing read, and -- on batches of <= 1.8 calls a read, where
 * the flush rule is looked at per finished row -- the reads around a row; otherwise every read's first call. */
typedef struct {
    uint32_t min_depth;  /* -d 10 */
    uint32_t min_cpgs;   /* -p 4  */
    uint8_t  min_qual;   /* -q 10 */
} mth_mhl_params_t;
int  mth_mhl_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_mhl_params_t *params);
/* BTreeMap<CpGPosition,f32> rows sorted by (tid,pos); *n_rows always set, arrays may be NULL;
 * coverage = number of reads of the segment the value comes from */
int  mth_mhl_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos, float *mhl, uint32_t *coverage);

/* ---- FDRP and qFDRP (fdrp.rs:176-246, qfdrp.rs:188-258): both from one pass ---------------------
 * Exact stream semantics (strict '<' flush by reads passing mapq with >= 1 CpG, +-201-bp window
 * drop, fill to max_depth in file order).  Beyond max_depth the reference replaces stored reads at
 * random from an OS-seeded RNG (fdrp.rs:90) -- not reproducible; here the draw is the counter-based
 * hash of (seed, tid, pos, n-th read) shared with the test oracle.  max_depth up to 16384: sites that hold more than 64 reads at
 * once are redone by a second pass with 256 slots in LDS, those beyond 256 by a third with max_depth rows per wave in HBM
 * scratch; max_depth above 16384 is refused with MTH_ERR_CAPACITY (never a truncated result).
 * The reference's own panic on this path (fdrp.rs:70-72, window index -1: a reverse-strand read of 203..403 reference bases that
 * passes min_qual, calls a CpG at its start - 1 and another one at start + 201) is reproduced as MTH_ERR_FORMAT at the next
 * synchronising call, on sorted and unsorted input alike (k_fdrp_guard; batches whose max_span is below 203 skip the pass). */
typedef struct {
    uint64_t min_depth;    /* -d 10 */
    uint64_t seed;         /* reservoir draws; any value */
    uint32_t max_depth;    /* -D 40 */
    int32_t  min_overlap;  /* -l 35 */
    uint8_t  min_qual;     /* -q 10 */
} mth_fdrp_params_t;
int  mth_fdrp_accumulate(mth_ctx_t *ctx, const mth_batch_t *batch, const mth_fdrp_params_t *params);
/* BTreeMap<CpGPosition,f32> rows of BOTH measures, sorted by (tid,pos); n_reads = num_sampled_read of
 * the segment the values come from; *n_rows always set, arrays may be NULL */
int  mth_fdrp_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos, float *fdrp, float *qfdrp,
                    uint32_t *n_reads);

/* ---- BAM record + XM decode on the device (the step before the hot path) ----------
 * Replaces, per record, BismarkRead::new (readutil.rs:24-53: start/end = first/last aligned reference
 * position) and get_cpgs (readutil.rs:323-345: XM z/Z at aligned query offsets, abspos for flags in
 * {0,99,147} else abspos-1; insertions / soft clips skipped), i.e. what bamutil.rs:4-11's record
 * iterator feeds every measure.  Input is the INFLATED BAM record stream (BGZF-decompressed bytes after
 * the header) and the byte offset of each record (rec_off[i] points at record i's block_size field,
 * rec_off[n_rec] = end of the last record); the walk over block_size fields stays on the host.  `mem`
 * says where raw and rec_off live; append != 0 adds the records after those of the previous calls (a file
 * streamed window by window), append == 0 starts over.  The decoded SoA stays in HBM, owned by the context:
 * the arrays of mth_batch_t for ALL records so far, in file order (64-bit offsets, 16-bit relpos).  A record without XM:Z or a malformed record -> MTH_ERR_FORMAT (the reference panics,
 * readutil.rs:46).  The --cpg-set filter (filter_isin, readutil.rs:87-95) is applied when set, see below. */
typedef struct {
    uint64_t n_reads, n_cpgs;
    const int32_t  *tid, *start, *end;   /* device pointers */
    const uint8_t  *mapq, *fwd;
    const uint64_t *cpg_off;             /* n_reads + 1 */
    const uint32_t *cpg_pos;             /* abspos | methylated << 31 */
    const uint16_t *cpg_rel;
} mth_decoded_t;
int  mth_decode_records(mth_ctx_t *ctx, const void *raw, uint64_t n_bytes, const uint64_t *rec_off, uint64_t n_rec,
                        int mem, int append, mth_decoded_t *out);
/* --cpg-set (get_target_cpgs / filter_isin, readutil.rs:347-374, 87-95): keep only the calls whose (tid, pos) is in the set;
 * relpos of the kept calls is unchanged.  keys_sorted = strictly ascending (uint64)tid << 32 | pos, host memory.
 * enabled != 0 with n_keys == 0 is the empty set (every call dropped, as HashSet::contains on an empty set);
 * enabled == 0 removes the filter.  Applies to the following mth_decode_records / mth_bgzf_decode calls. */
int  mth_decode_set_cpg_filter(mth_ctx_t *ctx, const uint64_t *keys_sorted, uint64_t n_keys, int enabled);
/* lpmd applies its mapq filter BEFORE BismarkRead::new (lpmd.rs:176-181): a record without XM:Z that the filter skips never
 * panics there.  Records without XM:Z whose mapq is below min_mapq are decoded with zero calls instead of raising
 * MTH_ERR_FORMAT (default 0: every such record is an error, as in the other six measures).  Applies to the following
 * mth_decode_records / mth_bgzf_decode calls. */
int  mth_decode_set_xm_min_mapq(mth_ctx_t *ctx, uint32_t min_mapq);
/* the inflate step alone: inflated bytes of the given blocks, concatenated, copied to dst_host (may be NULL); *n_out = size.
 * The 4 bytes after each payload (the gzip trailer's CRC32) must lie inside the file bytes given: they are verified. */
int  mth_bgzf_inflate(mth_ctx_t *ctx, const void *file, uint64_t n_bytes, const uint64_t *coff, const uint32_t *csize,
                      const uint32_t *isize, uint64_t n_blocks, void *dst_host, uint64_t *n_out);
/* The whole step on the device: BGZF inflate (one wave per block, RFC 1951 stored / fixed / dynamic blocks, checked
 * against ISIZE), record boundaries (one thread per block; htslib-family writers never let a record straddle a BGZF
 * block, which is verified -- MTH_ERR_UNALIGNED otherwise, nothing is appended then), record + XM decode as above.
 * `file` = n_bytes of the BAM file in HOST memory covering the blocks; per block: coff = offset of its DEFLATE payload
 * inside `file`, csize = payload bytes, isize = inflated bytes (blocks with isize 0 omitted); first_byte = offset in
 * the inflated stream of these blocks where the records start (the header's uncompressed size for the first call,
 * 0 afterwards).  Replaces bamutil.rs:4-11 (htslib's reader) + readutil.rs:24-53, 323-345 for a coordinate-sorted
 * Bismark BAM.  Every inflated block is checked against its ISIZE and its CRC32 (as htslib does); a mismatch is
 * MTH_ERR_FORMAT. */
int  mth_bgzf_decode(mth_ctx_t *ctx, const void *file, uint64_t n_bytes, const uint64_t *coff, const uint32_t *csize,
                     const uint32_t *isize, uint64_t n_blocks, uint64_t first_byte, int append, mth_decoded_t *out);
/* Overlap of the NEXT chunk's host-to-device copy with the current chunk's kernels: announces file[0, n_bytes) (host memory
 * that stays valid) as the chunk after the one the next mth_bgzf_decode / mth_bgzf_inflate call is given.  That call starts a
 * helper thread which copies the announced bytes into a second staging buffer on a side stream while its own kernels run;
 * the following call, given the SAME pointer and size, uses them instead of copying.  n_bytes = 0 withdraws the announcement. */
int  mth_bgzf_stage(mth_ctx_t *ctx, const void *file, uint64_t n_bytes);
/* size the decoded arrays for n_reads / n_cpgs in total (what is decoded so far is kept): spares the appending calls their
 * reallocations when the caller can estimate the file's totals */
int  mth_decode_reserve(mth_ctx_t *ctx, uint64_t n_reads, uint64_t n_cpgs);
/* copy the decoded arrays to the host (any pointer may be NULL) */
int  mth_decoded_fetch(mth_ctx_t *ctx, int32_t *tid, int32_t *start, int32_t *end, uint8_t *mapq, uint8_t *fwd,
                       uint64_t *cpg_off, uint32_t *cpg_pos, uint16_t *cpg_rel);
/* the decoded stream as runs of equal tid, in file order (a coordinate-sorted file has one run per contig): up to `cap`
 * runs are returned, *n_runs is the true number; *flags bit0 = some read has no aligned base (start < 0), bit1 = some
 * record has no contig (tid < 0) -- such records cannot enter a batch as they are --, bit2 = some read starts before its
 * predecessor on the same contig (the input is not coordinate-sorted) */
int  mth_decoded_contigs(mth_ctx_t *ctx, uint32_t cap, int32_t *tids, uint64_t *read_beg, uint64_t *read_end,
                         uint32_t *n_runs, uint32_t *flags);
/* Contig groups for the decoded stream (see "contig groups" above): the runs of mth_decoded_contigs (n_contigs of them, tids strictly
 * ascending, consecutive read ranges) are packed greedily into groups of at most 2^31 - 2^22 virtual positions; the reads' and calls'
 * positions of every group with more than one contig are shifted IN PLACE on the device (mth_decoded_fetch then returns the shifted
 * values) and the groups are registered.  Returns per group g < *n_groups: the contigs [first_contig[g], first_contig[g + 1]) and
 * batch_tid[g] = the group's handle, or the contig's own tid for a group of one; the caller batches a group with
 * mth_decoded_batch(ctx, read_beg[first_contig[g]], read_end[first_contig[g + 1] - 1], batch_tid[g], 0, -1, &b).
 * *n_groups = 0: nothing was changed (tids not ascending, a call at position -1, or a previous grouping in place). */
int  mth_decoded_group(mth_ctx_t *ctx, uint32_t n_contigs, const int32_t *tids, const uint64_t *read_beg, const uint64_t *read_end,
                       uint32_t *n_groups, uint32_t *first_contig /* [n_contigs + 1] */, int32_t *batch_tid /* [n_contigs] */);
/* re-order the decoded stream by (tid, start), stably, on the device -- for the measures whose result does not depend on the
 * record order (lpmd.rs:175-200, me.rs:106-125, pm.rs:101-121): an input that is not coordinate-sorted or not grouped by contig
 * can then be batched like a sorted one.  Every record must have a contig and an aligned base (flags bit0 / bit1 clear). */
int  mth_decoded_sort(mth_ctx_t *ctx);

/* ---- PDR, MHL, FDRP and qFDRP of a decoded stream that is NOT coordinate-sorted (mth_fileorder.hip) ----------------------------
 * The reference iterates the records in file order and never checks it (pdr.rs:139, mhl.rs:155, fdrp.rs:197, qfdrp.rs:209); these
 * four measures finalise a site when a later record's first CpG lies beyond it (pdr.rs:160-177 margin 150, passing records;
 * mhl.rs:162-173 any record with a CpG; fdrp.rs:212-223 passing records) and a site that is called again starts over, the last
 * segment with enough reads winning.  On a sorted file the per-contig batches compute that; for any other order this call replays
 * the stream itself, over the WHOLE decoded stream (mth_decode_records / mth_bgzf_decode, --cpg-set filter included) in the order
 * it was decoded: rows sorted by (tid, pos) as the reference's BTreeMap gives them.
 *   MTH_FO_PDR  v0 = pdr, c0 = n_concordant, c1 = n_discordant        (min_depth, min_cpgs, min_qual)
 *   MTH_FO_MHL  v0 = mhl, c0 = coverage                                (min_depth, min_cpgs, min_qual)
 *   MTH_FO_FDRP v0 = fdrp, v1 = qfdrp, c0 = stored reads               (min_depth, min_qual, max_depth <= 16384, min_overlap, seed)
 * Synchronous.  MTH_ERR_CAPACITY beyond 2^31 records / calls, MHL reads with > 1024 CpGs, --max-depth > 16384 (as on the sorted
 * path; up to 256 stored reads a site's slots live in LDS, beyond that in a per-wave row of HBM scratch). */
enum { MTH_FO_PDR = 0, MTH_FO_MHL = 1, MTH_FO_FDRP = 2 };
typedef struct {
    int32_t  measure;
    uint32_t min_depth, min_cpgs, max_depth;
    int32_t  min_overlap;
    uint8_t  min_qual;
    uint64_t seed;
} mth_fileorder_params_t;
int  mth_fileorder_run(mth_ctx_t *ctx, const mth_fileorder_params_t *params);
int  mth_fileorder_fetch(mth_ctx_t *ctx, uint64_t *n_rows, int32_t *tid, int32_t *pos, float *v0, float *v1, uint32_t *c0, uint32_t *c1);
/* reads [read_beg, read_end) of the decoded stream -- ONE contig's reads (or a region slice of them), or one contig GROUP's after
 * mth_decoded_group (tid = the group's handle, region_beg 0, region_end -1) -- as a
 * device-resident batch for the accumulate calls: 32-bit offsets rebased on the device, max_span reduced on
 * the device.  region_end < 0 = up to the last position these reads cover (reduced on the device too): what a
 * whole-contig batch should pass -- the reference emits every site it sees, also past the header's LN.
 * Valid until the next mth_decoded_batch / mth_decode_records call. */
int  mth_decoded_batch(mth_ctx_t *ctx, uint64_t read_beg, uint64_t read_end, int32_t tid, int32_t region_beg,
                       int32_t region_end, mth_batch_t *batch);

/* ---- `metheor tag`: the Bismark XM string of every record from the read's bases and the genome ----------
 * Replaces src/tag.rs:130-384 determine_xm_tag_string (per record) and the genome side of run(), tag.rs:419-431.
 * mth_tag_set_genome: n_refs contigs in header (tid) order; ref_len[t] = the header's LN (tag.rs:60-72), seq[t] / seq_len[t] =
 * the bases the FASTA gave for positions [0, seq_len[t]) (any case; HOST memory; copied).  A contig the FASTA lacks is the
 * caller's error to raise (tag.rs:427 expect).
 * mth_tag_records: a window of BAM records as mth_decode_records takes it (raw bytes + n_rec + 1 offsets); is_paired_end =
 * bamutil.rs:27-37 (flag 0x1 of the file's first record).  On return record i's string is xm[xm_off[i] .. xm_off[i] + xm_len[i])
 * (host memory owned by the context, valid until the next call).  MTH_ERR_FORMAT where the reference panics. */
typedef struct {
    uint64_t        n_records;
    const uint64_t *xm_off;     /* n_records + 1 slot starts */
    const uint32_t *xm_len;
    const char     *xm;
} mth_tag_out_t;
int  mth_tag_set_genome(mth_ctx_t *ctx, int32_t n_refs, const int64_t *ref_len, const uint8_t *const *seq, const int64_t *seq_len);
int  mth_tag_records(mth_ctx_t *ctx, const void *raw, uint64_t n_bytes, const uint64_t *rec_off, uint64_t n_rec, int mem,
                     int is_paired_end, mth_tag_out_t *out);

/* ---- measurement hooks (bench.py's roofline leg) -------------------------------------- */
/* when enabled, every kernel launch is bracketed by hipEvents on the launch stream */
int  mth_timing_enable(mth_ctx_t *ctx, int on);
int  mth_timing_reset(mth_ctx_t *ctx);
/* synchronises; name is one of mth_timing_kernel_name(i); avg over recorded launches */
int  mth_timing_get(mth_ctx_t *ctx, const char *kernel, double *avg_ms, uint64_t *launches);
int  mth_timing_num_kernels(void);
const char *mth_timing_kernel_name(int i);

#ifdef __cplusplus
}
#endif
#endif
