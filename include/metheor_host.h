/*
 * metheor_host.h -- C ABI of the HOST side of the engine (no GPU needed): BGZF/BAM reading and the
 * XM-tag decode that produce the SoA batches of metheor_hip.h.
 *
 * Reference interfaces replaced (paths under the reference repo):
 *   src/bamutil.rs:4-11    get_reader(input) -> bam::Reader (rust-htslib / C htslib; BAM or SAM text) => mth_host_open
 *   src/bamutil.rs:13-25   get_header / tid2chrom / chrom2tid                           => mth_host_n_refs / _ref_name / _ref_len / _ref_tid
 *   src/readutil.rs:24-53  BismarkRead::new(&Record)   (start/end, XM tag required)
 *   src/readutil.rs:323-345 get_cpgs (z/Z only, aligned bases only, strand rule flags in {0,99,147})
 *   src/readutil.rs:87-95, 347-374  filter_isin / get_target_cpgs (--cpg-set BED: col0 chrom, col1 start)
 *                                                                                        => mth_host_decode
 * Error behaviour mirrors the reference's panics as status codes + message (mth_host_last_error):
 *   missing file        "Error opening BAM file. file not found: <path>"     (bamutil.rs:7-9, tests/pdr-cli.rs:26-31)
 *   not a BAM           "Error opening BAM file. ..."                         (tests/cli_error_handling.rs:214-227)
 *   record without XM   "Error reading XM tag in BAM record. Make sure the reads are aligned using Bismark!" (readutil.rs:46,50)
 *   bad --cpg-set       "Could not read target CpG file."                     (readutil.rs:356)
 */
#ifndef METHEOR_HOST_H
#define METHEOR_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mth_host mth_host_t;

enum {
    MTH_HOST_OK = 0,
    MTH_HOST_ERR_OPEN = -101,     /* file missing / unreadable / not BGZF-BAM */
    MTH_HOST_ERR_FORMAT = -102,   /* truncated or corrupt BAM */
    MTH_HOST_ERR_XM = -103,       /* a record has no XM:Z tag */
    MTH_HOST_ERR_CPGSET = -104,   /* --cpg-set file unreadable / unknown contig / bad number */
    MTH_HOST_ERR_INVALID = -105,
    MTH_HOST_ERR_CONSUMER = -106  /* the window consumer of mth_host_decode_stream returned non-zero */
};

/* open a BAM file and read its header */
int  mth_host_open(const char *path, mth_host_t **out, char *errbuf, int errbuf_len);
void mth_host_close(mth_host_t *h);
const char *mth_host_last_error(const mth_host_t *h);
/* non-fatal findings of the host-side record decode so far (process-wide, sticky): bit 0 = a record's CIGAR holds a P (padding)
 * operation -- see MTH_NOTE_CIGAR_PAD in metheor_hip.h (rust-htslib is believed to panic there; this decoder takes P as "no base") */
uint32_t mth_host_notes(void);
int  mth_host_n_refs(const mth_host_t *h);
const char *mth_host_ref_name(const mth_host_t *h, int tid);
int64_t mth_host_ref_len(const mth_host_t *h, int tid);
int  mth_host_ref_tid(const mth_host_t *h, const char *name);   /* -1 if unknown */

/* the path the loaders read: the input itself for a BAM, the in-memory BAM a SAM text input was converted into otherwise */
const char *mth_host_path(const mth_host_t *h);
/* the header text as the file holds it (what bam::Header::from_template copies into a writer, tag.rs:403-406) */
const char *mth_host_header_text(const mth_host_t *h, uint64_t *n_bytes);
/* One BAM record (the bytes after its block_size field) as a SAM line ending in '\n' -- what bam::Writer with Format::Sam
 * prints (tag.rs:405-441); xm != NULL appends the field XM:Z:<xm> last, as push_aux does (tag.rs:437).  Returns the line's
 * length (copied to buf when it fits cap), or a negative status for a malformed record. */
int64_t mth_host_sam_format(const mth_host_t *h, const uint8_t *rec, uint32_t rec_len, const char *xm, uint32_t xm_len,
                            char *buf, int64_t cap);
/* FASTA access for `tag` (faidx::Reader::from_path + fetch_seq, tag.rs:412-431): the .fai next to the file when present,
 * otherwise one scan of the file.  A missing file reports "file not found: <path>" (tests/tag-cli.rs:42-58).
 * fetch: the bases [0, end_incl] of `name` clipped to the sequence (faidx_fetch_seq semantics); valid until the next fetch. */
typedef struct mth_fasta mth_fasta_t;
int  mth_host_fasta_open(const char *path, mth_fasta_t **out, char *errbuf, int errbuf_len);
void mth_host_fasta_close(mth_fasta_t *f);
const char *mth_host_fasta_last_error(const mth_fasta_t *f);
int  mth_host_fasta_fetch(mth_fasta_t *f, const char *name, int64_t end_incl, const uint8_t **seq, int64_t *len);

/* Decode ALL remaining records of the file into one SoA held by the handle, in file order.
 * cpg_set_path may be NULL.  After success the arrays below are valid until close/decode. */
/* lpmd.rs:176-181 filters on mapq BEFORE BismarkRead::new: records without XM:Z below min_mapq are decoded with zero calls
 * instead of failing the decode (default 0: every record without XM:Z is an error, readutil.rs:46) */
int  mth_host_set_xm_min_mapq(mth_host_t *h, int min_mapq);
int  mth_host_decode(mth_host_t *h, const char *cpg_set_path);
/* Streaming alternative to mth_host_decode for a device-side record decode (mth_decode_records in metheor_hip.h):
 * BGZF inflate (host threads) and the sequential walk over the records' block_size fields (bamutil.rs:4-11's
 * record iterator), but NO record / XM parsing on the host.  The consumer is called once per inflated window
 * (<= METHEOR_DECODE_WINDOW_MB, default 512 MiB) with the window's bytes and the byte offset of each complete
 * record in it (rec_off[n_rec] = end of the last record; a record straddling two windows is carried into the
 * next one).  buf / rec_off are only valid during the call.  A non-zero return aborts. */
typedef int (*mth_host_window_cb)(void *user, const uint8_t *buf, const uint64_t *rec_off, uint64_t n_rec);
int  mth_host_decode_stream(mth_host_t *h, mth_host_window_cb cb, void *user);
/* The --cpg-set BED file (readutil.rs:347-374: col0 chrom -- must be in the header --, col1 start) as strictly ascending keys
 * (uint64)tid << 32 | start, for mth_decode_set_cpg_filter.  Same errors as mth_host_decode's cpg_set_path.  Valid until close. */
int  mth_host_cpg_set_keys(mth_host_t *h, const char *cpg_set_path, const uint64_t **keys, uint64_t *n_keys);
/* For a consumer that inflates on its own (mth_bgzf_decode in metheor_hip.h): the file mapped read-only and the table
 * of its BGZF blocks that hold data -- payload offset in the file, payload bytes, inflated bytes -- plus the
 * uncompressed size of the BAM header (where the records start in the inflated stream).  Valid until close. */
typedef struct {
    const uint8_t  *file;
    uint64_t        file_bytes;
    const uint64_t *coff;
    const uint32_t *csize, *isize;
    uint64_t        n_blocks;
    uint64_t        header_bytes;
} mth_host_bgzf_t;
int  mth_host_bgzf_blocks(mth_host_t *h, mth_host_bgzf_t *out);
/* Byte-range sharding of a coordinate-sorted BAM whose records do not straddle BGZF blocks ("lets 8 GPUs'
 * host threads read disjoint BAM ranges instead of one streaming router" -- done here WITHOUT a .bai: BGZF blocks are
 * self-contained, so a shard can start at any block; the reference never uses the .bai files its fixtures ship).
 * The data blocks are cut into `world` runs of about equal compressed size.  Shard `rank` owns the genomic interval from
 * the first record of its run to the first record of the next run, in (tid, pos) order (sites / quartets / pairs are owned
 * by position, reads by their start: what mth_batch_t's region_beg / region_end mean).  To complete its sites it also
 * loads the blocks before its run that can hold reads starting within halo_bp of its interval (same contig) and the
 * blocks after it that hold reads starting exactly at the interval's end (a reverse read reports start - 1).
 * tid_beg = -1: from the start of the file; tid_end = INT32_MAX: to its end.  block_beg == block_end: nothing to load.
 * MTH_HOST_ERR_FORMAT if a run does not start at a record boundary (records straddle blocks). */
typedef struct {
    uint64_t block_beg, block_end;   /* BGZF blocks to inflate and decode, halo blocks included */
    uint64_t first_byte;             /* inflated bytes of block_beg before the first record (the BAM header, shard 0) */
    int32_t  tid_beg, pos_beg;       /* owned interval [(tid_beg, pos_beg), (tid_end, pos_end)) */
    int32_t  tid_end, pos_end;
} mth_host_shard_t;
int  mth_host_plan_shard(mth_host_t *h, int rank, int world, int64_t halo_bp, mth_host_shard_t *out);
/* The same kind of plan from the BAM index (.bai, SAM spec 5.2; SURVEY 8(f).2 -- the reference's fixtures ship
 * tests/test{1..6}.bam.bai, its reader bamutil.rs:4-11 never opens them): the BGZF blocks that can hold records overlapping
 * [beg - halo_bp, end] of reference `tid`, and the owned interval [(tid, beg), (tid, end)).  bai_path NULL: <bam>.bai, then
 * <bam without .bam>.bai.  MTH_HOST_ERR_OPEN without an index, MTH_HOST_ERR_FORMAT for one that is not this file's. */
int  mth_host_plan_region(mth_host_t *h, const char *bai_path, int32_t tid, int32_t beg, int32_t end, int64_t halo_bp,
                          mth_host_shard_t *out);
int64_t mth_host_n_reads(const mth_host_t *h);
int64_t mth_host_n_cpgs(const mth_host_t *h);
const int32_t  *mth_host_read_tid(const mth_host_t *h);
const int32_t  *mth_host_read_start(const mth_host_t *h);
const int32_t  *mth_host_read_end(const mth_host_t *h);
const uint8_t  *mth_host_read_mapq(const mth_host_t *h);
const uint8_t  *mth_host_read_fwd(const mth_host_t *h);
const uint64_t *mth_host_cpg_off(const mth_host_t *h);      /* n_reads + 1 */
const uint32_t *mth_host_cpg_pos(const mth_host_t *h);      /* abspos | methylated << 31 */
const uint16_t *mth_host_cpg_rel(const mth_host_t *h);

/* Tooling: write a synthetic single-contig Bismark-style BAM (read_len 'M' reads, XM:Z from the SoA's
 * calls, random sequence/quality bytes) -- the seeded generator used by the end-to-end benchmarks.
 * cpg_off/cpg_rel/cpg_pos as produced by mth_host_decode (cpg_pos bit 31 = methylated). */
int  mth_host_write_synthetic_bam(const char *path, const char *contig, int64_t contig_len, int64_t n_reads,
                                  int32_t read_len, const int32_t *start, const uint8_t *fwd, const uint8_t *mapq,
                                  const uint64_t *cpg_off, const uint16_t *cpg_rel, const uint32_t *cpg_pos,
                                  uint64_t seed, int nthreads);
/* the same over several contigs: tid[i] = contig of read i (reads grouped by contig, coordinate-sorted inside) */
int  mth_host_write_synthetic_bam_multi(const char *path, int32_t n_contigs, const char *const *contigs, const int64_t *contig_lens,
                                        int64_t n_reads, int32_t read_len, const int32_t *tid, const int32_t *start,
                                        const uint8_t *fwd, const uint8_t *mapq, const uint64_t *cpg_off,
                                        const uint16_t *cpg_rel, const uint32_t *cpg_pos, uint64_t seed, int nthreads);
/* the same base_reads reads of one contig on n_copies contigs (record i = read i % base_reads on contig i / base_reads): a large
 * test / bench file without a large SoA */
int  mth_host_write_synthetic_bam_repeat(const char *path, int32_t n_copies, const char *const *contigs, const int64_t *contig_lens,
                                         int64_t base_reads, int32_t read_len, const int32_t *start, const uint8_t *fwd,
                                         const uint8_t *mapq, const uint64_t *cpg_off, const uint16_t *cpg_rel,
                                         const uint32_t *cpg_pos, uint64_t seed, int nthreads);

/* Rust `{}` of an f32 (shortest round-trip digits, positional, "NaN"/"inf"); buf >= 64 bytes */
int  mth_host_format_f32(float v, char *buf);

#ifdef __cplusplus
}
#endif
#endif
