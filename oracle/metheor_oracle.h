/*
 * metheor_oracle.h -- C ABI of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference
 * algorithm (dohlee/metheor v0.1.9, src/{readutil,pdr,lpmd,mhl,me,pm,fdrp,qfdrp,tag}.rs)
 * used as the parity checker by tests/, __graft_entry__.smoke() and as the
 * `cpu_baseline` leg of bench.py.  Nothing in the product path
 * (metheor_amd/, include/metheor_hip.h) may include, link or call it.
 *
 * Parity pin: every known-answer test the reference holds for this path
 *  is asserted against this oracle in
 * tests/test_oracle_golden.py on the reference's own BAM fixtures.
 * Branches the reference's fixtures never reach (indels, soft clips,
 * reverse strand, multi-contig, flush re-open, reservoir sampling) are
 * restated from source and are "parity unpinned" -- see DESIGN.md.
 */
#ifndef METHEOR_ORACLE_H
#define METHEOR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Raw alignment records as a BAM reader hands them over (one entry per record,
 * stream order).  cigar[] holds BAM-packed ops (len<<4 | op).  A record whose
 * xm_off[i] == xm_off[i+1] has NO XM tag (the reference panics: readutil.rs:46,50). */
typedef struct {
    int64_t         n;
    const int32_t  *tid;
    const int32_t  *pos;        /* 0-based leftmost reference position */
    const uint16_t *flag;
    const uint8_t  *mapq;
    const uint32_t *cigar_off;  /* n+1 */
    const uint32_t *cigar;
    const uint32_t *xm_off;     /* n+1 */
    const char     *xm;
} orc_records_t;

/* Decoded reads == the reference's Vec<BismarkRead> in stream order, as SoA. */
typedef struct orc_reads orc_reads_t;

/* generic result table:  n rows; k positions per row; m counters per row */
typedef struct orc_result orc_result_t;

/* readutil.rs:24-53 + 323-345 (+ filter_isin 87-95 when set_n > 0).
 * set_* is the --cpg-set BED content already mapped to (tid,pos) (readutil.rs:347-374).
 * returns NULL and *err=1 when a record lacks the XM tag. */
orc_reads_t *orc_decode(const orc_records_t *rec, int64_t set_n, const int32_t *set_tid,
                        const int32_t *set_pos, int *err);
/* build the same object from an already decoded SoA (bench.py's synthetic input) */
orc_reads_t *orc_reads_from_soa(int64_t n_reads, const int32_t *tid, const int32_t *start,
                                const int32_t *end, const uint8_t *mapq, const uint8_t *fwd,
                                const uint64_t *cpg_off, const uint32_t *cpg_pos,
                                const uint16_t *cpg_rel);
void     orc_reads_free(orc_reads_t *);
int64_t  orc_reads_n(const orc_reads_t *);
int64_t  orc_reads_ncpg(const orc_reads_t *);
/* copy the SoA out (any pointer may be NULL) */
void     orc_reads_export(const orc_reads_t *, int32_t *tid, int32_t *start, int32_t *end,
                          uint8_t *mapq, uint8_t *fwd, uint64_t *cpg_off, uint32_t *cpg_pos,
                          uint16_t *cpg_rel);

/* per-read primitives (readutil.rs:134-145, 147-164, 166-224, 97-132) for unit tests */
int      orc_read_is_discordant(const orc_reads_t *, int64_t i);
/* counts[l-1] for l=1..cap ; returns max l with non-zero count */
int      orc_read_stretch_info(const orc_reads_t *, int64_t i, int32_t *counts, int cap);
void     orc_read_pairwise(const orc_reads_t *, int64_t i, int32_t min_d, int32_t max_d,
                           int32_t *n_conc, int32_t *n_disc);

/* pdr.rs:119-212.  rows sorted by (tid,pos); val=pdr; cnt = {n_concordant,n_discordant} */
orc_result_t *orc_pdr(const orc_reads_t *, uint32_t min_depth, uint64_t min_cpgs, uint8_t min_qual);

/* lpmd.rs:154-202.  globals[4] = {n_concordant,n_discordant,n_read,n_valid_read} exact (int64);
 * *lpmd uses the reference's wrapping-i32 arithmetic (lpmd.rs:11-12,51-55).
 * returned table (only when want_pairs): rows sorted by ((tid,pos1),(tid,pos2)); k=2 positions,
 * val = per-pair lpmd, cnt = {n_concordant,n_discordant} (lpmd.rs:89-122). */
orc_result_t *orc_lpmd(const orc_reads_t *, int32_t min_d, int32_t max_d, uint8_t min_qual,
                       int want_pairs, int64_t globals[4], float *lpmd);

/* mhl.rs:135-208 ; rows sorted; val = mhl; cnt = {coverage} */
orc_result_t *orc_mhl(const orc_reads_t *, uint32_t min_depth, uint64_t min_cpgs, uint8_t min_qual);

/* me.rs:90-132 + 42-55 / pm.rs:85-128 + 42-51.  k=4 positions; cnt = 16-bin histogram;
 * rows with depth < min_depth dropped (me.rs:82); rows sorted by key (reference order is
 * HashMap-random: compare as sets). which: 0 = ME, 1 = PM */
orc_result_t *orc_quartets(const orc_reads_t *, uint32_t min_depth, uint8_t min_qual, int which);

/* fdrp.rs:176-246 / qfdrp.rs:188-258.  which: 0 = FDRP, 1 = qFDRP.  val = (q)fdrp;
 * cnt = {num_sampled_read}.  The reference's reservoir branch draws from an OS-seeded RNG
 * (fdrp.rs:90) and is not reproducible: here j comes from orc_sample_j(seed,...) instead. */
orc_result_t *orc_fdrp(const orc_reads_t *, uint8_t min_qual, uint64_t min_depth,
                       uint64_t max_depth, int32_t min_overlap, uint64_t seed, int which);
/* the deterministic stand-in for rand::thread_rng().gen_range(1..=total) */
int32_t  orc_sample_j(uint64_t seed, int32_t tid, int32_t pos, int32_t num_total_read);

int64_t         orc_result_n(const orc_result_t *);
int             orc_result_k(const orc_result_t *);
int             orc_result_m(const orc_result_t *);
const int32_t  *orc_result_tid(const orc_result_t *);
const int32_t  *orc_result_pos(const orc_result_t *);  /* n*k */
const float    *orc_result_val(const orc_result_t *);
const uint32_t *orc_result_cnt(const orc_result_t *);  /* n*m */
void            orc_result_free(orc_result_t *);

/* tag.rs:130-384 determine_xm_tag_string for ONE record (oracle/tag_oracle.cpp).  seq = the read's bases as text
 * (rust-htslib seq().as_bytes()), ref_end = htslib's reference_end, contig = the bases [0, chromsize) of the record's
 * contig (tag.rs:423-430 fetch_seq).  Returns the XM string's length (written to out, no terminator), or -1 where the
 * reference panics. */
int64_t  orc_tag_xm(int32_t pos, int32_t ref_end, uint16_t flag, int is_paired_end, const uint32_t *cigar,
                    uint32_t n_cigar, const char *seq, int64_t l_seq, const char *contig, int64_t chromsize,
                    char *out, int64_t out_cap);

/* Rust `{}` of an f32 (shortest round-trip digits, never an exponent, "NaN", "inf").
 * returns strlen; buf must hold >= 64 bytes */
int      orc_format_f32(float v, char *buf);

#ifdef __cplusplus
}
#endif
#endif
