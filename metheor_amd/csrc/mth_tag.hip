// mth_tag.hip -- the `tag` subcommand's per-record function on the device.
//
// Replaces src/tag.rs:130-384 `determine_xm_tag_string`: the Bismark XM string of an alignment from the read's
// bases and the reference genome.  The reference builds, per record, two gapped strings (read / reference columns
// over the CIGAR's M, I and D runs -- nothing else is walked, tag.rs:185-237 -- with two flank columns on each side),
// reverse-complements them for reads from the G->A strand, and classifies every column whose reference base is C by
// the next two read-aligned reference bases (CG / CHG / CHH / unknown), skipping columns a deletion removed.
//
// Here: the genome is resident in HBM (all contigs concatenated, uploaded once); one thread per record, two kernels
// (columns per record -> 64-bit scan -> columns + letters).  The gapped columns of a record live in a scratch slice
// the thread writes and reads back itself (a column's letter looks at up to two later -- for the reverse complement,
// earlier -- read-aligned columns, further away behind deletions).  The letter of a column does not depend on the
// order the columns are visited in, so both strands are emitted in ascending column order, which is what the
// reference's final `rev()` (tag.rs:386-389) produces.  Where the reference panics (a base the complement table does
// not hold, tag.rs:24; an alignment outside the contig, tag.rs:158-170; no second context base, tag.rs:297; an
// unplaced record, tag.rs:155) the batch fails with MTH_ERR_FORMAT.
// Not the hot path: byte-walking threads (HBM/L2-latency bound like k_decode); no MFMA.
#include "mth_ctx.h"

namespace mth {

struct TagArgs {
    const uint8_t *raw;
    const uint64_t *off;              // n_rec + 1 byte offsets of the records
    uint32_t n_rec;
    const uint8_t *genome;            // contigs back to back, as fetched (any case)
    const uint64_t *g_off;            // n_refs + 1
    const int64_t *g_ln;              // header LN per tid (tag.rs:60-72 tid2size)
    int32_t n_refs;
    int32_t paired;                   // bamutil.rs:27-37 is_paired_end
    uint32_t *ncol;                   // pass 1 out: columns per record (flanks included)
    const unsigned long long *col_off;   // pass 2 in: exclusive scan of ncol
    uint8_t *cols;                    // pass 2 scratch: read column chars at [col_off, +ncol), reference chars at total + the same
    unsigned long long total;
    uint8_t *xm;                      // pass 2 out: a record's letters start at col_off[i]
    uint32_t *xm_len;
    uint32_t *err;
};

__device__ __forceinline__ uint32_t tg_u32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t tg_u16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint8_t tg_up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
// tag.rs:74-96; 0 = not in the table (the reference's HashMap index panics)
__device__ __forceinline__ uint8_t tg_comp(uint8_t c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N';
        case 'M': return 'K'; case 'R': return 'Y'; case 'W': return 'W'; case 'S': return 'S'; case 'Y': return 'R';
        case 'K': return 'M'; case 'V': return 'B'; case 'H': return 'D'; case 'D': return 'H'; case 'B': return 'V';
        case '-': return '-';
        default: return 0;
    }
}
__device__ __forceinline__ bool tg_h(uint8_t c) { return c == 'A' || c == 'T' || c == 'C'; }
// the context branches of tag.rs:298-337 / 343-383 for the context "C" c1 [c2] (n = 2 or 3 characters); 0 = no letter is pushed
__device__ __forceinline__ uint8_t tg_letter(uint8_t c1, uint8_t c2, int n, uint8_t read_base) {
    uint8_t hi, lo;
    if (c1 == 'G') { hi = 'Z'; lo = 'z'; }
    else if (n == 3 && tg_h(c1) && c2 == 'G') { hi = 'X'; lo = 'x'; }                          // CAG CTG CCG
    else if (n == 3 && tg_h(c1) && tg_h(c2)) { hi = 'H'; lo = 'h'; }                            // C[ATC][ATC]
    else if (c1 == '-' || c1 == 'N' || (n == 3 && (c2 == '-' || c2 == 'N'))) { hi = 'U'; lo = 'u'; }
    else return 0;
    return read_base == 'C' ? hi : (read_base == 'T' ? lo : (uint8_t)'.');
}

struct TagRec {
    int32_t tid, pos;
    uint32_t flag, n_cigar, l_seq;
    const uint8_t *cigar, *seq;
    bool bad;
};
__device__ __forceinline__ TagRec tg_parse(const TagArgs &a, uint32_t i) {
    TagRec r{};
    const uint64_t o0 = a.off[i], o1 = a.off[i + 1];
    const uint8_t *p = a.raw + o0 + 4;
    const uint32_t len = (uint32_t)(o1 - o0 - 4);
    r.bad = o1 < o0 + 4 + 32 || tg_u32(a.raw + o0) != len;
    if (r.bad) return r;
    r.tid = (int32_t)tg_u32(p); r.pos = (int32_t)tg_u32(p + 4);
    const uint32_t l_read_name = p[8];
    r.n_cigar = tg_u16(p + 12); r.flag = tg_u16(p + 14); r.l_seq = tg_u32(p + 16);
    const uint64_t o_cigar = 32ull + l_read_name;
    const uint64_t o_seq = o_cigar + 4ull * r.n_cigar;
    if (o_seq + ((uint64_t)r.l_seq + 1) / 2 + r.l_seq > len) { r.bad = true; return r; }
    r.cigar = p + o_cigar; r.seq = p + o_seq;
    return r;
}

// tag.rs:245-389 with a read column string (nr characters) SHORTER than the reference one (ng): the two target strings are
// built and indexed independently, exactly as the reference does; any index past a string's end is its panic (-1).
// Rare (the read ran out before its CIGAR did), so: one plain sequential routine, kept out of line.
__device__ __noinline__ int64_t tg_xm_unaligned(const uint8_t *R, uint32_t nr, const uint8_t *G, uint32_t ng, bool rc, uint8_t *out) {
    const uint32_t tlen = nr - 2u, glen = ng - 2u;               // target_read_seq.len(), target_ref_seq.len()
    if (rc) {                                                    // reverse_complement() maps every character (tag.rs:19-25)
        for (uint32_t t = 0; t < tlen; ++t) if (!tg_comp(R[t])) return -1;
        for (uint32_t t = 0; t < glen; ++t) if (!tg_comp(G[t])) return -1;
    }
    bool ok = true;
    auto rd = [&](uint32_t idx) -> uint8_t { if (idx >= tlen) { ok = false; return 0; } return rc ? tg_comp(R[tlen - 1u - idx]) : R[2u + idx]; };
    auto rf = [&](uint32_t idx) -> uint8_t { if (idx >= glen) { ok = false; return 0; } return rc ? tg_comp(G[glen - 1u - idx]) : G[2u + idx]; };
    uint32_t nx = 0;
    for (uint32_t idx = 0; idx + 2u < tlen && ok; ++idx) {       // tag.rs:265
        const uint8_t r0 = rd(idx);
        if (r0 == '-') continue;
        if (r0 == 'N') { out[nx++] = '.'; continue; }
        if (rf(idx) != 'C') { if (ok) out[nx++] = '.'; continue; }
        uint8_t c1 = 0, c2 = 0;
        int n = 1;
        if ((rd(idx + 1u) == '-' || rd(idx + 2u) == '-') && idx != tlen - 3u && idx != tlen - 4u) {      // tag.rs:271-296
            for (uint32_t k = 1; n != 3 && idx + k <= tlen - 1u; ++k) {
                if (rd(idx + k) != '-') { const uint8_t g = rf(idx + k); if (n == 1) c1 = g; else c2 = g; ++n; }
            }
            if (n < 2) ok = false;                               // tmp_target_ref_seq[1]
        } else {                                                 // tag.rs:340-383: skip(idx).take(3) may come up short; [idx + 1] may not
            c1 = rf(idx + 1u);
            n = 2;
            if (idx + 2u < glen) { c2 = rf(idx + 2u); n = 3; }
        }
        if (!ok) break;
        const uint8_t l = tg_letter(c1, c2, n, r0);
        if (l) out[nx++] = l;
    }
    if (!ok) return -1;
    if (rc) for (uint32_t x = 0, y = nx; x + 1u < y; ++x) { --y; const uint8_t t = out[x]; out[x] = out[y]; out[y] = t; }   // tag.rs:386-389
    return (int64_t)nx;
}

__global__ __launch_bounds__(256) void k_tag_count(const TagArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_rec) return;
    const TagRec r = tg_parse(a, i);
    uint64_t n = 4;
    if (!r.bad) {
        for (uint32_t k = 0; k < r.n_cigar; ++k) {
            const uint32_t c = tg_u32(r.cigar + 4 * k), op = c & 15u;
            if (op <= 2u) n += c >> 4;                           // M, I, D (tag.rs:188-235)
        }
    }
    if (r.bad || n >= (1ull << 31)) { atomicOr(a.err, (uint32_t)ERRB_FORMAT); n = 4; }
    a.ncol[i] = (uint32_t)n;
}

__global__ __launch_bounds__(256) void k_tag_xm(const TagArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_rec) return;
    a.xm_len[i] = 0;
    const TagRec r = tg_parse(a, i);
    if (r.bad) return;                                           // reported by the count pass
    const uint32_t ncol = a.ncol[i];
    uint8_t *R = a.cols + a.col_off[i], *G = R + a.total;
    uint32_t fail = 0;
    // tag.rs:136-144
    if (r.tid < 0 || r.tid >= a.n_refs) { atomicOr(a.err, (uint32_t)ERRB_TAGPANIC); return; }   // tid2size[&tid] panics
    int64_t reflen = 0, qwalk = 0;
    for (uint32_t k = 0; k < r.n_cigar; ++k) {
        const uint32_t c = tg_u32(r.cigar + 4 * k), op = c & 15u;
        if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) reflen += c >> 4;        // htslib bam_endpos
        if (op == 0u || op == 1u) qwalk += c >> 4;
    }
    const int64_t start = r.pos, end = r.pos + (reflen ? reflen : 1);
    const bool is_rev = r.flag & 16u, first = r.flag & 64u, last = r.flag & 128u;
    const bool rc = a.paired ? !((!is_rev && first) || (is_rev && last)) : is_rev;
    // tag.rs:151-173: the reference string is contig[max(start-2,0) .. min(end+2,LN)) padded with N to start-2 .. end+2
    const int64_t ln = a.g_ln[r.tid];
    const int64_t have = (int64_t)(a.g_off[r.tid + 1] - a.g_off[r.tid]);      // bases the FASTA actually gave
    const int64_t cs = start - 2 > 0 ? start - 2 : 0, ce = end + 2 < ln ? end + 2 : ln;
    const int64_t pad_s = 2 - start > 0 ? 2 - start : 0, pad_e = end - ln + 2 > 0 ? end - ln + 2 : 0;
    if (start < 0 || cs > ce || ce > have || pad_s > 2 || pad_e > 2) { atomicOr(a.err, (uint32_t)ERRB_TAGPANIC); return; }
    // A read shorter than its CIGAR's M + I (SEQ '*' beside a CIGAR: secondary alignments of bwa mem -a / bwa-meth) is not a panic in
    // the reference: chars().skip(a).take(b) just yields fewer characters (tag.rs:190-216), the read column string ends up
    // shorter than the reference one, and the two are then indexed independently (tag.rs:264-384).
    const bool short_read = qwalk > (int64_t)r.l_seq;
    const uint8_t *g = a.genome + a.g_off[r.tid];
    const int64_t reflen_str = pad_s + (ce - cs) + pad_e;
    auto ref_at = [&](int64_t k) -> uint8_t {                   // ref_seq[k]
        if (k < pad_s || k >= pad_s + (ce - cs)) return 'N';
        return tg_up(g[cs + (k - pad_s)]);
    };
    auto read_at = [&](uint32_t q) -> uint8_t {                 // rust-htslib seq().as_bytes(): "=ACMGRSVTWYHKDBN"
        const uint8_t b = r.seq[q >> 1];
        return (uint8_t)"=ACMGRSVTWYHKDBN"[(q & 1u) ? (b & 15u) : (b >> 4)];
    };
    if (reflen_str < 2) { atomicOr(a.err, (uint32_t)ERRB_TAGPANIC); return; }
    // tag.rs:175-243: the gapped columns
    uint32_t j = 2, jr = 2;                                       // next reference / read column (equal unless the read runs out)
    R[0] = '-'; R[1] = '-'; G[0] = ref_at(0); G[1] = ref_at(1);
    uint64_t uq = 0;
    int64_t ug = 2;
    for (uint32_t k = 0; k < r.n_cigar; ++k) {
        const uint32_t c = tg_u32(r.cigar + 4 * k), op = c & 15u, len = c >> 4;
        const uint32_t take = (op <= 1u) ? (uint32_t)(uq >= r.l_seq ? 0u : (r.l_seq - uq < len ? r.l_seq - uq : len)) : 0u;   // skip(uq).take(len)
        if (op == 0u) { for (uint32_t t = 0; t < take; ++t) R[jr++] = read_at((uint32_t)uq + t); for (uint32_t t = 0; t < len; ++t) G[j++] = ref_at(ug + t); uq += len; ug += len; }
        else if (op == 1u) { for (uint32_t t = 0; t < take; ++t) R[jr++] = read_at((uint32_t)uq + t); for (uint32_t t = 0; t < len; ++t) G[j++] = '-'; uq += len; }
        else if (op == 2u) { for (uint32_t t = 0; t < len; ++t) { R[jr++] = '-'; G[j++] = ref_at(ug + t); } ug += len; }
    }
    R[jr] = '-'; R[jr + 1] = '-'; G[j] = ref_at(reflen_str - 2); G[j + 1] = ref_at(reflen_str - 1);
    // (j + 2 == ncol by construction; jr == j unless short_read)
    if (short_read) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int64_t nx = tg_xm_unaligned(R, jr + 2u, G, j + 2u, rc, a.xm + a.col_off[i]);
        if (nx < 0) { atomicOr(a.err, (uint32_t)ERRB_TAGPANIC); return; }
        a.xm_len[i] = (uint32_t)nx;
        return;
    }
    if (rc) {                                                    // tag.rs:246-256: every character goes through the table
        for (uint32_t t = 0; t + 2 < ncol; ++t) {
            const uint8_t x = tg_comp(R[t]), y = tg_comp(G[t]);
            if (!x || !y) fail = 1;
            R[t] = x; G[t] = y;
        }
    }
    if (fail) { atomicOr(a.err, (uint32_t)ERRB_TAGPANIC); return; }
    // the thread reads its own columns back: every store above has to have landed first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // tag.rs:264-384, one column at a time; T index of column c: c - 2 (forward), ncol - 3 - c (reverse complement)
    const int32_t d = rc ? -1 : 1;
    const uint32_t m = ncol - 2;
    uint8_t *out = a.xm + a.col_off[i];
    uint32_t nx = 0;
    for (uint32_t c = 2; c + 2 < ncol; ++c) {
        const uint8_t r0 = R[c];
        if (r0 == '-') continue;
        if (r0 == 'N') { out[nx++] = '.'; continue; }
        if (G[c] != 'C') { out[nx++] = '.'; continue; }
        const uint32_t idx = rc ? ncol - 3u - c : c - 2u;
        uint8_t c1 = 0, c2 = 0;
        int n = 1;
        if ((R[(int32_t)c + d] == '-' || R[(int32_t)c + 2 * d] == '-') && idx != m - 3u && idx != m - 4u) {
            for (uint32_t k = 1; n != 3 && idx + k <= m - 1u; ++k) {                  // tag.rs:286-296
                const int32_t cc = (int32_t)c + d * (int32_t)k;
                if (R[cc] != '-') { if (n == 1) c1 = G[cc]; else c2 = G[cc]; ++n; }
            }
            if (n < 2) { fail = 1; break; }                      // tmp_target_ref_seq[1] panics
        } else {
            c1 = G[(int32_t)c + d]; c2 = G[(int32_t)c + 2 * d]; n = 3;
        }
        const uint8_t l = tg_letter(c1, c2, n, r0);
        if (l) out[nx++] = l;
    }
    if (fail) { atomicOr(a.err, (uint32_t)ERRB_TAGPANIC); return; }
    a.xm_len[i] = nx;
}

}  // namespace mth

using namespace mth;

extern "C" {

int mth_tag_set_genome(mth_ctx_t *ctx, int32_t n_refs, const int64_t *ref_len, const uint8_t *const *seq, const int64_t *seq_len) {
    if (!ctx || n_refs < 0 || (n_refs && (!ref_len || !seq || !seq_len))) return MTH_ERR_INVALID;
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    std::vector<uint64_t> off((size_t)n_refs + 1, 0);
    for (int32_t t = 0; t < n_refs; ++t) {
        if (seq_len[t] < 0 || (seq_len[t] && !seq[t])) return MTH_ERR_INVALID;
        off[t + 1] = off[t] + (uint64_t)seq_len[t];
    }
    MTH_HIP(ctx, ctx->tag_genome.reserve((size_t)off[n_refs] + 16, s));
    MTH_HIP(ctx, ctx->tag_goff.reserve(((size_t)n_refs + 1) * 8 + (size_t)n_refs * 8 + 16, s));
    for (int32_t t = 0; t < n_refs; ++t)
        if (seq_len[t]) MTH_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->tag_genome.p) + off[t], seq[t], (size_t)seq_len[t], hipMemcpyHostToDevice, s));
    MTH_HIP(ctx, hipMemcpyAsync(ctx->tag_goff.p, off.data(), ((size_t)n_refs + 1) * 8, hipMemcpyHostToDevice, s));
    if (n_refs) MTH_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->tag_goff.p) + ((size_t)n_refs + 1) * 8, ref_len, (size_t)n_refs * 8, hipMemcpyHostToDevice, s));
    MTH_HIP(ctx, hipStreamSynchronize(s));                      // the caller's buffers may go away
    ctx->tag_n_refs = n_refs;
    return MTH_OK;
}

int mth_tag_records(mth_ctx_t *ctx, const void *raw, uint64_t n_bytes, const uint64_t *rec_off, uint64_t n_rec, int mem,
                    int is_paired_end, mth_tag_out_t *out) {
    if (!ctx || !out || (n_rec && (!raw || !rec_off))) return MTH_ERR_INVALID;
    if (n_rec >= (1ull << 32) - 16) return fail(ctx, MTH_ERR_CAPACITY, "more than 2^32 records in one tag call: split the stream");
    if (ctx->tag_n_refs < 0) return fail(ctx, MTH_ERR_STATE, "mth_tag_set_genome has not been called");
    MTH_ENTER(ctx);
    hipStream_t s = ctx->stream;
    *out = mth_tag_out_t{};
    ctx->tag_h_off.assign(1, 0);
    ctx->tag_h_len.clear(); ctx->tag_h_xm.clear();
    out->xm_off = ctx->tag_h_off.data();
    if (n_rec == 0) return MTH_OK;
    const uint8_t *d_raw = (const uint8_t *)raw;
    const uint64_t *d_off = rec_off;
    if (mem == MTH_MEM_HOST) {
        MTH_HIP(ctx, ctx->dec_raw.reserve(n_bytes + 16, s));
        MTH_HIP(ctx, ctx->dec_recoff.reserve((n_rec + 1) * 8, s));
        if (n_bytes) MTH_HIP(ctx, hipMemcpyAsync(ctx->dec_raw.p, raw, n_bytes, hipMemcpyHostToDevice, s));
        MTH_HIP(ctx, hipMemcpyAsync(ctx->dec_recoff.p, rec_off, (n_rec + 1) * 8, hipMemcpyHostToDevice, s));
        d_raw = ctx->dec_raw.as<uint8_t>(); d_off = ctx->dec_recoff.as<uint64_t>();
    } else if (mem != MTH_MEM_DEVICE) {
        return fail(ctx, MTH_ERR_INVALID, "mem");
    }
    const uint32_t n = (uint32_t)n_rec, nb = (n + 255) / 256;
    MTH_HIP(ctx, ctx->tag_ncol.reserve((size_t)n * 4 + 16, s));
    MTH_HIP(ctx, ctx->tag_coloff.reserve(((size_t)n + 1) * 8 + 16, s));
    MTH_HIP(ctx, ctx->tag_xmlen.reserve((size_t)n * 4 + 16, s));
    TagArgs a{};
    a.raw = d_raw; a.off = d_off; a.n_rec = n;
    a.genome = ctx->tag_genome.as<uint8_t>(); a.g_off = ctx->tag_goff.as<uint64_t>();
    a.g_ln = reinterpret_cast<const int64_t *>(ctx->tag_goff.as<uint64_t>() + ctx->tag_n_refs + 1);
    a.n_refs = ctx->tag_n_refs; a.paired = is_paired_end ? 1 : 0;
    a.ncol = ctx->tag_ncol.as<uint32_t>(); a.xm_len = ctx->tag_xmlen.as<uint32_t>(); a.err = &ctx->d_state->err;
    hipLaunchKernelGGL(k_tag_count, dim3(nb), dim3(256), 0, s, a);
    unsigned long long total = 0;
    int rc = scan_u32_to_u64(ctx, a.ncol, n, 0ull, ctx->tag_coloff.as<unsigned long long>(), &total);   // synchronises
    if (rc) return rc;
    MTH_HIP(ctx, ctx->tag_cols.reserve((size_t)total * 2 + 16, s));
    MTH_HIP(ctx, ctx->tag_xm.reserve((size_t)total + 16, s));
    a.col_off = ctx->tag_coloff.as<unsigned long long>(); a.cols = ctx->tag_cols.as<uint8_t>(); a.total = total;
    a.xm = ctx->tag_xm.as<uint8_t>();
    hipLaunchKernelGGL(k_tag_xm, dim3(nb), dim3(256), 0, s, a);
    MTH_HIP(ctx, hipGetLastError());
    if ((rc = sync_and_check(ctx))) return rc;
    // results to the host: slot offsets, lengths, letters
    ctx->tag_h_off.resize((size_t)n + 1);
    ctx->tag_h_len.resize(n);
    ctx->tag_h_xm.resize((size_t)total);
    MTH_HIP(ctx, hipMemcpy(ctx->tag_h_off.data(), ctx->tag_coloff.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
    MTH_HIP(ctx, hipMemcpy(ctx->tag_h_len.data(), ctx->tag_xmlen.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (total) MTH_HIP(ctx, hipMemcpy(ctx->tag_h_xm.data(), ctx->tag_xm.p, (size_t)total, hipMemcpyDeviceToHost));
    out->n_records = n_rec; out->xm_off = ctx->tag_h_off.data(); out->xm_len = ctx->tag_h_len.data();
    out->xm = reinterpret_cast<const char *>(ctx->tag_h_xm.data());
    return MTH_OK;
}

}  // extern "C"
