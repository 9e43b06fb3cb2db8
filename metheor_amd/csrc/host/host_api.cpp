// host_api.cpp -- include/metheor_host.h: BAM -> decoded SoA (the reference's BismarkRead stream).
//
// The record decode itself lives in parallel_decode.cpp (multi-threaded inflate + decode); it follows
// src/readutil.rs:24-53 (BismarkRead::new) and 323-345 (get_cpgs) in ONE walk over the CIGAR with a
// cursor into the XM string (no per-base position vector):
//   M/=/X  consume query+reference : XM[q] in {z,Z} -> CpG at reference r (forward: flags exactly
//          0, 99 or 147) or r-1 (everything else), relpos = q          (readutil.rs:326-341)
//   I/S    consume query only      : reference position None -> skipped (readutil.rs:331)
//   D/N    consume reference only  : nothing yielded by reference_positions_full()
//   H/P    nothing
// start_pos / end_pos = first / last aligned reference position, -1 if none (readutil.rs:25-33).
#include <zlib.h>

#include <algorithm>
#include <climits>
#include <charconv>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../../include/metheor_host.h"
#include "bam_reader.h"
#include "bai_internal.h"
#include "parallel_decode.h"
#include "sam_text.h"
#include <sys/mman.h>
#include <unistd.h>

#include <thread>

using namespace mthh;

struct mth_host : DecodedSoA {
    BamReader reader;
    std::string path, last_error;
    std::unordered_map<std::string, int> name2tid;
    size_t header_bytes = 0;
    std::unique_ptr<BgzfMap> bgzf;
    std::vector<uint64_t> cpg_keys;
    int xm_min_mapq = 0;             // mth_host_set_xm_min_mapq
    int memfd = -1;                  // SAM input: the equivalent BAM lives in an anonymous memory file, `path` names it
    ~mth_host() { if (memfd >= 0) close(memfd); }
};

struct mth_fasta {
    Fasta fa;
    std::vector<uint8_t> seq;
    std::string last_error;
};

namespace {

inline uint64_t site_key(int32_t tid, int32_t pos) { return ((uint64_t)(uint32_t)tid << 32) | (uint32_t)pos; }

}  // namespace

// readutil.rs:347-374 get_target_cpgs: tab-split lines, col0 chrom (must be in the header), col1 start; HashSet<CpGPosition>
static int read_cpg_set(mth_host *h, const char *cpg_set_path, std::unordered_set<uint64_t> &target) {
    std::ifstream f(cpg_set_path);
    if (!f) { h->last_error = "Could not read target CpG file."; return MTH_HOST_ERR_CPGSET; }
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const size_t t1 = line.find('\t');
        if (t1 == std::string::npos) { h->last_error = "malformed --cpg-set line (needs chrom<TAB>start): " + line; return MTH_HOST_ERR_CPGSET; }
        const size_t t2 = line.find('\t', t1 + 1);
        const std::string chrom = line.substr(0, t1);
        const std::string num = line.substr(t1 + 1, t2 == std::string::npos ? std::string::npos : t2 - t1 - 1);
        int32_t pos = 0;
        const auto r = std::from_chars(num.data(), num.data() + num.size(), pos);
        if (r.ec != std::errc() || r.ptr != num.data() + num.size()) { h->last_error = "bad start in --cpg-set: " + line; return MTH_HOST_ERR_CPGSET; }
        const auto it = h->name2tid.find(chrom);
        if (it == h->name2tid.end()) { h->last_error = "unknown contig in --cpg-set: " + chrom; return MTH_HOST_ERR_CPGSET; }
        target.insert(site_key(it->second, pos));
    }
    return MTH_HOST_OK;
}

extern "C" {

int mth_host_open(const char *path, mth_host_t **out, char *errbuf, int errbuf_len) {
    if (!path || !out) return MTH_HOST_ERR_INVALID;
    *out = nullptr;
    auto *h = new mth_host;
    std::string open_path = path;
    // htslib opens SAM text through the same call (bamutil.rs:4-11; tests/tag-cli.rs feeds `tag` a .sam): the text is
    // converted once into the bytes of a BGZF-compressed BAM held in an anonymous memory file, and everything behind
    // this handle -- host inflate, device inflate, the shard planner -- reads that
    if (looks_like_sam(path)) {
        std::vector<uint8_t> bam;
        std::string err;
        if (!sam_text_to_bam(path, bam, err)) {
            const std::string msg = "Error opening BAM file. " + err;
            if (errbuf && errbuf_len > 0) { strncpy(errbuf, msg.c_str(), (size_t)errbuf_len - 1); errbuf[errbuf_len - 1] = 0; }
            delete h;
            return MTH_HOST_ERR_OPEN;
        }
        h->memfd = memfd_create("metheor_sam_as_bam", 0);
        bool ok = h->memfd >= 0;
        for (size_t o = 0; ok && o < bam.size();) { const ssize_t w = write(h->memfd, bam.data() + o, bam.size() - o); if (w <= 0) ok = false; else o += (size_t)w; }
        if (!ok) {
            if (errbuf && errbuf_len > 0) { strncpy(errbuf, "Error opening BAM file. cannot stage the SAM input in memory", (size_t)errbuf_len - 1); errbuf[errbuf_len - 1] = 0; }
            delete h;
            return MTH_HOST_ERR_OPEN;
        }
        open_path = "/proc/self/fd/" + std::to_string(h->memfd);
    }
    if (!h->reader.open(open_path)) {
        // bamutil.rs:7-9: panic!("Error opening BAM file. {}", error)
        const std::string msg = "Error opening BAM file. " + h->reader.error();
        if (errbuf && errbuf_len > 0) { strncpy(errbuf, msg.c_str(), (size_t)errbuf_len - 1); errbuf[errbuf_len - 1] = 0; }
        delete h;
        return MTH_HOST_ERR_OPEN;
    }
    for (size_t i = 0; i < h->reader.refs().size(); ++i) h->name2tid.emplace(h->reader.refs()[i].name, (int)i);
    h->path = open_path;
    h->header_bytes = h->reader.consumed();
    *out = h;
    return MTH_HOST_OK;
}

void mth_host_close(mth_host_t *h) { delete h; }
const char *mth_host_last_error(const mth_host_t *h) { return h ? h->last_error.c_str() : ""; }
uint32_t mth_host_notes(void) { return mthh::g_decode_notes.load(); }
int mth_host_n_refs(const mth_host_t *h) { return (int)h->reader.refs().size(); }
const char *mth_host_ref_name(const mth_host_t *h, int tid) {
    return (tid >= 0 && tid < (int)h->reader.refs().size()) ? h->reader.refs()[tid].name.c_str() : "";
}
int64_t mth_host_ref_len(const mth_host_t *h, int tid) {
    return (tid >= 0 && tid < (int)h->reader.refs().size()) ? h->reader.refs()[tid].length : -1;
}
int mth_host_ref_tid(const mth_host_t *h, const char *name) {
    auto it = h->name2tid.find(name);
    return it == h->name2tid.end() ? -1 : it->second;
}

const char *mth_host_path(const mth_host_t *h) { return h ? h->path.c_str() : ""; }

const char *mth_host_header_text(const mth_host_t *h, uint64_t *n_bytes) {
    if (n_bytes) *n_bytes = h ? h->reader.header_text().size() : 0;
    return h ? h->reader.header_text().data() : "";
}

int64_t mth_host_sam_format(const mth_host_t *h, const uint8_t *rec, uint32_t rec_len, const char *xm, uint32_t xm_len, char *buf, int64_t cap) {
    if (!h || !rec) return MTH_HOST_ERR_INVALID;
    std::string line;
    if (!sam_format_record(h->reader.refs(), rec, rec_len, xm, xm_len, line)) return MTH_HOST_ERR_FORMAT;
    if (buf && (int64_t)line.size() <= cap) memcpy(buf, line.data(), line.size());
    return (int64_t)line.size();
}

int mth_host_fasta_open(const char *path, mth_fasta_t **out, char *errbuf, int errbuf_len) {
    if (!path || !out) return MTH_HOST_ERR_INVALID;
    *out = nullptr;
    auto *f = new mth_fasta;
    std::string err;
    if (!f->fa.open(path, err)) {
        if (errbuf && errbuf_len > 0) { strncpy(errbuf, err.c_str(), (size_t)errbuf_len - 1); errbuf[errbuf_len - 1] = 0; }
        delete f;
        return MTH_HOST_ERR_OPEN;
    }
    *out = f;
    return MTH_HOST_OK;
}
void mth_host_fasta_close(mth_fasta_t *f) { delete f; }
const char *mth_host_fasta_last_error(const mth_fasta_t *f) { return f ? f->last_error.c_str() : ""; }
int mth_host_fasta_fetch(mth_fasta_t *f, const char *name, int64_t end_incl, const uint8_t **seq, int64_t *len) {
    if (!f || !name || !seq || !len) return MTH_HOST_ERR_INVALID;
    if (!f->fa.fetch(name, end_incl, f->seq, f->last_error)) return MTH_HOST_ERR_FORMAT;
    *seq = f->seq.data(); *len = (int64_t)f->seq.size();
    return MTH_HOST_OK;
}

int mth_host_set_xm_min_mapq(mth_host_t *h, int min_mapq) {
    if (!h) return MTH_HOST_ERR_INVALID;
    h->xm_min_mapq = min_mapq;
    return MTH_HOST_OK;
}

int mth_host_decode(mth_host_t *h, const char *cpg_set_path) {
    if (!h) return MTH_HOST_ERR_INVALID;
    h->tid.clear(); h->start.clear(); h->end.clear(); h->mapq.clear(); h->fwd.clear();
    h->cpg_off.assign(1, 0); h->cpg_pos.clear(); h->cpg_rel.clear();

    bool have_set = false;
    std::unordered_set<uint64_t> target;
    if (cpg_set_path) {
        const int rc = read_cpg_set(h, cpg_set_path, target);
        if (rc) return rc;
        have_set = true;
    }

    // records: multi-threaded inflate + decode (parallel_decode.cpp); METHEOR_THREADS overrides the thread count
    int nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    if (nthreads > 128) nthreads = 128;
    if (const char *e = getenv("METHEOR_THREADS")) { const int k = atoi(e); if (k >= 1 && k <= 1024) nthreads = k; }
    std::string err;
    int kind = 0;
    if (!parallel_decode(h->path, h->header_bytes, have_set ? &target : nullptr, nthreads, *h, err, kind, nullptr, h->xm_min_mapq)) {
        if (kind == 2) { h->last_error = err; return MTH_HOST_ERR_XM; }
        h->last_error = "Error reading BAM record. " + err;
        return MTH_HOST_ERR_FORMAT;
    }
    return MTH_HOST_OK;
}

int mth_host_cpg_set_keys(mth_host_t *h, const char *cpg_set_path, const uint64_t **keys, uint64_t *n_keys) {
    if (!h || !cpg_set_path || !keys || !n_keys) return MTH_HOST_ERR_INVALID;
    std::unordered_set<uint64_t> target;
    const int rc = read_cpg_set(h, cpg_set_path, target);
    if (rc) return rc;
    h->cpg_keys.assign(target.begin(), target.end());
    std::sort(h->cpg_keys.begin(), h->cpg_keys.end());
    *keys = h->cpg_keys.data(); *n_keys = h->cpg_keys.size();
    return MTH_HOST_OK;
}

int mth_host_bgzf_blocks(mth_host_t *h, mth_host_bgzf_t *out) {
    if (!h || !out) return MTH_HOST_ERR_INVALID;
    if (!h->bgzf) {
        std::unique_ptr<BgzfMap> m(new BgzfMap);
        std::string err;
        if (!bgzf_map(h->path, *m, err)) { h->last_error = "Error reading BAM record. " + err; return MTH_HOST_ERR_FORMAT; }
        h->bgzf = std::move(m);
    }
    out->file = h->bgzf->file; out->file_bytes = h->bgzf->file_bytes;
    out->coff = h->bgzf->coff.data(); out->csize = h->bgzf->csize.data(); out->isize = h->bgzf->isize.data();
    out->n_blocks = h->bgzf->coff.size(); out->header_bytes = h->header_bytes;
    return MTH_HOST_OK;
}

// first record of BGZF block `b` at inflated offset `off`: (refID, pos); refID -1 (no contig) sorts last
static bool peek_record(const BgzfMap &m, uint64_t b, uint64_t off, int32_t &tid, int32_t &pos) {
    const uint32_t isz = m.isize[b];
    if (off + 12 > isz) return false;
    std::vector<uint8_t> buf(isz);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(m.file + m.coff[b]); zs.avail_in = m.csize[b];
    zs.next_out = buf.data(); zs.avail_out = isz;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return false;
    auto rd = [&](uint64_t o) { int32_t v; memcpy(&v, buf.data() + o, 4); return v; };
    const int32_t block_size = rd(off);
    if (block_size < 32) return false;                       // not a record header: the run does not start at a record
    tid = rd(off + 4); pos = rd(off + 8);
    if (tid < -1 || pos < -1) return false;
    if (tid < 0) tid = INT32_MAX;
    return true;
}

int mth_host_plan_shard(mth_host_t *h, int rank, int world, int64_t halo_bp, mth_host_shard_t *out) {
    if (!h || !out || world < 1 || rank < 0 || rank >= world || halo_bp < 0) return MTH_HOST_ERR_INVALID;
    mth_host_bgzf_t bz;
    const int rc = mth_host_bgzf_blocks(h, &bz);
    if (rc != MTH_HOST_OK) return rc;
    const BgzfMap &m = *h->bgzf;
    memset(out, 0, sizeof *out);
    out->tid_beg = -1; out->tid_end = INT32_MAX;
    // data blocks [D, E): D holds the first record (at inflated offset d_off), E drops trailing empty blocks (the EOF marker)
    uint64_t D = 0, cum = 0, E = bz.n_blocks;
    while (D < E && cum + m.isize[D] <= h->header_bytes) cum += m.isize[D++];
    const uint64_t d_off = h->header_bytes - cum;
    while (E > D && m.isize[E - 1] == 0) --E;
    if (D >= E) { out->block_beg = out->block_end = 0; out->first_byte = 0; return MTH_HOST_OK; }   // no record at all
    auto cut = [&](int k) -> uint64_t {      // first block of run k: runs of about equal compressed size
        if (k <= 0) return D;
        if (k >= world) return E;
        const uint64_t total = m.coff[E - 1] + m.csize[E - 1] - m.coff[D];
        const uint64_t want = m.coff[D] + (uint64_t)((unsigned __int128)total * (unsigned)k / (unsigned)world);
        const auto it = std::lower_bound(m.coff.begin() + (ptrdiff_t)D, m.coff.begin() + (ptrdiff_t)E, want);
        return (uint64_t)(it - m.coff.begin());
    };
    auto first_of = [&](uint64_t b, int32_t &t, int32_t &p) { return peek_record(m, b, b == D ? d_off : 0, t, p); };
    const uint64_t B0 = cut(rank), B1 = cut(rank + 1);
    auto fail = [&]() { h->last_error = "a shard does not start at a record boundary (records straddle BGZF blocks)"; return MTH_HOST_ERR_FORMAT; };
    uint64_t L = D, R = E;
    if (rank > 0) {
        if (B0 >= E) { out->tid_beg = INT32_MAX; out->pos_beg = 0; L = E; }
        else {
            int32_t t, p;
            if (!first_of(B0, t, p)) return fail();
            out->tid_beg = t; out->pos_beg = p;
            // left halo: back to the block that holds the last record starting before p - halo_bp (or of an earlier contig)
            L = B0;
            while (L > D) {
                int32_t tj, pj;
                if (!first_of(L - 1, tj, pj)) return fail();
                --L;
                if (tj != t || (int64_t)pj < (int64_t)p - halo_bp) break;
            }
        }
    }
    if (rank < world - 1 && B1 < E) {
        int32_t t, p;
        if (!first_of(B1, t, p)) return fail();
        out->tid_end = t; out->pos_end = p;
        // right halo: the blocks that can hold reads starting exactly at p
        R = B1 + 1;
        while (R < E) {
            int32_t tj, pj;
            if (!first_of(R, tj, pj)) return fail();
            if (tj != t || pj != p) break;
            ++R;
        }
    }
    const bool empty = out->tid_beg == out->tid_end && out->pos_beg == out->pos_end;
    if (empty || L >= R) { out->block_beg = out->block_end = L; out->first_byte = 0; return MTH_HOST_OK; }
    if (L <= D) { out->block_beg = 0; out->first_byte = h->header_bytes; }    // from the top of the file: the header comes along
    else { out->block_beg = L; out->first_byte = 0; }
    out->block_end = R;
    return MTH_HOST_OK;
}

int mth_host_plan_region(mth_host_t *h, const char *bai_path, int32_t tid, int32_t beg, int32_t end, int64_t halo_bp, mth_host_shard_t *out) {
    if (!h || !out || tid < 0 || tid >= (int32_t)h->reader.refs().size() || beg < 0 || end < beg || halo_bp < 0) return MTH_HOST_ERR_INVALID;
    mth_host_bgzf_t bz;
    const int rc = mth_host_bgzf_blocks(h, &bz);
    if (rc != MTH_HOST_OK) return rc;
    const BgzfMap &m = *h->bgzf;
    BaiIndex bai;
    std::string path = bai_path ? bai_path : h->path + ".bai", err;
    if (!bai.load(path, err)) {
        // samtools also writes <name>.bai next to <name>.bam
        std::string alt = h->path;
        if (!bai_path && alt.size() > 4 && alt.compare(alt.size() - 4, 4, ".bam") == 0) { alt.replace(alt.size() - 4, 4, ".bai"); std::string e2; if (bai.load(alt, e2)) err.clear(); }
        if (!err.empty()) { h->last_error = err; return MTH_HOST_ERR_OPEN; }
    }
    if (bai.refs.size() != h->reader.refs().size()) { h->last_error = "the BAM index does not belong to this file (different number of references)"; return MTH_HOST_ERR_FORMAT; }
    memset(out, 0, sizeof *out);
    out->tid_beg = out->tid_end = tid; out->pos_beg = beg; out->pos_end = end;
    // records that can touch the region's sites: overlapping [beg - halo, end] (a reverse read starting AT end reports end - 1)
    uint64_t vlo = 0, vhi = 0;
    if (end == beg || !bai.query(tid, std::max<int64_t>(0, (int64_t)beg - halo_bp), (int64_t)end + 1, vlo, vhi)) { out->block_beg = out->block_end = 0; return MTH_HOST_OK; }
    // virtual offset -> block of the table: its payload starts right after the block's header, i.e. it is the first payload
    // offset above the block's own offset
    auto block_of = [&](uint64_t v) { return (uint64_t)(std::upper_bound(m.coff.begin(), m.coff.end(), v >> 16) - m.coff.begin()); };
    uint64_t L = block_of(vlo), R = block_of(vhi) + ((vhi & 0xffffu) ? 1u : 0u);
    R = std::min<uint64_t>(std::max(R, L + 1), bz.n_blocks);
    if (L >= bz.n_blocks) { out->block_beg = out->block_end = 0; return MTH_HOST_OK; }
    // the block that holds the first record also holds the tail of the header: load from the top of the file then
    uint64_t D = 0, cum = 0;
    while (D < bz.n_blocks && cum + m.isize[D] <= h->header_bytes) cum += m.isize[D++];
    if (L <= D) { out->block_beg = 0; out->first_byte = h->header_bytes; } else { out->block_beg = L; out->first_byte = 0; }
    out->block_end = R;
    return MTH_HOST_OK;
}

int mth_host_decode_stream(mth_host_t *h, mth_host_window_cb cb, void *user) {
    if (!h || !cb) return MTH_HOST_ERR_INVALID;
    h->tid.clear(); h->start.clear(); h->end.clear(); h->mapq.clear(); h->fwd.clear();
    h->cpg_off.assign(1, 0); h->cpg_pos.clear(); h->cpg_rel.clear();
    int nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    if (nthreads > 128) nthreads = 128;
    if (const char *e = getenv("METHEOR_THREADS")) { const int k = atoi(e); if (k >= 1 && k <= 1024) nthreads = k; }
    int cb_rc = 0;
    const WindowSink sink = [&](const uint8_t *buf, const uint64_t *rec_off, size_t n_rec, std::string &e) {
        cb_rc = cb(user, buf, rec_off, (uint64_t)n_rec);
        if (cb_rc != 0) { e = "window consumer failed"; return false; }
        return true;
    };
    std::string err;
    int kind = 0;
    if (!parallel_decode(h->path, h->header_bytes, nullptr, nthreads, *h, err, kind, &sink)) {
        if (cb_rc != 0) { h->last_error = "window consumer returned an error"; return MTH_HOST_ERR_CONSUMER; }
        h->last_error = "Error reading BAM record. " + err;
        return MTH_HOST_ERR_FORMAT;
    }
    return MTH_HOST_OK;
}

int64_t mth_host_n_reads(const mth_host_t *h) { return (int64_t)h->tid.size(); }
int64_t mth_host_n_cpgs(const mth_host_t *h) { return (int64_t)h->cpg_pos.size(); }
const int32_t *mth_host_read_tid(const mth_host_t *h) { return h->tid.data(); }
const int32_t *mth_host_read_start(const mth_host_t *h) { return h->start.data(); }
const int32_t *mth_host_read_end(const mth_host_t *h) { return h->end.data(); }
const uint8_t *mth_host_read_mapq(const mth_host_t *h) { return h->mapq.data(); }
const uint8_t *mth_host_read_fwd(const mth_host_t *h) { return h->fwd.data(); }
const uint64_t *mth_host_cpg_off(const mth_host_t *h) { return h->cpg_off.data(); }
const uint32_t *mth_host_cpg_pos(const mth_host_t *h) { return h->cpg_pos.data(); }
const uint16_t *mth_host_cpg_rel(const mth_host_t *h) { return h->cpg_rel.data(); }

// Rust `impl Display for f32`: shortest digits that round-trip, never an exponent.
// std::to_chars(float, chars_format::fixed) without a precision is exactly that.
int mth_host_format_f32(float v, char *buf) {
    if (std::isnan(v)) { memcpy(buf, "NaN", 4); return 3; }
    if (std::isinf(v)) { const char *s = v < 0 ? "-inf" : "inf"; const int n = (int)strlen(s); memcpy(buf, s, (size_t)n + 1); return n; }
    const auto r = std::to_chars(buf, buf + 60, v, std::chars_format::fixed);
    *r.ptr = 0;
    return (int)(r.ptr - buf);
}

}  // extern "C"
