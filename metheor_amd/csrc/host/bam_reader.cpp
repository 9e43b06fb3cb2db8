// bam_reader.cpp -- BGZF block inflate (zlib, raw deflate) + BAM header / record framing.
#include "bam_reader.h"

#include <zlib.h>

#include <cerrno>
#include <cstring>

namespace mthh {

BamReader::~BamReader() {
    if (fp_) fclose(fp_);
}

// One BGZF block: 18-byte gzip header with the BC extra subfield (BSIZE), raw deflate payload,
// CRC32 + ISIZE trailer.  SAM spec section 4.1.
bool BamReader::fill() {
    buf_.clear();
    off_ = 0;
    for (;;) {
        uint8_t hdr[18];
        const size_t got = fread(hdr, 1, sizeof hdr, fp_);
        if (got == 0) return true;  // clean EOF: buf_ stays empty
        if (got != sizeof hdr || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) {
            err_ = "not a BGZF file (bad block header)";
            return false;
        }
        const unsigned xlen = read_u16(hdr + 10);
        // the BC subfield is normally first; scan the extra field to be safe
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp_) != xlen - 6) { err_ = "truncated BGZF block"; return false; }
        int bsize = -1;
        for (unsigned o = 0; o + 4 <= xlen;) {
            const unsigned slen = read_u16(extra.data() + o + 2);
            if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = read_u16(extra.data() + o + 4);
            o += 4 + slen;
        }
        if (bsize < 0) { err_ = "not a BGZF file (no BC subfield)"; return false; }
        const long payload = (long)bsize + 1 - 12 - (long)xlen - 8;
        if (payload < 0) { err_ = "corrupt BGZF block size"; return false; }
        cbuf_.resize((size_t)payload + 8);
        if (fread(cbuf_.data(), 1, cbuf_.size(), fp_) != cbuf_.size()) { err_ = "truncated BGZF block"; return false; }
        const uint32_t isize = read_u32(cbuf_.data() + payload + 4);
        if (isize == 0) continue;  // empty block (the EOF marker): look for more
        buf_.resize(isize);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { err_ = "zlib init failed"; return false; }
        zs.next_in = cbuf_.data();
        zs.avail_in = (uInt)payload;
        zs.next_out = buf_.data();
        zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.avail_out != 0) { err_ = "corrupt BGZF block (inflate failed)"; return false; }
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), buf_.data(), isize) != read_u32(cbuf_.data() + payload)) {
            err_ = "corrupt BGZF block (CRC mismatch)";
            return false;
        }
        return true;
    }
}

bool BamReader::read_bytes(void *dst, size_t n, bool &eof_at_start) {
    eof_at_start = false;
    uint8_t *d = static_cast<uint8_t *>(dst);
    size_t done = 0;
    while (done < n) {
        if (off_ == buf_.size()) {
            if (!fill()) return false;
            if (buf_.empty()) {
                if (done == 0) { eof_at_start = true; return true; }
                err_ = "truncated BAM file";
                return false;
            }
        }
        const size_t take = std::min(n - done, buf_.size() - off_);
        memcpy(d + done, buf_.data() + off_, take);
        off_ += take;
        done += take;
        consumed_ += take;
    }
    return true;
}

bool BamReader::open(const std::string &path) {
    path_ = path;
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) {
        // rust-htslib: Error::FileNotFound displays "file not found: <path>"
        err_ = (errno == ENOENT ? "file not found: " : "unable to open: ") + path;
        return false;
    }
    bool eof;
    uint8_t magic[4];
    if (!read_bytes(magic, 4, eof) || eof || memcmp(magic, "BAM\1", 4) != 0) {
        err_ = "invalid BAM header: " + path + (err_.empty() ? "" : " (" + err_ + ")");
        return false;
    }
    uint8_t b4[4];
    if (!read_bytes(b4, 4, eof) || eof) { err_ = "invalid BAM header: " + path; return false; }
    const int32_t l_text = read_i32(b4);
    if (l_text < 0) { err_ = "invalid BAM header: " + path; return false; }
    text_.resize((size_t)l_text);
    if (l_text && (!read_bytes(&text_[0], (size_t)l_text, eof) || eof)) { err_ = "invalid BAM header: " + path; return false; }
    if (!read_bytes(b4, 4, eof) || eof) { err_ = "invalid BAM header: " + path; return false; }
    const int32_t n_ref = read_i32(b4);
    if (n_ref < 0) { err_ = "invalid BAM header: " + path; return false; }
    for (int32_t r = 0; r < n_ref; ++r) {
        if (!read_bytes(b4, 4, eof) || eof) { err_ = "invalid BAM header: " + path; return false; }
        const int32_t l_name = read_i32(b4);
        if (l_name <= 0 || l_name > 1 << 20) { err_ = "invalid BAM header: " + path; return false; }
        std::string name((size_t)l_name, '\0');
        if (!read_bytes(&name[0], (size_t)l_name, eof) || eof) { err_ = "invalid BAM header: " + path; return false; }
        name.resize(strlen(name.c_str()));
        if (!read_bytes(b4, 4, eof) || eof) { err_ = "invalid BAM header: " + path; return false; }
        refs_.push_back(BamRef{name, (int64_t)(uint32_t)read_i32(b4)});
    }
    return true;
}

bool BamReader::next(BamRecord &rec, bool &eof) {
    uint8_t b4[4];
    if (!read_bytes(b4, 4, eof)) return false;
    if (eof) return true;
    const int32_t block_size = read_i32(b4);
    if (block_size < 32) { err_ = "corrupt BAM record"; return false; }
    rec_.resize((size_t)block_size);
    bool e2;
    if (!read_bytes(rec_.data(), rec_.size(), e2) || e2) { if (err_.empty()) err_ = "truncated BAM file"; return false; }
    const uint8_t *p = rec_.data();
    rec.tid = read_i32(p);
    rec.pos = read_i32(p + 4);
    const uint32_t l_read_name = p[8];
    rec.mapq = p[9];
    rec.n_cigar = read_u16(p + 12);
    rec.flag = read_u16(p + 14);
    rec.l_seq = read_u32(p + 16);
    const size_t o_cigar = 32 + (size_t)l_read_name;
    const size_t o_aux = o_cigar + 4ull * rec.n_cigar + ((size_t)rec.l_seq + 1) / 2 + rec.l_seq;
    if (o_aux > rec_.size()) { err_ = "corrupt BAM record"; return false; }
    rec.cigar = reinterpret_cast<const uint32_t *>(p + o_cigar);
    rec.aux = p + o_aux;
    rec.aux_len = (uint32_t)(rec_.size() - o_aux);
    return true;
}

}  // namespace mthh
