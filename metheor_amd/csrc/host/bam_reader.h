// bam_reader.h -- minimal BGZF + BAM reader on zlib (the image has no htslib).
// Replaces what the reference gets from rust-htslib's bam::Reader (src/bamutil.rs:4-25).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace mthh {

struct BamRef {
    std::string name;
    int64_t length;
};

// One alignment record, as views into the reader's block buffer (valid until the next call).
struct BamRecord {
    int32_t tid, pos;
    uint16_t flag;
    uint8_t mapq;
    uint32_t n_cigar, l_seq;
    const uint32_t *cigar;      // BAM-packed: len << 4 | op  (may be unaligned: use read_u32)
    const uint8_t *aux;
    uint32_t aux_len;
};

class BamReader {
  public:
    ~BamReader();
    // returns false and sets error() on failure
    bool open(const std::string &path);
    bool next(BamRecord &rec, bool &eof);
    const std::vector<BamRef> &refs() const { return refs_; }
    const std::string &header_text() const { return text_; }
    const std::string &error() const { return err_; }
    // uncompressed bytes handed out so far (right after open(): the size of the BAM header)
    size_t consumed() const { return consumed_; }

  private:
    bool fill();                               // inflate the next BGZF block into buf_
    bool read_bytes(void *dst, size_t n, bool &eof_at_start);
    FILE *fp_ = nullptr;
    std::string path_, err_, text_;
    std::vector<BamRef> refs_;
    std::vector<uint8_t> buf_;                 // uncompressed bytes not yet consumed
    size_t off_ = 0, consumed_ = 0;
    std::vector<uint8_t> cbuf_, rec_;
};

inline uint32_t read_u32(const void *p) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}
inline int32_t read_i32(const void *p) { return (int32_t)read_u32(p); }
inline uint16_t read_u16(const void *p) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    return (uint16_t)(b[0] | (b[1] << 8));
}

}  // namespace mthh
