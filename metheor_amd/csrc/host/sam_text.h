// sam_text.h -- SAM text <-> BAM records and a FASTA reader (see sam_text.cpp)
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "bam_reader.h"

namespace mthh {

// true if the file is text that starts like SAM (an @XX header line, or an 11-field alignment line)
bool looks_like_sam(const std::string &path);
// the whole SAM file as the bytes of an equivalent BGZF-compressed BAM file (whole records per block)
bool sam_text_to_bam(const std::string &path, std::vector<uint8_t> &bam, std::string &err);
// one BAM record (the bytes after block_size) as a SAM line ending in '\n'; xm != nullptr appends "\tXM:Z:<xm>"
bool sam_format_record(const std::vector<BamRef> &refs, const uint8_t *rec, uint32_t len, const char *xm, uint32_t xm_len, std::string &out);

class Fasta {
  public:
    bool open(const std::string &path, std::string &err);
    // faidx_fetch_seq(name, 0, end_incl): the bases [0, end_incl] clipped to the sequence, white space dropped
    bool fetch(const std::string &name, int64_t end_incl, std::vector<uint8_t> &seq, std::string &err) const;

  private:
    struct Entry { int64_t length = 0, offset = 0, line_bases = 0, line_width = 0; };
    std::string path_;
    std::unordered_map<std::string, Entry> index_;
};

}  // namespace mthh
