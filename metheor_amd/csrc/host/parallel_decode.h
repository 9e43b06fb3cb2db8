// parallel_decode.h -- multi-threaded BGZF/BAM -> SoA decode (see parallel_decode.cpp)
#pragma once
#include <cstdint>
#include <string>
#include <unordered_set>
#include <vector>

namespace mthh {

struct DecodedSoA {
    std::vector<int32_t> tid, start, end;
    std::vector<uint8_t> mapq, fwd;
    std::vector<uint64_t> cpg_off;
    std::vector<uint32_t> cpg_pos;
    std::vector<uint16_t> cpg_rel;
};

// header_bytes: uncompressed size of the BAM header (records start right after it).
// target: --cpg-set keys (tid << 32 | pos) or nullptr.  err_kind: 1 format/IO, 2 record without XM.
bool parallel_decode(const std::string &path, size_t header_bytes, const std::unordered_set<uint64_t> *target,
                     int nthreads, DecodedSoA &out, std::string &err, int &err_kind);

}  // namespace mthh
