// bai_internal.h -- parsed .bai (private to the host library)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mthh {

struct BaiIndex {
    struct Chunk { uint32_t bin; uint64_t beg, end; };          // virtual offsets: compressed block start << 16 | offset inside the inflated block
    struct Ref { std::vector<Chunk> chunks; std::vector<uint64_t> ioffset; };
    std::vector<Ref> refs;
    bool load(const std::string &path, std::string &err);
    bool query(int32_t tid, int64_t beg, int64_t end, uint64_t &lo, uint64_t &hi) const;
};

}  // namespace mthh
