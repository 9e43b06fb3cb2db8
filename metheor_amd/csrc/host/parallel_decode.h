// parallel_decode.h -- multi-threaded BGZF/BAM -> SoA decode (see parallel_decode.cpp)
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <unordered_set>
#include <vector>

#include <atomic>

namespace mthh {

extern std::atomic<uint32_t> g_decode_notes;     // bit 0: a record with a CIGAR P operation was decoded (see MTH_NOTE_CIGAR_PAD)

struct DecodedSoA {
    std::vector<int32_t> tid, start, end;
    std::vector<uint8_t> mapq, fwd;
    std::vector<uint64_t> cpg_off;
    std::vector<uint32_t> cpg_pos;
    std::vector<uint16_t> cpg_rel;
};

// Optional consumer of the inflated windows: called once per window with the window's bytes and the byte offsets of
// its complete records (rec_off[n_rec] = end of the last one); when given, the host does NOT decode the records
// itself (the device does, mth_decode_records).  Return false to abort (err is reported).
using WindowSink = std::function<bool(const uint8_t *buf, const uint64_t *rec_off, size_t n_rec, std::string &err)>;

// The file mapped read-only plus the table of its BGZF blocks that hold data (payload offset / payload size /
// inflated size), for a consumer that inflates on its own (the device).  false + err for anything that is not BGZF.
struct BgzfMap {
    const uint8_t *file = nullptr;
    size_t file_bytes = 0;
    std::vector<uint64_t> coff;
    std::vector<uint32_t> csize, isize;
    ~BgzfMap();
};
bool bgzf_map(const std::string &path, BgzfMap &out, std::string &err);

// header_bytes: uncompressed size of the BAM header (records start right after it).
// target: --cpg-set keys (tid << 32 | pos) or nullptr.  err_kind: 1 format/IO, 2 record without XM.
bool parallel_decode(const std::string &path, size_t header_bytes, const std::unordered_set<uint64_t> *target,
                     int nthreads, DecodedSoA &out, std::string &err, int &err_kind, const WindowSink *sink = nullptr,
                     int xm_min_mapq = 0 /* records without XM:Z are an error from this mapq up */);

}  // namespace mthh
