// bai_index.cpp -- the BAM index (.bai, SAM spec section 5.2) as a way to load ONE genomic region of a coordinate-sorted
// BAM: SURVEY 8(f).2.  The reference's fixtures ship tests/test{1..6}.bam.bai but its reader (bamutil.rs:4-11) never
// opens them; here `metheor <measure> --region chr:beg-end` asks the index which BGZF blocks can hold records that
// overlap the region (plus the halo the measures need on its left) and only those blocks go to the GPU.
// The answer has the shape of a shard plan (mth_host_shard_t): blocks to load + the owned (tid, pos) interval.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/metheor_host.h"
#include "bai_internal.h"

namespace mthh {

static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

bool BaiIndex::load(const std::string &path, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "Error opening BAM index " + path + ": file not found"; return false; }
    char magic[4];
    int32_t n_ref = 0;
    bool ok = rd(f, magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(f, &n_ref, 4) && n_ref >= 0;
    refs.clear();
    for (int32_t r = 0; ok && r < n_ref; ++r) {
        Ref ref;
        int32_t n_bin = 0;
        ok = rd(f, &n_bin, 4) && n_bin >= 0;
        for (int32_t b = 0; ok && b < n_bin; ++b) {
            uint32_t bin = 0;
            int32_t n_chunk = 0;
            ok = rd(f, &bin, 4) && rd(f, &n_chunk, 4) && n_chunk >= 0 && n_chunk < (1 << 28);
            for (int32_t c = 0; ok && c < n_chunk; ++c) {
                uint64_t be[2];
                ok = rd(f, be, 16);
                if (ok && bin != 37450u) ref.chunks.push_back(Chunk{bin, be[0], be[1]});     // 37450: samtools' metadata pseudo-bin
            }
        }
        int32_t n_intv = 0;
        ok = ok && rd(f, &n_intv, 4) && n_intv >= 0;
        if (ok) { ref.ioffset.resize((size_t)n_intv); ok = n_intv == 0 || rd(f, ref.ioffset.data(), (size_t)n_intv * 8); }
        if (ok) refs.push_back(std::move(ref));
    }
    fclose(f);
    if (!ok) { err = "Error reading BAM index " + path + ": not a .bai file or truncated"; refs.clear(); return false; }
    return true;
}

// SAM spec 5.3: the bins that can hold records overlapping [beg, end)
static void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t> &bins) {
    bins.clear();
    if (end <= beg) return;
    --end;
    bins.push_back(0);
    for (int k = 1 + (int)(beg >> 26); k <= 1 + (int)(end >> 26); ++k) bins.push_back((uint32_t)k);
    for (int k = 9 + (int)(beg >> 23); k <= 9 + (int)(end >> 23); ++k) bins.push_back((uint32_t)k);
    for (int k = 73 + (int)(beg >> 20); k <= 73 + (int)(end >> 20); ++k) bins.push_back((uint32_t)k);
    for (int k = 585 + (int)(beg >> 17); k <= 585 + (int)(end >> 17); ++k) bins.push_back((uint32_t)k);
    for (int k = 4681 + (int)(beg >> 14); k <= 4681 + (int)(end >> 14); ++k) bins.push_back((uint32_t)k);
}

// virtual file offsets [lo, hi) that hold every record overlapping [beg, end) of reference `tid`; false = none
bool BaiIndex::query(int32_t tid, int64_t beg, int64_t end, uint64_t &lo, uint64_t &hi) const {
    if (tid < 0 || (size_t)tid >= refs.size() || end <= beg) return false;
    const Ref &r = refs[(size_t)tid];
    // linear index: no record overlapping a 16-kbp window starts before ioffset[window] (windows past the last entry: nothing
    // reaches that far, so the last entry is still a valid lower bound)
    uint64_t min_off = 0;
    if (!r.ioffset.empty()) min_off = r.ioffset[std::min<size_t>((size_t)(beg >> 14), r.ioffset.size() - 1)];
    std::vector<uint32_t> bins;
    reg2bins(beg, end, bins);
    std::sort(bins.begin(), bins.end());
    bool any = false;
    for (const Chunk &c : r.chunks) {
        if (!std::binary_search(bins.begin(), bins.end(), c.bin) || c.end <= min_off) continue;
        const uint64_t b = std::max(c.beg, min_off);
        if (!any) { lo = b; hi = c.end; any = true; } else { lo = std::min(lo, b); hi = std::max(hi, c.end); }
    }
    return any;
}

}  // namespace mthh
