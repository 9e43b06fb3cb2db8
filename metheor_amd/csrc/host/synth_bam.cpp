// synth_bam.cpp -- fast writer of a synthetic Bismark-style BAM from the decoded SoA (bench / test
// tooling: the SURVEY's "seeded synthetic Bismark BAM generator").  `read_len`M reads on one or several contigs,
// XM:Z strings with z/Z at the call offsets, random sequence and quality bytes so that the file
// compresses like a real BAM.  Records are built and deflated (BGZF, zlib level 6) in parallel.
// Block layout as htslib writes it (bgzf_flush_try before every record): a block holds whole records, at most
// 0xff00 inflated bytes -- no record straddles two BGZF blocks.
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/metheor_host.h"

namespace {

void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

constexpr size_t BGZF_BLOCK = 0xff00;   // htslib's BGZF_BLOCK_SIZE
// append `n` uncompressed bytes as BGZF blocks (<= 0xff00 bytes each)
// (one deflate state per thread, reset per block: deflateInit2 allocates ~270 KB through mmap every time, and 256 threads doing that
// for every 64-KB block spent their time in the kernel's address-space lock -- a 100 M-read file took 68 s)
struct Deflater {
    z_stream zs;
    bool ok;
    Deflater() { memset(&zs, 0, sizeof zs); ok = deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK; }
    ~Deflater() { if (ok) deflateEnd(&zs); }
};
void bgzf_append(std::vector<uint8_t> &out, const uint8_t *p, size_t n) {
    static thread_local Deflater df;
    size_t o = 0;
    do {
        const size_t take = std::min<size_t>(BGZF_BLOCK, n - o);
        uint8_t comp[70000];
        z_stream &zs = df.zs;
        deflateReset(&zs);
        zs.next_in = const_cast<uint8_t *>(p + o); zs.avail_in = (uInt)take;
        zs.next_out = comp; zs.avail_out = sizeof comp;
        deflate(&zs, Z_FINISH);
        const size_t clen = sizeof comp - zs.avail_out;
        const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
        out.insert(out.end(), hdr, hdr + 12);
        out.push_back('B'); out.push_back('C'); put16(out, 2); put16(out, (uint32_t)(clen + 25));
        out.insert(out.end(), comp, comp + clen);
        put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), p + o, (uInt)take));
        put32(out, (uint32_t)take);
        o += take;
    } while (o < n);
}

inline uint64_t rng_next(uint64_t &s) { s += 0x9e3779b97f4a7c15ULL; uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

}  // namespace

// base_n > 0: the arrays describe base_n reads of ONE contig and record i of the file is read i % base_n on contig i / base_n (the same
// reads on several contigs: a large file without a large SoA; bench.py's 100 M-read end-to-end leg)
static int write_core(const char *path, int32_t n_contigs, const char *const *contigs, const int64_t *contig_lens,
                      int64_t n_reads, int32_t read_len, const int32_t *tid, const int32_t *start,
                      const uint8_t *fwd, const uint8_t *mapq, const uint64_t *cpg_off,
                      const uint16_t *cpg_rel, const uint32_t *cpg_pos, uint64_t seed, int nthreads, int64_t base_n) {
    if (!path || !contigs || !contig_lens || n_contigs < 1 || n_reads < 0 || read_len <= 0) return MTH_HOST_ERR_INVALID;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    std::vector<uint8_t> head;
    std::string text = "@HD\tVN:1.0\tSO:coordinate\n";
    for (int32_t c = 0; c < n_contigs; ++c) text += std::string("@SQ\tSN:") + contigs[c] + "\tLN:" + std::to_string(contig_lens[c]) + "\n";
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put32(head, (uint32_t)text.size()); head.insert(head.end(), text.begin(), text.end());
    put32(head, (uint32_t)n_contigs);
    for (int32_t c = 0; c < n_contigs; ++c) {
        put32(head, (uint32_t)strlen(contigs[c]) + 1); head.insert(head.end(), contigs[c], contigs[c] + strlen(contigs[c]) + 1);
        put32(head, (uint32_t)contig_lens[c]);
    }
    std::vector<std::vector<uint8_t>> parts((size_t)nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] {
        const int64_t r0 = n_reads * t / nthreads, r1 = n_reads * (t + 1) / nthreads;
        std::vector<uint8_t> raw;          // the block being filled (whole records)
        raw.reserve(BGZF_BLOCK + 4096);
        std::vector<uint8_t> &out = parts[(size_t)t];
        uint64_t rs = seed ^ (0x51ed27ULL * (uint64_t)(t + 1));
        const uint32_t seqb = (uint32_t)(read_len + 1) / 2;
        out.reserve((size_t)(r1 - r0) * 190 + 65536);
        for (int64_t fi = r0; fi < r1; ++fi) {
            const int64_t i = base_n > 0 ? fi % base_n : fi;
            char name[32];
            const int ln = snprintf(name, sizeof name, "r%lld", (long long)fi) + 1;
            const uint32_t aux_len = 4 + 3 + (uint32_t)read_len + 1 + 6;
            const uint32_t bs = 32 + (uint32_t)ln + 4 + seqb + (uint32_t)read_len + aux_len;
            if (!raw.empty() && raw.size() + 4 + bs > BGZF_BLOCK) { bgzf_append(out, raw.data(), raw.size()); raw.clear(); }   // bgzf_flush_try
            put32(raw, bs); put32(raw, base_n > 0 ? (uint32_t)(fi / base_n) : (tid ? (uint32_t)tid[i] : 0u)); put32(raw, (uint32_t)start[i]);
            raw.push_back((uint8_t)ln); raw.push_back(mapq[i]); put16(raw, 4680); put16(raw, 1);
            put16(raw, fwd[i] ? 0 : 16); put32(raw, (uint32_t)read_len); put32(raw, 0xffffffffu); put32(raw, 0xffffffffu); put32(raw, 0);
            raw.insert(raw.end(), name, name + ln);
            put32(raw, ((uint32_t)read_len << 4) | 0);
            for (uint32_t k = 0; k < seqb; k += 8) { uint64_t x = rng_next(rs); for (uint32_t q = 0; q < 8 && k + q < seqb; ++q, x >>= 8) raw.push_back((uint8_t)((1u << (x & 3)) << 4 | (1u << ((x >> 2) & 3)))); }
            for (int32_t k = 0; k < read_len; k += 8) { uint64_t x = rng_next(rs); for (int q = 0; q < 8 && k + q < read_len; ++q, x >>= 8) raw.push_back((uint8_t)(2 + (x & 0xff) % 39)); }
            raw.insert(raw.end(), {'N', 'M', 'C', 0, 'X', 'M', 'Z'});
            const size_t xo = raw.size();
            raw.insert(raw.end(), (size_t)read_len, (uint8_t)'.');
            for (uint64_t c = cpg_off[i]; c < cpg_off[i + 1]; ++c)
                if (cpg_rel[c] < (uint32_t)read_len) raw[xo + cpg_rel[c]] = (cpg_pos[c] >> 31) ? 'Z' : 'z';
            raw.push_back(0);
            raw.insert(raw.end(), {'X', 'R', 'Z', 'C', 'T', 0});
        }
        if (!raw.empty()) bgzf_append(out, raw.data(), raw.size());
    });
    for (auto &x : th) x.join();
    // header, the threads' parts at their offsets (written in parallel: one fwrite of a 100 M-read file is ten seconds), EOF block
    std::vector<uint8_t> hb;
    bgzf_append(hb, head.data(), head.size());
    static const uint8_t eof_blk[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint64_t> at(parts.size() + 1);
    at[0] = hb.size();
    for (size_t t = 0; t < parts.size(); ++t) at[t + 1] = at[t] + parts[t].size();
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return MTH_HOST_ERR_OPEN;
    auto put = [&](const uint8_t *p, size_t n, uint64_t off) {
        while (n) {
            const ssize_t w = pwrite(fd, p, n, (off_t)off);
            if (w <= 0) return false;
            p += w; n -= (size_t)w; off += (uint64_t)w;
        }
        return true;
    };
    std::vector<char> okv(parts.size(), 1);
    std::vector<std::thread> wr;
    for (size_t t = 0; t < parts.size(); ++t) wr.emplace_back([&, t] { okv[t] = put(parts[t].data(), parts[t].size(), at[t]) ? 1 : 0; });
    bool ok = put(hb.data(), hb.size(), 0) && put(eof_blk, 28, at[parts.size()]);
    for (auto &x : wr) x.join();
    for (char o : okv) ok = ok && o;
    ok = (close(fd) == 0) && ok;
    return ok ? MTH_HOST_OK : MTH_HOST_ERR_OPEN;
}

extern "C" int mth_host_write_synthetic_bam_multi(const char *path, int32_t n_contigs, const char *const *contigs, const int64_t *contig_lens,
                                                  int64_t n_reads, int32_t read_len, const int32_t *tid, const int32_t *start,
                                                  const uint8_t *fwd, const uint8_t *mapq, const uint64_t *cpg_off,
                                                  const uint16_t *cpg_rel, const uint32_t *cpg_pos, uint64_t seed, int nthreads) {
    return write_core(path, n_contigs, contigs, contig_lens, n_reads, read_len, tid, start, fwd, mapq, cpg_off, cpg_rel, cpg_pos, seed, nthreads, 0);
}

extern "C" int mth_host_write_synthetic_bam_repeat(const char *path, int32_t n_copies, const char *const *contigs, const int64_t *contig_lens,
                                                   int64_t base_reads, int32_t read_len, const int32_t *start, const uint8_t *fwd,
                                                   const uint8_t *mapq, const uint64_t *cpg_off, const uint16_t *cpg_rel,
                                                   const uint32_t *cpg_pos, uint64_t seed, int nthreads) {
    if (n_copies < 1 || base_reads < 1) return MTH_HOST_ERR_INVALID;
    return write_core(path, n_copies, contigs, contig_lens, base_reads * n_copies, read_len, nullptr, start, fwd, mapq, cpg_off, cpg_rel, cpg_pos, seed,
                      nthreads, base_reads);
}

extern "C" int mth_host_write_synthetic_bam(const char *path, const char *contig, int64_t contig_len, int64_t n_reads,
                                            int32_t read_len, const int32_t *start, const uint8_t *fwd, const uint8_t *mapq,
                                            const uint64_t *cpg_off, const uint16_t *cpg_rel, const uint32_t *cpg_pos,
                                            uint64_t seed, int nthreads) {
    if (!contig) return MTH_HOST_ERR_INVALID;
    return mth_host_write_synthetic_bam_multi(path, 1, &contig, &contig_len, n_reads, read_len, nullptr, start, fwd, mapq, cpg_off, cpg_rel,
                                              cpg_pos, seed, nthreads);
}
