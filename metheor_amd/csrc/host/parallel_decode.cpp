// parallel_decode.cpp -- multi-threaded BGZF inflate + BAM record decode (the step
// before the hot path, the end-to-end limiter).  Same results as the single-threaded walk in
// host_api.cpp (readutil.rs:24-53, 323-345, 87-95), in file order.
//
//   1. mmap the file, hop over the BGZF block headers (BSIZE) -> table of blocks with their
//      uncompressed offsets (sequential, a few microseconds per MB)
//   2. windows of consecutive blocks (512 MiB uncompressed; METHEOR_DECODE_WINDOW_MB overrides): threads inflate the window's blocks
//      independently into one contiguous buffer (CRC32 checked); the tail of a record cut by the window
//      end is carried to the front of the next window
//   3. one sequential walk over the 4-byte block_size fields finds the record starts (cheap)
//   4. threads decode disjoint record ranges into thread-local SoA pieces
//   5. pieces are appended in order (parallel memcpy at precomputed offsets)
#include "parallel_decode.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "bam_reader.h"

namespace mthh {

std::atomic<uint32_t> g_decode_notes{0};      // non-fatal findings of the host decode (bit 0: a CIGAR P operation), mth_host_notes()

namespace {

struct Block { size_t coff; uint32_t csize, isize; size_t uoff; };   // payload offset/size in the file, uncompressed size/offset

struct Piece {   // one thread's decoded records of a window
    std::vector<int32_t> tid, start, end;
    std::vector<uint8_t> mapq, fwd;
    std::vector<uint32_t> ncpg, cpg_pos;
    std::vector<uint16_t> cpg_rel;
    void clear() { tid.clear(); start.clear(); end.clear(); mapq.clear(); fwd.clear(); ncpg.clear(); cpg_pos.clear(); cpg_rel.clear(); }
};

// persistent worker pool: run(n, f) executes f(0..n-1) on the pool's threads (the caller is worker 0)
class Pool {
  public:
    explicit Pool(int n) : n_(std::max(1, n)) {
        for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &x : th_) x.join();
    }
    int size() const { return n_; }
    template <class F>
    void run(int n, F f) {
        if (n <= 1 || n_ == 1) { for (int t = 0; t < n; ++t) f(t); return; }
        fn_ = [&f](int t) { f(t); };
        { std::lock_guard<std::mutex> g(m_); active_ = std::min(n, n_); pending_ = active_ - 1; ++gen_; }
        cv_.notify_all();
        f(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }
  private:
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            const bool mine = t < active_;
            g.unlock();
            if (mine) {
                fn_(t);
                std::lock_guard<std::mutex> g2(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_, active_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::function<void(int)> fn_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
};

// 0 = found, 1 = malformed aux block, 2 = no XM:Z.  Offsets are 64-bit: a crafted B-array count must not wrap the
// cursor back into the block (htslib's skip_aux rejects such a record; so do we).  Either failure is the SAME
// failure in the reference: Record::aux(b"XM") returns Err for a bad block as for a missing tag, and
// readutil.rs:46-48 panics with the XM text -- decode_record reports both as "no XM".
int find_xm(const uint8_t *aux, uint32_t len, const char *&xm, uint32_t &xm_len) {
    uint64_t o = 0;
    while (o + 3 <= len) {
        const uint8_t t0 = aux[o], t1 = aux[o + 1], ty = aux[o + 2];
        o += 3;
        switch (ty) {
            case 'A': case 'c': case 'C': o += 1; break;
            case 's': case 'S': o += 2; break;
            case 'i': case 'I': case 'f': o += 4; break;
            case 'Z': case 'H': {
                const uint64_t b = o;
                while (o < len && aux[o] != 0) ++o;
                if (o >= len) return 1;
                if (ty == 'Z' && t0 == 'X' && t1 == 'M') { xm = reinterpret_cast<const char *>(aux + b); xm_len = (uint32_t)(o - b); return 0; }
                o += 1;
                break;
            }
            case 'B': {
                if (o + 5 > len) return 1;
                const uint8_t sub = aux[o];
                const uint64_t cnt = read_u32(aux + o + 1);
                uint64_t w;
                switch (sub) {
                    case 'c': case 'C': w = 1; break;
                    case 's': case 'S': w = 2; break;
                    case 'i': case 'I': case 'f': w = 4; break;
                    default: return 1;
                }
                o += 5 + cnt * w;
                if (o > len) return 1;
                break;
            }
            default: return 1;
        }
        if (o > len) return 1;
    }
    return 2;   // fewer than 3 bytes left: htslib's bam_aux_get ends its search the same way (ENOENT)
}

// one record (p points at the 32-byte fixed part, len = block_size) -> appended to the piece
// returns 0 ok, 1 corrupt, 2 no XM
int decode_record(const uint8_t *p, uint32_t len, const std::unordered_set<uint64_t> *target, int xm_min_mapq, Piece &out) {
    const int32_t tid = read_i32(p), pos = read_i32(p + 4);
    const uint32_t l_read_name = p[8], n_cigar = read_u16(p + 12), l_seq = read_u32(p + 16);
    const uint8_t mapq = p[9];
    const uint16_t flag = read_u16(p + 14);
    const size_t o_cigar = 32 + (size_t)l_read_name;
    const size_t o_aux = o_cigar + 4ull * n_cigar + ((size_t)l_seq + 1) / 2 + l_seq;
    if (o_aux > len) return 1;
    const char *xm = nullptr;
    uint32_t xm_len = 0;
    if (find_xm(p + o_aux, (uint32_t)(len - o_aux), xm, xm_len) != 0) {
        if ((int)mapq >= xm_min_mapq) return 2;
        xm = nullptr; xm_len = 0;                  // lpmd.rs:176-181: skipped by the mapq filter before BismarkRead::new -> no calls
    }
    const bool forward = flag == 0 || flag == 99 || flag == 147;   // readutil.rs:332
    int32_t first = -1, last = -1;
    int64_t r = pos;
    uint32_t q = 0, n = 0;
    const uint8_t *cg = p + o_cigar;
    for (uint32_t c = 0; c < n_cigar; ++c) {
        const uint32_t w = read_u32(cg + 4 * c), op = w & 15u, ln = w >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (ln) { if (first < 0) first = (int32_t)r; last = (int32_t)(r + ln - 1); }
            const uint32_t qe = q + ln;
            for (; q < qe; ++q, ++r) {
                if (q >= xm_len) continue;
                const char ch = xm[q];
                if (ch != 'z' && ch != 'Z') continue;
                const int32_t ap = forward ? (int32_t)r : (int32_t)(r - 1);
                if (target && !target->count(((uint64_t)(uint32_t)tid << 32) | (uint32_t)ap)) continue;
                out.cpg_pos.push_back(((uint32_t)ap & 0x7fffffffu) | (ch == 'Z' ? 0x80000000u : 0u));
                out.cpg_rel.push_back((uint16_t)q);
                ++n;
            }
        } else if (op == 1 || op == 4) {
            q += ln;
        } else if (op == 2 || op == 3) {
            r += ln;
        } else if (op == 6) {
            g_decode_notes.fetch_or(1u, std::memory_order_relaxed);      // P: see MTH_NOTE_CIGAR_PAD (include/metheor_hip.h)
        }
    }
    out.tid.push_back(tid); out.start.push_back(first); out.end.push_back(last);
    out.mapq.push_back(mapq); out.fwd.push_back(forward ? 1 : 0); out.ncpg.push_back(n);
    return 0;
}

}  // namespace

namespace {
struct Stopwatch {   // METHEOR_TIMING=1: decoder phase totals on stderr
    double acc[6] = {0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int k) { const auto n = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double>(n - t).count(); t = n; }
    ~Stopwatch() {
        if (!getenv("METHEOR_TIMING")) return;
        static const char *nm[6] = {"block table", "inflate", "record walk", "record decode", "append", "other"};
        for (int k = 0; k < 6; ++k) fprintf(stderr, "[metheor timing]     decode/%-14s %.3f s\n", nm[k], acc[k]);
    }
};
}  // namespace

BgzfMap::~BgzfMap() { if (file && file_bytes) munmap(const_cast<uint8_t *>(file), file_bytes); }

// The block-table walk below hops from one BGZF header to the next: the first touch of every page of the mapping, one page fault at
// a time -- 0.7 s of a 17.8-GB file's 1.8-s run (bench.py e2e "large": startup_s 0.78 against 0.10 on the 1.8-GB file; VERDICT r05
// item 6).  The mapping is populated first, in slices, by as many threads as the host has (page-cache hits: the faults are
// independent): MADV_POPULATE_READ where the kernel has it (5.14), one read per page otherwise.
static void prefault_parallel(const uint8_t *file, size_t fsz) {
    if (fsz < ((size_t)64 << 20) || getenv("METHEOR_NO_PREFAULT")) return;
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt ? nt : 4u, 32u));
    const size_t page = 4096, slice = (((fsz + nt - 1) / nt) + page - 1) / page * page;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t b = (size_t)t * slice, e = std::min(fsz, b + slice);
        if (b >= e) break;
        th.emplace_back([file, b, e, page]() {
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
            if (madvise(const_cast<uint8_t *>(file) + b, e - b, MADV_POPULATE_READ) == 0) return;
            volatile uint8_t sink = 0;
            for (size_t o = b; o < e; o += page) sink = sink + file[o];
            (void)sink;
        });
    }
    for (auto &x : th) x.join();
}

bool bgzf_map(const std::string &path, BgzfMap &out, std::string &err) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); err = "cannot stat " + path; return false; }
    const size_t fsz = (size_t)st.st_size;
    const uint8_t *file = fsz ? static_cast<const uint8_t *>(mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0)) : nullptr;
    close(fd);
    if (fsz && file == MAP_FAILED) { err = "cannot mmap " + path; return false; }
    out.file = file; out.file_bytes = fsz;
    out.coff.clear(); out.csize.clear(); out.isize.clear();
    prefault_parallel(file, fsz);
    size_t o = 0;
    while (o < fsz) {
        if (o + 18 > fsz || file[o] != 31 || file[o + 1] != 139 || file[o + 2] != 8 || !(file[o + 3] & 4)) { err = "not a BGZF file (bad block header)"; return false; }
        const uint32_t xlen = read_u16(file + o + 10);
        if (o + 12 + xlen > fsz) { err = "truncated BGZF block"; return false; }
        int bsize = -1;
        for (uint32_t e = 0; e + 4 <= xlen;) {
            const uint8_t *x = file + o + 12 + e;
            const uint32_t slen = read_u16(x + 2);
            if (x[0] == 'B' && x[1] == 'C' && slen == 2 && e + 6 <= xlen) bsize = read_u16(x + 4);
            e += 4 + slen;
        }
        if (bsize < 0) { err = "not a BGZF file (no BC subfield)"; return false; }
        const size_t total = (size_t)bsize + 1;
        if (total < 12 + xlen + 8 || o + total > fsz) { err = "truncated BGZF block"; return false; }
        const uint32_t isize = read_u32(file + o + total - 4);
        if (isize) { out.coff.push_back(o + 12 + xlen); out.csize.push_back((uint32_t)(total - 12 - xlen - 8)); out.isize.push_back(isize); }
        o += total;
    }
    return true;
}

bool parallel_decode(const std::string &path, size_t header_bytes, const std::unordered_set<uint64_t> *target,
                     int nthreads, DecodedSoA &out, std::string &err, int &err_kind, const WindowSink *sink, int xm_min_mapq) {
    err_kind = 0;
    Stopwatch sw;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; err_kind = 1; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); err = "cannot stat " + path; err_kind = 1; return false; }
    const size_t fsz = (size_t)st.st_size;
    const uint8_t *file = fsz ? static_cast<const uint8_t *>(mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0)) : nullptr;
    close(fd);
    if (fsz && file == MAP_FAILED) { err = "cannot mmap " + path; err_kind = 1; return false; }
    struct Unmap { const uint8_t *p; size_t n; ~Unmap() { if (p && n) munmap(const_cast<uint8_t *>(p), n); } } unmap{file, fsz};

    // 1. block table
    std::vector<Block> blocks;
    size_t o = 0, uoff = 0;
    while (o < fsz) {
        if (o + 18 > fsz || file[o] != 31 || file[o + 1] != 139 || file[o + 2] != 8 || !(file[o + 3] & 4)) { err = "not a BGZF file (bad block header)"; err_kind = 1; return false; }
        const uint32_t xlen = read_u16(file + o + 10);
        if (o + 12 + xlen > fsz) { err = "truncated BGZF block"; err_kind = 1; return false; }
        int bsize = -1;
        for (uint32_t e = 0; e + 4 <= xlen;) {
            const uint8_t *x = file + o + 12 + e;
            const uint32_t slen = read_u16(x + 2);
            if (x[0] == 'B' && x[1] == 'C' && slen == 2 && e + 6 <= xlen) bsize = read_u16(x + 4);
            e += 4 + slen;
        }
        if (bsize < 0) { err = "not a BGZF file (no BC subfield)"; err_kind = 1; return false; }
        const size_t total = (size_t)bsize + 1;
        if (total < 12 + xlen + 8 || o + total > fsz) { err = "truncated BGZF block"; err_kind = 1; return false; }
        const uint32_t isize = read_u32(file + o + total - 4);
        if (isize) blocks.push_back(Block{o + 12 + xlen, (uint32_t)(total - 12 - xlen - 8), isize, uoff});
        uoff += isize;
        o += total;
    }
    const size_t total_u = uoff;
    sw.lap(0);
    if (header_bytes > total_u) { err = "truncated BAM file"; err_kind = 1; return false; }

    nthreads = std::max(1, nthreads);
    Pool pool(nthreads);
    std::vector<Piece> pieces((size_t)nthreads);
    // not a vector: no zero-fill of hundreds of MB per window
    std::unique_ptr<uint8_t[]> bufmem;
    size_t bufcap = 0, buflen = 0;
    std::vector<uint8_t> carry;
    std::vector<uint64_t> rec_off;
    std::atomic<int> fail{0};
    size_t WINDOW = 512u << 20;
    if (const char *e = getenv("METHEOR_DECODE_WINDOW_MB")) { const long k = atol(e); if (k >= 1 && k <= 65536) WINDOW = (size_t)k << 20; }
    size_t bi = 0;
    // skip whole blocks that lie entirely inside the header
    while (bi < blocks.size() && blocks[bi].uoff + blocks[bi].isize <= header_bytes) ++bi;
    size_t skip = bi < blocks.size() ? header_bytes - blocks[bi].uoff : 0;   // header bytes inside the first data block
    out.cpg_off.assign(1, 0);
    while (bi < blocks.size() || !carry.empty()) {
        // 2. the window's blocks
        size_t be = bi, wbytes = 0;
        while (be < blocks.size() && (wbytes == 0 || wbytes + blocks[be].isize <= WINDOW)) { wbytes += blocks[be].isize; ++be; }
        if (be == bi) {   // only a carried partial record is left: the file ends inside a record
            err = "truncated BAM file"; err_kind = 1; return false;
        }
        const size_t c0 = carry.size();
        buflen = c0 + wbytes;
        if (buflen > bufcap) { bufmem.reset(new uint8_t[buflen]); bufcap = buflen; }
        uint8_t *const buf = bufmem.get();
        if (c0) memcpy(buf, carry.data(), c0);
        const size_t ubase = blocks[bi].uoff;
        std::atomic<size_t> next{bi};
        // per-block completion flags: the calling thread walks record boundaries over the contiguous
        // prefix of finished blocks WHILE the other threads inflate (the walk is inherently sequential:
        // each record's start is known only from the previous record's block_size)
        const size_t nblk = be - bi;
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[nblk]);
        for (size_t k = 0; k < nblk; ++k) done[k].store(0, std::memory_order_relaxed);
        rec_off.clear();
        size_t p = c0 ? 0 : skip;
        bool walk_bad = false;
        auto inflate_some = [&]() {
            z_stream zs;
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b >= be || fail.load()) break;
                const Block &k = blocks[b];
                uint8_t *dst = buf + c0 + (k.uoff - ubase);
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { fail = 1; break; }
                zs.next_in = const_cast<uint8_t *>(file + k.coff); zs.avail_in = k.csize;
                zs.next_out = dst; zs.avail_out = k.isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.avail_out != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, k.isize) != read_u32(file + k.coff + k.csize)) { fail = 1; break; }
                done[b - bi].store(1, std::memory_order_release);
            }
        };
        auto walk = [&]() {   // caller thread
            size_t ready_blocks = 0, avail = c0;      // bytes of buf known to be complete
            for (;;) {
                while (ready_blocks < nblk && done[ready_blocks].load(std::memory_order_acquire)) {
                    avail += blocks[bi + ready_blocks].isize;
                    ++ready_blocks;
                }
                while (p + 4 <= avail) {
                    const int32_t bs = read_i32(buf + p);
                    if (bs < 32) { walk_bad = true; return; }
                    if (p + 4 + (size_t)bs > avail) break;
                    rec_off.push_back(p);
                    p += 4 + (size_t)bs;
                }
                if (ready_blocks == nblk || fail.load()) return;
                if (!done[ready_blocks].load(std::memory_order_acquire)) std::this_thread::yield();
            }
        };
        if (nthreads >= 4) {
            pool.run(nthreads, [&](int t) { if (t == 0) walk(); else inflate_some(); });
            if (!fail.load() && !walk_bad) { next.store(be); }   // all blocks were claimed; nothing left
        } else {
            pool.run(nthreads, [&](int) { inflate_some(); });
            if (!fail.load()) walk();
        }
        if (fail.load()) { err = "corrupt BGZF block"; err_kind = 1; return false; }
        if (walk_bad) { err = "corrupt BAM record"; err_kind = 1; return false; }
        sw.lap(1);
        const size_t end = buflen;
        carry.assign(buf + p, buf + end);
        bi = be;
        skip = 0;
        sw.lap(2);
        if (sink) {   // the records of this window go to the consumer (device decode) instead of the host threads
            const size_t n_win = rec_off.size();
            rec_off.push_back(p);
            if (!(*sink)(buf, rec_off.data(), n_win, err)) { err_kind = err_kind ? err_kind : 3; return false; }
            sw.lap(3);
            if (bi >= blocks.size() && !carry.empty()) { err = "truncated BAM file"; err_kind = 1; return false; }
            continue;
        }
        // 4. decode
        const size_t nrec = rec_off.size();
        const int nt = (int)std::min<size_t>((size_t)nthreads, std::max<size_t>(1, nrec / 2048));
        std::atomic<int> xm_missing{0}, corrupt{0};
        pool.run(nt, [&](int t) {
            Piece &pc = pieces[(size_t)t];
            pc.clear();
            const size_t r0 = nrec * (size_t)t / (size_t)nt, r1 = nrec * (size_t)(t + 1) / (size_t)nt;
            for (size_t r = r0; r < r1; ++r) {
                const uint8_t *q = buf + rec_off[r];
                const int rc = decode_record(q + 4, (uint32_t)read_i32(q), target, xm_min_mapq, pc);
                if (rc == 1) { corrupt = 1; return; }
                if (rc == 2) { xm_missing = 1; return; }
            }
        });
        if (corrupt.load()) { err = "corrupt BAM record"; err_kind = 1; return false; }
        if (xm_missing.load()) { err = "Error reading XM tag in BAM record. Make sure the reads are aligned using Bismark!"; err_kind = 2; return false; }
        sw.lap(3);
        // 5. append in order
        size_t add_r = 0, add_c = 0;
        std::vector<size_t> ro((size_t)nt + 1, 0), co((size_t)nt + 1, 0);
        for (int t = 0; t < nt; ++t) { ro[(size_t)t + 1] = ro[(size_t)t] + pieces[(size_t)t].tid.size(); co[(size_t)t + 1] = co[(size_t)t] + pieces[(size_t)t].cpg_pos.size(); }
        add_r = ro[(size_t)nt]; add_c = co[(size_t)nt];
        const size_t R0 = out.tid.size(), C0 = out.cpg_pos.size();
        out.tid.resize(R0 + add_r); out.start.resize(R0 + add_r); out.end.resize(R0 + add_r);
        out.mapq.resize(R0 + add_r); out.fwd.resize(R0 + add_r); out.cpg_off.resize(R0 + add_r + 1);
        out.cpg_pos.resize(C0 + add_c); out.cpg_rel.resize(C0 + add_c);
        pool.run(nt, [&](int t) {
            const Piece &pc = pieces[(size_t)t];
            const size_t r = R0 + ro[(size_t)t], cc = C0 + co[(size_t)t], n = pc.tid.size();
            if (n) {
                memcpy(&out.tid[r], pc.tid.data(), n * 4); memcpy(&out.start[r], pc.start.data(), n * 4);
                memcpy(&out.end[r], pc.end.data(), n * 4); memcpy(&out.mapq[r], pc.mapq.data(), n);
                memcpy(&out.fwd[r], pc.fwd.data(), n);
                uint64_t acc = cc;
                for (size_t i = 0; i < n; ++i) { acc += pc.ncpg[i]; out.cpg_off[r + i + 1] = acc; }
            }
            if (!pc.cpg_pos.empty()) {
                memcpy(&out.cpg_pos[cc], pc.cpg_pos.data(), pc.cpg_pos.size() * 4);
                memcpy(&out.cpg_rel[cc], pc.cpg_rel.data(), pc.cpg_rel.size() * 2);
            }
        });
        sw.lap(4);
        if (bi >= blocks.size() && !carry.empty()) { err = "truncated BAM file"; err_kind = 1; return false; }
    }
    return true;
}

}  // namespace mthh
