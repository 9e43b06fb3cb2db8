// cli_main.cpp -- the `metheor` executable: drop-in command line of dohlee/metheor v0.1.9 for the
// measures that run on the MI355X path.  Same subcommands, short/long flags and defaults as the
// reference's clap definitions (src/lib.rs:19-231), same TSV bytes (pdr.rs:95-116, lpmd.rs:137-151),
// same failure behaviour the reference's tests look for (tests/cli_error_handling.rs, tests/*-cli.rs):
// usage errors exit 2 with clap-style text on stderr; run-time failures (the reference panics) exit
// 101 with the panic message on stderr.  There is no CPU fallback: a measure whose device kernel is
// not built yet fails loudly.
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "../../../include/metheor_hip.h"
#include "../../../include/metheor_host.h"

namespace {

struct Opt {
    const char *long_name;
    char short_name;
    const char *value_name;   // upper-case placeholder
    const char *help;
    const char *def;          // nullptr: no default
    bool required;
    char kind;                // 's' string, 'B' u8, 'U' u32, 'Z' usize, 'I' i32
};
struct Cmd {
    const char *name;
    const char *about;
    std::vector<Opt> opts;
};

const Opt O_IN = {"input", 'i', "INPUT", "Input BAM file", nullptr, true, 's'};
const Opt O_CPG = {"cpg-set", 'c', "CPG_SET", "(Optional) Specify a predefined set of CpGs (in BED file) to be analyzed", nullptr, false, 's'};
// not in the reference (outer row): one run split over the GPUs of the node by genomic region
const Opt O_GPUS = {"gpus", 'G', "GPUS", "(MI355X extension) Number of GPUs to split the run over, by genomic region", "1", false, 'U'};
// not in the reference either: the .bai files its fixtures ship finally get a reader
const Opt O_REGION = {"region", 'r', "REGION", "(MI355X extension) Only this region: chr or chr:beg-end (1-based, inclusive); needs the BAM index", nullptr, false, 's'};
const Opt O_BAI = {"bai", 'b', "BAI", "(MI355X extension) BAM index for --region [default: <input>.bai]", nullptr, false, 's'};

// lib.rs:24-231
const std::vector<Cmd> &commands() {
    static const std::vector<Cmd> c = {
        {"pdr", "Compute proportion of discordant reads (PDR)",
         {O_IN, {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of PDR calculation", nullptr, true, 's'},
          {"min-depth", 'd', "MIN_DEPTH", "Minimum depth of CpG stretches to consider", "10", false, 'U'},
          {"min-cpgs", 'p', "MIN_CPGS", "Minimum number of consecutive CpGs in a CpG stretch to consider", "4", false, 'Z'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"pm", "Compute epipolymorphism",
         {O_IN, {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of PM calculation", nullptr, true, 's'},
          {"min-depth", 'd', "MIN_DEPTH", "Minimum depth of CpG quartets to consider", "10", false, 'U'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"me", "Compute methylation entropy",
         {O_IN, {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of PDR calculation", nullptr, true, 's'},
          {"min-depth", 'd', "MIN_DEPTH", "Minimum depth of CpG quartets to consider", "10", false, 'U'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"fdrp", "Compute fraction of discordant read pairs (FDRP)",
         {{"input", 'i', "INPUT", "Path to input BAM file", nullptr, true, 's'},
          {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of FDRP calculation", nullptr, true, 's'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'},
          {"min-depth", 'd', "MIN_DEPTH", "Minimum number of reads mapped to a CpG in order to be considered", "10", false, 'Z'},
          {"max-depth", 'D', "MAX_DEPTH", "Maximum number of reads to consider", "40", false, 'Z'},
          {"min-overlap", 'l', "MIN_OVERLAP", "Minimum overlap between two reads to consider in bp", "35", false, 'I'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"qfdrp", "Compute quantitative fraction of discordant read pairs (qFDRP)",
         {{"input", 'i', "INPUT", "Path to input BAM file", nullptr, true, 's'},
          {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of FDRP calculation", nullptr, true, 's'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'},
          {"min-depth", 'd', "MIN_DEPTH", "Minimum number of reads mapped to a CpG in order to be considered", "10", false, 'Z'},
          {"max-depth", 'D', "MAX_DEPTH", "Maximum number of reads to consider", "40", false, 'Z'},
          {"min-overlap", 'l', "MIN_OVERLAP", "Minimum overlap between two reads to consider in bp", "35", false, 'I'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"mhl", "Compute methylation haplotype load (MHL)",
         {O_IN, {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of MHL calculation", nullptr, true, 's'},
          {"min-depth", 'd', "MIN_DEPTH", "Minimum depth of CpG stretches to consider", "10", false, 'U'},
          {"min-cpgs", 'p', "MIN_CPGS", "Minimum number of consecutive CpGs in a CpG stretch to consider", "4", false, 'Z'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"lpmd", "Compute local pairwise methylation discordance (LPMD)",
         {{"input", 'i', "INPUT", "Path to input BAM file", nullptr, true, 's'},
          {"output", 'o', "OUTPUT", "Path to output table file summarizing the result of LPMD calculation", nullptr, true, 's'},
          {"pairs", 'p', "PAIRS", "(Optional) Concordance information for all CpG pairs", nullptr, false, 's'},
          {"min-distance", 'm', "MIN_DISTANCE", "Minimum distance between CpG pairs to consider", "2", false, 'I'},
          {"max-distance", 'M', "MAX_DISTANCE", "Maximum distance between CpG pairs to consider", "16", false, 'I'},
          {"min-qual", 'q', "MIN_QUAL", "Minimum quality for a read to be considered", "10", false, 'B'}, O_CPG, O_GPUS, O_REGION, O_BAI}},
        {"tag", "Add bismark XM tag to BAM file",
         {{"input", 'i', "INPUT", "", nullptr, true, 's'}, {"output", 'o', "OUTPUT", "", nullptr, true, 's'},
          {"genome", 'g', "GENOME", "", nullptr, true, 's'}}},
    };
    return c;
}

void print_main_help(FILE *f) {
    fprintf(f, "Summarizes the heterogeneity of DNA methylation states using BAM files.\n\nUsage: metheor <COMMAND>\n\nCommands:\n");
    for (const Cmd &c : commands()) fprintf(f, "  %-6s %s\n", c.name, c.about);
    fprintf(f, "  help   Print this message or the help of the given subcommand(s)\n\nOptions:\n"
               "  -h, --help     Print help\n  -V, --version  Print version\n");
}

std::string usage_line(const Cmd &c) {
    std::string u = std::string("Usage: metheor ") + c.name + " [OPTIONS]";
    for (const Opt &o : c.opts) if (o.required) u += std::string(" --") + o.long_name + " <" + o.value_name + ">";
    return u;
}

void print_cmd_help(FILE *f, const Cmd &c) {
    fprintf(f, "%s\n\n%s\n\nOptions:\n", c.about, usage_line(c).c_str());
    for (const Opt &o : c.opts) {
        std::string left = std::string("  -") + o.short_name + ", --" + o.long_name + " <" + o.value_name + ">";
        fprintf(f, "%-34s %s%s%s%s\n", left.c_str(), o.help, o.def ? " [default: " : "", o.def ? o.def : "", o.def ? "]" : "");
    }
    fprintf(f, "  -h, --help                       Print help\n");
}

[[noreturn]] void usage_error(const Cmd *c, const std::string &msg) {
    fprintf(stderr, "error: %s\n\n%s\n\nFor more information, try '--help'.\n", msg.c_str(),
            c ? usage_line(*c).c_str() : "Usage: metheor <COMMAND>");
    exit(2);
}

// the reference panics on run-time failures: message on stderr, exit status 101
// (with --gpus N the failing shard's thread ends the process: _exit, the other shards' threads are mid-run)
std::atomic<bool> g_threads{false};
[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "%s\n", msg.c_str());
    if (g_threads.load()) { fflush(nullptr); _exit(101); }
    exit(101);
}

bool parse_number(const Opt &o, const std::string &v, int64_t &out, std::string &why) {
    if (v.empty()) { why = "cannot parse integer from empty string"; return false; }
    char *end = nullptr;
    errno = 0;
    const long long x = strtoll(v.c_str(), &end, 10);
    if (*end != 0 || errno) { why = "invalid digit found in string"; return false; }
    int64_t lo = 0, hi = 0;
    switch (o.kind) {
        case 'B': lo = 0; hi = 255; break;
        case 'U': lo = 0; hi = 4294967295LL; break;
        case 'Z': lo = 0; hi = INT64_MAX; break;
        case 'I': lo = INT32_MIN; hi = INT32_MAX; break;
        default: break;
    }
    if (x < lo || x > hi) {
        why = o.kind == 'B' ? v + " is not in 0..=255"
            : (x < 0 && o.kind != 'I') ? "invalid digit found in string" : "number too large to fit in target type";
        return false;
    }
    out = x;
    return true;
}

struct Args {
    std::map<std::string, std::string> s;
    std::map<std::string, int64_t> n;
    bool has(const char *k) const { return s.count(k) != 0; }
};

Args parse_args(const Cmd &c, int argc, char **argv, int first) {
    Args a;
    if (first >= argc) { print_cmd_help(stderr, c); exit(2); }   // arg_required_else_help
    for (int i = first; i < argc; ++i) {
        std::string tok = argv[i];
        if (tok == "-h" || tok == "--help") { print_cmd_help(stdout, c); exit(0); }
        const Opt *o = nullptr;
        std::string val;
        bool have_val = false;
        if (tok.rfind("--", 0) == 0) {
            const size_t eq = tok.find('=');
            const std::string name = tok.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            for (const Opt &x : c.opts) if (name == x.long_name) o = &x;
            if (eq != std::string::npos) { val = tok.substr(eq + 1); have_val = true; }
        } else if (tok.size() >= 2 && tok[0] == '-') {
            for (const Opt &x : c.opts) if (tok[1] == x.short_name) o = &x;
            if (o && tok.size() > 2) { val = tok.substr(tok[2] == '=' ? 3 : 2); have_val = true; }
        }
        if (!o) usage_error(&c, "unexpected argument '" + tok + "' found");
        if (!have_val) {
            if (i + 1 >= argc) usage_error(&c, std::string("a value is required for '--") + o->long_name + " <" + o->value_name + ">' but none was supplied");
            val = argv[++i];
            // clap (no allow_hyphen_values / allow_negative_numbers in lib.rs): a separate token that
            // looks like a flag is never taken as a value ("-5" after --min-depth); --opt=-5 is
            if (val.size() > 1 && val[0] == '-')
                usage_error(&c, "unexpected argument '" + val + "' found");
        }
        if (o->kind == 's') a.s[o->long_name] = val;
        else {
            int64_t x;
            std::string why;
            if (!parse_number(*o, val, x, why))
                usage_error(&c, "invalid value '" + val + "' for '--" + o->long_name + " <" + o->value_name + ">': " + why);
            a.n[o->long_name] = x;
            a.s[o->long_name] = val;
        }
    }
    std::string missing;
    for (const Opt &o : c.opts) {
        if (o.required && !a.has(o.long_name)) missing += std::string("\n  --") + o.long_name + " <" + o.value_name + ">";
        if (o.def && !a.has(o.long_name)) { a.n[o.long_name] = strtoll(o.def, nullptr, 10); }
    }
    if (!missing.empty()) usage_error(&c, "the following required arguments were not provided:" + missing);
    return a;
}

// ---- decoded input as per-contig batches --------------------------------------------------------
// A contig's reads are a contiguous slice of the decoded SoA (the file is coordinate-sorted), so the
// batch points straight into the decoder's arrays; only the CSR offsets are rebased (4 B/read).  A
// contig that contains a read without any aligned base (start = -1) is copied without those reads.
// --gpus N: the run is N shards, one host thread and one device context each (shard r on device r mod the devices
// present, or mod METHEOR_DEVICES).  Every shard loads its own run of BGZF blocks (mth_host_plan_shard: no index, no
// router), owns a (tid, pos) interval and formats its rows into memory; the parts concatenated in shard order are the
// unsharded file.  The only exchange is the RCCL all-reduce of the four LPMD counters (mth_allreduce_lpmd).
// METHEOR_SHARD_HALO: bp of reads loaded before the interval (default 65536; must cover the longest alignment + FDRP's
// 201-bp window, checked).
struct Shard {
    int rank = 0, world = 1, device = 0;
    int64_t halo = 65536;
    mth_host_shard_t plan;
    int xm_min_mapq = 0;
    // --region chr[:beg-end]: the plan comes from the .bai (mth_host_plan_region) instead of a byte-range cut
    bool region = false;
    std::string region_name, bai;
    int32_t r_beg = 0, r_end = INT32_MAX;          // 0-based [r_beg, r_end); INT32_MAX = to the end of the contig's data
    bool order_free = false;                       // set by lpmd / me / pm: their result does not depend on the record order (see load())
    bool planned() const { return world > 1 || region; }
};
thread_local Shard g_shard;

// what one shard contributes to the output files (filled through open_memstream when world > 1)
struct Part { char *out = nullptr; size_t out_len = 0; char *pairs = nullptr; size_t pairs_len = 0; };
thread_local Part *g_part = nullptr;

// the shards' threads meet here once: the last to arrive runs the collective for all of them
struct Gang {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<mth_ctx_t *> ctxs;
    int arrived = 0, rc = MTH_OK;
    bool done = false;
} g_gang;

struct Contig {
    int32_t tid;
    int32_t region_beg = 0, region_end = -1;   // owned positions (-1: to the contig's end); a shard may own part of a contig
    const int32_t *start, *end;
    const uint8_t *mapq;
    const uint32_t *pos;
    const uint16_t *rel;
    size_t n_reads, n_cpgs;
    std::vector<uint32_t> off;
    int32_t max_span = 0, max_end = -1;
    uint64_t r0 = 0, r1 = 0;   // device-decoded input: the contig's reads in the decoded stream
    // owned copies (only when filtering was needed)
    std::vector<int32_t> o_start, o_end;
    std::vector<uint8_t> o_mapq;
    std::vector<uint32_t> o_pos;
    std::vector<uint16_t> o_rel;
};

struct Input {
    mth_host_t *h = nullptr;
    std::vector<Contig> contigs;
    // the reads re-ordered by (tid, start) -- order-free measures on an input that is not coordinate-sorted (sort_reads)
    std::vector<int32_t> s_tid, s_start, s_end;
    std::vector<uint8_t> s_mapq;
    std::vector<uint64_t> s_off;
    std::vector<uint32_t> s_pos;
    std::vector<uint16_t> s_rel;
    mth_ctx_t *ctx = nullptr;   // set when the records were decoded on the device (the batches live in its HBM)
    bool device = false;
    bool file_order = false;               // pdr / mhl / fdrp / qfdrp on input that is not coordinate-sorted: no batches, the decoded stream itself (mth_fileorder_run)
    std::vector<uint8_t> unbatched_mapq;   // mapq of the records that entered no batch (no contig / no aligned base): lpmd.rs:176-179 counts them
};

// METHEOR_TIMING=1: phase wall times on stderr (never stdout)
struct Phase {
    const char *name;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Phase(const char *n) : name(n) {}
    ~Phase() {
        if (!getenv("METHEOR_TIMING")) return;
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[metheor timing] %-28s %.3f s\n", name, s);
    }
};

// End of a run, output files closed.  Releasing gigabytes of device buffers, unmapping the input and unloading the HIP
// runtime cost tens of milliseconds that nobody reads the result of: the process leaves through _exit and the driver
// reclaims everything (METHEOR_TEARDOWN=1 keeps the orderly release, e.g. under a leak checker).
int finish(mth_ctx_t *ctx, mth_host_t *h) {
    {   // a CIGAR P operation was decoded: this engine takes it as "no query base, no reference base" (the SAM specification's meaning);
        // what the reference's BAM crate does with it cannot be checked here -- the note says what THIS engine did
        static std::atomic<bool> said{false};
        if ((((ctx ? mth_notes(ctx) : 0u) | mth_host_notes()) & MTH_NOTE_CIGAR_PAD) && !said.exchange(true))
            fprintf(stderr, "metheor: note: the input holds CIGAR 'P' (padding) operations; they were taken as covering no query base and no "
                            "reference base\n");
    }
    if (g_shard.world > 1) return 0;      // a shard's thread: main() writes the files and leaves
    if (getenv("METHEOR_TEARDOWN")) {
        Phase ph("teardown");
        { Phase p1("  device context destroy"); mth_ctx_destroy(ctx); }
        { Phase p2("  input close (munmap)"); mth_host_close(h); }
        return 0;
    }
    fflush(nullptr);
    _exit(0);
}

void check(mth_ctx_t *ctx, int rc) {
    if (rc == MTH_OK) return;
    std::string m = std::string("metheor (MI355X path): ") + mth_strerror(rc);
    if (ctx && mth_last_error(ctx)[0]) m += std::string(" -- ") + mth_last_error(ctx);
    die(m);
}

// HIP initialisation takes 70-140 ms: on the device load path it runs on its own thread while the main thread opens the
// file, parses --cpg-set and builds the BGZF block table.  Errors are reported by whoever joins (file errors first).
struct CtxFuture {
    std::thread th;
    mth_ctx_t *ctx = nullptr;
    int rc = MTH_OK;
    void start() {
        const int device = g_shard.device;
        th = std::thread([this, device] {
            Phase ph("device context (overlapped)");
            rc = mth_ctx_create(device, &ctx);
        });
    }
    void wait() { if (th.joinable()) th.join(); }
    mth_ctx_t *get() {
        wait();
        if (rc != MTH_OK) die(std::string("metheor (MI355X path): ") + mth_strerror(rc));
        return ctx;
    }
};

mth_ctx_t *make_ctx() {
    Phase ph("device context");
    mth_ctx_t *ctx = nullptr;
    const int rc = mth_ctx_create(g_shard.device, &ctx);
    if (rc != MTH_OK) die(std::string("metheor (MI355X path): ") + mth_strerror(rc));
    return ctx;
}

// Device-side record decode (mth_decode_records): the host inflates BGZF and walks the record boundaries, every
// window goes to the GPU as raw bytes, and the SoA the measures read is built in HBM -- no record / XM parsing and
// no SoA assembly on host threads (--cpg-set is applied by the decode kernel).  Not used with
// METHEOR_HOST_DECODE=1, or when the file has records the batches cannot hold as they are (no contig, no aligned
// base, contigs not grouped): those fall back to the host decoder below.
struct StreamState { mth_ctx_t *ctx = nullptr; bool first = true; int rc = MTH_OK; };
int window_to_device(void *user, const uint8_t *buf, const uint64_t *rec_off, uint64_t n_rec) {
    StreamState *st = static_cast<StreamState *>(user);
    mth_decoded_t d;
    st->rc = mth_decode_records(st->ctx, buf, rec_off[n_rec], rec_off, n_rec, MTH_MEM_HOST, st->first ? 0 : 1, &d);
    st->first = false;
    return st->rc == MTH_OK ? 0 : 1;
}

// Whole load path on the device (mth_bgzf_decode): the file's bytes go to the GPU as they are, BGZF inflate, record
// boundaries and record decode all happen there.  Needs every BGZF block to hold whole records (what htslib-family
// writers produce); returns false -- with the context reset -- for a file where that does not hold.

bool load_bgzf_on_device(Input &in) {
    mth_host_bgzf_t bz;
    if (mth_host_bgzf_blocks(in.h, &bz) != 0) die(mth_host_last_error(in.h));
    uint64_t blk_beg = 0;
    if (g_shard.planned()) {      // this shard's run of blocks (+ halo blocks), or the blocks the .bai names for --region, instead of the whole file
        if (g_shard.region) {
            const int tid = mth_host_ref_tid(in.h, g_shard.region_name.c_str());
            if (tid < 0) die("--region: the BAM header has no reference named '" + g_shard.region_name + "'");
            if (mth_host_plan_region(in.h, g_shard.bai.empty() ? nullptr : g_shard.bai.c_str(), tid, g_shard.r_beg, g_shard.r_end, g_shard.halo, &g_shard.plan) != 0)
                die(mth_host_last_error(in.h));
        } else if (mth_host_plan_shard(in.h, g_shard.rank, g_shard.world, g_shard.halo, &g_shard.plan) != 0) die(mth_host_last_error(in.h));
        blk_beg = g_shard.plan.block_beg;
        bz.n_blocks = g_shard.plan.block_end;
        bz.header_bytes = g_shard.plan.first_byte;
    }
    Phase ph("  device inflate + walk + decode");
    // Chunks of whole blocks.  The first is small (<= 512 MiB of file bytes) so that the GPU starts early; every later chunk
    // (<= 2 GiB; METHEOR_CHUNK_GROWTH makes the sizes grow geometrically instead -- measured slower, profiles/r02_e2e.md) is copied to the device by a helper thread on a side stream (mth_bgzf_stage) while the chunk before it is
    // being inflated and decoded -- only the first copy is exposed.  METHEOR_DEVICE_CHUNK_MB sets both sizes,
    // METHEOR_NO_STAGE=1 copies every chunk in line.
    size_t chunk_first = (size_t)512 << 20, chunk = (size_t)2 << 30, growth = 64;
    if (const char *e = getenv("METHEOR_DEVICE_CHUNK_MB")) { const long k = atol(e); if (k >= 1 && k <= 65536) { chunk_first = chunk = (size_t)k << 20; growth = 1; } }
    if (const char *e = getenv("METHEOR_FIRST_CHUNK_MB")) { const long k = atol(e); if (k >= 1 && k <= 65536) chunk_first = (size_t)k << 20; }
    if (const char *e = getenv("METHEOR_CHUNK_GROWTH")) { const long k = atol(e); if (k >= 1 && k <= 64) growth = (size_t)k; }
    const bool stage = !getenv("METHEOR_NO_STAGE");
    struct Chunk { uint64_t b0, b1, base, nbytes, ubytes; };
    std::vector<Chunk> chunks;
    uint64_t total_u = 0;
    for (uint64_t b0 = blk_beg; b0 < bz.n_blocks;) {
        size_t lim = chunk_first;
        for (size_t k = 0; k < chunks.size() && lim < chunk; ++k) lim = std::min(chunk, lim * growth);
        uint64_t b1 = b0, ubytes = 0;
        while (b1 < bz.n_blocks && (b1 == b0 || bz.coff[b1] + bz.csize[b1] - bz.coff[b0] <= lim)) { ubytes += bz.isize[b1]; ++b1; }
        chunks.push_back(Chunk{b0, b1, bz.coff[b0], bz.coff[b1 - 1] + bz.csize[b1 - 1] + 8 - bz.coff[b0], ubytes});   // incl. the last block's CRC32 + ISIZE trailer
        total_u += ubytes;
        b0 = b1;
    }
    if (chunks.empty()) chunks.push_back(Chunk{blk_beg, blk_beg, 0, 0, 0});
    bool first = true;
    uint64_t hdr_left = bz.header_bytes;      // header bytes still ahead of the next chunk's inflated stream
    uint64_t done_u = 0;
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
        const Chunk &c = chunks[ci];
        const uint64_t first_byte = std::min<uint64_t>(hdr_left, c.ubytes);
        if (c.b1 > c.b0 && hdr_left > c.ubytes && ci + 1 < chunks.size()) { hdr_left -= c.ubytes; continue; }   // a chunk made of header only
        if (stage && ci + 1 < chunks.size() && chunks[ci + 1].nbytes)
            check(in.ctx, mth_bgzf_stage(in.ctx, bz.file + chunks[ci + 1].base, chunks[ci + 1].nbytes));
        std::vector<uint64_t> rel((size_t)(c.b1 - c.b0));
        for (uint64_t k = c.b0; k < c.b1; ++k) rel[(size_t)(k - c.b0)] = bz.coff[k] - c.base;
        mth_decoded_t d;
        const int rc = mth_bgzf_decode(in.ctx, bz.file + c.base, c.nbytes, rel.data(), bz.csize + c.b0, bz.isize + c.b0, c.b1 - c.b0, first_byte, first ? 0 : 1, &d);
        hdr_left -= first_byte;
        if (rc == MTH_ERR_UNALIGNED) { check(in.ctx, mth_reset(in.ctx)); return false; }
        if (rc == MTH_ERR_FORMAT) {
            const std::string m = mth_last_error(in.ctx);
            if (m.find("XM") != std::string::npos) die("Error reading XM tag in BAM record. Make sure the reads are aligned using Bismark!");   // readutil.rs:46
            die("Error reading BAM record. corrupt BGZF block or BAM record");
        }
        check(in.ctx, rc);
        done_u += c.ubytes;
        if (first && ci + 1 < chunks.size() && done_u) {
            // size the decoded arrays once from the first chunk's yield (reads and calls per inflated byte, +3 %)
            const double f = 1.03 * (double)total_u / (double)done_u;
            check(in.ctx, mth_decode_reserve(in.ctx, (uint64_t)((double)d.n_reads * f) + 4096, (uint64_t)((double)d.n_cpgs * f) + 4096));
        }
        first = false;
    }
    return true;
}

bool load_on_device(Input &in, const char *cpg_set, CtxFuture &cf) {
    const uint64_t *keys = nullptr;
    uint64_t n_keys = 0;
    if (cpg_set && mth_host_cpg_set_keys(in.h, cpg_set, &keys, &n_keys) != 0) { cf.wait(); die(mth_host_last_error(in.h)); }   // readutil.rs:356
    {
        mth_host_bgzf_t bz;                      // builds (and caches) the block table while the context is being created
        if (!getenv("METHEOR_HOST_INFLATE") && mth_host_bgzf_blocks(in.h, &bz) != 0) { cf.wait(); die(mth_host_last_error(in.h)); }
    }
    // (touching the mapped file's pages from 16 threads while the HIP runtime starts was measured: it slows the start-up
    // it competes with and the copies gain nothing, profiles/r02_e2e.md)
    in.ctx = cf.get();
    if (cpg_set) check(in.ctx, mth_decode_set_cpg_filter(in.ctx, keys, n_keys, 1));
    check(in.ctx, mth_decode_set_xm_min_mapq(in.ctx, (uint32_t)g_shard.xm_min_mapq));
    const bool on_device = !getenv("METHEOR_HOST_INFLATE") && load_bgzf_on_device(in);
    if (!on_device && g_shard.planned()) die("--gpus N / --region need the device load path (a coordinate-sorted BAM whose records do not straddle BGZF blocks)");
    if (!on_device) {
        StreamState st;
        st.ctx = in.ctx;
        Phase ph("  inflate + device record decode");
        const int rc = mth_host_decode_stream(in.h, window_to_device, &st);
        if (st.rc == MTH_ERR_FORMAT) {
            const std::string m = mth_last_error(in.ctx);
            if (m.find("XM") != std::string::npos) die("Error reading XM tag in BAM record. Make sure the reads are aligned using Bismark!");   // readutil.rs:46
            die("Error reading BAM record. corrupt BAM record");
        }
        if (st.rc != MTH_OK) check(in.ctx, st.rc);
        if (rc != 0) die(mth_host_last_error(in.h));
        if (st.first) {   // no record at all: an empty decode
            mth_decoded_t d;
            check(in.ctx, mth_decode_records(in.ctx, nullptr, 0, nullptr, 0, MTH_MEM_HOST, 0, &d));
        }
    }
    Phase ph2("  contig ranges");
    // one run of equal tid per contig in a coordinate-sorted file; anything else the batches cannot hold as it is
    const uint32_t cap = (uint32_t)std::max(1, mth_host_n_refs(in.h)) + 1;
    std::vector<int32_t> tids(cap);
    std::vector<uint64_t> rb(cap), re(cap);
    uint32_t n_runs = 0, flags = 0;
    check(in.ctx, mth_decoded_contigs(in.ctx, cap, tids.data(), rb.data(), re.data(), &n_runs, &flags));
    {
        // not coordinate-sorted / contigs not grouped: the order-free measures sort the decoded stream where it is, on the device
        // (mth_decoded_sort); the others go on to the host path, which says why it refuses
        bool regroup = n_runs > cap || (flags & 4u);
        for (uint32_t k = 0; k < std::min(n_runs, cap) && !regroup; ++k)
            for (uint32_t j = 0; j < k; ++j) if (tids[j] == tids[k]) regroup = true;
        if (regroup && g_shard.order_free && !(flags & 3u) && !g_shard.planned()) {
            Phase ps("  device sort by (tid, start)");
            check(in.ctx, mth_decoded_sort(in.ctx));
            check(in.ctx, mth_decoded_contigs(in.ctx, cap, tids.data(), rb.data(), re.data(), &n_runs, &flags));
        } else if (regroup && !g_shard.order_free && !g_shard.planned()) {
            // pdr / mhl / fdrp / qfdrp finalise sites as the stream moves past them: on such input their result is a function of the
            // record order itself (pdr.rs:139-178, mhl.rs:155-173, fdrp.rs:197-223), which the decoded stream still has -- replayed as
            // it is (mth_fileorder.hip), no batches
            in.file_order = true;
            in.device = true;
            return true;
        }
    }
    if (flags || n_runs > cap) return false;                       // unaligned / contig-less records, or contigs not grouped
    for (uint32_t k = 0; k < n_runs; ++k) {
        for (const Contig &c : in.contigs) if (c.tid == tids[k]) return false;   // the host path reports it
        int32_t reg_beg = 0, reg_end = -1;
        if (g_shard.planned()) {      // the part of this contig inside the shard's / region's interval [(tid_beg, pos_beg), (tid_end, pos_end))
            const mth_host_shard_t &pl = g_shard.plan;
            if (tids[k] < pl.tid_beg || tids[k] > pl.tid_end) continue;               // halo reads of a neighbour's contig
            if (tids[k] == pl.tid_beg) reg_beg = pl.pos_beg;
            if (tids[k] == pl.tid_end && pl.pos_end != INT32_MAX) { reg_end = pl.pos_end; if (reg_end <= reg_beg) continue; }
        }
        in.contigs.emplace_back();
        Contig &c = in.contigs.back();
        c.tid = tids[k]; c.r0 = rb[k]; c.r1 = re[k]; c.n_reads = (size_t)(re[k] - rb[k]); c.n_cpgs = 0;
        c.region_beg = reg_beg; c.region_end = reg_end;
    }
    // Several contigs per batch (include/metheor_hip.h, "contig groups"): every pass pays its fixed costs per batch, and a human BAM
    // has 24 to a few thousand contigs.  The library shifts the decoded positions into one virtual coordinate space per group and maps
    // the rows back at the fetch, so nothing below this line knows.  Not under a shard / region plan (those own parts of contigs);
    // METHEOR_GROUP=0 keeps one batch per contig (A/B, tests).
    if (!g_shard.planned() && in.contigs.size() > 1 && !(getenv("METHEOR_GROUP") && atoi(getenv("METHEOR_GROUP")) == 0)) {
        Phase pg("  contig groups");
        const uint32_t nc = (uint32_t)in.contigs.size();
        std::vector<int32_t> gt(nc), bt(nc);
        std::vector<uint64_t> gb(nc), ge(nc);
        std::vector<uint32_t> first((size_t)nc + 1);
        for (uint32_t k = 0; k < nc; ++k) { gt[k] = in.contigs[k].tid; gb[k] = in.contigs[k].r0; ge[k] = in.contigs[k].r1; }
        uint32_t ng = 0;
        check(in.ctx, mth_decoded_group(in.ctx, nc, gt.data(), gb.data(), ge.data(), &ng, first.data(), bt.data()));
        if (ng) {
            std::vector<Contig> grouped((size_t)ng);
            for (uint32_t g = 0; g < ng; ++g) {
                Contig &c = grouped[g];
                c.tid = bt[g]; c.r0 = gb[first[g]]; c.r1 = ge[first[g + 1] - 1]; c.n_reads = (size_t)(c.r1 - c.r0); c.n_cpgs = 0;
                c.region_beg = 0; c.region_end = -1;
            }
            in.contigs.swap(grouped);
        }
    }
    in.device = true;
    return true;
}

// LPMD, ME and PM keep no per-site state that a later read could flush (lpmd.rs:175-200, me.rs:106-125, pm.rs:101-121 iterate the
// records in file order into hash maps that are only read at the end): their results do not depend on the order of the
// records, and the reference accepts any BAM.  For those subcommands an input that is not coordinate-sorted / not grouped by
// contig is sorted here, stably by (tid, start), and then batched like any other.  PDR, MHL, FDRP and qFDRP finalise sites as
// the stream moves past them (pdr.rs:160-177, mhl.rs:162-173, fdrp.rs:206-218): their output on unsorted input is a function
// of the record order itself, which batches cut by contig do not carry -- those subcommands keep refusing it, loudly.
bool reads_in_order(const int32_t *tid, const int32_t *st, int64_t n) {
    // Records without an aligned base never enter a batch: not looked at.  Records without a contig (a sorted BAM keeps them at
    // its end) do not either, but the batching loop below cuts the contigs' runs at them: one BETWEEN two placed records -- tids
    // 0, -1, 0 re-open contig 0's run; 0, -1, 1 is harmless but not what a sort gives -- counts as out of order, so that the
    // order-free measures sort the file (the stable sort puts such records first, where they are skipped and counted) instead of
    // dying on "not grouped by contig".
    int64_t p = -1;
    bool loose_seen = false;
    for (int64_t i = 0; i < n; ++i) {
        if (tid[i] < 0) { loose_seen = p >= 0; continue; }
        if (loose_seen) return false;
        if (st[i] < 0) continue;
        if (p >= 0 && (tid[i] < tid[p] || (tid[i] == tid[p] && st[i] < st[p]))) return false;
        p = i;
    }
    return true;
}

Input load(const std::string &path, const char *cpg_set) {
    Phase ph_all("load: open+decode+batch");
    Input in;
    char err[1024];
    const bool try_device = !getenv("METHEOR_HOST_DECODE");
    if (!try_device && g_shard.planned()) die("--gpus N / --region need the device load path (METHEOR_HOST_DECODE is set)");
    CtxFuture cf;
    if (try_device) cf.start();
    if (mth_host_open(path.c_str(), &in.h, err, sizeof err) != 0) { cf.wait(); die(err); }    // bamutil.rs:7-9
    if (try_device) {
        if (load_on_device(in, cpg_set, cf)) return in;
        if (g_shard.planned()) die("--gpus N / --region need the device load path (coordinate-sorted input, contigs grouped, records inside BGZF blocks)");
        in.contigs.clear();
    }
    {
        Phase ph("  host decode (BGZF+BAM+XM)");
        mth_host_set_xm_min_mapq(in.h, g_shard.xm_min_mapq);
        if (mth_host_decode(in.h, cpg_set) != 0) die(mth_host_last_error(in.h));
    }
    Phase ph2("  contig batches");
    const int64_t n = mth_host_n_reads(in.h);
    const int32_t *tid = mth_host_read_tid(in.h), *st = mth_host_read_start(in.h), *en = mth_host_read_end(in.h);
    const uint8_t *mq = mth_host_read_mapq(in.h);
    const uint64_t *off = mth_host_cpg_off(in.h);
    const uint32_t *pos = mth_host_cpg_pos(in.h);
    const uint16_t *rel = mth_host_cpg_rel(in.h);
    if (!reads_in_order(tid, st, n)) {
        if (!g_shard.order_free)
            die("input BAM is not coordinate-sorted (or not grouped by contig), and it had to be decoded on the host (records that straddle "
                "BGZF blocks, or METHEOR_HOST_DECODE): pdr, mhl, fdrp and qfdrp finalise a CpG as the reads move past it (pdr.rs:160-177, "
                "mhl.rs:162-173, fdrp.rs:206-218), so their output on such input depends on the record order itself, which only the device "
                "decode path replays (mth_fileorder_run) -- sort the file (samtools sort).  lpmd, me and pm take any order.");
        Phase ps("  sort by (tid, start)");
        std::vector<int64_t> perm((size_t)n);
        for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return tid[x] != tid[y] ? tid[x] < tid[y] : st[x] < st[y]; });
        in.s_tid.resize((size_t)n); in.s_start.resize((size_t)n); in.s_end.resize((size_t)n); in.s_mapq.resize((size_t)n);
        in.s_off.assign((size_t)n + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            const int64_t r = perm[(size_t)i];
            in.s_tid[(size_t)i] = tid[r]; in.s_start[(size_t)i] = st[r]; in.s_end[(size_t)i] = en[r]; in.s_mapq[(size_t)i] = mq[r];
            in.s_off[(size_t)i + 1] = in.s_off[(size_t)i] + (off[r + 1] - off[r]);
        }
        in.s_pos.resize((size_t)in.s_off[(size_t)n]); in.s_rel.resize((size_t)in.s_off[(size_t)n]);
        for (int64_t i = 0; i < n; ++i) {
            const int64_t r = perm[(size_t)i];
            std::copy(pos + off[r], pos + off[r + 1], in.s_pos.begin() + (ptrdiff_t)in.s_off[(size_t)i]);
            std::copy(rel + off[r], rel + off[r + 1], in.s_rel.begin() + (ptrdiff_t)in.s_off[(size_t)i]);
        }
        tid = in.s_tid.data(); st = in.s_start.data(); en = in.s_end.data(); mq = in.s_mapq.data();
        off = in.s_off.data(); pos = in.s_pos.data(); rel = in.s_rel.data();
    }
    for (int64_t i = 0; i < n;) {
        int64_t e = i;
        bool loose = false;
        while (e < n && tid[e] == tid[i]) { loose |= st[e] < 0; ++e; }
        if (tid[i] < 0) { in.unbatched_mapq.insert(in.unbatched_mapq.end(), mq + i, mq + e); i = e; continue; }   // records without a contig never enter a batch
        for (const Contig &c : in.contigs)
            if (c.tid == tid[i]) die("input BAM is not grouped by contig (coordinate-sorted input is required on the MI355X path)");
        in.contigs.emplace_back();
        Contig &c = in.contigs.back();
        c.tid = tid[i];
        if (!loose) {
            c.start = st + i; c.end = en + i; c.mapq = mq + i; c.pos = pos + off[i]; c.rel = rel + off[i];
            c.n_reads = (size_t)(e - i); c.n_cpgs = (size_t)(off[e] - off[i]);
            c.off.resize(c.n_reads + 1);
            int32_t ms = 0;
            if (off[e] - off[i] >= (1ull << 32) || (uint64_t)(e - i) >= (1ull << 32) - 1) die("metheor (MI355X path): a contig with 2^32 or more reads / CpG calls does not fit one batch on the host-decode path");
            for (int64_t r = i; r < e; ++r) { c.off[(size_t)(r - i)] = (uint32_t)(off[r] - off[i]); ms = std::max(ms, en[r] - st[r] + 1); c.max_end = std::max(c.max_end, en[r]); }
            c.off[c.n_reads] = (uint32_t)(off[e] - off[i]);
            c.max_span = ms;
        } else {
            c.off.push_back(0);
            for (int64_t r = i; r < e; ++r) {
                if (st[r] < 0) { in.unbatched_mapq.push_back(mq[r]); continue; }   // no aligned base: no CpG, no position
                c.o_start.push_back(st[r]); c.o_end.push_back(en[r]); c.o_mapq.push_back(mq[r]);
                for (uint64_t k = off[r]; k < off[r + 1]; ++k) { c.o_pos.push_back(pos[k]); c.o_rel.push_back(rel[k]); }
                if (c.o_pos.size() >= (1ull << 32)) die("metheor (MI355X path): a contig with 2^32 or more CpG calls does not fit one batch on the host-decode path");
                c.off.push_back((uint32_t)c.o_pos.size());
                c.max_span = std::max(c.max_span, en[r] - st[r] + 1);
                c.max_end = std::max(c.max_end, en[r]);
            }
            c.start = c.o_start.data(); c.end = c.o_end.data(); c.mapq = c.o_mapq.data(); c.pos = c.o_pos.data(); c.rel = c.o_rel.data();
            c.n_reads = c.o_start.size(); c.n_cpgs = c.o_pos.size();
        }
        i = e;
    }
    return in;
}

mth_batch_t make_batch(const Input &in, const Contig &c) {
    mth_batch_t b;
    memset(&b, 0, sizeof b);
    if (in.device) {
        // a whole contig (or a shard's last piece of it) ends where its reads end, not at the header's LN:
        // the reference has no notion of LN in this path and emits sites beyond it
        check(in.ctx, mth_decoded_batch(in.ctx, c.r0, c.r1, c.tid, c.region_beg, c.region_end >= 0 ? c.region_end : -1, &b));
        if (g_shard.planned() && (int64_t)b.max_span + 202 > g_shard.halo)
            die("an alignment spans " + std::to_string(b.max_span) + " bp: set METHEOR_SHARD_HALO to at least " + std::to_string(b.max_span + 202));
        return b;
    }
    b.tid = c.tid;
    b.region_beg = 0;
    b.region_end = (int32_t)std::min<int64_t>((int64_t)c.max_end + 2, INT32_MAX);      // to the end of the data, not the header's LN
    b.max_span = c.max_span;
    b.n_reads = (uint32_t)c.n_reads;
    b.n_cpgs = (uint32_t)c.n_cpgs;
    b.mem = MTH_MEM_HOST;
    b.read_start = c.start; b.read_end = c.end; b.read_mapq = c.mapq;
    b.cpg_off = c.off.data(); b.cpg_pos = c.pos; b.cpg_rel16 = c.rel;
    return b;
}

void submit(mth_ctx_t *ctx, const Input &in, const mth_pdr_lpmd_params_t &p) {
    for (const Contig &c : in.contigs) {
        const mth_batch_t b = make_batch(in, c);
        check(ctx, mth_pdr_lpmd_accumulate(ctx, &b, &p));
    }
}

// TSV lines are assembled by hand into a large buffer (one write per ~4 MiB instead of one per line)
struct LineWriter {
    FILE *f;
    std::vector<char> buf;
    explicit LineWriter(FILE *file) : f(file) { buf.reserve(4u << 20); }
    void str(const char *s) { buf.insert(buf.end(), s, s + strlen(s)); }
    void ch(char c) { buf.push_back(c); }
    void i64(long long v) { char t[24]; const int n = snprintf(t, sizeof t, "%lld", v); buf.insert(buf.end(), t, t + n); }
    void u32(uint32_t v) {
        char t[12]; int n = 0;
        do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) buf.push_back(t[--n]);
    }
    void i32(int32_t v) { if (v < 0) { ch('-'); u32((uint32_t)(-(int64_t)v)); } else u32((uint32_t)v); }
    void f32(float v) { char t[64]; const int n = mth_host_format_f32(v, t); buf.insert(buf.end(), t, t + n); }
    void eol() { ch('\n'); if (f && buf.size() > (4u << 20) - 256) flush(); }   // f == nullptr: an in-memory part
    void flush() { if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) die("Error writing to output file."); buf.clear(); }
};

// n rows -> f: contiguous row ranges are formatted by several threads into their own buffers (the f32 shortest
// round-trip formatting is most of a 700 k-line TSV's cost) and written out in row order
template <class Row>
void write_rows(FILE *f, uint64_t n, Row &&row) {
    int nt = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("METHEOR_THREADS")) { const int k = atoi(e); if (k >= 1 && k <= 1024) nt = k; }
    nt = std::max(1, std::min(nt / g_shard.world, 32));      // the shards of a --gpus N run format side by side
    if (n < 50000) nt = 1;
    std::vector<LineWriter> parts;
    parts.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) parts.emplace_back(nt == 1 ? f : nullptr);
    auto work = [&](int t) {
        const uint64_t r0 = n * (uint64_t)t / (uint64_t)nt, r1 = n * (uint64_t)(t + 1) / (uint64_t)nt;
        for (uint64_t i = r0; i < r1; ++i) row(parts[(size_t)t], i);
    };
    if (nt == 1) { work(0); parts[0].flush(); return; }
    // (every thread writing its own part with pwrite at its offset was measured SLOWER than this ordered append on tmpfs:
    // 22 against 15 ms for the 25-MB PDR table, profiles/r02_e2e.md)
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    for (auto &w : parts) { w.f = f; w.flush(); }
}

FILE *open_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "wb");   // create + truncate (pdr.rs:95-101)
    if (!f) die("called `Result::unwrap()` on an `Err` value: cannot open output file " + path + ": " + strerror(errno));
    return f;
}

// one shard of several: its rows are collected in memory (main() concatenates the parts in shard order)
FILE *open_output(const std::string &path, bool pairs = false) {
    if (g_shard.world > 1) {
        FILE *f = pairs ? open_memstream(&g_part->pairs, &g_part->pairs_len) : open_memstream(&g_part->out, &g_part->out_len);
        if (!f) die("cannot allocate an output part");
        return f;
    }
    FILE *f = open_file(path);
    static char buf[1 << 20];
    setvbuf(f, buf, _IOFBF, sizeof buf);
    return f;
}

// LPMDResult's counters are genome-wide: the shards meet, the last one to arrive runs the RCCL all-reduce for all
// of them (mth_allreduce_lpmd; lpmd.rs:11-12, 51-55), and every context then holds the totals
void gang_allreduce_lpmd(mth_ctx_t *ctx) {
    std::unique_lock<std::mutex> lk(g_gang.mu);
    g_gang.ctxs.resize((size_t)g_shard.world);
    g_gang.ctxs[(size_t)g_shard.rank] = ctx;
    if (++g_gang.arrived == g_shard.world) {
        g_gang.rc = mth_allreduce_lpmd(g_gang.ctxs.data(), g_shard.world);
        g_gang.done = true;
        g_gang.cv.notify_all();
    } else {
        g_gang.cv.wait(lk, [] { return g_gang.done; });
    }
    if (g_gang.rc != MTH_OK) check(g_gang.ctxs[0], g_gang.rc);
}

// pdr / mhl / fdrp / qfdrp of an input that is not coordinate-sorted: the stream replayed in file order on the device
// (mth_fileorder.hip); which: 0 = the PDR line (pdr.rs:102-116), 1 = value v0 (mhl.rs:114-132, fdrp.rs:162-173), 2 = value v1 (qfdrp.rs:174-185)
int run_file_order(const Args &a, Input &in, mth_ctx_t *ctx, const mth_fileorder_params_t &fp, int which) {
    uint64_t n = 0;
    {
        Phase ph("file-order replay (kernels, sync)");
        check(ctx, mth_fileorder_run(ctx, &fp));
        check(ctx, mth_fileorder_fetch(ctx, &n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    }
    Phase ph3("fetch + TSV write");
    std::vector<int32_t> tid(n), pos(n);
    std::vector<float> v0(n), v1(n);
    std::vector<uint32_t> c0(n), c1(n);
    check(ctx, mth_fileorder_fetch(ctx, &n, tid.data(), pos.data(), v0.data(), v1.data(), c0.data(), c1.data()));
    FILE *f = open_output(a.s.at("output"));
    write_rows(f, n, [&](LineWriter &w, uint64_t i) {
        w.str(mth_host_ref_name(in.h, tid[i])); w.ch('\t'); w.i32(pos[i]); w.ch('\t'); w.i32(pos[i] + 2); w.ch('\t');
        if (which == 0) { w.f32(v0[i]); w.ch('\t'); w.u32(c0[i]); w.ch('\t'); w.u32(c1[i]); }
        else w.f32(which == 2 ? v1[i] : v0[i]);
        w.eol();
    });
    if (fclose(f) != 0) die("Error writing to output file.");
    return finish(ctx, in.h);
}

int run_pdr(const Args &a) {
    Input in = load(a.s.at("input"), a.has("cpg-set") ? a.s.at("cpg-set").c_str() : nullptr);
    mth_ctx_t *ctx = in.ctx ? in.ctx : make_ctx();
    mth_pdr_lpmd_params_t p;
    memset(&p, 0, sizeof p);
    p.pdr_min_depth = (uint32_t)a.n.at("min-depth");
    p.pdr_min_cpgs = (uint32_t)std::min<int64_t>(a.n.at("min-cpgs"), UINT32_MAX);
    p.pdr_min_qual = (uint8_t)a.n.at("min-qual");
    p.want_pdr = 1;
    if (in.file_order) {
        mth_fileorder_params_t fp;
        memset(&fp, 0, sizeof fp);
        fp.measure = MTH_FO_PDR; fp.min_depth = p.pdr_min_depth; fp.min_cpgs = p.pdr_min_cpgs; fp.min_qual = p.pdr_min_qual;
        return run_file_order(a, in, ctx, fp, 0);
    }
    uint64_t n = 0;
    {
        Phase ph("H2D + kernels (sync)");
        submit(ctx, in, p);
        check(ctx, mth_pdr_count(ctx, &n));
    }
    {
        Phase ph3("fetch + TSV write");
        // the five columns in one page-locked allocation (device-to-host at the link's rate; nothing to zero-fill)
        void *pin = nullptr;
        check(ctx, mth_result_buffer_alloc(ctx, (size_t)n * 20, &pin));
        int32_t *tid = (int32_t *)pin, *pos = tid + n;
        float *pdr = (float *)(pos + n);
        uint32_t *nc = (uint32_t *)(pdr + n), *nd = nc + n;
        { Phase pf("  fetch"); check(ctx, mth_pdr_fetch(ctx, tid, pos, pdr, nc, nd)); }
        Phase pw("  format + write");
        FILE *f = open_output(a.s.at("output"));
        write_rows(f, n, [&](LineWriter &w, uint64_t i) {   // pdr.rs:102-116
            w.str(mth_host_ref_name(in.h, tid[i])); w.ch('\t'); w.i32(pos[i]); w.ch('\t'); w.i32(pos[i] + 2); w.ch('\t');
            w.f32(pdr[i]); w.ch('\t'); w.u32(nc[i]); w.ch('\t'); w.u32(nd[i]); w.eol();
        });
        if (fclose(f) != 0) die("Error writing to output file.");
        if (getenv("METHEOR_TEARDOWN")) check(ctx, mth_result_buffer_free(ctx, pin));   // (unpinning is a few ms: left to the process's end otherwise)
    }
    return finish(ctx, in.h);
}

int run_lpmd(const Args &a) {
    const std::string input = a.s.at("input");
    const int32_t mind = (int32_t)a.n.at("min-distance"), maxd = (int32_t)a.n.at("max-distance");
    // lpmd.rs:161-164
    fprintf(stderr, "Computing subset-LPMD with parameters input=%s, min_distance=%d, max_distance=%d\n", input.c_str(), mind, maxd);
    g_shard.xm_min_mapq = (int)a.n.at("min-qual");      // lpmd.rs:176-181: the mapq filter comes before BismarkRead::new (the XM panic)
    g_shard.order_free = true;
    Input in = load(input, a.has("cpg-set") ? a.s.at("cpg-set").c_str() : nullptr);
    mth_ctx_t *ctx = in.ctx ? in.ctx : make_ctx();
    mth_pdr_lpmd_params_t p;
    memset(&p, 0, sizeof p);
    p.lpmd_min_qual = (uint8_t)a.n.at("min-qual");
    p.lpmd_min_distance = mind; p.lpmd_max_distance = maxd;
    p.want_lpmd = 1;
    submit(ctx, in, p);
    if (!in.unbatched_mapq.empty()) {                            // lpmd.rs:176-179: every record counts, also those without a position
        uint64_t nv = 0;
        for (uint8_t q : in.unbatched_mapq) nv += q >= p.lpmd_min_qual ? 1u : 0u;
        check(ctx, mth_lpmd_add_unbatched(ctx, in.unbatched_mapq.size(), nv));
    }
    if (g_shard.world > 1) gang_allreduce_lpmd(ctx);             // every shard now holds the genome-wide counters
    int64_t g[4] = {0, 0, 0, 0};
    float lp = 0.f;
    check(ctx, mth_lpmd_global(ctx, g, &lp));
    if (getenv("METHEOR_DEBUG_COUNTS")) fprintf(stderr, "[metheor counts] n_concordant=%lld n_discordant=%lld n_read=%lld n_valid_read=%lld\n", (long long)g[0], (long long)g[1], (long long)g[2], (long long)g[3]);
    // records that never enter a batch only move n_read / n_valid_read (not part of the TSV)
    FILE *f = open_output(a.s.at("output"));
    char fb[64];
    mth_host_format_f32(lp, fb);
    if (g_shard.rank == 0) fprintf(f, "name\tlpmd\n%s\t%s\n", input.c_str(), fb);   // lpmd.rs:145-147
    if (fclose(f) != 0) die("Error writing to output file.");
    if (a.has("pairs")) {                                       // lpmd.rs:149-151, 89-122
        mth_lpmd_pairs_params_t pp;
        pp.min_distance = mind; pp.max_distance = maxd; pp.min_qual = p.lpmd_min_qual;
        for (const Contig &c : in.contigs) {
            const mth_batch_t b = make_batch(in, c);
            check(ctx, mth_lpmd_pairs_accumulate(ctx, &b, &pp));
        }
        uint64_t n = 0;
        check(ctx, mth_lpmd_pairs_fetch(ctx, &n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        std::vector<int32_t> tid(n), p1(n), p2(n);
        std::vector<float> v(n);
        std::vector<uint32_t> nc(n), nd(n);
        check(ctx, mth_lpmd_pairs_fetch(ctx, &n, tid.data(), p1.data(), p2.data(), v.data(), nc.data(), nd.data()));
        FILE *g = open_output(a.s.at("pairs"), true);
        if (g_shard.rank == 0) fprintf(g, "chrom\tcpg1\tcpg2\tlpmd\tn_concordant\tn_discordant\n");
        fflush(g);
        write_rows(g, n, [&](LineWriter &w, uint64_t i) {
            w.str(mth_host_ref_name(in.h, tid[i])); w.ch('\t'); w.i32(p1[i]); w.ch('\t'); w.i32(p2[i]); w.ch('\t');
            w.f32(v[i]); w.ch('\t'); w.u32(nc[i]); w.ch('\t'); w.u32(nd[i]); w.eol();
        });
        if (fclose(g) != 0) die("Error writing to output file.");
    }
    return finish(ctx, in.h);
}

// me.rs:68-88 / pm.rs:63-83: one line per quartet with depth >= min_depth,
// chrom, pos1..pos4, value (me.rs:57-65).  The reference iterates a HashMap (random order).
int run_quartet(const Args &a, bool want_me) {
    g_shard.order_free = true;
    Input in = load(a.s.at("input"), a.has("cpg-set") ? a.s.at("cpg-set").c_str() : nullptr);
    mth_ctx_t *ctx = in.ctx ? in.ctx : make_ctx();
    mth_quartet_params_t p;
    p.min_qual = (uint8_t)a.n.at("min-qual");
    {
        Phase ph("H2D + kernels (sync)");
        for (const Contig &c : in.contigs) {
            const mth_batch_t b = make_batch(in, c);
            check(ctx, mth_quartet_accumulate(ctx, &b, &p));
        }
    }
    {
        Phase ph("fetch + TSV write");
        const uint32_t min_depth = (uint32_t)a.n.at("min-depth");
        uint64_t n = 0;
        check(ctx, mth_quartet_fetch(ctx, min_depth, &n, nullptr, nullptr, nullptr, nullptr, nullptr));
        std::vector<int32_t> tid(n), pos(n * 4);
        std::vector<float> val(n);
        check(ctx, mth_quartet_fetch(ctx, min_depth, &n, tid.data(), pos.data(), nullptr, want_me ? val.data() : nullptr,
                                     want_me ? nullptr : val.data()));
        FILE *f = open_output(a.s.at("output"));
        write_rows(f, n, [&](LineWriter &w, uint64_t i) {
            w.str(mth_host_ref_name(in.h, tid[i]));
            for (int k = 0; k < 4; ++k) { w.ch('\t'); w.i32(pos[4 * i + k]); }
            w.ch('\t'); w.f32(val[i]); w.eol();
        });
        if (fclose(f) != 0) die("Error writing to output file.");
    }
    return finish(ctx, in.h);
}

// mhl.rs:101-133: chrom, pos, pos+2, mhl -- sorted by (tid,pos)
int run_mhl(const Args &a) {
    Input in = load(a.s.at("input"), a.has("cpg-set") ? a.s.at("cpg-set").c_str() : nullptr);
    mth_ctx_t *ctx = in.ctx ? in.ctx : make_ctx();
    mth_mhl_params_t p;
    p.min_depth = (uint32_t)a.n.at("min-depth");
    p.min_cpgs = (uint32_t)std::min<int64_t>(a.n.at("min-cpgs"), UINT32_MAX);
    p.min_qual = (uint8_t)a.n.at("min-qual");
    if (in.file_order) {
        mth_fileorder_params_t fp;
        memset(&fp, 0, sizeof fp);
        fp.measure = MTH_FO_MHL; fp.min_depth = p.min_depth; fp.min_cpgs = p.min_cpgs; fp.min_qual = p.min_qual;
        return run_file_order(a, in, ctx, fp, 1);
    }
    for (const Contig &c : in.contigs) {
        const mth_batch_t b = make_batch(in, c);
        check(ctx, mth_mhl_accumulate(ctx, &b, &p));
    }
    uint64_t n = 0;
    check(ctx, mth_mhl_fetch(ctx, &n, nullptr, nullptr, nullptr, nullptr));
    std::vector<int32_t> tid(n), pos(n);
    std::vector<float> val(n);
    check(ctx, mth_mhl_fetch(ctx, &n, tid.data(), pos.data(), val.data(), nullptr));
    FILE *f = open_output(a.s.at("output"));
    write_rows(f, n, [&](LineWriter &w, uint64_t i) {
        w.str(mth_host_ref_name(in.h, tid[i])); w.ch('\t'); w.i32(pos[i]); w.ch('\t'); w.i32(pos[i] + 2); w.ch('\t'); w.f32(val[i]); w.eol();
    });
    if (fclose(f) != 0) die("Error writing to output file.");
    return finish(ctx, in.h);
}

// fdrp.rs:148-174 / qfdrp.rs:160-186: chrom, pos, pos+2, value -- sorted by (tid,pos)
int run_fdrp(const Args &a, bool quantitative) {
    Input in = load(a.s.at("input"), a.has("cpg-set") ? a.s.at("cpg-set").c_str() : nullptr);
    mth_ctx_t *ctx = in.ctx ? in.ctx : make_ctx();
    mth_fdrp_params_t p;
    memset(&p, 0, sizeof p);
    p.min_qual = (uint8_t)a.n.at("min-qual");
    p.min_depth = (uint64_t)a.n.at("min-depth");
    p.max_depth = (uint32_t)std::min<int64_t>(a.n.at("max-depth"), UINT32_MAX);
    p.min_overlap = (int32_t)a.n.at("min-overlap");
    const char *seed = getenv("METHEOR_SEED");       // reservoir draws (the reference's are OS-seeded)
    p.seed = seed ? strtoull(seed, nullptr, 10) : 0;
    if (in.file_order) {
        mth_fileorder_params_t fp;
        memset(&fp, 0, sizeof fp);
        fp.measure = MTH_FO_FDRP; fp.min_depth = (uint32_t)std::min<uint64_t>(p.min_depth, UINT32_MAX); fp.max_depth = p.max_depth;
        fp.min_overlap = p.min_overlap; fp.min_qual = p.min_qual; fp.seed = p.seed;
        return run_file_order(a, in, ctx, fp, quantitative ? 2 : 1);
    }
    for (const Contig &c : in.contigs) {
        const mth_batch_t b = make_batch(in, c);
        check(ctx, mth_fdrp_accumulate(ctx, &b, &p));
    }
    uint64_t n = 0;
    check(ctx, mth_fdrp_fetch(ctx, &n, nullptr, nullptr, nullptr, nullptr, nullptr));
    std::vector<int32_t> tid(n), pos(n);
    std::vector<float> val(n);
    check(ctx, mth_fdrp_fetch(ctx, &n, tid.data(), pos.data(), quantitative ? nullptr : val.data(),
                              quantitative ? val.data() : nullptr, nullptr));
    FILE *f = open_output(a.s.at("output"));
    write_rows(f, n, [&](LineWriter &w, uint64_t i) {
        w.str(mth_host_ref_name(in.h, tid[i])); w.ch('\t'); w.i32(pos[i]); w.ch('\t'); w.i32(pos[i] + 2); w.ch('\t'); w.f32(val[i]); w.eol();
    });
    if (fclose(f) != 0) die("Error writing to output file.");
    return finish(ctx, in.h);
}

}  // namespace

// ---- metheor tag (src/tag.rs:386-443 run) ---------------------------------------------------------------------
// Same order of events as the reference: open the input (BAM or SAM text) and look at its first record's paired flag
// (bamutil.rs:27-37), refuse an output path whose parent is not a directory (tag.rs:396-403; a bare file name has the
// parent "" and is refused there too), create the output and write the header as it is (tag.rs:405-410), open the
// FASTA and fetch every @SQ contig (tag.rs:412-431, printing the two progress lines), then stream the records: the
// XM string of each is computed on the device (mth_tag_records), appended as the last optional field and the record is
// written as a SAM line (tag.rs:433-441).
struct TagState {
    mth_ctx_t *ctx = nullptr;
    mth_host_t *h = nullptr;
    FILE *out = nullptr;
    int paired = -1;
    std::string line, fail;
};
int tag_window(void *user, const uint8_t *buf, const uint64_t *rec_off, uint64_t n_rec) {
    TagState *st = static_cast<TagState *>(user);
    if (n_rec == 0) return 0;
    if (st->paired < 0) {                          // is_paired_end: the file's first record
        const uint64_t o = rec_off[0];
        st->paired = (rec_off[1] - o >= 36) ? ((buf[o + 4 + 14] | (buf[o + 4 + 15] << 8)) & 1) : 0;
    }
    mth_tag_out_t t{};
    const int rc = mth_tag_records(st->ctx, buf, rec_off[n_rec], rec_off, n_rec, MTH_MEM_HOST, st->paired, &t);
    if (rc != MTH_OK) {
        st->fail = std::string("metheor (MI355X path): ") + mth_strerror(rc) + (mth_last_error(st->ctx)[0] ? std::string(" -- ") + mth_last_error(st->ctx) : std::string());
        return 1;
    }
    std::vector<char> b;
    for (uint64_t i = 0; i < n_rec; ++i) {
        const uint8_t *rec = buf + rec_off[i] + 4;
        const uint32_t len = (uint32_t)(rec_off[i + 1] - rec_off[i] - 4);
        const int64_t cap = 4 * (int64_t)len + 256 + t.xm_len[i];
        if ((int64_t)b.size() < cap) b.resize((size_t)cap);
        int64_t n = mth_host_sam_format(st->h, rec, len, t.xm + t.xm_off[i], t.xm_len[i], b.data(), (int64_t)b.size());
        if (n > (int64_t)b.size()) {      // B:c / B:s arrays print up to 7 characters per 1-2 byte element: the formatter says how much it needs
            b.resize((size_t)n);
            n = mth_host_sam_format(st->h, rec, len, t.xm + t.xm_off[i], t.xm_len[i], b.data(), (int64_t)b.size());
        }
        if (n < 0 || n > (int64_t)b.size()) { st->fail = "metheor: cannot format a record as SAM (malformed BAM record)"; return 1; }
        if (fwrite(b.data(), 1, (size_t)n, st->out) != (size_t)n) { st->fail = "Error writing to output file."; return 1; }
    }
    return 0;
}

int run_tag(const Args &a) {
    const std::string input = a.s.at("input"), output = a.s.at("output"), genome = a.s.at("genome");
    CtxFuture cf;
    cf.start();
    mth_host_t *h = nullptr;
    char err[1024];
    if (mth_host_open(input.c_str(), &h, err, sizeof err) != 0) { cf.wait(); die(err); }   // bamutil.rs:4-11
    // tag.rs:396-403: PathBuf::from(output).parent() must be a directory
    {
        const size_t slash = output.rfind('/');
        const std::string dir = slash == std::string::npos ? std::string() : (slash == 0 ? std::string("/") : output.substr(0, slash));
        struct stat sb;
        if (dir.empty() || stat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) { cf.wait(); die("No such directory for output alignment file: " + dir); }
    }
    FILE *out = fopen(output.c_str(), "wb");
    if (!out) { cf.wait(); die("Error opening alignment file to write: " + output + ": " + strerror(errno)); }
    static char obuf[1 << 20];
    setvbuf(out, obuf, _IOFBF, sizeof obuf);
    {   // the header text as it is, newline-terminated (Header::from_template strips and re-adds the final newline)
        uint64_t n = 0;
        const char *t = mth_host_header_text(h, &n);
        std::string text(t, strnlen(t, (size_t)n));
        while (!text.empty() && text.back() == '\n') text.pop_back();
        if (!text.empty()) { text += '\n'; fwrite(text.data(), 1, text.size(), out); }
    }
    mth_fasta_t *fa = nullptr;
    if (mth_host_fasta_open(genome.c_str(), &fa, err, sizeof err) != 0) { cf.wait(); fflush(out); die(std::string("Error opening reference genome file: ") + err); }
    printf("Parsing reference genome...\n");
    mth_ctx_t *ctx = cf.get();
    {
        Phase ph("genome -> device");
        const int n_refs = mth_host_n_refs(h);
        std::vector<std::vector<uint8_t>> seqs((size_t)n_refs);
        std::vector<const uint8_t *> ptr((size_t)n_refs);
        std::vector<int64_t> ln((size_t)n_refs), got((size_t)n_refs);
        for (int t = 0; t < n_refs; ++t) {
            const uint8_t *p = nullptr;
            int64_t n = 0;
            ln[(size_t)t] = mth_host_ref_len(h, t);
            if (mth_host_fasta_fetch(fa, mth_host_ref_name(h, t), ln[(size_t)t], &p, &n) != 0) { fflush(out); die("Error fetching reference genome sequence.: " + std::string(mth_host_fasta_last_error(fa))); }
            seqs[(size_t)t].assign(p, p + n);
            ptr[(size_t)t] = seqs[(size_t)t].data(); got[(size_t)t] = n;
        }
        check(ctx, mth_tag_set_genome(ctx, n_refs, ln.data(), ptr.data(), got.data()));
    }
    printf("Done!\n");
    fflush(stdout);
    TagState st;
    st.ctx = ctx; st.h = h; st.out = out;
    {
        Phase ph("records -> XM -> SAM");
        const int rc = mth_host_decode_stream(h, tag_window, &st);
        if (rc != 0) { fflush(out); die(!st.fail.empty() ? st.fail : std::string("Error reading BAM record. ") + mth_host_last_error(h)); }
    }
    if (fclose(out) != 0) die("Error writing to output file.");
    mth_host_fasta_close(fa);
    return finish(ctx, h);
}

int main(int argc, char **argv) {
    if (argc < 2) { print_main_help(stderr); return 2; }   // arg_required_else_help (lib.rs:18)
    const std::string sub = argv[1];
    if (sub == "-h" || sub == "--help" || sub == "help") { print_main_help(stdout); return 0; }
    if (sub == "-V" || sub == "--version") { printf("metheor 0.1.9\n"); return 0; }
    const Cmd *cmd = nullptr;
    for (const Cmd &c : commands()) if (sub == c.name) cmd = &c;
    if (!cmd) {
        if (!sub.empty() && sub[0] == '-') usage_error(nullptr, "unexpected argument '" + sub + "' found");
        usage_error(nullptr, "unrecognized subcommand '" + sub + "'");
    }
    const Args a = parse_args(*cmd, argc, argv, 2);
    // HIP runtime setting for a process that keeps one or two streams busy per device (read by the runtime at its first call; a value
    // from the caller's environment wins): one hardware queue per device shortens the load phase by ~25 ms on the 10 M-read file
    // (profiles/r02_e2e.md, "Runtime settings"; HIP_FORCE_DEV_KERNARG, HSA_ENABLE_INTERRUPT=0, HSA_ENABLE_SDMA=0 and others: no gain)
    setenv("GPU_MAX_HW_QUEUES", "1", 0);
    auto run = [&]() -> int {
        if (sub == "pdr") return run_pdr(a);
        if (sub == "lpmd") return run_lpmd(a);
        if (sub == "mhl") return run_mhl(a);
        if (sub == "fdrp") return run_fdrp(a, false);
        if (sub == "qfdrp") return run_fdrp(a, true);
        if (sub == "me") return run_quartet(a, true);
        if (sub == "pm") return run_quartet(a, false);
        return -1;
    };
    if (sub == "tag") {
        if (const char *dev = getenv("METHEOR_DEVICE")) g_shard.device = atoi(dev);
        return run_tag(a);
    }
    int64_t halo = 65536;
    if (const char *e = getenv("METHEOR_SHARD_HALO")) { const long long k = atoll(e); if (k >= 0) halo = k; }
    if (a.has("region")) {       // chr | chr:beg-end (1-based, inclusive, commas allowed; as samtools view takes it)
        const std::string reg = a.s.at("region");
        g_shard.region = true; g_shard.region_name = reg;
        const size_t c = reg.rfind(':');
        if (c != std::string::npos && c + 1 < reg.size()) {
            std::string t;
            for (char ch : reg.substr(c + 1)) if (ch != ',') t.push_back(ch);
            long long b = 0, e = 0;
            char tail = 0;
            const int k = sscanf(t.c_str(), "%lld-%lld%c", &b, &e, &tail);
            if (k == 2 && b >= 1 && e >= b && e <= INT32_MAX) { g_shard.region_name = reg.substr(0, c); g_shard.r_beg = (int32_t)(b - 1); g_shard.r_end = (int32_t)e; }
            else if (k == 2 || reg.find('-', c) != std::string::npos) usage_error(cmd, "invalid value '" + reg + "' for '--region <REGION>': want chr or chr:beg-end with 1 <= beg <= end");
        }
        if (a.has("bai")) g_shard.bai = a.s.at("bai");
        if (a.n.at("gpus") > 1) usage_error(cmd, "the argument '--region <REGION>' cannot be used with '--gpus <GPUS>' above 1");
    } else if (a.has("bai")) usage_error(cmd, "the argument '--bai <BAI>' needs '--region <REGION>'");
    const int world = (int)std::min<int64_t>(a.n.at("gpus"), 4096);
    if (world < 1) usage_error(cmd, "invalid value '0' for '--gpus <GPUS>': at least one GPU");
    if (world == 1) {
        if (const char *dev = getenv("METHEOR_DEVICE")) g_shard.device = atoi(dev);
        g_shard.halo = halo;
        return run();
    }
    // --gpus N: shard r runs on its own thread with its own context on device r mod the devices in use
    int ndev = 0;
    mth_device_count(&ndev);
    if (const char *e = getenv("METHEOR_DEVICES")) { const int k = atoi(e); if (k >= 1 && k < ndev) ndev = k; }
    if (ndev < 1) die(std::string("metheor (MI355X path): ") + mth_strerror(MTH_ERR_NO_DEVICE));
    {   // the reference opens the BAM before anything else: report an unreadable input once, not once per shard
        mth_host_t *h = nullptr;
        char err[1024];
        if (mth_host_open(a.s.at("input").c_str(), &h, err, sizeof err) != 0) die(err);
        mth_host_close(h);
    }
    std::vector<Part> parts((size_t)world);
    std::vector<std::thread> th;
    g_threads.store(true);
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            g_shard.rank = r; g_shard.world = world; g_shard.device = r % ndev; g_shard.halo = halo;
            g_part = &parts[(size_t)r];
            run();
        });
    for (auto &t : th) t.join();
    auto write_parts = [&](const std::string &path, bool pairs) {
        FILE *f = open_file(path);
        for (const Part &p : parts) {
            const char *b = pairs ? p.pairs : p.out;
            const size_t n = pairs ? p.pairs_len : p.out_len;
            if (n && fwrite(b, 1, n, f) != n) die("Error writing to output file.");
        }
        if (fclose(f) != 0) die("Error writing to output file.");
    };
    write_parts(a.s.at("output"), false);
    if (sub == "lpmd" && a.has("pairs")) write_parts(a.s.at("pairs"), true);
    fflush(nullptr);
    _exit(0);
}
