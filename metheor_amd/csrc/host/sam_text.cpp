// sam_text.cpp -- SAM text in, SAM text out, FASTA in (the image has no htslib).
//
// What the reference gets from rust-htslib / htslib and this engine needs for a drop-in CLI:
//   * bam::Reader::from_path opens SAM text as readily as BAM (bamutil.rs:4-11; the reference's own `tag` golden feeds it
//     tests/test.chr19.noXM.sam, tests/tag-cli.rs:60-80): sam_text_to_bam() turns a SAM file into the bytes of an equivalent
//     BGZF-compressed BAM (SAM spec 1.4 -> 4.2), so that every loader behind mth_host_open -- host inflate, device inflate,
//     shard planner -- sees a BAM.
//   * bam::Writer::from_path(.., Format::Sam) + push_aux(b"XM", Aux::String) (tag.rs:405-441): sam_format_record() prints one
//     BAM record as a SAM line (htslib sam_format1's field rules), with the new XM:Z field appended last.
//   * faidx::Reader::from_path + fetch_seq(name, 0, LN) (tag.rs:412-431): Fasta -- the .fai next to the file if there is one,
//     otherwise one scan of the file (htslib would write the .fai; nothing is written here).
#include <zlib.h>

#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/metheor_host.h"
#include "bam_reader.h"
#include "sam_text.h"

namespace mthh {

namespace {

void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

constexpr size_t BGZF_BLOCK = 0xff00;
void bgzf_block(std::vector<uint8_t> &out, const uint8_t *p, size_t n) {
    uint8_t comp[70000];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<uint8_t *>(p); zs.avail_in = (uInt)n;
    zs.next_out = comp; zs.avail_out = sizeof comp;
    deflate(&zs, Z_FINISH);
    const size_t clen = sizeof comp - zs.avail_out;
    deflateEnd(&zs);
    const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    out.insert(out.end(), hdr, hdr + 12);
    out.push_back('B'); out.push_back('C'); put16(out, 2); put16(out, (uint32_t)(clen + 25));
    out.insert(out.end(), comp, comp + clen);
    put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n));
    put32(out, (uint32_t)n);
}

// SAM spec 5.3
int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

std::vector<std::string> split_tab(const std::string &s) {
    std::vector<std::string> f;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find('\t', a);
        f.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return f;
}

bool parse_i64(const std::string &s, int64_t &v) {
    if (s.empty()) return false;
    char *e = nullptr;
    errno = 0;
    v = strtoll(s.c_str(), &e, 10);
    return errno == 0 && e == s.c_str() + s.size();
}

const char *SEQ_CODES = "=ACMGRSVTWYHKDBN";
const char *CIGAR_CODES = "MIDNSHP=X";

// one optional field TAG:TYPE:VALUE -> BAM aux bytes (htslib sam_parse1: integers take the smallest type that holds them)
bool parse_aux(const std::string &a, std::vector<uint8_t> &out) {
    if (a.size() < 5 || a[2] != ':' || a[4] != ':') return false;
    out.push_back((uint8_t)a[0]); out.push_back((uint8_t)a[1]);
    const char ty = a[3];
    const std::string v = a.substr(5);
    auto put_int = [&](int64_t x) {
        if (x < 0) {
            if (x >= -128) { out.push_back('c'); out.push_back((uint8_t)(int8_t)x); }
            else if (x >= -32768) { out.push_back('s'); put16(out, (uint32_t)(uint16_t)(int16_t)x); }
            else { out.push_back('i'); put32(out, (uint32_t)(int32_t)x); }
        } else {
            if (x < 256) { out.push_back('C'); out.push_back((uint8_t)x); }
            else if (x < 65536) { out.push_back('S'); put16(out, (uint32_t)x); }
            else { out.push_back('I'); put32(out, (uint32_t)x); }
        }
    };
    switch (ty) {
        case 'A': if (v.size() != 1) return false; out.push_back('A'); out.push_back((uint8_t)v[0]); return true;
        case 'i': { int64_t x; if (!parse_i64(v, x) || x < INT32_MIN || x > (int64_t)UINT32_MAX) return false; put_int(x); return true; }
        case 'f': { char *e = nullptr; const float x = strtof(v.c_str(), &e); if (e != v.c_str() + v.size() || v.empty()) return false;
                    out.push_back('f'); uint32_t u; memcpy(&u, &x, 4); put32(out, u); return true; }
        case 'Z': case 'H': out.push_back((uint8_t)ty); out.insert(out.end(), v.begin(), v.end()); out.push_back(0); return true;
        case 'B': {
            if (v.empty()) return false;
            const char sub = v[0];
            if (!strchr("cCsSiIf", sub)) return false;
            std::vector<std::string> items;
            size_t p = 1;
            while (p < v.size()) {
                if (v[p] != ',') return false;
                const size_t q = v.find(',', p + 1);
                items.push_back(v.substr(p + 1, q == std::string::npos ? std::string::npos : q - p - 1));
                if (q == std::string::npos) break;
                p = q;
            }
            out.push_back('B'); out.push_back((uint8_t)sub); put32(out, (uint32_t)items.size());
            for (const auto &it : items) {
                if (sub == 'f') { char *e = nullptr; const float x = strtof(it.c_str(), &e); if (it.empty() || e != it.c_str() + it.size()) return false; uint32_t u; memcpy(&u, &x, 4); put32(out, u); continue; }
                int64_t x;
                if (!parse_i64(it, x)) return false;
                if (sub == 'c' || sub == 'C') out.push_back((uint8_t)x);
                else if (sub == 's' || sub == 'S') put16(out, (uint32_t)(uint16_t)x);
                else put32(out, (uint32_t)x);
            }
            return true;
        }
        default: return false;
    }
}

}  // namespace

bool looks_like_sam(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    if (n < 4) return false;
    buf[n] = 0;
    if ((uint8_t)buf[0] == 31 && (uint8_t)buf[1] == 139) return false;                 // gzip / BGZF
    if (buf[0] == '@') return buf[1] >= 'A' && buf[1] <= 'Z' && buf[2] >= 'A' && buf[2] <= 'Z' && (buf[3] == '\t' || buf[3] == '\n');
    // header-less SAM: a first line with >= 11 tab-separated fields whose 2nd and 4th are numbers
    const char *nl = strchr(buf, '\n');
    const std::string line(buf, nl ? (size_t)(nl - buf) : n);
    const auto f11 = split_tab(line);
    int64_t x;
    return f11.size() >= 11 && parse_i64(f11[1], x) && parse_i64(f11[3], x);
}

bool sam_text_to_bam(const std::string &path, std::vector<uint8_t> &bam, std::string &err) {
    std::ifstream in(path);
    if (!in) { err = "unable to open: " + path; return false; }
    std::string text, line;
    std::vector<BamRef> refs;
    std::unordered_map<std::string, int> name2tid;
    std::vector<uint8_t> raw;           // uncompressed BAM stream: header first, then the block being filled
    bool header_done = false;
    auto flush_header = [&]() {
        std::vector<uint8_t> head;
        head.insert(head.end(), {'B', 'A', 'M', 1});
        put32(head, (uint32_t)text.size()); head.insert(head.end(), text.begin(), text.end());
        put32(head, (uint32_t)refs.size());
        for (const auto &r : refs) { put32(head, (uint32_t)r.name.size() + 1); head.insert(head.end(), r.name.begin(), r.name.end()); head.push_back(0); put32(head, (uint32_t)r.length); }
        for (size_t o = 0; o < head.size(); o += BGZF_BLOCK) bgzf_block(bam, head.data() + o, std::min(BGZF_BLOCK, head.size() - o));
        header_done = true;
    };
    size_t lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '@' && !header_done) {
            text += line; text += '\n';
            if (line.compare(0, 3, "@SQ") == 0) {
                std::string sn; int64_t ln = -1;
                for (const auto &f : split_tab(line)) {
                    if (f.compare(0, 3, "SN:") == 0) sn = f.substr(3);
                    else if (f.compare(0, 3, "LN:") == 0) parse_i64(f.substr(3), ln);
                }
                if (sn.empty() || ln < 0) { err = "malformed @SQ line " + std::to_string(lineno) + ": " + path; return false; }
                name2tid.emplace(sn, (int)refs.size());
                refs.push_back(BamRef{sn, ln});
            }
            continue;
        }
        if (!header_done) flush_header();
        const auto f = split_tab(line);
        auto bad = [&](const char *what) { err = std::string("malformed SAM record (") + what + ") at line " + std::to_string(lineno) + ": " + path; return false; };
        if (f.size() < 11) return bad("fewer than 11 fields");
        int64_t flag, pos, mapq, pnext, tlen;
        if (!parse_i64(f[1], flag) || flag < 0 || flag > 65535) return bad("FLAG");
        if (!parse_i64(f[3], pos) || pos < 0 || pos > INT32_MAX) return bad("POS");
        if (!parse_i64(f[4], mapq) || mapq < 0 || mapq > 255) return bad("MAPQ");
        if (!parse_i64(f[7], pnext) || pnext < 0 || pnext > INT32_MAX) return bad("PNEXT");
        if (!parse_i64(f[8], tlen) || tlen < INT32_MIN || tlen > INT32_MAX) return bad("TLEN");
        int32_t tid = -1, mtid = -1;
        if (f[2] != "*") { const auto it = name2tid.find(f[2]); if (it == name2tid.end()) return bad("RNAME not in the header"); tid = it->second; }
        if (f[6] == "=") mtid = tid;
        else if (f[6] != "*") { const auto it = name2tid.find(f[6]); if (it == name2tid.end()) return bad("RNEXT not in the header"); mtid = it->second; }
        std::vector<uint32_t> cigar;
        int64_t reflen = 0, qlen = 0;
        if (f[5] != "*") {
            uint64_t num = 0; bool have = false;
            for (const char ch : f[5]) {
                if (ch >= '0' && ch <= '9') { num = num * 10 + (uint64_t)(ch - '0'); have = true; if (num >= (1u << 28)) return bad("CIGAR length"); continue; }
                const char *c = strchr(CIGAR_CODES, ch);
                if (!c || !have || !ch) return bad("CIGAR");
                const uint32_t op = (uint32_t)(c - CIGAR_CODES);
                cigar.push_back((uint32_t)num << 4 | op);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += (int64_t)num;
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += (int64_t)num;
                num = 0; have = false;
            }
            if (have) return bad("CIGAR");
        }
        if (cigar.size() > 65535) return bad("more than 65535 CIGAR operations");
        const std::string &seq = f[9], &qual = f[10];
        const uint32_t l_seq = seq == "*" ? 0u : (uint32_t)seq.size();
        if (l_seq && !cigar.empty() && qlen != (int64_t)l_seq) return bad("CIGAR and query sequence lengths differ");
        if (qual != "*" && qual.size() != l_seq) return bad("SEQ and QUAL lengths differ");
        if (f[0].empty() || f[0].size() > 254) return bad("QNAME");
        std::vector<uint8_t> aux;
        for (size_t k = 11; k < f.size(); ++k) if (!parse_aux(f[k], aux)) return bad("optional field");
        const int64_t p0 = pos - 1;                      // 0-based; POS 0 = unplaced -> -1
        const int64_t end = p0 + (reflen ? reflen : 1);
        const uint32_t l_qname = (uint32_t)f[0].size() + 1;
        const uint32_t bs = 32 + l_qname + 4u * (uint32_t)cigar.size() + (l_seq + 1) / 2 + l_seq + (uint32_t)aux.size();
        if (!raw.empty() && raw.size() + 4 + bs > BGZF_BLOCK) { bgzf_block(bam, raw.data(), raw.size()); raw.clear(); }     // whole records per block, as htslib writes them
        put32(raw, bs); put32(raw, (uint32_t)tid); put32(raw, (uint32_t)(int32_t)p0);
        raw.push_back((uint8_t)l_qname); raw.push_back((uint8_t)mapq); put16(raw, (uint32_t)reg2bin(p0 < 0 ? 0 : p0, end < 1 ? 1 : end));
        put16(raw, (uint32_t)cigar.size()); put16(raw, (uint32_t)flag); put32(raw, l_seq);
        put32(raw, (uint32_t)mtid); put32(raw, (uint32_t)(int32_t)(pnext - 1)); put32(raw, (uint32_t)(int32_t)tlen);
        raw.insert(raw.end(), f[0].begin(), f[0].end()); raw.push_back(0);
        for (const uint32_t c : cigar) put32(raw, c);
        for (uint32_t k = 0; k < l_seq; k += 2) {
            auto code = [&](char ch) { const char u = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch; const char *c = strchr(SEQ_CODES, u); return (c && u) ? (uint32_t)(c - SEQ_CODES) : 15u; };
            raw.push_back((uint8_t)(code(seq[k]) << 4 | (k + 1 < l_seq ? code(seq[k + 1]) : 0u)));
        }
        for (uint32_t k = 0; k < l_seq; ++k) raw.push_back(qual == "*" ? 0xffu : (uint8_t)(qual[k] - 33));
        raw.insert(raw.end(), aux.begin(), aux.end());
        // a single record larger than a block spills over several (htslib does the same)
        while (raw.size() > BGZF_BLOCK) { bgzf_block(bam, raw.data(), BGZF_BLOCK); raw.erase(raw.begin(), raw.begin() + (long)BGZF_BLOCK); }
    }
    if (!header_done) flush_header();
    if (!raw.empty()) bgzf_block(bam, raw.data(), raw.size());
    static const uint8_t eof_blk[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bam.insert(bam.end(), eof_blk, eof_blk + 28);
    return true;
}

// ---- BAM record -> SAM line (htslib sam_format1) ---------------------------------------------------------------
bool sam_format_record(const std::vector<BamRef> &refs, const uint8_t *p, uint32_t len, const char *xm, uint32_t xm_len, std::string &out) {
    if (len < 32) return false;
    const int32_t tid = read_i32(p), pos = read_i32(p + 4);
    const uint32_t l_qname = p[8], mapq = p[9], n_cigar = read_u16(p + 12), flag = read_u16(p + 14), l_seq = read_u32(p + 16);
    const int32_t mtid = read_i32(p + 20), mpos = read_i32(p + 24), tlen = read_i32(p + 28);
    const size_t o_cigar = 32 + (size_t)l_qname, o_seq = o_cigar + 4ull * n_cigar, o_qual = o_seq + ((size_t)l_seq + 1) / 2, o_aux = o_qual + l_seq;
    if (o_aux > len || l_qname == 0) return false;
    char num[32];
    auto put_int = [&](long long v) { out.append(num, (size_t)snprintf(num, sizeof num, "%lld", v)); };
    out.append(reinterpret_cast<const char *>(p + 32), strnlen(reinterpret_cast<const char *>(p + 32), l_qname - 1)); out += '\t';
    put_int(flag); out += '\t';
    if (tid >= 0 && tid < (int32_t)refs.size()) out += refs[(size_t)tid].name; else out += '*';
    out += '\t'; put_int((long long)pos + 1); out += '\t'; put_int(mapq); out += '\t';
    if (n_cigar == 0) out += '*';
    for (uint32_t k = 0; k < n_cigar; ++k) { const uint32_t c = read_u32(p + o_cigar + 4 * k); put_int(c >> 4); out += (c & 15u) < 9 ? CIGAR_CODES[c & 15u] : '?'; }
    out += '\t';
    if (mtid < 0) out += '*'; else if (mtid == tid) out += '='; else if (mtid < (int32_t)refs.size()) out += refs[(size_t)mtid].name; else out += '*';
    out += '\t'; put_int((long long)mpos + 1); out += '\t'; put_int(tlen); out += '\t';
    if (l_seq == 0) { out += "*\t*"; }
    else {
        for (uint32_t k = 0; k < l_seq; ++k) { const uint8_t b = p[o_seq + (k >> 1)]; out += SEQ_CODES[(k & 1u) ? (b & 15u) : (b >> 4)]; }
        out += '\t';
        if (p[o_qual] == 0xff) out += '*';
        else for (uint32_t k = 0; k < l_seq; ++k) out += (char)(p[o_qual + k] + 33);
    }
    // optional fields (SAM spec 4.2.4); every integer type prints as :i:
    size_t o = o_aux;
    while (o + 3 <= len) {
        out += '\t'; out += (char)p[o]; out += (char)p[o + 1]; out += ':';
        const uint8_t ty = p[o + 2];
        o += 3;
        auto need = [&](size_t n) { return o + n <= len; };
        switch (ty) {
            case 'A': if (!need(1)) return false; out += "A:"; out += (char)p[o]; o += 1; break;
            case 'c': if (!need(1)) return false; out += "i:"; put_int((int8_t)p[o]); o += 1; break;
            case 'C': if (!need(1)) return false; out += "i:"; put_int(p[o]); o += 1; break;
            case 's': if (!need(2)) return false; out += "i:"; put_int((int16_t)read_u16(p + o)); o += 2; break;
            case 'S': if (!need(2)) return false; out += "i:"; put_int(read_u16(p + o)); o += 2; break;
            case 'i': if (!need(4)) return false; out += "i:"; put_int(read_i32(p + o)); o += 4; break;
            case 'I': if (!need(4)) return false; out += "i:"; put_int(read_u32(p + o)); o += 4; break;
            case 'f': { if (!need(4)) return false; float x; const uint32_t u = read_u32(p + o); memcpy(&x, &u, 4); out += "f:"; out.append(num, (size_t)snprintf(num, sizeof num, "%g", x)); o += 4; break; }
            case 'Z': case 'H': {
                out += (char)ty; out += ':';
                const void *z = memchr(p + o, 0, len - o);
                if (!z) return false;
                out.append(reinterpret_cast<const char *>(p + o), (size_t)(static_cast<const uint8_t *>(z) - (p + o)));
                o = (size_t)(static_cast<const uint8_t *>(z) - p) + 1;
                break;
            }
            case 'B': {
                if (!need(5)) return false;
                const uint8_t sub = p[o];
                const uint64_t cnt = read_u32(p + o + 1);
                const uint64_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
                if (!w || o + 5 + cnt * w > len) return false;
                out += "B:"; out += (char)sub;
                o += 5;
                for (uint64_t k = 0; k < cnt; ++k, o += w) {
                    out += ',';
                    switch (sub) {
                        case 'c': put_int((int8_t)p[o]); break;
                        case 'C': put_int(p[o]); break;
                        case 's': put_int((int16_t)read_u16(p + o)); break;
                        case 'S': put_int(read_u16(p + o)); break;
                        case 'i': put_int(read_i32(p + o)); break;
                        case 'I': put_int(read_u32(p + o)); break;
                        default: { float x; const uint32_t u = read_u32(p + o); memcpy(&x, &u, 4); out.append(num, (size_t)snprintf(num, sizeof num, "%g", x)); }
                    }
                }
                break;
            }
            default: return false;
        }
    }
    if (xm) { out += "\tXM:Z:"; out.append(xm, xm_len); }      // push_aux(b"XM", Aux::String(..)) appends at the end (tag.rs:437)
    out += '\n';
    return true;
}

// ---- FASTA ------------------------------------------------------------------------------------------------------
bool Fasta::open(const std::string &path, std::string &err) {
    path_ = path;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = (errno == ENOENT ? "file not found: " : "unable to open: ") + path; return false; }   // rust-htslib Error::FileNotFound
    fclose(f);
    std::ifstream fai(path + ".fai");
    if (fai) {
        std::string line;
        while (std::getline(fai, line)) {
            const auto c = split_tab(line);
            Entry e;
            int64_t a, b, cc, d;
            if (c.size() < 5 || !parse_i64(c[1], a) || !parse_i64(c[2], b) || !parse_i64(c[3], cc) || !parse_i64(c[4], d) || cc <= 0 || d < cc) { err = "malformed FASTA index: " + path + ".fai"; return false; }
            e.length = a; e.offset = b; e.line_bases = cc; e.line_width = d;
            index_.emplace(c[0], e);
        }
        return true;
    }
    // no index: one pass over the file (name = up to the first white space, as faidx keys its entries)
    std::ifstream in(path, std::ios::binary);
    std::string line, name;
    Entry e{};
    int64_t off = 0;
    bool have = false, short_seen = false;
    auto close_entry = [&]() { if (have) index_.emplace(name, e); };
    while (std::getline(in, line)) {
        const int64_t raw_len = (int64_t)line.size() + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            close_entry();
            size_t k = 1;
            while (k < line.size() && !isspace((unsigned char)line[k])) ++k;
            name = line.substr(1, k - 1);
            e = Entry{}; e.offset = off + raw_len; have = true; short_seen = false;
        } else if (have && !line.empty()) {
            if (e.line_bases == 0) { e.line_bases = (int64_t)line.size(); e.line_width = raw_len; }
            else if (short_seen || (int64_t)line.size() > e.line_bases) { err = "FASTA lines of different length in one sequence (cannot be indexed): " + path; return false; }
            if ((int64_t)line.size() < e.line_bases) short_seen = true;
            e.length += (int64_t)line.size();
        }
        off += raw_len;
    }
    close_entry();
    return true;
}

bool Fasta::fetch(const std::string &name, int64_t end_incl, std::vector<uint8_t> &seq, std::string &err) const {
    seq.clear();
    const auto it = index_.find(name);
    if (it == index_.end()) { err = "sequence not in the FASTA: " + name; return false; }
    const Entry &e = it->second;
    const int64_t n = std::min<int64_t>(end_incl + 1, e.length);         // faidx_fetch_seq(name, 0, end): [0, end] clipped to the sequence
    if (n <= 0) return true;
    FILE *f = fopen(path_.c_str(), "rb");
    if (!f) { err = "unable to open: " + path_; return false; }
    const int64_t lines = (n + e.line_bases - 1) / e.line_bases;
    const int64_t bytes = n + lines * (e.line_width - e.line_bases);
    std::vector<uint8_t> buf((size_t)bytes);
    bool ok = fseeko(f, (off_t)e.offset, SEEK_SET) == 0;
    const size_t got = ok ? fread(buf.data(), 1, buf.size(), f) : 0;
    fclose(f);
    seq.reserve((size_t)n);
    for (size_t k = 0; k < got && (int64_t)seq.size() < n; ++k) if (isgraph(buf[k])) seq.push_back(buf[k]);   // faidx keeps isgraph() characters
    if ((int64_t)seq.size() != n) { err = "FASTA shorter than its index says: " + path_; return false; }
    return true;
}

}  // namespace mthh
