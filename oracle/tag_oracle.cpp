// tag_oracle.cpp -- CPU ORACLE of the `tag` subcommand's per-record function.  TEST INFRASTRUCTURE ONLY
// (same rules as metheor_oracle.cpp: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it).
//
// A literal restatement of dohlee/metheor v0.1.9 src/tag.rs:130-384 `determine_xm_tag_string` with the same
// intermediate strings (read / reference columns with '-' gaps, two flank columns on each side), including what
// the reference does NOT handle: only CIGAR M, I and D are walked (tag.rs:185-237; S, N, =, X, H, P fall into
// `_ => {}`), so a soft clip shifts the read bases and a reference skip shifts the reference bases exactly as there.
// Where the reference panics (rcmapping miss tag.rs:24, index past the collected context tag.rs:297, an alignment
// outside the contig tag.rs:158) the function returns -1.
//
// Parity pin: tests/test_tag_golden.py runs it over the reference's own 1000-read fixture
// (tests/test.chr19.noXM.sam -> tests/test.chr19.XM.sam == tests/test.chr19.metheor_tag_out.sam, tests/tag-cli.rs:60-80)
// with the chr19 bases rebuilt from the reads' MD:Z tags (the 58-MB hg38.chr19.fa is not shipped).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "metheor_oracle.h"

namespace {

// tag.rs:74-96
int rcmap(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N';
        case 'M': return 'K'; case 'R': return 'Y'; case 'W': return 'W'; case 'S': return 'S'; case 'Y': return 'R';
        case 'K': return 'M'; case 'V': return 'B'; case 'H': return 'D'; case 'D': return 'H'; case 'B': return 'V';
        case '-': return '-';
        default: return -1;   // HashMap index panics
    }
}
// tag.rs:19-25
bool reverse_complement(const std::string &seq, std::string &res) {
    res.clear();
    for (size_t k = seq.size(); k-- > 0;) {
        const int c = rcmap(seq[k]);
        if (c < 0) return false;
        res.push_back((char)c);
    }
    return true;
}
// tag.rs:31-50
bool is_chg_context(const std::string &s) { return s == "CAG" || s == "CTG" || s == "CCG"; }
bool is_chh_context(const std::string &s) {
    return s == "CAA" || s == "CAT" || s == "CAC" || s == "CTA" || s == "CTT" || s == "CTC" || s == "CCA" || s == "CCT" || s == "CCC";
}
bool is_unknown_context(const std::string &s) {
    for (char b : s) if (b == '-' || b == 'N') return true;
    return false;
}
char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// the six-way letter choice that closes every context branch (tag.rs:298-337, 343-383); 0 = nothing pushed
char context_letter(const std::string &ref_context, bool is_cg, char read_base) {
    char hi, lo;
    if (is_cg) { hi = 'Z'; lo = 'z'; }
    else if (is_chg_context(ref_context)) { hi = 'X'; lo = 'x'; }
    else if (is_chh_context(ref_context)) { hi = 'H'; lo = 'h'; }
    else if (is_unknown_context(ref_context)) { hi = 'U'; lo = 'u'; }
    else return 0;
    return read_base == 'C' ? hi : (read_base == 'T' ? lo : '.');
}

}  // namespace

extern "C" int64_t orc_tag_xm(int32_t pos, int32_t ref_end, uint16_t flag, int is_paired_end, const uint32_t *cigar,
                              uint32_t n_cigar, const char *seq, int64_t l_seq, const char *contig, int64_t chromsize,
                              char *out, int64_t out_cap) {
    const int64_t start = pos, end = ref_end;
    // tag.rs:141-144, 15-18
    const bool is_reverse = flag & 16, first = flag & 64, last = flag & 128;
    const bool flag_reverse_complement = is_paired_end ? !((!is_reverse && first) || (is_reverse && last)) : is_reverse;
    // tag.rs:147-150
    std::string read_seq(seq, (size_t)l_seq);
    for (char &c : read_seq) c = up(c);
    // tag.rs:154-164
    const int64_t clipped_start = std::max<int64_t>(start - 2, 0), clipped_end = std::min<int64_t>(end + 2, chromsize);
    if (clipped_start > clipped_end || clipped_end > chromsize) return -1;        // slice index panics
    std::string ref_seq(contig + clipped_start, (size_t)(clipped_end - clipped_start));
    for (char &c : ref_seq) c = up(c);
    // tag.rs:167-173
    static const char *padding[3] = {"", "N", "NN"};
    const int64_t pad_nbases_start = std::max<int64_t>(2 - start, 0), pad_nbases_end = std::max<int64_t>(end - chromsize + 2, 0);
    if (pad_nbases_start > 2 || pad_nbases_end > 2) return -1;                    // padding[] index panics
    ref_seq = std::string(padding[pad_nbases_start]) + ref_seq + padding[pad_nbases_end];
    if (ref_seq.size() < 2) return -1;                                            // .next().unwrap() / nth(1).unwrap()

    // tag.rs:175-185
    std::string tmp_read_seq = "--", tmp_ref_seq;
    tmp_ref_seq.push_back(ref_seq[0]);
    tmp_ref_seq.push_back(ref_seq[1]);
    size_t used_read_len = 0, used_ref_len = 2;
    auto skip_take = [](const std::string &s, size_t skip, size_t take) {         // chars().skip(a).take(b): short near the end, never a panic
        if (skip >= s.size()) return std::string();
        return s.substr(skip, take);
    };
    for (uint32_t k = 0; k < n_cigar; ++k) {                                       // tag.rs:187-237
        const uint32_t op = cigar[k] & 15u;
        const size_t length = cigar[k] >> 4;
        if (op == 0) {           // Cigar::Match
            tmp_read_seq += skip_take(read_seq, used_read_len, length);
            tmp_ref_seq += skip_take(ref_seq, used_ref_len, length);
            used_read_len += length; used_ref_len += length;
        } else if (op == 1) {    // Cigar::Ins
            tmp_read_seq += skip_take(read_seq, used_read_len, length);
            tmp_ref_seq += std::string(length, '-');
            used_read_len += length;
        } else if (op == 2) {    // Cigar::Del
            tmp_read_seq += std::string(length, '-');
            tmp_ref_seq += skip_take(ref_seq, used_ref_len, length);
            used_ref_len += length;
        }                        // _ => {}
    }
    // tag.rs:239-243
    tmp_read_seq += "--";
    tmp_ref_seq.push_back(ref_seq[ref_seq.size() - 2]);
    tmp_ref_seq.push_back(ref_seq[ref_seq.size() - 1]);

    // tag.rs:245-262
    std::string target_read_seq, target_ref_seq;
    if (flag_reverse_complement) {
        if (!reverse_complement(tmp_read_seq.substr(0, tmp_read_seq.size() - 2), target_read_seq)) return -1;
        if (!reverse_complement(tmp_ref_seq.substr(0, tmp_ref_seq.size() - 2), target_ref_seq)) return -1;
    } else {
        target_read_seq = tmp_read_seq.substr(2);
        target_ref_seq = tmp_ref_seq.substr(2);
    }
    auto rd = [&](size_t i, bool &ok) { if (i >= target_read_seq.size()) { ok = false; return '\0'; } return target_read_seq[i]; };   // char_at: nth().unwrap()
    auto rf = [&](size_t i, bool &ok) { if (i >= target_ref_seq.size()) { ok = false; return '\0'; } return target_ref_seq[i]; };

    std::string xm_tag;
    bool ok = true;
    const size_t tlen = target_read_seq.size();
    for (size_t idx = 0; idx + 2 < tlen && ok; ++idx) {                            // tag.rs:265  0..len-2
        const char r0 = rd(idx, ok);
        if (r0 == '-') continue;
        if (r0 == 'N') { xm_tag.push_back('.'); continue; }
        if (rf(idx, ok) == 'C') {
            if ((rd(idx + 1, ok) == '-' || rd(idx + 2, ok) == '-') && (idx != tlen - 3) && (idx != tlen - 4)) {   // tag.rs:271-274
                std::string tr, tg;                                                // tag.rs:275-296
                tr.push_back(r0);
                tg.push_back(rf(idx, ok));
                int flag_tmp = 0;
                size_t tmp_count = 1;
                while (flag_tmp != 2) {
                    if (idx + tmp_count > tlen - 1) break;
                    if (rd(idx + tmp_count, ok) != '-') {
                        tr.push_back(rd(idx + tmp_count, ok));
                        tg.push_back(rf(idx + tmp_count, ok));
                        flag_tmp += 1;
                    }
                    tmp_count += 1;
                }
                if (tg.size() < 2) { ok = false; break; }                          // tmp_target_ref_seq[1] panics
                const char c = context_letter(tg, tg[0] == 'C' && tg[1] == 'G', tr[0]);
                if (c) xm_tag.push_back(c);
            } else {                                                               // tag.rs:340-383
                const std::string ref_context = skip_take(target_ref_seq, idx, 3);
                const char c = context_letter(ref_context, rf(idx, ok) == 'C' && rf(idx + 1, ok) == 'G', r0);
                if (c) xm_tag.push_back(c);
            }
        } else {
            xm_tag.push_back('.');
        }
    }
    if (!ok) return -1;
    if (flag_reverse_complement) std::reverse(xm_tag.begin(), xm_tag.end());      // tag.rs:386-389
    if ((int64_t)xm_tag.size() > out_cap) return -1;
    memcpy(out, xm_tag.data(), xm_tag.size());
    return (int64_t)xm_tag.size();
}
