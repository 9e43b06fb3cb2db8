// metheor_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// A literal, single-threaded restatement of the reference's per-read CpG-pattern
// hot path (dohlee/metheor v0.1.9).  Stream order, hash-map + retain flush structure
// and f32 expressions follow the cited reference lines; nothing here is tuned.
// Every function cites the reference file:line it follows (paths under /root/reference).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// Third-party semantics restated (not vendored under /root/reference):
//  * rust-htslib 0.50.0 Record::reference_positions_full(): one Option<i64> per QUERY base;
//    None for insertions / soft clips; deletions and ref-skips yield nothing.
//  * rand 0.8.5 thread_rng().gen_range(1..=n) (fdrp.rs:90) is OS-seeded and therefore not
//    reproducible: replaced by the counter-based orc_sample_j() -- "parity unpinned" branch.
//  * itertools combinations(2): lexicographic (i<j) order.
#include "metheor_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <vector>

namespace {

struct Pos {  // readutil.rs:279-283 CpGPosition
    int32_t tid, pos;
    bool operator==(const Pos &o) const { return tid == o.tid && pos == o.pos; }
    bool operator<(const Pos &o) const {  // readutil.rs:311-315
        return tid != o.tid ? tid < o.tid : pos < o.pos;
    }
    bool is_before(const Pos &o, int32_t d) const {  // readutil.rs:290-296
        if (tid > o.tid) return false;
        if (tid < o.tid) return true;
        return pos + d < o.pos;
    }
};
struct PosHash {
    size_t operator()(const Pos &p) const {
        uint64_t x = ((uint64_t)(uint32_t)p.tid << 32) | (uint32_t)p.pos;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        return (size_t)x;
    }
};

struct CpG {  // readutil.rs:247-251
    int32_t relpos;
    Pos abspos;
    bool methylated;
};

struct Read {  // readutil.rs:15-21 BismarkRead (+ the record fields the drivers look at)
    int32_t tid, start_pos, end_pos;
    uint8_t mapq, fwd;
    std::vector<CpG> cpgs;
};

}  // namespace

struct orc_reads {
    std::vector<Read> reads;
};

struct orc_result {
    int k = 1, m = 0;
    std::vector<int32_t> tid, pos;
    std::vector<float> val;
    std::vector<uint32_t> cnt;
};

namespace {

// ---- readutil.rs:24-53 (BismarkRead::new) + 323-345 (get_cpgs) -------------------------
bool decode_one(const orc_records_t *rec, int64_t i, Read &out) {
    out.tid = rec->tid[i];
    out.mapq = rec->mapq[i];
    const uint16_t flag = rec->flag[i];
    // readutil.rs:332: "forward" iff flags is EXACTLY 0, 99 or 147
    out.fwd = (flag == 0 || flag == 99 || flag == 147) ? 1 : 0;
    // reference_positions_full(): one entry per query base
    std::vector<int64_t> refpos;  // -1 == None
    int64_t r = rec->pos[i];
    for (uint32_t c = rec->cigar_off[i]; c < rec->cigar_off[i + 1]; ++c) {
        const uint32_t op = rec->cigar[c] & 15u, len = rec->cigar[c] >> 4;
        switch (op) {
            case 0: case 7: case 8:  // M = X : query+ref
                for (uint32_t t = 0; t < len; ++t) refpos.push_back(r++);
                break;
            case 1: case 4:  // I S : query only -> None
                for (uint32_t t = 0; t < len; ++t) refpos.push_back(-1);
                break;
            case 2: case 3:  // D N : ref only -> nothing yielded
                r += len;
                break;
            default: break;  // H P
        }
    }
    // readutil.rs:25-33
    out.start_pos = -1; out.end_pos = -1;
    for (int64_t p : refpos) {
        if (p < 0) continue;
        if (out.start_pos == -1) out.start_pos = (int32_t)p;
        out.end_pos = (int32_t)p;
    }
    // readutil.rs:35-52 : XM must exist
    const uint32_t x0 = rec->xm_off[i], x1 = rec->xm_off[i + 1];
    if (x0 == x1) return false;
    // readutil.rs:326 : zip(reference_positions_full, xm.chars()).enumerate()
    const size_t nz = std::min<size_t>(refpos.size(), x1 - x0);
    out.cpgs.clear();
    for (size_t relpos = 0; relpos < nz; ++relpos) {
        const char c = rec->xm[x0 + relpos];
        if (c != 'z' && c != 'Z') continue;            // readutil.rs:327
        if (refpos[relpos] < 0) continue;              // readutil.rs:331
        const int32_t ap = (int32_t)(out.fwd ? refpos[relpos] : refpos[relpos] - 1);  // 332-340
        out.cpgs.push_back(CpG{(int32_t)relpos, Pos{out.tid, ap}, c == 'Z'});        // 254-260
    }
    return true;
}

// readutil.rs:134-145
bool is_discordant(const Read &r) {
    const bool init = r.cpgs[0].methylated;
    bool d = false;
    for (const CpG &c : r.cpgs) if (c.methylated != init) d = true;
    return d;
}

// readutil.rs:147-164 ; HashMap<i32,i32> restated as an ordered map (key order is irrelevant
// for the integer sums; it only fixes the f32 summation order in compute_mhl, see there)
std::map<int32_t, int32_t> stretch_info(const Read &r) {
    std::map<int32_t, int32_t> s;
    int32_t cur = 0;
    for (const CpG &c : r.cpgs) {
        if (c.methylated) {
            cur += 1;
            for (int32_t l = 1; l < cur + 1; ++l) s[l] += 1;
        } else {
            cur = 0;
        }
    }
    return s;
}

struct PairRec { Pos a, b; bool concordant; };

// readutil.rs:166-224
void pairwise(const Read &r, int32_t min_distance, int32_t max_distance, int32_t &n_conc,
              int32_t &n_disc, std::vector<PairRec> *pairs) {
    std::vector<CpG> anchors;
    int32_t min_anchor_pos = -1;
    n_conc = 0; n_disc = 0;
    for (const CpG &cpg : r.cpgs) {
        if (min_anchor_pos != -1) {
            while ((cpg.relpos - min_anchor_pos > max_distance) && !anchors.empty()) {
                anchors.erase(anchors.begin());
                min_anchor_pos = anchors.empty() ? -1 : anchors[0].relpos;
            }
        }
        for (const CpG &anchor : anchors) {
            if (cpg.relpos - anchor.relpos < min_distance) continue;
            if (anchor.methylated == cpg.methylated) {
                n_conc += 1;
                if (pairs) pairs->push_back(PairRec{anchor.abspos, cpg.abspos, true});
            } else {
                n_disc += 1;
                if (pairs) pairs->push_back(PairRec{anchor.abspos, cpg.abspos, false});
            }
        }
        if (min_anchor_pos == -1) min_anchor_pos = cpg.relpos;
        anchors.push_back(cpg);
    }
}

template <class K, class V, class H, class F>
void retain(std::unordered_map<K, V, H> &m, F keep) {  // HashMap::retain
    for (auto it = m.begin(); it != m.end();) {
        if (keep(it->first, it->second)) ++it; else it = m.erase(it);
    }
}

}  // namespace

extern "C" {

orc_reads_t *orc_decode(const orc_records_t *rec, int64_t set_n, const int32_t *set_tid,
                        const int32_t *set_pos, int *err) {
    if (err) *err = 0;
    auto *out = new orc_reads;
    out->reads.resize(rec->n);
    // readutil.rs:347-374 get_target_cpgs -> HashSet<CpGPosition>
    std::unordered_map<Pos, char, PosHash> target;
    for (int64_t i = 0; i < set_n; ++i) target[Pos{set_tid[i], set_pos[i]}] = 1;
    for (int64_t i = 0; i < rec->n; ++i) {
        if (!decode_one(rec, i, out->reads[i])) {
            if (err) *err = 1;
            delete out;
            return nullptr;
        }
        if (set_n > 0) {  // readutil.rs:87-95 filter_isin (relpos preserved)
            std::vector<CpG> kept;
            for (const CpG &c : out->reads[i].cpgs) if (target.count(c.abspos)) kept.push_back(c);
            out->reads[i].cpgs.swap(kept);
        }
    }
    return out;
}

orc_reads_t *orc_reads_from_soa(int64_t n, const int32_t *tid, const int32_t *start,
                                const int32_t *end, const uint8_t *mapq, const uint8_t *fwd,
                                const uint64_t *cpg_off, const uint32_t *cpg_pos,
                                const uint16_t *cpg_rel) {
    auto *out = new orc_reads;
    out->reads.resize(n);
    for (int64_t i = 0; i < n; ++i) {
        Read &r = out->reads[i];
        r.tid = tid[i]; r.start_pos = start[i]; r.end_pos = end[i];
        r.mapq = mapq[i]; r.fwd = fwd ? fwd[i] : 1;
        r.cpgs.reserve(cpg_off[i + 1] - cpg_off[i]);
        for (uint64_t c = cpg_off[i]; c < cpg_off[i + 1]; ++c)
            r.cpgs.push_back(CpG{(int32_t)cpg_rel[c], Pos{r.tid, (int32_t)(cpg_pos[c] & 0x7fffffffu)},
                                 (cpg_pos[c] >> 31) != 0});
    }
    return out;
}

void orc_reads_free(orc_reads_t *r) { delete r; }
int64_t orc_reads_n(const orc_reads_t *r) { return (int64_t)r->reads.size(); }
int64_t orc_reads_ncpg(const orc_reads_t *r) {
    int64_t n = 0;
    for (const Read &x : r->reads) n += (int64_t)x.cpgs.size();
    return n;
}
void orc_reads_export(const orc_reads_t *r, int32_t *tid, int32_t *start, int32_t *end,
                      uint8_t *mapq, uint8_t *fwd, uint64_t *cpg_off, uint32_t *cpg_pos,
                      uint16_t *cpg_rel) {
    uint64_t o = 0;
    for (size_t i = 0; i < r->reads.size(); ++i) {
        const Read &x = r->reads[i];
        if (tid) tid[i] = x.tid;
        if (start) start[i] = x.start_pos;
        if (end) end[i] = x.end_pos;
        if (mapq) mapq[i] = x.mapq;
        if (fwd) fwd[i] = x.fwd;
        if (cpg_off) cpg_off[i] = o;
        for (const CpG &c : x.cpgs) {
            if (cpg_pos) cpg_pos[o] = ((uint32_t)c.abspos.pos & 0x7fffffffu) | (c.methylated ? 0x80000000u : 0u);
            if (cpg_rel) cpg_rel[o] = (uint16_t)c.relpos;
            ++o;
        }
    }
    if (cpg_off) cpg_off[r->reads.size()] = o;
}

int orc_read_is_discordant(const orc_reads_t *r, int64_t i) {
    const Read &x = r->reads[i];
    if (x.cpgs.empty()) return -1;  // the reference would index out of bounds (readutil.rs:135)
    return is_discordant(x) ? 1 : 0;
}
int orc_read_stretch_info(const orc_reads_t *r, int64_t i, int32_t *counts, int cap) {
    auto s = stretch_info(r->reads[i]);
    int mx = 0;
    for (int l = 0; l < cap; ++l) counts[l] = 0;
    for (auto &kv : s) {
        if (kv.first <= cap) counts[kv.first - 1] = kv.second;
        mx = std::max(mx, (int)kv.first);
    }
    return mx;
}
void orc_read_pairwise(const orc_reads_t *r, int64_t i, int32_t min_d, int32_t max_d,
                       int32_t *n_conc, int32_t *n_disc) {
    pairwise(r->reads[i], min_d, max_d, *n_conc, *n_disc, nullptr);
}

// ---- pdr.rs:119-212 -----------------------------------------------------------------
orc_result_t *orc_pdr(const orc_reads_t *rd, uint32_t min_depth, uint64_t min_cpgs, uint8_t min_qual) {
    struct PDR { uint32_t n_conc = 0, n_disc = 0; };  // pdr.rs:12-16
    struct Out { float pdr; uint32_t c, d; };
    std::unordered_map<Pos, PDR, PosHash> cpg2reads;  // pdr.rs:131
    std::map<Pos, Out> result;                        // pdr.rs:136 (BTreeMap)
    auto finalize = [&](const Pos &cpg, const PDR &p) {
        if (p.n_conc + p.n_disc >= min_depth) {  // pdr.rs:163 / 200
            // pdr.rs:47-49
            const float pdr = (float)p.n_disc / ((float)p.n_conc + (float)p.n_disc);
            result[cpg] = Out{pdr, p.n_conc, p.n_disc};  // BTreeMap::insert overwrites
        }
    };
    for (const Read &br : rd->reads) {
        if (br.cpgs.size() < min_cpgs) continue;  // pdr.rs:147
        if (br.mapq < min_qual) continue;         // pdr.rs:150
        if (br.cpgs.empty()) continue;            // pdr.rs:155
        const Pos first = br.cpgs[0].abspos;      // pdr.rs:159
        retain(cpg2reads, [&](const Pos &cpg, const PDR &p) {  // pdr.rs:160-177
            if (cpg.is_before(first, 150)) { finalize(cpg, p); return false; }
            return true;
        });
        for (const CpG &c : br.cpgs) {  // pdr.rs:180-191 (state recomputed per CpG, as there)
            PDR &r = cpg2reads[c.abspos];
            if (is_discordant(br)) r.n_disc += 1; else r.n_conc += 1;
        }
    }
    for (auto &kv : cpg2reads) finalize(kv.first, kv.second);  // pdr.rs:199-210
    auto *res = new orc_result;
    res->k = 1; res->m = 2;
    for (auto &kv : result) {
        res->tid.push_back(kv.first.tid); res->pos.push_back(kv.first.pos);
        res->val.push_back(kv.second.pdr);
        res->cnt.push_back(kv.second.c); res->cnt.push_back(kv.second.d);
    }
    return res;
}

// ---- lpmd.rs:154-202 ----------------------------------------------------------------
orc_result_t *orc_lpmd(const orc_reads_t *rd, int32_t min_d, int32_t max_d, uint8_t min_qual,
                       int want_pairs, int64_t globals[4], float *lpmd) {
    int64_t n_read = 0, n_valid = 0, n_conc = 0, n_disc = 0;
    struct PairKey {
        Pos a, b;
        bool operator==(const PairKey &o) const { return a == o.a && b == o.b; }
        bool operator<(const PairKey &o) const {
            if (!(a == o.a)) return a < o.a;
            return b < o.b;
        }
    };
    struct PairHash {
        size_t operator()(const PairKey &k) const {
            return PosHash()(k.a) * 0x9e3779b97f4a7c15ULL ^ PosHash()(k.b);
        }
    };
    // lpmd.rs:13-14: two HashMaps keyed by the pair, BOTH upserted for every pair (76-77) whether or
    // not --pairs was asked for; never flushed.  Kept as-is: this is the reference's cost structure.
    std::unordered_map<PairKey, int32_t, PairHash> pair2n_conc, pair2n_disc;
    std::vector<PairRec> pv;
    for (const Read &br : rd->reads) {
        n_read += 1;                        // lpmd.rs:176
        if (br.mapq < min_qual) continue;   // lpmd.rs:177 (before XM decode; decode has no side effect)
        int32_t c, d;
        pv.clear();
        pairwise(br, min_d, max_d, c, d, &pv);                          // lpmd.rs:186-187
        n_valid += 1; n_conc += c; n_disc += d;                         // lpmd.rs:189-191
        for (const PairRec &p : pv) {                                   // lpmd.rs:192-194, 70-87
            int32_t &nc = pair2n_conc[PairKey{p.a, p.b}];
            int32_t &nd = pair2n_disc[PairKey{p.a, p.b}];
            if (p.concordant) nc += 1; else nd += 1;
        }
    }
    // print_pair_statistics sorts the keys (lpmd.rs:90-94)
    std::map<PairKey, std::pair<int32_t, int32_t>> pairs;
    if (want_pairs)
        for (auto &kv : pair2n_conc) pairs[kv.first] = std::make_pair(kv.second, pair2n_disc[kv.first]);
    globals[0] = n_conc; globals[1] = n_disc; globals[2] = n_read; globals[3] = n_valid;
    // lpmd.rs:11-12 the counters are i32 (wrapping in a release build); 51-55:
    const int32_t wc = (int32_t)(uint32_t)(uint64_t)n_conc, wd = (int32_t)(uint32_t)(uint64_t)n_disc;
    const int32_t wsum = (int32_t)((uint32_t)wc + (uint32_t)wd);
    *lpmd = (float)wd / (float)wsum;
    auto *res = new orc_result;
    res->k = 2; res->m = 2;
    for (auto &kv : pairs) {
        res->tid.push_back(kv.first.a.tid);
        res->pos.push_back(kv.first.a.pos); res->pos.push_back(kv.first.b.pos);
        const int32_t pc = kv.second.first, pd = kv.second.second;
        res->val.push_back((float)pd / ((float)pc + (float)pd));  // lpmd.rs:111
        res->cnt.push_back((uint32_t)pc); res->cnt.push_back((uint32_t)pd);
    }
    return res;
}

// ---- mhl.rs:135-208 -----------------------------------------------------------------
orc_result_t *orc_mhl(const orc_reads_t *rd, uint32_t min_depth, uint64_t min_cpgs, uint8_t min_qual) {
    struct Assoc {  // mhl.rs:12-17
        std::map<int32_t, int32_t> stretch;  // HashMap in the reference: its f32 sum order (mhl.rs:50)
                                             // is random there; ascending l is fixed here.
        std::vector<int32_t> num_cpgs;
        size_t max_num_cpgs = 0;
    };
    auto compute_mhl = [](const Assoc &a) -> float {  // mhl.rs:43-73
        float mhl = 0.0f, l_sum = 0.0f;
        for (size_t l = 1; l < a.max_num_cpgs + 1; ++l) l_sum += (float)l;
        for (auto &kv : a.stretch) {
            const int32_t l = kv.first;
            const float dom = (float)kv.second;
            float denom = 0.0f;
            for (int32_t nc : a.num_cpgs) if (nc >= l) denom += (float)(nc - l + 1);
            mhl += ((float)l * dom) / denom;
        }
        mhl /= l_sum;
        return mhl;
    };
    struct Out { float mhl; uint32_t cov; };
    std::unordered_map<Pos, Assoc, PosHash> cpg2reads;
    std::map<Pos, Out> result;
    auto finalize = [&](const Pos &cpg, const Assoc &a) {
        if ((uint32_t)a.num_cpgs.size() >= min_depth)  // mhl.rs:165 / 202
            result[cpg] = Out{compute_mhl(a), (uint32_t)a.num_cpgs.size()};
    };
    for (const Read &br : rd->reads) {
        if (!br.cpgs.empty()) {  // mhl.rs:162-173: flush BEFORE the filters, strict '<'
            const Pos first = br.cpgs[0].abspos;
            retain(cpg2reads, [&](const Pos &cpg, const Assoc &a) {
                if (cpg < first) { finalize(cpg, a); return false; }
                return true;
            });
        }
        if (br.mapq < min_qual) continue;         // mhl.rs:176
        if (br.cpgs.size() < min_cpgs) continue;  // mhl.rs:181
        for (const CpG &c : br.cpgs) {            // mhl.rs:185-192
            Assoc &a = cpg2reads[c.abspos];
            const size_t n = br.cpgs.size();      // add_num_cpgs mhl.rs:75-80
            a.num_cpgs.push_back((int32_t)n);
            if (n >= a.max_num_cpgs) a.max_num_cpgs = n;
            for (auto &kv : stretch_info(br)) a.stretch[kv.first] += kv.second;  // mhl.rs:36-41
        }
    }
    for (auto &kv : cpg2reads) finalize(kv.first, kv.second);  // mhl.rs:201-205
    auto *res = new orc_result;
    res->k = 1; res->m = 1;
    for (auto &kv : result) {
        res->tid.push_back(kv.first.tid); res->pos.push_back(kv.first.pos);
        res->val.push_back(kv.second.mhl); res->cnt.push_back(kv.second.cov);
    }
    return res;
}

// ---- me.rs:90-132 / pm.rs:85-128 ----------------------------------------------------
orc_result_t *orc_quartets(const orc_reads_t *rd, uint32_t min_depth, uint8_t min_qual, int which) {
    struct Q { Pos p[4]; bool operator<(const Q &o) const {
        for (int i = 0; i < 4; ++i) if (!(p[i] == o.p[i])) return p[i] < o.p[i];
        return false; } };
    struct Stat { uint32_t c[16] = {0}; };
    std::map<Q, Stat> q2s;  // HashMap in the reference (unordered output)
    for (const Read &br : rd->reads) {
        if (br.mapq < min_qual) continue;  // me.rs:115 / pm.rs:110
        const size_t n = br.cpgs.size();
        if (n < 4) continue;               // readutil.rs:101
        for (size_t i = 0; i < n - 3; ++i) {  // readutil.rs:105-129
            Q q{{br.cpgs[i].abspos, br.cpgs[i + 1].abspos, br.cpgs[i + 2].abspos, br.cpgs[i + 3].abspos}};
            int p = 0;
            if (br.cpgs[i].methylated) p += 8;
            if (br.cpgs[i + 1].methylated) p += 4;
            if (br.cpgs[i + 2].methylated) p += 2;
            if (br.cpgs[i + 3].methylated) p += 1;
            q2s[q].c[p] += 1;  // me.rs:121-125
        }
    }
    auto *res = new orc_result;
    res->k = 4; res->m = 16;
    for (auto &kv : q2s) {
        uint32_t total = 0;
        for (int i = 0; i < 16; ++i) total += kv.second.c[i];
        if (total < min_depth) continue;  // me.rs:82 / pm.rs:77
        float v;
        if (which == 0) {  // me.rs:42-55
            float me = 0.0f;
            for (int i = 0; i < 16; ++i) {
                const float p = (float)kv.second.c[i] / (float)total;
                if (kv.second.c[i] > 0) me += p * log2f(p);
            }
            me *= -0.25f;
            v = me;
        } else {  // pm.rs:42-51
            float pm = 1.0f;
            for (int i = 0; i < 16; ++i)
                pm -= ((float)kv.second.c[i] / (float)total) * ((float)kv.second.c[i] / (float)total);
            v = pm;
        }
        res->tid.push_back(kv.first.p[0].tid);
        for (int i = 0; i < 4; ++i) res->pos.push_back(kv.first.p[i].pos);
        res->val.push_back(v);
        for (int i = 0; i < 16; ++i) res->cnt.push_back(kv.second.c[i]);
    }
    return res;
}

// ---- fdrp.rs / qfdrp.rs --------------------------------------------------------------
int32_t orc_sample_j(uint64_t seed, int32_t tid, int32_t pos, int32_t num_total_read) {
    // splitmix64 over (seed, tid, pos, num_total_read) -> uniform in 1..=num_total_read.
    uint64_t z = seed ^ (((uint64_t)(uint32_t)tid << 32) | (uint32_t)pos);
    z += 0x9e3779b97f4a7c15ULL * (uint64_t)(uint32_t)num_total_read;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z = z ^ (z >> 31);
    return (int32_t)(z % (uint64_t)(uint32_t)num_total_read) + 1;
}

orc_result_t *orc_fdrp(const orc_reads_t *rd, uint8_t min_qual, uint64_t min_depth,
                       uint64_t max_depth, int32_t min_overlap, uint64_t seed, int which) {
    constexpr int32_t MAX_READ_LEN = 201;           // fdrp.rs:10
    constexpr int W = MAX_READ_LEN * 2 + 1;         // 403
    struct Arr { uint8_t b[W]; };
    struct Assoc {  // fdrp.rs:12-26
        std::vector<Arr> reads;
        int32_t num_total_read = 0, num_sampled_read = 0;
    };
    bool panicked = false;   // fdrp.rs:70-72: new_read[relative_pos] with relative_pos outside 0..=402 (a reverse-strand read whose first call
                             // sits at start - 1, added to its own site 202 bp further) is an index panic in the reference
    auto add_read = [&](Assoc &a, const Pos &site, const Read &br) {  // fdrp.rs:51-95
        Arr nr; memset(nr.b, 0, W);
        const int32_t s = MAX_READ_LEN + (br.start_pos - site.pos);
        const int32_t e = MAX_READ_LEN + (br.end_pos - site.pos);
        if (s < 0) return;                    // fdrp.rs:58
        if (e > MAX_READ_LEN * 2) return;     // fdrp.rs:61
        for (int32_t p = s; p < e + 1; ++p) nr.b[p] |= 1;
        for (const CpG &c : br.cpgs) {
            const int64_t rp = (int64_t)MAX_READ_LEN + ((int64_t)c.abspos.pos - site.pos);
            if (rp < 0 || rp >= W) { panicked = true; return; }
            nr.b[rp] |= 2;
            if (c.methylated) nr.b[rp] |= 4;
        }
        if (a.num_total_read < (int32_t)max_depth) {  // fdrp.rs:81-85
            a.num_sampled_read += 1; a.num_total_read += 1;
            a.reads.push_back(nr);
        } else {                                       // fdrp.rs:87-94
            a.num_total_read += 1;
            const int32_t j = orc_sample_j(seed, site.tid, site.pos, a.num_total_read);
            if (j <= (int32_t)max_depth) a.reads[j - 1] = nr;
        }
    };
    auto overlap_bases = [&](const Arr &x, const Arr &y) {  // fdrp.rs:97-107
        int32_t n = 0;
        for (int p = 0; p < W; ++p) n += (x.b[p] & y.b[p]) & 1;
        return n;
    };
    auto overlap_cpgs = [&](const Arr &x, const Arr &y) {   // qfdrp.rs:109-119
        int32_t n = 0;
        for (int p = 0; p < W; ++p) n += ((x.b[p] >> 1) & (y.b[p] >> 1)) & 1;
        return n;
    };
    auto hamming = [&](const Arr &x, const Arr &y) {        // qfdrp.rs:121-135 (fdrp.rs:109-122)
        float d = 0.0f;
        for (int p = 0; p < W; ++p)
            if (((x.b[p] & y.b[p]) & 3) == 3 && (((x.b[p] ^ y.b[p]) & 4) >> 2) == 1) d += 1.0f;
        return d;
    };
    auto compute = [&](const Assoc &a) -> float {  // fdrp.rs:124-145 / qfdrp.rs:137-157
        const size_t n = (size_t)a.num_sampled_read;
        float acc = 0.0f;
        for (size_t i = 0; i < n; ++i)
            for (size_t j = i + 1; j < n; ++j) {  // combinations(2): lexicographic
                const int32_t nb = overlap_bases(a.reads[i], a.reads[j]);
                if (which == 0) {
                    if (nb < min_overlap) continue;
                    if (hamming(a.reads[i], a.reads[j]) > 0.0f) acc += 1.0f;  // is_discordant
                } else {
                    const int32_t ncpg = overlap_cpgs(a.reads[i], a.reads[j]);
                    if (nb < min_overlap) continue;
                    acc += hamming(a.reads[i], a.reads[j]) / (float)ncpg;
                }
            }
        // (num_reads * (num_reads - 1)) as f32 / 2.0 ; usize arithmetic wraps in release
        const size_t prod = n * (n - 1);
        acc /= (float)prod / 2.0f;
        return acc;
    };
    struct Out { float v; uint32_t n; };
    std::map<Pos, Assoc> cpg2reads;  // BTreeMap fdrp.rs:194
    std::map<Pos, Out> result;
    auto finalize = [&](const Pos &cpg, const Assoc &a) {
        if ((size_t)a.num_sampled_read >= min_depth)  // fdrp.rs:215 / 240
            result[cpg] = Out{compute(a), (uint32_t)a.num_sampled_read};
    };
    for (const Read &br : rd->reads) {
        if (br.mapq < min_qual) continue;  // fdrp.rs:205
        if (br.cpgs.empty()) continue;     // fdrp.rs:208
        const Pos first = br.cpgs[0].abspos;
        for (auto it = cpg2reads.begin(); it != cpg2reads.end();) {  // fdrp.rs:212-223
            if (it->first < first) { finalize(it->first, it->second); it = cpg2reads.erase(it); }
            else ++it;
        }
        for (const CpG &c : br.cpgs) add_read(cpg2reads[c.abspos], c.abspos, br);  // fdrp.rs:225-231
        if (panicked) return nullptr;      // the caller reports the reference's panic
    }
    for (auto &kv : cpg2reads) finalize(kv.first, kv.second);  // fdrp.rs:239-243
    auto *res = new orc_result;
    res->k = 1; res->m = 1;
    for (auto &kv : result) {
        res->tid.push_back(kv.first.tid); res->pos.push_back(kv.first.pos);
        res->val.push_back(kv.second.v); res->cnt.push_back(kv.second.n);
    }
    return res;
}

int64_t orc_result_n(const orc_result_t *r) { return (int64_t)r->tid.size(); }
int orc_result_k(const orc_result_t *r) { return r->k; }
int orc_result_m(const orc_result_t *r) { return r->m; }
const int32_t *orc_result_tid(const orc_result_t *r) { return r->tid.data(); }
const int32_t *orc_result_pos(const orc_result_t *r) { return r->pos.data(); }
const float *orc_result_val(const orc_result_t *r) { return r->val.data(); }
const uint32_t *orc_result_cnt(const orc_result_t *r) { return r->cnt.data(); }
void orc_result_free(orc_result_t *r) { delete r; }

// Rust `impl Display for f32`: shortest digits that round-trip, positional notation.
int orc_format_f32(float v, char *buf) {
    if (std::isnan(v)) return sprintf(buf, "NaN");
    if (std::isinf(v)) return sprintf(buf, v < 0 ? "-inf" : "inf");
    // find the shortest precision whose scientific form parses back to v
    char sci[64];
    int prec = 0;
    for (; prec < 9; ++prec) {
        snprintf(sci, sizeof sci, "%.*e", prec, (double)v);
        if (strtof(sci, nullptr) == v) break;
    }
    // sci = d.ddddde[+-]xx  -> digits + decimal exponent
    char digits[32]; int nd = 0;
    const char *p = sci; bool neg = false;
    if (*p == '-') { neg = true; ++p; }
    for (; *p && *p != 'e'; ++p) if (*p != '.') digits[nd++] = *p;
    int exp10 = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd;  // strip trailing zeros
    int o = 0;
    if (neg) buf[o++] = '-';
    if (exp10 >= 0) {
        for (int i = 0; i <= exp10; ++i) buf[o++] = i < nd ? digits[i] : '0';
        if (nd > exp10 + 1) { buf[o++] = '.'; for (int i = exp10 + 1; i < nd; ++i) buf[o++] = digits[i]; }
    } else {
        buf[o++] = '0'; buf[o++] = '.';
        for (int i = 0; i < -exp10 - 1; ++i) buf[o++] = '0';
        for (int i = 0; i < nd; ++i) buf[o++] = digits[i];
    }
    buf[o] = 0;
    return o;
}

}  // extern "C"
