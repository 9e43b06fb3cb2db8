#!/usr/bin/env python3
"""Second, independent pin for the branches no reference fixture reaches.

A line-by-line Python transliteration of the reference's hot path, WRITTEN FROM THE RUST SOURCES under /root/reference/src
(file:line cited at every function) and NOT from oracle/metheor_oracle.cpp: dict + "retain" for the HashMap / BTreeMap state
machines, numpy.float32 for every f32 expression, Python ints wrapped by hand where Rust's release build wraps (i32 counters,
usize products).  It imports nothing from oracle/ or metheor_amd/.  Run in the build container only:

    python tools/gen_golden_unpinned.py            # writes tests/golden/unpinned_cases.json.gz

tests/test_unpinned_golden.py checks the C++ oracle against the file; tests/test_gpu_unpinned.py (-m gpu) checks the device path
(BAM -> device decode -> kernels -> TSV, through the `metheor` CLI) against it.

What is NOT in /root/reference and is restated from memory of the pinned dependency, rust-htslib 0.50.0 (Cargo.lock:939-941),
bam/ext.rs `BamRecordExtensions::reference_positions_full` = aligned_pairs_full() filtered to the entries that have a query
position, mapped to their reference position: one Option<i64> per QUERY base; M / = / X advance both and yield Some(ref); I and S
yield None per base; D and N advance the reference only (nothing yielded: no query base); H yields nothing; P panics
("Padding (Cigar::Pad) is not supported.") -- that last one is recalled, not verifiable here, and is not part of the cases.

CHECKLIST for a maintainer with `cargo` (rust-htslib 0.50.0, hts-sys 2.2.0): everything below is RECALLED, not read -- the crate is not
under /root/reference.  Each line is one assumption the 52 record sets depend on; the Rust one-liner next to it checks it.
  1. `Record::reference_positions_full()` yields exactly one item per query base (seq_len items; `H` and `P` add none).
  2. For `M`, `=`, `X` the item is Some(reference position), 0-based, ascending by one per base.
  3. For `I` and `S` the item is None (the base exists in the query, has no reference position).
  4. `D` and `N` yield no item and advance the reference position by their length.
  5. `Record::pos()` is the 0-based leftmost aligned position; the first non-None item equals it (readutil.rs:28-33 takes first / last
     non-None as start_pos / end_pos -- so a read that begins with `S` or `I` starts at pos(), not before it).
  6. `Record::flags()` is the raw u16 FLAG; the reference compares it with the literals 0, 99, 147 (readutil.rs:332-340), so 16, 83, 163
     and every flag with a secondary / duplicate / QC bit take the `abspos - 1` arm -- nothing here depends on rust-htslib for that.
  7. `Record::aux(b"XM")` returns the XM:Z string with one character per QUERY base (Bismark's convention), indexed by the same
     query index as reference_positions_full (readutil.rs:326-330).
  8. `Record::mapq()` is the raw u8; `Record::tid()` the i32 reference id, -1 for unmapped.
  9. `HeaderView::tid2name` / `tid` map ids and names in @SQ order (bamutil.rs:17-25).
     check: `for (i, p) in rec.reference_positions_full().enumerate() { println!("{i} {p:?}") }` on a record with CIGAR 3S5M2I4M3D6M2N5M.
"""
import json
import os
import sys

import numpy as np

f32 = np.float32


class Panic(Exception):
    pass


# ---------------------------------------------------------------------------------------------------------------------------
# records: dict(tid, pos, flag, mapq, cigar=[(op_char, len)], xm=str)
# ---------------------------------------------------------------------------------------------------------------------------
def reference_positions_full(rec):
    out = []
    g = rec["pos"]
    for op, n in rec["cigar"]:
        if op in "M=X":
            for _ in range(n):
                out.append(g)
                g += 1
        elif op in "IS":
            out.extend([None] * n)
        elif op in "DN":
            g += n
        elif op == "H":
            pass
        else:
            raise Panic("Padding (Cigar::Pad) is not supported.")
    return out


def bismark_read(rec):
    """readutil.rs:24-53 (BismarkRead::new) and 323-345 (get_cpgs)"""
    start_pos, end_pos = -1, -1                                     # :25-26
    for abspos in reference_positions_full(rec):                    # :28 .flatten() skips None
        if abspos is None:
            continue
        if start_pos == -1:                                         # :29
            start_pos = abspos
        end_pos = abspos                                            # :32
    if rec["xm"] is None:
        raise Panic("Error reading XM tag in BAM record. Make sure the reads are aligned using Bismark!")   # :46,50
    cpgs = []
    fwd = rec["flag"] in (0, 99, 147)                               # :332 exactly these three values
    for relpos, (abspos, c) in enumerate(zip(reference_positions_full(rec), rec["xm"])):     # :326 zip stops at the shorter
        if c != "z" and c != "Z":                                   # :327
            continue
        if abspos is not None:                                      # :331
            p = abspos if fwd else abspos - 1                       # :334 / :338
            cpgs.append(dict(relpos=relpos, abspos=(rec["tid"], p), methylated=(c == "Z")))   # CpG::new :254-260
    return dict(start_pos=start_pos, end_pos=end_pos, cpgs=cpgs)


def is_before(a, b, distance):
    """readutil.rs:304-310"""
    if a[0] > b[0]:
        return False
    if a[0] < b[0]:
        return True
    return a[1] + distance < b[1]


def concordance_state(br):
    """readutil.rs:134-145 -> True when Discordant"""
    init = br["cpgs"][0]["methylated"]
    res = False
    for cpg in br["cpgs"]:
        if cpg["methylated"] != init:
            res = True
    return res


def stretch_info(br):
    """readutil.rs:147-164"""
    info = {}
    cur = 0
    for cpg in br["cpgs"]:
        if cpg["methylated"]:
            cur += 1
            for l in range(1, cur + 1):
                info[l] = info.get(l, 0) + 1
        else:
            cur = 0
    return info


def pairwise(br, min_distance, max_distance):
    """readutil.rs:166-224"""
    anchors = []
    pairs = []
    min_anchor_pos = -1
    n_c = n_d = 0
    for cpg in br["cpgs"]:
        if min_anchor_pos != -1:                                                            # :183
            while (cpg["relpos"] - min_anchor_pos > max_distance) and anchors:              # :184
                anchors.pop(0)                                                              # :185
                if anchors:
                    min_anchor_pos = anchors[0]["relpos"]                                   # :188
                else:
                    min_anchor_pos = -1                                                     # :190
        for anchor in anchors:                                                              # :195
            if cpg["relpos"] - anchor["relpos"] < min_distance:                             # :196
                continue
            if anchor["methylated"] == cpg["methylated"]:
                n_c += 1
                pairs.append((anchor["abspos"], cpg["abspos"], False))
            else:
                n_d += 1
                pairs.append((anchor["abspos"], cpg["abspos"], True))
        if min_anchor_pos == -1:                                                            # :217
            min_anchor_pos = cpg["relpos"]
        anchors.append(cpg)                                                                 # :220
    return n_c, n_d, pairs


def quartets_and_patterns(br):
    """readutil.rs:97-132"""
    c = br["cpgs"]
    out = []
    if len(c) < 4:
        return out
    for i in range(len(c) - 3):
        q = (c[i]["abspos"], c[i + 1]["abspos"], c[i + 2]["abspos"], c[i + 3]["abspos"])
        p = 8 * c[i]["methylated"] + 4 * c[i + 1]["methylated"] + 2 * c[i + 2]["methylated"] + 1 * c[i + 3]["methylated"]
        out.append((q, int(p)))
    return out


def filter_isin(br, target):
    """readutil.rs:87-95"""
    if target is not None:
        br["cpgs"] = [c for c in br["cpgs"] if c["abspos"] in target]


# ---------------------------------------------------------------------------------------------------------------------------
def pdr(records, min_depth, min_cpgs, min_qual, target=None):
    """pdr.rs:119-212; value = (pdr f32, n_concordant, n_discordant), pdr.rs:47-49"""
    cpg2reads = {}
    result = {}

    def compute(nc, nd):
        return f32(nd) / (f32(nc) + f32(nd))

    for r in records:
        br = bismark_read(r)                                        # :140
        filter_isin(br, target)
        if len(br["cpgs"]) < min_cpgs:                              # :147
            continue
        if r["mapq"] < min_qual:                                    # :150
            continue
        positions = [c["abspos"] for c in br["cpgs"]]
        if not positions:                                           # :155
            continue
        first = positions[0]
        for cpg in list(cpg2reads.keys()):                          # :160 retain
            if is_before(cpg, first, 150):                          # :162
                nc, nd = cpg2reads[cpg]
                if nc + nd >= min_depth:                            # :163
                    result[cpg] = (compute(nc, nd), nc, nd)         # :164 BTreeMap insert overwrites
                del cpg2reads[cpg]
        disc = concordance_state(br)                                # :185 (per CpG in the reference; same value each time)
        for p in positions:                                         # :180
            e = cpg2reads.setdefault(p, [0, 0])
            if disc:
                e[1] += 1
            else:
                e[0] += 1
    for cpg, (nc, nd) in cpg2reads.items():                         # :199
        if nc + nd >= min_depth:
            result[cpg] = (compute(nc, nd), nc, nd)
    return dict(sorted(result.items()))


def wrap_i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >= (1 << 31) else x


def lpmd_from_counts(n_c, n_d):
    """lpmd.rs:51-55 on i32 fields that wrap in a release build (lpmd.rs:11-12, 45-49)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(wrap_i32(n_d)) / f32(wrap_i32(wrap_i32(n_c) + wrap_i32(n_d)))


def lpmd(records, min_distance, max_distance, min_qual, target=None):
    """lpmd.rs:154-202; the pairs table of lpmd.rs:89-122"""
    n_read = n_valid = n_c = n_d = 0
    pc, pd = {}, {}
    for r in records:
        n_read = wrap_i32(n_read + 1)                               # :176
        if r["mapq"] < min_qual:                                    # :177 -- BEFORE BismarkRead::new: no XM needed below the threshold
            continue
        br = bismark_read(r)                                        # :181
        filter_isin(br, target)
        c, d, pairs = pairwise(br, min_distance, max_distance)      # :186
        n_valid = wrap_i32(n_valid + 1)
        n_c = wrap_i32(n_c + c)
        n_d = wrap_i32(n_d + d)
        for a, b, disc in pairs:                                    # :192, add_pair_concordance :70-87
            k = (a, b)
            pc.setdefault(k, 0)
            pd.setdefault(k, 0)
            if disc:
                pd[k] += 1
            else:
                pc[k] += 1
    table = []
    for k in sorted(pc.keys()):                                     # :94 pairs.sort()
        with np.errstate(divide="ignore", invalid="ignore"):
            table.append((k, f32(pd[k]) / (f32(pc[k]) + f32(pd[k])), pc[k], pd[k]))      # :111
    return dict(n_read=n_read, n_valid_read=n_valid, n_concordant=n_c, n_discordant=n_d, lpmd=lpmd_from_counts(n_c, n_d), pairs=table)


def compute_mhl(stretch, num_cpgs, max_num_cpgs):
    """mhl.rs:43-73.  The reference iterates a HashMap (random order); ascending l here -- the f32 sum may differ in the last
    place between orders, which is why the bar for MHL is 1e-6 and not bit equality."""
    mhl = f32(0.0)
    l_sum = f32(0.0)
    for l in range(1, max_num_cpgs + 1):                            # :46
        l_sum = f32(l_sum + f32(l))
    for l in sorted(stretch.keys()):                                # :50
        dom = f32(stretch[l])
        denom = f32(0.0)
        for n in num_cpgs:                                          # :54
            if n >= l:
                denom = f32(denom + f32(n - l + 1))
        assert denom > 0.0                                          # :60
        mhl = f32(mhl + f32(f32(f32(l) * dom) / denom))             # :68
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(mhl / l_sum)                                     # :71


def mhl(records, min_depth, min_cpgs, min_qual, target=None):
    """mhl.rs:135-208"""
    cpg2reads = {}
    result = {}
    for r in records:
        br = bismark_read(r)                                        # :156
        filter_isin(br, target)
        if br["cpgs"]:                                              # :162 flush BEFORE the filters, strict <
            first = br["cpgs"][0]["abspos"]
            for cpg in list(cpg2reads.keys()):
                if cpg < first:                                     # :164 (tid, pos) order
                    st = cpg2reads[cpg]
                    if len(st["num_cpgs"]) >= min_depth:            # :165
                        result[cpg] = compute_mhl(st["stretch"], st["num_cpgs"], st["max"])
                    del cpg2reads[cpg]
        if r["mapq"] < min_qual:                                    # :176
            continue
        if len(br["cpgs"]) < min_cpgs:                              # :181
            continue
        si = stretch_info(br)
        n = len(br["cpgs"])
        for c in br["cpgs"]:                                        # :185
            st = cpg2reads.setdefault(c["abspos"], dict(stretch={}, num_cpgs=[], max=0))
            st["num_cpgs"].append(n)                                # add_num_cpgs :75-80
            if n >= st["max"]:
                st["max"] = n
            for l, cnt in si.items():                               # add_stretch_info :36-41
                st["stretch"][l] = st["stretch"].get(l, 0) + cnt
    for cpg, st in cpg2reads.items():                               # :201
        if len(st["num_cpgs"]) >= min_depth:
            result[cpg] = compute_mhl(st["stretch"], st["num_cpgs"], st["max"])
    return dict(sorted(result.items()))


def compute_me(counts):
    """me.rs:42-55"""
    me = f32(0.0)
    total = sum(counts)
    for c in counts:
        with np.errstate(divide="ignore", invalid="ignore"):
            p = f32(f32(c) / f32(total))
        if c > 0:
            me = f32(me + f32(p * f32(np.log2(p))))
    return f32(me * f32(-0.25))


def compute_pm(counts):
    """pm.rs:42-51"""
    total = sum(counts)
    pm = f32(1.0)
    for c in counts:
        with np.errstate(divide="ignore", invalid="ignore"):
            a = f32(f32(c) / f32(total))
        pm = f32(pm - f32(a * a))
    return pm


def me_pm(records, min_qual, target=None):
    """me.rs:90-132 / pm.rs:85-128 (the same loop); rows keyed by the quartet, depth filter applied by the writer (me.rs:82)"""
    q2 = {}
    for r in records:
        br = bismark_read(r)
        filter_isin(br, target)
        if r["mapq"] < min_qual:                                    # me.rs:114
            continue
        for q, p in quartets_and_patterns(br):
            q2.setdefault(q, [0] * 16)[p] += 1
    return {q: (cnt, compute_me(cnt), compute_pm(cnt)) for q, cnt in sorted(q2.items())}


MAX_READ_LEN = 201                                                  # fdrp.rs:10


class Assoc:
    """fdrp.rs:12-41 / qfdrp.rs the same"""

    def __init__(self, pos, max_depth):
        self.pos = pos
        self.reads = []
        self.num_total_read = 0
        self.num_sampled_read = 0
        self.max_depth = max_depth

    def add_read(self, br):
        """fdrp.rs:51-95"""
        new_read = np.zeros(MAX_READ_LEN * 2 + 1, np.uint8)
        s = MAX_READ_LEN + (br["start_pos"] - self.pos[1])
        e = MAX_READ_LEN + (br["end_pos"] - self.pos[1])
        if s < 0:                                                   # :58
            return
        if e > MAX_READ_LEN * 2:                                    # :61
            return
        for rp in range(s, e + 1):                                  # :65
            new_read[rp] |= 1
        for cpg in br["cpgs"]:                                      # :69
            rp = MAX_READ_LEN + (cpg["abspos"][1] - self.pos[1])    # get_relative_position :43-45, `as usize`
            if rp < 0 or rp >= len(new_read):
                raise Panic("index out of bounds")
            new_read[rp] |= 2
            if cpg["methylated"]:
                new_read[rp] |= 4
        if self.num_total_read < self.max_depth:                    # :81
            self.num_sampled_read += 1
            self.num_total_read += 1
            self.reads.append(new_read)
        else:
            raise Panic("reservoir branch (fdrp.rs:88-94, OS-seeded RNG): not reproducible, kept out of the cases")

    def overlap_bases(self, i, j):                                  # :97-107
        return int(((self.reads[i] & self.reads[j]) & 1).sum())

    def overlap_cpgs(self, i, j):                                   # qfdrp.rs:109-119
        return int((((self.reads[i] >> 1) & (self.reads[j] >> 1)) & 1).sum())

    def hamming(self, i, j):                                        # qfdrp.rs:121-135 (f32 count; exact below 2^24)
        r1, r2 = self.reads[i], self.reads[j]
        return int(((((r1 & r2) & 3) == 3) & ((((r1 ^ r2) & 4) >> 2) == 1)).sum())

    def denominator(self):
        n = self.num_sampled_read
        prod = (n * ((n - 1) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF            # usize, wrapping in a release build (:143)
        return f32(f32(prod) / f32(2.0))

    def compute_fdrp(self, min_overlap):
        """fdrp.rs:124-145"""
        n = self.num_sampled_read
        v = f32(0.0)
        for i in range(n):
            for j in range(i + 1, n):                               # combinations(2): lexicographic
                if self.overlap_bases(i, j) < min_overlap:
                    continue
                if self.hamming(i, j) > 0:                          # is_discordant :109-122
                    v = f32(v + f32(1.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            return f32(v / self.denominator())

    def compute_qfdrp(self, min_overlap):
        """qfdrp.rs:137-157"""
        n = self.num_sampled_read
        v = f32(0.0)
        for i in range(n):
            for j in range(i + 1, n):
                ob = self.overlap_bases(i, j)
                oc = self.overlap_cpgs(i, j)
                if ob < min_overlap:
                    continue
                with np.errstate(divide="ignore", invalid="ignore"):
                    v = f32(v + f32(f32(self.hamming(i, j)) / f32(oc)))
        with np.errstate(divide="ignore", invalid="ignore"):
            return f32(v / self.denominator())


def fdrp_qfdrp(records, min_qual, min_depth, max_depth, min_overlap, target=None):
    """fdrp.rs:176-246 and qfdrp.rs:188-258 (identical drivers): rows (fdrp, qfdrp, stored reads)"""
    cpg2reads = {}
    result = {}

    def fin(cpg, a):
        if a.num_sampled_read >= min_depth:                         # :216
            result[cpg] = (a.compute_fdrp(min_overlap), a.compute_qfdrp(min_overlap), a.num_sampled_read)

    for r in records:
        br = bismark_read(r)
        filter_isin(br, target)
        if r["mapq"] < min_qual:                                    # :206
            continue
        if not br["cpgs"]:                                          # :209
            continue
        first = br["cpgs"][0]["abspos"]
        for cpg in sorted(cpg2reads.keys()):                        # :213 retain on a BTreeMap, strict <
            if cpg < first:
                fin(cpg, cpg2reads[cpg])
                del cpg2reads[cpg]
        for c in br["cpgs"]:                                        # :227
            a = cpg2reads.get(c["abspos"])
            if a is None:
                a = cpg2reads[c["abspos"]] = Assoc(c["abspos"], max_depth)
            a.add_read(br)
    for cpg in sorted(cpg2reads.keys()):                            # :239
        fin(cpg, cpg2reads[cpg])
    return dict(sorted(result.items()))


# ---------------------------------------------------------------------------------------------------------------------------
# cases
# ---------------------------------------------------------------------------------------------------------------------------
def rec(tid, pos, cigar, xm, flag=0, mapq=40):
    ops = []
    num = ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            ops.append((ch, int(num)))
            num = ""
    return dict(tid=tid, pos=pos, flag=flag, mapq=mapq, cigar=ops, xm=xm)


def qlen(r):
    return sum(n for op, n in r["cigar"] if op in "M=XIS")


def xm_from(r, sites, levels, rng, drop=0.0):
    """a Bismark-like XM string for record r: 'z'/'Z' where the aligned reference base is the C (forward rule) or the G
    (reverse rule: the reference reports abspos - 1) of a site in `sites`, '.' elsewhere; calls on I / S bases are thrown in
    on purpose (they must be ignored, readutil.rs:331)"""
    fwd = r["flag"] in (0, 99, 147)
    out = []
    for ap in reference_positions_full(r):
        if ap is None:
            out.append("Z" if rng.random() < 0.15 else ".")       # a call on an inserted / clipped base
            continue
        site = ap if fwd else ap - 1
        if site in sites and rng.random() >= drop:
            out.append("Z" if rng.random() < levels[site] else "z")
        else:
            out.append("h" if rng.random() < 0.05 else ".")
    return "".join(out)


def random_case(seed, n_reads, contigs, span_choices, density, cigar_mix=True, flags=(0, 16), depth_cap=None):
    rng = np.random.default_rng(seed)
    records = []
    for tid, length in enumerate(contigs):
        pos = 1
        sites = set()
        while pos < length:
            pos += 2 + int(rng.geometric(density))
            sites.add(pos)
        levels = {s: (0.9 if rng.random() < 0.6 else 0.15) for s in sites}
        starts = np.sort(rng.integers(2, max(3, length - 40), size=n_reads))
        for s in starts:
            span = int(rng.choice(span_choices))
            shape = int(rng.integers(0, 8)) if cigar_mix else 0
            a = span // 2
            b = span - a
            if shape == 1:
                cg = "%dM%dI%dM" % (a, int(rng.integers(1, 4)), b)
            elif shape == 2:
                cg = "%dS%dM" % (int(rng.integers(1, 6)), span)
            elif shape == 3:
                cg = "%dM%dD%dM" % (a, int(rng.integers(1, 5)), b)
            elif shape == 4:
                cg = "%dH%d=%dX%dM%dS" % (2, a, 1, b - 1 if b > 1 else 1, int(rng.integers(1, 4)))
            elif shape == 5:
                cg = "%dM%dN%dM" % (a, int(rng.integers(20, 60)), b)
            else:
                cg = "%dM" % span
            fl = int(rng.choice(flags))
            mq = 42 if rng.random() < 0.85 else int(rng.integers(0, 10))
            r = rec(tid, int(s), cg, None, fl, mq)
            r["xm"] = xm_from(r, sites, levels, rng, drop=0.1)
            if rng.random() < 0.05:
                r["xm"] = r["xm"][:max(1, len(r["xm"]) - int(rng.integers(1, 5)))]      # XM shorter than the query (zip stops)
            records.append(r)
    return records


def hand_cases():
    cs = []
    # -- decode rules, one read each ---------------------------------------------------------------------------------------
    cs.append(("cigar_ins_softclip_del", [
        rec(0, 100, "3S5M2I5M3D4M", "ZZZ" + "Z.z.." + "ZZ" + ".Z..z" + "z..Z", 0),
        rec(0, 100, "3S5M2I5M3D4M", "ZZZ" + "Z.z.." + "ZZ" + ".Z..z" + "z..Z", 16),
        rec(0, 100, "3S5M2I5M3D4M", "ZZZ" + "Z.z.." + "ZZ" + ".Z..z" + "z..Z", 99),
        rec(0, 100, "3S5M2I5M3D4M", "ZZZ" + "Z.z.." + "ZZ" + ".Z..z" + "z..Z", 83),
        rec(0, 101, "2H4=1X4M50N6M2S", "Z..z" + "Z" + "z..Z" + "Z.z..Z" + "zz", 147),
        rec(0, 101, "2H4=1X4M50N6M2S", "Z..z" + "Z" + "z..Z" + "Z.z..Z" + "zz", 163),
    ], dict(pdr=[dict(min_depth=0, min_cpgs=0, min_qual=0)], lpmd=[dict(min_distance=0, max_distance=500, min_qual=0), dict(min_distance=2, max_distance=16, min_qual=0)],
            mhl=[dict(min_depth=0, min_cpgs=1, min_qual=0)], quartet=[dict(min_qual=0)], fdrp=[dict(min_qual=0, min_depth=0, max_depth=40, min_overlap=4)])))
    # -- PDR flush margin 150 / re-open / overwrite (pdr.rs:160-177) -------------------------------------------------------------
    # Coordinate-sorted (ties in start are in file order): site 1000 gets 3 reads; a 201-bp read with the same start whose FIRST CpG
    # is at 1155 (> 1000 + 150) flushes it (1010 stays: 1010 + 150 >= 1155); the next read, same start, calls 1000 again: a new
    # segment with one read, which overwrites the first at the final flush when min_depth <= 1 and does not when min_depth is 2 or 3.
    long_xm = "Z" + "." * 199 + "z"
    cs.append(("pdr_flush_reopen_overwrite", [
        rec(0, 1000, "30M", "Z" + "." * 9 + "z" + "." * 19),
        rec(0, 1000, "30M", "Z" + "." * 9 + "Z" + "." * 19),
        rec(0, 1000, "30M", "z" + "." * 9 + "z" + "." * 19),
        rec(0, 1000, "201M", "." * 155 + "Z" + "." * 44 + "z"),     # calls 1155 and 1200
        rec(0, 1000, "201M", long_xm),                               # calls 1000 and 1200: re-opens 1000
        rec(0, 1000, "201M", long_xm, mapq=3),                       # fails mapq: neither flushes nor counts
        rec(0, 1100, "201M", "." * 100 + "z" + "." * 100),          # first CpG 1200: flushes 1000 (again) and 1010
    ], dict(pdr=[dict(min_depth=d, min_cpgs=c, min_qual=10) for d in (0, 1, 2, 3, 4) for c in (0, 2)])))
    # the same idea in an order a coordinate sort would not give (a read starting at 1000 AFTER one starting at 1100): the
    # reference streams whatever order the file has (pdr.rs:139)
    cs.append(("pdr_flush_reopen_unsorted", [
        rec(0, 1000, "30M", "Z" + "." * 9 + "z" + "." * 19),
        rec(0, 1000, "30M", "Z" + "." * 9 + "Z" + "." * 19),
        rec(0, 1000, "30M", "z" + "." * 9 + "z" + "." * 19),
        rec(0, 1000, "201M", long_xm),
        rec(0, 1100, "60M", "." * 51 + "Z" + "." * 8),             # first CpG 1151: flushes 1000 only
        rec(0, 1000, "201M", long_xm, mapq=3),
        rec(0, 1000, "201M", long_xm),                               # re-opens 1000
        rec(0, 1100, "201M", "." * 100 + "z" + "." * 100),
    ], dict(pdr=[dict(min_depth=d, min_cpgs=c, min_qual=10) for d in (0, 1, 2, 3, 4) for c in (0, 2)]), False))
    cs.append(("pdr_flush_margin_sorted", [
        rec(0, 500, "20M", "Z" + "." * 9 + "Z" + "." * 9),
        rec(0, 500, "20M", "Z" + "." * 9 + "z" + "." * 9),
        rec(0, 700, "20M", "Z" + "." * 19),                         # 700 > 510 + 150: flushes 500 and 510
        rec(0, 700, "20M", "z" + "." * 19),
        rec(0, 700, "150M", "z" + "." * 149),
        rec(0, 700, "201M", "Z" + "." * 199 + "Z"),
        rec(0, 860, "41M", "." * 40 + "Z"),                         # first CpG 900 > 700 + 150: flushes 700
        rec(0, 860, "41M", "." * 40 + "z"),
    ], dict(pdr=[dict(min_depth=d, min_cpgs=0, min_qual=10) for d in (0, 1, 2, 3)])))
    # -- MHL: strict <, flusher = any read with a CpG, also one that fails mapq / min_cpgs (mhl.rs:162-181) -----------------------------
    # every read starts at 100 (sorted; file order is the tie order)
    cs.append(("mhl_flusher_rules", [
        rec(0, 100, "20M", "ZZ.Z" + "." * 16),                      # sites 100 101 103 (not biology; the map does not care)
        rec(0, 100, "20M", "Zz.Z" + "." * 16),
        rec(0, 100, "20M", "ZZ.z" + "." * 16),
        rec(0, 100, "20M", ".Z" + "." * 18, mapq=0),                 # fails mapq, first CpG 101: flushes 100 only
        rec(0, 100, "20M", "ZZ.Z" + "." * 16),                      # re-opens 100
        rec(0, 100, "20M", "zZ.Z" + "." * 16),
        rec(0, 100, "20M", "..Z" + "." * 17),                       # one CpG (102): flushes 100 and 101; counts only when min_cpgs <= 1
        rec(0, 100, "20M", "...ZZZZZ" + "." * 12),                  # first CpG 103: flushes 102
    ], dict(mhl=[dict(min_depth=d, min_cpgs=c, min_qual=10) for d in (0, 1, 2, 3) for c in (1, 2, 3)])))
    cs.append(("mhl_flusher_rules_unsorted", [
        rec(0, 100, "20M", "ZZ.Z" + "." * 16),
        rec(0, 100, "20M", "Zz.Z" + "." * 16),
        rec(0, 100, "20M", "ZZ.z" + "." * 16),
        rec(0, 101, "20M", "Z" + "." * 19, mapq=0),
        rec(0, 100, "20M", "ZZ.Z" + "." * 16),
        rec(0, 100, "20M", "zZ.Z" + "." * 16),
        rec(0, 102, "20M", "." + "Z" + "." * 18),
        rec(0, 103, "20M", "ZZZZZ" + "." * 15),
    ], dict(mhl=[dict(min_depth=d, min_cpgs=c, min_qual=10) for d in (0, 1, 2, 3) for c in (1, 2, 3)]), False))
    # -- FDRP / qFDRP: strict <, flusher = mapq-passing reads with a CpG; window drop; NaN rows; call outside coverage --------------------
    a = "Z" + "." * 9 + "z" + "." * 9 + "Z" + "." * 19
    b = "z" + "." * 9 + "z" + "." * 9 + "Z" + "." * 19
    c = "Z" + "." * 9 + "Z" + "." * 9 + "z" + "." * 19
    cs.append(("fdrp_flush_reopen_window", [
        rec(0, 300, "40M", a), rec(0, 300, "40M", b), rec(0, 300, "40M", c), rec(0, 300, "40M", a, mapq=5),
        rec(0, 300, "40M", "." * 10 + "z" + "." * 9 + "Z" + "." * 19),      # first CpG 310: flushes 300
        rec(0, 300, "40M", b),                                      # re-opens 300
        rec(0, 300, "40M", c),
        rec(0, 305, "36M", "." * 5 + "Z" + "." * 9 + "Z" + "." * 20, flag=16),       # reverse: sites 309, 319
        rec(0, 311, "30M", "Z" + "." * 29, flag=16),                # reverse: call at 310 = start - 1 (outside its coverage)
        rec(0, 311, "30M", "z" + "." * 29, flag=16),
        rec(0, 320, "250M", "Z" + "." * 248 + "z"),                 # span 250: dropped from site 320's window (end > +201), kept at 569
        rec(0, 320, "30M", "z" + "." * 29),
        rec(0, 560, "30M", "." * 9 + "Z" + "." * 20),
    ], dict(fdrp=[dict(min_qual=10, min_depth=d, max_depth=40, min_overlap=o) for d in (0, 1, 2, 3) for o in (0, 10, 35)])))
    # -- two contigs, the flush across the tid boundary (readutil.rs:304-310: a smaller tid is always "before") ---------------------------
    cs.append(("two_contigs_flush", [
        rec(0, 50, "30M", "Z" + "." * 9 + "z" + "." * 9 + "Z" + "." * 4 + "Z" + "." * 4),
        rec(0, 50, "30M", "Z" + "." * 9 + "Z" + "." * 9 + "Z" + "." * 4 + "Z" + "." * 4),
        rec(0, 60, "30M", "z" + "." * 9 + "Z" + "." * 4 + "Z" + "." * 4 + "z" + "." * 9, flag=16),
        rec(1, 10, "30M", "Z" + "." * 9 + "z" + "." * 9 + "Z" + "." * 4 + "Z" + "." * 4),
        rec(1, 10, "30M", "z" + "." * 9 + "z" + "." * 9 + "Z" + "." * 4 + "z" + "." * 4),
        rec(1, 10, "30M", "z" + "." * 9 + "z" + "." * 9 + "Z" + "." * 4 + "z" + "." * 4, mapq=1),
    ], dict(pdr=[dict(min_depth=0, min_cpgs=0, min_qual=10), dict(min_depth=2, min_cpgs=4, min_qual=10)],
            lpmd=[dict(min_distance=2, max_distance=16, min_qual=10), dict(min_distance=5, max_distance=4, min_qual=10)],
            mhl=[dict(min_depth=0, min_cpgs=1, min_qual=10), dict(min_depth=2, min_cpgs=4, min_qual=10)],
            quartet=[dict(min_qual=10)], fdrp=[dict(min_qual=10, min_depth=0, max_depth=40, min_overlap=10)])))
    return cs


def enc_key(k):
    return [int(k[0]), int(k[1])]


def bits(v):
    return int(np.float32(v).view(np.uint32))


def run_case(name, records, params):
    out = dict(name=name, records=[dict(tid=r["tid"], pos=r["pos"], flag=r["flag"], mapq=r["mapq"],
                                        cigar="".join("%d%s" % (n, op) for op, n in r["cigar"]), xm=r["xm"]) for r in records],
               decode=[], expect={})
    for r in records:
        br = bismark_read(r)
        out["decode"].append(dict(start=br["start_pos"], end=br["end_pos"],
                                  cpgs=[[c["relpos"], c["abspos"][1], int(c["methylated"])] for c in br["cpgs"]]))
    ex = out["expect"]
    for p in params.get("pdr", []):
        res = pdr(records, **p)
        ex.setdefault("pdr", []).append(dict(params=p, rows=[[k[0], k[1], bits(v[0]), v[1], v[2]] for k, v in res.items()]))
    for p in params.get("lpmd", []):
        res = lpmd(records, **p)
        ex.setdefault("lpmd", []).append(dict(params=p, counts=[res["n_concordant"], res["n_discordant"], res["n_read"], res["n_valid_read"]],
                                               lpmd_bits=bits(res["lpmd"]),
                                               pairs=[[k[0][0], k[0][1], k[1][1], bits(v), c, d] for k, v, c, d in res["pairs"]]))
    for p in params.get("mhl", []):
        res = mhl(records, **p)
        ex.setdefault("mhl", []).append(dict(params=p, rows=[[k[0], k[1], bits(v)] for k, v in res.items()]))
    for p in params.get("quartet", []):
        res = me_pm(records, **p)
        ex.setdefault("quartet", []).append(dict(params=p, rows=[[q[0][0], q[0][1], q[1][1], q[2][1], q[3][1], cnt, bits(me), bits(pm)]
                                                                for q, (cnt, me, pm) in res.items()]))
    for p in params.get("fdrp", []):
        res = fdrp_qfdrp(records, **p)
        ex.setdefault("fdrp", []).append(dict(params=p, rows=[[k[0], k[1], bits(v[0]), bits(v[1]), v[2]] for k, v in res.items()]))
    return out


ALL = dict(pdr=[dict(min_depth=0, min_cpgs=0, min_qual=10), dict(min_depth=3, min_cpgs=2, min_qual=10)],
           lpmd=[dict(min_distance=2, max_distance=16, min_qual=10), dict(min_distance=0, max_distance=40, min_qual=0)],
           mhl=[dict(min_depth=0, min_cpgs=1, min_qual=10), dict(min_depth=3, min_cpgs=2, min_qual=10)],
           quartet=[dict(min_qual=10)],
           fdrp=[dict(min_qual=10, min_depth=0, max_depth=64, min_overlap=10), dict(min_qual=10, min_depth=3, max_depth=64, min_overlap=35)])


def main():
    cases = []
    for hc in hand_cases():
        c = run_case(hc[0], hc[1], hc[2])
        c["sorted"] = hc[3] if len(hc) > 3 else True
        cases.append(c)
    # seeded mixes: indels / clips / skips / =X / H, both strand rules and the paired flags, short XM, low mapq, two contigs,
    # long spans (re-opening under the 150-bp margin), dense CpGs
    specs = [
        (1, 60, [3000], [30, 40, 50], 0.05, True, (0, 16)),
        (2, 60, [3000], [30, 40, 50], 0.05, True, (99, 147, 83, 163)),
        (3, 80, [2000, 1500], [25, 36, 60], 0.08, True, (0, 16)),
        (4, 70, [4000], [120, 180, 200], 0.03, False, (0, 16)),           # spans > 150: PDR re-open territory; <= 201: FDRP window holds
        (5, 70, [4000], [150, 190, 201], 0.02, True, (0, 16, 99, 83)),
        (6, 120, [1200], [30, 45], 0.12, False, (0, 16)),                 # deep and dense
        (7, 50, [2500, 800, 1200], [40, 75], 0.06, True, (0, 16)),
        (8, 90, [1500], [20, 28], 0.2, True, (0,)),                       # very dense: many quartets, long stretches
        (9, 90, [6000], [100, 160, 230], 0.02, False, (0, 16)),           # spans beyond 201: FDRP window drops + NaN rows
        (10, 40, [2000], [60], 0.04, True, (16,)),
    ]
    prng = np.random.default_rng(77)
    for k in range(11, 45):                                               # 34 more mixes drawn from the same pools
        ncont = int(prng.choice([1, 1, 2, 3]))
        specs.append((k, int(prng.integers(25, 70)), [int(prng.integers(600, 3000)) for _ in range(ncont)],
                      [[25, 36, 50], [30, 45], [100, 151, 180], [150, 200, 201], [60, 75], [170, 210, 260]][int(prng.integers(0, 6))],
                      float(prng.choice([0.02, 0.04, 0.08, 0.15])), bool(prng.integers(0, 2)),
                      [(0, 16), (0,), (16,), (99, 147, 83, 163), (0, 16, 99, 83)][int(prng.integers(0, 5))]))
    for seed, n, contigs, spans, dens, mix, flags in specs:
        records = random_case(1000 + seed, n, contigs, spans, dens, mix, flags)
        params = dict(ALL)
        try:
            cases.append(run_case("seeded_%d" % seed, records, params))
        except Panic as e:                                          # fdrp.rs:69-71 index -1: leave FDRP out of that case, keep the rest
            params = {k: v for k, v in ALL.items() if k != "fdrp"}
            c = run_case("seeded_%d" % seed, records, params)
            c["fdrp_panics"] = str(e)
            cases.append(c)
        cases[-1]["sorted"] = True
    # lpmd.rs:11-12 / 51-55: i32 counters that wrap; the expression on the wrapped values
    table = []
    for nc, nd in [(48, 48), (0, 0), (0, 7), (2147483647, 1), (2147483648, 5), (3000000000, 3000000000), (4294967296 + 17, 4), (2147483000, 2147483000),
                   (1, 2147483647), (5000000000, 123456789), (4294967295, 1), (2147483647, 2147483647)]:
        table.append([nc, nd, bits(lpmd_from_counts(nc, nd))])
    out = dict(about="expected values from tools/gen_golden_unpinned.py: a Python transliteration of /root/reference/src written from the Rust, "
                     "independent of oracle/; floats are IEEE-754 binary32 bit patterns", cases=cases, lpmd_from_counts=table)
    import gzip
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "unpinned_cases.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:                  # (mtime 0: the same bytes on every run)
        f.write(json.dumps(out, separators=(",", ":")).encode())
    n_rec = sum(len(c["records"]) for c in cases)
    print("wrote %s: %d cases, %d records, %d bytes" % (path, len(cases), n_rec, os.path.getsize(path)))
    for c in cases:
        print("  %-28s %4d records  %s%s" % (c["name"], len(c["records"]), {k: [len(x.get("rows", x.get("pairs", []))) for x in v] for k, v in c["expect"].items()},
                                             "  FDRP PANICS" if "fdrp_panics" in c else ""))


if __name__ == "__main__":
    sys.exit(main())
