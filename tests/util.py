"""helpers shared by the parity tests (oracle side + device side)"""
import numpy as np

from metheor_amd import shard


def oracle_soa_from_contig(c):
    from metheor_amd import synth
    return synth.to_oracle_soa(c)


def contig_from_oracle_soa(soa, tid, length):
    """oracle Reads.soa() dict restricted to one tid -> contig dict the device Batch takes"""
    sel = np.nonzero(soa["tid"] == tid)[0]
    assert len(sel) == 0 or (np.diff(sel) == 1).all(), "tid must be contiguous"
    i0, i1 = (int(sel[0]), int(sel[-1]) + 1) if len(sel) else (0, 0)
    o0, o1 = int(soa["cpg_off"][i0]), int(soa["cpg_off"][i1])
    rel = soa["cpg_rel"][o0:o1]
    return dict(tid=tid, length=length, read_start=soa["start"][i0:i1], read_end=soa["end"][i0:i1],
                read_mapq=soa["mapq"][i0:i1], read_fwd=soa["fwd"][i0:i1],
                cpg_off=(soa["cpg_off"][i0:i1 + 1] - soa["cpg_off"][i0]).astype(np.uint32),
                cpg_pos=soa["cpg_pos"][o0:o1],
                cpg_rel=rel.astype(np.uint8) if (len(rel) == 0 or rel.max() < 256) else rel)


from metheor_amd.batches import device_batch  # noqa: E402,F401  (moved into the package: bench.py and smoke() must not import tests)


def contig_to_records(c, name="chrS"):
    """synthetic contig dict -> oracle.bamio.Records (150M reads with XM strings) so that the same
    input can be written as a real BAM and pushed through the CLI"""
    from oracle import bamio
    n = len(c["read_start"])
    off = c["cpg_off"].astype(np.int64)
    rl = int(c["read_end"][0] - c["read_start"][0] + 1) if n else 150
    xms = []
    for i in range(n):
        xm = bytearray(b"." * rl)
        for k in range(off[i], off[i + 1]):
            xm[int(c["cpg_rel"][k])] = ord("Z") if (int(c["cpg_pos"][k]) >> 31) else ord("z")
        xms.append(bytes(xm))
    flag = np.where(c["read_fwd"] == 1, 0, 16)
    return bamio.Records([(name, int(c["length"]))], np.zeros(n, np.int32), c["read_start"], flag, c["read_mapq"],
                         [[(rl << 4) | 0]] * n, xms)


def oracle_tsv_pdr(reads, names, **kw):
    from oracle import pyoracle
    t = reads.pdr(**kw)
    return "".join("%s\t%d\t%d\t%s\t%d\t%d\n" % (names[ti], p, p + 2, pyoracle.format_f32(v), c[0], c[1])
                   for ti, p, v, c in zip(t.tid, t.pos[:, 0], t.val, t.cnt))


def oracle_tsv_lpmd(reads, input_name, **kw):
    from oracle import pyoracle
    return "name\tlpmd\n%s\t%s\n" % (input_name, pyoracle.format_f32(reads.lpmd(**kw)["lpmd"]))


def subset_reads(c, mask):
    """keep the reads where mask is True (CSR rebuilt)"""
    mask = np.asarray(mask, bool)
    off = c["cpg_off"].astype(np.int64)
    n = np.diff(off)
    keep_call = np.repeat(mask, n)
    out = dict(c)
    for k in ("read_start", "read_end", "read_mapq", "read_fwd"):
        out[k] = c[k][mask]
    no = np.zeros(int(mask.sum()) + 1, np.int64)
    np.cumsum(n[mask], out=no[1:])
    out["cpg_off"] = no.astype(np.uint32)
    out["cpg_pos"] = c["cpg_pos"][keep_call]
    out["cpg_rel"] = c["cpg_rel"][keep_call]
    return out


def reblock_aligned(src, dst):
    """rewrite a BAM so that every BGZF block holds whole records (what htslib writes): the header in blocks of its own,
    then records packed greedily into <= 0xff00-byte blocks"""
    import gzip
    import struct
    raw = gzip.decompress(open(src, "rb").read())
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o); o += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, o); o += 8 + l_name
    with open(dst, "wb") as fh:
        for k in range(0, o, 0xff00):
            fh.write(_bamio()._bgzf_block(raw[k:min(k + 0xff00, o)]))
        blk = bytearray()
        while o < len(raw):
            bs, = struct.unpack_from("<i", raw, o)
            r = raw[o:o + 4 + bs]
            if blk and len(blk) + len(r) > 0xff00:
                fh.write(_bamio()._bgzf_block(bytes(blk))); blk = bytearray()
            blk += r
            o += 4 + bs
        if blk:
            fh.write(_bamio()._bgzf_block(bytes(blk)))
        fh.write(_bamio()._bgzf_block(b""))


def _bamio():
    from oracle import bamio
    return bamio


# ---- oracle text of a whole `metheor <sub>` run (the TSV bytes the reference writes) ---------------------
CLI_PARAMS = {   # command-line flags -> oracle keyword arguments
    "pdr": {"-d": "min_depth", "-p": "min_cpgs", "-q": "min_qual"},
    "mhl": {"-d": "min_depth", "-p": "min_cpgs", "-q": "min_qual"},
    "me": {"-d": "min_depth", "-q": "min_qual"}, "pm": {"-d": "min_depth", "-q": "min_qual"},
    "fdrp": {"-q": "min_qual", "-d": "min_depth", "-D": "max_depth", "-l": "min_overlap"},
    "qfdrp": {"-q": "min_qual", "-d": "min_depth", "-D": "max_depth", "-l": "min_overlap"},
    "lpmd": {"-m": "min_distance", "-M": "max_distance", "-q": "min_qual"},
}


def oracle_kwargs(sub, flags):
    """["-d", "3", "-p", "1"] -> {"min_depth": 3, "min_cpgs": 1}"""
    m = CLI_PARAMS[sub]
    return {m[flags[i]]: int(flags[i + 1]) for i in range(0, len(flags), 2)}


def oracle_text(reads, names, sub, input_name="", seed=0, **kw):
    """-> (output text, pairs-table text or None) of `metheor <sub>` as the oracle computes it"""
    from oracle import pyoracle
    f = pyoracle.format_f32
    if sub == "pdr":
        return oracle_tsv_pdr(reads, names, **kw), None
    if sub == "lpmd":
        res = reads.lpmd(pairs=True, **kw)
        t = res["pairs"]
        pairs = "chrom\tcpg1\tcpg2\tlpmd\tn_concordant\tn_discordant\n" + "".join(
            "%s\t%d\t%d\t%s\t%d\t%d\n" % (names[ti], p[0], p[1], f(v), c[0], c[1]) for ti, p, v, c in zip(t.tid, t.pos, t.val, t.cnt))
        return "name\tlpmd\n%s\t%s\n" % (input_name, f(res["lpmd"])), pairs
    if sub in ("mhl", "fdrp", "qfdrp"):
        t = reads.mhl(**kw) if sub == "mhl" else getattr(reads, sub)(seed=seed, **kw)
        return "".join("%s\t%d\t%d\t%s\n" % (names[ti], p[0], p[0] + 2, f(v)) for ti, p, v in zip(t.tid, t.pos, t.val)), None
    if sub in ("me", "pm"):
        t = getattr(reads, sub)(**kw)
        return "".join("%s\t%d\t%d\t%d\t%d\t%s\n" % (names[ti], p[0], p[1], p[2], p[3], f(v)) for ti, p, v in zip(t.tid, t.pos, t.val)), None
    raise ValueError(sub)


def assert_tsv_equals_oracle(sub, got, want):
    """byte equality for the sorted outputs; ME / PM are written in HashMap order by the reference (compare as sets of
    lines), and ME's value within 1e-6 (log2f last-ulp differences; the bar of BASELINE.json's north_star)"""
    if sub in ("me", "pm"):
        g, w = sorted(got.splitlines()), sorted(want.splitlines())
        assert len(g) == len(w), (sub, len(g), len(w))
        if sub == "pm":
            assert g == w, sub
        else:
            for a, b in zip(g, w):
                fa, fb = a.split("\t"), b.split("\t")
                assert fa[:5] == fb[:5] and abs(float(fa[5]) - float(fb[5])) <= 1e-6, (a, b)
    else:
        assert got == want, sub
