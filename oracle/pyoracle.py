"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("metheor_oracle.cpp", "tag_oracle.cpp", "metheor_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class _Records(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p),
                ("mapq", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("xm_off", C.c_void_p), ("xm", C.c_char_p)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i64, i32, u32, u64, u8 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_uint64, C.c_uint8
        L.orc_decode.restype = vp
        L.orc_decode.argtypes = [C.POINTER(_Records), i64, vp, vp, C.POINTER(C.c_int)]
        L.orc_reads_from_soa.restype = vp
        L.orc_reads_from_soa.argtypes = [i64] + [vp] * 8
        L.orc_reads_free.argtypes = [vp]
        L.orc_reads_n.restype = i64; L.orc_reads_n.argtypes = [vp]
        L.orc_reads_ncpg.restype = i64; L.orc_reads_ncpg.argtypes = [vp]
        L.orc_reads_export.argtypes = [vp] * 9
        L.orc_read_is_discordant.restype = C.c_int; L.orc_read_is_discordant.argtypes = [vp, i64]
        L.orc_read_stretch_info.restype = C.c_int; L.orc_read_stretch_info.argtypes = [vp, i64, vp, C.c_int]
        L.orc_read_pairwise.argtypes = [vp, i64, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.orc_pdr.restype = vp; L.orc_pdr.argtypes = [vp, u32, u64, u8]
        L.orc_lpmd.restype = vp
        L.orc_lpmd.argtypes = [vp, i32, i32, u8, C.c_int, C.POINTER(i64 * 4), C.POINTER(C.c_float)]
        L.orc_mhl.restype = vp; L.orc_mhl.argtypes = [vp, u32, u64, u8]
        L.orc_quartets.restype = vp; L.orc_quartets.argtypes = [vp, u32, u8, C.c_int]
        L.orc_fdrp.restype = vp; L.orc_fdrp.argtypes = [vp, u8, u64, u64, i32, u64, C.c_int]
        L.orc_sample_j.restype = i32; L.orc_sample_j.argtypes = [u64, i32, i32, i32]
        L.orc_result_n.restype = i64; L.orc_result_n.argtypes = [vp]
        L.orc_result_k.restype = C.c_int; L.orc_result_k.argtypes = [vp]
        L.orc_result_m.restype = C.c_int; L.orc_result_m.argtypes = [vp]
        for f in ("tid", "pos", "val", "cnt"):
            getattr(L, "orc_result_" + f).restype = vp
            getattr(L, "orc_result_" + f).argtypes = [vp]
        L.orc_result_free.argtypes = [vp]
        L.orc_format_f32.restype = C.c_int; L.orc_format_f32.argtypes = [C.c_float, C.c_char_p]
        L.orc_tag_xm.restype = i64
        L.orc_tag_xm.argtypes = [i32, i32, C.c_uint16, C.c_int, vp, u32, C.c_char_p, i64, C.c_char_p, i64, C.c_char_p, i64]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class ReferencePanic(RuntimeError):
    """the reference panics on this input (the message names where)"""


class Table:
    """result rows: tid[n], pos[n,k], val[n] (f32), cnt[n,m] (u32)"""

    def __init__(self, h):
        L = lib()
        n, k, m = L.orc_result_n(h), L.orc_result_k(h), L.orc_result_m(h)

        def arr(p, dt, cnt):
            if cnt == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()

        self.tid = arr(L.orc_result_tid(h), np.int32, n)
        self.pos = arr(L.orc_result_pos(h), np.int32, n * k).reshape(n, k)
        self.val = arr(L.orc_result_val(h), np.float32, n)
        self.cnt = arr(L.orc_result_cnt(h), np.uint32, n * m).reshape(n, m)
        L.orc_result_free(h)

    def __len__(self):
        return len(self.tid)


class Reads:
    """the reference's Vec<BismarkRead> (decoded), held by the oracle library"""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_reads_free(self.h)
            self.h = None

    @classmethod
    def decode(cls, rec, cpg_set=None):
        """rec: oracle.bamio.Records ; cpg_set: iterable of (tid,pos) or None"""
        cigar_off, cigar, xm_off, xm = rec.packed()
        keep = [rec.tid, rec.pos, rec.flag, rec.mapq, cigar_off, cigar, xm_off]
        r = _Records(len(rec), *[_ptr(np.ascontiguousarray(a)) for a in keep], xm)
        st = sp = None
        ns = 0
        if cpg_set is not None:
            cs = list(cpg_set)
            ns = len(cs)
            st = np.array([c[0] for c in cs], dtype=np.int32)
            sp = np.array([c[1] for c in cs], dtype=np.int32)
            if ns == 0:  # an empty set filters everything (HashSet::contains is always false)
                st = np.array([-2], dtype=np.int32); sp = np.array([-2], dtype=np.int32); ns = 1
        err = C.c_int(0)
        h = lib().orc_decode(C.byref(r), ns, _ptr(st), _ptr(sp), C.byref(err))
        if not h:
            raise RuntimeError("Error reading XM tag in BAM record. Make sure the reads are aligned using Bismark!")
        return cls(h)

    @classmethod
    def from_soa(cls, tid, start, end, mapq, fwd, cpg_off, cpg_pos, cpg_rel):
        a = [np.ascontiguousarray(tid, np.int32), np.ascontiguousarray(start, np.int32),
             np.ascontiguousarray(end, np.int32), np.ascontiguousarray(mapq, np.uint8),
             np.ascontiguousarray(fwd, np.uint8), np.ascontiguousarray(cpg_off, np.uint64),
             np.ascontiguousarray(cpg_pos, np.uint32), np.ascontiguousarray(cpg_rel, np.uint16)]
        return cls(lib().orc_reads_from_soa(len(a[0]), *[_ptr(x) for x in a]))

    def __len__(self):
        return lib().orc_reads_n(self.h)

    def soa(self):
        n, nc = len(self), lib().orc_reads_ncpg(self.h)
        out = dict(tid=np.zeros(n, np.int32), start=np.zeros(n, np.int32), end=np.zeros(n, np.int32),
                   mapq=np.zeros(n, np.uint8), fwd=np.zeros(n, np.uint8),
                   cpg_off=np.zeros(n + 1, np.uint64), cpg_pos=np.zeros(nc, np.uint32),
                   cpg_rel=np.zeros(nc, np.uint16))
        lib().orc_reads_export(self.h, *[_ptr(out[k]) for k in
                                         ("tid", "start", "end", "mapq", "fwd", "cpg_off", "cpg_pos", "cpg_rel")])
        return out

    # ---- per-read primitives -------------------------------------------------------
    def is_discordant(self, i):
        return lib().orc_read_is_discordant(self.h, i)

    def stretch_info(self, i, cap=512):
        c = np.zeros(cap, np.int32)
        mx = lib().orc_read_stretch_info(self.h, i, _ptr(c), cap)
        return c[:mx]

    def pairwise(self, i, min_d, max_d):
        a, b = C.c_int32(0), C.c_int32(0)
        lib().orc_read_pairwise(self.h, i, min_d, max_d, C.byref(a), C.byref(b))
        return a.value, b.value

    # ---- the compute_helper()s -----------------------------------------------------
    def pdr(self, min_depth=10, min_cpgs=4, min_qual=10):
        return Table(lib().orc_pdr(self.h, min_depth, min_cpgs, min_qual))

    def lpmd(self, min_distance=2, max_distance=16, min_qual=10, pairs=False):
        g = (C.c_int64 * 4)()
        v = C.c_float(0)
        t = Table(lib().orc_lpmd(self.h, min_distance, max_distance, min_qual, int(pairs), C.byref(g), C.byref(v)))
        return dict(lpmd=np.float32(v.value), n_concordant=g[0], n_discordant=g[1], n_read=g[2],
                    n_valid_read=g[3], pairs=t)

    def mhl(self, min_depth=10, min_cpgs=4, min_qual=10):
        return Table(lib().orc_mhl(self.h, min_depth, min_cpgs, min_qual))

    def me(self, min_depth=10, min_qual=10):
        return Table(lib().orc_quartets(self.h, min_depth, min_qual, 0))

    def pm(self, min_depth=10, min_qual=10):
        return Table(lib().orc_quartets(self.h, min_depth, min_qual, 1))

    def fdrp(self, min_qual=10, min_depth=10, max_depth=40, min_overlap=35, seed=0):
        return self._fdrp(min_qual, min_depth, max_depth, min_overlap, seed, 0)

    def qfdrp(self, min_qual=10, min_depth=10, max_depth=40, min_overlap=35, seed=0):
        return self._fdrp(min_qual, min_depth, max_depth, min_overlap, seed, 1)

    def _fdrp(self, min_qual, min_depth, max_depth, min_overlap, seed, which):
        h = lib().orc_fdrp(self.h, min_qual, min_depth, max_depth, min_overlap, seed, which)
        if not h:
            raise ReferencePanic("fdrp.rs:70-72: window index out of bounds (reverse-strand read, first call at start - 1, its own site 202 bp further)")
        return Table(h)


def format_f32(v):
    buf = C.create_string_buffer(80)
    lib().orc_format_f32(float(np.float32(v)), buf)
    return buf.value.decode()


def sample_j(seed, tid, pos, total):
    return lib().orc_sample_j(seed, tid, pos, total)


def ref_end(pos, cigar):
    """htslib bam_endpos: pos + reference bases consumed by M, D, N, =, X (pos + 1 when that is zero)"""
    rl = sum(int(c) >> 4 for c in cigar if (int(c) & 15) in (0, 2, 3, 7, 8))
    return pos + (rl if rl else 1)


def tag_xm(pos, flag, cigar, seq, contig, is_paired_end=False):
    """tag.rs:130-384 for one record; cigar = BAM-packed ops, seq / contig = bytes.  None where the reference panics."""
    cg = np.ascontiguousarray(cigar, dtype=np.uint32)
    out = C.create_string_buffer(len(seq) + 8)
    n = lib().orc_tag_xm(int(pos), int(ref_end(pos, cg)), int(flag), int(bool(is_paired_end)), _ptr(cg), len(cg),
                         bytes(seq), len(seq), bytes(contig), len(contig), out, len(seq) + 8)
    return None if n < 0 else out.raw[:n]
