"""ctypes binding of include/metheor_host.h (libmetheor_host.so: BGZF/BAM reader + XM decode, no GPU)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SYMBOLS = [
    "mth_host_open", "mth_host_close", "mth_host_last_error", "mth_host_notes", "mth_host_n_refs", "mth_host_ref_name",
    "mth_host_ref_len", "mth_host_ref_tid", "mth_host_set_xm_min_mapq", "mth_host_decode", "mth_host_decode_stream", "mth_host_bgzf_blocks", "mth_host_plan_shard", "mth_host_plan_region", "mth_host_cpg_set_keys", "mth_host_n_reads", "mth_host_n_cpgs",
    "mth_host_read_tid", "mth_host_read_start", "mth_host_read_end", "mth_host_read_mapq",
    "mth_host_read_fwd", "mth_host_cpg_off", "mth_host_cpg_pos", "mth_host_cpg_rel", "mth_host_format_f32", "mth_host_write_synthetic_bam",
    "mth_host_write_synthetic_bam_multi", "mth_host_write_synthetic_bam_repeat", "mth_host_header_text", "mth_host_path", "mth_host_sam_format", "mth_host_fasta_open", "mth_host_fasta_close",
    "mth_host_fasta_last_error", "mth_host_fasta_fetch",
]


class _Bgzf(C.Structure):      # mth_host_bgzf_t
    _fields_ = [("file", C.c_void_p), ("file_bytes", C.c_uint64), ("coff", C.c_void_p), ("csize", C.c_void_p),
                ("isize", C.c_void_p), ("n_blocks", C.c_uint64), ("header_bytes", C.c_uint64)]


class _Shard(C.Structure):     # mth_host_shard_t
    _fields_ = [("block_beg", C.c_uint64), ("block_end", C.c_uint64), ("first_byte", C.c_uint64),
                ("tid_beg", C.c_int32), ("pos_beg", C.c_int32), ("tid_end", C.c_int32), ("pos_end", C.c_int32)]


WINDOW_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)   # mth_host_window_cb


class HostError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


def library_path():
    return os.path.join(_HERE, "libmetheor_host.so")


def lib():
    global _LIB
    if _LIB is None:
        p = library_path()
        if not os.path.exists(p):
            raise ImportError("%s is missing: build it with `python -m metheor_amd.build`" % p)
        L = C.CDLL(p)
        vp = C.c_void_p
        L.mth_host_open.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
        L.mth_host_close.argtypes = [vp]; L.mth_host_close.restype = None
        L.mth_host_last_error.argtypes = [vp]; L.mth_host_last_error.restype = C.c_char_p
        L.mth_host_notes.argtypes = []; L.mth_host_notes.restype = C.c_uint32
        L.mth_host_n_refs.argtypes = [vp]
        L.mth_host_ref_name.argtypes = [vp, C.c_int]; L.mth_host_ref_name.restype = C.c_char_p
        L.mth_host_ref_len.argtypes = [vp, C.c_int]; L.mth_host_ref_len.restype = C.c_int64
        L.mth_host_ref_tid.argtypes = [vp, C.c_char_p]
        L.mth_host_decode.argtypes = [vp, C.c_char_p]
        for f in ("n_reads", "n_cpgs"):
            getattr(L, "mth_host_" + f).argtypes = [vp]; getattr(L, "mth_host_" + f).restype = C.c_int64
        for f in ("read_tid", "read_start", "read_end", "read_mapq", "read_fwd", "cpg_off", "cpg_pos", "cpg_rel"):
            getattr(L, "mth_host_" + f).argtypes = [vp]; getattr(L, "mth_host_" + f).restype = vp
        L.mth_host_bgzf_blocks.argtypes = [vp, vp]
        L.mth_host_plan_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp]
        L.mth_host_plan_region.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, vp]
        L.mth_host_format_f32.argtypes = [C.c_float, C.c_char_p]
        L.mth_host_write_synthetic_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int32] + [vp] * 6 + [C.c_uint64, C.c_int]
        L.mth_host_write_synthetic_bam_multi.argtypes = [C.c_char_p, C.c_int32, vp, vp, C.c_int64, C.c_int32] + [vp] * 7 + [C.c_uint64, C.c_int]
        L.mth_host_write_synthetic_bam_repeat.argtypes = [C.c_char_p, C.c_int32, vp, vp, C.c_int64, C.c_int32] + [vp] * 6 + [C.c_uint64, C.c_int]
        L.mth_host_path.argtypes = [vp]; L.mth_host_path.restype = C.c_char_p
        L.mth_host_header_text.argtypes = [vp, C.POINTER(C.c_uint64)]; L.mth_host_header_text.restype = vp
        L.mth_host_sam_format.argtypes = [vp, vp, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int64]; L.mth_host_sam_format.restype = C.c_int64
        L.mth_host_decode_stream.argtypes = [vp, WINDOW_CB, vp]
        L.mth_host_fasta_open.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
        L.mth_host_fasta_close.argtypes = [vp]; L.mth_host_fasta_close.restype = None
        L.mth_host_fasta_last_error.argtypes = [vp]; L.mth_host_fasta_last_error.restype = C.c_char_p
        L.mth_host_fasta_fetch.argtypes = [vp, C.c_char_p, C.c_int64, C.POINTER(vp), C.POINTER(C.c_int64)]
        _LIB = L
    return _LIB


def format_f32(v):
    buf = C.create_string_buffer(80)
    lib().mth_host_format_f32(float(np.float32(v)), buf)
    return buf.value.decode()


def write_synthetic_bam(path, c, contig="chr19", seed=0, threads=0):
    """c: a metheor_amd.synth contig dict -> BAM file (fast C++ writer, realistic byte content)"""
    n = len(c["read_start"])
    rl = int(c["read_end"][0] - c["read_start"][0] + 1) if n else 150
    arrs = [np.ascontiguousarray(c["read_start"], np.int32), np.ascontiguousarray(c["read_fwd"], np.uint8),
            np.ascontiguousarray(c["read_mapq"], np.uint8), np.ascontiguousarray(c["cpg_off"], np.uint64),
            np.ascontiguousarray(c["cpg_rel"], np.uint16), np.ascontiguousarray(c["cpg_pos"], np.uint32)]
    rc = lib().mth_host_write_synthetic_bam(os.fsencode(path), contig.encode(), int(c["length"]), n, rl,
                                            *[a.ctypes.data_as(C.c_void_p) for a in arrs], seed, threads)
    if rc != 0:
        raise HostError(rc, "cannot write " + path)


def write_synthetic_bam_repeat(path, c, names, seed=0, threads=0):
    """the reads of contig dict c on len(names) contigs (one copy each) -> one BAM file, without concatenating the SoA"""
    n = len(c["read_start"])
    rl = int(c["read_end"][0] - c["read_start"][0] + 1) if n else 150
    nm = (C.c_char_p * len(names))(*[x.encode() for x in names])
    ln = (C.c_int64 * len(names))(*[int(c["length"])] * len(names))
    arrs = [np.ascontiguousarray(c["read_start"], np.int32), np.ascontiguousarray(c["read_fwd"], np.uint8),
            np.ascontiguousarray(c["read_mapq"], np.uint8), np.ascontiguousarray(c["cpg_off"], np.uint64),
            np.ascontiguousarray(c["cpg_rel"], np.uint16), np.ascontiguousarray(c["cpg_pos"], np.uint32)]
    rc = lib().mth_host_write_synthetic_bam_repeat(os.fsencode(path), len(names), nm, ln, n, rl,
                                                   *[a.ctypes.data_as(C.c_void_p) for a in arrs], seed, threads)
    if rc != 0:
        raise HostError(rc, "cannot write " + path)


def write_synthetic_bam_multi(path, contigs, names, seed=0, threads=0):
    """contigs: metheor_amd.synth contig dicts in tid order (c["tid"] indexes names / the header) -> one BAM file"""
    from . import synth
    tid, start, end, mapq, fwd, off, pos, rel = synth.concat_oracle_soa(contigs)
    n = len(start)
    rl = int(end[0] - start[0] + 1) if n else 150
    lens = {c["tid"]: int(c["length"]) for c in contigs}
    nm = (C.c_char_p * len(names))(*[x.encode() for x in names])
    ln = (C.c_int64 * len(names))(*[lens.get(t, 1000) for t in range(len(names))])
    arrs = [np.ascontiguousarray(tid, np.int32), np.ascontiguousarray(start, np.int32), np.ascontiguousarray(fwd, np.uint8),
            np.ascontiguousarray(mapq, np.uint8), np.ascontiguousarray(off, np.uint64), np.ascontiguousarray(rel, np.uint16),
            np.ascontiguousarray(pos, np.uint32)]
    rc = lib().mth_host_write_synthetic_bam_multi(os.fsencode(path), len(names), nm, ln, n, rl,
                                                  *[a.ctypes.data_as(C.c_void_p) for a in arrs], seed, threads)
    if rc != 0:
        raise HostError(rc, "cannot write " + path)


class BamFile:
    def __init__(self, path):
        self.L = lib()
        self.h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = self.L.mth_host_open(os.fsencode(path), C.byref(self.h), err, 1024)
        if rc != 0:
            self.h = None
            raise HostError(rc, err.value.decode())
        self.refs = [(self.L.mth_host_ref_name(self.h, t).decode(), self.L.mth_host_ref_len(self.h, t))
                     for t in range(self.L.mth_host_n_refs(self.h))]

    def close(self):
        if self.h:
            self.L.mth_host_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bgzf_blocks(self):
        """the file's BGZF block table: dict(coff, csize, isize (numpy), header_bytes)"""
        bz = _Bgzf()
        rc = self.L.mth_host_bgzf_blocks(self.h, C.byref(bz))
        if rc != 0:
            raise HostError(rc, self.L.mth_host_last_error(self.h).decode())
        n = int(bz.n_blocks)

        def arr(p, dt):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy() if n else np.zeros(0, dt)
        return dict(coff=arr(bz.coff, np.uint64), csize=arr(bz.csize, np.uint32), isize=arr(bz.isize, np.uint32),
                    header_bytes=int(bz.header_bytes), file_bytes=int(bz.file_bytes))

    def plan_region(self, tid, beg, end, halo_bp=65536, bai=None):
        """mth_host_plan_region: the BGZF blocks the .bai index names for [beg - halo, end] of reference tid + the owned interval"""
        sh = _Shard()
        rc = self.L.mth_host_plan_region(self.h, os.fsencode(bai) if bai else None, int(tid), int(beg), int(end), int(halo_bp), C.byref(sh))
        if rc != 0:
            raise HostError(rc, self.L.mth_host_last_error(self.h).decode())
        return dict(block_beg=int(sh.block_beg), block_end=int(sh.block_end), first_byte=int(sh.first_byte),
                    tid_beg=int(sh.tid_beg), pos_beg=int(sh.pos_beg), tid_end=int(sh.tid_end), pos_end=int(sh.pos_end))

    def plan_shard(self, rank, world, halo_bp=65536):
        """mth_host_plan_shard: the BGZF blocks shard `rank` of `world` loads and the (tid, pos) interval it owns"""
        sh = _Shard()
        rc = self.L.mth_host_plan_shard(self.h, int(rank), int(world), int(halo_bp), C.byref(sh))
        if rc != 0:
            raise HostError(rc, self.L.mth_host_last_error(self.h).decode())
        return dict(block_beg=int(sh.block_beg), block_end=int(sh.block_end), first_byte=int(sh.first_byte),
                    beg=(int(sh.tid_beg), int(sh.pos_beg)), end=(int(sh.tid_end), int(sh.pos_end)))

    def decode(self, cpg_set=None):
        """-> SoA dict with the same keys as oracle.pyoracle.Reads.soa()"""
        rc = self.L.mth_host_decode(self.h, os.fsencode(cpg_set) if cpg_set else None)
        if rc != 0:
            raise HostError(rc, self.L.mth_host_last_error(self.h).decode())
        n, nc = self.L.mth_host_n_reads(self.h), self.L.mth_host_n_cpgs(self.h)

        def arr(name, dt, cnt):
            p = getattr(self.L, "mth_host_" + name)(self.h)
            if cnt == 0 or not p:
                return np.zeros(cnt, dtype=dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()

        return dict(tid=arr("read_tid", np.int32, n), start=arr("read_start", np.int32, n),
                    end=arr("read_end", np.int32, n), mapq=arr("read_mapq", np.uint8, n),
                    fwd=arr("read_fwd", np.uint8, n), cpg_off=arr("cpg_off", np.uint64, n + 1),
                    cpg_pos=arr("cpg_pos", np.uint32, nc), cpg_rel=arr("cpg_rel", np.uint16, nc))

    def staged_path(self):
        """SAM input: the path of the in-memory BAM the text was converted into (valid while this object lives)"""
        p = self.L.mth_host_path(self.h)
        return p.decode()

    def header_text(self):
        n = C.c_uint64(0)
        p = self.L.mth_host_header_text(self.h, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def windows(self):
        """mth_host_decode_stream: [(bytes of the window, record offsets uint64[n_rec + 1])] -- host inflate + record walk only"""
        out = []

        def cb(_user, buf, off, n_rec):
            o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(n_rec + 1,)).copy()
            out.append((C.string_at(buf, int(o[-1])), o))
            return 0
        rc = self.L.mth_host_decode_stream(self.h, WINDOW_CB(cb), None)
        if rc != 0:
            raise HostError(rc, self.L.mth_host_last_error(self.h).decode())
        return out

    def sam_line(self, raw, o0, o1, xm=None):
        """the record at raw[o0:o1] (block_size field included) as a SAM line; xm (bytes) is appended as XM:Z"""
        rec = (C.c_uint8 * (o1 - o0 - 4)).from_buffer_copy(raw[o0 + 4:o1])
        cap = 4 * (o1 - o0) + 256 + (len(xm) if xm else 0)
        buf = C.create_string_buffer(cap)
        n = self.L.mth_host_sam_format(self.h, rec, o1 - o0 - 4, xm, len(xm) if xm else 0, buf, cap)
        if n > cap:       # large signed B arrays print longer than 4 characters per byte: the formatter returns what it needs
            cap = int(n)
            buf = C.create_string_buffer(cap)
            n = self.L.mth_host_sam_format(self.h, rec, o1 - o0 - 4, xm, len(xm) if xm else 0, buf, cap)
        if n < 0 or n > cap:
            raise HostError(int(n), "cannot format the record")
        return buf.raw[:n]


class Fasta:
    def __init__(self, path):
        self.L = lib()
        self.h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = self.L.mth_host_fasta_open(os.fsencode(path), C.byref(self.h), err, 1024)
        if rc != 0:
            self.h = None
            raise HostError(rc, err.value.decode())

    def fetch(self, name, end_incl):
        p, n = C.c_void_p(), C.c_int64(0)
        rc = self.L.mth_host_fasta_fetch(self.h, name.encode(), int(end_incl), C.byref(p), C.byref(n))
        if rc != 0:
            raise HostError(rc, self.L.mth_host_fasta_last_error(self.h).decode())
        return C.string_at(p, n.value) if n.value else b""

    def close(self):
        if self.h:
            self.L.mth_host_fasta_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
