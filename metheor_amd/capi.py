"""ctypes binding of include/metheor_hip.h (one Python method per C entry point)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MTH_MEM_HOST, MTH_MEM_DEVICE = 0, 1

# every symbol include/metheor_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "mth_abi_version", "mth_ctx_create", "mth_ctx_destroy", "mth_ctx_set_stream", "mth_ctx_sync",
    "mth_strerror", "mth_last_error", "mth_notes", "mth_batch_prepare", "mth_batch_release", "mth_reset", "mth_pdr_lpmd_accumulate", "mth_pdr_count",
    "mth_pdr_fetch", "mth_result_buffer_alloc", "mth_result_buffer_free", "mth_pdr_device_view", "mth_lpmd_global", "mth_lpmd_add_unbatched", "mth_lpmd_from_counts",
    "mth_lpmd_export_device", "mth_device_count", "mth_allreduce_lpmd", "mth_rccl_unique_id", "mth_rccl_init_rank",
    "mth_allreduce_lpmd_rank", "mth_quartet_accumulate", "mth_quartet_fetch", "mth_mhl_accumulate", "mth_mhl_fetch", "mth_fdrp_accumulate", "mth_fdrp_fetch", "mth_lpmd_pairs_accumulate", "mth_lpmd_pairs_fetch",
    "mth_decode_records", "mth_decode_set_cpg_filter", "mth_decode_set_xm_min_mapq", "mth_bgzf_inflate", "mth_bgzf_decode", "mth_bgzf_stage", "mth_decode_reserve", "mth_decoded_fetch", "mth_decoded_contigs", "mth_decoded_sort", "mth_decoded_group", "mth_group_define", "mth_group_clear", "mth_fileorder_run", "mth_fileorder_fetch", "mth_decoded_batch", "mth_tag_set_genome", "mth_tag_records",
    "mth_timing_enable", "mth_timing_reset", "mth_timing_get", "mth_timing_num_kernels",
    "mth_timing_kernel_name",
]


class MthError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("metheor_hip error %d: %s" % (status, msg))
        self.status = status


class mth_batch_t(C.Structure):
    _fields_ = [("tid", C.c_int32), ("region_beg", C.c_int32), ("region_end", C.c_int32),
                ("max_span", C.c_int32), ("n_reads", C.c_uint32), ("n_cpgs", C.c_uint32),
                ("mem", C.c_int32),
                ("read_start", C.c_void_p), ("read_end", C.c_void_p), ("read_mapq", C.c_void_p),
                ("read_fwd", C.c_void_p), ("cpg_off", C.c_void_p), ("cpg_pos", C.c_void_p),
                ("cpg_rel", C.c_void_p), ("cpg_rel16", C.c_void_p)]


class mth_decoded_t(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_cpgs", C.c_uint64), ("tid", C.c_void_p), ("start", C.c_void_p),
                ("end", C.c_void_p), ("mapq", C.c_void_p), ("fwd", C.c_void_p), ("cpg_off", C.c_void_p),
                ("cpg_pos", C.c_void_p), ("cpg_rel", C.c_void_p)]


class mth_pdr_lpmd_params_t(C.Structure):
    _fields_ = [("pdr_min_depth", C.c_uint32), ("pdr_min_cpgs", C.c_uint32),
                ("pdr_min_qual", C.c_uint8), ("lpmd_min_qual", C.c_uint8),
                ("want_pdr", C.c_uint8), ("want_lpmd", C.c_uint8),
                ("lpmd_min_distance", C.c_int32), ("lpmd_max_distance", C.c_int32)]


class mth_quartet_params_t(C.Structure):
    _fields_ = [("min_qual", C.c_uint8)]


class mth_mhl_params_t(C.Structure):
    _fields_ = [("min_depth", C.c_uint32), ("min_cpgs", C.c_uint32), ("min_qual", C.c_uint8)]


class mth_fdrp_params_t(C.Structure):
    _fields_ = [("min_depth", C.c_uint64), ("seed", C.c_uint64), ("max_depth", C.c_uint32),
                ("min_overlap", C.c_int32), ("min_qual", C.c_uint8)]


class mth_fileorder_params_t(C.Structure):
    _fields_ = [("measure", C.c_int32), ("min_depth", C.c_uint32), ("min_cpgs", C.c_uint32), ("max_depth", C.c_uint32),
                ("min_overlap", C.c_int32), ("min_qual", C.c_uint8), ("seed", C.c_uint64)]


class mth_lpmd_pairs_params_t(C.Structure):
    _fields_ = [("min_distance", C.c_int32), ("max_distance", C.c_int32), ("min_qual", C.c_uint8)]


class mth_tag_out_t(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("xm_off", C.c_void_p), ("xm_len", C.c_void_p), ("xm", C.c_void_p)]


def library_path():
    # METHEOR_HIP_LIB: another build of the same library (A/B timing of kernel variants on one box)
    return os.environ.get("METHEOR_HIP_LIB") or os.path.join(_HERE, "libmetheor_hip.so")


def lib():
    """load libmetheor_hip.so; fails loudly when it has not been built (no fallback)"""
    global _LIB
    if _LIB is None:
        p = library_path()
        if not os.path.exists(p):
            raise ImportError("%s is missing: build it with `python -m metheor_amd.build` "
                              "(hipcc, gfx950). There is no CPU fallback." % p)
        # In a Python process PyTorch brings its own bundled ROCm runtime (libamdhip64.so.7 under
        # torch/lib).  Two HIP/HSA runtimes in one process break each other, so make sure torch's
        # is the one already loaded before this library's NEEDED libamdhip64.so.7 is resolved.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(p)
        vp = C.c_void_p
        L.mth_abi_version.restype = C.c_int
        L.mth_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.mth_ctx_destroy.argtypes = [vp]; L.mth_ctx_destroy.restype = None
        L.mth_ctx_set_stream.argtypes = [vp, vp]
        L.mth_ctx_sync.argtypes = [vp]
        L.mth_strerror.restype = C.c_char_p; L.mth_strerror.argtypes = [C.c_int]
        L.mth_last_error.restype = C.c_char_p; L.mth_last_error.argtypes = [vp]
        L.mth_notes.restype = C.c_uint32; L.mth_notes.argtypes = [vp]
        L.mth_batch_prepare.argtypes = [vp, C.POINTER(mth_batch_t), C.POINTER(mth_batch_t)]
        L.mth_batch_release.argtypes = [vp, C.POINTER(mth_batch_t)]
        L.mth_reset.argtypes = [vp]
        L.mth_pdr_lpmd_accumulate.argtypes = [vp, C.POINTER(mth_batch_t), C.POINTER(mth_pdr_lpmd_params_t)]
        L.mth_pdr_count.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.mth_pdr_fetch.argtypes = [vp] * 6
        L.mth_result_buffer_alloc.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p)]
        L.mth_result_buffer_free.argtypes = [vp, vp]
        L.mth_pdr_device_view.argtypes = [vp, C.POINTER(C.c_uint64)] + [C.POINTER(vp)] * 4
        L.mth_lpmd_global.argtypes = [vp, C.POINTER(C.c_int64 * 4), C.POINTER(C.c_float)]
        L.mth_lpmd_add_unbatched.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.mth_lpmd_from_counts.restype = C.c_float
        L.mth_lpmd_from_counts.argtypes = [C.c_int64, C.c_int64]
        L.mth_lpmd_export_device.argtypes = [vp, vp]
        L.mth_device_count.argtypes = [C.POINTER(C.c_int)]
        L.mth_allreduce_lpmd.argtypes = [C.POINTER(vp), C.c_int]
        L.mth_rccl_unique_id.argtypes = [vp]
        L.mth_rccl_init_rank.argtypes = [vp, vp, C.c_int, C.c_int]
        L.mth_allreduce_lpmd_rank.argtypes = [vp]
        L.mth_quartet_accumulate.argtypes = [vp, C.POINTER(mth_batch_t), C.POINTER(mth_quartet_params_t)]
        L.mth_quartet_fetch.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)] + [vp] * 5
        L.mth_mhl_accumulate.argtypes = [vp, C.POINTER(mth_batch_t), C.POINTER(mth_mhl_params_t)]
        L.mth_mhl_fetch.argtypes = [vp, C.POINTER(C.c_uint64)] + [vp] * 4
        L.mth_fdrp_accumulate.argtypes = [vp, C.POINTER(mth_batch_t), C.POINTER(mth_fdrp_params_t)]
        L.mth_fdrp_fetch.argtypes = [vp, C.POINTER(C.c_uint64)] + [vp] * 5
        L.mth_lpmd_pairs_accumulate.argtypes = [vp, C.POINTER(mth_batch_t), C.POINTER(mth_lpmd_pairs_params_t)]
        L.mth_lpmd_pairs_fetch.argtypes = [vp, C.POINTER(C.c_uint64)] + [vp] * 6
        L.mth_decode_records.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(mth_decoded_t)]
        L.mth_decode_set_cpg_filter.argtypes = [vp, vp, C.c_uint64, C.c_int]
        L.mth_decode_set_xm_min_mapq.argtypes = [vp, C.c_uint32]
        L.mth_bgzf_inflate.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]
        L.mth_bgzf_decode.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(mth_decoded_t)]
        L.mth_bgzf_stage.argtypes = [vp, vp, C.c_uint64]
        L.mth_decode_reserve.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.mth_tag_set_genome.argtypes = [vp, C.c_int32, vp, vp, vp]
        L.mth_tag_records.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(mth_tag_out_t)]
        L.mth_decoded_fetch.argtypes = [vp] * 9
        L.mth_decoded_contigs.argtypes = [vp, C.c_uint32, vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.mth_decoded_sort.argtypes = [vp]
        L.mth_decoded_group.argtypes = [vp, C.c_uint32, vp, vp, vp, C.POINTER(C.c_uint32), vp, vp]
        L.mth_group_define.argtypes = [vp, C.c_uint32, vp, vp, C.POINTER(C.c_int32)]
        L.mth_group_clear.argtypes = [vp]
        L.mth_fileorder_run.argtypes = [vp, C.POINTER(mth_fileorder_params_t)]
        L.mth_fileorder_fetch.argtypes = [vp, C.POINTER(C.c_uint64)] + [vp] * 6
        L.mth_decoded_batch.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(mth_batch_t)]
        L.mth_timing_enable.argtypes = [vp, C.c_int]
        L.mth_timing_reset.argtypes = [vp]
        L.mth_timing_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mth_timing_num_kernels.restype = C.c_int
        L.mth_timing_kernel_name.restype = C.c_char_p; L.mth_timing_kernel_name.argtypes = [C.c_int]
        _LIB = L
    return _LIB


class PdrLpmdParams:
    """pdr.rs:82-89 / lpmd.rs:125-133 arguments with the reference's clap defaults (lib.rs)"""

    def __init__(self, min_depth=10, min_cpgs=4, min_qual=10, min_distance=2, max_distance=16,
                 lpmd_min_qual=10, want_pdr=True, want_lpmd=True):
        self.c = mth_pdr_lpmd_params_t(min_depth, min_cpgs, min_qual, lpmd_min_qual,
                                       int(want_pdr), int(want_lpmd), min_distance, max_distance)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DeviceBatch:
    """an mth_batch_t filled in by the library (device pointers it owns)"""

    def __init__(self):
        self.c = mth_batch_t()


class PreparedBatch:
    """what Engine.batch_prepare returns: a batch (mem = MTH_MEM_PREPARED) every *_accumulate method takes in place of the original --
    device-resident, its read index built once and shared by the measures.  Keeps the original alive; release() frees the index and the
    device copies of a host batch; what is not released, the engine's close frees (the context keeps a registry of its prepared
    batches and validates a handle by look-up)."""

    def __init__(self, eng, orig):
        self.eng, self.orig = eng, orig
        self.c = mth_batch_t()
        eng._check(eng.L.mth_batch_prepare(eng.h, C.byref(orig.c), C.byref(self.c)))
        self.n_reads, self.n_cpgs, self.tid = orig.n_reads, orig.n_cpgs, orig.tid

    def release(self):
        if self.eng is not None and getattr(self.eng, "h", None):
            self.eng._check(self.eng.L.mth_batch_release(self.eng.h, C.byref(self.c)))
        self.eng = None


class Batch:
    """mth_batch_t over numpy arrays (host) or torch CUDA tensors (device); keeps them alive"""

    def __init__(self, tid, region_beg, region_end, read_start, read_end, read_mapq, cpg_off,
                 cpg_pos, cpg_rel, read_fwd=None, max_span=None):
        arrs = dict(read_start=read_start, read_end=read_end, read_mapq=read_mapq, cpg_off=cpg_off,
                    cpg_pos=cpg_pos, cpg_rel=cpg_rel, read_fwd=read_fwd)
        dev = _is_torch(read_start)
        self.keep = []
        ptr = {}
        want = dict(read_start=("int32", 4), read_end=("int32", 4), read_mapq=("uint8", 1),
                    cpg_off=("uint32", 4), cpg_pos=("uint32", 4), read_fwd=("uint8", 1))
        for k, a in arrs.items():
            if a is None:
                ptr[k] = None
                continue
            if dev:
                assert a.is_cuda and a.is_contiguous(), k
                if k == "cpg_rel":
                    assert a.element_size() in (1, 2)
                else:
                    assert a.element_size() == want[k][1], (k, a.dtype)
                self.keep.append(a)
                ptr[k] = a.data_ptr() if a.numel() else None
            else:
                if k == "cpg_rel":
                    a = np.ascontiguousarray(a)
                    assert a.dtype in (np.uint8, np.uint16), a.dtype
                else:
                    a = np.ascontiguousarray(a, dtype=want[k][0])
                self.keep.append(a)
                ptr[k] = a.ctypes.data if a.size else None
        rel = self.keep[[k for k, v in arrs.items() if v is not None].index("cpg_rel")]
        rel16 = (rel.element_size() if dev else rel.dtype.itemsize) == 2
        n_reads = int(read_start.shape[0])
        n_cpgs = int(cpg_pos.shape[0])
        if max_span is None:
            if dev:
                max_span = int((read_end - read_start).max().item()) + 1 if n_reads else 0
            else:
                max_span = int((np.asarray(read_end, np.int64) - np.asarray(read_start, np.int64)).max()) + 1 if n_reads else 0
        # a zero-length cpg_rel still has to say which width it is
        dummy = self.keep[0].data_ptr() if dev else self.keep[0].ctypes.data
        relp = ptr["cpg_rel"] or (dummy if n_reads else None)
        self.c = mth_batch_t(tid, region_beg, region_end, max(max_span, 0), n_reads, n_cpgs,
                             MTH_MEM_DEVICE if dev else MTH_MEM_HOST,
                             ptr["read_start"], ptr["read_end"], ptr["read_mapq"], ptr["read_fwd"],
                             ptr["cpg_off"], ptr["cpg_pos"],
                             None if rel16 else relp, relp if rel16 else None)
        self.n_reads, self.n_cpgs, self.tid = n_reads, n_cpgs, tid


class Engine:
    """one mth_ctx_t (one GPU).  Methods map 1:1 onto the C entry points."""

    def __init__(self, device=0, stream=None):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.mth_ctx_create(device, C.byref(self.h))
        if rc != 0:
            raise MthError(rc, self.L.mth_strerror(rc).decode())
        if stream is not None:
            self._check(self.L.mth_ctx_set_stream(self.h, C.c_void_p(stream)))
        # Batches whose kernels may still be queued or may be REPLAYED from their arrays at a later entry point (pipelined PDR + LPMD
        # batches; queued ME / PM and pairs batches: include/metheor_hip.h).  The engine keeps them alive until its next synchronising
        # call, so that a caller who drops or recycles a temporary batch right after the accumulate call cannot have the replay read
        # freed memory.  Keyed by id: a loop over one resident batch holds it once.
        self._inflight = {}

    def _hold(self, batch):
        self._inflight[id(batch)] = batch

    def _settled(self):
        self._inflight.clear()

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.L.mth_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.L.mth_strerror(rc).decode()
            last = self.L.mth_last_error(self.h).decode()
            raise MthError(rc, msg + (" (" + last + ")" if last else ""))

    def reset(self):
        self._check(self.L.mth_reset(self.h))

    def batch_prepare(self, batch):
        """one device-resident copy and ONE read index for all the measures of this batch (include/metheor_hip.h, "prepared batches")"""
        return PreparedBatch(self, batch)

    def notes(self):
        """non-fatal findings of the device decode so far (bit 0: a CIGAR P operation was decoded)"""
        return int(self.L.mth_notes(self.h))

    def sync(self):
        self._check(self.L.mth_ctx_sync(self.h))
        self._settled()

    def pdr_lpmd_accumulate(self, batch, params):
        self._hold(batch)
        self._check(self.L.mth_pdr_lpmd_accumulate(self.h, C.byref(batch.c), C.byref(params.c)))

    def pdr_count(self):
        n = C.c_uint64(0)
        self._check(self.L.mth_pdr_count(self.h, C.byref(n)))
        self._settled()
        return n.value

    def pdr_fetch(self):
        n = self.pdr_count()
        out = dict(tid=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), pdr=np.zeros(n, np.float32),
                   n_concordant=np.zeros(n, np.uint32), n_discordant=np.zeros(n, np.uint32))
        self._check(self.L.mth_pdr_fetch(self.h, *[out[k].ctypes.data_as(C.c_void_p) for k in
                                                   ("tid", "pos", "pdr", "n_concordant", "n_discordant")]))
        return out

    def lpmd_global(self):
        g = (C.c_int64 * 4)()
        v = C.c_float(0)
        self._check(self.L.mth_lpmd_global(self.h, C.byref(g), C.byref(v)))
        self._settled()
        return dict(n_concordant=g[0], n_discordant=g[1], n_read=g[2], n_valid_read=g[3],
                    lpmd=np.float32(v.value))

    def lpmd_add_unbatched(self, n_read, n_valid_read):
        """records left out of the batches (no contig / no aligned base): lpmd.rs:176-179 still counts them"""
        self._check(self.L.mth_lpmd_add_unbatched(self.h, int(n_read), int(n_valid_read)))

    def lpmd_export_device(self, dst_ptr):
        """dst_ptr: device address of 4 x int64 (e.g. torch_tensor.data_ptr())"""
        self._check(self.L.mth_lpmd_export_device(self.h, C.c_void_p(dst_ptr)))

    # ---- the exchange step (mth_rccl.hip): one RCCL all-reduce of the four LPMD counters ----
    @staticmethod
    def rccl_unique_id():
        """128 bytes made by rank 0; ship them to every rank (torch.distributed.broadcast, MPI, a file)"""
        buf = (C.c_uint8 * 128)()
        rc = lib().mth_rccl_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise MthError(rc, lib().mth_strerror(rc).decode())
        return bytes(buf)

    def rccl_init_rank(self, unique_id, rank, world):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.L.mth_rccl_init_rank(self.h, C.cast(buf, C.c_void_p), int(rank), int(world)))

    def allreduce_lpmd_rank(self):
        """collective over the ranks of rccl_init_rank, enqueued on this context's stream; afterwards lpmd_global()
        returns the node-wide counters"""
        self._check(self.L.mth_allreduce_lpmd_rank(self.h))

    def lpmd_from_counts(self, n_conc, n_disc):
        return np.float32(self.L.mth_lpmd_from_counts(int(n_conc), int(n_disc)))

    def quartet_accumulate(self, batch, min_qual=10):
        p = mth_quartet_params_t(min_qual)
        self._hold(batch)
        self._check(self.L.mth_quartet_accumulate(self.h, C.byref(batch.c), C.byref(p)))

    def quartet_fetch(self, min_depth=10):
        """rows (depth >= min_depth) of the HashMap<Quartet,...>: tid, pos[n,4], cnt[n,16], me, pm"""
        n = C.c_uint64(0)
        self._check(self.L.mth_quartet_fetch(self.h, min_depth, C.byref(n), None, None, None, None, None))
        self._settled()
        k = n.value
        out = dict(tid=np.zeros(k, np.int32), pos=np.zeros((k, 4), np.int32), cnt=np.zeros((k, 16), np.uint32),
                   me=np.zeros(k, np.float32), pm=np.zeros(k, np.float32))
        self._check(self.L.mth_quartet_fetch(self.h, min_depth, C.byref(n), *[out[x].ctypes.data_as(C.c_void_p) for x in
                                                                              ("tid", "pos", "cnt", "me", "pm")]))
        return out

    def mhl_accumulate(self, batch, min_depth=10, min_cpgs=4, min_qual=10):
        p = mth_mhl_params_t(min_depth, min_cpgs, min_qual)
        self._check(self.L.mth_mhl_accumulate(self.h, C.byref(batch.c), C.byref(p)))

    def mhl_fetch(self):
        n = C.c_uint64(0)
        self._check(self.L.mth_mhl_fetch(self.h, C.byref(n), None, None, None, None))
        k = n.value
        out = dict(tid=np.zeros(k, np.int32), pos=np.zeros(k, np.int32), mhl=np.zeros(k, np.float32), cov=np.zeros(k, np.uint32))
        self._check(self.L.mth_mhl_fetch(self.h, C.byref(n), *[out[x].ctypes.data_as(C.c_void_p) for x in ("tid", "pos", "mhl", "cov")]))
        return out

    def fdrp_accumulate(self, batch, min_qual=10, min_depth=10, max_depth=40, min_overlap=35, seed=0):
        p = mth_fdrp_params_t(min_depth, seed, max_depth, min_overlap, min_qual)
        self._check(self.L.mth_fdrp_accumulate(self.h, C.byref(batch.c), C.byref(p)))

    def fdrp_fetch(self):
        n = C.c_uint64(0)
        self._check(self.L.mth_fdrp_fetch(self.h, C.byref(n), None, None, None, None, None))
        k = n.value
        out = dict(tid=np.zeros(k, np.int32), pos=np.zeros(k, np.int32), fdrp=np.zeros(k, np.float32),
                   qfdrp=np.zeros(k, np.float32), n_reads=np.zeros(k, np.uint32))
        self._check(self.L.mth_fdrp_fetch(self.h, C.byref(n), *[out[x].ctypes.data_as(C.c_void_p) for x in
                                                                ("tid", "pos", "fdrp", "qfdrp", "n_reads")]))
        return out

    def lpmd_pairs_accumulate(self, batch, min_distance=2, max_distance=16, min_qual=10):
        p = mth_lpmd_pairs_params_t(min_distance, max_distance, min_qual)
        self._hold(batch)
        self._check(self.L.mth_lpmd_pairs_accumulate(self.h, C.byref(batch.c), C.byref(p)))

    def lpmd_pairs_fetch(self):
        n = C.c_uint64(0)
        self._check(self.L.mth_lpmd_pairs_fetch(self.h, C.byref(n), None, None, None, None, None, None))
        self._settled()
        k = n.value
        out = dict(tid=np.zeros(k, np.int32), pos1=np.zeros(k, np.int32), pos2=np.zeros(k, np.int32),
                   lpmd=np.zeros(k, np.float32), n_concordant=np.zeros(k, np.uint32), n_discordant=np.zeros(k, np.uint32))
        self._check(self.L.mth_lpmd_pairs_fetch(self.h, C.byref(n), *[out[x].ctypes.data_as(C.c_void_p) for x in
                                                                      ("tid", "pos1", "pos2", "lpmd", "n_concordant", "n_discordant")]))
        return out

    # ---- device-side BAM record decode (mth_decode.hip) ----
    def decode_records(self, raw, rec_off, append=False):
        """raw: the inflated BAM record stream (numpy uint8 / bytes, or a torch CUDA uint8 tensor); rec_off: uint64
        byte offsets of the records (n + 1).  The decoded SoA stays on the device; returns (n_reads, n_cpgs)."""
        dev = _is_torch(raw)
        if dev:
            assert raw.is_cuda and raw.is_contiguous() and rec_off.is_cuda and rec_off.is_contiguous()
            self._dec_keep = (raw, rec_off)
            rp, op, nb, nr = raw.data_ptr(), rec_off.data_ptr(), raw.numel(), rec_off.numel() - 1
        else:
            raw = np.frombuffer(raw, np.uint8) if isinstance(raw, (bytes, bytearray, memoryview)) else np.ascontiguousarray(raw, np.uint8)
            rec_off = np.ascontiguousarray(rec_off, np.uint64)
            self._dec_keep = (raw, rec_off)
            rp, op, nb, nr = raw.ctypes.data, rec_off.ctypes.data, raw.size, rec_off.size - 1
        d = mth_decoded_t()
        self._check(self.L.mth_decode_records(self.h, rp, nb, op, max(nr, 0), 1 if dev else 0, int(bool(append)), C.byref(d)))
        self._decoded = d
        return int(d.n_reads), int(d.n_cpgs)

    def bgzf_inflate(self, file_bytes, coff, csize, isize):
        """inflate the given BGZF blocks on the device; returns the concatenated inflated bytes (numpy uint8)"""
        fb = np.frombuffer(file_bytes, np.uint8) if isinstance(file_bytes, (bytes, bytearray, memoryview)) else np.ascontiguousarray(file_bytes, np.uint8)
        coff = np.ascontiguousarray(coff, np.uint64); csize = np.ascontiguousarray(csize, np.uint32); isize = np.ascontiguousarray(isize, np.uint32)
        out = np.zeros(int(isize.astype(np.uint64).sum()), np.uint8)
        n = C.c_uint64(0)
        self._check(self.L.mth_bgzf_inflate(self.h, fb.ctypes.data, fb.size, coff.ctypes.data, csize.ctypes.data, isize.ctypes.data,
                                            len(coff), out.ctypes.data, C.byref(n)))
        assert n.value == len(out)
        return out

    def bgzf_decode(self, file_bytes, coff, csize, isize, first_byte, append=False):
        """file_bytes: the BAM file (bytes / uint8 array, host); coff/csize/isize: per-BGZF-block payload offset, payload
        size, inflated size.  Inflate + record walk + decode on the device; returns (n_reads, n_cpgs)."""
        fb = np.frombuffer(file_bytes, np.uint8) if isinstance(file_bytes, (bytes, bytearray, memoryview)) else np.ascontiguousarray(file_bytes, np.uint8)
        coff = np.ascontiguousarray(coff, np.uint64); csize = np.ascontiguousarray(csize, np.uint32); isize = np.ascontiguousarray(isize, np.uint32)
        d = mth_decoded_t()
        self._check(self.L.mth_bgzf_decode(self.h, fb.ctypes.data, fb.size, coff.ctypes.data, csize.ctypes.data, isize.ctypes.data,
                                           len(coff), int(first_byte), int(bool(append)), C.byref(d)))
        self._decoded = d
        return int(d.n_reads), int(d.n_cpgs)

    def tag_set_genome(self, contigs):
        """contigs: [(header LN, bases as bytes)] in tid order (mth_tag_set_genome; tag.rs:419-431)"""
        n = len(contigs)
        ln = (C.c_int64 * max(n, 1))(*[int(c[0]) for c in contigs])
        got = (C.c_int64 * max(n, 1))(*[len(c[1]) for c in contigs])
        bufs = [np.frombuffer(bytes(c[1]), np.uint8) for c in contigs]
        ptr = (C.c_void_p * max(n, 1))(*[b.ctypes.data if len(b) else None for b in bufs])
        self._check(self.L.mth_tag_set_genome(self.h, n, ln, ptr, got))

    def tag_records(self, raw, rec_off, is_paired_end=False):
        """raw: the inflated BAM records (bytes), rec_off: uint64[n_rec + 1]; -> list of XM strings (bytes), tag.rs:130-384"""
        rb = np.frombuffer(raw, np.uint8)
        off = np.ascontiguousarray(rec_off, np.uint64)
        t = mth_tag_out_t()
        n = len(off) - 1
        self._check(self.L.mth_tag_records(self.h, rb.ctypes.data if len(rb) else None, len(rb), off.ctypes.data, max(n, 0), 0,
                                           int(bool(is_paired_end)), C.byref(t)))
        if n <= 0:
            return []
        xo = np.ctypeslib.as_array(C.cast(t.xm_off, C.POINTER(C.c_uint64)), shape=(n + 1,))
        xl = np.ctypeslib.as_array(C.cast(t.xm_len, C.POINTER(C.c_uint32)), shape=(n,))
        return [C.string_at(t.xm + int(xo[i]), int(xl[i])) for i in range(n)]

    def decoded_fetch(self):
        d = self._decoded
        nr, nc = int(d.n_reads), int(d.n_cpgs)
        out = dict(tid=np.zeros(nr, np.int32), start=np.zeros(nr, np.int32), end=np.zeros(nr, np.int32),
                   mapq=np.zeros(nr, np.uint8), fwd=np.zeros(nr, np.uint8), cpg_off=np.zeros(nr + 1, np.uint64),
                   cpg_pos=np.zeros(nc, np.uint32), cpg_rel=np.zeros(nc, np.uint16))
        self._check(self.L.mth_decoded_fetch(self.h, *[out[k].ctypes.data_as(C.c_void_p) for k in
                                                       ("tid", "start", "end", "mapq", "fwd", "cpg_off", "cpg_pos", "cpg_rel")]))
        return out

    def decode_set_cpg_filter(self, sites):
        """sites: iterable of (tid, pos) or None (no filter); an empty iterable filters every call"""
        if sites is None:
            self._check(self.L.mth_decode_set_cpg_filter(self.h, None, 0, 0))
            return
        keys = np.array(sorted({(int(t) << 32) | (int(p) & 0xffffffff) for t, p in sites}), np.uint64)
        self._check(self.L.mth_decode_set_cpg_filter(self.h, keys.ctypes.data if len(keys) else None, len(keys), 1))

    def decoded_contigs(self, cap=65536):
        """runs of equal tid in the decoded stream: (tids, read_beg, read_end, flags)"""
        tids, beg, end = np.zeros(cap, np.int32), np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n, fl = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.mth_decoded_contigs(self.h, cap, tids.ctypes.data, beg.ctypes.data, end.ctypes.data, C.byref(n), C.byref(fl)))
        k = min(n.value, cap)
        return tids[:k], beg[:k], end[:k], fl.value

    def group_define(self, tids, voff):
        """contigs `tids` laid out at virtual offsets `voff` (ascending): the handle to use as the tid of their one batch"""
        t, v = np.ascontiguousarray(tids, np.int32), np.ascontiguousarray(voff, np.int64)
        assert len(t) == len(v)
        h = C.c_int32(0)
        self._check(self.L.mth_group_define(self.h, len(t), t.ctypes.data, v.ctypes.data, C.byref(h)))
        self.__dict__.setdefault("_groups", {})[h.value] = (t.copy(), v.copy())
        return h.value

    def group_copy(self, other, handle):
        """define on this engine the group `handle` of engine `other` (must come out under the same handle)"""
        g = other._groups[handle]
        assert self.group_define(*g) == handle

    def group_clear(self):
        self._check(self.L.mth_group_clear(self.h))

    def decoded_group(self, tids, read_beg, read_end):
        """pack the decoded stream's contigs into groups (positions shifted in place): [(batch_tid, read_beg, read_end, [tids])], [] if not grouped"""
        t = np.ascontiguousarray(tids, np.int32); b = np.ascontiguousarray(read_beg, np.uint64); e = np.ascontiguousarray(read_end, np.uint64)
        n = len(t)
        first, bt, ng = np.zeros(n + 1, np.uint32), np.zeros(max(n, 1), np.int32), C.c_uint32(0)
        self._check(self.L.mth_decoded_group(self.h, n, t.ctypes.data, b.ctypes.data, e.ctypes.data, C.byref(ng), first.ctypes.data, bt.ctypes.data))
        return [(int(bt[g]), int(b[first[g]]), int(e[first[g + 1] - 1]), [int(x) for x in t[first[g]:first[g + 1]]]) for g in range(ng.value)]

    def decoded_sort(self):
        """the decoded stream re-ordered by (tid, start) on the device (order-free measures on unsorted input)"""
        self._check(self.L.mth_decoded_sort(self.h))

    def fileorder(self, measure, min_depth=10, min_cpgs=4, min_qual=10, max_depth=40, min_overlap=35, seed=0):
        """PDR (0) / MHL (1) / FDRP + qFDRP (2) of the decoded stream in FILE order: the reference's stream semantics on input that is
        not coordinate-sorted (mth_fileorder.hip); rows sorted by (tid, pos)"""
        p = mth_fileorder_params_t(int(measure), int(min_depth), int(min_cpgs), int(max_depth), int(min_overlap), int(min_qual), int(seed))
        self._check(self.L.mth_fileorder_run(self.h, C.byref(p)))
        n = C.c_uint64(0)
        self._check(self.L.mth_fileorder_fetch(self.h, C.byref(n), None, None, None, None, None, None))
        k = n.value
        out = dict(tid=np.zeros(k, np.int32), pos=np.zeros(k, np.int32), v0=np.zeros(k, np.float32), v1=np.zeros(k, np.float32),
                   c0=np.zeros(k, np.uint32), c1=np.zeros(k, np.uint32))
        self._check(self.L.mth_fileorder_fetch(self.h, C.byref(n), *[out[x].ctypes.data_as(C.c_void_p) for x in ("tid", "pos", "v0", "v1", "c0", "c1")]))
        return out

    def decoded_batch(self, read_beg, read_end, tid, region_beg, region_end):
        """device-resident batch over reads [read_beg, read_end) of the decoded stream (one contig)"""
        b = DeviceBatch()
        self._check(self.L.mth_decoded_batch(self.h, int(read_beg), int(read_end), int(tid), int(region_beg), int(region_end), C.byref(b.c)))
        return b

    def timing_enable(self, on=True):
        self._check(self.L.mth_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._check(self.L.mth_timing_reset(self.h))

    def timing(self):
        out = {}
        for i in range(self.L.mth_timing_num_kernels()):
            name = self.L.mth_timing_kernel_name(i)
            ms, n = C.c_double(0), C.c_uint64(0)
            self._check(self.L.mth_timing_get(self.h, name, C.byref(ms), C.byref(n)))
            out[name.decode()] = (ms.value, n.value)
        return out


def device_count():
    n = C.c_int(0)
    lib().mth_device_count(C.byref(n))
    return n.value


def allreduce_lpmd(engines):
    """one process, one Engine per GPU (or several per GPU): one RCCL all-reduce of the four LPMD counters between the
    distinct devices; afterwards every engine's lpmd_global() returns the totals"""
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    rc = lib().mth_allreduce_lpmd(arr, len(engines))
    if rc != 0:
        engines[0]._check(rc)
