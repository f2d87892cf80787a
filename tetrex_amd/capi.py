"""ctypes view of the C-ABI declared in include/txq.h (tetrex_amd/libtxq.so).

Mirrors the reference's seam for the probe path (see include/txq.h for the file:line of
each replaced interface).  There is deliberately no fallback: a missing library or a missing
GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtxq.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)

TXQ_MERGED_BIN = 0xFFFFFFFFFFFFFFFF

# every symbol include/txq.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "txq_init", "txq_shutdown", "txq_last_error", "txq_device_count",
    "txq_index_upload", "txq_index_upload_subtrees", "txq_index_get_info", "txq_index_free", "txq_index_supports_dense", "txq_index_memory", "txq_index_set_tag", "txq_index_get_tag", "txq_index_create_ibf",
    "txq_index_download_words", "txq_probe", "txq_probe_device", "txq_emplace_device",
    "txq_run_programs", "txq_run_programs_device", "txq_session_begin", "txq_session_set_aux_index", "txq_session_stage", "txq_session_end",
    "txq_malloc", "txq_free", "txq_memcpy_h2d", "txq_memcpy_d2h", "txq_synchronize", "txq_host_alloc", "txq_host_free",
]


class TxqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("txq error %d: %s" % (code, msg))
        self.code = code


class IbfDesc(C.Structure):
    _fields_ = [("bins", C.c_uint64), ("tech_bins", C.c_uint64), ("bin_size", C.c_uint64),
                ("hash_shift", C.c_uint64), ("bin_words", C.c_uint64), ("hash_funs", C.c_uint64),
                ("words", u64p)]


class IndexDesc(C.Structure):
    _fields_ = [("n_ibf", C.c_uint64), ("ibf", C.POINTER(IbfDesc)),
                ("next_ibf_id", C.POINTER(u64p)), ("tb_to_user_bin", C.POINTER(u64p)),
                ("user_bins", C.c_uint64)]


class IndexInfo(C.Structure):
    _fields_ = [("user_bins", C.c_uint64), ("mask_words", C.c_uint64), ("shard_word0", C.c_uint64),
                ("shard_words", C.c_uint64), ("n_ibf", C.c_uint64), ("device_bytes", C.c_uint64),
                ("is_hibf", C.c_int), ("device", C.c_int), ("join_or", C.c_int), ("shard_rank", C.c_int), ("n_shards", C.c_int),
                ("reserved", C.c_int)]


_LIB = None


def lib():
    """Load libtxq.so; raises if it has not been built (there is no fallback path)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make` (or __graft_entry__.build()); "
                              "tetrex_amd has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.txq_last_error.restype = C.c_char_p
        L.txq_init.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.txq_index_upload.argtypes = [C.POINTER(IndexDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.txq_index_upload_subtrees.argtypes = [C.POINTER(IndexDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.txq_index_get_info.argtypes = [C.c_void_p, C.POINTER(IndexInfo)]
        L.txq_index_free.argtypes = [C.c_void_p]
        L.txq_index_create_ibf.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.txq_index_download_words.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.txq_probe.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
        L.txq_probe_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.txq_emplace_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.txq_run_programs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, u64p]
        L.txq_run_programs_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.txq_session_begin.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.txq_session_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, u32p, u32p, C.c_size_t, C.POINTER(C.c_uint8)]
        L.txq_session_end.argtypes = [C.c_void_p, u64p]
        L.txq_session_set_aux_index.argtypes = [C.c_void_p, C.c_void_p]
        L.txq_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.txq_free.argtypes = [C.c_void_p]
        L.txq_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.txq_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.txq_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.txq_host_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


_QLIB = None


def _query_lib():
    global _QLIB
    if _QLIB is None:
        lib()  # libtxq.so first, so the query library binds to the same copy
        path = os.path.join(_HERE, "libtetrex_query.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: run `make`" % path)
        L = C.CDLL(path)
        L.txe_last_error.restype = C.c_char_p
        L.txe_query_masks.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t,
                                      u64p, C.POINTER(C.c_int), u64p]
        L.txe_query_masks_text.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                           u64p, C.POINTER(C.c_int), u64p]
        _QLIB = L
    return _QLIB


def check(rc):
    if rc != 0:
        raise TxqError(rc, lib().txq_last_error().decode(errors="replace"))


def init(device=0):
    dev = C.c_int(device)
    check(lib().txq_init(1, C.byref(dev)))


def init_devices(devices):
    """One process, several GPUs (txq_init(N, ids)): shard r of an index lives on devices[r % N]."""
    ids = (C.c_int * len(devices))(*[int(d) for d in devices])
    check(lib().txq_init(len(devices), ids))


def shutdown():
    check(lib().txq_shutdown())


def device_count():
    n = lib().txq_device_count()
    if n < 0:
        check(n)
    return n


def device_count_safe():
    """Number of visible GPUs, 0 when the HIP runtime reports none (or errors)."""
    n = lib().txq_device_count()
    return n if n > 0 else 0


def synchronize():
    check(lib().txq_synchronize())


class HostBuffer:
    """Page-locked host memory from txq_host_alloc, viewed as a numpy array."""

    def __init__(self, shape, dtype=np.uint64):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(lib().txq_host_alloc(C.byref(p), self.nbytes))
        self.ptr = p.value
        self.array = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            check(lib().txq_host_free(self.ptr))
            self.ptr = None


class DeviceBuffer:
    """A raw HBM allocation owned through txq_malloc/txq_free."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().txq_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            check(lib().txq_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes))
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            check(lib().txq_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().txq_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ibf_desc(bins, bin_size, hash_funs, words):
    d = IbfDesc()
    d.bins = bins
    d.bin_words = (bins + 63) // 64
    d.tech_bins = d.bin_words * 64
    d.bin_size = bin_size
    d.hash_shift = 64 - int(bin_size).bit_length()
    d.hash_funs = hash_funs
    d.words = words.ctypes.data_as(u64p) if words is not None else None
    return d


class Index:
    """An (H)IBF resident in HBM (txq_index)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        info = IndexInfo()
        check(lib().txq_index_get_info(self._h, C.byref(info)))
        self.info = info

    # -- construction ---------------------------------------------------------------
    @classmethod
    def upload_ibf(cls, bins, bin_size, hash_funs, words, shard_rank=0, n_shards=1):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        d = _ibf_desc(bins, bin_size, hash_funs, words)
        assert words.size == bin_size * d.bin_words
        desc = IndexDesc(1, C.pointer(d), None, None, bins)
        h = C.c_void_p()
        check(lib().txq_index_upload(C.byref(desc), shard_rank, n_shards, C.byref(h)))
        return cls(h.value)

    @classmethod
    def upload_hibf(cls, user_bins, ibfs, shard_rank=0, n_shards=1, subtrees=False):
        """ibfs: list of dicts {bins, bin_size, hash_funs, words, next_ibf_id, tb_to_user}.  subtrees: txq_index_upload_subtrees
        (a general tree is sharded by sub-trees: full-width masks that are ORed, info.join_or == 1)."""
        n = len(ibfs)
        keep = []
        descs = (IbfDesc * n)()
        nxt = (u64p * n)()
        tbu = (u64p * n)()
        for i, f in enumerate(ibfs):
            w = np.ascontiguousarray(f["words"], dtype=np.uint64)
            a = np.ascontiguousarray(f["next_ibf_id"], dtype=np.uint64)
            b = np.ascontiguousarray(f["tb_to_user"], dtype=np.uint64)
            keep += [w, a, b]
            descs[i] = _ibf_desc(f["bins"], f["bin_size"], f["hash_funs"], w)
            nxt[i] = a.ctypes.data_as(u64p)
            tbu[i] = b.ctypes.data_as(u64p)
        desc = IndexDesc(n, descs, nxt, tbu, user_bins)
        h = C.c_void_p()
        check((lib().txq_index_upload_subtrees if subtrees else lib().txq_index_upload)(C.byref(desc), shard_rank, n_shards, C.byref(h)))
        return cls(h.value)

    @classmethod
    def create_ibf(cls, bins, bin_size, hash_funs, shard_rank=0, n_shards=1):
        h = C.c_void_p()
        check(lib().txq_index_create_ibf(bins, bin_size, hash_funs, shard_rank, n_shards, C.byref(h)))
        return cls(h.value)

    def free(self):
        if self._h:
            lib().txq_index_free(self._h)
            self._h = None

    @property
    def tag(self):
        """txq_index_get_tag: the host layer's note on the index (bits 0-1: 0 nothing known, 1 state lists saturate on it —
        dense steps pay — 2 states thin out)."""
        L = lib()
        L.txq_index_get_tag.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        v = C.c_uint64()
        check(L.txq_index_get_tag(self._h, C.byref(v)))
        return int(v.value)

    @tag.setter
    def tag(self, value):
        L = lib()
        L.txq_index_set_tag.argtypes = [C.c_void_p, C.c_uint64]
        check(L.txq_index_set_tag(self._h, int(value)))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # -- queries ---------------------------------------------------------------------
    @property
    def shard_words(self):
        return int(self.info.shard_words)

    def probe(self, kmers, out=None):
        """Host-buffer batched bulk_contains: (n, shard_words) uint64.  `out` may be a preallocated
        C-contiguous uint64 array of that shape (e.g. HostBuffer.array: pinned, no bounce copy)."""
        k = np.ascontiguousarray(kmers, dtype=np.uint64)
        if out is None:
            out = np.zeros((k.size, self.shard_words), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.flags.c_contiguous and out.size == k.size * self.shard_words
        check(lib().txq_probe(self._h, k.ctypes.data_as(u64p), k.size, out.ctypes.data_as(u64p)))
        return out

    def probe_device(self, d_kmers, n, d_masks, d_alive=None, stream=None):
        check(lib().txq_probe_device(self._h, d_kmers, n, d_masks, d_alive, stream))

    def emplace_device(self, d_values, d_bins_of, n, stream=None):
        check(lib().txq_emplace_device(self._h, d_values, d_bins_of, n, stream))

    def download_words_rows(self, bin_size):
        """The shard's bit matrix, row-major [bin_size][shard_words] (flat IBF only)."""
        out = np.zeros(bin_size * self.shard_words, dtype=np.uint64)
        check(lib().txq_index_download_words(self._h, out.ctypes.data_as(u64p), out.size))
        return out

    def session(self, n_programs):
        return Session(self, n_programs)

    def query_masks_gapped(self, regexes, dna, k, reduction=0, augment=True, dgram=None, min_gap=0, max_gap=0,
                           ops_per_query_per_stage=0):
        """query_masks with -a (augment) and, when `dgram` (a GPU-resident flat IBF over d-gram codes)
        is given, -g."""
        from .host import GapOptions
        Lq = _query_lib()
        Lq.txe_query_masks_gapped.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(GapOptions), C.c_int, C.c_uint, C.c_uint,
                                              C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t, u64p, C.POINTER(C.c_int), u64p]
        n = len(regexes)
        arr = (C.c_char_p * n)(*[r.encode() for r in regexes])
        masks = np.zeros((n, self.shard_words), dtype=np.uint64)
        status = (C.c_int * n)()
        stats = (C.c_uint64 * 8)()
        g = GapOptions(int(augment), int(dgram is not None), min_gap, max_gap)
        rc = Lq.txe_query_masks_gapped(self._h, dgram._h if dgram is not None else None, C.byref(g), int(dna), k, reduction,
                                       arr, n, ops_per_query_per_stage, masks.ctypes.data_as(u64p), status, stats)
        if rc < 0:
            raise TxqError(rc, Lq.txe_last_error().decode(errors="replace"))
        keys = ("stages", "ops", "kmers", "states", "pruned", "feedback_queries", "expand_us", "execute_us")
        Lq.txe_last_dense_ops.restype = C.c_uint64
        Lq.txe_last_tracked_queries.restype = C.c_uint64
        return masks, list(status), dict(zip(keys, (int(x) for x in stats)), dense_ops=int(Lq.txe_last_dense_ops()), tracked_queries=int(Lq.txe_last_tracked_queries()))

    def query_masks(self, regexes, dna, k, reduction=0, ops_per_query_per_stage=0):
        """Whole queries on this GPU-resident index through the C++ host (libtetrex_query.so):
        returns (masks [n, shard_words], status list, stats dict)."""
        Lq = _query_lib()
        n = len(regexes)
        masks = np.zeros((n, self.shard_words), dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        stats = (C.c_uint64 * 8)()
        text = "\n".join(regexes)
        if n and text.count("\n") == n - 1:  # one text, one motif per line: no array of n C strings to build
            raw = text.encode()
            rc = Lq.txe_query_masks_text(self._h, int(dna), k, reduction, raw, len(raw), n, ops_per_query_per_stage,
                                         masks.ctypes.data_as(u64p), status.ctypes.data_as(C.POINTER(C.c_int)), stats)
        else:
            arr = (C.c_char_p * n)(*[r.encode() for r in regexes])
            rc = Lq.txe_query_masks(self._h, int(dna), k, reduction, arr, n, ops_per_query_per_stage,
                                    masks.ctypes.data_as(u64p), status.ctypes.data_as(C.POINTER(C.c_int)), stats)
        if rc < 0:
            raise TxqError(rc, Lq.txe_last_error().decode(errors="replace"))
        keys = ("stages", "ops", "kmers", "states", "pruned", "feedback_queries", "expand_us", "execute_us")
        Lq.txe_last_dense_ops.restype = C.c_uint64
        Lq.txe_last_tracked_queries.restype = C.c_uint64
        return masks, status.tolist(), dict(zip(keys, (int(x) for x in stats)), dense_ops=int(Lq.txe_last_dense_ops()), tracked_queries=int(Lq.txe_last_tracked_queries()))

    def supports_dense(self):
        """txq_index_supports_dense: 0 no dense steps, 1 through the descent, 2 fused (tracked programs, too)."""
        return int(lib().txq_index_supports_dense(self._h))

    def run_programs(self, blob, n_programs):
        buf = np.frombuffer(blob, dtype=np.uint8)
        # 8-byte aligned copy
        al = np.zeros((buf.size + 7) // 8, dtype=np.uint64)
        al.view(np.uint8)[:buf.size] = buf
        out = np.zeros((n_programs, self.shard_words), dtype=np.uint64)
        check(lib().txq_run_programs(self._h, al.ctypes.data, buf.size, n_programs, out.ctypes.data_as(u64p)))
        return out


def query_masks_sharded(shards, regexes, dna, k, reduction=0, ops_per_query_per_stage=0):
    """Whole queries on ALL column shards of an index at once (libtetrex_query.so txe_query_masks_sharded): one
    frontier expansion drives every shard, the final masks are joined.  shards: Index objects, shard r of
    len(shards).  Returns (full masks [n, mask_words], status list, stats dict)."""
    Lq = _query_lib()
    Lq.txe_query_masks_sharded.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_int, C.c_uint, C.c_uint,
                                           C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t, u64p, C.POINTER(C.c_int), u64p]
    n = len(regexes)
    arr = (C.c_char_p * n)(*[r.encode() for r in regexes])
    hs = (C.c_void_p * len(shards))(*[s._h for s in shards])
    masks = np.zeros((n, int(shards[0].info.mask_words)), dtype=np.uint64)
    status = (C.c_int * n)()
    stats = (C.c_uint64 * 8)()
    rc = Lq.txe_query_masks_sharded(hs, None, len(shards), None, int(dna), k, reduction, arr, n, ops_per_query_per_stage,
                                    masks.ctypes.data_as(u64p), status, stats)
    if rc < 0:
        raise TxqError(rc, Lq.txe_last_error().decode(errors="replace"))
    keys = ("stages", "ops", "kmers", "states", "pruned", "feedback_queries", "expand_us", "execute_us")
    Lq.txe_last_dense_ops.restype = C.c_uint64
    Lq.txe_last_tracked_queries.restype = C.c_uint64
    return masks, list(status), dict(zip(keys, (int(x) for x in stats)), dense_ops=int(Lq.txe_last_dense_ops()), tracked_queries=int(Lq.txe_last_tracked_queries()))


def _aligned(blob):
    buf = np.frombuffer(blob, dtype=np.uint8)
    al = np.zeros((buf.size + 7) // 8, dtype=np.uint64)
    al.view(np.uint8)[:buf.size] = buf
    return al, buf.size


class Session:
    """Staged execution of a batch of programs (txq_session_*)."""

    def __init__(self, index, n_programs):
        self.index = index
        self.n = n_programs
        h = C.c_void_p()
        check(lib().txq_session_begin(index._h, n_programs, C.byref(h)))
        self._h = h

    def set_aux_index(self, aux):
        self._aux = aux  # keep it alive
        check(lib().txq_session_set_aux_index(self._h, aux._h if aux is not None else None))

    def stage(self, blob, query_program=(), query_slot=()):
        al, size = _aligned(blob)
        qp = np.ascontiguousarray(query_program, dtype=np.uint32)
        qs = np.ascontiguousarray(query_slot, dtype=np.uint32)
        alive = np.zeros(max(qp.size, 1), dtype=np.uint8)
        check(lib().txq_session_stage(self._h, al.ctypes.data, size, qp.ctypes.data_as(u32p), qs.ctypes.data_as(u32p),
                                      qp.size, alive.ctypes.data_as(C.POINTER(C.c_uint8))))
        return alive[:qp.size].astype(bool)

    def end(self):
        out = np.zeros((self.n, self.index.shard_words), dtype=np.uint64)
        h, self._h = self._h, None
        check(lib().txq_session_end(h, out.ctypes.data_as(u64p)))
        return out

    def __del__(self):
        try:
            if self._h:
                lib().txq_session_end(self._h, None)
                self._h = None
        except Exception:
            pass
