"""ctypes view of the C++ host front-end (include/txh.h, tetrex_amd/libtetrex_host.so)."""
import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TETREX_HOST_LIB: load another build of the same library (e.g. an AddressSanitizer build on the CPU)
LIB_PATH = os.environ.get("TETREX_HOST_LIB") or os.path.join(_HERE, "libtetrex_host.so")
_LIB = None
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


class HostError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `make`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.txh_last_error.restype = C.c_char_p
        L.txh_translate.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.txh_preprocess.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.c_uint, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.txh_kgraph.argtypes = [C.c_char_p, C.c_uint, C.c_int, i32p, i32p, i32p, C.c_int32]
        L.txh_compile_batch.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint64,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.txh_blob_data.restype = C.c_void_p
        L.txh_blob_data.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.txh_blob_stats.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.txh_blob_free.argtypes = [C.c_void_p]
        L.txh_record_values.restype = C.c_int64
        L.txh_record_values.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_char_p, C.c_size_t, C.c_int, u64p, C.c_size_t]
        _LIB = L
    return _LIB


def _err():
    return HostError(lib().txh_last_error().decode(errors="replace"))


def translate(rx):
    buf = C.create_string_buffer(1 << 16)
    if lib().txh_translate(rx.encode(), buf, len(buf)) < 0:
        raise _err()
    return buf.value.decode()


def preprocess(rx, dna, k, reduction=0):
    a = C.create_string_buffer(1 << 16)
    b = C.create_string_buffer(1 << 16)
    if lib().txh_preprocess(rx.encode(), int(dna), k, reduction, a, len(a), b, len(b)) < 0:
        raise _err()
    return a.value.decode(), b.value.decode()


def kgraph(postfix, k, reduced=False):
    cap = 1 << 18
    lab, na, nb = (C.c_int32 * cap)(), (C.c_int32 * cap)(), (C.c_int32 * cap)()
    n = lib().txh_kgraph(postfix.encode(), k, int(reduced), lab, na, nb, cap)
    if n < 0:
        raise _err()
    return dict(labels=list(lab[:n]), succ=list(zip(na[:n], nb[:n])))


def kgraph_fused(postfix, k):
    """The k-graph the expansion works on: unions of single residues fused into class nodes (include/txh.h)."""
    L = lib()
    i32p = C.POINTER(C.c_int32)
    L.txh_kgraph_fused.argtypes = [C.c_char_p, C.c_uint, i32p, i32p, i32p, C.c_int32, C.c_char_p, C.c_size_t]
    cap = 1 << 18
    lab, na, nb = (C.c_int32 * cap)(), (C.c_int32 * cap)(), (C.c_int32 * cap)()
    buf = C.create_string_buffer(1 << 20)
    n = L.txh_kgraph_fused(postfix.encode(), k, lab, na, nb, cap, buf, len(buf))
    if n < 0:
        raise _err()
    return dict(labels=list(lab[:n]), succ=list(zip(na[:n], nb[:n])), members=buf.value.decode().split("\n")[:n])


def kgraph_dot(postfix, k, reduced=False, augment=False):
    L = lib()
    L.txh_kgraph_dot.argtypes = [C.c_char_p, C.c_uint, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 22)
    if L.txh_kgraph_dot(postfix.encode(), k, int(reduced), int(augment), buf, len(buf)) < 0:
        raise _err()
    return buf.value.decode()


def compile_batch(regexes, dna, k, reduction, bins):
    """Returns (blob bytes, status list, stats array [n,4] = ops, slots, states, probe ops)."""
    n = len(regexes)
    arr = (C.c_char_p * n)(*[r.encode() for r in regexes])
    status = (C.c_int * n)()
    h = C.c_void_p()
    rc = lib().txh_compile_batch(arr, n, int(dna), k, reduction, bins, C.byref(h), status)
    if rc < 0:
        raise _err()
    try:
        size = C.c_size_t()
        p = lib().txh_blob_data(h, C.byref(size))
        blob = C.string_at(p, size.value)
        stats = np.zeros((n, 4), dtype=np.uint64)
        if n and lib().txh_blob_stats(h, stats.ctypes.data_as(u64p), n) != 0:
            raise _err()
    finally:
        lib().txh_blob_free(h)
    return blob, list(status), stats


STAGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t,
                       C.POINTER(C.c_uint8))


class GapOptions(C.Structure):
    _fields_ = [("augment", C.c_int), ("dgram_loaded", C.c_int), ("min_gap", C.c_uint64), ("max_gap", C.c_uint64)]


def dgram_values(seq, min_gap, max_gap):
    L = lib()
    L.txh_dgram_values.restype = C.c_int64
    L.txh_dgram_values.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, u64p, C.c_size_t]
    s = seq.encode() if isinstance(seq, str) else seq
    cap = max(1, len(s) * (max_gap - min_gap + 1))
    out = np.zeros(cap, dtype=np.uint64)
    n = L.txh_dgram_values(s, len(s), min_gap, max_gap, out.ctypes.data_as(u64p), cap)
    return out[:n].copy()


class DenseOptions(C.Structure):
    _fields_ = [("enabled", C.c_int), ("min_states", C.c_uint32), ("sparse_below", C.c_uint32), ("max_blocks", C.c_uint32),
                ("slot_bytes", C.c_uint64), ("pool_bytes", C.c_uint64), ("tracked", C.c_int)]


def run_staged(regexes, dna, k, reduction, bins, stage, ops_per_query_per_stage=0, ops_per_stage=0, gaps=None, dense=None):
    """Drive the C++ staged expansion with a Python executor.

    stage(blob: bytes, query_program: list, query_slot: list) -> iterable of answers: False/0 = the slot has no bit set,
    True/1 = alive, or 1 + floor(log2(bits set)) as txq_session_stage answers (what the expansion reads mask fills from).
    dense: None, or dict(min_states=, sparse_below=, max_blocks=, slot_bytes=, pool_bytes=, tracked=) to switch dense DP
    steps on (the executor then gets version-4 blobs); tracked: 1 = the executor keeps live lists, 2 = every query uses them."""
    L = lib()
    L.txh_run_staged_dense.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_size_t,
                                       C.c_size_t, C.POINTER(GapOptions), C.POINTER(DenseOptions), STAGE_FN, C.c_void_p,
                                       C.POINTER(C.c_int), u64p]
    d = None
    if dense is not None:
        d = DenseOptions(1, dense.get("min_states", 0), dense.get("sparse_below", 0), dense.get("max_blocks", 0),
                         dense.get("slot_bytes", 0), dense.get("pool_bytes", 0), int(dense.get("tracked", 0)))
    g = None
    if gaps is not None:  # dict(augment=, dgram_loaded=, min_gap=, max_gap=)
        g = GapOptions(int(gaps.get("augment", 0)), int(gaps.get("dgram_loaded", 0)), gaps.get("min_gap", 0), gaps.get("max_gap", 0))
    n = len(regexes)
    arr = (C.c_char_p * n)(*[r.encode() for r in regexes])
    status = (C.c_int * n)()
    stats = (C.c_uint64 * 8)()
    err = []

    def cb(user, blob, size, qp, qs, nq, alive):
        try:
            res = stage(C.string_at(blob, size), [qp[i] for i in range(nq)], [qs[i] for i in range(nq)])
            for i, a in enumerate(res):
                alive[i] = min(255, int(a))
            return 0
        except Exception as e:  # noqa: BLE001 - reported through the return code
            err.append(e)
            return -1

    rc = L.txh_run_staged_dense(arr, n, int(dna), k, reduction, bins, ops_per_query_per_stage, ops_per_stage,
                                C.byref(g) if g is not None else None, C.byref(d) if d is not None else None, STAGE_FN(cb), None,
                                status, stats)
    if err:
        raise err[0]
    if rc < 0:
        raise _err()
    keys = ("stages", "ops", "kmers", "states", "pruned", "feedback_queries", "expand_us", "execute_us")
    return list(status), dict(zip(keys, (int(x) for x in stats)))


def join_shard_masks(mask_words, shard_word0, shard_masks):
    """host/compiler.hpp join_shard_masks: shard r = array [n, words_r] holding words [shard_word0[r], +words_r) of every mask."""
    L = lib()
    L.txh_join_shard_masks.argtypes = [C.c_size_t, C.c_uint64, C.c_size_t, u64p, u64p, C.POINTER(u64p), u64p]
    R = len(shard_masks)
    parts = [np.ascontiguousarray(m, dtype=np.uint64) for m in shard_masks]
    n = parts[0].shape[0] if R else 0
    w0 = np.array(shard_word0, dtype=np.uint64)
    ws = np.array([p.shape[1] for p in parts], dtype=np.uint64)
    ptrs = (u64p * R)(*[p.ctypes.data_as(u64p) for p in parts])
    out = np.zeros((n, mask_words), dtype=np.uint64)
    if L.txh_join_shard_masks(n, mask_words, R, w0.ctypes.data_as(u64p), ws.ctypes.data_as(u64p), ptrs, out.ctypes.data_as(u64p)) != 0:
        raise _err()
    return out


def regex_find_all(pattern, text, posix):
    """The verification matcher (host/matcher.hpp): [(start, length), ...] of successive non-overlapping matches."""
    L = lib()
    L.txh_regex_find_all.restype = C.c_int64
    L.txh_regex_find_all.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, u64p, C.c_size_t]
    t = text.encode() if isinstance(text, str) else text
    cap = 2 * (len(t) + 2)
    out = np.zeros(cap, dtype=np.uint64)
    n = L.txh_regex_find_all(pattern.encode(), int(posix), t, len(t), out.ctypes.data_as(u64p), cap)
    if n < 0:
        raise _err()
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def regex_required_literal(pattern, posix=True):
    """The matcher's prefilter string for `pattern` (host/matcher.hpp required_literal)."""
    L = lib()
    L.txh_regex_required_literal.restype = C.c_int64
    L.txh_regex_required_literal.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(4096)
    n = L.txh_regex_required_literal(pattern.encode(), int(posix), buf, 4096)
    if n < 0:
        raise _err()
    return buf.raw[:n].decode()


def record_values(seq, k, dna=True, reduction=0, wraparound=False):
    s = seq.encode() if isinstance(seq, str) else seq
    cap = len(s) + 2
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().txh_record_values(int(dna), k, reduction, s, len(s), int(wraparound), out.ctypes.data_as(u64p), cap)
    return [int(x) for x in out[:n]]


def record_values_array(seq, k, dna=True, reduction=0, wraparound=False):
    """record_values as a uint64 numpy array (no per-value Python objects: for whole sequences)."""
    s = seq.encode() if isinstance(seq, str) else seq
    cap = len(s) + 2
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().txh_record_values(int(dna), k, reduction, s, len(s), int(wraparound), out.ctypes.data_as(u64p), cap)
    return out[:n].copy()


def parse_blob(blob):
    """Decode a txq_program.h blob (version 1, 2 or 4): (kmers uint64[], [(n_slots, ops array [n,4] =
    kmer,dst,a,b)]).  Version-2/4 ops are in level order, which is also a valid sequential order.
    The dense table of a version-4 blob: blob_dense()."""
    magic, ver = struct.unpack_from("<2I", blob, 0)
    assert magic == 0x50515854 and ver in (1, 2, 4)
    if ver == 1:
        _, _, n_prog, n_kmers, n_ops, _, k_off, p_off, o_off = struct.unpack_from("<6I3Q", blob, 0)
        stride = 4
    else:
        _, _, n_prog, n_kmers, n_ops, _, k_off, p_off, o_off, _, _ = struct.unpack_from("<6I5Q", blob, 0)
        stride = 6
    kmers = np.frombuffer(blob, dtype="<u8", count=n_kmers, offset=k_off)
    progs = np.frombuffer(blob, dtype="<u4", count=n_prog * stride, offset=p_off).reshape(n_prog, stride)
    ops = np.frombuffer(blob, dtype="<u4", count=n_ops * 4, offset=o_off).reshape(n_ops, 4)
    out = []
    for row in progs:
        first, cnt, n_slots = int(row[0]), int(row[1]), int(row[2])
        out.append((n_slots, ops[first:first + cnt]))
    return kmers, out


def blob_aux_kmers(blob):
    """Number of trailing k-mer table entries that belong to the auxiliary (d-gram) index."""
    magic, ver = struct.unpack_from("<2I", blob, 0)
    return struct.unpack_from("<6I5Q", blob, 0)[-1] if ver >= 2 else 0


DENSE_OP = 0xFFFFFFFE
DENSE_SLOT_BIT = 0x40000000


def blob_dense(blob):
    """Dense part of a version-4 blob: (params dict(k, bits, alphabet, canonical), table uint32[n, 16] =
    kind, dst, src, r_mask, shape[11], reserved, per-program dense slot counts); None for older versions."""
    magic, ver = struct.unpack_from("<2I", blob, 0)
    if ver != 4:
        return None
    _, _, n_prog, _, _, _, _, p_off, _, _, _, d_off, n_dense, k, bits, alphabet, canonical, _ = struct.unpack_from("<6I5QQ6I", blob, 0)
    table = np.frombuffer(blob, dtype="<u4", count=n_dense * 16, offset=d_off).reshape(n_dense, 16)
    progs = np.frombuffer(blob, dtype="<u4", count=n_prog * 6, offset=p_off).reshape(n_prog, 6)
    return dict(k=k, bits=bits, alphabet=alphabet, canonical=canonical), table, [int(r[5]) for r in progs]


def blob_levels(blob):
    """Level tables of a version-2/4 blob: list of per-program end-index lists."""
    magic, ver = struct.unpack_from("<2I", blob, 0)
    if ver < 2:
        return None
    _, _, n_prog, n_kmers, n_ops, n_lv, k_off, p_off, o_off, l_off, _ = struct.unpack_from("<6I5Q", blob, 0)
    progs = np.frombuffer(blob, dtype="<u4", count=n_prog * 6, offset=p_off).reshape(n_prog, 6)
    lv = np.frombuffer(blob, dtype="<u4", count=n_lv, offset=l_off)
    return [list(int(x) for x in lv[int(r[3]):int(r[3]) + int(r[4])]) for r in progs]


# ---- .ibf index files ------------------------------------------------------------------------
def _index_api():
    L = lib()
    if not hasattr(L, "_index_ready"):
        L.txh_index_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.txh_index_from_ibf.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, u64p, C.c_char_p,
                                         C.POINTER(C.c_void_p)]
        L.txh_index_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.txh_index_words.restype = C.c_int64
        L.txh_index_words.argtypes = [C.c_void_p, C.c_uint64, u64p, C.c_size_t]
        L.txh_index_maps.restype = C.c_int64
        L.txh_index_maps.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p, C.c_size_t]
        L.txh_index_serialise.restype = C.c_void_p
        L.txh_index_serialise.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.txh_index_free.argtypes = [C.c_void_p]
        L._index_ready = True
    return L


class IndexFile:
    """A parsed / constructed TetRex index image (host/index_file.hpp)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def parse(cls, data):
        L = _index_api()
        h = C.c_void_p()
        if L.txh_index_parse(data, len(data), C.byref(h)) != 0:
            raise _err()
        return cls(h.value)

    @classmethod
    def load(cls, path):
        """read_index_file: the file is mapped, its bit matrices are not copied (what `tetrex query` does)."""
        L = _index_api()
        L.txh_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        if L.txh_index_load(str(path).encode(), C.byref(h)) != 0:
            raise _err()
        return cls(h.value)

    @classmethod
    def from_ibf(cls, k, dna, reduction, hash_count, bins, bin_size, words, paths):
        L = _index_api()
        w = np.ascontiguousarray(words, dtype=np.uint64)
        h = C.c_void_p()
        if L.txh_index_from_ibf(k, int(dna), reduction, hash_count, bins, bin_size, w.ctypes.data_as(u64p),
                                "\n".join(paths).encode(), C.byref(h)) != 0:
            raise _err()
        return cls(h.value)

    def describe(self):
        import json
        buf = C.create_string_buffer(1 << 22)
        if _index_api().txh_index_describe(self._h, buf, len(buf)) < 0:
            raise _err()
        return json.loads(buf.value.decode())

    def words(self, ibf_id=0):
        L = _index_api()
        n = L.txh_index_words(self._h, ibf_id, None, 0)
        if n < 0:
            raise _err()
        out = np.zeros(n, dtype=np.uint64)
        L.txh_index_words(self._h, ibf_id, out.ctypes.data_as(u64p), n)
        return out

    def maps(self, ibf_id):
        L = _index_api()
        a = np.zeros(1 << 20, dtype=np.uint64)
        b = np.zeros(1 << 20, dtype=np.uint64)
        n = L.txh_index_maps(self._h, ibf_id, a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), a.size)
        if n < 0:
            raise _err()
        return a[:n].copy(), b[:n].copy()

    def serialise(self):
        size = C.c_size_t()
        p = _index_api().txh_index_serialise(self._h, C.byref(size))
        if not p:
            raise _err()
        return C.string_at(p, size.value)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.serialise())

    def __del__(self):
        try:
            if self._h:
                _index_api().txh_index_free(self._h)
                self._h = None
        except Exception:
            pass
