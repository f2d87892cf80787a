"""ctypes view of the C++ host front-end (include/txh.h, tetrex_amd/libtetrex_host.so)."""
import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtetrex_host.so")
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


def record_values(seq, k, dna=True, reduction=0, wraparound=False):
    s = seq.encode() if isinstance(seq, str) else seq
    cap = len(s) + 2
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().txh_record_values(int(dna), k, reduction, s, len(s), int(wraparound), out.ctypes.data_as(u64p), cap)
    return [int(x) for x in out[:n]]


def parse_blob(blob):
    """Decode a txq_program.h blob: (kmers uint64[], [(n_slots, ops array [n,4] = kmer,dst,a,b)])."""
    magic, ver, n_prog, n_kmers, n_ops, _, k_off, p_off, o_off = struct.unpack_from("<6I3Q", blob, 0)
    assert magic == 0x50515854 and ver == 1
    kmers = np.frombuffer(blob, dtype="<u8", count=n_kmers, offset=k_off)
    progs = np.frombuffer(blob, dtype="<u4", count=n_prog * 4, offset=p_off).reshape(n_prog, 4)
    ops = np.frombuffer(blob, dtype="<u4", count=n_ops * 4, offset=o_off).reshape(n_ops, 4)
    out = []
    for first, cnt, n_slots, _ in progs:
        out.append((int(n_slots), ops[first:first + cnt]))
    return kmers, out
