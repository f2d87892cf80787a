"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (see oracle/txo_capi.cpp)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("TETREX_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # e.g. the ASan build
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.txo_last_error.restype = C.c_char_p
        L.txo_compute_bitcount.restype = C.c_uint64
        L.txo_compute_bitcount.argtypes = [C.c_uint64, C.c_float]
        L.txo_hash_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, u64p]
        L.txo_ibf_new.restype = C.c_void_p
        L.txo_ibf_new.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_int, C.c_uint, C.c_uint]
        L.txo_ibf_set_words.argtypes = [C.c_void_p, u64p, C.c_uint64]
        L.txo_ibf_words.restype = u64p
        L.txo_ibf_words.argtypes = [C.c_void_p, u64p]
        L.txo_ibf_shape.argtypes = [C.c_void_p, u64p]
        L.txo_ibf_emplace.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_uint64]
        L.txo_ibf_emplace_pairs.argtypes = [C.c_void_p, u64p, C.POINTER(C.c_uint32), C.c_uint64]
        L.txo_probe.argtypes = [C.c_void_p, u64p, C.c_uint64, u64p, C.c_int]
        L.txo_hibf_new.restype = C.c_void_p
        L.txo_hibf_new.argtypes = [C.c_uint64, C.c_int, C.c_uint, C.c_uint]
        L.txo_hibf_add_ibf.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint, u64p, u64p, u64p]
        L.txo_hibf_emplace.argtypes = [C.c_void_p, C.c_uint64, u64p, C.c_uint64, C.c_uint64]
        L.txo_hibf_words.restype = u64p
        L.txo_hibf_words.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.txo_index_free.argtypes = [C.c_void_p]
        L.txo_index_bins.restype = C.c_uint64
        L.txo_index_bins.argtypes = [C.c_void_p]
        L.txo_decompose.restype = C.c_int64
        L.txo_decompose.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_char_p, C.c_uint64, C.c_int, u64p, C.c_uint64]
        L.txo_update_kmers.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_char_p, C.c_uint64, u64p, u64p, u64p]
        L.txo_encoder_tables.argtypes = [C.c_uint, C.c_char_p, C.c_char_p]
        for f in (L.txo_translate, L.txo_trim_regex):
            f.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.txo_reduce_alphabet.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_size_t]
        ip = C.POINTER(C.c_int)
        L.txo_kgraph.argtypes = [C.c_char_p, C.c_uint, C.c_int, ip, ip, ip, C.c_int, ip, C.c_int, ip]
        L.txo_query.argtypes = [C.c_void_p, C.c_char_p, u64p, u64p]
        L.txo_query_aug.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, u64p, u64p]
        L.txo_dgram_codes.restype = C.c_int64
        L.txo_dgram_codes.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, u64p, C.c_uint64]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    pass


def _err():
    return OracleError(lib().txo_last_error().decode())


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def compute_bitcount(n, fpr):
    return int(lib().txo_compute_bitcount(n, fpr))


def hash_rows(value, bin_size, h):
    out = np.zeros(5, dtype=np.uint64)
    if lib().txo_hash_rows(value, bin_size, h, out.ctypes.data_as(u64p)) != 0:
        raise _err()
    return [int(x) for x in out[:h]]


class Index:
    """An oracle-side IBF or HIBF plus its k-mer encoder."""

    def __init__(self, handle, is_hibf):
        if not handle:
            raise _err()
        self._h = C.c_void_p(handle)
        self.is_hibf = is_hibf

    @classmethod
    def ibf(cls, bins, bin_size, h, dna=True, k=3, reduction=0):
        return cls(lib().txo_ibf_new(bins, bin_size, h, int(dna), k, reduction), False)

    @classmethod
    def hibf(cls, user_bins, dna=False, k=4, reduction=0):
        return cls(lib().txo_hibf_new(user_bins, int(dna), k, reduction), True)

    def __del__(self):
        try:
            if self._h:
                lib().txo_index_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def bins(self):
        return int(lib().txo_index_bins(self._h))

    @property
    def words_per_mask(self):
        return (self.bins + 63) // 64

    def shape(self):
        out = np.zeros(6, dtype=np.uint64)
        lib().txo_ibf_shape(self._h, out.ctypes.data_as(u64p))
        keys = ("bins", "tech_bins", "bin_size", "hash_shift", "bin_words", "hash_funs")
        return dict(zip(keys, (int(x) for x in out)))

    def set_words(self, words):
        a, p = _u64(words)
        if lib().txo_ibf_set_words(self._h, p, a.size) != 0:
            raise _err()

    def words(self):
        n = C.c_uint64()
        p = lib().txo_ibf_words(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def emplace(self, values, bin_id):
        a, p = _u64(values)
        if lib().txo_ibf_emplace(self._h, p, a.size, bin_id) != 0:
            raise _err()

    def emplace_pairs(self, values, bins_of):
        a, p = _u64(values)
        b = np.ascontiguousarray(bins_of, dtype=np.uint32)
        if lib().txo_ibf_emplace_pairs(self._h, p, b.ctypes.data_as(C.POINTER(C.c_uint32)), a.size) != 0:
            raise _err()

    def probe(self, values, threads=1):
        a, p = _u64(values)
        out = np.zeros((a.size, self.words_per_mask), dtype=np.uint64)
        if lib().txo_probe(self._h, p, a.size, out.ctypes.data_as(u64p), threads) != 0:
            raise _err()
        return out

    # HIBF construction helpers -----------------------------------------------------
    def add_ibf(self, bins, bin_size, h, next_ibf_id, tb_to_user, words=None):
        nx, pn = _u64(next_ibf_id)
        tb, pt = _u64(tb_to_user)
        assert nx.size == bins and tb.size == bins
        pw = None
        if words is not None:
            w, pw = _u64(words)
        r = lib().txo_hibf_add_ibf(self._h, bins, bin_size, h, pw, pn, pt)
        if r < 0:
            raise _err()
        return r

    def hibf_emplace(self, ibf_id, values, tb):
        a, p = _u64(values)
        if lib().txo_hibf_emplace(self._h, ibf_id, p, a.size, tb) != 0:
            raise _err()

    def hibf_words(self, ibf_id):
        n = C.c_uint64()
        p = lib().txo_hibf_words(self._h, ibf_id, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def query_aug(self, regex, augment=True, dgram=None, min_gap=0, max_gap=0):
        """Query with -a (and -g when `dgram`, an oracle flat IBF over d-gram codes, is given)."""
        mask = np.zeros(self.words_per_mask, dtype=np.uint64)
        stats = np.zeros(5, dtype=np.uint64)
        rc = lib().txo_query_aug(self._h, regex.encode(), int(augment), dgram._h if dgram is not None else None,
                                 min_gap, max_gap, mask.ctypes.data_as(u64p), stats.ctypes.data_as(u64p))
        if rc != 0:
            raise _err()
        keys = ("probes", "states", "quirk_merges", "dgram_probes", "gap_nodes")
        return mask, dict(zip(keys, (int(x) for x in stats)))

    def expected_mask(self, regex, augment=False):
        """The mask a correct implementation must return: the reference's (restated) result where that is well defined
        (no quirk merges), otherwise the result under well-defined merges.  Returns (mask, quirk_merges)."""
        if augment:
            mask, st = self.query_aug(regex, augment=True)
        else:
            mask, st = self.query(regex, with_stats=True)
        if st["quirk_merges"]:
            mask = self.query(regex, well_defined=True, augment=augment)
        return mask, st["quirk_merges"]

    def query(self, regex, with_stats=False, well_defined=False, augment=False):
        """well_defined=True: states are keyed by what they are, so the merges that make the reference's result
        implementation-defined (quirk_merges) do not happen — NOT the reference's rule, but the only defined answer for
        those queries; identical to the default wherever quirk_merges == 0."""
        mask = np.zeros(self.words_per_mask, dtype=np.uint64)
        stats = np.zeros(3, dtype=np.uint64)
        L = lib()
        if well_defined:
            L.txo_query_well_defined.argtypes = [C.c_void_p, C.c_char_p, C.c_int, u64p, u64p]
            rc = L.txo_query_well_defined(self._h, regex.encode(), int(augment), mask.ctypes.data_as(u64p), stats.ctypes.data_as(u64p))
        else:
            rc = L.txo_query(self._h, regex.encode(), mask.ctypes.data_as(u64p), stats.ctypes.data_as(u64p))
        if rc != 0:
            raise _err()
        if with_stats:
            return mask, dict(probes=int(stats[0]), states=int(stats[1]), quirk_merges=int(stats[2]))
        return mask


def dgram_codes(seq, min_gap, max_gap):
    s = seq.encode() if isinstance(seq, str) else seq
    cap = max(1, len(s) * (max_gap - min_gap + 1))
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().txo_dgram_codes(s, len(s), min_gap, max_gap, out.ctypes.data_as(u64p), cap)
    return out[:n].copy()


def decompose(seq, k, dna=True, reduction=0, quirk=False):
    s = seq.encode() if isinstance(seq, str) else seq
    cap = max(len(s) + 2, 1)
    out = np.zeros(cap, dtype=np.uint64)
    n = lib().txo_decompose(int(dna), k, reduction, s, len(s), int(quirk), out.ctypes.data_as(u64p), cap)
    return [int(x) for x in out[:n]]


def update_kmers(seq, k, dna=True, reduction=0, start=0):
    s = seq.encode() if isinstance(seq, str) else seq
    fwd = np.zeros(len(s), dtype=np.uint64)
    canon = np.zeros(len(s), dtype=np.uint64)
    st = C.c_uint64(start)
    lib().txo_update_kmers(int(dna), k, reduction, s, len(s), C.byref(st), fwd.ctypes.data_as(u64p), canon.ctypes.data_as(u64p))
    return [int(x) for x in fwd], [int(x) for x in canon]


def encoder_tables(reduction):
    a = C.create_string_buffer(256)
    r = C.create_string_buffer(256)
    lib().txo_encoder_tables(reduction, a, r)
    return bytes(a.raw), bytes(r.raw)


def _str_call(fn, *args):
    buf = C.create_string_buffer(1 << 16)
    n = fn(*args, buf, len(buf))
    if n < 0:
        raise _err()
    return buf.value.decode()


def translate(rx):
    return _str_call(lib().txo_translate, rx.encode())


def trim_regex(rx):
    return _str_call(lib().txo_trim_regex, rx.encode())


def reduce_alphabet(rx, reduction):
    return _str_call(lib().txo_reduce_alphabet, rx.encode(), reduction)


def kgraph(postfix, k, reduced=False):
    cap_n, cap_a = 1 << 16, 1 << 17
    labels = (C.c_int * cap_n)()
    succ = (C.c_int * (2 * cap_n))()
    ranks = (C.c_int * cap_n)()
    arcs = (C.c_int * (2 * cap_a))()
    na = C.c_int()
    n = lib().txo_kgraph(postfix.encode(), k, int(reduced), labels, succ, ranks, cap_n, arcs, cap_a, C.byref(na))
    if n < 0:
        raise _err()
    return dict(
        labels=list(labels[:n]),
        succ=[(succ[2 * i], succ[2 * i + 1]) for i in range(n)],
        ranks=list(ranks[:n]),
        arcs=[(arcs[2 * i], arcs[2 * i + 1]) for i in range(na.value)],
    )


def read_fasta(path):
    """Tiny FASTA reader for the test fixtures (handles files without a trailing newline)."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    recs, name, seq = [], None, []
    with op(path, "rt") as f:
        text = f.read()
    # the reference's toy files put '>' of the next record right after the sequence
    for line in text.replace(">", "\n>").splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:], []
        else:
            seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def read_legacy_fixture(path):
    """Decode the reference's legacy (seqan3/sdsl era) index fixture test/data/ibf_idx.ibf.

    Layout: u64 bin_count | u64 bin_size | u8 h | seqan3 IBF {u64 bins,
    technical_bins, bin_size, hash_shift, bin_words, hash_funs} | sdsl bit_vector
    {u8 width, f32 growth, u64 bits} | data words | u8 k | str molecule | vec<str> paths | ...
    """
    import struct
    d = open(path, "rb").read()
    bin_count, bin_size, h = struct.unpack_from("<QQB", d, 0)
    bins, tb, bs, shift, bw, hf = struct.unpack_from("<6Q", d, 17)
    width, growth, bits = struct.unpack_from("<BfQ", d, 17 + 48)
    off = 17 + 48 + 13
    nwords = (bits + 63) // 64
    words = np.frombuffer(d, dtype="<u8", count=nwords, offset=off).copy()
    off += nwords * 8
    k = d[off]
    off += 1
    (ln,) = struct.unpack_from("<Q", d, off)
    mol = d[off + 8: off + 8 + ln].decode()
    return dict(bin_count=bin_count, bin_size=bin_size, h=h, bins=bins, tech_bins=tb, hash_shift=shift,
                bin_words=bw, hash_funs=hf, bits=bits, words=words, k=k, molecule=mol)
