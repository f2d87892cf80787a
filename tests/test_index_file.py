"""CPU tests of the .ibf codec (host/index_file.cpp): the reference's own legacy fixture, the
written variant, and hand-assembled files in the other plausible seqan::hibf layouts."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import random_words


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


def test_reads_the_reference_fixture(host, oracle):
    ix = host.IndexFile.load(os.path.join(GOLDEN, "ibf_idx.ibf"))
    d = ix.describe()
    assert (d["k"], d["molecule"], d["is_hibf"], d["bins"], d["hash_count"]) == (3, "na", False, 2, 3)
    assert d["format"] == "legacy-seqan3"
    assert d["ibfs"][0] == dict(bins=2, tech_bins=64, bin_size=64, hash_shift=57, bin_words=1, hash_funs=3)
    assert [os.path.basename(p) for p in d["paths"]] == ["file1.fa", "file2.fa"]
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    assert np.array_equal(ix.words(), fx["words"])


def test_ibf_round_trip(host):
    for bins, m, dna, k, red in ((5, 106, True, 3, 0), (1024, 777, False, 4, 0), (130, 64, False, 5, 1), (64, 1, False, 12, 2)):
        words = random_words(bins, m, 0.3, bins)
        paths = ["/data/bin_%04d.fa.gz" % i for i in range(bins)]
        a = host.IndexFile.from_ibf(k, dna, red, 3, bins, m, words, paths)
        blob = a.serialise()
        b = host.IndexFile.parse(blob)
        d = b.describe()
        assert (d["k"], d["molecule"], d["reduction"], d["bins"], d["paths"]) == (k, "na" if dna else "aa", red, bins, paths)
        assert d["ibfs"][0]["bin_size"] == m and d["ibfs"][0]["hash_shift"] == 64 - m.bit_length()
        assert np.array_equal(b.words(), words)
        assert b.serialise() == blob
        # index_params header (load_params): u8 k | str molecule | u8 is_hibf
        assert blob[0] == k and blob[1:9] == struct.pack("<Q", 2) and blob[9:11] in (b"na", b"aa") and blob[11] == 0


def _s(x):
    return struct.pack("<Q", len(x)) + x


def _vs(xs):
    return struct.pack("<Q", len(xs)) + b"".join(_s(x) for x in xs)


def _hibf_ibf(f, variant):
    out = b""
    if variant["ibfver"]:
        out += struct.pack("<I", 1)
    out += struct.pack("<6Q", f["bins"], f["tech"], f["m"], 64 - f["m"].bit_length(), f["tech"] // 64, f["h"])
    bits = f["tech"] * f["m"]
    words = f["words"].tobytes()
    nw = len(words) // 8
    if variant["pad"]:
        words += b"\0" * (((nw + 7) // 8 * 8 - nw) * 8)
        nw = (nw + 7) // 8 * 8
    occ = struct.pack("<Q", f["bins"]) + struct.pack("<%dQ" % f["bins"], *range(f["bins"])) + b"\1"
    if variant["occ"] == 2:
        out += occ
    if variant["bv"] == 0:
        out += struct.pack("<Q", bits) + words
    elif variant["bv"] == 1:
        out += struct.pack("<Q", nw) + words + struct.pack("<Q", bits)
    else:
        out += struct.pack("<QQ", bits, nw) + words
    if variant["occ"] == 1:
        out += occ
    return out


def _decomposer(k, dna):
    if dna:
        return struct.pack("<6BQ", k, 2, 3, k, 0, 2 * k - 2, (1 << (2 * k)) - 1)
    return struct.pack("<6BQ", k, 5, 31, k, 0, 20, (1 << (5 * k)) - 1) + bytes(512)


@pytest.mark.parametrize("variant", [
    dict(ibfver=1, bv=0, pad=0, occ=0), dict(ibfver=0, bv=0, pad=0, occ=0), dict(ibfver=1, bv=1, pad=0, occ=0),
    dict(ibfver=1, bv=2, pad=0, occ=0), dict(ibfver=1, bv=0, pad=0, occ=1), dict(ibfver=1, bv=0, pad=0, occ=2),
    dict(ibfver=1, bv=0, pad=1, occ=1), dict(ibfver=0, bv=1, pad=1, occ=2),
])
def test_reader_accepts_the_plausible_hibf_library_layouts(host, variant):
    bins, m, k = 100, 333, 4
    words = random_words(bins, m, 0.4, 1)
    f = dict(bins=bins, tech=128, m=m, h=3, words=words)
    paths = [b"/p/%d.fa" % i for i in range(bins)]
    blob = bytes([k]) + _s(b"aa") + b"\0" + _vs(paths) + b"\0" + struct.pack("<QQB", bins, 0, 3) + _vs(paths)
    blob += _hibf_ibf(f, variant) + _decomposer(k, False)
    ix = host.IndexFile.parse(blob)
    d = ix.describe()
    assert d["bins"] == bins and not d["is_hibf"] and d["ibfs"][0]["bin_size"] == m
    assert np.array_equal(ix.words(), words)


@pytest.mark.parametrize("prev", [0, 1])
def test_reader_parses_an_hibf(host, prev):
    k, user_bins = 5, 6
    variant = dict(ibfver=1, bv=0, pad=0, occ=0)
    root = dict(bins=3, tech=64, m=50, h=2, words=random_words(3, 50, 0.5, 2))
    child = dict(bins=4, tech=64, m=20, h=2, words=random_words(4, 20, 0.5, 3))
    MERGED = 0xFFFFFFFFFFFFFFFF
    nxt = [[0, 1, 0], [1, 1, 1, 1]]
    tbu = [[0, MERGED, 1], [2, 3, 4, 5]]
    paths = [b"/p/%d.fa" % i for i in range(user_bins)]
    blob = bytes([k]) + _s(b"aa") + b"\1" + _vs(paths) + b"\0" + struct.pack("<QfB", user_bins, 0.05, 2) + _vs(paths)
    blob += struct.pack("<I", 1) + struct.pack("<QQ", user_bins, 2) + _hibf_ibf(root, variant) + _hibf_ibf(child, variant)
    blob += struct.pack("<Q", 2) + b"".join(struct.pack("<Q", len(v)) + struct.pack("<%dQ" % len(v), *v) for v in nxt)
    if prev:
        blob += struct.pack("<Q", 2) + struct.pack("<4Q", 0, 0, 0, 1)
    blob += struct.pack("<Q", 2) + b"".join(struct.pack("<Q", len(v)) + struct.pack("<%dQ" % len(v), *v) for v in tbu)
    blob += _decomposer(k, False)
    ix = host.IndexFile.parse(blob)
    d = ix.describe()
    assert d["is_hibf"] and d["bins"] == user_bins and len(d["ibfs"]) == 2
    assert np.array_equal(ix.words(1), child["words"])
    a, b = ix.maps(0)
    assert list(b) == tbu[0] and int(a[1]) == 1
    again = host.IndexFile.parse(ix.serialise())
    assert again.describe()["ibfs"] == d["ibfs"] and np.array_equal(again.words(0), root["words"])


def test_reader_rejects_garbage(host):
    for blob in (b"", b"\x03", b"not an index at all" * 10, bytes(1000)):
        with pytest.raises(host.HostError):
            host.IndexFile.parse(blob)
    good = host.IndexFile.from_ibf(4, False, 0, 3, 10, 20, random_words(10, 20, 0.5, 1), ["p%d" % i for i in range(10)]).serialise()
    for cut in (len(good) - 1, len(good) // 2, 30):
        with pytest.raises(host.HostError):
            host.IndexFile.parse(good[:cut])
    with pytest.raises(host.HostError):
        host.IndexFile.parse(good + b"\0")


def test_mapped_load_equals_parsed_bytes(tmp_path, host):
    """`tetrex query` maps the index file instead of reading it (read_index_file): same description, same words, and the
    image can be serialised again — for a flat IBF, an HIBF and the legacy fixture."""
    import os
    from conftest import GOLDEN
    rng = np.random.default_rng(8)
    words = rng.integers(0, 1 << 63, size=37 * 2, dtype=np.uint64)
    flat = host.IndexFile.from_ibf(5, False, 1, 3, 100, 37, words, ["bin%d.fa" % i for i in range(100)])
    p = tmp_path / "flat.ibf"
    flat.save(p)
    for path in (str(p), os.path.join(GOLDEN, "ibf_idx.ibf")):
        a = host.IndexFile.load(path)
        b = host.IndexFile.parse(open(path, "rb").read())
        assert a.describe() == b.describe()
        assert np.array_equal(a.words(), b.words())
        assert a.serialise() == b.serialise()
    with pytest.raises(host.HostError):
        host.IndexFile.load(str(tmp_path / "missing.ibf"))
    (tmp_path / "empty.ibf").write_bytes(b"")
    with pytest.raises(host.HostError):
        host.IndexFile.load(str(tmp_path / "empty.ibf"))
