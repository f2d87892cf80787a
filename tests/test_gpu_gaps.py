"""-a / -g on the GPU: augmented queries through libtetrex_query + a device-resident d-gram index
(the session's auxiliary index), and the `tetrex track` / `tetrex query -a -g` command line."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_host_gaps import GAPPED, PLANTED, AA

pytestmark = pytest.mark.gpu
TETREX = os.path.join(ROOT, "bin", "tetrex")


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def library(oracle):
    rng = np.random.default_rng(0)
    bins = 200
    seqs = ["".join(rng.choice(list(AA), size=400)) for _ in range(bins)]
    for b in (5, 40, 177):
        at = int(rng.integers(20, 300))
        seqs[b] = seqs[b][:at] + PLANTED + seqs[b][at + len(PLANTED):]
    ox = oracle.Index.ibf(bins, oracle.compute_bitcount(400, 0.05), 3, dna=False, k=4)
    for b, s in enumerate(seqs):
        ox.emplace(oracle.decompose(s, 4, dna=False), b)
    codes = [oracle.dgram_codes(s, 1, 10) for s in seqs]
    dg = oracle.Index.ibf(bins, oracle.compute_bitcount(max(len(c) for c in codes), 0.05), 3, dna=False, k=4)
    for b, c in enumerate(codes):
        dg.emplace(c, b)
    return dict(bins=bins, seqs=seqs, ox=ox, dg=dg)


@pytest.mark.parametrize("shards", [1, 2])
def test_augmented_queries_on_device(capi, oracle, library, shards):
    ox, dg, bins = library["ox"], library["dg"], library["bins"]
    so, sd = ox.shape(), dg.shape()
    for r in range(shards):
        ix = capi.Index.upload_ibf(bins, so["bin_size"], 3, ox.words(), shard_rank=r, n_shards=shards)
        dx = capi.Index.upload_ibf(bins, sd["bin_size"], 3, dg.words(), shard_rank=r, n_shards=shards)
        lo, nw = int(ix.info.shard_word0), ix.shard_words
        for aux, mn, mx in ((None, 0, 0), (dx, 1, 10)):
            got, status, stats = ix.query_masks_gapped(GAPPED, False, 4, augment=True, dgram=aux, min_gap=mn, max_gap=mx)
            assert not any(status)
            for i, q in enumerate(GAPPED):
                want, st = ox.query_aug(q, True, dg if aux is not None else None, mn, mx)
                if st["quirk_merges"] == 0:
                    assert np.array_equal(got[i], want[lo:lo + nw]), (q, aux is not None, r)
        # an auxiliary index over other bins / another shard is refused
        other = capi.Index.upload_ibf(64, 8, 2, np.zeros(8, dtype=np.uint64))
        sess = ix.session(1)
        with pytest.raises(capi.TxqError):
            sess.set_aux_index(other)
        del sess  # a session that never ran a stage has no result to fetch
        other.free(); dx.free(); ix.free()


def _run(*args, cwd=None):
    r = subprocess.run([TETREX, *args], capture_output=True, text=True, cwd=cwd, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_track_and_query_with_gaps_on_the_command_line(library, tmp_path):
    files = []
    for b, s in enumerate(library["seqs"][:80]):
        p = tmp_path / ("bin%03d.fa" % b)
        p.write_text(">rec%d\n%s\n" % (b, s))
        files.append(str(p))
    rc, so, se = _run("index", "-i", "-k", "4", str(tmp_path / "lib"), *files)
    assert rc == 0 and "DONE" in se
    rc, so, se = _run("track", "-l", "1", "-u", "10", str(tmp_path / "lib.dg"), *files)
    assert rc == 0, se
    assert os.path.getsize(tmp_path / "lib.dg") > 1000
    hit = lambda out: sorted({os.path.basename(l.split("\t")[0]) for l in out.splitlines() if l})
    # the motif occurs in bins 5 and 40 (of the first 80); every variant must verify exactly those
    q = "LMAEG.{4}WWYN"
    for flags in ([], ["-a"], ["-a", "-g", str(tmp_path / "lib.dg")]):
        rc, so, se = _run("query", "-v", *flags, str(tmp_path / "lib.ibf"), q)
        assert rc == 0, se
        assert hit(so) == ["bin005.fa", "bin040.fa"], (flags, se)
    # a motif with the wrong gap length: -a alone widens the candidate set, -g narrows it again
    counts = []
    for flags in ([], ["-a"], ["-a", "-g", str(tmp_path / "lib.dg")]):
        rc, so, se = _run("query", "-v", *flags, str(tmp_path / "lib.ibf"), "LMAEG.{6}WWYN")
        assert hit(so) == []
        counts.append(int(se.split("Narrowed Search to ")[1].split()[0]))
    assert counts[1] >= 2 and counts[1] >= counts[0] and counts[2] <= counts[1] - 1
