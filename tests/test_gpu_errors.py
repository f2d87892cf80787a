"""The C-ABI refuses malformed input on the host, before anything reaches the GPU (a bad descriptor
or blob must never become an out-of-bounds access on the device)."""
import ctypes as C

import numpy as np
import pytest

from helpers import random_words, make_blob, MERGED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _upload_raw(capi, desc_fields, words, user_bins=None, n_shards=1, rank=0):
    d = capi.IbfDesc()
    for k, v in desc_fields.items():
        setattr(d, k, v)
    d.words = words.ctypes.data_as(capi.u64p) if words is not None else None
    desc = capi.IndexDesc(1, C.pointer(d), None, None, desc_fields["bins"] if user_bins is None else user_bins)
    h = C.c_void_p()
    return capi.lib().txq_index_upload(C.byref(desc), rank, n_shards, C.byref(h)), h


def test_inconsistent_ibf_descriptors_are_rejected(capi):
    words = random_words(100, 50, 0.3, 1)
    good = dict(bins=100, tech_bins=128, bin_size=50, hash_shift=64 - (50).bit_length(), bin_words=2, hash_funs=3)
    rc, h = _upload_raw(capi, good, words)
    assert rc == 0
    capi.lib().txq_index_free(h)
    for field, bad in (("tech_bins", 64), ("bin_words", 1), ("hash_shift", 10), ("hash_funs", 0), ("hash_funs", 6),
                       ("bins", 0), ("bin_size", 0)):
        f = dict(good)
        f[field] = bad
        rc, _ = _upload_raw(capi, f, words)
        assert rc == -1, field
        assert capi.lib().txq_last_error()
    assert _upload_raw(capi, good, None)[0] == -1            # no words
    assert _upload_raw(capi, good, words, user_bins=99)[0] == -1
    assert _upload_raw(capi, good, words, n_shards=2, rank=2)[0] == -1
    assert _upload_raw(capi, good, words, n_shards=0)[0] == -1


def test_hibf_trees_are_validated(capi):
    w = random_words(4, 16, 0.5, 2)

    def ibf(nxt, tbu):
        return dict(bins=4, bin_size=16, hash_funs=2, words=w, next_ibf_id=np.array(nxt, dtype=np.uint64),
                    tb_to_user=np.array(tbu, dtype=np.uint64))
    ok = [ibf([1, 0, 0, 0], [MERGED, 0, 1, 2]), ibf([0, 0, 0, 0], [3, 4, 5, 6])]
    capi.Index.upload_hibf(7, ok).free()
    bad_trees = [
        [ibf([5, 0, 0, 0], [MERGED, 0, 1, 2]), ibf([0] * 4, [3, 4, 5, 6])],          # child out of range
        [ibf([0, 0, 0, 0], [MERGED, 0, 1, 2]), ibf([0] * 4, [3, 4, 5, 6])],          # root is its own child
        [ibf([1, 1, 0, 0], [MERGED, MERGED, 1, 2]), ibf([0] * 4, [3, 4, 5, 6])],     # two parents
        [ibf([0, 0, 0, 0], [0, 1, 2, 3]), ibf([0] * 4, [3, 4, 5, 6])],               # unreachable IBF
        [ibf([1, 0, 0, 0], [MERGED, 0, 1, 99]), ibf([0] * 4, [3, 4, 5, 6])],         # user bin out of range
    ]
    for tree in bad_trees:
        with pytest.raises(capi.TxqError) as e:
            capi.Index.upload_hibf(7, tree)
        assert e.value.code == -1


def test_session_feedback_queries_are_bounds_checked(capi):
    ix = capi.Index.upload_ibf(64, 8, 2, np.zeros(8, dtype=np.uint64))
    blob = make_blob(np.zeros(0, dtype=np.uint64), [(4, [(0xFFFFFFFF, 3, 1, 0)]), (3, [])])
    empty = make_blob(np.zeros(0, dtype=np.uint64), [(4, []), (3, [])])
    for qp, qs, bad in (([2], [0], empty), ([0], [4], empty), ([1], [3], empty),
                        ([], [], make_blob(np.zeros(0, dtype=np.uint64), [(3, [])]))):  # (the last: wrong program count for this session)
        sess = ix.session(2)
        assert list(sess.stage(blob, [0, 0], [3, 1])) == [True, True]
        with pytest.raises(capi.TxqError):
            sess.stage(bad, qp, qs)
        # a session with a failed stage takes no further stage and hands out no masks (half-executed state must
        # not come back as plausible masks with TXQ_OK)
        with pytest.raises(capi.TxqError) as e:
            sess.stage(empty)
        assert e.value.code == -4
        with pytest.raises(capi.TxqError) as e:
            sess.end()
        assert e.value.code == -4
    sess = ix.session(2)
    assert list(sess.stage(blob, [0, 0], [3, 1])) == [True, True]
    out = sess.end()
    assert out.shape == (2, 1) and not out.any()
    ix.free()  # (every failed session was closed by its end(): the index has no session left)


def test_emplace_and_download_guards(capi):
    ix = capi.Index.create_ibf(100, 50, 3)
    with pytest.raises(capi.TxqError):
        capi.check(capi.lib().txq_index_download_words(ix._h, np.zeros(3, dtype=np.uint64).ctypes.data_as(capi.u64p), 3))
    # bins outside the index are skipped, not written
    vals = capi.DeviceBuffer.from_numpy(np.arange(10, dtype=np.uint64))
    bins = capi.DeviceBuffer.from_numpy(np.full(10, 1000, dtype=np.uint32))
    ix.emplace_device(vals.ptr, bins.ptr, 10)
    capi.synchronize()
    assert not ix.download_words_rows(50).any()
    ix.free()
    # the bits of an index cannot change under an open session
    ix = capi.Index.create_ibf(100, 50, 3)
    sess = ix.session(1)
    with pytest.raises(capi.TxqError) as e:
        ix.emplace_device(vals.ptr, bins.ptr, 10)
    assert e.value.code == -4
    capi.check(capi.lib().txq_session_end(sess._h, None))
    sess._h = None
    ix.emplace_device(vals.ptr, bins.ptr, 10)
    capi.synchronize()
    ix.free()
    hx = capi.Index.upload_hibf(4, [dict(bins=4, bin_size=16, hash_funs=2, words=random_words(4, 16, 0.5, 2),
                                          next_ibf_id=np.zeros(4, dtype=np.uint64), tb_to_user=np.arange(4, dtype=np.uint64))])
    with pytest.raises(capi.TxqError):
        hx.emplace_device(vals.ptr, bins.ptr, 10)
    hx.free()
