"""libtxq.so inside a PyTorch-ROCm process: torch owns device memory and the stream
(plumbing), the HIP kernels of libtxq run on torch's stream against torch's buffers."""
import numpy as np
import pytest

from helpers import random_words, oracle_ibf_from_words, splitmix64

pytestmark = pytest.mark.gpu


def test_probe_on_torch_buffers_and_stream(oracle):
    import torch
    assert torch.cuda.is_available()
    from tetrex_amd import capi
    capi.init(0)
    bins, m, h, n = 1024, 4099, 3, 10000
    words = random_words(bins, m, 0.4, 3)
    ix = capi.Index.upload_ibf(bins, m, h, words)
    kmers = splitmix64(4, n) >> np.uint64(44)
    dk = torch.from_numpy(kmers.view(np.int64)).cuda()
    dm = torch.empty((n, ix.shard_words), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ix.probe_device(dk.data_ptr(), n, dm.data_ptr(), None, stream)
    ev1.record()
    torch.cuda.synchronize()
    assert ev0.elapsed_time(ev1) > 0
    got = dm.cpu().numpy().view(np.uint64)
    want = oracle_ibf_from_words(oracle, bins, m, h, words).probe(kmers)
    assert np.array_equal(got, want)
    ix.free()
