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


def test_no_device_memory_is_left_behind():
    """Indexes (flat and hierarchical), sessions, host-buffer probes and pinned buffers created and freed
    a few hundred times: the device's free memory (hipMemGetInfo through torch) ends where it started."""
    import torch
    import oracle as O
    from helpers import random_hibf
    from tetrex_amd import capi
    capi.init(0)
    words = random_words(300, 4099, 0.3, 1)
    ox, descs, values = random_hibf(O, 2, user_bins=120, levels=3)
    kmers = splitmix64(5, 5000) >> np.uint64(44)

    def cycle():
        ix = capi.Index.upload_ibf(300, 4099, 3, words)
        ix.probe(kmers)
        ix.query_masks(["LMAEG", "A.CD[EK]"], False, 4)
        ix.free()
        hx = capi.Index.upload_hibf(120, descs)
        hx.probe(kmers)
        hx.query_masks(["LMAEG"], False, 4)
        hx.free()
        hb = capi.HostBuffer((1000, 5))
        hb.free()
        db = capi.DeviceBuffer(1 << 20)
        db.free()

    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(200):
        cycle()
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_before - free_after < (8 << 20), (free_before, free_after)
