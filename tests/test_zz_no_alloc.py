"""The allocation-free brute-force search (orbx_hamming_knn2_device_static) against the workspace
variant (orbx_hamming_knn2_device), through the C ABI: bit-exact (indices and integer distances),
on two calls in a row, the second with a smaller input, so that whatever the first call left in
the library's fixed device array cannot leak into the second result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


def _pair(A, torch, nq, nt, seed, base):
    from orb_slam3_amd.matcher import knn_match2_device_static
    dev = torch.device("cuda", 0)
    q = synth.random_descriptors(nq, seed)
    t, _ = synth.planted_map(q, nt, seed + 1)
    t[3] = t[11]                                   # equal rows: ties go to the lower index
    dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    ref, _ = A.knn_match2_device(dq, dt, base_index=base)
    out = torch.full((nq, 4), -7, dtype=torch.int32, device=dev)
    knn_match2_device_static(dq, dt, out, base_index=base)
    torch.cuda.synchronize()
    return out.cpu().numpy(), ref.cpu().numpy()


def test_static_knn2_equals_workspace_variant_twice(oracle):
    import torch
    A = _gpu()
    # first call: many chunks, fills a large part of the fixed array
    got, ref = _pair(A, torch, 3000, 200_000, 21, 1000)
    assert np.array_equal(got, ref)
    # second call, smaller in both dimensions: fewer chunks, shorter rows of partial results
    got2, ref2 = _pair(A, torch, 130, 5_000, 33, 0)
    assert np.array_equal(got2, ref2)
    q = synth.random_descriptors(130, 33)
    t, _ = synth.planted_map(q, 5_000, 34)
    t[3] = t[11]
    assert np.array_equal(got2, oracle.knn2(q, t))


def test_static_knn2_rejects_oversize_and_handles_empty():
    import torch
    from orb_slam3_amd import _lib
    A = _gpu()
    L = A.lib()
    dev = torch.device("cuda", 0)
    limit = L.orbx_knn2_static_max_queries()
    # one partial row set (16 bytes per query) plus one span counter per 256 queries must fit the 16 MiB array
    assert 1_000_000 < limit <= (16 << 20) // 16 and limit * 16 + (limit // 256) * 4 <= (16 << 20)
    dq = torch.zeros((8, 32), dtype=torch.uint8, device=dev)
    out = torch.zeros((8, 4), dtype=torch.int32, device=dev)
    # a query count beyond the limit is refused before any pointer is touched
    st = L.orbx_hamming_knn2_device_static(dq.data_ptr(), limit + 1, dq.data_ptr(), 8, 0, out.data_ptr(), None)
    assert st == _lib.ORBX_ERR_INVALID
    assert b"exceed" in L.orbx_last_error()
    # an empty train set yields "no neighbour" records
    st = L.orbx_hamming_knn2_device_static(dq.data_ptr(), 8, None, 0, 0, out.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream)
    assert st == _lib.ORBX_OK
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert (o[:, 0] == -1).all() and (o[:, 2] == -1).all()


def test_static_knn2_allocates_nothing_and_works_when_the_array_limits_the_workers():
    """300 000 queries leave room for three worker row sets in the 16 MiB array, far fewer than the chip would take
    (the merge is lexicographic, so the result does not depend on the number of workers); device memory in use is
    the same before and after the calls."""
    import torch
    from orb_slam3_amd import matcher
    A = _gpu()
    L = A.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(11)
    nq, nt = 300_000, 10_000                                     # 5 spans, 6 workers wanted, room for 3
    dq = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g)
    dt = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev, generator=g)
    dt[17] = dt[5]                                              # equal rows: the lower index must win
    out_s = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    want, _ = A.knn_match2_device(dq, dt)
    L.orbx_knn2_static_max_queries()                            # resolves the array's address outside the measured calls
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(2):
        matcher.knn_match2_device_static(dq, dt, out_s)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, "the allocation-free search changed the device's free memory"
    assert torch.equal(out_s, want)
