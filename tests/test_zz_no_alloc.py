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
    assert limit == (16 << 20) // 16
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
