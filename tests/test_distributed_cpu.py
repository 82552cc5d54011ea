"""The N > 1 paths on CPU with gloo (world_size 2): map sharding + all-gather + merge for the
brute-force matcher, and frame sharding for batched extraction.  The HIP kernels are replaced
by explicit CPU stand-ins (the oracle as the local search, a numpy lexicographic merge), so
this covers the partitioning and the collective plumbing, not the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    from orb_slam3_amd.dist import frame_shard, shard_bounds
    for n in (0, 1, 7, 8, 1024, 10_000_000, 10_000_003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0
            for (b0, n0), (b1, _) in zip(spans[:-1], spans[1:]):
                assert b0 + n0 == b1
            assert spans[-1][0] + spans[-1][1] == n
    assert [frame_shard(1024, 8, r) for r in (0, 7)] == [(0, 128), (896, 1024)]


def _numpy_merge(parts: torch.Tensor) -> torch.Tensor:
    p = parts.numpy().astype(np.int64)
    world, nq, _ = p.shape
    out = np.zeros((nq, 4), np.int32)
    for q in range(nq):
        c = [(p[w, q, 1], p[w, q, 0]) for w in range(world) if p[w, q, 0] >= 0]
        c += [(p[w, q, 3], p[w, q, 2]) for w in range(world) if p[w, q, 2] >= 0]
        c.sort()
        c += [(2**31 - 1, -1)] * 2
        out[q] = [c[0][1], c[0][0], c[1][1], c[1][0]]
    return torch.from_numpy(out)


def _worker(rank, world, port, q, t, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from orb_slam3_amd.dist import shard_bounds, sharded_knn2
    base, n = shard_bounds(len(t), world, rank)

    def local(qq, tt, b):
        r = O.knn2(qq.numpy(), tt.numpy())
        r[:, 0] = np.where(r[:, 0] >= 0, r[:, 0] + b, -1)
        r[:, 2] = np.where(r[:, 2] >= 0, r[:, 2] + b, -1)
        return torch.from_numpy(r)

    res = sharded_knn2(torch.from_numpy(q), torch.from_numpy(t[base:base + n]), base, local_search=local, merge=_numpy_merge)
    ret[rank] = res.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_knn2_gloo(oracle, world):
    from orb_slam3_amd import synth
    q = synth.random_descriptors(40, 5)
    t, _ = synth.planted_map(q, 3001, 6)
    t[10] = t[2900]                      # a tie across shards: the lower global index must win
    t[11] = q[0]; t[2000] = q[0]
    ref = oracle.knn2(q, t)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, q, t, ret), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(ret[r], ref), f"rank {r}"
