"""Multi-GPU partitioning of the two shardable workloads (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm):

  * batched extraction: frames are independent -> contiguous blocks of frames per rank, no
    data-path collective;
  * brute-force Hamming kNN against a large map: the MAP descriptors are sharded in
    contiguous index ranges, every rank searches its shard for all queries (global indices),
    one all-gather of the per-rank top-2 (nq x 16 bytes per rank), then the same lexicographic
    (distance, index) merge on every rank -- identical to a single sequential scan because
    the shards are contiguous index ranges.
"""
from __future__ import annotations


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous shard [base, base + n) of n_total items for `rank`; the last rank takes the
    remainder."""
    assert world >= 1 and 0 <= rank < world and n_total >= 0
    per = n_total // world
    base = rank * per
    n = per if rank < world - 1 else n_total - base
    return base, n


def frame_shard(n_frames: int, world: int, rank: int):
    """Frames [first, last) handled by `rank` for a batched extraction job."""
    base, n = shard_bounds(n_frames, world, rank)
    return base, base + n


def sharded_knn2(d_query, d_train_local, base_index: int, group=None, local_search=None, merge=None,
                 gather_buffer=None):
    """2-NN of every query against a map sharded over the ranks of `group`.

    d_train_local is this rank's contiguous shard, base_index its first global row.  Returns
    an int32 [nq, 4] tensor (idx0, dist0, idx1, dist1) with global indices, identical on every
    rank.  local_search / merge default to the HIP kernels (orbx_hamming_knn2_device,
    orbx_knn2_merge_device); tests inject CPU stand-ins to exercise the collective plumbing
    over gloo.
    """
    import torch
    import torch.distributed as dist
    if local_search is None:
        from .matcher import knn_match2_device
        local_search = lambda q, t, b: knn_match2_device(q, t, base_index=b)[0]
    if merge is None:
        from .matcher import knn2_merge_device
        merge = knn2_merge_device
    part = local_search(d_query, d_train_local, base_index)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return merge(part.unsqueeze(0))
    nq = part.shape[0]
    if gather_buffer is None:
        gather_buffer = torch.empty((world * nq, part.shape[1]), dtype=part.dtype, device=part.device)
    dist.all_gather_into_tensor(gather_buffer.view(world * nq, -1), part.contiguous(), group=group)   # rank-major rows
    return merge(gather_buffer.view(world, nq, -1))
