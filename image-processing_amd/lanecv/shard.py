"""Data-parallel sharding of independent clips over the GPUs of a node.

The lane path is stateless per frame (the reference keeps only host-side smoothing state per clip,
`Simple Lane Detection with OpenCV_5(mp4).py:124-138`), so clips shard with no data-path collective:
clip c goes to rank c mod world, every rank runs its own pipeline, and the only exchange is a final
gather of the padded segment lists to rank 0.  Works with any torch.distributed backend ("nccl" is
RCCL over xGMI on ROCm; "gloo" on the CPU for tests).
"""
import numpy as np


def clip_assignment(nclips, world, rank):
    """Clip-major round robin: the clips rank `rank` owns, in processing order."""
    return list(range(rank, nclips, world))


def gather_segments(seg, cnt, dst=0, device=None):
    """Gathers per-frame padded segments to rank `dst`.

    seg: (n, cap, 4) int32, cnt: (n,) int32 numpy arrays of this rank (same n and cap on every rank).
    Returns on rank dst a list over ranks of (seg, cnt) numpy pairs, elsewhere None.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [(seg, cnt)]
    world, rank = dist.get_world_size(), dist.get_rank()
    st = torch.from_numpy(np.ascontiguousarray(seg))
    ct = torch.from_numpy(np.ascontiguousarray(cnt))
    if device is not None:
        st, ct = st.to(device), ct.to(device)
    seg_all = [torch.empty_like(st) for _ in range(world)] if rank == dst else None
    cnt_all = [torch.empty_like(ct) for _ in range(world)] if rank == dst else None
    dist.gather(st, seg_all, dst=dst)
    dist.gather(ct, cnt_all, dst=dst)
    if rank != dst:
        return None
    return [(s.cpu().numpy(), c.cpu().numpy()) for s, c in zip(seg_all, cnt_all)]


def unpad(seg, cnt):
    """Padded (n, cap, 4) + counts -> list of (k, 4) arrays, one per frame."""
    return [seg[i, :cnt[i]].copy() for i in range(len(cnt))]
