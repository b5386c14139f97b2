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


def compact(seg, cnt):
    """Padded (n, cap, 4) segments + (n,) counts -> the (sum(cnt), 4) rows of all frames back to back (frame order kept)."""
    seg = np.asarray(seg)
    cnt = np.asarray(cnt)
    keep = np.arange(seg.shape[1], dtype=cnt.dtype)[None, :] < cnt[:, None]
    return seg[keep]


class RowSink:
    """A rank's whole result as it accumulates: every batch's segments, unpadded, back to back in one array that is
    allocated (and touched, so that no page of it is faulted in later) before the first batch arrives.  rows() and
    counts() are what gather_compact takes."""

    def __init__(self, max_frames, cap):
        self._rows = np.zeros((max(1, int(max_frames) * int(cap)), 4), np.int32)
        self._counts = np.zeros(max(1, int(max_frames)), np.int32)
        self._k = self._n = 0

    def append(self, seg, cnt):
        """seg (n, cap, 4) padded segments and cnt (n,) counts of one batch (counts above cap are cut to cap)."""
        cnt = np.minimum(np.asarray(cnt), np.asarray(seg).shape[1]).astype(np.int32, copy=False)
        rows = compact(seg, cnt)
        n, k = len(cnt), len(rows)
        if self._n + n > len(self._counts) or self._k + k > len(self._rows):
            raise ValueError("RowSink is full: %d frames / %d rows held, %d / %d more" % (self._n, self._k, n, k))
        self._rows[self._k:self._k + k] = rows
        self._counts[self._n:self._n + n] = cnt
        self._k += k
        self._n += n

    def rows(self):
        return self._rows[:self._k]

    def counts(self):
        return self._counts[:self._n]


def gather_compact(rows, counts, dst=0, device=None):
    """Final gather of a rank's whole result to rank `dst` without the padding: `rows` (k, 4) int32 are the segments
    of all its frames back to back, `counts` (n,) int32 the segments per frame, in processing order (clip after clip).
    Ranks may hold different n and k.  Returns on dst a list over ranks of (rows, counts), elsewhere None.
    Three collectives in all -- the sizes, the counts, the rows -- whatever the number of frames."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [(rows, counts)]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else "cpu"
    sizes = torch.tensor([len(counts), len(rows)], dtype=torch.int64, device=dev)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [tuple(int(v) for v in t.cpu().tolist()) for t in all_sizes]
    max_n = max(1, max(n for n, _ in all_sizes))
    max_k = max(1, max(k for _, k in all_sizes))
    ct = torch.zeros(max_n, dtype=torch.int32)
    ct[:len(counts)] = torch.from_numpy(counts)
    rt = torch.zeros((max_k, 4), dtype=torch.int32)
    rt[:len(rows)] = torch.from_numpy(rows)
    ct, rt = ct.to(dev), rt.to(dev)
    c_all = [torch.empty_like(ct) for _ in range(world)] if rank == dst else None
    r_all = [torch.empty_like(rt) for _ in range(world)] if rank == dst else None
    dist.gather(ct, c_all, dst=dst)
    dist.gather(rt, r_all, dst=dst)
    if rank != dst:
        return None
    return [(r_all[r].cpu().numpy()[:k], c_all[r].cpu().numpy()[:n]) for r, (n, k) in enumerate(all_sizes)]


def split_clips(rows, counts, clips, frames_per_clip):
    """What one rank sent through gather_compact, cut back into its clips: {clip: list over frames of (k, 4) arrays}."""
    out = {}
    ends = np.cumsum(counts)
    starts = ends - counts
    for i, c in enumerate(clips):
        fr = range(i * frames_per_clip, (i + 1) * frames_per_clip)
        out[c] = [rows[starts[t]:ends[t]] for t in fr]
    return out


def unpad(seg, cnt):
    """Padded (n, cap, 4) + counts -> list of (k, 4) arrays, one per frame."""
    return [seg[i, :cnt[i]].copy() for i in range(len(cnt))]
