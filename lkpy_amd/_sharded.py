"""
One-process-per-GPU drivers for the two legs of the path that shard WITHOUT a data-path
collective (last row):

* item-kNN model build -- output rows (items) are independent (`compute_similarities` is a
  ``par_iter`` over rows, src/accel/knn/item_train.rs:56-70): every rank holds both normalised
  orientations (200 MB on ML-25M: replicated), builds a contiguous block of output rows
  balanced by multiply-accumulates (``lk_iknn_plan_create_rows``), and the blocks are stitched
  in rank order -- row offsets rebased by the running entry count, nothing else touched;
* dense scoring + top-N -- users are independent; every rank scores a contiguous block of users
  against the replicated item factors.

The only communication is the collection of the finished blocks (sizes by ``all_gather``, the
blocks themselves point-to-point to the destination rank), after the timed compute.  The
arithmetic sits behind a callable so that the sharding / stitching logic runs on CPU (gloo,
world_size 2) in the tests with the oracle standing in; the product callables are the HIP
kernels (:func:`iknn_build_sharded`, :func:`score_topk_sharded`).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def balanced_ranges(weights: np.ndarray, world: int) -> list[tuple[int, int]]:
    """
    ``world`` contiguous row ranges covering ``[0, len(weights))`` whose weight sums are as equal
    as a prefix-sum cut allows (rows stay in order: the stitched result needs no permutation).
    """
    n = len(weights)
    cum = np.concatenate([[0], np.cumsum(weights.astype(np.float64))])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(cum, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _send(t: torch.Tensor, dst: int, group):
    t = t.contiguous()
    if t.is_cuda and dist.get_backend(group) != "nccl":
        # RCCL point-to-point is ordered on the current stream; a host-staged backend (gloo: the
        # two-process tests on one GPU) reads the buffer from the host side -- the kernels that
        # produced it must have finished
        torch.cuda.current_stream(t.device).synchronize()
    dist.send(t, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)


def _recv(t: torch.Tensor, src: int, group):
    dist.recv(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    if t.is_cuda and dist.get_backend(group) != "nccl":
        torch.cuda.current_stream(t.device).synchronize()


def stitch_csr_blocks(ptr: torch.Tensor, idx: torch.Tensor, val: torch.Tensor, n_rows_total: int,
                      ranges, group=None, dst: int = 0):
    """
    Collect per-rank CSR row blocks (local offsets starting at 0, rows ``ranges[rank]``) on rank
    ``dst`` as ONE CSR: (offsets int64 [n_rows_total + 1], indices int32, values f32); other
    ranks get ``None``.  Blocks travel point-to-point with their exact sizes.
    """
    world, rank = _world(group)
    if world == 1:
        return ptr.to(torch.int64), idx, val
    dev = ptr.device
    mine = torch.tensor([int(idx.shape[0])], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = [int(c.item()) for c in counts]
    if rank != dst:
        _send(ptr.to(torch.int64), dst, group)
        if counts[rank] > 0:
            _send(idx, dst, group)
            _send(val, dst, group)
        return None
    out_ptr = torch.zeros(n_rows_total + 1, dtype=torch.int64, device=dev)
    out_idx = torch.empty(sum(counts), dtype=torch.int32, device=dev)
    out_val = torch.empty(sum(counts), dtype=torch.float32, device=dev)
    base = 0
    for r in range(world):
        lo, hi = ranges[r]
        if r == rank:
            p, i, v = ptr.to(torch.int64), idx, val
        else:
            p = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev)
            _recv(p, r, group)
            i = torch.empty(counts[r], dtype=torch.int32, device=dev)
            v = torch.empty(counts[r], dtype=torch.float32, device=dev)
            if counts[r] > 0:
                _recv(i, r, group)
                _recv(v, r, group)
        out_ptr[lo + 1 : hi + 1] = p[1:] + base  # offsets rebased by the entries before
        out_idx[base : base + counts[r]] = i
        out_val[base : base + counts[r]] = v
        base += counts[r]
    return out_ptr, out_idx, out_val


def build_rows_sharded(build_rows, n_rows: int, row_weights: np.ndarray, group=None,
                       collect: bool = True, dst: int = 0):
    """
    ``build_rows(lo, hi) -> (ptr, idx, val)`` (torch tensors, local offsets) is called for this
    rank's block; returns ``(ranges, stitched | None)`` -- stitched only on ``dst`` and only when
    ``collect`` (a serving deployment keeps the blocks where they are).
    """
    world, rank = _world(group)
    ranges = balanced_ranges(row_weights, world)
    lo, hi = ranges[rank]
    ptr, idx, val = build_rows(lo, hi)
    if not collect:
        return ranges, (ptr, idx, val)
    return ranges, stitch_csr_blocks(ptr, idx, val, n_rows, ranges, group, dst)


def iknn_row_weights(ui_h_indptr: np.ndarray, iu_h_indptr: np.ndarray,
                     iu_indices: np.ndarray) -> np.ndarray:
    "multiply-accumulates of every output row: sum over the item's users of their row lengths"
    ulen = np.diff(ui_h_indptr).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(ulen[iu_indices])])
    return (csum[iu_h_indptr[1:]] - csum[iu_h_indptr[:-1]]).astype(np.float64)


def iknn_build_sharded(ui, iu, min_sim: float, save_nbrs=None, group=None, collect: bool = True):
    """
    Item-kNN build over all ranks of ``group`` (product path: ``_device.iknn_build`` with
    ``rows=``).  ``ui`` / ``iu``: the normalised DeviceCSR orientations, replicated.  Returns
    ``(ranges, (ptr, idx, val) | None)``.
    """
    from . import _device as D

    w = iknn_row_weights(ui.h_indptr, iu.h_indptr, iu.indices.cpu().numpy())

    def build(lo, hi):
        out = D.iknn_build(ui, iu, min_sim, save_nbrs, rows=(lo, hi))
        return out.indptr, out.indices, out.values

    return build_rows_sharded(build, iu.shape[0], w, group, collect)


def shard_rows_even(n: int, world: int) -> list[tuple[int, int]]:
    per = (n + world - 1) // world
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world)]


def topk_sharded(score_block, n_users: int, n: int, group=None, collect: bool = True,
                 dst: int = 0):
    """
    ``score_block(lo, hi) -> (idx int32 [hi-lo x n], score f32 [hi-lo x n])`` for this rank's
    users; collected (rank order = user order) on ``dst`` when ``collect``.
    """
    world, rank = _world(group)
    ranges = shard_rows_even(n_users, world)
    lo, hi = ranges[rank]
    idx, sc = score_block(lo, hi)
    if world == 1 or not collect:
        return ranges, (idx, sc)
    if rank != dst:
        if hi > lo:
            _send(idx, dst, group)
            _send(sc, dst, group)
        return ranges, None
    out_i = torch.empty((n_users, n), dtype=torch.int32, device=idx.device)
    out_s = torch.empty((n_users, n), dtype=torch.float32, device=idx.device)
    for r, (a, b) in enumerate(ranges):
        if r == rank:
            out_i[a:b], out_s[a:b] = idx, sc
        elif b > a:
            _recv(out_i[a:b], r, group)
            _recv(out_s[a:b], r, group)
    return ranges, (out_i, out_s)


def score_topk_sharded(users: torch.Tensor, items: torch.Tensor, k: int, n: int,
                       excl_ptr: torch.Tensor | None = None, excl_items: torch.Tensor | None = None,
                       group=None, collect: bool = True):
    "Dense scoring + top-N over all ranks (product path: ``lk_score_topk`` per user block)."
    from . import _device as D

    def block(lo, hi):
        ep = None
        if excl_ptr is not None:
            ep = excl_ptr[lo : hi + 1].contiguous()  # absolute offsets into excl_items
        return D.score_topk(users[lo:hi].contiguous(), items, k, n, ep, excl_items)

    return topk_sharded(block, users.shape[0], n, group, collect)
