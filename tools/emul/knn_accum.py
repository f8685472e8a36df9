"""
Python model of what csrc/iknn_score.hip does per TARGET -- the part that decides bit parity with
the reference accumulator (src/accel/knn/accum.rs) and cannot be looked at without a GPU:

* `KfHeap`: std's BinaryHeap push (sift_up) / pop (swap last to root, sift_down_to_bottom,
  sift_up) on (weight, value) arrays with the REVERSED ordering of accum.rs:170-184;
* the slot kernel's IN-PLACE Partial -> Full conversion: the reference pops the vector from the
  back and pushes every element onto an empty heap; the kernel reverses the vector in place and
  sifts element k up into the heap formed by the k before it;
* the candidate-list kernel's ROUNDS: histories are cut into rounds of CAP rows, a round's hits
  arrive in arbitrary order, are sorted back by history position (rank counting), appended while
  the vector has room, and the rest replayed one by one on the heap; the accumulator is carried
  from round to round;
* the final sums: sequential float32, product rounded before it is added.

`score_target(hits, max_nbrs, ...)` is compared with the C oracle in tests/test_emul.py.
TEST INFRASTRUCTURE (tools/, tests/ only).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class KfHeap:
    "struct KfHeap of csrc/iknn_score.hip, statement for statement"

    def __init__(self, cap: int):
        self.w = np.zeros(cap, f32)
        self.v = np.zeros(cap, f32)
        self.len = 0

    def sift_up(self, pos: int, ew, ev):
        while pos > 0:
            parent = (pos - 1) >> 1
            if ew >= self.w[parent]:
                break
            self.w[pos], self.v[pos] = self.w[parent], self.v[parent]
            pos = parent
        self.w[pos], self.v[pos] = ew, ev

    def push(self, ew, ev):
        self.sift_up(self.len, ew, ev)
        self.len += 1

    def pop(self):
        ew, ev = self.w[self.len - 1], self.v[self.len - 1]
        self.len -= 1
        if self.len == 0:
            return
        end = self.len
        pos, child = 0, 1
        limit = end - 2 if end >= 2 else 0
        while child <= limit and end >= 2:
            if self.w[child] >= self.w[child + 1]:
                child += 1
            self.w[pos], self.v[pos] = self.w[child], self.v[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1:
            self.w[pos], self.v[pos] = self.w[child], self.v[child]
            pos = child
        self.sift_up(pos, ew, ev)

    def heapify_in_place(self):
        "the slot kernel's Partial -> Full: reverse, then sift element k up into [0, k)"
        n = self.len
        self.w[:n] = self.w[:n][::-1].copy()
        self.v[:n] = self.v[:n][::-1].copy()
        for k in range(1, n):
            self.sift_up(k, self.w[k], self.v[k])

    def heapify_by_pushes(self):
        "the list kernel's (and the reference's): staged copy, pushed from the back"
        n = self.len
        tw, tv = self.w[:n].copy(), self.v[:n].copy()
        self.len = 0
        for i in range(n - 1, -1, -1):
            self.push(tw[i], tv[i])


def score_target(hits, max_nbrs: int, min_nbrs: int, explicit: bool, cap: int = 256,
                 in_place: bool = False, rng: np.random.Generator | None = None):
    """
    ``hits``: [(history position, weight, value)] of ONE target in history order.  Processed as
    the list kernel does: rounds of ``cap`` history rows, each round's hits shuffled (``rng``:
    the order the atomics happened to produce), ranked back into history order, fed to the
    accumulator.  Returns (score or NaN, count).
    """
    h = KfHeap(max_nbrs + 1)
    full = False
    hits = sorted(hits, key=lambda t: t[0])
    n_rounds = (max((p for p, _, _ in hits), default=-1) // cap) + 1
    for rnd in range(max(n_rounds, 1)):
        batch = [(p - rnd * cap, w, v) for p, w, v in hits if rnd * cap <= p < (rnd + 1) * cap]
        if rng is not None and len(batch) > 1:
            batch = [batch[i] for i in rng.permutation(len(batch))]
        # rank = number of hits with a smaller position (positions are distinct)
        pos = np.array([b[0] for b in batch], np.int64)
        order = np.empty(len(batch), np.int64)
        for i, p in enumerate(pos):
            order[int(np.sum(pos < p))] = i
        sw = [batch[i][1] for i in order]
        sv = [batch[i][2] for i in order]
        i0 = 0
        if not full:
            i0 = min(len(sw), max_nbrs - h.len)
            for i in range(i0):
                h.w[h.len], h.v[h.len] = sw[i], sv[i]
                h.len += 1
        if i0 < len(sw):
            if not full:
                if in_place:
                    h.heapify_in_place()
                else:
                    h.heapify_by_pushes()
                full = True
            for i in range(i0, len(sw)):
                if sw[i] > h.w[0]:
                    h.push(sw[i], sv[i])
                    while h.len > max_nbrs:
                        h.pop()
    n = h.len
    if n < min_nbrs or n == 0:
        return float("nan"), n
    tw, ws = f32(0), f32(0)
    for i in range(n):
        tw = f32(tw + h.w[i])
        ws = f32(ws + f32(h.w[i] * h.v[i]))
    return float(f32(ws / tw) if explicit else tw), n


# ---- cg_reduce8 of csrc/als_cg.hip ---------------------------------------------------------------
def cg_reduce8(a: np.ndarray) -> np.ndarray:
    """
    ``a``: [64 lanes, 8 items] partial dot products.  The fold of cg_reduce8 (lane ^ 1 keeps 4
    of the 8 items, lane ^ 2 two, lane ^ 4 one, then 8 / 16 / 32): returns, per lane, the value it
    ends up with.  Exchanges are modelled as exact lane permutations, sums in float64.
    """
    lanes = np.arange(64)
    a = a.astype(np.float64)
    b0, b1, b2 = (lanes & 1) != 0, (lanes & 2) != 0, (lanes & 4) != 0
    b = np.empty((64, 4))
    for m in range(4):
        keep = np.where(b0, a[:, m + 4], a[:, m])
        send = np.where(b0, a[:, m], a[:, m + 4])
        b[:, m] = keep + send[lanes ^ 1]
    c = np.empty((64, 2))
    for m in range(2):
        keep = np.where(b1, b[:, m + 2], b[:, m])
        send = np.where(b1, b[:, m], b[:, m + 2])
        c[:, m] = keep + send[lanes ^ 2]
    keep = np.where(b2, c[:, 1], c[:, 0])
    send = np.where(b2, c[:, 0], c[:, 1])
    d = keep + send[lanes ^ 4]
    d = d + d[(lanes & ~15) | ((lanes + 8) & 15)]  # row_ror:8 inside the row of 16
    d = d + d[lanes ^ 16]
    d = d + d[lanes ^ 32]
    return d


def cg_rev3(j: int) -> int:
    return ((j & 1) << 2) | (j & 2) | ((j & 4) >> 2)
