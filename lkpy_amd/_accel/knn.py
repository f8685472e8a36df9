"""``lenskit._accel.knn`` stand-in (src/lenskit/_accel/knn.pyi:8-29)."""
from __future__ import annotations

import numpy as np
import pyarrow as pa
import torch

from .. import _device as D
from ..matrix import SparseIndexType
from ..parallel import AccelTask
from ._util import as_csr_arrays, f32_with_nulls, i32_with_nulls, nullable_i32


def _sim_chunk(offsets: np.ndarray, indices: np.ndarray, values: np.ndarray, dim: int):
    """
    One output chunk exactly as ``ArrowCSRConsumer::complete`` builds it
    (src/accel/sparse/consumer.rs:96-130): ``LargeList<rows: Struct{index: Int32 with the
    lenskit.sparse_index extension (dimension), value: Float32}>``, all fields non-nullable.
    The three buffers are wrapped, not copied.
    """
    idx_t = SparseIndexType(dim)
    idx = pa.ExtensionArray.from_storage(idx_t, pa.array(indices, pa.int32()))
    fields = [pa.field("index", idx_t, nullable=False),
              pa.field("value", pa.float32(), nullable=False)]
    rows = pa.StructArray.from_arrays([idx, pa.array(values, pa.float32())], fields=fields)
    ltype = pa.large_list(pa.field("rows", rows.type, nullable=False))
    return pa.LargeListArray.from_arrays(pa.array(offsets, pa.int64()), rows, type=ltype)


def compute_similarities(ui_ratings, iu_ratings, shape, min_sim: float,
                         save_nbrs: int | None) -> AccelTask:
    """
    Item-item similarity build (src/accel/knn/item_train.rs:33-152).  Returns a task
    yielding a LIST of ``pa.LargeListArray`` chunks (one chunk here; rows in item order,
    sorted by column) -- what the reference's caller feeds to ``pa.chunked_array(...)
    .combine_chunks()`` and ``SparseRowArray.from_array`` (src/lenskit/knn/item.py:173-177).
    """
    nu, ni = shape
    uo, uidx, uval, ushape = as_csr_arrays(ui_ratings)
    io, iidx, ival, ishape = as_csr_arrays(iu_ratings)
    if uval is None or ival is None:
        raise TypeError("compute_similarities needs sparse matrices with values")
    assert ushape == (nu, ni) and ishape == (ni, nu)  # item_train.rs:51-54

    def run(task: AccelTask):
        dev = D.device()
        dt = np.int64 if (uo.dtype == np.int64 or io.dtype == np.int64) else np.int32
        ui = D.DeviceCSR.from_arrays(uo.astype(dt), uidx, uval, ushape, dev)
        iu = D.DeviceCSR.from_arrays(io.astype(dt), iidx, ival, ishape, dev)
        ctl = D.TaskCtl()
        task.attach(ctl)  # cancel() / current_progress() reach the running build kernel
        out = D.iknn_build(ui, iu, min_sim, save_nbrs, ctl=ctl)
        task.set_progress(ni)
        return [_sim_chunk(out.indptr.cpu().numpy(), D.to_host(out.indices),
                           D.to_host(out.values), ni)]

    return AccelTask(run, total=ni)


def _score(sims, ref_items, ref_rates, tgt_items, max_nbrs, min_nbrs):
    so, sidx, sval, sshape = as_csr_arrays(sims)
    if sval is None:
        raise TypeError("invalid similarity matrix: no values")
    assert sshape[0] == sshape[1]  # item_score.rs:113-118
    dev = D.device()
    # (np.array: Arrow buffers are read-only views; torch wants writable host memory)
    dsims = D.DeviceCSR(torch.from_numpy(np.array(so, dtype=np.int64)).to(dev),
                        torch.from_numpy(np.array(sidx, dtype=np.int32)).to(dev),
                        torch.from_numpy(np.array(sval, dtype=np.float32)).to(dev),
                        sshape, None)
    # reference items are walked from the raw value buffer, nulls included
    # (item_score.rs:38-49); the device scorer SKIPS invalid (negative) entries instead
    ri = nullable_i32(ref_items)
    ti = nullable_i32(tgt_items)
    rr = None
    if ref_rates is not None:
        if isinstance(ref_rates, (pa.Array, pa.ChunkedArray)):
            if isinstance(ref_rates, pa.ChunkedArray):
                ref_rates = ref_rates.combine_chunks()
            if not pa.types.is_floating(ref_rates.type):
                raise TypeError(f"invalid reference rating type {ref_rates.type}, expected Float32")
            rr_np = ref_rates.fill_null(0).to_numpy(zero_copy_only=False)
        else:
            rr_np = np.asarray(ref_rates)
        # (an Arrow buffer's NumPy view is read-only: torch wants a writable array -- a copy)
        rr = torch.from_numpy(np.array(rr_np, dtype=np.float32, order="C")).to(dev)
    one = lambda n: torch.tensor([0, n], dtype=torch.int64, device=dev)  # noqa: E731
    s, c = D.iknn_score_batch(dsims, one(len(ri)), torch.from_numpy(ri).to(dev), rr,
                              one(len(ti)), torch.from_numpy(ti).to(dev), max_nbrs, min_nbrs)
    # (pa.FloatArray with nulls, pa.Int32Array with nulls for null targets): accum.rs:180-240;
    # the caller does scores.to_numpy(zero_copy_only=False, writable=True) (knn/item.py:281,288)
    return f32_with_nulls(s.cpu().numpy()), i32_with_nulls(c.cpu().numpy())


def score_explicit(sims, ref_items, ref_rates, tgt_items, max_nbrs: int, min_nbrs: int):
    "-> (pa.FloatArray, pa.Int32Array), nulls where unscored -- item_score.rs:23-69."
    return _score(sims, ref_items, ref_rates, tgt_items, max_nbrs, min_nbrs)


def score_implicit(sims, ref_items, tgt_items, max_nbrs: int, min_nbrs: int):
    "-> (pa.FloatArray, pa.Int32Array) -- item_score.rs:72-111."
    return _score(sims, ref_items, None, tgt_items, max_nbrs, min_nbrs)


def _user_score(tgt_items, nbr_rows, nbr_sims, ratings, max_nbrs, min_nbrs, explicit: bool):
    ro, ridx, rval, rshape = as_csr_arrays(ratings)
    if explicit and rval is None:
        raise TypeError("invalid ratings matrix: no values")  # CSRMatrix::from_arrow
    dev = D.device()
    drat = D.DeviceCSR(torch.from_numpy(np.array(ro, dtype=np.int64)).to(dev),
                       torch.from_numpy(np.array(ridx, dtype=np.int32)).to(dev),
                       torch.from_numpy(np.array(rval, dtype=np.float32)).to(dev)
                       if explicit else None, rshape, None)
    ti = nullable_i32(tgt_items)
    # null neighbours (either array) are skipped: user_score.rs:41-44,80-83
    nr = nullable_i32(nbr_rows)
    if isinstance(nbr_sims, (pa.Array, pa.ChunkedArray)):
        if isinstance(nbr_sims, pa.ChunkedArray):
            nbr_sims = nbr_sims.combine_chunks()
        if not pa.types.is_floating(nbr_sims.type):
            raise TypeError(f"invalid neighbor sims type {nbr_sims.type}, expected float32")
        null = nbr_sims.is_null().to_numpy(zero_copy_only=False) if nbr_sims.null_count else None
        ns = np.array(nbr_sims.fill_null(0).to_numpy(zero_copy_only=False), dtype=np.float32)
        if null is not None:
            nr = np.where(null, -1, nr).astype(np.int32)
    else:
        ns = np.ascontiguousarray(nbr_sims, dtype=np.float32)
    one = lambda n: torch.tensor([0, n], dtype=torch.int64, device=dev)  # noqa: E731
    s, _c = D.uknn_score_batch(drat, one(len(nr)), torch.from_numpy(nr).to(dev),
                               torch.from_numpy(ns).to(dev), one(len(ti)),
                               torch.from_numpy(ti).to(dev), max_nbrs, min_nbrs)
    return f32_with_nulls(s.cpu().numpy())


def user_score_items_explicit(tgt_items, nbr_rows, nbr_sims, ratings, max_nbrs: int,
                              min_nbrs: int):
    "-> pa.FloatArray with nulls -- src/accel/knn/user_score.rs:21-58."
    return _user_score(tgt_items, nbr_rows, nbr_sims, ratings, max_nbrs, min_nbrs, True)


def user_score_items_implicit(tgt_items, nbr_rows, nbr_sims, ratings, max_nbrs: int,
                              min_nbrs: int):
    "-> pa.FloatArray with nulls -- src/accel/knn/user_score.rs:60-98 (structure-only ratings)."
    return _user_score(tgt_items, nbr_rows, nbr_sims, ratings, max_nbrs, min_nbrs, False)
