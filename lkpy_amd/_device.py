"""
Device-side plumbing: torch tensors as HBM buffers, the current torch stream as the
HIP stream, thin typed wrappers over the C ABI (``include/lkamd.h``).

Nothing in here computes: every arithmetic step is a hand-written HIP kernel behind
``lkpy_amd/_lkamd.so``.  PyTorch only owns memory, streams and (elsewhere) the RCCL
process group.
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _native
from ._native import check


def device(dev=None) -> torch.device:
    _native.require_gpu()
    if dev is None:
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise _native.BackendUnavailable(f"lkpy_amd runs on a HIP device only (got {dev})")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _ptr(t) -> ctypes.c_void_p:
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def padded_dim(k: int) -> int:
    kp = _native.load().lk_padded_dim(int(k))
    if kp == 0:
        raise ValueError(f"unsupported embedding size {k} (supported: 1..1024)")
    return kp


def to_device_padded(mat: np.ndarray, dev) -> torch.Tensor:
    "Host [n x k] float32 -> device [n x KP] with zero pad columns."
    mat = np.ascontiguousarray(mat, dtype=np.float32)
    n, k = mat.shape
    kp = padded_dim(k)
    src = torch.from_numpy(mat).to(dev)
    if kp == k:
        return src.contiguous()
    dst = torch.empty((n, kp), dtype=torch.float32, device=dev)
    check(_native.load().lk_pad_rows(_ptr(src), n, k, k, _ptr(dst), kp, _stream()), "lk_pad_rows")
    return dst


_NP_OF = {torch.float32: np.float32, torch.int32: np.int32, torch.int64: np.int64,
          torch.uint8: np.uint8, torch.float64: np.float64}


def _advise_hugepages(arr: np.ndarray) -> None:
    """
    Ask for transparent huge pages under a freshly allocated destination (``madvise``,
    ``MADV_HUGEPAGE``): a 9.2 GB result is 2.3 M first-touch faults of 4 KiB pages -- host time
    that the download team pays while the link waits -- against 4 600 faults of 2 MiB pages.
    Best effort: silently a no-op where the kernel does not offer it (LK_DOWNLOAD_THP=0: off).
    """
    if os.environ.get("LK_DOWNLOAD_THP", "1") == "0":
        return
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        page = 2 << 20
        beg = (arr.ctypes.data + page - 1) // page * page
        end = (arr.ctypes.data + arr.nbytes) // page * page
        if end > beg:
            libc.madvise(ctypes.c_void_p(beg), ctypes.c_size_t(end - beg), 14)  # MADV_HUGEPAGE
    except Exception:  # noqa: BLE001 -- an optimisation hint only
        pass


def to_host(t: torch.Tensor, threads: int = 0, index_bound: int | None = None) -> np.ndarray:
    """
    A device tensor as a host NumPy array.  Large results (>= 64 MB) go through ``lk_download``
    (pinned staging ring + a team of host threads: PCIe speed into pageable memory instead of
    the ~12 GB/s of a plain copy into fresh pages); small ones are a plain ``.cpu()``.
    ``index_bound``: an int32 tensor whose values are known to lie in [0, index_bound) -- with
    index_bound <= 65 536 it crosses the link as uint16 (``lk_download_i32_narrow``).
    """
    nbytes = t.numel() * t.element_size()
    if t.is_cuda and (256 << 10) <= nbytes < (64 << 20) and t.dtype in _NP_OF:
        # mid-sized results (the [B x n] lists of a batch recommend call: 8 MB): one DMA into a
        # pinned block of torch's caching host allocator -- a plain ``.cpu()`` lands in fresh
        # pageable pages at 10 ... 40 GB/s (0.2 ... 0.8 ms for those 8 MB, measured); the array
        # keeps the block alive and hands it back to the cache when it is dropped
        host = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host.numpy()
    if not t.is_cuda or nbytes < (64 << 20) or t.dtype not in _NP_OF:
        return t.cpu().numpy()
    t = t.contiguous()
    out = np.empty(tuple(t.shape), dtype=_NP_OF[t.dtype])
    _advise_hugepages(out)
    lib = _native.require_gpu()
    if (index_bound is not None and index_bound <= 65536 and t.dtype == torch.int32
            and os.environ.get("LK_DOWNLOAD_NARROW", "1") != "0"):
        tmp = torch.empty(t.numel(), dtype=torch.int16, device=t.device)
        check(lib.lk_download_i32_narrow(out.ctypes.data_as(ctypes.c_void_p), _ptr(t), t.numel(),
                                         _ptr(tmp), int(threads), _stream()),
              "lk_download_i32_narrow")
        return out
    check(lib.lk_download(out.ctypes.data_as(ctypes.c_void_p), _ptr(t), nbytes, int(threads),
                          _stream()), "lk_download")
    return out


def to_host_unpadded(mat: torch.Tensor, k: int) -> np.ndarray:
    "Device [n x KP] -> host [n x k] float32."
    n, kp = mat.shape
    if kp == k:
        return mat.cpu().numpy()
    dst = torch.empty((n, k), dtype=torch.float32, device=mat.device)
    check(
        _native.load().lk_unpad_rows(_ptr(mat), n, k, kp, _ptr(dst), k, _stream()),
        "lk_unpad_rows",
    )
    return dst.cpu().numpy()


@dataclass
class DeviceCSR:
    "CSR in HBM: the SparseRowArray layout (offsets i32/i64, indices i32, values f32)."

    indptr: torch.Tensor
    indices: torch.Tensor
    values: torch.Tensor
    shape: tuple[int, int]
    h_indptr: np.ndarray  # host copy of the offsets (plans are built from it)

    @property
    def nnz(self) -> int:
        return int(self.indices.shape[0])

    @property
    def is64(self) -> bool:
        return self.indptr.dtype == torch.int64

    @classmethod
    def from_arrays(cls, indptr, indices, values, shape, dev) -> "DeviceCSR":
        indptr = np.ascontiguousarray(indptr)
        if indptr.dtype not in (np.int32, np.int64):
            indptr = indptr.astype(np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        # zero-copy views of Arrow buffers are read-only; torch.from_numpy wants writable memory
        indptr, indices, values = (a if a.flags.writeable else a.copy()
                                   for a in (indptr, indices, values))
        return cls(
            torch.from_numpy(indptr).to(dev),
            torch.from_numpy(indices).to(dev),
            torch.from_numpy(values).to(dev),
            (int(shape[0]), int(shape[1])),
            indptr,
        )

    @classmethod
    def from_scipy(cls, mat, dev) -> "DeviceCSR":
        mat = mat.tocsr()
        mat.sort_indices()
        return cls.from_arrays(mat.indptr, mat.indices, mat.data, mat.shape, dev)


class TaskCtl:
    """
    Cancel + progress words of a long-running call (``lk_task_ctl``; the ``AccelTask``
    protocol of src/accel/tasks/mod.rs:62-106): ``cancel()`` may be called from any thread
    while the kernels run, ``progress()`` reads the live row count from pinned host memory.
    """

    def __init__(self):
        lib = _native.require_gpu()
        self._h = ctypes.c_void_p(0)
        check(lib.lk_task_ctl_create(ctypes.byref(self._h)), "lk_task_ctl_create")

    def cancel(self):
        _native.load().lk_task_ctl_cancel(self._h)

    @property
    def cancelled(self) -> bool:
        return bool(_native.load().lk_task_ctl_cancelled(self._h))

    def reset(self):
        _native.load().lk_task_ctl_reset(self._h)

    def progress(self) -> tuple[int, int]:
        "(rows done, rows total) of the call in flight (or of the last finished one)"
        d, t = ctypes.c_int64(0), ctypes.c_int64(0)
        check(_native.load().lk_task_ctl_progress(self._h, ctypes.byref(d), ctypes.byref(t)))
        return int(d.value), int(t.value)

    def __del__(self):
        try:
            if self._h:
                _native.load().lk_task_ctl_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


class Gramian:
    "``M^T M + reg I`` (lk_gramian) with a reusable workspace."

    def __init__(self, k: int, dev):
        self.k = int(k)
        self.kp = padded_dim(k)
        lib = _native.load()
        self.ws = torch.empty(lib.lk_gramian_workspace_bytes(self.k), dtype=torch.uint8, device=dev)
        self.dev = dev

    def __call__(self, m: torch.Tensor, reg: float, out: torch.Tensor | None = None):
        n, ld = m.shape
        assert ld == self.kp and m.dtype == torch.float32 and m.is_contiguous()
        if out is None:
            out = torch.empty((self.k, self.k), dtype=torch.float32, device=self.dev)
        check(
            _native.load().lk_gramian(
                _ptr(m), n, self.k, ld, float(reg), _ptr(out), out.stride(0), _ptr(self.ws),
                _stream()
            ),
            "lk_gramian",
        )  # fmt: skip
        return out


def als_order_mode(x) -> str:
    "``auto`` / ``reference`` / ``accurate`` from a bool, a string or None (= LK_ALS_RHS_ORDER)"
    if x is None:
        x = os.environ.get("LK_ALS_RHS_ORDER", "") or "auto"
    if x is True:
        return "reference"
    if x is False:
        return "accurate"
    x = str(x).lower()
    if x in ("hybrid", "default"):
        x = "auto"
    if x not in ("auto", "reference", "accurate"):
        raise ValueError(f"unknown LK_ALS_RHS_ORDER {x!r} (auto / reference / accurate)")
    return x


class ALSPlan:
    "Row schedule + workspace of one CSR orientation (lk_als_plan)."

    def __init__(self, csr: DeviceCSR, k: int, solver: int = _native.SOLVER_AUTO,
                 reference_order: "bool | str | None" = None):
        """``reference_order`` -- how the two long sums of a row (normal matrix, right-hand side)
        are ordered (include/lkamd.h, ``lk_als_plan_create_ex``; INTEGRATION.md, ``LK_ALS_RHS_ORDER``):

        * ``"auto"`` (the default): rows longer than ``LK_ALS_REF_LEN`` (2048) entries in the
          reference's own order (LK_ALS_PLAN_HYBRID_ORDER), the others in the tuned kernels' order;
        * ``"reference"`` / ``True``: strict -- every row of more than 256 entries
          (LK_ALS_PLAN_REFERENCE_ORDER + the rhs workspace);
        * ``"accurate"`` / ``False``: the tuned kernels' own summation on every row (round 4's
          default: closer to the exact solution on rows of 10^4+ entries, up to 7e-2 from the
          reference there);
        * ``None``: what ``LK_ALS_RHS_ORDER`` says (default ``auto``).
        """
        lib = _native.require_gpu()
        self.csr = csr
        self.k = int(k)
        self.kp = padded_dim(k)
        self._h = ctypes.c_void_p(0)
        hp = csr.h_indptr
        self.order_mode = als_order_mode(reference_order)
        if int(solver) == _native.SOLVER_CG or self.kp > 256:
            self.order_mode = "accurate"  # (no reference arithmetic to reproduce / no slab path)
        self.reference_order = self.order_mode == "reference"
        flags = {"accurate": 0, "reference": 1, "auto": 2}[self.order_mode]
        check(
            lib.lk_als_plan_create_ex(
                ctypes.byref(self._h), hp.ctypes.data_as(ctypes.c_void_p),
                1 if hp.dtype == np.int64 else 0, csr.shape[0], self.k, int(solver), flags
            ),
            "lk_als_plan_create",
        )  # fmt: skip
        dev = csr.indices.device
        self.ws = torch.empty(lib.lk_als_plan_workspace_bytes(self._h), dtype=torch.uint8,
                              device=dev)
        self.frob = torch.zeros(1, dtype=torch.float32, device=dev)
        self.solver = int(lib.lk_als_plan_solver(self._h))
        # rows with <= 64 entries at padded k > 64: Woodbury paths (csrc/als_wb.hip, als_wb64_kernel)
        # when there are enough of them to pay for Z = other @ OtOr^-1 (LK_ALS_WB_MIN_ROWS; 0
        # disables)
        self.short_rows = int(lib.lk_als_plan_short_rows(self._h))        # <= 16 entries
        # rows the Woodbury kernels take: <= 64 entries at padded k = 256 (<= 128 with the
        # 128 x 128 variant, counted by the caller); at k = 128 <= 16, or <= 32 / 64 with
        # LK_ALS_WB64_K128 (the library applies the same rule)
        self.woodbury_rows = int(lib.lk_als_plan_woodbury_rows(self._h))
        wb_min = int(os.environ.get("LK_ALS_WB_MIN_ROWS", "4096"))
        self.use_wb = (64 < self.kp <= 256 and self.solver == _native.SOLVER_CHOLESKY
                       and wb_min > 0 and self.woodbury_rows >= wb_min)
        self._negative_values = None  # not scanned yet (one reduction + one host sync)
        if self.use_wb and self.negative_values:
            # the Woodbury kernels take sqrt(v) of every confidence increment: with negative
            # values (use_ratings=True and negative ratings) they would flag rows the dense
            # sposv path still solves -- such matrices keep the dense kernels for every row
            self.use_wb = False
        self._z = None
        self._z_leader = None  # another slice's plan whose Z this one uses (share_z_from)
        self._z_shared_set = False
        self._yref = None
        if self.reference_order:
            self.set_rhs_order("reference")

    @property
    def negative_values(self) -> bool:
        """
        Does the matrix hold a confidence value below zero?  Scanned on first use and remembered:
        whoever is about to switch the Woodbury kernels on asks -- this plan for itself, an
        ``ALSPlanGroup`` for the half-epoch as a whole (its decision can differ from every
        slice's own: slices each below LK_ALS_WB_MIN_ROWS whose sum is above it).
        """
        if self._negative_values is None:
            v = self.csr.values
            self._negative_values = bool(v is not None and v.numel() > 0 and float(v.min()) < 0.0)
        return self._negative_values

    def set_rhs_order(self, order: str):
        """
        ``"reference"``: every half-epoch forms the right-hand side in the reference's own
        summation order (one sequential float32 chain per feature: implicit.rs:116-117; csrc/
        als_rhs.hip) and the dense kernels solve with it; ``"accurate"`` (default): the solve
        kernels' own slotted / chunked sum.  INTEGRATION.md, ``LK_ALS_RHS_ORDER``.
        """
        if order not in ("reference", "accurate"):
            raise ValueError(f"unknown right-hand-side order {order!r}")
        if order == "accurate" and getattr(self, "reference_order", False):
            raise ValueError("a reference-order plan cannot switch back: build another plan")
        if getattr(self, "order_mode", "accurate") == "auto":
            raise ValueError("a hybrid-order plan owns its right-hand-side buffer: build a plan "
                             "with reference_order='accurate' or 'reference' instead")
        if order == "reference" and self.solver != _native.SOLVER_CHOLESKY:
            return  # the CG option has no reference arithmetic to reproduce
        if order == "reference" and self._yref is None:
            self._yref = torch.zeros((max(self.csr.shape[0], 1), self.kp), dtype=torch.float32,
                                     device=self.csr.indices.device)
        elif order == "accurate":
            self._yref = None
        check(_native.load().lk_als_plan_set_rhs_workspace(
            self._h, _ptr(self._yref) if self._yref is not None else None),
            "lk_als_plan_set_rhs_workspace")

    def long_rows(self) -> int:
        "rows of the plan that are pre-reduced in chunks (the first tasks of the longest-first order)"
        return int(_native.load().lk_als_plan_long_rows(self._h))

    def yref_tasks(self) -> "torch.Tensor | None":
        """Hybrid plans: the right-hand sides the last half-epoch formed in the reference's order,
        [long_rows x KP], row t = the t-th longest row (a view of the workspace); else None."""
        lib = _native.load()
        ptr = lib.lk_als_plan_yref(self._h, _ptr(self.ws))
        if not ptr:
            return None
        off = int(ptr) - int(self.ws.data_ptr())
        n = self.long_rows() * self.kp
        return self.ws[off : off + 4 * n].view(torch.float32).view(-1, self.kp)

    def share_z_from(self, leader: "ALSPlan"):
        """
        Row slices of one half-epoch share one Z = other @ OtOr^-1: ``leader`` (launched first in
        every half-epoch) owns the buffer and forms Z, this plan reads it and copies the leader's
        "OtOr is not positive definite" flag at every launch (lk_als_plan_set_z_shared).
        """
        self._z_leader = leader
        check(_native.load().lk_als_plan_set_z_leader(leader._h, 1), "lk_als_plan_set_z_leader")

    def set_external_z(self, z: torch.Tensor, flag: torch.Tensor):
        """
        Z = other @ OtOr^-1 formed OUTSIDE the plan for every half-epoch (:class:`ShardedZ`: each
        rank forms its share of the rows, one all-gather; ``LK_ALS_Z=sharded``): the plan reads
        ``z`` and copies ``flag`` ("OtOr is not positive definite") into its status word at every
        launch (lk_als_plan_set_z_shared) instead of running the n_cols x KP x KP GEMM itself.
        """
        assert z.shape[1] == self.kp and z.shape[0] == self.csr.shape[1] and z.is_contiguous()
        self._z_ext = (z, flag)  # kept alive with the plan
        check(_native.load().lk_als_plan_set_z_shared(self._h, _ptr(z), _ptr(flag)),
              "lk_als_plan_set_z_shared")
        self._z_shared_set = True

    def set_ctl(self, ctl: "TaskCtl | None"):
        "Attach (or detach) a cancel / progress block; check_status then reports a cancel."
        self._ctl = ctl  # keep it alive as long as the plan refers to it
        check(_native.load().lk_als_plan_set_ctl(self._h, ctl._h if ctl is not None else None))

    def set_cg(self, tol: float, max_iter: int = 0):
        check(_native.load().lk_als_plan_set_cg(self._h, float(tol), int(max_iter)))

    def cg_stats(self):
        "(CG iterations, non-empty rows) of the last CG half-epoch on this plan (blocking)"
        import ctypes

        it, rows = ctypes.c_int64(0), ctypes.c_int64(0)
        check(_native.load().lk_als_plan_cg_stats(self._h, _ptr(self.ws), _stream(),
                                                   ctypes.byref(it), ctypes.byref(rows)))
        return int(it.value), int(rows.value)

    def half_epoch(self, this: torch.Tensor, other: torch.Tensor, otor: torch.Tensor):
        """
        One ALS half-epoch on the current stream; ``this`` ([rows x KP]) is updated in
        place.  Asynchronous: returns the device scalar holding sqrt(sum ||delta||^2).
        """
        csr = self.csr
        assert this.shape == (csr.shape[0], self.kp) and other.shape == (csr.shape[1], self.kp)
        assert this.is_contiguous() and other.is_contiguous() and otor.is_contiguous()
        if self.use_wb and getattr(self, "_z_ext", None) is not None:
            pass  # Z arrives from outside (set_external_z)
        elif self.use_wb and self._z_leader is not None:
            if not self._z_shared_set:
                ld = self._z_leader
                assert ld._z is not None, "the leading slice's half-epoch must be launched first"
                lib = _native.load()
                check(lib.lk_als_plan_set_z_shared(self._h, _ptr(ld._z),
                                                   lib.lk_als_plan_z_flag(ld._h, _ptr(ld.ws))),
                      "lk_als_plan_set_z_shared")
                self._z_shared_set = True
        elif self.use_wb and self._z is None:
            # the library forms Z = other @ OtOr^-1 itself at every half-epoch (OtOr^-1 to float64
            # accuracy on the device, csrc/spd_inverse.hip; Z on the scoring GEMM): this side only
            # lends it the [n_cols x KP] buffer -- no library factorisation, no host round trip
            self._z = torch.empty((csr.shape[1], self.kp), dtype=torch.float32,
                                  device=other.device)
            check(_native.load().lk_als_plan_set_z_workspace(self._h, _ptr(self._z)),
                  "lk_als_plan_set_z_workspace")
        check(
            _native.load().lk_als_implicit_half_epoch(
                self._h, _ptr(csr.indptr), _ptr(csr.indices), _ptr(csr.values),
                csr.shape[0], csr.shape[1], self.k, _ptr(this), self.kp, _ptr(other), self.kp,
                _ptr(otor), otor.stride(0), _ptr(self.ws), _ptr(self.frob), _stream()
            ),
            "lk_als_implicit_half_epoch",
        )  # fmt: skip
        return self.frob

    def half_epoch_explicit(self, this: torch.Tensor, other: torch.Tensor, reg: float):
        """
        One explicit-feedback (biased-MF) half-epoch (lk_als_explicit_half_epoch;
        src/accel/als/explicit.rs:33-119): the CSR values are the bias-normalised ratings.
        """
        csr = self.csr
        assert this.shape == (csr.shape[0], self.kp) and other.shape == (csr.shape[1], self.kp)
        assert this.is_contiguous() and other.is_contiguous()
        check(
            _native.load().lk_als_explicit_half_epoch(
                self._h, _ptr(csr.indptr), _ptr(csr.indices), _ptr(csr.values),
                csr.shape[0], csr.shape[1], self.k, _ptr(this), self.kp, _ptr(other), self.kp,
                float(np.float32(reg)), _ptr(self.ws), _ptr(self.frob), _stream()
            ),
            "lk_als_explicit_half_epoch",
        )  # fmt: skip
        return self.frob

    def enable_timing(self, enable: bool = True):
        check(_native.load().lk_als_plan_enable_timing(self._h, 1 if enable else 0))

    def get_timing(self):
        "-> (ms in the chunk kernel, ms in the solve kernel, half-epochs recorded); resets."
        a, b, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int32(0)
        check(
            _native.load().lk_als_plan_get_timing(
                self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)
            )
        )
        return a.value, b.value, n.value

    def check_status(self):
        "Synchronise and raise RuntimeError('ALS solve error: ...') on a failed solve."
        check(_native.load().lk_als_check_status(self._h, _ptr(self.ws), _stream()))

    def __del__(self):
        try:
            if self._h:
                _native.load().lk_als_plan_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


class ShardedZ:
    """
    Z = other @ OtOr^-1 (the Woodbury kernels' operand at padded k = 128 / 256) formed ONCE ACROSS
    THE RANKS instead of once per rank: every rank inverts the k x k OtOr (spd_inverse.hip,
    microseconds), multiplies its own 1 / world share of the rows of ``other`` (the scoring GEMM)
    and the shares are all-gathered in place.  With the default (``LK_ALS_Z=replicated``) every
    rank runs the whole n_cols x KP x KP GEMM -- work that does not shrink with the number of
    GPUs (the cap on cfg5's scaling); sharded it costs one more all-gather of
    the size of the factor gather.  Same bits either way: a row of Z depends on that row of
    ``other`` and on OtOr^-1 only.
    """

    def __init__(self, n_rows: int, k: int, dev):
        lib = _native.require_gpu()
        self.k, self.kp = int(k), padded_dim(k)
        assert self.kp in (128, 256)
        self.z = torch.empty((n_rows, self.kp), dtype=torch.float32, device=dev)
        self.ginv = torch.empty((self.kp, self.kp), dtype=torch.float32, device=dev)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ws = torch.empty(lib.lk_spd_inverse_workspace_bytes(self.k), dtype=torch.uint8,
                              device=dev)

    def form(self, other: torch.Tensor, otor: torch.Tensor, rank: int, world: int, comm):
        lib = _native.load()
        n = other.shape[0]
        assert n == self.z.shape[0] and n % world == 0 and other.is_contiguous()
        self.flag.zero_()
        check(lib.lk_spd_inverse(_ptr(otor), otor.stride(0), self.k, _ptr(self.ginv),
                                 _ptr(self.flag), _ptr(self.ws), _stream()), "lk_spd_inverse")
        share = n // world
        lo, hi = rank * share, (rank + 1) * share
        if share:
            check(lib.lk_score_dense(_ptr(other[lo:hi]), self.kp, share, _ptr(self.ginv), self.kp,
                                     self.kp, self.k, _ptr(self.z[lo:hi]), self.kp, _stream()),
                  "lk_score_dense")
            comm.all_gather_rows(self.z, lo, hi)
        return self.z, self.flag


class ALSPlanGroup:
    """
    The plans of the ROW SLICES of one orientation on one rank (the sharded engine cuts a rank's
    rows into slices so that the all-gather of one slice runs under the solve of the next),
    presented as one plan to whoever only asks about it (bench / tests): timing and statistics
    are summed, settings go to every slice.  The slices share one Z (the first slice leads).
    """

    def __init__(self, plans: list[ALSPlan], n_cols: int):
        from types import SimpleNamespace

        assert len(plans) >= 1
        self.plans = plans
        p0 = plans[0]
        self.k, self.kp, self.solver = p0.k, p0.kp, p0.solver
        lens = np.concatenate([np.diff(p.csr.h_indptr) for p in plans])
        h = np.zeros(len(lens) + 1, dtype=p0.csr.h_indptr.dtype)
        np.cumsum(lens, out=h[1:])
        # the rank's rows in slice order (row lengths / shapes only: the slices are not contiguous)
        self.csr = SimpleNamespace(h_indptr=h, indices=p0.csr.indices, values=p0.csr.values,
                                   shape=(len(lens), n_cols),
                                   full_h_indptr=getattr(p0.csr, "full_h_indptr", None))
        self.short_rows = sum(p.short_rows for p in plans)
        self.woodbury_rows = sum(p.woodbury_rows for p in plans)
        wb_min = int(os.environ.get("LK_ALS_WB_MIN_ROWS", "4096"))
        # the Woodbury decision belongs to the half-epoch, not to a slice of it
        use_wb = (64 < self.kp <= 256 and self.solver == _native.SOLVER_CHOLESKY and wb_min > 0
                  and self.woodbury_rows >= wb_min)
        if use_wb:
            # the slices are views into ONE values array (offsets are not rebased): scan each
            # distinct array once, whether or not a slice had reason to look on its own
            seen = {}
            for p in plans:
                v = p.csr.values
                key = None if v is None else (v.data_ptr(), v.numel())
                if key not in seen:
                    seen[key] = p.negative_values
                else:
                    p._negative_values = seen[key]
            use_wb = not any(seen.values())
        for p in plans:
            p.use_wb = use_wb
        for p in plans[1:]:
            p.share_z_from(p0)

    @property
    def use_wb(self) -> bool:
        return bool(self.plans[0].use_wb)

    def set_cg(self, tol: float, max_iter: int = 0):
        for p in self.plans:
            p.set_cg(tol, max_iter)

    def set_rhs_order(self, order: str):
        for p in self.plans:
            p.set_rhs_order(order)

    def cg_stats(self):
        st = [p.cg_stats() for p in self.plans]
        return sum(s[0] for s in st), sum(s[1] for s in st)

    def set_ctl(self, ctl):
        for p in self.plans:
            p.set_ctl(ctl)

    def enable_timing(self, enable: bool = True):
        for p in self.plans:
            p.enable_timing(enable)

    def get_timing(self):
        "-> (chunk ms, solve ms summed over the slices, half-epochs recorded)"
        ts = [p.get_timing() for p in self.plans]
        return sum(t[0] for t in ts), sum(t[1] for t in ts), ts[0][2]

    def check_status(self):
        for p in self.plans:
            p.check_status()


def iknn_build(ui: DeviceCSR, iu: DeviceCSR, min_sim: float, save_nbrs=None,
               rows: tuple[int, int] | None = None, ctl: "TaskCtl | None" = None,
               timing: dict | None = None) -> DeviceCSR:
    """
    Item-item similarity build (lk_iknn_build_count / _fill, then lk_iknn_truncate_* when
    ``save_nbrs`` is set): ``ui`` users x items and ``iu`` items x users hold the normalised
    ratings; returns the similarity matrix as a device CSR with int64 offsets, rows sorted
    by column.  ``rows = (begin, end)`` builds only that block of output rows (one rank's
    shard: rows are independent, no collective); the result then has ``end - begin`` rows.
    """
    lib = _native.require_gpu()
    n_users, n_items = ui.shape
    r0, r1 = (0, n_items) if rows is None else (int(rows[0]), int(rows[1]))
    n_rows = r1 - r0
    assert iu.shape == (n_items, n_users)
    assert ui.h_indptr.dtype == iu.h_indptr.dtype
    dev = ui.indices.device
    is64 = 1 if ui.h_indptr.dtype == np.int64 else 0
    h = ctypes.c_void_p(0)
    check(
        lib.lk_iknn_plan_create_rows(
            ctypes.byref(h), ui.h_indptr.ctypes.data_as(ctypes.c_void_p),
            iu.h_indptr.ctypes.data_as(ctypes.c_void_p), is64, n_users, n_items, r0, r1
        ),
        "lk_iknn_plan_create_rows",
    )  # fmt: skip
    try:
        if ctl is not None:
            check(lib.lk_iknn_plan_set_ctl(h, ctl._h), "lk_iknn_plan_set_ctl")
        if timing is not None:
            check(lib.lk_iknn_plan_enable_timing(h, 1), "lk_iknn_plan_enable_timing")
        ws = torch.empty(lib.lk_iknn_plan_workspace_bytes(h), dtype=torch.uint8, device=dev)
        out_ptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
        total = ctypes.c_int64(0)
        ms = float(np.float32(min_sim))  # cast to f32 at the boundary (item_train.rs:37)
        check(
            lib.lk_iknn_build_count(
                h, _ptr(ui.indptr), _ptr(ui.indices), _ptr(ui.values), _ptr(iu.indptr),
                _ptr(iu.indices), _ptr(iu.values), ms, -1, _ptr(ws), _ptr(out_ptr),
                ctypes.byref(total), _stream()
            ),
            "lk_iknn_build_count",
        )  # fmt: skip
        nnz = int(total.value)
        out_idx = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
        out_val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)[:nnz]
        check(
            lib.lk_iknn_build_fill(
                h, _ptr(ui.indptr), _ptr(ui.indices), _ptr(ui.values), _ptr(iu.indptr),
                _ptr(iu.indices), _ptr(iu.values), ms, -1, _ptr(ws), _ptr(out_ptr),
                _ptr(out_idx), _ptr(out_val), _stream()
            ),
            "lk_iknn_build_fill",
        )  # fmt: skip
        torch.cuda.current_stream().synchronize()
        if timing is not None:
            ms, nl = ctypes.c_double(0), ctypes.c_int32(0)
            check(lib.lk_iknn_plan_get_timing(h, ctypes.byref(ms), ctypes.byref(nl)))
            timing["build_kernel_ms"] = ms.value
            timing["build_kernel_launches"] = nl.value
    finally:
        lib.lk_iknn_plan_destroy(h)
    full = DeviceCSR(out_ptr, out_idx, out_val, (n_rows, n_items), None)
    if save_nbrs is None or int(save_nbrs) <= 0:
        return full
    # item_train.rs:139-151: per-row top-save_nbrs, ties in order of first encounter
    del ws
    tws = torch.empty(lib.lk_iknn_truncate_workspace_bytes(n_rows, nnz), dtype=torch.uint8,
                      device=dev)
    new_ptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    check(
        lib.lk_iknn_truncate_count(
            _ptr(full.indptr), _ptr(full.indices), _ptr(full.values), _ptr(iu.indptr), is64,
            _ptr(iu.indices), n_rows, r0, nnz, int(save_nbrs), _ptr(tws), _ptr(new_ptr),
            ctypes.byref(total), _stream()
        ),
        "lk_iknn_truncate_count",
    )  # fmt: skip
    nnz2 = int(total.value)
    t_idx = torch.empty(max(nnz2, 1), dtype=torch.int32, device=dev)[:nnz2]
    t_val = torch.empty(max(nnz2, 1), dtype=torch.float32, device=dev)[:nnz2]
    check(
        lib.lk_iknn_truncate_fill(
            _ptr(full.indptr), _ptr(full.indices), _ptr(full.values), n_rows, nnz, _ptr(tws),
            _ptr(new_ptr), _ptr(t_idx), _ptr(t_val), _stream()
        ),
        "lk_iknn_truncate_fill",
    )  # fmt: skip
    torch.cuda.current_stream().synchronize()
    return DeviceCSR(new_ptr, t_idx, t_val, (n_rows, n_items), None)


def score_topk(users: torch.Tensor, items: torch.Tensor, k: int, n: int,
               excl_ptr: torch.Tensor | None = None, excl_items: torch.Tensor | None = None):
    """
    Batched dense scoring + top-N (lk_score_topk): ``users`` [B x KP], ``items`` [I x KP]
    padded device matrices; exclusion CSR (int64 offsets, int32 items) optional.
    Returns (indices int32 [B x n] with -1 padding, scores f32 [B x n] with NaN padding).
    """
    lib = _native.require_gpu()
    B, kp = users.shape
    I = items.shape[0]
    assert kp == padded_dim(k) and items.shape[1] == kp
    assert users.is_contiguous() and items.is_contiguous()
    dev = users.device
    n = int(n)
    cols = I if n < 0 else n  # n < 0: rank every candidate (TopNRanker without n)
    ws = torch.empty(lib.lk_score_topk_workspace_bytes(B, I, n), dtype=torch.uint8, device=dev)
    out_idx = torch.empty((B, cols), dtype=torch.int32, device=dev)
    out_sc = torch.empty((B, cols), dtype=torch.float32, device=dev)
    if excl_ptr is not None:
        assert excl_ptr.dtype == torch.int64 and excl_items.dtype == torch.int32
    check(
        lib.lk_score_topk(
            _ptr(users), kp, B, _ptr(items), kp, I, int(k), int(n), _ptr(excl_ptr),
            _ptr(excl_items), _ptr(ws), _ptr(out_idx), _ptr(out_sc), _stream()
        ),
        "lk_score_topk",
    )  # fmt: skip
    return out_idx, out_sc


def argtopn(scores: torch.Tensor, n: int) -> torch.Tensor:
    """
    Per-row top-N indices (lk_argtopn) of a [rows x len] f32 device matrix, -1 padding;
    ``n < 0`` ranks every valid entry (``argsort_descending``).  Lists of up to 4096 take the
    selection kernel, longer ones (and ``n < 0``) the full stable sort.
    """
    lib = _native.require_gpu()
    assert scores.dtype == torch.float32 and scores.is_contiguous() and scores.dim() == 2
    rows, ln = scores.shape
    n = int(n)
    cols = ln if n < 0 else min(n, ln)
    out = torch.empty((rows, cols), dtype=torch.int32, device=scores.device)
    if cols == 0 or rows == 0:
        return out
    wb = lib.lk_argtopn_workspace_bytes(rows, ln, n if n < 0 else cols)
    ws = torch.empty(wb, dtype=torch.uint8, device=scores.device) if wb else None
    check(lib.lk_argtopn(_ptr(scores), rows, ln, n if n < 0 else cols, _ptr(ws), _ptr(out),
                         _stream()), "lk_argtopn")
    return out


def iknn_score_batch(sims: DeviceCSR, ref_ptr, ref_items, ref_rates, tgt_ptr, tgt_items,
                     max_nbrs: int, min_nbrs: int):
    """
    Item-kNN scoring of a batch of queries (lk_iknn_score_batch).  ``sims``: similarity CSR
    (int64 offsets); ``ref_*``/``tgt_*``: CSR-style (int64 offsets, int32 items) history and
    target lists, negative items are nulls; ``ref_rates`` f32 (explicit) or None (implicit).
    Returns (scores f32 with NaN for null, counts int32 with -1 for null targets).
    """
    lib = _native.require_gpu()
    n_items = sims.shape[0]
    nq = int(ref_ptr.shape[0]) - 1
    dev = sims.indices.device
    assert sims.indptr.dtype == torch.int64
    ws = torch.empty(lib.lk_iknn_score_workspace_bytes(n_items, nq, int(max_nbrs)),
                     dtype=torch.uint8, device=dev)
    out_s = torch.empty(int(tgt_items.shape[0]), dtype=torch.float32, device=dev)
    out_c = torch.empty(int(tgt_items.shape[0]), dtype=torch.int32, device=dev)
    check(
        lib.lk_iknn_score_batch(
            _ptr(sims.indptr), _ptr(sims.indices), _ptr(sims.values), n_items, nq,
            _ptr(ref_ptr), _ptr(ref_items), _ptr(ref_rates), _ptr(tgt_ptr), _ptr(tgt_items),
            int(max_nbrs), int(min_nbrs), _ptr(ws), _ptr(out_s), _ptr(out_c), _stream()
        ),
        "lk_iknn_score_batch",
    )  # fmt: skip
    return out_s, out_c


def iknn_recommend(sims: DeviceCSR, ref_ptr, ref_items, ref_rates, item_bias, max_nbrs: int,
                   min_nbrs: int, n: int, query_hits: np.ndarray, exclude_refs: bool = True):
    """
    Item-kNN top-``n`` recommendations for a batch of queries (lk_iknn_recommend): every item is
    scored with the reference accumulator's arithmetic (bit for bit), ``item_bias`` (the item
    means of the explicit model, device f32 [n_items], or None) is added, the query's own items
    are struck out (``exclude_refs``), and the ``n`` best scored items are returned.
    ``ref_ptr`` / ``ref_items`` / ``ref_rates``: device CSR-style history lists as for
    :func:`iknn_score_batch`; ``query_hits``: HOST int64 [queries], per query the summed
    similarity-row lengths of its history items (``item_counts[ref_items]`` summed per query).
    Returns device (item numbers int32 [B x n], -1 padded; scores f32 [B x n], NaN padded).
    """
    lib = _native.require_gpu()
    n_items = sims.shape[0]
    nq = int(ref_ptr.shape[0]) - 1
    dev = sims.indices.device
    assert sims.indptr.dtype == torch.int64
    query_hits = np.ascontiguousarray(query_hits, dtype=np.int64)
    assert query_hits.shape == (nq,)
    max_hits = int(query_hits.max()) if nq else 0
    cols = n_items if n < 0 else int(n)
    out_i = torch.empty((nq, cols), dtype=torch.int32, device=dev)
    out_s = torch.empty((nq, cols), dtype=torch.float32, device=dev)
    if nq == 0 or cols == 0:
        return out_i, out_s
    ws = torch.empty(lib.lk_iknn_recommend_workspace_bytes(n_items, nq, max_hits, int(max_nbrs),
                                                           int(n)),
                     dtype=torch.uint8, device=dev)
    check(
        lib.lk_iknn_recommend(
            _ptr(sims.indptr), _ptr(sims.indices), _ptr(sims.values), n_items, nq, _ptr(ref_ptr),
            _ptr(ref_items), _ptr(ref_rates), _ptr(item_bias), int(max_nbrs), int(min_nbrs),
            int(n), 1 if exclude_refs else 0, query_hits.ctypes.data_as(ctypes.c_void_p),
            max_hits, _ptr(ws), _ptr(out_i), _ptr(out_s), _stream()
        ),
        "lk_iknn_recommend",
    )  # fmt: skip
    return out_i, out_s


def knn_score_last_stats():
    "(queries on the candidate-list kernel, queries on the slot kernel, longest target list) of the last call"
    import ctypes

    out = (ctypes.c_int64 * 3)()
    _native.load().lk_knn_score_last_stats(out)
    return tuple(int(x) for x in out)


def uknn_score_batch(ratings: DeviceCSR, nbr_ptr, nbr_rows, nbr_sims, tgt_ptr, tgt_items,
                     max_nbrs: int, min_nbrs: int):
    """
    User-kNN scoring of a batch of queries (lk_uknn_score_batch; src/accel/knn/user_score.rs):
    ``ratings`` users x items CSR (int64 offsets; ``values`` None = implicit feedback),
    neighbours / targets as CSR-style (int64 offsets) lists.  Returns (scores, counts).
    """
    lib = _native.require_gpu()
    n_users, n_items = ratings.shape
    nq = int(nbr_ptr.shape[0]) - 1
    dev = ratings.indices.device
    assert ratings.indptr.dtype == torch.int64
    ws = torch.empty(lib.lk_iknn_score_workspace_bytes(n_items, nq, int(max_nbrs)),
                     dtype=torch.uint8, device=dev)
    out_s = torch.empty(int(tgt_items.shape[0]), dtype=torch.float32, device=dev)
    out_c = torch.empty(int(tgt_items.shape[0]), dtype=torch.int32, device=dev)
    check(
        lib.lk_uknn_score_batch(
            _ptr(ratings.indptr), _ptr(ratings.indices), _ptr(ratings.values), n_users, n_items,
            nq, _ptr(nbr_ptr), _ptr(nbr_rows), _ptr(nbr_sims), _ptr(tgt_ptr), _ptr(tgt_items),
            int(max_nbrs), int(min_nbrs), _ptr(ws), _ptr(out_s), _ptr(out_c), _stream()
        ),
        "lk_uknn_score_batch",
    )  # fmt: skip
    return out_s, out_c


def csr_rows_dot(csr: DeviceCSR, x: torch.Tensor) -> torch.Tensor:
    """
    ``out[q][r] = <row r of csr, x[:, q]>`` for the dense columns of ``x`` ([n_cols x B],
    row-major) -- the neighbour similarities of user-kNN (lk_csr_rows_dot).  Returns [B x rows].
    """
    lib = _native.require_gpu()
    n_rows, n_cols = csr.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[0] == n_cols
    B = int(x.shape[1])
    out = torch.empty((B, n_rows), dtype=torch.float32, device=x.device)
    check(
        lib.lk_csr_rows_dot(_ptr(csr.indptr), 1 if csr.is64 else 0, _ptr(csr.indices),
                            _ptr(csr.values), n_rows, _ptr(x), B, B, _ptr(out), n_rows,
                            _stream()),
        "lk_csr_rows_dot",
    )
    return out


def ease_gram(cooc: DeviceCSR, item_counts: torch.Tensor, reg: float) -> torch.Tensor:
    """
    ``X^T X + reg I`` as a dense [n x n] f32 device matrix (lk_ease_gram) from the off-diagonal
    co-occurrence counts (``iknn_build`` of the binary matrix, threshold 0.5) and the item counts.
    """
    lib = _native.require_gpu()
    n = int(cooc.shape[0])
    out = torch.empty((n, n), dtype=torch.float32, device=cooc.indices.device)
    assert cooc.indptr.dtype == torch.int64 and item_counts.dtype == torch.int32
    check(
        lib.lk_ease_gram(_ptr(cooc.indptr), _ptr(cooc.indices), _ptr(cooc.values),
                         _ptr(item_counts), n, float(reg), _ptr(out), n, _stream()),
        "lk_ease_gram",
    )
    return out


def ease_score_batch(hist_ptr: torch.Tensor, hist_items: torch.Tensor,
                     weights: torch.Tensor) -> torch.Tensor:
    "Sum of the history items' weight rows per query, [B x n_items] f32 (lk_ease_score_batch)."
    lib = _native.require_gpu()
    n = int(weights.shape[0])
    B = int(hist_ptr.shape[0]) - 1
    assert hist_ptr.dtype == torch.int64 and hist_items.dtype == torch.int32
    assert weights.dtype == torch.float32 and weights.is_contiguous()
    out = torch.empty((B, n), dtype=torch.float32, device=weights.device)
    for lo in range(0, B, 65535):
        hi = min(B, lo + 65535)
        check(
            lib.lk_ease_score_batch(_ptr(hist_ptr[lo:]), _ptr(hist_items), hi - lo,
                                    _ptr(weights), n, n, _ptr(out[lo:]), n, _stream()),
            "lk_ease_score_batch",
        )
    return out


def score_dense(users: torch.Tensor, items: torch.Tensor, k: int) -> torch.Tensor:
    "All (user, item) scores, [B x I] f32 (lk_score_dense)."
    lib = _native.require_gpu()
    B, kp = users.shape
    I = items.shape[0]
    assert kp == padded_dim(k) and items.shape[1] == kp
    out = torch.empty((B, I), dtype=torch.float32, device=users.device)
    check(
        lib.lk_score_dense(_ptr(users), kp, B, _ptr(items), kp, I, int(k), _ptr(out), I, _stream()),
        "lk_score_dense",
    )
    return out


def fold_in(hist: DeviceCSR, items: torch.Tensor, otor: torch.Tensor, k: int,
            solver: int = _native.SOLVER_AUTO, pending: list | None = None) -> torch.Tensor:
    """
    Batched new-user embeddings (``ImplicitMFScorer._train_new_row``,
    src/lenskit/als/_implicit.py:101-130): one ALS row solve per history row of ``hist``
    (queries x items CSR, values = weight or weight*rating) against ``items`` and
    ``otor`` = Q^T Q + user_reg I.  Returns [n_queries x KP]; empty histories give zeros.
    ``pending``: the launch is left unchecked and its plan appended there -- the caller runs
    ``check_status`` once its own launches are queued behind this one (no host wait in between).

    The plan sums in the kernels' own (``accurate``) order whatever ``LK_ALS_RHS_ORDER`` says for
    training: the hybrid order reproduces the RUST half-epoch's sequential float32 sums
    (src/accel/als/implicit.rs:110-117), but a fold-in is the reference's PYTHON path --
    ``(M.T * ratings) @ M`` and ``M.T @ (ratings + 1)`` through NumPy's BLAS, ``cho_factor`` -- which
    has no such chain.  Measured on 10 000 ML-25M-shaped histories against that restatement
    (``oracle.als_fold_in``): worst row 3.0e-5 / median 2.2e-6 in this order, 4.0e-5 / 2.7e-6 in the
    hybrid one; and the launch is 0.24 instead of 0.29 ms (no chain kernel, no slab sums).
    """
    plan = ALSPlan(hist, k, solver, reference_order="accurate")
    out = torch.zeros((hist.shape[0], plan.kp), dtype=torch.float32, device=items.device)
    plan.half_epoch(out, items, otor)
    if pending is None:
        plan.check_status()
    else:
        pending.append(plan)
    return out


def fold_in_explicit(hist: DeviceCSR, items: torch.Tensor, reg: float, k: int) -> torch.Tensor:
    """
    Batched new-user embeddings of the biased-MF model (``_train_bias_row_cholesky``,
    src/lenskit/als/_explicit.py:121-149): one explicit row solve per history row of ``hist``
    (queries x items CSR, values = bias-normalised ratings).  Returns [n_queries x KP].
    """
    plan = ALSPlan(hist, k, _native.SOLVER_CHOLESKY, reference_order="accurate")  # (as fold_in)
    out = torch.zeros((hist.shape[0], plan.kp), dtype=torch.float32, device=items.device)
    plan.half_epoch_explicit(out, items, reg)
    plan.check_status()
    return out


def gather_rows(csr: DeviceCSR, rows: np.ndarray, *, scale: float = 1.0,
                with_values: bool = True, col_bias: torch.Tensor | None = None) -> DeviceCSR:
    """
    The rows ``rows`` (HOST int32, -1 = an empty row) of a device CSR as a new device CSR with
    int64 offsets (lk_csr_gather_rows): the histories of a batch of queries cut out of the
    training matrix in one launch (src/lenskit/basic/history.py:37-95 per query in the
    reference).  Values are ``csr.values * scale`` in float32, or the constant ``scale`` when the
    matrix holds none (src/lenskit/als/_implicit.py:83-92); ``col_bias`` (device f32 per column)
    is subtracted before the scaling (item.py:268-271); ``with_values=False``: structure only.  The offsets are prefix sums of the host copy of ``csr``'s offsets: a few thousand
    integers on the host, nothing of the size of the matrix.
    """
    lib = _native.require_gpu()
    hp = csr.h_indptr
    assert hp is not None, "gather_rows needs the host copy of the offsets"
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    B = int(rows.shape[0])
    safe = np.where(rows >= 0, rows, 0)
    lens = np.where(rows >= 0, hp[safe + 1] - hp[safe], 0).astype(np.int64)
    ptr = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    dev = csr.indices.device
    nnz = int(ptr[-1])
    # one upload for the two small host arrays (offsets, row numbers)
    packed = np.empty(2 * (B + 1) + B + (B & 1), dtype=np.int32)
    packed[:2 * (B + 1)] = ptr.view(np.int32)
    packed[2 * (B + 1):2 * (B + 1) + B] = rows
    d_packed = torch.from_numpy(packed).to(dev)
    d_ptr = d_packed[:2 * (B + 1)].view(torch.int64)
    d_rows = d_packed[2 * (B + 1):2 * (B + 1) + B]
    out_idx = torch.empty(nnz, dtype=torch.int32, device=dev)
    out_val = torch.empty(nnz, dtype=torch.float32, device=dev) if with_values else None
    if B and nnz:
        check(
            lib.lk_csr_gather_rows(_ptr(csr.indptr), 1 if csr.is64 else 0, _ptr(csr.indices),
                                   _ptr(csr.values), B, _ptr(d_rows), _ptr(d_ptr), _ptr(col_bias),
                                   float(np.float32(scale)), _ptr(out_idx), _ptr(out_val),
                                   _stream()),
            "lk_csr_gather_rows",
        )
    return DeviceCSR(d_ptr, out_idx, out_val, (B, csr.shape[1]), ptr)


def csr_transpose(csr: DeviceCSR, with_values: bool = True) -> DeviceCSR:
    """
    Stable transpose on the device (lk_csr_transpose; ``SparseRowArray.transpose`` /
    ``_accel.data.transpose_csr``, src/lenskit/data/matrix.py:512-530,
    src/accel/data/transpose.rs:19-108): entries of an output row keep the input's entry
    order, i.e. ascending source row.  Offsets keep the input's width.
    """
    lib = _native.require_gpu()
    n_rows, n_cols = csr.shape
    nnz = int(csr.indices.shape[0])
    dev = csr.indices.device
    is64 = 1 if csr.indptr.dtype == torch.int64 else 0
    wb = lib.lk_csr_transpose_workspace_bytes(nnz, n_cols, is64)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    t_ptr = torch.empty(n_cols + 1, dtype=csr.indptr.dtype, device=dev)
    t_idx = torch.empty(nnz, dtype=torch.int32, device=dev)
    perm = torch.empty(nnz, dtype=csr.indptr.dtype, device=dev) if with_values else None
    check(
        lib.lk_csr_transpose(_ptr(csr.indptr), is64, _ptr(csr.indices), n_rows, n_cols, nnz,
                             _ptr(t_ptr), _ptr(t_idx), _ptr(perm), _ptr(ws), wb, _stream()),
        "lk_csr_transpose",
    )
    vals = csr.values[perm.long()] if with_values and csr.values is not None else None
    out = DeviceCSR(t_ptr, t_idx, vals, (n_cols, n_rows), None)
    out.perm = perm
    return out


def iknn_prepare(ratings, explicit: bool = True, dev=None):
    """
    Item-kNN rating normalisation (``ItemKNNScorer._center_ratings`` / ``_normalize_rows``,
    src/lenskit/knn/item.py:202-228) with the data on the device: ``ratings`` is the
    users x items matrix (SciPy; for implicit feedback the caller passes the interaction
    matrix of ones, like the reference) and ``explicit`` selects the item-mean centring.
    Returns (ui DeviceCSR, iu DeviceCSR, item means | None, all_zero flag) -- the two
    orientations the similarity build consumes -- bit-identical to the reference's SciPy
    preparation: the structure comes from the stable device transpose, every elementwise
    step and the sequential sum of squares from csrc/iknn_prepare.hip, and the two per-item
    vectors whose rounding depends on the host's NumPy (``np.add.reduceat`` sums, sqrt /
    reciprocal of the norms) from the very calls the reference makes, on [n_items] arrays.
    """
    import scipy.sparse as sps

    lib = _native.require_gpu()
    dev = device(dev)
    csr = sps.csr_array(ratings).astype(np.float32)
    csr.sort_indices()
    n_users, n_items = csr.shape
    dcsr = DeviceCSR.from_arrays(csr.indptr, csr.indices, csr.data, csr.shape, dev)
    t = csr_transpose(dcsr)  # item-major values, offsets, users, permutation
    is64 = 1 if t.indptr.dtype == torch.int64 else 0
    t_ptr = t.indptr.cpu().numpy()
    counts = np.diff(t_ptr)
    means = d_means = None
    if explicit:
        # rmat.sum(axis=0) on the CSC matrix == np.add.reduceat over the non-empty items
        # (scipy.sparse._compressed._cs_matrix.sum -> _minor_reduce), in the host's NumPy
        vals_items = t.values.cpu().numpy()
        nonempty = np.flatnonzero(counts)
        sums = np.zeros(n_items, dtype=np.float32)
        if len(nonempty):
            sums[nonempty] = np.add.reduceat(vals_items, t_ptr[nonempty])
        means = np.zeros(n_items, dtype=np.float32)
        np.divide(sums, counts, out=means, where=counts > 0)
        d_means = torch.from_numpy(means).to(dev)
    nnz = t.nnz
    cent = torch.empty(nnz, dtype=torch.float32, device=dev)
    sumsq = torch.empty(n_items, dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    check(
        lib.lk_iknn_prep_center(_ptr(t.indptr), is64, _ptr(t.values), _ptr(d_means), n_items,
                                _ptr(cent), _ptr(sumsq), _ptr(flag), _stream()),
        "lk_iknn_prep_center",
    )
    norms = np.sqrt(sumsq.cpu().numpy())  # spla.norm(rmat, 2, axis=0)
    recip = np.true_divide(1.0, np.maximum(norms, np.finfo("f4").smallest_normal))
    d_recip = torch.from_numpy(np.ascontiguousarray(recip, dtype=np.float32)).to(dev)
    v_items = torch.empty(nnz, dtype=torch.float32, device=dev)
    v_users = torch.empty(nnz, dtype=torch.float32, device=dev)
    check(
        lib.lk_iknn_prep_scale(_ptr(t.indptr), is64, _ptr(t.perm), _ptr(cent), _ptr(d_recip),
                               n_items, _ptr(v_items), _ptr(v_users), _stream()),
        "lk_iknn_prep_scale",
    )
    all_zero = explicit and int(flag.item()) == 0
    ui = DeviceCSR(dcsr.indptr, dcsr.indices, v_users, (n_users, n_items), dcsr.h_indptr)
    iu = DeviceCSR(t.indptr, t.indices, v_items, (n_items, n_users), t_ptr)
    return ui, iu, means, all_zero
