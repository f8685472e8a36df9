"""
``run_accel_task`` / ``AccelTask`` protocol -- mirror of src/lenskit/parallel/_task.py:25-147
and the Rust pyclass src/accel/tasks/mod.rs:33-106: ``invoke(*, pool=None)`` runs ONCE on a
helper thread, ``cancel()`` and ``current_progress()`` are called from the main thread while
it runs, failures are re-raised as ``RuntimeError("accelerator task failed with exception")``.
"""

from __future__ import annotations

import threading
from typing import Any, Callable, Generic, TypeVar

R = TypeVar("R")


class AccelTask(Generic[R]):
    """
    A unit of GPU work with the reference's task protocol (src/accel/tasks/mod.rs:33-106).

    ``invoke`` runs the work (once, on a helper thread; the C-ABI calls release the GIL like
    ``py.detach`` in implicit.rs:75).  ``cancel()`` and ``current_progress()`` may be called
    from the main thread meanwhile: they go through the device-visible control words of
    ``lk_task_ctl`` (``include/lkamd.h``) -- the running kernels poll the cancel word and skip
    every row not yet started; the live row count is read from pinned host memory.  A
    cancelled ``invoke`` raises ``KeyboardInterrupt`` (``LK_E_CANCELLED``).
    """

    def __init__(self, fn: Callable[["AccelTask"], R], total: int | None = None):
        self._fn = fn
        self._cancel = threading.Event()
        self._done = 0
        self._total = total
        self._invoked = False
        self._ctl = None  # lkpy_amd._device.TaskCtl, created by the work function

    def invoke(self, *, pool=None) -> R:
        if self._invoked:
            raise RuntimeError("task already invoked")
        self._invoked = True
        if self._cancel.is_set():
            raise KeyboardInterrupt("cancelled")
        return self._fn(self)

    def attach(self, ctl) -> None:
        "called by the work function once the device control block exists"
        self._ctl = ctl
        if self._cancel.is_set():
            ctl.cancel()

    def cancel(self) -> None:
        "tasks/mod.rs:88-95; seen by the running kernels, not only between launches"
        self._cancel.set()
        ctl = self._ctl
        if ctl is not None:
            ctl.cancel()

    @property
    def cancelled(self) -> bool:
        return self._cancel.is_set()

    def set_progress(self, done: int):
        self._done = done

    def current_progress(self):
        "tasks/mod.rs:97-105: rows completed (and the total, when known)"
        ctl = self._ctl
        done = self._done
        if ctl is not None:
            try:
                done = max(done, ctl.progress()[0])
            except Exception:
                pass
        return (done, self._total) if self._total is not None else done


class AccelTaskThread(threading.Thread, Generic[R]):
    def __init__(self, task: AccelTask[R]):
        super().__init__(name="lkpy-amd-accel-task", daemon=True)
        self.task = task
        self.result: Any = None
        self.error: BaseException | None = None

    def run(self):
        try:
            self.result = self.task.invoke()
        except BaseException as e:  # noqa: BLE001
            self.error = e


def run_accel_task(task: AccelTask[R], progress=None) -> R:
    thread = AccelTaskThread(task)
    thread.start()
    try:
        while thread.is_alive():
            thread.join(0.2)  # the reference polls at 5 Hz (_task.py:34-40)
            if progress is not None:
                cur = task.current_progress()
                done = cur[0] if isinstance(cur, tuple) else cur
                try:
                    progress.update(completed=done)
                except Exception:
                    pass
    except KeyboardInterrupt:
        task.cancel()  # _task.py:54-57
        thread.join()
        raise
    if thread.error is not None:
        raise RuntimeError("accelerator task failed with exception") from thread.error
    return thread.result
