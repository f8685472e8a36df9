"""
GPU: the C ABI used the way INTEGRATION.md tells a maintainer to use it.

``tests/host_abi_driver.py`` extracts the binding sketch from INTEGRATION.md, executes it verbatim in
a FRESH interpreter (raw ``ctypes`` on ``_lkamd.so``; neither ``lkpy_amd`` nor torch is imported
there) and calls ``train_implicit_matrix`` -- i.e. ``lk_als_implicit_half_epoch_host_ctl``, the
host-pointer entry with the reference's argument list (src/accel/als/implicit.rs:35-84) and task
controls (src/accel/tasks/mod.rs:62-106) -- at k = 25 / 64 / 128 / 256 with 32- and 64-bit offsets,
an empty row, a 5000-entry row (hybrid summation order), a non-positive-definite ``otor``, and
cancel + progress from another thread; the CPU oracle checks every row at 1e-4.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
RTOL = 1.0e-4


@pytest.fixture(scope="module")
def report(gpu):
    env = dict(os.environ, LK_AMD_LIBRARY=str(ROOT / "lkpy_amd" / "_lkamd.so"))
    env.pop("LK_ALS_RHS_ORDER", None)
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "host_abi_driver.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    rep = json.loads(line)
    print("\nhost C-ABI through INTEGRATION.md's sketch:", json.dumps(rep)[:3000])
    return rep


def test_sketch_solves_every_k_and_offset_width(report):
    assert [(c["k"], c["offsets"]) for c in report["cases"]] == [(25, 32), (64, 64), (128, 32),
                                                                  (256, 64)]
    for c in report["cases"]:
        assert c["row_rel_max"] < RTOL, c          # every row, the long one included
        assert c["long_row_rel"] < 3e-5, c         # (hybrid order: an order inside the tolerance)
        assert c["empty_row_zero"], c              # implicit.rs:98-101
        assert abs(c["frob"] - c["want_frob"]) <= 1e-4 * c["want_frob"], c
        assert c["progress"] == [c["rows"], c["rows"]], c   # rows done = rows, at the end


def test_plain_host_entry_takes_the_woodbury_kernels(report):
    w = report["woodbury_host"]
    assert w["rc"] == 0, w
    assert w["row_rel_max"] < RTOL, w


def test_errors_are_the_references(report):
    assert report["not_spd"].startswith("ALS solve error"), report["not_spd"]   # implicit.rs:79
    assert report["type_error"] is True


def test_progress_and_cancel_through_the_task_object(report):
    pr = report["progress"]
    assert pr["final"] == [120_000, 120_000] and pr["monotone"] and pr["finite"], pr
    assert pr["frob"] is not None and pr["frob"] > 0
    ca = report["cancel"]
    # the race is real (a fast box may finish first); when the cancel won the call says so, has
    # stopped early, and the rows it did solve were written back in place
    if ca["error"]:
        assert ca["error"] == ["KeyboardInterrupt"], ca
        assert 0 < ca["rows_done"] < ca["rows_total"], ca
        assert 0 < ca["rows_written"] < ca["rows_total"], ca
    else:
        assert ca["rows_done"] == ca["rows_total"], ca
