"""Where the host time of the batch recommend call goes (bench.py's `recommend` leg): every piece of
``scorer.recommend_batch(lookup.batch(users), n)`` timed on the host with a device synchronisation
after it, on the ML-25M-shaped model (a few epochs are enough: the times do not depend on the
factors).  python tools/recommend_profile.py [users]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from lkpy_amd import _device as D  # noqa: E402
from lkpy_amd import synth  # noqa: E402
from lkpy_amd.als import ImplicitMFScorer  # noqa: E402
from lkpy_amd.basic import UserTrainingHistoryLookup  # noqa: E402
from lkpy_amd.data import Dataset, Vocabulary  # noqa: E402
from lkpy_amd.training import TrainingOptions  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
SHORT = len(sys.argv) > 2 and sys.argv[2] == "short"  # profiling runs: the whole call only
r = synth.ml25m_like()
nu, ni = r.shape
rows = np.repeat(np.arange(nu, dtype=np.int32), np.diff(r.indptr))
ds = Dataset(Vocabulary(np.arange(nu), "user", reorder=False),
             Vocabulary(np.arange(ni), "item", reorder=False), rows, r.indices, {"rating": r.data})
sc = ImplicitMFScorer(embedding_size=64, epochs=5, weight=40.0)
sc.train(ds, TrainingOptions(rng=42))
lk = UserTrainingHistoryLookup()
lk.train(ds)
users = np.random.default_rng(11).choice(ds.users.ids(), B, replace=False)
sc.recommend_batch(lk.batch(users[:256]), 100)
st = sc._device_state()
sync = torch.cuda.synchronize


def t(label, fn, reps=7):
    best, res = 1e9, None
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        res = fn()
        sync()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:38s} {best * 1e3:8.3f} ms", flush=True)
    return res


if SHORT:
    t("whole call", lambda: sc.recommend_batch(lk.batch(users), 100), reps=10)
    sys.exit(0)
hb = t("lookup.batch (vocabulary, lengths)", lambda: lk.batch(users))
src = lk._device_matrix()["csr"]
src = D.DeviceCSR(src.indptr, src.indices, None, src.shape, src.h_indptr)
hist = t("gather_rows (host prefix + kernel)", lambda: D.gather_rows(src, hb.user_nums, scale=40.0))
plan = t("ALSPlan(hist)", lambda: D.ALSPlan(hist, 64, reference_order="accurate"))
import ctypes  # noqa: E402

from lkpy_amd import _native  # noqa: E402

lib = _native.load()
hp = hist.h_indptr


def raw_plan():
    h = ctypes.c_void_p(0)
    lib.lk_als_plan_create_ex(ctypes.byref(h), hp.ctypes.data_as(ctypes.c_void_p), 1, len(hp) - 1,
                              64, 2, 2)
    return h


h = t("  lk_als_plan_create_ex alone", raw_plan)
t("  lk_als_plan_destroy", lambda: lib.lk_als_plan_destroy(raw_plan()))
wsb = lib.lk_als_plan_workspace_bytes(h)
print("  workspace bytes", wsb)
t("  torch.empty(workspace)", lambda: torch.empty(wsb, dtype=torch.uint8, device=st["device"]))
t("  torch.zeros(1)", lambda: torch.zeros(1, dtype=torch.float32, device=st["device"]))
t("  torch.zeros(B x 64)", lambda: torch.zeros((B, 64), dtype=torch.float32, device=st["device"]))
u = torch.zeros((B, plan.kp), dtype=torch.float32, device=st["device"])
t("plan.half_epoch", lambda: plan.half_epoch(u, st["Q"], st["OtOr"]))
t("plan.check_status", lambda: plan.check_status())
for split in ("1", "0"):
    import os
    os.environ["LK_TOPK_SPLIT"] = split
    idx, scv = t(f"score_topk (LK_TOPK_SPLIT={split})",
                 lambda: D.score_topk(u, st["Q"], 64, 100, hist.indptr, hist.indices))
os.environ["LK_TOPK_SPLIT"] = "1"
t("cat + to_host", lambda: D.to_host(torch.cat([idx.view(torch.float32), scv], dim=1)))
t("whole call", lambda: sc.recommend_batch(lk.batch(users), 100))
if len(sys.argv) > 2 and sys.argv[2] == "longexcl":
    import os
    for le in ("4096", "1024", "256", "64", "1"):
        os.environ["LK_TOPK_LONG_EXCL"] = le
        t(f"whole call, LK_TOPK_LONG_EXCL={le}", lambda: sc.recommend_batch(lk.batch(users), 100))
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "cprofile":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        sc.recommend_batch(lk.batch(users), 100)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    sys.exit(0)
for b2 in (1000, 2000, 5000, 20000, 50000):
    if b2 <= nu:
        us = np.random.default_rng(1).choice(ds.users.ids(), b2, replace=False)
        t(f"whole call, {b2} users", lambda: sc.recommend_batch(lk.batch(us), 100), reps=4)
