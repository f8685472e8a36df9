#!/usr/bin/env python3
"""The cfg3 batch-score leg alone (10 000 users x 100 targets, `lk_iknn_score_batch`): time, and with
--parity the bench's oracle check.  LK_KNN_SCORE_COMPACT=0: the general table sizes."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parity", action="store_true")
    args = ap.parse_args()
    import torch

    import bench
    from lkpy_amd import _device as D
    from lkpy_amd import _knn_bench, synth

    dev = torch.device("cuda:0")
    ratings = synth.ml25m_like()
    dui, diu, means, _ = D.iknn_prepare(ratings, True, dev)
    sims = D.iknn_build(dui, diu, 1.0e-6, 100)
    kw = {}
    if args.parity:
        kw["checker"] = getattr(bench, "knn_score_cpu_and_parity", None)
    res = _knn_bench._score_batch_leg(D, ratings, means, sims, dev, **{k: v for k, v in kw.items() if v})
    res.pop("roofline", None)
    print(json.dumps(res)[:900])


if __name__ == "__main__":
    main()
