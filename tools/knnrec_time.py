#!/usr/bin/env python3
"""The cfg3 recommend leg alone (10 000 users, every item scored, top-100; csrc/iknn_recommend.hip):
time + (with --parity) the bench's oracle check of 2 048 users.  LK_AMD_LIBRARY selects a variant
build (tools/build_variant.py ... -DLK_REC_RW=2048)."""
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
    res = _knn_bench._recommend_leg(D, ratings, means, sims, dev,
                                    checker=bench.knn_recommend_cpu_and_parity if args.parity else None)
    res.pop("roofline", None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
