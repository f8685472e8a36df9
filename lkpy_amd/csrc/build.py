#!/usr/bin/env python3
"""Compile the HIP sources of this directory into ``lkpy_amd/_lkamd.so`` for gfx950.

hipcc cross-compiles without a GPU; the shared object stays in-tree (git-ignored)
so that it travels to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE.parent / "_lkamd.so"
OBJ = HERE / "_obj"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [*os.environ.get("LK_EXTRA_FLAGS", "").split(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    srcs = sorted(HERE.glob("*.hip"))
    hdrs = sorted(HERE.glob("*.h")) + [HERE.parent.parent / "include" / "lkamd.h"]
    OBJ.mkdir(exist_ok=True)
    jobs = []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        if force or _stale(o, [s, *hdrs]):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC, *FLAGS, "-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
