#!/usr/bin/env python3
"""
Condense a rocprofv3 capture made by tools/prof_*.sh (CSV output) into the small text
files that are committed under profiles/:

    python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>

writes <name>_kernel_stats.csv (the `--kernel-trace --stats` table, verbatim columns),
<name>_dispatches.csv (per-dispatch duration / grid / registers of our kernels) and
<name>_counters.csv (mean PMC value per kernel and counter, one pass per group).
"""
import sys
from pathlib import Path

import pandas as pd


def main(src, dst):
    src, dst = Path(src), Path(dst)
    dst.parent.mkdir(parents=True, exist_ok=True)
    stats = list(src.glob("stats/*kernel_stats.csv"))
    if stats:
        df = pd.read_csv(stats[0])
        df.to_csv(f"{dst}_kernel_stats.csv", index=False)
    kt = list(src.glob("stats/*kernel_trace.csv"))
    if kt:
        df = pd.read_csv(kt[0])
        df["Duration_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
        df = df[df.Kernel_Name.str.contains("lk::")]
        cols = ["Kernel_Name", "Duration_us", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count",
                "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size"]  # fmt: skip
        df[cols].to_csv(f"{dst}_dispatches.csv", index=False, float_format="%.3f")
    rows = []
    for d in sorted(src.glob("pmc*")):
        cc = list(d.glob("*counter_collection.csv"))
        if not cc:
            continue
        df = pd.read_csv(cc[0])
        df = df[df.Kernel_Name.str.contains("lk::")]
        g = df.groupby(["Kernel_Name", "Grid_Size", "Counter_Name"]).Counter_Value.agg(
            ["mean", "count"]
        )
        g = g.reset_index()
        g.insert(0, "pass", d.name)
        rows.append(g)
    if rows:
        pd.concat(rows).to_csv(f"{dst}_counters.csv", index=False, float_format="%.6g")


if __name__ == "__main__":
    main(*sys.argv[1:3])
