"""Print calls / average / max duration of the kernels whose names contain one of the given
substrings from a rocprofv3 --kernel-trace --stats output directory:
    python tools/trace_kernels.py <dir> cand_select score_filter ..."""
import sys
from pathlib import Path

import pandas as pd

src = Path(sys.argv[1])
f = next(src.rglob("*kernel_stats.csv"))
d = pd.read_csv(f)
pat = sys.argv[2:]
for _, r in d.iterrows():
    if not pat or any(p in r["Name"] for p in pat):
        print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):4d}  avg {r['AverageNs'] / 1e3:9.1f} us  "
              f"max {r['MaxNs'] / 1e3:9.1f} us")
