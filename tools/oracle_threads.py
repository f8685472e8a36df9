"""How many concurrent sposv callers does SciPy's bundled OpenBLAS serve?  (oracle/lk_oracle.c caps
the port's row-parallel loop at 32 threads; BASELINE.md's protocol names $(nproc).)  Runs one
implicit half-epoch of the oracle on a synthetic matrix in a CHILD process per thread count --
a crash of OpenBLAS' buffer pool ends the child, not the caller -- and prints seconds per count.

    python tools/oracle_threads.py 8 16 32 64 128
"""
import os
import subprocess
import sys

CHILD = r"""
import sys, time, numpy as np, scipy.sparse as sps
sys.path.insert(0, %r)
from oracle import lk_oracle as lko
from lkpy_amd import synth
r = synth.ml25m_like(seed=3, scale=float(sys.argv[1]))
ui = sps.csr_array((np.full(r.nnz, 40.0, np.float32), r.indices, r.indptr), shape=r.shape)
rng = np.random.default_rng(0)
k = 64
Q = lko.als_initial_params(rng, ui.shape[1], k); P = lko.als_initial_params(rng, ui.shape[0], k)
otor = lko.implicit_otor(Q, 0.1)
t = lko.num_threads()
lko.als_half_epoch(ui, P.copy(), Q, otor, t)
t0 = time.perf_counter(); lko.als_half_epoch(ui, P, Q, otor, t); dt = time.perf_counter() - t0
print("threads", t, "seconds", round(dt, 3), "nnz", ui.nnz, flush=True)
"""

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scale = os.environ.get("LK_ORACLE_THREADS_SCALE", "0.2")
    for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
        env = dict(os.environ, OMP_NUM_THREADS=str(n), LKO_MAX_THREADS=str(n))
        p = subprocess.run([sys.executable, "-c", CHILD % root, scale], env=env,
                           capture_output=True, text=True)
        print(n, "rc", p.returncode, p.stdout.strip(), p.stderr.strip()[-300:], flush=True)
