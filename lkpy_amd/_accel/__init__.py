"""
Function seam: drop-in stand-ins for the functions of the reference's native module
``lenskit._accel`` (Rust / PyO3) that sit on the hot path, with the same argument lists
(SURVEY.md section 8b).  Host buffers in, host buffers out; the arithmetic runs in the HIP
kernels behind ``lkpy_amd/_lkamd.so`` -- there is no CPU fallback.
"""
from . import als, data, knn  # noqa: F401
