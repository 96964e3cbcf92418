"""multigriddet_amd - MI355X-native (gfx950) hot path of MultiGridDet.

Host-side mirror of the reference's `multigriddet.*` Python API on top of libmgd_hip.so
(hand-written HIP kernels behind the C-ABI in include/mgd_hip.h).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in device memory: the HIP runtime's default on this ROCm (7.2) for gfx950, stated here so that an
# environment that turns it off is a decision and not an accident - a step is ~390 dependent launches and the argument
# fetch sits in front of every one of them (measured on MI355X, same box: 11.74 ms per train step with 1, 12.11 with 0;
# batch-1 inference 1 043 vs 953 images/s).  Only effective if set before the process's first HIP call.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
