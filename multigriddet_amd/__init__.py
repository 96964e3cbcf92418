"""multigriddet_amd - MI355X-native (gfx950) hot path of MultiGridDet.

Host-side mirror of the reference's `multigriddet.*` Python API on top of libmgd_hip.so
(hand-written HIP kernels behind the C-ABI in include/mgd_hip.h).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
