"""Ablation timing of the 128-channel gather-GEMM (conv_gemm8_kernel) on one layer, on the DIAGNOSTIC library
(libmgd_hip_diag.so, mgd_diag_set_flags).  Flag bits select its ablation instantiation: 32 no epilogue, 64 no MFMA, 128 no
pixel-fragment reads, 256 no weight-fragment loads, 512 no LDS-DMA, 1024 dispatch only, 2048 no K-loop.
usage: python3 tools/ablate_gemm.py <flags> cin cout H [k]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import _lib
lib = _lib.use_diag()
from multigriddet_amd import ops
flags, ci, co, h = (int(v) for v in sys.argv[1:5])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
lib.mgd_diag_set_flags(flags)
ops.CONV_FORM = 8      # conv_gemm8_kernel
dev = torch.device("cuda:0")
x = torch.randn(16, h, h, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, k * k, ci, device=dev) * 0.05
pk = ops.PackedConv(co, ci, k, 1, dev); pk.refresh(w)
y = torch.empty(16, h, h, co, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.conv_fwd(x, pk, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv_fwd(x, pk, out=y)
e1.record(); torch.cuda.synchronize()
print(f"flags={flags:>4} {ci}->{co}@{h} k{k}: {e0.elapsed_time(e1)*1e3/20:.1f} us")
