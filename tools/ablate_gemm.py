#!/usr/bin/env python3
"""Ablation timing of the 128-channel gather-GEMM (conv_gemm8_kernel) on one layer.  MGD_DBG bits select its ablation build:
32 no epilogue, 64 no MFMA, 128 no pixel-fragment reads, 256 no weight-fragment loads, 512 no LDS-DMA.
usage: MGD_DBG=<bits> python tools/ablate_gemm.py cin cout H [k]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
ci, co, h = (int(v) for v in sys.argv[1:4])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
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
print(f"DBG={os.environ.get('MGD_DBG','0'):>4} {ci}->{co}@{h} k{k}: {e0.elapsed_time(e1)*1e3/20:.1f} us")
