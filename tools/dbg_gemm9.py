#!/usr/bin/env python3
"""Debug: forward conv through the gather-GEMM against torch on a few shapes; prints max error / NaN count and where.
usage: dbg_gemm9.py N H W Ci Co k"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops

N, H, W, Ci, Co, k = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(N, H, W, Ci, generator=g).to(torch.bfloat16)
w = torch.randn(Co, k * k, Ci, generator=g) / (k * Ci ** 0.5)
pk = ops.PackedConv(Co, Ci, k, 1, dev)
pk.refresh(w.to(dev))
# poison LDS-sized garbage? not possible from here; run twice
for rep in range(2):
    y = ops.conv_fwd(x.to(dev), pk)
    torch.cuda.synchronize()
wr = w.to(torch.bfloat16).float().view(Co, k, k, Ci).permute(0, 3, 1, 2)
ref = F.conv2d(x.float().permute(0, 3, 1, 2), wr, padding=k // 2).permute(0, 2, 3, 1)
yc = y.float().cpu()
bad = ~torch.isfinite(yc)
err = (yc - ref).abs()
err[bad] = 0
tol = 0.02 * ref.abs().max().item() + 1e-3
wrong = (err > tol) | bad
print(f"shape {N}x{H}x{W} {Ci}->{Co} k{k}: nan/inf {int(bad.sum())}, wrong {int(wrong.sum())} of {yc.numel()}, max err {err.max().item():.4f} tol {tol:.4f}")
if wrong.any():
    idx = wrong.nonzero()
    pix = (idx[:, 0] * H + idx[:, 1]) * W + idx[:, 2]
    up = torch.unique(pix)
    print("  wrong pixels (flat):", up[:20].tolist(), "... count", len(up), " channels:", torch.unique(idx[:, 3])[:16].tolist())
    print("  first (n,h,w,c):", idx[:8].tolist())
