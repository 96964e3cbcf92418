#!/usr/bin/env python3
"""Weight gradient of the 3x3 stride-1 layers with Ci >= 128: per-tap blocks (v2) against the kernel-row patch form
(MGD_WGRAD_ROW, read once per process) - run once per setting; MGD_DBG=16 drops the atomic epilogue of the patch form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
from tools.bench_conv import timeit
dev = torch.device("cuda:0")
for ci, co, h in ((128, 256, 76), (256, 512, 38), (512, 1024, 19)):
    x = torch.randn(16, h, h, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(16, h, h, co, device=dev).to(torch.bfloat16)
    dw = torch.zeros(co, 9, ci, device=dev)
    t = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 1), iters=20)
    print(f"{ci}->{co} @{h}: {t:.1f} us  {2.0 * 16 * h * h * 9 * ci * co / t / 1e6:.0f} TFLOP/s")
