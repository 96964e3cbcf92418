"""Split-K sweep of the weight-gradient kernel for one layer shape: python tools/bench_wgrad_splits.py cin cout k s H"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
from tools.bench_conv import timeit
ci, co, k, s, h = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
B = 16
x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
dy = torch.randn(B, h // s, h // s, co, device=dev).to(torch.bfloat16)
dw = torch.zeros(co, k * k, ci, device=dev)
fl = 2.0 * B * (h // s) ** 2 * k * k * ci * co
auto = ops.wgrad_splits(B * (h // s) ** 2, co, ci, k * k)
for sp in sorted({1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, auto}):
    t = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s, splits=sp))
    print(f"splits {sp:3d}{' (auto)' if sp == auto else ''}: {t:8.1f} us  {fl / t / 1e6:6.0f} TF/s")
