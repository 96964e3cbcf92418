"""Per-launch times of the gather-GEMM launches of ONE folded inference forward (HIP events around every launch; they add
a few microseconds of gap each, so the sum is above the un-instrumented forward).  usage: infer_layers.py [batch] [size]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
from multigriddet_amd.models import build_multigriddet_darknet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 608
dev = torch.device("cuda:0")
model, _ = build_multigriddet_darknet(input_shape=(S, S, 3), num_classes=80)
model.fold_bn(True)
img = torch.from_numpy(np.random.default_rng(0).random((B, S, S, 3), dtype=np.float32)).to(dev)
for _ in range(3):
    model(img, training=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    model(img, training=False)
e1.record()
torch.cuda.synchronize()
print(f"forward, batch {B}: {e0.elapsed_time(e1) / 10:.3f} ms")
ops.PROFILE = []
model(img, training=False)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
tot = 0.0
for (a, b, fl, variant, what) in prof:
    us = a.elapsed_time(b) * 1e3
    tot += us
    print(f"{us:8.1f} us  {fl / 1e9:7.2f} GFLOP  {fl / us / 1e6:7.1f} TF/s  {variant}")
print(f"sum of {len(prof)} conv launches: {tot / 1e3:.3f} ms")
