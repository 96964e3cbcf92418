#!/usr/bin/env python3
"""Timing of the per-tap weight-gradient kernel on one layer; MGD_DBG=16 swaps its fp32 atomics for plain stores, 32 drops
the epilogue.  usage: [MGD_DBG=16|32] python tools/ablate_wgrad.py cin cout H [k]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
ci, co, h = (int(v) for v in sys.argv[1:4])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
x = torch.randn(16, h, h, ci, device=dev).to(torch.bfloat16)
dy = torch.randn(16, h, h, co, device=dev).to(torch.bfloat16)
dw = torch.zeros(co, k * k, ci, device=dev)
for _ in range(3): ops.conv_wgrad(x, dy, dw, k, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv_wgrad(x, dy, dw, k, 1)
e1.record(); torch.cuda.synchronize()
print(f"DBG={os.environ.get('MGD_DBG','0'):>4} wgrad {ci}->{co}@{h} k{k}: {e0.elapsed_time(e1)*1e3/20:.1f} us")
