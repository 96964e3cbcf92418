"""Timing of the per-tap weight-gradient kernel on one layer on the DIAGNOSTIC library (libmgd_hip_diag.so): flag 16 swaps its
fp32 atomics for plain stores, 32 drops the epilogue, 131072 puts every LDS-DMA lane out of range, ... (conv_wgrad.hip).
usage: python3 tools/ablate_wgrad.py <flags> cin cout H [k]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import _lib
lib = _lib.use_diag()
from multigriddet_amd import ops
flags, ci, co, h = (int(v) for v in sys.argv[1:5])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
lib.mgd_diag_set_flags(flags)
ops.WGRAD_FORM = int(os.environ.get("WG_FORM", "0"))
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
print(f"flags={flags:>4} wgrad {ci}->{co}@{h} k{k}: {e0.elapsed_time(e1)*1e3/20:.1f} us")
