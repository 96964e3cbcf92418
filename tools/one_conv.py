"""Run ONE conv shape repeatedly (for rocprofv3 --pmc passes).  usage: one_conv.py cin cout k s H [B] [mode] [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops

ci, co, k, s, h = (int(v) for v in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 16
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev = torch.device("cuda:0")
ho = h // s
x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
dy = torch.randn(B, ho, ho, co, device=dev).to(torch.bfloat16)
w = torch.randn(co, k * k, ci, device=dev) * 0.05
pk = ops.PackedConv(co, ci, k, s, dev)
pk.refresh(w)
y = torch.empty(B, ho, ho, co, dtype=torch.bfloat16, device=dev)
dx = torch.empty(B, h, h, ci, dtype=torch.bfloat16, device=dev)
dw = torch.zeros(co, k * k, ci, device=dev)
stats = torch.zeros(ops.STATS_REPLICAS, 2, co, device=dev)
ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)          # slab workspace of the kernel-row weight gradient
for _ in range(iters):
    if mode == "fwd":
        ops.conv_fwd(x, pk, out=y, stats=stats)
    elif mode == "dgrad":
        ops.conv_dgrad(dy, pk, (h, h), out=dx)
    else:
        ops.conv_wgrad(x, dy, dw, k, s, ws=ws, row_blocks=0)
torch.cuda.synchronize()
print("done", mode, ci, co, k, s, h, B)
