"""Stem conv (3 -> 32, 608x608, batch 16): direct fp32 kernel vs bf16 im2col + 1x1 GEMM (forward and weight gradient)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
from tools.bench_conv import timeit
dev = torch.device("cuda:0")
B, S = 16, 608
img = torch.rand(B, S, S, 3, device=dev)
w = torch.randn(32, 9, 3, device=dev) * 0.1
y = torch.empty(B, S, S, 32, dtype=torch.bfloat16, device=dev)
stats = torch.zeros(ops.STATS_REPLICAS, 2, 32, device=dev)
t_direct = timeit(lambda: ops.stem_fwd(img, w.view(32, 27), out=y, stats=stats))
col = torch.empty(B, S, S, 32, dtype=torch.bfloat16, device=dev)
t_col = timeit(lambda: ops.stem_im2col(img, col))
pk = ops.PackedConv(32, 32, 1, 1, dev, need_dgrad=False)
pk.refresh(torch.randn(32, 1, 32, device=dev) * 0.1)
t_gemm = timeit(lambda: ops.conv_fwd(col, pk, out=y, stats=stats))
dy = torch.randn(B, S, S, 32, device=dev).to(torch.bfloat16)
dw = torch.zeros(32, 27, device=dev)
t_wd = timeit(lambda: ops.stem_wgrad(img, dy, dw))
dw2 = torch.zeros(32, 1, 32, device=dev)
t_wg = timeit(lambda: ops.conv_wgrad(col, dy, dw2, 1, 1))
da = torch.randn(B, S, S, 32, device=dev).to(torch.bfloat16)
sc, sh, mu, iv = (torch.rand(32, device=dev) + 0.5 for _ in range(4))
sums = torch.zeros((ops.STATS_REPLICAS + 1) * 2 * 32, device=dev)
dg, db = torch.zeros(32, device=dev), torch.zeros(32, device=dev)
t_ap = timeit(lambda: ops.bn_act_bwd(da, y, sc, sh, mu, iv, sums, dg, db, dy, reduced=True))
t_fu = timeit(lambda: ops.stem_wgrad_bn(img, da, y, sc, sh, mu, iv, sums, dg, db, dw))
print(f"wgrad + BN backward: apply {t_ap:.1f} + wgrad {t_wd:.1f} = {t_ap + t_wd:.1f} us | fused {t_fu:.1f} us")
print(f"forward: direct {t_direct:.1f} us | im2col {t_col:.1f} + gemm {t_gemm:.1f} = {t_col + t_gemm:.1f} us")
print(f"wgrad  : direct {t_wd:.1f} us | gemm on the im2col image {t_wg:.1f} us")
