"""Parity check of ONE gather-GEMM kernel variant, selected through the library's environment switches (they are read once
per process, hence a process per variant: tests/test_gpu_gemm_variants.py).  For every shape: forward with the BatchNorm
statistics epilogue, forward with bias + LeakyReLU + residual (folded inference), data gradient with residual addend, all
against fp32 torch on the same bf16-rounded operands; prints one line per check and exits non-zero on the first mismatch.
usage: gemm_variant_check.py <expected kernel family substring>"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops  # noqa: E402

expect = sys.argv[1]
dev = torch.device("cuda:0")
SHAPES = [  # N, H, W, Ci, Co, k   (stride 1; enough tiles / K-steps for the persistent, ping-pong and streaming forms)
    (4, 76, 76, 128, 256, 3),      # 18 K-steps, 181 tiles of 256 x 128, ragged last tile
    (16, 19, 19, 512, 1024, 3),    # 72 K-steps, 4 channel tiles
    (8, 38, 38, 256, 512, 3),      # 36 K-steps
    (3, 30, 52, 128, 256, 3),      # non-square map, pixel count not a multiple of 64
]


def bf(t):
    return t.to(torch.bfloat16).float()


def ref_conv(x, w, k):
    co, T, ci = w.shape
    wr = bf(w).view(co, k, k, ci).permute(0, 3, 1, 2)
    return F.conv2d(x.permute(0, 3, 1, 2), wr, padding=k // 2).permute(0, 2, 3, 1)


seen = set()
for (N, H, W, Ci, Co, k) in SHAPES:
    g = torch.Generator().manual_seed(31 + H + Ci)
    x = bf(torch.randn(N, H, W, Ci, generator=g))
    w = torch.randn(Co, k * k, Ci, generator=g) / (k * Ci ** 0.5)
    bias = torch.randn(Co, generator=g) * 0.5
    pk = ops.PackedConv(Co, Ci, k, 1, dev)
    pk.refresh(w.to(dev))
    xd = x.to(dev).to(torch.bfloat16)
    y_ref = ref_conv(x, w, k)
    tol = 0.02 * y_ref.abs().max().item() + 1e-3
    # forward + BatchNorm statistics
    stats = torch.zeros(ops.STATS_REPLICAS, 2, Co, device=dev)
    y = ops.conv_fwd(xd, pk, stats=stats)
    fam = ops.L.load().mgd_last_kernel().decode()
    seen.add(fam)
    torch.cuda.synchronize()
    err = (y.float().cpu() - y_ref).abs().max().item()
    yb = y.float().cpu().view(-1, Co)
    st = stats.sum(0).cpu()
    ok = err <= tol and np.allclose(st[0].numpy(), yb.sum(0).numpy(), rtol=2e-3, atol=3e-2) and \
        np.allclose(st[1].numpy(), (yb * yb).sum(0).numpy(), rtol=2e-3, atol=3e-2)
    print(f"{fam}: fwd+stats {N}x{H}x{W} {Ci}->{Co}: err {err:.4f} tol {tol:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
    # folded-inference epilogue
    res = bf(torch.randn(N, H, W, Co, generator=g))
    z = y_ref + bias
    ref2 = torch.where(z > 0, z, 0.1 * z) + res
    out = ops.conv_fwd(xd, pk, bias=bias.to(dev), act_slope=0.1, addend=res.to(dev).to(torch.bfloat16))
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref2).abs().max().item()
    ok = err <= 0.02 * ref2.abs().max().item() + 1e-3
    print(f"{fam}: fwd bias+leaky+residual: err {err:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
    # data gradient + residual addend (the transposed conv: Co -> Ci)
    dy = bf(torch.randn(N, H, W, Co, generator=g))
    add = bf(torch.randn(N, H, W, Ci, generator=g))
    wr = bf(w).view(Co, k, k, Ci).permute(0, 3, 1, 2)
    dx_ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), wr, padding=k // 2).permute(0, 2, 3, 1) + add
    dx = ops.conv_dgrad(dy.to(dev).to(torch.bfloat16), pk, (H, W), addend=add.to(dev).to(torch.bfloat16))
    seen.add(ops.L.load().mgd_last_kernel().decode())
    torch.cuda.synchronize()
    err = (dx.float().cpu() - dx_ref).abs().max().item()
    ok = err <= 0.02 * dx_ref.abs().max().item() + 1e-3
    print(f"dgrad+addend {Co}->{Ci}: err {err:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
if not any(expect in f for f in seen):
    print(f"variant not exercised: expected a kernel family containing {expect!r}, saw {sorted(seen)}")
    sys.exit(2)
print("all ok; kernel families:", sorted(seen))
