"""Debug helper: forward conv (3x3, stride 1) against torch fp32 on several shapes; prints where errors sit."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops

dev = torch.device("cuda:0")
shapes = [(2, 20, 20, 64, 128), (1, 8, 16, 64, 128), (2, 19, 19, 128, 128), (1, 38, 38, 128, 256), (2, 76, 76, 128, 256),
          (2, 19, 19, 512, 256), (3, 5, 7, 64, 128), (1, 16, 16, 256, 128), (1, 19, 19, 256, 512), (2, 9, 9, 64, 256)]
for (N, H, W, Ci, Co) in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, Ci, generator=g).to(torch.bfloat16)
    w = torch.randn(Co, 9, Ci, generator=g) / (3 * Ci ** 0.5)
    pk = ops.PackedConv(Co, Ci, 3, 1, dev)
    pk.refresh(w.to(dev))
    y = ops.conv_fwd(x.to(dev), pk)
    torch.cuda.synchronize()
    wr = w.to(torch.bfloat16).float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1)
    err = (y.float().cpu() - ref).abs()
    tol = 0.02 * ref.abs().max().item() + 1e-3
    bad = err > tol
    print(f"N{N} {H}x{W} {Ci}->{Co}: max err {err.max().item():.3f} tol {tol:.3f} bad {int(bad.sum())}/{bad.numel()}")
    if bad.any():
        idx = bad.nonzero()
        print("   bad n:", sorted(set(idx[:, 0].tolist())), " rows:", sorted(set(idx[:, 1].tolist()))[:40])
        print("   cols:", sorted(set(idx[:, 2].tolist()))[:40], " co range:", int(idx[:, 3].min()), int(idx[:, 3].max()),
              " co%16 set:", sorted(set((idx[:, 3] % 16).tolist())), " co//16:", sorted(set((idx[:, 3] // 16).tolist())))
