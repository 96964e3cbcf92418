"""Weight-gradient kernel forms at the benchmark shapes (batch 16) on RANDOM data against an fp32 reference computed tap by tap on
the device (torch einsum over the zero-padded input - checker only), per tap: max |dW - ref| / max |ref|.
usage: python3 tools/check_wgrad_full.py [form:arg ...]   (default 0:0 4:0 5:0 5:1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops  # noqa: E402

LAYERS = [(128, 256, 3, 76), (256, 512, 3, 38), (512, 1024, 3, 19), (64, 128, 3, 152), (256, 128, 1, 76)]


def reference(x, dy, k):
    N, H, W, Ci = x.shape
    xf, dyf = x.float(), dy.float()
    if k == 1:
        return torch.einsum("nhwo,nhwi->oi", dyf, xf)[:, None, :]
    xp = torch.nn.functional.pad(xf, (0, 0, 1, 1, 1, 1))
    taps = []
    for t in range(9):
        dh, dw = t // 3, t % 3
        taps.append(torch.einsum("nhwo,nhwi->oi", dyf, xp[:, dh:dh + H, dw:dw + W, :]))
    return torch.stack(taps, 1)


def main():
    forms = [tuple(int(v) for v in f.split(":")) for f in sys.argv[1:]] or [(0, 0), (4, 0), (5, 0), (5, 1)]
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    ws_all = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    torch.manual_seed(1)
    for (ci, co, k, h) in LAYERS:
        x = torch.randn(16, h, h, ci, device=dev).to(torch.bfloat16)
        dy = torch.randn(16, h, h, co, device=dev).to(torch.bfloat16)
        ref = reference(x, dy, k)
        scale = ref.abs().max().item()
        for f in forms:
            ops.WGRAD_FORM, ops.WGRAD_FORM_ARG = f[0], 0
            ops.WGRAD_ROW_BLOCKS = 0
            ws = ws_all if (f[0] == 5 and f[1] == 1) else None
            dw = torch.zeros(co, k * k, ci, device=dev)
            try:
                ops.conv_wgrad(x, dy, dw, k, 1, ws=ws)
            except ops.L.MgdError as e:
                print(f"{ci}->{co} k{k} @{h} form {f}: refused ({e})")
                continue
            torch.cuda.synchronize()
            per_tap = [((dw[:, t] - ref[:, t]).abs().max().item() / scale) for t in range(k * k)]
            print(f"{ci:4d}->{co:4d} k{k} @{h:3d} form {f[0]}:{f[1]} [{ops.L.load().mgd_last_kernel().decode()}]  max err / max ref per tap: "
                  + " ".join(f"{v:.1e}" for v in per_tap), flush=True)
    ops.WGRAD_FORM = 0


if __name__ == "__main__":
    main()
