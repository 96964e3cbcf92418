"""A/B of weight-gradient kernel forms (mgd_wgrad_desc.form : form_arg) at the benchmark shape (608 x 608, batch 16), every form
in one process with interleaved rounds; each form's result is also checked against the library's own dispatch (max relative
difference of dW).  usage: python3 tools/bench_wgrad_forms.py [form:arg ...]   (default 0:0 5:0)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops  # noqa: E402

LAYERS = [  # cin, cout, k, Hin   (stride 1)
    (128, 256, 3, 76), (256, 512, 3, 38), (512, 1024, 3, 19),
    (256, 128, 1, 76), (512, 256, 1, 38), (1024, 512, 1, 19),
    (128, 256, 3, 38), (256, 704, 3, 19), (128, 352, 3, 38), (64, 128, 3, 152),
]


def main():
    forms = [tuple(int(v) for v in f.split(":")) for f in sys.argv[1:]] or [(0, 0), (5, 0), (5, 1)]   # 5:1 = kernel-row form with slabs
    B, rounds, iters = 16, 5, 5
    dev = torch.device("cuda:0")
    ops.WGRAD_ROW_BLOCKS = 0                           # each launch alone: one block per CU
    global WS
    WS = torch.empty(64 << 20, dtype=torch.float32, device=dev)      # 256 MB: slabs of the kernel-row form
    print("layer                 " + "  ".join(f"{f[0]:>2}:{f[1]}     us  TF/s  reldiff" for f in forms))
    for (ci, co, k, h) in LAYERS:
        x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, h, h, co, device=dev).to(torch.bfloat16)
        fl = 2.0 * B * h * h * k * k * ci * co
        best, diff, fam = {}, {}, {}
        ref = None
        for r in range(rounds):
            for f in forms:
                ops.WGRAD_FORM, ops.WGRAD_FORM_ARG = f[0], 0
                ws = WS if (f[0] == 5 and f[1] == 1) else None
                dw = torch.zeros(co, k * k, ci, device=dev)
                try:
                    ops.conv_wgrad(x, dy, dw, k, 1, ws=ws)
                except ops.L.MgdError:
                    best[f] = None
                    continue
                fam[f] = ops.L.load().mgd_last_kernel().decode()
                torch.cuda.synchronize()
                if r == 0:
                    if ref is None:
                        ref = dw.clone()
                    diff[f] = ((dw - ref).abs().max() / ref.abs().max()).item()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.conv_wgrad(x, dy, dw, k, 1, ws=ws)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) * 1e3 / iters
                best[f] = t if best.get(f) is None else min(best[f], t)
        ops.WGRAD_FORM = ops.WGRAD_FORM_ARG = 0
        cells = "  ".join(f"{best[f]:8.1f} {fl / best[f] / 1e6:5.0f}  {diff.get(f, 0):7.1e}" if best.get(f) else "       -     -        -" for f in forms)
        print(f"{ci:4d}->{co:4d} k{k} @{h:3d}   {cells}   [{fam.get(forms[-1], '')}]", flush=True)


if __name__ == "__main__":
    main()
