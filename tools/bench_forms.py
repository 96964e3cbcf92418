"""A/B of gather-GEMM kernel forms on the large 3x3 layers at the benchmark shape (608 x 608, batch 16), forward and data
gradient, every form in the same process with interleaved rounds (cdna_hip_programming.md rule 24).  Forms are (form,
form_arg) pairs of mgd_conv_desc; 0:0 is the library's own dispatch.
usage: python3 tools/bench_forms.py [form:arg ...]     (default: 0:0 12:0 12:1 12:2)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops  # noqa: E402

LAYERS = [  # cin, cout, k, s, Hin
    (128, 256, 3, 1, 76), (256, 512, 3, 1, 38), (512, 1024, 3, 1, 19),
    (128, 256, 3, 2, 152), (256, 512, 3, 2, 76), (512, 1024, 3, 2, 38),
    (64, 128, 3, 1, 152),
    (256, 128, 1, 1, 76), (512, 256, 1, 1, 38), (1024, 512, 1, 1, 19),
]


def main():
    forms = [tuple(int(v) for v in f.split(":")) for f in sys.argv[1:]] or [(0, 0), (12, 0), (12, 1), (12, 2)]
    B, rounds, iters = 16, 5, 5
    dev = torch.device("cuda:0")
    ops.CONV_FORM_SOFT = False
    print("layer                      pass  " + "  ".join(f"{f[0]:>2}:{f[1]}   us  TF/s" for f in forms))
    for (ci, co, k, s, h) in LAYERS:
        ho = h // s
        x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, ho, ho, co, device=dev).to(torch.bfloat16)
        w = torch.randn(co, k * k, ci, device=dev) * 0.05
        pk = ops.PackedConv(co, ci, k, s, dev)
        pk.refresh(w)
        y = torch.empty(B, ho, ho, co, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(B, h, h, ci, dtype=torch.bfloat16, device=dev)
        stats = torch.zeros(ops.STATS_REPLICAS, 2, co, device=dev)
        fl = 2.0 * B * ho * ho * k * k * ci * co
        for pname, fn in (("fwd", lambda: ops.conv_fwd(x, pk, out=y, stats=stats)), ("dgrad", lambda: ops.conv_dgrad(dy, pk, (h, h), out=dx))):
            best = {}
            fam = {}
            for r in range(rounds):
                for f in forms:
                    ops.CONV_FORM, ops.CONV_FORM_ARG = f
                    try:
                        fn()
                    except ops.L.MgdError:
                        best[f] = None
                        continue
                    fam[f] = ops.L.load().mgd_last_kernel().decode()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) * 1e3 / iters
                    best[f] = t if best.get(f) is None else min(best[f], t)
            ops.CONV_FORM = ops.CONV_FORM_ARG = 0
            cells = "  ".join(f"{best[f]:8.1f} {fl / best[f] / 1e6:5.0f}" if best.get(f) else "       -     -" for f in forms)
            print(f"{ci:4d}->{co:4d} k{k} s{s} @{h:3d}  {pname:5s} {cells}   [{fam.get(forms[0], '')}]", flush=True)


if __name__ == "__main__":
    main()
