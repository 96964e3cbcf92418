"""Per-layer micro-benchmark of the conv engine at the benchmark shapes (608x608, batch 16):
forward gather-GEMM, data-gradient and weight-gradient for every distinct conv of the graph, timed with
HIP events on the launch stream.  Prints one line per (layer shape, pass) with us, TFLOP/s and the
algorithmic HBM GB/s (activations in + out, bf16).  Usage: python tools/bench_conv.py [size] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops  # noqa: E402
from multigriddet_amd.engine import conv_specs, STAGES  # noqa: E402


def layer_shapes(S):
    """(cin, cout, k, s, Hin) for every conv, in graph order."""
    out = []
    specs = conv_specs()
    h = S
    i = 0
    out.append((specs[0], h)); i = 1
    for f, n in STAGES:
        out.append((specs[i], h)); i += 1
        h //= 2
        for _ in range(n):
            out.append((specs[i], h)); out.append((specs[i + 1], h)); i += 2
    g = S // 32
    for sc in range(3):
        for j in range(5):
            out.append((specs[i + j], g))
        i += 5
        if sc < 2:
            out.append((specs[i], g)); i += 1
            g *= 2
    return out


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 608
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device("cuda:0")
    seen = {}
    for sp, h in layer_shapes(S):
        if sp["role"] == "stem":
            continue
        key = (sp["cin"], sp["cout"], sp["k"], sp["s"], h)
        seen[key] = seen.get(key, 0) + 1
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)       # slab workspace of the kernel-row weight gradient
    totf = 0.0
    print(f"{'cin':>5} {'cout':>5} k s {'H':>4} {'n':>2} | {'fwd us':>8} {'TF/s':>6} {'GB/s':>6} | {'dgrad us':>8} {'TF/s':>6} | {'wgrad us':>8} {'TF/s':>6}")
    for (ci, co, k, s, h), n in seen.items():
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
        fl = 2.0 * B * ho * ho * k * k * ci * co
        skip = os.environ.get("BENCH_SKIP", "")       # e.g. "wgrad" or "fwd,dgrad": passes not to time (reported as 1 us)
        t_f = timeit(lambda: ops.conv_fwd(x, pk, out=y, stats=stats)) if "fwd" not in skip else 1.0
        t_d = timeit(lambda: ops.conv_dgrad(dy, pk, (h, h), out=dx)) if "dgrad" not in skip else 1.0
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s, ws=ws, row_blocks=0)) if "wgrad" not in skip else 1.0
        by = 2.0 * (x.numel() + y.numel())
        print(f"{ci:5d} {co:5d} {k} {s} {h:4d} {n:2d} | {t_f:8.1f} {fl / t_f / 1e6:6.0f} {by / t_f / 1e3:6.0f} | "
              f"{t_d:8.1f} {fl / t_d / 1e6:6.0f} | {t_w:8.1f} {fl / t_w / 1e6:6.0f}", flush=True)
        tot["fwd"] += n * t_f; tot["dgrad"] += n * t_d; tot["wgrad"] += n * t_w
        totf += n * fl
    for kname, v in tot.items():
        print(f"total {kname}: {v / 1e3:.2f} ms  ({totf / v / 1e6:.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
