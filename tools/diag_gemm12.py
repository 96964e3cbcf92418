"""Diagnostics of the phased gather-GEMM (conv_gemm12_kernel) on one layer: time with the weight DMA / pixel DMA / epilogue
switched off and the per-block stamps {prologue, K-loop, epilogue} - the diagnostic instantiation of libmgd_hip_diag.so, flags
through mgd_diag_set_flags (G12_* of conv_gemm12.hip).
usage: python3 tools/diag_gemm12.py cin cout H [shape] [k]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import _lib as L  # noqa: E402
LIB = L.use_diag()
from multigriddet_amd import ops  # noqa: E402
G12_STAMP, G12_NO_W, G12_NO_X, G12_NO_EPI = 0x08000000, 0x10000000, 0x20000000, 0x40000000
import ctypes as C  # noqa: E402


def main():
    ci, co, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    shape = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    B = 16
    dev = torch.device("cuda:0")
    x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, k * k, ci, device=dev) * 0.05
    pk = ops.PackedConv(co, ci, k, 1, dev)
    pk.refresh(w)
    y = torch.empty(B, h, h, co, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(ops.STATS_REPLICAS, 2, co, device=dev)
    stamps = torch.zeros(4096 * 4, dtype=torch.int64, device=dev)
    dh, dw = ops.taps_fwd(k)
    fl = 2.0 * B * h * h * k * k * ci * co

    def launch(flags):
        d = ops._desc(x, pk.fwd, y, B, h, h, ci, h, h, h, h, co, 1, 1, (0, 0), dh, dw, pk.fwd_kpad, pk.fwd_copad, stats=stats)
        LIB.mgd_diag_set_flags(flags)
        d.form, d.form_arg = 12, shape
        d.partial = stamps.data_ptr()
        L.check(L.load().mgd_conv_gather_gemm(C.byref(d), L.stream_ptr()), "diag")

    names = {0: "full", G12_NO_W: "no weight DMA", G12_NO_X: "no pixel DMA", G12_NO_W | G12_NO_X: "no DMA", G12_NO_EPI: "no epilogue",
             G12_NO_W | G12_NO_X | G12_NO_EPI: "no DMA, no epilogue"}
    res = {f: 1e9 for f in names}
    for r in range(4):
        for f in names:
            launch(f)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                launch(f)
            e1.record()
            torch.cuda.synchronize()
            res[f] = min(res[f], e0.elapsed_time(e1) * 200)
    for f, n in names.items():
        print(f"{ci}->{co}@{h} shape {shape}: {n:22s} {res[f]:7.1f} us  {fl / res[f] / 1e6:6.0f} TF/s")
    launch(G12_STAMP)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4)
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    nk = pk.fwd_kpad // 64
    print(f"stamps over {len(s)} blocks (shader cycles, 100 MHz clock? see s_memtime): prologue {np.median(pro):.0f}, K-loop {np.median(loop):.0f} "
          f"({np.median(loop) / nk:.0f} per K-step of {nk}), epilogue {np.median(epi):.0f}; start spread {np.ptp(s[:, 0]):.0f}, "
          f"end-to-end {s[:, 3].max() - t0:.0f}")


if __name__ == "__main__":
    main()
