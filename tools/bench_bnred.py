"""dgrad with / without the fused BN-backward reduction vs the stand-alone reduce pass, per layer shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
from tools.bench_conv import timeit

dev = torch.device("cuda:0")
B = 16
print(f"{'ci':>5} {'co':>5} k {'H':>4} | dgrad us | +bnred us | +addend us | reduce us | apply us")
for ci, co, k, h in [(32, 64, 3, 304), (64, 128, 3, 152), (128, 64, 1, 152), (128, 256, 3, 76), (256, 128, 1, 76),
                     (256, 512, 3, 38), (512, 256, 1, 38), (512, 1024, 3, 19)]:
    dy = torch.randn(B, h, h, co, device=dev).to(torch.bfloat16)
    w = torch.randn(co, k * k, ci, device=dev) * 0.05
    pk = ops.PackedConv(co, ci, k, 1, dev); pk.refresh(w)
    dx = torch.empty(B, h, h, ci, dtype=torch.bfloat16, device=dev)
    add = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
    y = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
    sc = torch.rand(ci, device=dev) + 0.5; sh = torch.randn(ci, device=dev)
    mu = torch.randn(ci, device=dev); iv = torch.rand(ci, device=dev) + 0.5
    sums = torch.zeros((ops.STATS_REPLICAS + 1) * 2 * ci, device=dev)
    dg = torch.zeros(ci, device=dev); db = torch.zeros(ci, device=dev)
    dyo = torch.empty_like(y)
    t0 = timeit(lambda: ops.conv_dgrad(dy, pk, (h, h), out=dx))
    t1 = timeit(lambda: ops.conv_dgrad(dy, pk, (h, h), out=dx, bnred=(y, sc, sh, mu, iv, sums)))
    t2 = timeit(lambda: ops.conv_dgrad(dy, pk, (h, h), out=dx, addend=add))
    lib = ops.L.load()
    import ctypes as C
    def red():
        ops.L.check(lib.mgd_bn_act_bwd_reduce(ops.L.ptr(dx), ops.L.ptr(y), ops.L.ptr(sc), ops.L.ptr(sh), ops.L.ptr(mu),
                    ops.L.ptr(iv), ops.L.ptr(sums), ops.STATS_REPLICAS, C.c_int64(y.numel() // ci), ci,
                    C.c_float(0.1), ops.L.stream_ptr()), "red")
    t3 = timeit(red)
    t4 = timeit(lambda: ops.bn_act_bwd(dx, y, sc, sh, mu, iv, sums, dg, db, dyo, reduced=True))
    print(f"{ci:5d} {co:5d} {k} {h:4d} | {t0:8.1f} | {t1:9.1f} | {t2:10.1f} | {t3:9.1f} | {t4:8.1f}")
