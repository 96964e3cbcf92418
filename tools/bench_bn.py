"""Micro-benchmark of the BN/activation elementwise kernels at the benchmark shapes (608x608, batch 16):
us and algorithmic HBM GB/s per (P, C).  Usage: python tools/bench_bn.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops
from tools.bench_conv import layer_shapes, timeit

dev = torch.device("cuda:0")
B = 16
seen = {}
for sp, h in layer_shapes(608):
    if not sp["bn"]:
        continue
    ho = h // sp["s"]
    key = (B * ho * ho, sp["cout"])
    seen[key] = seen.get(key, 0) + 1
tot = [0.0, 0.0, 0.0]
print(f"{'P':>9} {'C':>5} {'n':>2} | {'fwd us':>8} {'GB/s':>6} | {'reduce us':>9} {'GB/s':>6} | {'apply us':>9} {'GB/s':>6}")
for (P, C), n in seen.items():
    y = torch.randn(P, C, device=dev).to(torch.bfloat16)
    da = torch.randn(P, C, device=dev).to(torch.bfloat16)
    res = torch.randn(P, C, device=dev).to(torch.bfloat16)
    out = torch.empty_like(y)
    dy = torch.empty_like(y)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    scale, shift, smean, sinv = (torch.empty(C, device=dev) for _ in range(4))
    stats = torch.zeros(ops.STATS_REPLICAS, 2, C, device=dev)
    stats[0, 1] = float(P)
    ops.bn_finalize(stats, float(P), gamma, beta, mm, mv, scale, shift, smean, sinv)
    sums = torch.zeros((ops.STATS_REPLICAS + 1) * 2 * C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    L = ops.L
    import ctypes as Ct
    lib = L.load()
    t_f = timeit(lambda: ops.bn_act_fwd(y, scale, shift, out, residual=res))
    t_r = timeit(lambda: L.check(lib.mgd_bn_act_bwd_reduce(L.ptr(da), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv), L.ptr(sums), ops.STATS_REPLICAS, Ct.c_int64(P), C, Ct.c_float(0.1), L.stream_ptr())))
    t_a = timeit(lambda: L.check(lib.mgd_bn_act_bwd_apply(L.ptr(da), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv), L.ptr(sums), ops.STATS_REPLICAS, L.ptr(dg), L.ptr(db), L.ptr(dy), Ct.c_int64(P), C, Ct.c_float(0.1), 0, L.stream_ptr())))
    by = 2.0 * P * C
    t_c = timeit(lambda: out.copy_(y))                       # the device's own copy at this size: the achievable rate
    t_ff = timeit(lambda: L.check(lib.mgd_bn_act_fwd_fused(L.ptr(stats), ops.STATS_REPLICAS, Ct.c_float(P), L.ptr(gamma), L.ptr(beta), L.ptr(mm), L.ptr(mv),
                                                             L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv), Ct.c_float(1e-3), Ct.c_float(0.99), 1,
                                                             L.ptr(y), None, L.ptr(out), Ct.c_int64(P), C, Ct.c_float(0.1), L.stream_ptr())))
    print(f"{P:9d} {C:5d} {n:2d} | {t_f:8.1f} {3 * by / t_f / 1e3:6.0f} | {t_r:9.1f} {2 * by / t_r / 1e3:6.0f} | {t_a:9.1f} {3 * by / t_a / 1e3:6.0f} | "
          f"fused fwd (no residual) {t_ff:7.1f} us {2 * by / t_ff / 1e3:6.0f} GB/s | copy {t_c:7.1f} us {2 * by / t_c / 1e3:6.0f} GB/s", flush=True)
    tot[0] += n * t_f; tot[1] += n * t_r; tot[2] += n * t_a
print(f"totals per step: fwd {tot[0] / 1e3:.2f} ms (with residual on all), reduce {tot[1] / 1e3:.2f} ms, apply {tot[2] / 1e3:.2f} ms")
