#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the gather-GEMM K-loop (stamped build, MGD_DBG=2).
usage: MGD_DBG=2 python tools/stamps.py cin cout k s H [B]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops, _lib
ci, co, k, s, h = (int(v) for v in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 16
dev = torch.device("cuda:0")
x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, k * k, ci, device=dev) * 0.05
pk = ops.PackedConv(co, ci, k, s, dev); pk.refresh(w)
y = torch.empty(B, h // s, h // s, co, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
buf = (C.c_ulonglong * 8)()
ops.conv_fwd(x, pk, out=y); torch.cuda.synchronize(); lib.mgd_debug_read_stamps(buf)
n = 5
for _ in range(n):
    ops.conv_fwd(x, pk, out=y)
torch.cuda.synchronize()
lib.mgd_debug_read_stamps(buf)
tw, tb, ti, tc, nks = (buf[i] for i in range(5))
tot = tw + tb + ti + tc
print(f"wave K-steps {nks}: cycles per wave-K-step: wait_vmcnt {tw/nks:.0f}  zero+barrier {tb/nks:.0f}  dma-issue {ti/nks:.0f}  reads+mfma {tc/nks:.0f}  total {tot/nks:.0f}")
print("shares: wait %.1f%% barrier %.1f%% issue %.1f%% compute %.1f%%" % (100*tw/tot, 100*tb/tot, 100*ti/tot, 100*tc/tot))
