#!/usr/bin/env python3
"""Diagnostic: where a wave of the resident-patch gather-GEMM spends its cycles (stamped build, MGD_DBG=2).
usage: MGD_DBG=2 [MGD_PGEMM=1] python tools/stamps_pg.py cin cout H [B]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import ops, _lib
ci, co, h = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda:0")
x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, 9, ci, device=dev) * 0.05
pk = ops.PackedConv(co, ci, 3, 1, dev); pk.refresh(w)
y = torch.empty(B, h, h, co, dtype=torch.bfloat16, device=dev)
lib = _lib.load()
buf = (C.c_ulonglong * 8)()
ops.conv_fwd(x, pk, out=y); torch.cuda.synchronize(); lib.mgd_debug_read_stamps(buf)
n = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.conv_fwd(x, pk, out=y)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
lib.mgd_debug_read_stamps(buf)
pro, tw, tb, rest, epi, nt = (buf[i] for i in range(6))
nk = 9 * ci // 64
tot = pro + tw + tb + rest + epi
print(f"{ci}->{co} @{h} B{B}: {us:.1f} us/launch (stamped); wave-tiles {nt}, K-steps/tile {nk}; share of wave cycles: "
      f"prologue {100*pro/tot:.1f}%  wait-vmcnt {100*tw/tot:.1f}%  barrier {100*tb/tot:.1f}%  rest(issue+reads+mfma) {100*rest/tot:.1f}%  epilogue {100*epi/tot:.1f}%")
print(f"   cycles per wave-tile-Kstep: wait {tw/nt/nk:.0f} barrier {tb/nt/nk:.0f} rest {rest/nt/nk:.0f}; per wave-tile: prologue {pro/nt:.0f} epilogue {epi/nt:.0f}")
