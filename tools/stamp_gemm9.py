"""Phase times of the ping-pong gather-GEMM (stamped instantiation of the DIAGNOSTIC library, flag 4096):
   python3 tools/stamp_gemm9.py cin cout k H [B]
Prints, for waves 0-3 (group A) and 4-7 (group B), the mean s_memtime ticks per K-step spent in each segment."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import _lib
_lib.use_diag().mgd_diag_set_flags(4096)
from multigriddet_amd import ops
ops.CONV_FORM, ops.CONV_FORM_ARG = 10, 8      # ping-pong form, 128-pixel tiles

ci, co, k, h = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda:0")
x = torch.randn(B, h, h, ci, device=dev).to(torch.bfloat16)
w = torch.randn(co, k * k, ci, device=dev) * 0.05
pk = ops.PackedConv(co, ci, k, 1, dev)
pk.refresh(w)
y = torch.empty(B, h, h, co, dtype=torch.bfloat16, device=dev)
stats = torch.zeros(ops.STATS_REPLICAS, 2, co, device=dev)
lib = _lib.load()
buf = (C.c_ulonglong * 24)()
for _ in range(3):
    ops.conv_fwd(x, pk, out=y, stats=stats)
torch.cuda.synchronize()
lib.mgd_debug_stamps(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv_fwd(x, pk, out=y, stats=stats)
e1.record()
torch.cuda.synchronize()
lib.mgd_debug_stamps(buf)
v = list(buf)
print(f"{ci}->{co} k{k} @{h} B{B}: {e0.elapsed_time(e1) * 100:.1f} us per launch (stamped build)")
names = {0: ["wait stage s+1", "read kk0 issue", "issue stage s+2", "barrier 1", "multiply(+reads kk1)", "barrier 2"],
         1: ["wait stage s+1", "read kk0 issue", "issue stage s+2", "barrier 1", "multiply(+reads kk1)", "barrier 2"]}
for g in range(2):
    n = max(1, v[g * 8 + 7])
    tot = sum(v[g * 8 + q] for q in range(6))
    print(f"group {'AB'[g]}: {n} wave-steps, {tot / n:.0f} ticks per step")
    for q in range(6):
        print(f"    {names[g][q]:24s} {v[g * 8 + q] / n:8.0f}")
n = max(1, v[23])
tn = ["row tables (+barriers)", "prologue issue", "first wait", "K-loop", "drain + barrier", "epilogue"]
print(f"per tile ({n} tiles, wave 0): " + ", ".join(f"{tn[q]} {v[16 + q] / n:.0f}" for q in range(6)))
