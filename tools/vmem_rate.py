"""(DIAGNOSTIC library, libmgd_hip_diag.so.)  How fast can ONE compute unit issue vector-memory instructions?  One workgroup per
CU (256 workgroups), 1 / 2 / 4 / 8 waves each, every wave issues iters x 8 instructions of one kind and nothing else
(mgd_debug_vmem_rate); HIP events around the launch.  Prints cycles per wave-instruction and CU (at the 2.4 GHz nominal clock)
and the byte rate that corresponds to - the figure the gather-GEMM K-step of DESIGN.md (Round 4 item 2) is priced against.
usage: python3 tools/vmem_rate.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import _lib as L  # noqa: E402

KINDS = {0: ("buffer_load_dwordx4 ... lds, L2-resident", 1024), 1: ("buffer_load_dwordx4 ... lds, all lanes out of range", 1024),
         2: ("buffer_load_dwordx4 -> VGPR, L2-resident", 1024), 6: ("buffer_load_dwordx4 -> VGPR, all lanes out of range", 1024),
         5: ("global_load_dwordx4 -> VGPR, L2-resident", 1024), 3: ("buffer_load_dword ... lds, L2-resident", 256),
         4: ("buffer_load_dword -> VGPR, L2-resident", 256)}


def main():
    lib = L.use_diag()
    lib.mgd_debug_vmem_rate.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    window = 2 << 20                                   # 2 MiB: inside every XCD's L2
    buf = torch.zeros(window + 65536, dtype=torch.uint8, device=dev)
    out = torch.zeros(1024, device=dev)
    clk = 2.4e9
    print(f"{'kind':58s} waves/CU   us/launch   cycles per wave-instruction and CU   B/clk/CU   chip TB/s")
    for kind, (name, nbytes) in KINDS.items():
        for waves in (1, 2, 4, 8):
            iters = 4000 // waves
            for _ in range(2):
                L.check(lib.mgd_debug_vmem_rate(buf.data_ptr(), window, out.data_ptr(), 256, waves, iters, kind, L.stream_ptr()), "vmem_rate")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                L.check(lib.mgd_debug_vmem_rate(buf.data_ptr(), window, out.data_ptr(), 256, waves, iters, kind, L.stream_ptr()), "vmem_rate")
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / 3
            n = iters * 8 * waves                      # wave-instructions per CU
            cyc = t * clk / n
            print(f"{name:58s} {waves:8d} {t * 1e6:11.1f} {cyc:38.1f} {nbytes / cyc:10.1f} {nbytes / cyc * 256 * clk / 1e12:11.2f}", flush=True)


if __name__ == "__main__":
    main()
