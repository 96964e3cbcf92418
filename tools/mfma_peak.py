"""(DIAGNOSTIC library, libmgd_hip_diag.so.)  The matrix pipe's ceiling on this box: a bare v_mfma_f32_16x16x32_bf16 stream (mgd_debug_mfma_peak: register operands,
8 or 16 independent accumulators per wave, no memory traffic) at 1 - 4 workgroups of 4 waves per CU, timed with HIP events.
Prints TFLOP/s = launches' MFMA FLOPs / time; 2 500 would be the dense bf16 figure at 2.4 GHz."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd import _lib as L

lib = L.use_diag()
dev = torch.device("cuda:0")
out = torch.zeros(1024, device=dev)
for nacc in (8, 16):
    for per_cu in (1, 2, 3, 4, 8):
        blocks, iters = 256 * per_cu, 20000 // per_cu
        fl = blocks * 4 * iters * nacc * 2.0 * 16 * 16 * 32
        for _ in range(2):
            L.check(lib.mgd_debug_mfma_peak(L.ptr(out), blocks, iters, nacc, L.stream_ptr()), "mfma_peak")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(lib.mgd_debug_mfma_peak(L.ptr(out), blocks, iters, nacc, L.stream_ptr()), "mfma_peak")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{nacc:2d} accumulators, {per_cu} workgroups per CU ({per_cu} waves per SIMD): {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s", flush=True)

print("weight-gradient K-step skeleton, three workgroups per CU (16 MFMAs per wave and step):")
names = {0: "bare MFMAs", 1: "+ 24 fragment reads (bursts, lgkmcnt(0))", 9: "+ 24 fragment reads (interleaved, counted)", 2: "+ barrier", 4: "+ 6 LDS-DMA (out of range)",
         3: "reads (bursts) + barrier", 11: "reads (interleaved) + barrier", 7: "reads (bursts) + barrier + LDS-DMA", 15: "reads (interleaved) + barrier + LDS-DMA",
         23: "reads (bursts) + barrier + LDS-DMA + setprio", 31: "reads (interleaved) + barrier + LDS-DMA + setprio",
         32: "+ 6 register loads (out of range) + 6 ds_write_b128", 35: "reads (bursts) + barrier + register loads + ds_write",
         43: "reads (interleaved) + barrier + register loads + ds_write",
         2048: "+ 6 LDS-DMA behind one M0 set-up per group (immediate offsets)", 2051: "reads (bursts) + barrier + LDS-DMA (immediate offsets)"}
for mode in (0, 1, 9, 2, 4, 2048, 32, 3, 11, 7, 2051, 15, 23, 31, 35, 43):
    blocks, iters = 768, 4000
    fl = blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 32
    for _ in range(2):
        L.check(lib.mgd_debug_wgrad_skeleton(L.ptr(out), blocks, iters, mode, L.stream_ptr()), "skeleton")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        L.check(lib.mgd_debug_wgrad_skeleton(L.ptr(out), blocks, iters, mode, L.stream_ptr()), "skeleton")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"mode {mode:4d} {names[mode]:52s}: {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s", flush=True)

print("candidate tiles (mode bits 7 = reads in bursts + barrier + LDS-DMA, 15 = reads interleaved):")
shapes = {1: ("256 x 128, 8 waves of 64 x 64, 6 LDS-DMA, one workgroup per CU", 256, 8, 32),
          2: ("256 x 256, 8 waves of 128 x 64, 8 LDS-DMA, one workgroup per CU", 256, 8, 64),
          3: ("128 x 128, 4 waves of 64 x 64, 8 LDS-DMA, two workgroups per CU", 512, 4, 32)}
for shp, (name, blocks, nw, mf) in shapes.items():
    for bits in (0, 7, 15):
        iters = 4000
        fl = blocks * nw * iters * mf * 2.0 * 16 * 16 * 32
        mode = 64 * shp + bits
        for _ in range(2):
            L.check(lib.mgd_debug_wgrad_skeleton(L.ptr(out), blocks, iters, mode, L.stream_ptr()), "skeleton")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            L.check(lib.mgd_debug_wgrad_skeleton(L.ptr(out), blocks, iters, mode, L.stream_ptr()), "skeleton")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{name}, bits {bits:2d}: {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s", flush=True)

print("gather-GEMM K-step skeleton (ds_read_b128 fragments, one barrier per 64-deep step):")
gshapes = {5: ("128 x 128, 4 waves of 32 x 128: fragment reads + barrier only, three per CU", 768, 4, 32),
           0: ("128 x 128, 4 waves of 32 x 128, 4 LDS-DMA + 4 register weight loads (conv_gemm8_kernel), three per CU", 768, 4, 32),
           1: ("128 x 128, 4 waves of 32 x 128, weights from LDS, 6 LDS-DMA, three per CU", 768, 4, 32),
           2: ("256 x 128, 8 waves of 64 x 64, weights from LDS, 6 LDS-DMA, one per CU", 256, 8, 32),
           3: ("256 x 256, 8 waves of 128 x 64, weights from LDS, 6 LDS-DMA (of 8), one per CU", 256, 8, 64),
           4: ("128 x 256, 4 waves of 64 x 128, 8 register weight loads + 6 LDS-DMA (of 8), two per CU", 512, 4, 64)}
for shp, (name, blocks, nw, mf) in gshapes.items():
    iters = 3000
    fl = blocks * nw * iters * mf * 2.0 * 16 * 16 * 32
    for _ in range(2):
        L.check(lib.mgd_debug_gemm_skeleton(L.ptr(out), blocks, iters, shp, L.stream_ptr()), "gemm skeleton")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        L.check(lib.mgd_debug_gemm_skeleton(L.ptr(out), blocks, iters, shp, L.stream_ptr()), "gemm skeleton")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name}: {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s", flush=True)
