// Diagnostic kernels of libmgd_hip_diag.so (never part of libmgd_hip.so): the bare MFMA stream and the K-step skeletons of the
// weight gradient and of the gather-GEMM (tools/mfma_peak.py), the diagnostic flag word.  See include/mgd_hip_diag.h.
#include "../conv_common.hpp"
#include "../../../include/mgd_hip_diag.h"

namespace {

template <int NACC>
__global__ __launch_bounds__(256) MGD_VGPR_MFMA void mfma_peak_kernel(float* out, int iters) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (bf16_t)(0x3f80 + threadIdx.x + i); b[i] = (bf16_t)(0x3c00 + 3 * threadIdx.x + i); }
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  f32x4 t = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) t += acc[i];
  if (t[0] == 123.456f) out[threadIdx.x] = t[1] + t[2] + t[3];
}

// Diagnostic: the weight gradient's K-step rebuilt piece by piece around the bare MFMA stream (128 x 64 tile: 16 MFMAs, 24
// transposed fragment reads, 6 LDS-DMA instructions and one barrier per wave and step; 48 KB of LDS so that three blocks share a
// CU).  MODE bits: 1 = fragment reads (burst of 12, lgkmcnt(0), 12 under the first 8 MFMAs, lgkmcnt(0) - as conv_wgrad4_kernel),
// 2 = s_barrier per step, 4 = six LDS-DMA instructions per step with every lane out of range, 8 = the reads one or two per MFMA
// gap with counted waits instead, 16 = s_setprio 1 around the MFMAs.
// MT x NT: 16 x 16 tiles per wave; NW waves per workgroup; D1 + D2 LDS-DMA instructions per wave and step; LKB KiB of LDS
template <int MODE, int MT = 4, int NT = 2, int NW = 4, int D1 = 4, int D2 = 2, int LKB = 48>
__global__ __launch_bounds__(64 * NW) MGD_VGPR_MFMA void wgrad_skel_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 48 * 1024 / 4; i += 64 * NW) ((unsigned*)smem)[i] = 0x3f803f80u + i;
  __syncthreads();
  const unsigned base = lds_addr(smem) + (lane & 3) * 8 + ((lane >> 2) & 15) * 288;   // 288-byte rows: the eight rows of a 32-lane group on disjoint banks
  unsigned ra[2][MT][2], rb[2][NT][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MT; ++m) ra[kk][m][h] = base + kk * 8192 + h * 1024 + ((m ^ (lane & 3)) * 32);
#pragma unroll
      for (int n = 0; n < NT; ++n) rb[kk][n][h] = base + 16384 + kk * 4096 + h * 1024 + ((n ^ (lane & 1)) * 32);
    }
  i32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)out);
  srd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(((unsigned long long)out >> 32) & 0xFFFFu));
  srd[2] = 0;                                  // num_records 0: every lane out of range
  srd[3] = 0x00020000;
  unsigned vo4[D1], vo2[D2];
#pragma unroll
  for (int i = 0; i < D1; ++i) vo4[i] = lane * 16 + i * 64;
#pragma unroll
  for (int i = 0; i < D2; ++i) vo2[i] = lane * 16 + i * 64;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x4 fa[2][MT][2], fb[2][NT][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int m = 0; m < MT; ++m) fa[kk][m][0] = fa[kk][m][1] = s16x4{(short)(0x3f80 + lane), 1, 2, 3};
#pragma unroll
    for (int n = 0; n < NT; ++n) fb[kk][n][0] = fb[kk][n][1] = s16x4{(short)(0x3c00 + lane), 1, 2, 3};
  }
  auto mfma1 = [&](int kk, int m, int n) {
    s16x8 av = __builtin_shufflevector(fa[kk][m][0], fa[kk][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
    s16x8 bv = __builtin_shufflevector(fb[kk][n][0], fb[kk][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[m][n], 0, 0, 0);
  };
  auto rd = [&](int kk, int q) {
    if (q < 2 * NT) tr_read_asm<0>(fb[kk][q >> 1][q & 1], rb[kk][q >> 1][q & 1]);
    else { const int r = q - 2 * NT; tr_read_asm<0>(fa[kk][r >> 1][r & 1], ra[kk][r >> 1][r & 1]); }
  };
  auto touch_half = [&](int kk) {
#pragma unroll
    for (int m = 0; m < MT; ++m) { touch(fa[kk][m][0]); touch(fa[kk][m][1]); }
#pragma unroll
    for (int n = 0; n < NT; ++n) { touch(fb[kk][n][0]); touch(fb[kk][n][1]); }
  };
  constexpr int R1 = 2 * MT + 2 * NT;
  f32x4 stg[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) stg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE & 2) __builtin_amdgcn_s_barrier();
    if (MODE & 4) {
      dma_rows_asm<D1, 1024>(vo4, srd, lds_addr(smem) + (wave & 3) * 1024);
      dma_rows_asm<D2, 1024>(vo2, srd, lds_addr(smem) + 16384 + (wave & 3) * 1024);
    }
    if (MODE & 2048) {                         // six LDS-DMA pieces behind ONE M0 set-up, told apart by the immediate offset
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %3, %2, 0 offen lds\n\tbuffer_load_dwordx4 %3, %2, 0 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %3, %2, 0 offen offset:2048 lds\n\tbuffer_load_dwordx4 %3, %2, 0 offen offset:3072 lds\n\t"
                   "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %3, %2, 0 offen lds\n\tbuffer_load_dwordx4 %3, %2, 0 offen offset:1024 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(lds_addr(smem) + (wave & 3) * 4096), "s"(srd), "v"(vo4[0]), "s"(lds_addr(smem) + 16384 + (wave & 3) * 2048)
                   : "memory");
    }
    if (MODE & 32) {                           // the same six pieces as register loads + ds_write_b128 of the previous step's
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(smem) + 24576 + (wave & 3) * 1024 + i * 4096 + lane * 16), "v"(stg[i]) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(stg[i]) : "v"(vo4[i % D1]), "s"(srd) : "memory");
    }
    if (MODE & 16) __builtin_amdgcn_s_setprio(1);
    if ((MODE & 1) && !(MODE & 8)) {
#pragma unroll
      for (int q = 0; q < R1; ++q) rd(0, q);
      wait_lgkm_dyn(0);
#pragma unroll
      for (int q = 0; q < R1; ++q) rd(1, q);
      touch_half(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) mfma1(0, m, n);
      wait_lgkm_dyn(0);
      touch_half(1);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) mfma1(1, m, n);
    } else if (MODE & 1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) rd(0, q);
      int q1 = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        wait_lgkm_dyn(2 * (MT - 1 - m) + q1);
        if (m == 0) {
#pragma unroll
          for (int n = 0; n < NT; ++n) { touch(fb[0][n][0]); touch(fb[0][n][1]); }
        }
        touch(fa[0][m][0]); touch(fa[0][m][1]);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          mfma1(0, m, n);
          const int g = m * NT + n;
          while (q1 < (g + 1) * R1 / (MT * NT)) { rd(1, q1); ++q1; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        wait_lgkm_dyn(2 * (MT - 1 - m));
        if (m == 0) {
#pragma unroll
          for (int n = 0; n < NT; ++n) { touch(fb[1][n][0]); touch(fb[1][n][1]); }
        }
        touch(fa[1][m][0]); touch(fa[1][m][1]);
#pragma unroll
        for (int n = 0; n < NT; ++n) mfma1(1, m, n);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) mfma1(kk, m, n);
    }
    if (MODE & 16) __builtin_amdgcn_s_setprio(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 t = acc[0][0];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) t += acc[m][n];
  if (t[0] == 123.456f) out[threadIdx.x] = t[1] + t[2] + t[3];
}

template <int MODE, int MT = 4, int NT = 2, int NW = 4, int D1 = 4, int D2 = 2, int LKB = 48>
static void launch_skel(float* out, int blocks, int iters, hipStream_t st) {
  auto k = wgrad_skel_kernel<MODE, MT, NT, NW, D1, D2, LKB>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), LKB * 1024, st, out, iters);
}

// mode = bits (see wgrad_skel_kernel) + 64 * shape: 0 = today's 128 x 64 tile (4 waves of 64 x 32, 6 LDS-DMA per wave and step,
// 48 KB: three workgroups per CU); 1 = 256 x 128, 8 waves of 64 x 64, 6 LDS-DMA, 96 KB (one workgroup per CU); 2 = 256 x 256, 8
// waves of 128 x 64, 8 LDS-DMA, 128 KB; 3 = 128 x 128, 4 waves of 64 x 64, 8 LDS-DMA, 64 KB (two workgroups per CU)
}  // namespace

extern "C" int mgd_debug_wgrad_skeleton(float* out, int blocks, int iters, int mode, void* stream) {
  MGD_REQUIRE(out && blocks >= 1 && iters >= 1, "wgrad_skeleton: arguments");
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: launch_skel<0>(out, blocks, iters, st); break;
    case 1: launch_skel<1>(out, blocks, iters, st); break;
    case 2: launch_skel<2>(out, blocks, iters, st); break;
    case 3: launch_skel<3>(out, blocks, iters, st); break;
    case 4: launch_skel<4>(out, blocks, iters, st); break;
    case 7: launch_skel<7>(out, blocks, iters, st); break;
    case 9: launch_skel<9>(out, blocks, iters, st); break;
    case 11: launch_skel<11>(out, blocks, iters, st); break;
    case 15: launch_skel<15>(out, blocks, iters, st); break;
    case 23: launch_skel<23>(out, blocks, iters, st); break;
    case 31: launch_skel<31>(out, blocks, iters, st); break;
    case 32: launch_skel<32>(out, blocks, iters, st); break;
    case 35: launch_skel<35>(out, blocks, iters, st); break;
    case 2048: launch_skel<2048>(out, blocks, iters, st); break;
    case 2051: launch_skel<2051>(out, blocks, iters, st); break;
    case 43: launch_skel<43>(out, blocks, iters, st); break;
    case 64 + 0: launch_skel<0, 4, 4, 8, 4, 2, 96>(out, blocks, iters, st); break;
    case 64 + 7: launch_skel<7, 4, 4, 8, 4, 2, 96>(out, blocks, iters, st); break;
    case 64 + 15: launch_skel<15, 4, 4, 8, 4, 2, 96>(out, blocks, iters, st); break;
    case 128 + 0: launch_skel<0, 8, 4, 8, 4, 4, 128>(out, blocks, iters, st); break;
    case 128 + 7: launch_skel<7, 8, 4, 8, 4, 4, 128>(out, blocks, iters, st); break;
    case 128 + 15: launch_skel<15, 8, 4, 8, 4, 4, 128>(out, blocks, iters, st); break;
    case 192 + 0: launch_skel<0, 4, 4, 4, 4, 4, 64>(out, blocks, iters, st); break;
    case 192 + 7: launch_skel<7, 4, 4, 4, 4, 4, 64>(out, blocks, iters, st); break;
    case 192 + 15: launch_skel<15, 4, 4, 4, 4, 4, 64>(out, blocks, iters, st); break;
    default: MGD_REQUIRE(false, "wgrad_skeleton: mode %d not built", mode);
  }
  MGD_CHECK_LAUNCH("wgrad_skeleton");
  return MGD_OK;
}

namespace {

// Diagnostic: the gather-GEMM's K-step around the bare MFMA stream.  Per wave and 64-deep step: 2 * MT * NT MFMAs; pixel
// fragments by ds_read_b128 (2 * NT); weight fragments either as WL register loads (buffer_load_dwordx4, out of range - the form
// of conv_gemm8_kernel: 2 * MT of them) or by ds_read_b128 from LDS (WL = 0: 2 * MT reads - weights staged by LDS-DMA like the
// pixels); DX LDS-DMA instructions (out of range); one barrier.  NW waves per workgroup, LKB KiB of LDS (sets workgroups per CU).
template <int MT, int NT, int NW, int DX, int WL, int LKB>
__global__ __launch_bounds__(64 * NW) MGD_VGPR_MFMA void gemm_skel_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 32 * 1024 / 4; i += 64 * NW) ((unsigned*)smem)[i] = 0x3f803f80u + i;
  __syncthreads();
  // conflict-free 16-byte fragment reads: row = lane & 15 (128-byte rows, chunk XOR-swizzled by the row), k-group = lane >> 4
  const unsigned base = lds_addr(smem) + (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  i32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)out);
  srd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(((unsigned long long)out >> 32) & 0xFFFFu));
  srd[2] = 0;
  srd[3] = 0x00020000;
  unsigned vo[DX > 0 ? DX : 1];
#pragma unroll
  for (int i = 0; i < DX; ++i) vo[i] = lane * 16 + i * 64;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 wf[2][MT], xf[NT];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 8; ++i) wf[kk][m][i] = (bf16_t)(0x3f80 + lane + i);
  bf16x8 wn[2][MT];                               // WL: the NEXT step's weight fragments, in flight under this step's MFMAs
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    if constexpr (DX > 0) dma_rows_asm<DX, 1024>(vo, srd, lds_addr(smem) + 16384 + (wave & 3) * 1024);
    if constexpr (WL > 0) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(wn[kk][m]) : "v"(lane * 16u + (unsigned)(kk * MT + m) * 1024u), "s"(srd) : "memory");
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[n]) : "v"(base + (n & 7) * 2048), "n"(0));
      if constexpr (WL == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][m]) : "v"(base + (m & 7) * 2048), "n"(64));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(xf[n]));
#pragma unroll
      for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(wf[kk][m]));
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][m], xf[n], acc[m][n], 0, 0, 0);
    }
    if constexpr (WL > 0) {                       // they have had the whole step: wait, then they are the current set
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          asm volatile("" : "+v"(wn[kk][m]));
          wf[kk][m] = wn[kk][m];
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 t = acc[0][0];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) t += acc[m][n];
  if (t[0] == 123.456f) out[threadIdx.x] = t[1] + t[2] + t[3];
}

template <int MT, int NT, int NW, int DX, int WL, int LKB>
static void launch_gemm_skel(float* out, int blocks, int iters, hipStream_t st) {
  auto k = gemm_skel_kernel<MT, NT, NW, DX, WL, LKB>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), LKB * 1024, st, out, iters);
}

// shape: 0 = conv_gemm8_kernel's step (4 waves of 32 x 128: MT 2, NT 8, 4 LDS-DMA + 4 register weight loads, 48 KB: three per CU);
// 1 = the same with the weights read from LDS (4 + 4 LDS-DMA); 2 = 256 x 128 block, 8 waves of 64 x 64, weights from LDS, 2 + 4
// LDS-DMA per wave, one workgroup per CU; 3 = 256 x 256 block, 8 waves of 128 x 64, weights from LDS, 4 + 4 LDS-DMA, one per CU;
// 4 = 128 x 256 block, 4 waves of 64 x 128 (MT 4, NT 8), register weights (8) + 8 LDS-DMA, two per CU
}  // namespace

extern "C" int mgd_debug_gemm_skeleton(float* out, int blocks, int iters, int shape, void* stream) {
  MGD_REQUIRE(out && blocks >= 1 && iters >= 1, "gemm_skeleton: arguments");
  hipStream_t st = (hipStream_t)stream;
  switch (shape) {
    case 0: launch_gemm_skel<2, 8, 4, 4, 1, 48>(out, blocks, iters, st); break;
    case 1: launch_gemm_skel<2, 8, 4, 6, 0, 48>(out, blocks, iters, st); break;
    case 2: launch_gemm_skel<4, 4, 8, 6, 0, 100>(out, blocks, iters, st); break;
    case 3: launch_gemm_skel<8, 4, 8, 6, 0, 130>(out, blocks, iters, st); break;
    case 4: launch_gemm_skel<4, 8, 4, 6, 1, 70>(out, blocks, iters, st); break;
    case 5: launch_gemm_skel<2, 8, 4, 0, 0, 48>(out, blocks, iters, st); break;      // reads + barrier only
    default: MGD_REQUIRE(false, "gemm_skeleton: shape %d not built", shape);
  }
  MGD_CHECK_LAUNCH("gemm_skeleton");
  return MGD_OK;
}
extern "C" int mgd_debug_mfma_peak(float* out, int blocks, int iters, int nacc, void* stream) {
  MGD_REQUIRE(out && blocks >= 1 && iters >= 1, "mfma_peak: arguments");
  if (nacc == 16) hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(mfma_peak_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  MGD_CHECK_LAUNCH("mfma_peak");
  return MGD_OK;
}

// ---- vector-memory issue rate of a CU (tools/vmem_rate.py): NW waves of ONE workgroup per CU each issue iters x 8 vector-memory
// instructions of one kind and nothing else (a vmcnt(0) per group of 8).  kind: 0 = buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB
// per wave-instruction) over an L2-resident window, 1 = the same with every lane out of range (no memory traffic at all),
// 2 = buffer_load_dwordx4 into registers, 3 = buffer_load_dword ... lds (256 B per wave-instruction), 4 = buffer_load_dword
// into a register, 5 = global_load_dwordx4 (flat addressing) into registers, 6 = buffer_load_dwordx4 into registers, every lane
// out of range.
namespace {
template <int NW, int KIND>
__global__ __launch_bounds__(64 * NW) void vmem_rate_kernel(const unsigned char* buf, float* out, int iters, unsigned window) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  i32x4 srd;
  {
    const unsigned long long p = (unsigned long long)buf;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    srd[2] = __builtin_amdgcn_readfirstlane((int)window);
    srd[3] = 0x00020000;
  }
  constexpr bool oob = KIND == 1 || KIND == 6;
  constexpr int W = (KIND == 3 || KIND == 4) ? 4 : 16;              // bytes per lane
  unsigned vo = oob ? 0xFFFFFFF0u : (unsigned)((blockIdx.x * NW + wave) * 8192 % (window - 8192) + lane * W);
  const unsigned lds = lds_addr(smem) + wave * 8192;
  f32x4 r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_nop 4" ::: "memory");                                // H1: the descriptor words came from v_readfirstlane
  if constexpr (KIND == 0 || KIND == 1 || KIND == 3) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds) : "memory");
  }
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0 || KIND == 1) {
      asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %0, %1, 0 offen offset:2048 lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:3072 lds\n\t"
                   "buffer_load_dwordx4 %0, %1, 0 offen lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %0, %1, 0 offen offset:2048 lds\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:3072 lds\n\t"
                   "s_waitcnt vmcnt(0)" ::"v"(vo), "s"(srd) : "memory");
    } else if constexpr (KIND == 3) {
      asm volatile("buffer_load_dword %0, %1, 0 offen lds\n\tbuffer_load_dword %0, %1, 0 offen offset:256 lds\n\t"
                   "buffer_load_dword %0, %1, 0 offen offset:512 lds\n\tbuffer_load_dword %0, %1, 0 offen offset:768 lds\n\t"
                   "buffer_load_dword %0, %1, 0 offen lds\n\tbuffer_load_dword %0, %1, 0 offen offset:256 lds\n\t"
                   "buffer_load_dword %0, %1, 0 offen offset:512 lds\n\tbuffer_load_dword %0, %1, 0 offen offset:768 lds\n\t"
                   "s_waitcnt vmcnt(0)" ::"v"(vo), "s"(srd) : "memory");
    } else if constexpr (KIND == 2 || KIND == 6) {
      asm volatile("buffer_load_dwordx4 %0, %8, %9, 0 offen\n\tbuffer_load_dwordx4 %1, %8, %9, 0 offen offset:1024\n\t"
                   "buffer_load_dwordx4 %2, %8, %9, 0 offen offset:2048\n\tbuffer_load_dwordx4 %3, %8, %9, 0 offen offset:3072\n\t"
                   "buffer_load_dwordx4 %4, %8, %9, 0 offen\n\tbuffer_load_dwordx4 %5, %8, %9, 0 offen offset:1024\n\t"
                   "buffer_load_dwordx4 %6, %8, %9, 0 offen offset:2048\n\tbuffer_load_dwordx4 %7, %8, %9, 0 offen offset:3072\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(vo), "s"(srd) : "memory");
    } else if constexpr (KIND == 4) {
      float t[8];
      asm volatile("buffer_load_dword %0, %8, %9, 0 offen\n\tbuffer_load_dword %1, %8, %9, 0 offen offset:256\n\t"
                   "buffer_load_dword %2, %8, %9, 0 offen offset:512\n\tbuffer_load_dword %3, %8, %9, 0 offen offset:768\n\t"
                   "buffer_load_dword %4, %8, %9, 0 offen\n\tbuffer_load_dword %5, %8, %9, 0 offen offset:256\n\t"
                   "buffer_load_dword %6, %8, %9, 0 offen offset:512\n\tbuffer_load_dword %7, %8, %9, 0 offen offset:768\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                   : "v"(vo), "s"(srd) : "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j][0] += t[j];
    } else {
      asm volatile("global_load_dwordx4 %0, %8, %9 offset:0\n\tglobal_load_dwordx4 %1, %8, %9 offset:1024\n\t"
                   "global_load_dwordx4 %2, %8, %9 offset:2048\n\tglobal_load_dwordx4 %3, %8, %9 offset:3072\n\t"
                   "global_load_dwordx4 %4, %8, %9 offset:0\n\tglobal_load_dwordx4 %5, %8, %9 offset:1024\n\t"
                   "global_load_dwordx4 %6, %8, %9 offset:2048\n\tglobal_load_dwordx4 %7, %8, %9 offset:3072\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(vo), "s"(buf) : "memory");
    }
  }
  f32x4 t = r[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) t += r[j];
  if (t[0] == 123.456f) out[threadIdx.x] = t[1] + smem[lane];          // keep everything alive
}

template <int NW>
void launch_vmem_rate(const unsigned char* buf, float* out, int blocks, int iters, int kind, unsigned window, hipStream_t st) {
#define MGD_VR(K) hipLaunchKernelGGL((vmem_rate_kernel<NW, K>), dim3(blocks), dim3(64 * NW), NW * 8192, st, buf, out, iters, window)
  switch (kind) {
    case 0: MGD_VR(0); break; case 1: MGD_VR(1); break; case 2: MGD_VR(2); break; case 3: MGD_VR(3); break;
    case 4: MGD_VR(4); break; case 5: MGD_VR(5); break; default: MGD_VR(6); break;
  }
#undef MGD_VR
}
}  // namespace

extern "C" int mgd_debug_vmem_rate(const void* buf, unsigned window, float* out, int blocks, int waves, int iters, int kind,
                                   void* stream) {
  MGD_REQUIRE(buf && out && blocks >= 1 && iters >= 1 && kind >= 0 && kind <= 6 && window >= (1u << 20),
              "vmem_rate: arguments (window >= 1 MiB)");
  hipStream_t st = (hipStream_t)stream;
  switch (waves) {
    case 1: launch_vmem_rate<1>((const unsigned char*)buf, out, blocks, iters, kind, window, st); break;
    case 2: launch_vmem_rate<2>((const unsigned char*)buf, out, blocks, iters, kind, window, st); break;
    case 4: launch_vmem_rate<4>((const unsigned char*)buf, out, blocks, iters, kind, window, st); break;
    case 8: launch_vmem_rate<8>((const unsigned char*)buf, out, blocks, iters, kind, window, st); break;
    default: MGD_REQUIRE(false, "vmem_rate: waves must be 1, 2, 4 or 8");
  }
  MGD_CHECK_LAUNCH("vmem_rate");
  return MGD_OK;
}

static int g_diag_flags = 0;
extern "C" int mgd_diag_set_flags(int flags) {
  g_diag_flags = flags;
  return MGD_OK;
}
extern "C" int mgd_diag_flags_value(void) {
  return g_diag_flags;
}
