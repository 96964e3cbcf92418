// Inference-side letterbox on the device: uint8 HWC frame -> bicubic resize -> centred paste on a 128-grey canvas ->
// /255 -> fp32 NHWC model input.  Replaces letterbox_resize + preprocess_image of the reference
// (multigriddet/utils/preprocessing.py:12-90), which runs PIL's Image.resize(BICUBIC) on the host.
//
// PIL's resampler is integer arithmetic: per output column/row a window [xmin, xmin+xmax) of source pixels and 22-bit
// fixed-point coefficients (ImagingResample: precompute_coeffs + normalize_coeffs_8bpc), a horizontal pass into an
// 8-bit intermediate, then a vertical pass; each sum starts at 1 << 21 and is shifted right by 22 and clamped to
// [0, 255].  The host builds the same coefficient tables in double precision (multigriddet_amd/utils/preprocessing.py)
// and the two kernels below repeat the two passes, so the result equals PIL's bit for bit.
// HBM-bound: reads H*W*3 bytes, writes Hd*Wd*12 bytes.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[y][xx][c] = horizontal pass, y in [0, H)
__global__ void letterbox_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int H, int W, int nw,
                                   const int32_t* __restrict__ kx, const int32_t* __restrict__ bx, int ksx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)H * nw) return;
  int y = (int)(i / nw), xx = (int)(i - (long long)y * nw);
  int xmin = bx[2 * xx], xmax = bx[2 * xx + 1];
  const int32_t* k = kx + (long long)xx * ksx;
  const uint8_t* row = src + ((long long)y * W + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xmax; ++x) {
    int kv = k[x];
    s0 += row[3 * x] * kv; s1 += row[3 * x + 1] * kv; s2 += row[3 * x + 2] * kv;
  }
  uint8_t* o = tmp + i * 3;
  o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// dst[yd][xd][c] = canvas (fill) outside the pasted image, else vertical pass of tmp, / 255
__global__ void letterbox_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ dst, int Hd, int Wd, int nh, int nw,
                                   int dy, int dx, const int32_t* __restrict__ ky, const int32_t* __restrict__ by, int ksy,
                                   float fill) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Hd * Wd) return;
  int yd = (int)(i / Wd), xd = (int)(i - (long long)yd * Wd);
  int yy = yd - dy, xx = xd - dx;
  float* o = dst + i * 3;
  if (yy < 0 || yy >= nh || xx < 0 || xx >= nw) {
    float f = fill / 255.0f;
    o[0] = f; o[1] = f; o[2] = f;
    return;
  }
  int ymin = by[2 * yy], ymax = by[2 * yy + 1];
  const int32_t* k = ky + (long long)yy * ksy;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < ymax; ++y) {
    const uint8_t* p = tmp + ((long long)(y + ymin) * nw + xx) * 3;
    int kv = k[y];
    s0 += p[0] * kv; s1 += p[1] * kv; s2 += p[2] * kv;
  }
  o[0] = (float)clip8(s0) / 255.0f; o[1] = (float)clip8(s1) / 255.0f; o[2] = (float)clip8(s2) / 255.0f;
}

}  // namespace

extern "C" size_t mgd_letterbox_workspace_size(int H, int nw) { return (size_t)(H > 0 ? H : 0) * (size_t)(nw > 0 ? nw : 0) * 3; }

extern "C" int mgd_letterbox_u8(const uint8_t* src, int H, int W, float* dst, int Hd, int Wd, int nh, int nw, int dy, int dx,
                                const int32_t* kx, const int32_t* bx, int ksx, const int32_t* ky, const int32_t* by, int ksy,
                                float fill, void* ws, size_t ws_bytes, void* stream) {
  MGD_REQUIRE(src && dst && kx && bx && ky && by && ws, "letterbox: null pointer");
  MGD_REQUIRE(H >= 1 && W >= 1 && Hd >= 1 && Wd >= 1 && nh >= 1 && nw >= 1 && ksx >= 1 && ksy >= 1, "letterbox: sizes");
  MGD_REQUIRE(dy >= 0 && dx >= 0 && dy + nh <= Hd && dx + nw <= Wd, "letterbox: pasted image %dx%d at (%d,%d) exceeds canvas %dx%d",
              nh, nw, dy, dx, Hd, Wd);
  if (ws_bytes < mgd_letterbox_workspace_size(H, nw)) return mgd_set_error(MGD_ENOSPC, "letterbox: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(letterbox_h_kernel, dim3(cdiv((long long)H * nw, 256)), dim3(256), 0, st, src, (uint8_t*)ws, H, W, nw, kx,
                     bx, ksx);
  hipLaunchKernelGGL(letterbox_v_kernel, dim3(cdiv((long long)Hd * Wd, 256)), dim3(256), 0, st, (const uint8_t*)ws, dst, Hd, Wd,
                     nh, nw, dy, dx, ky, by, ksy, fill);
  MGD_CHECK_LAUNCH("letterbox_u8");
  return MGD_OK;
}
