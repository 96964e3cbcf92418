// On-device batch augmentation on gfx950 (HBM-bound copies + tiny box lists).
// Replaces tf_random_mosaic (reference multigriddet/data/generators.py:561-1009; effective code =
// process_entire_batch :794-998), tf_random_gridmask (:1164-1282) and tf_random_mixup (:1012-1161).
// Random draws are made by the caller on the host (see multigriddet_amd/data/augment.py) and arrive as
// small device arrays, so that the CPU oracle can replay exactly the same draws.
#include "common.h"

namespace {

// ordered append of kept boxes: one block (256 threads) walks candidates in order
__device__ int block_ordered_append(bool keep, const float (&box)[5], float* out, int base, int cap, int* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long m = __ballot(keep);
  int inwave = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __popcll(m);
  __syncthreads();
  int off = base;
  for (int w = 0; w < wave; ++w) off += wsum[w];
  if (keep) {
    int pos = off + inwave;
    if (pos < cap)
#pragma unroll
      for (int j = 0; j < 5; ++j) out[pos * 5 + j] = box[j];
  }
  int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + tot;
}

__global__ __launch_bounds__(256) void mosaic_image_kernel(const float* __restrict__ img, const int* __restrict__ src,
                                                           const int* __restrict__ crop, float* __restrict__ out, int B,
                                                           int S) {
  // one thread = one pixel (3 floats); rows are contiguous so loads/stores stay coalesced
  long long npx = (long long)B * S * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (long long)gridDim.x * blockDim.x) {
    int x = (int)(i % S);
    long long t = i / S;
    int y = (int)(t % S);
    int b = (int)(t / S);
    int cx = crop[b * 2], cy = crop[b * 2 + 1];
    int q = y < cy ? (x < cx ? 0 : 3) : (x < cx ? 1 : 2);
    int sb = src[b * 4 + q];
    const float* p = img + (((long long)sb * S + y) * S + x) * 3;
    float* o = out + i * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
}

__global__ __launch_bounds__(256) void mosaic_boxes_kernel(const float* __restrict__ boxes, const int* __restrict__ src,
                                                           const int* __restrict__ crop, float min_wh,
                                                           float* __restrict__ out, int S, int M_in, int M_out,
                                                           int* overflow) {
  __shared__ int wsum[4];
  const int b = blockIdx.x;
  const float cx = (float)crop[b * 2], cy = (float)crop[b * 2 + 1], W = (float)S, H = (float)S;
  float* ob = out + (long long)b * M_out * 5;
  for (int i = threadIdx.x; i < M_out * 5; i += 256) ob[i] = 0.f;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < 4; ++q) {
    const float* sbx = boxes + (long long)src[b * 4 + q] * M_in * 5;
    for (int t0 = 0; t0 < M_in; t0 += 256) {
      int t = t0 + threadIdx.x;
      bool keep = false;
      float bx[5] = {0, 0, 0, 0, 0};
      if (t < M_in) {
        float x1 = sbx[t * 5], y1 = sbx[t * 5 + 1], x2 = sbx[t * 5 + 2], y2 = sbx[t * 5 + 3];
        bool valid = (x2 - x1) * (y2 - y1) > 0.0f;
        float lx = (q == 0 || q == 1) ? 0.f : cx, hx = (q == 0 || q == 1) ? cx : W;
        float ly = (q == 0 || q == 3) ? 0.f : cy, hy = (q == 0 || q == 3) ? cy : H;
        keep = valid && y2 > ly && y1 < hy && x2 > lx && x1 < hx;
        float nx1 = fmaxf(x1, lx), ny1 = fmaxf(y1, ly), nx2 = fminf(x2, hx), ny2 = fminf(y2, hy);
        keep = keep && (nx2 - nx1) >= min_wh && (ny2 - ny1) >= min_wh;
        bx[0] = nx1; bx[1] = ny1; bx[2] = nx2; bx[3] = ny2; bx[4] = sbx[t * 5 + 4];
      }
      base = block_ordered_append(keep, bx, ob, base, M_out, wsum);
    }
  }
  if (threadIdx.x == 0 && base > M_out) atomicExch(overflow, 1);
}

__global__ __launch_bounds__(256) void gridmask_image_kernel(float* __restrict__ img, const int* __restrict__ apply,
                                                             const int* __restrict__ par, int B, int S) {
  long long npx = (long long)B * S * S;
  const int hh = (int)ceilf(sqrtf(2.0f * (float)S * (float)S));
  const int off = (hh - S) / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (long long)gridDim.x * blockDim.x) {
    long long t = i / S;
    int y = (int)(t % S);
    int b = (int)(t / S);
    if (!apply[b]) continue;
    int d = par[b * 3], l = par[b * 3 + 1], st = par[b * 3 + 2];
    int r = ((y + off - st) % d + d) % d;
    float keep = r < l ? 1.f : 0.f;      // the reference multiplies by (1 - mask): only the stripes survive
    float* o = img + i * 3;
    o[0] = fminf(fmaxf(o[0] * keep, 0.f), 255.f);
    o[1] = fminf(fmaxf(o[1] * keep, 0.f), 255.f);
    o[2] = fminf(fmaxf(o[2] * keep, 0.f), 255.f);
  }
}

__global__ __launch_bounds__(256) void gridmask_boxes_kernel(float* __restrict__ boxes, const int* __restrict__ apply,
                                                             const int* __restrict__ par, int S, int M, float frac) {
  __shared__ int wsum[4];
  const int b = blockIdx.x;
  if (!apply[b]) return;
  const int hh = (int)ceilf(sqrtf(2.0f * (float)S * (float)S));
  const int off = (hh - S) / 2;
  const int d = par[b * 3], l = par[b * 3 + 1], st = par[b * 3 + 2];
  float* bb = boxes + (long long)b * M * 5;
  extern __shared__ float stage[];   // M*5 compacted copy
  for (int i = threadIdx.x; i < M * 5; i += 256) stage[i] = 0.f;
  __syncthreads();
  int base = 0;
  for (int t0 = 0; t0 < M; t0 += 256) {
    int t = t0 + threadIdx.x;
    bool keep = false;
    float bx[5] = {0, 0, 0, 0, 0};
    if (t < M) {
#pragma unroll
      for (int j = 0; j < 5; ++j) bx[j] = bb[t * 5 + j];
      int x1 = (int)bx[0], y1 = (int)bx[1], x2 = (int)bx[2], y2 = (int)bx[3];
      float area = (float)((x2 - x1) * (y2 - y1));
      int xa = max(x1, 0), xb = min(x2, S), ya = max(y1, 0), yb = min(y2, S);
      int rows = 0;
      for (int y = ya; y < yb; ++y) {
        int r = ((y + off - st) % d + d) % d;
        rows += r < l ? 0 : 1;            // un-inverted mask: 1 outside the stripes
      }
      float valid = (float)rows * (float)max(xb - xa, 0);
      keep = valid > area * frac;
    }
    base = block_ordered_append(keep, bx, stage, base, M, wsum);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < M * 5; i += 256) bb[i] = stage[i];
}

__global__ __launch_bounds__(256) void mixup_image_kernel(const float* __restrict__ img, const int* __restrict__ partner,
                                                          const float* __restrict__ lam, float* __restrict__ out, int B,
                                                          long long per) {
  long long n = (long long)B * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int b = (int)(i / per);
    long long r = i - (long long)b * per;
    float la = lam[b];
    out[i] = la * img[i] + (1.0f - la) * img[(long long)partner[b] * per + r];
  }
}

__global__ __launch_bounds__(256) void mixup_boxes_kernel(const float* __restrict__ boxes, const int* __restrict__ partner,
                                                          float* __restrict__ out, int M_in, int M_out, int* overflow) {
  __shared__ int wsum[4];
  const int b = blockIdx.x;
  float* ob = out + (long long)b * M_out * 5;
  for (int i = threadIdx.x; i < M_out * 5; i += 256) ob[i] = 0.f;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < 2; ++q) {
    const float* sb = boxes + (long long)(q == 0 ? b : partner[b]) * M_in * 5;
    for (int t0 = 0; t0 < M_in; t0 += 256) {
      int t = t0 + threadIdx.x;
      bool keep = false;
      float bx[5] = {0, 0, 0, 0, 0};
      if (t < M_in) {
#pragma unroll
        for (int j = 0; j < 5; ++j) bx[j] = sb[t * 5 + j];
        keep = (bx[2] - bx[0]) * (bx[3] - bx[1]) > 0.0f;
      }
      base = block_ordered_append(keep, bx, ob, base, M_out, wsum);
    }
  }
  if (threadIdx.x == 0 && base > M_out && overflow) atomicExch(overflow, 1);
}

inline int grid_px(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mgd_mosaic(const float* images, const float* boxes, int B, int S, int M_in, const int32_t* src_idx,
                          const int32_t* crop_xy, float min_wh, float* out_images, float* out_boxes, int M_out,
                          int32_t* overflow, void* stream) {
  MGD_REQUIRE(images && boxes && src_idx && crop_xy && out_images && out_boxes && overflow, "mosaic: null pointer");
  MGD_REQUIRE(B >= 4, "mosaic: needs a batch of at least 4 images (reference generators.py:611), got %d", B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mosaic_image_kernel, dim3(grid_px((long long)B * S * S)), dim3(256), 0, st, images, src_idx,
                     crop_xy, out_images, B, S);
  hipLaunchKernelGGL(mosaic_boxes_kernel, dim3(B), dim3(256), 0, st, boxes, src_idx, crop_xy, min_wh, out_boxes, S,
                     M_in, M_out, overflow);
  MGD_CHECK_LAUNCH("mosaic");
  return MGD_OK;
}

extern "C" int mgd_gridmask(float* images, float* boxes, int B, int S, int M, const int32_t* apply,
                            const int32_t* d_l_off, float keep_frac, void* stream) {
  MGD_REQUIRE(images && boxes && apply && d_l_off, "gridmask: null pointer");
  MGD_REQUIRE((size_t)M * 5 * 4 <= 60 * 1024, "gridmask: M=%d too large", M);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gridmask_image_kernel, dim3(grid_px((long long)B * S * S)), dim3(256), 0, st, images, apply,
                     d_l_off, B, S);
  hipLaunchKernelGGL(gridmask_boxes_kernel, dim3(B), dim3(256), (size_t)M * 5 * 4, st, boxes, apply, d_l_off, S, M,
                     keep_frac);
  MGD_CHECK_LAUNCH("gridmask");
  return MGD_OK;
}

extern "C" int mgd_mixup(const float* images, const float* boxes, int B, int S, int M_in, const int32_t* partner,
                         const float* lam, float* out_images, float* out_boxes, int M_out, void* stream) {
  MGD_REQUIRE(images && boxes && partner && lam && out_images && out_boxes, "mixup: null pointer");
  hipStream_t st = (hipStream_t)stream;
  long long per = (long long)S * S * 3;
  hipLaunchKernelGGL(mixup_image_kernel, dim3(grid_px((long long)B * per)), dim3(256), 0, st, images, partner, lam,
                     out_images, B, per);
  hipLaunchKernelGGL(mixup_boxes_kernel, dim3(B), dim3(256), 0, st, boxes, partner, out_boxes, M_in, M_out,
                     (int*)nullptr);
  MGD_CHECK_LAUNCH("mixup");
  return MGD_OK;
}
