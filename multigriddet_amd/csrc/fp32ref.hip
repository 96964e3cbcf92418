// Strict-parity (fp32) execution of the graph: fp32 NHWC activations, fp32 OHWI weights, fp32 accumulation - the
// reference's default numeric type (Keras Conv2D / BatchNormalization / LeakyReLU in float32, models/layers.py:43-95;
// mixed precision is opt-in there, trainers/trainer.py:122-129).  These kernels exist so that the 69-conv graph can be
// compared with the oracle end to end under a tight bound (Network(precision="fp32")); they are direct, one thread per
// output element, and make no attempt at speed - the bf16 MFMA path is the product's fast path.
#include "common.h"

namespace {

struct C32 {
  const float* x; const float* w; float* y; const float* bias; const float* addend;
  int N, H, W, Ci, Ho, Wo, Co, k, s;
};

// y[n,ho,wo,co] = bias[co] + sum_{kh,kw,ci} x[n, ho*s + kh - 1, wo*s + kw - 1, ci] * w[co][kh*k+kw][ci]   (k = 3)
// 'same' for stride 1; stride 2 = ZeroPadding2D(((1,0),(1,0))) + 'valid' (models/backbones/darknet.py:33-34), i.e. the same
// index arithmetic with pad 1 on the top/left only.  k = 1: no offset.
__global__ void conv_f32_fwd_kernel(C32 a) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long tot = (long long)a.N * a.Ho * a.Wo * a.Co;
  if (i >= tot) return;
  int co = (int)(i % a.Co);
  long long p = i / a.Co;
  int wo = (int)(p % a.Wo); p /= a.Wo;
  int ho = (int)(p % a.Ho);
  int n = (int)(p / a.Ho);
  const int pad = a.k / 2;
  float acc = a.bias ? a.bias[co] : 0.f;
  for (int kh = 0; kh < a.k; ++kh) {
    int h = ho * a.s + kh - pad;
    if ((unsigned)h >= (unsigned)a.H) continue;
    for (int kw = 0; kw < a.k; ++kw) {
      int w = wo * a.s + kw - pad;
      if ((unsigned)w >= (unsigned)a.W) continue;
      const float* xp = a.x + (((long long)n * a.H + h) * a.W + w) * a.Ci;
      const float* wp = a.w + ((long long)co * a.k * a.k + kh * a.k + kw) * a.Ci;
      for (int ci = 0; ci < a.Ci; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
    }
  }
  a.y[i] = acc;
}

// dx[n,h,w,ci] = addend + sum over (kh,kw,co) with ho*s + kh - pad == h, wo*s + kw - pad == w of dy[n,ho,wo,co] * w[co][kh,kw][ci]
__global__ void conv_f32_dgrad_kernel(C32 a) {     // here a.x = dy (N,Ho,Wo,Co), a.y = dx (N,H,W,Ci)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long tot = (long long)a.N * a.H * a.W * a.Ci;
  if (i >= tot) return;
  int ci = (int)(i % a.Ci);
  long long p = i / a.Ci;
  int w = (int)(p % a.W); p /= a.W;
  int h = (int)(p % a.H);
  int n = (int)(p / a.H);
  const int pad = a.k / 2;
  float acc = a.addend ? a.addend[i] : 0.f;
  for (int kh = 0; kh < a.k; ++kh) {
    int t = h + pad - kh;
    if (t < 0 || t % a.s) continue;
    int ho = t / a.s;
    if (ho >= a.Ho) continue;
    for (int kw = 0; kw < a.k; ++kw) {
      int u = w + pad - kw;
      if (u < 0 || u % a.s) continue;
      int wo = u / a.s;
      if (wo >= a.Wo) continue;
      const float* dp = a.x + (((long long)n * a.Ho + ho) * a.Wo + wo) * a.Co;
      const float* wp = a.w + ((long long)(kh * a.k + kw)) * a.Ci + ci;
      for (int co = 0; co < a.Co; ++co) acc = fmaf(dp[co], wp[(long long)co * a.k * a.k * a.Ci], acc);
    }
  }
  a.y[i] = acc;
}

// dw[co][kh,kw][ci] += sum over a slice of the pixels; grid.y = pixel slices, one fp32 atomic per thread
__global__ void conv_f32_wgrad_kernel(C32 a, const float* dy, float* dw, int per_slice) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long tot = (long long)a.Co * a.k * a.k * a.Ci;
  if (i >= tot) return;
  int ci = (int)(i % a.Ci);
  int t = (int)((i / a.Ci) % (a.k * a.k));
  int co = (int)(i / ((long long)a.Ci * a.k * a.k));
  int kh = t / a.k, kw = t % a.k;
  const int pad = a.k / 2;
  long long P = (long long)a.N * a.Ho * a.Wo;
  long long p0 = (long long)blockIdx.y * per_slice, p1 = p0 + per_slice < P ? p0 + per_slice : P;
  float acc = 0.f;
  for (long long p = p0; p < p1; ++p) {
    int wo = (int)(p % a.Wo);
    int ho = (int)((p / a.Wo) % a.Ho);
    int n = (int)(p / ((long long)a.Wo * a.Ho));
    int h = ho * a.s + kh - pad, w = wo * a.s + kw - pad;
    if ((unsigned)h >= (unsigned)a.H || (unsigned)w >= (unsigned)a.W) continue;
    acc = fmaf(dy[p * a.Co + co], a.x[(((long long)n * a.H + h) * a.W + w) * a.Ci + ci], acc);
  }
  atomicAdd(dw + i, acc);
}

// per-channel sum / sum of squares: block = one 256-row slice of one channel group of 64 channels
__global__ void bn_stats_f32_kernel(const float* y, long long P, int C, float* stats) {
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int lane_row = threadIdx.x >> 6;                  // 4 rows in parallel
  long long p0 = (long long)blockIdx.y * 1024, p1 = p0 + 1024 < P ? p0 + 1024 : P;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (long long p = p0 + lane_row; p < p1; p += 4) {
      float v = y[p * C + c];
      s1 += v; s2 = fmaf(v, v, s2);
    }
  if (c < C) { atomicAdd(stats + c, s1); atomicAdd(stats + C + c, s2); }
}

__global__ void bn_act_fwd_f32_kernel(const float* y, const float* scale, const float* shift, const float* res, float* a,
                                      long long n, int C, float slope) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = (int)(i % C);
  float z = fmaf(y[i], scale[c], shift[c]);
  float v = z > 0.f ? z : z * slope;
  a[i] = res ? v + res[i] : v;
}

// sums[0..C) = sum dz, sums[C..2C) = sum dz * yhat,  dz = da * leaky'(y*scale+shift), yhat = (y - mean) * invstd
__global__ void bn_act_bwd_reduce_f32_kernel(const float* da, const float* y, const float* scale, const float* shift,
                                             const float* mean, const float* invstd, long long P, int C, float slope,
                                             float* sums) {
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int lane_row = threadIdx.x >> 6;
  long long p0 = (long long)blockIdx.y * 1024, p1 = p0 + 1024 < P ? p0 + 1024 : P;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    float sc = scale[c], sh = shift[c], mu = mean[c], iv = invstd[c];
    for (long long p = p0 + lane_row; p < p1; p += 4) {
      float yv = y[p * C + c];
      float z = fmaf(yv, sc, sh);
      float dz = z > 0.f ? da[p * C + c] : da[p * C + c] * slope;
      s1 += dz; s2 = fmaf(dz, (yv - mu) * iv, s2);
    }
    atomicAdd(sums + c, s1); atomicAdd(sums + C + c, s2);
  }
}

// training: dy = scale * (dz - s1/P - yhat * s2/P); frozen: dy = scale * dz.  (dgamma = s2, dbeta = s1: written by thread c < C of block 0)
__global__ void bn_act_bwd_apply_f32_kernel(const float* da, const float* y, const float* scale, const float* shift,
                                            const float* mean, const float* invstd, const float* sums, float* dgamma,
                                            float* dbeta, float* dy, long long n, long long P, int C, float slope,
                                            int frozen) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!frozen && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) { dbeta[c] += sums[c]; dgamma[c] += sums[C + c]; }
  if (i >= n) return;
  int c = (int)(i % C);
  float yv = y[i];
  float z = fmaf(yv, scale[c], shift[c]);
  float dz = z > 0.f ? da[i] : da[i] * slope;
  if (frozen) { dy[i] = dz * scale[c]; return; }
  float yhat = (yv - mean[c]) * invstd[c];
  float inv = 1.0f / (float)P;
  dy[i] = scale[c] * (dz - sums[c] * inv - yhat * sums[C + c] * inv);
}

__global__ void upcat_fwd_f32_kernel(const float* u, const float* skip, float* out, int N, int h, int w, int Cu, int Cs) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int C = Cu + Cs;
  long long tot = (long long)N * 4 * h * w * C;
  if (i >= tot) return;
  int c = (int)(i % C);
  long long p = i / C;
  int x = (int)(p % (2 * w)); p /= 2 * w;
  int y = (int)(p % (2 * h));
  int n = (int)(p / (2 * h));
  out[i] = c < Cu ? u[(((long long)n * h + y / 2) * w + x / 2) * Cu + c]
                  : skip[(((long long)n * 2 * h + y) * 2 * w + x) * Cs + (c - Cu)];
}

__global__ void upcat_bwd_f32_kernel(const float* dout, float* du, float* dskip, int N, int h, int w, int Cu, int Cs) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int C = Cu + Cs;
  long long nu = (long long)N * h * w * Cu, ns = (long long)N * 4 * h * w * Cs;
  if (i < nu) {
    int c = (int)(i % Cu);
    long long p = i / Cu;
    int x = (int)(p % w); p /= w;
    int y = (int)(p % h);
    int n = (int)(p / h);
    float s = 0.f;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) s += dout[(((long long)n * 2 * h + 2 * y + dy) * 2 * w + 2 * x + dx) * C + c];
    du[i] = s;
  } else if (i < nu + ns) {
    long long j = i - nu;
    int c = (int)(j % Cs);
    long long p = j / Cs;
    dskip[j] = dout[p * C + Cu + c];
  }
}

__global__ void bias_grad_f32_kernel(const float* dy, float* db, long long P, int C) {
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int lane_row = threadIdx.x >> 6;
  long long p0 = (long long)blockIdx.y * 1024, p1 = p0 + 1024 < P ? p0 + 1024 : P;
  float s = 0.f;
  if (c < C) {
    for (long long p = p0 + lane_row; p < p1; p += 4) s += dy[p * C + c];
    atomicAdd(db + c, s);
  }
}

static C32 make(const float* x, const float* w, float* y, const float* bias, const float* addend, int N, int H, int W, int Ci,
                int Co, int k, int s) {
  C32 a;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.addend = addend;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.k = k; a.s = s;
  a.Ho = s == 2 ? H / 2 : H; a.Wo = s == 2 ? W / 2 : W;
  return a;
}

}  // namespace

#define MGD_F32_GEOM(name)                                                                                            \
  MGD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && Ci >= 1 && Co >= 1 && (k == 1 || k == 3) && (s == 1 || s == 2) &&        \
                  (s == 1 || (H % 2 == 0 && W % 2 == 0)),                                                             \
              name ": N=%d H=%d W=%d Ci=%d Co=%d k=%d s=%d", N, H, W, Ci, Co, k, s)

extern "C" int mgd_conv2d_f32_fwd(const float* x, const float* w, float* y, const float* bias, int N, int H, int W, int Ci,
                                  int Co, int k, int s, void* stream) {
  MGD_REQUIRE(x && w && y, "conv2d_f32_fwd: null pointer");
  MGD_F32_GEOM("conv2d_f32_fwd");
  C32 a = make(x, w, y, bias, nullptr, N, H, W, Ci, Co, k, s);
  long long tot = (long long)N * a.Ho * a.Wo * Co;
  hipLaunchKernelGGL(conv_f32_fwd_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, a);
  MGD_CHECK_LAUNCH("conv2d_f32_fwd");
  return MGD_OK;
}

extern "C" int mgd_conv2d_f32_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N, int H, int W,
                                    int Ci, int Co, int k, int s, void* stream) {
  MGD_REQUIRE(dy && w && dx, "conv2d_f32_dgrad: null pointer");
  MGD_F32_GEOM("conv2d_f32_dgrad");
  C32 a = make(dy, w, dx, nullptr, addend, N, H, W, Ci, Co, k, s);
  long long tot = (long long)N * H * W * Ci;
  hipLaunchKernelGGL(conv_f32_dgrad_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, a);
  MGD_CHECK_LAUNCH("conv2d_f32_dgrad");
  return MGD_OK;
}

extern "C" int mgd_conv2d_f32_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Ci, int Co, int k,
                                    int s, void* stream) {
  MGD_REQUIRE(x && dy && dw, "conv2d_f32_wgrad: null pointer");
  MGD_F32_GEOM("conv2d_f32_wgrad");
  C32 a = make(x, nullptr, nullptr, nullptr, nullptr, N, H, W, Ci, Co, k, s);
  long long tot = (long long)Co * k * k * Ci, P = (long long)N * a.Ho * a.Wo;
  int slices = (int)(P / 2048 > 0 ? (P / 2048 < 1024 ? P / 2048 : 1024) : 1);
  int per = cdiv(P, slices);
  hipLaunchKernelGGL(conv_f32_wgrad_kernel, dim3(cdiv(tot, 256), cdiv(P, per)), dim3(256), 0, (hipStream_t)stream, a, dy, dw,
                     per);
  MGD_CHECK_LAUNCH("conv2d_f32_wgrad");
  return MGD_OK;
}

extern "C" int mgd_bn_stats_f32(const float* y, int64_t P, int C, float* stats, void* stream) {
  MGD_REQUIRE(y && stats && P >= 1 && C >= 1, "bn_stats_f32: bad arguments");
  hipLaunchKernelGGL(bn_stats_f32_kernel, dim3(cdiv(C, 64), cdiv(P, 1024)), dim3(256), 0, (hipStream_t)stream, y, (long long)P,
                     C, stats);
  MGD_CHECK_LAUNCH("bn_stats_f32");
  return MGD_OK;
}

extern "C" int mgd_bn_act_fwd_f32(const float* y, const float* scale, const float* shift, const float* residual, float* a,
                                  int64_t P, int C, float slope, void* stream) {
  MGD_REQUIRE(y && scale && shift && a, "bn_act_fwd_f32: null pointer");
  long long n = (long long)P * C;
  hipLaunchKernelGGL(bn_act_fwd_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, scale, shift, residual, a,
                     n, C, slope);
  MGD_CHECK_LAUNCH("bn_act_fwd_f32");
  return MGD_OK;
}

extern "C" int mgd_bn_act_bwd_f32(const float* da, const float* y, const float* scale, const float* shift,
                                  const float* save_mean, const float* save_invstd, float* sums /*[2C], zeroed*/,
                                  float* dgamma, float* dbeta, float* dy, int64_t P, int C, float slope, int frozen,
                                  void* stream) {
  MGD_REQUIRE(da && y && scale && shift && dy && (frozen || (save_mean && save_invstd && sums && dgamma && dbeta)),
              "bn_act_bwd_f32: null pointer");
  long long n = (long long)P * C;
  if (!frozen)
    hipLaunchKernelGGL(bn_act_bwd_reduce_f32_kernel, dim3(cdiv(C, 64), cdiv(P, 1024)), dim3(256), 0, (hipStream_t)stream, da, y,
                       scale, shift, save_mean, save_invstd, (long long)P, C, slope, sums);
  hipLaunchKernelGGL(bn_act_bwd_apply_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, da, y, scale, shift,
                     save_mean, save_invstd, sums, dgamma, dbeta, dy, n, (long long)P, C, slope, frozen);
  MGD_CHECK_LAUNCH("bn_act_bwd_f32");
  return MGD_OK;
}

extern "C" int mgd_upsample_concat_fwd_f32(const float* u, const float* skip, float* out, int N, int h, int w, int Cu, int Cs,
                                           void* stream) {
  MGD_REQUIRE(u && skip && out, "upsample_concat_fwd_f32: null pointer");
  long long tot = (long long)N * 4 * h * w * (Cu + Cs);
  hipLaunchKernelGGL(upcat_fwd_f32_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, u, skip, out, N, h, w, Cu,
                     Cs);
  MGD_CHECK_LAUNCH("upsample_concat_fwd_f32");
  return MGD_OK;
}

extern "C" int mgd_upsample_concat_bwd_f32(const float* dout, float* du, float* dskip, int N, int h, int w, int Cu, int Cs,
                                           void* stream) {
  MGD_REQUIRE(dout && du && dskip, "upsample_concat_bwd_f32: null pointer");
  long long tot = (long long)N * h * w * Cu + (long long)N * 4 * h * w * Cs;
  hipLaunchKernelGGL(upcat_bwd_f32_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, dout, du, dskip, N, h, w,
                     Cu, Cs);
  MGD_CHECK_LAUNCH("upsample_concat_bwd_f32");
  return MGD_OK;
}

extern "C" int mgd_bias_grad_f32(const float* dy, float* dbias, int64_t P, int C, void* stream) {
  MGD_REQUIRE(dy && dbias, "bias_grad_f32: null pointer");
  hipLaunchKernelGGL(bias_grad_f32_kernel, dim3(cdiv(C, 64), cdiv(P, 1024)), dim3(256), 0, (hipStream_t)stream, dy, dbias,
                     (long long)P, C);
  MGD_CHECK_LAUNCH("bias_grad_f32");
  return MGD_OK;
}
