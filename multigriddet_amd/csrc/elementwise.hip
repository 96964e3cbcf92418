// HBM-bound elementwise / reduction kernels around the conv engine (gfx950):
// training-mode BatchNorm finalize, BN+LeakyReLU(+residual) forward and backward, nearest-upsample +
// concat, bias gradient, dtype casts, Adam / SGD.  All activation traffic is 16-byte vectorised bf16.
// Replaces Keras BatchNormalization/LeakyReLU/Add/UpSampling2D/Concatenate and their gradients
// (reference multigriddet/models/layers.py:94-95, models/backbones/darknet.py:39,
// models/heads/multigrid_head.py:296-298) and keras.optimizers.Adam (config/model_builder.py:86-96).
#include "common.h"

namespace {

// 16-byte load with the non-temporal hint for operands the BatchNorm passes stream once (they are dead afterwards: no reason to
// keep their lines in L2 / the memory-side cache beside the tensors the next kernel reads; 11.70 -> 11.63 ms per step, same box.
// The same hint on the optimiser's 1.2 GB per step measured nothing.)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// Sum of the R replicas of a per-channel pair (stats[r][0][c], stats[r][1][c]), replicas in ascending order (same
// rounding as the plain loop).  The loads of eight replicas are issued before the first add: written as a plain
// `for r: s += stats[r]` loop the compiler kept one load in flight at a time and the 16-replica fold cost ~8 us of
// pure latency in EVERY BatchNorm launch (the 9-10 us floor of the small layers in the rocprof trace).
__device__ __forceinline__ void fold_replicas(const float* base, int R, int C, int c, float& s0, float& s1) {
  s0 = 0.f;
  s1 = 0.f;
  for (int r0 = 0; r0 < R; r0 += 16) {        // 16 replicas (the engine's count) = 32 loads in flight, ONE round trip
    float a[16], b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const bool ok = r0 + j < R;
      a[j] = ok ? base[((long long)(r0 + j) * 2 + 0) * C + c] : 0.f;
      b[j] = ok ? base[((long long)(r0 + j) * 2 + 1) * C + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (r0 + j < R) { s0 += a[j]; s1 += b[j]; }
    }
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ stats, int R, int C, float count, const float* gamma,
                                   const float* beta, float* mm, float* mv, float* scale, float* shift, float* smean,
                                   float* sinv, float eps, float mom, int training) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    double s = 0.0, q = 0.0;
    for (int r = 0; r < R; ++r) {
      s += (double)stats[((long long)r * 2 + 0) * C + c];
      q += (double)stats[((long long)r * 2 + 1) * C + c];
    }
    double m = s / (double)count;
    double v = q / (double)count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (mm) mm[c] = mm[c] * mom + mean * (1.f - mom);
    if (mv) mv[c] = mv[c] * mom + var * (1.f - mom);
  } else {
    mean = mm[c];
    var = mv[c];
  }
  float inv = 1.0f / sqrtf(var + eps);
  float sc = gamma[c] * inv;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (smean) smean[c] = mean;
  if (sinv) sinv[c] = inv;
}

// a = leaky(y*scale+shift) (+res).  One thread = 8 channels of one pixel.
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const bf16_t* res,
                                                         bf16_t* __restrict__ a, long long nvec, int CV, float slope) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    int cv = (int)(i % CV);
    uint4 v = *(const uint4*)(y + i * 8);
    float f[8];
    unpack8(v, f);
    const float4* sp = (const float4*)(scale + cv * 8);
    const float4* hp = (const float4*)(shift + cv * 8);
    float4 s0 = sp[0], s1 = sp[1], h0 = hp[0], h1 = hp[1];
    float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float z = fmaf(f[j], sc[j], sh[j]);
      f[j] = z > 0.f ? z : z * slope;
    }
    if (res) {
      float g[8];
      unpack8(*(const uint4*)(res + i * 8), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j];
    }
    *(uint4*)(a + i * 8) = pack8(f);
  }
}


// Fused BatchNorm finalize + apply + LeakyReLU (+ residual): every block folds the R statistic replicas of
// its <= 256-channel window (thread t <-> channel), block x == 0 also publishes scale/shift/mean/invstd for
// the backward pass and updates the moving statistics.  Saves one launch per BN layer (66 per step).
__global__ __launch_bounds__(256) void bn_act_fwd_fused_kernel(const float* __restrict__ stats, int R, float count,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* mm, float* mv,
                                                               float* scale, float* shift, float* smean, float* sinv,
                                                               float eps, float mom, int training,
                                                               const bf16_t* __restrict__ y, const bf16_t* res,
                                                               bf16_t* __restrict__ a, long long P, int C, float slope) {
  __shared__ float prm[2][256];
  const int CV = C >> 3;
  const int CVB = CV < 32 ? CV : 32;
  const int PL = 256 / CVB;
  const int oct = blockIdx.y * 32 + (threadIdx.x % CVB);
  const int pl = threadIdx.x / CVB;
  const bool active = oct < CV && pl < PL;
  // The first tile of activations is requested BEFORE the statistics are folded: the parameter phase is a chain of
  // dependent round trips (replicas -> barrier -> parameters) that the data loads do not depend on, and a small layer
  // is one tile per thread - so its launch costs one memory round trip instead of three (it matters most inside the
  // backward pass, where the weight-gradient stream multiplies the latency of every round trip).
  // A block owns CONTIGUOUS chunks of PL * U pixels (consecutive blocks sweep memory in address order, like a copy).
  constexpr int U = 4;
  const long long stride = PL;
  const long long pstart = (long long)blockIdx.x * PL * U + pl;
  const long long pstep = (long long)gridDim.x * PL * U;
  uint4 yv[U], rv[U];
  auto fetch = [&](long long p0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long p = p0 + u * stride;
      if (p < P) {
        long long e = p * C + oct * 8;
        yv[u] = nt_load16(y + e);     // y is dead after this pass (until the backward pass)
        if (res) rv[u] = *(const uint4*)(res + e);
      }
    }
  };
  if (active && pstart < P) fetch(pstart);
  {
    int c = blockIdx.y * 256 + threadIdx.x;
    float sc = 0.f, sh = 0.f;
    if (c < C && threadIdx.x < CVB * 8) {
      float mean, var;
      if (training) {
        float s, q;
        fold_replicas(stats, R, C, c, s, q);
        mean = s / count;
        var = fmaxf(q / count - mean * mean, 0.f);
      } else {
        mean = mm[c];
        var = mv[c];
      }
      float inv = 1.0f / sqrtf(var + eps);
      sc = gamma[c] * inv;
      sh = beta[c] - mean * sc;
      if (blockIdx.x == 0) {
        scale[c] = sc; shift[c] = sh;
        if (smean) smean[c] = mean;
        if (sinv) sinv[c] = inv;
        if (training) {
          if (mm) mm[c] = mm[c] * mom + mean * (1.f - mom);
          if (mv) mv[c] = mv[c] * mom + var * (1.f - mom);
        }
      }
    }
    prm[0][threadIdx.x] = sc;
    prm[1][threadIdx.x] = sh;
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
  {
    int lo = (threadIdx.x % CVB) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = prm[0][lo + j]; sh[j] = prm[1][lo + j]; }
  }
  for (long long p0 = pstart; p0 < P; p0 += pstep) {
    if (p0 != pstart) fetch(p0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long p = p0 + u * stride;
      if (p < P) {
        float f[8];
        unpack8(yv[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = fmaf(f[j], sc[j], sh[j]);
          f[j] = z > 0.f ? z : z * slope;
        }
        if (res) {
          float g[8];
          unpack8(rv[u], g);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += g[j];
        }
        *(uint4*)(a + p * C + oct * 8) = pack8(f);
      }
    }
  }
}

// BN + LeakyReLU backward, two passes over (da, y):
//   REDUCE: per-channel sums of dyh = da * leaky'(z) and dyh * yhat  -> sums[R][2][C] (atomics, R replicas)
//   APPLY : dy = scale * (dyh - mean(dyh) - yhat * mean(dyh * yhat)); block 0 of each channel window also
//           folds the replicas into dgamma / dbeta (what used to be a separate finalize launch).
// Thread = one channel octet x one pixel lane; a block covers a window of <= 32 octets (256 channels).
// Per-channel parameters are loaded once per thread (two float4 per array); the grid is sized so that a
// thread owns >= 8 pixels, otherwise those parameter loads dominate small tensors (measured: ~45 us floor).
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <bool APPLY>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ smean,
                                                         const float* __restrict__ sinv, float* sums, int R,
                                                         float* dgamma, float* dbeta, bf16_t* __restrict__ dy,
                                                         long long P, int C, float slope, int frozen) {
  __shared__ float red[2][256];
  const int CV = C >> 3;
  const int CVB = CV < 32 ? CV : 32;
  const int PL = 256 / CVB;
  const int oct = blockIdx.y * 32 + (threadIdx.x % CVB);
  const int pl = threadIdx.x / CVB;
  const bool active = oct < CV && pl < PL;
  // first tile and per-channel parameters requested before the replica fold (see bn_act_fwd_fused_kernel)
  constexpr int U = 4;                        // pixels in flight per thread: 8 x 16-byte loads before any use
  const long long stride = PL;                // contiguous chunks of PL * U pixels per block (see bn_act_fwd_fused_kernel)
  const long long pstart = (long long)blockIdx.x * PL * U + pl;
  const long long pstep = (long long)gridDim.x * PL * U;
  uint4 gv[U], yv[U];
  auto fetch = [&](long long p0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long p = p0 + u * stride;
      if (p < P) {
        long long e = p * C + oct * 8;
        gv[u] = nt_load16(da + e);    // both operands are dead after this pass
        yv[u] = nt_load16(y + e);
      }
    }
  };
  float sc[8], sh[8], mu[8], iv[8], pa[8], pb[8];
  if (active && pstart < P) fetch(pstart);
  if (APPLY) {
    // per-channel coefficients, computed once per block (thread t <-> channel blockIdx.y*256 + t) and shared through
    // LDS:  dy = g * (z > 0 ? sc : sc*slope) - (pa + pb * y),  z = y*sc + sh,
    //       pb = sc * m2 * invstd,  pa = sc * m1 - pb * mean   (m1, m2 = the two batch means; 0 when frozen)
    __shared__ float prm[4][256];
    int c = blockIdx.y * 256 + threadIdx.x;
    float csc = 0.f, csh = 0.f, cpa = 0.f, cpb = 0.f;
    if (c < C && threadIdx.x < CVB * 8) {
      csc = scale[c];
      csh = shift[c];
      if (!frozen) {
        const float cmu = smean[c], civ = sinv[c];
        float sa = 0.f, sb = 0.f;
        fold_replicas(sums, R, C, c, sa, sb);
        if (blockIdx.x == 0) {
          if (dbeta) dbeta[c] += sa;
          if (dgamma) dgamma[c] += sb;
        }
        const float invP = 1.0f / (float)P;
        cpb = csc * (sb * invP) * civ;
        cpa = csc * (sa * invP) - cpb * cmu;
      }
    }
    prm[0][threadIdx.x] = csc; prm[1][threadIdx.x] = csh; prm[2][threadIdx.x] = cpa; prm[3][threadIdx.x] = cpb;
    __syncthreads();
    if (active) {
      int lo = (threadIdx.x % CVB) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = prm[0][lo + j]; sh[j] = prm[1][lo + j]; pa[j] = prm[2][lo + j]; pb[j] = prm[3][lo + j]; }
    }
  } else if (active) {
    load8(scale + oct * 8, sc);
    load8(shift + oct * 8, sh);
    load8(smean + oct * 8, mu);
    load8(sinv + oct * 8, iv);
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (active) {
    for (long long p0 = pstart; p0 < P; p0 += pstep) {
      if (p0 != pstart) fetch(p0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long p = p0 + u * stride;
        if (p < P) {
          long long e = p * C + oct * 8;
          float g[8], v[8], o[8];
          unpack8(gv[u], g);
          unpack8(yv[u], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = fmaf(v[j], sc[j], sh[j]);
            float d = z > 0.f ? g[j] : g[j] * slope;
            if (APPLY) {
              o[j] = fmaf(sc[j], d, -fmaf(pb[j], v[j], pa[j]));
            } else {
              float yh = (v[j] - mu[j]) * iv[j];
              s1[j] += d;
              s2[j] += d * yh;
            }
          }
          if (APPLY) *(uint4*)(dy + e) = pack8(o);
        }
      }
    }
  }
  if (!APPLY) {
    red[0][threadIdx.x] = 0.f;
    red[1][threadIdx.x] = 0.f;
    __syncthreads();
    if (active) {
      int lo = (threadIdx.x % CVB) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&red[0][lo + j], s1[j]);
        atomicAdd(&red[1][lo + j], s2[j]);
      }
    }
    __syncthreads();
    int c = blockIdx.y * 256 + threadIdx.x;
    if (threadIdx.x < CVB * 8 && c < C) {
      int rep = blockIdx.x % R;
      atomicAdd(sums + ((long long)rep * 2 + 0) * C + c, red[0][threadIdx.x]);
      atomicAdd(sums + ((long long)rep * 2 + 1) * C + c, red[1][threadIdx.x]);
    }
  }
}

__global__ __launch_bounds__(256) void upcat_fwd_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ s,
                                                        bf16_t* __restrict__ out, int N, int h, int w, int Cu, int Cs) {
  const int Ct = Cu + Cs, CV = Ct >> 3, H = 2 * h, W = 2 * w;
  long long nvec = (long long)N * H * W * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    int cv = (int)(i % CV);
    long long p = i / CV;
    int x = (int)(p % W);
    long long t = p / W;
    int yy = (int)(t % H);
    int n = (int)(t / H);
    int c = cv * 8;
    uint4 v;
    if (c < Cu) v = *(const uint4*)(u + (((long long)n * h + (yy >> 1)) * w + (x >> 1)) * Cu + c);
    else v = *(const uint4*)(s + p * Cs + (c - Cu));
    *(uint4*)(out + p * Ct + c) = v;
  }
}

__global__ __launch_bounds__(256) void upcat_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ du,
                                                        bf16_t* __restrict__ ds, int N, int h, int w, int Cu, int Cs) {
  const int Ct = Cu + Cs, H = 2 * h, W = 2 * w;
  const int CVu = Cu >> 3, CVs = Cs >> 3;
  long long nu = (long long)N * h * w * CVu, ns = (long long)N * H * W * CVs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nu + ns; i += (long long)gridDim.x * blockDim.x) {
    if (i < nu) {
      int cv = (int)(i % CVu);
      long long p = i / CVu;
      int x = (int)(p % w);
      long long t = p / w;
      int yy = (int)(t % h);
      int n = (int)(t / h);
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float f[8];
          unpack8(*(const uint4*)(dout + (((long long)n * H + 2 * yy + dy) * W + 2 * x + dx) * Ct + cv * 8), f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
      *(uint4*)(du + p * Cu + cv * 8) = pack8(acc);
    } else {
      long long k = i - nu;
      int cv = (int)(k % CVs);
      long long p = k / CVs;
      *(uint4*)(ds + p * Cs + cv * 8) = *(const uint4*)(dout + p * Ct + Cu + cv * 8);
    }
  }
}

__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ dy, float* dbias, long long P, int C) {
  // thread = channel octet x pixel lane (same mapping idea as bn bwd); C <= 256
  __shared__ float red[256];
  red[threadIdx.x] = 0.f;
  __syncthreads();
  const int CV = C >> 3;
  const int PL = 256 / CV;
  const int oct = threadIdx.x % CV, pl = threadIdx.x / CV;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < PL) {
    constexpr int U = 4;                       // loads in flight per thread (the loop is otherwise one dependent load)
    const long long stride = (long long)gridDim.x * PL;
    for (long long p0 = (long long)blockIdx.x * PL + pl; p0 < P; p0 += stride * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long p = p0 + u * stride;
        v[u] = p < P ? *(const uint4*)(dy + p * C + oct * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += f[j];
      }
    }
  }
  if (pl < PL)
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&red[oct * 8 + j], s[j]);
  __syncthreads();
  if (threadIdx.x < C) atomicAdd(dbias + threadIdx.x, red[threadIdx.x]);
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      float4 v = *(const float4*)(in + i);
      uint2 p;
      p.x = pack2bf(v.x, v.y);
      p.y = pack2bf(v.z, v.w);
      *(uint2*)(out + i) = p;
    } else {
      for (long long j = i; j < n; ++j) out[j] = f2bf(in[j]);
    }
  }
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = bf2f(in[i]);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr_t,
                                                   float b1, float b2, float eps, float gscale, float lr_wd,
                                                   const float* __restrict__ hyper) {
  if (hyper) { lr_t = hyper[0]; lr_wd = hyper[1]; }
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      float4 P = *(float4*)(p + i), G = *(const float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
      float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w},
            vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gr = gg[j] * gscale;
        mm[j] = b1 * mm[j] + (1.f - b1) * gr;
        vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
        pp[j] = pp[j] - lr_wd * pp[j] - lr_t * mm[j] / (sqrtf(vv[j]) + eps);
      }
      *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (long long j = i; j < n; ++j) {
        float gr = g[j] * gscale;
        float mj = b1 * m[j] + (1.f - b1) * gr, vj = b2 * v[j] + (1.f - b2) * gr * gr;
        m[j] = mj; v[j] = vj;
        p[j] = p[j] - lr_wd * p[j] - lr_t * mj / (sqrtf(vj) + eps);
      }
    }
  }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, long long n,
                           float lr, float mu, int nesterov, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gr = g[i] * gscale;
    float vel = mu * mom[i] - lr * gr;          // Keras SGD: v = mu*v - lr*g; p += v (nesterov: p += mu*v - lr*g)
    mom[i] = vel;
    p[i] += nesterov ? mu * vel - lr * gr : vel;
  }
}

inline int grid_for(long long n, int per_block) {
  long long b = (n + per_block - 1) / per_block;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int mgd_bn_finalize(const float* stats, int replicas, int C, float count, const float* gamma,
                               const float* beta, float* moving_mean, float* moving_var, float* scale, float* shift,
                               float* save_mean, float* save_invstd, float eps, float momentum, int training,
                               void* stream) {
  MGD_REQUIRE(gamma && beta && scale && shift, "bn_finalize: null pointer");
  MGD_REQUIRE(training ? (stats != nullptr && replicas >= 1 && count > 0) : (moving_mean && moving_var),
              "bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, stats, replicas, C,
                     count, gamma, beta, moving_mean, moving_var, scale, shift, save_mean, save_invstd, eps, momentum,
                     training);
  MGD_CHECK_LAUNCH("bn_finalize");
  return MGD_OK;
}

extern "C" int mgd_bn_act_fwd(const void* y, const float* scale, const float* shift, const void* residual, void* a,
                              int64_t P, int C, float slope, void* stream) {
  MGD_REQUIRE(y && scale && shift && a, "bn_act_fwd: null pointer");
  MGD_REQUIRE(C % 8 == 0, "bn_act_fwd: C=%d must be a multiple of 8", C);
  long long nvec = P * (C / 8);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, scale, shift, (const bf16_t*)residual, (bf16_t*)a, nvec, C / 8, slope);
  MGD_CHECK_LAUNCH("bn_act_fwd");
  return MGD_OK;
}

static void bn_bwd_grid(int64_t P, int C, int* gx, int* gy) {
  int CV = C / 8, CVB = CV < 32 ? CV : 32, PL = 256 / CVB;
  *gy = cdiv(CV, 32);
  // chunks of PL * 4 pixels (one memory round trip per thread); small tensors: one chunk per block, else every block the
  // same number (>= 2) of chunks with at most 8 blocks per CU
  const long long chunks = (P + (long long)PL * 4 - 1) / ((long long)PL * 4);
  const long long cap = 256 * 8 / *gy;
  long long g = chunks;
  if (chunks > cap / 2) {
    long long iters = (chunks + cap - 1) / cap;
    if (iters < 2) iters = 2;
    g = (chunks + iters - 1) / iters;
  }
  if (g < 1) g = 1;
  *gx = (int)g;
}

extern "C" int mgd_bn_act_bwd_reduce(const void* da, const void* y, const float* scale, const float* shift,
                                     const float* save_mean, const float* save_invstd, float* sums, int replicas,
                                     int64_t P, int C, float slope, void* stream) {
  MGD_REQUIRE(da && y && scale && shift && save_mean && save_invstd && sums, "bn_act_bwd_reduce: null pointer");
  MGD_REQUIRE(C % 8 == 0 && replicas >= 1, "bn_act_bwd_reduce: C=%d replicas=%d", C, replicas);
  int gx, gy;
  bn_bwd_grid(P, C, &gx, &gy);
  hipLaunchKernelGGL(bn_act_bwd_kernel<false>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)da, (const bf16_t*)y, scale, shift, save_mean, save_invstd, sums, replicas,
                     (float*)nullptr, (float*)nullptr, (bf16_t*)nullptr, (long long)P, C, slope, 0);
  MGD_CHECK_LAUNCH("bn_act_bwd_reduce");
  return MGD_OK;
}

extern "C" int mgd_bn_act_bwd_apply(const void* da, const void* y, const float* scale, const float* shift,
                                    const float* save_mean, const float* save_invstd, const float* sums, int replicas,
                                    float* dgamma, float* dbeta, void* dy, int64_t P, int C, float slope, int frozen,
                                    void* stream) {
  MGD_REQUIRE(da && y && scale && shift && save_mean && save_invstd && dy, "bn_act_bwd_apply: null pointer");
  MGD_REQUIRE(C % 8 == 0, "bn_act_bwd_apply: C=%d", C);
  MGD_REQUIRE(frozen || (sums && replicas >= 1), "bn_act_bwd_apply: sums required unless frozen");
  int gx, gy;
  bn_bwd_grid(P, C, &gx, &gy);
  hipLaunchKernelGGL(bn_act_bwd_kernel<true>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)da,
                     (const bf16_t*)y, scale, shift, save_mean, save_invstd, (float*)sums, replicas, dgamma, dbeta,
                     (bf16_t*)dy, (long long)P, C, slope, frozen);
  MGD_CHECK_LAUNCH("bn_act_bwd_apply");
  return MGD_OK;
}


extern "C" int mgd_bn_act_fwd_fused(const float* stats, int replicas, float count, const float* gamma,
                                    const float* beta, float* moving_mean, float* moving_var, float* scale,
                                    float* shift, float* save_mean, float* save_invstd, float eps, float momentum,
                                    int training, const void* y, const void* residual, void* a, int64_t P, int C,
                                    float slope, void* stream) {
  MGD_REQUIRE(gamma && beta && scale && shift && y && a, "bn_act_fwd_fused: null pointer");
  MGD_REQUIRE(C % 8 == 0, "bn_act_fwd_fused: C=%d must be a multiple of 8", C);
  MGD_REQUIRE(training ? (stats != nullptr && replicas >= 1 && count > 0) : (moving_mean && moving_var),
              "bn_act_fwd_fused: missing statistics");
  int gx, gy;
  bn_bwd_grid(P, C, &gx, &gy);
  hipLaunchKernelGGL(bn_act_fwd_fused_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, stats, replicas, count,
                     gamma, beta, moving_mean, moving_var, scale, shift, save_mean, save_invstd, eps, momentum, training,
                     (const bf16_t*)y, (const bf16_t*)residual, (bf16_t*)a, (long long)P, C, slope);
  MGD_CHECK_LAUNCH("bn_act_fwd_fused");
  return MGD_OK;
}

extern "C" int mgd_upsample_concat_fwd(const void* u, const void* skip, void* out, int N, int h, int w, int Cu, int Cs,
                                       void* stream) {
  MGD_REQUIRE(u && skip && out && Cu % 8 == 0 && Cs % 8 == 0, "upsample_concat_fwd: bad arguments");
  long long nvec = (long long)N * 4 * h * w * ((Cu + Cs) / 8);
  hipLaunchKernelGGL(upcat_fwd_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u,
                     (const bf16_t*)skip, (bf16_t*)out, N, h, w, Cu, Cs);
  MGD_CHECK_LAUNCH("upsample_concat_fwd");
  return MGD_OK;
}

extern "C" int mgd_upsample_concat_bwd(const void* dout, void* du, void* dskip, int N, int h, int w, int Cu, int Cs,
                                       void* stream) {
  MGD_REQUIRE(dout && du && dskip && Cu % 8 == 0 && Cs % 8 == 0, "upsample_concat_bwd: bad arguments");
  long long nvec = (long long)N * h * w * (Cu / 8) + (long long)N * 4 * h * w * (Cs / 8);
  hipLaunchKernelGGL(upcat_bwd_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dout, (bf16_t*)du, (bf16_t*)dskip, N, h, w, Cu, Cs);
  MGD_CHECK_LAUNCH("upsample_concat_bwd");
  return MGD_OK;
}

extern "C" int mgd_bias_grad(const void* dy, float* dbias, int64_t P, int C, void* stream) {
  MGD_REQUIRE(dy && dbias && C % 8 == 0 && C <= 256, "bias_grad: C=%d must be a multiple of 8 and <= 256", C);
  int PL = 256 / (C / 8);
  hipLaunchKernelGGL(bias_grad_kernel, dim3(grid_for(P, PL)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     dbias, (long long)P, C);
  MGD_CHECK_LAUNCH("bias_grad");
  return MGD_OK;
}

extern "C" int mgd_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  MGD_REQUIRE(in && out, "f32_to_bf16: null pointer");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out,
                     (long long)n);
  MGD_CHECK_LAUNCH("f32_to_bf16");
  return MGD_OK;
}
extern "C" int mgd_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
  MGD_REQUIRE(in && out, "bf16_to_f32: null pointer");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                     out, (long long)n);
  MGD_CHECK_LAUNCH("bf16_to_f32");
  return MGD_OK;
}

extern "C" int mgd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale, float weight_decay, void* stream) {
  MGD_REQUIRE(p && g && m && v && step >= 1, "adam: bad arguments");
  double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                     (float)lr_t, beta1, beta2, eps, grad_scale, lr * weight_decay, (const float*)nullptr);
  MGD_CHECK_LAUNCH("adam");
  return MGD_OK;
}

extern "C" int mgd_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                 float beta1, float beta2, float eps, float grad_scale, void* stream) {
  MGD_REQUIRE(p && g && m && v && hyper, "adam_dev: null pointer");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                     0.f, beta1, beta2, eps, grad_scale, 0.f, hyper);
  MGD_CHECK_LAUNCH("adam_dev");
  return MGD_OK;
}

extern "C" int mgd_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, int nesterov,
                            float grad_scale, void* stream) {
  MGD_REQUIRE(p && g && mom, "sgd: null pointer");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, mom, (long long)n,
                     lr, momentum, nesterov, grad_scale);
  MGD_CHECK_LAUNCH("sgd");
  return MGD_OK;
}
