// Stem convolution (3 -> 32 channels) and the weight-packing kernels of the convolution engine (gfx950).
#include "conv_common.hpp"

namespace {

__global__ __launch_bounds__(256) MGD_VGPR_MFMA void stem_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       bf16_t* __restrict__ y, float* stats, int reps, int N, int H,
                                                       int W, const float* __restrict__ bias, float act_slope) {
  constexpr int TH = 4, TW = 64, PR = TH + 2, PCF = (TW + 2) * 3;      // patch rows, floats per patch row
  __shared__ float patch[PR * PCF];
  __shared__ float red[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
  // A fragments: lane (fr = row of the 16-row tile, fq = k group) holds w[chan(m, fr)][fq*8 .. +7].  The rows are
  // permuted, chan(m, row) = (row / 4) * 8 + m * 4 + row % 4, so that the accumulators of a lane (rows fq*4 .. +3 of
  // both tiles) are the eight consecutive channels fq*8 .. +7: one 16-byte store per pixel and lane, a wave writes
  // 16 pixels x 64 B contiguously.
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 wf[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float t[8];
    const int co = (fr >> 2) * 8 + m * 4 + (fr & 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int k = fq * 8 + i;
      t[i] = k < 27 ? w[co * 27 + k] : 0.f;
    }
    wf[m] = __builtin_bit_cast(bf16x8, pack8(t));
  }
  // B fragment gather offsets (patch floats relative to the pixel's row start): k -> (k / 9) rows down, (k % 9) floats right
  int koff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int k = fq * 8 + i;
    koff[i] = k < 27 ? (k / 9) * PCF + (k % 9) : -1;
  }
  // folded inference (mgd_stem_fwd_act): the BatchNorm shift of this lane's eight channels, LeakyReLU on (acc + shift)
  float bsh[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) bsh[m][r] = bias ? bias[fq * 8 + m * 4 + r] : 0.f;
  float s1[2][4], s2[2][4];                        // BatchNorm statistics, carried over all tiles of the block
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = 0.f;
  const int ntiles = N * tilesH * tilesW;
  // the next tile's patch is fetched into registers while the current one is computed and stored
  constexpr int NPV = (PR * PCF + 255) / 256;
  float pv[NPV];
  auto fetch = [&](int tile) {
    int b = tile;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      int i = tid + u * 256;
      int pr = i / PCF, off = i - pr * PCF;
      int hh = th * TH - 1 + pr;
      int col = (tw * TW - 1) * 3 + off;            // float index inside the image row
      pv[u] = 0.f;
      if (i < PR * PCF && (unsigned)hh < (unsigned)H && (unsigned)col < (unsigned)(W * 3))
        pv[u] = img[((long long)n * H + hh) * W * 3 + col];
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int b = tile;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH;
  const int n = b / tilesH;
  const int h0 = th * TH, w0 = tw * TW;
  __syncthreads();                                  // previous tile's patch reads are done
#pragma unroll
  for (int u = 0; u < NPV; ++u) {
    int i = tid + u * 256;
    if (i < PR * PCF) patch[i] = pv[u];
  }
  __syncthreads();
  if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
  const int hy = h0 + wave;
  if (hy < H) {
#pragma unroll
    for (int j = 0; j < TW / 16; ++j) {
      const int px = j * 16 + fr;
      const float* base = patch + wave * PCF + px * 3;
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = koff[i] >= 0 ? base[koff[i]] : 0.f;
      bf16x8 xf = __builtin_bit_cast(bf16x8, pack8(t));
      const bool ok = w0 + px < W;
      bf16_t* yr = y + (((long long)n * H + hy) * W + w0 + px) * 32;
      uint4 st;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = acc[r] + bsh[m][r];
            acc[r] = z > 0.f ? z : z * act_slope;
          }
        }
        uint2 pk;
        pk.x = pack2bf(acc[0], acc[1]);
        pk.y = pack2bf(acc[2], acc[3]);
        if (m == 0) { st.x = pk.x; st.y = pk.y; } else { st.z = pk.x; st.w = pk.y; }
        if (ok) {
          float v0 = __uint_as_float(pk.x << 16), v1 = __uint_as_float(pk.x & 0xffff0000u);
          float v2 = __uint_as_float(pk.y << 16), v3 = __uint_as_float(pk.y & 0xffff0000u);
          s1[m][0] += v0; s1[m][1] += v1; s1[m][2] += v2; s1[m][3] += v3;
          s2[m][0] = fmaf(v0, v0, s2[m][0]); s2[m][1] = fmaf(v1, v1, s2[m][1]);
          s2[m][2] = fmaf(v2, v2, s2[m][2]); s2[m][3] = fmaf(v3, v3, s2[m][3]);
        }
      }
      if (ok) *(uint4*)(yr + fq * 8) = st;
    }
  }
  }   // tiles
  if (stats) {
    // lanes with equal fq hold the same channels: fold the 16 pixel lanes, one partial row per wave, then per block
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x1 = s1[m][r], x2 = s2[m][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
        if (fr == 0) {
          red[wave][0][fq * 8 + m * 4 + r] = x1;
          red[wave][1][fq * 8 + m * 4 + r] = x2;
        }
      }
    __syncthreads();
    if (tid < 64) {
      int which = tid >> 5, c = tid & 31;
      float t = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
      int rep = blockIdx.x % reps;
      atomicAdd(stats + ((long long)rep * 2 + which) * 32 + c, t);
    }
  }
}

// dW[co][tap][c] = sum_p dy[p][co] * img[p (+) tap][c] ; 864 outputs reduced over all pixels.
// thread = (channel co = tid & 31, pixel lane = tid >> 5): 27 private accumulators.
// Stem weight gradient on the matrix cores: dW[co][k] += sum_p dy[p][co] * x[p (+) tap][c], k = tap*3 + c (27 of 32).
// A block walks tiles of 4 rows x 64 pixels; per tile it stages the haloed fp32 image patch and the bf16 dy tile in
// LDS, each wave takes one row (two 32-pixel K-steps): the A fragments (dy transposed) come from ds_read_b64_tr_b16,
// the B fragments (the im2col matrix transposed, never materialised) are eight patch words per lane at stride 3.
// Accumulators stay in registers over all tiles of the block; one 32x27 fp32 atomic flush per block.  Image rounded to
// bf16 like the forward pass.
// FUSE_BN: `dy` is not materialised - the block reads da (gradient wrt the activated stem output) and y (raw stem
// output) and applies the BatchNorm + LeakyReLU backward on the way into LDS,
//   dy = scale * (dyh - mean(dyh) - yhat * mean(dyh * yhat)),  dyh = da * leaky'(y*scale+shift),  yhat = (y-mu)*invstd,
// with the two means from the replicated sums the producer of da left (fused reduction); block 0 also adds
// dbeta / dgamma.  Saves the stem's BatchNorm-backward pass: 756 MB read + 378 MB written at 608 x 608, batch 16.
struct StemBn {
  const bf16_t* y;
  const float *scale, *shift, *mean, *invstd, *sums;
  float *dgamma, *dbeta;
  int R;
  float slope;
};

template <bool FUSE_BN>
__global__ __launch_bounds__(256) MGD_VGPR_MFMA void stem_wgrad_kernel(const float* __restrict__ img, const bf16_t* __restrict__ dy,
                                                         float* dw, int N, int H, int W, StemBn bn) {
  constexpr int TH = 4, TW = 64, PR = TH + 2, PCF = (TW + 2) * 3;
  __shared__ float patch[PR * PCF];
  __shared__ __attribute__((aligned(16))) unsigned char dyt[TH * TW * 64];     // [row][pixel][32 ch] bf16, tr-swizzled
  __shared__ float red[32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
  const int ntiles = N * tilesH * tilesW;
  const int fr = lane & 15, fq = lane >> 4;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pl = lane & 3;
  // B gather: lane = column k (per 16-wide tile nt) and pixel group fq*8.. ; offsets of the k-th im2col column in the patch
  int koff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    int k = nt * 16 + fr;
    koff[nt] = k < 27 ? (k / 9) * PCF + (k % 9) : -1;
  }
  // A (dy^T) transposed reads: MFMA k index = pixel kk*32 + 8g + qq (+4 for the upper half), channel group m
  int o_rd[2][2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int r0 = kk * 32 + 8 * g + qq + 4 * h;
#pragma unroll
      for (int m = 0; m < 2; ++m) o_rd[kk][h][m] = (wave * TW + r0) * 64 + ((m ^ tr_swz(r0, 2)) * 32) + pl * 8;
    }
  f32x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const unsigned dyt_a = lds_addr(dyt);
  // fused BN backward: this thread always stages the same 8 channels (chunk tid & 3)
  float bsc[8], bsh[8], bmu[8], biv[8], bm1[8], bm2[8];
  // software pipeline over the block's tiles: the next tile's global loads (image patch, dy or da + y) are issued into
  // registers right after the current tile is published to LDS and land while the matrix cores work on it
  constexpr int NPV = (PR * PCF + 255) / 256, NDV = TH * TW * 4 / 256;
  float pv[NPV];
  uint4 gv[NDV], yv[NDV];
  unsigned inside = 0;                                // bit u: chunk u of the fetched tile is a real pixel
  auto fetch = [&](int t) {
    int b = t;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
    const int h0 = th * TH, w0 = tw * TW;
    inside = 0;
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      int i = tid + u * 256;
      int pr = i / PCF, off = i - pr * PCF;
      int hh = h0 - 1 + pr;
      int col = (w0 - 1) * 3 + off;
      pv[u] = 0.f;
      if (i < PR * PCF && (unsigned)hh < (unsigned)H && (unsigned)col < (unsigned)(W * 3))
        pv[u] = img[((long long)n * H + hh) * W * 3 + col];
    }
#pragma unroll
    for (int u = 0; u < NDV; ++u) {                   // 16-byte chunks of the dy tile
      int i = tid + u * 256;
      int ch = i & 3, px = (i >> 2) % TW, r = i / (4 * TW);
      gv[u] = make_uint4(0, 0, 0, 0);
      if (FUSE_BN) yv[u] = make_uint4(0, 0, 0, 0);
      if (h0 + r < H && w0 + px < W) {
        const long long e = (((long long)n * H + h0 + r) * W + w0 + px) * 32 + ch * 8;
        gv[u] = *(const uint4*)(dy + e);
        if (FUSE_BN) yv[u] = *(const uint4*)(bn.y + e);
        inside |= 1u << u;
      }
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  if (FUSE_BN) {      // after the first tile's loads are in flight: the fold is a chain of dependent round trips
    const int c0 = (tid & 3) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bsc[j] = bn.scale[c0 + j]; bsh[j] = bn.shift[c0 + j]; bmu[j] = bn.mean[c0 + j]; biv[j] = bn.invstd[c0 + j];
    }
    if (tid < 64) {                                   // fold the R replicas: red[0..31] = sum dyh, red[32..63] = sum dyh*yhat
      int which = tid >> 5, c = tid & 31;
      float t = 0.f;
      for (int r0 = 0; r0 < bn.R; r0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = r0 + j < bn.R ? bn.sums[((long long)(r0 + j) * 2 + which) * 32 + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
      }
      red[tid] = t;
      if (blockIdx.x == 0) {
        if (which == 0 && bn.dbeta) bn.dbeta[c] += t;
        if (which == 1 && bn.dgamma) bn.dgamma[c] += t;
      }
    }
    __syncthreads();
    const float invP = 1.0f / ((float)N * (float)H * (float)W);
#pragma unroll
    for (int j = 0; j < 8; ++j) { bm1[j] = red[c0 + j] * invP; bm2[j] = red[32 + c0 + j] * invP; }
  }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();                                  // previous tile's reads are done
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      int i = tid + u * 256;
      if (i < PR * PCF) patch[i] = pv[u];
    }
#pragma unroll
    for (int u = 0; u < NDV; ++u) {
      int i = tid + u * 256;
      int ch = i & 3, px = (i >> 2) % TW, r = i / (4 * TW);
      uint4 v = gv[u];
      if (FUSE_BN) {
        float gq[8], yq[8], o[8];
        unpack8(v, gq);
        unpack8(yv[u], yq);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = fmaf(yq[j], bsc[j], bsh[j]);
          float dd = z > 0.f ? gq[j] : gq[j] * bn.slope;
          float yh = (yq[j] - bmu[j]) * biv[j];
          o[j] = bsc[j] * (dd - bm1[j] - yh * bm2[j]);
        }
        v = pack8(o);
        if (!(inside >> u & 1)) v = make_uint4(0, 0, 0, 0);   // pixels past the image edge still meet real image taps
      }
      int slot = ((((ch >> 1) ^ tr_swz(px, 2)) << 1) | (ch & 1));
      *(uint4*)(dyt + (r * TW + px) * 64 + slot * 16) = v;
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x4 fa[2][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        tr_read_asm<0>(fa[m][0], dyt_a + o_rd[kk][0][m]);
        tr_read_asm<0>(fa[m][1], dyt_a + o_rd[kk][1][m]);
      }
      bf16x8 xf[2];
      const float* base = patch + wave * PCF + (kk * 32 + fq * 8) * 3;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float tv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tv[i] = koff[nt] >= 0 ? base[koff[nt] + i * 3] : 0.f;
        xf[nt] = __builtin_bit_cast(bf16x8, pack8(tv));
      }
      wait_lgkm_dyn(0);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        touch(fa[m][0]); touch(fa[m][1]);
        s16x8 av = __builtin_shufflevector(fa[m][0], fa[m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), xf[nt], acc[m][nt], 0, 0, 0);
      }
    }
  }
  // fold the four waves, then one atomic per (co, k)
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 256) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[(m * 16 + fq * 4 + r) * 32 + nt * 16 + fr], acc[m][nt][r]);
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 256) {
    int co = i >> 5, k = i & 31;
    if (k < 27) atomicAdd(dw + co * 27 + k, red[i]);
  }
}


// Stem as a GEMM: im2col of the fp32 image into bf16 [P][32] (k = (kh*3+kw)*3 + c for k < 27, zeros after),
// so that the 3->32 conv and its weight gradient run on the MFMA kernels (1x1, Ci = 32) instead of the
// direct VALU kernels above (0.6 + 1.5 ms per step at 608^2 x 16).  One thread = one 8-wide k octet.
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int N,
                                                          int H, int W) {
  const long long nvec = (long long)N * H * W * 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    int oct = (int)(i & 3);
    long long p = i >> 2;
    int wx = (int)(p % W);
    long long t = p / W;
    int hy = (int)(t % H);
    int n = (int)(t / H);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int k = oct * 8 + j;
      float v = 0.f;
      if (k < 27) {
        int tap = k / 3, c = k - tap * 3;
        int hh = hy + tap / 3 - 1, ww = wx + tap % 3 - 1;
        if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = img[(((long long)n * H + hh) * W + ww) * 3 + c];
      }
      f[j] = v;
    }
    *(uint4*)(out + i * 8) = pack8(f);
  }
}

struct PackJob {
  const float* w;
  bf16_t* out;
  int Co, T, Ci, transpose, ntaps_out, rows_pad, K_pad, pad_;
  unsigned long long srccode;
  long long begin;   // first flat element index of this job
};

// Layout of a packed image [rows_pad][K_pad]:
//  * rows_pad % 128 != 0 (64- and 32-row tiles, patch kernels): row-major;
//  * rows_pad % 128 == 0: MFMA-FRAGMENT ORDER.  Per (128-row tile cot, 64-deep K-step ks) one 16-KiB block of 1024 16-byte
//    chunks, chunk ((wcm*2 + kk)*64 + lane) = row cot*128 + wcm*16 + (lane & 15), columns ks*64 + (kk*4 + (lane >> 4))*8 .. +8
//    (wcm = 16-row group 0..7, kk = 32-deep half).  A wave's A operand of a K-step is then plain coalesced 16-byte global
//    loads (conv_gemm8_kernel) or, staged through LDS (conv_gemm6_kernel), a linear copy read back conflict-free.
__device__ __forceinline__ long long packed_elem(int r, int k, int rows_pad, int K_pad) {
  if (rows_pad & 127) return (long long)r * K_pad + k;
  const int nk = K_pad >> 6;
  const int cot = r >> 7, wcm = (r >> 4) & 7, fr = r & 15;
  const int ks = k >> 6, kk = (k >> 5) & 1, fq = (k >> 3) & 3;
  return ((((long long)cot * nk + ks) * 16 + wcm * 2 + kk) * 64 + fq * 16 + fr) * 8 + (k & 7);
}

// One block iteration = one 32-row x 64-column tile of one packed image (output rows r0.., columns t*cin + c0..); the
// tile is read along the source's contiguous index (ci for forward images, the OUTPUT-row index for transposed /
// data-gradient images) and goes through LDS, so both sides are coalesced: 128-byte source rows, and on the output side
// 128-byte bf16 rows (row-major images, two columns per thread) or, in fragment order, one 16-byte chunk per thread with
// 16 consecutive rows = 256 contiguous bytes per 16 lanes.  Only the valid region is written: the zero padding of the
// images is written once at allocation and never changes.  `begin` of a job = index of its first tile.
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs, long long total) {
  __shared__ float tile[64][33];
  for (long long bt = blockIdx.x; bt < total; bt += gridDim.x) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].begin <= bt) lo = mid; else hi = mid - 1;
    }
    const PackJob J = jobs[lo];
    const int rows = J.transpose ? J.Ci : J.Co, cin = J.transpose ? J.Co : J.Ci;
    const int tr = (rows + 31) >> 5, tcn = (cin + 63) >> 6;
    int e = (int)(bt - J.begin);
    const int rt = e % tr; e /= tr;
    const int ct = e % tcn;
    const int t = e / tcn;
    const int st = (int)((J.srccode >> (4 * t)) & 15);
    const int r0 = rt * 32, c0 = ct * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool even = !((J.Ci | J.Co) & 1) && !(((uintptr_t)J.w) & 7);   // paired 4-byte stores / 8-byte loads are aligned
    const bool frag = !(J.rows_pad & 127);
    if (J.transpose) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int c = c0 + ty + 8 * j, r = r0 + tx;
        tile[ty + 8 * j][tx] = (c < cin && r < rows) ? J.w[((long long)c * J.T + st) * J.Ci + r] : 0.f;
      }
      __syncthreads();
    } else if (frag) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int r = r0 + ty + 8 * j, c = c0 + 2 * tx;
        float v0 = 0.f, v1 = 0.f;
        if (r < rows && c < cin) {
          const float* wsrc = J.w + ((long long)r * J.T + st) * J.Ci + c;
          if (even) { float2 v = *(const float2*)wsrc; v0 = v.x; v1 = v.y; }
          else { v0 = wsrc[0]; if (c + 1 < cin) v1 = wsrc[1]; }
        }
        tile[2 * tx][ty + 8 * j] = v0;
        tile[2 * tx + 1][ty + 8 * j] = v1;
      }
      __syncthreads();
    }
    if (frag) {
      if (!(cin & 7)) {
        // one 16-byte chunk per thread: rows r0 + (tid & 31), columns c0 + 8*(tid >> 5) .. +8
        const int rl = threadIdx.x & 31, q = threadIdx.x >> 5;
        const int r = r0 + rl, c = c0 + q * 8;
        if (r < rows && c < cin) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = tile[q * 8 + j][rl];
          *(uint4*)(J.out + packed_elem(r, t * cin + c, J.rows_pad, J.K_pad)) = pack8(f);
        }
      } else {                                        // channel counts that are no multiple of 8 (the 255-channel heads)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = r0 + tx, c = c0 + ty + 8 * j;
          if (r < rows && c < cin) J.out[packed_elem(r, t * cin + c, J.rows_pad, J.K_pad)] = f2bf(tile[ty + 8 * j][tx]);
        }
      }
      __syncthreads();
    } else if (J.transpose) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int r = r0 + ty + 8 * j, c = c0 + 2 * tx;
        if (r < rows && c < cin) {
          bf16_t* o = J.out + (long long)r * J.K_pad + t * cin + c;
          if (even) {
            *(uint32_t*)o = pack2bf(tile[2 * tx][ty + 8 * j], tile[2 * tx + 1][ty + 8 * j]);
          } else {                                    // odd channel counts: element-wise, nothing outside the tap's columns
            o[0] = f2bf(tile[2 * tx][ty + 8 * j]);
            if (c + 1 < cin) o[1] = f2bf(tile[2 * tx + 1][ty + 8 * j]);
          }
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int r = r0 + ty + 8 * j, c = c0 + 2 * tx;
        if (r < rows && c < cin) {
          const float* wsrc = J.w + ((long long)r * J.T + st) * J.Ci + c;
          bf16_t* o = J.out + (long long)r * J.K_pad + t * cin + c;
          if (even) {
            float2 v = *(const float2*)wsrc;
            *(uint32_t*)o = pack2bf(v.x, v.y);
          } else {
            o[0] = f2bf(wsrc[0]);
            if (c + 1 < cin) o[1] = f2bf(wsrc[1]);
          }
        }
      }
    }
  }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Co, int T, int Ci,
                                    int transpose, int ntaps_out, unsigned long long srccode, int rows_pad,
                                    int K_pad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long tot = (long long)rows_pad * K_pad;
  if (i >= tot) return;
  int r = (int)(i / K_pad), k = (int)(i - (long long)r * K_pad);
  int rows = transpose ? Ci : Co, cin = transpose ? Co : Ci;
  float v = 0.f;
  if (r < rows && k < ntaps_out * cin) {
    int t = k / cin, c = k - t * cin;
    int st = (int)((srccode >> (4 * t)) & 15);
    v = transpose ? w[((long long)c * T + st) * Ci + r] : w[((long long)r * T + st) * Ci + c];
  }
  out[packed_elem(r, k, rows_pad, K_pad)] = f2bf(v);
}

}  // namespace

extern "C" int mgd_stem_fwd(const float* image, const float* w, void* y, float* stats, int stats_replicas, int N,
                            int H, int W, void* stream) {
  MGD_REQUIRE(image && w && y, "stem_fwd: null pointer");
  MGD_REQUIRE(!stats || stats_replicas >= 1, "stem_fwd: stats_replicas");
  MGD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && (long long)N * H * W * 3 < (1ll << 31), "stem_fwd: N=%d H=%d W=%d", N, H, W);
  long long grid = (long long)N * ((H + 3) / 4) * ((W + 63) / 64);
  if (grid > 256 * 8) grid = 256 * 8;               // persistent blocks: the statistics leave once per block
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, image, w, (bf16_t*)y, stats,
                     stats_replicas > 0 ? stats_replicas : 1, N, H, W, (const float*)nullptr, 0.f);
  MGD_CHECK_LAUNCH("stem_fwd");
  return MGD_OK;
}

extern "C" int mgd_stem_fwd_act(const float* image, const float* w, const float* bias, float act_slope, void* y, int N,
                                int H, int W, void* stream) {
  MGD_REQUIRE(image && w && bias && y, "stem_fwd_act: null pointer");
  MGD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && (long long)N * H * W * 3 < (1ll << 31), "stem_fwd_act: N=%d H=%d W=%d", N, H, W);
  long long grid = (long long)N * ((H + 3) / 4) * ((W + 63) / 64);
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, image, w, (bf16_t*)y,
                     (float*)nullptr, 1, N, H, W, bias, act_slope);
  MGD_CHECK_LAUNCH("stem_fwd_act");
  return MGD_OK;
}

extern "C" int mgd_stem_wgrad(const float* image, const void* dy, float* dw, int N, int H, int W, void* stream) {
  MGD_REQUIRE(image && dy && dw, "stem_wgrad: null pointer");
  MGD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && (long long)N * H * W * 3 < (1ll << 31), "stem_wgrad: N=%d H=%d W=%d", N, H, W);
  long long tiles = (long long)N * ((H + 3) / 4) * ((W + 63) / 64);
  int grid = (int)(tiles < 256 * 4 ? tiles : 256 * 4);
  hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, image, (const bf16_t*)dy,
                     dw, N, H, W, StemBn{});
  MGD_CHECK_LAUNCH("stem_wgrad");
  return MGD_OK;
}

extern "C" int mgd_stem_wgrad_bn(const float* image, const void* da, const void* y, const float* scale, const float* shift,
                                 const float* save_mean, const float* save_invstd, const float* sums, int replicas,
                                 float* dgamma, float* dbeta, float slope, float* dw, int N, int H, int W, void* stream) {
  MGD_REQUIRE(image && da && y && scale && shift && save_mean && save_invstd && sums && dw && replicas >= 1,
              "stem_wgrad_bn: null pointer / replicas");
  MGD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && (long long)N * H * W * 3 < (1ll << 31), "stem_wgrad_bn: N=%d H=%d W=%d", N, H, W);
  long long tiles = (long long)N * ((H + 3) / 4) * ((W + 63) / 64);
  int grid = (int)(tiles < 256 * 4 ? tiles : 256 * 4);
  StemBn bn{(const bf16_t*)y, scale, shift, save_mean, save_invstd, sums, dgamma, dbeta, replicas, slope};
  hipLaunchKernelGGL(stem_wgrad_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, image, (const bf16_t*)da,
                     dw, N, H, W, bn);
  MGD_CHECK_LAUNCH("stem_wgrad_bn");
  return MGD_OK;
}

extern "C" int mgd_pack_weights(const float* w, void* out, int Co, int T, int Ci, int transpose, int ntaps_out,
                                const int32_t* src_tap_host, int rows_pad, int K_pad, void* stream) {
  MGD_REQUIRE(w && out && src_tap_host, "pack: null pointer");
  MGD_REQUIRE(ntaps_out >= 1 && ntaps_out <= 9 && T <= 9, "pack: taps");
  unsigned long long code = 0;
  for (int t = 0; t < ntaps_out; ++t) {
    MGD_REQUIRE(src_tap_host[t] >= 0 && src_tap_host[t] < T, "pack: src_tap out of range");
    code |= (unsigned long long)src_tap_host[t] << (4 * t);
  }
  int rows = transpose ? Ci : Co, cin = transpose ? Co : Ci;
  MGD_REQUIRE(rows_pad >= rows && K_pad >= ntaps_out * cin, "pack: padded sizes too small");
  // images of 128-row tiles are written in MFMA-fragment order (packed_elem): blocks of 128 rows x 64 K
  MGD_REQUIRE((rows_pad & 127) || K_pad % 64 == 0, "pack: K_pad=%d must be a multiple of 64 for 128-row (fragment-order) images", K_pad);
  long long tot = (long long)rows_pad * K_pad;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)out,
                     Co, T, Ci, transpose, ntaps_out, code, rows_pad, K_pad);
  MGD_CHECK_LAUNCH("pack_weights");
  return MGD_OK;
}

extern "C" int mgd_stem_im2col(const float* image, void* out, int N, int H, int W, void* stream) {
  MGD_REQUIRE(image && out, "stem_im2col: null pointer");
  long long nvec = (long long)N * H * W * 4;
  long long g = (nvec + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, image, (bf16_t*)out, N, H, W);
  MGD_CHECK_LAUNCH("stem_im2col");
  return MGD_OK;
}

extern "C" int mgd_pack_weights_batch(const mgd_pack_job* jobs_dev, int njobs, int64_t total, void* stream) {
  MGD_REQUIRE(jobs_dev && njobs >= 1 && total >= 1, "pack_batch: bad arguments");
  static_assert(sizeof(mgd_pack_job) == sizeof(PackJob), "mgd_pack_job layout");
  long long g = total;               // total = number of 32x64 tiles over all jobs
  if (g > 256 * 64) g = 256 * 64;
  hipLaunchKernelGGL(pack_batch_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs_dev, njobs,
                     (long long)total);
  MGD_CHECK_LAUNCH("pack_batch");
  return MGD_OK;
}

// Diagnostic: reads and clears the phase stamps of the stamped conv_gemm9_kernel build (MGD_DBG=4096, MGD_GEMM9_PP=4).
// Workspace of the latency form (mgd_conv_desc.latency with splitk > 1): uncached device memory owned by the library, one
// buffer per device, grown on demand (the first 16 KiB, the tile tickets, zero-filled).  Not for use during stream capture:
// call once with the largest size before capturing.
