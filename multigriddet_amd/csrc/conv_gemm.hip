// Convolution engine for gfx950: implicit-GEMM ("gather-GEMM") on v_mfma_f32_16x16x32_bf16.
//
// Forward, data-gradient and 1x1 all run through conv_gather_gemm_kernel; the weight gradient has
// its own kernel (both GEMM operands are strided along the contraction index there, so fragments
// are fetched with ds_read_b64_tr_b16).  Replaces Keras Conv2D + autodiff as used by
// DarknetConv2D_BN_Leaky (reference multigriddet/models/layers.py:43-49,88-95).
//
// Data layout in HBM: activations NHWC bf16; packed weights bf16 [Co_pad][K_pad], K = tap*Ci + ci;
// master weights / weight gradients fp32 [Co][taps][Ci] (OHWI).
//
// Block = 256 threads = 4 waves (one per SIMD).  MFMA roles: A := weight tile (rows = output
// channels), B := gathered pixel tile (cols = pixels), so each lane ends with 4 consecutive output
// channels of one pixel per accumulator tile - 8-byte packed bf16 pieces that are staged through LDS
// and leave as whole 16-byte/256-byte NHWC rows.
#include "conv_common.hpp"

extern "C" int64_t mgd_latency_workspace_size(int tiles, int ranges);

namespace {

template <int WC, int WP, int MT, int NT, int NST>
__global__ __launch_bounds__(64 * WC * WP) MGD_VGPR_MFMA void conv_gemm2_kernel(GemmArgs a) {
  constexpr int BNC = WC * MT * 16;
  constexpr int BMP = WP * NT * 16;
  constexpr int NTHR = 64 * WC * WP;      // 4 waves (128-pixel tile, 2 blocks/CU) or 8 waves (256-pixel tile)
  constexpr int RPR = NTHR / 8;            // tile rows covered by one LDS-DMA round of the block
  static_assert(BMP % RPR == 0 && BNC % RPR == 0, "tile/threads");
  constexpr int WCH = BNC / RPR;  // weight glds per thread per stage
  constexpr int XCH = BMP / RPR;  // pixel glds per thread per stage
  constexpr int LPS = WCH + XCH;  // loads per stage per wave
  constexpr int STAGE = (BNC + BMP) * ROWB;
  constexpr int EPI_MAX = BMP * (BNC * 4 + 16);    // fp32 epilogue tile (largest user of the region)
  constexpr int AUX = NST * STAGE > EPI_MAX ? NST * STAGE : EPI_MAX;   // then row_dst (BMP x 8 B) + colred (256 x 4 B)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* row_dst = (long long*)(smem + AUX);
  float* colred = (float*)(smem + AUX + BMP * 8);
  uint2* row_src = (uint2*)(smem + AUX + BMP * 8 + 1024);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wc = wave / WP, wp = wave % WP;

  const int L = xcd_remap(blockIdx.x, a.nblk);
  const int tc = L % a.tilesC, tp = L / a.tilesC;
  const int co0 = tc * BNC;
  const int pix0 = tp * BMP;

  make_row_tables(a, pix0, tid, BMP, row_dst, row_src);
  __syncthreads();

  // bf16 epilogue operands that live in HBM (residual-gradient addend, y of the fused BN-backward reduction) are
  // fetched here, before the K-loop, so their latency hides under the main loop instead of being paid eight times
  // in a row by the epilogue.  (Fetching them from inside the K-loop, a few steps before its end, was tried: the
  // extra branch cost the loop its schedule and every layer 30-60 %.)
  constexpr int CPB = BNC / 8;                 // 16-byte bf16 chunks per output row
  constexpr int EPC = BMP * CPB / NTHR;        // chunks per thread
  static_assert(BMP * CPB % NTHR == 0 && NTHR % CPB == 0, "epilogue mapping");
  const bool bnred = a.bn_y != nullptr && !a.dst_f32;
  const bool addpre = a.addend != nullptr && !a.dst_f32;
  float bnp[4][8];                             // scale, shift, mean, invstd of my 8 channels (bnred only)
  if (bnred) {
    const int c = co0 + (tid % CPB) * 8;
    const float* ps[4] = {a.bn_scale, a.bn_shift, a.bn_mean, a.bn_invstd};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
      if (c < a.Co) { lo = *(const f32x4*)(ps[k] + c); hi = *(const f32x4*)(ps[k] + c + 4); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { bnp[k][j] = lo[j]; bnp[k][4 + j] = hi[j]; }
    }
  }
  uint4 ypre[EPC], apre[EPC];
  if (bnred || addpre) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EPC; ++it) {
      const int q = tid + it * NTHR;
      const int r = q / CPB, ch = q - r * CPB;
      const long long off = row_dst[r];
      const int c = co0 + ch * 8;
      const bool ok = off >= 0 && c < a.Co;
      ypre[it] = (bnred && ok) ? *(const uint4*)(a.bn_y + off + c) : make_uint4(0, 0, 0, 0);
      apre[it] = (addpre && ok) ? *(const uint4*)(a.addend + off + c) : make_uint4(0, 0, 0, 0);
    }
  }

  // thread -> LDS slot (tid & 7) of rows (tid >> 3) + RPR i ; source chunk kc = slot ^ (row & 7).
  // Address generation is kept off the critical path (the first version spent 5.7 VALU instructions
  // per MFMA on it): per row a 32-bit byte offset and a 9-bit tap-validity mask are computed once; per
  // K-step a load costs a mask test, a select and an add.  Out-of-image taps load from a zero page.  (They used to
  // load from offset 0 with the owning lane zeroing its LDS slot once the DMA had landed: in front of that ds_write hipcc
  // puts s_waitcnt vmcnt(0) - it may alias the LDS-DMA writes in flight - and the fix-up sat between the wait and the
  // barrier of most K-steps of a 3x3 layer; without it the 128-tile launches are 5-9 % faster.)
  const int rlo = tid >> 3;
  const int kc = (tid & 7) ^ (rlo & 7);
  unsigned xoff[XCH];
  unsigned vmask[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const uint2 rs = row_src[rlo + RPR * i];
    xoff[i] = rs.x;
    vmask[i] = rs.y;
  }
  int tap = (kc * 8) / a.Ci;
  int cch = (kc * 8) - tap * a.Ci;
  unsigned woff[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) woff[i] = (unsigned)(((long long)(co0 + rlo + RPR * i) * a.K_pad + kc * 8) * 2);
  const char* xbase = (const char*)a.src;
  const char* wbase = (const char*)a.wpk;
  // my LDS slots (byte offsets inside a stage) for the zero fix-up

  const void* zero = (const void*)g_zero_page;
  asm volatile("" : "+s"(zero));
  auto issue = [&](int ks, int buf) {
    unsigned char* wb = smem + buf * STAGE + wave * 1024;
    unsigned char* xb = smem + buf * STAGE + BNC * ROWB + wave * 1024;
#pragma unroll
    for (int i = 0; i < WCH; ++i)
      glds16(wbase + (MGD_DBG(a, 1) ? 0u : woff[i] + (unsigned)ks * (BK * 2)), wb + i * (RPR * ROWB));
    int dh = (int)((a.tapcode >> (4 * tap)) & 3) - 1;
    int dw = (int)((a.tapcode >> (4 * tap + 2)) & 3) - 1;
    int toff = ((dh * a.Ws + dw) * a.Ci + cch) * 2;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      bool v = (vmask[i] >> tap) & 1u;
      unsigned off = xoff[i] + (unsigned)toff;
      if (MGD_DBG(a, 1)) off = 0u;
      glds16(v ? (const void*)(xbase + off) : zero, xb + i * (RPR * ROWB));
    }
    cch += BK;
    while (cch >= a.Ci) {
      cch -= a.Ci;
      ++tap;
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K_pad / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);

  const int fr = lane & 15, fq = lane >> 4;
  // fragment read offsets inside a stage (k-half kk flips bit 6 of the byte offset)
  int wro[MT], xro[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) wro[m] = lds_off((wc * MT + m) * 16 + fr, fq);
#pragma unroll
  for (int n = 0; n < NT; ++n) xro[n] = BNC * ROWB + lds_off((wp * NT + n) * 16 + fr, fq);

  for (int ks = 0; ks < nk; ++ks) {
    int pending = min(NST - 2, nk - 1 - ks);
    if (NST >= 4 && pending >= 2) wait_vmcnt<2 * LPS>();
    else if (NST >= 3 && pending == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    const int cur = ks % NST;
    unsigned char* sb = smem + cur * STAGE;
    __builtin_amdgcn_s_barrier();
    if (ks + NST - 1 < nk) issue(ks + NST - 1, (ks + NST - 1) % NST);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[MT], xf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) wf[m] = *(const bf16x8*)(sb + (wro[m] ^ (kk << 6)));
#pragma unroll
      for (int n = 0; n < NT; ++n) xf[n] = *(const bf16x8*)(sb + (xro[n] ^ (kk << 6)));
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[n], acc[m][n], 0, 0, 0);
    }
  }
  __syncthreads();

  // ---- epilogue (same as v1)
  const int esz = a.dst_f32 ? 4 : 2;
  const int EROW = BNC * esz + 16;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int cl = (wc * MT + m) * 16 + fq * 4;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (a.bias) {
      int c = co0 + cl;
      if (c + 3 < a.Co) { b0 = a.bias[c]; b1 = a.bias[c + 1]; b2 = a.bias[c + 2]; b3 = a.bias[c + 3]; }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      int pl = (wp * NT + n) * 16 + fr;
      f32x4 v = acc[m][n];
      v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
      if (a.act_slope != 0.f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * a.act_slope;
      }
      if (a.dst_f32) {
        *(f32x4*)(smem + pl * EROW + cl * 4) = v;
      } else {
        uint2 p;
        p.x = pack2bf(v[0], v[1]);
        p.y = pack2bf(v[2], v[3]);
        *(uint2*)(smem + pl * EROW + cl * 2) = p;
      }
    }
  }
  lds_barrier();        // LDS tile published; no vmcnt wait (see lds_barrier)
  const int CPR = BNC * esz / 16;
  if (a.dst_f32) {
    for (int q = tid; q < BMP * CPR; q += NTHR) {
      int r = q / CPR, ch = q - r * CPR;
      long long off = row_dst[r];
      int c = co0 + ch * 4;
      if (off < 0 || c >= a.Co) continue;
      *(uint4*)((float*)a.dst + off + c) = *(const uint4*)(smem + r * EROW + ch * 16);
    }
    return;
  }
  // bf16 rows.  NTHR % CPB == 0, so a thread owns the same 8 channels in every row it writes, and both per-channel
  // reductions ride on the output loop in registers:
  //   stats  (forward, BatchNorm batch statistics):  r1 = sum y, r2 = sum y^2 of the bf16-ROUNDED values;
  //   bnred  (data gradient; BatchNorm backward of the layer that consumes this tensor as its da):
  //          r1 = sum dyh, r2 = sum dyh * yhat,  dyh = da * leaky'(y*scale+shift).
  // The residual-gradient addend and y were fetched before the K-loop (apre / ypre).
  const bool stats = a.stats != nullptr;
  float r1[8], r2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r1[j] = r2[j] = 0.f;
#pragma unroll
  for (int it = 0; it < EPC; ++it) {
    const int q = tid + it * NTHR;
    const int r = q / CPB, ch = q - r * CPB;
    const long long off = row_dst[r];
    const int c = co0 + ch * 8;
    if (off < 0 || c >= a.Co) continue;
    uint4 v = *(const uint4*)(smem + r * EROW + ch * 16);
    if (addpre) {
      float f[8], g[8];
      unpack8(v, f);
      unpack8(apre[it], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j];
      v = pack8(f);
    }
    *(uint4*)((bf16_t*)a.dst + off + c) = v;
    if (stats) {
      float d[8];
      unpack8(v, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) { r1[j] += d[j]; r2[j] = fmaf(d[j], d[j], r2[j]); }
    } else if (bnred) {
      float d[8], yv[8];
      unpack8(v, d);
      unpack8(ypre[it], yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = fmaf(yv[j], bnp[0][j], bnp[1][j]);
        float dd = z > 0.f ? d[j] : d[j] * a.bn_slope;
        r1[j] += dd;
        r2[j] = fmaf(dd * (yv[j] - bnp[2][j]), bnp[3][j], r2[j]);
      }
    }
  }
  if (stats || bnred) {
    // lanes with equal (lane % CPB) own the same channels: butterfly over the other lane bits, one partial row per
    // wave in LDS (above the bf16 tile, inside the fp32-sized epilogue region), then 2*BNC threads fold the waves
    // and issue one global atomic each into replica (block % R).
    float* wred = (float*)(smem + BMP * (BNC * 2 + 16));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = CPB; o < 64; o <<= 1) { r1[j] += __shfl_xor(r1[j], o, 64); r2[j] += __shfl_xor(r2[j], o, 64); }
    }
    if (lane < CPB) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wred[wave * 2 * BNC + lane * 8 + j] = r1[j];
        wred[wave * 2 * BNC + BNC + lane * 8 + j] = r2[j];
      }
    }
    lds_barrier();      // NOT __syncthreads(): that would wait for the tile's global stores to be acknowledged
    if (tid < 2 * BNC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NTHR / 64; ++w) t += wred[w * 2 * BNC + tid];
      int which = tid / BNC, col = tid - which * BNC;
      if (co0 + col < a.Co) {
        int rep = blockIdx.x % a.stats_replicas;
        float* dstp = stats ? a.stats : a.bn_sums;
        atomicAdd(dstp + ((long long)rep * 2 + which) * a.Co + co0 + col, t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the gather-GEMM kernels: accumulators -> LDS tile -> whole NHWC rows, with bias, fp32
// output, the residual-gradient addend, and the two per-channel reductions that ride on the output loop
// (BatchNorm batch statistics in the forward pass; the BatchNorm-backward sums of the consumer layer in the
// data gradient).  prefetch() fetches the HBM operands of the bf16 epilogue before the K-loop.

// ------------------------------------------------------------------------------------------------
// v6: producer / consumer form of the gather-GEMM.  The v2 K-loop is bound by LDS-DMA issue + landing latency and the
// block barrier per K-step (MGD_DBG=2 stamps: a wave spends 24 % of a K-step issuing its 8 LDS-DMA pieces, 42 %
// waiting for the stage, 10 % in the barrier, 24 % on fragment reads + MFMA).  Here a block is 8 waves: waves 0-3
// only load (each its quarter of every stage), waves 4-7 only compute (64x64 wave tiles as in v2), with ONE
// s_barrier per K-step as the hand-off over a 3-stage LDS ring: loader waves wait for stage s to land, everyone
// meets at the barrier, then the loaders issue stage s+2 into the slot the consumers have just left while the
// consumers compute stage s - a wave's LDS-DMA issue time (~840 ticks per stage) no longer sits in front of its
// own MFMAs.  One block per CU; out-of-image taps come from a zero page.  (A first version handed stages over
// through per-wave LDS counters instead of the barrier: each side then waited about as long for the other's
// counter as it worked, and it lost to this form on every layer.)
template <int WC, int WP, int MT, int NT, int NS>
__global__ __launch_bounds__(512) void conv_gemm6_kernel(GemmArgs a) {
  using Epi = GemmEpilogue<WC, WP, MT, NT>;
  constexpr int BNC = WC * MT * 16;
  constexpr int BMP = WP * NT * 16;
  static_assert(WC * WP == 4 && BMP == 128, "4 consumer waves, 128-pixel tile");
  constexpr int WCH = BNC / 32;                 // weight pieces per loader wave per stage
  constexpr int XCH = BMP / 32;                 // pixel pieces per loader wave per stage
  constexpr int LPS = WCH + XCH;
  constexpr int STAGE = (BNC + BMP) * ROWB;
  constexpr int RING = NS * STAGE;
  constexpr int EPI_MAX = BMP * (BNC * 4 + 16);
  constexpr int AUX = RING > EPI_MAX ? RING : EPI_MAX;
  static_assert(NS == 3 || NS == 4, "3- or 4-stage ring");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* row_dst = (long long*)(smem + AUX);
  uint2* row_src = (uint2*)(smem + AUX + BMP * 8);

  const int tid = threadIdx.x;
  const bool loader = tid < 256;
  const int ltid = tid & 255, lane = tid & 63, wave = ltid >> 6;
  const int L = xcd_remap(blockIdx.x, a.nblk);
  const int tc = L % a.tilesC, tp = L / a.tilesC;
  const int co0 = tc * BNC;
  const int pix0 = tp * BMP;
  const int nk = a.K_pad / BK;

  make_row_tables(a, pix0, tid, BMP, row_dst, row_src);
  __syncthreads();

  if (loader) {
    // thread -> LDS slot (ltid & 7) of rows (ltid >> 3) + 32 i ; source chunk kc = slot ^ (row & 7)
    const int rlo = ltid >> 3;
    const int kc = (ltid & 7) ^ (rlo & 7);
    unsigned xoff[XCH], vmask[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const uint2 rs = row_src[rlo + 32 * i];
      xoff[i] = rs.x;
      vmask[i] = rs.y;
    }
    int tap = (kc * 8) / a.Ci;
    int cch = (kc * 8) - tap * a.Ci;
    // weights: the packed image is in fragment order (packed_elem) - the tile of K-step s is one contiguous 16-KiB block,
    // copied as it lies (piece i of this thread = chunks i*256 + ltid), and the consumers read it back lane-linear
    static_assert(BNC == 128, "fragment-ordered weight tiles are 128 rows");
    const char* xbase = (const char*)a.src;
    const char* wbase = (const char*)a.wpk + ((size_t)tc * nk * 16384) + (size_t)ltid * 16;
    const void* zero = (const void*)g_zero_page;

    auto issue = [&](int s, int slot) {
      unsigned char* wb = smem + slot * STAGE + wave * 1024;
      unsigned char* xb = wb + BNC * ROWB;
#pragma unroll
      for (int i = 0; i < WCH; ++i) glds16(wbase + (size_t)s * 16384 + i * 4096, wb + i * (32 * ROWB));
      int dh = (int)((a.tapcode >> (4 * tap)) & 3) - 1;
      int dw = (int)((a.tapcode >> (4 * tap + 2)) & 3) - 1;
      int toff = ((dh * a.Ws + dw) * a.Ci + cch) * 2;
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        bool v = (vmask[i] >> tap) & 1u;
        const void* g = v ? (const void*)(xbase + (long long)xoff[i] + toff) : zero;
        glds16(g, xb + i * (32 * ROWB));
      }
      cch += BK;
      while (cch >= a.Ci) { cch -= a.Ci; ++tap; }
    };
    // look-ahead LA = NS - 1 stages: after the barrier of K-step s the slot of stage s-1 is free and stage s+LA goes
    // into it, so a stage has LA-1 whole K-steps to land before the loaders wait for it
    constexpr int LA = NS - 1;
    int islot = 0;
#pragma unroll
    for (int p = 0; p < LA; ++p)
      if (p < nk) { issue(p, islot); if (++islot == NS) islot = 0; }
    for (int s = 0; s < nk; ++s) {
      const int younger = min(LA - 1, nk - 1 - s);                 // stages that may stay in flight behind stage s
      if (younger >= 2) wait_vmcnt<2 * LPS>(); else if (younger == 1) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                                // consumers: stage s is yours, stage s-1's slot is free
      if (s + LA < nk) { issue(s + LA, islot); if (++islot == NS) islot = 0; }
    }
    __syncthreads();
    return;
  }

  // ---------------------------------------------------------------- consumers
  const int wc = wave / WP, wp = wave % WP;
  Epi epi;
  epi.prefetch(a, row_dst, co0, ltid, false);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  int wro[MT], xro[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) wro[m] = (wc * MT + m) * 2048 + lane * 16;   // fragment (m, kk) = 1 KiB, lane-linear
#pragma unroll
  for (int n = 0; n < NT; ++n) xro[n] = BNC * ROWB + lds_off((wp * NT + n) * 16 + fr, fq);
  int slot = 0;
  for (int s = 0; s < nk; ++s) {
    __builtin_amdgcn_s_barrier();
    const unsigned char* sb = smem + slot * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[MT], xf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) wf[m] = *(const bf16x8*)(sb + wro[m] + kk * 1024);
#pragma unroll
      for (int n = 0; n < NT; ++n) xf[n] = *(const bf16x8*)(sb + (xro[n] ^ (kk << 6)));
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[n], acc[m][n], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my fragment reads of this slot are done before the next barrier
    if (++slot == NS) slot = 0;
  }
  __syncthreads();
  epi.run(a, acc, smem, row_dst, co0, ltid);
}

// ------------------------------------------------------------------------------------------------
// Gather-GEMM for 128-channel tiles: the WEIGHT operand comes straight from global memory.
// An LDS-DMA instruction costs its wave ~125 cycles of issue time per 1-KiB piece; with both operands staged that way a wave
// spent eight of them per K-step (4 weight + 4 pixel pieces) in front of 32 MFMA = 512 cycles.  The weights need no gather
// and no transposition: the packed image of a 128-row tile is stored in MFMA-fragment order (packed_elem), so a wave's A
// operand of a K-step is eight plain, fully coalesced global_load_dwordx4 (vector-memory path: L1/TA, a fraction of the
// issue cost), loaded NST - 1 K-steps ahead into spare register sets.  The LDS-DMA ring carries the gathered pixel tile
// only: half the pieces, half the ring (16 KB per stage), no LDS reads for the weight fragments - and the block needs so
// little LDS (40 KB) and so few registers (165) that THREE blocks share a CU (three waves per SIMD).  Measured against the
// form with both operands in the ring (conv_gemm2_kernel<2,2,4,4,2>, since removed), 608x608 batch 16: 128->256 at 76x76
// 79 -> 68 us, 256->512 at 38x38 74 -> 64 us, 64->128 at 152x152 103 -> 85 us; a deeper ring (NST = 3, 4) is not faster.
// Tile 128 x 128, 4 waves of 32 channels x 128 pixels (WC = 4) or 64 x 64 (WC = 2), epilogue = GemmEpilogue.
template <int NST, int WPE, bool ABL = false, int WC = 2, bool UNI = false>   // ABL: ablation build (MGD_DBG bits switch parts of the loop off); WC: waves along the channels;
//                                                                             UNI: (tap, channel) of a K-step is wave-uniform (ntaps == 1 or Ci % 64 == 0)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void conv_gemm8_kernel(GemmArgs a) {
  constexpr int WP = 4 / WC, MT = 8 / WC, NT = 8 / WP, BNC = 128, BMP = 128, NTHR = 256;
  constexpr int RPR = NTHR / 8, XCH = BMP / RPR;          // 32 rows per LDS-DMA round, 4 pixel pieces per wave and stage
  constexpr int STAGE = BMP * ROWB;
  using Epi = GemmEpilogue<WC, WP, MT, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* row_dst = (long long*)(smem + a.aux);   // behind max(ring, epilogue tile): the fp32 tile of the heads is larger
  uint2* row_src = (uint2*)(smem + a.aux + BMP * 8);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;
  int bid = (int)blockIdx.x;
  if (a.ncls > 1) {                                   // tap classes in one launch: this block's class (wave-uniform)
    // class-major: block b works on class b / nblk.  (korder & 2, measurement: the classes of one tile as neighbours on one
    // XCD, so that they share the source rows in its L2 - 82 us against 62 on the 512-channel layers: blocks with 1 / 2 / 2 / 4
    // taps side by side leave the CUs unevenly loaded)
    int cls;
    if (a.korder & 2) { bid = xcd_remap(bid, a.nblk * a.ncls); cls = bid % a.ncls; bid /= a.ncls; }
    else { cls = bid / a.nblk; bid = xcd_remap(bid - cls * a.nblk, a.nblk); }
    // (static indices + selects: a runtime index into the by-value argument block sends the whole block to scratch memory -
    // 424 bytes per lane and 15 % of the step, measured)
#define MGD_PICK(f) (cls == 0 ? a.f[0] : cls == 1 ? a.f[1] : cls == 2 ? a.f[2] : a.f[3])
    a.wpk = MGD_PICK(c_wpk); a.tapcode = MGD_PICK(c_tapcode); a.K_pad = MGD_PICK(c_K_pad); a.ntaps = MGD_PICK(c_ntaps);
    a.out_off_h = MGD_PICK(c_off_h); a.out_off_w = MGD_PICK(c_off_w); a.rowmask = MGD_PICK(c_rowmask); a.colmask = MGD_PICK(c_colmask);
#undef MGD_PICK
  }
  const int L = a.ncls > 1 ? bid : xcd_remap(bid, a.nblk);
  const int tc = L % a.tilesC, tp = L / a.tilesC;
  const int co0 = tc * BNC, pix0 = tp * BMP;
  if (ABL && (a.dbg & 1024)) return;                  // dispatch cost alone
  const int nk_all = a.K_pad / BK;
  constexpr int ks_lo = 0;
  const int nk = (ABL && (a.dbg & 2048)) ? 0 : nk_all;   // 2048: prologue + epilogue, no K-loop

  make_row_tables(a, pix0, tid, BMP, row_dst, row_src);
  __syncthreads();
  Epi epi;                                           // its HBM operands are fetched late (run<.., LATE>): fetched here they
  //                                                    cost 52 bytes of scratch under the 256-register cap of two waves per SIMD

  // pixel rows of this thread (as conv_gemm2_kernel): byte offset + tap-validity mask once, per K-step a select and an add
  const int rlo = tid >> 3;
  const int kc = (tid & 7) ^ (rlo & 7);
  unsigned xoff[XCH], vmask[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const uint2 rs = row_src[rlo + RPR * i];
    xoff[i] = rs.x;
    vmask[i] = rs.y;
  }
  int tap = (kc * 8 + ks_lo * BK) / a.Ci;
  int cch = (kc * 8 + ks_lo * BK) - tap * a.Ci;
  // out-of-image taps are fetched from a zero page (64-bit per-lane source addresses): conv_gemm2_kernel loads them from
  // offset 0 and overwrites the LDS slot with zeros afterwards, and in front of that ds_write hipcc puts s_waitcnt vmcnt(0)
  // (it may alias the LDS-DMA writes in flight) - harmless with two stages, but it drains any deeper ring on every K-step
  // whose tap leaves the image for some lane, i.e. on most of them
  const char* xbase = (const char*)a.src;
  const void* zero = (const void*)g_zero_page;
  asm volatile("" : "+s"(zero));
  // UNI (round 3): tap and first channel of a K-step are the same for every lane, so they live in SGPRs, the pixel rows go
  // out as buffer_load ... lds through a raw descriptor over the activation tensor (a lane whose tap leaves the image, or
  // whose K index is padding, presents an out-of-range offset and the hardware writes zeros): per LDS-DMA a mask test, a
  // 32-bit select and an add instead of a shift, a 64-bit select and a 64-bit add - 40 -> 20 vector instructions per K-step
  i32x4 srd = {0, 0, 0, 0};
  int s_tap = 0, s_c0 = 0, s_toff = 0, s_k0 = ks_lo * BK;
  auto tap_off = [&](int tp_) {
    const int dh = (int)((a.tapcode >> (4 * tp_)) & 3) - 1;
    const int dw = (int)((a.tapcode >> (4 * tp_ + 2)) & 3) - 1;
    return (dh * a.Ws + dw) * a.Ci * 2;
  };
  if (UNI) {
    const unsigned long long p = (unsigned long long)a.src;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    srd[2] = __builtin_amdgcn_readfirstlane((unsigned)((long long)a.N * a.Hs * a.Ws * a.Ci * 2));
    srd[3] = 0x00020000;
    s_tap = (ks_lo * BK) / a.Ci;
    s_c0 = ks_lo * BK - s_tap * a.Ci;
    s_toff = tap_off(s_tap) + s_c0 * 2;
#pragma unroll
    for (int i = 0; i < XCH; ++i) xoff[i] += kc * 16;
  }
  const int Kreal = a.ntaps * a.Ci;
  const unsigned lds_w = lds_addr(smem) + wave * 1024;
  const int nchunk = a.Ci / BK;                        // chunk-major order (a.korder): 64-channel chunks per tap
  int s_chunk = 0, s_step = ks_lo;
  auto issue = [&](int buf) -> int {                   // stages go out in order; returns the K block of the packed image it fetched
    if constexpr (UNI) {
      const unsigned bit = 1u << s_tap;
      const bool kin = s_k0 + kc * 8 < Kreal;             // K padding of the last step reads as zeros
      unsigned vo[XCH];
#pragma unroll
      for (int i = 0; i < XCH; ++i) vo[i] = ((vmask[i] & bit) && kin) ? xoff[i] + (unsigned)s_toff : 0xFFFFFFF0u;
      dma_rows_asm<XCH, RPR * ROWB>(vo, srd, lds_w + buf * STAGE);
      if (a.korder & 1) {                                  // (chunk, tap): K_pad == ntaps * Ci, no padding
        const int wk = s_tap * nchunk + s_chunk;
        if (++s_tap == a.ntaps) { s_tap = 0; ++s_chunk; }
        s_toff = tap_off(s_tap) + s_chunk * (BK * 2);
        return wk;
      }
      const int wk = s_k0 / BK;
      s_k0 += BK;
      s_c0 += BK;
      s_toff += BK * 2;
      if (s_c0 >= a.Ci) { s_c0 -= a.Ci; ++s_tap; s_toff = tap_off(s_tap) + s_c0 * 2; }
      return wk;
    } else {
      unsigned char* xb = smem + buf * STAGE + wave * 1024;
      const int dh = (int)((a.tapcode >> (4 * tap)) & 3) - 1;
      const int dw = (int)((a.tapcode >> (4 * tap + 2)) & 3) - 1;
      const int toff = ((dh * a.Ws + dw) * a.Ci + cch) * 2;
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        const bool v = (vmask[i] >> tap) & 1u;
        glds16(v ? (const void*)(xbase + (xoff[i] + (unsigned)toff)) : zero, xb + i * (RPR * ROWB));
      }
      cch += BK;
      while (cch >= a.Ci) { cch -= a.Ci; ++tap; }
      return s_step++;
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // MGD_DBG bits of the ablation build: 32 no epilogue, 64 no MFMA, 128 no pixel-fragment reads, 256 no weight-fragment
  // loads, 512 no LDS-DMA; all constant false in the product build
  const bool abl_e = ABL && (a.dbg & 32), abl_m = ABL && (a.dbg & 64), abl_x = ABL && (a.dbg & 128),
             abl_a = ABL && (a.dbg & 256), abl_d = ABL && (a.dbg & 512);

  const int fr = lane & 15, fq = lane >> 4;
  int xro[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) xro[n] = lds_off((wp * NT + n) * 16 + fr, fq);
  // weight fragments: block (tc, ks) = 1024 chunks of 16 B; this wave's eight start at wc*512.  NST register sets: the
  // set of step ks + NST - 1 is requested together with ring stage ks + NST - 1, so both operands have NST - 1 K-steps
  // to arrive (with a single step of distance the round trip of the weight loads sets the K-step, whatever the ring depth)
  const uint4* wl = (const uint4*)a.wpk + ((size_t)tc * nk_all * 16 + (size_t)wc * (MT * 2)) * 64 + lane;
  bf16x8 af[NST][MT][2];
  auto load_a = [&](bf16x8 (&f)[MT][2], int ks) {
    const uint4* w = wl + (size_t)ks * 1024;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) f[m][kk] = __builtin_bit_cast(bf16x8, w[(m * 2 + kk) * 64]);
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (ks_lo + s < nk) { int wk = ks_lo + s; if (!abl_d) wk = issue(s); load_a(af[s], wk); }

  constexpr int GRP = XCH + 2 * MT;                    // vector-memory instructions per stage: 4 LDS-DMA + 8 fragment loads
  for (int ks0 = ks_lo; ks0 < nk; ks0 += NST) {
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int ks = ks0 + j;
      if (ks >= nk) break;
      const int younger = min(NST - 2, nk - 1 - ks);   // stage groups behind this one that may stay in flight
      if (NST >= 4 && younger >= 2) wait_vmcnt_tracked<2 * GRP>();
      else if (NST >= 3 && younger == 1) wait_vmcnt_tracked<GRP>();
      else wait_vmcnt_tracked<0>();
      unsigned char* sb = smem + j * STAGE;
      __builtin_amdgcn_s_barrier();
      // the first half's pixel fragments are requested before the next stage goes out: their LDS round trip then runs
      // under the issue of the LDS-DMA and weight loads (all eight at once would not fit the 168 registers of three
      // waves per SIMD)
      bf16x8 xf[NT];
      if (!abl_x) {
#pragma unroll
        for (int n = 0; n < NT; ++n) xf[n] = *(const bf16x8*)(sb + xro[n]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int jn = (j + NST - 1) % NST;               // static after unrolling
      if (ks + NST - 1 < nk) {
        int wk = ks + NST - 1;
        if (!abl_d) wk = issue(jn);
        if (!abl_a) load_a(af[jn], wk);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (kk == 1 && !abl_x) {
#pragma unroll
          for (int n = 0; n < NT; ++n) xf[n] = *(const bf16x8*)(sb + (xro[n] ^ 64));
        }
        if (!abl_m) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][m][kk], xf[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  if (abl_e) return;
  epi.template run<false, true>(a, acc, smem, row_dst, co0, tid);
}

template <int NST, int WPE, int WC = 2, bool UNI = false>
int launch_gemm8(GemmArgs& a, hipStream_t st) {
  a.tilesC = a.Co_pad / 128;
  a.nblk = a.tilesC * cdiv(a.M, 128);
  size_t ring = (size_t)NST * 128 * ROWB;
  size_t epi = a.dst_f32 ? (size_t)128 * (128 * 4 + 16) : (size_t)128 * (128 * 2 + 16) + 4 * 2 * 128 * 4;
  const int grid8 = a.nblk * (a.ncls > 1 ? a.ncls : 1);
  a.aux = (int)(ring > epi ? ring : epi);
  size_t lds = (size_t)a.aux + 128 * 16 + 64;
  auto k = conv_gemm8_kernel<NST, WPE, false, WC, UNI>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
#ifdef MGD_DIAG
  if (a.dbg & 0xFE0) {                                        // ablation instantiation (tools/ablate_gemm.py)
    auto ka = conv_gemm8_kernel<NST, WPE, true, WC>;
    (void)hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(ka, dim3(grid8), dim3(256), lds, st, a);
    return 0;
  }
#endif
  hipLaunchKernelGGL(k, dim3(grid8), dim3(256), lds, st, a);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Gather-GEMM with a hand-counted memory pipeline (round 3).  conv_gemm8_kernel keeps one stage in flight and drains
// vmcnt(0) every K-step: hipcc cannot count LDS-DMA and register loads on one queue (it waits vmcnt(0) at the first use
// of a loaded register), so a deeper ring never reached the hardware, and a K-step of a wave lasts one full load round
// trip (~2600 cycles against 512 cycles of MFMA; three blocks per CU hide about half).  Here every vector-memory
// instruction of the K-loop is issued from inline asm - the weight fragments as global_load_dwordx4 (SGPR base + lane
// offset + immediate), the gathered pixel rows as buffer_load_dwordx4 ... lds - so the compiler sees no memory traffic in
// the loop, and the waits are mine: s_waitcnt vmcnt((NST - 2) * GRP) leaves NST - 2 whole stages in flight across the
// barrier.  The pixel source is a raw buffer descriptor over the activation tensor: a lane whose tap leaves the image (or
// whose K index is padding) presents an out-of-range offset and the hardware writes zeros - no zero page, no 64-bit
// select.  Tap and channel of a K-step are wave-uniform (requires ntaps == 1 or Ci % 64 == 0): one scalar byte offset per
// K-step, per DMA a mask test, a select and an add.
// Tile 128 channels x 16*NT pixels (NT = 8, 6, 4: the host picks the pixel tile that fills the 512 block slots best),
// 4 waves of 32 channels x 16*NT pixels, NST-deep ring of pixel stages, NST register sets of weight fragments, 2 blocks
// per CU (<= 256 registers).  Persistent: the grid is min(tiles, 512) and a block walks tiles blockIdx.x + i * gridDim.x.
// diagnostic build (STAMP): per-phase s_memtime deltas of waves 0 and WC/2 of every block, summed into g_stamps[group][8]
// (+ a step count in slot 7); read and reset through mgd_debug_stamps().  Never instantiated on the product path.
__device__ unsigned long long g_stamps[3][8];      // [2]: per tile - tables, prologue issue, first wait, K-loop, drain, epilogue, tiles

template <int WC, int NT, int NST, bool PP = false, bool STAMP = false>
__global__ __launch_bounds__(64 * WC) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_gemm9_kernel(GemmArgs a) {
  constexpr int WP = 1, MT = 2, BNC = 32 * WC, BMP = 16 * NT, NTHR = 64 * WC;
  constexpr int RPR = NTHR / 8, XCH = BMP / RPR;           // rows per DMA round of the block; pixel pieces per wave and stage
  static_assert(BMP % RPR == 0, "pixel tile must be whole DMA rounds");
  constexpr int STAGE = BMP * ROWB;
  constexpr int GRP = 4 + XCH;                              // vector-memory instructions per stage and wave
  static_assert(NST >= 2 && NST <= 5 && (!PP || (NST == 4 && WC == 8)), "ring depth");
  constexpr int DIST = NST - 1 - (PP ? 1 : 0);              // stages between the one being multiplied and the one being issued
  using Epi = GemmEpilogue<WC, WP, MT, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* row_dst = (long long*)(smem + a.aux);
  uint2* row_src = (uint2*)(smem + a.aux + BMP * 8);        // per pixel row: byte offset of its centre pixel, tap-validity mask

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = a.K_pad / BK;
  const int Kreal = a.ntaps * a.Ci;
  const unsigned lds0 = lds_addr(smem) + wave * 1024;

  i32x4 srd;
  {
    const unsigned long long p = (unsigned long long)a.src;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    srd[2] = __builtin_amdgcn_readfirstlane((unsigned)((long long)a.N * a.Hs * a.Ws * a.Ci * 2));
    srd[3] = 0x00020000;
  }
  const unsigned OOB = 0xFFFFFFF0u;                         // beyond any tensor the host admits (< 4 GiB): reads as zeros

  const int rlo = tid >> 3;
  const int kc = (tid & 7) ^ (rlo & 7);
  const int fr = lane & 15, fq = lane >> 4;
  int xro[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) xro[n] = lds_off(n * 16 + fr, fq);
  const unsigned lane16 = lane * 16;

  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // STAMP only
  for (int t = blockIdx.x; t < a.nblk; t += gridDim.x) {
    Epi epi;                                                   // per tile: none of its registers lives across the K-loop
    const unsigned long long tt0 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    const int L = xcd_remap(t, a.nblk);
    const int tc = L % a.tilesC, tp = L / a.tilesC;
    const int co0 = tc * BNC, pix0 = tp * BMP;
    lds_barrier();                                             // the previous tile's epilogue has left the ring and the row tables
    // row tables, one pixel per thread: destination element offset, source byte offset, and which taps stay inside the
    // image - the set of taps with a valid row AND a valid column (a.rowmask / a.colmask: taps by dh + 1 / dw + 1, 9 bits each)
    if (tid < BMP) {
      int m = pix0 + tid;
      long long off = -1;
      unsigned xo = 0, vm = 0;
      if (m < a.M) {
        int hw = a.Hg * a.Wg;
        int n = m / hw, rem = m - n * hw;
        int ig = rem / a.Wg, jg = rem - ig * a.Wg;
        int hd = ig * a.out_stride + a.out_off_h, wd = jg * a.out_stride + a.out_off_w;
        off = (((long long)n * a.Hd + hd) * a.Wd + wd) * a.Co;
        int hs = ig * a.in_stride, ws = jg * a.in_stride;
        xo = (unsigned)(((((long long)n * a.Hs + hs) * a.Ws + ws) * a.Ci) * 2);
        unsigned rsel = 0, csel = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if ((unsigned)(hs + j - 1) < (unsigned)a.Hs) rsel |= (a.rowmask >> (9 * j)) & 0x1FFu;
          if ((unsigned)(ws + j - 1) < (unsigned)a.Ws) csel |= (a.colmask >> (9 * j)) & 0x1FFu;
        }
        vm = rsel & csel;
      }
      row_dst[tid] = off;                                      // read by the epilogue, many barriers from here
      row_src[tid] = make_uint2(xo, vm);
    }
    lds_barrier();
    unsigned xoff[XCH], vmask[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const uint2 rs = row_src[rlo + RPR * i];
      xoff[i] = rs.x + kc * 16;
      vmask[i] = rs.y;
    }
    // wave-uniform K-step state: tap bit, first channel, byte offset of (tap, channel) relative to the centre pixel
    int s_tap = 0, s_c0 = 0, s_k0 = 0;
    auto tap_off = [&](int tp_) {
      const int dh = (int)((a.tapcode >> (4 * tp_)) & 3) - 1;
      const int dw = (int)((a.tapcode >> (4 * tp_ + 2)) & 3) - 1;
      return (dh * a.Ws + dw) * a.Ci * 2;
    };
    int s_toff = tap_off(0);
    const int nchunk = a.Ci / BK;
    int s_chunk = 0;
    // weight fragments of (tile tc, K-step ks): 16 KiB, this wave's four fragments start at wave * 4 KiB
    // (a block of 8 waves spans two consecutive 128-row tiles of the image)
    const char* abase = (const char*)a.wpk + ((size_t)(tc * (WC / 4) + (wave >> 2)) * nk * 16 + (size_t)(wave & 3) * 4) * 1024;
    int s_issued = 0;
    // Issues the next stage.  Past the last K-step it issues a DUMMY stage instead - every pixel lane out of range (no memory
    // traffic, zeros into a ring slot nobody reads again), the weight fragments of the last step once more - so that every
    // K-step of the loop below is the same straight code with the same wait count: no branch between a load and its wait.
    auto issue = [&](int buf, bf16x8 (&f)[2][2]) {
      const bool real = s_issued < nk;
      const unsigned bit = 1u << s_tap;
      const bool kin = real && (s_k0 + kc * 8 < Kreal);       // K padding of the last step reads as zeros
      unsigned vo[XCH];
#pragma unroll
      for (int i = 0; i < XCH; ++i) vo[i] = ((vmask[i] & bit) && kin) ? xoff[i] + (unsigned)s_toff : OOB;
      dma_rows_asm<XCH, RPR * ROWB>(vo, srd, lds0 + buf * STAGE);
      // K block of the packed image: the step number (tap-major) or (tap, chunk) of the chunk-major order (GemmArgs::korder)
      const int wk = (a.korder & 1) ? (real ? s_tap * nchunk + s_chunk : nk - 1) : min(s_issued, nk - 1);
      load_a4_asm(f, lane16, abase + (size_t)wk * 16384);
      ++s_issued;
      if (real) {
        if (a.korder & 1) {
          if (++s_tap == a.ntaps) { s_tap = 0; ++s_chunk; }
          s_toff = tap_off(s_tap) + s_chunk * (BK * 2);
        } else {
          s_k0 += BK;
          s_c0 += BK;
          s_toff += BK * 2;
          if (s_c0 >= a.Ci) { s_c0 -= a.Ci; ++s_tap; s_toff = tap_off(s_tap) + s_c0 * 2; }
        }
      }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 af[NST][2][2];
    // One K-step on register set / ring stage J: wait until all but WS vector-memory operations of this wave have completed,
    // meet the block, read the pixel fragments, issue stage J + NST - 1, multiply (MUL: the padded steps of the last group
    // only keep the pipeline's count).  J is compile-time and the body has no branch between a load and its wait: a register
    // set that is still in flight must never pass through a compiler-made copy (a v_mov of a pending load destination reads
    // garbage) - 'if' ladders around the waits and a 'break' inside the unrolled body both made hipcc merge differently
    // allocated copies of the sets in front of the wait.
    // vmcnt counts LDS-DMA and register loads together, but the two kinds do NOT retire in issue order with respect to
    // each other (measured: 2 DMA + 4 register loads per stage, a wait for all but the 6 youngest let a multiply read a
    // stage whose DMA had not landed - the younger register loads had overtaken it).  Each kind does retire in order, so a
    // count proves "the older stage is complete" as long as the survivors of EITHER kind alone must all be younger: at
    // most min(DMA, register loads) per stage left in flight.
    constexpr int SAFE = XCH < 4 ? XCH : 4;
    constexpr int WS = (DIST - 1) * SAFE;                      // steady state: what may stay in flight behind the wait
    auto mults = [&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const unsigned char* sb = smem + J * STAGE;
      // pixel fragments in groups of NH tiles (all of them up to 128 pixels; halves of a 192-pixel tile, whose 12 + 12
      // fragments beside 96 accumulators and the weight sets would spill); group (0, 0) was read by the caller
      constexpr int NH = NT > 8 ? NT / 2 : NT;
      return [&, sb](bf16x8 (&xf)[NH]) {
#pragma unroll
        for (int h = 0; h < NT / NH; ++h)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            if (h + kk) {
#pragma unroll
              for (int n = 0; n < NH; ++n) xf[n] = *(const bf16x8*)(sb + (xro[h * NH + n] ^ (kk << 6)));
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < NH; ++n)
                acc[m][h * NH + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[J][m][kk], xf[n], acc[m][h * NH + n], 0, 0, 0);
          }
      };
    };
    constexpr int NH = NT > 8 ? NT / 2 : NT;
    // One K-step on register set / ring stage J (compile-time).  No branch lies between a load and its wait: a register
    // set that is still in flight must never pass through a compiler-made copy (a v_mov of a pending load destination
    // reads garbage) - 'if' ladders around the waits and a 'break' inside the unrolled body both made hipcc merge
    // differently allocated copies of the sets in front of the wait.  mul: the padded steps of the last group only keep
    // the pipeline's count.
    auto step = [&](auto jc, bool mul) {
      constexpr int J = decltype(jc)::value;
      constexpr int JN = (J + DIST) % NST;
      wait_a4<WS>(af[J]);
      __builtin_amdgcn_s_barrier();
      const unsigned char* sb = smem + J * STAGE;
      bf16x8 xf[NH];
#pragma unroll
      for (int n = 0; n < NH; ++n) xf[n] = *(const bf16x8*)(sb + xro[n]);
      issue(JN, af[JN]);
      if (mul) mults(jc)(xf);
    };
    // Ping-pong form (8 waves, two per SIMD): waves 4-7 run one phase behind waves 0-3, so that on every SIMD one wave
    // multiplies while its partner reads fragments and issues the next stage - the two never want the matrix pipe at once,
    // and each wave's memory phase hides under the other's MFMAs.  Two barriers per K-step separate the phases:
    //   waves 0-3:  [wait s+1, read s, issue s+2] | [multiply s]              | [wait s+2, read s+1, issue s+3] | ...
    //   waves 4-7:        (one barrier behind)    | [wait s+1, read s, issue] | [multiply s]                    | ...
    // A stage is issued in step s, waited for at the start of step s+1 - a whole step later, by every wave, with nothing
    // else in flight: vmcnt(0), so the order in which LDS-DMA and register loads retire does not matter - and read in step
    // s+2, one phase after the last wave's wait.  A stage is read during three phases (waves 0-3: one, waves 4-7: two, the
    // second k-half inside their multiply phase), hence four ring slots for two stages in flight.
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return STAMP ? __builtin_amdgcn_s_memtime() : 0ull; };
    auto step_pp = [&](auto jc, bool mul) {
      constexpr int J = decltype(jc)::value;
      constexpr int JN = (J + DIST) % NST, J1 = (J + 1) % NST;
      const unsigned char* sb = smem + J * STAGE;
      const unsigned long long t0 = now();
      wait_a4<0>(af[J1]);
      const unsigned long long t1 = now();
      bf16x8 xf[NH];
#pragma unroll
      for (int n = 0; n < NH; ++n) xf[n] = *(const bf16x8*)(sb + xro[n]);
      const unsigned long long t2 = now();
      issue(JN, af[JN]);
      const unsigned long long t3 = now();
      phase_barrier();
      const unsigned long long t4 = now();
      if (mul) mults(jc)(xf);
      const unsigned long long t5 = now();
      phase_barrier();
      if (STAMP) {
        const unsigned long long t6 = now();
        st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += t5 - t4;
        st_acc[5] += t6 - t5; st_acc[7] += 1;
      }
    };
    using J0 = std::integral_constant<int, 0>;
    using J1_ = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    using J3 = std::integral_constant<int, 3>;
    using J4 = std::integral_constant<int, 4>;
    const unsigned long long tt1 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
#pragma unroll
    for (int st = 0; st < DIST; ++st) issue(st, af[st]);        // nk >= DIST (host)
    const unsigned long long tt2 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long tt3 = 0;
    const int full = nk / NST, rem = nk - full * NST;
    auto run_steps = [&](auto&& stp) {
      for (int g = 0; g < full; ++g) {
        stp(J0{}, true);
        stp(J1_{}, true);
        if constexpr (NST >= 3) stp(J2{}, true);
        if constexpr (NST >= 4) stp(J3{}, true);
        if constexpr (NST >= 5) stp(J4{}, true);
      }
      if (rem) {                                               // last, partial group: same steps, the padded ones skip the MFMAs
        stp(J0{}, true);
        stp(J1_{}, rem > 1);
        if constexpr (NST >= 3) stp(J2{}, rem > 2);
        if constexpr (NST >= 4) stp(J3{}, rem > 3);
        if constexpr (NST >= 5) stp(J4{}, false);
      }
    };
    if constexpr (PP) {
      static_assert(!PP || (DIST == 2 && NST == 4), "ping-pong form: two stages in flight, four ring slots");
      wait_a4<0>(af[0]);
      tt3 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
      phase_barrier();
      if (wave >= WC / 2) {
        phase_barrier();
        run_steps(step_pp);
      } else {
        run_steps(step_pp);
        phase_barrier();
      }
    } else {
      run_steps(step);
    }
    if (STAMP) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ph_acc[q] += st_acc[q];
    }
    const unsigned long long tt4 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    // the dummy stages still in flight write zeros into the ring: they must have landed before the epilogue reuses it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 7" ::: "memory");
    lds_barrier();                                             // every wave has read its last fragments: the ring is free
    const unsigned long long tt5 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    epi.template run_grouped<(Epi::EPC > 8 ? Epi::EPC / 2 : Epi::EPC)>(a, acc, smem, row_dst, co0, tid);
    if (STAMP) {
      const unsigned long long tt6 = __builtin_amdgcn_s_memtime();
      tl_acc[0] += tt1 - tt0; tl_acc[1] += tt2 - tt1; tl_acc[2] += tt3 - tt2; tl_acc[3] += tt4 - tt3; tl_acc[4] += tt5 - tt4;
      tl_acc[5] += tt6 - tt5; tl_acc[7] += 1;
    }
  }
  if (STAMP && PP && lane == 0 && (wave == 0 || wave == WC / 2)) {
#pragma unroll
    for (int q = 0; q < 8; ++q) atomicAdd(&g_stamps[wave ? 1 : 0][q], ph_acc[q]);
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) atomicAdd(&g_stamps[2][q], tl_acc[q]);
    }
  }
}

template <int WC, int NT, int NST, bool PP = false, bool STAMP = false>
int launch_gemm9(GemmArgs& a, hipStream_t st, int grid_cap) {
  constexpr int BMP = 16 * NT, BNC = 32 * WC;
  a.tilesC = a.Co_pad / BNC;
  a.nblk = a.tilesC * cdiv(a.M, BMP);
  size_t ring = (size_t)NST * BMP * ROWB;
  size_t epi = (size_t)BMP * (BNC * 2 + 16) + (size_t)WC * 2 * BNC * 4;
  a.aux = (int)(ring > epi ? ring : epi);
  size_t lds = (size_t)a.aux + BMP * 16 + 64;
  auto k = conv_gemm9_kernel<WC, NT, NST, PP, STAMP>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  int grid = a.nblk;
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WC), lds, st, a);
  return 0;
}

// (WC, NT) -> instance; the 8-wave form needs whole DMA rounds of 64 rows: NT = 4, 8, 12
template <int NST>
int launch_gemm9_cfg(int wc, int nt, GemmArgs& a, hipStream_t st, int grid_cap) {
  (void)wc;                                                    // 4 waves (the 8-wave forms are the ping-pong instantiations)
  if (nt == 12) return launch_gemm9<4, 12, NST>(a, st, grid_cap);
  if (nt == 8) return launch_gemm9<4, 8, NST>(a, st, grid_cap);
  if (nt == 6) return launch_gemm9<4, 6, NST>(a, st, grid_cap);
  return launch_gemm9<4, 4, NST>(a, st, grid_cap);
}

// ------------------------------------------------------------------------------------------------
// Latency form of the gather-GEMM (round 3): small-batch inference.  At batch 1 a 608 x 608 forward is 75 DEPENDENT launches
// of 5 - 6 000 pixels each; the kernel trace (tools/trace_timeline.py) shows no idle gaps - the time is the kernels' own:
// 8 - 42 us each for 0.4 - 3.4 GFLOP.  Such a launch has 12 - 90 tiles of 128 x 128 for 256 CUs, and each block walks its
// whole K-loop (up to 72 steps) with one or two stages in flight: a step costs a memory round trip (the weights are touched
// once per forward: they come from HBM / the memory-side cache, 1 - 2 us away) and a CU pulls its tile's share of the
// weights alone - 512 -> 1024 at 19 x 19 streams 9.4 MB through 24 CUs.  What the layer needs is the opposite shape:
//  * every CU pulls weights: K is cut into `splitk` ranges, (tile, range) blocks of 128 channels x 64 pixels - a few
//    hundred blocks of 2 - 12 K-steps;
//  * everything a block will read is in flight at once: weights AND pixel rows come by LDS-DMA into an NST-deep ring (24
//    KiB a stage), so all of a wave's vector-memory operations are of one kind and retire in order - the counted wait
//    s_waitcnt vmcnt((NST - 2) * 6) is exact (with weight fragments loaded to registers the two kinds overtake each other and
//    the count has to assume the worst, see conv_gemm9_kernel) - and nothing in flight lives in a register, so the loop needs
//    no static register sets: a plain loop over the block's steps, dummy stages (all lanes out of range: no traffic, same
//    count) behind the last one.  A wave reads back only the weight fragments it fetched itself;
//  * the ranges meet in the kernel, not in a second launch: a block stores its fp32 partial tile in fragment order (16
//    bytes per lane, coalesced), takes a ticket of its tile (fence, atomic), and the block that draws the last ticket adds all
//    ranges IN RANGE ORDER (deterministic, whoever comes last), applies bias / LeakyReLU / residual and writes bf16.  Tickets
//    are left at zero for the next launch.
// bf16 output without BatchNorm statistics / fused reductions; (tap, channel) of a K-step wave-uniform (ntaps == 1 or
// Ci % 64 == 0).  One block per CU (the ring is 144 KiB).
template <int NT, int NST>
__global__ __launch_bounds__(256) MGD_VGPR_MFMA void conv_gemm11_kernel(GemmArgs a) {
  constexpr int WC = 4, WP = 1, MT = 2, BNC = 128, BMP = 16 * NT, NTHR = 256;
  constexpr int RPR = NTHR / 8, XCH = BMP / RPR;             // 32 rows per DMA round; pixel pieces per wave and stage
  constexpr int PIXB = BMP * ROWB, STAGE = PIXB + 4 * 4096;  // pixel rows, then each wave's four weight fragments
  constexpr int GRP = 4 + XCH, DIST = NST - 1, WS = (DIST - 1) * GRP;
  static_assert(BMP % RPR == 0 && NST >= 2 && WS <= 63, "ring");
  using Epi = GemmEpilogue<WC, WP, MT, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* row_dst = (long long*)(smem + a.aux);
  uint2* row_src = (uint2*)(smem + a.aux + BMP * 8);
  int* last_flag = (int*)(smem + a.aux + BMP * 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = a.K_pad / BK;
  const int Kreal = a.ntaps * a.Ci;
  const int S = a.splitk > 1 ? a.splitk : 1;
  const int tilesP = a.nblk / a.tilesC;
  // blocks that share weights (same channel tile and K range, consecutive pixel tiles) sit on one XCD
  const int L = xcd_remap(blockIdx.x, a.nblk * S);
  const int tp = L % tilesP, rest = L / tilesP;
  const int tc = rest % a.tilesC, split = rest / a.tilesC;
  const int tile = tc * tilesP + tp;
  const int co0 = tc * BNC, pix0 = tp * BMP;
  const int kbase = nk / S, kextra = nk - kbase * S;
  const int k0 = split * kbase + min(split, kextra);
  const int nkb = kbase + (split < kextra ? 1 : 0);           // >= 1 (host: S <= nk)

  auto make_srd = [&](const void* ptr, long long bytes) {
    const unsigned long long p = (unsigned long long)ptr;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000;
    return r;
  };
  const i32x4 srd = make_srd(a.src, (long long)a.N * a.Hs * a.Ws * a.Ci * 2);
  const i32x4 wsrd = make_srd(a.wpk, (long long)a.Co_pad * a.K_pad * 2);
  const unsigned OOB = 0xFFFFFFF0u;

  make_row_tables(a, pix0, tid, BMP, row_dst, row_src);
  lds_barrier();
  // the residual rows of the tile are requested now: their round trip runs under the K-loop instead of in front of the
  // stores (register loads next to the LDS-DMA stream only make the counted waits stricter: each kind retires in order)
  Epi epi;
  epi.prefetch(a, row_dst, co0, tid, false);
  const int rlo = tid >> 3;
  const int kc = (tid & 7) ^ (rlo & 7);
  unsigned xoff[XCH], vmask[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const uint2 rs = row_src[rlo + RPR * i];
    xoff[i] = rs.x + kc * 16;
    vmask[i] = rs.y;
  }
  const int fr = lane & 15, fq = lane >> 4;
  int xro[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) xro[n] = lds_off(n * 16 + fr, fq);

  // wave-uniform K-step state, starting at the block's first step
  auto tap_off = [&](int tp_) {
    const int dh = (int)((a.tapcode >> (4 * tp_)) & 3) - 1;
    const int dw = (int)((a.tapcode >> (4 * tp_ + 2)) & 3) - 1;
    return (dh * a.Ws + dw) * a.Ci * 2;
  };
  int s_k0 = k0 * BK, s_tap = 0, s_c0 = s_k0;
  if (a.ntaps > 1) { s_tap = s_k0 / a.Ci; s_c0 = s_k0 - s_tap * a.Ci; }
  int s_toff = tap_off(s_tap) + s_c0 * 2;
  int s_issued = 0;
  const unsigned ldsp = lds_addr(smem) + wave * 1024;
  const unsigned ldsw = lds_addr(smem) + PIXB + wave * 4096;
  // this wave's four fragments of (channel tile tc, K-step k0): 1 KiB each, lane l's 16 bytes at + 16 l
  const unsigned wlane = (unsigned)((((size_t)tc * nk + k0) * 16 + (size_t)wave * 4) * 1024) + lane * 16;
  auto issue = [&](int buf) {
    const bool real = s_issued < nkb;
    const unsigned bit = 1u << s_tap;
    const bool kin = real && (s_k0 + kc * 8 < Kreal);         // K padding of the last step reads as zeros
    unsigned vo[XCH], wv[4];
#pragma unroll
    for (int i = 0; i < XCH; ++i) vo[i] = ((vmask[i] & bit) && kin) ? xoff[i] + (unsigned)s_toff : OOB;
    const unsigned wo = wlane + (unsigned)s_issued * 16384u;
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = real ? wo + i * 1024 : OOB;
    dma_rows_asm<4, 1024>(wv, wsrd, ldsw + buf * STAGE);
    dma_rows_asm<XCH, RPR * ROWB>(vo, srd, ldsp + buf * STAGE);
    ++s_issued;
    if (real) {
      s_k0 += BK;
      s_c0 += BK;
      s_toff += BK * 2;
      if (s_c0 >= a.Ci) { s_c0 -= a.Ci; ++s_tap; s_toff = tap_off(s_tap) + s_c0 * 2; }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int st = 0; st < DIST; ++st) issue(st);
  int buf = 0, nbuf = DIST;
  for (int s = 0; s < nkb; ++s) {
    wait_vmcnt<WS>();                                          // stage s has landed (this wave's pieces; the barrier: everyone's)
    __builtin_amdgcn_s_barrier();
    const unsigned char* sb = smem + buf * STAGE;
    const unsigned char* wb = sb + PIXB + wave * 4096 + lane * 16;
    bf16x8 xf[NT], wf[MT];
#pragma unroll
    for (int n = 0; n < NT; ++n) xf[n] = *(const bf16x8*)(sb + xro[n]);
#pragma unroll
    for (int m = 0; m < MT; ++m) wf[m] = *(const bf16x8*)(wb + m * 2048);
    issue(nbuf);                                               // into the slot everyone left before this barrier
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk) {
#pragma unroll
        for (int n = 0; n < NT; ++n) xf[n] = *(const bf16x8*)(sb + (xro[n] ^ 64));
#pragma unroll
        for (int m = 0; m < MT; ++m) wf[m] = *(const bf16x8*)(wb + m * 2048 + 1024);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[n], acc[m][n], 0, 0, 0);
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
    nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
  }
  // the dummy stages still in flight write zeros into the ring: they must have landed before the epilogue reuses it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_nop 7" ::: "memory");
  lds_barrier();

  if (S > 1) {
    // Partial tiles cross XCDs (each has its own L2, not coherent with the others for ordinary device memory inside a
    // kernel): the workspace is UNCACHED device memory (mgd_latency_workspace) and is stored / loaded at system scope, ordered
    // by the stores' completion, the block barrier and the ticket atomic.  (Release / acquire FENCES make ordinary memory work
    // too - buffer_wbl2 + buffer_inv of the whole L2 per block: measured 80 - 100 us a launch; scope bits alone on ordinary
    // memory did not: ranges were read stale.)
    constexpr int TILE_F4 = BNC * BMP / 4;                     // f32x4 elements per partial tile
    f32x4* mine = (f32x4*)a.partial + ((size_t)split * a.nblk + tile) * TILE_F4 + tid;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        // (s_nop 1: a store of more than 64 bits reads its data registers for two more wait states; hipcc pads that for
        // its own stores but not behind an asm statement, and re-filled v[4:7] from the accumulators right behind each store)
        store_sys_b128(mine + (m * NT + n) * NTHR, acc[m][n]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // written through before the ticket
    __syncthreads();
    if (tid == 0) {
      const unsigned t = atomicAdd(a.tickets + tile, 1u);
      *last_flag = t == (unsigned)(S - 1);
    }
    __syncthreads();
    if (!*last_flag) return;
    const f32x4* all = (const f32x4*)a.partial + (size_t)tile * TILE_F4 + tid;
    const size_t rstride = (size_t)a.nblk * TILE_F4;
    // in range order (the sum does not depend on who came last); four whole ranges (32 loads a lane) in flight at a time:
    // every round is a full trip to memory, and the chain store -> ticket -> loads is what K ranges cost (~6 us)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < S; r0 += 4) {
      f32x4 v[4][MT * NT];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4* pr = all + (size_t)min(r0 + r, S - 1) * rstride;
#pragma unroll
        for (int q = 0; q < MT * NT; ++q) load_sys_b128(v[r][q], pr + q * NTHR);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < MT * NT; ++q) {
          asm volatile("" : "+v"(v[r][q]));                    // consumers stay behind the wait
          if (r0 + r < S) acc[q / NT][q % NT] += v[r][q];
        }
      }
    }
    if (tid == 0) a.tickets[tile] = 0u;
  }
  epi.template run<false, false>(a, acc, smem, row_dst, co0, tid);
}

template <int NT, int NST>
int launch_gemm11(GemmArgs& a, hipStream_t st) {
  constexpr int BMP = 16 * NT, STAGE = BMP * ROWB + 4 * 4096;
  size_t ring = (size_t)NST * STAGE;
  size_t epi = (size_t)BMP * (128 * 2 + 16) + (size_t)4 * 2 * 128 * 4;
  a.aux = (int)(ring > epi ? ring : epi);
  size_t lds = (size_t)a.aux + BMP * 16 + 64;
  auto k = conv_gemm11_kernel<NT, NST>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(a.nblk * (a.splitk > 1 ? a.splitk : 1)), dim3(256), lds, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// "Patch" form of the 3x3 gather-GEMM for the thin early layers (CI, CO <= 64: all nine taps of the weights fit in
// LDS).  The v2 kernel walks K = 9*CI in 64-deep steps and is bound by one LDS-DMA round trip per step (1.63 us), i.e.
// by nothing the layer itself needs: the input is re-read nine times through L2.  Here a persistent block keeps the
// weights in LDS as ready-made A fragments (9 x CI/32 x CO/16 KiB), and per tile of 8 x 16 output pixels (4 x 16 at
// stride 2) stages the haloed input patch ONCE (plain 16-byte loads, padded pixel pitch -> conflict-free
// ds_read_b128), then runs the nine taps out of LDS.  Same epilogue as the other forms (GemmEpilogue).
template <int CI, int CO, int S>
__global__ __launch_bounds__(256) MGD_VGPR_MFMA void conv_patch_kernel(GemmArgs a) {
  constexpr int MT = CO / 16, KS = CI / 32, NT = 2 / S;
  constexpr int TR = 8 / S, TC = 16;                 // output tile: 8 x 16 pixels (4 x 16 at stride 2), NT rows per wave
  constexpr int BMP = TR * TC;
  using Epi = GemmEpilogue<1, 4, MT, NT>;
  constexpr int PRW = (TR - 1) * S + 3, PCL = (TC - 1) * S + 3;     // patch rows / cols
  constexpr int PB = CI * 2, PITCH = PB + 16;        // bytes per patch pixel, padded
  constexpr int CPP = PB / 16;                       // 16-byte chunks per pixel
  constexpr int NCH = PRW * PCL * CPP;               // chunks per patch
  constexpr int PL = (NCH + 255) / 256;              // chunks per thread
  constexpr int WBYTES = 9 * KS * MT * 1024;
  constexpr int PBYTES = ((PRW * PCL * PITCH + 15) / 16) * 16;
  constexpr int EBYTES = BMP * (CO * 2 + 16) + 4 * 2 * CO * 4;
  constexpr int UBYTES = PBYTES > EBYTES ? PBYTES : EBYTES;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                          // A fragments: [(t*KS + ks)*MT + m][lane] x 16 B
  unsigned char* pt = smem + WBYTES;                 // patch ...
  unsigned char* el = pt;                            // ... reused by the epilogue tile + partial sums
  long long* row_dst = (long long*)(smem + WBYTES + UBYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // weights -> LDS once per block: lane (fr = co, fq = ci group) of fragment (t, ks, m) = 16 contiguous bytes of the
  // packed image row co, at k = t*CI + ks*32 + fq*8
  for (int f = wave; f < 9 * KS * MT; f += 4) {
    int m = f % MT, ks = (f / MT) % KS, t = f / (MT * KS);
    *(uint4*)(wl + f * 1024 + lane * 16) =
        *(const uint4*)(a.wpk + (long long)(m * 16 + fr) * a.K_pad + t * CI + ks * 32 + fq * 8);
  }
  const int tilesW = (a.Wg + TC - 1) / TC, tilesH = (a.Hg + TR - 1) / TR;
  const int ntiles = a.N * tilesH * tilesW;
  int pbase[NT];                                      // my pixel fragments: tile rows NT*wave .. (slot = row*16 + col)
#pragma unroll
  for (int n = 0; n < NT; ++n) pbase[n] = (((wave * NT + n) * S) * PCL + fr * S) * PITCH + fq * 16;
  // patch chunks of this thread: (patch pixel, 16-byte chunk) -> LDS offset, constant over tiles
  int pp_r[PL], pp_c[PL], pp_o[PL], pp_ch[PL];
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    int i = tid + j * 256;
    int ch = i % CPP, pp = i / CPP;
    pp_r[j] = i < NCH ? pp / PCL : -100000;
    pp_c[j] = pp % PCL;
    pp_o[j] = pp * PITCH + ch * 16;
    pp_ch[j] = ch * 8;
  }
  uint4 pre[PL];
  auto fetch = [&](int t) {                           // haloed input patch of tile t -> registers
    int b = t;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      int hh = th * TR * S - 1 + pp_r[j], ww = tw * TC * S - 1 + pp_c[j];
      pre[j] = make_uint4(0, 0, 0, 0);
      if ((unsigned)hh < (unsigned)a.Hs && (unsigned)ww < (unsigned)a.Ws)
        pre[j] = *(const uint4*)(a.src + (((long long)n * a.Hs + hh) * a.Ws + ww) * CI + pp_ch[j]);
    }
  };
  // per-channel sums (BatchNorm statistics / fused BN-backward sums) stay in registers over all tiles of the block
  Epi epi;
  epi.init_deferred();
  int t = blockIdx.x;
  if (t < ntiles) fetch(t);
  for (; t < ntiles; t += gridDim.x) {
    int b = t;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
    const int h0 = th * TR, w0 = tw * TC;
    __syncthreads();                                  // previous tile's epilogue is done with the shared region
    if (tid < BMP) {
      int r = tid >> 4, c = tid & 15;
      long long off = -1;
      if (h0 + r < a.Hg && w0 + c < a.Wg) off = (((long long)n * a.Hd + h0 + r) * a.Wd + w0 + c) * a.Co;
      row_dst[tid] = off;
    }
#pragma unroll
    for (int j = 0; j < PL; ++j)
      if (pp_r[j] >= 0) *(uint4*)(pt + pp_o[j]) = pre[j];
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);      // next tile's patch flies under this tile's MFMAs
    epi.prefetch(a, row_dst, 0, tid, false);                     // and so do the epilogue's HBM operands
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) acc[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int toff = ((tp / 3) * PCL + (tp % 3)) * PITCH;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wf[MT], xf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) wf[m] = *(const bf16x8*)(wl + ((tp * KS + ks) * MT + m) * 1024 + lane * 16);
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) xf[nn] = *(const bf16x8*)(pt + pbase[nn] + toff + ks * 64);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int nn = 0; nn < NT; ++nn)
            acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[nn], acc[m][nn], 0, 0, 0);
      }
    }
    __syncthreads();                                  // every wave is done with the patch: the epilogue reuses its LDS
    epi.template run<true>(a, acc, el, row_dst, 0, tid);
  }
  __syncthreads();
  epi.flush(a, el, 0, tid);
}

// ------------------------------------------------------------------------------------------------
// Stride-2 data gradient of the first down-sampling layer (dy: 64 channels at 304 x 304 -> dx: 32 channels at 608 x
// 608) in patch form, all four output-parity classes in one launch.  The generic path runs the four classes as four
// gather-GEMM launches with K = 64..256 (1-4 K-steps per block, each launch re-reading dy): 402 us against a 125 us
// HBM floor.  Here a persistent block keeps the nine transposed taps in LDS, stages the (8+1) x (16+1) dy patch of a
// tile once, and accumulates the four classes side by side (tap kh feeds output rows of parity kh != 1, from dy row
// i + (kh == 0)); each class then leaves through the common epilogue (addend, fused BatchNorm-backward sums).
struct Dgrad2Args {
  GemmArgs g;                 // dst / addend / bn_* / Co (= dx channels) / stats_replicas for the epilogue; src = dy
  const bf16_t* wpk[4];       // class images [(ph, pw)] : [32 rows][ntaps_c * 64]
  int K_pad[4];
  int Ho, Wo, H, W;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv_patch_dgrad2_kernel(Dgrad2Args d) {
  constexpr int KS = CIN / 32, MT = COUT / 16, NT = 2;
  constexpr int TR = 8, TC = 16, BMP = TR * TC;
  using Epi = GemmEpilogue<1, 4, MT, NT>;
  constexpr int PRW = TR + 1, PCL = TC + 1;
  constexpr int PB = CIN * 2, PITCH = PB + 16, CPP = PB / 16;
  constexpr int NCH = PRW * PCL * CPP, PL = (NCH + 255) / 256;
  constexpr int WBYTES = 9 * KS * MT * 1024;
  constexpr int PBYTES = ((PRW * PCL * PITCH + 15) / 16) * 16;
  constexpr int EBYTES = BMP * (COUT * 2 + 16) + 4 * 2 * COUT * 4;
  const GemmArgs& a = d.g;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;
  unsigned char* pt = smem + WBYTES;                 // dy patch (kept over both output-row parities)
  unsigned char* el = pt + PBYTES;                   // epilogue tile + partial sums
  long long* row_dst = (long long*)(el + EBYTES);      // [2][BMP]: the two column parities of the current row parity

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // transposed taps -> LDS as A fragments [(q*KS + ks)*MT + m], q = kh*3 + kw; tap (kh, kw) lives in the image of class
  // (ph, pw) = (kh != 1, kw != 1) at slot (kh == 2) * (pw ? 2 : 1) + (kw == 2)
  for (int f = wave; f < 9 * KS * MT; f += 4) {
    int m = f % MT, ks = (f / MT) % KS, q = f / (MT * KS);
    int kh = q / 3, kw = q - kh * 3;
    int ph = kh != 1, pw = kw != 1;
    int slot = (kh == 2 ? 1 : 0) * (pw ? 2 : 1) + (kw == 2 ? 1 : 0);
    int c = ph * 2 + pw;
    *(uint4*)(wl + f * 1024 + lane * 16) =
        *(const uint4*)(d.wpk[c] + (long long)(m * 16 + fr) * d.K_pad[c] + slot * CIN + ks * 32 + fq * 8);
  }
  const int tilesW = (d.Wo + TC - 1) / TC, tilesH = (d.Ho + TR - 1) / TR;
  const int ntiles = a.N * tilesH * tilesW;
  int pbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) pbase[n] = ((wave * NT + n) * PCL + fr) * PITCH + fq * 16;
  int pp_r[PL], pp_c[PL], pp_o[PL], pp_ch[PL];
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    int i = tid + j * 256;
    int ch = i % CPP, pp = i / CPP;
    pp_r[j] = i < NCH ? pp / PCL : 100000;
    pp_c[j] = pp % PCL;
    pp_o[j] = pp * PITCH + ch * 16;
    pp_ch[j] = ch * 8;
  }
  uint4 pre[PL];
  auto fetch = [&](int t) {
    int b = t;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      int hh = th * TR + pp_r[j], ww = tw * TC + pp_c[j];
      pre[j] = make_uint4(0, 0, 0, 0);
      if (hh < d.Ho && ww < d.Wo)
        pre[j] = *(const uint4*)(a.src + (((long long)n * d.Ho + hh) * d.Wo + ww) * CIN + pp_ch[j]);
    }
  };
  Epi epi;                                           // fused BN-backward sums carried over all tiles, one flush per block
  epi.init_deferred();
  int t = blockIdx.x;
  if (t < ntiles) fetch(t);
  for (; t < ntiles; t += gridDim.x) {
    int b = t;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH;
    const int n = b / tilesH;
    const int h0 = th * TR, w0 = tw * TC;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PL; ++j)
      if (pp_r[j] < 100000) *(uint4*)(pt + pp_o[j]) = pre[j];
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    // output rows of parity ph take the taps kh = 1 (ph = 0, dy row i) or kh = 0, 2 (ph = 1, dy rows i+1, i); the two
    // column parities of a row parity are accumulated side by side and flushed before the other row parity starts
    // (all four at once needed 254 VGPRs = one wave per SIMD)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      __syncthreads();                                // previous epilogues are done with row_dst and the tile
      {
        int pw = tid >> 7, sl = tid & 127, r = sl >> 4, cc = sl & 15;
        long long off = -1;
        if (h0 + r < d.Ho && w0 + cc < d.Wo)
          off = (((long long)n * d.H + 2 * (h0 + r) + ph) * d.W + 2 * (w0 + cc) + pw) * a.Co;
        row_dst[tid] = off;
      }
      __syncthreads();
      epi.prefetch(a, row_dst, 0, tid, false);        // HBM operands of the first epilogue fly under the MFMAs
      f32x4 acc[2][MT][NT];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int nn = 0; nn < NT; ++nn) acc[c][m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int kh = q / 3, kw = q % 3;
        if ((kh != 1) != (ph == 1)) continue;
        const int c = kw != 1;
        const int toff = ((kh == 0) * PCL + (kw == 0)) * PITCH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 wf[MT], xf[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m) wf[m] = *(const bf16x8*)(wl + ((q * KS + ks) * MT + m) * 1024 + lane * 16);
#pragma unroll
          for (int nn = 0; nn < NT; ++nn) xf[nn] = *(const bf16x8*)(pt + pbase[nn] + toff + ks * 64);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
              acc[c][m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[nn], acc[c][m][nn], 0, 0, 0);
        }
      }
      epi.template run<true>(a, acc[0], el, row_dst, 0, tid);
      epi.prefetch(a, row_dst + BMP, 0, tid, false);  // (a second resident copy of the BN parameters spilled registers)
      __syncthreads();                                // the tile is free again
      epi.template run<true>(a, acc[1], el, row_dst + BMP, 0, tid);
    }
  }
  __syncthreads();
  epi.flush(a, el, 0, tid);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient.  dW[co][tap][ci] += sum_p dy[p][co] * src[p (+) tap][ci]
// LDS tiles are [pixel][channel] exactly as they come from NHWC memory; MFMA fragments need
// [channel][8 consecutive pixels], fetched with ds_read_b64_tr_b16 (4 pixel rows x 16 channels per
// 16-lane group, delivered column-major).
template <int WC, int WP, int MT, int NT, int NST>
int launch_gemm2(GemmArgs& a, hipStream_t st) {
  constexpr int BNC = WC * MT * 16, BMP = WP * NT * 16;
  a.tilesC = a.Co_pad / BNC;
  int tilesP = cdiv(a.M, BMP);
  a.nblk = a.tilesC * tilesP;
  size_t ring = (size_t)NST * (BNC + BMP) * ROWB;
  size_t epi = (size_t)BMP * (BNC * 4 + 16);       // must match AUX in the kernel
  ring = (ring > epi ? ring : epi) + (size_t)BMP * 16 + 1024;
  auto k = conv_gemm2_kernel<WC, WP, MT, NT, NST>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(a.nblk), dim3(64 * WC * WP), ring, st, a);
  return 0;
}


template <int WC, int WP, int MT, int NT, int NS>
int launch_gemm6(GemmArgs& a, hipStream_t st) {
  constexpr int BNC = WC * MT * 16, BMP = WP * NT * 16;
  a.tilesC = a.Co_pad / BNC;
  a.nblk = a.tilesC * cdiv(a.M, BMP);
  size_t ring = (size_t)NS * (BNC + BMP) * ROWB;
  size_t epi = (size_t)BMP * (BNC * 4 + 16);
  size_t lds = (ring > epi ? ring : epi) + (size_t)BMP * 16 + 64;
  auto k = conv_gemm6_kernel<WC, WP, MT, NT, NS>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(a.nblk), dim3(512), lds, st, a);
  return 0;
}

template <int CI, int CO, int S>
int launch_patch(GemmArgs& a, hipStream_t st) {
  constexpr int MT = CO / 16, KS = CI / 32;
  constexpr int TR = 8 / S, TC = 16;
  constexpr int PRW = (TR - 1) * S + 3, PCL = (TC - 1) * S + 3, PITCH = CI * 2 + 16;
  size_t pbytes = (size_t)((PRW * PCL * PITCH + 15) / 16) * 16, ebytes = (size_t)(TR * TC) * (CO * 2 + 16) + (size_t)4 * 2 * CO * 4;
  size_t lds = (size_t)9 * KS * MT * 1024 + (pbytes > ebytes ? pbytes : ebytes) + (size_t)(TR * TC) * 8;
  a.tilesC = 1;
  long long tiles = (long long)a.N * cdiv(a.Hg, TR) * cdiv(a.Wg, TC);
  a.nblk = (int)tiles;
  auto k = conv_patch_kernel<CI, CO, S>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  int per_cu = (int)(160 * 1024 / (lds + 512));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  long long grid = 256ll * per_cu;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, a);
  return 0;
}

}  // namespace

// The counted-pipeline form's tile shape: a launch is modelled as rounds x K-steps x bytes per step of a CU's resident blocks
// (the K-loop is bound by the CU's vector-memory path, ~25 B/clk delivered); wider tiles move fewer bytes per FLOP but
// quantise worse on the 512 block slots.  nt_forced != 0: that pixel tile.
static int counted_tile(const mgd_conv_desc* d, int M, int nt_forced) {
  int nt = 8;
  double best = 1e30;
  const int cn[4] = {12, 8, 6, 4};
  for (int c = 0; c < 4; ++c) {
    if (nt_forced && cn[c] != nt_forced) continue;
    const long long tiles = (long long)(d->Co_pad / 128) * cdiv(M, cn[c] * 16);
    const double rounds = (double)((tiles + 511) / 512);
    const double bytes = 2 * (4 * 4096.0 + cn[c] * 16 * 128.0);               // per K-step and CU (two blocks)
    const double mfma = 2.0 * cn[c] * 2 * 16 * 2 * 1.6;                        // cycles per K-step and SIMD at 2 waves, derated
    const double cost = rounds * (bytes / 25.0 > mfma ? bytes / 25.0 : mfma) + rounds * 600.0;   // + epilogue / prologue per round
    if (cost < best - 1e-9) { best = cost; nt = cn[c]; }
  }
  return nt;
}

// validation of a descriptor and its translation into the kernels' argument block
static int conv_args(const mgd_conv_desc* d, GemmArgs& a) {
  MGD_REQUIRE(d && d->src && d->wpk && d->dst, "conv: null pointer");
  MGD_REQUIRE(d->Ci % 8 == 0 && d->Ci >= 8, "conv: Ci=%d must be a multiple of 8", d->Ci);
  MGD_REQUIRE(d->Co % 8 == 0, "conv: Co=%d must be a multiple of 8", d->Co);
  MGD_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9, "conv: ntaps=%d", d->ntaps);
  MGD_REQUIRE(d->K_pad % BK == 0 && d->K_pad >= d->ntaps * d->Ci, "conv: K_pad=%d too small/unaligned", d->K_pad);
  MGD_REQUIRE(d->Co_pad >= d->Co && d->Co_pad % 32 == 0, "conv: Co_pad=%d", d->Co_pad);
  MGD_REQUIRE(!(d->stats && d->dst_f32), "conv: stats epilogue needs bf16 output");
  MGD_REQUIRE(!d->stats || d->stats_replicas >= 1, "conv: stats_replicas");
  MGD_REQUIRE((d->Hg - 1) * d->out_stride + d->out_off_h < d->Hd && (d->Wg - 1) * d->out_stride + d->out_off_w < d->Wd,
              "conv: iteration grid exceeds destination");
  MGD_REQUIRE((long long)d->N * d->Hg * d->Wg < (1ll << 31), "conv: too many pixels");
  MGD_REQUIRE(d->splitk <= 1 || d->latency, "conv: K ranges (splitk) exist in the latency form only");
  bool ok;
  a.src = (const bf16_t*)d->src; a.wpk = (const bf16_t*)d->wpk; a.dst = d->dst; a.bias = d->bias;
  a.addend = (const bf16_t*)d->addend; a.stats = d->stats;
  a.N = d->N; a.Hs = d->Hs; a.Ws = d->Ws; a.Ci = d->Ci; a.Hg = d->Hg; a.Wg = d->Wg; a.Hd = d->Hd; a.Wd = d->Wd;
  a.Co = d->Co; a.in_stride = d->in_stride; a.out_stride = d->out_stride; a.out_off_h = d->out_off_h;
  a.out_off_w = d->out_off_w; a.ntaps = d->ntaps;
  a.tapcode = make_tapcode(d->ntaps, d->dh, d->dw, &ok);
  MGD_REQUIRE(ok, "conv: tap offsets must lie in [-1,1]");
  a.K_pad = d->K_pad; a.Co_pad = d->Co_pad; a.dst_f32 = d->dst_f32; a.stats_replicas = d->stats_replicas;
  a.M = d->N * d->Hg * d->Wg;
  a.rowmask = a.colmask = 0;                      // taps by row / column offset: the row tables' validity masks
  for (int t = 0; t < d->ntaps; ++t) {
    a.rowmask |= 1u << (9 * (d->dh[t] + 1) + t);
    a.colmask |= 1u << (9 * (d->dw[t] + 1) + t);
  }
  a.splitk = 0; a.slab_elems = 0; a.partial = nullptr; a.tickets = nullptr;
  a.dbg = MGD_DIAG_FLAGS;
  a.bn_y = (const bf16_t*)d->bn_y; a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.bn_mean = d->bn_mean;
  a.bn_invstd = d->bn_invstd; a.bn_sums = d->bn_sums; a.bn_slope = d->bn_slope;
  a.act_slope = d->act_slope;
  MGD_REQUIRE(d->act_slope == 0.f || (!d->dst_f32 && !d->stats && !d->bn_y), "conv: act_slope is for bf16 inference output (no stats / fused reductions)");
  MGD_REQUIRE(!d->bn_y || (d->bn_scale && d->bn_shift && d->bn_mean && d->bn_invstd && d->bn_sums && d->stats_replicas >= 1),
              "conv: fused BN-backward reduction needs scale/shift/mean/invstd/sums and stats_replicas");
  // 32-bit byte offsets (SGPR base + VGPR offset) address the source tensor and the packed weights in every kernel below
  MGD_REQUIRE((long long)d->N * d->Hs * d->Ws * d->Ci * 2 < (1ll << 32), "conv: source tensor exceeds 32-bit byte addressing (N*Hs*Ws*Ci*2 >= 4 GiB)");
  MGD_REQUIRE((long long)d->Co_pad * d->K_pad * 2 < (1ll << 32), "conv: packed weights exceed 32-bit byte addressing");

  a.ncls = 0;
  a.korder = 0;
  return MGD_OK;
}

extern "C" int mgd_conv_gather_gemm(const mgd_conv_desc* d, void* stream) {
  GemmArgs a;
  { const int rc = conv_args(d, a); if (rc != MGD_OK) return rc; }
  hipStream_t st = (hipStream_t)stream;
  const int nk = d->K_pad / BK;
  const int form = d->form;                       // 0: the rules below; anything else: that form or MGD_EINVAL
  const bool frag = d->Co_pad % 128 == 0;         // the packed image is in fragment order (packed_elem)
  const bool uni = d->ntaps == 1 || d->Ci % 64 == 0;       // (tap, channel) of a K-step is wave-uniform
  // K order (GemmArgs::korder).  Chunk-major is the default of the 128 x 128 form only: three blocks per CU on the large maps
  // are what overflows the L2 in tap-major order (256 -> 128 data gradient at 76 x 76: 61 -> 57 us); the counted / ping-pong /
  // phased forms run the small maps, where it costs 1 - 4 % (a tap_off per K-step) and saves nothing.  A forced form may
  // ask for the other order: form_arg + 256 = tap-major, + 512 = chunk-major (measurement, tools/bench_forms.py).
  const int farg = d->form_arg & 255;
  const bool kmaj_ok = d->Ci % BK == 0 && d->ntaps > 1 && d->K_pad == d->ntaps * d->Ci;
  const int kmaj_small = (kmaj_ok && form != MGD_CONV_AUTO && (d->form_arg & 512)) ? 1 : 0;
  const int kmaj_large = (kmaj_ok && !(form != MGD_CONV_AUTO && (d->form_arg & 256))) ? 1 : 0;
  a.korder = kmaj_small;
  const long long tiles128 = (long long)(d->Co_pad / 128) * cdiv(a.M, 128);

  // ---- latency form (a request of its own: mgd_conv_desc.latency) ----
  if (d->latency) {
    MGD_REQUIRE(form == MGD_CONV_AUTO, "conv: latency and a forced form exclude each other");
    MGD_REQUIRE(frag && !d->dst_f32 && !d->stats && !d->bn_y && uni,
                "conv: the latency form needs 128-channel weight tiles, bf16 output without statistics / fused reductions and wave-uniform K-steps (ntaps == 1 or Ci %% 64 == 0)");
    const int S = d->splitk > 1 ? d->splitk : 1;
    MGD_REQUIRE(S <= nk, "conv: more K ranges than K-steps");
    a.tilesC = d->Co_pad / 128;
    a.nblk = a.tilesC * cdiv(a.M, 64);
    a.splitk = S;
    if (S > 1) {
      MGD_REQUIRE(a.nblk <= 4096, "conv: latency form with K ranges: at most 4096 tiles (got %d)", a.nblk);
      MGD_REQUIRE(d->partial && d->partial_bytes >= mgd_latency_workspace_size(a.nblk, S), "conv: latency-form workspace too small");
      a.tickets = (unsigned*)d->partial;
      a.partial = d->partial + 4096;
    }
    launch_gemm11<4, 6>(a, st);
    MGD_CHECK_LAUNCH("conv_gather_gemm(latency form)");
    return MGD_OK;
  }

  // ---- patch form: the thin early layers (3x3 in the standard tap order, 32 -> 64 at stride 1 / 2, 64 -> 32 at stride 1) ----
  {
    bool std9p = d->ntaps == 9 && d->out_stride == 1 && d->out_off_h == 0 && d->out_off_w == 0 && d->Hd == d->Hg &&
                 d->Wd == d->Wg && !d->dst_f32 && d->K_pad >= 9 * d->Ci &&
                 (d->in_stride == 1 || d->in_stride == 2) && d->Hs == d->Hg * d->in_stride && d->Ws == d->Wg * d->in_stride;
    for (int t = 0; t < 9 && std9p; ++t) std9p = d->dh[t] == t / 3 - 1 && d->dw[t] == t % 3 - 1;
    const bool p3264 = std9p && d->Ci == 32 && d->Co == 64 && d->Co_pad == 64;
    const bool p6432 = std9p && d->Ci == 64 && d->Co == 32 && d->Co_pad == 32 && d->in_stride == 1;
    MGD_REQUIRE(form != MGD_CONV_PATCH || p3264 || p6432, "conv: the patch form is built for 3x3 32 -> 64 (stride 1 / 2) and 64 -> 32 (stride 1)");
    if ((form == MGD_CONV_AUTO || form == MGD_CONV_PATCH) && (p3264 || p6432)) {
      if (p6432) launch_patch<64, 32, 1>(a, st);             // the stride-1 data gradient of a 32 -> 64 layer
      else if (d->in_stride == 1) launch_patch<32, 64, 1>(a, st);
      else launch_patch<32, 64, 2>(a, st);
      MGD_CHECK_LAUNCH("conv_gather_gemm(patch)");
      return MGD_OK;
    }
  }

  // ---- thin channel tiles (row-major packed image) ----
  if (!frag) {
    MGD_REQUIRE(form == MGD_CONV_AUTO || form == MGD_CONV_THIN, "conv: Co_pad=%d (not a multiple of 128) runs on the thin-tile form only", d->Co_pad);
    if (d->Co_pad % 64 == 0) launch_gemm2<1, 4, 4, 2, 2>(a, st); else launch_gemm2<1, 4, 2, 2, 2>(a, st);
    MGD_CHECK_LAUNCH("conv_gather_gemm");
    return MGD_OK;
  }
  MGD_REQUIRE(form != MGD_CONV_THIN, "conv: the thin-tile form reads row-major packed images (Co_pad %% 128 != 0)");

  // ---- phased 8-wave form (conv_gemm12.hip).  Own rule, measured at 608 x 608, batch 16 (tools/bench_forms.py): with
  // 256 x 192 tiles it wins where one round of tiles fills the CUs and the K-loop is long enough to pay its prologue /
  // epilogue: 256 -> 512 at 38 x 38 and its stride-2 entry 55 -> 50 us; it ties on 18-step tiles (128 -> 256 at 76 x 76) and
  // loses wherever tiles quantise badly on 256 CUs. ----
  {
    const bool can = !d->dst_f32 && d->Ci % 64 == 0 && d->K_pad == d->ntaps * d->Ci;
    bool want = form == MGD_CONV_PHASED;
    int shape = farg;
    if (form == MGD_CONV_AUTO && can && d->Co_pad % 256 == 0 && d->ntaps > 1 && nk >= 36) {
      const long long t192 = (long long)(d->Co_pad / 256) * cdiv(a.M, 192);
      const long long rounds = (t192 + 255) / 256;
      if (rounds == 1 && t192 >= 230) { want = true; shape = 1; }
    }
    if (want) {
      a.partial = d->partial;                     // (diagnostic library: stamp buffer)
      MGD_REQUIRE(can && mgd::launch_gemm12(a, shape, 0, st) == MGD_OK,
                  "conv: the phased form needs bf16 output, Ci %% 64 == 0, K_pad == ntaps * Ci and Co_pad a multiple of its channel tile (shape %d)", shape);
      MGD_CHECK_LAUNCH("conv_gather_gemm(phased, 8 waves)");
      return MGD_OK;
    }
  }

  // ---- counted pipeline (conv_gemm9_kernel): bf16 output, wave-uniform K-steps.  Rules from tools/bench_conv.py at 608 x 608,
  // batch 16, each launch alone:
  //  * long K-loops on 256-channel blocks (nk >= 64, or nk >= 36 with at most 512 tiles of 128 x 128): ping-pong form -
  //    512 -> 1024 at 19 x 19 64 -> 56 us, its stride-2 entry 66 -> 57 us, the data gradient of 256 -> 512 at 38 x 38 65 -> 55 us;
  //  * launches of 128 - 512 tiles with short K-loops (the head's 3x3 convolutions): 4-wave form, three stages -
  //    128 -> 256 at 38 x 38 25.7 -> 21.9 us, 128 -> 352 33.6 -> 30.9 us, 256 -> 704 at 19 x 19 35.7 -> 31.1 us;
  //  * (both rules need enough tiles to occupy the chip: at batch 1 the 19 x 19 / 38 x 38 layers have 12 - 48 tiles) ----
  {
    const bool can9 = !d->dst_f32 && d->ntaps > 1 && d->Ci % 64 == 0 && nk >= 2;
    bool pp = form == MGD_CONV_PINGPONG, c4 = form == MGD_CONV_COUNTED;
    MGD_REQUIRE(!(pp || c4) || can9, "conv: the counted-pipeline forms need bf16 output, ntaps > 1, Ci %% 64 == 0 and at least two K-steps");
    MGD_REQUIRE(!pp || (d->Co_pad % 256 == 0 && nk >= 3), "conv: the ping-pong form needs Co_pad %% 256 == 0 and at least three K-steps");
    if (form == MGD_CONV_AUTO && can9 && nk >= 4) {
      if (d->Co_pad % 256 == 0 && tiles128 >= 256 && (nk >= 64 || (nk >= 36 && tiles128 <= 512))) pp = true;
      else if (tiles128 >= 128 && tiles128 <= 512 && nk <= 36) c4 = true;
    }
    if (pp) {
      const int nt = farg == 12 ? 12 : 8;
#ifdef MGD_DIAG
      if (nt == 8 && (a.dbg & 4096)) { launch_gemm9<8, 8, 4, true, true>(a, st, 256); MGD_CHECK_LAUNCH("conv_gather_gemm(counted pipeline, ping-pong)"); return MGD_OK; }
#endif
      if (nt == 8) launch_gemm9<8, 8, 4, true>(a, st, 256); else launch_gemm9<8, 12, 4, true>(a, st, 256);
      MGD_CHECK_LAUNCH("conv_gather_gemm(counted pipeline, ping-pong)");
      return MGD_OK;
    }
    if (c4) {
      const int fa = form == MGD_CONV_COUNTED ? farg : 0;
      MGD_REQUIRE(fa == 0 || fa == 12 || fa == 8 || fa == 6 || fa == 4, "conv: counted pipeline: pixel tile / 16 must be 12, 8, 6 or 4");
      const int nt = counted_tile(d, a.M, fa);
      // ring depth: the one (3, 4 or 2) that pads the K-loop least (the loop runs whole groups of NST steps)
      int nst = 0, bestpad = 1 << 30;
      const int cand[3] = {3, 4, 2};
      for (int c = 0; c < 3; ++c) {
        const int pad = (nk + cand[c] - 1) / cand[c] * cand[c] - nk;
        if (nk >= cand[c] - 1 && pad < bestpad) { bestpad = pad; nst = cand[c]; }
      }
      if (nst == 4 && (nk < 3 || nt == 12)) nst = 3;         // four stages of 192 pixels: two blocks would not fit a CU
      if (nst == 3 && nk < 2) nst = 2;
      if (nst == 2) launch_gemm9_cfg<2>(4, nt, a, st, 512);
      else if (nst == 4) launch_gemm9_cfg<4>(4, nt, a, st, 512);
      else launch_gemm9_cfg<3>(4, nt, a, st, 512);
      MGD_CHECK_LAUNCH("conv_gather_gemm(counted pipeline)");
      return MGD_OK;
    }
  }

  // ---- producer / consumer form: launches that do not fill the CUs once and have a K-loop worth pipelining - one deep-ring
  // block per CU hides the latency of a long K-loop better than a third of the CUs' worth of barrier-synchronous blocks
  // (1024 -> 512 at 19 x 19, 184 tiles: 66 us against 108) ----
  MGD_REQUIRE(form != MGD_CONV_PRODCONS || (nk >= 4 && !d->dst_f32), "conv: the producer/consumer form needs bf16 output and at least four K-steps");
  if (form == MGD_CONV_PRODCONS || (form == MGD_CONV_AUTO && nk >= 4 && !d->dst_f32 && tiles128 <= 256)) {
    launch_gemm6<2, 2, 4, 4, 4>(a, st);
    MGD_CHECK_LAUNCH("conv_gather_gemm(producer/consumer)");
    return MGD_OK;
  }

  // ---- everything else: weights straight from global memory, three blocks per CU; four waves along the channels (32 x 128
  // wave tiles): every weight fragment is loaded by exactly one wave.  form_arg 1: per-lane taps (the round-2 K-step, also
  // taken when a K-step may straddle taps) ----
  MGD_REQUIRE(form == MGD_CONV_AUTO || form == MGD_CONV_GLOBALW, "conv: unknown kernel form %d", form);
  a.korder = kmaj_large;
  if (uni && !(form == MGD_CONV_GLOBALW && farg == 1) && !MGD_DBG(a, 0xFE1)) launch_gemm8<2, 3, 4, true>(a, st);
  else launch_gemm8<2, 3, 4>(a, st);
  MGD_CHECK_LAUNCH("conv_gather_gemm(global weight fragments)");
  return MGD_OK;
}

// Several tap classes of ONE geometry in one launch (see GemmArgs::ncls): the descriptors may differ in wpk, K_pad, the taps
// and the output offset only.  Runs on conv_gemm8_kernel (128-channel tiles, wave-uniform K-steps); anything else: MGD_EINVAL,
// the caller launches the classes one by one.  The class with the longest contraction goes first (it ends last).
extern "C" int mgd_conv_gather_gemm_classes(const mgd_conv_desc* d, int n, void* stream) {
  MGD_REQUIRE(d && n >= 2 && n <= 4, "conv classes: 2 to 4 descriptors");
  GemmArgs a[4];
  for (int c = 0; c < n; ++c) { const int rc = conv_args(d + c, a[c]); if (rc != MGD_OK) return rc; }
  const mgd_conv_desc& r = d[0];
  MGD_REQUIRE(r.Co_pad % 128 == 0 && r.Ci % 64 == 0 && !r.dst_f32 && !r.latency && (r.form == MGD_CONV_AUTO || r.form == MGD_CONV_GLOBALW),
              "conv classes: 128-channel weight tiles, Ci %% 64 == 0, bf16 output, library dispatch");
  for (int c = 1; c < n; ++c) {
    const mgd_conv_desc& q = d[c];
    MGD_REQUIRE(q.src == r.src && q.dst == r.dst && q.bias == r.bias && q.addend == r.addend && q.stats == r.stats && q.N == r.N &&
                    q.Hs == r.Hs && q.Ws == r.Ws && q.Ci == r.Ci && q.Hg == r.Hg && q.Wg == r.Wg && q.Hd == r.Hd && q.Wd == r.Wd &&
                    q.Co == r.Co && q.in_stride == r.in_stride && q.out_stride == r.out_stride && q.Co_pad == r.Co_pad &&
                    q.dst_f32 == r.dst_f32 && q.stats_replicas == r.stats_replicas && q.bn_y == r.bn_y && q.bn_sums == r.bn_sums &&
                    q.act_slope == r.act_slope && !q.latency && q.form == r.form,
                "conv classes: descriptor %d differs from descriptor 0 in more than weights / taps / output offset", c);
  }
  int order[4] = {0, 1, 2, 3};
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (d[order[j]].K_pad > d[order[i]].K_pad) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  GemmArgs g = a[order[0]];
  g.ncls = n;
  // descriptor 0 may force the form (MGD_CONV_GLOBALW) with form_arg bits 256 = tap-major K order, 4 = tile-major blocks (measurement)
  g.korder = (d[0].form == MGD_CONV_GLOBALW ? ((d[0].form_arg & 256) ? 0 : 1) | ((d[0].form_arg & 4) ? 2 : 0) : 1);
  for (int c = 0; c < n; ++c) {
    const GemmArgs& s = a[order[c]];
    g.c_wpk[c] = s.wpk; g.c_tapcode[c] = s.tapcode; g.c_K_pad[c] = s.K_pad; g.c_ntaps[c] = s.ntaps;
    g.c_off_h[c] = s.out_off_h; g.c_off_w[c] = s.out_off_w; g.c_rowmask[c] = s.rowmask; g.c_colmask[c] = s.colmask;
  }
  launch_gemm8<2, 3, 4, true>(g, (hipStream_t)stream);
  MGD_CHECK_LAUNCH("conv_gather_gemm(global weight fragments, tap classes)");
  return MGD_OK;
}

extern "C" int mgd_conv_dgrad_s2_patch(const mgd_dgrad_s2_desc* d, void* stream) {
  MGD_REQUIRE(d && d->dy && d->dx && d->wpk[0] && d->wpk[1] && d->wpk[2] && d->wpk[3], "dgrad_s2_patch: null pointer");
  MGD_REQUIRE(d->Co == 64 && d->Ci == 32, "dgrad_s2_patch: built for 64 -> 32 channels (got %d -> %d)", d->Co, d->Ci);
  MGD_REQUIRE(d->H == 2 * d->Ho && d->W == 2 * d->Wo && d->N >= 1, "dgrad_s2_patch: geometry");
  MGD_REQUIRE((long long)d->N * d->H * d->W * d->Ci < (1ll << 31), "dgrad_s2_patch: tensor too large");
  MGD_REQUIRE(!d->bn_y || (d->bn_scale && d->bn_shift && d->bn_mean && d->bn_invstd && d->bn_sums && d->stats_replicas >= 1),
              "dgrad_s2_patch: fused BN-backward reduction needs scale/shift/mean/invstd/sums and stats_replicas");
  Dgrad2Args k;
  GemmArgs& a = k.g;
  a = GemmArgs{};
  a.src = (const bf16_t*)d->dy; a.dst = d->dx; a.addend = (const bf16_t*)d->addend;
  a.N = d->N; a.Co = d->Ci; a.Ci = d->Co; a.dst_f32 = 0; a.stats_replicas = d->stats_replicas > 0 ? d->stats_replicas : 1;
  a.bn_y = (const bf16_t*)d->bn_y; a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.bn_mean = d->bn_mean;
  a.bn_invstd = d->bn_invstd; a.bn_sums = d->bn_sums; a.bn_slope = d->bn_slope;
  for (int c = 0; c < 4; ++c) {
    k.wpk[c] = (const bf16_t*)d->wpk[c];
    k.K_pad[c] = d->K_pad[c];
    int ntap = (c & 2 ? 2 : 1) * (c & 1 ? 2 : 1);
    MGD_REQUIRE(d->K_pad[c] >= ntap * 64, "dgrad_s2_patch: K_pad[%d]=%d", c, d->K_pad[c]);
  }
  k.Ho = d->Ho; k.Wo = d->Wo; k.H = d->H; k.W = d->W;
  constexpr int WB = 9 * 2 * 2 * 1024, PITCH = 64 * 2 + 16;
  size_t pbytes = (size_t)((9 * 17 * PITCH + 15) / 16) * 16, ebytes = (size_t)128 * (32 * 2 + 16) + 4 * 2 * 32 * 4;
  size_t lds = WB + pbytes + ebytes + 2 * 128 * 8;
  auto kern = conv_patch_dgrad2_kernel<64, 32>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  long long tiles = (long long)d->N * cdiv(d->Ho, 8) * cdiv(d->Wo, 16);
  int per_cu = (int)(160 * 1024 / (lds + 512));
  if (per_cu > 4) per_cu = 4;
  long long grid = 256ll * per_cu;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, k);
  MGD_CHECK_LAUNCH("conv_dgrad_s2_patch");
  return MGD_OK;
}

// Caller-owned workspace of the latency form.  Partial tiles written by blocks on one XCD are read by a block on another
// inside the same kernel, which ordinary (L2-cached) device memory does not guarantee: the workspace must be UNCACHED device
// memory.  The library owns no buffer: a caller (one model instance per stream) allocates its own, passes it in
// mgd_conv_desc.partial and frees it; two callers never share tickets.
extern "C" int64_t mgd_latency_workspace_size(int tiles, int ranges) {
  if (tiles < 1 || ranges < 1) return 0;
  return 16384 + (int64_t)ranges * tiles * 128 * 64 * 4;
}

extern "C" int mgd_uncached_alloc(int64_t bytes, void** out) {
  MGD_REQUIRE(out && bytes >= 16384, "uncached_alloc: at least the 16 KiB of tickets");
  void* p = nullptr;
  MGD_REQUIRE(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) == hipSuccess && p, "uncached_alloc: allocation of %lld bytes failed", (long long)bytes);
  if (hipMemset(p, 0, 16384) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {      // the ticket area starts at zero
    (void)hipFree(p);
    return mgd_set_error(MGD_EINVAL, "uncached_alloc: memset failed");
  }
  *out = p;
  return MGD_OK;
}

extern "C" int mgd_uncached_free(void* p) {
  if (!p) return MGD_OK;
  MGD_REQUIRE(hipFree(p) == hipSuccess, "uncached_free: hipFree failed");
  return MGD_OK;
}

// test hook: the 4096 tile tickets of a workspace, copied to the host after the stream has drained (all zero between launches)
extern "C" int mgd_latency_tickets(const void* workspace, unsigned* out4096, void* stream) {
  MGD_REQUIRE(workspace && out4096, "latency_tickets: null pointer");
  MGD_REQUIRE(hipStreamSynchronize((hipStream_t)stream) == hipSuccess && hipMemcpy(out4096, workspace, 16384, hipMemcpyDeviceToHost) == hipSuccess, "latency_tickets: copy");
  return MGD_OK;
}

#ifdef MGD_DIAG
// diagnostic library only: reads and clears the phase-time accumulators of the stamped ping-pong build (include/mgd_hip_diag.h)
extern "C" int mgd_debug_stamps(unsigned long long* out24) {
  MGD_REQUIRE(out24, "debug_stamps: null pointer");
  unsigned long long z[24] = {0};
  if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_stamps), sizeof(z)) != hipSuccess) return MGD_EINVAL;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return MGD_EINVAL;
  return MGD_OK;
}
#endif
