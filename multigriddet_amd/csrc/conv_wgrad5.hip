// Kernel-row weight gradient (round 4): dW[co][tap][ci] += sum_p dy[p][co] * x[p (+) tap][ci] for the stride-1 'same' layers
// with Ci, Co >= 128 (3x3 in the standard tap order) and for 1x1 layers.  Replaces the Conv2D kernel gradient of Keras autodiff
// (reference multigriddet/models/layers.py:43-49).
//
// What bounds the per-tap form (conv_wgrad4_kernel, 128 x 64 tile of ONE tap, three blocks per CU): a CU retires one 1-KiB
// LDS-DMA wave-instruction per ~35 cycles whatever it fetches - measured again this round on the gather-GEMM, where a build
// with every DMA lane out of range runs as long as the real one - and a per-tap block stages 24 of them per 256 cycles of
// MFMA work: 30 % of the matrix peak at best, 27 % measured.  Bytes per MFMA have to come down:
//   * ONE dy tile and ONE input tile serve the THREE taps of a kernel row: on a stride-1 'same' layer the flat pixel index is
//     linear in memory, so tap dw of output pixel k reads input row k + dw of the same staged tile;
//   * block tile 128 co x 128 ci x 3 taps, 8 waves (2 x 4) of 64 co x 32 ci x 3 taps: 4 DMA pieces per wave and K-step for
//     48 MFMAs per wave - 1 140 cycles of DMA issue against 1 536 of MFMA per CU and K-step;
//   * one block per CU, 4-stage LDS ring, ONE phase per K-step in the rotated two-group schedule of conv_gemm12.hip
//     (waves 0-3: R(k) M(k) | barrier, waves 4-7: M(k-1) R(k) | barrier): R reads every fragment of stage k (transposed LDS
//     reads, ds_read_b64_tr_b16) into registers, issues the DMA of stage k+3 and waits - counted - for stage k+1.
// First version of this file (masks on the fragments, XOR-swizzled tiles, fp32 atomics): correct at the first run and SLOWER
// than the per-tap form, 101 against 80 us on 128 -> 256 at 76 x 76 - 42 us of it the epilogue (252 blocks x 192 KiB of atomics
// = 48 MB at the chip's ~1.15 TB/s of added bytes, all blocks at once), and a K-step of 3 300 cycles: ~150 vector
// instructions (border masks, address XORs) + 40 fragment reads next to 48 MFMAs, twice the issue slots the MFMAs leave.
// Hence:
//   * PADDED K INDEX instead of masks.  The contraction runs over u = p + (p / W): every image row is followed by one PAD
//     position whose dy and x are zero (an out-of-range DMA lane).  Tap dw = -1 of a pixel in column 0 and tap dw = +1 of a
//     pixel in column W-1 then read the pad - zero - instead of the neighbouring row's end, and the K-loop carries no mask
//     at all; the price is (W+1)/W more K-steps (1.3 % at 76 x 76, 5 % at 19 x 19).  The kernel row dh is applied at the DMA
//     too: an input row is staged only if its output pixel's row h has a partner h + dh inside the image.
//   * CHUNK-MAJOR tiles: [32-byte channel chunk (8)][pixel position (64)][32 B]; the pixel order inside a column is a fixed
//     permutation (pos) that keeps the eight rows of a half-wave's transposed read on eight different bank groups for every
//     tap shift.  A fragment's channel group and ring stage are IMMEDIATE offsets of the read: no address arithmetic in the loop.
//   * SLABS instead of atomics when the caller provides a workspace: every block stores its fp32 tile with plain stores to
//     slab[split] (the stores run at ~5x the atomic rate) and wgrad5_reduce_kernel adds the slabs into dW.
//
// K-step geometry.  A step covers padded indices u0 .. u0+61; both tiles hold the 64 indices u0-1 .. u0+62 (tile row e <-> u0-1+e),
// dy rows 0 and 63 forced to zero (they belong to the neighbouring steps), so tap dw of dy row e reads input row e + dw, and the
// two reads that leave the tile (row -1, row 64) are clamped: their dy factor is zero, any finite value will do.
// Hazards: as conv_gemm12.hip (WAR: a stage is re-filled two intervals after its fragment reads, which end with lgkmcnt(0)
// in front of the barrier; RAW: the counted wait of R(k) covers stage k+1, first read in interval k+1, behind barrier k).
#include "conv_common.hpp"

namespace {

using mgd::WgradArgs;

constexpr int W5_KP = 62;                                      // padded indices per K-step

__host__ __device__ constexpr int w5_pos(int row) {           // tile row -> position inside a chunk column
  return (row & 3) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3) | ((row >> 4) << 4);
}
__host__ __device__ constexpr int w5_row(int pos) {           // its inverse
  return (pos & 3) | (((pos >> 3) & 1) << 2) | (((pos >> 2) & 1) << 3) | ((pos >> 4) << 4);
}

// NTAP: 3 (one kernel row of a 3x3 layer per block) or 1 (1x1 layer); SLAB: plain stores to a.slab[split] instead of atomics
template <int NTAP, bool SLAB, int NS = 4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad5_kernel(WgradArgs a) {
  constexpr int MT = 4, NT = 2, WI = 4;                        // 8 waves: 2 (co) x 4 (ci), wave tile 64 co x 32 ci
  constexpr int KP = W5_KP;
  constexpr int COL = 64 * 32;                                 // bytes of one chunk column (64 positions x 32 B)
  constexpr int TILE = 8 * COL;                                // 16 KiB: 64 rows x 128 channels
  constexpr int XBASE = NS * TILE;                             // [dy0 .. dy(NS-1)][x0 .. x(NS-1)]
  static_assert(NS == 3 || NS == 4, "ring depth");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WI, wi = wave % WI;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int tco = b % a.tilesCo; b /= a.tilesCo;
  const int tci = b % a.tilesCi; b /= a.tilesCi;
  constexpr int nrow = NTAP == 3 ? 3 : 1;
  const int krow = b % nrow;                                   // kernel row (dh + 1)
  const int split = b / nrow;
  const int co0 = tco * 128, ci0 = tci * 128;
  const int dh = NTAP == 3 ? krow - 1 : 0;
  const int W = a.Wg, HW = a.Hg * a.Wg;
  const int Wp = NTAP == 3 ? W + 1 : (1 << 30);                // padded row length (1x1: no taps, no pads)
  const int pbeg = split * a.chunk;
  const int pend = min(a.P, pbeg + a.chunk);
  // padded range of the block: u = p + p / W
  const int ubeg = NTAP == 3 ? pbeg + pbeg / W : pbeg;
  const int uend = NTAP == 3 ? (pend - 1) + (pend - 1) / W + 1 : pend;
  const int nk = (uend - ubeg + KP - 1) / KP;
  const int nk3 = (nk + NS - 1) / NS * NS;                     // the loop runs whole ring revolutions; steps past the end see zeros
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // fixed descriptors: dy over the block's pixel range (rows past it read as zeros), x from the block's first source pixel on
  auto make_srd = [&](unsigned long long base, long long rec) {
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)(rec > 0 ? (rec < 0x7FFFFFF0ll ? rec : 0x7FFFFFF0ll) : 0));
    r[3] = 0x00020000;
    return r;
  };
  const i32x4 osrd = make_srd((unsigned long long)a.dy + (unsigned long long)((long long)pbeg * a.Co * 2), (long long)(pend - pbeg) * a.Co * 2);
  // (the input descriptor starts ONE pixel early: tile row 0 of the first step is pixel pbeg - 1, and offsets are unsigned)
  const i32x4 xsrd = make_srd((unsigned long long)((long long)(unsigned long long)a.src + ((long long)pbeg - 1 + (long long)dh * W) * a.Ci * 2), 0x7FFFFFF0ll);

  // staging: a DMA piece = 1 KiB per wave = 32 positions x 32 B of ONE chunk column; wave w fills column w (16 channels) of
  // both tiles, piece j = positions 32 j .. 32 j + 31; lane l: position 32 j + (l >> 1), 16-byte half l & 1
  const int chan = wave * 16 + (lane & 1) * 8;                 // this lane's first channel inside the 128-channel tile
  const unsigned o_c = co0 + chan < a.Co ? (unsigned)((co0 + chan) * 2) : OOB;
  const unsigned x_c = ci0 + chan < a.Ci ? (unsigned)((ci0 + chan) * 2) : OOB;
  int e_row[2], t_uw[2], t_p[2], t_pq[2];                     // per staged row: tile row, column in the padded row, pixel - pbeg, pixel inside its image
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    e_row[j] = w5_row(32 * j + (lane >> 1));
    const int u = ubeg - 1 + e_row[j];                         // (u = -1 only in front of pixel 0: never valid)
    const int R = u >= 0 ? u / Wp : -1;
    t_uw[j] = u - R * Wp;
    const int p = u - R;                                       // pixel of a non-pad position
    t_p[j] = p - pbeg;
    t_pq[j] = ((p % HW) + HW) % HW;
  }
  const int adv_w = KP % Wp, adv_R = KP / Wp;
  const int q_lo = dh < 0 ? W : 0, q_hi = dh > 0 ? HW - W : HW;   // output pixels whose row h has a partner h + dh
  const unsigned ldsb = lds_addr(smem);
  int issued = 0;
  auto issue = [&](int st) {
    unsigned ov[2], xv[2], od[2], xd[2];
    const bool live = issued < nk;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool real = NTAP == 1 || t_uw[j] < W;              // not a pad position
      const int p = t_p[j] + pbeg;
      const bool inp = p >= 0 && p < a.P;
      const bool okx = live && real && inp && t_pq[j] >= q_lo && t_pq[j] < q_hi;
      const bool oko = live && real && t_p[j] >= 0 && e_row[j] >= 1 && e_row[j] <= KP;
      ov[j] = (oko && o_c != OOB) ? (unsigned)(t_p[j] * a.Co * 2) + o_c : OOB;
      xv[j] = (okx && x_c != OOB) ? (unsigned)((t_p[j] + 1) * a.Ci * 2) + x_c : OOB;
      // next K-step: u += KP
      t_uw[j] += adv_w;
      const int wrap = t_uw[j] >= Wp ? 1 : 0;
      t_uw[j] -= wrap ? Wp : 0;
      const int dp = KP - adv_R - wrap;
      t_p[j] += dp;
      t_pq[j] += dp;
      t_pq[j] -= t_pq[j] >= HW ? HW : 0;
      od[j] = ldsb + st * TILE + wave * COL + j * 1024;
      xd[j] = ldsb + XBASE + st * TILE + wave * COL + j * 1024;
    }
    dma_to<2>(ov, od, osrd);
    dma_to<2>(xv, xd, xsrd);
    ++issued;
  };

  f32x4 acc[NTAP][MT][NT];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: half h of the fragment of 32-row sub-step kk covers tile rows r0 = 32 kk + 8 g + q + 4 h (g = lane >> 4,
  // q = (lane & 15) >> 2), 8 bytes at pp = lane & 3 of the row's 32-byte chunk; chunk column (wc*4 + m) resp. (wi*2 + n) and
  // the ring stage are the instruction's immediate offset
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  unsigned a_rd[2][2], b_rd[NTAP][2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r0 = kk * 32 + 8 * g + qq + 4 * h;
      a_rd[kk][h] = ldsb + wc * MT * COL + w5_pos(r0) * 32 + pp * 8;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        int r = r0 + (NTAP == 3 ? t - 1 : 0);
        r = r < 0 ? 0 : (r > 63 ? 63 : r);                     // rows -1 / 64 meet the zero dy rows 0 / 63: any finite value
        b_rd[t][kk][h] = ldsb + XBASE + wi * NT * COL + w5_pos(r) * 32 + pp * 8;
      }
    }

  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x4 fa[2][MT][2], fb[NTAP][2][NT][2];
  // diagnostic library (tools/ablate_wgrad.py): 256 = no fragment reads, 512 = no DMA issue, 1024 = no MFMAs, 2048 = no barrier
  const bool dg_nr = MGD_DBG(a, 256), dg_nd = MGD_DBG(a, 512), dg_nm = MGD_DBG(a, 1024), dg_nb = MGD_DBG(a, 2048);
  auto R = [&](auto ST) {
    constexpr int st = decltype(ST)::value;
    if (!dg_nr)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        tr_read_asm<st * TILE + 0 * COL>(fa[kk][0][h], a_rd[kk][h]);
        tr_read_asm<st * TILE + 1 * COL>(fa[kk][1][h], a_rd[kk][h]);
        tr_read_asm<st * TILE + 2 * COL>(fa[kk][2][h], a_rd[kk][h]);
        tr_read_asm<st * TILE + 3 * COL>(fa[kk][3][h], a_rd[kk][h]);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          tr_read_asm<st * TILE + 0 * COL>(fb[t][kk][0][h], b_rd[t][kk][h]);
          tr_read_asm<st * TILE + 1 * COL>(fb[t][kk][1][h], b_rd[t][kk][h]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    if (!dg_nd) issue((st + NS - 1) % NS);
    // the pieces of the NS - 2 youngest stages stay in flight: stage k+1 has landed (NS = 4: it was issued two intervals ago -
    // the operands stream from HBM / the memory-side cache, a single interval does not cover that round trip)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int m = 0; m < MT; ++m) touch(fa[kk][m][h]);
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) touch(fb[t][kk][n][h]);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto M = [&]() {
    if (dg_nm) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const s16x8 av = __builtin_shufflevector(fa[kk][m][0], fa[kk][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const s16x8 bv = __builtin_shufflevector(fb[t][kk][n][0], fb[t][kk][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[t][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                  acc[t][m][n], 0, 0, 0);
          }
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (!dg_nb) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
  bar();                                                       // stage 0 is complete for everyone
  if (wave < 4) {
    for (int t = 0; t < nk3; t += NS) {
      R(S0{}); M(); bar();
      R(S1{}); M(); bar();
      R(S2{}); M(); bar();
      if constexpr (NS == 4) { R(S3{}); M(); bar(); }
    }
  } else {
    R(S0{}); bar();
    for (int t = 0; t < nk3; t += NS) {
      M(); R(S1{}); bar();
      M(); R(S2{}); bar();
      if constexpr (NS == 4) { M(); R(S3{}); bar(); }
      M();
      if (t + NS < nk3) { R(S0{}); bar(); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the dummy stages' zero writes land before the wave ends

  const int fr = lane & 15, fq = lane >> 4;
  if (MGD_DBG(a, 32)) return;                                  // diagnostic library: no epilogue (tools/ablate_wgrad.py)
  float* out = SLAB ? a.slab + (long long)split * a.Co * a.ntaps * a.Ci : a.dw;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int tap = NTAP == 3 ? krow * 3 + t : 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wc * MT + m) * 16 + fq * 4 + r;
        if (co >= a.Co) continue;
        float* row = out + ((long long)co * a.ntaps + tap) * a.Ci;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int ci = ci0 + (wi * NT + n) * 16 + fr;
          if (ci >= a.Ci) continue;
          if constexpr (SLAB) row[ci] = acc[t][m][n][r];       // 64 contiguous bytes per 16 lanes
          else atomicAdd(row + ci, acc[t][m][n][r]);
        }
      }
  }
}

// dW += sum over the splits' slabs (every element of every slab was written: blocks cover all tiles x kernel rows x splits)
__global__ __launch_bounds__(256) void wgrad5_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n4, int splits,
                                                            long long stride4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* s = (const f32x4*)slab + i;
  f32x4 v = ((const f32x4*)dw)[i];
  int k = 0;
  for (; k + 4 <= splits; k += 4) {
    const f32x4 a0 = s[(long long)k * stride4], a1 = s[(long long)(k + 1) * stride4], a2 = s[(long long)(k + 2) * stride4],
                a3 = s[(long long)(k + 3) * stride4];
    v += (a0 + a1) + (a2 + a3);
  }
  for (; k < splits; ++k) v += s[(long long)k * stride4];
  ((f32x4*)dw)[i] = v;
}

template <int NTAP, bool SLAB>
void launch5(WgradArgs& a, int nblk, hipStream_t st) {
  auto k = conv_wgrad5_kernel<NTAP, SLAB>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(k, dim3(nblk), dim3(512), 2 * 4 * 8 * 64 * 32, st, a);           // NS = 4 stages x (dy + x) tiles of 16 KiB
}

// Split plan of the kernel-row form: blocks = tiles x splits cut to ONE round of 256; a.chunk / a.splits are filled in.
bool wgrad5_plan(WgradArgs& a) {
  if (!(a.ntaps == 9 || a.ntaps == 1) || a.Wg < 8 || a.Hg * a.Wg < 64 || a.in_stride != 1 || a.Hs != a.Hg || a.Ws != a.Wg) return false;
  a.tilesCo = cdiv(a.Co, 128);
  a.tilesCi = cdiv(a.Ci, 128);
  const int tiles = a.tilesCo * a.tilesCi * (a.ntaps == 9 ? 3 : 1);
  // one block per CU, one round; a.splits > 0: the caller's cap on the number of blocks (a launch that shares the chip with
  // another stream leaves CUs free: a block holds every register of its CU)
  const int cap = a.splits > 0 && a.splits < 256 ? a.splits : 256;
  int splits = cap / tiles;
  if (splits < 1) splits = 1;
  a.chunk = cdiv(cdiv(a.P, splits), 4 * W5_KP) * 4 * W5_KP;    // about whole ring revolutions of K-steps
  a.splits = cdiv(a.P, a.chunk);
  // 32-bit offsets inside a block's pixel range
  return (long long)(a.chunk + 4 * a.Wg + 256) * (a.Co > a.Ci ? a.Co : a.Ci) * 2 < (1ll << 31);
}

}  // namespace

namespace mgd {

long long wgrad5_workspace_bytes(WgradArgs a) {
  if (!wgrad5_plan(a)) return 0;
  return (long long)a.splits * a.Co * a.ntaps * a.Ci * 4;
}

// Geometry: stride-1 'same' layer (the caller checked), ntaps == 9 in the standard order (dh = t / 3 - 1, dw = t % 3 - 1) or
// ntaps == 1, W >= 8.  With a.slab (>= wgrad5_workspace_bytes) the blocks store slabs and a second launch adds them into dW.
int launch_wgrad5(WgradArgs& a, hipStream_t st) {
  if (!wgrad5_plan(a)) return MGD_EINVAL;
  const int tiles = a.tilesCo * a.tilesCi * (a.ntaps == 9 ? 3 : 1);
  const int nblk = tiles * a.splits;
  const long long elems = (long long)a.Co * a.ntaps * a.Ci;
  const bool slab = a.slab != nullptr && a.slab_bytes >= a.splits * elems * 4 && a.Ci % 16 == 0 && elems % 4 == 0;
  if (a.ntaps == 9) { if (slab) launch5<3, true>(a, nblk, st); else launch5<3, false>(a, nblk, st); }
  else { if (slab) launch5<1, true>(a, nblk, st); else launch5<1, false>(a, nblk, st); }
  if (slab) {
    const long long n4 = elems / 4;
    hipLaunchKernelGGL(wgrad5_reduce_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, (const float*)a.slab, a.dw, n4, a.splits, n4);
  }
  return MGD_OK;
}

}  // namespace mgd
