// Weight gradient of the convolution engine (gfx950): per-tap blocks on v_mfma_f32_16x16x32_bf16 with transposed LDS reads.
#include "conv_common.hpp"

extern "C" int64_t mgd_conv_wgrad_workspace_size(const mgd_wgrad_desc* d);

namespace {

using mgd::WgradArgs;


// (one raw barrier per 64-pixel K-step, the next stage's loads in flight during the MFMAs).
template <int WC, int WI, int MT, int NT>
__global__ __launch_bounds__(256) MGD_VGPR_MFMA void conv_wgrad2_kernel(WgradArgs a) {
  constexpr int BCO = WC * MT * 16;
  constexpr int BCI = WI * NT * 16;
  static_assert(WC * WI == 4, "4 waves");
  constexpr int RBO = BCO * 2, RBI = BCI * 2;
  constexpr int OCH = RBO / 64, ICH = RBI / 64;        // LDS-DMA instructions per wave per stage
  constexpr int ORPI = 1024 / RBO, IRPI = 1024 / RBI;  // rows per wave-instruction
  constexpr int STAGE = 64 * (RBO + RBI);

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WI, wi = wave % WI;

  // the blocks of one pixel range (all channel tiles x taps: they stage the same dy / input rows at about the same time) get
  // consecutive logical ids = one XCD, so that the re-reads hit its L2 (MGD_DBG & 65536: plain block order)
  int b = MGD_DBG(a, 65536) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int tco = b % a.tilesCo; b /= a.tilesCo;
  const int tci = b % a.tilesCi; b /= a.tilesCi;
  const int tap = b % a.ntaps;
  const int split = b / a.ntaps;
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int dh = (int)((a.tapcode >> (4 * tap)) & 3) - 1;
  const int dw = (int)((a.tapcode >> (4 * tap + 2)) & 3) - 1;
  const int pbeg = split * a.chunk;
  const int pend = min(a.P, pbeg + a.chunk);
  const int nk = (pend - pbeg + 63) / 64;
  const void* zero = (const void*)g_zero_page;
  asm volatile("" : "+s"(zero));

  // per-thread (row-in-group, logical chunk) for both tiles
  const int o_rl = lane / (RBO / 16), o_s = lane % (RBO / 16);
  const int i_rl = lane / (RBI / 16), i_s = lane % (RBI / 16);

  // Address generation off the critical path.  K runs over PIXELS here, so the rows a lane stages change every K-step;
  // the first version re-derived (image, row, column) of every row from its flat pixel index per step - two divisions
  // with correction loops per LDS-DMA instruction, 510 VALU + 395 SALU instructions per 32 MFMA (ISA count); carrying the
  // coordinates left 173 + 83, and an ablation build still put HALF of the kernel's time into this skeleton (128->256 at
  // 76x76: 53 of 106 us with MFMA, fragment reads and LDS-DMA all switched off) - with three waves per SIMD every VALU
  // instruction costs 12 cycles of a SIMD's issue.  Now the source ADDRESS itself is carried (a lane's rows advance by
  // exactly 64 pixels per K-step: one uniform 64-bit increment, plus a constant when the column / the row wraps), and
  // (row, column) survive only for the validity test, done as two unsigned range compares against per-tap bounds.
  // Uniform values are pinned in SGPRs (a scalar re-load from the argument segment inside the loop drains lgkmcnt).
  const int Wg = sgpr(a.Wg), Hg = sgpr(a.Hg);
  const int q64 = sgpr(64 / a.Wg), r64 = sgpr(64 % a.Wg);
  const int wraps = sgpr((64 / a.Wg + 1 + a.Hg - 1) / a.Hg);            // image wraps one 64-pixel advance can cross
  const int sst = a.in_stride;
  // valid source rows / columns of this tap, as ranges of the OUTPUT coordinate: lo <= x <= lo + span
  const int lo_i = __builtin_amdgcn_readfirstlane(dh < 0 ? (-dh + sst - 1) / sst : 0);
  const int lo_j = __builtin_amdgcn_readfirstlane(dw < 0 ? (-dw + sst - 1) / sst : 0);
  const int hi_i = min(a.Hg - 1, (a.Hs - 1 - dh) / sst), hi_j = min(a.Wg - 1, (a.Ws - 1 - dw) / sst);
  const unsigned span_i = (unsigned)__builtin_amdgcn_readfirstlane(hi_i - lo_i);
  const unsigned span_j = (unsigned)__builtin_amdgcn_readfirstlane(hi_j - lo_j);
  const bool tap_ok = hi_i >= lo_i && hi_j >= lo_j;
  long long dstep = (long long)a.Co * 128;                                                  // dy: 64 pixels further
  long long xstep = ((long long)(64 / a.Wg) * sst * a.Ws + (long long)(64 % a.Wg) * sst) * a.Ci * 2;   // x: the same advance
  long long xrow = ((long long)sst * a.Ws - (long long)a.Wg * sst) * a.Ci * 2;               // column wrapped: next output row
  long long ximg = ((long long)a.Hs - (long long)a.Hg * sst) * a.Ws * a.Ci * 2;              // row wrapped: next image
  asm volatile("" : "+s"(dstep), "+s"(xstep), "+s"(xrow), "+s"(ximg));
  const char* o_ad[OCH];         // address of the lane's dy chunk at the next K-step to issue
  int o_left[OCH];               // pixels left in the block's range from this row on (<= 0: past the end / channel padding)
#pragma unroll
  for (int i = 0; i < OCH; ++i) {
    const int r = (i * 4 + wave) * ORPI + o_rl;
    const int ch = (((o_s >> 1) ^ tr_swz(r, RBO / 32)) << 1) | (o_s & 1);
    const int c = co0 + ch * 8;
    o_left[i] = c < a.Co ? pend - (pbeg + r) : -(1 << 30);
    o_ad[i] = (const char*)a.dy + ((long long)(pbeg + r) * a.Co + c) * 2;
  }
  const char* x_ad[ICH];
  int x_left[ICH], x_i[ICH], x_j[ICH];
#pragma unroll
  for (int i = 0; i < ICH; ++i) {
    const int r = (i * 4 + wave) * IRPI + i_rl;
    const int ch = (((i_s >> 1) ^ tr_swz(r, RBI / 32)) << 1) | (i_s & 1);
    const int c = ci0 + ch * 8;
    const int pix = pbeg + r, hw = a.Hg * a.Wg;
    const int n = pix / hw, rem = pix - n * hw;
    x_left[i] = (c < a.Ci && tap_ok) ? pend - pix : -(1 << 30);
    x_i[i] = rem / a.Wg;
    x_j[i] = rem - x_i[i] * a.Wg;
    // may point outside the tensor where the tap leaves the image: such rows are never fetched
    x_ad[i] = (const char*)a.src + ((((long long)n * a.Hs + (x_i[i] * sst + dh)) * a.Ws + (x_j[i] * sst + dw)) * a.Ci + c) * 2;
  }

  // Staging the NEXT K-step (steps go out in order, each exactly once) is split in two: prep() picks the eight source
  // addresses and advances the carried state - plain VALU work, placed behind the first MFMAs of the PREVIOUS step, where
  // it runs in the shadow of the matrix pipe - and fire() is the eight LDS-DMA instructions alone, right after the barrier:
  // with a two-deep ring the DMA round trip is the step's critical path, and nothing may sit between barrier and issue.
  const void* og[OCH];
  const void* xg[ICH];
  auto prep = [&]() {
#pragma unroll
    for (int i = 0; i < OCH; ++i) {
      og[i] = o_left[i] > 0 ? (const void*)o_ad[i] : zero;
      o_left[i] -= 64;
      o_ad[i] += dstep;
    }
#pragma unroll
    for (int i = 0; i < ICH; ++i) {
      const bool v = x_left[i] > 0 && (unsigned)(x_i[i] - lo_i) <= span_i && (unsigned)(x_j[i] - lo_j) <= span_j;
      xg[i] = v ? (const void*)x_ad[i] : zero;
      x_left[i] -= 64;
      x_ad[i] += xstep;
      x_j[i] += r64;
      x_i[i] += q64;
      const bool cj = x_j[i] >= Wg;                      // branch-free carries: exec-mask branches cost more than the selects
      x_j[i] -= cj ? Wg : 0;
      x_i[i] += cj ? 1 : 0;
      x_ad[i] += cj ? xrow : 0ll;
      for (int w = 0; w < wraps; ++w) {
        const bool cn = x_i[i] >= Hg;
        x_i[i] -= cn ? Hg : 0;
        x_ad[i] += cn ? ximg : 0ll;
      }
    }
  };
  auto fire = [&](int buf) {
    unsigned char* ob = smem + buf * STAGE;
    unsigned char* ib = ob + 64 * RBO;
#pragma unroll
    for (int i = 0; i < OCH; ++i) glds16(og[i], ob + (i * 4 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < ICH; ++i) glds16(xg[i], ib + (i * 4 + wave) * 1024);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  prep();
  if (nk > 0) fire(0);
  prep();                   // addresses of step 1
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  // fragment read offsets inside a stage: MFMA k index = pixel row kk*32 + 8g + qq (+4 for the upper half)
  int o_rd[2][2][MT], i_rd[2][2][NT];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r0 = kk * 32 + 8 * g + qq + 4 * h;
#pragma unroll
      for (int m = 0; m < MT; ++m) o_rd[kk][h][m] = r0 * RBO + (((wc * MT + m) ^ tr_swz(r0, RBO / 32)) * 32) + pp * 8;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        i_rd[kk][h][n] = 64 * RBO + r0 * RBI + (((wi * NT + n) ^ tr_swz(r0, RBI / 32)) * 32) + pp * 8;
    }
  // The transposed fragment reads go out from inline asm: in front of a ds_read_b64_tr_b16 it can see, the compiler
  // puts s_waitcnt vmcnt(0) (it cannot tell the read from the LDS-DMA writes in flight), which made every K-step
  // wait for the NEXT stage's loads before computing the current one.  Waits are explicit instead.
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const unsigned smem_a = lds_addr(smem);
  for (int ks = 0; ks < nk; ++ks) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const bool more = ks + 1 < nk;
    const unsigned sb = smem_a + (ks & 1) * STAGE;
    s16x4 fa[2][MT][2], fb[2][NT][2];
    auto read_half = [&](int kk) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        tr_read_asm<0>(fa[kk][m][0], sb + o_rd[kk][0][m]);
        tr_read_asm<0>(fa[kk][m][1], sb + o_rd[kk][1][m]);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        tr_read_asm<0>(fb[kk][n][0], sb + i_rd[kk][0][n]);
        tr_read_asm<0>(fb[kk][n][1], sb + i_rd[kk][1][n]);
      }
    };
    auto mfma_half = [&](int kk) {
#pragma unroll
      for (int m = 0; m < MT; ++m) { touch(fa[kk][m][0]); touch(fa[kk][m][1]); }
#pragma unroll
      for (int n = 0; n < NT; ++n) { touch(fb[kk][n][0]); touch(fb[kk][n][1]); }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        s16x8 av = __builtin_shufflevector(fa[kk][m][0], fa[kk][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          s16x8 bv = __builtin_shufflevector(fb[kk][n][0], fb[kk][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                             acc[m][n], 0, 0, 0);
        }
      }
    };
    if (more) fire((ks + 1) & 1);
    read_half(0);
    wait_lgkm_dyn(0);
    read_half(1);          // in flight under the MFMAs of the first half
    mfma_half(0);
    prep();                // addresses of step ks + 2
    wait_lgkm_dyn(0);
    mfma_half(1);
  }
  // epilogue straight from the accumulators: a lane's (m, n, r) element belongs to channel row (wc*MT + m)*16 + fq*4 + r
  // and column (wi*NT + n)*16 + fr, so the 16 lanes of equal fq add 64 contiguous bytes - one memory-side atomic request,
  // the same as from an LDS-staged tile, without the LDS round trip and its two barriers
  const int fr = lane & 15, fq = lane >> 4;
  if MGD_DBG(a, 32) return;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + (wc * MT + m) * 16 + fq * 4 + r;
      if (co >= a.Co) continue;
      float* row = a.dw + ((long long)co * a.ntaps + tap) * a.Ci;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ci = ci0 + (wi * NT + n) * 16 + fr;
        if (ci >= a.Ci) continue;
        if MGD_DBG(a, 16) row[ci] = acc[m][n][r]; else atomicAdd(row + ci, acc[m][n][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient v4 (round 3): conv_wgrad2_kernel with the address arithmetic taken out of the vector ALU.
// PMC counters of v2 on 128 -> 256 at 76 x 76 (tools/prof_conv.sh): 9 VALU instructions per MFMA - 125 per K-step and wave
// carrying (row, column, image) of the staged pixels, testing tap bounds and selecting 64-bit source addresses - the SIMDs'
// vector issue 60 % busy, the matrix pipe 24 %.  For stride-1 'same' convolutions none of it is needed:
//  * the staged rows of a K-step are 64 CONSECUTIVE pixels, and both operands are linear in the flat pixel index (the
//    input shifted by the tap: pixel p + dh*W + dw) - so each lane's byte offset is a CONSTANT, and the K-step advance moves
//    the base of a raw buffer descriptor held in SGPRs (three scalar instructions per operand and step);
//  * the end of the block's pixel range is the descriptor's num_records: rows past it read as zeros in hardware;
//  * the only per-pixel fact left, "does this tap stay inside the image at pixel q", is one bit: the block builds the bit
//    map of ITS tap over one image (H*W bits, <= 23 words at 76 x 76... 181 words) in LDS once, and a staged input row costs an
//    index update, an LDS word, a bit test and a select per K-step.
// Same tiles, ring, transposed fragment reads and atomic epilogue as v2.
// NR: ring depth.  Both operands come by LDS-DMA, so a wave's vector-memory operations retire in order and a counted wait is
// exact: NR >= 3 leaves NR - 2 whole stages in flight across the barrier (dummy stages behind the last step - the descriptors'
// num_records have run out, every lane is out of range: no traffic, same count).
// KP: pixels per K-step (64 or 32).  32 halves the ring, so that a 128 x 128 tile (256 bytes of LDS-DMA per MFMA instead of the
// 384 of 128 x 64) still runs three blocks per CU: its fragment reads are then pipelined over the two halves of the channel
// rows instead of the two 32-pixel halves of the step.
template <int WC, int WI, int MT, int NT, int NR = 2, int KP = 64, bool IL = false>
__global__ __launch_bounds__(256) MGD_VGPR_MFMA void conv_wgrad4_kernel(WgradArgs a) {
  constexpr int BCO = WC * MT * 16;
  constexpr int BCI = WI * NT * 16;
  static_assert(WC * WI == 4, "4 waves");
  constexpr int RBO = BCO * 2, RBI = BCI * 2;
  static_assert(KP == 64 || (KP == 32 && MT % 2 == 0), "pixels per K-step");
  constexpr int KK = KP / 32;                          // 32-pixel MFMA k-steps per stage
  constexpr int OCH = KP * RBO / 4096, ICH = KP * RBI / 4096;   // LDS-DMA instructions per wave per stage
  constexpr int ORPI = 1024 / RBO, IRPI = 1024 / RBI;  // rows per wave-instruction
  constexpr int STAGE = KP * (RBO + RBI);

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int GRP = OCH + ICH;                       // LDS-DMA instructions per wave and stage
  unsigned* mbits = (unsigned*)(smem + NR * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WI, wi = wave % WI;

  // the blocks of one pixel range (all channel tiles x taps: they stage the same dy / input rows at about the same time) get
  // consecutive logical ids = one XCD, so that the re-reads hit its L2 (MGD_DBG & 65536: plain block order)
  int b = MGD_DBG(a, 65536) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int tco = b % a.tilesCo; b /= a.tilesCo;
  const int tci = b % a.tilesCi; b /= a.tilesCi;
  const int tap = b % a.ntaps;
  const int split = b / a.ntaps;
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int dh = (int)((a.tapcode >> (4 * tap)) & 3) - 1;
  const int dw = (int)((a.tapcode >> (4 * tap + 2)) & 3) - 1;
  const int pbeg = split * a.chunk;
  const int pend = min(a.P, pbeg + a.chunk);
  const int nk = (pend - pbeg + KP - 1) / KP;
  const int HW = a.Hg * a.Wg;
  const unsigned OOB = 0xFFFFFFF0u;

  // bit q of the map: the tap's source pixel of output pixel q (of one image) lies inside the image
  const bool masked = (dh != 0 || dw != 0) && !MGD_DBG(a, 16777216);   // diagnostic 16777216: no border bit map (wrong sums at the borders)
  if (masked) {
    for (int w = tid; w < (HW + 31) / 32; w += 256) {
      const int q0 = w * 32;
      int i = q0 / a.Wg, j = q0 - i * a.Wg;
      unsigned bits = 0;
      for (int t = 0; t < 32; ++t) {
        const bool ok = q0 + t < HW && (unsigned)(i + dh) < (unsigned)a.Hg && (unsigned)(j + dw) < (unsigned)a.Wg;
        bits |= (ok ? 1u : 0u) << t;
        if (++j == a.Wg) { j = 0; ++i; }
      }
      mbits[w] = bits;
    }
  }

  // raw buffer descriptors over the block's pixel range; the K-loop moves base and num_records by one 64-pixel step
  const long long dstep = (long long)a.Co * 2 * KP, xstep = (long long)a.Ci * 2 * KP;
  unsigned long long obase = (unsigned long long)a.dy + (unsigned long long)((long long)pbeg * a.Co * 2);
  // (the input base may lie before the tensor for the upper taps of the first pixels: those rows are masked, never fetched)
  unsigned long long xbase = (unsigned long long)((long long)(unsigned long long)a.src + ((long long)pbeg + (long long)dh * a.Ws + dw) * a.Ci * 2);
  long long orec = (long long)(pend - pbeg) * a.Co * 2, xrec = (long long)(pend - pbeg) * a.Ci * 2;
  if MGD_DBG(a, 131072) orec = xrec = 0;       // diagnostic: every LDS-DMA lane out of range - the K-loop without memory traffic
  if MGD_DBG(a, 262144) { orec = min(orec, 4096ll); xrec = min(xrec, 4096ll); }   // diagnostic: only the first rows are fetched (cache hits)
  auto make_srd = [&](unsigned long long base, long long rec) {
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)(rec > 0 ? rec : 0));
    r[3] = 0x00020000;
    return r;
  };

  // per-thread constant byte offsets (row in the 64-pixel step, logical chunk) for both tiles
  const int o_rl = lane / (RBO / 16), o_s = lane % (RBO / 16);
  const int i_rl = lane / (RBI / 16), i_s = lane % (RBI / 16);
  unsigned o_off[OCH], x_off[ICH];
  int x_q[ICH];                                        // the staged input row's pixel index inside its image
#pragma unroll
  for (int i = 0; i < OCH; ++i) {
    const int r = (i * 4 + wave) * ORPI + o_rl;
    const int ch = (((o_s >> 1) ^ tr_swz(r, RBO / 32)) << 1) | (o_s & 1);
    const int c = co0 + ch * 8;
    o_off[i] = c < a.Co ? (unsigned)((r * a.Co + c) * 2) : OOB;
  }
  const int adv = KP % HW;
#pragma unroll
  for (int i = 0; i < ICH; ++i) {
    const int r = (i * 4 + wave) * IRPI + i_rl;
    const int ch = (((i_s >> 1) ^ tr_swz(r, RBI / 32)) << 1) | (i_s & 1);
    const int c = ci0 + ch * 8;
    x_off[i] = c < a.Ci ? (unsigned)((r * a.Ci + c) * 2) : OOB;
    x_q[i] = (pbeg + r) % HW;
  }
  __syncthreads();                                     // bit map published

  // prep(): the input rows' offsets of the NEXT stage to issue (bit test against the map), behind the first MFMAs of the
  // previous step; fire(): the six LDS-DMA instructions, right after the barrier, then the descriptors move on
  // (the map word of a row is fetched one prep() AHEAD: its LDS round trip then lies under a whole K-step instead of in front
  // of the select that needs it - with the fetch and its lgkmcnt(0) inside prep() the lookup cost 6 % of the kernel)
  unsigned xv[ICH], xw[ICH];
#pragma unroll
  for (int i = 0; i < ICH; ++i) xw[i] = masked ? mbits[x_q[i] >> 5] : 0xFFFFFFFFu;
  auto prep = [&]() {
#pragma unroll
    for (int i = 0; i < ICH; ++i) {
      const unsigned ok = (xw[i] >> (x_q[i] & 31)) & 1u;
      xv[i] = ok ? x_off[i] : OOB;
      x_q[i] += adv;
      x_q[i] -= x_q[i] >= HW ? HW : 0;
      if (masked) xw[i] = mbits[x_q[i] >> 5];
    }
  };
  const unsigned smem_a = lds_addr(smem);
  auto fire = [&](int buf) {
    const unsigned ob = smem_a + buf * STAGE + wave * 1024;
    dma_rows_asm<OCH, 4096>(o_off, make_srd(obase, orec), ob);
    dma_rows_asm<ICH, 4096>(xv, make_srd(xbase, xrec), ob + KP * RBO);
    obase += dstep; orec -= dstep;
    xbase += xstep; xrec -= xstep;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (NR == 2) {
    prep();
    if (nk > 0) fire(0);
    prep();                   // offsets of step 1
  } else {
#pragma unroll
    for (int st = 0; st < NR - 1; ++st) { prep(); fire(st); }
    prep();                   // offsets of stage NR - 1
  }
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  int o_rd[KK][2][MT], i_rd[KK][2][NT];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r0 = kk * 32 + 8 * g + qq + 4 * h;
#pragma unroll
      for (int m = 0; m < MT; ++m) o_rd[kk][h][m] = r0 * RBO + (((wc * MT + m) ^ tr_swz(r0, RBO / 32)) * 32) + pp * 8;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        i_rd[kk][h][n] = KP * RBO + r0 * RBI + (((wi * NT + n) ^ tr_swz(r0, RBI / 32)) * 32) + pp * 8;
    }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // fragment read addresses are per-lane constants; the ring slot is an immediate offset (the loop is unrolled by two)
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MT; ++m) o_rd[kk][h][m] += smem_a;
#pragma unroll
      for (int n = 0; n < NT; ++n) i_rd[kk][h][n] += smem_a;
    }
  auto kstep = [&](auto bc, int ks) {
    constexpr int BUF = decltype(bc)::value;
    if (!MGD_DBG(a, 2097152)) {                  // diagnostic: 2097152 = no wait / barrier per K-step
      wait_vmcnt<(NR - 2) * GRP>();
      __builtin_amdgcn_s_barrier();
    }
    const bool more = ks + 1 < nk;
    const bool no_reads = MGD_DBG(a, 4194304);     // diagnostic: no fragment reads (MFMAs on whatever the registers hold)
    // the ring slot as the instruction's immediate offset where it fits its 16 bits, else added to the address
    constexpr int IMM = BUF * STAGE < 65536 ? BUF * STAGE : 0;
    constexpr unsigned EXTRA = (unsigned)(BUF * STAGE - IMM);
    const bool fire_late = MGD_DBG(a, 1048576);    // diagnostics: 524288 = no LDS-DMA at all, 1048576 = issue it behind the first MFMAs
    auto do_fire = [&]() {
      if MGD_DBG(a, 524288) return;
      if constexpr (NR == 2) { if (more) fire(BUF ^ 1); }
      else fire((BUF + NR - 1) % NR);        // into the slot every wave left before this barrier (a dummy stage past the end)
    };
    const bool fire_mid = MGD_DBG(a, 33554432);    // diagnostic: LDS-DMA issue in the shadow of the first fragment reads' latency
    if (!fire_late && !fire_mid) do_fire();
    if constexpr (KK == 2) {
      s16x4 fa[2][MT][2], fb[2][NT][2];
      auto read_half = [&](int kk) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          tr_read_asm<IMM>(fa[kk][m][0], (unsigned)o_rd[kk][0][m] + EXTRA);
          tr_read_asm<IMM>(fa[kk][m][1], (unsigned)o_rd[kk][1][m] + EXTRA);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          tr_read_asm<IMM>(fb[kk][n][0], (unsigned)i_rd[kk][0][n] + EXTRA);
          tr_read_asm<IMM>(fb[kk][n][1], (unsigned)i_rd[kk][1][n] + EXTRA);
        }
      };
      auto mfma_half = [&](int kk) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { touch(fa[kk][m][0]); touch(fa[kk][m][1]); }
#pragma unroll
        for (int n = 0; n < NT; ++n) { touch(fb[kk][n][0]); touch(fb[kk][n][1]); }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          s16x8 av = __builtin_shufflevector(fa[kk][m][0], fa[kk][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            s16x8 bv = __builtin_shufflevector(fb[kk][n][0], fb[kk][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                               acc[m][n], 0, 0, 0);
          }
        }
      };
      if constexpr (IL) {
        // interleaved form: the pixel-operand fragments first, then the channel rows one by one; a row's MFMAs start as soon
        // as ITS fragments are back (LDS operations return in order: counted lgkmcnt), and the second half's reads go out one
        // or two per MFMA gap instead of as a burst in front of the first MFMA
        constexpr int R1 = 2 * MT + 2 * NT, G = MT * NT;
        auto rd = [&](int kk, int q) {           // read q of half kk: 0 .. 2NT-1 the pixel operand, then the rows
          if (q < 2 * NT) tr_read_asm<IMM>(fb[kk][q >> 1][q & 1], (unsigned)i_rd[kk][q & 1][q >> 1] + EXTRA);
          else { const int r = q - 2 * NT; tr_read_asm<IMM>(fa[kk][r >> 1][r & 1], (unsigned)o_rd[kk][r & 1][r >> 1] + EXTRA); }
        };
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
          s16x8 av = __builtin_shufflevector(fa[0][m][0], fa[0][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            s16x8 bv = __builtin_shufflevector(fb[0][n][0], fb[0][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                               acc[m][n], 0, 0, 0);
            const int g = m * NT + n;
            while (q1 < (g + 1) * R1 / G) { rd(1, q1); ++q1; }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        prep();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          wait_lgkm_dyn(2 * (MT - 1 - m));       // (the bit-map word prep() may have asked for is younger: waited for too)
          if (m == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n) { touch(fb[1][n][0]); touch(fb[1][n][1]); }
          }
          touch(fa[1][m][0]); touch(fa[1][m][1]);
          s16x8 av = __builtin_shufflevector(fa[1][m][0], fa[1][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            s16x8 bv = __builtin_shufflevector(fb[1][n][0], fb[1][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                               acc[m][n], 0, 0, 0);
          }
        }
        wait_lgkm_dyn(0);
      } else {
      if (!no_reads) read_half(0);
      if (fire_mid) do_fire();
      wait_lgkm_dyn(0);
      if (!no_reads) read_half(1);          // in flight under the MFMAs of the first half
      mfma_half(0);
      if (fire_late) do_fire();
      prep();                // offsets of step ks + 2 (its LDS word is back long before the next fire)
      wait_lgkm_dyn(0);
      mfma_half(1);
      }
    } else {
      // one 32-pixel k-step per stage: the halves are the lower / upper channel-row tiles of the wave
      constexpr int MH = MT / 2;
      s16x4 fa[MT][2], fb[NT][2];
      auto read_rows = [&](int h) {
#pragma unroll
        for (int m = h * MH; m < (h + 1) * MH; ++m) {
          tr_read_asm<IMM>(fa[m][0], (unsigned)o_rd[0][0][m] + EXTRA);
          tr_read_asm<IMM>(fa[m][1], (unsigned)o_rd[0][1][m] + EXTRA);
        }
      };
      auto mfma_rows = [&](int h) {
#pragma unroll
        for (int m = h * MH; m < (h + 1) * MH; ++m) { touch(fa[m][0]); touch(fa[m][1]); }
        if (h == 0) {
#pragma unroll
          for (int n = 0; n < NT; ++n) { touch(fb[n][0]); touch(fb[n][1]); }
        }
#pragma unroll
        for (int m = h * MH; m < (h + 1) * MH; ++m) {
          s16x8 av = __builtin_shufflevector(fa[m][0], fa[m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            s16x8 bv = __builtin_shufflevector(fb[n][0], fb[n][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                               acc[m][n], 0, 0, 0);
          }
        }
      };
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        tr_read_asm<IMM>(fb[n][0], (unsigned)i_rd[0][0][n] + EXTRA);
        tr_read_asm<IMM>(fb[n][1], (unsigned)i_rd[0][1][n] + EXTRA);
      }
      read_rows(0);
      wait_lgkm_dyn(0);
      read_rows(1);          // in flight under the MFMAs of the lower rows
      mfma_rows(0);
      prep();
      wait_lgkm_dyn(0);
      mfma_rows(1);
    }
  };
  if constexpr (NR == 2) {
    for (int ks = 0; ks < nk; ks += 2) {
      kstep(std::integral_constant<int, 0>{}, ks);
      if (ks + 1 < nk) kstep(std::integral_constant<int, 1>{}, ks + 1);
    }
  } else {
    for (int ks = 0; ks < nk; ks += NR) {
      kstep(std::integral_constant<int, 0>{}, ks);
      if (ks + 1 < nk) kstep(std::integral_constant<int, 1>{}, ks + 1);
      if (ks + 2 < nk) kstep(std::integral_constant<int, 2>{}, ks + 2);
      if constexpr (NR >= 4) { if (ks + 3 < nk) kstep(std::integral_constant<int, 3>{}, ks + 3); }
    }
    wait_vmcnt<0>();                          // the dummy stages' zero writes land before the wave ends
  }
  const int fr = lane & 15, fq = lane >> 4;
  if MGD_DBG(a, 32) return;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + (wc * MT + m) * 16 + fq * 4 + r;
      if (co >= a.Co) continue;
      float* row = a.dw + ((long long)co * a.ntaps + tap) * a.Ci;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ci = ci0 + (wi * NT + n) * 16 + fr;
        if (ci >= a.Ci) continue;
        if MGD_DBG(a, 16) row[ci] = acc[m][n][r]; else atomicAdd(row + ci, acc[m][n][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient v3 ("patch" form) for 3x3 convs with few input channels (Ci = 32 or 64), stride 1 or 2.
// With so few channels the per-tap blocks of v2 move 3x more LDS-DMA bytes per FLOP than a 128x128 tile and
// re-read dy nine times.  Here one block owns ALL nine taps of a [64 co] x [Ci] slice: per K-step (an
// R x TW = 4 x 16 patch of output pixels of one image) it stages the dy tile and ONE haloed input patch
// ((R-1)s+3) x ((TW-1)s+3) pixels; the nine taps read the same patch at shifted addresses.  Patch pixels
// are stored at a padded pitch (Ci*2 + 32 bytes) so that the transposed fragment reads of any tap spread
// over the LDS banks without a per-address swizzle; out-of-image pixels and the pad lanes are DMA'd from a
// zero page.  9 x MT x NT accumulators per wave; the epilogue adds them to dW with 64-byte-contiguous fp32
// atomics straight from registers (the memory-side atomic unit works on 64-byte requests anyway).
struct Wgrad3Args {
  const bf16_t* src;
  const bf16_t* dy;
  float* dw;
  int N, Hs, Ws, Ci, Hg, Wg, Co;
  int s, R, TW, PC, PP;     // stride, output tile, patch columns, patch pixels
  int tilesH, tilesW, ntiles, per_split, tilesCo;
  int tilesCi;              // ROW form: Ci slices of BCI channels
  int nst;                  // ring stages (2..6), one block per CU
  int dbg;                  // diagnostics: 16 = no atomics
  int stage;                // bytes per ring stage
};

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction only takes an immediate)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
#define MGD_VMC(k) case k: wait_vmcnt<k>(); break;
    MGD_VMC(0) MGD_VMC(1) MGD_VMC(2) MGD_VMC(3) MGD_VMC(4) MGD_VMC(5) MGD_VMC(6) MGD_VMC(7) MGD_VMC(8) MGD_VMC(9)
    MGD_VMC(10) MGD_VMC(11) MGD_VMC(12) MGD_VMC(13) MGD_VMC(14) MGD_VMC(15) MGD_VMC(16) MGD_VMC(17) MGD_VMC(18)
    MGD_VMC(19) MGD_VMC(20) MGD_VMC(21) MGD_VMC(22) MGD_VMC(23) MGD_VMC(24) MGD_VMC(25) MGD_VMC(26) MGD_VMC(27)
    MGD_VMC(28) MGD_VMC(29) MGD_VMC(30) MGD_VMC(31) MGD_VMC(32) MGD_VMC(33) MGD_VMC(34) MGD_VMC(35) MGD_VMC(36)
    MGD_VMC(37) MGD_VMC(38) MGD_VMC(39) MGD_VMC(40) MGD_VMC(41) MGD_VMC(42) MGD_VMC(43) MGD_VMC(44) MGD_VMC(45)
    MGD_VMC(46) MGD_VMC(47) MGD_VMC(48) MGD_VMC(49) MGD_VMC(50) MGD_VMC(51) MGD_VMC(52) MGD_VMC(53) MGD_VMC(54)
    MGD_VMC(55) MGD_VMC(56) MGD_VMC(57) MGD_VMC(58) MGD_VMC(59) MGD_VMC(60)
#undef MGD_VMC
    default: wait_vmcnt<0>(); break;
  }
}

// dy-tile chunk swizzle for eight consecutive pixel rows per 32-lane half: with 128-byte rows the row parity already
// selects the bank half, the XOR supplies the other two bits; 256-byte rows all start on bank 0, the XOR supplies three.
template <int RBO>
__device__ __forceinline__ int w3_swz(int p) {
  return RBO == 128 ? ((p >> 1) & 3) : (p & (RBO / 32 - 1));
}

// PPW = patch pieces (1 KiB LDS-DMA wave-instructions) per wave and stage: every wave issues exactly
// OPW + PPW loads per stage, so the counted vmcnt waits are wave-uniform.
// ROW = true: the form for Ci >= 128 (stride 1).  Nine taps of a 128 x 128 slice do not fit the register file, so a
// block owns the THREE taps of one kernel row (dh = block index) of a [BCO co] x [BCI ci] slice: the patch is the R
// pixel rows h0 + dh - 1 .. of the tile with one halo column either side, read at the three column shifts.  Against
// the per-tap blocks of v2 that is one staged dy tile + one input patch (40 KB) per 3 x 128 x 128 x 64 MACs instead
// of per 128 x 128 x 64, and a third of the barriers.  Measured (MGD_WGRAD_ROW=1, tools/bench_wgrad_row.py): the K-loop
// is 22 % faster than v2's (128->256 at 76x76: 88 us against ~113), but a block's partial result is three tiles, so the
// fp32-atomic epilogue moves 3x the bytes (49 MB at the memory-side atomic rate of ~1.2 TB/s = 42 us, a third of the
// launch) and the total ties with v2: 130 / 126 / 141 us against 127 / 122 / 148 us.  Hence opt-in.
template <int MT, int NT, int PPW, bool ROW = false>
__global__ __launch_bounds__(256) void conv_wgrad3_kernel(Wgrad3Args a) {
  constexpr int NTAP = ROW ? 3 : 9;
  constexpr int BCO = 2 * MT * 16, BCI = 2 * NT * 16;
  constexpr int RBO = BCO * 2;                 // dy-tile row bytes
  constexpr int PB = BCI * 2, PBP = PB + 32;   // patch pixel bytes, padded pitch
  constexpr int ORPI = 1024 / RBO;             // dy rows per 1-KiB piece
  constexpr int OPW = (64 / ORPI) / 4;         // dy pieces per wave
  constexpr int LPS = OPW + PPW;               // loads per wave and stage
  static_assert(OPW >= 1, "dy tile too narrow");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wi = wave & 1;
  int bx = MGD_DBG(a, 65536) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);   // one split's blocks on one XCD (see conv_wgrad2_kernel)
  const int tco = bx % a.tilesCo; bx /= a.tilesCo;
  int ci0 = 0, dh = 0;
  if (ROW) { ci0 = (bx % a.tilesCi) * BCI; bx /= a.tilesCi; dh = bx % 3; bx /= 3; }
  const int split = bx;
  const int co0 = tco * BCO;
  const int kt0 = split * a.per_split;
  const int kt1 = min(a.ntiles, kt0 + a.per_split);
  const void* zero = (const void*)g_zero_page;
  const int NST = a.nst;

  // ---- per-lane constants of the dy pieces: pixel slot -> (r, c) of the tile, source chunk (tr swizzle as in v2)
  unsigned dy_off[OPW], dy_rc[OPW];
#pragma unroll
  for (int i = 0; i < OPW; ++i) {
    int p = (i * 4 + wave) * ORPI + lane / (RBO / 16);
    int o_s = lane % (RBO / 16);
    int ch = (((o_s >> 1) ^ w3_swz<RBO>(p)) << 1) | (o_s & 1);
    int r = p / a.TW, c = p - r * a.TW;
    bool ok = r < a.R && co0 + ch * 8 < a.Co;
    dy_off[i] = (unsigned)((((long long)r * a.Wg + c) * a.Co + co0 + ch * 8) * 2);
    dy_rc[i] = ok ? ((unsigned)r << 16 | (unsigned)c) : 0x7fff0000u;
  }
  // ---- patch pieces: LDS byte -> (patch pixel, 16-byte chunk); chunks >= PB/16 are the pad
  unsigned p_off[PPW], p_rc[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    int bo = (j * 4 + wave) * 1024 + lane * 16;
    int pp = bo / PBP, chunk = (bo - pp * PBP) >> 4;
    int pr = pp / a.PC, pc = pp - pr * a.PC;
    bool ok = pp < a.PP && chunk < PB / 16;
    p_off[j] = (unsigned)((((long long)pr * a.Ws + pc) * a.Ci + ci0 + chunk * 8) * 2);
    p_rc[j] = ok ? ((unsigned)pr << 16 | (unsigned)pc) : 0x7fff0000u;
  }

  // next tile to issue (wave-uniform counters instead of divisions per K-step)
  int in_, ith, itw;
  {
    const int tiles_img = a.tilesH * a.tilesW;
    in_ = kt0 / tiles_img;
    int rem = kt0 - in_ * tiles_img;
    ith = rem / a.tilesW;
    itw = rem - ith * a.tilesW;
  }
  auto issue = [&](int buf) {
    const int h0 = ith * a.R, w0 = itw * a.TW;
    unsigned char* ob = smem + buf * a.stage;
    unsigned char* ib = ob + 64 * RBO;
    const char* dbase = (const char*)a.dy + ((((long long)in_ * a.Hg + h0) * a.Wg + w0) * a.Co) * 2;
#pragma unroll
    for (int i = 0; i < OPW; ++i) {
      int r = (int)(dy_rc[i] >> 16), c = (int)(dy_rc[i] & 0xffffu);
      bool v = h0 + r < a.Hg && w0 + c < a.Wg;
      const void* g = v ? (const void*)(dbase + dy_off[i]) : zero;
      glds16(g, ob + (i * 4 + wave) * 1024);
    }
    const int hb = ROW ? h0 + dh - 1 : h0 * a.s - 1, wb = w0 * a.s - 1;
    const char* sbase = (const char*)a.src + ((((long long)in_ * a.Hs + hb) * a.Ws + wb) * a.Ci) * 2;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      int pr = (int)(p_rc[j] >> 16), pc = (int)(p_rc[j] & 0xffffu);
      bool v = (unsigned)(hb + pr) < (unsigned)a.Hs && (unsigned)(wb + pc) < (unsigned)a.Ws;
      const void* g = v ? (const void*)(sbase + p_off[j]) : zero;
      glds16(g, ib + (j * 4 + wave) * 1024);
    }
    if (++itw == a.tilesW) {
      itw = 0;
      if (++ith == a.tilesH) { ith = 0; ++in_; }
    }
  };

  f32x4 acc[NTAP][MT][NT];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing: MFMA k index = pixel slot kk*32 + 8g + qq (+4 for the upper half)
  const int g = lane >> 4, qq = (lane & 15) >> 2, pl = lane & 3;
  int o_rd[2][2][MT];   // dy^T fragments  [kk][half][m]
  int p_rd[2][2];       // patch base      [kk][half]   (+ n*32 + tap offset)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // a 32-lane half reads eight CONSECUTIVE pixels (the k index is summed over, any pixel <-> k-slot map that the
      // two operands share is valid): the padded patch pitch (96 / 160 / 288 bytes = 24 / 40 / 72 dwords, all = 8 mod
      // 16) then spreads them over all 64 banks, and w3_swz does the same for the dy tile.  With the (0-3, 8-11) pixel
      // groups of the plain tr-read layout every read of both operands was 2-way conflicted (SQ_LDS_BANK_CONFLICT =
      // half of SQ_LDS_IDX_ACTIVE, profiles/r01_lds_conflicts.txt)
      int p = kk * 32 + 16 * (g >> 1) + 4 * (g & 1) + qq + 8 * h;
      const int swz = w3_swz<RBO>(p);
#pragma unroll
      for (int m = 0; m < MT; ++m)
        o_rd[kk][h][m] = p * RBO + (((wc * MT + m) ^ swz) * 32) + pl * 8;
      int r = p / a.TW, c = p - r * a.TW;
      int pp = ROW ? (r < a.R ? r * a.PC + c + 1 : 1) : (r < a.R ? (r * a.s + 1) * a.PC + c * a.s + 1 : a.PC + 1);
      p_rd[kk][h] = 64 * RBO + pp * PBP + wi * NT * 32 + pl * 8;
    }
  int toff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) toff[t] = ROW ? (t - 1) * PBP : ((t / 3 - 1) * a.PC + (t % 3 - 1)) * PBP;

  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int nkt = kt1 - kt0;
  int ibuf = 0;           // ring slot of the next issue
  for (int s = 0; s < NST - 1 && s < nkt; ++s) { issue(ibuf); ibuf = ibuf + 1 == NST ? 0 : ibuf + 1; }
  int cbuf = 0;           // ring slot being computed
  // The fragment reads are issued from inline asm: for a ds_read_b64_tr_b16 that the compiler can see it puts
  // an s_waitcnt vmcnt(0) in front (it cannot tell the read from the LDS-DMA writes in flight), which would
  // serialise every K-step with the loads of the stages behind it.  With asm the waits are ours: the 18
  // (k-half, tap) groups of a K-step are software-pipelined DEPTH groups ahead with counted lgkmcnt.
  constexpr int DEPTH = ROW ? 1 : (MT * NT >= 4 ? 2 : 3);
  constexpr int NG = 2 * NTAP;
  const unsigned smem_a = lds_addr(smem);
  for (int it = 0; it < nkt; ++it) {
    const int ahead = min(NST - 2, nkt - 1 - it);     // younger stages that may stay in flight
    if (ROW) {                                        // 3-stage ring: one younger stage or none (no branch tree in the loop)
      if (ahead > 0) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
    } else {
      wait_vmcnt_dyn(ahead * LPS);
    }
    __builtin_amdgcn_s_barrier();
    if (it + NST - 1 < nkt) { issue(ibuf); ibuf = ibuf + 1 == NST ? 0 : ibuf + 1; }
    const unsigned sb = smem_a + cbuf * a.stage;
    cbuf = cbuf + 1 == NST ? 0 : cbuf + 1;

    s16x4 ofr[2][MT][2];            // dy^T fragments [kk][m][half]
    s16x4 xbr[DEPTH + 1][NT][2];    // patch fragments, ring over groups
    auto readA = [&](int kk) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        tr_read_asm<0>(ofr[kk][m][0], sb + o_rd[kk][0][m]);
        tr_read_asm<0>(ofr[kk][m][1], sb + o_rd[kk][1][m]);
      }
    };
    auto readB = [&](int gi) {
      const int kk = gi / NTAP, t = gi % NTAP, slot = gi % (DEPTH + 1);
      const unsigned alo = sb + p_rd[kk][0] + toff[t], ahi = sb + p_rd[kk][1] + toff[t];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n == 0)      { tr_read_asm<0>(xbr[slot][n][0], alo); tr_read_asm<0>(xbr[slot][n][1], ahi); }
        else if (n == 1) { tr_read_asm<32>(xbr[slot][n][0], alo); tr_read_asm<32>(xbr[slot][n][1], ahi); }
        else if (n == 2) { tr_read_asm<64>(xbr[slot][n][0], alo); tr_read_asm<64>(xbr[slot][n][1], ahi); }
        else             { tr_read_asm<96>(xbr[slot][n][0], alo); tr_read_asm<96>(xbr[slot][n][1], ahi); }
      }
    };
    readA(0);
#pragma unroll
    for (int gi = 0; gi < DEPTH; ++gi) readB(gi);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int kk = gi / NTAP, t = gi % NTAP, slot = gi % (DEPTH + 1);
      if (gi + DEPTH < NG) {
        if (gi + DEPTH == NTAP) readA(1);
        readB(gi + DEPTH);
      }
      // LDS operations issued after group gi's reads: the younger B groups, and the kk = 1 dy fragments if
      // they went out after this group's reads (gi < 9) and have gone out already (gi + DEPTH >= 9)
      const int younger = (NG - 1 - gi < DEPTH ? NG - 1 - gi : DEPTH) * 2 * NT + ((gi < NTAP && gi + DEPTH >= NTAP) ? 2 * MT : 0);
      wait_lgkm_dyn(younger);
#pragma unroll
      for (int n = 0; n < NT; ++n) { touch(xbr[slot][n][0]); touch(xbr[slot][n][1]); }
      if (t == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { touch(ofr[kk][m][0]); touch(ofr[kk][m][1]); }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        s16x8 av = __builtin_shufflevector(ofr[kk][m][0], ofr[kk][m][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          s16x8 bv = __builtin_shufflevector(xbr[slot][n][0], xbr[slot][n][1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[t][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av),
                                                                __builtin_bit_cast(bf16x8, bv), acc[t][m][n], 0, 0, 0);
        }
      }
    }
  }

  if MGD_DBG(a, 16) { if (acc[0][0][0][0] == 123.456f) a.dw[0] = 1.f; return; }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int co = co0 + (wc * MT + m) * 16 + fq * 4 + r;
          int ci = ci0 + (wi * NT + n) * 16 + fr;
          int tap = ROW ? dh * 3 + t : t;
          if (co < a.Co && ci < a.Ci) {
            float* q = a.dw + ((long long)co * 9 + tap) * a.Ci + ci;
            // MGD_DBG=32: workgroup-scope atomics - measured identical in time (and result): no faster L2-side path
            if MGD_DBG(a, 32) __hip_atomic_fetch_add(q, acc[t][m][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(q, acc[t][m][n][r]);
          }
        }
}

// ------------------------------------------------------------------------------------------------
// Stem: 3x3, 3 -> 32, fp32 image in, bf16 out.  HBM-bound (writes 64 B per pixel); direct VALU.
// Stem forward on the matrix cores, straight from the fp32 image (no im2col image in HBM): a block owns 4 rows x 64
// pixels of one image; the haloed fp32 patch (6 x 66 pixels x 3 channels) is staged in LDS with coalesced loads; a
// wave takes one row and builds, 16 pixels at a time, the B fragment of v_mfma_f32_16x16x32_bf16 from eight LDS
// words per lane (k = tap*3 + c; the three taps of a kernel row are nine consecutive floats of the patch row) -
// K = 27 of 32, weights as two A fragments held in registers.  Same arithmetic as the im2col + GEMM path it replaces
// (bf16-rounded image and weights, fp32 accumulation), reads 71 MB + writes 378 MB instead of 71+378 + 378+378 MB.
template <int WC, int WI, int MT, int NT>
int launch_wgrad(WgradArgs& a, hipStream_t st) {
  constexpr int BCO = WC * MT * 16, BCI = WI * NT * 16;
  a.tilesCo = cdiv(a.Co, BCO);
  a.tilesCi = cdiv(a.Ci, BCI);
  size_t lds = (size_t)64 * (BCO + BCI) * 2 * 2;       // the two-stage ring; the epilogue needs no LDS
  auto k = conv_wgrad2_kernel<WC, WI, MT, NT>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    attr = true;
  }
  int nblk = a.tilesCo * a.tilesCi * a.ntaps * a.splits;
  hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, st, a);
  return 0;
}

template <int WC, int WI, int MT, int NT, int NR = 2, int KP = 64, bool IL = false>
int launch_wgrad4(WgradArgs& a, hipStream_t st) {
  constexpr int BCO = WC * MT * 16, BCI = WI * NT * 16;
  a.tilesCo = cdiv(a.Co, BCO);
  a.tilesCi = cdiv(a.Ci, BCI);
  size_t lds = (size_t)KP * (BCO + BCI) * 2 * NR + (size_t)((a.Hg * a.Wg + 31) / 32) * 4 + 16;   // ring + the tap's bit map
  auto k = conv_wgrad4_kernel<WC, WI, MT, NT, NR, KP, IL>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    attr = true;
  }
  int nblk = a.tilesCo * a.tilesCi * a.ntaps * a.splits;
  hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, st, a);
  return 0;
}

template <int MT, int NT, int PPW>
int launch_wgrad3(Wgrad3Args& a, hipStream_t st) {
  constexpr int BCO = 2 * MT * 16;
  a.tilesCo = cdiv(a.Co, BCO);
  a.stage = 64 * BCO * 2 + 4 * PPW * 1024;
  constexpr int OPW = (64 / (1024 / (BCO * 2))) / 4;
  // measured: two blocks per CU win for Ci = 32 (short MFMA phases), one block with a deep ring for Ci = 64
  int per_cu = (NT == 1 && 2 * a.stage <= 78 * 1024) ? 2 : 1;
  a.nst = std::min(6, ((per_cu == 2 ? 78 : 156) * 1024) / a.stage);           // ring as deep as LDS allows
  while (a.nst > 2 && (a.nst - 2) * (OPW + PPW) > 60) --a.nst;   // vmcnt is a 6-bit counter
  size_t lds = (size_t)a.nst * a.stage;
  int splits = std::max(1, 256 * per_cu / a.tilesCo);
  a.per_split = cdiv(a.ntiles, splits);
  splits = cdiv(a.ntiles, a.per_split);
  auto k = conv_wgrad3_kernel<MT, NT, PPW>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(a.tilesCo * splits), dim3(256), lds, st, a);
  return 0;
}

}  // namespace

extern "C" int mgd_conv_wgrad(const mgd_wgrad_desc* d, void* stream) {
  MGD_REQUIRE(d && d->src && d->dy && d->dw, "wgrad: null pointer");
  MGD_REQUIRE(d->Ci % 8 == 0 && d->Co % 8 == 0, "wgrad: channels must be multiples of 8");
  MGD_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9 && d->splits >= 1, "wgrad: ntaps/splits");
  MGD_REQUIRE((long long)d->N * d->Hg * d->Wg < (1ll << 31), "wgrad: too many pixels");
  bool ok;
  WgradArgs a;
  a.src = (const bf16_t*)d->src; a.dy = (const bf16_t*)d->dy; a.dw = d->dw;
  a.N = d->N; a.Hs = d->Hs; a.Ws = d->Ws; a.Ci = d->Ci; a.Hg = d->Hg; a.Wg = d->Wg; a.Co = d->Co;
  a.in_stride = d->in_stride; a.ntaps = d->ntaps;
  a.tapcode = make_tapcode(d->ntaps, d->dh, d->dw_off, &ok);
  MGD_REQUIRE(ok, "wgrad: tap offsets must lie in [-1,1]");
  a.P = d->N * d->Hg * d->Wg;
  a.splits = d->splits;
  a.chunk = ((cdiv(a.P, a.splits) + 63) / 64) * 64;
  a.splits = cdiv(a.P, a.chunk);
  a.rcp_hw = 1.0f / (float)(d->Hg * d->Wg);
  a.rcp_w = 1.0f / (float)d->Wg;
  a.dbg = MGD_DIAG_FLAGS;
  a.slab = d->partial; a.slab_bytes = d->partial_bytes;
  hipStream_t st = (hipStream_t)stream;
  const int co = d->Co, ci = d->Ci, form = d->form;
  MGD_REQUIRE(form == MGD_WGRAD_AUTO || form == MGD_WGRAD_PERTAP || form == MGD_WGRAD_PATCH || form == MGD_WGRAD_DESC || form == MGD_WGRAD_ROW,
              "wgrad: unknown kernel form %d", form);
  // patch form: 3x3 in the standard tap order, Ci = 32 or 64, output map at least 16 wide
  bool std9 = d->ntaps == 9;
  for (int t = 0; t < 9 && std9; ++t) std9 = d->dh[t] == t / 3 - 1 && d->dw_off[t] == t % 3 - 1;
  // stride-1 'same' geometry (every 1x1, every stride-1 3x3): descriptor-addressed form, no per-pixel address arithmetic.
  // 128 x 64 tiles: 48 KB of LDS, so three blocks share a CU - 8 % faster over the graph than 128 x 128 with two, although a
  // block stages half as many MACs per LDS-DMA byte (three waves per SIMD hide the ring's round trips better).
  const bool lin = d->in_stride == 1 && d->Hs == d->Hg && d->Ws == d->Wg &&
                   (long long)a.chunk * (co > ci ? co : ci) * 2 < (1ll << 31) && d->Hg * d->Wg <= 64 * 1024;
  // kernel-row form (conv_wgrad5.hip): one dy tile and one input tile for the three taps of a kernel row, 128 x 128 x 3 blocks
  {
    const bool can_row = lin && (std9 || d->ntaps == 1) && d->Wg >= 8;
    MGD_REQUIRE(form != MGD_WGRAD_ROW || can_row, "wgrad: the kernel-row form needs a stride-1 'same' layer, 3x3 in the standard tap order or 1x1, at least 8 pixels wide");
    // Own rule (tools/bench_wgrad_forms.py, 608 x 608, batch 16, each launch alone): with a slab workspace it beats the per-tap
    // form on the 3x3 layers with Co >= 128, Ci >= 64 and enough pixels for one block per CU - 128 -> 256 at 76 x 76 82 -> 73 us,
    // 256 -> 512 at 38 x 38 77 -> 72, 512 -> 1024 at 19 x 19 89 -> 77, 64 -> 128 at 152 x 152 124 -> 99 (patch form) - and loses
    // with atomics, on the 1x1 layers and on the head's small maps.
    // (the head's 3x3 layers leave a block only ~10 K-steps: there it loses 38 -> 42 us; hence the pixels-per-block floor)
    const long long row_tiles = (long long)cdiv(co, 128) * cdiv(ci, 128) * 3;
    const bool want_row = form == MGD_WGRAD_AUTO && can_row && std9 && co >= 128 && ci >= 64 && d->partial &&
                          (long long)a.P * row_tiles >= 256ll * 1500 && d->partial_bytes >= mgd_conv_wgrad_workspace_size(d);
    if (form == MGD_WGRAD_ROW || want_row) {
      a.splits = d->form_arg;                     // kernel-row form: form_arg = cap on its blocks (0: 256, one per CU)
      MGD_REQUIRE(mgd::launch_wgrad5(a, st) == MGD_OK, "wgrad: the kernel-row form refused the geometry");
      MGD_CHECK_LAUNCH("conv_wgrad(kernel row)");
      return MGD_OK;
    }
  }
  const bool can_patch = std9 && (ci == 32 || ci == 64) && co >= 32 && d->Wg >= 16 &&
      (d->in_stride == 1 || d->in_stride == 2) && d->Hs == d->Hg * d->in_stride && d->Ws == d->Wg * d->in_stride &&
      (long long)d->N * cdiv(d->Hg, 4) * cdiv(d->Wg, 16) * cdiv(co, 64) >= 256 * 16;   // >= 16 K-steps per block
  MGD_REQUIRE(form != MGD_WGRAD_PATCH || can_patch, "wgrad: the patch form needs a standard 3x3 layer with Ci = 32 / 64 and enough pixels");
  if ((form == MGD_WGRAD_AUTO || form == MGD_WGRAD_PATCH) && can_patch) {
    Wgrad3Args w;
    w.dbg = MGD_DIAG_FLAGS;
    w.src = a.src; w.dy = a.dy; w.dw = a.dw;
    w.N = d->N; w.Hs = d->Hs; w.Ws = d->Ws; w.Ci = ci; w.Hg = d->Hg; w.Wg = d->Wg; w.Co = co;
    w.s = d->in_stride; w.R = 4; w.TW = 16;
    w.PC = (w.TW - 1) * w.s + 3;
    w.PP = ((w.R - 1) * w.s + 3) * w.PC;
    w.tilesH = cdiv(w.Hg, w.R); w.tilesW = cdiv(w.Wg, w.TW);
    w.ntiles = w.N * w.tilesH * w.tilesW;
    // patch pieces per wave = ceil(PP * pitch / 4096): 108 or 297 pixels at 96 / 160 bytes
    if (ci == 32) { if (w.s == 1) launch_wgrad3<2, 1, 3>(w, st); else launch_wgrad3<2, 1, 7>(w, st); }
    else          { if (w.s == 1) launch_wgrad3<2, 2, 5>(w, st); else launch_wgrad3<2, 2, 12>(w, st); }
    MGD_CHECK_LAUNCH("conv_wgrad3");
    return MGD_OK;
  }
  const bool can_desc = lin && co > 32 && ci > 32;
  MGD_REQUIRE(form != MGD_WGRAD_DESC || can_desc, "wgrad: the descriptor-addressed form needs a stride-1 'same' layer with Ci, Co > 32");
  if ((form == MGD_WGRAD_AUTO || form == MGD_WGRAD_DESC) && can_desc) {
    // ring depth (form_arg 2 / 3 / 4; 0 = by layer): the 3x3 launches have nine tap blocks per tile and fill every block slot -
    // there three blocks per CU with a 2-stage ring beat two with three stages (128->256 at 76x76: 86.5 against 108.6 us, one
    // block with four stages 133.4); the 1x1 launches do not fill the slots and take the deeper ring (26.1 -> 21.6 us)
    const int fa = form == MGD_WGRAD_DESC ? d->form_arg : 0;
    MGD_REQUIRE(fa == 0 || fa == 2 || fa == 3 || fa == 4, "wgrad: ring depth (form_arg) must be 0, 2, 3 or 4");
    const int ring = fa ? fa : (d->ntaps == 1 ? 3 : 2);
    if (co > 64 && ci > 64) {
      if (ring == 3) launch_wgrad4<2, 2, 4, 2, 3>(a, st); else if (ring == 4) launch_wgrad4<2, 2, 4, 2, 4>(a, st);
      else launch_wgrad4<2, 2, 4, 2>(a, st);
    } else {
      if (ring == 3) launch_wgrad4<2, 2, 2, 2, 3>(a, st); else launch_wgrad4<2, 2, 2, 2>(a, st);
    }
    MGD_CHECK_LAUNCH("conv_wgrad(descriptor-addressed)");
    return MGD_OK;
  }
  if (co > 64 && ci > 64) launch_wgrad<2, 2, 4, 2>(a, st);
  else if (co > 32 && ci > 32) launch_wgrad<2, 2, 2, 2>(a, st);
  else if (ci <= 32) launch_wgrad<2, 2, 2, 1>(a, st);
  else launch_wgrad<2, 2, 1, 2>(a, st);
  MGD_CHECK_LAUNCH("conv_wgrad");
  return MGD_OK;
}

// Bytes of workspace (mgd_wgrad_desc.partial) with which the kernel-row form stores per-split slabs instead of issuing fp32
// atomics; 0 when that form cannot run the geometry.
extern "C" int64_t mgd_conv_wgrad_workspace_size(const mgd_wgrad_desc* d) {
  if (!d || d->ntaps < 1 || d->ntaps > 9) return 0;
  WgradArgs a{};
  a.N = d->N; a.Hs = d->Hs; a.Ws = d->Ws; a.Ci = d->Ci; a.Hg = d->Hg; a.Wg = d->Wg; a.Co = d->Co;
  a.in_stride = d->in_stride; a.ntaps = d->ntaps;
  a.P = d->N * d->Hg * d->Wg;
  return mgd::wgrad5_workspace_bytes(a);
}
