// Phased 8-wave gather-GEMM (round 4): forward / data gradient of the 3x3 and 1x1 layers with Ci % 64 == 0 on large tiles.
// Replaces Keras Conv2D + its autodiff data gradient (reference multigriddet/models/layers.py:43-49) for those layers.
//
// Why another form.  conv_gemm8_kernel (128 x 128 tile, 4 waves, one stage in flight, vmcnt(0) + barrier per K-step, three
// blocks per CU) and the round-3 forms built on 64 x 64 / 32 x 128 wave tiles all land at 0.9 - 1.0 PFLOP/s alone: a CU pulls
// 32 KiB of operands through its vector-memory path per 128 x 128 x 64 block-step, twice the bytes per FLOP of a 256 x 256
// tile, and every K-step ends in a full drain.  This kernel is the other structure:
//   * ONE block of 8 waves per CU; block tile BC channels x BP pixels (256 x 256, 256 x 192, 128 x 384), wave tiles of
//     128 x 64 or 64 x 96 - 0.375 - 0.42 fragment reads and 0.11 - 0.15 LDS-DMA instructions per MFMA;
//   * BOTH operands by LDS-DMA (buffer_load_dwordx4 ... lds): one kind of vector-memory operation, so s_waitcnt vmcnt(N)
//     counts exactly; the weights come from the fragment-ordered packed image (every 16-row x 32-deep fragment is 1 KiB
//     contiguous: one DMA instruction per fragment, lane-linear conflict-free ds_read_b128), the gathered pixel rows through
//     a raw descriptor over the activation tensor with the XOR swizzle on the per-lane SOURCE address (out-of-image taps are
//     out-of-range offsets: the hardware writes zeros);
//   * two stages of (BC + BP) x 128 B; a K-step (one stage, 64 deep) is TWO phases of MT*NT MFMAs per wave:
//         P0: R = read all pixel fragments b + weight fragments a0, issue DMA;     M = a0 x b
//         P1: R = read weight fragments a1, issue DMA, counted wait;              M = a1 x b
//     waves 0-3 run  R(k) M(k) | barrier,  waves 4-7 run  M(k-1) R(k) | barrier : ONE barrier per phase, and on every SIMD
//     one wave multiplies while its partner reads fragments and issues DMA (first version of this file: four phases of
//     MT*NT/2 MFMAs with two barriers each and the groups one barrier apart - stamps gave 2 860 - 3 000 cycles per K-step
//     of 2 048 MFMA cycles, with or without any DMA traffic: ~100 cycles of every 356-cycle barrier interval were the
//     barrier itself);
//   * DMA stays in flight across the barriers: s_waitcnt vmcnt(N) with N > 0, never 0 in the loop.
//
// Schedule.  Phase k = 2 t + P works on K-step t in stage t & 1; interval k ends with barrier k.
//     R(P0 of t): read b(all), a0;  issue  A_a1(t+1), B1(t+1)  -> stage (t+1) & 1
//     R(P1 of t): read a1;          issue  B0(t+2), A_a0(t+2)  -> stage t & 1;   s_waitcnt vmcnt(pieces of THIS R)
//   (A_a0 / A_a1: the fragments of the waves' first / second half of weight rows, B0 / B1: first / second half of the pixel
//   pieces).  Every R-section ends with s_waitcnt lgkmcnt(0) in front of the next barrier (waves 4-7) or of its own
//   M-section (waves 0-3), so the fragment reads of phase k have returned before barrier k completes.
//   WAR  a region is overwritten by DMA issued in a LATER interval than its last read: b and a0 are last read in P0 of t
//        -> B0(t+2), A_a0(t+2) go out in P1 of t; a1 is last read in P1 of t -> A_a1(t+2) in P0 of t+1; B1(t+2) rides
//        with it only to balance the DMA issue (4 + 4 pieces per K-step at 256 x 256).
//   RAW  stage t+1 is complete when every wave has passed the wait of R(P1 of t) - it leaves only that R's own pieces in
//        flight, everything of stage t+1 is older - and barrier 2t+1; its first read is R(P0 of t+1) in interval 2t+2.
//        "Read a staged buffer one phase after the wait that retires it" (cdna_hip_programming.md, 8-phase template).
// K-steps past the end are DUMMY stages (every lane out of range: zeros land in LDS, nothing leaves the CU), so the counts
// are the same in every iteration; the loop runs over pairs of K-steps (stage parity static), an odd K-loop gets one dummy.
//
// Asm hazard rules used here (conv_common.hpp, "inline-asm hazard checklist"): every DMA statement opens with s_nop 4 (its
// SGPR operands may come from v_readfirstlane), M0 is saved, written and restored inside the statement that uses it with
// s_nop 0 between each write and its DMA, and no DMA has a register destination.
#include "conv_common.hpp"

namespace {

template <int WM, int WN, int MT, int NT>
struct G12 {
  static constexpr int BC = WM * MT * 16, BP = WN * NT * 16;
  static constexpr int NAH = BC / 128;                        // DMA pieces (8 fragments of 1 KiB, one per wave) per half of the weights
  static constexpr int NB = BP / 64, NB0 = (NB + 1) / 2, NB1 = NB / 2;   // pixel pieces (64 rows) per stage, by half
  static constexpr int ABYTES = BC * ROWB, STAGE = (BC + BP) * ROWB;
  static constexpr int INFL = NB0 + NAH;                      // pieces of R(P1) that stay in flight behind its wait
  static constexpr int EPC = BP * (BC / 8) / 512;             // epilogue chunks per thread
  static constexpr int EPI = BP * (BC * 2 + 16) + 8 * 2 * BC * 4;
  static constexpr int AUX = 2 * STAGE > EPI ? 2 * STAGE : EPI;
  static constexpr int LDS = AUX + BP * 16 + 64;
  static_assert(WM * WN == 8 && MT % 2 == 0 && BC % 128 == 0 && BP % 64 == 0, "tile shape");
  static_assert(NB1 >= 1 && NB0 <= 3 && NAH <= 3 && LDS <= 160 * 1024, "pieces / LDS");
};

// DG: instantiation of the diagnostic library only (flags of mgd_diag_set_flags: G12_NO_W = weight DMA out of range, G12_NO_X =
// pixel DMA out of range, G12_NO_EPI = no epilogue, G12_STAMP = per-block s_memtime stamps {start, first stage landed, K-loop
// done, end} to (unsigned long long*)a.partial; tools/diag_gemm12.py)
constexpr int G12_STAMP = 0x08000000, G12_NO_W = 0x10000000, G12_NO_X = 0x20000000, G12_NO_EPI = 0x40000000;
template <int WM, int WN, int MT, int NT, bool DG = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_gemm12_kernel(GemmArgs a) {
  using T = G12<WM, WN, MT, NT>;
  using Epi = GemmEpilogue<WM, WN, MT, NT>;
  constexpr int BC = T::BC, BP = T::BP, NB = T::NB, NAH = T::NAH, MH = MT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long* row_dst = (long long*)(smem + T::AUX);
  uint2* row_src = (uint2*)(smem + T::AUX + BP * 8);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  unsigned long long stamp[4] = {0, 0, 0, 0};
  auto take_stamp = [&](int i) {
    if constexpr (DG) {
      if (a.dbg & G12_STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i]) :: "memory");
    }
  };
  take_stamp(0);
  const bool dg_w = DG && (a.dbg & G12_NO_W), dg_x = DG && (a.dbg & G12_NO_X);
  const int L = xcd_remap(blockIdx.x, a.nblk);
  const int tc = L % a.tilesC, tp = L / a.tilesC;
  const int co0 = tc * BC, pix0 = tp * BP;
  const int nk = sgpr(a.K_pad / BK);
  const int Ci = sgpr(a.Ci), Ws = sgpr(a.Ws);
  const unsigned tc_lo = (unsigned)sgpr((int)(unsigned)a.tapcode), tc_hi = (unsigned)sgpr((int)(unsigned)(a.tapcode >> 32));

  make_row_tables(a, pix0, tid, BP, row_dst, row_src);
  lds_barrier();

  auto make_srd = [&](const void* ptr, long long bytes) {
    const unsigned long long p = (unsigned long long)ptr;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000;
    return r;
  };
  const i32x4 xsrd = make_srd(a.src, (long long)a.N * a.Hs * a.Ws * a.Ci * 2);
  const i32x4 wsrd = make_srd(a.wpk, (long long)a.Co_pad * a.K_pad * 2);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned ldsb = lds_addr(smem);

  // pixel pieces: piece i holds rows 64 i + (tid >> 3); this thread fills 16-byte slot tid & 7 of its row = logical chunk
  // kc = slot ^ (row & 7) of the K-step (the read side applies the same XOR: lds_off)
  const int rlo = tid >> 3;
  const int kc = (tid & 7) ^ (rlo & 7);
  unsigned xoff[NB], vmask[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const uint2 rs = row_src[rlo + 64 * i];
    xoff[i] = rs.x + kc * 16;
    vmask[i] = rs.y;
  }
  // weight pieces: in half h (0: the fragments a0 of every wave row, 1: a1) this wave copies fragment f = i * 8 + wave of
  // piece i: k-half kk = f & 1 of 16-row group G = wm_ * MT + h * MH + m_  ((f >> 1) = wm_ * MH + m_), 1 KiB at byte
  // ((co0/128 + (G >> 3)) * nk + ks) * 16384 + ((G & 7) * 2 + kk) * 1024 of the packed image, to the same place of the stage
  unsigned a_src[2][NAH], a_dst[2][NAH];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < NAH; ++i) {
      const int f = i * 8 + wave, kk = f & 1, idx = f >> 1;
      const int G = (idx / MH) * MT + h * MH + idx % MH;
      a_dst[h][i] = (unsigned)((G >> 3) * 16384 + ((G & 7) * 2 + kk) * 1024);
      a_src[h][i] = ((unsigned)(co0 / 128 + (G >> 3)) * (unsigned)nk) * 16384u + (unsigned)(((G & 7) * 2 + kk) * 1024);
    }
  const unsigned lane16 = lane * 16u;

  auto tap_off = [&](int tp_) {
    const unsigned code = (tp_ < 8 ? tc_lo >> (4 * tp_) : tc_hi >> (4 * (tp_ - 8))) & 15u;
    const int dh = (int)(code & 3) - 1, dw = (int)(code >> 2) - 1;
    return (dh * Ws + dw) * Ci * 2;
  };
  // wave-uniform state of the next pixel stage to go out (K-step s_kb): its tap and the byte offset of (tap, first channel)
  int s_kb = 0, s_tap = 0, s_c0 = 0, s_toff = tap_off(0);
  // chunk-major K order (GemmArgs::korder): the pixel stream and the two weight-half streams each walk (chunk, tap)
  const int nchunk = Ci / BK, ntaps = a.ntaps;
  const bool kmaj = (a.korder & 1) != 0;
  int s_chunk = 0, w_tap0 = 0, w_chunk0 = 0, w_tap1 = 0, w_chunk1 = 0;

  // H = 0 / 1: first / second group of the stage's pixel pieces (NB0 / NB1 of them); both go out in order, B1 closes the stage
  auto issue_b = [&](auto H, int st) {
    constexpr int h = decltype(H)::value;
    constexpr int NP = h ? T::NB1 : T::NB0, P0 = h ? T::NB0 : 0;
    const bool real = s_kb < nk && !dg_x;
    const unsigned bit = 1u << s_tap;
    unsigned vo[NP], dd[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      vo[i] = ((vmask[P0 + i] & bit) && real) ? xoff[P0 + i] + (unsigned)s_toff : OOB;
      dd[i] = ldsb + st * T::STAGE + T::ABYTES + (P0 + i) * 8192 + wave * 1024;
    }
    dma_to<NP>(vo, dd, xsrd);
    if constexpr (h == 1) {
      ++s_kb;
      if (kmaj) {
        if (++s_tap == ntaps) { s_tap = 0; ++s_chunk; }
        s_toff = tap_off(s_tap) + s_chunk * (BK * 2);
      } else {
        s_c0 += BK;
        s_toff += BK * 2;
        if (s_c0 >= Ci) { s_c0 = 0; ++s_tap; s_toff = tap_off(s_tap); }
      }
    }
  };
  auto issue_a = [&](auto H, int ks, int st) {
    constexpr int h = decltype(H)::value;
    const bool real = ks < nk && !dg_w;
    // each half is requested for K-steps 0, 1, 2, ... in order: its own (tap, chunk) walk gives the K block of the image
    int& wt = h ? w_tap1 : w_tap0;
    int& wc_ = h ? w_chunk1 : w_chunk0;
    const int wk = kmaj ? wt * nchunk + wc_ : ks;
    if (kmaj) { if (++wt == ntaps) { wt = 0; ++wc_; } }
    unsigned vo[NAH], dd[NAH];
#pragma unroll
    for (int i = 0; i < NAH; ++i) {                           // (everything in the per-lane offset: the range check covers it)
      vo[i] = real ? a_src[h][i] + (unsigned)wk * 16384u + lane16 : OOB;
      dd[i] = ldsb + st * T::STAGE + a_dst[h][i];
    }
    dma_to<NAH>(vo, dd, wsrd);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses.  Weights: fragment (G, kk) of the block's A region = 1 KiB at (G >> 3) * 16384 + ((G & 7) * 2 + kk) *
  // 1024, lane l's 16 bytes at + 16 l.  Pixels: row (wn * NT + n) * 16 + (lane & 15), chunk (lane >> 4) + 4 kk at its slot.
  const int fr = lane & 15, fq = lane >> 4;
  const int a_rd = lane * 16;
  const int b_rd0 = T::ABYTES + lds_off(wn * NT * 16 + fr, fq), b_rd1 = T::ABYTES + (lds_off(wn * NT * 16 + fr, fq) ^ 64);
  bf16x8 af[MH][2], bfr[NT][2];
  auto read_b = [&](int st) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) bfr[n][kk] = *(const bf16x8*)(smem + (kk ? b_rd1 : b_rd0) + st * T::STAGE + n * 2048);
  };
  auto read_a = [&](auto HH, int st) {
    constexpr int h = decltype(HH)::value;
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int G = wm * MT + h * MH + m;
        af[m][kk] = *(const bf16x8*)(smem + a_rd + st * T::STAGE + (G >> 3) * 16384 + ((G & 7) * 2 + kk) * 1024);
      }
  };
  // R-section of phase (P, K-step t in stage st)
  auto R = [&](auto P, auto ST, int t) {
    constexpr int p = decltype(P)::value, st = decltype(ST)::value;
    if constexpr (p == 0) {
      read_b(st);
      read_a(H0{}, st);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(H1{}, t + 1, st ^ 1);
      issue_b(H1{}, st ^ 1);
    } else {
      read_a(H1{}, st);
      __builtin_amdgcn_sched_barrier(0);
      issue_b(H0{}, st);
      issue_a(H0{}, t + 2, st);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::INFL) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // M-section of phase P: (the weight half now in af) x (all pixel fragments) x 64-deep K
  auto M = [&](auto P) {
    constexpr int p = decltype(P)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[p * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][kk], bfr[n][kk], acc[p * MH + m][n], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: stage 0 whole, then what R(P1 of K-step -1) would have sent of stage 1
  issue_b(H0{}, 0); issue_a(H0{}, 0, 0);
  issue_a(H1{}, 0, 0); issue_b(H1{}, 0);
  issue_b(H0{}, 1); issue_a(H0{}, 1, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::INFL) : "memory");
  take_stamp(1);
  bar();                                                      // stage 0 is complete for everyone

  const int nk2 = (nk + 1) & ~1;                              // an odd K-loop: one dummy stage of zeros
  if (wave < 4) {
    for (int t = 0; t < nk2; t += 2) {
      R(H0{}, H0{}, t);     M(H0{}); bar();
      R(H1{}, H0{}, t);     M(H1{}); bar();
      R(H0{}, H1{}, t + 1); M(H0{}); bar();
      R(H1{}, H1{}, t + 1); M(H1{}); bar();
    }
  } else {
    R(H0{}, H0{}, 0); bar();
    for (int t = 0; t < nk2; t += 2) {
      M(H0{}); R(H1{}, H0{}, t);     bar();
      M(H1{}); R(H0{}, H1{}, t + 1); bar();
      M(H0{}); R(H1{}, H1{}, t + 1); bar();
      M(H1{});
      if (t + 2 < nk2) { R(H0{}, H0{}, t + 2); bar(); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the dummy stages still in flight write zeros into the ring
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  take_stamp(2);
  constexpr int G = T::EPC % 4 == 0 ? 4 : (T::EPC % 3 == 0 ? 3 : 2);
  Epi epi;
  if (!(DG && (a.dbg & G12_NO_EPI))) epi.template run_grouped<G>(a, acc, smem, row_dst, co0, tid);
  if constexpr (DG) {
    if (a.dbg & G12_NO_EPI) {                                 // keep the accumulators alive
      f32x4 t = acc[0][0];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) t += acc[m][n];
      if (t[0] == 123.456f) ((float*)a.dst)[tid] = t[1];
    }
    if (a.dbg & G12_STAMP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      take_stamp(3);
      if (tid == 0) {
        unsigned long long* o = (unsigned long long*)a.partial + (size_t)blockIdx.x * 4;
        o[0] = stamp[0]; o[1] = stamp[1]; o[2] = stamp[2]; o[3] = stamp[3];
      }
    }
  }
}

template <int WM, int WN, int MT, int NT, bool DG = false>
int launch12(GemmArgs& a, hipStream_t st) {
  using T = G12<WM, WN, MT, NT>;
  if (a.Co_pad % T::BC) return MGD_EINVAL;
  a.tilesC = a.Co_pad / T::BC;
  a.nblk = a.tilesC * cdiv(a.M, T::BP);
  auto k = conv_gemm12_kernel<WM, WN, MT, NT, DG>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(a.nblk), dim3(512), T::LDS, st, a);
  return MGD_OK;
}

}  // namespace

namespace mgd {

// shape: 0 = 256 channels x 256 pixels (wave tiles 128 x 64), 1 = 256 x 192 (64 x 96), 2 = 128 x 384 (64 x 96).
// Needs bf16 output, Ci % 64 == 0 (a K-step never straddles taps; K_pad == ntaps * Ci), Co_pad % (block channels) == 0.
int launch_gemm12(GemmArgs& a, int shape, int kranges, hipStream_t st) {
  (void)kranges;
  if (a.dst_f32 || a.Ci % 64 || a.K_pad != a.ntaps * a.Ci) return MGD_EINVAL;
#ifdef MGD_DIAG
  if (a.dbg & (G12_STAMP | G12_NO_W | G12_NO_X | G12_NO_EPI)) {
    switch (shape) {
      case 0: return launch12<2, 4, 8, 4, true>(a, st);
      case 1: return launch12<4, 2, 4, 6, true>(a, st);
      case 2: return launch12<2, 4, 4, 6, true>(a, st);
      default: return MGD_EINVAL;
    }
  }
#endif
  switch (shape) {
    case 0: return launch12<2, 4, 8, 4>(a, st);
    case 1: return launch12<4, 2, 4, 6>(a, st);
    case 2: return launch12<2, 4, 4, 6>(a, st);
    default: return MGD_EINVAL;
  }
}

}  // namespace mgd
