// Shared pieces of the convolution engine (gfx950): argument blocks, LDS layout helpers, the inline-asm wrappers with their
// hazard rules, and the common epilogue.  Included by conv_gemm.hip, conv_gemm12.hip, conv_wgrad.hip, conv_stem.hip and the
// diagnostic library (diag/).  Everything except the argument blocks has internal linkage (one copy per translation unit).
#pragma once
#include "common.h"
#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include <type_traits>

// Diagnostic switches exist only in libmgd_hip_diag.so (the same sources built with -DMGD_DIAG, see diag/): there
// MGD_DBG(a, bits) tests the flags mgd_diag_set_flags() stored; in the product library it is the constant 0, so no
// diagnostic branch, ablation instantiation or stamp survives in a product kernel, and nothing reads the environment.
#ifdef MGD_DIAG
extern "C" int mgd_diag_flags_value(void);
#define MGD_DBG(a, bits) (((a).dbg & (bits)) != 0)
#define MGD_DIAG_FLAGS mgd_diag_flags_value()
#else
#define MGD_DBG(a, bits) (false)
#define MGD_DIAG_FLAGS 0
#endif

namespace mgd {

constexpr int BK = 64;          // K elements per stage (2 MFMA k-steps of 32)
constexpr int ROWB = BK * 2;    // bytes per LDS tile row (128)

struct GemmArgs {
  const bf16_t* src;
  const bf16_t* wpk;
  void* dst;
  const float* bias;
  const bf16_t* addend;
  float* stats;
  int N, Hs, Ws, Ci, Hg, Wg, Hd, Wd, Co;
  int in_stride, out_stride, out_off_h, out_off_w;
  int ntaps;
  unsigned long long tapcode;  // 4 bits per tap: (dh+1) | (dw+1)<<2
  int K_pad, Co_pad, dst_f32, stats_replicas;
  int M;        // N*Hg*Wg
  int tilesC;   // Co_pad / BNC
  int nblk;
  int aux;      // conv_gemm8_kernel: byte offset of row_dst in LDS (behind the ring / the epilogue tile)
  int dbg;      // diagnostic only (MGD_DBG): 1 = all LDS-DMA loads hit one cache line; 32...2048: ablation build of conv_gemm8_kernel
  // fused BN-backward reduction of the CONSUMER layer over the tile just produced (dst = da of that layer)
  const bf16_t* bn_y;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
  float* bn_sums;
  float bn_slope;
  float act_slope;   // != 0: LeakyReLU on (acc + bias) before the addend (BatchNorm-folded inference)
  unsigned rowmask, colmask;   // conv_gemm9_kernel: 3 x 9 bits, the taps with dh + 1 == j / dw + 1 == j (bits 9j .. 9j+8)
  int splitk;                  // conv_gemm8_kernel: K ranges per tile (blocks = nblk * splitk); fp32 partial tiles go to
  long long slab_elems;        //   (float*)dst + range * slab_elems
  float* partial;              // conv_gemm11_kernel: fp32 partial tiles [range][tile][128 x 64 in fragment order]
  unsigned* tickets;           //   and one arrival counter per tile (zero between launches)
  // conv_gemm8_kernel, ncls > 1: ONE launch over several tap classes of one geometry (the four output-parity classes of a
  // stride-2 data gradient): block b works on class b / nblk with that class's packed image, taps and output offset
  int ncls;
  // K-step order (kernels with wave-uniform K-steps, Ci % 64 == 0): 0 = tap-major (tap, then the 64-channel chunks of the
  // pixel rows: the packed image's own order), 1 = chunk-major (64-channel chunk, then its taps).  Chunk-major re-reads one
  // 128-byte segment of every pixel row for all taps before moving on: the re-use distance is a quarter / an eighth of the
  // tap-major one and stays inside an XCD's 4-MB L2 with 96 blocks in flight (measured: 256 -> 128 at 76 x 76, data gradient,
  // 338 -> ... MB fetched per launch for 47 MB of operand).
  int korder;
  const bf16_t* c_wpk[4];
  unsigned long long c_tapcode[4];
  int c_K_pad[4], c_ntaps[4], c_off_h[4], c_off_w[4];
  unsigned c_rowmask[4], c_colmask[4];
};

struct WgradArgs {
  const bf16_t* src;
  const bf16_t* dy;
  float* dw;
  int N, Hs, Ws, Ci, Hg, Wg, Co;
  int in_stride, ntaps;
  unsigned long long tapcode;
  int P;          // N*Hg*Wg
  int chunk;      // pixels per split (multiple of 64)
  int splits, tilesCo, tilesCi;
  float rcp_hw, rcp_w;
  int dbg;        // diagnostics (MGD_DBG): 16 = plain stores instead of atomics, 32 = no epilogue at all
  float* slab;            // conv_wgrad5_kernel: caller's workspace for the splits' fp32 slabs (null: fp32 atomics)
  long long slab_bytes;
};

// conv_wgrad5.hip: kernel-row weight gradient (128 x 128 x 3 taps per block, 8 waves); MGD_EINVAL when the geometry does not apply
int launch_wgrad5(WgradArgs& a, hipStream_t st);
long long wgrad5_workspace_bytes(WgradArgs a);

// conv_gemm12.hip: the phased 8-wave gather-GEMM (shape: see the launcher); returns 0 or MGD_EINVAL when the shape does not apply
int launch_gemm12(GemmArgs& a, int shape, int kranges, hipStream_t st);

}  // namespace mgd

namespace {

using mgd::GemmArgs;
using mgd::BK;
using mgd::ROWB;

// A kernel that promises at least two waves per SIMD has at most 256 registers, and only then does hipcc put MFMA
// accumulators in ordinary VGPRs.  With launch_bounds(256) alone the budget is 512 (256 VGPR + 256 AGPR), the MFMAs are
// selected in their AGPR form, and every loop whose accumulators cross a control-flow merge moves ALL of them AGPR <-> VGPR
// each iteration: conv_wgrad4_kernel carried 32 v_accvgpr_write + 32 v_accvgpr_read per 16 MFMAs (round 3, ISA listing).
#define MGD_VGPR_MFMA __attribute__((amdgpu_waves_per_eu(2)))

__device__ __forceinline__ int lds_off(int row, int kc) { return row * ROWB + ((kc ^ (row & 7)) << 4); }

// Row tables of one pixel tile, one pixel per thread (tid < BMP): the destination element offset of the pixel's output row
// (-1 past the end) and, for the gather, the byte offset of its centre source pixel with the 9-bit set of taps that stay
// inside the image = taps with a valid row AND a valid column (a.rowmask / a.colmask: the taps by dh + 1 / dw + 1).
// Round 3: every gather-GEMM builds its staged rows from these tables - before, each thread derived (image, row, column) of
// each of its 2-4 staged rows itself and looped over the taps: two divisions and a 9-iteration loop per row, 1 500 vector
// instructions per wave and tile, as many as an 18-step K-loop (PMC: profiles/r03_pmc_instruction_mix.txt).
__device__ __forceinline__ void make_row_tables(const GemmArgs& a, int pix0, int tid, int bmp, long long* row_dst, uint2* row_src) {
  if (tid < bmp) {
    const int m = pix0 + tid;
    long long off = -1;
    unsigned xo = 0, vm = 0;
    if (m < a.M) {
      const int hw = a.Hg * a.Wg;
      const int n = m / hw, rem = m - n * hw;
      const int ig = rem / a.Wg, jg = rem - ig * a.Wg;
      const int hd = ig * a.out_stride + a.out_off_h, wd = jg * a.out_stride + a.out_off_w;
      off = (((long long)n * a.Hd + hd) * a.Wd + wd) * a.Co;
      const int hs = ig * a.in_stride, ws = jg * a.in_stride;
      xo = (unsigned)(((((long long)n * a.Hs + hs) * a.Ws + ws) * a.Ci) * 2);
      unsigned rsel = 0, csel = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if ((unsigned)(hs + j - 1) < (unsigned)a.Hs) rsel |= (a.rowmask >> (9 * j)) & 0x1FFu;
        if ((unsigned)(ws + j - 1) < (unsigned)a.Ws) csel |= (a.colmask >> (9 * j)) & 0x1FFu;
      }
      vm = rsel & csel;
    }
    row_dst[tid] = off;
    row_src[tid] = make_uint2(xo, vm);
  }
}

// ------------------------------------------------------------------------------------------------
// Gather-GEMM: LDS-DMA (global_load_lds) staging into an NST-deep LDS ring with counted vmcnt and one raw
// s_barrier per K-step - the loads of NST-1 stages stay in flight across barriers, which hides the L2/HBM
// latency a register-staged loop exposes (~2 us per K-step).
// LDS image per stage: W tile [BNC][64 bf16] then X tile [128][64 bf16], rows of 128 B, the 16-byte
// chunk kc of row r stored at slot kc ^ (r & 7).  LDS-DMA writes lane-linear (base + lane*16), so the
// swizzle is applied on the per-lane SOURCE address; zero padding comes from a zero page in HBM.
// All LDS lives in one dynamic array (a second __shared__ object makes hipcc drain vmcnt early).
__device__ uint4 g_zero_page[8];

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// Workgroup barrier for data handed over through LDS only: waits for this wave's LDS operations, not for its global
// loads / stores.  __syncthreads() also drains vmcnt: after an epilogue's store loop that is the full HBM write
// acknowledgement latency (stamps: 16-19 thousand cycles per 128x128 tile, as long as an 18-step K-loop), during which the
// block holds its CU slot for nothing.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// phase boundary of a hand-scheduled loop: hipcc may move neither MFMAs nor LDS reads across it
__device__ __forceinline__ void phase_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  lds_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the same wait through the builtin: hipcc's own wait-count bookkeeping sees it (after an asm wait it still assumes the
// loads outstanding and adds a full s_waitcnt vmcnt(0) in front of their first use)
template <int N>
__device__ __forceinline__ void wait_vmcnt_tracked() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// ================================================================================================
// Inline-asm wrappers.  EVERY vector-memory instruction issued from inline asm in the product library lives in this block
// (tests/test_asm_hazards.py greps the sources for strays and checks the disassembly of the built library).
//
// INLINE-ASM HAZARD CHECKLIST (gfx950; hipcc pads nothing inside an asm statement and does not count its memory operations -
// cdna_hip_programming.md 5.7).  Each rule names the bug that taught it.
//  H1  VALU write of an SGPR (v_readfirstlane, v_readlane, v_cmp with an SGPR destination) -> vector-memory instruction that
//      reads it as descriptor, scalar offset or base: FIVE wait states.  Every statement below that takes "s" operands opens
//      with `s_nop 4`.  (Round 3: conv_gemm10_kernel faulted on launches of more than 128 blocks - "Memory access fault by
//      GPU node-2, address (nil)" - with an SGPR base straight from v_readfirstlane.)
//  H2  s_mov / s_add of M0 -> LDS-DMA (buffer_load ... lds, global_load_lds) that uses it: ONE wait state (`s_nop 0`).  M0 is
//      the compiler's: save it, write it and restore it inside the SAME statement.
//  H3  store of more than 64 bits (global_store_dwordx3/x4, buffer_store_dwordx3/x4): its data registers are read for TWO more
//      wait states, so the statement ends with `s_nop 1`.  (Round 3: the latency form's partial tiles differed from run to run
//      until the nop went in - the next accumulator was moved into v[4:7] right behind the store.)
//  H4  an asm load with a register destination is invisible to hipcc's wait counting: "=&v" outputs, the wait in a statement
//      that names every destination "+v" (wait_a4, touch), and nothing of the compiler's between load and wait may merge
//      differently allocated paths (no `if` ladder, no `break`: hipcc moves pending destinations through v_mov copies there -
//      NaNs on large shapes only, round 3).
//  H5  LDS-DMA and register loads do NOT retire in issue order with respect to each other: a counted vmcnt only proves the
//      older stage when ONE kind of load is in flight (conv_gemm11 / conv_gemm12: all DMA), or use vmcnt(0) a whole step
//      after the issue (the ping-pong form).
//  H6  LDS-DMA data is ordered for a ds_read only by the issuing waves' vmcnt wait FOLLOWED by a barrier the reader has
//      passed; an LDS region is re-staged only after the reads of it have returned (lgkmcnt) and a barrier.
//  H7  ds_read_b64_tr_b16 from asm: EXEC all ones, 8-byte aligned addresses, `s_waitcnt lgkmcnt` + an empty "+v" statement
//      (touch) before the first consumer - hipcc hoists register-only MFMAs above an asm wait otherwise.

// XCH pixel pieces of one stage: piece i goes to LDS byte address lds_dst + i * STRIDE (wave-uniform; lane l lands at + 16 l).
// M0 is the compiler's: saved, set and restored inside ONE statement.
#define MGD_DMA_FIRST "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, 0 offen lds\n\t"
#define MGD_DMA_NEXT(k) "s_add_u32 m0, m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %" #k ", %2, 0 offen lds\n\t"
#define MGD_DMA_LAST "s_mov_b32 m0, %0"
template <int XCH, int STRIDE>
__device__ __forceinline__ void dma_rows_asm(const unsigned (&v)[XCH], i32x4 srd, unsigned lds_dst) {
  unsigned keep;
  static_assert(XCH >= 1 && XCH <= 6, "pixel pieces per wave and stage");
  if constexpr (XCH == 1)
    asm volatile(MGD_DMA_FIRST MGD_DMA_LAST : "=&s"(keep) : "s"(lds_dst), "s"(srd), "n"(STRIDE), "v"(v[0]) : "memory", "scc");
  else if constexpr (XCH == 2)
    asm volatile(MGD_DMA_FIRST MGD_DMA_NEXT(5) MGD_DMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(srd), "n"(STRIDE), "v"(v[0]), "v"(v[1]) : "memory", "scc");
  else if constexpr (XCH == 3)
    asm volatile(MGD_DMA_FIRST MGD_DMA_NEXT(5) MGD_DMA_NEXT(6) MGD_DMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(srd), "n"(STRIDE), "v"(v[0]), "v"(v[1]), "v"(v[2]) : "memory", "scc");
  else if constexpr (XCH == 4)
    asm volatile(MGD_DMA_FIRST MGD_DMA_NEXT(5) MGD_DMA_NEXT(6) MGD_DMA_NEXT(7) MGD_DMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(srd), "n"(STRIDE), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory", "scc");
  else if constexpr (XCH == 5)
    asm volatile(MGD_DMA_FIRST MGD_DMA_NEXT(5) MGD_DMA_NEXT(6) MGD_DMA_NEXT(7) MGD_DMA_NEXT(8) MGD_DMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(srd), "n"(STRIDE), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4])
                 : "memory", "scc");
  else
    asm volatile(MGD_DMA_FIRST MGD_DMA_NEXT(5) MGD_DMA_NEXT(6) MGD_DMA_NEXT(7) MGD_DMA_NEXT(8) MGD_DMA_NEXT(9) MGD_DMA_LAST
                 : "=&s"(keep) : "s"(lds_dst), "s"(srd), "n"(STRIDE), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5])
                 : "memory", "scc");
}

// NP pieces of 1 KiB per wave: piece i = 64 lanes x 16 B from (srd + v[i]) to LDS byte d[i] + 16 * lane (d wave-uniform)
#define MGD_Q_OPEN "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
#define MGD_Q_PIECE(d, v) "s_mov_b32 m0, %" #d "\n\ts_nop 0\n\tbuffer_load_dwordx4 %" #v ", %1, 0 offen lds\n\t"
#define MGD_Q_CLOSE "s_mov_b32 m0, %0"
template <int NP>
__device__ __forceinline__ void dma_to(const unsigned (&v)[NP], const unsigned (&d)[NP], i32x4 srd) {
  unsigned keep;
  static_assert(NP >= 1 && NP <= 3, "pieces per statement");
  if constexpr (NP == 1)
    asm volatile(MGD_Q_OPEN MGD_Q_PIECE(2, 3) MGD_Q_CLOSE : "=&s"(keep) : "s"(srd), "s"(d[0]), "v"(v[0]) : "memory");
  else if constexpr (NP == 2)
    asm volatile(MGD_Q_OPEN MGD_Q_PIECE(2, 4) MGD_Q_PIECE(3, 5) MGD_Q_CLOSE
                 : "=&s"(keep) : "s"(srd), "s"(d[0]), "s"(d[1]), "v"(v[0]), "v"(v[1]) : "memory");
  else
    asm volatile(MGD_Q_OPEN MGD_Q_PIECE(2, 5) MGD_Q_PIECE(3, 6) MGD_Q_PIECE(4, 7) MGD_Q_CLOSE
                 : "=&s"(keep) : "s"(srd), "s"(d[0]), "s"(d[1]), "s"(d[2]), "v"(v[0]), "v"(v[1]), "v"(v[2]) : "memory");
}

// the four weight fragments of a K-step (2 channel groups x 2 k-halves), 1 KiB apart in the fragment-ordered image
// (s_nop 4: the base may come straight from v_readfirstlane - a VALU write of an SGPR needs five wait states before a
// vector-memory instruction reads it as an address, and hipcc pads nothing inside an asm statement; without it the loads of
// conv_gemm10_kernel used a stale SGPR pair now and then: memory access faults that came and went with the launch size)
__device__ __forceinline__ void load_a4_asm(bf16x8 (&f)[2][2], unsigned lane16, const void* sbase) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5 offset:0\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
               "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
               : "=&v"(f[0][0]), "=&v"(f[0][1]), "=&v"(f[1][0]), "=&v"(f[1][1]) : "v"(lane16), "s"(sbase) : "memory");
}

// counted wait that also pins the register set it releases: nothing may read f before this statement
template <int N>
__device__ __forceinline__ void wait_a4(bf16x8 (&f)[2][2]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]) : "n"(N) : "memory");
}


// 16-byte store / load at system scope (sc0 sc1: write-through / cache-bypassing) for tiles that cross XCDs inside a launch
// (latency form).  H3: the store ends with s_nop 1.  H4: the load's destination is "=&v", wait with s_waitcnt vmcnt + touch.
__device__ __forceinline__ void store_sys_b128(const void* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load_sys_b128(f32x4& v, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
}

// ================================================================================================
template <int WC, int WP, int MT, int NT>
struct GemmEpilogue {
  static constexpr int BNC = WC * MT * 16, BMP = WP * NT * 16, NTHR = 64 * WC * WP;
  static constexpr int CPB = BNC / 8;            // 16-byte bf16 chunks per output row
  static constexpr int EPC = BMP * CPB / NTHR;   // chunks per thread
  static_assert(BMP * CPB % NTHR == 0 && NTHR % CPB == 0, "epilogue mapping");
  uint4 ypre[EPC], apre[EPC];
  float bnp[4][8];
  bool bnred, addpre;
  float pr1[8], pr2[8];      // per-channel partial sums carried across tiles (persistent kernels: run<true> + flush)

  __device__ __forceinline__ void init_deferred() {
#pragma unroll
    for (int j = 0; j < 8; ++j) pr1[j] = pr2[j] = 0.f;
  }

  __device__ __forceinline__ void prefetch(const GemmArgs& a, const long long* row_dst, int co0, int tid,
                                           bool sync = true) {
    bnred = a.bn_y != nullptr && !a.dst_f32;
    addpre = a.addend != nullptr && !a.dst_f32;
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
    if (bnred || addpre) {
      if (sync) __syncthreads();
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
    } else {
      // defined on every path, HERE: left undefined, hipcc materialises the zeros at kernel entry and carries them (in
      // scratch, in the persistent kernels) across the K-loop; their reloads in the store loop then wait vmcnt(0)
#pragma unroll
      for (int it = 0; it < EPC; ++it) ypre[it] = apre[it] = make_uint4(0, 0, 0, 0);
    }
  }

  // call after a __syncthreads() that follows the last LDS read of the K-loop.  DEFER: the per-channel sums stay in
  // this thread's registers (pr1/pr2) instead of being reduced and added to global memory - flush() does that once.
  // LATE: fetch the HBM operands (prefetch()) only after the accumulators have gone to LDS - for kernels that run the
  // epilogue with little register headroom and gain nothing from fetching earlier
  template <bool DEFER = false, bool LATE = false>
  __device__ __forceinline__ void run(const GemmArgs& a, f32x4 (&acc)[MT][NT], unsigned char* smem,
                                      const long long* row_dst, int co0, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int fr = lane & 15, fq = lane >> 4;
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
    lds_barrier();
    if (LATE) {
      __builtin_amdgcn_sched_barrier(0);          // keep the fetches below the point where the accumulators die
      prefetch(a, row_dst, co0, tid, false);
    }
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
    const bool stats = a.stats != nullptr;
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r1[j] = r2[j] = 0.f;
    // all LDS reads of the pass are issued before the first store: one LDS round trip per tile instead of two per chunk
    long long offs[EPC];
    uint4 vals[EPC];
#pragma unroll
    for (int it = 0; it < EPC; ++it) offs[it] = row_dst[(tid + it * NTHR) / CPB];
#pragma unroll
    for (int it = 0; it < EPC; ++it) {
      const int q = tid + it * NTHR;
      const int r = q / CPB, ch = q - r * CPB;
      vals[it] = *(const uint4*)(smem + r * EROW + ch * 16);
    }
#pragma unroll
    for (int it = 0; it < EPC; ++it) {
      const int q = tid + it * NTHR;
      const int r = q / CPB, ch = q - r * CPB;
      const long long off = offs[it];
      const int c = co0 + ch * 8;
      if (off < 0 || c >= a.Co) continue;
      uint4 v = vals[it];
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
    if (DEFER) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { pr1[j] += r1[j]; pr2[j] += r2[j]; }
      return;
    }
    reduce_and_add(a, r1, r2, smem, co0, tid, stats, bnred);
  }

  // The same epilogue in groups of G chunks per thread (EPC % G == 0): large tiles have 12 chunks per thread, and their
  // HBM operands (12 + 12 uint4) beside the tile's values and offsets would not fit the register file.  HBM operands are
  // fetched per group, after the accumulators have gone to LDS (the LATE form of run()).
  template <int G>
  __device__ __forceinline__ void run_grouped(const GemmArgs& a, f32x4 (&acc)[MT][NT], unsigned char* smem,
                                              const long long* row_dst, int co0, int tid) {
    static_assert(EPC % G == 0, "epilogue groups");
    const int lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int EROW = BNC * 2 + 16;
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
        uint2 p;
        p.x = pack2bf(v[0], v[1]);
        p.y = pack2bf(v[2], v[3]);
        *(uint2*)(smem + pl * EROW + cl * 2) = p;
      }
    }
    lds_barrier();
    __builtin_amdgcn_sched_barrier(0);
    bnred = a.bn_y != nullptr;
    addpre = a.addend != nullptr;
    const bool stats = a.stats != nullptr;
    const int ch = tid % CPB;                        // NTHR % CPB == 0: the same 8 channels in every row this thread writes
    const int c = co0 + ch * 8;
    if (bnred) {
      const float* ps[4] = {a.bn_scale, a.bn_shift, a.bn_mean, a.bn_invstd};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (c < a.Co) { lo = *(const f32x4*)(ps[k] + c); hi = *(const f32x4*)(ps[k] + c + 4); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { bnp[k][j] = lo[j]; bnp[k][4 + j] = hi[j]; }
      }
    }
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r1[j] = r2[j] = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < EPC; g0 += G) {
      long long offs[G];
      uint4 vals[G], yv4[G], av4[G];
#pragma unroll
      for (int i = 0; i < G; ++i) offs[i] = row_dst[(tid + (g0 + i) * NTHR) / CPB];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int r = (tid + (g0 + i) * NTHR) / CPB;
        vals[i] = *(const uint4*)(smem + r * EROW + ch * 16);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const bool ok = offs[i] >= 0 && c < a.Co;
        yv4[i] = (bnred && ok) ? *(const uint4*)(a.bn_y + offs[i] + c) : make_uint4(0, 0, 0, 0);
        av4[i] = (addpre && ok) ? *(const uint4*)(a.addend + offs[i] + c) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const long long off = offs[i];
        if (off < 0 || c >= a.Co) continue;
        uint4 v = vals[i];
        if (addpre) {
          float f[8], g[8];
          unpack8(v, f);
          unpack8(av4[i], g);
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
          unpack8(yv4[i], yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = fmaf(yv[j], bnp[0][j], bnp[1][j]);
            float dd = z > 0.f ? d[j] : d[j] * a.bn_slope;
            r1[j] += dd;
            r2[j] = fmaf(dd * (yv[j] - bnp[2][j]), bnp[3][j], r2[j]);
          }
        }
      }
    }
    reduce_and_add(a, r1, r2, smem, co0, tid, stats, bnred);
  }

  // one flush per block of a persistent kernel (all threads; smem = the epilogue region, free at this point)
  __device__ __forceinline__ void flush(const GemmArgs& a, unsigned char* smem, int co0, int tid) {
    const bool stats = a.stats != nullptr;
    const bool bnr = a.bn_y != nullptr && !a.dst_f32;
    reduce_and_add(a, pr1, pr2, smem, co0, tid, stats, bnr);
  }

  __device__ __forceinline__ void reduce_and_add(const GemmArgs& a, float (&r1)[8], float (&r2)[8], unsigned char* smem,
                                                 int co0, int tid, bool stats, bool bnred) {
    const int lane = tid & 63, wave = tid >> 6;
    if (stats || bnred) {
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
      lds_barrier();    // NOT __syncthreads(): that would wait for the tile's global stores to be acknowledged
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
};
// ------------------------------------------------------------------------------------------------
// keeps a wave-uniform value in an SGPR and opaque to the compiler: it can then neither be re-loaded from the kernel
// argument segment inside a K-loop (a scalar load there forces s_waitcnt lgkmcnt(0) in front of the MFMAs and with it
// the just-issued fragment reads of the NEXT step) nor folded back into a longer expression
__device__ __forceinline__ int sgpr(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
// 32-byte chunk swizzle of the [pixel][channel] LDS tiles read with ds_read_b64_tr_b16.  A 32-lane group of that read
// touches one chunk of each of the pixel rows {q, q + 8 : q = 0..3} (+4 for the upper half), and LDS has 64 banks = 256 bytes:
// 256-byte rows all start on bank 0 and need eight distinct chunk slots; 128-byte rows alternate between the two bank halves,
// so the FOUR rows of equal parity {0, 2, 8, 10} need four distinct slots; of 64-byte rows only {q, q + 8} share banks.
// (Round 3: the 128- and 64-byte cases used the low bits of the 256-byte formula, which gives rows q and q + 8 the same slot:
// every read of such a tile was a 2-way conflict, 24 % of the weight gradient's LDS cycles - SQ_LDS_BANK_CONFLICT.)
__device__ __forceinline__ int tr_swz(int row, int nchunk32) {
  if (nchunk32 >= 8) return ((row & 3) | (((row >> 3) & 1) << 2)) & (nchunk32 - 1);
  if (nchunk32 == 4) return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
  return nchunk32 == 2 ? (row >> 3) & 1 : 0;
}

__device__ __forceinline__ s16x4 ds_read_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// ds_read_b64_tr_b16 the compiler does not see (no automatic waits: pair with wait_lgkm_dyn + touch)
template <int OFF>
__device__ __forceinline__ void tr_read_asm(s16x4& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// orders the consumers of v after the preceding wait (the asm "modifies" v)
__device__ __forceinline__ void touch(s16x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void wait_lgkm_dyn(int n) {   // n is a compile-time constant after unrolling
  switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt lgkmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); break;
    default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient v2: same MFMA/tr-read structure, operands staged by LDS-DMA into a 2-deep ring

inline unsigned long long make_tapcode(int ntaps, const int32_t* dh, const int32_t* dw, bool* ok) {
  unsigned long long code = 0;
  *ok = true;
  for (int t = 0; t < ntaps; ++t) {
    if (dh[t] < -1 || dh[t] > 1 || dw[t] < -1 || dw[t] > 1) *ok = false;
    code |= (unsigned long long)(((dh[t] + 1) & 3) | (((dw[t] + 1) & 3) << 2)) << (4 * t);
  }
  return code;
}

}  // namespace
