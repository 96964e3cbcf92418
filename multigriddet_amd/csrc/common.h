// Shared helpers for libmgd_hip.so (gfx950 only; no other back end is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mgd_hip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

int mgd_set_error(int code, const char* fmt, ...);

#define MGD_REQUIRE(cond, ...)                        \
  do {                                                \
    if (!(cond)) return mgd_set_error(MGD_EINVAL, __VA_ARGS__); \
  } while (0)

extern thread_local const char* mgd_last_launch_name;     // error.cpp: the kernel family of this thread's last launch
#define MGD_CHECK_LAUNCH(name)                                                              \
  do {                                                                                      \
    mgd_last_launch_name = name;                                                            \
    hipError_t e__ = hipGetLastError();                                                     \
    if (e__ != hipSuccess) return mgd_set_error(MGD_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// Wave-wide reductions on the DPP data path (the result is wave-uniform: lane 63 read back through an SGPR).  Four steps inside
// the rows of 16 lanes (quad permutes, half-row and row mirrors), then row_bcast:15 / row_bcast:31 carry the row totals into
// rows 1, 3 and 2, 3.  Seven VALU instructions; the __shfl_xor butterfly these replace is six ds_bpermute round trips through
// the LDS crossbar per reduction.  Call with every lane of the wave active (as the butterfly needed too).
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               ROWMASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(0.f, v);           // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(0.f, v);           // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(0.f, v);          // row_half_mirror
  v += dpp_f32<0x140>(0.f, v);          // row_mirror
  v += dpp_f32<0x142, 0xA>(0.f, v);     // row_bcast:15 into rows 1, 3
  v += dpp_f32<0x143, 0xC>(0.f, v);     // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v, v));
  v = fmaxf(v, dpp_f32<0x4E>(v, v));
  v = fmaxf(v, dpp_f32<0x141>(v, v));
  v = fmaxf(v, dpp_f32<0x140>(v, v));
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of logical
// tile ids so that neighbouring tiles hit the same L2.  Bijective for any nblk.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
