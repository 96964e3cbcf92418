/*
 * libmgd_hip_diag.so - the DIAGNOSTIC build of libmgd_hip.so (gfx950): the same sources compiled with -DMGD_DIAG plus
 * csrc/diag/conv_diag.hip.  It exports everything include/mgd_hip.h declares and, in addition, the entry points below.
 * Only in this library do the convolution kernels carry ablation switches, stamped instantiations and skeleton loops; the
 * product library has none of them and reads no environment variable.  tools/ load it explicitly (multigriddet_amd._lib.load_diag);
 * the product path never does.  No reference counterpart.
 */
#ifndef MGD_HIP_DIAG_H
#define MGD_HIP_DIAG_H

#include "mgd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic flag word the kernels of this library test (the MGD_DBG bits of DESIGN.md: 16 = plain stores instead of the
 * weight gradient's atomics, 32 ... 2048 = parts of conv_gemm8_kernel switched off, 4096 = stamped ping-pong build, ...). */
int mgd_diag_set_flags(int flags);
int mgd_diag_flags_value(void);

/* Reads and clears the 3 x 8 phase-time accumulators of the stamped ping-pong gather-GEMM (flag 4096; tools/stamp_gemm9.py). */
int mgd_debug_stamps(unsigned long long* out24);
/* `blocks` workgroups of 4 waves each issue iters x nacc (8 or 16) independent v_mfma_f32_16x16x32_bf16 on register operands -
 * the matrix pipe's ceiling at the clock the part holds under that load (tools/mfma_peak.py). */
int mgd_debug_mfma_peak(float* out, int blocks, int iters, int nacc, void* stream);
/* The weight gradient's K-step rebuilt around that stream, one ingredient per `mode` bit (fragment reads, barrier, LDS-DMA
 * issue, interleaved reads, s_setprio): what each costs next to 16 MFMAs (tools/mfma_peak.py). */
int mgd_debug_wgrad_skeleton(float* out, int blocks, int iters, int mode, void* stream);
/* The same for the gather-GEMM's K-step (ds_read_b128 fragments, weights as register loads or from LDS), by tile shape. */
int mgd_debug_gemm_skeleton(float* out, int blocks, int iters, int shape, void* stream);

/* Vector-memory issue rate of a CU: `blocks` workgroups of `waves` (1, 2, 4, 8) waves each issue iters x 8 vector-memory
 * instructions of one kind and nothing else.  kind 0 = buffer_load_dwordx4 ... lds over an L2-resident `window` of buf, 1 = the
 * same with every lane out of range, 2 = buffer_load_dwordx4 into registers, 3 = buffer_load_dword ... lds, 4 = buffer_load_dword
 * into a register, 5 = global_load_dwordx4, 6 = kind 2 with every lane out of range (tools/vmem_rate.py). */
int mgd_debug_vmem_rate(const void* buf, unsigned window, float* out, int blocks, int waves, int iters, int kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif
