// Evaluation hot spot on gfx950: IoU matrix and greedy prediction -> ground-truth matching of the mAP computation
// (reference multigriddet/evaluation/metrics.py:28-71 calculate_iou_matrix, :73-219 match_predictions_to_gt[_cached]).
// float64 throughout, like the reference (boxes arrive as Python floats), so every TP/FP decision is identical.
#include "common.h"

#pragma clang fp contract(off)   // numpy evaluates a*b + c in two roundings; a TP/FP decision must not depend on an FMA

namespace {

// mode 0: boxes are xyxy (metrics.py:56-69, the cached-IoU path).
// mode 1: the un-cached path calls BoxUtils.box_iou (utils/boxes.py:16-57), which reads the SAME xyxy numbers as
//         (cx, cy, w, h) - reproduced because the reference's per-scale metrics (APS/APM/APL) always take that path.
__device__ __forceinline__ double eval_iou(const double* a, const double* b, int mode) {
  if (mode == 0) {
    double x1 = fmax(a[0], b[0]), y1 = fmax(a[1], b[1]);
    double x2 = fmin(a[2], b[2]), y2 = fmin(a[3], b[3]);
    double inter = fmax(0.0, x2 - x1) * fmax(0.0, y2 - y1);
    double area1 = (a[2] - a[0]) * (a[3] - a[1]), area2 = (b[2] - b[0]) * (b[3] - b[1]);
    double uni = area1 + area2 - inter;
    return uni > 0.0 ? inter / uni : 0.0;
  }
  double ax0 = a[0] - a[2] / 2, ay0 = a[1] - a[3] / 2, ax1 = a[0] + a[2] / 2, ay1 = a[1] + a[3] / 2;
  double bx0 = b[0] - b[2] / 2, by0 = b[1] - b[3] / 2, bx1 = b[0] + b[2] / 2, by1 = b[1] + b[3] / 2;
  double ixmin = fmax(ax0, bx0), iymin = fmax(ay0, by0), ixmax = fmin(ax1, bx1), iymax = fmin(ay1, by1);
  if (ixmax <= ixmin || iymax <= iymin) return 0.0;
  double inter = (ixmax - ixmin) * (iymax - iymin);
  double uni = a[2] * a[3] + b[2] * b[3] - inter;
  return uni > 0.0 ? inter / uni : 0.0;
}

__global__ void iou_matrix_kernel(const double* __restrict__ b1, const double* __restrict__ b2, double* __restrict__ out,
                                  int n, int m, int mode) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * m) return;
  int r = (int)(i / m), c = (int)(i - (long long)r * m);
  out[i] = eval_iou(b1 + (long long)r * 4, b2 + (long long)c * 4, mode);
}

// One block per (image, class) group, one wavefront per IoU threshold.  The group's predictions are contiguous and
// in descending score order, its ground truths contiguous; wave t walks the predictions, lanes hold the IoU with
// the still-unmatched ground truths (64 at a time), the best one (first maximum, strictly positive) is matched if
// it reaches threshold t.  tp[t][p] = 1 for true positives, 0 for false positives.
__global__ __launch_bounds__(1024) void eval_match_kernel(const double* __restrict__ pb, const int* __restrict__ pstart,
                                                          const double* __restrict__ gb, const int* __restrict__ gstart,
                                                          const double* __restrict__ thr, int NT, int mode,
                                                          unsigned char* __restrict__ tp, long long P) {
  extern __shared__ unsigned char matched[];     // [NT][ng]
  const int g = blockIdx.x, lane = threadIdx.x & 63, t = threadIdx.x >> 6;
  const int p0 = pstart[g], p1 = pstart[g + 1], g0 = gstart[g], ng = gstart[g + 1] - g0;
  if (t >= NT) return;
  unsigned char* mt = matched + (size_t)t * ng;
  for (int j = lane; j < ng; j += 64) mt[j] = 0;
  __builtin_amdgcn_wave_barrier();
  const double th = thr[t];
  for (int p = p0; p < p1; ++p) {
    const double* a = pb + (long long)p * 4;
    double best = 0.0;
    int bi = -1;
    for (int c = 0; c < ng; c += 64) {
      int j = c + lane;
      double v = -1.0;
      if (j < ng && !mt[j]) v = eval_iou(a, gb + (long long)(g0 + j) * 4, mode);
      // wave arg-max, first index on ties
      int idx = j;
      for (int o = 32; o > 0; o >>= 1) {
        long long bits = __double_as_longlong(v);
        int lo = __shfl_xor((int)(bits & 0xffffffffll), o, 64), hi = __shfl_xor((int)(bits >> 32), o, 64);
        double ov = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if (v > best) { best = v; bi = idx; }
    }
    bool is_tp = bi >= 0 && best >= th;
    if (lane == 0) {
      tp[(long long)t * P + p] = is_tp ? 1 : 0;
      if (is_tp) mt[bi] = 1;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
}

}  // namespace

extern "C" int mgd_iou_matrix(const double* boxes1, const double* boxes2, double* out, int n, int m, int mode,
                              void* stream) {
  MGD_REQUIRE(n >= 0 && m >= 0 && (mode == 0 || mode == 1), "iou_matrix: n=%d m=%d mode=%d", n, m, mode);
  if (n == 0 || m == 0) return MGD_OK;
  MGD_REQUIRE(boxes1 && boxes2 && out, "iou_matrix: null pointer");
  long long tot = (long long)n * m;
  hipLaunchKernelGGL(iou_matrix_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes1,
                     boxes2, out, n, m, mode);
  MGD_CHECK_LAUNCH("iou_matrix");
  return MGD_OK;
}

extern "C" int mgd_eval_match(const double* pred_boxes, const int32_t* group_pred_start, const double* gt_boxes,
                              const int32_t* group_gt_start, int num_groups, int max_group_gts,
                              const double* thresholds, int num_thresholds, int mode, uint8_t* tp,
                              long long num_preds, void* stream) {
  MGD_REQUIRE(num_groups >= 0 && num_thresholds >= 1 && num_thresholds <= 16, "eval_match: groups=%d thresholds=%d (1..16)",
              num_groups, num_thresholds);
  MGD_REQUIRE(mode == 0 || mode == 1, "eval_match: mode=%d", mode);
  if (num_groups == 0 || num_preds == 0) return MGD_OK;
  MGD_REQUIRE(pred_boxes && group_pred_start && gt_boxes && group_gt_start && thresholds && tp, "eval_match: null pointer");
  size_t lds = (size_t)num_thresholds * (size_t)(max_group_gts > 0 ? max_group_gts : 1);
  MGD_REQUIRE(lds <= 60 * 1024, "eval_match: %d ground truths in one (image, class) group exceed the LDS budget",
              max_group_gts);
  hipLaunchKernelGGL(eval_match_kernel, dim3(num_groups), dim3(64 * num_thresholds), lds, (hipStream_t)stream, pred_boxes,
                     group_pred_start, gt_boxes, group_gt_start, thresholds, num_thresholds, mode, tp, num_preds);
  MGD_CHECK_LAUNCH("eval_match");
  return MGD_OK;
}
