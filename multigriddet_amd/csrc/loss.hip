// MultiGridLoss forward + backward on gfx950, one wavefront per grid cell.
// Replaces MultiGridLoss.compute_loss and helpers (reference multigriddet/losses/multigrid_loss.py:233-443,
// 445-492 IoU, 494-703 ignore mask, 729-757 MSE, 759-799 anchor BCE, 829-859 class BCE, 861-928
// objectness, 930-1043 variance consensus; losses/focal_loss.py:40-77 sigmoid focal, :80-114 softmax focal;
// losses/iou_losses.py:36-237 GIoU / DIoU / CIoU localisation, wired at multigrid_loss.py:353-364).
//
// GIoU/DIoU/CIoU and the softmax focal loss exist in two forms (cfg.iou_compat / cfg.softmax_compat):
//  * 0 = "tf_ref": what the reference's TensorFlow code computes.  It multiplies a per-cell loss of shape [B,H,W] by the
//    object mask of shape [B,H,W,1]; broadcasting aligns trailing dimensions, so the product is the 4-D tensor
//    P[i,j,k,l] = loss[j,k,l] * mask[i,j,k] - defined only for H == W and (B == 1 or B == H) - and its sum is
//    sum_{b,h,w} loss[b,h,w] * Wt[b,h] with Wt[b,h] = sum_i mask[i,b,h] (B == H) or sum_j mask[0,j,h] (B == 1).
//    DIoU/CIoU subtract a keepdims [B,H,W,1] centre-distance term from the [B,H,W] IoU the same way, which adds
//    W * sum_{positive cells} d2 / (c2 + eps).  The boxes are the raw tensors "as given" (grid offsets / log ratios).
//    The softmax focal loss additionally multiplies by class_weights [1,1,1,C] along the W axis (C == 1 or C == W).
//  * 1 = "fixed": the per-cell loss times the cell's own mask, boxes decoded to grid-cell units
//    (xy = tanh(.15p)+sigmoid(.15p) against the target offset, wh = exp(t) * assigned anchor / stride).
//
// HBM-bound: algorithmic bytes = read y_pred + y_true (2 x B*cells*F*4) + write the gradient.
// A cell's F = 5+A+C channels sit on the 64 lanes (coalesced 256-byte loads), the per-image list of
// ground-truth boxes (= every positive cell, as in the reference) is staged once per block in LDS,
// and the ignore-mask IoU search runs lanes-over-GT with a wave max.  All reference quirks are kept:
// the transposed 'ij' grid (:545-551), pixel anchors multiplied by the stride again (:572),
// anchor_scale and object_scale applied twice (:349/:433, :908/:432).
#include "common.h"

namespace {

constexpr int MAXL = 4, MAXA = 8;
constexpr int CELLS_PER_BLOCK = 64;     // upper bound; the launch picks fewer on the small grids (see mgd_loss_fwd_bwd)
constexpr int GT_LDS = 1024;
constexpr float KEPS = 1e-7f;

struct LossArgs {
  mgd_loss_cfg cfg;
  const float* yp[MAXL];
  const float* yt[MAXL];
  float* gf[MAXL];       // f32 gradient (may be null)
  bf16_t* gb[MAXL];      // bf16 gradient (may be null)
  const float* class_w;  // may be null
  int* gt_count;         // [L][B]
  int* npos;             // [L]
  int* ncenter;          // [L]
  float4* gt;            // [sum_l B*g*g]
  float* assigned;       // [sum_l B*g*g]
  double* acc;           // [8]
  float* wt;             // tf_ref broadcast weights Wt[l][b][h] (see the header comment); [sum_l B*gh]
  long long wt_off[MAXL + 1];
  long long cell_off[MAXL + 1];
  float* components;
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float bce_logits(float z, float x) {
  return fmaxf(x, 0.f) - x * z + log1pf(__expf(-fabsf(x)));
}
__device__ __forceinline__ float xy_act(float p) { return tanhf(0.15f * p) + sigmoidf(0.15f * p); }
__device__ __forceinline__ float xy_act_grad(float p) {
  float t = tanhf(0.15f * p), s = sigmoidf(0.15f * p);
  return 0.15f * (1.f - t * t) + 0.15f * s * (1.f - s);
}

__device__ __forceinline__ float norm_factor(const mgd_loss_cfg& c, int l, int npos) {
  float f = 1.f;
  float B = (float)c.B;
  for (int i = 0; i < c.norm_positives; ++i) f *= fmaxf((float)npos, 1.f);
  for (int i = 0; i < c.norm_batch; ++i) f *= B;
  for (int i = 0; i < c.norm_grid; ++i) f *= B * (float)c.grid_h[l] * (float)c.grid_w[l];
  return fmaxf(f, 1.f);
}

// ---- K1: collect the per-image GT list (every positive cell) and the counters
// One launch for all scales: blockIdx.y = scale (blocks past a small scale's cells leave at once).
__global__ void loss_prep_kernel(LossArgs a) {
  const mgd_loss_cfg& c = a.cfg;
  const int l = blockIdx.y;
  const int gh = c.grid_h[l], gw = c.grid_w[l], F = 5 + c.A + c.C;
  long long ncell = (long long)c.B * gh * gw;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncell) return;
  const float* t = a.yt[l] + i * F;
  a.assigned[a.cell_off[l] + i] = 0.f;
  if (!(t[4] > 0.5f)) return;
  int b = (int)(i / (gh * gw));
  int rem = (int)(i - (long long)b * gh * gw);
  int row = rem / gw, col = rem - row * gw;
  float sx = (float)c.in_w / (float)gw, sy = (float)c.in_h / (float)gh;
  // transposed grid: x gets the row index, y the column index (reference :545-551)
  float gx = (t[0] + (float)row) * sx, gy = (t[1] + (float)col) * sy;
  int k = 0;
  float best = t[5];
  for (int j = 1; j < c.A; ++j)
    if (t[5 + j] > best) { best = t[5 + j]; k = j; }
  float w = __expf(t[2]) * c.anchors[l][k][0] * sx, h = __expf(t[3]) * c.anchors[l][k][1] * sy;
  int slot = atomicAdd(a.gt_count + l * c.B + b, 1);
  a.gt[a.cell_off[l] + (long long)b * gh * gw + slot] = make_float4(gx, gy, w, h);
  atomicAdd(a.npos + l, 1);
  if (a.wt) atomicAdd(a.wt + a.wt_off[l] + (c.B == 1 ? 0 : row) * gh + col, 1.0f);   // Wt[b'=row][h'=col] (B==H) / Wt[0][col] (B==1)
  if (t[0] >= 0.f && t[0] < 1.f && t[1] >= 0.f && t[1] < 1.f) atomicAdd(a.ncenter + l, 1);
}


// ---- forward-mode dual numbers for the IoU losses: lane j < 4 carries d/d(pred channel j)
struct Dual { float v, d; };
__device__ __forceinline__ Dual dc(float v) { return Dual{v, 0.f}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  float q = a.v / b.v;
  return Dual{q, (a.d - q * b.d) / b.v};
}
// TF gradient conventions: maximum passes the gradient to x where x >= y, minimum where x <= y
__device__ __forceinline__ Dual dmax(Dual a, Dual b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ Dual dmin(Dual a, Dual b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ Dual datan2(Dual y, Dual x) {
  float den = x.v * x.v + y.v * y.v;
  return Dual{atan2f(y.v, x.v), den > 0.f ? (x.v * y.d - y.v * x.d) / den : 0.f};
}

// One cell of GIoU / DIoU / CIoU (reference iou_losses.py:58-95, 121-160, 186-237) on centre-format boxes.
// Returns the two parts of 1 - metric:  `cellpart` = the [B,H,W]-shaped part (1 - iou (+ enclosure / aspect terms)),
// `distpart` = d2 / (c2 + eps), which the reference computes with keepdims (DIoU / CIoU only, else 0).
__device__ __forceinline__ void iou_loss_cell(int kind, Dual tx, Dual ty, Dual tw, Dual th, Dual px, Dual py, Dual pw,
                                              Dual ph, Dual& cellpart, Dual& distpart) {
  const Dual half = dc(0.5f), zero = dc(0.f), eps = dc(KEPS), one = dc(1.f);
  Dual tminx = tx - tw * half, tmaxx = tx + tw * half, tminy = ty - th * half, tmaxy = ty + th * half;
  Dual pminx = px - pw * half, pmaxx = px + pw * half, pminy = py - ph * half, pmaxy = py + ph * half;
  Dual iw = dmax(dmin(tmaxx, pmaxx) - dmax(tminx, pminx), zero);
  Dual ih = dmax(dmin(tmaxy, pmaxy) - dmax(tminy, pminy), zero);
  Dual inter = iw * ih;
  Dual uni = tw * th + pw * ph - inter;
  Dual iou = inter / (uni + eps);
  Dual ew = dmax(dmax(tmaxx, pmaxx) - dmin(tminx, pminx), zero);
  Dual eh = dmax(dmax(tmaxy, pmaxy) - dmin(tminy, pminy), zero);
  distpart = zero;
  if (kind == 1) {                       // GIoU
    Dual earea = ew * eh;
    cellpart = one - (iou - (earea - uni) / (earea + eps));
    return;
  }
  Dual dx = tx - px, dy = ty - py;
  distpart = (dx * dx + dy * dy) / (ew * ew + eh * eh + eps);
  cellpart = one - iou;
  if (kind == 3) {                       // CIoU: + alpha * v, alpha = v / (1 - iou + v + eps), differentiated through
    Dual da = datan2(tw, th) - datan2(pw, ph);
    Dual v = dc(4.0f / (3.14159265358979323846f * 3.14159265358979323846f)) * da * da;
    Dual alpha = v / (one - iou + v + eps);
    cellpart = cellpart + alpha * v;
  }
}

// ---- K2: per-cell loss and gradient.  grid = (cells chunks, B, 1) per scale.
struct CellsPerBlock { int v[MAXL]; };

// One launch for all scales: blockIdx.z = scale, blockIdx.y = image, blockIdx.x = chunk of cpb.v[scale] cells; the
// three scales are independent and the small ones are latency bound, so they run under the large one.
__global__ __launch_bounds__(256) void loss_cell_kernel(LossArgs a, CellsPerBlock cpbs) {
  const mgd_loss_cfg& c = a.cfg;
  const int l = blockIdx.z, cpb = cpbs.v[l];
  const int gh = c.grid_h[l], gw = c.grid_w[l], A = c.A, C = c.C, F = 5 + A + C;
  if ((int)blockIdx.x * cpb >= gh * gw) return;      // block-uniform: this scale has fewer chunks than the widest one
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float4 gts[GT_LDS];
  __shared__ float wsum[4][4];

  const int ngt = a.gt_count[l * c.B + b];
  const float4* gtg = a.gt + a.cell_off[l] + (long long)b * gh * gw;
  const bool in_lds = ngt <= GT_LDS;
  if (in_lds)
    for (int i = threadIdx.x; i < ngt; i += 256) gts[i] = gtg[i];
  __syncthreads();

  const float nf = norm_factor(c, l, a.npos[l]);
  const float inv_nf = 1.0f / nf;
  const float sx = (float)c.in_w / (float)gw, sy = (float)c.in_h / (float)gh;
  const float gscale = c.grad_out_scale;

  float s_loc = 0.f, s_obj = 0.f, s_anc = 0.f, s_cls = 0.f;
  const int cell_beg = blockIdx.x * cpb;
  const int cell_end = min(gh * gw, cell_beg + cpb);
  for (int cell = cell_beg + wave; cell < cell_end; cell += 4) {
    const long long gcell = (long long)b * gh * gw + cell;
    const float* p = a.yp[l] + gcell * F;
    const float* t = a.yt[l] + gcell * F;
    const int row = cell / gw, col = cell - row * gw;
    // channel ch0 = lane, ch1 = 64 + lane
    const int ch0 = lane, ch1 = 64 + lane;
    float p0 = ch0 < F ? p[ch0] : 0.f, t0 = ch0 < F ? t[ch0] : 0.f;
    float p1 = ch1 < F ? p[ch1] : 0.f, t1 = ch1 < F ? t[ch1] : 0.f;
    // broadcast the header channels
    float pxr = __shfl(p0, 0, 64), pyr = __shfl(p0, 1, 64), pwr = __shfl(p0, 2, 64), phr = __shfl(p0, 3, 64);
    float tob = __shfl(t0, 4, 64);
    const float obj = tob > 0.5f ? 1.f : 0.f;
    // predicted boxes (absolute), transposed grid as in the reference
    float ax = xy_act(pxr), ay = xy_act(pyr);
    float bx = (ax + (float)row) * sx, by = (ay + (float)col) * sy;
    float ew = __expf(pwr), eh = __expf(phr);
    float iou_a[MAXA];
#pragma unroll
    for (int j = 0; j < MAXA; ++j) iou_a[j] = 0.f;
    for (int g0 = 0; g0 < ngt; g0 += 64) {
      int g = g0 + lane;
      if (g < ngt) {
        float4 G = in_lds ? gts[g] : gtg[g];
        float gx0 = G.x - G.z / 2.0f, gx1 = G.x + G.z / 2.0f, gy0 = G.y - G.w / 2.0f, gy1 = G.y + G.w / 2.0f;
        float garea = G.z * G.w;
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
          if (j < A) {
            float w = ew * c.anchors[l][j][0] * sx, h = eh * c.anchors[l][j][1] * sy;
            float iw = fmaxf(fminf(bx + w / 2.0f, gx1) - fmaxf(bx - w / 2.0f, gx0), 0.f);
            float ih = fmaxf(fminf(by + h / 2.0f, gy1) - fmaxf(by - h / 2.0f, gy0), 0.f);
            float inter = iw * ih;
            float iou = inter / (w * h + garea - inter + KEPS);
            iou_a[j] = fmaxf(iou_a[j], iou);
          }
        }
      }
    }
    float max_iou = 0.f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
      if (j < A) {
        iou_a[j] = wave_max(iou_a[j]);
        max_iou = j == 0 ? iou_a[0] : fmaxf(max_iou, iou_a[j]);
      }
    const float ignore = (max_iou > c.ignore_thresh && tob < 0.5f) ? 1.f : 0.f;
    // assigned-anchor IoU (argmax of the true anchor one-hot, first max)
    int kstar = 0;
    {
      float best = __shfl(t0, 5, 64);
      for (int j = 1; j < A; ++j) {
        float v = __shfl(t0, 5 + j, 64);
        if (v > best) { best = v; kstar = j; }
      }
    }
    float assigned = 0.f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
      if (j == kstar) assigned = iou_a[j];
    assigned *= obj;
    if (lane == 0) a.assigned[a.cell_off[l] + gcell] = assigned;

    // objectness target / weight
    float tgt = tob;
    if (c.use_iou_aware_objectness) {
      float piou = fminf(fmaxf(assigned, 0.f), 1.f);
      float blended = c.iou_objectness_ratio * powf(piou + KEPS, c.iou_objectness_power) +
                      (1.f - c.iou_objectness_ratio) * tob;
      tgt = obj * blended + (1.f - obj) * tgt;
    }
    float wobj = obj * c.object_scale + (1.f - obj) * (1.f - ignore) * c.no_object_scale;
    if (c.trainable_nms_weight > 0.f)
      wobj += (1.f - obj) * ignore * c.trainable_nms_weight *
              powf(fminf(fmaxf(max_iou, 0.f), 1.f) + KEPS, c.trainable_nms_power);

    // IoU localisation (loss_option 3 + one of the flags): lanes 0-3 differentiate with respect to their own channel
    const bool iou_mode = c.iou_loss != 0;
    float iou_val = 0.f, iou_grad = 0.f;
    const float t_x = __shfl(t0, 0, 64), t_y = __shfl(t0, 1, 64), t_w = __shfl(t0, 2, 64), t_h = __shfl(t0, 3, 64);
    if (iou_mode && lane < 4) {
      Dual px{pxr, lane == 0 ? 1.f : 0.f}, py{pyr, lane == 1 ? 1.f : 0.f};
      Dual pw{pwr, lane == 2 ? 1.f : 0.f}, ph{phr, lane == 3 ? 1.f : 0.f};
      Dual tx = dc(t_x), ty = dc(t_y), tw = dc(t_w), th = dc(t_h);
      float wcell = obj, wdist = obj;
      if (c.iou_compat == 0) {           // tf_ref: raw tensors, broadcast weights
        wcell = a.wt[a.wt_off[l] + (c.B == 1 ? 0 : b) * gh + row];
        wdist = obj * (float)gw;
      } else {                           // fixed: decoded boxes in grid-cell units, assigned anchor
        float aw = 0.f, ah = 0.f;
#pragma unroll
        for (int j = 0; j < MAXA; ++j)
          if (j == kstar) { aw = c.anchors[l][j][0] / sx; ah = c.anchors[l][j][1] / sy; }
        px = Dual{xy_act(pxr), px.d * xy_act_grad(pxr)};
        py = Dual{xy_act(pyr), py.d * xy_act_grad(pyr)};
        pw = Dual{ew * aw, pw.d * ew * aw};
        ph = Dual{eh * ah, ph.d * eh * ah};
        tw = dc(__expf(t_w) * aw);
        th = dc(__expf(t_h) * ah);
      }
      if (wcell != 0.f || wdist != 0.f) {
        Dual cp, dp;
        iou_loss_cell(c.iou_loss, tx, ty, tw, th, px, py, pw, ph, cp, dp);
        iou_val = wcell * cp.v + wdist * dp.v;
        iou_grad = wcell * cp.d + wdist * dp.d;
      }
    }
    // softmax focal classification (use_softmax_loss): one value per cell, reduced over the class channels
    const bool smx = c.use_softmax_focal != 0;
    float smx_w = 0.f, smx_lse = 0.f, smx_pt = 0.f, smx_ce = 0.f, smx_sy = 0.f;
    if (smx) {
      const bool c0 = ch0 >= 5 + A && ch0 < F, c1 = ch1 >= 5 + A && ch1 < F;
      float m = wave_max(fmaxf(c0 ? p0 : -3.0e38f, c1 ? p1 : -3.0e38f));
      float e0 = c0 ? __expf(p0 - m) : 0.f, e1 = c1 ? __expf(p1 - m) : 0.f;
      float se = wave_sum(e0 + e1);
      smx_lse = m + __logf(se);
      smx_sy = wave_sum((c0 ? t0 : 0.f) + (c1 ? t1 : 0.f));
      float syx = wave_sum((c0 ? t0 * p0 : 0.f) + (c1 ? t1 * p1 : 0.f));
      smx_pt = wave_sum((c0 ? t0 * e0 : 0.f) + (c1 ? t1 * e1 : 0.f)) / se;
      smx_ce = -syx + smx_sy * smx_lse;
      if (c.softmax_compat == 0) {       // tf_ref: Wt[b,h] * class_weights[w] (C == W) or class_weights[0] (C == 1)
        float cw = a.class_w ? a.class_w[C == 1 ? 0 : col] : 1.f;
        smx_w = a.wt[a.wt_off[l] + (c.B == 1 ? 0 : b) * gh + row] * cw;
      } else {                           // fixed: the cell's own mask, class weight of the true class
        float cwt = wave_sum((c0 ? t0 * (a.class_w ? a.class_w[ch0 - 5 - A] : 1.f) : 0.f) +
                             (c1 ? t1 * (a.class_w ? a.class_w[ch1 - 5 - A] : 1.f) : 0.f));
        smx_w = obj * cwt;
      }
      if (lane == 0 && smx_w != 0.f) s_cls += smx_w * powf(fmaxf(1.f - smx_pt, 0.f), c.focal_gamma) * smx_ce;
    }

    // per-channel loss / gradient
    float g0v = 0.f, g1v = 0.f;
    auto channel = [&](int ch, float pv, float tv, float& gout) {
      gout = 0.f;
      if (ch >= F) return;
      if (ch < 4 && iou_mode) {
        if (ch == 0) s_loc += iou_val;
        gout = c.coord_scale * iou_grad * inv_nf;
      } else if (ch >= 5 + A && smx) {
        if (smx_w != 0.f) {
          float om = fmaxf(1.f - smx_pt, 0.f);
          float mod = powf(om, c.focal_gamma);
          float sc = __expf(pv - smx_lse);                       // softmax_c
          float dce = -tv + smx_sy * sc;
          float dpt = sc * (tv - smx_pt);
          float dmod = om > 0.f ? -c.focal_gamma * powf(om, c.focal_gamma - 1.f) * dpt : 0.f;
          gout = c.class_scale * smx_w * (dmod * smx_ce + mod * dce) * inv_nf;
        }
      } else if (ch < 2) {
        float act = xy_act(pv);
        float d = act - tv;
        s_loc += obj * d * d;
        gout = c.coord_scale * obj * 2.f * d * xy_act_grad(pv) * inv_nf;
      } else if (ch < 4) {
        float d = pv - tv;
        s_loc += obj * d * d;
        gout = c.coord_scale * obj * 2.f * d * inv_nf;
      } else if (ch == 4) {
        s_obj += bce_logits(tgt, pv) * wobj;
        gout = c.object_scale * wobj * (sigmoidf(pv) - tgt) * inv_nf;
      } else if (ch < 5 + A) {
        float m = obj * (1.f - ignore);
        s_anc += bce_logits(tv, pv) * m;
        gout = c.anchor_scale * c.anchor_scale * m * (sigmoidf(pv) - tv) * inv_nf;
      } else {
        float cw = a.class_w ? a.class_w[ch - 5 - A] : 1.f;
        if (c.use_focal_loss) {
          float pr = sigmoidf(pv);
          float pt = tv * pr + (1.f - tv) * (1.f - pr);
          float om = fmaxf(1.f - pt, 0.f);
          float mod = powf(om, c.focal_gamma);
          float at = tv * c.focal_alpha + (1.f - tv) * (1.f - c.focal_alpha);
          float bce = bce_logits(tv, pv);
          s_cls += mod * at * bce * cw * obj;
          float dmod = om > 0.f ? -c.focal_gamma * powf(om, c.focal_gamma - 1.f) * (2.f * tv - 1.f) * pr * (1.f - pr) : 0.f;
          gout = c.class_scale * cw * obj * at * (dmod * bce + mod * (pr - tv)) * inv_nf;
        } else {
          float ts = c.label_smoothing > 0.f ? tv * (1.f - c.label_smoothing) + c.label_smoothing / (float)C : tv;
          s_cls += bce_logits(ts, pv) * cw * obj;
          gout = c.class_scale * cw * obj * (sigmoidf(pv) - ts) * inv_nf;
        }
      }
    };
    channel(ch0, p0, t0, g0v);
    channel(ch1, p1, t1, g1v);
    g0v *= gscale;
    g1v *= gscale;
    if (a.gf[l]) {
      float* g = a.gf[l] + gcell * F;
      if (ch0 < F) g[ch0] = g0v;
      if (ch1 < F) g[ch1] = g1v;
    }
    if (a.gb[l]) {
      bf16_t* g = a.gb[l] + gcell * F;
      if (ch0 < F) g[ch0] = f2bf(g0v);
      if (ch1 < F) g[ch1] = f2bf(g1v);
    }
  }
  s_loc = wave_sum(s_loc); s_obj = wave_sum(s_obj); s_anc = wave_sum(s_anc); s_cls = wave_sum(s_cls);
  if (lane == 0) { wsum[wave][0] = s_loc; wsum[wave][1] = s_obj; wsum[wave][2] = s_anc; wsum[wave][3] = s_cls; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x];
    v *= inv_nf;
    if (threadIdx.x == 2) v *= c.anchor_scale;   // reference accumulates anchor_scale * anchor_loss (:349,:390)
    atomicAdd(a.acc + threadIdx.x, (double)v);
  }
}

// ---- K2' (round 4): the same per-cell loss for the configurations WITHOUT an IoU localisation loss or softmax focal
// classification (every default), laid out for the vector ALU.  loss_cell_kernel spends one wave per cell - 1 090 wave
// instructions for 88 channels, 8 of them the cell's own scalars computed on all 64 lanes, and the kernel is bound by the
// vector ALU (PMC: 1.3e8 VALU instructions per launch = the whole 228 us).  Here a block of 256 threads owns 64 cells:
//   A1  thread (q, cell) = (tid / 64, tid % 64): IoU of the cell's predicted box under every anchor against the ground-truth
//       boxes g = q, q + 4, ... of the image (list in LDS, broadcast reads) -> partial maxima in LDS;
//   A2  thread = cell (64 threads): ignore mask, objectness target / weight, the losses and gradients of the 5 + A header
//       channels; the cell's `obj` goes to LDS;
//   B   thread = (cell, class) elements, 256 at a time: the class channels, element-wise.
// Same arithmetic per channel as loss_cell_kernel (the sums are added in another order).
constexpr int C2_CELLS = 64;
__global__ __launch_bounds__(256) void loss_cell2_kernel(LossArgs a) {
  const mgd_loss_cfg& c = a.cfg;
  const int l = blockIdx.z;
  const int gh = c.grid_h[l], gw = c.grid_w[l], A = c.A, C = c.C, F = 5 + A + C;
  const int cell_beg = blockIdx.x * C2_CELLS;
  if (cell_beg >= gh * gw) return;                    // block-uniform: this scale has fewer chunks than the widest one
  const int ncell = min(C2_CELLS, gh * gw - cell_beg);
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float4 gts[GT_LDS];
  __shared__ float part[4][MAXA][C2_CELLS];
  __shared__ float cell_obj[C2_CELLS];
  __shared__ float wsum[4][4];

  const int ngt = a.gt_count[l * c.B + b];
  const float4* gtg = a.gt + a.cell_off[l] + (long long)b * gh * gw;
  const bool in_lds = ngt <= GT_LDS;
  if (in_lds)
    for (int i = tid; i < ngt; i += 256) gts[i] = gtg[i];

  const float nf = norm_factor(c, l, a.npos[l]);
  const float inv_nf = 1.0f / nf;
  const float sx = (float)c.in_w / (float)gw, sy = (float)c.in_h / (float)gh;
  const float gscale = c.grad_out_scale;
  const long long cell0 = (long long)b * gh * gw + cell_beg;     // first cell of the block in the scale's tensors
  const float* P = a.yp[l] + cell0 * F;
  const float* T = a.yt[l] + cell0 * F;

  // ---- A1: partial IoU maxima
  const int ci = lane;                                 // cell of this thread in phases A1 / A2
  const bool cell_ok = ci < ncell;
  float pxr = 0.f, pyr = 0.f, pwr = 0.f, phr = 0.f;
  if (cell_ok) {
    const float* p = P + (long long)ci * F;
    pxr = p[0]; pyr = p[1]; pwr = p[2]; phr = p[3];
  }
  const int row = (cell_beg + ci) / gw, col = (cell_beg + ci) - row * gw;
  const float ax = xy_act(pxr), ay = xy_act(pyr);
  const float bx = (ax + (float)row) * sx, by = (ay + (float)col) * sy;   // transposed grid as in the reference
  const float ew = __expf(pwr), eh = __expf(phr);
  float iou_a[MAXA];
#pragma unroll
  for (int j = 0; j < MAXA; ++j) iou_a[j] = 0.f;
  __syncthreads();                                     // gts staged
  for (int g = wave; g < ngt; g += 4) {
    const float4 G = in_lds ? gts[g] : gtg[g];
    const float gx0 = G.x - G.z / 2.0f, gx1 = G.x + G.z / 2.0f, gy0 = G.y - G.w / 2.0f, gy1 = G.y + G.w / 2.0f;
    const float garea = G.z * G.w;
#pragma unroll
    for (int j = 0; j < MAXA; ++j) {
      if (j < A) {
        const float w = ew * c.anchors[l][j][0] * sx, h = eh * c.anchors[l][j][1] * sy;
        const float iw = fmaxf(fminf(bx + w / 2.0f, gx1) - fmaxf(bx - w / 2.0f, gx0), 0.f);
        const float ih = fmaxf(fminf(by + h / 2.0f, gy1) - fmaxf(by - h / 2.0f, gy0), 0.f);
        const float inter = iw * ih;
        const float iou = inter / (w * h + garea - inter + KEPS);
        iou_a[j] = fmaxf(iou_a[j], iou);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXA; ++j)
    if (j < A) part[wave][j][ci] = iou_a[j];
  __syncthreads();

  // ---- A2: the cell's scalars and its 5 + A header channels (wave 0)
  float s_loc = 0.f, s_obj = 0.f, s_anc = 0.f, s_cls = 0.f;
  if (wave == 0) {
    float obj = 0.f;
    if (cell_ok) {
      const float* p = P + (long long)ci * F;
      const float* t = T + (long long)ci * F;
      float max_iou = 0.f;
#pragma unroll
      for (int j = 0; j < MAXA; ++j)
        if (j < A) {
          iou_a[j] = fmaxf(fmaxf(part[0][j][ci], part[1][j][ci]), fmaxf(part[2][j][ci], part[3][j][ci]));
          max_iou = j == 0 ? iou_a[0] : fmaxf(max_iou, iou_a[j]);
        }
      // the cell's 5 + A header values of both tensors: two 16-byte loads each when the row pitch keeps them aligned (they are
      // 352 bytes apart from lane to lane: every dword load of a wave touches 64 cache lines)
      float th_[8], ph_[8];
      if ((F & 3) == 0 && A <= 3 && ((((unsigned long long)t) | ((unsigned long long)p)) & 15ull) == 0) {
        const float4 t0 = ((const float4*)t)[0], t1 = ((const float4*)t)[1];
        const float4 p1 = ((const float4*)p)[1];
        th_[0] = t0.x; th_[1] = t0.y; th_[2] = t0.z; th_[3] = t0.w; th_[4] = t1.x; th_[5] = t1.y; th_[6] = t1.z; th_[7] = t1.w;
        ph_[4] = p1.x; ph_[5] = p1.y; ph_[6] = p1.z; ph_[7] = p1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          th_[j] = j < 5 + A ? t[j] : 0.f;
          ph_[j] = (j >= 4 && j < 5 + A) ? p[j] : 0.f;
        }
      }
      const float t_x = th_[0], t_y = th_[1], t_w = th_[2], t_h = th_[3], tob = th_[4];
      obj = tob > 0.5f ? 1.f : 0.f;
      const float ignore = (max_iou > c.ignore_thresh && tob < 0.5f) ? 1.f : 0.f;
      float tanc[MAXA], panc[MAXA];
#pragma unroll
      for (int j = 0; j < MAXA; ++j) {
        tanc[j] = j < A ? (j < 3 ? th_[5 + j] : t[5 + j]) : 0.f;
        panc[j] = j < A ? (j < 3 ? ph_[5 + j] : p[5 + j]) : 0.f;
      }
      int kstar = 0;
      {
        float best = tanc[0];
#pragma unroll
        for (int j = 1; j < MAXA; ++j)
          if (j < A && tanc[j] > best) { best = tanc[j]; kstar = j; }
      }
      float assigned = 0.f;
#pragma unroll
      for (int j = 0; j < MAXA; ++j)
        if (j == kstar) assigned = iou_a[j];
      assigned *= obj;
      a.assigned[a.cell_off[l] + cell0 + ci] = assigned;
      float tgt = tob;
      if (c.use_iou_aware_objectness) {
        const float piou = fminf(fmaxf(assigned, 0.f), 1.f);
        const float blended = c.iou_objectness_ratio * powf(piou + KEPS, c.iou_objectness_power) +
                              (1.f - c.iou_objectness_ratio) * tob;
        tgt = obj * blended + (1.f - obj) * tgt;
      }
      float wobj = obj * c.object_scale + (1.f - obj) * (1.f - ignore) * c.no_object_scale;
      if (c.trainable_nms_weight > 0.f)
        wobj += (1.f - obj) * ignore * c.trainable_nms_weight *
                powf(fminf(fmaxf(max_iou, 0.f), 1.f) + KEPS, c.trainable_nms_power);
      float g[5 + MAXA];
      {                                                 // x, y: squared error on the activated offsets
        const float dx = ax - t_x, dy = ay - t_y;
        s_loc += obj * dx * dx;
        s_loc += obj * dy * dy;
        g[0] = c.coord_scale * obj * 2.f * dx * xy_act_grad(pxr) * inv_nf;
        g[1] = c.coord_scale * obj * 2.f * dy * xy_act_grad(pyr) * inv_nf;
        const float dw = pwr - t_w, dh = phr - t_h;     // w, h: squared error on the raw log-ratios
        s_loc += obj * dw * dw;
        s_loc += obj * dh * dh;
        g[2] = c.coord_scale * obj * 2.f * dw * inv_nf;
        g[3] = c.coord_scale * obj * 2.f * dh * inv_nf;
      }
      {
        const float pob = ph_[4];
        s_obj += bce_logits(tgt, pob) * wobj;
        g[4] = c.object_scale * wobj * (sigmoidf(pob) - tgt) * inv_nf;
      }
      const float m = obj * (1.f - ignore);
#pragma unroll
      for (int j = 0; j < MAXA; ++j)
        if (j < A) {
          s_anc += bce_logits(tanc[j], panc[j]) * m;
          g[5 + j] = c.anchor_scale * c.anchor_scale * m * (sigmoidf(panc[j]) - tanc[j]) * inv_nf;
        }
      float* gf = a.gf[l] ? a.gf[l] + (cell0 + ci) * F : nullptr;
      bf16_t* gb = a.gb[l] ? a.gb[l] + (cell0 + ci) * F : nullptr;
#pragma unroll
      for (int j = 0; j < 5 + MAXA; ++j)
        if (j < 5 + A) {
          const float v = g[j] * gscale;
          if (gf) gf[j] = v;
          if (gb) gb[j] = f2bf(v);
        }
    }
    cell_obj[ci] = obj;
  }
  __syncthreads();

  // ---- B: class channels, element (cell, class) = e / C, e % C.  U elements per thread and round, all of their loads issued
  // before the first is used: one element per round was one dependent memory round trip per element (20 per thread).
  {
    constexpr int U = 4;
    const int nel = ncell * C;
    const int dq = 256 / C, dr = 256 - dq * C;
    int cell = tid / C, k = tid - cell * C;
    const float* Pc = P + 5 + A;
    const float* Tc = T + 5 + A;
    float* gfc = a.gf[l] ? a.gf[l] + cell0 * F + 5 + A : nullptr;
    bf16_t* gbc = a.gb[l] ? a.gb[l] + cell0 * F + 5 + A : nullptr;
    for (int e0 = tid; e0 < nel; e0 += 256 * U) {
      long long off[U];
      int kk[U], cc[U];
      float pv[U], tv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cc[u] = cell; kk[u] = k;
        off[u] = (long long)cell * F + k;
        const bool ok = e0 + u * 256 < nel;
        pv[u] = ok ? Pc[off[u]] : 0.f;
        tv[u] = ok ? Tc[off[u]] : 0.f;
        cell += dq;
        k += dr;
        if (k >= C) { k -= C; ++cell; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e0 + u * 256 >= nel) break;
        const float obj = cell_obj[cc[u]];
        const float cw = a.class_w ? a.class_w[kk[u]] : 1.f;
        float gout;
        if (c.use_focal_loss) {
          const float pr = sigmoidf(pv[u]);
          const float pt = tv[u] * pr + (1.f - tv[u]) * (1.f - pr);
          const float om = fmaxf(1.f - pt, 0.f);
          const float mod = powf(om, c.focal_gamma);
          const float at = tv[u] * c.focal_alpha + (1.f - tv[u]) * (1.f - c.focal_alpha);
          const float bce = bce_logits(tv[u], pv[u]);
          s_cls += mod * at * bce * cw * obj;
          const float dmod = om > 0.f ? -c.focal_gamma * powf(om, c.focal_gamma - 1.f) * (2.f * tv[u] - 1.f) * pr * (1.f - pr) : 0.f;
          gout = c.class_scale * cw * obj * at * (dmod * bce + mod * (pr - tv[u])) * inv_nf;
        } else {
          const float ts = c.label_smoothing > 0.f ? tv[u] * (1.f - c.label_smoothing) + c.label_smoothing / (float)C : tv[u];
          s_cls += bce_logits(ts, pv[u]) * cw * obj;
          gout = c.class_scale * cw * obj * (sigmoidf(pv[u]) - ts) * inv_nf;
        }
        gout *= gscale;
        if (gfc) gfc[off[u]] = gout;
        if (gbc) gbc[off[u]] = f2bf(gout);
      }
    }
  }
  s_loc = wave_sum(s_loc); s_obj = wave_sum(s_obj); s_anc = wave_sum(s_anc); s_cls = wave_sum(s_cls);
  if (lane == 0) { wsum[wave][0] = s_loc; wsum[wave][1] = s_obj; wsum[wave][2] = s_anc; wsum[wave][3] = s_cls; }
  __syncthreads();
  if (tid < 4) {
    float v = wsum[0][tid] + wsum[1][tid] + wsum[2][tid] + wsum[3][tid];
    v *= inv_nf;
    if (tid == 2) v *= c.anchor_scale;                 // reference accumulates anchor_scale * anchor_loss (:349,:390)
    atomicAdd(a.acc + tid, (double)v);
  }
}

// ---- K3: variance consensus (reference :930-1043), kernel size 3.  One wave per centre cell.
__global__ __launch_bounds__(256) void loss_consensus_kernel(LossArgs a, int l) {
  const mgd_loss_cfg& c = a.cfg;
  const int gh = c.grid_h[l], gw = c.grid_w[l], A = c.A, C = c.C, F = 5 + A + C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ncell = (long long)c.B * gh * gw;
  const long long gcell = (long long)blockIdx.x * 4 + wave;
  if (gcell >= ncell) return;
  const float* t = a.yt[l] + gcell * F;
  float t0 = t[0], t1 = t[1], tob = t[4];
  if (!(tob > 0.5f && t0 >= 0.f && t0 < 1.f && t1 >= 0.f && t1 < 1.f)) return;
  const int b = (int)(gcell / (gh * gw));
  const int cell = (int)(gcell - (long long)b * gh * gw);
  const int row = cell / gw, col = cell - row * gw;
  const float cx0 = t0 + (float)row, cy0 = t1 + (float)col;   // transposed grid again (:937)
  // weights over the 3x3 patch (computed redundantly by every lane)
  float w[9];
  long long nb[9];
  float wsumv = 0.f;
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    int rr = row + p / 3 - 1, cc = col + p % 3 - 1;
    w[p] = 0.f;
    nb[p] = -1;
    if (rr < 0 || rr >= gh || cc < 0 || cc >= gw) continue;
    long long n = (long long)b * gh * gw + rr * gw + cc;
    nb[p] = n;
    const float* tn = a.yt[l] + n * F;
    float m = tn[4] > 0.5f ? 1.f : 0.f;
    float dx = fabsf(tn[0] + (float)rr - cx0), dy = fabsf(tn[1] + (float)cc - cy0);
    float same = fmaxf(dx, dy) < c.consensus_center_tolerance ? 1.f : 0.f;
    float gm = m * same;
    if (gm > 0.f) {
      float iou = fmaxf(a.assigned[a.cell_off[l] + n], c.consensus_min_iou);
      w[p] = powf(iou, c.consensus_iou_power) * gm;
    }
    wsumv += w[p];
  }
  float inv = 1.0f / (wsumv + KEPS);
#pragma unroll
  for (int p = 0; p < 9; ++p) w[p] *= inv;
  float wtot = 0.f;
#pragma unroll
  for (int p = 0; p < 9; ++p) wtot += w[p];
  const float normalizer = fmaxf((float)a.ncenter[l], 1.f);
  float s_coord = 0.f, s_objv = 0.f, s_clsv = 0.f;
  for (int ch = lane; ch < F; ch += 64) {
    if (ch >= 5 && ch < 5 + A) continue;      // anchor logits take no part
    const bool is_box = ch < 4, is_obj = ch == 4;
    float v[9], dv[9];
    float cons = 0.f;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      float raw = nb[p] >= 0 ? a.yp[l][nb[p] * F + ch] : 0.f;
      if (is_box) { v[p] = nb[p] >= 0 ? raw : 0.f; dv[p] = 1.f; }
      else {
        float s = sigmoidf(raw);
        v[p] = nb[p] >= 0 ? s : 0.f;           // zero-padded patch of probabilities
        dv[p] = s * (1.f - s);
      }
      cons += w[p] * v[p];
    }
    float var = 0.f, lin = 0.f;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      float d = v[p] - cons;
      var += w[p] * d * d;
      lin += w[p] * d;
    }
    float coef = is_box ? c.consensus_coord_scale / normalizer
                        : (is_obj ? c.consensus_obj_scale / normalizer
                                  : c.consensus_class_scale / (normalizer * (float)C));
    if (is_box) s_coord += var; else if (is_obj) s_objv += var; else s_clsv += var;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      if (nb[p] < 0 || w[p] == 0.f) continue;
      float d = v[p] - cons;
      float g = 2.f * w[p] * d;
      if (!c.consensus_stop_gradient) g -= 2.f * w[p] * lin * wtot;   // d cons / d v_p = w_p
      g *= dv[p] * coef * c.grad_out_scale;
      atomicAdd(a.gf[l] + nb[p] * F + ch, g);
    }
  }
  s_coord = wave_sum(s_coord); s_objv = wave_sum(s_objv); s_clsv = wave_sum(s_clsv);
  if (lane == 0) {
    atomicAdd(a.acc + 4, (double)(s_coord / normalizer));
    atomicAdd(a.acc + 5, (double)(s_objv / normalizer));
    atomicAdd(a.acc + 6, (double)(s_clsv / (normalizer * (float)C)));
  }
}

__global__ void loss_finalize_kernel(LossArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const mgd_loss_cfg& c = a.cfg;
  float v[8];
  for (int i = 0; i < 7; ++i) v[i] = (float)a.acc[i];
  float tot = c.coord_scale * v[0] + c.object_scale * v[1] + c.anchor_scale * v[2] + c.class_scale * v[3];
  if (c.use_consensus_loss)
    tot += c.consensus_coord_scale * v[4] + c.consensus_obj_scale * v[5] + c.consensus_class_scale * v[6];
  v[7] = tot;
  for (int i = 0; i < 8; ++i) a.components[i] = v[i];
}

struct WsLayout {
  size_t gt_count, npos, ncenter, acc, wt, gt, assigned, gscratch, total;
  long long cell_off[MAXL + 1];
  long long wt_off[MAXL + 1];
};

WsLayout ws_layout(const mgd_loss_cfg* c) {
  WsLayout w;
  long long cells = 0;
  for (int l = 0; l < c->L; ++l) {
    w.cell_off[l] = cells;
    cells += (long long)c->B * c->grid_h[l] * c->grid_w[l];
  }
  w.cell_off[c->L] = cells;
  long long nwt = 0;
  for (int l = 0; l < c->L; ++l) {
    w.wt_off[l] = nwt;
    nwt += (long long)c->B * c->grid_h[l];
  }
  w.wt_off[c->L] = nwt;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
  w.acc = take(8 * sizeof(double));
  w.gt_count = take((size_t)c->L * c->B * 4);
  w.npos = take(MAXL * 4);
  w.ncenter = take(MAXL * 4);
  w.wt = take((size_t)nwt * 4);          // zeroed with the header (everything before `gt`)
  w.gt = take((size_t)cells * 16);
  w.assigned = take((size_t)cells * 4);
  w.gscratch = take(c->use_consensus_loss ? (size_t)cells * (5 + c->A + c->C) * 4 : 0);
  w.total = o;
  return w;
}

}  // namespace

extern "C" size_t mgd_loss_workspace_size(const mgd_loss_cfg* cfg) {
  if (!cfg || cfg->L < 1 || cfg->L > MAXL) return 0;
  return ws_layout(cfg).total;
}

extern "C" int mgd_loss_fwd_bwd(const mgd_loss_cfg* cfg, const float* const* y_pred_host,
                                const float* const* y_true_host, const float* class_weights,
                                float* const* grad_f32_host, void* const* grad_bf16_host, float* components,
                                void* ws, size_t ws_bytes, void* stream) {
  MGD_REQUIRE(cfg && y_pred_host && y_true_host && components && ws, "loss: null pointer");
  MGD_REQUIRE(cfg->L >= 1 && cfg->L <= MAXL && cfg->A >= 1 && cfg->A <= MAXA, "loss: L=%d A=%d unsupported", cfg->L,
              cfg->A);
  MGD_REQUIRE(5 + cfg->A + cfg->C <= 128, "loss: 5+A+C=%d exceeds 128 channels", 5 + cfg->A + cfg->C);
  MGD_REQUIRE(cfg->loss_option >= 1 && cfg->loss_option <= 3, "loss: loss_option=%d", cfg->loss_option);
  MGD_REQUIRE(cfg->use_focal_loss == 0 || cfg->use_focal_loss == 1, "loss: use_focal_loss");
  MGD_REQUIRE(cfg->iou_loss >= 0 && cfg->iou_loss <= 3 && (cfg->iou_loss == 0 || cfg->loss_option == 3),
              "loss: iou_loss=%d needs loss_option 3", cfg->iou_loss);
  MGD_REQUIRE((cfg->iou_compat | 1) == 1 && (cfg->softmax_compat | 1) == 1 && (cfg->use_softmax_focal | 1) == 1,
              "loss: iou_compat / softmax_compat / use_softmax_focal must be 0 or 1");
  const bool need_wt = (cfg->iou_loss != 0 && cfg->iou_compat == 0) || (cfg->use_softmax_focal && cfg->softmax_compat == 0);
  if (need_wt) {
    // the reference multiplies [B,H,W] by [B,H,W,1]: TensorFlow broadcasts that only for H == W and B in {1, H}
    // (iou_losses.py:70-93, 140-158; multigrid_loss.py:815-828) and raises InvalidArgumentError otherwise
    for (int l = 0; l < cfg->L; ++l) {
      MGD_REQUIRE(cfg->grid_h[l] == cfg->grid_w[l] && (cfg->B == 1 || cfg->B == cfg->grid_h[l]),
                  "loss: tf_ref broadcast of [B,H,W] * [B,H,W,1] is undefined for B=%d, grid %dx%d (scale %d); "
                  "use compat='fixed'", cfg->B, cfg->grid_h[l], cfg->grid_w[l], l);
      MGD_REQUIRE(!(cfg->use_softmax_focal && cfg->softmax_compat == 0) || cfg->C == 1 || cfg->C == cfg->grid_w[l],
                  "loss: tf_ref softmax focal multiplies class_weights [1,1,1,C] along W: needs C == 1 or C == W "
                  "(C=%d, W=%d)", cfg->C, cfg->grid_w[l]);
    }
  }
  WsLayout w = ws_layout(cfg);
  if (ws_bytes < w.total) return mgd_set_error(MGD_ENOSPC, "loss: workspace %zu < %zu", ws_bytes, w.total);
  LossArgs a;
  a.cfg = *cfg;
  char* base = (char*)ws;
  a.acc = (double*)(base + w.acc);
  a.gt_count = (int*)(base + w.gt_count);
  a.npos = (int*)(base + w.npos);
  a.ncenter = (int*)(base + w.ncenter);
  a.gt = (float4*)(base + w.gt);
  a.assigned = (float*)(base + w.assigned);
  a.wt = need_wt ? (float*)(base + w.wt) : nullptr;
  for (int l = 0; l <= cfg->L; ++l) a.wt_off[l] = w.wt_off[l];
  for (int l = cfg->L + 1; l <= MAXL; ++l) a.wt_off[l] = 0;
  a.class_w = class_weights;
  a.components = components;
  float* scratch = (float*)(base + w.gscratch);
  bool scratch_used[MAXL] = {false, false, false, false};
  for (int l = 0; l <= cfg->L; ++l) a.cell_off[l] = w.cell_off[l];
  for (int l = 0; l < MAXL; ++l) { a.yp[l] = a.yt[l] = nullptr; a.gf[l] = nullptr; a.gb[l] = nullptr; }
  const int F = 5 + cfg->A + cfg->C;
  for (int l = 0; l < cfg->L; ++l) {
    a.yp[l] = y_pred_host[l];
    a.yt[l] = y_true_host[l];
    MGD_REQUIRE(a.yp[l] && a.yt[l], "loss: y_pred/y_true[%d] null", l);
    a.gf[l] = grad_f32_host ? grad_f32_host[l] : nullptr;
    a.gb[l] = grad_bf16_host ? (bf16_t*)grad_bf16_host[l] : nullptr;
    if (cfg->use_consensus_loss && !a.gf[l]) {   // consensus scatters into an f32 image
      a.gf[l] = scratch + w.cell_off[l] * F;
      scratch_used[l] = true;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(base, 0, w.gt, st) != hipSuccess) return mgd_set_error(MGD_ELAUNCH, "loss: memset failed");
  long long max_cell = 0;
  for (int l = 0; l < cfg->L; ++l) max_cell = std::max(max_cell, (long long)cfg->B * cfg->grid_h[l] * cfg->grid_w[l]);
  hipLaunchKernelGGL(loss_prep_kernel, dim3(cdiv(max_cell, 256), cfg->L), dim3(256), 0, st, a);
  LossArgs al = a;
  CellsPerBlock cpbs;
  int max_gx = 1;
  for (int l = 0; l < MAXL; ++l) cpbs.v[l] = CELLS_PER_BLOCK;
  for (int l = 0; l < cfg->L; ++l) {
    if (scratch_used[l]) al.gb[l] = nullptr;    // bf16 image is produced after the consensus scatter
    // a wave walks its cells one after the other and every cell is a dependent chain of global loads, so the launch
    // is latency bound: 64 cells per block left the 19x19 grid on 96 blocks (69 us for 5776 cells).  Aim for >= 2048
    // blocks; the per-block cost that grows with it is the GT list staged in LDS (<= 16 KB, usually ~1.5 KB).
    const int cells = cfg->grid_h[l] * cfg->grid_w[l];
    int cpb = CELLS_PER_BLOCK;
    while (cpb > 4 && (long long)cdiv(cells, cpb) * cfg->B < 2048) cpb >>= 1;
    cpbs.v[l] = cpb;
    max_gx = std::max(max_gx, (int)cdiv(cells, cpb));
  }
  if (cfg->iou_loss == 0 && !cfg->use_softmax_focal) {
    long long max_cells = 0;
    for (int l = 0; l < cfg->L; ++l) max_cells = std::max(max_cells, (long long)cfg->grid_h[l] * cfg->grid_w[l]);
    hipLaunchKernelGGL(loss_cell2_kernel, dim3(cdiv(max_cells, C2_CELLS), cfg->B, cfg->L), dim3(256), 0, st, al);
  } else {
    hipLaunchKernelGGL(loss_cell_kernel, dim3(max_gx, cfg->B, cfg->L), dim3(256), 0, st, al, cpbs);
  }
  if (cfg->use_consensus_loss) {
    for (int l = 0; l < cfg->L; ++l) {
      if (!a.gf[l]) continue;
      long long ncell = (long long)cfg->B * cfg->grid_h[l] * cfg->grid_w[l];
      hipLaunchKernelGGL(loss_consensus_kernel, dim3(cdiv(ncell, 4)), dim3(256), 0, st, a, l);
    }
    for (int l = 0; l < cfg->L; ++l) {
      if (!a.gb[l] || !a.gf[l]) continue;
      long long n = (long long)cfg->B * cfg->grid_h[l] * cfg->grid_w[l] * F;
      int rc = mgd_f32_to_bf16(a.gf[l], a.gb[l], n, stream);
      if (rc != MGD_OK) return rc;
    }
  }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, a);
  MGD_CHECK_LAUNCH("loss_fwd_bwd");
  return MGD_OK;
}
