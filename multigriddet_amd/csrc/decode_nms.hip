// Batched decode -> correct_boxes -> confidence filter, and greedy NMS -> top-k -> xyxy, on gfx950.
// Replaces MultiGridDecoder._decode_single_scale / correct_boxes / handle_predictions /
// _convert_to_xyxy (reference multigriddet/postprocess/multigrid_decode.py:100-235, 237-345, 397-422)
// and StandardNMS / DIoUNMS / ClusterNMS (reference multigriddet/postprocess/nms.py:83-231, 320-385).
//
// HBM-bound on the head tensors (2.67 MB per 608^2 image); everything after the confidence filter
// lives in LDS / L2.  The reference runs this per image on the host behind a device->host copy; here
// all images of a batch are processed by one launch each (one block per image for compaction/NMS).
// Float pair tests are evaluated in fp32 in numpy's operation order with contraction off, so the
// keep/suppress decisions match the reference's numpy NMS on fp32 boxes bit for bit.
#include "common.h"

namespace {

constexpr int MAXL = 4, MAXA = 8;

struct DecArgs {
  mgd_decode_cfg cfg;
  const float* yp[MAXL];
  const float* image_hw;   // [B][2]
  int cells[MAXL + 1];     // prefix of gh*gw
  float* dboxes;           // dense [B][T][4]
  float* dscores;          // dense [B][T]
  int* dcls;               // dense [B][T]
  float* oboxes;
  float* oscores;
  int* ocls;
  int* ocount;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void decode_kernel(DecArgs a) {
  const mgd_decode_cfg& c = a.cfg;
  const int T = a.cells[c.L];
  const int A = c.A, C = c.C, F = 5 + A + C;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)c.B * T) return;
  const int b = (int)(i / T);
  const int k = (int)(i - (long long)b * T);
  int l = 0;
  while (l + 1 < c.L && k >= a.cells[l + 1]) ++l;
  const int gh = c.grid_h[l], gw = c.grid_w[l];
  const int cell = k - a.cells[l];
  const int row = cell / gw, col = cell - row * gw;
  const float* p = a.yp[l] + ((long long)b * gh * gw + cell) * F;

  // anchor probabilities
  int ai = 0;
  float amax = p[5];
  for (int j = 1; j < A; ++j)
    if (p[5 + j] > amax) { amax = p[5 + j]; ai = j; }
  float best_anchor;
  if (c.use_softmax) {
    float s = 0.f;
    for (int j = 0; j < A; ++j) s += expf(p[5 + j] - amax);
    best_anchor = 1.0f / s;
  } else {
    best_anchor = sigm(amax);
  }
  // class probabilities
  int ci = 0;
  float cmax = p[5 + A];
  for (int j = 1; j < C; ++j) {
    float v = p[5 + A + j];
    if (v > cmax) { cmax = v; ci = j; }
  }
  float best_class;
  if (c.use_softmax) {
    float s = 0.f;
    for (int j = 0; j < C; ++j) s += expf(p[5 + A + j] - cmax);
    best_class = 1.0f / s;
  } else {
    best_class = sigm(cmax);
  }
  float score = sigm(p[4]);
  if (c.rescore) score = score * best_anchor * best_class;

  float ax = tanhf(0.15f * p[0]) + sigm(0.15f * p[0]);
  float ay = tanhf(0.15f * p[1]) + sigm(0.15f * p[1]);
  // reference divides x by grid_h and y by grid_w, w by input_h and h by input_w (:116,:155,:163)
  float bx = (ax + (float)col) / (float)gh;
  float by = (ay + (float)row) / (float)gw;
  float bw = c.anchors[l][ai][0] * expf(p[2]) / (float)c.in_h;
  float bh = c.anchors[l][ai][1] * expf(p[3]) / (float)c.in_w;

  // correct_boxes (:185-235)
  float ih = a.image_hw[b * 2 + 0], iw = a.image_hw[b * 2 + 1];
  float mh = (float)c.in_h, mw = (float)c.in_w;
  float r = fminf(mh / ih, mw / iw);
  float nh = rintf(ih * r), nw = rintf(iw * r);
  float offx = (mw - nw) / 2.0f / mw, offy = (mh - nh) / 2.0f / mh;
  float scx = mw / nw, scy = mh / nh;
  bx = (bx - offx) * scx;
  by = (by - offy) * scy;
  bw = bw * scx;
  bh = bh * scy;
  bx = bx - bw / 2.0f;
  by = by - bh / 2.0f;
  bx *= iw; by *= ih; bw *= iw; bh *= ih;

  long long o = (long long)b * T + k;
  *(float4*)(a.dboxes + o * 4) = make_float4(bx, by, bw, bh);
  a.dscores[o] = score;
  a.dcls[o] = c.tag_scale ? (ci | (l << 16)) : ci;     // per-scale NMS: the scale rides in the upper half of the class id
}

// ordered compaction: one block per image
__global__ __launch_bounds__(1024) void compact_kernel(DecArgs a) {
  const mgd_decode_cfg& c = a.cfg;
  const int T = a.cells[c.L];
  const int b = blockIdx.x;
  __shared__ int wsum[16];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k0 = 0; k0 < T; k0 += 1024) {
    int k = k0 + threadIdx.x;
    bool keep = k < T && a.dscores[(long long)b * T + k] >= c.confidence;
    unsigned long long m = __ballot(keep);
    int inwave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (keep) {
      int pos = off + inwave;
      if (pos < c.cap) {
        long long src = (long long)b * T + k, dst = (long long)b * c.cap + pos;
        *(float4*)(a.oboxes + dst * 4) = *(const float4*)(a.dboxes + src * 4);
        a.oscores[dst] = a.dscores[src];
        a.ocls[dst] = a.dcls[src];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) a.ocount[b] = min(base, c.cap);
  // candidates past the count read as zeros (the caller hands over uninitialised buffers)
  for (int pos = min(base, c.cap) + (int)threadIdx.x; pos < c.cap; pos += 1024) {
    const long long dst = (long long)b * c.cap + pos;
    *(float4*)(a.oboxes + dst * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    a.oscores[dst] = 0.f;
    a.ocls[dst] = 0;
  }
}

struct NmsArgs {
  const float* boxes;   // [B][cap][4]
  const float* scores;  // [B][cap]
  const int* cls;       // [B][cap]
  const int* count;     // [B]
  int B, cap, method, max_boxes, return_xyxy, npow2, per_scale;
  float thr;
  const float* image_hw;
  void* out_boxes;
  float* out_scores;
  int* out_cls;
  int* out_count;
  float* sorted;        // ws [B][cap][4]
};

// Output slots past the image's detection count read as zeros (out_boxes is 16 bytes per slot in either format).
__device__ __forceinline__ void zero_tail(const NmsArgs& a, int b, int K, int tid) {
  for (int k = K + tid; k < a.max_boxes; k += 1024) {
    const long long o = (long long)b * a.max_boxes + k;
    *(float4*)((float*)a.out_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    a.out_scores[o] = 0.f;
    a.out_cls[o] = 0;
  }
}

__device__ __forceinline__ float pair_metric(const float4 A, const float4 Bx, int method) {
  float x1 = A.x, y1 = A.y, w1 = A.z, h1 = A.w, x2 = Bx.x, y2 = Bx.y, w2 = Bx.z, h2 = Bx.w;
  float ixmin = fmaxf(x1, x2), iymin = fmaxf(y1, y2);
  float ixmax = fminf(x1 + w1, x2 + w2), iymax = fminf(y1 + h1, y2 + h2);
  float iw = fmaxf(0.0f, ixmax - ixmin), ih = fmaxf(0.0f, iymax - iymin);
  float inter = iw * ih;
  float area1 = w1 * h1, area2 = w2 * h2;
  float uni = area1 + area2 - inter;
  float iou = inter / (uni + 1e-8f);
  if (method == 0) return iou;
  float c1x = x1 + w1 / 2.0f, c1y = y1 + h1 / 2.0f, c2x = x2 + w2 / 2.0f, c2y = y2 + h2 / 2.0f;
  float dx = c1x - c2x, dy = c1y - c2y;
  float cd = dx * dx + dy * dy;
  float ex = fmaxf(x1 + w1, x2 + w2) - fminf(x1, x2), ey = fmaxf(y1 + h1, y2 + h2) - fminf(y1, y2);
  float ed = ex * ex + ey * ey;
  return iou - cd / (ed + 1e-8f);
}

// block-wide (1024 threads) bitonic sort of np2 (power of two) 64-bit keys in LDS, descending
__device__ __forceinline__ void bitonic_desc(unsigned long long* keys, int np2, int tid) {
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 1024) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long x = keys[i], y = keys[ixj];
          bool desc = (i & k) == 0;
          if (desc ? x < y : x > y) { keys[i] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
}

// one block (1024 threads) per image: bitonic sort by (score, index) descending in LDS, then greedy
// suppression with an alive bitmask; kept boxes come out in descending score order.
__global__ __launch_bounds__(1024) void nms_kernel(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = (unsigned long long*)smem;
  unsigned int* alive = (unsigned int*)(smem + (size_t)a.npow2 * 8);
  __shared__ int kept_n;
  __shared__ int kept_idx[1024];
  const int b = blockIdx.x;
  const int n = min(a.count[b], a.cap);
  const int tid = threadIdx.x;
  const float* bx = a.boxes + (long long)b * a.cap * 4;
  const float* sc = a.scores + (long long)b * a.cap;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = tid; i < np2; i += 1024)
    keys[i] = i < n ? (((unsigned long long)__float_as_uint(fmaxf(sc[i], 0.f)) << 32) | (unsigned)i) : 0ull;
  __syncthreads();
  bitonic_desc(keys, np2, tid);
  // gather boxes in sorted order (global scratch, L2 resident)
  float* sb = a.sorted + (long long)b * a.cap * 4;
  int* sseg = (int*)(a.sorted + (long long)a.B * a.cap * 4) + (long long)b * a.cap;   // per-scale mode only
  for (int i = tid; i < n; i += 1024) {
    int src = (int)(keys[i] & 0xffffffffu);
    *(float4*)(sb + (long long)i * 4) = *(const float4*)(bx + (long long)src * 4);
    if (a.per_scale) sseg[i] = a.cls[(long long)b * a.cap + src] >> 16;
  }
  const int nw = (n + 31) >> 5;
  for (int i = tid; i < nw; i += 1024) {
    int rem = n - i * 32;
    alive[i] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
  }
  if (tid == 0) kept_n = 0;
  __threadfence_block();
  __syncthreads();
  int i = 0, K = 0;
  const int kmax = min(a.max_boxes, 1024);
  while (K < kmax) {
    // next alive index >= i (uniform: every thread walks the same words)
    int w = i >> 5;
    unsigned int m = w < nw ? (alive[w] & (0xffffffffu << (i & 31))) : 0u;
    while (!m && ++w < nw) m = alive[w];
    if (!m) break;
    i = w * 32 + __ffs(m) - 1;
    if (tid == 0) kept_idx[K] = i;
    ++K;
    float4 cur = *(const float4*)(sb + (long long)i * 4);
    const int cseg = a.per_scale ? sseg[i] : 0;
    for (int j = i + 1 + tid; j < n; j += 1024) {
      if (alive[j >> 5] & (1u << (j & 31))) {
        if (a.per_scale && sseg[j] != cseg) continue;      // a box only suppresses boxes of its own scale
        float4 o = *(const float4*)(sb + (long long)j * 4);
        if (!(pair_metric(cur, o, a.method) < a.thr)) atomicAnd(&alive[j >> 5], ~(1u << (j & 31)));
      }
    }
    ++i;
    __syncthreads();
  }
  if (tid == 0) a.out_count[b] = K;
  zero_tail(a, b, K, tid);
  __syncthreads();
  float ih = a.image_hw[b * 2], iw = a.image_hw[b * 2 + 1];
  for (int k = tid; k < K; k += 1024) {
    int si = kept_idx[k];
    int src = (int)(keys[si] & 0xffffffffu);
    float4 B4 = *(const float4*)(bx + (long long)src * 4);
    long long o = (long long)b * a.max_boxes + k;
    if (a.return_xyxy) {
      float x0 = B4.x, y0 = B4.y, x1 = B4.x + B4.z, y1 = B4.y + B4.w;
      x0 = fminf(fmaxf(x0, 0.f), iw); y0 = fminf(fmaxf(y0, 0.f), ih);
      x1 = fminf(fmaxf(x1, 0.f), iw); y1 = fminf(fmaxf(y1, 0.f), ih);
      int* ob = (int*)a.out_boxes + o * 4;
      ob[0] = (int)floorf(x0 + 0.5f); ob[1] = (int)floorf(y0 + 0.5f);
      ob[2] = (int)floorf(x1 + 0.5f); ob[3] = (int)floorf(y1 + 0.5f);
    } else {
      *(float4*)((float*)a.out_boxes + o * 4) = B4;
    }
    a.out_scores[o] = sc[src];
    a.out_cls[o] = a.per_scale ? (a.cls[(long long)b * a.cap + src] & 0xffff) : a.cls[(long long)b * a.cap + src];
  }
}

__device__ __forceinline__ void write_box(const NmsArgs& a, long long o, double x, double y, double w, double h, float ih,
                                          float iw) {
  if (a.return_xyxy) {
    double x0 = x, y0 = y, x1 = x + w, y1 = y + h;
    x0 = fmin(fmax(x0, 0.0), (double)iw); y0 = fmin(fmax(y0, 0.0), (double)ih);
    x1 = fmin(fmax(x1, 0.0), (double)iw); y1 = fmin(fmax(y1, 0.0), (double)ih);
    int* ob = (int*)a.out_boxes + o * 4;
    ob[0] = (int)floor(x0 + 0.5); ob[1] = (int)floor(y0 + 0.5);
    ob[2] = (int)floor(x1 + 0.5); ob[3] = (int)floor(y1 + 0.5);
  } else {
    float* ob = (float*)a.out_boxes + o * 4;
    ob[0] = (float)x; ob[1] = (float)y; ob[2] = (float)w; ob[3] = (float)h;
  }
}

// SoftNMS (reference nms.py:234-317; sigma 0.5, score threshold 1e-3): fixed order = scores descending (never
// re-sorted); box i, unless its own decayed score has fallen below the threshold (then it is zeroed), multiplies
// the score of every later box by exp(-iou^2 / sigma).  Survivors leave in ORIGINAL candidate order with their
// decayed scores; if more than max_boxes survive, the top max_boxes by decayed score (postprocess _filter_boxes).
// ws per image: sorted boxes [cap][4] f32 | soft scores [cap] f32 | original index [cap] i32.
__global__ __launch_bounds__(1024) void soft_nms_kernel(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = (unsigned long long*)smem;
  __shared__ int surv;
  const int b = blockIdx.x;
  const int n = min(a.count[b], a.cap);
  const int tid = threadIdx.x;
  const float* bx = a.boxes + (long long)b * a.cap * 4;
  const float* sc = a.scores + (long long)b * a.cap;
  float* sb = a.sorted + (long long)b * a.cap * 6;
  float* ss = sb + (long long)a.cap * 4;
  int* so = (int*)(ss + a.cap);
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = tid; i < np2; i += 1024)
    keys[i] = i < n ? (((unsigned long long)__float_as_uint(fmaxf(sc[i], 0.f)) << 32) | (unsigned)i) : 0ull;
  if (tid == 0) surv = 0;
  __syncthreads();
  bitonic_desc(keys, np2, tid);
  for (int i = tid; i < n; i += 1024) {
    int src = (int)(keys[i] & 0xffffffffu);
    *(float4*)(sb + (long long)i * 4) = *(const float4*)(bx + (long long)src * 4);
    ss[i] = sc[src];
    so[i] = src;
  }
  __threadfence_block();
  __syncthreads();
  const float sthr = 0.001f, sigma = 0.5f;
  for (int i = 0; i < n; ++i) {
    const float s = ss[i];
    if (s < sthr) {                 // position i is never touched again inside the loop
      if (tid == 0) ss[i] = 0.f;
      continue;
    }
    if (i + 1 >= n) break;
    const float4 cur = *(const float4*)(sb + (long long)i * 4);
    for (int j = i + 1 + tid; j < n; j += 1024) {
      float iou = pair_metric(cur, *(const float4*)(sb + (long long)j * 4), 0);
      ss[j] *= expf(-(iou * iou) / sigma);
    }
    __threadfence_block();
    __syncthreads();
  }
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < n; i += 1024) mine += ss[i] >= sthr ? 1 : 0;
  if (mine) atomicAdd(&surv, mine);
  __syncthreads();
  const int K = surv;
  const bool by_score = K > a.max_boxes;
  for (int i = tid; i < np2; i += 1024) {
    unsigned long long k = 0ull;
    if (i < n && ss[i] >= sthr) {
      unsigned hi = by_score ? __float_as_uint(ss[i]) : (0x7fffffffu - (unsigned)so[i]);
      k = ((unsigned long long)hi << 32) | (unsigned)(i + 1);
    }
    keys[i] = k;
  }
  __syncthreads();
  bitonic_desc(keys, np2, tid);
  const int Ko = min(K, a.max_boxes);
  if (tid == 0) a.out_count[b] = Ko;
  zero_tail(a, b, Ko, tid);
  const float ih = a.image_hw[b * 2], iw = a.image_hw[b * 2 + 1];
  for (int k = tid; k < Ko; k += 1024) {
    int i = (int)(keys[k] & 0xffffffffu) - 1;
    int src = so[i];
    float4 B4 = *(const float4*)(bx + (long long)src * 4);
    long long o = (long long)b * a.max_boxes + k;
    write_box(a, o, B4.x, B4.y, B4.z, B4.w, ih, iw);
    a.out_scores[o] = ss[i];
    a.out_cls[o] = a.cls[(long long)b * a.cap + src];
  }
}

// Weighted Boxes Fusion as the reference runs it on one model's detections (wbf.py:79-199, called from
// multigrid_decode.py:281-287 with iou_thr = nms_threshold): per class (ascending id), boxes in descending score
// order; the first unused box seeds a cluster and every later unused box of the class with IoU(seed, box) >= thr
// joins it (IoU against the SEED, fp32, no epsilon, 0 when the boxes do not overlap).  Cluster box = score-weighted
// mean (float64 in the reference), cluster score = mean score.  More than max_boxes clusters -> top max_boxes by
// score.  ws per image: sorted boxes [cap][4] f32 | scores [cap] f32 | class [cap] i32 | cluster box [cap][4] f64 |
// cluster score [cap] f32 | cluster class [cap] i32.
__device__ __forceinline__ float wbf_iou(const float4 A, const float4 Bx) {
  float ixmin = fmaxf(A.x, Bx.x), iymin = fmaxf(A.y, Bx.y);
  float ixmax = fminf(A.x + A.z, Bx.x + Bx.z), iymax = fminf(A.y + A.w, Bx.y + Bx.w);
  if (ixmax <= ixmin || iymax <= iymin) return 0.0f;
  float inter = (ixmax - ixmin) * (iymax - iymin);
  float uni = A.z * A.w + Bx.z * Bx.w - inter;
  return uni > 0.f ? inter / uni : 0.0f;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) {
    long long bits = __double_as_longlong(v);
    int lo = __shfl_xor((int)(bits & 0xffffffffll), o, 64), hi = __shfl_xor((int)(bits >> 32), o, 64);
    v += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  return v;
}

__global__ __launch_bounds__(1024) void wbf_kernel(NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = (unsigned long long*)smem;
  unsigned int* alive = (unsigned int*)(smem + (size_t)a.npow2 * 8);
  __shared__ double part[16][6];
  const int b = blockIdx.x;
  const int n = min(a.count[b], a.cap);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* bx = a.boxes + (long long)b * a.cap * 4;
  const float* sc = a.scores + (long long)b * a.cap;
  const int* cl = a.cls + (long long)b * a.cap;
  unsigned char* wsb = (unsigned char*)a.sorted + (long long)b * a.cap * 64;
  float* sb = (float*)wsb;
  float* ss = sb + (long long)a.cap * 4;
  int* scl = (int*)(ss + a.cap);
  double* cb = (double*)(wsb + (long long)a.cap * 24);
  float* cs = (float*)(wsb + (long long)a.cap * 56);
  int* cc = (int*)(wsb + (long long)a.cap * 60);
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = tid; i < np2; i += 1024) {
    unsigned long long k = 0ull;
    if (i < n) {
      unsigned c = (unsigned)min(max(cl[i], 0), 65534);
      k = ((unsigned long long)(65535u - c) << 48) | ((unsigned long long)__float_as_uint(fmaxf(sc[i], 0.f)) << 16) |
          (unsigned long long)(i & 0xffff);
    }
    keys[i] = k;
  }
  __syncthreads();
  bitonic_desc(keys, np2, tid);
  for (int i = tid; i < n; i += 1024) {
    int src = (int)(keys[i] & 0xffffull);
    *(float4*)(sb + (long long)i * 4) = *(const float4*)(bx + (long long)src * 4);
    ss[i] = sc[src];
    scl[i] = cl[src];
  }
  const int nw = (n + 31) >> 5;
  for (int i = tid; i < nw; i += 1024) {
    int rem = n - i * 32;
    alive[i] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
  }
  __threadfence_block();
  __syncthreads();
  int i = 0, K = 0;
  while (true) {
    int w = i >> 5;
    unsigned int m = w < nw ? (alive[w] & (0xffffffffu << (i & 31))) : 0u;
    while (!m && ++w < nw) m = alive[w];
    if (!m) break;
    i = w * 32 + __ffs(m) - 1;
    __syncthreads();                       // everyone has read alive[] before it changes
    const float4 seed = *(const float4*)(sb + (long long)i * 4);
    const int c = scl[i];
    double acc[6] = {0, 0, 0, 0, 0, 0};    // sum s*x, s*y, s*w, s*h, sum s, count
    if (tid == 0) {
      double s0 = (double)ss[i];
      acc[0] = s0 * seed.x; acc[1] = s0 * seed.y; acc[2] = s0 * seed.z; acc[3] = s0 * seed.w; acc[4] = s0; acc[5] = 1.0;
    }
    for (int j = i + 1 + tid; j < n; j += 1024) {
      if (scl[j] != c) break;              // classes are contiguous in the sorted order
      if (alive[j >> 5] & (1u << (j & 31))) {
        float4 o = *(const float4*)(sb + (long long)j * 4);
        if (wbf_iou(seed, o) >= a.thr) {
          atomicAnd(&alive[j >> 5], ~(1u << (j & 31)));
          double sj = (double)ss[j];
          acc[0] += sj * o.x; acc[1] += sj * o.y; acc[2] += sj * o.z; acc[3] += sj * o.w; acc[4] += sj; acc[5] += 1.0;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      double v = wave_sum_f64(acc[q]);
      if (lane == 0) part[wave][q] = v;
    }
    __syncthreads();
    if (tid == 0) {
      double t[6] = {0, 0, 0, 0, 0, 0};
      for (int wv = 0; wv < 16; ++wv)
        for (int q = 0; q < 6; ++q) t[q] += part[wv][q];
      cb[(long long)K * 4 + 0] = t[0] / t[4]; cb[(long long)K * 4 + 1] = t[1] / t[4];
      cb[(long long)K * 4 + 2] = t[2] / t[4]; cb[(long long)K * 4 + 3] = t[3] / t[4];
      cs[K] = (float)(t[4] / t[5]);
      cc[K] = c;
    }
    ++K;
    ++i;
    __threadfence_block();
    __syncthreads();
  }
  // clusters are in (class ascending, seed score descending) order = the reference's output order
  const float ih = a.image_hw[b * 2], iw = a.image_hw[b * 2 + 1];
  const int Ko = min(K, a.max_boxes);
  if (tid == 0) a.out_count[b] = Ko;
  zero_tail(a, b, Ko, tid);
  if (K > a.max_boxes) {
    int kp2 = 1;
    while (kp2 < K) kp2 <<= 1;
    for (int k = tid; k < kp2; k += 1024)
      keys[k] = k < K ? (((unsigned long long)__float_as_uint(fmaxf(cs[k], 0.f)) << 32) | (unsigned)k) : 0ull;
    __syncthreads();
    bitonic_desc(keys, kp2, tid);
  }
  for (int k = tid; k < Ko; k += 1024) {
    int src = K > a.max_boxes ? (int)(keys[k] & 0xffffffffu) : k;
    long long o = (long long)b * a.max_boxes + k;
    write_box(a, o, cb[(long long)src * 4], cb[(long long)src * 4 + 1], cb[(long long)src * 4 + 2], cb[(long long)src * 4 + 3],
              ih, iw);
    a.out_scores[o] = cs[src];
    a.out_cls[o] = cc[src];
  }
}
#pragma clang fp contract(fast)

size_t dec_ws(const mgd_decode_cfg* c, size_t* o_scores, size_t* o_cls) {
  long long T = 0;
  for (int l = 0; l < c->L; ++l) T += (long long)c->grid_h[l] * c->grid_w[l];
  size_t boxes = ((size_t)c->B * T * 16 + 255) & ~(size_t)255;
  size_t scores = ((size_t)c->B * T * 4 + 255) & ~(size_t)255;
  if (o_scores) *o_scores = boxes;
  if (o_cls) *o_cls = boxes + scores;
  return boxes + 2 * scores;
}

}  // namespace

extern "C" size_t mgd_decode_workspace_size(const mgd_decode_cfg* cfg) {
  if (!cfg || cfg->L < 1 || cfg->L > MAXL) return 0;
  return dec_ws(cfg, nullptr, nullptr);
}

extern "C" int mgd_decode(const mgd_decode_cfg* cfg, const float* const* y_pred_host, const float* image_hw,
                          float* cand_boxes, float* cand_scores, int32_t* cand_cls, int32_t* cand_count, void* ws,
                          size_t ws_bytes, void* stream) {
  MGD_REQUIRE(cfg && y_pred_host && image_hw && cand_boxes && cand_scores && cand_cls && cand_count && ws,
              "decode: null pointer");
  MGD_REQUIRE(cfg->L >= 1 && cfg->L <= MAXL && cfg->A >= 1 && cfg->A <= MAXA && cfg->cap >= 1, "decode: L/A/cap");
  size_t os, oc;
  size_t need = dec_ws(cfg, &os, &oc);
  if (ws_bytes < need) return mgd_set_error(MGD_ENOSPC, "decode: workspace %zu < %zu", ws_bytes, need);
  DecArgs a;
  a.cfg = *cfg;
  a.cells[0] = 0;
  for (int l = 0; l < MAXL; ++l) a.yp[l] = nullptr;
  for (int l = 0; l < cfg->L; ++l) {
    a.yp[l] = y_pred_host[l];
    MGD_REQUIRE(a.yp[l] != nullptr, "decode: y_pred[%d] null", l);
    a.cells[l + 1] = a.cells[l] + cfg->grid_h[l] * cfg->grid_w[l];
  }
  a.image_hw = image_hw;
  a.dboxes = (float*)ws;
  a.dscores = (float*)((char*)ws + os);
  a.dcls = (int*)((char*)ws + oc);
  a.oboxes = cand_boxes; a.oscores = cand_scores; a.ocls = cand_cls; a.ocount = cand_count;
  hipStream_t st = (hipStream_t)stream;
  long long tot = (long long)cfg->B * a.cells[cfg->L];
  hipLaunchKernelGGL(decode_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(compact_kernel, dim3(cfg->B), dim3(1024), 0, st, a);
  MGD_CHECK_LAUNCH("decode");
  return MGD_OK;
}

extern "C" size_t mgd_nms_workspace_size(int B, int cap) { return (size_t)B * cap * 24; }   // soft: + score + index

extern "C" int mgd_nms(const float* cand_boxes, const float* cand_scores, const int32_t* cand_cls,
                       const int32_t* cand_count, int B, int cap, int method, float threshold, int max_boxes,
                       const float* image_hw, int return_xyxy, void* out_boxes, float* out_scores, int32_t* out_cls,
                       int32_t* out_count, void* ws, size_t ws_bytes, void* stream) {
  MGD_REQUIRE(cand_boxes && cand_scores && cand_cls && cand_count && image_hw && out_boxes && out_scores && out_cls &&
                  out_count && ws,
              "nms: null pointer");
  const int per_scale = (method & MGD_NMS_PER_SCALE) ? 1 : 0;
  method &= ~MGD_NMS_PER_SCALE;
  MGD_REQUIRE(method >= 0 && method <= 2, "nms: method %d unknown (0=iou, 1=diou, 2=soft)", method);
  MGD_REQUIRE(!(per_scale && method == 2), "nms: per-scale suppression is for the greedy methods (iou / diou)");
  MGD_REQUIRE(max_boxes >= 1 && max_boxes <= 1024, "nms: max_boxes=%d must be in [1,1024]", max_boxes);
  MGD_REQUIRE(cap >= 1 && cap <= 16384, "nms: cap=%d must be in [1,16384]", cap);
  if (ws_bytes < (size_t)B * cap * 24) return mgd_set_error(MGD_ENOSPC, "nms: workspace too small");
  NmsArgs a;
  a.boxes = cand_boxes; a.scores = cand_scores; a.cls = cand_cls; a.count = cand_count;
  a.B = B; a.cap = cap; a.method = method; a.max_boxes = max_boxes; a.return_xyxy = return_xyxy; a.thr = threshold;
  a.image_hw = image_hw; a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_cls = out_cls;
  a.out_count = out_count; a.sorted = (float*)ws; a.per_scale = per_scale;
  int np2 = 1;
  while (np2 < cap) np2 <<= 1;
  a.npow2 = np2;
  size_t lds = (size_t)np2 * 8 + (size_t)(np2 / 32 + 1) * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  if (method == 2) {
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute((const void*)soft_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr2 = true;
    }
    hipLaunchKernelGGL(soft_nms_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, a);
  }
  MGD_CHECK_LAUNCH("nms");
  return MGD_OK;
}

extern "C" size_t mgd_wbf_workspace_size(int B, int cap) { return (size_t)B * cap * 64; }

extern "C" int mgd_wbf(const float* cand_boxes, const float* cand_scores, const int32_t* cand_cls,
                       const int32_t* cand_count, int B, int cap, float iou_threshold, int max_boxes,
                       const float* image_hw, int return_xyxy, void* out_boxes, float* out_scores, int32_t* out_cls,
                       int32_t* out_count, void* ws, size_t ws_bytes, void* stream) {
  MGD_REQUIRE(cand_boxes && cand_scores && cand_cls && cand_count && image_hw && out_boxes && out_scores && out_cls &&
                  out_count && ws,
              "wbf: null pointer");
  MGD_REQUIRE(max_boxes >= 1 && max_boxes <= 1024, "wbf: max_boxes=%d must be in [1,1024]", max_boxes);
  MGD_REQUIRE(cap >= 1 && cap <= 16384, "wbf: cap=%d must be in [1,16384]", cap);
  if (ws_bytes < (size_t)B * cap * 64) return mgd_set_error(MGD_ENOSPC, "wbf: workspace too small");
  NmsArgs a;
  a.boxes = cand_boxes; a.scores = cand_scores; a.cls = cand_cls; a.count = cand_count;
  a.B = B; a.cap = cap; a.method = 0; a.max_boxes = max_boxes; a.return_xyxy = return_xyxy; a.thr = iou_threshold;
  a.image_hw = image_hw; a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_cls = out_cls;
  a.out_count = out_count; a.sorted = (float*)ws;
  int np2 = 1;
  while (np2 < cap) np2 <<= 1;
  a.npow2 = np2;
  size_t lds = (size_t)np2 * 8 + (size_t)(np2 / 32 + 1) * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wbf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(wbf_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, a);
  MGD_CHECK_LAUNCH("wbf");
  return MGD_OK;
}
