// 3x3 multi-grid y_true target builder on gfx950 (HBM-bound: the cost is writing the dense tensors).
//
// mode 0 (T1) replaces tf_preprocess_true_boxes  (reference multigriddet/data/generators.py:2696-3390)
// mode 1 (T2) replaces preprocess_true_boxes     (reference multigriddet/data/generators.py:3393-3473)
//
// T1: (a) one thread per box picks layer/anchor/cell exactly as the TF graph does (fp32, same
// operation order), and claims its <= 9 cells with atomicMax on a packed order key; the key is the
// position of the candidate in the reference's flattened (box, candidate) update list, so the
// winner is the LAST update - what TF-CPU's tensor_scatter_nd_update does with duplicate indices.
// (b) one pass over every output element writes either zeros or the winner's row, so y_true is
// written exactly once, 16 bytes per lane.
// T2 is order-dependent per image (its ">= 3 cells" skip rule reads what earlier boxes wrote), so one
// wavefront per image replays the boxes in order and keeps a per-cell owner table; the dense write
// pass is shared with T1.
#include "common.h"

namespace {

constexpr int MAXL = 4, MAXA = 8;

struct TgtArgs {
  const float* boxes;  // [B][M][5]
  int B, M, L, A, C, in_h, in_w, mode;
  int gh[MAXL], gw[MAXL];
  long long cell_off[MAXL + 1];  // prefix of B*gh*gw over layers (owner table offsets)
  float anchors[MAXL][MAXA][2];
  float* y[MAXL];
  int* owner;      // [sum_l B*gh*gw]  packed key (0 = empty)
  float* boxinfo;  // [B*M][12]: tx (double), ty (double), tw, th, layer, anchor, cls, valid, pad, pad
  int* assign;     // optional [B][M][4]
};

// fp32, no contraction: must match the TF graph's rounding sequence
#pragma clang fp contract(off)
__device__ void t1_select(const TgtArgs& a, float x1, float y1, float x2, float y2, int& layer, int& k, int& row,
                          int& col, float& tx, float& ty, float& tw, float& th, bool& valid) {
  float bx = (x1 + x2) / 2.0f, by = (y1 + y2) / 2.0f;
  float bw = x2 - x1, bh = y2 - y1;
  valid = (bw * bh) > 0.0f;
  layer = 0; k = 0; row = 0; col = 0; tx = ty = tw = th = 0.f;
  if (!valid) return;
  float best_layer_iol = -1.f;
  for (int l = 0; l < a.L; ++l) {
    float bl = -1.f;
    int bk = 0;
    for (int j = 0; j < a.A; ++j) {
      float aw = a.anchors[l][j][0], ah = a.anchors[l][j][1];
      float inter = fminf(bw, aw) * fminf(bh, ah);
      float iol = inter / (fmaxf(bw * bh, aw * ah) + 1e-7f);
      if (iol > bl) { bl = iol; bk = j; }          // first maximum wins (tf.argmax)
    }
    if (bl > best_layer_iol) { best_layer_iol = bl; layer = l; k = bk; }
  }
  float cx = bx * ((float)a.gw[layer] / (float)a.in_w);
  float cy = by * ((float)a.gh[layer] / (float)a.in_h);
  col = (int)cx;                                   // tf.cast(float32 -> int32) truncates
  row = (int)cy;
  tx = cx - (float)col;
  ty = cy - (float)row;
  tw = logf(fmaxf(bw / a.anchors[layer][k][0], 1e-3f));
  th = logf(fmaxf(bh / a.anchors[layer][k][1], 1e-3f));
}

__global__ void t1_claim_kernel(TgtArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.M) return;
  int b = i / a.M, t = i - b * a.M;
  const float* bx = a.boxes + (long long)i * 5;
  int layer, k, row, col;
  float tx, ty, tw, th;
  bool valid;
  t1_select(a, bx[0], bx[1], bx[2], bx[3], layer, k, row, col, tx, ty, tw, th, valid);
  float* info = a.boxinfo + (long long)i * 12;
  ((double*)info)[0] = (double)tx; ((double*)info)[1] = (double)ty; info[4] = tw; info[5] = th;
  info[6] = (float)layer; info[7] = (float)k; info[8] = (float)(int)bx[4]; info[9] = valid ? 1.f : 0.f;
  if (a.assign) {
    int* as = a.assign + (long long)i * 4;
    as[0] = valid ? layer : -1; as[1] = valid ? k : -1; as[2] = valid ? row : -1; as[3] = valid ? col : -1;
  }
  if (!valid) return;
  int gh = a.gh[layer], gw = a.gw[layer];
  for (int cand = 0; cand < 9; ++cand) {
    int ki = cand / 3 - 1, kj = cand % 3 - 1;
    int r = row + ki, q = col + kj;
    if (r < 0 || r >= gh || q < 0 || q >= gw) continue;
    int key = (t * 9 + cand) + 1;                  // order within the image; larger = later update
    atomicMax(a.owner + a.cell_off[layer] + ((long long)b * gh + r) * gw + q, key);
  }
}

// T2: one wavefront per image, boxes in order.  owner key as above; `cnt` rule reproduced.
__device__ float round3(float v) { return rintf(v * 1000.0f) / 1000.0f; }   // np.round(x, 3) on float32

__global__ void t2_assign_kernel(TgtArgs a) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;   // 64
  for (int t = 0; t < a.M; ++t) {
    const float* bx = a.boxes + ((long long)b * a.M + t) * 5;
    float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
    float cxp = floorf((x1 + x2) / 2.0f), cyp = floorf((y1 + y2) / 2.0f);   // '// 2' on float32
    float bw = x2 - x1, bh = y2 - y1;
    float* info = a.boxinfo + ((long long)b * a.M + t) * 12;
    bool valid = !((bw * bh) <= 0.0f);
    int layer = 0, k = 0;
    if (valid) {
      float best = -1.f;
      for (int l = 0; l < a.L; ++l)
        for (int j = 0; j < a.A; ++j) {
          float aw = a.anchors[l][j][0], ah = a.anchors[l][j][1];
          float iol = round3((fminf(bw, aw) * fminf(bh, ah)) / fmaxf(bw * bh, aw * ah));
          if (iol > best) { best = iol; layer = l; k = j; }   // stable argsort(-iol)[0]
        }
    }
    int gh = a.gh[layer], gw = a.gw[layer];
    // numpy 2: float32 * (int / np.int32 -> float64) is computed in double
    double cx = (double)cxp * ((double)gh / (double)a.in_h);
    double cy = (double)cyp * ((double)gw / (double)a.in_w);
    int ci = (int)cx, cj = (int)cy;   // ci = column, cj = row
    if (lane == 0) {
      ((double*)info)[0] = cx - ci; ((double*)info)[1] = cy - cj;   // kept in double: one rounding, as numpy
      float rw = bw / a.anchors[layer][k][0], rh = bh / a.anchors[layer][k][1];
      info[4] = logf(rw >= 1e-3f ? rw : 1e-3f);
      info[5] = logf(rh >= 1e-3f ? rh : 1e-3f);
      info[6] = (float)layer; info[7] = (float)k; info[8] = (float)(int)bx[4]; info[9] = valid ? 1.f : 0.f;
      if (a.assign) {
        int* as = a.assign + ((long long)b * a.M + t) * 4;
        as[0] = valid ? layer : -1; as[1] = valid ? k : -1; as[2] = valid ? cj : -1; as[3] = valid ? ci : -1;
      }
      if (valid) {
        int count = 0;
        for (int ki = -1; ki <= 1; ++ki) {
          int kii = ci + ki;
          for (int kj = -1; kj <= 1; ++kj) {
            int kjj = cj + kj;
            if (kii < 0 || kii >= gh) continue;    // reference compares the column with grid_shapes[0]
            if (kjj < 0 || kjj >= gw) continue;
            int* own = a.owner + a.cell_off[layer] + ((long long)b * gh + kjj) * gw + kii;
            if (*own != 0 && count >= 3) continue; // occupied (obj == 1) and this box already wrote >= 3
            *own = (t * 9 + (ki + 1) * 3 + (kj + 1)) + 1;
            ++count;
          }
        }
      }
    }
    __builtin_amdgcn_s_barrier();
  }
}

// Dense write: one thread = VEC consecutive channels of one cell (VEC = 4 when F % 4 == 0, else 1).
template <int VEC>
__global__ __launch_bounds__(256) void write_dense_kernel(TgtArgs a, int layer) {
  const int gh = a.gh[layer], gw = a.gw[layer];
  const int F = 5 + a.A + a.C, FV = F / VEC;
  const long long ncell = (long long)a.B * gh * gw;
  const long long nvec = ncell * FV;
  float* y = a.y[layer];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    long long cell = i / FV;
    int c0 = (int)(i - cell * FV) * VEC;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    int key = a.owner[a.cell_off[layer] + cell];
    if (key != 0) {
      int b = (int)(cell / ((long long)gh * gw));
      int t = (key - 1) / 9, cand = (key - 1) - t * 9;
      const float* info = a.boxinfo + ((long long)b * a.M + t) * 12;
      const double txd = ((const double*)info)[0], tyd = ((const double*)info)[1];
      int k = (int)info[7], cls = (int)info[8];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        int c = c0 + j;
        float val = 0.f;
        if (a.mode == 0) {
          int ki = cand / 3 - 1, kj = cand % 3 - 1;           // ki = row offset, kj = column offset
          if (c == 0) val = (float)(-kj) + (float)tyd;        // x <- -kj + ty   (reference :3337)
          else if (c == 1) val = (float)(-ki) + (float)txd;   // y <- -ki + tx   (reference :3338)
          else if (c == 2) val = info[4];
          else if (c == 3) val = info[5];
        } else {
          int ki = cand / 3 - 1, kj = cand % 3 - 1;           // ki = column offset, kj = row offset
          if (c == 0) val = (float)((double)(-ki) + txd);
          else if (c == 1) val = (float)((double)(-kj) + tyd);
          else if (c == 2) val = info[4];
          else if (c == 3) val = info[5];
        }
        if (c == 4) val = 1.f;
        else if (c >= 5 && c < 5 + a.A) val = (c - 5 == k) ? 1.f : 0.f;
        else if (c >= 5 + a.A) val = (c - 5 - a.A == cls && cls >= 0 && cls < a.C) ? 1.f : 0.f;
        o[j] = val;
      }
    }
    if (VEC == 4) *(float4*)(y + cell * F + c0) = make_float4(o[0], o[1], o[2], o[3]);
    else y[cell * F + c0] = o[0];
  }
}
#pragma clang fp contract(fast)

}  // namespace

static size_t tgt_ws(int B, int M, int L, const int32_t* ghw, long long* offs) {
  long long cells = 0;
  for (int l = 0; l < L; ++l) {
    if (offs) offs[l] = cells;
    cells += (long long)B * ghw[2 * l] * ghw[2 * l + 1];
  }
  if (offs) offs[L] = cells;
  size_t owner = ((size_t)cells * 4 + 255) & ~(size_t)255;
  return owner + (size_t)B * M * 12 * 4;
}

extern "C" size_t mgd_build_targets_workspace_size(int B, int M, int L, const int32_t* grid_hw_host) {
  if (L < 1 || L > MAXL || !grid_hw_host) return 0;
  return tgt_ws(B, M, L, grid_hw_host, nullptr);
}

extern "C" int mgd_build_targets(const float* boxes, int B, int M, const float* anchors_host, int L, int A, int C,
                                 int in_h, int in_w, const int32_t* grid_hw_host, float* const* y_true_host,
                                 int32_t* assign, int mode, void* ws, size_t ws_bytes, void* stream) {
  MGD_REQUIRE(boxes && anchors_host && grid_hw_host && y_true_host && ws, "build_targets: null pointer");
  MGD_REQUIRE(L >= 1 && L <= MAXL && A >= 1 && A <= MAXA, "build_targets: L=%d A=%d unsupported", L, A);
  MGD_REQUIRE(mode == 0 || mode == 1, "build_targets: mode");
  MGD_REQUIRE(M >= 1 && M * 9 + 1 < (1 << 30), "build_targets: M");
  TgtArgs a;
  a.boxes = boxes; a.B = B; a.M = M; a.L = L; a.A = A; a.C = C; a.in_h = in_h; a.in_w = in_w; a.mode = mode;
  long long offs[MAXL + 1];
  size_t need = tgt_ws(B, M, L, grid_hw_host, offs);
  if (ws_bytes < need) return mgd_set_error(MGD_ENOSPC, "build_targets: workspace %zu < %zu", ws_bytes, need);
  for (int l = 0; l < L; ++l) {
    a.gh[l] = grid_hw_host[2 * l]; a.gw[l] = grid_hw_host[2 * l + 1];
    a.cell_off[l] = offs[l];
    a.y[l] = y_true_host[l];
    MGD_REQUIRE(a.y[l] != nullptr, "build_targets: y_true[%d] null", l);
    for (int j = 0; j < A; ++j) {
      a.anchors[l][j][0] = anchors_host[(l * A + j) * 2];
      a.anchors[l][j][1] = anchors_host[(l * A + j) * 2 + 1];
    }
  }
  a.cell_off[L] = offs[L];
  a.owner = (int*)ws;
  size_t owner_bytes = ((size_t)offs[L] * 4 + 255) & ~(size_t)255;
  a.boxinfo = (float*)((char*)ws + owner_bytes);
  a.assign = assign;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(a.owner, 0, (size_t)offs[L] * 4, st) != hipSuccess)
    return mgd_set_error(MGD_ELAUNCH, "build_targets: memset failed");
  if (mode == 0) hipLaunchKernelGGL(t1_claim_kernel, dim3(cdiv((long)B * M, 128)), dim3(128), 0, st, a);
  else hipLaunchKernelGGL(t2_assign_kernel, dim3(B), dim3(64), 0, st, a);
  for (int l = 0; l < L; ++l) {
    const int F = 5 + A + C, vec = (F % 4 == 0) ? 4 : 1;
    long long nvec = (long long)B * a.gh[l] * a.gw[l] * (F / vec);
    long long g = (nvec + 255) / 256;
    if (g > 2048) g = 2048;
    if (vec == 4) hipLaunchKernelGGL(write_dense_kernel<4>, dim3((int)g), dim3(256), 0, st, a, l);
    else hipLaunchKernelGGL(write_dense_kernel<1>, dim3((int)g), dim3(256), 0, st, a, l);
  }
  MGD_CHECK_LAUNCH("build_targets");
  return MGD_OK;
}
