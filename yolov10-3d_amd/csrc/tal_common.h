// Shared pieces of the task-aligned assigners (2D: utils/tal.py:19-264, 3D: utils/tal.py:355-753): level walking, CIoU,
// the per-GT top-k, conflict resolution and target-score normalisation kernels.  Included by tal_loss3d.hip and tal_loss2d.hip.
#pragma once
#include "common.h"

namespace {

constexpr int MAXL = 4;
constexpr int GTW = 40;   // floats per GT record: valid, label, box(4), c2(2), s2(2), c3(2), s3(3), depth, hbin, hres, kps(24) = 42 -> see layout
// record layout
constexpr int G_VALID = 0, G_LABEL = 1, G_BOX = 2, G_KPS = 6;  // 6..29 keypoints; raw 17-vector is read from the padded gt tensor

struct Levels {
  const void* map[MAXL];   // (B, H, W, >=no) NHWC maps, channel 0 of this head set at the pointer
  void* grad[MAXL];        // same geometry, pixel stride gsw
  long psw[MAXL];          // pixel stride of map (elements)
  long gsw[MAXL];
  int H[MAXL], W[MAXL], a0[MAXL];  // a0: first anchor index of the level
  float stride[MAXL];
  int nl, A, B, nc, no;
};

template <typename T>
__device__ __forceinline__ const T* anchor_ptr(const Levels& L, int b, int a, float& ax, float& ay, float& st, int& lvl) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < MAXL; ++i) if (i < L.nl && a >= L.a0[i]) l = i;
  int r = a - L.a0[l];
  int hy = r / L.W[l], hx = r - hy * L.W[l];
  ax = hx + 0.5f;
  ay = hy + 0.5f;
  st = L.stride[l];
  lvl = l;
  return (const T*)L.map[l] + (((long)b * L.H[l] + hy) * L.W[l] + hx) * L.psw[l];
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// GT rows in use: the padded target tensor has `n` rows per image (its capacity); `n_used` (device, may be null) holds the largest
// per-image box count, written by y3d_pad_targets — the host never learns it, so the step has no host sync (the reference's
// `counts.max()` sizes the tensor on the host, loss.py:801-803).  Rows in [n_used, n) are all-zero padding in every image.
__device__ __forceinline__ int rows_used(const int* __restrict__ n_used, int n) {
  if (!n_used) return n;
  int v = *n_used;
  return v < n ? v : n;
}

// utils/metrics.py:78-134 bbox_iou(xywh=False, CIoU=True), box1 = gt, box2 = prediction
__device__ __forceinline__ float ciou_f(const float* g, float x21, float y21, float x22, float y22) {
  const float eps = 1e-7f;
  float x11 = g[0], y11 = g[1], x12 = g[2], y12 = g[3];
  float w1 = x12 - x11, h1 = y12 - y11 + eps;
  float w2 = x22 - x21, h2 = y22 - y21 + eps;
  float iw = fminf(x12, x22) - fmaxf(x11, x21);
  float ih = fminf(y12, y22) - fmaxf(y11, y21);
  float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cw = fmaxf(x12, x22) - fminf(x11, x21);
  float ch = fmaxf(y12, y22) - fminf(y11, y21);
  float c2 = cw * cw + ch * ch + eps;
  float dx = x21 + x22 - x11 - x12, dy = y21 + y22 - y11 - y12;
  float rho2 = (dx * dx + dy * dy) / 4.f;
  float da = atanf(w2 / h2) - atanf(w1 / h1);
  float v = (float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846)) * (da * da);
  float alpha = v / (v - iou + (1.f + eps));
  return iou - (rho2 / c2 + v * alpha);
}

// one block per (b, g): k passes of arg-max with (value desc, index asc) order; cand[b][g][j] = anchor index or -1
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ align, const float* __restrict__ rec, int* __restrict__ cand,
                                                   Levels L, int n, int k, const int* __restrict__ n_used, int lds_row, int constrain) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ int chosen[16];
  // block id -> (row g, image b) with the ROW as the slow index: the rows in use (g < n_used, known on the device only) are the first
  // B * n_used blocks of the grid, so the dispatcher spreads them over the whole chip; with the image as the slow index the few
  // working blocks of every image sat next to 64 - n_used empty ones and piled up on a fraction of the CUs (2x slower)
  const int B_ = gridDim.x / n;
  const int g_ = blockIdx.x / B_, b = blockIdx.x - g_ * B_;
  const int bg = b * n + g_;
  const float* r = rec + (long)bg * GTW;
  int* out = cand + (long)bg * k;
  if (g_ >= rows_used(n_used, n)) return;  // never read: resolve_kernel stops at the same bound
  if (r[G_VALID] == 0.f) {
    if (threadIdx.x < k) out[threadIdx.x] = -1;
    return;
  }
  const float* row = align + (long)bg * L.A;
  // the row is scanned k times: keep it in LDS when it fits (8400 anchors: 33 KB) - one pass over global memory instead of k
  extern __shared__ float srow[];
  const bool in_lds = lds_row != 0;
  if (in_lds) {
    for (int a = threadIdx.x; a < L.A; a += 256) srow[a] = row[a];
    __syncthreads();
  }
  for (int j = 0; j < k; ++j) {
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int a = threadIdx.x; a < L.A; a += 256) {
      float v;
      if (in_lds) {
        v = srow[a];  // chosen anchors were overwritten with -2 (metrics are >= 0)
      } else {
        bool skip = false;
        for (int t = 0; t < j; ++t) skip |= (chosen[t] == a);
        if (skip) continue;
        v = row[a];
      }
      if (v > bv || (v == bv && a < bi)) { bv = v; bi = a; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        float v2 = sv[threadIdx.x + s];
        int i2 = si[threadIdx.x + s];
        if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      chosen[j] = si[0];
      if (in_lds && si[0] < L.A) srow[si[0]] = -2.f;
    }
    __syncthreads();
  }
  if (threadIdx.x < k) {
    int a = chosen[threadIdx.x];
    // mask_pos = mask_topk * mask_in_gts * mask_gt (tal.py:492-497): a candidate outside the box is dropped
    float ax, ay, st;
    int lvl;
    (void)b;
    int l = 0;
    for (int i = 1; i < MAXL; ++i) if (i < L.nl && a >= L.a0[i]) l = i;
    int rr = a - L.a0[l];
    int hy = rr / L.W[l], hx = rr - hy * L.W[l];
    ax = (hx + 0.5f) * L.stride[l];
    ay = (hy + 0.5f) * L.stride[l];
    (void)st; (void)lvl;
    float d0 = ax - r[G_BOX], d1 = ay - r[G_BOX + 1], d2 = r[G_BOX + 2] - ax, d3 = r[G_BOX + 3] - ay;
    // (`constrain_anchors: False`, tal.py:492-496: mask_pos = mask_topk * mask_gt - every top-k candidate of a valid box counts)
    out[threadIdx.x] = (!constrain || fminf(fminf(d0, d1), fminf(d2, d3)) > 1e-9f) ? a : -1;
  }
}

// per anchor: how many GTs selected it; the winner; per-GT normalisers by atomicMax on the (non-negative) float bits
__global__ void resolve_kernel(const int* __restrict__ cand, const float* __restrict__ align, const float* __restrict__ sim,
                               unsigned char* __restrict__ fg, int* __restrict__ gt_idx, unsigned* __restrict__ pa, unsigned* __restrict__ po,
                               int B, int n, int A, int k, const int* __restrict__ n_used) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * A) return;
  int b = (int)(i / A), a = (int)(i - (long)b * A);
  int cnt = 0, gsel = 0;
  const int ne = rows_used(n_used, n);  // rows >= ne are padding in every image: they select nothing and their sim is 0
  for (int g = 0; g < ne; ++g) {
    const int* c = cand + ((long)b * n + g) * k;
    bool hit = false;
    for (int j = 0; j < k; ++j) hit |= (c[j] == a);
    if (hit) { if (cnt == 0) gsel = g; ++cnt; }
  }
  if (cnt > 1) {  // tal.py:741-748: arg-max of `overlaps` (= similarities) over ALL gts, first maximum
    float bv = -1.f;
    for (int g = 0; g < ne; ++g) {
      float v = sim[((long)b * n + g) * A + a];
      if (v > bv) { bv = v; gsel = g; }
    }
  }
  fg[i] = cnt > 0;
  gt_idx[i] = cnt > 0 ? gsel : 0;
  if (cnt > 0) {
    float al = align[((long)b * n + gsel) * A + a], sm = sim[((long)b * n + gsel) * A + a];
    atomicMax(pa + b * n + gsel, __float_as_uint(al));
    atomicMax(po + b * n + gsel, __float_as_uint(sm));
  }
}

// target_scores (B, A, nc) and block partials (sum of scores, n_fg)
__global__ __launch_bounds__(256) void scores_kernel(const unsigned char* __restrict__ fg, const int* __restrict__ gt_idx,
                                                     const float* __restrict__ align, const float* __restrict__ rec,
                                                     const unsigned* __restrict__ pa, const unsigned* __restrict__ po,
                                                     float* __restrict__ tscores, float* __restrict__ part, int B, int n, int A, int nc,
                                                     float eps) {
  __shared__ float s0[256], s1[256];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float ts = 0.f, nf = 0.f;
  if (i < (long)B * A) {
    int b = (int)(i / A), a = (int)(i - (long)b * A);
    int lab = -1;
    float norm = 0.f;
    if (fg[i]) {
      int g = gt_idx[i];
      lab = (int)rec[((long)b * n + g) * GTW + G_LABEL];
      float al = align[((long)b * n + g) * A + a];
      norm = al * __uint_as_float(po[b * n + g]) / (__uint_as_float(pa[b * n + g]) + eps);
      ts = norm;
      nf = 1.f;
    }
    for (int c = 0; c < nc; ++c) tscores[i * nc + c] = c == lab ? norm : 0.f;
  }
  s0[threadIdx.x] = ts;
  s1[threadIdx.x] = nf;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { s0[threadIdx.x] += s0[threadIdx.x + s]; s1[threadIdx.x] += s1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = s0[0]; part[blockIdx.x * 2 + 1] = s1[0]; }
}

// scal[0] = max(sum target_scores, 1), scal[1] = n_fg
__global__ __launch_bounds__(64) void scal_kernel(const float* __restrict__ part, int nblk, float* __restrict__ scal) {
  // one wave: 64 lanes stride over the block partials in fp64, folded through LDS in a fixed order (it was one thread: 74 us)
  __shared__ double sh[64][2];
  const int t = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int i = t; i < nblk; i += 64) { a += part[i * 2]; b += part[i * 2 + 1]; }
  sh[t][0] = a;
  sh[t][1] = b;
  __syncthreads();
  if (t == 0) {
    a = 0.0; b = 0.0;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) { a += sh[k][0]; b += sh[k][1]; }
    scal[0] = (float)(a > 1.0 ? a : 1.0);
    scal[1] = (float)b;
  }
}


}  // namespace
