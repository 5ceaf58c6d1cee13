// 2D task-aligned assignment and the YOLOv8/v10 2D detection loss (BCE + CIoU + DFL) with its gradient, on the NHWC head maps.
// Reference: utils/tal.py:19-264 TaskAlignedAssigner (alpha 0.5, beta 6), utils/loss.py:73-113 BboxLoss, :157-257
// v8DetectionLoss, nn/modules/block.py:44-62 DFL, utils/metrics.py:78-134 CIoU.  Map channel order: [4 x 16 DFL bins | nc cls].
// One thread owns one anchor; fp32 in the reference's operation order; integer outputs reproduce the reference's.
#include "common.h"
#include "tal_common.h"

namespace {

constexpr int RM = 16;  // reg_max

// gt: (B, n, 5) = cls | box xyxy px
__global__ void gt2d_prep_kernel(const float* __restrict__ gt, float* __restrict__ rec, unsigned* __restrict__ pa, unsigned* __restrict__ po, int B,
                                 int n, int nc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  pa[i] = 0u;  // the atomicMax accumulators of resolve_kernel, zeroed by a kernel (not hipMemsetAsync: tal_loss3d.hip, gt_prep_kernel)
  po[i] = 0u;
  const float* g = gt + (long)i * 5;
  float* r = rec + (long)i * GTW;
  r[G_VALID] = (g[1] + g[2] + g[3] + g[4]) > 0.f ? 1.f : 0.f;
  int lab = (int)g[0];
  lab = lab < 0 ? 0 : (lab >= nc ? nc - 1 : lab);
  r[G_LABEL] = (float)lab;
  for (int j = 0; j < 4; ++j) r[G_BOX + j] = g[1 + j];
}

// DFL expectation of one side: softmax over 16 logits . arange(16)   (block.py:59-62, loss.py:197-204)
template <typename T, bool VEC = false>
__device__ __forceinline__ float dfl_expect(const T* z, float* prob) {
  float v[RM], mx = -INFINITY;
  if (VEC) {  // 16-byte aligned rows: 2 (bf16) / 4 (fp32) chunk loads instead of 16 scalar ones
#pragma unroll
    for (int k = 0; k < RM; k += TT<T>::CE) Chunk<T>::unpack(*(const uint4*)(z + k), v + k);
#pragma unroll
    for (int k = 0; k < RM; ++k) mx = fmaxf(mx, v[k]);
  } else {
#pragma unroll
    for (int k = 0; k < RM; ++k) { v[k] = TT<T>::ld(z + k); mx = fmaxf(mx, v[k]); }
  }
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < RM; ++k) { v[k] = expf(v[k] - mx); se += v[k]; }
  float e = 0.f;
#pragma unroll
  for (int k = 0; k < RM; ++k) {
    float pk = v[k] / se;
    if (prob) prob[k] = pk;
    e += pk * (float)k;
  }
  return e;
}

template <typename T>
__global__ __launch_bounds__(256) void metric2d_kernel(Levels L, const float* __restrict__ rec, float* __restrict__ align,
                                                       float* __restrict__ ovl, int n, float alpha, float beta,
                                                       const int* __restrict__ n_used) {
  extern __shared__ float sg[];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < n * GTW; i += blockDim.x) sg[i] = rec[(long)b * n * GTW + i];
  __syncthreads();
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= L.A) return;
  float ax, ay, st;
  int lvl;
  const T* p = anchor_ptr<T>(L, b, a, ax, ay, st, lvl);
  float dl = dfl_expect<T>(p, nullptr), dtp = dfl_expect<T>(p + RM, nullptr), dr = dfl_expect<T>(p + 2 * RM, nullptr),
        db = dfl_expect<T>(p + 3 * RM, nullptr);
  float bx1 = (ax - dl) * st, by1 = (ay - dtp) * st, bx2 = (ax + dr) * st, by2 = (ay + db) * st;
  float apx = ax * st, apy = ay * st;
  const int ne = rows_used(n_used, n);
  for (int g = 0; g < ne; ++g) {
    const float* r = sg + g * GTW;
    float al = 0.f, ov = 0.f;
    if (r[G_VALID] != 0.f) {
      float d0 = apx - r[G_BOX], d1 = apy - r[G_BOX + 1], d2 = r[G_BOX + 2] - apx, d3 = r[G_BOX + 3] - apy;
      if (fminf(fminf(d0, d1), fminf(d2, d3)) > 1e-9f) {
        float s = sigmoid_f(TT<T>::ld(p + 4 * RM + (int)r[G_LABEL]));
        ov = fmaxf(ciou_f(r + G_BOX, bx1, by1, bx2, by2), 0.f);
        float sa = alpha == 0.5f ? sqrtf(s) : (alpha == 1.f ? s : powf(s, alpha));
        float ob = beta == 1.f ? ov : powf(ov, beta);
        al = sa * ob;
      }
    }
    align[((long)b * n + g) * L.A + a] = al;
    ovl[((long)b * n + g) * L.A + a] = ov;
  }
}

// CIoU(box1 = prediction p, box2 = target t) and its gradient wrt the four prediction coordinates (alpha is a constant,
// metrics.py:127-129; the clamp / max / min sub-gradients follow the strict inequalities)
__device__ __forceinline__ float ciou_grad(const float* pb, const float* tb, float* gp) {
  const float eps = 1e-7f;
  float x11 = pb[0], y11 = pb[1], x12 = pb[2], y12 = pb[3];
  float x21 = tb[0], y21 = tb[1], x22 = tb[2], y22 = tb[3];
  float w1 = x12 - x11, h1 = y12 - y11 + eps, w2 = x22 - x21, h2 = y22 - y21 + eps;
  float iw = fminf(x12, x22) - fmaxf(x11, x21), ih = fminf(y12, y22) - fmaxf(y11, y21);
  float iwc = fmaxf(iw, 0.f), ihc = fmaxf(ih, 0.f);
  float inter = iwc * ihc;
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cw = fmaxf(x12, x22) - fminf(x11, x21), ch = fmaxf(y12, y22) - fminf(y11, y21);
  float c2 = cw * cw + ch * ch + eps;
  float sx = x21 + x22 - x11 - x12, sy = y21 + y22 - y11 - y12;
  float rho2 = (sx * sx + sy * sy) / 4.f;
  const float k4 = (float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846));
  float da = atanf(w2 / h2) - atanf(w1 / h1);
  float v = k4 * (da * da);
  float alpha = v / (v - iou + (1.f + eps));
  // d(iw)/d(x11, x12), d(ih)/d(y11, y12)
  float diw_x11 = (iw > 0.f && x11 > x21) ? -1.f : 0.f, diw_x12 = (iw > 0.f && x12 < x22) ? 1.f : 0.f;
  float dih_y11 = (ih > 0.f && y11 > y21) ? -1.f : 0.f, dih_y12 = (ih > 0.f && y12 < y22) ? 1.f : 0.f;
  float dinter[4] = {diw_x11 * ihc, dih_y11 * iwc, diw_x12 * ihc, dih_y12 * iwc};
  float dw1[4] = {-1.f, 0.f, 1.f, 0.f}, dh1[4] = {0.f, -1.f, 0.f, 1.f};
  float dcw[4] = {x11 < x21 ? -1.f : 0.f, 0.f, x12 > x22 ? 1.f : 0.f, 0.f};
  float dch[4] = {0.f, y11 < y21 ? -1.f : 0.f, 0.f, y12 > y22 ? 1.f : 0.f};
  float drho[4] = {-sx / 2.f, -sy / 2.f, -sx / 2.f, -sy / 2.f};
  float den = h1 * h1 + w1 * w1;
  float dat_w = h1 / den, dat_h = -w1 / den;  // d atan(w1/h1) / d w1, d h1
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float duni = dw1[j] * h1 + w1 * dh1[j] - dinter[j];
    float diou = (dinter[j] * uni - inter * duni) / (uni * uni);
    float dc2 = 2.f * cw * dcw[j] + 2.f * ch * dch[j];
    float dpen = (drho[j] * c2 - rho2 * dc2) / (c2 * c2);
    float dv = k4 * 2.f * da * (-(dat_w * dw1[j] + dat_h * dh1[j]));
    gp[j] = diou - dpen - alpha * dv;
  }
  return iou - (rho2 / c2 + v * alpha);
}

struct Loss2W { float box, cls, dfl; };

// items: [0] box (CIoU), [1] cls (BCE), [2] dfl — already divided by target_scores_sum and multiplied by the gains
// VEC: every map / gradient row is 16-byte aligned and nc is a chunk multiple (the launcher checks): rows move as 16-byte chunks.  One
// thread owns one anchor's 4*RM + nc channels; with scalar 2-byte accesses a wave touched 64 cache lines per instruction and the
// 33 600-anchor hi-res maps ran at 0.16 TB/s.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void loss2d_kernel(Levels L, const unsigned char* __restrict__ fg, const int* __restrict__ gt_idx,
                                                     const float* __restrict__ tscores, const float* __restrict__ gt,
                                                     const float* __restrict__ scal, Loss2W w, float gscale, float* __restrict__ part, int n) {
  __shared__ float sh[3][256];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float l[3] = {0.f, 0.f, 0.f};
  if (i < (long)L.B * L.A) {
    const int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    float ax, ay, st;
    int lvl;
    const T* p = anchor_ptr<T>(L, b, a, ax, ay, st, lvl);
    int r = a - L.a0[lvl];
    int hy = r / L.W[lvl], hx = r - hy * L.W[lvl];
    T* gp = (T*)L.grad[lvl] + (((long)b * L.H[lvl] + hy) * L.W[lvl] + hx) * L.gsw[lvl];
    const int nc = L.nc;
    const float tss = scal[0];
    float wsum = 0.f;
    if (VEC) {
      constexpr int CE = TT<T>::CE;
      for (int c = 0; c < nc; c += CE) {
        float xv[CE], gv[CE];
        Chunk<T>::unpack(*(const uint4*)(p + 4 * RM + c), xv);
#pragma unroll
        for (int j = 0; j < CE; ++j) {
          const float x = xv[j], t = tscores[i * nc + c + j];
          l[1] += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
          gv[j] = (sigmoid_f(x) - t) / tss * w.cls * gscale;
          wsum += t;
        }
        *(uint4*)(gp + 4 * RM + c) = Chunk<T>::pack(gv);
      }
    } else
    for (int c = 0; c < nc; ++c) {
      float x = TT<T>::ld(p + 4 * RM + c), t = tscores[i * nc + c];
      l[1] += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      TT<T>::st(gp + 4 * RM + c, (sigmoid_f(x) - t) / tss * w.cls * gscale);
      wsum += t;
    }
    l[1] = l[1] / tss * w.cls;
    if (fg[i]) {
      const float* g = gt + ((long)b * n + gt_idx[i]) * 5;
      float prob[4][RM], e[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) e[s] = dfl_expect<T, VEC>(p + s * RM, prob[s]);
      float pb[4] = {ax - e[0], ay - e[1], ax + e[2], ay + e[3]};            // grid units
      float tb[4] = {g[1] / st, g[2] / st, g[3] / st, g[4] / st};
      float gc[4];
      float ci = ciou_grad(pb, tb, gc);
      l[0] = (1.f - ci) * wsum / tss * w.box;
      // d loss / d side: box = (ax - l, ay - t, ax + r, ay + b)
      float kbox = -wsum / tss * w.box;
      float dside[4] = {-gc[0] * kbox, -gc[1] * kbox, gc[2] * kbox, gc[3] * kbox};
      // DFL: target ltrb clamped to [0, reg_max - 1 - 0.01]  (tal.py:328-331, loss.py:99-113)
      float tl[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
      float dsum = 0.f;
      const float kd = wsum / tss * w.dfl / 4.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float t = fminf(fmaxf(tl[s], 0.f), (float)(RM - 1) - 0.01f);
        int il = (int)t, ir = il + 1;
        float wl = (float)ir - t, wr = 1.f - wl;
        float lpl = logf(prob[s][0]), lpr = lpl;  // placeholders, set below
#pragma unroll
        for (int k = 0; k < RM; ++k) {
          if (k == il) lpl = logf(prob[s][k]);
          if (k == ir) lpr = logf(prob[s][k]);
        }
        dsum += -(lpl * wl + lpr * wr);
        float gk[RM];
#pragma unroll
        for (int k = 0; k < RM; ++k) {
          float gd = prob[s][k] - (k == il ? wl : 0.f) - (k == ir ? wr : 0.f);      // d CE-mix / d logit
          float gb = dside[s] * prob[s][k] * ((float)k - e[s]);                     // d box term / d logit through the expectation
          gk[k] = (gd * kd + gb) * gscale;
        }
        if (VEC) {
#pragma unroll
          for (int k = 0; k < RM; k += TT<T>::CE) *(uint4*)(gp + s * RM + k) = Chunk<T>::pack(gk + k);
        } else {
#pragma unroll
          for (int k = 0; k < RM; ++k) TT<T>::st(gp + s * RM + k, gk[k]);
        }
      }
      l[2] = dsum / 4.f * wsum / tss * w.dfl;
    } else if (VEC) {
#pragma unroll
      for (int c = 0; c < 4 * RM; c += TT<T>::CE) *(uint4*)(gp + c) = make_uint4(0, 0, 0, 0);
    } else {
      for (int c = 0; c < 4 * RM; ++c) TT<T>::st(gp + c, 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) sh[j][threadIdx.x] = l[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int j = 0; j < 3; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) part[(long)blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss2d_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ items) {
  // 3 loss items x nblk block partials: 240 threads = 80 row lanes x 3 items, fp64, folded through LDS in a fixed order
  __shared__ double sh[80][3];
  const int t = threadIdx.x;
  const int j = t % 3, r = t / 3;
  if (r < 80) {
    double a = 0.0;
    for (int i = r; i < nblk; i += 80) a += part[(long)i * 3 + j];
    sh[r][j] = a;
  }
  __syncthreads();
  if (t < 3) {
    double a = 0.0;
#pragma unroll 8
    for (int k = 0; k < 80; ++k) a += sh[k][t];
    items[t] = (float)a;
  }
}

int fill2d(Levels& L, int dtype, int nl, const void* const* maps, const int64_t* psw, void* const* grads, const int64_t* gsw, const int* H,
           const int* W, const float* strides, int B, int nc) {
  Y3D_CHECK(nl >= 1 && nl <= MAXL, "tal2d: 1..%d levels", MAXL);
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "tal2d: bad dtype");
  int a0 = 0;
  for (int i = 0; i < MAXL; ++i) {
    bool v = i < nl;
    L.map[i] = v ? maps[i] : nullptr; L.psw[i] = v ? psw[i] : 0;
    L.grad[i] = (v && grads) ? grads[i] : nullptr; L.gsw[i] = (v && gsw) ? gsw[i] : 0;
    L.H[i] = v ? H[i] : 1; L.W[i] = v ? W[i] : 1; L.a0[i] = a0; L.stride[i] = v ? strides[i] : 1.f;
    if (v) a0 += H[i] * W[i];
  }
  L.nl = nl; L.A = a0; L.B = B; L.nc = nc; L.no = nc + 4 * RM;
  return Y3D_OK;
}

}  // namespace

extern "C" {

int y3d_tal2d_assign(int dtype, int nl, const void* const* maps, const int64_t* psw, const int* H, const int* W, const float* strides,
                     int B, int nc, const float* gt, int n, int topk, float alpha, float beta, float* scratch, uint8_t* fg_mask,
                     int* target_gt_idx, float* target_scores, float* scal, const int* n_used, void* stream) {
  Levels L;
  if (fill2d(L, dtype, nl, maps, psw, nullptr, nullptr, H, W, strides, B, nc)) return Y3D_ERR_INVALID;
  Y3D_CHECK(n >= 1 && n <= 64, "tal2d_assign: 1..64 ground-truth boxes per image (got %d)", n);
  Y3D_CHECK(topk >= 1 && topk <= 16, "tal2d_assign: topk in 1..16");
  hipStream_t st = (hipStream_t)stream;
  const int A = L.A;
  float* rec = scratch;
  float* align = rec + (long)B * n * GTW;
  float* ovl = align + (long)B * n * A;
  int* cand = (int*)(ovl + (long)B * n * A);
  unsigned* pa = (unsigned*)(cand + (long)B * n * topk);
  unsigned* po = pa + (long)B * n;
  float* part = (float*)(po + (long)B * n);
  hipLaunchKernelGGL(gt2d_prep_kernel, dim3(cdiv((long)B * n, 64)), dim3(64), 0, st, gt, rec, pa, po, B, n, nc);
  dim3 gm(cdiv(A, 256), B);
  size_t sm = (size_t)n * GTW * sizeof(float);
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(metric2d_kernel<bf16_t>, gm, dim3(256), sm, st, L, rec, align, ovl, n, alpha, beta, n_used);
  else hipLaunchKernelGGL(metric2d_kernel<float>, gm, dim3(256), sm, st, L, rec, align, ovl, n, alpha, beta, n_used);
  {
    const bool fits = (size_t)A * sizeof(float) <= 96 * 1024;  // the metric row in LDS (33.6 KB at 640x640; 1280x1280: 131 KB, global passes)
    hipLaunchKernelGGL(topk_kernel, dim3(B * n), dim3(256), fits ? (size_t)A * sizeof(float) : 0, st, align, rec, cand, L, n, topk, n_used, fits ? 1 : 0, 1);
  }
  int nblk = cdiv((long)B * A, 256);
  hipLaunchKernelGGL(resolve_kernel, dim3(nblk), dim3(256), 0, st, cand, align, ovl, fg_mask, target_gt_idx, pa, po, B, n, A, topk, n_used);
  hipLaunchKernelGGL(scores_kernel, dim3(nblk), dim3(256), 0, st, fg_mask, target_gt_idx, align, rec, pa, po, target_scores, part, B, n, A, nc, 1e-9f);
  hipLaunchKernelGGL(scal_kernel, dim3(1), dim3(64), 0, st, part, nblk, scal);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_loss2d(int dtype, int nl, const void* const* maps, const int64_t* psw, void* const* grads, const int64_t* gsw, const int* H,
               const int* W, const float* strides, int B, int nc, const float* gt, int n, const uint8_t* fg_mask,
               const int* target_gt_idx, const float* target_scores, const float* scal, float w_box, float w_cls, float w_dfl,
               float grad_scale, float* partials, float* items, void* stream) {
  Levels L;
  if (fill2d(L, dtype, nl, maps, psw, grads, gsw, H, W, strides, B, nc)) return Y3D_ERR_INVALID;
  Loss2W w{w_box, w_cls, w_dfl};
  hipStream_t st = (hipStream_t)stream;
  int nblk = cdiv((long)B * L.A, 256);
  const int ce = dtype == Y3D_BF16 ? 8 : 4;
  bool vec = nc % ce == 0;
  for (int i = 0; i < nl; ++i)
    vec = vec && psw[i] % ce == 0 && gsw[i] % ce == 0 && ((uintptr_t)maps[i] & 15) == 0 && ((uintptr_t)grads[i] & 15) == 0;
#define Y3D_L2D(T, V) hipLaunchKernelGGL((loss2d_kernel<T, V>), dim3(nblk), dim3(256), 0, st, L, fg_mask, target_gt_idx, target_scores, gt, scal, w, grad_scale, partials, n)
  if (dtype == Y3D_BF16) { if (vec) Y3D_L2D(bf16_t, true); else Y3D_L2D(bf16_t, false); }
  else { if (vec) Y3D_L2D(float, true); else Y3D_L2D(float, false); }
#undef Y3D_L2D
  hipLaunchKernelGGL(loss2d_final_kernel, dim3(1), dim3(256), 0, st, partials, nblk, items);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
