// Task-aligned assignment (3D) and the 3D detection loss with its gradient, fused for the NHWC head maps.
//
// Reference: utils/tal.py:355-753 TaskAlignedAssigner3d (use_2d + use_3d, 'l1' keypoint metric, constrain_anchors),
// utils/keypoint_utils.py:11-118, utils/metrics.py:78-134 (CIoU), utils/loss.py:821-963, 1112-1136 DDDetectionLoss.
// The reference materialises (B, n, A, 8, 3) keypoint differences and ~60 small torch ops; here one thread owns one anchor:
// it decodes its 38 logits straight from the per-level maps, builds its 8 box corners in registers and walks the (<= 64) GT
// records of its image held in LDS.  All arithmetic is fp32 in the reference's operation order (no FMA contraction), so the
// integer outputs (fg_mask, target_gt_idx) reproduce the reference's on identical inputs; ties go to the lowest index.
//
// Stages (all tiny next to the convolutions, HBM/latency-bound):
//   gt_prep   (B*n threads)       GT record: label, box, 8 keypoints
//   metric    (B x A threads)     align[b][g][a], sim[b][g][a]
//   topk      (one block per (b,g)) k passes of a block-wide arg-max (value desc, index asc) -> candidate anchors
//   resolve   (B x A threads)     anchor -> fg flag, gt index (multi-assigned: arg-max of sim over ALL gts), atomicMax of the
//                                 per-GT normalisers (order independent)
//   scores    (B x A threads)     normalised target scores + block partials of sum(target_scores), n_fg
//   loss      (B x A threads)     six loss partial sums and d(loss)/d(map) for all 38 channels
#include "common.h"
#include "tal_common.h"

namespace {

// utils/keypoint_utils.py:11-118 (explicit form of Rx(pi/2) @ Ry(-ry))
__device__ __forceinline__ void keypoints(float cx, float cy, float dep, float sh, float sw, float sl, int bin, float res,
                                          const float* cal, float* k) {
  const float PI = 3.14159265358979323846f;
  float cu = cal[0], cv = cal[1], fu = cal[2], fv = cal[3], tx = cal[4], ty = cal[5];
  float X = (cx - cu) * dep / fu + tx;
  float Y = (cy - cv) * dep / fv + ty;
  float hl = sl / 2.f, hw = sw / 2.f, hh = sh / 2.f;
  float ang = (float)bin * (float)(2.0 * 3.14159265358979323846 / 12.0) + res;
  if (ang > PI) ang = ang - (float)(2.0 * 3.14159265358979323846);
  float ry = ang + atan2f(cx - cu, fu);
  if (ry > PI) ry = ry - (float)(2.0 * 3.14159265358979323846);
  if (ry < -PI) ry = ry + (float)(2.0 * 3.14159265358979323846);
  float a = -ry;
  float ca = cosf(a), sa = sinf(a);
  const float cxr = 6.123233995736766e-17f, sxr = 1.0f;  // cos(pi/2), sin(pi/2) as the reference's Python floats
  const float sgx[8] = {1, 1, -1, -1, 1, 1, -1, -1};
  const float sgy[8] = {1, -1, 1, -1, 1, -1, 1, -1};
  const float sgz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float px = sgx[i] * hl, py = sgy[i] * hw, pz = sgz[i] * hh;
    float ox = ca * px + (sxr * sa) * py - (cxr * sa) * pz;
    float oy = cxr * py + sxr * pz;
    float oz = sa * px - (sxr * ca) * py + (cxr * ca) * pz;
    k[i * 3 + 0] = ox + X;
    k[i * 3 + 1] = oy + Y;
    k[i * 3 + 2] = oz + dep;
  }
}

// gt: (B, n, 17) = cls | box xyxy px | c2(2) | s2(2) | c3(2) | s3(3) | depth | hbin | hres ; rec: (B, n, GTW)
__global__ void gt_prep_kernel(const float* __restrict__ gt, const float* __restrict__ calib, const float* __restrict__ mean_sizes,
                               float* __restrict__ rec, unsigned* __restrict__ pa, unsigned* __restrict__ po, int B, int n, int nc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  // the per-box maxima that resolve_kernel builds with atomicMax start from zero.  Zeroed HERE, by a kernel, not by hipMemsetAsync: inside
  // a captured hipGraph of the whole training step the memset nodes were seen NOT to take effect on later replays (stale maxima of the
  // previous replay's other head set, whose scratch shares the address: tools/probe/graph_debug2.py), while kernel nodes always run
  pa[i] = 0u;
  po[i] = 0u;
  int b = i / n;
  const float* g = gt + (long)i * 17;
  float* r = rec + (long)i * GTW;
  float bs = g[1] + g[2] + g[3] + g[4];
  r[G_VALID] = bs > 0.f ? 1.f : 0.f;
  int lab = (int)g[0];
  lab = lab < 0 ? 0 : (lab >= nc ? nc - 1 : lab);
  r[G_LABEL] = (float)lab;
  for (int j = 0; j < 4; ++j) r[G_BOX + j] = g[1 + j];
  float sh = mean_sizes[lab * 3 + 0] + g[11], sw = mean_sizes[lab * 3 + 1] + g[12], sl = mean_sizes[lab * 3 + 2] + g[13];
  keypoints(g[9], g[10], g[14], sh, sw, sl, (int)g[15], g[16], calib + b * 6, r + G_KPS);
}

template <typename T>
__global__ __launch_bounds__(256) void metric_kernel(Levels L, const float* __restrict__ rec, const float* __restrict__ calib,
                                                     const float* __restrict__ mean_sizes, float* __restrict__ align,
                                                     float* __restrict__ sim, int n, float alpha, float beta, float gamma,
                                                     const int* __restrict__ n_used, int mode) {
  extern __shared__ float sg[];  // [n][GTW]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < n * GTW; i += blockDim.x) sg[i] = rec[(long)b * n * GTW + i];
  __syncthreads();
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= L.A) return;
  float ax, ay, st;
  int lvl;
  const T* p = anchor_ptr<T>(L, b, a, ax, ay, st, lvl);
  const int nc = L.nc;
  float sc[8];
  int amax = 0;
  float best = -1.f;
  for (int c = 0; c < nc && c < 8; ++c) {
    sc[c] = sigmoid_f(TT<T>::ld(p + c));
    if (sc[c] > best) { best = sc[c]; amax = c; }
  }
  float o2x = TT<T>::ld(p + nc + 0), o2y = TT<T>::ld(p + nc + 1), s2x = TT<T>::ld(p + nc + 2), s2y = TT<T>::ld(p + nc + 3);
  float ccx = ax + o2x, ccy = ay + o2y;
  float bx1 = (ccx - s2x / 2.f) * st, by1 = (ccy - s2y / 2.f) * st, bx2 = (ccx + s2x / 2.f) * st, by2 = (ccy + s2y / 2.f) * st;
  float apx = ax * st, apy = ay * st;
  float c3x = apx + TT<T>::ld(p + nc + 4) * st, c3y = apy + TT<T>::ld(p + nc + 5) * st;
  float sh = mean_sizes[amax * 3 + 0] + TT<T>::ld(p + nc + 6), sw = mean_sizes[amax * 3 + 1] + TT<T>::ld(p + nc + 7),
        sl = mean_sizes[amax * 3 + 2] + TT<T>::ld(p + nc + 8);
  int hb = 0;
  float hbv = TT<T>::ld(p + nc + 9);
  for (int j = 1; j < 12; ++j) {
    float v = TT<T>::ld(p + nc + 9 + j);
    if (v > hbv) { hbv = v; hb = j; }
  }
  float hres = TT<T>::ld(p + nc + 21 + hb);
  float dep = TT<T>::ld(p + nc + 33);
  float kp[24];
  keypoints(c3x, c3y, dep, sh, sw, sl, hb, hres, calib + b * 6, kp);
  const int ne = rows_used(n_used, n);
  for (int g = 0; g < ne; ++g) {
    const float* r = sg + g * GTW;
    float al = 0.f, sm = 0.f;
    if (r[G_VALID] != 0.f) {
      // tal.py:709-726: anchor centre strictly inside the box (eps 1e-9); with `constrain_anchors: False` (mode bit 8 clear) the metric
      // is computed for every anchor of a valid box (tal.py:476-483)
      float d0 = apx - r[G_BOX], d1 = apy - r[G_BOX + 1], d2 = r[G_BOX + 2] - apx, d3 = r[G_BOX + 3] - apy;
      if (!(mode & 8) || fminf(fminf(d0, d1), fminf(d2, d3)) > 1e-9f) {
        float s = sc[(int)r[G_LABEL]];
        float ov = fmaxf(ciou_f(r + G_BOX, bx1, by1, bx2, by2), 0.f);
        float dist = 0.f;
        if (mode & 4) {  // kps_dist_metric "l2": 1 / exp(0.5 * sum(d^2) / 24)  (tal.py:468-470)
#pragma unroll
          for (int j = 0; j < 24; ++j) { const float d = kp[j] - r[G_KPS + j]; dist += d * d; }
          dist = 0.5f * (dist / 24.f);
        } else {         // "l1": 1 / exp(sum|d| / 24)  (tal.py:465-467)
#pragma unroll
          for (int j = 0; j < 24; ++j) dist += fabsf(kp[j] - r[G_KPS + j]);
          dist = dist / 24.f;
        }
        const float smk = 1.f / expf(dist);
        float sa = alpha == 0.5f ? sqrtf(s) : (alpha == 1.f ? s : powf(s, alpha));
        float ob = beta == 1.f ? ov : powf(ov, beta);
        float sgm = gamma == 1.f ? smk : powf(smk, gamma);
        // tal.py:473-484: box + keypoint metric ("overlaps" = similarities), keypoint-only, or box-only ("overlaps" = CIoU)
        if ((mode & 3) == 3) { al = sa * ob * sgm; sm = smk; }
        else if (mode & 2) { al = sa * sgm; sm = smk; }
        else { al = sa * ob; sm = ov; }
      }
    }
    align[((long)b * n + g) * L.A + a] = al;
    sim[((long)b * n + g) * L.A + a] = sm;
  }
}

// utils/loss.py:795-810 `preprocess`: ragged per-box rows [batch_idx | cls | xywh in [0,1] | ...] -> (B, cap, width) zero-padded per image
// in order of appearance, boxes scaled to pixels and converted to xyxy; n_used[0] = min(largest per-image box count, cap), n_used[1] =
// the largest count itself (device side: the reference sizes the tensor with a host-side counts.max()).  One block per image.
__global__ void zero_ints_kernel(int* __restrict__ p, int n) {  // (instead of hipMemsetAsync: see gt_prep_kernel)
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void pad_targets_kernel(const float* __restrict__ rows, int nbox, int width, float* __restrict__ out, int cap,
                                                          float sx, float sy, int* __restrict__ n_used) {
  __shared__ int smax[256];
  const int b = blockIdx.x, rw = width + 1;
  float* o = out + (long)b * cap * width;
  for (int i = threadIdx.x; i < cap * width; i += 256) o[i] = 0.f;
  __syncthreads();
  int mx = 0;
  for (int i = threadIdx.x; i < nbox; i += 256) {
    if ((int)rows[(long)i * rw] != b) continue;
    int pos = 0;
    for (int j = 0; j < i; ++j) pos += ((int)rows[(long)j * rw] == b);
    mx = max(mx, pos + 1);
    if (pos >= cap) continue;
    const float* r = rows + (long)i * rw + 1;
    float* d = o + (long)pos * width;
    d[0] = r[0];
    float x = r[1] * sx, y = r[2] * sy, w = r[3] * sx, h = r[4] * sy;
    d[1] = x - w / 2.f;
    d[2] = y - h / 2.f;
    d[3] = x + w / 2.f;
    d[4] = y + h / 2.f;
    for (int c = 5; c < width; ++c) d[c] = r[c];
  }
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && smax[0] > 0) {
    atomicMax(n_used, min(smax[0], cap));  // rows the assigner walks
    atomicMax(n_used + 1, smax[0]);        // the true largest count: > cap means boxes were dropped (the host raises on it)
  }
}

struct LossW { float loss2d, cls, depth, offset3d, size3d, heading; };

// per anchor: six loss terms (already divided by the normalisers) and the gradient of their sum times `gscale` wrt the 38 logits
// NC = number of classes as a template parameter: the gradient row gr[] is indexed by nc + j, and with a run-time nc the array
// lived in scratch (176 B per lane)
template <typename T, int NC>
__global__ __launch_bounds__(256) void loss_kernel(Levels L, const unsigned char* __restrict__ fg, const int* __restrict__ gt_idx,
                                                   const float* __restrict__ tscores, const float* __restrict__ gt,
                                                   const float* __restrict__ scal, LossW w, float gscale, float* __restrict__ part,
                                                   int n) {
  __shared__ float sh[6][256];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float l[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < (long)L.B * L.A) {
    const int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    float ax, ay, st;
    int lvl;
    const T* p = anchor_ptr<T>(L, b, a, ax, ay, st, lvl);
    int r = a - L.a0[lvl];
    int hy = r / L.W[lvl], hx = r - hy * L.W[lvl];
    T* gp = (T*)L.grad[lvl] + (((long)b * L.H[lvl] + hy) * L.W[lvl] + hx) * L.gsw[lvl];
    constexpr int nc = NC;
    const float tss = scal[0], nfg = scal[1];
    float gr[40];
#pragma unroll
    for (int c = 0; c < 40; ++c) gr[c] = 0.f;
    // classification: BCE with logits against the normalised target scores, dense (loss.py:888)
#pragma unroll
    for (int c = 0; c < nc; ++c) {
      float x = TT<T>::ld(p + c), t = tscores[i * nc + c];
      float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      l[1] += bce;
      gr[c] = (sigmoid_f(x) - t) / tss * w.cls;
    }
    l[1] = l[1] / tss * w.cls;
    if (fg[i]) {
      const float* g = gt + ((long)b * n + gt_idx[i]) * 17;
      const float apx = ax * st, apy = ay * st;
      const float inv2 = 1.f / (nfg * 2.f);
      // 2D box: mean L1 of offset and of size over the fg elements, / tss (loss.py:913-926)
      float acc2 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float po = TT<T>::ld(p + nc + j) * st, to = g[5 + j] - (j == 0 ? apx : apy);
        float ps = TT<T>::ld(p + nc + 2 + j) * st, tsz = g[7 + j];
        acc2 += fabsf(po - to) + fabsf(ps - tsz);
        gr[nc + j] = (po > to ? 1.f : (po < to ? -1.f : 0.f)) * st * inv2 / tss * w.loss2d;
        gr[nc + 2 + j] = (ps > tsz ? 1.f : (ps < tsz ? -1.f : 0.f)) * st * inv2 / tss * w.loss2d;
      }
      l[0] = acc2 * inv2 / tss * w.loss2d;
      // depth with Laplacian aleatoric uncertainty (loss.py:1112-1119)
      float d = TT<T>::ld(p + nc + 33), u = TT<T>::ld(p + nc + 34), td = g[14];
      float e = 1.4142f * expf(-0.5f * u), ad = fabsf(d - td);
      l[2] = (e * ad + 0.5f * u) / tss * w.depth;
      gr[nc + 33] = e * (d > td ? 1.f : (d < td ? -1.f : 0.f)) / tss * w.depth;
      gr[nc + 34] = (-0.5f * e * ad + 0.5f) / tss * w.depth;
      // 3D offset: mean L1 over fg elements (loss.py:936-940)
      float acc3 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float po = TT<T>::ld(p + nc + 4 + j) * st, to = g[9 + j] - (j == 0 ? apx : apy);
        acc3 += fabsf(po - to);
        gr[nc + 4 + j] = (po > to ? 1.f : (po < to ? -1.f : 0.f)) * st * inv2 / tss * w.offset3d;
      }
      l[3] = acc3 * inv2 / tss * w.offset3d;
      // 3D size: summed L1 (loss.py:942-946)
      float acc4 = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float ps = TT<T>::ld(p + nc + 6 + j), tsz = g[11 + j];
        acc4 += fabsf(ps - tsz);
        gr[nc + 6 + j] = (ps > tsz ? 1.f : (ps < tsz ? -1.f : 0.f)) / tss * w.size3d;
      }
      l[4] = acc4 / tss * w.size3d;
      // heading: cross entropy over 12 bins + L1 on the residual of the target bin (loss.py:1122-1136)
      int tb = (int)g[15];
      float hv[12], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 12; ++j) { hv[j] = TT<T>::ld(p + nc + 9 + j); mx = fmaxf(mx, hv[j]); }
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 12; ++j) se += expf(hv[j] - mx);
      float lse = mx + logf(se);
      float pr = TT<T>::ld(p + nc + 21 + tb), tr = g[16];
      float hvt = hv[0];
#pragma unroll
      for (int j = 1; j < 12; ++j) hvt = j == tb ? hv[j] : hvt;
      l[5] = (lse - hvt + fabsf(pr - tr)) / tss * w.heading;
#pragma unroll
      for (int j = 0; j < 12; ++j) gr[nc + 9 + j] = (expf(hv[j] - lse) - (j == tb ? 1.f : 0.f)) / tss * w.heading;
      const float gres = (pr > tr ? 1.f : (pr < tr ? -1.f : 0.f)) / tss * w.heading;
#pragma unroll
      for (int j = 0; j < 12; ++j) gr[nc + 21 + j] = j == tb ? gres : 0.f;  // static indices: a run-time gr[.. + tb] puts gr[] in scratch
    }
#pragma unroll
    for (int c = 0; c < NC + 35; ++c) TT<T>::st(gp + c, gr[c] * gscale);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) sh[j][threadIdx.x] = l[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int j = 0; j < 6; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[(long)blockIdx.x * 6 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ items) {
  // 6 loss items x nblk block partials: 240 threads = 40 row lanes x 6 items, fp64, folded through LDS in a fixed order
  __shared__ double sh[40][6];
  const int t = threadIdx.x;
  const int j = t % 6, r = t / 6;
  if (r < 40) {
    double a = 0.0;
    for (int i = r; i < nblk; i += 40) a += part[(long)i * 6 + j];
    sh[r][j] = a;
  }
  __syncthreads();
  if (t < 6) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 40; ++k) a += sh[k][t];
    items[t] = (float)a;
  }
}

}  // namespace

extern "C" {

// floats of scratch the assigner needs: records + align + sim + candidates + normalisers + partials
int y3d_tal3d_scratch_floats(int B, int n, int A, int topk) {
  long blocks = ((long)B * A + 255) / 256;
  long v = (long)B * n * GTW + 2L * B * n * A + (long)B * n * topk + 2L * B * n + 2 * blocks + 8;
  return v < (1L << 31) ? (int)v : -1;
}

static int fill_levels(Levels& L, int dtype, int nl, const void* const* maps, const int64_t* psw, void* const* grads, const int64_t* gsw,
                       const int* H, const int* W, const float* strides, int B, int nc, int no) {
  Y3D_CHECK(nl >= 1 && nl <= MAXL, "tal3d: 1..%d levels", MAXL);
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "tal3d: bad dtype");
  Y3D_CHECK(nc >= 1 && nc <= 5 && no == nc + 35, "tal3d: nc in 1..5 and no == nc + 35");
  int a0 = 0;
  for (int i = 0; i < nl; ++i) {
    L.map[i] = maps[i]; L.psw[i] = psw[i];
    L.grad[i] = grads ? grads[i] : nullptr; L.gsw[i] = gsw ? gsw[i] : 0;
    L.H[i] = H[i]; L.W[i] = W[i]; L.a0[i] = a0; L.stride[i] = strides[i];
    a0 += H[i] * W[i];
  }
  for (int i = nl; i < MAXL; ++i) { L.map[i] = nullptr; L.grad[i] = nullptr; L.psw[i] = L.gsw[i] = 0; L.H[i] = L.W[i] = 1; L.a0[i] = a0; L.stride[i] = 1.f; }
  L.nl = nl; L.A = a0; L.B = B; L.nc = nc; L.no = no;
  return Y3D_OK;
}

int y3d_pad_targets(const float* rows, int nbox, int width, int B, int cap, float scale_x, float scale_y, float* out, int* n_used,
                    void* stream) {
  Y3D_CHECK(B >= 1 && cap >= 1 && width >= 5 && nbox >= 0, "pad_targets: B, cap >= 1, width >= 5 (cls + box + ...)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(64), 0, st, n_used, 2);
  hipLaunchKernelGGL(pad_targets_kernel, dim3(B), dim3(256), 0, st, rows, nbox, width, out, cap, scale_x, scale_y, n_used);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_tal3d_assign(int dtype, int nl, const void* const* maps, const int64_t* psw, const int* H, const int* W, const float* strides,
                     int B, int nc, const float* gt, int n, const float* calib, const float* mean_sizes, int topk, float alpha,
                     float beta, float gamma, int mode, float* scratch, uint8_t* fg_mask, int* target_gt_idx, float* target_scores, float* scal,
                     const int* n_used, void* stream) {
  Levels L;
  if (fill_levels(L, dtype, nl, maps, psw, nullptr, nullptr, H, W, strides, B, nc, nc + 35)) return Y3D_ERR_INVALID;
  Y3D_CHECK(n >= 1 && n <= 64, "tal3d_assign: 1..64 ground-truth boxes per image (got %d)", n);
  Y3D_CHECK(topk >= 1 && topk <= 16, "tal3d_assign: topk in 1..16");
  Y3D_CHECK((mode & 3) != 0 && (mode & ~15) == 0, "tal3d_assign: mode = %d: either the box metric (1) or the keypoint metric (2) or both must be selected", mode);
  hipStream_t st = (hipStream_t)stream;
  const int A = L.A;
  float* rec = scratch;
  float* align = rec + (long)B * n * GTW;
  float* sim = align + (long)B * n * A;
  int* cand = (int*)(sim + (long)B * n * A);
  unsigned* pa = (unsigned*)(cand + (long)B * n * topk);
  unsigned* po = pa + (long)B * n;
  float* part = (float*)(po + (long)B * n);
  hipLaunchKernelGGL(gt_prep_kernel, dim3(cdiv((long)B * n, 64)), dim3(64), 0, st, gt, calib, mean_sizes, rec, pa, po, B, n, nc);
  dim3 gm(cdiv(A, 256), B);
  size_t sm = (size_t)n * GTW * sizeof(float);
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(metric_kernel<bf16_t>, gm, dim3(256), sm, st, L, rec, calib, mean_sizes, align, sim, n, alpha, beta, gamma, n_used, mode);
  else hipLaunchKernelGGL(metric_kernel<float>, gm, dim3(256), sm, st, L, rec, calib, mean_sizes, align, sim, n, alpha, beta, gamma, n_used, mode);
  {
    const bool fits = (size_t)A * sizeof(float) <= 96 * 1024;  // the metric row in LDS (33.6 KB at 640x640; 1280x1280: 131 KB, global passes)
    hipLaunchKernelGGL(topk_kernel, dim3(B * n), dim3(256), fits ? (size_t)A * sizeof(float) : 0, st, align, rec, cand, L, n, topk, n_used, fits ? 1 : 0, (mode & 8) ? 1 : 0);
  }
  int nblk = cdiv((long)B * A, 256);
  hipLaunchKernelGGL(resolve_kernel, dim3(nblk), dim3(256), 0, st, cand, align, sim, fg_mask, target_gt_idx, pa, po, B, n, A, topk, n_used);
  hipLaunchKernelGGL(scores_kernel, dim3(nblk), dim3(256), 0, st, fg_mask, target_gt_idx, align, rec, pa, po, target_scores, part, B, n, A, nc, 1e-9f);
  hipLaunchKernelGGL(scal_kernel, dim3(1), dim3(64), 0, st, part, nblk, scal);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_loss3d(int dtype, int nl, const void* const* maps, const int64_t* psw, void* const* grads, const int64_t* gsw, const int* H,
               const int* W, const float* strides, int B, int nc, const float* gt, int n, const uint8_t* fg_mask,
               const int* target_gt_idx, const float* target_scores, const float* scal, float w_loss2d, float w_cls, float w_depth,
               float w_offset3d, float w_size3d, float w_heading, float grad_scale, float* partials, float* items, void* stream) {
  Levels L;
  if (fill_levels(L, dtype, nl, maps, psw, grads, gsw, H, W, strides, B, nc, nc + 35)) return Y3D_ERR_INVALID;
  LossW w{w_loss2d, w_cls, w_depth, w_offset3d, w_size3d, w_heading};
  hipStream_t st = (hipStream_t)stream;
  int nblk = cdiv((long)B * L.A, 256);
#define Y3D_L3D(T, NC) hipLaunchKernelGGL((loss_kernel<T, NC>), dim3(nblk), dim3(256), 0, st, L, fg_mask, target_gt_idx, target_scores, gt, scal, w, grad_scale, partials, n)
#define Y3D_L3D_NC(T) switch (nc) { case 1: Y3D_L3D(T, 1); break; case 2: Y3D_L3D(T, 2); break; case 3: Y3D_L3D(T, 3); break; case 4: Y3D_L3D(T, 4); break; default: Y3D_L3D(T, 5); break; }
  if (dtype == Y3D_BF16) { Y3D_L3D_NC(bf16_t) } else { Y3D_L3D_NC(float) }
#undef Y3D_L3D_NC
#undef Y3D_L3D
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partials, nblk, items);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
