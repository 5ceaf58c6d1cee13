// Eval-side selection kernels of the NMS-free YOLOv10-3D head (HBM / latency bound, integer outputs):
//   * top-K cells of the max-class logit per image and level      (v10Detect3d.select_candidates, head.py:686-692)
//   * (k1+k2-1)^2 zero-padded input patches around those cells    (extract_patches, head.py:663-684)
//   * scatter of the per-candidate regression outputs + dense cls into the (B, H, W, no) map (head.py:709-713)
//   * decode to (B, no, A): xyxy px boxes, centre-3d px            (decode / inference, head.py:755-797)
//   * v10_3Dpostprocess: top-k over anchors, then over k x nc      (utils/ops.py:867-880)
// Ties are broken towards the lowest index (the reference's CUDA radix select leaves the order unspecified).
#include "common.h"

namespace {

// Order-preserving key of a float: larger value <=> larger key (-0 counts as +0).  Inputs are finite or +-inf (a NaN would sort first).
__device__ __forceinline__ unsigned topk_key(float v) {
  const unsigned u = __float_as_uint(v + 0.f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

// words of LDS work space block_topk needs for K winners: 256 histogram bins, 32 per-wave sums, 8 control words, K values + K indices
__host__ __device__ constexpr int topk_ws(int K) { return 296 + 2 * K; }

// The K best of `n` LDS values by (value desc, index asc), written in that order (n >= K; `vals` is left untouched).
// Radix select instead of K arg-max rounds (first form: one barrier + one wave reduction per winner, 2.9 us each - 148 us for the
// postprocess of a batch, 78 us for the 50 cells of an 80x80 level): four 8-bit passes over the keys find the key T of the K-th value
// (LDS histogram of the elements that match the prefix found so far, wave 0 walks the 256 bins from the top); then ONE ordered
// compaction - every thread owns a contiguous chunk, a block scan of the (> T, == T) counts places all elements above T and the
// lowest-index `need` elements equal to T - and a rank sort of the K survivors.  ~12 barriers in total, independent of K.
__device__ void block_topk(const float* vals, int n, int K, int* out_idx, float* out_val, int* ws) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  int* hist = ws;                     // [256]
  int* wsum = ws + 256;               // [2][16]
  int* ctl = ws + 288;                // [0] bin, [1] winners still to take inside it
  float* selv = (float*)(ws + 296);   // [K]
  int* seli = ws + 296 + K;           // [K]
  unsigned prefix = 0;
  int need = K;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned mask = pass ? 0xffffffffu << (shift + 8) : 0u;
    for (int i = tid; i < 256; i += nt) hist[i] = 0;
    __syncthreads();
    for (int a = tid; a < n; a += nt) {
      const unsigned k = topk_key(vals[a]);
      if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
    }
    __syncthreads();
    if (wave == 0) {  // lane L owns bins 255 - 4L .. 252 - 4L; the bin where the count from the top reaches `need`
      const int h0 = hist[255 - 4 * lane], h1 = hist[254 - 4 * lane], h2 = hist[253 - 4 * lane], h3 = hist[252 - 4 * lane];
      const int s = h0 + h1 + h2 + h3;
      int inc = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
      }
      int c = inc - s;  // elements in the bins above this lane's
      if (c < need && need <= inc) {
        int bin = 255 - 4 * lane;
        if (need > c + h0) { c += h0; --bin; if (need > c + h1) { c += h1; --bin; if (need > c + h2) { c += h2; --bin; } } }
        ctl[0] = bin;
        ctl[1] = need - c;
      }
    }
    __syncthreads();
    prefix |= (unsigned)ctl[0] << shift;
    need = ctl[1];
  }
  // prefix = key of the K-th value; `need` of the elements with that key are winners (the lowest indices), all elements above it are
  const unsigned T = prefix;
  const int above = K - need;
  const int chunk = ((n + nt - 1) / nt) | 1;  // odd: consecutive threads start on different LDS banks
  const int a0 = min(n, tid * chunk), a1 = min(n, a0 + chunk);
  int gt = 0, eq = 0;
  for (int a = a0; a < a1; ++a) {
    const unsigned k = topk_key(vals[a]);
    gt += k > T;
    eq += k == T;
  }
  int ig = gt, ie = eq;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(ig, d), u = __shfl_up(ie, d);
    if (lane >= d) { ig += t; ie += u; }
  }
  if (lane == 63) { wsum[wave] = ig; wsum[16 + wave] = ie; }
  __syncthreads();
  int pg = ig - gt, pe = ie - eq;
  for (int w = 0; w < wave; ++w) { pg += wsum[w]; pe += wsum[16 + w]; }
  for (int a = a0; a < a1; ++a) {
    const float v = vals[a];
    const unsigned k = topk_key(v);
    if (k > T) { selv[pg] = v; seli[pg] = a; ++pg; }
    else if (k == T) {
      if (pe < need) { selv[above + pe] = v; seli[above + pe] = a; }
      ++pe;
    }
  }
  __syncthreads();
  for (int j = tid; j < K; j += nt) {
    // ranked by the SAME total order the selection used (integer keys, index asc): a permutation of 0..K-1 for any input - with float
    // compares every NaN ranked 0 and some out_idx slots stayed unwritten (round-3 advisor finding)
    const float v = selv[j];
    const unsigned k = topk_key(v);
    const int i = seli[j];
    int r = 0;
    for (int q = 0; q < K; ++q) {
      const unsigned k2 = topk_key(selv[q]);
      r += (k2 > k) || (k2 == k && seli[q] < i);
    }
    out_idx[r] = i;
    if (out_val) out_val[r] = v;
  }
  __syncthreads();  // out_idx / out_val (often LDS) are complete for every thread
}

template <typename T>
__global__ __launch_bounds__(1024) void topk_cells_kernel(const T* __restrict__ cls, long psw, int HW, int nc, int K, int* __restrict__ out) {
  extern __shared__ float sm[];  // [HW] + selection work space
  float* vals = sm;
  int* ws = (int*)(sm + HW);
  const int b = blockIdx.x;
  for (int a = threadIdx.x; a < HW; a += blockDim.x) {
    const T* p = cls + ((long)b * HW + a) * psw;
    float m = TT<T>::ld(p);
    for (int c = 1; c < nc; ++c) m = fmaxf(m, TT<T>::ld(p + c));
    vals[a] = m;
  }
  __syncthreads();
  block_topk(vals, HW, K, out + (long)b * K, nullptr, ws);
}

// patches[(b*K + j)][py][px][c] = x[b][row - pad + py][col - pad + px][c]  (zero outside the map)
template <typename T>
__global__ void patch_gather_kernel(const T* __restrict__ x, long xsb, long xsh, long xsw, const int* __restrict__ idx, T* __restrict__ out,
                                    int B, int H, int W, int C, int K, int ps) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE, pad = ps / 2;
  long total = (long)B * K * ps * ps * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % cpr) * CE;
    long r = i / cpr;
    int px = (int)(r % ps);
    r /= ps;
    int py = (int)(r % ps);
    long bk = r / ps;
    int b = (int)(bk / K);
    int cell = idx[bk];
    int hh = cell / W - pad + py, ww = cell % W - pad + px;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = *(const uint4*)(x + (long)b * xsb + (long)hh * xsh + (long)ww * xsw + c);
    *(uint4*)(out + i * CE) = v;
  }
}

// map[b][cell][0:nc] = cls, map[b][cell][nc:no] = reg of the candidate sitting on that cell, else 0
template <typename T>
__global__ void head_scatter_kernel(const T* __restrict__ cls, long csw, const T* __restrict__ reg, long rsw, const int* __restrict__ idx,
                                    T* __restrict__ map, int B, int HW, int nc, int no, int K) {
  long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / HW), cell = (int)(i - (long)b * HW);
    T* m = map + i * no;
    const T* cp = cls + i * csw;
    for (int c = 0; c < nc; ++c) m[c] = cp[c];
    int hit = -1;
    for (int j = 0; j < K; ++j)
      if (idx[(long)b * K + j] == cell) hit = j;  // indices are unique per image
    if (hit >= 0) {
      const T* rp = reg + ((long)b * K + hit) * rsw;
      for (int c = nc; c < no; ++c) m[c] = rp[c - nc];
    } else {
      for (int c = nc; c < no; ++c) TT<T>::st(m + c, 0.f);
    }
  }
}

struct DecL {
  const void* map[4];
  int H[4], W[4], a0[4];
  float stride[4];
  int nl, A, nc, no;
};

// y[b][c][a] (fp32): cls | xyxy px | centre-3d px | s3d | hd | dep | dep_un    (head.py:755-764)
template <typename T>
__global__ void head_decode_kernel(DecL L, float* __restrict__ y, int B) {
  long total = (long)B * L.A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    int l = 0;
    for (int k = 1; k < 4; ++k) if (k < L.nl && a >= L.a0[k]) l = k;
    int r = a - L.a0[l];
    int hy = r / L.W[l], hx = r - hy * L.W[l];
    float ax = hx + 0.5f, ay = hy + 0.5f, st = L.stride[l];
    const T* p = (const T*)L.map[l] + (((long)b * L.H[l] + hy) * L.W[l] + hx) * L.no;
    float* o = y + (long)b * L.no * L.A + a;
    const int nc = L.nc;
    for (int c = 0; c < nc; ++c) o[(long)c * L.A] = TT<T>::ld(p + c);
    float o2x = TT<T>::ld(p + nc), o2y = TT<T>::ld(p + nc + 1), s2x = TT<T>::ld(p + nc + 2) * st, s2y = TT<T>::ld(p + nc + 3) * st;
    float cx = (o2x + ax) * st, cy = (o2y + ay) * st;
    o[(long)(nc + 0) * L.A] = cx - s2x / 2.f;
    o[(long)(nc + 1) * L.A] = cy - s2y / 2.f;
    o[(long)(nc + 2) * L.A] = cx + s2x / 2.f;
    o[(long)(nc + 3) * L.A] = cy + s2y / 2.f;
    o[(long)(nc + 4) * L.A] = (TT<T>::ld(p + nc + 4) + ax) * st;
    o[(long)(nc + 5) * L.A] = (TT<T>::ld(p + nc + 5) + ay) * st;
    for (int c = nc + 6; c < L.no; ++c) o[(long)c * L.A] = TT<T>::ld(p + c);
  }
}

// 2D head (Detect.inference head.py:53-79, DFL block.py:59-62, dist2bbox tal.py:315-325): per anchor the expectation of the softmax
// over the 16 bins of each box side, (lt, rb) distances -> xywh box in pixels, sigmoid class scores.
// map (B, H, W, 64 + nc) NHWC with channels [4 x 16 DFL bins | nc logits]  ->  y[b][c][a] fp32, c = x, y, w, h, scores
template <typename T>
__global__ void head2d_decode_kernel(DecL L, float* __restrict__ y, int B) {
  constexpr int RM = 16;
  long total = (long)B * L.A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / L.A), a = (int)(i - (long)b * L.A);
    int l = 0;
    for (int k = 1; k < 4; ++k) if (k < L.nl && a >= L.a0[k]) l = k;
    int r = a - L.a0[l];
    int hy = r / L.W[l], hx = r - hy * L.W[l];
    float ax = hx + 0.5f, ay = hy + 0.5f, st = L.stride[l];
    const T* p = (const T*)L.map[l] + (((long)b * L.H[l] + hy) * L.W[l] + hx) * L.no;
    const int nc = L.nc;
    const int C = 4 + nc;
    float* o = y + (long)b * C * L.A + a;
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v[RM], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < RM; ++j) { v[j] = TT<T>::ld(p + s * RM + j); mx = fmaxf(mx, v[j]); }
      float se = 0.f, ex = 0.f;
#pragma unroll
      for (int j = 0; j < RM; ++j) { float e = expf(v[j] - mx); se += e; ex += e * (float)j; }
      d[s] = ex / se;
    }
    float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    o[0] = (x1 + x2) / 2.f * st;
    o[(long)L.A] = (y1 + y2) / 2.f * st;
    o[2L * L.A] = (x2 - x1) * st;
    o[3L * L.A] = (y2 - y1) * st;
    for (int c = 0; c < nc; ++c) o[(long)(4 + c) * L.A] = 1.f / (1.f + expf(-TT<T>::ld(p + 4 * RM + c)));
  }
}

// ---- f3: image side of KITTIDataset.__getitem__ (data/datasets/kitti.py:132-206) --------------------------------------------------
// per output pixel: mirror (FLIP_LEFT_RIGHT), mixup blend (Pillow ImagingBlend: in1 + 0.5 * (in2 - in1) in float, truncated), affine
// crop with bilinear resampling in Pillow's arithmetic (Geometry.c: the pixel centre through the 2x3 matrix in double precision,
// outside the source -> 0, clamped neighbours, truncation to uint8).  One thread per output pixel, all three bands.
struct AugP {
  const unsigned char* const* src;   // [B] (H, W, 3) uint8 RGB
  const unsigned char* const* src2;  // [B] mixup partner or null entries (may be null altogether)
  const int* hw;                     // [B][2] source height, width
  const int* flip;                   // [B]
  const double* tinv;                // [B][6] output -> source affine map (the dataset's trans_inv)
  int B, oh, ow, mode;               // mode 0: (B, 3, oh, ow) fp32 = value / 255 (the reference's tensor); 1: (B, oh, ow, 3) uint8
};

__device__ __forceinline__ double aug_src(const AugP& p, int b, int H, int W, int fl, int y, int x, int band) {
  const int xs = fl ? W - 1 - x : x;
  const long o = ((long)y * W + xs) * 3 + band;
  const unsigned char* s1 = p.src[b];
  const unsigned char* s2 = p.src2 ? p.src2[b] : nullptr;
  int a = s1[o];
  if (s2) {
    // (UINT8)((int)in1 + alpha * ((int)in2 - (int)in1)) with alpha = 0.5f: float arithmetic, truncation
    float f = (float)a + 0.5f * (float)((int)s2[o] - a);
    a = (int)f;
  }
  return (double)a;
}

__global__ __launch_bounds__(256) void kitti_aug_kernel(AugP p, void* __restrict__ out) {
  const long total = (long)p.B * p.oh * p.ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long)p.oh * p.ow));
    const int r = (int)(i - (long)b * p.oh * p.ow);
    const int oy = r / p.ow, ox = r - oy * p.ow;
    const int H = p.hw[b * 2], W = p.hw[b * 2 + 1], fl = p.flip[b];
    const double* t = p.tinv + b * 6;
    const double xc = ox + 0.5, yc = oy + 0.5;
    double xin = t[0] * xc + t[1] * yc + t[2];
    double yin = t[3] * xc + t[4] * yc + t[5];
    int v[3] = {0, 0, 0};
    if (!(xin < 0.0 || xin >= (double)W || yin < 0.0 || yin >= (double)H)) {
      xin -= 0.5;
      yin -= 0.5;
      const int x = xin < 0.0 ? (int)floor(xin) : (int)xin, y = yin < 0.0 ? (int)floor(yin) : (int)yin;
      const double dx = xin - x, dy = yin - y;
      const int yc0 = y < 0 ? 0 : (y >= H ? H - 1 : y);
      const int x0 = x < 0 ? 0 : (x >= W ? W - 1 : x), x1 = x + 1 < 0 ? 0 : (x + 1 >= W ? W - 1 : x + 1);
      const bool y1ok = y + 1 >= 0 && y + 1 < H;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double a0 = aug_src(p, b, H, W, fl, yc0, x0, c), a1 = aug_src(p, b, H, W, fl, yc0, x1, c);
        double v1 = a0 + (a1 - a0) * dx, v2 = v1;
        if (y1ok) {
          const double b0 = aug_src(p, b, H, W, fl, y + 1, x0, c), b1 = aug_src(p, b, H, W, fl, y + 1, x1, c);
          v2 = b0 + (b1 - b0) * dx;
        }
        v[c] = (int)(unsigned char)(v1 + (v2 - v1) * dy);
      }
    }
    if (p.mode == 0) {
      float* o = (float*)out + (long)b * 3 * p.oh * p.ow + r;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[(long)c * p.oh * p.ow] = (float)v[c] / 255.0f;
    } else {
      unsigned char* o = (unsigned char*)out + i * 3;
      o[0] = (unsigned char)v[0]; o[1] = (unsigned char)v[1]; o[2] = (unsigned char)v[2];
    }
  }
}

// ---- f2: one-to-many depth fusion of the validator (models/yolov10_3D/val.py:78-102) -------------------------------------------------
// one block per one-to-one detection: collect the votes (itself + matching one-to-many rows, in index order), then the weighted
// gaussian kernel density at NPROP proposals (double precision, as scikit-learn), first maximum of the log-density
__global__ __launch_bounds__(256) void kde_fusion_kernel(const float* __restrict__ O, const float* __restrict__ M, float* __restrict__ out, int K, int KM,
                                                         int C, float thres, float iou_thres, int nprop) {
  extern __shared__ float sm[];       // (one dynamic-LDS symbol per translation unit: float[], viewed as doubles here; 16-byte aligned)
  double* xs = (double*)sm;           // [KM + 1] votes
  double* ws = xs + KM + 1;           // [KM + 1] weights
  double* bv = ws + KM + 1;           // [256]
  int* bi = (int*)(bv + 256);         // [256]
  __shared__ int nv;
  __shared__ float s_lo, s_hi;
  const int b = blockIdx.y, j = blockIdx.x;
  const float* o = O + ((long)b * K + j) * C;
  float* dst = out + ((long)b * K + j) * C;
  for (int c = threadIdx.x; c < C; c += 256) dst[c] = o[c];
  if (threadIdx.x == 0) {
    const float eps = 1e-7f;
    const float ax1 = o[0], ay1 = o[1], ax2 = o[2], ay2 = o[3], lab = o[C - 1];
    const float areaA = (ax2 - ax1) * (ay2 - ay1);
    int n = 0;
    float ssum = 0.f, lo = 0.f, hi = 0.f;
    auto vote = [&](float d, float u, float c) {
      const float s = expf(-u);
      if (s > thres && c == lab) {
        xs[n] = (double)d;
        ws[n] = (double)s;  // normalised below with the float32 sum, as the reference
        ssum = n == 0 ? s : ssum + s;
        lo = n == 0 ? d : fminf(lo, d);
        hi = n == 0 ? d : fmaxf(hi, d);
        ++n;
      }
    };
    vote(o[C - 4], o[C - 3], lab);
    for (int m = 0; m < KM; ++m) {
      const float* q = M + ((long)b * KM + m) * C;
      float iw = fminf(ax2, q[2]) - fmaxf(ax1, q[0]), ih = fminf(ay2, q[3]) - fmaxf(ay1, q[1]);
      iw = iw < 0.f ? 0.f : iw;
      ih = ih < 0.f ? 0.f : ih;
      const float inter = iw * ih;
      const float iou = inter / (areaA + (q[2] - q[0]) * (q[3] - q[1]) - inter + eps);
      if (iou > iou_thres) vote(q[C - 4], q[C - 3], q[C - 1]);
    }
    for (int k = 0; k < n; ++k) ws[k] = (double)((float)ws[k] / ssum);
    nv = n;
    s_lo = lo;
    s_hi = hi;
  }
  __syncthreads();
  const int n = nv;
  if (n <= 1) return;  // uniform
  const double h = pow((double)n * 3.0 / 4.0, -0.2);
  const float lo = s_lo, hi = s_hi;
  const float step = (hi - lo) / (float)(nprop - 1);  // np.linspace on float32 end points
  double best = -INFINITY;
  int besti = 0x7fffffff;
  for (int k = threadIdx.x; k < nprop; k += 256) {
    const float pf = k == nprop - 1 ? hi : (float)k * step + lo;
    const double pp = (double)pf;
    double dens = 0.0;
    for (int t = 0; t < n; ++t) {
      const double z = (pp - xs[t]) / h;
      dens += ws[t] * exp(-0.5 * z * z);
    }
    const double ld = log(dens);
    if (ld > best) { best = ld; besti = k; }  // k ascends per thread: first maximum
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = besti;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const double v2 = bv[threadIdx.x + s];
      const int i2 = bi[threadIdx.x + s];
      if (v2 > bv[threadIdx.x] || (v2 == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v2; bi[threadIdx.x] = i2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int k = bi[0];
    dst[C - 4] = k == nprop - 1 ? hi : (float)k * step + lo;
  }
}

// preds y (B, C, A) fp32 with `nc` score rows first: reg (B,K,C-nc), scores (B,K), labels (B,K) int64
// hi-res maps, stage 1: block (seg, b) scores the anchors of its segment (max over classes) in LDS and keeps their K best
__global__ __launch_bounds__(1024) void postprocess_seg_kernel(const float* __restrict__ y, int A, int C, int nc, int K, int boxes_first, int seglen,
                                                              float* __restrict__ scratch) {
  extern __shared__ float sm[];
  float* vals = sm;            // [seglen]
  int* ws = (int*)(sm + seglen);  // [topk_ws(K)]
  int* top = ws + topk_ws(K);  // [K]
  float* tv = (float*)(top + K);  // [K]
  const int seg = blockIdx.x, nseg = gridDim.x, b = blockIdx.y;
  const float* yb = y + (long)b * C * A;
  const int s0 = boxes_first ? C - nc : 0;
  const int a0 = seg * seglen;
  const int n = min(seglen, A - a0);
  for (int a = threadIdx.x; a < n; a += blockDim.x) {
    float m = yb[(long)s0 * A + a0 + a];
    for (int c = 1; c < nc; ++c) m = fmaxf(m, yb[(long)(s0 + c) * A + a0 + a]);
    vals[a] = m;
  }
  __syncthreads();
  const int kk = min(K, n);
  block_topk(vals, n, kk, top, tv, ws);
  float* cv = scratch + ((long)b * nseg + seg) * K;
  int* ci = (int*)(scratch + (long)gridDim.y * nseg * K) + ((long)b * nseg + seg) * K;
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    cv[i] = i < kk ? tv[i] : -INFINITY;
    ci[i] = i < kk ? a0 + top[i] : 0;
  }
}

__global__ __launch_bounds__(1024) void postprocess_kernel(const float* __restrict__ y, int A, int C, int nc, int K, int boxes_first,
                                                          float* __restrict__ reg, float* __restrict__ scores, long* __restrict__ labels,
                                                          float* __restrict__ scratch, int nseg) {
  extern __shared__ float sm[];
  // scratch == nullptr: the A per-anchor scores live in LDS.  Otherwise (hi-res maps) postprocess_seg_kernel has already picked the
  // K best anchors of each of `nseg` anchor segments: scratch holds their values [B][nseg*K] and anchor indices [B][nseg*K]
  const int ncand = scratch ? nseg * K : A;
  float* vals = sm;           // [ncand]
  int* ws = (int*)(sm + ncand);  // [topk_ws(K)]
  int* top = ws + topk_ws(K);   // [K]
  float* sc2 = (float*)(top + K);  // [K*nc]
  int* top2 = (int*)(sc2 + K * nc);  // [K]
  float* val2 = (float*)(top2 + K);  // [K]
  const int b = blockIdx.x;
  const float* yb = y + (long)b * C * A;
  const int s0 = boxes_first ? C - nc : 0;   // first score row
  const int r0 = boxes_first ? 0 : nc;       // first regression row
  const int nr = C - nc;
  if (scratch) {
    const float* cv = scratch + (long)b * ncand;
    for (int a = threadIdx.x; a < ncand; a += blockDim.x) vals[a] = cv[a];
  } else {
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
      float m = yb[(long)s0 * A + a];
      for (int c = 1; c < nc; ++c) m = fmaxf(m, yb[(long)(s0 + c) * A + a]);
      vals[a] = m;
    }
  }
  __syncthreads();
  block_topk(vals, ncand, K, top, nullptr, ws);
  if (scratch) {  // candidate position -> anchor index (positions are ordered by segment, then by rank: ties keep the lowest anchor first)
    const int* ci = (const int*)(scratch + (long)gridDim.x * ncand) + (long)b * ncand;
    for (int i = threadIdx.x; i < K; i += blockDim.x) top[i] = ci[top[i]];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < K * nc; i += blockDim.x) sc2[i] = yb[(long)(s0 + i % nc) * A + top[i / nc]];
  __syncthreads();
  block_topk(sc2, K * nc, K, top2, val2, ws);
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    scores[(long)b * K + i] = val2[i];
    labels[(long)b * K + i] = top2[i] % nc;
  }
  for (int i = threadIdx.x; i < K * nr; i += blockDim.x) {
    int j = i / nr, c = i - j * nr;
    reg[((long)b * K + j) * nr + c] = yb[(long)(r0 + c) * A + top[top2[j] / nc]];
  }
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}


// KITTI decode of the post-processed rows (data/datasets/kitti.py:519-576): one thread per detection.  The reference promotes the
// float32 predictions to float64 through the calibration constants; the float32 steps (heading angle, sigmoid, exp, size residual)
// are kept in float32 here as there.
__global__ __launch_bounds__(256) void kitti_decode_kernel(const float* __restrict__ preds, int B, int K, const double* __restrict__ calib,
                                                           const double* __restrict__ ratio, const double* __restrict__ inv_trans,
                                                           const double* __restrict__ mean_size, int nc, int use_camera_dis, double threshold,
                                                           double* __restrict__ out, unsigned char* __restrict__ keep) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * K) return;
  const int i = idx / K;
  const float* r = preds + (long)idx * 37;
  const double cu = calib[i * 6 + 0], cv = calib[i * 6 + 1], fu = calib[i * 6 + 2], fv = calib[i * 6 + 3], tx = calib[i * 6 + 4], ty = calib[i * 6 + 5];
  const double rw = ratio[i * 2 + 0], rh = ratio[i * 2 + 1];
  int cid = (int)r[36];
  const int cm = cid < 0 ? 0 : (cid >= nc ? nc - 1 : cid);  // the reference would raise on a label outside the mean-size table
  // heading: first maximum of the 12 bin logits, residual of that bin (decode_helper.py:12-18, float32)
  int bin = 0;
  float best = r[9];
  for (int k = 1; k < 12; ++k) if (r[9 + k] > best) { best = r[9 + k]; bin = k; }
  const float PI_F = 3.14159265358979323846f;
  float ang = (float)bin * (float)(2.0 * 3.14159265358979323846 / 12.0) + r[21 + bin];
  if (ang > PI_F) ang = ang - (float)(2.0 * 3.14159265358979323846);
  const double alpha = (double)ang;
  const double x1 = (double)r[0] / rw, y1 = (double)r[1] / rh, x2 = (double)r[2] / rw, y2 = (double)r[3] / rh;
  const double xc = (x1 + x2) / 2;
  const float h = r[6] + (float)mean_size[cm * 3 + 0], w = r[7] + (float)mean_size[cm * 3 + 1], l = r[8] + (float)mean_size[cm * 3 + 2];
  const double depth = (double)r[33];
  const double sigma = (double)expf(-r[34]);
  double u, v;
  if (inv_trans) {
    const double* t = inv_trans + i * 6;
    u = t[0] * (double)r[4] + t[1] * (double)r[5] + t[2];
    v = t[3] * (double)r[4] + t[4] * (double)r[5] + t[5];
  } else {
    u = (double)((r[4] * 1242.f) / 1280.f);
    v = (double)((r[5] * 375.f) / 384.f);
  }
  double lx, ly, lz;
  if (use_camera_dis) {  // kitti_utils.py:286-299
    const double fd = sqrt((u - cu) * (u - cu) + (v - cv) * (v - cv) + fu * fu);
    lx = ((u - cu) * depth) / fd + tx;
    ly = ((v - cv) * depth) / fd + ty;
    lz = sqrt(depth * depth - lx * lx - ly * ly);
  } else {               // kitti_utils.py:241-251
    lx = ((u - cu) * depth) / fu + tx;
    ly = ((v - cv) * depth) / fv + ty;
    lz = depth;
  }
  ly += (double)h / 2;
  const double PI_D = 3.14159265358979323846;
  double ry = alpha + atan2(xc - cu, fu);  // kitti_utils.py:311-325
  if (ry > PI_D) ry -= 2 * PI_D;
  if (ry < -PI_D) ry += 2 * PI_D;
  const float sg = 1.f / (1.f + expf(-r[35]));
  const double score = (double)sg * sigma;
  double* o = out + (long)idx * 14;
  o[0] = (double)cid; o[1] = alpha; o[2] = x1; o[3] = y1; o[4] = x2; o[5] = y2; o[6] = (double)h; o[7] = (double)w; o[8] = (double)l;
  o[9] = lx; o[10] = ly; o[11] = lz; o[12] = ry; o[13] = score;
  keep[idx] = score < threshold ? 0 : 1;
}

}  // namespace

extern "C" {

int y3d_topk_cells(int dtype, const void* cls, int64_t psw, int B, int HW, int nc, int K, int* out_idx, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "topk_cells: bad dtype");
  Y3D_CHECK(K >= 1 && K <= HW, "topk_cells: need 1 <= K <= H*W (K=%d, H*W=%d)", K, HW);
  size_t sm = (size_t)(HW + topk_ws(K)) * 4;
  Y3D_CHECK(sm <= 160 * 1024, "topk_cells: map of %d cells does not fit LDS", HW);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) {
    (void)hipFuncSetAttribute((const void*)topk_cells_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(topk_cells_kernel<bf16_t>, dim3(B), dim3(HW >= 2048 ? 1024 : 256), sm, st, (const bf16_t*)cls, (long)psw, HW, nc, K, out_idx);
  } else {
    (void)hipFuncSetAttribute((const void*)topk_cells_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(topk_cells_kernel<float>, dim3(B), dim3(HW >= 2048 ? 1024 : 256), sm, st, (const float*)cls, (long)psw, HW, nc, K, out_idx);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_patch_gather(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, const int* idx, void* out, int B, int H, int W,
                     int C, int K, int ps, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "patch_gather: bad dtype");
  Y3D_CHECK(C % ce == 0 && ((uintptr_t)x & 15) == 0 && xsb % ce == 0 && xsh % ce == 0 && xsw % ce == 0, "patch_gather: alignment");
  long total = (long)B * K * ps * ps * (C / ce);
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(patch_gather_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)xsb, (long)xsh, (long)xsw, idx, (bf16_t*)out, B, H, W, C, K, ps);
  else hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)xsb, (long)xsh, (long)xsw, idx, (float*)out, B, H, W, C, K, ps);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_head3d_scatter(int dtype, const void* cls, int64_t csw, const void* reg, int64_t rsw, const int* idx, void* map, int B, int HW,
                       int nc, int no, int K, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "head3d_scatter: bad dtype");
  long total = (long)B * HW;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(head_scatter_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)cls, (long)csw, (const bf16_t*)reg, (long)rsw, idx, (bf16_t*)map, B, HW, nc, no, K);
  else hipLaunchKernelGGL(head_scatter_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)cls, (long)csw, (const float*)reg, (long)rsw, idx, (float*)map, B, HW, nc, no, K);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_head3d_decode(int dtype, int nl, const void* const* maps, const int* H, const int* W, const float* strides, int B, int nc,
                      float* y, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "head3d_decode: bad dtype");
  Y3D_CHECK(nl >= 1 && nl <= 4, "head3d_decode: 1..4 levels");
  DecL L;
  int a0 = 0;
  for (int i = 0; i < 4; ++i) {
    L.map[i] = i < nl ? maps[i] : nullptr;
    L.H[i] = i < nl ? H[i] : 1; L.W[i] = i < nl ? W[i] : 1; L.a0[i] = a0; L.stride[i] = i < nl ? strides[i] : 1.f;
    if (i < nl) a0 += H[i] * W[i];
  }
  L.nl = nl; L.A = a0; L.nc = nc; L.no = nc + 35;
  long total = (long)B * a0;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(head_decode_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, L, y, B);
  else hipLaunchKernelGGL(head_decode_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, L, y, B);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_head2d_decode(int dtype, int nl, const void* const* maps, const int* H, const int* W, const float* strides, int B, int nc,
                      float* y, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "head2d_decode: bad dtype");
  Y3D_CHECK(nl >= 1 && nl <= 4 && nc >= 1, "head2d_decode: 1..4 levels, nc >= 1");
  DecL L;
  int a0 = 0;
  for (int i = 0; i < 4; ++i) {
    L.map[i] = i < nl ? maps[i] : nullptr;
    L.H[i] = i < nl ? H[i] : 1; L.W[i] = i < nl ? W[i] : 1; L.a0[i] = a0; L.stride[i] = i < nl ? strides[i] : 1.f;
    if (i < nl) a0 += H[i] * W[i];
  }
  L.nl = nl; L.A = a0; L.nc = nc; L.no = nc + 64;
  long total = (long)B * a0;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(head2d_decode_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, L, y, B);
  else hipLaunchKernelGGL(head2d_decode_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, L, y, B);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_v10_postprocess_scratch_floats(int B, int A, int nc, int max_det) {
  return (size_t)(A + topk_ws(max_det) + max_det * (nc + 3)) * 4 <= 160 * 1024 ? 0 : B * A;
}

int y3d_v10_postprocess(const float* y, int B, int C, int A, int nc, int max_det, int boxes_first, float* reg, float* scores,
                        int64_t* labels, float* scratch, void* stream) {
  Y3D_CHECK(max_det >= 1 && max_det <= A && nc >= 1 && nc < C, "v10_postprocess: bad sizes");
  const bool fits = y3d_v10_postprocess_scratch_floats(B, A, nc, max_det) == 0;
  Y3D_CHECK(fits || scratch, "v10_postprocess: %d anchors do not fit LDS and no scratch was given (y3d_v10_postprocess_scratch_floats)", A);
  if (fits) scratch = nullptr;
  // hi-res: segments of anchors whose K best are merged by the second stage; candidates (values + indices) must fit the B*A scratch
  int nseg = 0, seglen = 0;
  if (!fits) {
    nseg = 16;
    while (nseg > 1 && (long)nseg * max_det * 2 > A) nseg >>= 1;
    seglen = cdiv(A, nseg);
    Y3D_CHECK((long)nseg * max_det * 2 <= A && (size_t)(seglen + topk_ws(max_det) + 2 * max_det) * 4 <= 160 * 1024, "v10_postprocess: %d anchors with max_det %d", A, max_det);
    size_t sm1 = (size_t)(seglen + topk_ws(max_det) + 2 * max_det) * 4;
    (void)hipFuncSetAttribute((const void*)postprocess_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(postprocess_seg_kernel, dim3(nseg, B), dim3(1024), sm1, (hipStream_t)stream, y, A, C, nc, max_det, boxes_first, seglen, scratch);
  }
  size_t sm = (size_t)((fits ? A : nseg * max_det) + topk_ws(max_det) + max_det * (nc + 3)) * 4;
  Y3D_CHECK(sm <= 160 * 1024, "v10_postprocess: max_det * nc = %d does not fit LDS", max_det * nc);
  (void)hipFuncSetAttribute((const void*)postprocess_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(postprocess_kernel, dim3(B), dim3((fits ? A : nseg * max_det) >= 2048 ? 1024 : 256), sm, (hipStream_t)stream, y, A, C, nc, max_det, boxes_first, reg, scores, (long*)labels, scratch,
                     nseg);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_kitti_image_aug(const unsigned char* const* src, const unsigned char* const* src2, const int* hw, const int* flip, const double* trans_inv,
                        int B, int out_h, int out_w, int mode, void* out, void* stream) {
  Y3D_CHECK(src && hw && flip && trans_inv && out, "kitti_image_aug: null argument");
  Y3D_CHECK(B >= 1 && out_h >= 1 && out_w >= 1 && (mode == 0 || mode == 1), "kitti_image_aug: bad sizes / mode");
  AugP p{src, src2, hw, flip, trans_inv, B, out_h, out_w, mode};
  hipLaunchKernelGGL(kitti_aug_kernel, dim3(ew_grid((long)B * out_h * out_w)), dim3(256), 0, (hipStream_t)stream, p, out);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_kde_depth_fusion(const float* predsO, int B, int K, const float* predsM, int KM, int C, float thres, float iou_thres, int nprop,
                         float* out, void* stream) {
  Y3D_CHECK(predsO && predsM && out && B >= 1 && K >= 1 && KM >= 1 && C >= 8 && nprop >= 2, "kde_depth_fusion: bad arguments");
  size_t sm = (size_t)(2 * (KM + 1) + 256) * sizeof(double) + 256 * sizeof(int);
  Y3D_CHECK(sm <= 64 * 1024, "kde_depth_fusion: %d one-to-many rows do not fit LDS", KM);
  hipLaunchKernelGGL(kde_fusion_kernel, dim3(K, B), dim3(256), sm, (hipStream_t)stream, predsO, predsM, out, K, KM, C, thres, iou_thres, nprop);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_kitti_decode(const float* preds, int B, int K, const double* calib, const double* ratio, const double* inv_trans,
                     const double* mean_size, int nc, int use_camera_dis, double threshold, double* out, unsigned char* keep, void* stream) {
  Y3D_CHECK(B >= 1 && K >= 1 && nc >= 1, "kitti_decode: B, K, nc must be positive");
  Y3D_CHECK(preds && calib && ratio && mean_size && out && keep, "kitti_decode: null argument");
  hipLaunchKernelGGL(kitti_decode_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, preds, B, K, calib, ratio, inv_trans, mean_size,
                     nc, use_camera_dis, threshold, out, keep);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
