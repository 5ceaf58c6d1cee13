// Depth-wise k x k convolution (3x3 s1/s2, 7x7 s1), NHWC, VALU stencil — HBM-bound, no MFMA.
// Reference call sites: SCDown.cv2 (block.py:824), CIB (block.py:747-751), RepVGGDW (block.py:705-706),
// Attention.pe (block.py:783), v10Detect cv3 (head.py:511-513); all `nn.Conv2d(c, c, k, s, groups=c, bias=False)`.
// A block owns a 64-channel slab x a contiguous pixel range; every thread owns one 16-byte channel chunk and
// strides over the pixels, the filter slab lives in LDS.  The forward also emits BatchNorm partial sums.
#include "common.h"

namespace {

struct DwP {
  const void* x;   // gathered tensor
  const float* w;  // packed [taps][C] fp32
  void* y;
  float* part;     // optional BN partials [nblk][C][2]
  long xsb, xsh, xsw, ysw;
  int B, Hg, Wg, Hq, Wq, C, kh, kw, stride, pad;
  long M;
  int px_per_block;
};

// K = 3 / 7: the filter rows are unrolled and a row's K loads are issued together (predicated, zero outside the map); with runtime
// tap loops every load waited for the previous one (1 TB/s on the 3x3 layers).  K = 0: any filter size (runtime loops).
template <typename T, bool DGRAD, int K>
__global__ __launch_bounds__(256) void dwconv_kernel(DwP p) {
  constexpr int CE = TT<T>::CE;
  constexpr int CT = 64 / CE;
  constexpr int PT = 256 / CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sw = (float*)smem;  // [taps][64]
  const int taps = p.kh * p.kw;
  float* sred = sw + taps * 64;  // [PT][64][2]
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int cs = blockIdx.y * 64;
  const int c = cs + ct * CE;
  for (int i = threadIdx.x; i < taps * 64; i += 256) {
    int t = i / 64, cc = i % 64;
    sw[i] = (cs + cc < p.C) ? p.w[(long)t * p.C + cs + cc] : 0.f;
  }
  __syncthreads();
  const T* __restrict__ X = (const T*)p.x;
  T* __restrict__ Y = (T*)p.y;
  float s1[CE], s2[CE];
#pragma unroll
  for (int j = 0; j < CE; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (c < p.C) {
    long pbeg = (long)blockIdx.x * p.px_per_block;
    long pend = pbeg + p.px_per_block < p.M ? pbeg + p.px_per_block : p.M;
    const int HWq = p.Hq * p.Wq;
    for (long m = pbeg + pt; m < pend; m += PT) {
      const int mi = (int)m;  // M < 2^31 (checked by the launcher)
      int b = mi / HWq;
      int rem = mi - b * HWq;
      int hq = rem / p.Wq, wq = rem - hq * p.Wq;
      float acc[CE];
#pragma unroll
      for (int j = 0; j < CE; ++j) acc[j] = 0.f;
      const T* xb = X + (long)b * p.xsb + c;
      if (K > 0) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
          int hh;
          bool okh;
          if (DGRAD) { int t = hq + p.pad - r; hh = t / p.stride; okh = t >= 0 && hh * p.stride == t && hh < p.Hg; }
          else { hh = hq * p.stride - p.pad + r; okh = hh >= 0 && hh < p.Hg; }
          uint4 raw[K > 0 ? K : 1];
#pragma unroll
          for (int q = 0; q < K; ++q) {
            int ww;
            bool okw;
            if (DGRAD) { int t = wq + p.pad - q; ww = t / p.stride; okw = t >= 0 && ww * p.stride == t && ww < p.Wg; }
            else { ww = wq * p.stride - p.pad + q; okw = ww >= 0 && ww < p.Wg; }
            raw[q] = (okh && okw) ? *(const uint4*)(xb + (long)hh * p.xsh + (long)ww * p.xsw) : make_uint4(0, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < K; ++q) {
            float v[CE];
            Chunk<T>::unpack(raw[q], v);
            const float* wt = sw + (r * K + q) * 64 + ct * CE;
#pragma unroll
            for (int j = 0; j < CE; ++j) acc[j] += v[j] * wt[j];
          }
        }
      } else
      for (int r = 0; r < p.kh; ++r) {
        int hh;
        bool okh;
        if (DGRAD) { int t = hq + p.pad - r; hh = t / p.stride; okh = t >= 0 && hh * p.stride == t && hh < p.Hg; }
        else { hh = hq * p.stride - p.pad + r; okh = hh >= 0 && hh < p.Hg; }
        if (!okh) continue;
        for (int q = 0; q < p.kw; ++q) {
          int ww;
          bool okw;
          if (DGRAD) { int t = wq + p.pad - q; ww = t / p.stride; okw = t >= 0 && ww * p.stride == t && ww < p.Wg; }
          else { ww = wq * p.stride - p.pad + q; okw = ww >= 0 && ww < p.Wg; }
          if (!okw) continue;
          float v[CE];
          Chunk<T>::unpack(*(const uint4*)(xb + (long)hh * p.xsh + (long)ww * p.xsw), v);
          const float* wt = sw + (r * p.kw + q) * 64 + ct * CE;
#pragma unroll
          for (int j = 0; j < CE; ++j) acc[j] += v[j] * wt[j];
        }
      }
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        acc[j] = TT<T>::rnd(acc[j]);
        s1[j] += acc[j];
        s2[j] += acc[j] * acc[j];
      }
      *(uint4*)(Y + m * p.ysw + c) = Chunk<T>::pack(acc);
    }
  }
  if (p.part) {
#pragma unroll
    for (int j = 0; j < CE; ++j) { sred[(pt * 64 + ct * CE + j) * 2] = s1[j]; sred[(pt * 64 + ct * CE + j) * 2 + 1] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < 64 && cs + threadIdx.x < p.C) {
      float a = 0.f, b2 = 0.f;
      for (int r = 0; r < PT; ++r) { a += sred[(r * 64 + threadIdx.x) * 2]; b2 += sred[(r * 64 + threadIdx.x) * 2 + 1]; }
      p.part[((long)blockIdx.x * p.C + cs + threadIdx.x) * 2] = a;
      p.part[((long)blockIdx.x * p.C + cs + threadIdx.x) * 2 + 1] = b2;
    }
  }
}

struct DwWP {
  const void* x;
  const void* dy;
  float* slab;  // [nblk][taps][C]
  long xsb, xsh, xsw, dsw;
  int B, H, W, Ho, Wo, C, kh, kw, stride, pad;
  long M;
  int px_per_block;
};

template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwWP p) {
  constexpr int CE = TT<T>::CE;
  constexpr int CT = 64 / CE;
  constexpr int PT = 256 / CT;
  constexpr int TG = 9;  // taps per pass
  __shared__ float sh[PT][64];
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int cs = blockIdx.y * 64;
  const int c = cs + ct * CE;
  const int taps = p.kh * p.kw;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ D = (const T*)p.dy;
  long pbeg = (long)blockIdx.x * p.px_per_block;
  long pend = pbeg + p.px_per_block < p.M ? pbeg + p.px_per_block : p.M;
  const int HWo = p.Ho * p.Wo;
  for (int t0 = 0; t0 < taps; t0 += TG) {
    float acc[TG][CE];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
      for (int j = 0; j < CE; ++j) acc[t][j] = 0.f;
    if (c < p.C) {
      for (long m = pbeg + pt; m < pend; m += PT) {
        int b = (int)(m / HWo);
        int rem = (int)(m - (long)b * HWo);
        int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        float d[CE];
        Chunk<T>::unpack(*(const uint4*)(D + m * p.dsw + c), d);
        const T* xb = X + (long)b * p.xsb + c;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          int tap = t0 + t;
          if (tap < taps) {
            int r = tap / p.kw, q = tap - r * p.kw;
            int hh = ho * p.stride - p.pad + r, ww = wo * p.stride - p.pad + q;
            if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W) {
              float v[CE];
              Chunk<T>::unpack(*(const uint4*)(xb + (long)hh * p.xsh + (long)ww * p.xsw), v);
#pragma unroll
              for (int j = 0; j < CE; ++j) acc[t][j] += d[j] * v[j];
            }
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t0 + t < taps) {  // uniform
#pragma unroll
        for (int j = 0; j < CE; ++j) sh[pt][ct * CE + j] = acc[t][j];
        __syncthreads();
        if (threadIdx.x < 64 && cs + threadIdx.x < p.C) {
          float a = 0.f;
          for (int r = 0; r < PT; ++r) a += sh[r][threadIdx.x];
          p.slab[((long)blockIdx.x * taps + t0 + t) * p.C + cs + threadIdx.x] = a;
        }
        __syncthreads();
      }
    }
  }
}

// slab[nblk][taps][C] -> grad OIHW [C][1][kh][kw]
__global__ void dw_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nblk, int taps, int C, int accumulate) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= taps * C) return;
  int c = idx % C, t = idx / C;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += slab[(long)b * taps * C + idx];
  long o = (long)c * taps + t;
  grad[o] = accumulate ? grad[o] + s : s;
}

__global__ void dw_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int taps) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= taps * C) return;
  int c = idx % C, t = idx / C;
  out[idx] = w[(long)c * taps + t];
}

}  // namespace

extern "C" {

int y3d_dw_blocks(int64_t M) {
  long n = (M + 255) / 256;
  return (int)(n < 1 ? 1 : (n > 2048 ? 2048 : n));
}

int y3d_dw_pack_weight(const float* w_oihw, float* out, int C, int kh, int kw, void* stream) {
  hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv((long)C * kh * kw, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, out, C, kh * kw);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

#define DW_LAUNCH_K(T, D, K) hipLaunchKernelGGL((dwconv_kernel<T, D, K>), grid, dim3(256), sm, st, p)
#define DW_LAUNCH(T)                                                                                             \
  do {                                                                                                           \
    const int kk = (p.kh == p.kw && (p.kh == 3 || p.kh == 7)) ? p.kh : 0;                                        \
    if (dgrad) { if (kk == 3) DW_LAUNCH_K(T, true, 3); else if (kk == 7) DW_LAUNCH_K(T, true, 7); else DW_LAUNCH_K(T, true, 0); }     \
    else { if (kk == 3) DW_LAUNCH_K(T, false, 3); else if (kk == 7) DW_LAUNCH_K(T, false, 7); else DW_LAUNCH_K(T, false, 0); }        \
  } while (0)

static int dw_launch(int dtype, bool dgrad, const DwP& p, hipStream_t st) {
  Y3D_CHECK(p.M < (1L << 31), "dwconv: more than 2^31 pixels");
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  int pt = 256 / (64 / ce);
  size_t sm = (size_t)(p.kh * p.kw * 64 + pt * 64 * 2) * sizeof(float);
  dim3 grid(y3d_dw_blocks(p.M), cdiv(p.C, 64));  // blocks past the end write zero partials
  if (dtype == Y3D_BF16) {
    DW_LAUNCH(bf16_t);
  } else {
    DW_LAUNCH(float);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_dwconv2d_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int C,
                     const float* w_packed, void* y, int64_t ysw, int Ho, int Wo, int kh, int kw, int stride, int pad,
                     float* stat_partials, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "dwconv2d_fwd: bad dtype");
  Y3D_CHECK(C % ce == 0, "dwconv2d_fwd: C=%d not a multiple of %d", C, ce);
  Y3D_CHECK(Ho == (H + 2 * pad - kh) / stride + 1 && Wo == (W + 2 * pad - kw) / stride + 1, "dwconv2d_fwd: output dims");
  Y3D_CHECK((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && xsb % ce == 0 && xsh % ce == 0 && xsw % ce == 0 && ysw % ce == 0, "dwconv2d_fwd: alignment");
  DwP p;
  p.x = x; p.w = w_packed; p.y = y; p.part = stat_partials;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.Hg = H; p.Wg = W; p.Hq = Ho; p.Wq = Wo; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.M = (long)B * Ho * Wo;
  int nblk = y3d_dw_blocks(p.M);
  p.px_per_block = (int)((p.M + nblk - 1) / nblk);
  return dw_launch(dtype, false, p, (hipStream_t)stream);
}

int y3d_dwconv2d_bwd_data(int dtype, const void* dy, int64_t dsb, int64_t dsh, int64_t dsw, int B, int Ho, int Wo, int C,
                          const float* w_packed, void* dx, int64_t xsw, int H, int W, int kh, int kw, int stride, int pad,
                          void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "dwconv2d_bwd_data: bad dtype");
  Y3D_CHECK(C % ce == 0, "dwconv2d_bwd_data: C=%d not a multiple of %d", C, ce);
  Y3D_CHECK((((uintptr_t)dy | (uintptr_t)dx) & 15) == 0 && dsb % ce == 0 && dsh % ce == 0 && dsw % ce == 0 && xsw % ce == 0, "dwconv2d_bwd_data: alignment");
  DwP p;
  p.x = dy; p.w = w_packed; p.y = dx; p.part = nullptr;
  p.xsb = dsb; p.xsh = dsh; p.xsw = dsw; p.ysw = xsw;
  p.B = B; p.Hg = Ho; p.Wg = Wo; p.Hq = H; p.Wq = W; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.M = (long)B * H * W;
  int nblk = y3d_dw_blocks(p.M);
  p.px_per_block = (int)((p.M + nblk - 1) / nblk);
  return dw_launch(dtype, true, p, (hipStream_t)stream);
}

int y3d_dwconv2d_bwd_weight(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int C,
                            const void* dy, int64_t dsw, int Ho, int Wo, int kh, int kw, int stride, int pad, float* slab,
                            float* grad_oihw, int accumulate, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "dwconv2d_bwd_weight: bad dtype");
  Y3D_CHECK(C % ce == 0, "dwconv2d_bwd_weight: C=%d not a multiple of %d", C, ce);
  Y3D_CHECK((((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && xsb % ce == 0 && xsh % ce == 0 && xsw % ce == 0 && dsw % ce == 0, "dwconv2d_bwd_weight: alignment");
  DwWP p;
  p.x = x; p.dy = dy; p.slab = slab;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.dsw = dsw;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.M = (long)B * Ho * Wo;
  int nblk = y3d_dw_blocks(p.M);
  p.px_per_block = (int)((p.M + nblk - 1) / nblk);
  dim3 grid(nblk, cdiv(C, 64));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), 0, st, p);
  Y3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(cdiv((long)C * kh * kw, 256)), dim3(256), 0, st, slab, grad_oihw, nblk, kh * kw, C, accumulate);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
