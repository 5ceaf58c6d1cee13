// Depth-wise k x k convolution (3x3 s1/s2, 7x7 s1), NHWC, VALU stencil — HBM-bound, no MFMA.
// Reference call sites: SCDown.cv2 (block.py:824), CIB (block.py:747-751), RepVGGDW (block.py:705-706),
// Attention.pe (block.py:783), v10Detect cv3 (head.py:511-513); all `nn.Conv2d(c, c, k, s, groups=c, bias=False)`.
// A block owns a 64-channel slab x a contiguous pixel range; every thread owns one 16-byte channel chunk and
// strides over the pixels, the filter slab lives in LDS.  The forward also emits BatchNorm partial sums.
#include "common.h"

namespace {

// acc[j] += a[j] * b[j] for CE channels, two per instruction (v_pk_fma_f32): these kernels are VALU-bound on the 7x7 layers (49 taps x
// 8 channels per 16 bytes of output); a fused multiply-add also drops the separate rounding of the product
template <int CE>
__device__ __forceinline__ void pk_fma(float* acc, const float* a, const float* b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < CE; j += 2) {
    const f32x2_t c2 = {acc[j], acc[j + 1]}, a2 = {a[j], a[j + 1]}, b2 = {b[j], b[j + 1]};
    const f32x2_t o2 = __builtin_elementwise_fma(a2, b2, c2);
    acc[j] = o2[0];
    acc[j + 1] = o2[1];
  }
}

struct DwP {
  const void* x;   // gathered tensor
  const float* w;  // packed [taps][C] fp32
  void* y;
  float* part;     // optional BN partials [nblk][C][2]
  long xsb, xsh, xsw, ysw;
  int B, Hg, Wg, Hq, Wq, C, kh, kw, stride, pad;
  long M;
  int px_per_block;
  // eval epilogue (AFF): z = act(y * scale[c] + shift[c] (+ res if res_mode 2)) (+ res if res_mode 1) - the folded BatchNorm + SiLU + residual
  // that y3d_bn_act_fwd would apply in a second pass, on the value ROUNDED as the pre-BatchNorm tensor would have been stored
  const float *scale, *shift;
  const void* res;
  long rsw;
  int act, res_mode;
};

// K = 3 / 7: the filter rows are unrolled and a row's K loads are issued together (predicated, zero outside the map); with runtime
// tap loops every load waited for the previous one (1 TB/s on the 3x3 layers).  K = 0: any filter size (runtime loops).
template <typename T, bool DGRAD, int K, bool AFF = false>
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
  float esc[CE], esf[CE];
  if (AFF && c < p.C) {
#pragma unroll
    for (int j = 0; j < CE; ++j) { esc[j] = p.scale[c + j]; esf[j] = p.shift[c + j]; }
  }
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
      if (K == 3 && DGRAD && p.stride == 2) {
        // stride-2 data gradient: only the taps whose parity matches the pixel reach dy - rows r0, r0 + 2 with r0 = (hq + pad) & 1
        // (one or two of three), columns likewise: at most 4 tap slots instead of 9, no division (the generic form below spent its
        // time on 18 integer divisions and 9 mostly masked taps per pixel: 1.2 TB/s against 4.1 for the forward)
        const int r0 = (hq + p.pad) & 1, q0 = (wq + p.pad) & 1;
        uint4 raw[2][2];
        int tapi[2][2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int r = r0 + 2 * rr, th = hq + p.pad - r, hh = th >> 1;
          const bool okh = r < 3 && th >= 0 && hh < p.Hg;
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int q = q0 + 2 * qq, tw = wq + p.pad - q, ww = tw >> 1;
            const bool ok = okh && q < 3 && tw >= 0 && ww < p.Wg;
            raw[rr][qq] = ok ? *(const uint4*)(xb + (long)hh * p.xsh + (long)ww * p.xsw) : make_uint4(0, 0, 0, 0);
            tapi[rr][qq] = ok ? r * 3 + q : 0;
          }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            float v[CE];
            Chunk<T>::unpack(raw[rr][qq], v);
            pk_fma<CE>(acc, v, sw + tapi[rr][qq] * 64 + ct * CE);
          }
      } else if (K > 0) {
        // K = 7: one filter row (7 loads) in flight at a time -- unrolled over all 49 taps the kernel needed 512 VGPRs and spilled
#pragma unroll K == 3 ? 3 : 1
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
            pk_fma<CE>(acc, v, sw + (r * K + q) * 64 + ct * CE);
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
          pk_fma<CE>(acc, v, sw + (r * p.kw + q) * 64 + ct * CE);
        }
      }
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        acc[j] = TT<T>::rnd(acc[j]);
        s1[j] += acc[j];
        s2[j] += acc[j] * acc[j];
      }
      if (AFF) {
        float r[CE];
        if (p.res_mode) Chunk<T>::unpack(*(const uint4*)((const T*)p.res + m * p.rsw + c), r);
#pragma unroll
        for (int j = 0; j < CE; ++j) {
          float u = acc[j] * esc[j] + esf[j];
          if (p.res_mode == 2) u += r[j];
          if (p.act) u = silu_f(u);
          if (p.res_mode == 1) u += r[j];
          acc[j] = u;
        }
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
  int px_per_block, nblk, nslab;
};

// K = 3 / 7: filter width known at compile time; K = 0: any width up to 8.  A block owns ONE filter row of a 64-channel
// slab over a pixel range: the small maps these layers run on (20x20 .. 80x80) need the parallelism -- walking the rows in passes
// inside a block left the 7x7 layers latency-bound at 150 us for 13 MB.  Pixel coordinates advance incrementally (no divisions in
// the loop), a row's loads are issued together (predicated) and two pixels are in flight per thread.
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwWP p) {
  constexpr int CE = TT<T>::CE;
  constexpr int CT = 64 / CE;
  constexpr int PT = 256 / CT;
  constexpr int RP = 1;                                // filter rows per block
  constexpr int KW = K > 0 ? K : 8;                    // accumulator columns (K = 0: kw <= 8)
  __shared__ float sh[4][RP * KW * 64];
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int kh = K > 0 ? K : p.kh, kw = K > 0 ? K : p.kw;
  // XCD-aware block order: workgroup L runs on XCD L % 8; within one XCD's stream the filter rows of a (pixel range, channel slab)
  // tile are consecutive, so the kh blocks that read the same dy tile and overlapping x rows share that XCD's L2
  const int L = blockIdx.x, jx = L >> 3;
  const int row = jx % kh, tile = (jx / kh) * 8 + (L & 7);
  if (tile >= p.nblk * p.nslab) return;
  const int bx = tile % p.nblk;
  const int cs = (tile / p.nblk) * 64;
  const int c = cs + ct * CE;
  const int taps = kh * kw;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ D = (const T*)p.dy;
  const long pbeg = (long)bx * p.px_per_block;
  const long pend = pbeg + p.px_per_block < p.M ? pbeg + p.px_per_block : p.M;
  const int HWo = p.Ho * p.Wo;
  {
    const int r0 = row * RP;
    float acc[RP][KW][CE];
#pragma unroll
    for (int r = 0; r < RP; ++r)
#pragma unroll
      for (int q = 0; q < KW; ++q)
#pragma unroll
        for (int j = 0; j < CE; ++j) acc[r][q][j] = 0.f;
    if (c < p.C) {
      long m = pbeg + pt;
      int b = (int)(m / HWo);
      int rem = (int)(m - (long)b * HWo);
      int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      auto one = [&](const uint4& dv, int bb, int hh0, int ww0) {
        float d[CE];
        Chunk<T>::unpack(dv, d);
        const T* xb = X + (long)bb * p.xsb + c;
        // all loads of the pass first (predicated, zero outside the map), then the FMAs: with the load inside the bounds test every
        // tap waited for the previous one (the 7x7 layers spent 150 us on 13 MB)
        uint4 raw[RP][KW];
#pragma unroll
        for (int r = 0; r < RP; ++r) {
          const int hh = hh0 + r0 + r;
          const bool okh = r0 + r < kh && hh >= 0 && hh < p.H;
#pragma unroll
          for (int q = 0; q < KW; ++q) {
            const int ww = ww0 + q;
            const bool ok = okh && q < kw && ww >= 0 && ww < p.W;
            raw[r][q] = ok ? *(const uint4*)(xb + (long)hh * p.xsh + (long)ww * p.xsw) : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < RP; ++r)
#pragma unroll
          for (int q = 0; q < KW; ++q) {
            float v[CE];
            Chunk<T>::unpack(raw[r][q], v);
            pk_fma<CE>(acc[r][q], d, v);
          }
      };
      auto advance = [&]() {
        wo += PT;
        while (wo >= p.Wo) { wo -= p.Wo; ++ho; }
        while (ho >= p.Ho) { ho -= p.Ho; ++b; }
      };
      for (; m + PT < pend; m += 2 * PT) {
        const uint4 d0 = *(const uint4*)(D + m * p.dsw + c), d1 = *(const uint4*)(D + (m + PT) * p.dsw + c);
        const int b0 = b, h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
        advance();
        const int b1 = b, h1 = ho * p.stride - p.pad, w1 = wo * p.stride - p.pad;
        advance();
        one(d0, b0, h0, w0);
        one(d1, b1, h1, w1);
      }
      if (m < pend) one(*(const uint4*)(D + m * p.dsw + c), b, ho * p.stride - p.pad, wo * p.stride - p.pad);
    }
    // fold the 32 pixel rows of the block: 8 of them sit in one wave (lanes 8 apart) -> shuffles; the 4 waves meet in LDS once per pass
    // (the first version folded tap by tap through LDS: 2 barriers + a 32-step loop per tap, ~50 us of the 7x7 layers' 150)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < RP; ++r)
#pragma unroll
      for (int q = 0; q < KW; ++q)
#pragma unroll
        for (int j = 0; j < CE; ++j) {
          float v = acc[r][q][j];
          if (CT == 8) v = lane_xor8_sum(v);  // lane = (pixel row % (64 / CT)) * CT + channel chunk: fold the pixel rows of the wave
          v = lane_xor32_sum(lane_xor16_sum(v));
          if (lane < CT) sh[wave][(r * KW + q) * 64 + ct * CE + j] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < RP * KW * 64; i += 256) {
      const int t = i >> 6, cc = i & 63;
      const int r = t / KW, q = t - r * KW;
      if (r0 + r < kh && q < kw && cs + cc < p.C)
        p.slab[((long)bx * taps + (r0 + r) * kw + q) * p.C + cs + cc] = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
    }
  }
}

// slab[nblk][taps][C] -> grad OIHW [C][1][kh][kw]; 8 lanes share the block loop of an element and fold through LDS
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nblk, int taps, int C,
                                                              int accumulate) {
  __shared__ float sh[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + e;
  const int n = taps * C;
  float s = 0.f;
  if (idx < n)
    for (int b = sl; b < nblk; b += 8) s += slab[(long)b * n + idx];
  sh[sl][e] = s;
  __syncthreads();
  if (sl != 0 || idx >= n) return;
#pragma unroll
  for (int j = 1; j < 8; ++j) s += sh[j][e];
  const int c = idx % C, t = idx / C;
  const long o = (long)c * taps + t;
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

// 64 pixels (two per thread) per block: the stencil loop is a chain of load rounds (one per filter row), so small maps need many
// blocks to cover the latency -- with 256 pixels per block the 7x7 layers @20x20 ran 200 blocks on 256 CUs at 0.08 TB/s
int y3d_dw_blocks(int64_t M) {
  long n = (M + 63) / 64;
  return (int)(n < 1 ? 1 : (n > 2048 ? 2048 : n));
}

// weight gradient: every block ends with a slab of taps x C partial sums, so fewer, longer blocks
int y3d_dw_wgrad_blocks(int64_t M) {
  long n = (M + 127) / 128;
  return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}

int y3d_dw_pack_weight(const float* w_oihw, float* out, int C, int kh, int kw, void* stream) {
  hipLaunchKernelGGL(dw_pack_kernel, dim3(cdiv((long)C * kh * kw, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, out, C, kh * kw);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

#define DW_LAUNCH_K(T, D, K) hipLaunchKernelGGL((dwconv_kernel<T, D, K>), grid, dim3(256), sm, st, p)
#define DW_LAUNCH_A(T, K) hipLaunchKernelGGL((dwconv_kernel<T, false, K, true>), grid, dim3(256), sm, st, p)
#define DW_LAUNCH(T)                                                                                             \
  do {                                                                                                           \
    const int kk = (p.kh == p.kw && (p.kh == 3 || p.kh == 7)) ? p.kh : 0;                                        \
    if (!dgrad && p.scale) { if (kk == 3) DW_LAUNCH_A(T, 3); else if (kk == 7) DW_LAUNCH_A(T, 7); else DW_LAUNCH_A(T, 0); }           \
    else if (dgrad) { if (kk == 3) DW_LAUNCH_K(T, true, 3); else if (kk == 7) DW_LAUNCH_K(T, true, 7); else DW_LAUNCH_K(T, true, 0); }     \
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
  p.scale = nullptr; p.shift = nullptr; p.res = nullptr; p.rsw = 0; p.act = 0; p.res_mode = 0;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.Hg = H; p.Wg = W; p.Hq = Ho; p.Wq = Wo; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.M = (long)B * Ho * Wo;
  int nblk = y3d_dw_blocks(p.M);
  p.px_per_block = (int)((p.M + nblk - 1) / nblk);
  return dw_launch(dtype, false, p, (hipStream_t)stream);
}

int y3d_dwconv2d_fwd_affine(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int C,
                            const float* w_packed, const float* scale, const float* shift, int act, int res_mode, const void* res, int64_t rsw,
                            void* z, int64_t zsw, int Ho, int Wo, int kh, int kw, int stride, int pad, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "dwconv2d_fwd_affine: bad dtype");
  Y3D_CHECK(C % ce == 0, "dwconv2d_fwd_affine: C=%d not a multiple of %d", C, ce);
  Y3D_CHECK(Ho == (H + 2 * pad - kh) / stride + 1 && Wo == (W + 2 * pad - kw) / stride + 1, "dwconv2d_fwd_affine: output dims");
  Y3D_CHECK((((uintptr_t)x | (uintptr_t)z) & 15) == 0 && xsb % ce == 0 && xsh % ce == 0 && xsw % ce == 0 && zsw % ce == 0, "dwconv2d_fwd_affine: alignment");
  Y3D_CHECK(scale && shift && res_mode >= 0 && res_mode <= 2, "dwconv2d_fwd_affine: scale / shift missing or bad residual mode");
  Y3D_CHECK(res_mode == 0 || (res && (((uintptr_t)res) & 15) == 0 && rsw % ce == 0), "dwconv2d_fwd_affine: residual missing or misaligned");
  DwP p;
  p.x = x; p.w = w_packed; p.y = z; p.part = nullptr;
  p.scale = scale; p.shift = shift; p.res = res; p.rsw = rsw; p.act = act; p.res_mode = res_mode;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = zsw;
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
  p.scale = nullptr; p.shift = nullptr; p.res = nullptr; p.rsw = 0; p.act = 0; p.res_mode = 0;
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
  int nblk = y3d_dw_wgrad_blocks(p.M);
  p.px_per_block = (int)((p.M + nblk - 1) / nblk);
  p.nblk = nblk; p.nslab = cdiv(C, 64);
  dim3 grid(cdiv(nblk * p.nslab, 8) * 8 * kh);
  hipStream_t st = (hipStream_t)stream;
  Y3D_CHECK(kw <= 8 || (kh == kw && (kh == 3 || kh == 7)), "dwconv2d_bwd_weight: kernels wider than 8 taps are not supported");
#define Y3D_DWW(T)                                                                                                          \
  do {                                                                                                                      \
    if (kh == 3 && kw == 3) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 3>), grid, dim3(256), 0, st, p);                      \
    else if (kh == 7 && kw == 7) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 7>), grid, dim3(256), 0, st, p);                 \
    else hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 0>), grid, dim3(256), 0, st, p);                                         \
  } while (0)
  if (dtype == Y3D_BF16) Y3D_DWW(bf16_t); else Y3D_DWW(float);
#undef Y3D_DWW
  Y3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(cdiv((long)C * kh * kw, 32)), dim3(256), 0, st, slab, grad_oihw, nblk, kh * kw, C, accumulate);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
