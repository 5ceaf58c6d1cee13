// HBM-bound glue of the YOLOv10 graph, NHWC, 16-byte chunks per lane:
//   * 5x5 stride-1 max pool + its backward (SPPF, block.py:171-178)
//   * nearest 2x upsample + backward (yaml `nn.Upsample(None, 2, "nearest")`)
//   * strided channel copy (Concat conv.py:404 / torch.cat in C2f, SPPF, PSA, heads)
//   * NCHW fp32 image -> NHWC compute dtype with channel padding (model boundary)
//   * small-Cout 1x1 projection with bias (head final nn.Conv2d, head.py:637) fwd / bwd
#include "common.h"

namespace {

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

// ---------------------------------------------------------------- max pool k x k, stride 1, pad k/2
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, long xsb, long xsh, long xsw, T* __restrict__ y, long ysw,
                                   unsigned char* __restrict__ arg, int B, int H, int W, int C, int k) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE, pad = k / 2;
  long total = (long)B * H * W * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    int b = (int)(px / (H * W));
    int rem = (int)(px - (long)b * H * W);
    int h = rem / W, w = rem - h * W;
    float best[CE];
    int bi[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    for (int r = 0; r < k; ++r) {
      int hh = h - pad + r;
      if (hh < 0 || hh >= H) continue;
      for (int q = 0; q < k; ++q) {
        int ww = w - pad + q;
        if (ww < 0 || ww >= W) continue;
        float v[CE];
        Chunk<T>::unpack(*(const uint4*)(x + (long)b * xsb + (long)hh * xsh + (long)ww * xsw + c), v);
#pragma unroll
        for (int j = 0; j < CE; ++j)
          if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = r * k + q; }  // first max wins (ATen max_pool2d order), NaN propagates
      }
    }
    *(uint4*)(y + px * ysw + c) = Chunk<T>::pack(best);
    if (arg) {
#pragma unroll
      for (int j = 0; j < CE; ++j) arg[px * C + c + j] = (unsigned char)bi[j];
    }
  }
}

// dx[p] = sum over output windows o containing p whose arg-max is p
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, long dsw, const unsigned char* __restrict__ arg,
                                   T* __restrict__ dx, long xsw, int B, int H, int W, int C, int k) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE, pad = k / 2;
  long total = (long)B * H * W * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    int b = (int)(px / (H * W));
    int rem = (int)(px - (long)b * H * W);
    int h = rem / W, w = rem - h * W;
    float acc[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) acc[j] = 0.f;
    for (int r = 0; r < k; ++r) {
      int ho = h + pad - r;  // output row whose tap r lands on h
      if (ho < 0 || ho >= H) continue;
      for (int q = 0; q < k; ++q) {
        int wo = w + pad - q;
        if (wo < 0 || wo >= W) continue;
        long po = ((long)b * H + ho) * W + wo;
        float d[CE];
        Chunk<T>::unpack(*(const uint4*)(dy + po * dsw + c), d);
        const unsigned char* a = arg + po * C + c;
#pragma unroll
        for (int j = 0; j < CE; ++j)
          if (a[j] == r * k + q) acc[j] += d[j];
      }
    }
    *(uint4*)(dx + px * xsw + c) = Chunk<T>::pack(acc);
  }
}

// ---------------------------------------------------------------- nearest 2x upsample
template <typename T>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ x, long xsb, long xsh, long xsw, T* __restrict__ y, long ysw, int B,
                                      int H, int W, int C) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE;
  long total = (long)B * (2 * H) * (2 * W) * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    int b = (int)(px / (4 * H * W));
    int rem = (int)(px - (long)b * 4 * H * W);
    int h = rem / (2 * W), w = rem - h * (2 * W);
    *(uint4*)(y + px * ysw + c) = *(const uint4*)(x + (long)b * xsb + (long)(h >> 1) * xsh + (long)(w >> 1) * xsw + c);
  }
}

template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dy, long dsb, long dsh, long dsw, T* __restrict__ dx, long xsw, int B,
                                      int H, int W, int C) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE;
  long total = (long)B * H * W * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    int b = (int)(px / (H * W));
    int rem = (int)(px - (long)b * H * W);
    int h = rem / W, w = rem - h * W;
    float acc[CE], v[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) acc[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        Chunk<T>::unpack(*(const uint4*)(dy + (long)b * dsb + (long)(2 * h + r) * dsh + (long)(2 * w + q) * dsw + c), v);
#pragma unroll
        for (int j = 0; j < CE; ++j) acc[j] += v[j];
      }
    *(uint4*)(dx + px * xsw + c) = Chunk<T>::pack(acc);
  }
}

// ---------------------------------------------------------------- strided [P][C] copy (concat slices)
template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ x, long xsw, T* __restrict__ y, long ysw, long P, int C) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE;
  long total = P * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    *(uint4*)(y + px * ysw + c) = *(const uint4*)(x + px * xsw + c);
  }
}

// y = a + b (grad accumulation of fan-out tensors), all [P][C] strided
template <typename T>
__global__ void add2d_kernel(const T* __restrict__ a, long asw, const T* __restrict__ b, long bsw, T* __restrict__ y, long ysw,
                             long P, int C) {
  constexpr int CE = TT<T>::CE;
  const int cpr = C / CE;
  long total = P * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    float u[CE], v[CE];
    Chunk<T>::unpack(*(const uint4*)(a + px * asw + c), u);
    Chunk<T>::unpack(*(const uint4*)(b + px * bsw + c), v);
#pragma unroll
    for (int j = 0; j < CE; ++j) u[j] += v[j];
    *(uint4*)(y + px * ysw + c) = Chunk<T>::pack(u);
  }
}

// ---------------------------------------------------------------- layout conversion at the model boundary
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int H, int W, int Cpad) {
  long total = (long)B * H * W * Cpad;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % Cpad);
    long px = idx / Cpad;
    int b = (int)(px / (H * W));
    int hw = (int)(px - (long)b * H * W);
    float v = c < C ? x[((long)b * C + c) * H * W + hw] : 0.f;
    TT<T>::st(y + idx, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, long xsw, float* __restrict__ y, int B, int C, int H, int W) {
  long total = (long)B * C * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int hw = (int)(idx % (H * W));
    long bc = idx / (H * W);
    int c = (int)(bc % C);
    int b = (int)(bc / C);
    y[idx] = TT<T>::ld(x + ((long)b * H * W + hw) * xsw + c);
  }
}

// ---------------------------------------------------------------- small-Cout projection (1x1 conv + bias)
// y[p][co] = b[co] + sum_ci x[p][ci] * w[co][ci],  Cout <= 24 per launch slice; 8 lanes per pixel
template <typename T, int CN>
__global__ __launch_bounds__(256) void proj_fwd_kernel(const T* __restrict__ x, long xsw, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y, long ysw, long P,
                                                       int Cin, int Cout) {
  constexpr int CE = TT<T>::CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sw = (float*)smem;  // [Cout][Cin]
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = w[i];
  __syncthreads();
  const int sub = threadIdx.x & 7;
  const int cpr = Cin / CE;
  long px = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  float acc[CN];
#pragma unroll
  for (int o = 0; o < CN; ++o) acc[o] = 0.f;
  if (px < P) {
    for (int ch = sub; ch < cpr; ch += 8) {
      float v[CE];
      Chunk<T>::unpack(*(const uint4*)(x + px * xsw + ch * CE), v);
#pragma unroll
      for (int o = 0; o < CN; ++o)
        if (o < Cout) {
          const float* wr = sw + o * Cin + ch * CE;
#pragma unroll
          for (int j = 0; j < CE; ++j) acc[o] += v[j] * wr[j];
        }
    }
  }
#pragma unroll
  for (int o = 0; o < CN; ++o) {
    acc[o] += __shfl_xor(acc[o], 1);
    acc[o] += __shfl_xor(acc[o], 2);
    acc[o] += __shfl_xor(acc[o], 4);
  }
  if (px < P && sub == 0) {
#pragma unroll
    for (int o = 0; o < CN; ++o)
      if (o < Cout) TT<T>::st(y + px * ysw + o, acc[o] + (bias ? bias[o] : 0.f));
  }
}

// dx[p][ci] = sum_co dy[p][co] * w[co][ci]
template <typename T, int CN>
__global__ __launch_bounds__(256) void proj_bwd_data_kernel(const T* __restrict__ dy, long dsw, const float* __restrict__ w,
                                                            T* __restrict__ dx, long xsw, long P, int Cin, int Cout) {
  constexpr int CE = TT<T>::CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sw = (float*)smem;
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = w[i];
  __syncthreads();
  const int cpr = Cin / CE;
  long total = P * cpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    float acc[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) acc[j] = 0.f;
#pragma unroll
    for (int o = 0; o < CN; ++o)
      if (o < Cout) {
        float d = TT<T>::ld(dy + px * dsw + o);
        const float* wr = sw + o * Cin + c;
#pragma unroll
        for (int j = 0; j < CE; ++j) acc[j] += d * wr[j];
      }
    *(uint4*)(dx + px * xsw + c) = Chunk<T>::pack(acc);
  }
}

// slab[blk][co][ci] = sum over the block's pixels of dy[p][co] * x[p][ci];  bias slab[blk][co] = sum dy[p][co]
template <typename T>
__global__ __launch_bounds__(256) void proj_bwd_weight_kernel(const T* __restrict__ x, long xsw, const T* __restrict__ dy, long dsw,
                                                              float* __restrict__ slab, float* __restrict__ bslab, long P,
                                                              int Cin, int Cout, int px_per_block) {
  constexpr int CE = TT<T>::CE;
  constexpr int CT = 64 / CE;   // chunk threads covering a 64-channel slab of Cin
  constexpr int PT = 256 / CT;
  constexpr int OG = 4;         // output channels per pass
  __shared__ float sh[PT][64];
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int cs = blockIdx.y * 64;
  const int c = cs + ct * CE;
  long pbeg = (long)blockIdx.x * px_per_block;
  long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
  for (int o0 = 0; o0 < Cout; o0 += OG) {
    float acc[OG][CE], bacc[OG];
#pragma unroll
    for (int o = 0; o < OG; ++o) {
      bacc[o] = 0.f;
#pragma unroll
      for (int j = 0; j < CE; ++j) acc[o][j] = 0.f;
    }
    if (c < Cin) {
      for (long px = pbeg + pt; px < pend; px += PT) {
        float v[CE];
        Chunk<T>::unpack(*(const uint4*)(x + px * xsw + c), v);
#pragma unroll
        for (int o = 0; o < OG; ++o)
          if (o0 + o < Cout) {
            float d = TT<T>::ld(dy + px * dsw + o0 + o);
            bacc[o] += d;
#pragma unroll
            for (int j = 0; j < CE; ++j) acc[o][j] += d * v[j];
          }
      }
    }
#pragma unroll
    for (int o = 0; o < OG; ++o) {
      if (o0 + o < Cout) {
#pragma unroll
        for (int j = 0; j < CE; ++j) sh[pt][ct * CE + j] = acc[o][j];
        __syncthreads();
        if (threadIdx.x < 64 && cs + threadIdx.x < Cin) {
          float a = 0.f;
          for (int r = 0; r < PT; ++r) a += sh[r][threadIdx.x];
          slab[((long)blockIdx.x * Cout + o0 + o) * Cin + cs + threadIdx.x] = a;
        }
        __syncthreads();
        if (bslab && blockIdx.y == 0) {  // bias: every chunk-thread column saw the same dy; use column ct == 0
          sh[pt][ct] = bacc[o];
          __syncthreads();
          if (threadIdx.x == 0) {
            float a = 0.f;
            for (int r = 0; r < PT; ++r) a += sh[r][0];
            bslab[(long)blockIdx.x * Cout + o0 + o] = a;
          }
          __syncthreads();
        }
      }
    }
  }
}

// out[i] (+)= sum_b slab[b][i]
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int nblk, long n, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += slab[(long)b * n + idx];
  out[idx] = accumulate ? out[idx] + s : s;
}

// Stem im2col: NCHW fp32 image (3 channels) -> [B][Ho][Wo][32] in the compute dtype, k = (r*3 + q)*3 + ci for the 27 taps x channels
// of a 3x3 stride-2 pad-1 conv, 5 zero columns.  The stem conv then runs as a dense 1x1 conv with K = 32 (forward, weight gradient)
// instead of a 9-tap conv over an 8-channel-padded image whose MFMA tiles are 86 % padding (measured 23-62 TFLOP/s).
template <typename T>
__global__ void stem_im2col_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
  constexpr int CE = TT<T>::CE;
  constexpr int CPP = 32 / CE;  // chunks per output pixel
  long total = (long)B * Ho * Wo * CPP;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % CPP);
    long pidx = i / CPP;
    const int ow = (int)(pidx % Wo);
    pidx /= Wo;
    const int oh = (int)(pidx % Ho);
    const int b = (int)(pidx / Ho);
    float v[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) {
      const int k = ch * CE + j;
      const int tap = k / 3, ci = k - tap * 3;
      const int r = tap / 3, q = tap - r * 3;
      const int ih = 2 * oh - 1 + r, iw = 2 * ow - 1 + q;
      v[j] = (k < 27 && ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[(((long)b * 3 + ci) * H + ih) * W + iw] : 0.f;
    }
    *(uint4*)(out + i * CE) = Chunk<T>::pack(v);
  }
}

// The same from the dataset's uint8 image, NCHW (hwc = 0) or NHWC as decoded (hwc = 1): v = u8 / 255 in fp32 -- the reference divides
// on the host (data/datasets/kitti.py:204-205: astype(float32) / 255) and ships 4 bytes per sample; bit-identical to that float image.
template <typename T>
__global__ void stem_im2col_u8_kernel(const unsigned char* __restrict__ x, int hwc, T* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
  constexpr int CE = TT<T>::CE;
  constexpr int CPP = 32 / CE;
  long total = (long)B * Ho * Wo * CPP;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % CPP);
    long pidx = i / CPP;
    const int ow = (int)(pidx % Wo);
    pidx /= Wo;
    const int oh = (int)(pidx % Ho);
    const int b = (int)(pidx / Ho);
    float v[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) {
      const int k = ch * CE + j;
      const int tap = k / 3, ci = k - tap * 3;
      const int r = tap / 3, q = tap - r * 3;
      const int ih = 2 * oh - 1 + r, iw = 2 * ow - 1 + q;
      float u = 0.f;
      if (k < 27 && ih >= 0 && ih < H && iw >= 0 && iw < W)
        u = (float)(hwc ? x[(((long)b * H + ih) * W + iw) * 3 + ci] : x[(((long)b * 3 + ci) * H + ih) * W + iw]) / 255.f;
      v[j] = u;
    }
    *(uint4*)(out + i * CE) = Chunk<T>::pack(v);
  }
}

}  // namespace

#define DISPATCH_T(dtype, KERNEL, grid, block, sm, st, ...)                                   \
  do {                                                                                        \
    if ((dtype) == Y3D_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, grid, block, sm, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, grid, block, sm, st, __VA_ARGS__);                 \
    Y3D_LAUNCH_CHECK();                                                                       \
  } while (0)


// diagnostic (tools/cu_contention.py): `n` workgroups that stay resident - one wave each polling `flag` between sleeps, the others
// copying inside `buf` at a trickle - until *flag != 0 or ~50 ms have passed (every wave reaches the exit: bounded loop)
__global__ __launch_bounds__(256) void occupy_cus_kernel(const int* __restrict__ flag, float* __restrict__ buf, long nbuf) {
  __shared__ int stop;
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  const long chunk = nbuf / (2 * (long)gridDim.x);
  float* src = buf + (long)blockIdx.x * 2 * chunk;
  float* dst = src + chunk;
  for (int it = 0; it < 50000; ++it) {  // ~1 us per round
    if (threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) stop = 1;
    const long o = ((long)it * 256 + threadIdx.x) % (chunk > 256 ? chunk - 256 : 1);
    dst[o] = src[o] + 1.f;
    __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (stop) break;
  }
}

static int chk(const char* what, int dtype, int C, const void* a, long s0, long s1, long s2) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "%s: bad dtype", what);
  Y3D_CHECK(C % ce == 0, "%s: C=%d not a multiple of %d", what, C, ce);
  Y3D_CHECK((((uintptr_t)a) & 15) == 0 && s0 % ce == 0 && s1 % ce == 0 && s2 % ce == 0, "%s: not 16-byte aligned", what);
  return Y3D_OK;
}

extern "C" {

int y3d_maxpool_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, void* y, int64_t ysw, uint8_t* argmax,
                    int B, int H, int W, int C, int k, void* stream) {
  if (chk("maxpool_fwd x", dtype, C, x, xsb, xsh, xsw) || chk("maxpool_fwd y", dtype, C, y, ysw, 0, 0)) return Y3D_ERR_INVALID;
  Y3D_CHECK(k % 2 == 1 && k <= 15, "maxpool_fwd: k");
  long total = (long)B * H * W * (C / (dtype == Y3D_BF16 ? 8 : 4));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)xsb, (long)xsh, (long)xsw, (bf16_t*)y, (long)ysw, argmax, B, H, W, C, k);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)xsb, (long)xsh, (long)xsw, (float*)y, (long)ysw, argmax, B, H, W, C, k);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_maxpool_bwd(int dtype, const void* dy, int64_t dsw, const uint8_t* argmax, void* dx, int64_t xsw, int B, int H, int W,
                    int C, int k, void* stream) {
  if (chk("maxpool_bwd dy", dtype, C, dy, dsw, 0, 0) || chk("maxpool_bwd dx", dtype, C, dx, xsw, 0, 0)) return Y3D_ERR_INVALID;
  long total = (long)B * H * W * (C / (dtype == Y3D_BF16 ? 8 : 4));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (long)dsw, argmax, (bf16_t*)dx, (long)xsw, B, H, W, C, k);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (long)dsw, argmax, (float*)dx, (long)xsw, B, H, W, C, k);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_upsample2x_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, void* y, int64_t ysw, int B, int H, int W,
                       int C, void* stream) {
  if (chk("upsample2x_fwd x", dtype, C, x, xsb, xsh, xsw) || chk("upsample2x_fwd y", dtype, C, y, ysw, 0, 0)) return Y3D_ERR_INVALID;
  long total = (long)B * 4 * H * W * (C / (dtype == Y3D_BF16 ? 8 : 4));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(upsample2x_fwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)xsb, (long)xsh, (long)xsw, (bf16_t*)y, (long)ysw, B, H, W, C);
  else hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)xsb, (long)xsh, (long)xsw, (float*)y, (long)ysw, B, H, W, C);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_upsample2x_bwd(int dtype, const void* dy, int64_t dsb, int64_t dsh, int64_t dsw, void* dx, int64_t xsw, int B, int H,
                       int W, int C, void* stream) {
  if (chk("upsample2x_bwd dy", dtype, C, dy, dsb, dsh, dsw) || chk("upsample2x_bwd dx", dtype, C, dx, xsw, 0, 0)) return Y3D_ERR_INVALID;
  long total = (long)B * H * W * (C / (dtype == Y3D_BF16 ? 8 : 4));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (long)dsb, (long)dsh, (long)dsw, (bf16_t*)dx, (long)xsw, B, H, W, C);
  else hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (long)dsb, (long)dsh, (long)dsw, (float*)dx, (long)xsw, B, H, W, C);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_occupy_cus(int n, const int* flag, float* buf, int64_t nbuf, void* stream) {
  Y3D_CHECK(n >= 1 && n <= 256 && flag && buf && nbuf >= 1024L * n, "occupy_cus: 1..256 workgroups, a flag and a buffer of >= 1024 floats per workgroup");
  hipLaunchKernelGGL(occupy_cus_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, flag, buf, (long)nbuf);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_copy2d(int dtype, const void* x, int64_t xsw, void* y, int64_t ysw, int64_t P, int C, void* stream) {
  if (chk("copy2d x", dtype, C, x, xsw, 0, 0) || chk("copy2d y", dtype, C, y, ysw, 0, 0)) return Y3D_ERR_INVALID;
  long total = P * (C / (dtype == Y3D_BF16 ? 8 : 4));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(copy2d_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)xsw, (bf16_t*)y, (long)ysw, (long)P, C);
  else hipLaunchKernelGGL(copy2d_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)xsw, (float*)y, (long)ysw, (long)P, C);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_add2d(int dtype, const void* a, int64_t asw, const void* b, int64_t bsw, void* y, int64_t ysw, int64_t P, int C, void* stream) {
  if (chk("add2d a", dtype, C, a, asw, 0, 0) || chk("add2d b", dtype, C, b, bsw, 0, 0) || chk("add2d y", dtype, C, y, ysw, 0, 0)) return Y3D_ERR_INVALID;
  long total = P * (C / (dtype == Y3D_BF16 ? 8 : 4));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(add2d_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (long)asw, (const bf16_t*)b, (long)bsw, (bf16_t*)y, (long)ysw, (long)P, C);
  else hipLaunchKernelGGL(add2d_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (long)asw, (const float*)b, (long)bsw, (float*)y, (long)ysw, (long)P, C);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_nchw_to_nhwc(int dtype, const float* x_nchw, void* y_nhwc, int B, int C, int H, int W, int Cpad, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "nchw_to_nhwc: bad dtype");
  Y3D_CHECK(Cpad >= C, "nchw_to_nhwc: Cpad < C");
  long total = (long)B * H * W * Cpad;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, (bf16_t*)y_nhwc, B, C, H, W, Cpad);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, (float*)y_nhwc, B, C, H, W, Cpad);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_stem_im2col(int dtype, const float* x_nchw, void* out, int B, int H, int W, int Ho, int Wo, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "stem_im2col: bad dtype");
  Y3D_CHECK(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "stem_im2col: output size of a 3x3 stride-2 pad-1 conv expected");
  long total = (long)B * Ho * Wo * (dtype == Y3D_BF16 ? 4 : 8);
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, (bf16_t*)out, B, H, W, Ho, Wo);
  else hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, (float*)out, B, H, W, Ho, Wo);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_stem_im2col_u8(int dtype, const uint8_t* x, int hwc, void* out, int B, int H, int W, int Ho, int Wo, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "stem_im2col_u8: bad dtype");
  Y3D_CHECK(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "stem_im2col_u8: output size of a 3x3 stride-2 pad-1 conv expected");
  long total = (long)B * Ho * Wo * (dtype == Y3D_BF16 ? 4 : 8);
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(stem_im2col_u8_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, hwc, (bf16_t*)out, B, H, W, Ho, Wo);
  else hipLaunchKernelGGL(stem_im2col_u8_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, hwc, (float*)out, B, H, W, Ho, Wo);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_nhwc_to_nchw(int dtype, const void* x_nhwc, int64_t xsw, float* y_nchw, int B, int C, int H, int W, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "nhwc_to_nchw: bad dtype");
  long total = (long)B * C * H * W;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_nhwc, (long)xsw, y_nchw, B, C, H, W);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x_nhwc, (long)xsw, y_nchw, B, C, H, W);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_fwd(int dtype, const void* x, int64_t xsw, const float* w, const float* bias, void* y, int64_t ysw, int64_t P,
                 int Cin, int Cout, void* stream) {
  if (chk("proj_fwd x", dtype, Cin, x, xsw, 0, 0)) return Y3D_ERR_INVALID;
  Y3D_CHECK(Cout >= 1 && Cout <= 24, "proj_fwd: Cout=%d must be in 1..24 per call", Cout);
  Y3D_CHECK((size_t)Cout * Cin * 4 <= 64 * 1024, "proj_fwd: weight slab too large for LDS");
  dim3 grid(cdiv(P, 32)), block(256);
  size_t sm = (size_t)Cout * Cin * 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) {
    if (Cout <= 4) hipLaunchKernelGGL((proj_fwd_kernel<bf16_t, 4>), grid, block, sm, st, (const bf16_t*)x, (long)xsw, w, bias, (bf16_t*)y, (long)ysw, (long)P, Cin, Cout);
    else hipLaunchKernelGGL((proj_fwd_kernel<bf16_t, 24>), grid, block, sm, st, (const bf16_t*)x, (long)xsw, w, bias, (bf16_t*)y, (long)ysw, (long)P, Cin, Cout);
  } else {
    if (Cout <= 4) hipLaunchKernelGGL((proj_fwd_kernel<float, 4>), grid, block, sm, st, (const float*)x, (long)xsw, w, bias, (float*)y, (long)ysw, (long)P, Cin, Cout);
    else hipLaunchKernelGGL((proj_fwd_kernel<float, 24>), grid, block, sm, st, (const float*)x, (long)xsw, w, bias, (float*)y, (long)ysw, (long)P, Cin, Cout);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_bwd_data(int dtype, const void* dy, int64_t dsw, const float* w, void* dx, int64_t xsw, int64_t P, int Cin, int Cout,
                      void* stream) {
  if (chk("proj_bwd_data dx", dtype, Cin, dx, xsw, 0, 0)) return Y3D_ERR_INVALID;
  Y3D_CHECK(Cout >= 1 && Cout <= 24, "proj_bwd_data: Cout=%d must be in 1..24 per call", Cout);
  size_t sm = (size_t)Cout * Cin * 4;
  long total = P * (Cin / (dtype == Y3D_BF16 ? 8 : 4));
  dim3 grid(ew_grid(total)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) {
    if (Cout <= 4) hipLaunchKernelGGL((proj_bwd_data_kernel<bf16_t, 4>), grid, block, sm, st, (const bf16_t*)dy, (long)dsw, w, (bf16_t*)dx, (long)xsw, (long)P, Cin, Cout);
    else hipLaunchKernelGGL((proj_bwd_data_kernel<bf16_t, 24>), grid, block, sm, st, (const bf16_t*)dy, (long)dsw, w, (bf16_t*)dx, (long)xsw, (long)P, Cin, Cout);
  } else {
    if (Cout <= 4) hipLaunchKernelGGL((proj_bwd_data_kernel<float, 4>), grid, block, sm, st, (const float*)dy, (long)dsw, w, (float*)dx, (long)xsw, (long)P, Cin, Cout);
    else hipLaunchKernelGGL((proj_bwd_data_kernel<float, 24>), grid, block, sm, st, (const float*)dy, (long)dsw, w, (float*)dx, (long)xsw, (long)P, Cin, Cout);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_blocks(int64_t P) {
  long n = (P + 511) / 512;
  return (int)(n < 1 ? 1 : (n > 128 ? 128 : n));
}

int y3d_proj_bwd_weight(int dtype, const void* x, int64_t xsw, const void* dy, int64_t dsw, float* slab, float* bias_slab,
                        float* grad_w, float* grad_b, int accumulate, int64_t P, int Cin, int Cout, void* stream) {
  if (chk("proj_bwd_weight x", dtype, Cin, x, xsw, 0, 0)) return Y3D_ERR_INVALID;
  int nblk = y3d_proj_blocks(P);
  int ppb = (int)((P + nblk - 1) / nblk);
  dim3 grid(nblk, cdiv(Cin, 64));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(proj_bwd_weight_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (long)xsw, (const bf16_t*)dy, (long)dsw, slab, bias_slab, (long)P, Cin, Cout, ppb);
  else hipLaunchKernelGGL(proj_bwd_weight_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (long)xsw, (const float*)dy, (long)dsw, slab, bias_slab, (long)P, Cin, Cout, ppb);
  Y3D_LAUNCH_CHECK();
  long n = (long)Cout * Cin;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, grad_w, nblk, n, accumulate);
  Y3D_LAUNCH_CHECK();
  if (bias_slab && grad_b) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, st, bias_slab, grad_b, nblk, (long)Cout, accumulate);
    Y3D_LAUNCH_CHECK();
  }
  return Y3D_OK;
}

int y3d_slab_reduce(const float* slab, float* out, int nblk, int64_t n, int accumulate, void* stream) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, slab, out, nblk, (long)n, accumulate);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
