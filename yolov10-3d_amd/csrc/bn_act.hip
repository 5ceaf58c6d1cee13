// BatchNorm (training statistics or running statistics) + SiLU + residual, NHWC, HBM-bound.
// Reference semantics: nn/modules/conv.py:120-122 (act(bn(conv(x)))), BN eps/momentum from
// utils/torch_utils.py:327-337, residual adds block.py:342,758,816-817, RepVGGDW block.py:711.
//
//   u = y*scale[c] + shift[c] (+ res if RES_PRE);  z = act(u) (+ res if RES_POST)
//
// Every thread owns one 16-byte channel chunk of one pixel; per-channel reductions are two-level
// (per-block partial slabs, then a finalize kernel) so results are bitwise reproducible.
#include "common.h"
#include "fp8_common.h"

namespace {

// partial[nblk][C][2] (sum, sumsq) -> mean/invstd/scale/shift (+ running stats update)
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ run_mean,
                                   float* __restrict__ run_var, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
  // 8 channels x 32 row-threads per block: the partial slab has up to ceil(B*H*W/128) rows
  __shared__ double sh[32][8][2];
  int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  int c = blockIdx.x * 8 + cx;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // four rows in flight per thread (the loop is latency-bound: up to 6 400 partial rows for a 160x160 map); the summation
    // order stays fixed, so the result is still bitwise reproducible
    int b = ry;
    for (; b + 96 < nblk; b += 128) {
      float2 v0 = *(const float2*)(part + ((long)b * C + c) * 2);
      float2 v1 = *(const float2*)(part + ((long)(b + 32) * C + c) * 2);
      float2 v2 = *(const float2*)(part + ((long)(b + 64) * C + c) * 2);
      float2 v3 = *(const float2*)(part + ((long)(b + 96) * C + c) * 2);
      s += (double)v0.x; q += (double)v0.y;
      s += (double)v1.x; q += (double)v1.y;
      s += (double)v2.x; q += (double)v2.y;
      s += (double)v3.x; q += (double)v3.y;
    }
    for (; b < nblk; b += 32) {
      float2 v = *(const float2*)(part + ((long)b * C + c) * 2);
      s += (double)v.x;
      q += (double)v.y;
    }
  }
  sh[ry][cx][0] = s;
  sh[ry][cx][1] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    for (int r = 1; r < 32; ++r) { s += sh[r][cx][0]; q += sh[r][cx][1]; }
    double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float sc = gamma[c] * invstd;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    scale_out[c] = sc;
    shift_out[c] = beta[c] - (float)mean * sc;
    if (run_mean) {
      double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
  }
}

// eval: scale/shift from the running statistics
__global__ void bn_eval_scale_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps,
                                     float* __restrict__ scale_out, float* __restrict__ shift_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sc = gamma[c] / sqrtf(run_var[c] + eps);
  scale_out[c] = sc;
  shift_out[c] = beta[c] - run_mean[c] * sc;
}

template <typename T, int ACT, int RES>  // RES: 0 none, 1 post-activation, 2 pre-activation
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, long ysw, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const T* __restrict__ res, long rsw, T* __restrict__ z, long zsw, long P,
                                  int C, int px_per_block, int slabw) {
  // channel-stationary: a block owns a slab of `slabw` channels (64, or up to the whole pixel row: slab_width) and a run of pixels,
  // scale/shift live in registers, two pixels in flight
  constexpr int CE = TT<T>::CE;
  // a slab narrower than 64 channels (C = 32, or the last slab of C = 96) spends its spare chunk-threads on more pixels: with a
  // fixed 8 chunk-threads the 32-channel tensors of the first two stages ran with half of every wave idle (2.8-3.8 TB/s)
  const int left = (C - (int)blockIdx.y * slabw) / CE;
  const int CT = left < slabw / CE ? left : slabw / CE, PT = 256 / CT;
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int c = blockIdx.y * slabw + ct * CE;
  if (pt >= PT) return;
  float sc[CE], sf[CE];
#pragma unroll
  for (int j = 0; j < CE; ++j) { sc[j] = scale[c + j]; sf[j] = shift[c + j]; }
  const long pbeg = (long)blockIdx.x * px_per_block;
  const long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
  auto one = [&](long px, const uint4& yv, const uint4& rv) {
    float v[CE], r[CE];
    Chunk<T>::unpack(yv, v);
    if (RES) Chunk<T>::unpack(rv, r);
#pragma unroll
    for (int j = 0; j < CE; ++j) {
      float u = v[j] * sc[j] + sf[j];
      if (RES == 2) u += r[j];
      if (ACT) u = silu_f(u);
      if (RES == 1) u += r[j];
      v[j] = u;
    }
    *(uint4*)(z + px * zsw + c) = Chunk<T>::pack(v);
  };
  long px = pbeg + pt;
  for (; px + PT < pend; px += 2 * PT) {
    const uint4 y0 = *(const uint4*)(y + px * ysw + c), y1 = *(const uint4*)(y + (px + PT) * ysw + c);
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (RES) { r0 = *(const uint4*)(res + px * rsw + c); r1 = *(const uint4*)(res + (px + PT) * rsw + c); }
    one(px, y0, r0);
    one(px + PT, y1, r1);
  }
  if (px < pend) {
    uint4 r0 = make_uint4(0, 0, 0, 0);
    if (RES) r0 = *(const uint4*)(res + px * rsw + c);
    one(px, *(const uint4*)(y + px * ysw + c), r0);
  }
}

// the forward pass with a SECOND output: the fp8 (OCP-MX) copy of z that the fp8 MFMA convolution of the next layer reads - e4m3 codes
// q[pixel][C] + one E8M0 byte per 32 channels s[pixel][C / 32], quantised from the bf16-rounded z (so that q is exactly what
// y3d_fp8_quantize_act makes of the z tensor).  The 839 MB head activation is then read once less per step than with a separate
// quantising pass, and written 1.5x.  bf16, C % 64 == 0, no residual.
template <int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_q_kernel(const bf16_t* __restrict__ y, long ysw, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, bf16_t* __restrict__ z, long zsw,
                                                           unsigned char* __restrict__ q, unsigned char* __restrict__ s, long P, int C, int px_per_block) {
  const int ct = threadIdx.x & 7, pt = threadIdx.x >> 3;  // 8 chunk-threads per 64-channel slab, 32 pixel-threads
  const int c = blockIdx.y * 64 + ct * 8;
  float sc[8], sf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[c + j]; sf[j] = shift[c + j]; }
  const long pbeg = (long)blockIdx.x * px_per_block;
  const long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
  const int CS = fp8_scale_pitch(C);
  // every quad runs the same number of iterations (the pixel index depends on threadIdx.x >> 3 only): the DPP folds see all four lanes
  for (long px = pbeg + pt; px < pend; px += 32) {
    float v[8];
    Chunk<bf16_t>::unpack(*(const uint4*)(y + px * ysw + c), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = v[j] * sc[j] + sf[j];
      if (ACT) u = silu_f(u);
      v[j] = u;
    }
    const uint4 zz = Chunk<bf16_t>::pack(v);
    *(uint4*)(z + px * zsw + c) = zz;
    Chunk<bf16_t>::unpack(zz, v);  // the values as stored
    unsigned lo, hi;
    const int sbyte = mx_quantize8(v, lo, hi);
    *(uint2*)(q + px * C + c) = make_uint2(lo, hi);
    if ((ct & 3) == 0) s[px * CS + (c >> 5)] = (unsigned char)sbyte;
  }
}

// backward pass 1: g = dz * act'(u); per-block partial sums of g and g*xhat.
// Block = 256 threads = (64/CE... ) organised as CT chunk-threads x PT pixel-threads over a 64-channel slab.
template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* __restrict__ y, long ysw, const T* __restrict__ dz, long dsw,
                                         const T* __restrict__ res, long rsw, const float* __restrict__ scale,
                                         const float* __restrict__ shift, const float* __restrict__ mean,
                                         const float* __restrict__ invstd, float* __restrict__ part, long P, int C, int px_per_block,
                                         int slabw) {
  constexpr int CE = TT<T>::CE;
  // chunk-threads per slab: slabw / CE, fewer when the slab is narrower (see bn_act_fwd_kernel); the rest are pixel-threads
  const int left = (C - (int)blockIdx.y * slabw) / CE;
  const int CT = left < slabw / CE ? left : slabw / CE, PT = 256 / CT, SW = CT * CE;
  __shared__ float sh[256 * CE * 2];  // [PT][SW][2]: PT * SW <= 256 * CE
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int c = blockIdx.y * slabw + ct * CE;
  float s1[CE], s2[CE];
#pragma unroll
  for (int j = 0; j < CE; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (pt < PT) {
    float sc[CE], sf[CE], mu[CE], is[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) { sc[j] = scale[c + j]; sf[j] = shift[c + j]; mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
    long pbeg = (long)blockIdx.x * px_per_block;
    long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
    auto one = [&](const uint4& yv, const uint4& dv, const uint4& rv) {
      float v[CE], d[CE], r[CE];
      Chunk<T>::unpack(yv, v);
      Chunk<T>::unpack(dv, d);
      if (RES == 2) Chunk<T>::unpack(rv, r);
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        float g = d[j];
        if (ACT) {
          float u = v[j] * sc[j] + sf[j];
          if (RES == 2) u += r[j];
          g *= silu_grad_f(u);
        }
        s1[j] += g;
        s2[j] += g * (v[j] - mu[j]) * is[j];
      }
    };
    long px = pbeg + pt;
    for (; px + PT < pend; px += 2 * PT) {  // two pixels in flight
      const uint4 y0 = *(const uint4*)(y + px * ysw + c), y1 = *(const uint4*)(y + (px + PT) * ysw + c);
      const uint4 d0 = *(const uint4*)(dz + px * dsw + c), d1 = *(const uint4*)(dz + (px + PT) * dsw + c);
      uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
      if (RES == 2) { r0 = *(const uint4*)(res + px * rsw + c); r1 = *(const uint4*)(res + (px + PT) * rsw + c); }
      one(y0, d0, r0);
      one(y1, d1, r1);
    }
    if (px < pend) {
      uint4 r0 = make_uint4(0, 0, 0, 0);
      if (RES == 2) r0 = *(const uint4*)(res + px * rsw + c);
      one(*(const uint4*)(y + px * ysw + c), *(const uint4*)(dz + px * dsw + c), r0);
    }
  }
  if (pt < PT) {
#pragma unroll
    for (int j = 0; j < CE; ++j) { sh[(pt * SW + ct * CE + j) * 2] = s1[j]; sh[(pt * SW + ct * CE + j) * 2 + 1] = s2[j]; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < SW; t += 256) {
    int cc = blockIdx.y * slabw + t;
    if (cc < C) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < PT; ++r) { a += sh[(r * SW + t) * 2]; b += sh[(r * SW + t) * 2 + 1]; }
      *(float2*)(part + ((long)blockIdx.x * C + cc) * 2) = make_float2(a, b);
    }
  }
}

// partial[nblk][C][2] -> dgamma (sum g*xhat), dbeta (sum g), and the two per-channel means used by pass 2
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int accumulate, float* __restrict__ mg, float* __restrict__ mgx) {
  __shared__ double sh[32][8][2];
  int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  int c = blockIdx.x * 8 + cx;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // four rows in flight per thread, fixed summation order (as bn_finalize_kernel): up to 2048 partial rows
    int b = ry;
    for (; b + 96 < nblk; b += 128) {
      const float2 v0 = *(const float2*)(part + ((long)b * C + c) * 2), v1 = *(const float2*)(part + ((long)(b + 32) * C + c) * 2);
      const float2 v2 = *(const float2*)(part + ((long)(b + 64) * C + c) * 2), v3 = *(const float2*)(part + ((long)(b + 96) * C + c) * 2);
      s += (double)v0.x; q += (double)v0.y;
      s += (double)v1.x; q += (double)v1.y;
      s += (double)v2.x; q += (double)v2.y;
      s += (double)v3.x; q += (double)v3.y;
    }
    for (; b < nblk; b += 32) {
      const float2 v = *(const float2*)(part + ((long)b * C + c) * 2);
      s += (double)v.x;
      q += (double)v.y;
    }
  }
  sh[ry][cx][0] = s;
  sh[ry][cx][1] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    for (int r = 1; r < 32; ++r) { s += sh[r][cx][0]; q += sh[r][cx][1]; }
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    mg[c] = (float)(s / count);
    mgx[c] = (float)(q / count);
  }
}

// backward pass 2: dy = scale * (g - mean(g) - xhat * mean(g*xhat))      [TRAIN]
//                  dy = scale * g                                         [eval / frozen stats]
// optionally also emits g itself (gradient of a pre-activation residual).
// Channel-stationary layout (as in pass 1): a block owns a 64-channel slab and a run of pixels; a thread keeps its chunk's six
// per-channel constants in registers and walks the pixels, two in flight.  (The first version re-read the constants per element
// and did a 64-bit division per chunk: 2.3 TB/s on the 839 MB head tensors against 4.2 TB/s for pass 1.)
template <typename T, int ACT, int RES, bool TRAIN>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ y, long ysw, const T* __restrict__ dz, long dsw,
                                        const T* __restrict__ res, long rsw, const float* __restrict__ scale,
                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                        const float* __restrict__ invstd, const float* __restrict__ mg,
                                        const float* __restrict__ mgx, T* __restrict__ dy, long dysw, T* __restrict__ dres,
                                        long drsw, long P, int C, int px_per_block, int slabw) {
  constexpr int CE = TT<T>::CE;
  const int left = (C - (int)blockIdx.y * slabw) / CE;
  const int CT = left < slabw / CE ? left : slabw / CE, PT = 256 / CT;  // chunk-threads per slab, pixel-threads (see bn_act_fwd_kernel)
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int c = blockIdx.y * slabw + ct * CE;
  if (pt >= PT) return;
  // dy = a * g + b * y + k   with  a = scale, b = -scale * invstd^2... kept explicit for exactness with the reference formula
  float sc[CE], sf[CE], mu[CE], is[CE], m1[CE], m2[CE];
#pragma unroll
  for (int j = 0; j < CE; ++j) {
    sc[j] = scale[c + j]; sf[j] = shift[c + j];
    if (TRAIN) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; m1[j] = mg[c + j]; m2[j] = mgx[c + j]; }
  }
  const long pbeg = (long)blockIdx.x * px_per_block;
  const long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
  auto one = [&](long px, const uint4& yv, const uint4& dv, const uint4& rv) {
    float v[CE], d[CE], r[CE], o[CE];
    Chunk<T>::unpack(yv, v);
    Chunk<T>::unpack(dv, d);
    if (RES == 2) Chunk<T>::unpack(rv, r);
#pragma unroll
    for (int j = 0; j < CE; ++j) {
      float g = d[j];
      if (ACT) {
        float u = v[j] * sc[j] + sf[j];
        if (RES == 2) u += r[j];
        g *= silu_grad_f(u);
      }
      d[j] = g;
      if (TRAIN) {
        float xh = (v[j] - mu[j]) * is[j];
        o[j] = sc[j] * (g - m1[j] - xh * m2[j]);
      } else {
        o[j] = sc[j] * g;
      }
    }
    *(uint4*)(dy + px * dysw + c) = Chunk<T>::pack(o);
    if (RES == 2 && dres) *(uint4*)(dres + px * drsw + c) = Chunk<T>::pack(d);
  };
  long px = pbeg + pt;
  for (; px + PT < pend; px += 2 * PT) {
    const uint4 y0 = *(const uint4*)(y + px * ysw + c), y1 = *(const uint4*)(y + (px + PT) * ysw + c);
    const uint4 d0 = *(const uint4*)(dz + px * dsw + c), d1 = *(const uint4*)(dz + (px + PT) * dsw + c);
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (RES == 2) { r0 = *(const uint4*)(res + px * rsw + c); r1 = *(const uint4*)(res + (px + PT) * rsw + c); }
    one(px, y0, d0, r0);
    one(px + PT, y1, d1, r1);
  }
  if (px < pend) {
    uint4 r0 = make_uint4(0, 0, 0, 0);
    if (RES == 2) r0 = *(const uint4*)(res + px * rsw + c);
    one(px, *(const uint4*)(y + px * ysw + c), *(const uint4*)(dz + px * dsw + c), r0);
  }
}

// column sums of a [P][C] tensor into partial[nblk][C][2] (second slot zero) — bias gradients
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, long xsw, float* __restrict__ part, long P, int C, int px_per_block) {
  __shared__ float sh[4][64];
  int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  int c = blockIdx.y * 64 + cx;
  float s = 0.f;
  long pbeg = (long)blockIdx.x * px_per_block;
  long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
  if (c < C)
    for (long px = pbeg + ry; px < pend; px += 4) s += TT<T>::ld(x + px * xsw + c);
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    s += sh[1][cx] + sh[2][cx] + sh[3][cx];
    part[((long)blockIdx.x * C + c) * 2] = s;
    part[((long)blockIdx.x * C + c) * 2 + 1] = 0.f;
  }
}

// Channels of a pixel row one workgroup covers.  64 (a wave instruction = eight 128-byte pieces, one pixel pitch apart) streams the narrow
// tensors at 5.5-6.5 TB/s, but the 2048-channel head tensors (839 MB at stride 8) at 4.1 / 4.2 TB/s forward / apply in a stand-alone probe:
// there a workgroup takes the WHOLE 4 KB row and walks consecutive pixels - one contiguous stream per workgroup - 5.0-5.4 / 5.1-5.4 TB/s
// (tools/probe/bn_stream_probe.cpp, profiles/r04_bn_stream_probe.txt; these kernels, tools/bn_bench.py ab, profiles/r04_bn_bench_ab.txt:
// forward 377 -> 339 us, apply 631 -> 499 us on the stride-8 tensor (548 -> 474 on another box), 98 -> 87 and 163 -> 127 us at stride 16; the reduce pass does not
// gain, y3d_bn_bwd_blocks).  bf16, C >= 512: the row, or the largest whole fraction of it that keeps >= 84 % of the 256 threads busy
// (chunk-threads x pixel-threads).
int g_wide_slabs = 1;

inline int slab_width(int dtype, int C) {
  if (!g_wide_slabs || dtype != Y3D_BF16 || C < 512) return 64;
  for (int k = 1; k <= 8; ++k) {
    if (C % k != 0 || (C / k) % 8 != 0) continue;
    const int ct = C / k / 8;
    if (ct > 256) continue;
    if (ct * (256 / ct) >= 216) return C / k;
  }
  return 64;
}

// ~8 workgroups per CU as (pixel runs) x (slabs); every pixel-thread of a run has at least two pixels
inline dim3 slab_grid(long P, int C, int* ppb, int slabw = 64, int ce = 8) {
  const int nslab = cdiv(C, slabw);
  const int ct = (C < slabw ? C : slabw) / ce, minrun = slabw == 64 ? 64 : 2 * (256 / ct);
  long npx = 2048 / nslab;
  if (npx < 1) npx = 1;
  if (npx > (P + minrun - 1) / minrun) npx = (P + minrun - 1) / minrun;
  *ppb = (int)((P + npx - 1) / npx);
  return dim3((unsigned)npx, nslab);
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

template <typename T, int ACT>
int launch_fwd(int res_mode, const void* y, long ysw, const float* scale, const float* shift, const void* res, long rsw,
               void* z, long zsw, long P, int C, hipStream_t st) {
  int ppb;
  const int sw = slab_width(TT<T>::CE == 8 ? Y3D_BF16 : Y3D_F32, C);
  dim3 g = slab_grid(P, C, &ppb, sw, TT<T>::CE), b(256);
  if (res_mode == 0) hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACT, 0>), g, b, 0, st, (const T*)y, ysw, scale, shift, (const T*)res, rsw, (T*)z, zsw, P, C, ppb, sw);
  else if (res_mode == 1) hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACT, 1>), g, b, 0, st, (const T*)y, ysw, scale, shift, (const T*)res, rsw, (T*)z, zsw, P, C, ppb, sw);
  else hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACT, 2>), g, b, 0, st, (const T*)y, ysw, scale, shift, (const T*)res, rsw, (T*)z, zsw, P, C, ppb, sw);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

extern "C" {

int y3d_set_bn_wide_slabs(int enable) {
  const int old = g_wide_slabs;
  g_wide_slabs = enable ? 1 : 0;
  return old;
}

int y3d_bn_finalize(const float* partials, int nblk, int C, int64_t count, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                    float* shift, void* stream) {
  Y3D_CHECK(nblk > 0 && C > 0 && count > 0, "bn_finalize: empty");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, partials, nblk, C, (double)count,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_bn_eval_scale(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, void* stream) {
  hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C, gamma, beta, running_mean,
                     running_var, eps, scale, shift);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

static int ew_check(const char* what, int dtype, const void* a, int64_t asw, int C) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "%s: bad dtype", what);
  Y3D_CHECK(C % ce == 0, "%s: C=%d not a multiple of %d", what, C, ce);
  Y3D_CHECK(a == nullptr || ((((uintptr_t)a) & 15) == 0 && asw % ce == 0), "%s: tensor not 16-byte aligned", what);
  return Y3D_OK;
}

int y3d_bn_act_fwd(int dtype, const void* y, int64_t ysw, const float* scale, const float* shift, int act, int res_mode,
                   const void* res, int64_t rsw, void* z, int64_t zsw, int64_t P, int C, void* stream) {
  if (ew_check("bn_act_fwd y", dtype, y, ysw, C) || ew_check("bn_act_fwd z", dtype, z, zsw, C) ||
      ew_check("bn_act_fwd res", dtype, res, rsw, C)) return Y3D_ERR_INVALID;
  Y3D_CHECK(res_mode == 0 || res != nullptr, "bn_act_fwd: residual missing");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) return act ? launch_fwd<bf16_t, 1>(res_mode, y, ysw, scale, shift, res, rsw, z, zsw, P, C, st)
                                    : launch_fwd<bf16_t, 0>(res_mode, y, ysw, scale, shift, res, rsw, z, zsw, P, C, st);
  return act ? launch_fwd<float, 1>(res_mode, y, ysw, scale, shift, res, rsw, z, zsw, P, C, st)
             : launch_fwd<float, 0>(res_mode, y, ysw, scale, shift, res, rsw, z, zsw, P, C, st);
}

int y3d_bn_act_fwd_q(const void* y, int64_t ysw, const float* scale, const float* shift, int act, void* z, int64_t zsw, uint8_t* q, uint8_t* s,
                     int64_t P, int C, void* stream) {
  if (ew_check("bn_act_fwd_q y", Y3D_BF16, y, ysw, C) || ew_check("bn_act_fwd_q z", Y3D_BF16, z, zsw, C)) return Y3D_ERR_INVALID;
  Y3D_CHECK(C % 64 == 0 && q && s && ((((uintptr_t)q) & 7) == 0), "bn_act_fwd_q: C = %d must be a multiple of 64, q 8-byte aligned", C);
  int ppb;
  dim3 g = slab_grid(P, C, &ppb), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (act) hipLaunchKernelGGL((bn_act_fwd_q_kernel<1>), g, b, 0, st, (const bf16_t*)y, (long)ysw, scale, shift, (bf16_t*)z, (long)zsw, q, s, (long)P, C, ppb);
  else hipLaunchKernelGGL((bn_act_fwd_q_kernel<0>), g, b, 0, st, (const bf16_t*)y, (long)ysw, scale, shift, (bf16_t*)z, (long)zsw, q, s, (long)P, C, ppb);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_bn_bwd_blocks(int64_t P, int C) {
  // pixel runs of >= 128 pixels; (runs) x (64-channel slabs) ~ 8 workgroups per CU, at most 2048 rows for the finalize pass.  (Runs of
  // >= 512 pixels left the 20x20 / 40x40 tensors with 200 workgroups of 16 dependent pixel steps each: 15-17 us for 13 MB, twice the
  // apply pass that writes as well.)
  // The reduce pass keeps 64-channel slabs on every tensor: row-wide slabs (slab_width) in 1024 runs measured 353 against 350 us on the 839 MB
  // head tensor and 172 against 159 us on 204 800 x 896 (tools/bn_bench.py ab) - its two read streams already run at 4.8 TB/s.
  long n = (P + 127) / 128;
  long cap = 2048 / cdiv(C, 64);
  if (cap < 64) cap = 64;
  if (n > cap) n = cap;
  return (int)(n < 1 ? 1 : n);
}

#define BWD_REDUCE(T, A, R)                                                                                                  \
  hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, A, R>), grid, dim3(256), 0, st, (const T*)y, ysw, (const T*)dz, dsw,        \
                     (const T*)res, rsw, scale, shift, mean, invstd, partials, (long)P, C, ppb, sw)

int y3d_bn_act_bwd_reduce(int dtype, const void* y, int64_t ysw, const void* dz, int64_t dsw, const void* res, int64_t rsw,
                          const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                          int res_mode, float* partials, int64_t P, int C, void* stream) {
  if (ew_check("bn_act_bwd_reduce y", dtype, y, ysw, C) || ew_check("bn_act_bwd_reduce dz", dtype, dz, dsw, C) ||
      ew_check("bn_act_bwd_reduce res", dtype, res, rsw, C)) return Y3D_ERR_INVALID;
  int nblk = y3d_bn_bwd_blocks(P, C);
  int ppb = (int)((P + nblk - 1) / nblk);
  const int sw = 64;  // see y3d_bn_bwd_blocks
  dim3 grid(nblk, cdiv(C, sw));
  hipStream_t st = (hipStream_t)stream;
  int r2 = (res_mode == 2 && act) ? 2 : 0;
  if (dtype == Y3D_BF16) {
    if (act) { if (r2) BWD_REDUCE(bf16_t, 1, 2); else BWD_REDUCE(bf16_t, 1, 0); }
    else BWD_REDUCE(bf16_t, 0, 0);
  } else {
    if (act) { if (r2) BWD_REDUCE(float, 1, 2); else BWD_REDUCE(float, 1, 0); }
    else BWD_REDUCE(float, 0, 0);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_bn_bwd_finalize(const float* partials, int nblk, int C, int64_t count, float* dgamma, float* dbeta, int accumulate,
                        float* mean_g, float* mean_gx, void* stream) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, partials, nblk, C,
                     (double)count, dgamma, dbeta, accumulate, mean_g, mean_gx);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

#define BWD_APPLY(T, A, R, TR)                                                                                               \
  hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, A, R, TR>), grid, dim3(256), 0, st, (const T*)y, ysw, (const T*)dz, dsw,    \
                     (const T*)res, rsw, scale, shift, mean, invstd, mean_g, mean_gx, (T*)dy, dysw, (T*)dres, drsw, (long)P, C, ppb, sw)
#define BWD_APPLY_T(T)                                                        \
  do {                                                                        \
    if (train) {                                                              \
      if (act) { if (r2) BWD_APPLY(T, 1, 2, true); else BWD_APPLY(T, 1, 0, true); } \
      else { if (r2) BWD_APPLY(T, 0, 2, true); else BWD_APPLY(T, 0, 0, true); }     \
    } else {                                                                  \
      if (act) { if (r2) BWD_APPLY(T, 1, 2, false); else BWD_APPLY(T, 1, 0, false); } \
      else { if (r2) BWD_APPLY(T, 0, 2, false); else BWD_APPLY(T, 0, 0, false); }     \
    }                                                                         \
  } while (0)

int y3d_bn_act_bwd_apply(int dtype, const void* y, int64_t ysw, const void* dz, int64_t dsw, const void* res, int64_t rsw,
                         const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* mean_g, const float* mean_gx, int act, int res_mode, int train, void* dy, int64_t dysw,
                         void* dres, int64_t drsw, int64_t P, int C, void* stream) {
  if (ew_check("bn_act_bwd_apply y", dtype, y, ysw, C) || ew_check("bn_act_bwd_apply dz", dtype, dz, dsw, C) ||
      ew_check("bn_act_bwd_apply dy", dtype, dy, dysw, C) || ew_check("bn_act_bwd_apply res", dtype, res, rsw, C) ||
      ew_check("bn_act_bwd_apply dres", dtype, dres, drsw, C)) return Y3D_ERR_INVALID;
  int r2 = res_mode == 2 ? 2 : 0;
  int ppb;
  const int sw = slab_width(dtype, C);
  dim3 grid = slab_grid(P, C, &ppb, sw, dtype == Y3D_BF16 ? 8 : 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) BWD_APPLY_T(bf16_t); else BWD_APPLY_T(float);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_colsum_partials(int dtype, const void* x, int64_t xsw, float* partials, int64_t P, int C, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "colsum: bad dtype");
  int nblk = y3d_bn_bwd_blocks(P, C);
  int ppb = (int)((P + nblk - 1) / nblk);
  dim3 grid(nblk, cdiv(C, 64));
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)xsw, partials, (long)P, C, ppb);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)xsw, partials, (long)P, C, ppb);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
