// fp8 (OCP e4m3fn) conv weights with per-output-channel power-of-two scales — BASELINE.json configs[4] ("fp8 MFMA weights").
//
// The reference has no 8-bit path; this is the build's own weight format (SURVEY §7 "Hard parts"):
//   scale[co] = 2^ceil(log2(absmax(w[co, :]) / 448))   (1 for an all-zero row; 448 = largest finite e4m3fn value)
//   code[co, k] = e4m3fn(w[co, k] / scale[co])          round-to-nearest-even, saturating at +-448
//   w_eff[co, k] = value(code) * scale[co]
// A power-of-two scale (the MX formats' E8M0 choice) makes w_eff EXACTLY representable in bf16: an e4m3 value has 4 significant
// bits, bf16 keeps 8, and the scale only moves the exponent.  The matrix cores therefore compute "fp8 weight x bf16 activation"
// products exactly when they are fed w_eff as a bf16 operand (v_mfma_f32_16x16x32_bf16 multiplies exactly and accumulates in fp32):
// the result is bit-identical to a mixed fp8 x bf16 MFMA, which the ISA does not have (its fp8 forms need BOTH operands in fp8, at the
// bf16 rate; only the block-scaled K=128 forms run faster, and they need fp8 activations).  What the format buys today is the
// 1-byte weight store (checkpoints / serving: `codes` + `scale`), with the training semantics of quantisation-aware fp8 weights:
// forward and data gradient see w_eff, the weight gradient goes to the fp32 master (straight-through).
#include "common.h"

namespace {

__device__ __forceinline__ float fp8w_scale_of(float amax) {
  if (!(amax > 0.f)) return 1.f;
  // smallest power of two s with amax / s <= 448
  int e;
  const float m = frexpf(amax / 448.f, &e);  // amax / 448 = m * 2^e, m in [0.5, 1)
  return ldexpf(1.f, m == 0.5f ? e - 1 : e);
}

// RNE to e4m3fn through the hardware conversion (v_cvt_pk_fp8_f32), input clamped to the finite range first
__device__ __forceinline__ unsigned fp8w_encode(float x) {
  x = fminf(fmaxf(x, -448.f), 448.f);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xffu;
}
__device__ __forceinline__ float fp8w_decode(unsigned code) { return __builtin_amdgcn_cvt_f32_fp8((int)code, 0); }

// one block per output-channel row of one weight: desc[t] = {src fp32 (rows, K), w_eff fp32 or 0, codes uint8 or 0, scale fp32 (rows), rows, K}
__global__ __launch_bounds__(256) void mt_fp8w_kernel(const long* __restrict__ desc, const int* __restrict__ row_begin, int nt) {
  __shared__ float red[256];
  __shared__ float s_scale;
  int lo = 0, hi = nt - 1;
  const int row_g = blockIdx.x;
  while (lo < hi) {  // largest t with row_begin[t] <= row_g
    const int mid = (lo + hi + 1) >> 1;
    if (row_begin[mid] <= row_g) lo = mid; else hi = mid - 1;
  }
  const long* d = desc + (long)lo * 6;
  const int row = row_g - row_begin[lo];
  const int K = (int)d[5];
  const float* __restrict__ w = (const float*)d[0] + (long)row * K;
  float* __restrict__ weff = d[1] ? (float*)d[1] + (long)row * K : nullptr;
  unsigned char* __restrict__ codes = d[2] ? (unsigned char*)d[2] + (long)row * K : nullptr;
  float amax = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(w[k]));
  red[threadIdx.x] = amax;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    s_scale = fp8w_scale_of(red[0]);
    ((float*)d[3])[row] = s_scale;
  }
  __syncthreads();
  const float sc = s_scale;
  for (int k = threadIdx.x; k < K; k += 256) {
    const unsigned c = fp8w_encode(w[k] / sc);
    if (codes) codes[k] = (unsigned char)c;
    if (weff) weff[k] = fp8w_decode(c) * sc;
  }
}

__global__ __launch_bounds__(256) void fp8w_dequant_kernel(const unsigned char* __restrict__ codes, const float* __restrict__ scale, float* __restrict__ w,
                                                           long n, int K) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) w[i] = fp8w_decode(codes[i]) * scale[i / K];
}

}  // namespace

extern "C" {

int y3d_mt_fp8w_quantize(const int64_t* desc, const int* row_begin, int ntensors, int nrows, void* stream) {
  Y3D_CHECK(desc && row_begin && ntensors >= 1 && nrows >= 1, "mt_fp8w_quantize: empty table");
  hipLaunchKernelGGL(mt_fp8w_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, (const long*)desc, row_begin, ntensors);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_fp8w_dequantize(const uint8_t* codes, const float* scale, float* w, int rows, int K, void* stream) {
  Y3D_CHECK(codes && scale && w && rows >= 1 && K >= 1, "fp8w_dequantize: bad arguments");
  const long n = (long)rows * K;
  long nb = (n + 255) / 256;
  hipLaunchKernelGGL(fp8w_dequant_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, codes, scale, w, n, K);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
