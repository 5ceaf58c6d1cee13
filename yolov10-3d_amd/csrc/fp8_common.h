// OCP-MX block quantisation helpers shared by the activation quantiser (conv3x3_fp8.hip) and the BatchNorm + SiLU pass that writes the
// fp8 copy of its output next to the bf16 one (bn_act.hip): e4m3 codes + one E8M0 scale byte per 32 channels.
#pragma once
#include "common.h"

// bytes per pixel of the E8M0 scale tensor of a C-channel activation: C / 32 scale bytes, padded to whole dwords (the convolution's
// LDS-DMA fetches a pixel's scales as aligned dwords; C = 320: 10 -> 12)
__host__ __device__ inline int fp8_scale_pitch(int C) { return ((C >> 5) + 3) & ~3; }

// maximum over the four lanes of a quad (a 32-channel block = four threads of 8 channels): two quad-permute DPP ops
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
  return v;
}

// E8M0 byte of the smallest power of two s with amax / s <= 448 (448 = 1.75 * 2^8); amax == 0 -> 127 (s = 1)
__device__ __forceinline__ int mx_scale_byte(float amax) {
  const unsigned bits = __float_as_uint(amax);
  if ((bits & 0x7fffffffu) == 0) return 127;
  int e = (int)((bits >> 23) & 255) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);  // biased: (E + 127) - 8 (+1 when the mantissa exceeds 1.75)
  return e < 0 ? 0 : (e > 254 ? 254 : e);
}

// this thread's 8 channels f[0..7] of a 32-channel block (the quad's other three threads hold the rest) -> 8 e4m3 codes in (lo, hi), returns
// the block's scale byte.  Must be called by all four lanes of the quad.
__device__ __forceinline__ int mx_quantize8(const float* f, unsigned& lo, unsigned& hi) {
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
  amax = quad_max(amax);
  const int sbyte = mx_scale_byte(amax);
  const float inv = __uint_as_float((unsigned)(254 - sbyte) << 23);  // 2^-(sbyte - 127)
  float g[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = fminf(fmaxf(f[k] * inv, -448.f), 448.f);
  lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(g[0], g[1], 0, false);
  lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(g[2], g[3], (int)lo, true);
  hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(g[4], g[5], 0, false);
  hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(g[6], g[7], (int)hi, true);
  return sbyte;
}
