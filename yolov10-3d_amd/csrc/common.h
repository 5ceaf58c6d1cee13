// Shared device/host helpers for the y3d HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/y3d.h"

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define Y3D_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int CE = 4;    // elements per 16-byte chunk
  static constexpr int BKE = 32;  // elements per 128-byte K row
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <> struct TT<bf16_t> {
  static constexpr int CE = 8;
  static constexpr int BKE = 64;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// 16-byte chunk <-> CE floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Chunk<bf16_t> {
  __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    uint4 u;
    u.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
    u.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
    u.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
    u.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
    return u;
  }
};

// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE `/` expands to ~10 VALU instructions (div_scale, rcp, four FMAs,
// div_fmas, div_fixup), which made the BatchNorm + SiLU kernels VALU-bound next to their 2-3 streams of HBM traffic
__device__ __forceinline__ float fast_sigmoid_f(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float silu_f(float u) { return u * fast_sigmoid_f(u); }
__device__ __forceinline__ float silu_grad_f(float u) {
  float s = fast_sigmoid_f(u);
  return s * (1.f + u * (1.f - s));
}

// v + v[lane ^ 16], v + v[lane ^ 32]: gfx950 row swaps on the VALU (v_permlane16_swap / v_permlane32_swap exchange the odd 16- / 32-lane
// groups of one register with the even groups of the other), v + v[lane ^ 8]: a DPP rotate inside the 16-lane row.  No LDS crossbar
// (ds_bpermute) traffic: the depth-wise weight-gradient epilogue folds 56 values per lane and was LDS-bound with shuffles.
__device__ __forceinline__ float lane_xor8_sum(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
}
__device__ __forceinline__ float lane_xor16_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_xor32_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane of the row gets the total.
// Four v_add_f32 with DPP operand swizzles (quad_perm xor-1, xor-2, row_half_mirror, row_mirror) - no LDS crossbar round trips.
__device__ __forceinline__ float wave_xor_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}

// host-side error plumbing (y3d_api.cpp)
void y3d_set_error(const char* fmt, ...);
#define Y3D_CHECK(cond, ...)          \
  do {                                \
    if (!(cond)) {                    \
      y3d_set_error(__VA_ARGS__);     \
      return Y3D_ERR_INVALID;         \
    }                                 \
  } while (0)
#define Y3D_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) {                                  \
      y3d_set_error("HIP launch: %s", hipGetErrorString(e_)); \
      return Y3D_ERR_HIP;                                    \
    }                                                        \
  } while (0)

#define Y3D_HIP(call)                                                 \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      y3d_set_error("%s: %s", #call, hipGetErrorString(e_));          \
      return Y3D_ERR_HIP;                                             \
    }                                                                 \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
