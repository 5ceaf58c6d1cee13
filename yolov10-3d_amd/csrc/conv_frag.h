// MFMA operand fragments read from an LDS tile of 128-byte rows whose 16-byte chunks are XOR-swizzled by (row & 7):
// conflict-free for ds_read_b128 on gfx950 for any 16 consecutive rows (checked against the instruction's lane groups).
#pragma once
#include "common.h"

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  // one K sub-step = 32 elements = 4 chunks; lane reads chunk (ks*4 + lane>>4) of row (lane&15)
  static constexpr int KSUB = 2;
  typedef bf16x8_t type;
  __device__ static __forceinline__ type load(const char* tile, int row, int ks, int lane) {
    int r = row + (lane & 15);
    int c = ks * 4 + (lane >> 4);
#ifdef Y3D_PROBE_NOLDS
    return __builtin_bit_cast(bf16x8_t, make_uint4(r, c, r, c));
#else
    const uint4* p = (const uint4*)(tile + r * 128 + ((c ^ (r & 7)) << 4));
    return __builtin_bit_cast(bf16x8_t, *p);
#endif
  }
  // byte offset of this lane's piece of K sub-step ks inside a 128-byte row whose chunks are XOR-swizzled by swz
  __device__ static __forceinline__ int coff(int ks, int lane, int swz) { return ((ks * 4 + (lane >> 4)) ^ swz) << 4; }
  __device__ static __forceinline__ type ld(const char* p) { return __builtin_bit_cast(bf16x8_t, *(const uint4*)p); }
  __device__ static __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag<float> {
  // one K sub-step = 4 elements = 1 chunk; lane reads element (lane>>4) of chunk ks of row (lane&15)
  static constexpr int KSUB = 8;
  typedef float type;
  __device__ static __forceinline__ type load(const char* tile, int row, int ks, int lane) {
    int r = row + (lane & 15);
    return *(const float*)(tile + r * 128 + ((ks ^ (r & 7)) << 4) + ((lane >> 4) << 2));
  }
  __device__ static __forceinline__ int coff(int ks, int lane, int swz) { return ((ks ^ swz) << 4) + ((lane >> 4) << 2); }
  __device__ static __forceinline__ type ld(const char* p) { return *(const float*)p; }
  __device__ static __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

