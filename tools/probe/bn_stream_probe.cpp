// How should an NHWC BatchNorm + SiLU pass walk a [P][C] bf16 tensor?  (csrc/bn_act.hip: a block owns a 64-channel slab and a run of pixels,
// so a wave instruction touches eight 128-byte pieces C * 2 bytes apart, two pixels in flight per thread: 4.2 TB/s forward, 4.7 reduce, 5.0
// apply on the 839 MB head tensor against 6.3 TB/s for a plain copy.)  Variants: channels per slab (= contiguous bytes per pixel row a block
// covers), pixels in flight per thread, workgroups per CU.   hipcc -O3 --offload-arch=gfx950 bn_stream_probe.cpp -o bin/bn_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float silu(float u) { return u * __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float silu_grad(float u) {
  float s = __builtin_amdgcn_rcpf(1.f + __expf(-u));
  return s * (1.f + u * (1.f - s));
}
__device__ __forceinline__ void unpack(const uint4& v, float* f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 pack(const float* f) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const f2_t a = {f[2 * i], f[2 * i + 1]}; w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, b2_t)); }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// MODE 0: forward z = silu(y * sc + sf);  1: reduce (sum g, sum g xhat) -> one atomic-free dummy store;  2: apply
template <int MODE, int NFL>
__global__ __launch_bounds__(256) void pass_kernel(const bf16_t* __restrict__ y, const bf16_t* __restrict__ dz, bf16_t* __restrict__ z,
                                                   const float* __restrict__ cst, float* __restrict__ part, long P, int C, int slabw, int ppb) {
  const int CT = slabw / 8, PT = 256 / CT;
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int c = blockIdx.y * slabw + ct * 8;
  float sc[8], sf[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = cst[c + j]; sf[j] = cst[C + c + j]; mu[j] = cst[2 * C + c + j]; is[j] = cst[3 * C + c + j]; }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const long pbeg = (long)blockIdx.x * ppb;
  const long pend = pbeg + ppb < P ? pbeg + ppb : P;
  auto one = [&](long px, const uint4& yv, const uint4& dv) {
    float v[8], d[8], o[8];
    unpack(yv, v);
    if (MODE) unpack(dv, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = v[j] * sc[j] + sf[j];
      if (MODE == 0) o[j] = silu(u);
      else {
        const float g = d[j] * silu_grad(u);
        const float xh = (v[j] - mu[j]) * is[j];
        if (MODE == 1) { s1[j] += g; s2[j] += g * xh; }
        else o[j] = sc[j] * (g - mu[j] - xh * is[j]);
      }
    }
    if (MODE != 1) *(uint4*)(z + px * C + c) = pack(o);
  };
  long px = pbeg + pt;
  for (; px + (NFL - 1) * PT < pend; px += NFL * PT) {
    uint4 yv[NFL], dv[NFL];
#pragma unroll
    for (int i = 0; i < NFL; ++i) {
      yv[i] = *(const uint4*)(y + (px + i * PT) * C + c);
      if (MODE) dv[i] = *(const uint4*)(dz + (px + i * PT) * C + c);
    }
#pragma unroll
    for (int i = 0; i < NFL; ++i) one(px + i * PT, yv[i], MODE ? dv[i] : yv[i]);
  }
  for (; px < pend; px += PT) one(px, *(const uint4*)(y + px * C + c), MODE ? *(const uint4*)(dz + px * C + c) : make_uint4(0, 0, 0, 0));
  if (MODE == 1) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) a += s1[j] + s2[j];
    part[((long)blockIdx.x * gridDim.y + blockIdx.y) * 256 + threadIdx.x] = a;
  }
}

template <int MODE>
void launch(int nfl, dim3 g, const bf16_t* y, const bf16_t* dz, bf16_t* z, const float* cst, float* part, long P, int C, int slabw, int ppb) {
  if (nfl == 2) hipLaunchKernelGGL((pass_kernel<MODE, 2>), g, dim3(256), 0, 0, y, dz, z, cst, part, P, C, slabw, ppb);
  else if (nfl == 4) hipLaunchKernelGGL((pass_kernel<MODE, 4>), g, dim3(256), 0, 0, y, dz, z, cst, part, P, C, slabw, ppb);
  else hipLaunchKernelGGL((pass_kernel<MODE, 8>), g, dim3(256), 0, 0, y, dz, z, cst, part, P, C, slabw, ppb);
}

int main() {
  struct Shape { long P; int C; } shapes[] = {{204800, 2048}, {51200, 2048}, {12800, 2048}, {204800, 1024}};
  for (auto sh : shapes) {
    const long n = sh.P * sh.C;
    bf16_t *y, *dz, *z;
    float *cst, *part;
    hipMalloc(&y, n * 2); hipMalloc(&dz, n * 2); hipMalloc(&z, n * 2);
    hipMalloc(&cst, sh.C * 16); hipMalloc(&part, 64 << 20);
    std::vector<bf16_t> h(n);
    unsigned s = 12345;
    for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (bf16_t)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }
    hipMemcpy(y, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(dz, h.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<float> hc(sh.C * 4, 0.75f);
    hipMemcpy(cst, hc.data(), sh.C * 16, hipMemcpyHostToDevice);
    printf("P = %ld, C = %d (%.0f MB per tensor)\n", sh.P, sh.C, n * 2 / 1e6);
    for (int slabw : {64, 1024, 2048}) {
      if (slabw > sh.C) continue;
      for (int nfl : {2, 4, 8})
        for (int wgs : {512, 1024, 2048}) {
          const int nslab = sh.C / slabw;
          long npx = wgs / nslab;
          if (npx < 1) npx = 1;
          const int PT = 256 / (slabw / 8);
          if (npx > sh.P / (PT * nfl)) npx = sh.P / (PT * nfl);
          if (npx < 1) continue;
          const int ppb = (int)((sh.P + npx - 1) / npx);
          dim3 g((unsigned)npx, nslab);
          float us[3];
          for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto go = [&]() {
              if (mode == 0) launch<0>(nfl, g, y, dz, z, cst, part, sh.P, sh.C, slabw, ppb);
              else if (mode == 1) launch<1>(nfl, g, y, dz, z, cst, part, sh.P, sh.C, slabw, ppb);
              else launch<2>(nfl, g, y, dz, z, cst, part, sh.P, sh.C, slabw, ppb);
            };
            for (int i = 0; i < 3; ++i) go();
            hipEventRecord(e0, 0);
            const int reps = 10;
            for (int i = 0; i < reps; ++i) go();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[mode] = ms * 1e3f / reps;
          }
          const double mb = n * 2 / 1e6;
          printf("  slab %4d ch  in flight %d  wgs %5u x %-3d: fwd %7.1f us %5.2f TB/s | reduce %7.1f us %5.2f | apply %7.1f us %5.2f\n", slabw, nfl, g.x,
                 g.y, us[0], 2 * mb / us[0], us[1], 2 * mb / us[1], us[2], 3 * mb / us[2]);
          fflush(stdout);
        }
    }
    hipFree(y); hipFree(dz); hipFree(z); hipFree(cst); hipFree(part);
  }
  return 0;
}
