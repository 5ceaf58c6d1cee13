// VALU issue rates on gfx950 without matrix work beside them: scalar v_fma_f32 vs packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, v_exp_f32,
// v_rcp_f32 (MI355X_MICROARCH.md prices packed f32 as an anti-lever BESIDE MFMAs; the BatchNorm / projection passes have no MFMA to hide
// behind and are 60-70 % VALU-active, profiles/r04_pmc_projg.txt).   hipcc -O3 --offload-arch=gfx950 valu_rate.cpp -o bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 v = {x[i], x[i + 1]};
        const f2 aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 v = {x[i], x[i + 1]};
        const f2 aa = {a, a};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(aa));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        unsigned u;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x[i]), "v"(x[i + 1]));
        x[i] = __uint_as_float(u);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int elems_per_instr, int instrs_per_iter) {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000, blocks = 2048;  // 8 workgroups per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * instrs_per_iter;          // wave instructions
  const double cyc = ms * 1e-3 * 2.4e9 * 1024 / winstr;                          // SIMD cycles per wave instruction at 2.4 GHz
  printf("%-18s %8.3f ms  %5.2f cycles per wave instruction (at 2.4 GHz)  %6.1f G elements/s\n", name, ms, cyc, winstr * 64 * elems_per_instr / ms / 1e6);
  hipFree(out);
}
int main() {
  run<0>("v_fma_f32", 1, 16);
  run<1>("v_pk_fma_f32", 2, 8);
  run<2>("v_pk_mul_f32", 2, 8);
  run<5>("v_mul_f32", 1, 16);
  run<3>("v_exp_f32", 1, 16);
  run<4>("v_rcp_f32", 1, 16);
  run<6>("v_cvt_pk_bf16_f32", 2, 8);
  return 0;
}
