// LDS-DMA issue/throughput microbenchmark: cycles per 1 KB wave-instruction for different source address patterns,
// L2-resident source, 1..8 issuing waves per workgroup (one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ROWB, int DEPTH>  // contiguous bytes per source row; DEPTH = pieces left in flight per wave between batches
__global__ __launch_bounds__(512) void dma_kernel(const char* src, long pitch, int nwaves, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= nwaves) return;
  constexpr int LPR = ROWB / 16;           // lanes per row
  const int row = lane / LPR, piece = lane % LPR;
  const char* base = src + (long)blockIdx.x * 262144 + (long)wave * 32768;  // per-wave rows, reused -> L2/L1 hits after first touch
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // walk a 256 KB per-CU window (32 KB per wave): misses the vector L1, hits L2
      const long r = ((long)(it * 8 + j) * (64 / LPR) + row) % (32768 / ROWB);
      const char* s = base + r * ROWB + piece * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)(smem + wave * 8192 + j * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int ROWB, int DEPTH>
void run(const char* d, long pitch, unsigned long long* dout) {
  hipFuncSetAttribute((const void*)dma_kernel<ROWB, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int nw = 2; nw <= 8; nw *= 2) {
    int iters = 200;
    hipLaunchKernelGGL((dma_kernel<ROWB, DEPTH>), dim3(256), dim3(512), 65536, 0, d, pitch, nw, iters, dout);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((dma_kernel<ROWB, DEPTH>), dim3(256), dim3(512), 65536, 0, d, pitch, nw, iters, dout);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2048);
    hipMemcpy(h.data(), dout, 2048 * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) { s += h[b * 8 + w]; ++n; }
    double cyc = s / n;  // per wave
    printf("row %4d B depth %d+8, %d waves/CU: %.0f cycles per wave for %d pieces -> %.1f cycles/piece/wave, %.1f cycles/piece/CU (%.1f B/clk/CU)\n", ROWB, DEPTH, nw, cyc,
           iters * 8, cyc / (iters * 8), cyc / (iters * 8) / nw, 1024.0 * nw * iters * 8 / cyc);
  }
}
int main() {
  long pitch = 4096; size_t bytes = (size_t)256 * 262144 + 65536;
  char* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  unsigned long long* dout; hipMalloc(&dout, 2048 * 8);
  run<64, 0>(d, pitch, dout); run<64, 2>(d, pitch, dout); run<64, 4>(d, pitch, dout); run<64, 8>(d, pitch, dout); run<64, 16>(d, pitch, dout);
  return 0;
}
