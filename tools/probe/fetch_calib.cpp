// What does rocprofv3's FETCH_SIZE report for the access shapes of the headline kernel?  (MI355X_MICROARCH.md, HBM section: wide
// coalesced reads are tallied at HALF their bytes on gfx950; "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern".)  Every kernel below reads each byte of an 839 MB buffer - one stride-8 head tensor: 204 800 pixels x 4 KB -
// exactly once, by LDS-DMA (16 B per lane):
//   contig  : a wave instruction fetches 1 KB of consecutive bytes                       (the guide's calibrated shape)
//   piece64 : a wave instruction fetches the SAME 64-byte slab of 16 consecutive pixels  (4 KB apart: the halo gather of conv3x3_wide)
//   piece128: the same with 128-byte slabs of 8 pixels
// Run each under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with 839 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
template <int PIECE>  // bytes per pixel per instruction: 0 = contiguous
__global__ __launch_bounds__(256) void gather_kernel(const char* src, long npix, int pixbytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  if (PIECE == 0) {
    const long total = npix * pixbytes / 1024;
    for (long i = gw; i < total; i += nw) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
      if ((i / nw) % 8 == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    constexpr int LPP = PIECE ? PIECE / 16 : 1, PPI = 64 / LPP;  // lanes per pixel, pixels per instruction
    const int slabs = pixbytes / PIECE;
    const long groups = npix / PPI, total = groups * slabs;
    for (long i = gw; i < total; i += nw) {
      const long grp = i / slabs;
      const int slab = (int)(i - grp * slabs);
      const long pix = grp * PPI + lane / LPP;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pix * pixbytes + slab * PIECE + (lane % LPP) * 16),
                                       (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
      if ((i / nw) % 8 == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main(int argc, char** argv) {
  const long npix = 204800;
  const int pixbytes = 4096;
  char* d;
  if (hipMalloc(&d, npix * pixbytes) != hipSuccess) return 1;
  (void)hipMemset(d, 1, npix * pixbytes);
  const char* mode = argc > 1 ? argv[1] : "contig";
  // a second 1 GB buffer is written between the runs so that nothing of the first stays in the 256 MB Infinity Cache
  char* flush;
  (void)hipMalloc(&flush, 1L << 30);
  for (int it = 0; it < 3; ++it) {
    (void)hipMemset(flush, it, 1L << 30);
    (void)hipDeviceSynchronize();
    if (!strcmp(mode, "contig")) hipLaunchKernelGGL(gather_kernel<0>, dim3(2048), dim3(256), 4096, 0, d, npix, pixbytes);
    else if (!strcmp(mode, "piece64")) hipLaunchKernelGGL(gather_kernel<64>, dim3(2048), dim3(256), 4096, 0, d, npix, pixbytes);
    else hipLaunchKernelGGL(gather_kernel<128>, dim3(2048), dim3(256), 4096, 0, d, npix, pixbytes);
    (void)hipDeviceSynchronize();
  }
  printf("%s: %.1f MB read per launch\n", mode, npix * (double)pixbytes / 1e6);
  return 0;
}
