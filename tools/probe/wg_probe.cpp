// Stand-alone timing probe for the resident weight-gradient tile kernel (16 groups of 128x128 @80x80, B=32, bf16).
// Variants: -DY3D_WGP_NODMA (no LDS-DMA), -DY3D_WGP_NOMFMA (no MFMA), -DY3D_WGP_NOLDS (no fragment reads).  Not part of the library.
#include "../../yolov10-3d_amd/csrc/conv3x3_wgrad_tile.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  int B = 32, H = 80, W = 80, G = 16, Cg = 128, Cn = 128, th = 8;
  if (argc > 1) H = W = atoi(argv[1]);
  if (argc > 2) G = atoi(argv[2]);
  if (argc > 3) Cg = Cn = atoi(argv[3]);
  long C = (long)G * Cg, nx = (long)B * H * W * C;
  std::vector<unsigned short> hx(nx);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; union { float f; unsigned u; } cv; cv.f = f; v = (unsigned short)(cv.u >> 16); }
  void *dx, *dy; float* slab;
  int ns = y3d_wgrad_tile_splits(th, B, H, W, Cg, Cn, G);
  hipMalloc(&dx, nx * 2); hipMalloc(&dy, nx * 2); hipMalloc(&slab, (size_t)ns * G * Cn * 9 * Cg * 4);
  hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dy, hx.data(), nx * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() { return y3d_conv3x3_wgrad_tile_launch(th, dx, (long)H * W * C, (long)W * C, C, dy, C, B, H, W, Cg, Cn, G, slab, ns, nullptr); };
  for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  double fl = 2.0 * B * H * W * (double)G * Cn * Cg * 9;
  printf("%s H=%d G=%d C=%d splits=%d: %.3f ms  %.1f TFLOP/s\n", argv[0], H, G, Cg, ns, ms, fl / ms / 1e9);
  return 0;
}
