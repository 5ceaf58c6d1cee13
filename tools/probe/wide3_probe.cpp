// Stand-alone timing / trace probe of csrc/conv3x3_wide3.hip on the headline shape (16 groups of 128->128 @80x80, B=32, bf16, random
// operands).  -DY3D_W3_TRACE: s_memtime stamps of one halo-role and one weight-role wave around every L / M segment of two slabs.
#include "conv3x3_wide3_probe.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  int B = 32, H = 80, W = 80, G = 16, Cg = 128, Cn = 128, th = 16;
  if (argc > 1) H = W = atoi(argv[1]);
  if (argc > 2) G = atoi(argv[2]);
  if (argc > 3) Cg = Cn = atoi(argv[3]);
  if (argc > 4) th = atoi(argv[4]);
  long C = (long)G * Cg, nx = (long)B * H * W * C, nw = (long)G * Cn * 9 * Cg;
  std::vector<unsigned short> hx(nx), hw(nw);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; union { float f; unsigned u; } cv; cv.f = f; return (unsigned short)(cv.u >> 16); };
  if (getenv("ZERO")) { for (auto& v : hx) v = 0; } else for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  void *dx, *dw, *dy; float* part;
  hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy, nx * 2);
  hipMalloc(&part, (size_t)B * ((H + th - 1) / th) * ((W + 15) / 16) * C * 2 * 4);
  hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() { return y3d_conv3x3_wide3_launch(th, dx, (long)H * W * C, (long)W * C, C, B, H, W, Cg, Cn, G, dw, 9 * Cg, dy, C, getenv("NOPART") ? nullptr : part, 0, nullptr, nullptr, 0, nullptr); };
  for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed: %s\n", y3d_last_error()); return 1; }
  hipDeviceSynchronize();
  int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  double fl = 2.0 * B * H * W * (double)G * Cn * Cg * 9;
#ifdef Y3D_W3_TRACE
  {
    std::vector<unsigned> tr(2 * 18 * 2 * 5);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(y3d_w3_trace), tr.size() * 4);
    for (int w = 0; w < 2; ++w) {
      printf("wave %d (%s role): per phase: L | wait at barrier 1 | M | wait at barrier 2   (s_memtime ticks)\n", w * 4, w ? "weights" : "halo");
      for (int st = 0; st < 18; ++st)
        for (int ph = 0; ph < 2; ++ph) {
          unsigned* q = &tr[((w * 18 + st) * 2 + ph) * 5];
          printf("  st %2d.%d @%7u: L %5u  b1 %5u  M %5u  b2 %5u   total %5u\n", st, ph, q[0] - tr[0], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[4] - q[0]);
        }
    }
  }
#endif
  printf("%s H=%d G=%d C=%d th=%d: %.3f ms  %.1f TFLOP/s\n", argv[0], H, G, Cg, th, ms, fl / ms / 1e9);
  return 0;
}
