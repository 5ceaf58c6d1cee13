// Stand-alone timing probe for the resident-halo 3x3 kernel on the headline shape (16 groups of 128->128 @80x80, B=32, bf16).
// Built in variants (-DY3D_PROBE_NODMA / _NOMFMA / _NOLDS) to see which resource bounds the loop.  Not part of the library.
#include "../../yolov10-3d_amd/csrc/conv3x3_tile.hip"
#include "conv3x3_wide_v2_probe.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int y3d_conv3x3_tile_launch(int dtype, int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G,
                            const void* w, int Ktot, void* y, long ysw, float* part, int flip, const float* scale, const float* shift, int act,
                            void* stream);
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
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  void *dx, *dw, *dy; float* part;
  hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy, nx * 2);
  hipMalloc(&part, (size_t)B * (H / th) * ((W + 15) / 16) * C * 2 * 4);
  hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() { return y3d_conv3x3_tile_launch(1, th, dx, (long)H * W * C, (long)W * C, C, B, H, W, Cg, Cn, G, dw, 9 * Cg, dy, C, part, 0, nullptr, nullptr, 0, nullptr); };
  for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed: %s\n", y3d_last_error()); return 1; }
  hipDeviceSynchronize();
  int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  double fl = 2.0 * B * H * W * (double)G * Cn * Cg * 9;
#ifdef Y3D_PROBE_STAMP
  {
    int nwg = B * (H / th) * ((W + 15) / 16) * G; if (nwg > 16384) nwg = 16384;
    std::vector<unsigned long long> st(nwg * 4);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(y3d_probe_stamps), nwg * 32);
    double a = 0, b = 0, c = 0; unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < nwg; ++i) { a += st[i*4+1] - st[i*4]; b += st[i*4+2] - st[i*4+1]; c += st[i*4+3] - st[i*4+2]; if (st[i*4] < lo) lo = st[i*4]; if (st[i*4+3] > hi) hi = st[i*4+3]; }
    printf("stamps (s_memtime ticks, 100 MHz?): prologue %.0f  loop %.0f  epilogue %.0f  per WG; kernel span %llu ticks; sum/256 CUs = %.0f\n", a / nwg, b / nwg, c / nwg, hi - lo, (a + b + c) / 256);
  }
#endif
#ifdef Y3D_PROBE_TRACE
  {
    std::vector<unsigned> tr(2 * 18 * 6);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(y3d_probe_trace), tr.size() * 4);
    for (int w = 0; w < 2; ++w) {
      printf("wave %d (%s role): per stage: issue | half0 | half1 | wait | barrier   (cycles)\n", w * 4, w ? "weights" : "halo");
      for (int st = 0; st < 18; ++st) {
        unsigned* q = &tr[(w * 18 + st) * 6];
        printf("  st %2d @%7u: %5u %5u %5u %5u %5u  total %5u\n", st, q[0] - tr[0], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - q[0]);
      }
    }
  }
#endif
  printf("%s H=%d G=%d C=%d th=%d: %.3f ms  %.1f TFLOP/s\n", argv[0], H, G, Cg, th, ms, fl / ms / 1e9);
  return 0;
}
