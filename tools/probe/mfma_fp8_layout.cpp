// Operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3), checked with exact small-integer data before the
// fp8 convolution kernel relies on them (the guide: "other dtypes: check the map with exact integer data").
// Hypothesis: lane l holds A[row l & 15][k = 32 * (l >> 4) + j] and B[k = 32 * (l >> 4) + j][col l & 15], j = 0..31 (byte j of the 8 dwords);
// its scale byte (E8M0: 2^(byte - 127), byte OPSEL of the scale VGPR) applies to exactly those 32 K elements; C/D as the bf16 16x16 form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

static unsigned char e4m3(float v) {  // exact for the small values used here
  if (v == 0.f) return 0;
  unsigned char s = v < 0 ? 0x80 : 0;
  v = fabsf(v);
  int e; float m = frexpf(v, &e);  // v = m * 2^e, m in [0.5, 1)
  int E = e - 1 + 7;               // 1.xxx * 2^(e-1)
  int M = (int)roundf((m * 2.f - 1.f) * 8.f);
  return s | (unsigned char)(E << 3) | (unsigned char)M;
}

template <int OA, int OB>
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned* sa, const unsigned* sb, float* C) {
  const int l = threadIdx.x, r = l & 15, kb = l >> 4;
  i32x8_t a, b;
  for (int d = 0; d < 8; ++d) {
    unsigned va = 0, vb = 0;
    for (int j = 0; j < 4; ++j) {
      va |= (unsigned)A[r * 128 + 32 * kb + 4 * d + j] << (8 * j);
      vb |= (unsigned)B[(32 * kb + 4 * d + j) * 16 + r] << (8 * j);
    }
    a[d] = (int)va; b[d] = (int)vb;
  }
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
  for (int i = 0; i < 4; ++i) C[(kb * 4 + i) * 16 + r] = c[i];  // row = (lane >> 4) * 4 + reg, col = lane & 15
}

int main() {
  std::vector<unsigned char> A(16 * 128), B(128 * 16);
  std::vector<float> Af(16 * 128), Bf(128 * 16);
  unsigned s = 7;
  const float vals[8] = {0.f, 1.f, -1.f, 2.f, 0.5f, -3.f, 1.5f, -0.25f};
  for (int i = 0; i < 16 * 128; ++i) { s = s * 1664525u + 1013904223u; Af[i] = vals[(s >> 13) & 7]; A[i] = e4m3(Af[i]); }
  for (int i = 0; i < 128 * 16; ++i) { s = s * 1664525u + 1013904223u; Bf[i] = vals[(s >> 11) & 7]; B[i] = e4m3(Bf[i]); }
  // scale bytes: different per (row, k-block) and per byte position, small exponents around 127
  std::vector<unsigned> sa(64), sb(64);
  for (int l = 0; l < 64; ++l) {
    unsigned wa = 0, wb = 0;
    for (int o = 0; o < 4; ++o) {
      wa |= (unsigned)(127 + ((l * 3 + o * 5) % 5) - 2) << (8 * o);
      wb |= (unsigned)(127 + ((l * 7 + o * 3) % 4) - 1) << (8 * o);
    }
    sa[l] = wa; sb[l] = wb;
  }
  unsigned char *dA, *dB; unsigned *dsa, *dsb; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 1024);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int oa = cfg & 1 ? 2 : 0, ob = cfg & 2 ? 3 : 1;
    if (cfg == 0) hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    if (cfg == 1) hipLaunchKernelGGL((k<2, 1>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    if (cfg == 2) hipLaunchKernelGGL((k<0, 3>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    if (cfg == 3) hipLaunchKernelGGL((k<2, 3>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int kb = 0; kb < 4; ++kb) {
          const int ea = (int)((sa[kb * 16 + i] >> (8 * oa)) & 255) - 127, eb = (int)((sb[kb * 16 + j] >> (8 * ob)) & 255) - 127;
          double part = 0;
          for (int kk = 0; kk < 32; ++kk) part += (double)Af[i * 128 + 32 * kb + kk] * Bf[(32 * kb + kk) * 16 + j];
          ref += part * ldexp(1.0, ea + eb);
        }
        if (fabs(ref - C[i * 16 + j]) > 1e-6 * (1 + fabs(ref))) { if (bad < 3) printf("  cfg %d mismatch C[%d][%d] = %g, expected %g\n", cfg, i, j, C[i * 16 + j], ref); ++bad; }
      }
    printf("opsel_a %d opsel_b %d: %s (%d of 256 wrong)\n", oa, ob, bad ? "FAIL" : "PASS", bad);
    bad_total += bad;
  }
  printf("%s\n", bad_total ? "LAYOUT HYPOTHESIS REJECTED" : "layout + scale semantics confirmed: lane l -> row/col l&15, k block l>>4 (32 consecutive k), scale byte OPSEL of the lane's scale VGPR = E8M0 of that block");
  return bad_total != 0;
}
