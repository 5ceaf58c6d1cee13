// Operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3), found with exact small-integer data before the
// fp8 convolution kernel relies on them (the guide: "other dtypes: check the map with exact integer data").
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
  int e; float m = frexpf(v, &e);
  int E = e - 1 + 7;
  int M = (int)roundf((m * 2.f - 1.f) * 8.f);
  return s | (unsigned char)(E << 3) | (unsigned char)M;
}

// operands are handed over per lane exactly as given: a[l][8 dwords], b[l][8 dwords], scale words
template <int OA, int OB>
__global__ void k(const int* a_, const int* b_, const unsigned* sa, const unsigned* sb, float* C) {
  const int l = threadIdx.x;
  i32x8_t a, b;
  for (int d = 0; d < 8; ++d) { a[d] = a_[l * 8 + d]; b[d] = b_[l * 8 + d]; }
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
  for (int i = 0; i < 4; ++i) C[l * 4 + i] = c[i];  // raw: lane, register
}

int *dA, *dB; unsigned *dsa, *dsb; float* dC;
std::vector<float> run(int oa, int ob, const std::vector<int>& a, const std::vector<int>& b, const std::vector<unsigned>& sa, const std::vector<unsigned>& sb) {
  hipMemcpy(dA, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
#define L(OA, OB) if (oa == OA && ob == OB) hipLaunchKernelGGL((k<OA, OB>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
  L(0, 0) L(1, 0) L(2, 0) L(3, 0) L(0, 1) L(0, 2) L(0, 3)
  std::vector<float> C(256);
  hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  return C;
}

int main() {
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 1024);
  const unsigned one = e4m3(1.f);
  const unsigned ones4 = one * 0x01010101u;
  std::vector<unsigned> unit(64, 0x7f7f7f7fu);
  // ---- E1: where does lane L byte j of A go?  A = one non-zero element, B = all ones -> row sums.  C raw [lane][reg]
  printf("E1: A has a single 1.0 at (lane La, byte ja); B all ones; unit scales.  Non-zero C[lane][reg]:\n");
  for (int La : {0, 5, 16, 37, 63})
    for (int ja : {0, 9, 31}) {
      std::vector<int> a(512, 0), b(512, (int)ones4);
      a[La * 8 + ja / 4] = (int)(one << (8 * (ja % 4)));
      auto C = run(0, 0, a, b, unit, unit);
      printf("  La %2d ja %2d ->", La, ja);
      int n = 0;
      for (int i = 0; i < 256; ++i) if (C[i] != 0.f) { if (n < 4) printf(" C[lane %d reg %d]=%g", i / 4, i % 4, C[i]); ++n; }
      printf("  (%d non-zero)\n", n);
    }
  // ---- E2: pairing of k between A and B: A single 1.0 at (La, ja), B single 1.0 at (Lb, jb): C non-zero iff same k
  printf("E2: single 1.0 in A at (La, ja) and in B at (Lb, jb): C non-zero iff they share k\n");
  for (int La : {3, 35})
    for (int ja : {0, 17}) {
      int hits = 0, Lb_hit = -1, jb_hit = -1;
      for (int Lb = 0; Lb < 64; ++Lb)
        for (int jb = 0; jb < 32; ++jb) {
          if ((Lb & 15) != 7) continue;  // one B column is enough
          std::vector<int> a(512, 0), b(512, 0);
          a[La * 8 + ja / 4] = (int)(one << (8 * (ja % 4)));
          b[Lb * 8 + jb / 4] = (int)(one << (8 * (jb % 4)));
          auto C = run(0, 0, a, b, unit, unit);
          for (int i = 0; i < 256; ++i) if (C[i] != 0.f) { ++hits; Lb_hit = Lb; jb_hit = jb; }
        }
      printf("  A(La %d, ja %d) meets B(Lb %d, jb %d)  [%d hits]\n", La, ja, Lb_hit, jb_hit, hits);
    }
  // ---- E3: scales.  A, B all ones (C = 128 everywhere with unit scales).  One lane's scale_a byte `by` = 128 (x2), opsel o.
  printf("E3: all-ones operands; lane Ls has scale_a byte `by` = 128 (x2), others 127; C values that differ from 128:\n");
  {
    std::vector<int> a(512, (int)ones4), b(512, (int)ones4);
    for (int o : {0, 1, 2, 3})
      for (int by : {0, 1, 2, 3})
        for (int Ls : {2, 21, 40}) {
          std::vector<unsigned> sa(64, 0x7f7f7f7fu);
          sa[Ls] = (0x7f7f7f7fu & ~(0xffu << (8 * by))) | (128u << (8 * by));
          auto C = run(o, 0, a, b, sa, unit);
          int n = 0; int first = -1; float v = 0;
          for (int i = 0; i < 256; ++i) if (C[i] != 128.f) { if (first < 0) { first = i; v = C[i]; } ++n; }
          if (n) printf("  opsel_a %d byte %d lane %2d: %d entries changed, first C[lane %d reg %d] = %g\n", o, by, Ls, n, first / 4, first % 4, v);
        }
    printf("  (same for scale_b, opsel_b)\n");
    for (int o : {0, 1, 2, 3})
      for (int by : {0, 1, 2, 3})
        for (int Ls : {2, 21, 40}) {
          std::vector<unsigned> sb(64, 0x7f7f7f7fu);
          sb[Ls] = (0x7f7f7f7fu & ~(0xffu << (8 * by))) | (128u << (8 * by));
          auto C = run(0, o, a, b, unit, sb);
          int n = 0; int first = -1; float v = 0;
          for (int i = 0; i < 256; ++i) if (C[i] != 128.f) { if (first < 0) { first = i; v = C[i]; } ++n; }
          if (n) printf("  opsel_b %d byte %d lane %2d: %d entries changed, first C[lane %d reg %d] = %g\n", o, by, Ls, n, first / 4, first % 4, v);
        }
  }
  // ---- E3b: WHICH k block does lane (c, s)'s scale byte apply to?  A ones only in the k block `kd` of every row; B all ones
  printf("E3b: data only in k block kd (A), scale x2 in lane 16*s + 3 -> changed?\n");
  for (int kd = 0; kd < 4; ++kd)
    for (int which = 0; which < 2; ++which) {
      printf("  %s scale, data block %d: scale lanes that matter:", which ? "B" : "A", kd);
      for (int sl = 0; sl < 4; ++sl) {
        std::vector<int> a(512, 0), b(512, (int)ones4);
        for (int l = 16 * kd; l < 16 * kd + 16; ++l) for (int d = 0; d < 8; ++d) a[l * 8 + d] = (int)ones4;
        std::vector<unsigned> sa(64, 0x7f7f7f7fu), sb(64, 0x7f7f7f7fu);
        (which ? sb : sa)[16 * sl + 3] = 0x80808080u;
        auto C = run(0, 0, a, b, sa, sb);
        int n = 0;
        for (int i = 0; i < 256; ++i) n += C[i] != 32.f;
        if (n) printf(" s=%d(%d changed)", sl, n);
      }
      printf("\n");
    }
  // ---- E4: random small-integer operands and random scale bytes per lane and byte: the full formula
  //      C[i][j] = sum_kb 2^(sa[lane kb*16+i].byte[oa] - 127) * 2^(sb[lane kb*16+j].byte[ob] - 127) * sum_{k in block kb} A[i][k] B[k][j]
  {
    const float vals[8] = {0.f, 1.f, -1.f, 2.f, 0.5f, -3.f, 1.5f, -0.25f};
    std::vector<float> Af(16 * 128), Bf(128 * 16);
    unsigned s = 7;
    for (auto& v : Af) { s = s * 1664525u + 1013904223u; v = vals[(s >> 13) & 7]; }
    for (auto& v : Bf) { s = s * 1664525u + 1013904223u; v = vals[(s >> 11) & 7]; }
    std::vector<int> a(512), b(512);
    for (int l = 0; l < 64; ++l)
      for (int d = 0; d < 8; ++d) {
        unsigned va = 0, vb = 0;
        for (int j = 0; j < 4; ++j) {
          const int byte = 4 * d + j, kk = 64 * (byte >> 4) + 16 * (l >> 4) + (byte & 15);  // the K index of (lane, byte): see the summary below
          va |= (unsigned)e4m3(Af[(l & 15) * 128 + kk]) << (8 * j);
          vb |= (unsigned)e4m3(Bf[kk * 16 + (l & 15)]) << (8 * j);
        }
        a[l * 8 + d] = (int)va; b[l * 8 + d] = (int)vb;
      }
    for (int mode = 0; mode < 8; ++mode) {
      std::vector<unsigned> sa(64), sb(64);
      for (int l = 0; l < 64; ++l) {
        unsigned wa = 0, wb = 0;
        for (int o = 0; o < 4; ++o) {
          s = s * 1664525u + 1013904223u; const unsigned ra = 124 + ((s >> 9) % 7);
          s = s * 1664525u + 1013904223u; const unsigned rb = 124 + ((s >> 9) % 7);
          unsigned xa = (mode == 1 ? 127u : ra), xb = (mode == 0 ? 127u : rb);
          if (mode == 3) { xa = 127; xb = 127; }                       // unit scales: the data layout alone
          if (mode == 4) { xa = 127; xb = 126; }                       // uniform x0.5 on B
          if (mode == 5) { xa = 127; xb = 127 + (rb & 1); }            // B scales in {1, 2}
          if (mode == 6) { xa = 127; xb = 126 + (rb & 1); }            // B scales in {0.5, 1}
          if (mode == 7) { xa = 125 + (ra % 5); xb = 127; }            // A scales 2^-2 .. 2^2
          wa |= xa << (8 * o);
          wb |= xb << (8 * o);
        }
        sa[l] = wa; sb[l] = wb;
      }
      for (int o : {0, 2}) {
        const int oa = (mode == 1 || mode >= 3) ? 0 : o, ob = mode == 0 ? 0 : (mode == 1 ? o : 0);
        auto C = run(oa, ob, a, b, sa, sb);
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
            const float got = C[(j + 16 * (i >> 2)) * 4 + (i & 3)];
            if (fabs(ref - got) > 1e-6 * (1 + fabs(ref))) { if (bad < 2) printf("    C[%d][%d] = %g, expected %g\n", i, j, got, ref); ++bad; }
          }
        printf("E4 mode %d (0: A scales, 1: B scales, 2: both, 3: unit, 4: B x0.5, 5-7: per-lane subsets) opsel_a %d opsel_b %d: %s (%d of 256 wrong)\n", mode, oa, ob, bad ? "FAIL" : "PASS", bad);
      }
    }
  }
  printf("summary: lane l, byte j of the 32-byte operand holds K = 64 * (j >> 4) + 16 * (l >> 4) + (j & 15) of row / column l & 15; the E8M0 byte OPSEL of lane\n"
         "         l's scale VGPR scales the 32 K values [32 * (l >> 4), +32) of row / column l & 15; C/D: row = 4 * (lane >> 4) + reg, col = lane & 15\n");
  return 0;
}
