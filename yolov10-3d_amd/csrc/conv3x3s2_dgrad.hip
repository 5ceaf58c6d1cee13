// Data gradient of a 3x3 stride-2 pad-1 convolution (the down-sampling Convs of the backbone / neck: reference nn/modules/conv.py:120-122
// under autograd), bf16, all four output-parity classes from ONE resident dy tile.
//
//   dx[y][x][ci] = sum over the taps (r, q) with (y + 1 - r), (x + 1 - q) even of  dy[(y + 1 - r) / 2][(x + 1 - q) / 2][co] * w[co][ci][r][q]
//
// With y = 2 hy + py, x = 2 wx + px a dx pixel of class (py, px) reads dy at (hy + dr, wx + dq), dr, dq in {0, 1}: 1 / 2 / 2 / 4 taps.
// The generic kernel (conv_gemm.hip MODE 2) runs each class as its own implicit GEMM: dy is gathered once per tap (9 reads of every
// dy pixel through L2, 945 MB for the 320x320 layer of S-3D) by workgroups that live for one to four K steps - 335 us for 315 MB of
// tensor traffic.  Here a persistent workgroup keeps the weights of its 32 input channels in LDS (9 taps x Cout x 32) and walks
// 8 x 16 dy tiles: the (8 + 1) x (16 + 1) halo of a tile arrives by LDS-DMA (double buffered, the next tile in flight under the
// MFMAs and the stores of the current one: the counted s_waitcnt leaves exactly the 16 stores of a wave outstanding), the four
// distinct windows feed all nine taps, and the 2 x 2 dx pixels of every dy pixel leave the same workgroup back to back (their
// 64-byte records interleave in memory).  The taps of a class and the K steps inside a tap are accumulated in the generic kernel's
// order: bit-identical results.
#include "common.h"
#include "conv_frag.h"

namespace {

struct S2P {
  const bf16_t* dy;
  const bf16_t* w;   // dgrad packing [Cin][Kpad], K index = tap * Cout + co
  bf16_t* dx;
  int dsw, xsw;      // pixel strides in elements
  int B, Ho, Wo, H, W, Cout, Cin, Kpad;
  int nty, ntx, ntiles;
  unsigned dbytes, xbytes;
};

template <int N> __device__ __forceinline__ void s2_wvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((ext_vector_type(2))) unsigned s2_u32x2;

// NSLAB: ceil(Cout / 64) (64-channel K slabs: 128-byte LDS rows; the chunks past Cout - 96 output channels of the M widths - are zeros:
// weights below, halo by out-of-range DMA lanes)
template <int NSLAB>
__global__ __launch_bounds__(256) void conv3x3s2_dgrad_kernel(S2P p) {
  constexpr int TH = 8, HW = 17, SLOTS = 160;       // (TH + 1) * 17 = 153 halo pixels, padded to 20 DMA instructions of 8 slots
  constexpr int TILE = NSLAB * SLOTS * 128;         // bytes per halo buffer
  constexpr int NI = NSLAB * 20 / 4;                // DMA instructions per wave per tile
  constexpr int WB = 9 * NSLAB * 32 * 128;          // resident weights: [tap][slab][ci 32][128 B]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  char* sH = smem + WB;                             // [2][slab][160 slots][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci0 = blockIdx.y * 32;
  const int nwk = gridDim.x, wk = blockIdx.x;
  if (wk >= p.ntiles) return;

  for (int i = tid; i < 9 * NSLAB * 32 * 8; i += 256) {
    const int c = i & 7, r = (i >> 3) & 31, ts = i >> 8;  // ts = tap * NSLAB + slab
    const int tap = ts / NSLAB, sl = ts - tap * NSLAB;
    const int ci = ci0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ci < p.Cin && sl * 64 + c * 8 < p.Cout) v = *(const uint4*)(p.w + (long)ci * p.Kpad + tap * p.Cout + sl * 64 + c * 8);
    *(uint4*)(sW + (ts * 32 + r) * 128 + ((c ^ (r & 7)) << 4)) = v;
  }

  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.dx, 0, (int)p.xbytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  auto decode = [&](int t, int& b, int& hy0, int& wx0) {
    const int tx = t % p.ntx, t2 = t / p.ntx;
    wx0 = tx * 16;
    hy0 = (t2 % p.nty) * TH;
    b = t2 / p.nty;
  };
  // DMA instruction ii (0 .. 20 NSLAB - 1): slab ii / 20, slots 8 (ii % 20) .. + 7; lane l: slot + (l >> 3), chunk (l & 7) ^ (l >> 3).
  // The lane's halo pixel and element offset per instruction are computed once (per tile: two compares, an add, a select each).
  int p_off[NI], p_rc[NI];  // (row << 8) | column, row 255 = never valid
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int ii = wave * NI + n;
    const int sl = ii / 20, s8 = ii - sl * 20;
    const int P = s8 * 8 + (lane >> 3);
    const int row = P / HW, col = P - row * HW;
    p_off[n] = (row * p.Wo + col) * p.dsw + sl * 64 + (((lane & 7) ^ (lane >> 3)) << 3);
    p_rc[n] = ((((P < (TH + 1) * HW) & (sl * 64 + (((lane & 7) ^ (lane >> 3)) << 3) < p.Cout)) ? row : 255) << 8) | col;
  }
  auto issue = [&](int t, int buf) {
    int b, hy0, wx0;
    const bool live = t < p.ntiles;
    decode(live ? t : 0, b, hy0, wx0);
    const int base = ((b * p.Ho + hy0) * p.Wo + wx0) * p.dsw;
    const int hlim = live ? p.Ho : 0;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int ii = wave * NI + n;  // uniform
      const int sl = ii / 20, s8 = ii - sl * 20;
      const int r = p_rc[n] >> 8, c = p_rc[n] & 255;
      const bool ok = (hy0 + r < hlim) & (wx0 + c < p.Wo);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(sH + buf * TILE + (sl * SLOTS + s8 * 8) * 128), 16,
                                               ok ? (unsigned)(base + p_off[n]) * 2u : OOB, 0, 0, 0);
    }
  };
  issue(wk, 0);

  const int lp = lane & 15, lq = lane >> 4;
  int buf = 0;
  bool first = true;
#pragma unroll 1
  for (int t = wk; t < p.ntiles; t += nwk) {
    if (first) { s2_wvm<0>(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); first = false; }
    else s2_wvm<16>();
    __builtin_amdgcn_s_barrier();
    issue(t + nwk, buf ^ 1);
    int b, hy0, wx0;
    decode(t, b, hy0, wx0);
    const char* hb = sH + buf * TILE;
    f32x4_t acc[2][4][2];  // [row of the wave][class 2 py + px][ci tile]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[r][c][a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // class (py, px), its taps in the generic kernel's order (jr, jq): filter row kr = py ? 2 jr : 1, window row dr = (py && kr == 0)
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int py = cls >> 1, px = cls & 1;
#pragma unroll
      for (int jr = 0; jr < (py ? 2 : 1); ++jr)
#pragma unroll
        for (int jq = 0; jq < (px ? 2 : 1); ++jq) {
          const int kr = py ? 2 * jr : 1, kq = px ? 2 * jq : 1;
          const int dr = (py && kr == 0) ? 1 : 0, dq = (px && kq == 0) ? 1 : 0;
          const int tap = kr * 3 + kq;
#pragma unroll
          for (int sl = 0; sl < NSLAB; ++sl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              bf16x8_t fa[2], fb[2];
#pragma unroll
              for (int a = 0; a < 2; ++a) fa[a] = Frag<bf16_t>::load(sW + (tap * NSLAB + sl) * 32 * 128, a * 16, ks, lane);
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                const int P = (wave * 2 + r + dr) * HW + dq + lp;
                fb[r] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(hb + (sl * SLOTS + P) * 128 + (((ks * 4 + lq) ^ (P & 7)) << 4)));
              }
#pragma unroll
              for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[r][cls][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[r], acc[r][cls][a], 0, 0, 0);
            }
        }
    }
    // stores: lane = dy pixel (hy0 + 2 wave + r, wx0 + lp), 4 input channels ci0 + 16 a + 4 lq .. + 3 of each of its 2 x 2 dx pixels
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        const int y = 2 * (hy0 + wave * 2 + r) + (cls >> 1), x = 2 * (wx0 + lp) + (cls & 1);
        const bool inb = (y < p.H) & (x < p.W);
        const unsigned pix = (unsigned)((b * p.H + y) * p.W + x) * (unsigned)p.xsw;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int ci = ci0 + a * 16 + 4 * lq;
          const f32x4_t v = acc[r][cls][a];
          const s2_u32x2 u = {(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
          __builtin_amdgcn_raw_buffer_store_b64(u, rx, (inb & (ci < p.Cin)) ? (pix + ci) * 2u : OOB, 0, 0);
        }
      }
    buf ^= 1;
  }
  s2_wvm<0>();
}

}  // namespace

extern "C" int y3d_get_stream1x1(void);

int y3d_conv3x3s2_dgrad_ok(int dtype, int B, int Ho, int Wo, int H, int W, int Cout, int Cin, long dsw, long xsw) {
  if (!y3d_get_stream1x1() || dtype != Y3D_BF16 || Cout < 32 || Cout > 128 || Cout % 8 != 0 || Cin % 4 != 0 || Cin < 8) return 0;
  if (Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return 0;
  if (((long)B * Ho * Wo * dsw + Cout) * 2 >= (1L << 32) - 64 || ((long)B * H * W * xsw + Cin) * 2 >= (1L << 32) - 64) return 0;
  return 1;
}

int y3d_conv3x3s2_dgrad_launch(const void* dy, long dsw, int B, int Ho, int Wo, int Cout, const void* w_packed_dgrad, int Kpad, void* dx, long xsw,
                               int H, int W, int Cin, void* stream) {
  Y3D_CHECK(((uintptr_t)dy & 15) == 0 && dsw % 8 == 0 && ((uintptr_t)w_packed_dgrad & 15) == 0 && Kpad % 8 == 0 && ((uintptr_t)dx & 7) == 0 && xsw % 4 == 0,
            "conv3x3s2_dgrad: operand alignment");
  S2P p;
  p.dy = (const bf16_t*)dy; p.w = (const bf16_t*)w_packed_dgrad; p.dx = (bf16_t*)dx; p.dsw = (int)dsw; p.xsw = (int)xsw;
  p.B = B; p.Ho = Ho; p.Wo = Wo; p.H = H; p.W = W; p.Cout = Cout; p.Cin = Cin; p.Kpad = Kpad;
  p.nty = cdiv(Ho, 8); p.ntx = cdiv(Wo, 16); p.ntiles = B * p.nty * p.ntx;
  p.dbytes = (unsigned)((((long)B * Ho * Wo - 1) * dsw + Cout) * 2);
  p.xbytes = (unsigned)((((long)B * H * W - 1) * xsw + Cin) * 2);
  const int nslab = cdiv(Cout, 64);
  const size_t lds = (size_t)9 * nslab * 32 * 128 + 2 * (size_t)nslab * 160 * 128;
  const int nci = cdiv(Cin, 32);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int nwk = 256 * per_cu / nci;
  if (nwk < 1) nwk = 1;
  if (nwk > p.ntiles) nwk = p.ntiles;
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3s2_dgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv3x3s2_dgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (nslab == 2) hipLaunchKernelGGL(conv3x3s2_dgrad_kernel<2>, dim3(nwk, nci), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(conv3x3s2_dgrad_kernel<1>, dim3(nwk, nci), dim3(256), lds, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}
