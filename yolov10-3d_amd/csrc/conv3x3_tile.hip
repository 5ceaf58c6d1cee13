// 3x3 stride-1 "same" convolution with the input halo tile RESIDENT in LDS — the fast path for the shapes that carry the
// YOLOv10-3D step (head 3x3 convs: 83 % of S-3D forward FLOPs; Bottleneck 3x3s of the body).
//
// One workgroup owns TH x 16 output pixels x 128 output channels of one group.  Per 64-channel (bf16; 32 for fp32) input slab
// the (TH+2) x 18 halo tile is brought into LDS ONCE and all nine filter taps read their shifted windows straight out of it
// as MFMA B-operand fragments (no im2col copy anywhere, each input pixel fetched once per slab instead of nine times); the
// 128 x 64 weight tile of the current tap streams through a second LDS ring.  Both rings are filled by LDS-DMA
// (`global_load_lds_dwordx4`: no VGPR staging, no ds_write); the XOR swizzle that keeps `ds_read_b128` conflict-free is applied
// on the per-lane SOURCE address (the DMA destination is lane-linear), zero padding comes from a 16-byte zero page.
// Global->LDS traffic per workgroup is ~3x lower than the generic implicit GEMM of conv_gemm.hip (204 vs 64 FLOP/B at TH=16),
// which moves the kernel from L1-fill-bound towards MFMA-bound.
//
// The data gradient of such a conv is the same kernel on dy with the taps flipped (`flip`).
#include "common.h"
#include "conv_frag.h"

namespace {

__device__ uint4 y3d_zero_page[4];  // zero-initialised: source of every padded / out-of-range 16-byte chunk

struct C3P {
  const void* x;
  const void* w;   // packed [G][Cn][9][Cg] (forward) or [G][Cn][9][Cg] of the dgrad packing; row pitch Ktot
  void* y;
  float* part;     // optional BN partials [B*nty*ntx][G*Cn][2]
  const float* scale;  // optional per-channel affine (+SiLU) epilogue (eval-mode BatchNorm)
  const float* shift;
  int act;
  long xsb, xsh, xsw, ysw;
  int B, H, W;
  int Cg, Cn, G;
  int Ktot;
  int ntx, nty, ntc;
  int flip;
};

template <typename T, int TH>
__global__ __launch_bounds__(TH * 32) void conv3x3_tile_kernel(C3P p) {
  constexpr int CE = TT<T>::CE;
  constexpr int CSE = TT<T>::BKE;          // channels per slab (128 bytes)
  constexpr int NW = TH / 2;               // waves: (TH/4) pixel-row groups x 2 channel halves
  constexpr int NT = NW * 64;
  constexpr int HWD = 18;
  constexpr int NPIX = (TH + 2) * HWD;
  constexpr int HCH = NPIX * 8;            // 16-byte chunks of one halo slab
  constexpr int HR = (HCH + NT - 1) / NT;  // DMA rounds per halo slab
  constexpr int HBYTES = HCH * 16;         // exact: lanes past the end of the last round are masked off
  constexpr int WR = 1024 / NT;            // DMA rounds per 128 x 128 B weight tile
  constexpr int TPS = TH == 16 ? 2 : 1;    // filter taps per barrier: 64 MFMAs per wave between barriers at TH=16
  constexpr int WBYTES = TPS * 16384;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;               // [2][HBYTES]
  char* sW = smem + 2 * HBYTES;  // [2][TPS][16 KB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wc = wave & 1;
  const int g = blockIdx.z;
  int tc, tx, ty, b, tile_lin;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tc = lin % p.ntc;
    tile_lin = lin / p.ntc;
    tx = tile_lin % p.ntx;
    int t2 = tile_lin / p.ntx;
    ty = t2 % p.nty;
    b = t2 / p.nty;
  }
  const int x0 = tx * 16, y0 = ty * TH, c0 = tc * 128;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ Wt = (const T*)p.w;
  const T* zero = (const T*)y3d_zero_page;

  // ---- DMA source pointers, fixed for the whole kernel (the slab / tap offset is added per issue) -------------------------
  const T* hsrc[HR];
#pragma unroll
  for (int rd = 0; rd < HR; ++rd) {
    int chunk = rd * NT + tid;
    int P = chunk >> 3, s = chunk & 7;
    int hy = P / HWD, hx = P - hy * HWD;
    int yy = y0 + hy - 1, xx = x0 + hx - 1;
    bool inb = chunk < HCH && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    hsrc[rd] = inb ? X + (long)b * p.xsb + (long)yy * p.xsh + (long)xx * p.xsw + (long)g * p.Cg + ((s ^ (P & 7)) * CE) : nullptr;
  }
  const T* wsrc[WR];
#pragma unroll
  for (int rd = 0; rd < WR; ++rd) {
    int chunk = rd * NT + tid;
    int n = chunk >> 3, s = chunk & 7;
    wsrc[rd] = (c0 + n < p.Cn) ? Wt + ((long)(g * p.Cn + c0 + n)) * p.Ktot + ((s ^ (n & 7)) * CE) : nullptr;
  }
  auto issue_w = [&](int lin, char* dstbase) {  // lin = slab * 9 + tap
    const int slab = lin / 9, tap = lin - slab * 9;
    const long off = (long)(p.flip ? 8 - tap : tap) * p.Cg + (long)slab * CSE;
#pragma unroll
    for (int rd = 0; rd < WR; ++rd) {
      const T* src = wsrc[rd] ? wsrc[rd] + off : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dstbase + (rd * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto issue_h = [&](int slab, int buf, int rd) {
    // rd is wave-uniform; lanes past the end of the image of the last round are masked off
    if (rd * NT + tid < HCH) {
      const T* src = hsrc[rd] ? hsrc[rd] + (long)slab * CSE : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sH + buf * HBYTES + (rd * NT + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];  // [channel tile][pixel-row tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nslab = p.Cg / CSE;
  const int S = nslab * 9;  // linear (slab, tap) steps
#pragma unroll
  for (int rd = 0; rd < HR; ++rd) issue_h(0, 0, rd);
#pragma unroll
  for (int t = 0; t < TPS; ++t)
    if (t < S) issue_w(t, sW + t * 16384);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Fragments are double buffered by hand: while the 16 MFMAs of one K sub-step run, the LDS reads of the next sub-step are
  // already in flight (also across taps and, for the halo operand, across the barrier: the halo slab is stable within a slab).
  // Only the four weight fragments of a stage's first sub-step wait for the barrier that publishes the DMA.
  constexpr int KS = Frag<T>::KSUB;
  typename Frag<T>::type fb[2][4], fa[2][4];
  auto load_b = [&](typename Frag<T>::type* dst, int lin, int ks) {
    const int slab = lin / 9, tap = lin - slab * 9;
    const int r = tap / 3, q = tap - r * 3;
    const char* hb = sH + (slab & 1) * HBYTES;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) dst[pt] = Frag<T>::load(hb, (4 * wp + pt + r) * HWD + q, ks, lane);
  };
  auto load_a = [&](typename Frag<T>::type* dst, const char* wb, int ks) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) dst[ct] = Frag<T>::load(wb, wc * 64 + ct * 16, ks, lane);
  };
  load_b(fb[0], 0, 0);
  int hslab = 0, hrd = 0;  // slab whose halo is being streamed in, next DMA round
  for (int s0 = 0, stg = 0; s0 < S; s0 += TPS, ++stg) {
    // prefetch the next stage's weight tiles; bring in the next slab's halo as soon as its buffer is free
    char* wnext = sW + ((stg + 1) & 1) * WBYTES;
#pragma unroll
    for (int t = 0; t < TPS; ++t)
      if (s0 + TPS + t < S) issue_w(s0 + TPS + t, wnext + t * 16384);
    {
      // every tap of this stage lies in slab >= k, so slab k-1's halo buffer is free: stream slab k+1's halo into it, a few
      // DMA rounds per stage (all rounds have landed at least one stage before the first tap of slab k+1 is prefetched)
      const int k = s0 / 9;
      if (k + 1 < nslab) {
        if (hslab != k + 1) { hslab = k + 1; hrd = 0; }
        constexpr int RPS = TPS == 2 ? (HR + 2) / 3 : (HR + 5) / 6;
#pragma unroll
        for (int rd = 0; rd < HR; ++rd)
          if (rd >= hrd && rd < hrd + RPS) issue_h(k + 1, (k + 1) & 1, rd);
        hrd += RPS;
      }
    }
    const char* wb = sW + (stg & 1) * WBYTES;
    load_a(fa[0], wb, 0);
#pragma unroll
    for (int step = 0; step < TPS * KS; ++step) {
      const int t = step / KS, ks = step - t * KS;
      const int cur = step & 1, nxt = cur ^ 1;
      if (s0 + t < S) {  // uniform: the last stage may hold fewer taps
        if (ks + 1 < KS) {
          load_b(fb[nxt], s0 + t, ks + 1);
          load_a(fa[nxt], wb + t * 16384, ks + 1);
        } else if (t + 1 < TPS) {
          if (s0 + t + 1 < S) {
            load_b(fb[nxt], s0 + t + 1, 0);
            load_a(fa[nxt], wb + (t + 1) * 16384, 0);
          }
        } else if (s0 + TPS < S) {
          load_b(fb[nxt], s0 + TPS, 0);  // first sub-step of the next stage: halo operand only
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = Frag<T>::mma(fa[cur][ct], fb[cur][pt], acc[ct][pt]);
      }
    }
    if ((TPS * KS) & 1) {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) fb[0][pt] = fb[1][pt];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: store, optional BN partial sums over the valid pixels -----------------------------------------------------
  T* __restrict__ Y = (T*)p.y;
  const int lc = (lane >> 4) * 4, lp = lane & 15;
  const bool xok = x0 + lp < p.W;
  float ssum[4][4], ssq[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int co = c0 + wc * 64 + ct * 16 + lc;
    float sv[4] = {1.f, 1.f, 1.f, 1.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (co + j < p.Cn) { sv[j] = p.scale[g * p.Cn + co + j]; hv[j] = p.shift[g * p.Cn + co + j]; }
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int yy = y0 + 4 * wp + pt;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = acc[ct][pt][j];
        if (p.scale) { u = u * sv[j] + hv[j]; if (p.act) u = silu_f(u); }
        v[j] = xok ? TT<T>::rnd(u) : 0.f;
        ssum[ct][j] += v[j];
        ssq[ct][j] += v[j] * v[j];
      }
      if (xok) {
        T* dst = Y + (((long)b * p.H + yy) * p.W + x0 + lp) * p.ysw + (long)g * p.Cn + co;
        if (co + 3 < p.Cn) {
          if (sizeof(T) == 2) {
            uint2 u;
            u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *(uint2*)dst = u;
          } else {
            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (co + j < p.Cn) TT<T>::st(dst + j, v[j]);
        }
      }
    }
  }
  if (p.part) {
    float* red = (float*)smem;  // [NW/2][128][2]; every LDS tile read finished at the last barrier
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = wave_xor_sum16(ssum[ct][j]);
        float q2 = wave_xor_sum16(ssq[ct][j]);
        if (lp == 0) {
          int cl = wc * 64 + ct * 16 + lc + j;
          red[(wp * 128 + cl) * 2 + 0] = s;
          red[(wp * 128 + cl) * 2 + 1] = q2;
        }
      }
    __syncthreads();
    if (tid < 128 && c0 + tid < p.Cn) {
      float s = 0.f, q2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW / 2; ++w) { s += red[(w * 128 + tid) * 2]; q2 += red[(w * 128 + tid) * 2 + 1]; }
      float* dst = p.part + ((long)tile_lin * (p.G * p.Cn) + g * p.Cn + c0 + tid) * 2;
      dst[0] = s;
      dst[1] = q2;
    }
  }
}

template <typename T, int TH>
int launch_tile(const C3P& p, hipStream_t st) {
  constexpr int NW = TH / 2, NT = NW * 64;
  constexpr int HCH = (TH + 2) * 18 * 8;
  size_t sm = 2 * (size_t)HCH * 16 + 2 * (size_t)(TH == 16 ? 2 : 1) * 16384;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_tile_kernel<T, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  dim3 grid(p.B * p.nty * p.ntx * p.ntc, 1, p.G);
  hipLaunchKernelGGL((conv3x3_tile_kernel<T, TH>), grid, dim3(NT), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

// tile height the resident-halo kernel would use for this geometry, 0 if the generic implicit GEMM must be used
int y3d_tile_height(int dtype, int H, int W, int Cg, int kh, int kw, int stride, int pad) {
  int cse = dtype == Y3D_BF16 ? 64 : 32;
  if (kh != 3 || kw != 3 || stride != 1 || pad != 1) return 0;
  if (Cg % cse != 0) return 0;
  if (W < 8) return 0;
  if (H % 16 == 0) return 16;
  if (H % 8 == 0) return 8;
  // TH = 4 (two waves per workgroup) is built and tested but measured slower than the generic implicit GEMM on 20x20 maps
  // (262 vs 691 TFLOP/s, 512->2048, B=32): too few waves per CU to hide the DMA / LDS latency
  return 0;
}

int y3d_conv3x3_tile_launch(int dtype, int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G,
                            const void* w, int Ktot, void* y, long ysw, float* part, int flip, const float* scale, const float* shift, int act,
                            void* stream) {
  C3P p;
  p.x = x; p.w = w; p.y = y; p.part = part; p.scale = scale; p.shift = shift; p.act = act;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.H = H; p.W = W; p.Cg = Cg; p.Cn = Cn; p.G = G; p.Ktot = Ktot;
  p.ntx = cdiv(W, 16); p.nty = H / th; p.ntc = cdiv(Cn, 128); p.flip = flip;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) {
    if (th == 16) return launch_tile<bf16_t, 16>(p, st);
    if (th == 8) return launch_tile<bf16_t, 8>(p, st);
    return launch_tile<bf16_t, 4>(p, st);
  }
  if (th == 16) return launch_tile<float, 16>(p, st);
  if (th == 8) return launch_tile<float, 8>(p, st);
  return launch_tile<float, 4>(p, st);
}
