// 3x3 stride-1 "same" convolution with the input halo tile RESIDENT in LDS — the fast path for the shapes that carry the
// YOLOv10-3D step (head 3x3 convs: 83 % of S-3D forward FLOPs; Bottleneck 3x3s of the body).
//
// One workgroup owns TH x 16 output pixels x 128 output channels of one group.  Per 64-channel (bf16; 32 for fp32) input slab
// the (TH+2) x 18 halo tile is brought into LDS ONCE and all nine filter taps read their shifted windows straight out of it
// as MFMA B-operand fragments (no im2col copy anywhere, each input pixel fetched once per slab instead of nine times); the
// 128 x 64 weight tile of each tap streams through an LDS ring (4 slots at TH=16: three taps in flight).  Both are filled by
// LDS-DMA (`global_load_lds_dwordx4`: no VGPR staging, no ds_write); the XOR swizzle that keeps `ds_read_b128` conflict-free is
// applied on the per-lane SOURCE address (the DMA destination is lane-linear), zero padding comes from a 16-byte zero page.
//
// Pipeline (one stage = one filter tap = 32 MFMAs per wave at bf16): the nine taps of a slab are unrolled, so every LDS offset
// and every wait count is a compile-time constant.  The DMA of tap t+3 and one round of the NEXT slab's halo are issued at the
// top of stage t; the stage ends with a COUNTED `s_waitcnt vmcnt(N)` that only retires tap t+2 (the younger loads stay in
// flight across the raw `s_barrier`), so tap t+1's weight fragments can be prefetched during stage t like the halo fragments.
//
// Output channels are permuted inside the MFMA row index (row r of channel tile ct is channel (r>>2)*16 + ct*4 + (r&3)) so a
// lane ends with 16 CONSECUTIVE channels of one pixel: two 16-byte stores per pixel row instead of eight 8-byte ones.
//
// The data gradient of such a conv is the same kernel on dy with the taps flipped (`flip`).
#include "common.h"
#include <cstdlib>
#include "conv_frag.h"

namespace {

__device__ uint4 y3d_zero_page[4];  // zero-initialised: source of every padded / out-of-range 16-byte chunk
#ifdef Y3D_PROBE_STAMP
__device__ unsigned long long y3d_probe_stamps[16384 * 4];
#define Y3D_STAMP(i) if (threadIdx.x == 0 && blockIdx.z * gridDim.x + blockIdx.x < 16384) y3d_probe_stamps[(blockIdx.z * gridDim.x + blockIdx.x) * 4 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define Y3D_STAMP(i)
#endif

struct C3P {
  const void* x;
  const void* w;   // packed [G][Cn][9][Cg] (forward) or the dgrad packing; row pitch Ktot
  void* y;
  float* part;     // optional BN partials [B*nty*ntx][G*Cn][2]
  const float* scale;  // per-channel affine (+SiLU) epilogue (eval-mode BatchNorm), EPI = 1 only
  const float* shift;
  int act;
  long xsb, xsh, xsw, ysw;
  int B, H, W;
  int Cg, Cn, G;
  int Ktot;
  int ntx, nty, ntc;
  int flip;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// XOR value of weight-tile row n: conflict-free ds_read_b128 for the permuted row order the A fragments are read in
__device__ __forceinline__ int wswz(int n) { return (((n >> 1) & 1) << 2) | ((4 - ((n >> 4) & 3)) & 3); }

template <typename T, int TH, int EPI>
__global__ __launch_bounds__(TH * 32, TH == 16 ? 1 : 2) void conv3x3_tile_kernel(C3P p) {
  constexpr int CE = TT<T>::CE;
  constexpr int CSE = TT<T>::BKE;          // channels per slab (128 bytes)
  constexpr int NW = TH / 2;               // waves: (TH/4) pixel-row groups x 2 channel halves
  constexpr int NT = NW * 64;
  constexpr int HWD = 18;
  constexpr int NPIX = (TH + 2) * HWD;
  constexpr int HCH = NPIX * 8;            // 16-byte chunks of one halo slab
  constexpr int HR = (HCH + NT - 1) / NT;  // DMA rounds per halo slab (the last one may be partial)
  constexpr int HFULL = HCH / NT;          // rounds every wave takes part in (what the wait counts may rely on)
  constexpr int HBYTES = HCH * 16;
  constexpr int WR = 1024 / NT;            // DMA rounds per 128 x 128 B weight tile
  constexpr int RD = TH == 16 ? 4 : 2;     // weight ring slots (TH = 8 keeps two workgroups per CU instead)
  constexpr int D = RD - 1;                // taps in flight
  constexpr bool EARLY = RD >= 4;          // tap t+1 is published one stage early: its fragments are prefetched inside stage t
  constexpr int KS = Frag<T>::KSUB;
  static_assert(HR <= 7 && KS % 2 == 0, "halo rounds must be issued by stage 6; fragment double buffer needs an even step count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;               // [2][HBYTES]
  char* sW = smem + 2 * HBYTES;  // [RD][16 KB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wc = wave & 1;
  const int g = blockIdx.z;
  Y3D_STAMP(0);
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
  const int nslab = p.Cg / CSE;

  // ---- DMA source pointers, fixed for the whole kernel (the slab / tap offset is added per issue) -------------------------
  const T* hsrc[HR];
#pragma unroll
  for (int rd = 0; rd < HR; ++rd) {
    int chunk = rd * NT + tid;
    int P = chunk >> 3, s = chunk & 7;
    int hy = P / HWD, hx = P - hy * HWD;
    int yy = y0 + hy - 1, xx = x0 + hx - 1;
    bool inb = chunk < HCH && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    hsrc[rd] = inb ? X + (long)b * p.xsb + (long)yy * p.xsh + (long)xx * p.xsw + (long)g * p.Cg + ((s ^ (hx & 7)) * CE) : nullptr;
  }
  const T* wsrc[WR];
#pragma unroll
  for (int rd = 0; rd < WR; ++rd) {
    int chunk = rd * NT + tid;
    int n = chunk >> 3, s = chunk & 7;
    wsrc[rd] = (c0 + n < p.Cn) ? Wt + ((long)(g * p.Cn + c0 + n)) * p.Ktot + ((s ^ wswz(n)) * CE) : nullptr;
  }
  // every wave issues exactly WR instructions per call (the wait counts depend on it); steps past the end read the zero page
  auto issue_w = [&](int slab, int tap, int slot) {
    const long off = (long)(p.flip ? 8 - tap : tap) * p.Cg + (long)slab * CSE;
    const bool live = slab < nslab;
#pragma unroll
    for (int rd = 0; rd < WR; ++rd) {
      const T* src = (live && wsrc[rd]) ? wsrc[rd] + off : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sW + slot * 16384 + (rd * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto issue_h = [&](int slab, int rd) {
    // rd is a compile-time constant after unrolling; in the last (partial) round the lanes past the end are masked off
    if (rd * NT + tid < HCH) {
      const T* src = (slab < nslab && hsrc[rd]) ? hsrc[rd] + (long)slab * CSE : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sH + (slab & 1) * HBYTES + (rd * NT + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];  // [channel tile][pixel-row tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing: a lane-dependent byte offset per (column shift q | K sub-step) + compile-time immediates ------------
  // halo rows are swizzled by their COLUMN (hx & 7): the 16 pixels of a fragment are 16 consecutive columns of one halo row
  const int lp16 = lane & 15;
  const int arow = wc * 64 + (lp16 >> 2) * 16 + (lp16 & 3);  // + ct * 4: weight-tile row of this lane's A-fragment row
  int bo[3][KS], ao[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int q = 0; q < 3; ++q) bo[q][ks] = (4 * wp * HWD + q + lp16) * 128 + Frag<T>::coff(ks, lane, (q + lp16) & 7);
    ao[ks] = arow * 128 + Frag<T>::coff(ks, lane, wswz(arow));  // wswz(arow + ct * 4) == wswz(arow)
  }
  typename Frag<T>::type fb[2][4], fa[2][4];
  auto load_b = [&](typename Frag<T>::type* dst, int slab, int tap, int ks) {
    const int r = tap / 3, q = tap - r * 3;
    const char* hb = sH + (slab & 1) * HBYTES + bo[q][ks];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) dst[pt] = Frag<T>::ld(hb + (pt + r) * (HWD * 128));
  };
  auto load_a = [&](typename Frag<T>::type* dst, int slot, int ks) {
    const char* wb = sW + slot * 16384 + ao[ks];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) dst[ct] = Frag<T>::ld(wb + ct * 512);
  };

  // ---- prologue: halo of slab 0, taps 0 .. D-1 ------------------------------------------------------------------------------
#pragma unroll
  for (int rd = 0; rd < HR; ++rd) issue_h(0, rd);
#pragma unroll
  for (int t = 0; t < D; ++t) issue_w(t / 9, t % 9, t % RD);
  wait_vm<EARLY ? (D - 2) * WR : (D - 1) * WR>();  // halo + tap 0 (+ tap 1 when it is read inside stage 0)
  __builtin_amdgcn_s_barrier();
  Y3D_STAMP(1);
  load_b(fb[0], 0, 0, 0);
  if (EARLY) load_a(fa[0], 0, 0);

#pragma unroll 1
  for (int k = 0; k < nslab; ++k) {
    const int kr = RD == 4 ? (k & 3) : (k & 1);  // ring slot of (k, tap) = (k * 9 + tap) % RD = (kr + tap) % RD  (9 % 4 == 1 % 2 == 1)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // ---- issue: one halo round of the next slab (its buffer was last read in slab k-1), then the weights of tap t+D ------
#ifndef Y3D_PROBE_NODMA
      if (t < HR) issue_h(k + 1, t);
      {
        const int t2 = t + D;
        issue_w(t2 < 9 ? k : k + 1, t2 < 9 ? t2 : t2 - 9, (kr + t2) % RD);
      }
#endif
      const int slot = (kr + t) % RD;
      if (!EARLY) load_a(fa[0], slot, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < KS) {
          load_b(fb[nxt], k, t, ks + 1);
          load_a(fa[nxt], slot, ks + 1);
        } else {
          // first sub-step of the next tap: the halo operand is stable (the next slab's halo was retired by an earlier stage's
          // wait); the weight operand only when its tap was published by the previous stage's barrier
          if (t < 8) load_b(fb[nxt], k, t + 1, 0); else load_b(fb[nxt], k + 1, 0, 0);
          if (EARLY) load_a(fa[nxt], (kr + t + 1) % RD, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) {
#ifdef Y3D_PROBE_NOMFMA
            asm volatile("" ::"v"(fa[cur][ct]), "v"(fb[cur][pt]));
#else
            acc[ct][pt] = Frag<T>::mma(fa[cur][ct], fb[cur][pt], acc[ct][pt]);
#endif
          }
      }
      // ---- retire tap t+1 (t+2 when EARLY): everything issued after it may stay in flight across the barrier -------------------
      // EARLY: tap t+2 was issued in stage t-1; younger: this stage's halo round (full rounds only) + this stage's weights
      if (EARLY) {
        if (t < HFULL) wait_vm<WR + 1>(); else wait_vm<WR>();
      } else {
        wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  wait_vm<0>();  // the dummy loads past the end must land before the LDS is reused or the workgroup retires

  // ---- epilogue: this lane holds channels cb .. cb+15 of pixel (y0 + 4 wp + pt, x0 + lp) ------------------------------------
  Y3D_STAMP(2);
  T* __restrict__ Y = (T*)p.y;
  const int lq = lane >> 4, lp = lane & 15;
  const int cl = wc * 64 + lq * 16;  // first of this lane's 16 channels inside the 128-channel tile
  const bool cok = c0 + cl < p.Cn;   // Cn % 16 == 0
  const bool xok = x0 + lp < p.W;
  float ssum[16], ssq[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
  float sv[16], hv[16];
  if (EPI == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = cok ? p.scale[g * p.Cn + c0 + cl + i] : 1.f; hv[i] = cok ? p.shift[g * p.Cn + c0 + cl + i] : 0.f; }
  }
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int yy = y0 + 4 * wp + pt;
    float v[16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = acc[ct][pt][j];
        if (EPI == 1) { u = u * sv[ct * 4 + j] + hv[ct * 4 + j]; if (p.act) u = silu_f(u); }
        u = xok ? TT<T>::rnd(u) : 0.f;
        v[ct * 4 + j] = u;
        if (EPI == 0) { ssum[ct * 4 + j] += u; ssq[ct * 4 + j] += u * u; }
      }
#ifdef Y3D_PROBE_NOEPI
    if (xok && cok && v[0] == 123.456f) {
#else
    if (xok && cok) {
#endif
      T* dst = Y + (((long)b * p.H + yy) * p.W + x0 + lp) * p.ysw + (long)g * p.Cn + c0 + cl;
      if (sizeof(T) == 2) {
        ((uint4*)dst)[0] = Chunk<T>::pack(v);
        ((uint4*)dst)[1] = Chunk<T>::pack(v + 8);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ((uint4*)dst)[i] = Chunk<T>::pack(v + 4 * i);
      }
    }
  }
  if (EPI == 0 && p.part) {
    float* red = (float*)smem;  // [NW/2][128][2]; all LDS tile reads finished before the last barrier
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s = wave_xor_sum16(ssum[i]);
      float q2 = wave_xor_sum16(ssq[i]);
      if (lp == i) {  // spread the 16 writes over the 16 lanes of the row
        red[(wp * 128 + cl + i) * 2 + 0] = s;
        red[(wp * 128 + cl + i) * 2 + 1] = q2;
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
  Y3D_STAMP(3);
}

template <typename T, int TH, int EPI>
int launch_tile(const C3P& p, hipStream_t st) {
  constexpr int NW = TH / 2, NT = NW * 64;
  constexpr int HCH = (TH + 2) * 18 * 8;
  constexpr int RD = TH == 16 ? 4 : 2;
  size_t sm = 2 * (size_t)HCH * 16 + (size_t)RD * 16384;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_tile_kernel<T, TH, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  dim3 grid(p.B * p.nty * p.ntx * p.ntc, 1, p.G);
  hipLaunchKernelGGL((conv3x3_tile_kernel<T, TH, EPI>), grid, dim3(NT), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

template <typename T, int TH>
int launch_tile_epi(const C3P& p, hipStream_t st) {
  return p.scale ? launch_tile<T, TH, 1>(p, st) : launch_tile<T, TH, 0>(p, st);
}

}  // namespace

// conv3x3_wide3.hip: the bf16 production kernel (persistent, 512-pixel tiles, SIMD partners half a phase apart); this file's kernel stays as
// the fp32 (parity mode) path and serves the bf16 shapes with too few 512-pixel tiles to fill the chip
int y3d_conv3x3_wide3_launch(int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G, const void* w,
                             int Ktot, void* y, long ysw, float* part, int flip, const float* scale, const float* shift, int act, void* stream);

// conv3x3_flat.hip: the same pipeline over the flattened padded pixel space (tiles of 512 consecutive positions): the better tiling where
// 16-pixel tile rows / 8- or 16-row tiles do not divide the map (40x40, 20x20).  Tile height code -1.
int y3d_conv3x3_flat_ok(int B, int H, int W, int Cg, int Cn, int G);
int y3d_conv3x3_flat_tiles(int B, int H, int W);
int y3d_conv3x3_flat_launch(const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G, const void* w, int Ktot, void* y,
                            long ysw, float* part, int flip, const float* scale, const float* shift, int act, void* stream);

// with fewer 512-pixel tiles than this (half the CUs) the 256-pixel tiles of this file's kernel fill the chip better
static int v2_max_tiles() { return 129; }

// tile height the resident-halo kernels would use for this geometry, 0 if the generic implicit GEMM must be used
int y3d_tile_height(int dtype, int B, int H, int W, int Cg, int Cn, int G, int kh, int kw, int stride, int pad) {
  int cse = 32;  // channels per K slab: 32 bf16 (64-byte rows, wide kernel) or 32 fp32 (128-byte rows)
  if (kh != 3 || kw != 3 || stride != 1 || pad != 1) return 0;
  // bf16: the persistent kernel (conv3x3_wide3.hip) zero-fills a partial last 32-channel slab: any multiple of 8 from 40 channels
  // on (80 -> 80 of the X widths ran on the generic implicit GEMM: 270 TFLOP/s); the fp32 tile kernel keeps whole slabs
  if (dtype == Y3D_BF16 ? (Cg % 8 != 0 || Cg < 40 || (Cg < 64 && Cg % 32 != 0 && Cn <= 64)) : (Cg % cse != 0 || Cg < 64)) return 0;
  if (Cn % 16 != 0) return 0;
  if (W < 8) return 0;
  if (dtype == Y3D_BF16 && y3d_conv3x3_flat_ok(B, H, W, Cg, Cn, G)) {
    // the flat tiling when it needs at least 4 % fewer tiles than the rectangular one (and enough of them to fill the chip).  Same-box A/B
    // (tools/conv_bench.py, B = 32): 512 -> 2048 @20x20 forward 0.341 -> 0.252 ms (710 -> 959 TFLOP/s), 256 -> 2048 @40x40 0.484 -> 0.459,
    // 16 x (128 -> 128) @40x40 0.292 -> 0.282; the 20x20 data gradients (124 tiles: one round of half the CUs either way) do not move
    const int thr = H % 16 == 0 ? 16 : 8;
    const long rect = (long)cdiv(B, 32 / thr) * cdiv(H, thr) * cdiv(W, 16), flat = y3d_conv3x3_flat_tiles(B, H, W);
    if (flat * 104 < rect * 100 && flat * G * cdiv(Cn, 128) >= v2_max_tiles()) return -1;
  }
  if (H % 16 == 0) return 16;
  if (H % 8 == 0) return 8;
  // ragged height (20x20: the stride-32 level of a 640x640 image): the persistent bf16 kernel runs ceil(H / 8) row tiles and masks the
  // rows past the map in its epilogue (20 rows in 24: 83 % useful MFMAs, against 440-650 TFLOP/s for the generic kernel on the P5
  // head layers) - when there are enough tiles to fill the chip; the 256-pixel tile kernel (smaller shapes) needs H % TH == 0
  if (dtype == Y3D_BF16 && H > 8 && (H % 8) >= 4 && (long)G * cdiv(B, 4) * cdiv(H, 8) * cdiv(W, 16) * cdiv(Cn, 128) >= v2_max_tiles()) return 8;
  return 0;
}

int y3d_conv3x3_tile_launch(int dtype, int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G,
                            const void* w, int Ktot, void* y, long ysw, float* part, int flip, const float* scale, const float* shift, int act,
                            void* stream) {
  C3P p;
  p.x = x; p.w = w; p.y = y; p.part = part; p.scale = scale; p.shift = shift; p.act = act;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.H = H; p.W = W; p.Cg = Cg; p.Cn = Cn; p.G = G; p.Ktot = Ktot;
  p.ntx = cdiv(W, 16); p.nty = cdiv(H, th); p.ntc = cdiv(Cn, 128); p.flip = flip;
  hipStream_t st = (hipStream_t)stream;
  if (th < 0) return y3d_conv3x3_flat_launch(x, xsb, xsh, xsw, B, H, W, Cg, Cn, G, w, Ktot, y, ysw, part, flip, scale, shift, act, stream);
  if (dtype == Y3D_BF16) {
    // the persistent kernel runs one 512-pixel tile per CU at a time: with fewer tiles than half the CUs (128 -> 128 @40x40, B = 32:
    // 120) the 256-pixel tiles of this file's kernel fill the chip better (628 / 671 against 448 / 482 TFLOP/s forward / dgrad)
    const long wide_tiles = (long)G * cdiv(B, 32 / th) * p.nty * p.ntx * p.ntc;
    if (wide_tiles < v2_max_tiles() && Cg % 64 == 0 && H % th == 0)  // this kernel's K slab is a 128-byte row: 64 bf16 channels
      return th == 16 ? launch_tile_epi<bf16_t, 16>(p, st) : launch_tile_epi<bf16_t, 8>(p, st);
    return y3d_conv3x3_wide3_launch(th, x, xsb, xsh, xsw, B, H, W, Cg, Cn, G, w, Ktot, y, ysw, part, flip, scale, shift, act, stream);
  }
  return th == 16 ? launch_tile_epi<float, 16>(p, st) : launch_tile_epi<float, 8>(p, st);
}
