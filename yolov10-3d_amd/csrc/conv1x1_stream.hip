// 1x1 stride-1 convolution (and its data gradient) as a STREAMING GEMM:  y[m][n] = sum_k x[m][k] * w[n][k]   (bf16, fp32 accumulate).
//
// Reference: nn/modules/conv.py:120-122 (`Conv.forward`, k = 1) - C2f/SCDown/PSA/SPPF cv1/cv2, about half of the body's launches.
// These layers carry 64..512 flop per byte: on this part they are bound by how many bytes a CU keeps in flight, not by the matrix
// cores.  The generic implicit-GEMM kernel (conv_gemm.hip) gives each workgroup one 128-pixel tile: two to twelve K steps, one
// register-staged prefetch deep, then the workgroup retires - measured 2.2 TB/s of fabric traffic and 150-260 TFLOP/s.  Here:
//   * workgroups are PERSISTENT (one or two per CU) and walk pixel tiles m = worker, worker + nworkers, ...;
//   * the weight tile [BN][K] is loaded into LDS ONCE per workgroup and stays there (K <= 768 at BN = 64, K <= 384 at BN = 128 ...);
//   * the pixel operand streams through a ring of NS stages (128 pixels x 64 channels = 16 KB each) by LDS-DMA
//     (buffer_load ... lds, 16 B per lane, bank swizzle applied on the GLOBAL side: lane l of a DMA instruction fetches chunk
//     (l & 7) ^ (l >> 3) of row l >> 3), NS - 1 stages in flight ACROSS tile boundaries, one counted s_waitcnt + one raw barrier
//     per stage;
//   * rows past the end of the tensor get an out-of-range buffer offset and read zeros (hardware range check);
//   * vmcnt counts loads, stores and LDS-DMA together, in issue order (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)"): the epilogue's
//     stores sit in the same FIFO BEHIND the DMA rounds of the next stages.  A fixed wait count would make every tile boundary wait
//     for its own stores and for the whole ring (measured: the forward with statistics ran 1.5x the data gradient).  So every store
//     is a buffer store issued by EVERY lane (dead lanes get an out-of-range offset): the number of stores a wave issued in the last
//     NS - 1 stages is known exactly, and the wait count is 2 (NS - 2) plus that number;
//   * the epilogue is the generic kernel's (bias / folded affine + SiLU / bf16 rounding / 8-byte stores / BatchNorm partial sums
//     - ONE row per worker: y3d_conv2d_stat_rows returns the worker count for the shapes this kernel takes, so the finalize pass folds
//     at most 512 rows instead of one per 128-pixel tile: 25 600 on the 320x320 layer, 90 us of bn_finalize);
// The K order of the accumulation (32-element MFMA steps, ascending) is the generic kernel's, so the outputs are bit-identical to it.
#include "common.h"
#include "conv_frag.h"
#include <cstdlib>

namespace {

struct PwP {
  const bf16_t* x;
  const bf16_t* w;  // packed [N][Kpad]
  const float *bias, *scale, *shift;
  bf16_t* y;
  float* part;  // [nmt][N][2] or null
  long ysw;
  int xsw;      // pixel stride of x in elements
  int M, K, N, Kpad;
  int nkc, nmt, nnt, act;
  int G, PC;    // groups (blockIdx.y; K, N, Kpad are per group, the operands of group g start g * K / g * N channels further), channels of a partial row
  unsigned xbytes, ybytes, pbytes, wbytes;
};

template <int N> __device__ __forceinline__ void pw_wvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt takes an immediate: a uniform run-time count goes through a switch (vmcnt is a 6-bit field)
__device__ __forceinline__ void pw_wvm_n(int n) {
#define PW_C(i) case i: pw_wvm<i>(); break;
#define PW_C8(i) PW_C(i) PW_C(i + 1) PW_C(i + 2) PW_C(i + 3) PW_C(i + 4) PW_C(i + 5) PW_C(i + 6) PW_C(i + 7)
  switch (n) {
    PW_C8(0) PW_C8(8) PW_C8(16) PW_C8(24) PW_C8(32) PW_C8(40) PW_C8(48) PW_C(56) PW_C(57) PW_C(58) PW_C(59) PW_C(60) PW_C(61) PW_C(62)
    default: pw_wvm<63>(); break;
  }
#undef PW_C8
#undef PW_C
}
typedef __attribute__((ext_vector_type(2))) unsigned pw_u32x2;

// AFF: bias or folded affine (+ SiLU) in the epilogue.  A template parameter, not a run-time test: the compiler puts the
// s_waitcnt vmcnt(0) for those per-channel loads at the join point of the branch, where it runs - and drains the ring - even
// when nothing was loaded.
// WRES: the weight tile [BN][K] is resident in LDS; otherwise (K x BN too large: the 40x40 / 20x20 layers with K, N >= 256) its 64-deep
// chunk travels in the ring next to the pixel chunk (a stage is 16 KB of pixels + BN x 128 B of weights, the weights out of L2).
template <int BN, int WP, int WC, int NS, bool AFF, bool WRES>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(PwP p) {
  {  // grouped 1x1 (the M head's second layer: 14 groups of 64 -> 64): blockIdx.y picks the group, everything below sees ITS operands
    const int g = blockIdx.y;
    p.x += (long)g * p.K;
    p.w += (long)g * p.N * p.Kpad;
    p.y += (long)g * p.N;
    if (p.bias) p.bias += g * p.N;
    if (p.scale) { p.scale += g * p.N; p.shift += g * p.N; }
    if (p.part) p.part += (long)g * p.N * 2;
  }
  constexpr int BM = 128;
  constexpr int TP = BM / WP / 16, TC = BN / WC / 16;
  static_assert(WP * WC == 8 && TP >= 1 && TC >= 1, "8 waves");
  constexpr int STAGE = BM * 128 + (WRES ? 0 : BN * 128);
  constexpr int ND = 2 + (WRES ? 0 : BN / 64);   // DMA instructions per wave per stage
  static_assert(WRES || BN % 64 == 0, "streamed weight tiles: whole DMA instructions per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                              // [nkc][BN][128 B]  (resident weights)
  char* sA = smem + (WRES ? (size_t)p.nkc * BN * 128 : 0);      // [NS][128 pixel rows | BN weight rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WP, wc = wave / WP;
  // XCD x gets workers x, x + 8, ...; the nnt channel tiles of one worker sit on the same XCD (they stream the same pixels)
  const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
  const int nt = idx % p.nnt, wk = (idx / p.nnt) * 8 + xcd, nwk = (gridDim.x / p.nnt);
  const int n0 = nt * BN;
  const int mine = wk < p.nmt ? (p.nmt - wk + nwk - 1) / nwk : 0;
  const int S = mine * p.nkc;
  if (S == 0) {  // a worker without tiles (the grid is a multiple of 8): its row of BatchNorm partials is zeros
    if (p.part && tid < BN && n0 + tid < p.N) *(float2*)(p.part + ((long)wk * p.PC + n0 + tid) * 2) = make_float2(0.f, 0.f);
    return;
  }

  // ---- resident weights ----------------------------------------------------------------------------------------------------------
  if (WRES)
  for (int i = tid; i < p.nkc * BN * 8; i += 512) {
    const int c = i & 7, r = (i >> 3) % BN, kc = i / (8 * BN);
    const int k = kc * 64 + c * 8, n = n0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < p.N && k < p.Kpad) v = *(const uint4*)(p.w + (long)n * p.Kpad + k);
    *(uint4*)(sW + (kc * BN + r) * 128 + ((c ^ (r & 7)) << 4)) = v;
  }

  // ---- DMA issue cursor ----------------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;  // row inside an 8-row DMA instruction, global chunk it fetches
  int is = 0, i_tile = wk, i_kc = 0, i_slot = 0;
  auto issue = [&]() {
    const bool live = is < S;
    const int k = i_kc * 64 + lchunk * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 16 * wave + 8 * h;
      const int m = i_tile * BM + r + lrow;
      const bool ok = live & (m < p.M) & (k < p.K);
      const unsigned off = (unsigned)(m * p.xsw + k) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sA + i_slot * STAGE + r * 128), 16, ok ? off : OOB, 0, 0, 0);
    }
    if (!WRES) {
#pragma unroll
      for (int h = 0; h < BN / 64; ++h) {
        const int r = (BN / 8) * wave + 8 * h;  // weight rows of this instruction: r .. r + 7
        const int n = n0 + r + lrow;
        const bool ok = live & (n < p.N) & (k < p.K);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sA + i_slot * STAGE + (BM + r) * 128), 16,
                                                 ok ? (unsigned)(n * p.Kpad + k) * 2u : OOB, 0, 0, 0);
      }
    }
    ++is;
    if (++i_kc == p.nkc) { i_kc = 0; i_tile += nwk; }
    if (++i_slot == NS) i_slot = 0;
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue();

  f32x4_t acc[TC][TP];
  const int lc = (lane >> 4) * 4, lp = lane & 15;
  int c_tile = wk, c_kc = 0, c_slot = 0;
  int hist[NS - 1];  // stores this wave issued in each of the last NS - 1 stages (younger than the DMA round the next wait is for)
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) hist[i] = 0;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.ybytes, 0x00020000);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // weights visible after the first stage's barrier (a __syncthreads() here would drain the DMA rounds)

  // BatchNorm partial sums: per-lane over ALL tiles of this worker, folded once at the end into the worker's row (the finalize pass sums
  // rows, in double).  A per-tile fold (16-lane DPP chain,
  // LDS hand-off, store) sat on the critical path of every stage: with K = 32 a stage is only two MFMAs per wave.
  float ssum[TC][4], ssq[TC][4];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }

#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    {
      int young = ND * (NS - 2);
#pragma unroll
      for (int i = 0; i < NS - 1; ++i) young += hist[i];
      pw_wvm_n(young);
#pragma unroll
      for (int i = 0; i < NS - 2; ++i) hist[i] = hist[i + 1];
      hist[NS - 2] = 0;
    }
    __builtin_amdgcn_s_barrier();
    issue();
    if (c_kc == 0) {
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const char* tA = sA + c_slot * STAGE + (wp * (BM / WP)) * 128;
    const char* tW = WRES ? sW + (c_kc * BN + wc * (BN / WC)) * 128 : sA + c_slot * STAGE + (BM + wc * (BN / WC)) * 128;
    const int nks = p.K - c_kc * 64 > 32 ? 2 : 1;  // K is a multiple of 8: chunks past K were zero-filled by the range check
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8_t fb[TP], fa[TC];
#pragma unroll
      for (int b = 0; b < TP; ++b) fb[b] = Frag<bf16_t>::load(tA, b * 16, ks, lane);
#pragma unroll
      for (int a = 0; a < TC; ++a) fa[a] = Frag<bf16_t>::load(tW, a * 16, ks, lane);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (++c_slot == NS) c_slot = 0;
    if (++c_kc < p.nkc) continue;
    c_kc = 0;
    // ---- epilogue of tile c_tile -------------------------------------------------------------------------------------------------
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      const int cl = wc * (BN / WC) + a * 16 + lc, co = n0 + cl;
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {1.f, 1.f, 1.f, 1.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
      if (AFF && p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (co + j < p.N) bv[j] = p.bias[co + j];
      }
      if (AFF && p.scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (co + j < p.N) { sv[j] = p.scale[co + j]; hv[j] = p.shift[co + j]; }
      }
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        const int m = c_tile * BM + wp * (BM / WP) + b * 16 + lp;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = acc[a][b][j];
          if (AFF) {
            u += bv[j];
            if (p.scale) { u = u * sv[j] + hv[j]; if (p.act) u = silu_f(u); }
          }
          v[j] = bf2f(f2bf(u));
          ssum[a][j] += v[j];
          ssq[a][j] += v[j] * v[j];
        }
        // N % 4 == 0 (launcher): the four channels of a lane are all inside or all outside
        const pw_u32x2 u = {(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
        __builtin_amdgcn_raw_buffer_store_b64(u, ry, ((m < p.M) & (co < p.N)) ? (unsigned)(m * (int)p.ysw + co) * 2u : OOB, 0, 0);
      }
    }
    hist[NS - 2] += TC * TP;
    c_tile += nwk;
  }
  pw_wvm<0>();  // the trailing (out-of-range) DMA rounds must land before the LDS is released
  if (p.part) {
    // rows >= M contributed exact zeros (zero-filled operands, no bias together with statistics)
    __syncthreads();
    float* red = (float*)sA;  // [WP][BN][2]: the ring is idle now
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s = wave_xor_sum16(ssum[a][j]), q = wave_xor_sum16(ssq[a][j]);
        if (lp == 0) {
          const int cl = wc * (BN / WC) + a * 16 + lc + j;
          red[(wp * BN + cl) * 2] = s;
          red[(wp * BN + cl) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WP; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
      float* dst = p.part + ((long)wk * p.PC + n0 + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

struct PwPlan {
  int bn, ns, nnt, grid, wres;
  size_t lds;
};

// Resident weights: BN = the smallest tile covering N whose weights leave room for a 3-stage ring; otherwise the largest that fits.
// More than two channel tiles would stream the pixels again and again (and a 64-channel tile with a six-stage ring measured 20-60 %
// slower than the generic kernel on the 40x40 / 20x20 layers): those layers stream their weight chunk through the ring instead.
bool pw_plan(int M, int K, int N, PwPlan* pl, int G = 1) {
  const int nkc = cdiv(K, 64);
  const size_t cap = 160 * 1024;
  const int nmt = cdiv(M, 128);
  const int bns[4] = {256, 128, 64, 32};
  int bn = 0;
  for (int i = 0; i < 4; ++i) {
    const int b = bns[i];
    if (i < 3 && bns[i + 1] >= N) continue;  // a smaller tile still covers N
    if ((size_t)b * nkc * 128 + 3 * 16384 <= cap) { bn = b; break; }
  }
  const bool can_stream = N > 64 && nkc >= 2;
  // two resident channel tiles stream the pixels twice: one streamed 256-channel tile measured 15-20 % faster at 40x40 (384 -> 256,
  // 256 -> 256), slower on the 20x20 maps (too few pixel tiles to amortise the weight chunks)
  bool wres = bn != 0 && (cdiv(N, bn) == 1 || (cdiv(N, bn) == 2 && nmt < 200));
  if (!wres && !can_stream) {
    if (!bn) return false;
    wres = true;
  }
  pl->wres = wres;
  if (wres) {
    const size_t wb = (size_t)bn * nkc * 128;
    pl->bn = bn;
    pl->ns = wb + 4 * 16384 <= cap ? 4 : 3;
    pl->lds = wb + pl->ns * 16384;
  } else {
    pl->bn = N > 128 ? 256 : 128;
    pl->ns = pl->bn == 256 ? 3 : 4;
    pl->lds = (size_t)pl->ns * (16384 + pl->bn * 128);
  }
  pl->nnt = cdiv(N, pl->bn);
  const int per_cu = pl->lds <= 80 * 1024 ? 2 : 1;
  int wpx = (32 * per_cu) / (pl->nnt * G);  // workers per XCD (and group)
  if (wpx < 1) wpx = 1;
  if (wpx > cdiv(nmt, 8)) wpx = cdiv(nmt, 8);
  pl->grid = 8 * wpx * pl->nnt;
  return true;
}

template <int BN, int WP, int WC, int NS, bool AFF, bool WRES>
void pw_launch_1(const PwP& p, const PwPlan& pl, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv1x1_stream_kernel<BN, WP, WC, NS, AFF, WRES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((conv1x1_stream_kernel<BN, WP, WC, NS, AFF, WRES>), dim3(pl.grid, p.G), dim3(512), pl.lds, st, p);
}

template <int BN, int WP, int WC>
void pw_launch_ns(const PwP& p, const PwPlan& pl, hipStream_t st) {
  const bool aff = p.bias || p.scale;
  if (pl.ns == 4) { if (aff) pw_launch_1<BN, WP, WC, 4, true, true>(p, pl, st); else pw_launch_1<BN, WP, WC, 4, false, true>(p, pl, st); }
  else { if (aff) pw_launch_1<BN, WP, WC, 3, true, true>(p, pl, st); else pw_launch_1<BN, WP, WC, 3, false, true>(p, pl, st); }
}

}  // namespace

static int g_stream1x1 = 1;

extern "C" int y3d_get_stream1x1(void) { return g_stream1x1; }

extern "C" int y3d_set_stream1x1(int enable) {
  const int old = g_stream1x1;
  g_stream1x1 = enable ? 1 : 0;
  return old;
}

// Does the streaming kernel take this GEMM?  (bf16, K a multiple of 32, dense pixel rows, 32-bit byte offsets, at most `max_nnt`
// channel tiles: every extra channel tile streams the pixel operand again)
int y3d_conv1x1_stream_ok(int dtype, long M, int K, int N, long xsw, int G) {
  if (!g_stream1x1 || dtype != Y3D_BF16 || K % 8 != 0 || K < 32 || N < 8 || N % 4 != 0 || M < 128 || G < 1 || G > 64) return 0;
  if ((M * xsw + (long)G * K) * 2 >= (1L << 32) - 64 || M * N * G * 2 >= (1L << 31)) return 0;
  PwPlan pl;
  if (!pw_plan((int)M, K, N, &pl, G)) return 0;
  return 1;
}

// rows of BatchNorm partials the kernel writes for this GEMM (= its worker count)
int y3d_conv1x1_stream_rows(long M, int K, int N, int G) {
  PwPlan pl;
  if (!pw_plan((int)M, K, N, &pl, G)) return 0;
  return pl.grid / pl.nnt;
}

int y3d_conv1x1_stream_launch(const void* x, long xsw, const void* w, int Kpad, const float* bias, const float* scale, const float* shift, int act,
                              void* y, long ysw, float* part, long M, int K, int N, int G, void* stream) {
  PwPlan pl;
  Y3D_CHECK(pw_plan((int)M, K, N, &pl, G), "conv1x1_stream: no tile plan for K=%d N=%d", K, N);
  Y3D_CHECK(((uintptr_t)x & 15) == 0 && xsw % 8 == 0 && ((uintptr_t)w & 15) == 0 && Kpad % 8 == 0 && ((uintptr_t)y & 7) == 0 && ysw % 4 == 0,
            "conv1x1_stream: operand alignment");
  PwP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y; p.part = part;
  p.ysw = ysw; p.xsw = (int)xsw; p.M = (int)M; p.K = K; p.N = N; p.Kpad = Kpad;
  p.nkc = cdiv(K, 64); p.nmt = cdiv(M, 128); p.nnt = pl.nnt; p.act = act; p.G = G; p.PC = G * N;
  p.xbytes = (unsigned)(((M - 1) * xsw + K) * 2);
  Y3D_CHECK(((M - 1) * ysw + (long)G * N) * 2 < (1L << 32) - 64 && M * ysw < (1L << 31), "conv1x1_stream: output tensor beyond 32-bit byte offsets");
  p.ybytes = (unsigned)(((M - 1) * ysw + N) * 2);
  p.pbytes = part ? (unsigned)((long)p.nmt * N * 8) : 0u;
  hipStream_t st = (hipStream_t)stream;
  p.wbytes = (unsigned)((long)N * Kpad * 2);
  if (!pl.wres) {
    const bool aff = bias || scale;
    if (pl.bn == 256) { if (aff) pw_launch_1<256, 2, 4, 3, true, false>(p, pl, st); else pw_launch_1<256, 2, 4, 3, false, false>(p, pl, st); }
    else { if (aff) pw_launch_1<128, 4, 2, 4, true, false>(p, pl, st); else pw_launch_1<128, 4, 2, 4, false, false>(p, pl, st); }
    Y3D_LAUNCH_CHECK();
    return Y3D_OK;
  }
  switch (pl.bn) {
    case 256: pw_launch_ns<256, 2, 4>(p, pl, st); break;
    case 128: pw_launch_ns<128, 4, 2>(p, pl, st); break;
    case 64: pw_launch_ns<64, 8, 1>(p, pl, st); break;
    default: pw_launch_ns<32, 8, 1>(p, pl, st); break;
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}
