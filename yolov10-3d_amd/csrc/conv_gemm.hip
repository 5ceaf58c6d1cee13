// Dense / grouped convolution as an im2col-free implicit GEMM on the CDNA4 matrix cores.
//
//   forward : y[p][co]   = sum_{tap,ci} x[gather(p,tap)][ci] * w[co][tap][ci]      (reference: nn/modules/conv.py:120-122,
//   dgrad   : dx[p][ci]  = sum_{tap,co} dy[gatherT(p,tap)][co] * w[co][tap][ci]      F.conv2d and its autograd backward)
//   wgrad   : dw[co][tap][ci] = sum_p dy[p][co] * x[gather(p,tap)][ci]
//
// Layout: activations NHWC (channels contiguous) so that for one filter tap a GEMM-K run is a
// contiguous channel vector; weights are pre-packed K-contiguous per output channel.  Tiles are
// staged global -> registers -> LDS (16-byte chunks, XOR-swizzled 128-byte rows, halo/zero padding
// resolved in the loader) and double buffered with one barrier per K step.  MFMA: 16x16x32 bf16
// (or the exact-f32 16x16x4 form for the fp32 parity mode); the weight tile is the MFMA A operand
// and the pixel tile the B operand, so each lane ends up with 4 consecutive output channels of one
// pixel (one 8/16-byte store).  The epilogue optionally emits per-block BatchNorm partial sums
// (sum, sum of squares per channel), which makes the BN statistics bitwise reproducible (no atomics).
#include "common.h"
#include "conv_frag.h"
#include <cstdlib>

namespace {

struct ConvP {
  const void* x;   // gathered tensor (fwd: input, dgrad: dy)
  const void* w;   // packed weights [G][Cn][Kpad]
  const float* bias;
  const float* scale;  // optional per-channel affine (+SiLU) epilogue: eval-mode BatchNorm folded into the conv launch
  const float* shift;
  int act;
  void* y;         // output tensor, pixel-dense
  float* part;     // optional BN partials [gridDim.x][G*Cn][2]
  long xsb, xsh, xsw;
  long ysw;
  int B, Hg, Wg;   // spatial dims of the gathered tensor
  int Hq, Wq;      // spatial dims of the output tensor
  int Cg, Cn, G;   // per-group channels: gathered (GEMM K side), output (GEMM N side)
  int kh, kw, stride, pad;
  int Ktot, Kpad;  // taps*Cg and its padding to a chunk multiple (row pitch of packed weights)
  int M;           // B*Hq*Wq
  int ntx, nty;    // tile grid (pixel tiles x channel tiles); the launch is 1-D over ntx*nty, remapped per XCD
};

// BP x BC output tile (pixels x channels), 256 threads = WP x WC waves.
// MODE 0: forward.  MODE 1: data gradient (transposed gather, strides through tap masks).  MODE 2: data gradient of a 3x3
// stride-2 pad-1 conv by PARITY CLASS (blockIdx.y = 2*(row parity) + column parity of the dx pixel): a dx pixel only receives
// the taps whose parity matches, 1 / 2 / 2 / 4 of the 9, so each class is a dense GEMM over its own taps instead of a 9-tap
// GEMM with 3/4 of the operands masked to zero (measured 69-130 TFLOP/s for MODE 1 on these layers).
template <typename T, int BP, int BC, int WP, int WC, int MODE>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvP p) {
  constexpr bool DGRAD = MODE != 0;
  constexpr int CE = TT<T>::CE;
  constexpr int BKE = TT<T>::BKE;
  constexpr int RA = BP / 32;  // pixel rows per thread in the loader
  constexpr int RB = BC / 32;
  constexpr int TP = BP / WP / 16;
  constexpr int TC = BC / WC / 16;
  static_assert(WP * WC == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                      // [2][BP][128B]  pixel tile
  char* sB = smem + 2 * BP * 128;       // [2][BC][128B]  weight tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave / WC, wc = wave % WC;
  const int g = blockIdx.z;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of tiles and
  // walk the channel tiles of one pixel tile back to back: the blocks that re-read the same input pixels share one L2.
  int tile_x, tile_y;
  {
    const int nwg = p.ntx * p.nty, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_y = lin % p.nty;
    tile_x = lin / p.nty;
  }
  const int p0 = tile_x * BP;
  const int c0 = tile_y * BC;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ Wt = (const T*)p.w;
  // MODE 2: the dx pixels of this parity class, its taps (rows r = py ? {0,2} : {1}, columns likewise) and its compacted K extent
  const int py = MODE == 2 ? (int)(blockIdx.y >> 1) : 0, px = MODE == 2 ? (int)(blockIdx.y & 1) : 0;
  const int Hc = MODE == 2 ? (p.Hq - py + 1) / 2 : p.Hq, Wc = MODE == 2 ? (p.Wq - px + 1) / 2 : p.Wq;
  const int Mc = MODE == 2 ? p.B * Hc * Wc : p.M;
  const int ntq = px ? 2 : 1;
  const int Kc = MODE == 2 ? (py ? 2 : 1) * ntq * p.Cg : p.Ktot;
  if (MODE == 2 && p0 >= Mc) return;  // the grid is sized for the largest class

  // ---- loader state: this thread owns chunk column cc of rows (tid>>3) + 32*i -------------------
  // Per row: a base pointer and a bit mask of the filter taps that land inside the gathered tensor, both computed once.
  // Per K step the address is base + (tap offset + channel), i.e. one 64-bit add and one mask test per 16-byte chunk.
  const int cc = tid & 7;
  const int r0 = tid >> 3;
  // dgrad: (t/s)*xsh == t*(xsh/s) for the taps the mask admits (no 64-bit division for the strides the models use)
  const long SH = !DGRAD || p.stride == 1 ? p.xsh : (p.stride == 2 ? p.xsh >> 1 : p.xsh / p.stride);
  const long SW = !DGRAD || p.stride == 1 ? p.xsw : (p.stride == 2 ? p.xsw >> 1 : p.xsw / p.stride);
  const T* aptr[RA];
  unsigned long long amask[RA];
  // pixel coordinates of the first row by division, of the following ones (32 pixels further each) by carry
  int rb_ = (p0 + r0) / (Hc * Wc), rh_, rw_;
  {
    const int rem = (p0 + r0) - rb_ * (Hc * Wc);
    rh_ = rem / Wc;
    rw_ = rem - rh_ * Wc;
  }
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = p0 + r0 + 32 * i;
    aptr[i] = X;
    amask[i] = 0ull;
    if (i) {
      rw_ += 32;
      while (rw_ >= Wc) { rw_ -= Wc; ++rh_; }
      while (rh_ >= Hc) { rh_ -= Hc; ++rb_; }
    }
    if (m < Mc) {
      int b = rb_;
      int hq = rh_;
      int wq = rw_;
      if (MODE == 2) { hq = 2 * hq + py; wq = 2 * wq + px; }
      int a_h = DGRAD ? hq + p.pad : hq * p.stride - p.pad;
      int a_w = DGRAD ? wq + p.pad : wq * p.stride - p.pad;
      aptr[i] = X + (long)b * p.xsb + (long)g * p.Cg + (long)a_h * SH + (long)a_w * SW;
      unsigned long long mk = 0ull;
      for (int r = 0; r < p.kh; ++r)
        for (int q = 0; q < p.kw; ++q) {
          bool ok;
          if (DGRAD) {
            // strides 1 and 2 without an integer division: 9 taps x 4 rows x 2 divisions per thread were most of a workgroup's life on
            // the short-K launches (3x3 stride-2 data gradient by parity class at 320x320: 25 600 workgroups of one to four K steps)
            int th = a_h - r, tw = a_w - q;
            int hh, ww;
            if (p.stride == 1) { hh = th; ww = tw; }
            else if (p.stride == 2) { hh = th >> 1; ww = tw >> 1; }
            else { hh = th / p.stride; ww = tw / p.stride; }
            ok = th >= 0 && tw >= 0 && hh * p.stride == th && ww * p.stride == tw && hh < p.Hg && ww < p.Wg;
          } else {
            int hh = a_h + r, ww = a_w + q;
            ok = hh >= 0 && ww >= 0 && hh < p.Hg && ww < p.Wg;
          }
          if (ok) mk |= 1ull << (r * p.kw + q);
        }
      amask[i] = mk;
    }
  }
  const T* bptr[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = c0 + r0 + 32 * i;
    bptr[i] = n < p.Cn ? Wt + ((long)(g * p.Cn + n)) * p.Kpad + cc * CE : nullptr;
  }
  // K position of this thread's chunk: k = kt*BKE + cc*CE -> (tap index, channel ci)
  // (MODE 2: kpos runs over the class's compacted K = (class tap j, channel); the j-th class tap is filter tap (kr, kq))
  int kpos = cc * CE;
  int kci = kpos % p.Cg;
  int ktap = kpos / p.Cg;
  int kr = ktap / p.kw, kq = ktap - kr * p.kw;
  int kj = ktap;
  auto class_tap = [&]() {
    const int jr = kj / ntq, jq = kj - jr * ntq;
    kr = py ? 2 * jr : 1;
    kq = px ? 2 * jq : 1;
    ktap = kr * 3 + kq;
  };
  if (MODE == 2) class_tap();

  const int nk = (Kc + BKE - 1) / BKE;
  uint4 ra[RA], rb[RB];

  auto gload = [&](int kt) {
    const bool kvalid = kpos < Kc;
    const long koff = (DGRAD ? -((long)kr * SH + (long)kq * SW) : ((long)kr * SH + (long)kq * SW)) + kci;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kvalid && ((amask[i] >> ktap) & 1ull)) v = *(const uint4*)(aptr[i] + koff);
      ra[i] = v;
    }
    const bool wvalid = MODE == 2 ? kvalid : kpos < p.Kpad;
    const long woff = MODE == 2 ? (long)ktap * p.Cg + kci - cc * CE : (long)kt * BKE;  // bptr already holds this thread's chunk column
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (wvalid && bptr[i]) v = *(const uint4*)(bptr[i] + woff);
      rb[i] = v;
    }
    // advance to the next K step
    kpos += BKE;
    kci += BKE;
    while (kci >= p.Cg) {
      kci -= p.Cg;
      if (MODE == 2) {
        ++kj;
        class_tap();
      } else {
        ++ktap;
        if (++kq == p.kw) { kq = 0; ++kr; }
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int r = r0 + 32 * i;
      *(uint4*)(sA + buf * BP * 128 + r * 128 + ((cc ^ (r & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int r = r0 + 32 * i;
      *(uint4*)(sB + buf * BC * 128 + r * 128 + ((cc ^ (r & 7)) << 4)) = rb[i];
    }
  };

  f32x4_t acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* tA = sA + cur * BP * 128 + (wp * (BP / WP)) * 128;
    const char* tB = sB + cur * BC * 128 + (wc * (BC / WC)) * 128;
#pragma unroll
    for (int ks = 0; ks < Frag<T>::KSUB; ++ks) {
      typename Frag<T>::type fb[TP], fa[TC];
#pragma unroll
      for (int b = 0; b < TP; ++b) fb[b] = Frag<T>::load(tA, b * 16, ks, lane);
#pragma unroll
      for (int a = 0; a < TC; ++a) fa[a] = Frag<T>::load(tB, a * 16, ks, lane);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = Frag<T>::mma(fa[a], fb[b], acc[a][b]);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  T* __restrict__ Y = (T*)p.y;
  const int lc = (lane >> 4) * 4;  // channel sub-offset inside a 16x16 tile
  const int lp = lane & 15;        // pixel sub-offset
  float ssum[TC][4], ssq[TC][4];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }

#pragma unroll
  for (int a = 0; a < TC; ++a) {
    const int co = c0 + wc * (BC / WC) + a * 16 + lc;  // first of 4 consecutive channels
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    float sv[4] = {1.f, 1.f, 1.f, 1.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (co + j < p.Cn) bv[j] = p.bias[g * p.Cn + co + j];
    }
    if (p.scale) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (co + j < p.Cn) { sv[j] = p.scale[g * p.Cn + co + j]; hv[j] = p.shift[g * p.Cn + co + j]; }
    }
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int m = p0 + wp * (BP / WP) + b * 16 + lp;
      long opix = m;  // output pixel index (pixel-dense tensor)
      if (MODE == 2) {
        const int bb = m / (Hc * Wc), rem = m - bb * (Hc * Wc), hy = rem / Wc, wx = rem - hy * Wc;
        opix = ((long)bb * p.Hq + 2 * hy + py) * p.Wq + 2 * wx + px;
      }
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = acc[a][b][j] + bv[j];
        if (p.scale) { u = u * sv[j] + hv[j]; if (p.act) u = silu_f(u); }
        v[j] = TT<T>::rnd(u);
        ssum[a][j] += v[j];
        ssq[a][j] += v[j] * v[j];
      }
      if (m < Mc) {
        T* dst = Y + opix * p.ysw + (long)g * p.Cn + co;
        if (co + 3 < p.Cn && ((p.Cn | p.ysw) & 3) == 0) {
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
    // rows >= M and channels >= Cn contributed exact zeros (zero-filled operands, no bias with stats)
    float* red = (float*)smem;  // [WP][BC][2], safe: all LDS tile reads finished at the last barrier
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = wave_xor_sum16(ssum[a][j]);
        float q = wave_xor_sum16(ssq[a][j]);
        if (lp == 0) {
          int cl = wc * (BC / WC) + a * 16 + lc + j;
          red[(wp * BC + cl) * 2 + 0] = s;
          red[(wp * BC + cl) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < BC && c0 + tid < p.Cn) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WP; ++w) { s += red[(w * BC + tid) * 2]; q += red[(w * BC + tid) * 2 + 1]; }
      float* dst = p.part + ((long)tile_x * (p.G * p.Cn) + g * p.Cn + c0 + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad: slab[split][g][co][k] = sum over the split's pixel range of dy[p][co] * x[gather(p, tap(k))][ci(k)]
// ---------------------------------------------------------------------------------------------------
struct WgradP {
  const void* x;
  const void* dy;
  float* slab;     // [nsplit][G*Cn][Ktot]
  long xsb, xsh, xsw;
  long dsw;        // dy pixel stride (pixel-dense)
  int B, H, W, Ho, Wo;
  int Cg, Cn, G;
  int kh, kw, stride, pad;
  int Ktot, M, nsplit, chunk_px;  // chunk_px: pixels per split (multiple of BPK)
};

// LDS operand tiles are [pixel][W channels] with a padded row pitch; fragments are read transposed (the reduction runs over
// pixels).  W = 128 / 64 / 32 channels: narrow layers (the stem, 32/64-channel 1x1 convs) get narrow tiles, so that their
// workgroups carry only useful bytes and several of them fit a CU (their pixel loop is latency-bound).
template <typename T, int W> struct TFrag;
template <int W> struct TFrag<bf16_t, W> {
  // one MFMA k-sub-step = 32 pixels.  A/B fragment: two transposed 4x16 block reads.  The pixel <-> MFMA-k assignment is a free
  // permutation of the reduction index (A and B use the same one): k = 8g+j  <->  pixel row k0 + 16*(j>>2) + 4g + (j&3), so
  // that the two 16-lane groups of a 32-lane half read 8 CONSECUTIVE rows per instruction; the pitch (in dwords) is 8 * odd, so
  // those 8 rows x 32 bytes cover all 64 banks once.
  typedef bf16x8_t type;
  static constexpr int KSUB_PX = 32;
  static constexpr int PITCH = W * 2 + 32;  // 288 / 160 / 96 bytes
  __device__ static __forceinline__ type load(const char* tile, int i0, int k0, int lane) {
    int grp = lane >> 4, li = lane & 15;
    int q = li >> 2, pp = li & 3;
    const char* a0 = tile + (k0 + 4 * grp + q) * PITCH + (i0 + 4 * pp) * 2;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 16 * PITCH));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
  __device__ static __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <int W> struct TFrag<float, W> {
  // one MFMA k-sub-step = 4 pixels; row r+1 lands 16 banks after row r: the 2 rows x 16 floats of a half are conflict-free
  typedef float type;
  static constexpr int KSUB_PX = 4;
  static constexpr int PITCH = W * 4 + 64;  // 576 / 320 / 192 bytes
  __device__ static __forceinline__ type load(const char* tile, int i0, int k0, int lane) {
    return *(const float*)(tile + (k0 + (lane >> 4)) * PITCH + (i0 + (lane & 15)) * 4);
  }
  __device__ static __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// WD (co) x WX (k index) output tile, reduction over pixels in steps of BPK; 4 waves as 2x2 quadrants.
template <typename T, int WD, int WX>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  constexpr int CE = TT<T>::CE;
  constexpr int BPK = TT<T>::BKE;            // 64 pixels (bf16) / 32 pixels (fp32) per step
  constexpr int PD = TFrag<T, WD>::PITCH, PX = TFrag<T, WX>::PITCH;  // LDS row pitches (padded against bank conflicts)
  constexpr int CPD = WD * sizeof(T) / 16, CPX = WX * sizeof(T) / 16;  // chunks per row
  constexpr int ND = BPK * CPD / 256 > 0 ? BPK * CPD / 256 : 1;        // chunks per thread per step (4 / 2 / 1)
  constexpr int NX = BPK * CPX / 256 > 0 ? BPK * CPX / 256 : 1;
  constexpr int TA = WD / 32, TB = WX / 32;  // 16 x 16 sub-tiles per wave quadrant
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sD = smem;                   // [2][BPK][PD]  dy tile  (pixel x co)
  char* sX = smem + 2 * BPK * PD;    // [2][BPK][PX]  x tile   (pixel x k)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int g = blockIdx.z % p.G;
  const int split = blockIdx.z / p.G;
  const int j0 = blockIdx.x * WX;  // k-index tile origin
  const int i0 = blockIdx.y * WD;  // co tile origin
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ D = (const T*)p.dy;

  const int mbeg = split * p.chunk_px;
  const int mend = min(p.M, mbeg + p.chunk_px);
  const int nsteps = (mend - mbeg + BPK - 1) / BPK;

  // chunk q = tid + 256*i of an operand tile -> row q / CPR, column chunk q % CPR (CPR divides 256: the column is fixed per thread;
  // fp32 tiles of 32 pixels x 32 channels have fewer chunks than threads: the upper threads idle for that operand)
  const int ccd = tid % CPD, rd0 = tid / CPD;
  const int ccx = tid % CPX, rx0 = tid / CPX;
  constexpr int RSD = 256 / CPD, RSX = 256 / CPX;
  const bool tdv = BPK * CPD >= 256 || tid < BPK * CPD, txv = BPK * CPX >= 256 || tid < BPK * CPX;
  // x-tile column: k index -> (tap, ci) fixed for the whole kernel
  const int kx = j0 + ccx * CE;
  const bool kx_ok = txv && kx < p.Ktot;
  int tap = 0, ci = 0, tr = 0, tq = 0;
  if (kx_ok) { tap = kx / p.Cg; ci = kx - tap * p.Cg; tr = tap / p.kw; tq = tap - tr * p.kw; }
  const int cd = i0 + ccd * CE;  // dy-tile column: output channel
  const bool cd_ok = tdv && cd < p.Cn;
  const int HW = p.Ho * p.Wo;

  uint4 rd[ND], rx[NX];
  // per-row pixel coordinates of the x rows, decoded once and advanced by BPK pixels per step (no divisions in the loop)
  int rb_[NX], rh_[NX], rw_[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    int m = mbeg + rx0 + RSX * i;
    int b = m / HW;
    int rem = m - b * HW;
    rb_[i] = b;
    rh_[i] = rem / p.Wo;
    rw_[i] = rem - rh_[i] * p.Wo;
  }
  const T* dcol = D + (long)g * p.Cn + cd;
  const T* xcol = X + (long)g * p.Cg + ci + (long)(tr - p.pad) * p.xsh + (long)(tq - p.pad) * p.xsw;
  auto gload = [&](int st) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      int m = mbeg + st * BPK + rd0 + RSD * i;
      uint4 vd = make_uint4(0, 0, 0, 0);
      if (m < mend && cd_ok) vd = *(const uint4*)(dcol + (long)m * p.dsw);
      rd[i] = vd;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int m = mbeg + st * BPK + rx0 + RSX * i;
      uint4 vx = make_uint4(0, 0, 0, 0);
      if (m < mend && kx_ok) {
        int hh = rh_[i] * p.stride - p.pad + tr, ww = rw_[i] * p.stride - p.pad + tq;
        if (hh >= 0 && ww >= 0 && hh < p.H && ww < p.W)
          vx = *(const uint4*)(xcol + (long)rb_[i] * p.xsb + (long)(rh_[i] * p.stride) * p.xsh + (long)(rw_[i] * p.stride) * p.xsw);
      }
      rx[i] = vx;
      rw_[i] += BPK;
      while (rw_[i] >= p.Wo) { rw_[i] -= p.Wo; ++rh_[i]; }
      while (rh_[i] >= p.Ho) { rh_[i] -= p.Ho; ++rb_[i]; }
    }
  };
  auto lstore = [&](int buf) {
    if (tdv) {
#pragma unroll
      for (int i = 0; i < ND; ++i) *(uint4*)(sD + buf * BPK * PD + (rd0 + RSD * i) * PD + ccd * 16) = rd[i];
    }
    if (txv) {
#pragma unroll
      for (int i = 0; i < NX; ++i) *(uint4*)(sX + buf * BPK * PX + (rx0 + RSX * i) * PX + ccx * 16) = rx[i];
    }
  };

  f32x4_t acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    if (st + 1 < nsteps) gload(st + 1);
    const char* tD = sD + cur * BPK * PD;
    const char* tX = sX + cur * BPK * PX;
#pragma unroll
    for (int k0 = 0; k0 < BPK; k0 += TFrag<T, WD>::KSUB_PX) {
      typename TFrag<T, WD>::type fa[TA], fb[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) fa[a] = TFrag<T, WD>::load(tD, wi * (WD / 2) + a * 16, k0, lane);
#pragma unroll
      for (int b = 0; b < TB; ++b) fb[b] = TFrag<T, WX>::load(tX, wj * (WX / 2) + b * 16, k0, lane);
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = TFrag<T, WD>::mma(fa[a], fb[b], acc[a][b]);
    }
    if (st + 1 < nsteps) lstore(cur ^ 1);
    __syncthreads();
  }
  // D[i][j]: row i = co = (lane>>4)*4 + reg, col j = k index = lane&15
  float* slab = p.slab + ((long)split * p.G * p.Cn + (long)g * p.Cn) * p.Ktot;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      int k = j0 + wj * (WX / 2) + b * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int co = i0 + wi * (WD / 2) + a * 16 + (lane >> 4) * 4 + r;
        if (co < p.Cn && k < p.Ktot) slab[(long)co * p.Ktot + k] = acc[a][b][r];
      }
    }
}

// slab[split][co][tap][ci]  ->  grad OIHW [co][ci][tap]  (accumulate optional).  SL split-lanes per element group: a narrow layer has
// few elements and many splits, so the split loop is shared by SL threads and folded through LDS.  A thread owns FOUR consecutive
// elements (one float4 per slab: 512-byte runs per slab row and wave instead of 128 - the fold reads nsplit x n floats and was at
// 2.7 TB/s with scalar loads); n is a multiple of 4 because Cg is a multiple of the 16-byte chunk.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nsplit, int Ctot,
                                                           int taps, int Cg, int Cg_real, int accumulate) {
  constexpr int GPB = 256 / SL;  // float4 groups per block
  __shared__ float4 sh[SL][GPB];
  const int e = threadIdx.x % GPB, sl = threadIdx.x / GPB;
  const long idx = ((long)blockIdx.x * GPB + e) * 4;
  const long n = (long)Ctot * taps * Cg;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < n) {
    // four slabs in flight per lane (the partial sums keep a fixed order: bitwise reproducible)
    float4 s0 = s, s1 = s, s2 = s, s3 = s;
    auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int k = sl;
    for (; k + 3 * SL < nsplit; k += 4 * SL) {
      const float4 v0 = *(const float4*)(slab + (long)k * n + idx), v1 = *(const float4*)(slab + (long)(k + SL) * n + idx);
      const float4 v2 = *(const float4*)(slab + (long)(k + 2 * SL) * n + idx), v3 = *(const float4*)(slab + (long)(k + 3 * SL) * n + idx);
      add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
    }
    for (; k < nsplit; k += SL) add(s0, *(const float4*)(slab + (long)k * n + idx));
    s.x = (s0.x + s1.x) + (s2.x + s3.x); s.y = (s0.y + s1.y) + (s2.y + s3.y);
    s.z = (s0.z + s1.z) + (s2.z + s3.z); s.w = (s0.w + s1.w) + (s2.w + s3.w);
  }
  if (SL > 1) {
    sh[sl][e] = s;
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int j = 1; j < SL; ++j) { const float4 v = sh[j][e]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  if (idx >= n) return;
  const int ci = (int)(idx % Cg);  // the four elements share (co, tap): Cg % 4 == 0
  const int tap = (int)((idx / Cg) % taps);
  const int co = (int)(idx / ((long)Cg * taps));
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (ci + j >= Cg_real) break;  // channel padding
    const long o = ((long)co * Cg_real + ci + j) * taps + tap;
    grad[o] = accumulate ? grad[o] + v[j] : v[j];
  }
}

// one element per thread: the form for few, large slabs (head layers: nsplit < 8, 2.4 M elements) - there the transposing
// 4-byte writes dominate and four of them per thread were slower (49 -> 60 us)
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nsplit, int Ctot,
                                                           int taps, int Cg, int Cg_real, int accumulate) {
  constexpr int EPB = 256 / SL;  // elements per block
  __shared__ float sh[SL][EPB];
  const int e = threadIdx.x % EPB, sl = threadIdx.x / EPB;
  long idx = (long)blockIdx.x * EPB + e;
  long n = (long)Ctot * taps * Cg;
  float s = 0.f;
  if (idx < n) {
    // four slabs in flight per lane (the partial sums keep a fixed order: bitwise reproducible)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = sl;
    for (; k + 3 * SL < nsplit; k += 4 * SL) {
      s0 += slab[(long)k * n + idx];
      s1 += slab[(long)(k + SL) * n + idx];
      s2 += slab[(long)(k + 2 * SL) * n + idx];
      s3 += slab[(long)(k + 3 * SL) * n + idx];
    }
    for (; k < nsplit; k += SL) s0 += slab[(long)k * n + idx];
    s = (s0 + s1) + (s2 + s3);
  }
  if (SL > 1) {
    sh[sl][e] = s;
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int j = 1; j < SL; ++j) s += sh[j][e];
  }
  if (idx >= n) return;
  int ci = idx % Cg;
  int tap = (idx / Cg) % taps;
  int co = idx / ((long)Cg * taps);
  if (ci >= Cg_real) return;  // channel padding
  long o = ((long)co * Cg_real + ci) * taps + tap;
  grad[o] = accumulate ? grad[o] + s : s;
}

inline void launch_wgrad_reduce(const float* slab, float* grad, int nsplit, int Ctot, int taps, int Cg, int Cg_real, int accumulate, hipStream_t st) {
  long n = (long)Ctot * taps * Cg;
  long ng = (n + 3) / 4;  // float4 groups
  // split lanes per element group: enough loads in flight for the slab stream (nsplit x n floats) whatever the layer's shape
  if (n <= 65536 && nsplit >= 64) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(cdiv(ng, 16)), dim3(256), 0, st, slab, grad, nsplit, Ctot, taps, Cg, Cg_real, accumulate);
  else if (n <= 1048576 && nsplit >= 32) hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3(cdiv(ng, 32)), dim3(256), 0, st, slab, grad, nsplit, Ctot, taps, Cg, Cg_real, accumulate);
  else if (nsplit >= 8) hipLaunchKernelGGL(wgrad_reduce_kernel<2>, dim3(cdiv(ng, 128)), dim3(256), 0, st, slab, grad, nsplit, Ctot, taps, Cg, Cg_real, accumulate);
  else hipLaunchKernelGGL(wgrad_reduce_scalar_kernel<1>, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, grad, nsplit, Ctot, taps, Cg, Cg_real, accumulate);
}

// OIHW fp32 -> packed [Cout][taps][Cg_pad] (forward) in T, row pitch Kpad
template <typename T>
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cg, int Cg_pad, int taps, int Kpad) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long n = (long)Cout * Kpad;
  if (idx >= n) return;
  int k = idx % Kpad;
  int co = idx / Kpad;
  float v = 0.f;
  if (k < taps * Cg_pad) {
    int tap = k / Cg_pad, ci = k - tap * Cg_pad;
    if (ci < Cg) v = w[((long)co * Cg + ci) * taps + tap];
  }
  TT<T>::st(out + idx, v);
}

// OIHW fp32 -> packed for dgrad: [G][Cg][taps][Cn] in T (row = input channel, K = (tap, co)), row pitch Kpad
template <typename T>
__global__ void pack_w_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int G, int Cn, int Cg, int taps, int Kpad) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long n = (long)G * Cg * Kpad;
  if (idx >= n) return;
  int k = idx % Kpad;
  int ci = (idx / Kpad) % Cg;
  int g = idx / ((long)Kpad * Cg);
  float v = 0.f;
  if (k < taps * Cn) {
    int tap = k / Cn, co = k - tap * Cn;
    v = w[((long)(g * Cn + co) * Cg + ci) * taps + tap];
  }
  TT<T>::st(out + idx, v);
}

// Multi-tensor weight packing: every conv weight of the model, forward layout (mode 0: [Cout][taps][Cg_pad], as pack_w_fwd_kernel) or
// data-gradient layout (mode 1: [G][Cg][taps][Cn], as pack_w_dgrad_kernel), in ONE launch per step instead of two tiny launches per
// conv.  desc[t] = {src fp32 OIHW, dst, a, b, c, taps, Kpad, mode}: mode 0: a = Cout, b = Cg, c = Cg_pad; mode 1: a = G, b = Cn, c = Cg.
// A workgroup owns `chunk` consecutive OUTPUT elements of one tensor.  Both layouts are transposes of the OIHW source (tap <-> channel,
// output <-> input channel): gathering element by element read 4 bytes out of every 36..4608 (170 + 250 us per step on S-3D for
// 110 MB of traffic).  The data-gradient layout is staged through LDS: the source is read in runs that are contiguous in OIHW (the
// R * taps weights that R consecutive input channels own in one filter), the packed rows are written contiguously (250 -> 117 us).
template <typename T>
__global__ __launch_bounds__(256) void mt_pack_w_kernel(const long* __restrict__ desc, const int* __restrict__ ctensor, const int* __restrict__ coff,
                                                        int chunk) {
  constexpr int CAP = 4096;
  __shared__ float st[CAP + 64];
  const long* d = desc + (long)ctensor[blockIdx.x] * 8;
  const float* __restrict__ w = (const float*)d[0];
  T* __restrict__ out = (T*)d[1];
  const int a = (int)d[2], b = (int)d[3], c = (int)d[4], taps = (int)d[5], Kpad = (int)d[6], mode = (int)d[7];
  const long n = mode == 0 ? (long)a * Kpad : (long)a * c * Kpad;
  const long beg = (long)coff[blockIdx.x] * chunk;
  const long end = beg + chunk < n ? beg + chunk : n;
  const int tid = threadIdx.x;
  const long rb = beg / Kpad, re = (end - 1) / Kpad;  // packed rows this chunk touches
  if (mode == 0) {
    // row = output channel co, columns (tap, ci): consecutive lanes read OIHW addresses `taps` floats apart, all inside one filter
    // (b * taps contiguous floats) - the lines are reused from L1/L2.  (Staging whole filters through LDS measured slower: 118 vs 89 us.)
    for (long idx = beg + tid; idx < end; idx += 256) {
      const int k = (int)(idx % Kpad), co = (int)(idx / Kpad);
      float v = 0.f;
      if (k < taps * c) {
        const int tap = k / c, ci = k - tap * c;
        if (ci < b) v = w[((long)co * b + ci) * taps + tap];
      }
      TT<T>::st(out + idx, v);
    }
  } else {
    // row = (g, ci); its columns are (tap, co).  Per pass: R consecutive input channels x CB output channels.
    const int CB = b < 256 ? b : 256;
    int R = CAP / (taps * (CB + 1));
    if (R < 1) R = 1;
    const int P = CB + 1;  // LDS pitch (odd: the transposing writes spread over the banks)
    for (long r0 = rb; r0 <= re;) {
      const int g = (int)(r0 / c), ci0 = (int)(r0 - (long)g * c);
      int nr = R;
      if (nr > c - ci0) nr = c - ci0;
      if (nr > re - r0 + 1) nr = (int)(re - r0 + 1);
      const int seg = nr * taps;
      for (int co0 = 0; co0 < b; co0 += CB) {
        const int cb = b - co0 < CB ? b - co0 : CB;
        __syncthreads();
        for (int i = tid; i < cb * seg; i += 256) {
          const int col = i / seg, j = i - col * seg;
          st[j * P + col] = w[((long)(g * b + co0 + col) * c + ci0) * taps + j];
        }
        __syncthreads();
        for (int o = tid; o < seg * cb; o += 256) {
          const int j = o / cb, col = o - j * cb;
          const int cil = j / taps, tap = j - cil * taps;
          const long idx = (r0 + cil) * Kpad + (long)tap * b + co0 + col;
          if (idx >= beg && idx < end) TT<T>::st(out + idx, st[j * P + col]);
        }
      }
      // row padding (Kpad - taps * b < 8 elements)
      for (int o = tid; o < nr * (Kpad - taps * b); o += 256) {
        const int rl = o / (Kpad - taps * b), k = taps * b + o - rl * (Kpad - taps * b);
        const long idx = (r0 + rl) * Kpad + k;
        if (idx >= beg && idx < end) TT<T>::st(out + idx, 0.f);
      }
      r0 += nr;
    }
  }
}

template <typename T, int MODE>
int launch_conv_mode(ConvP p, hipStream_t st) {
  dim3 block(256);
  const unsigned ny = MODE == 2 ? 4 : 1;  // parity classes
  p.ntx = MODE == 2 ? cdiv((long)p.B * ((p.Hq + 1) / 2) * ((p.Wq + 1) / 2), 128) : cdiv(p.M, 128);
  // channel tile: 128 wide, unless that leaves most of the chip idle - the stride-32 level (20x20, B = 32: 100 pixel tiles) ran a
  // 4 608-deep reduction on 100 workgroups (512 -> 128 3x3: 97 us, 155 TFLOP/s); narrower tiles put 2-4x as many workgroups to work
  // on the same reduction depth
  const long t128 = (long)p.ntx * cdiv(p.Cn, 128) * ny * p.G;
  int bc = p.Cn > 64 ? 128 : (p.Cn > 32 ? 64 : 32);
  if (p.Cn % 32 == 0) {
    if (bc == 128 && t128 < 192) bc = 64;
    if (bc == 64 && (long)p.ntx * cdiv(p.Cn, 64) * ny * p.G < 192 && p.Cn > 32) bc = 32;
  }
  p.nty = cdiv(p.Cn, bc);
  if (bc == 128) {
    size_t sm = 2 * (128 + 128) * 128;
    hipLaunchKernelGGL((conv_gemm_kernel<T, 128, 128, 2, 2, MODE>), dim3(p.ntx * p.nty, ny, p.G), block, sm, st, p);
  } else if (bc == 64) {
    size_t sm = 2 * (128 + 64) * 128;
    hipLaunchKernelGGL((conv_gemm_kernel<T, 128, 64, 4, 1, MODE>), dim3(p.ntx * p.nty, ny, p.G), block, sm, st, p);
  } else {
    size_t sm = 2 * (128 + 32) * 128;
    hipLaunchKernelGGL((conv_gemm_kernel<T, 128, 32, 4, 1, MODE>), dim3(p.ntx * p.nty, ny, p.G), block, sm, st, p);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

template <typename T, bool DGRAD>
int launch_conv(ConvP p, hipStream_t st) {
  if (DGRAD && p.kh == 3 && p.kw == 3 && p.stride == 2 && p.pad == 1 && p.Cg % TT<T>::CE == 0) return launch_conv_mode<T, 2>(p, st);
  return launch_conv_mode<T, DGRAD ? 1 : 0>(p, st);
}

}  // namespace

// conv3x3_tile.hip
int y3d_tile_height(int dtype, int B, int H, int W, int Cg, int Cn, int G, int kh, int kw, int stride, int pad);
int y3d_conv3x3_flat_tiles(int B, int H, int W);
int y3d_conv3x3_tile_launch(int dtype, int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G,
                            const void* w, int Ktot, void* y, long ysw, float* part, int flip, const float* scale, const float* shift, int act,
                            void* stream);

// conv3x3_wgrad_tile.hip
int y3d_wgrad_tile_height(int dtype, int H, int W, int Cg, int Cn, int kh, int kw, int stride, int pad);
int y3d_wgrad_tile_splits(int th, int B, int H, int W, int Cg, int Cn, int G);
int y3d_conv3x3_wgrad_tile_launch(int th, const void* x, long xsb, long xsh, long xsw, const void* dy, long dsw, int B, int H, int W, int Cg,
                                  int Cn, int G, float* slab, int nsplit, void* stream);

// conv1x1_stream.hip
int y3d_conv1x1_stream_ok(int dtype, long M, int K, int N, long xsw, int G = 1);
int y3d_conv1x1_stream_launch(const void* x, long xsw, const void* w, int Kpad, const float* bias, const float* scale, const float* shift, int act,
                              void* y, long ysw, float* part, long M, int K, int N, int G, void* stream);
// wgrad1x1_stream.hip
int y3d_wgrad1x1_stream_ok(int dtype, long M, int Cg, int Cn, long xsw, long dsw);
int y3d_wgrad1x1_stream_launch(const void* x, long xsw, const void* dy, long dsw, long M, int Cg, int Cn, float* slab, int nsplit, int chunk_px,
                               void* stream);
// conv3x3s2_dgrad.hip
int y3d_conv3x3s2_dgrad_ok(int dtype, int B, int Ho, int Wo, int H, int W, int Cout, int Cin, long dsw, long xsw);
int y3d_conv3x3s2_dgrad_launch(const void* dy, long dsw, int B, int Ho, int Wo, int Cout, const void* w_packed_dgrad, int Kpad, void* dx, long xsw,
                               int H, int W, int Cin, void* stream);
// conv3x3_small.hip
int y3d_conv3x3_small_ok(int dtype, int B, int H, int W, int Cin, int Cout, int rows);
int y3d_conv3x3_small_launch(const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cin, int Cout, const void* w, int Ktot, void* y,
                             long ysw, float* part, int rows, int flip, const float* scale, const float* shift, int act, const void* res, long rsw,
                             void* stream);
int y3d_conv3x3_small_rows(int B, int H, int W, int Cin, int Cout);
int y3d_conv3x3_small_s2_ok(int dtype, int B, int H, int W, int Cin, int Cout);
int y3d_conv3x3_small_s2_rows(int B, int H, int W, int Cin, int Cout);
int y3d_conv3x3_small_s2_launch(const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cin, int Cout, const void* w, int Ktot, void* y,
                                long ysw, float* part, int rows, const float* scale, const float* shift, int act, void* stream);
int y3d_conv1x1_stream_rows(long M, int K, int N, int G = 1);
extern "C" int y3d_conv2d_stat_rows(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad);
// wgrad3x3_small.hip
int y3d_wgrad3x3_small_ok(int dtype, int B, int H, int W, int Cin, int Cout);
int y3d_wgrad3x3_small_splits(int B, int H, int W);
int y3d_wgrad3x3_small_launch(const void* x, long xsb, long xsh, long xsw, const void* dy, long dsw, int B, int H, int W, int Cin, int Cout,
                              float* slab, int nsplit, void* stream);
static inline bool dense_pixels(int B, int H, int W, long sb, long sh, long sw) {
  return (H == 1 || sh == (long)W * sw) && (B == 1 || sb == (long)H * W * sw);
}

static int g_tile_kernels = 1;
extern "C" int y3d_conv2d_wgrad_splits(int dtype, int B, int Ho, int Wo, int Cout, int Cin_g, int groups, int kh, int kw);

extern "C" {

int y3d_get_tile_kernels(void) { return g_tile_kernels; }

int y3d_set_tile_kernels(int enable) {
  int old = g_tile_kernels;
  g_tile_kernels = enable ? 1 : 0;
  return old;
}

int y3d_conv_stat_blocks(int B, int Ho, int Wo) { return cdiv((long)B * Ho * Wo, 128); }

int y3d_conv2d_wgrad_plan(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad) {
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && y3d_wgrad3x3_small_ok(dtype, B, H, W, Cin, Cout))
    return y3d_wgrad3x3_small_splits(B, H, W);
  int th = (g_tile_kernels && groups > 0) ? y3d_wgrad_tile_height(dtype, H, W, Cin / groups, Cout / groups, kh, kw, stride, pad) : 0;
  if (th) return y3d_wgrad_tile_splits(th, B, H, W, Cin / groups, Cout / groups, groups);
  int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && groups == 1 && y3d_wgrad1x1_stream_ok(dtype, (long)B * H * W, Cin, Cout, Cin, Cout)) {
    // the streaming kernel keeps one 512-thread workgroup per CU busy (148 KB of LDS ring): one round of workgroups, at least four
    // 64-pixel steps each; every further split is one more fp32 slab to write and to fold
    const int target = 256;
    const long tiles = (long)cdiv(Cin, Cin <= 64 ? 64 : 128) * cdiv(Cout, Cout <= 64 ? 64 : 128);
    // rounded down, as y3d_wgrad_tile_splits: 15 tiles (640 -> 320, X widths) x 18 splits = 270 workgroups = a second round for 14 of them
    long want = target / tiles, maxs = cdiv((long)B * H * W, 4 * 64);
    if (want > maxs) want = maxs;
    return (int)(want < 1 ? 1 : want);
  }
  return y3d_conv2d_wgrad_splits(dtype, B, Ho, Wo, Cout, Cin / groups, groups, kh, kw);
}

int y3d_conv2d_stat_rows(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad) {
  // the persistent kernels write one row per workgroup (a few hundred rows whatever the map size)
  if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && groups >= 1 && y3d_conv1x1_stream_ok(dtype, (long)B * H * W, Cin / groups, Cout / groups, Cin, groups))
    return y3d_conv1x1_stream_rows((long)B * H * W, Cin / groups, Cout / groups, groups);
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && y3d_conv3x3_small_ok(dtype, B, H, W, Cin, Cout, 1))
    return y3d_conv3x3_small_rows(B, H, W, Cin, Cout);
  if (kh == 3 && kw == 3 && stride == 2 && pad == 1 && groups == 1 && y3d_conv3x3_small_s2_ok(dtype, B, H, W, Cin, Cout))
    return y3d_conv3x3_small_s2_rows(B, H, W, Cin, Cout);
  int th = (g_tile_kernels && groups > 0) ? y3d_tile_height(dtype, B, H, W, Cin / groups, Cout / groups, groups, kh, kw, stride, pad) : 0;
  if (th < 0) return y3d_conv3x3_flat_tiles(B, H, W);  // conv3x3_flat.hip: one row per 512-position tile of the flat padded space
  if (th) return B * cdiv(H, th) * cdiv(W, 16);
  int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  return cdiv((long)B * Ho * Wo, 128);
}

int y3d_conv_kpad(int dtype, int k_total) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  return cdiv(k_total, ce) * ce;
}

int y3d_pack_weight_fwd(int dtype, const float* w_oihw, void* out, int Cout, int Cin_g, int Cin_g_pad, int kh, int kw, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(Cin_g_pad >= Cin_g && Cin_g_pad % ce == 0, "pack_weight_fwd: Cin_g_pad=%d must be a multiple of %d", Cin_g_pad, ce);
  int taps = kh * kw, Kpad = taps * Cin_g_pad;
  long n = (long)Cout * Kpad;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16)
    hipLaunchKernelGGL(pack_w_fwd_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, w_oihw, (bf16_t*)out, Cout, Cin_g, Cin_g_pad, taps, Kpad);
  else
    hipLaunchKernelGGL(pack_w_fwd_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, w_oihw, (float*)out, Cout, Cin_g, Cin_g_pad, taps, Kpad);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_pack_weight_dgrad(int dtype, const float* w_oihw, void* out, int Cout, int Cin_g, int groups, int kh, int kw, void* stream) {
  Y3D_CHECK(Cout % groups == 0, "pack_weight_dgrad: Cout %% groups");
  int Cn = Cout / groups, taps = kh * kw;
  int Kpad = y3d_conv_kpad(dtype, taps * Cn);
  long n = (long)groups * Cin_g * Kpad;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16)
    hipLaunchKernelGGL(pack_w_dgrad_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, w_oihw, (bf16_t*)out, groups, Cn, Cin_g, taps, Kpad);
  else
    hipLaunchKernelGGL(pack_w_dgrad_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, w_oihw, (float*)out, groups, Cn, Cin_g, taps, Kpad);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

static int check_align(const char* what, const void* ptr, long s0, long s1, long s2, int ce) {
  Y3D_CHECK(((uintptr_t)ptr & 15) == 0, "%s: pointer not 16-byte aligned", what);
  Y3D_CHECK(s0 % ce == 0 && s1 % ce == 0 && s2 % ce == 0, "%s: strides (%ld,%ld,%ld) not multiples of %d elements", what, s0, s1, s2, ce);
  return Y3D_OK;
}

static int conv2d_fwd_impl(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                           const void* w_packed, const float* bias, const float* scale, const float* shift, int act, void* y, int64_t ysw,
                           int Ho, int Wo, int Cout, int groups, int kh, int kw, int stride, int pad, float* stat_partials, void* stream,
                           const void* res = nullptr, int64_t rsw = 0) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "conv2d_fwd: bad dtype %d", dtype);
  Y3D_CHECK(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && groups > 0, "conv2d_fwd: empty shape");
  Y3D_CHECK(Cin % groups == 0 && Cout % groups == 0, "conv2d_fwd: channels not divisible by groups");
  Y3D_CHECK((Cin / groups) % ce == 0, "conv2d_fwd: Cin/groups=%d must be a multiple of %d (pad the input)", Cin / groups, ce);
  Y3D_CHECK(Ho == (H + 2 * pad - kh) / stride + 1 && Wo == (W + 2 * pad - kw) / stride + 1, "conv2d_fwd: output dims (%d,%d) inconsistent", Ho, Wo);
  Y3D_CHECK(ysw >= Cout, "conv2d_fwd: ysw < Cout");
  Y3D_CHECK(kh * kw <= 64, "conv2d_fwd: at most 64 filter taps");
  Y3D_CHECK(!(bias && stat_partials), "conv2d_fwd: bias and BN partials are mutually exclusive");
  if (check_align("conv2d_fwd x", x, xsb, xsh, xsw, ce)) return Y3D_ERR_INVALID;
  Y3D_CHECK(((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)y & 7) == 0, "conv2d_fwd: w/y alignment");
  Y3D_CHECK((long)B * Ho * Wo < (1L << 31), "conv2d_fwd: too many pixels");
  ConvP p;
  p.x = x; p.w = w_packed; p.bias = bias; p.scale = scale; p.shift = shift; p.act = act; p.y = y; p.part = stat_partials;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.Hg = H; p.Wg = W; p.Hq = Ho; p.Wq = Wo;
  p.Cg = Cin / groups; p.Cn = Cout / groups; p.G = groups;
  p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.Ktot = kh * kw * p.Cg; p.Kpad = p.Ktot; p.M = B * Ho * Wo;
  if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && dense_pixels(B, H, W, xsb, xsh, xsw) &&
      y3d_conv1x1_stream_ok(dtype, p.M, p.Cg, p.Cn, xsw, groups))
    return y3d_conv1x1_stream_launch(x, xsw, w_packed, p.Kpad, bias, scale, shift, act, y, ysw, stat_partials, p.M, p.Cg, p.Cn, groups, stream);
  // y3d_conv2d_stat_rows sized the caller's partial buffer for the streaming kernel: an operand it cannot take (a view that is not
  // pixel-dense, or beyond 32-bit byte offsets) must not fall through to the generic kernel's one-row-per-tile layout
  Y3D_CHECK(!(stat_partials && kh == 1 && kw == 1 && stride == 1 && pad == 0 && y3d_conv1x1_stream_ok(dtype, p.M, p.Cg, p.Cn, Cin, groups)),
            "conv2d_fwd: 1x1 input view must be pixel-dense and below 4 GB for the BatchNorm partial layout of this shape");
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && !bias) {
    const int rows = stat_partials ? y3d_conv2d_stat_rows(dtype, B, H, W, Cin, Cout, groups, kh, kw, stride, pad) : 1;
    if (y3d_conv3x3_small_ok(dtype, B, H, W, Cin, Cout, rows))
      return y3d_conv3x3_small_launch(x, xsb, xsh, xsw, B, H, W, Cin, Cout, w_packed, p.Ktot, y, ysw, stat_partials, rows, 0, scale, shift, act, res, rsw, stream);
  }
  Y3D_CHECK(!res, "conv2d_fwd_affine_res: no kernel with a residual epilogue takes this geometry (ask y3d_conv2d_fwd_affine_res_ok first)");
  if (kh == 3 && kw == 3 && stride == 2 && pad == 1 && groups == 1 && !bias && y3d_conv3x3_small_s2_ok(dtype, B, H, W, Cin, Cout)) {
    const int rows = stat_partials ? y3d_conv2d_stat_rows(dtype, B, H, W, Cin, Cout, groups, kh, kw, stride, pad) : 1;
    return y3d_conv3x3_small_s2_launch(x, xsb, xsh, xsw, B, H, W, Cin, Cout, w_packed, p.Ktot, y, ysw, stat_partials, rows, scale, shift, act, stream);
  }
  if (!bias && g_tile_kernels) {
    int th = y3d_tile_height(dtype, B, H, W, p.Cg, p.Cn, groups, kh, kw, stride, pad);
    if (th) return y3d_conv3x3_tile_launch(dtype, th, x, xsb, xsh, xsw, B, H, W, p.Cg, p.Cn, groups, w_packed, p.Ktot, y, ysw, stat_partials, 0, scale, shift, act, stream);
  }
  if (dtype == Y3D_BF16) return launch_conv<bf16_t, false>(p, (hipStream_t)stream);
  return launch_conv<float, false>(p, (hipStream_t)stream);
}

int y3d_conv2d_fwd(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                   const void* w_packed, const float* bias, void* y, int64_t ysw, int Ho, int Wo, int Cout, int groups,
                   int kh, int kw, int stride, int pad, float* stat_partials, void* stream) {
  return conv2d_fwd_impl(dtype, x, xsb, xsh, xsw, B, H, W, Cin, w_packed, bias, nullptr, nullptr, 0, y, ysw, Ho, Wo, Cout, groups, kh, kw,
                         stride, pad, stat_partials, stream);
}

int y3d_conv2d_fwd_affine(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                          const void* w_packed, const float* scale, const float* shift, int act, void* y, int64_t ysw, int Ho, int Wo,
                          int Cout, int groups, int kh, int kw, int stride, int pad, void* stream) {
  Y3D_CHECK(scale && shift, "conv2d_fwd_affine: scale/shift required");
  return conv2d_fwd_impl(dtype, x, xsb, xsh, xsw, B, H, W, Cin, w_packed, nullptr, scale, shift, act, y, ysw, Ho, Wo, Cout, groups, kh, kw,
                         stride, pad, nullptr, stream);
}

int y3d_conv2d_fwd_affine_res_ok(int dtype, int B, int H, int W, int Cin, int Cout, int groups, int kh, int kw, int stride, int pad) {
  return kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && y3d_conv3x3_small_ok(dtype, B, H, W, Cin, Cout, 1);
}

int y3d_conv2d_fwd_affine_res(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                              const void* w_packed, const float* scale, const float* shift, int act, const void* res, int64_t rsw, void* y,
                              int64_t ysw, int Ho, int Wo, int Cout, int groups, int kh, int kw, int stride, int pad, void* stream) {
  Y3D_CHECK(scale && shift && res, "conv2d_fwd_affine_res: scale / shift / residual required");
  Y3D_CHECK(y3d_conv2d_fwd_affine_res_ok(dtype, B, H, W, Cin, Cout, groups, kh, kw, stride, pad), "conv2d_fwd_affine_res: geometry not supported");
  return conv2d_fwd_impl(dtype, x, xsb, xsh, xsw, B, H, W, Cin, w_packed, nullptr, scale, shift, act, y, ysw, Ho, Wo, Cout, groups, kh, kw,
                         stride, pad, nullptr, stream, res, rsw);
}

int y3d_conv2d_bwd_data(int dtype, const void* dy, int64_t dsb, int64_t dsh, int64_t dsw, int B, int Ho, int Wo, int Cout,
                        const void* w_packed_dgrad, void* dx, int64_t xsw, int H, int W, int Cin, int groups, int kh, int kw,
                        int stride, int pad, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "conv2d_bwd_data: bad dtype %d", dtype);
  Y3D_CHECK(Cin % groups == 0 && Cout % groups == 0, "conv2d_bwd_data: channels not divisible by groups");
  Y3D_CHECK((Cout / groups) % ce == 0, "conv2d_bwd_data: Cout/groups=%d must be a multiple of %d", Cout / groups, ce);
  Y3D_CHECK(xsw >= Cin, "conv2d_bwd_data: xsw < Cin");
  Y3D_CHECK(stride >= 1 && dsh % stride == 0 && dsw % stride == 0, "conv2d_bwd_data: stride %d must divide the dy strides", stride);
  Y3D_CHECK(kh * kw <= 64, "conv2d_bwd_data: at most 64 filter taps");
  if (check_align("conv2d_bwd_data dy", dy, dsb, dsh, dsw, ce)) return Y3D_ERR_INVALID;
  ConvP p;
  p.x = dy; p.w = w_packed_dgrad; p.bias = nullptr; p.scale = nullptr; p.shift = nullptr; p.act = 0; p.y = dx; p.part = nullptr;
  p.xsb = dsb; p.xsh = dsh; p.xsw = dsw; p.ysw = xsw;
  p.B = B; p.Hg = Ho; p.Wg = Wo; p.Hq = H; p.Wq = W;
  p.Cg = Cout / groups; p.Cn = Cin / groups; p.G = groups;
  p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.Ktot = kh * kw * p.Cg; p.Kpad = y3d_conv_kpad(dtype, p.Ktot); p.M = B * H * W;
  if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && dense_pixels(B, Ho, Wo, dsb, dsh, dsw) &&
      y3d_conv1x1_stream_ok(dtype, p.M, p.Cg, p.Cn, dsw, groups))
    return y3d_conv1x1_stream_launch(dy, dsw, w_packed_dgrad, p.Kpad, nullptr, nullptr, nullptr, 0, dx, xsw, nullptr, p.M, p.Cg, p.Cn, groups, stream);
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && Ho == H && Wo == W && y3d_conv3x3_small_ok(dtype, B, H, W, Cout, Cin, 1))
    return y3d_conv3x3_small_launch(dy, dsb, dsh, dsw, B, H, W, Cout, Cin, w_packed_dgrad, p.Kpad, dx, xsw, nullptr, 1, 1, nullptr, nullptr, 0, nullptr, 0, stream);
  if (kh == 3 && kw == 3 && stride == 2 && pad == 1 && groups == 1 && dense_pixels(B, Ho, Wo, dsb, dsh, dsw) &&
      y3d_conv3x3s2_dgrad_ok(dtype, B, Ho, Wo, H, W, Cout, Cin, dsw, xsw))
    return y3d_conv3x3s2_dgrad_launch(dy, dsw, B, Ho, Wo, Cout, w_packed_dgrad, p.Kpad, dx, xsw, H, W, Cin, stream);
  if (g_tile_kernels) {
    // a 3x3 s1 p1 data gradient is the same conv on dy with flipped taps (Ho == H, Wo == W)
    int th = y3d_tile_height(dtype, B, Ho, Wo, p.Cg, p.Cn, groups, kh, kw, stride, pad);
    if (th && Ho == H && Wo == W)
      return y3d_conv3x3_tile_launch(dtype, th, dy, dsb, dsh, dsw, B, H, W, p.Cg, p.Cn, groups, w_packed_dgrad, p.Kpad, dx, xsw, nullptr, 1, nullptr, nullptr, 0, stream);
  }
  if (dtype == Y3D_BF16) return launch_conv<bf16_t, true>(p, (hipStream_t)stream);
  return launch_conv<float, true>(p, (hipStream_t)stream);
}

static inline int wgrad_tile_w(int n) { return n <= 32 ? 32 : (n <= 64 ? 64 : 128); }  // operand tile width of the generic wgrad kernel

int y3d_conv2d_wgrad_splits(int dtype, int B, int Ho, int Wo, int Cout, int Cin_g, int groups, int kh, int kw) {
  int bpk = dtype == Y3D_BF16 ? 64 : 32;
  long M = (long)B * Ho * Wo;
  long tiles = (long)cdiv(kh * kw * Cin_g, wgrad_tile_w(kh * kw * Cin_g)) * cdiv(Cout / groups, wgrad_tile_w(Cout / groups)) * groups;
  // every extra split is one more fp32 slab to write and re-read, so: ~2 workgroups per CU when the slab is large, ~4 when it is
  // small (the pixel loop of a narrow layer is latency-bound: 128 splits left half of the CUs idle on the stem), at least
  // 4 K-steps per split, at most 32 MB of slabs
  long slab = (long)Cout * kh * kw * Cin_g * 4;
  long want = cdiv(slab <= (1 << 20) ? 1024 : 512, tiles);
  long maxs = cdiv(M, 4 * bpk);
  if (want > maxs) want = maxs;
  long cap = (32L << 20) / slab;
  if (cap < 128) cap = 128;
  if (want > cap) want = cap;
  if (want > 1024) want = 1024;
  if (want < 1) want = 1;
  return (int)want;
}

int y3d_conv2d_bwd_weight(int dtype, const void* x, int64_t xsb, int64_t xsh, int64_t xsw, int B, int H, int W, int Cin,
                          int Cin_real, const void* dy, int64_t dsw, int Ho, int Wo, int Cout, int groups, int kh, int kw,
                          int stride, int pad, float* slab, int nsplit, float* grad_oihw, int accumulate, void* stream) {
  int ce = dtype == Y3D_BF16 ? 8 : 4;
  int bpk = dtype == Y3D_BF16 ? 64 : 32;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "conv2d_bwd_weight: bad dtype %d", dtype);
  Y3D_CHECK(Cin % groups == 0 && Cout % groups == 0, "conv2d_bwd_weight: channels not divisible by groups");
  Y3D_CHECK((Cin / groups) % ce == 0 && (Cout / groups) % ce == 0, "conv2d_bwd_weight: per-group channels must be multiples of %d", ce);
  Y3D_CHECK(nsplit >= 1, "conv2d_bwd_weight: nsplit");
  Y3D_CHECK(Cin_real <= Cin && (Cin_real == Cin || groups == 1), "conv2d_bwd_weight: Cin_real");
  if (check_align("conv2d_bwd_weight x", x, xsb, xsh, xsw, ce)) return Y3D_ERR_INVALID;
  Y3D_CHECK(((uintptr_t)dy & 15) == 0 && dsw % ce == 0, "conv2d_bwd_weight: dy alignment");
  WgradP p;
  p.x = x; p.dy = dy; p.slab = slab;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.dsw = dsw;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  p.Cg = Cin / groups; p.Cn = Cout / groups; p.G = groups;
  p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.Ktot = kh * kw * p.Cg; p.M = B * Ho * Wo; p.nsplit = nsplit;
  p.chunk_px = cdiv(cdiv(p.M, nsplit), bpk) * bpk;
  hipStream_t st = (hipStream_t)stream;
  if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && groups == 1 && Cin_real == Cin && y3d_wgrad3x3_small_ok(dtype, B, H, W, Cin, Cout)) {
    int rc = y3d_wgrad3x3_small_launch(x, xsb, xsh, xsw, dy, dsw, B, H, W, Cin, Cout, slab, nsplit, stream);
    if (rc) return rc;
    launch_wgrad_reduce(slab, grad_oihw, nsplit, Cout, 9, Cin, Cin, accumulate, st);
    Y3D_LAUNCH_CHECK();
    return Y3D_OK;
  }
  {
    int th = g_tile_kernels ? y3d_wgrad_tile_height(dtype, H, W, p.Cg, p.Cn, kh, kw, stride, pad) : 0;
    if (th && Cin_real == Cin) {
      Y3D_CHECK(nsplit == y3d_wgrad_tile_splits(th, B, H, W, p.Cg, p.Cn, groups), "conv2d_bwd_weight: nsplit must come from y3d_conv2d_wgrad_plan");
      int rc = y3d_conv3x3_wgrad_tile_launch(th, x, xsb, xsh, xsw, dy, dsw, B, H, W, p.Cg, p.Cn, groups, slab, nsplit, stream);
      if (rc) return rc;
      long n2 = (long)Cout * kh * kw * p.Cg;
      (void)n2;
      launch_wgrad_reduce(slab, grad_oihw, nsplit, Cout, kh * kw, p.Cg, p.Cg, accumulate, st);
      Y3D_LAUNCH_CHECK();
      return Y3D_OK;
    }
  }
  if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && groups == 1 && Cin_real == Cin && dense_pixels(B, H, W, xsb, xsh, xsw) &&
      y3d_wgrad1x1_stream_ok(dtype, p.M, Cin, Cout, xsw, dsw)) {
    int rc = y3d_wgrad1x1_stream_launch(x, xsw, dy, dsw, p.M, Cin, Cout, slab, nsplit, p.chunk_px, stream);
    if (rc) return rc;
    launch_wgrad_reduce(slab, grad_oihw, nsplit, Cout, 1, Cin, Cin, accumulate, st);
    Y3D_LAUNCH_CHECK();
    return Y3D_OK;
  }
  const int wd = wgrad_tile_w(p.Cn), wx = wgrad_tile_w(p.Ktot);
  dim3 grid(cdiv(p.Ktot, wx), cdiv(p.Cn, wd), groups * nsplit);
#define Y3D_WGRAD(T, WD, WX)                                                                                         \
  hipLaunchKernelGGL((conv_wgrad_kernel<T, WD, WX>), grid, dim3(256),                                                  \
                     2 * TT<T>::BKE * (size_t)(TFrag<T, WD>::PITCH + TFrag<T, WX>::PITCH), st, p)
#define Y3D_WGRAD_T(T)                                                                                               \
  do {                                                                                                               \
    if (wd == 128) { if (wx == 128) Y3D_WGRAD(T, 128, 128); else if (wx == 64) Y3D_WGRAD(T, 128, 64); else Y3D_WGRAD(T, 128, 32); } \
    else if (wd == 64) { if (wx == 128) Y3D_WGRAD(T, 64, 128); else if (wx == 64) Y3D_WGRAD(T, 64, 64); else Y3D_WGRAD(T, 64, 32); } \
    else { if (wx == 128) Y3D_WGRAD(T, 32, 128); else if (wx == 64) Y3D_WGRAD(T, 32, 64); else Y3D_WGRAD(T, 32, 32); }    \
  } while (0)
  if (dtype == Y3D_BF16) Y3D_WGRAD_T(bf16_t); else Y3D_WGRAD_T(float);
#undef Y3D_WGRAD_T
#undef Y3D_WGRAD
  Y3D_LAUNCH_CHECK();
  int cg_real = groups == 1 ? Cin_real : p.Cg;
  launch_wgrad_reduce(slab, grad_oihw, nsplit, Cout, kh * kw, p.Cg, cg_real, accumulate, st);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_mt_pack_weights(int dtype, const int64_t* desc, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "mt_pack_weights: bad dtype");
  Y3D_CHECK(nchunks >= 1 && chunk >= 256, "mt_pack_weights: empty chunk table");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(mt_pack_w_kernel<bf16_t>, dim3(nchunks), dim3(256), 0, st, (const long*)desc, chunk_tensor, chunk_off, chunk);
  else hipLaunchKernelGGL(mt_pack_w_kernel<float>, dim3(nchunks), dim3(256), 0, st, (const long*)desc, chunk_tensor, chunk_off, chunk);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
