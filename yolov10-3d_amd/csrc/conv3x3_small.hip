// 3x3 stride-1 "same" convolution of the NARROW body layers (8 .. 64 input channels in multiples of 8 - 32 / 64 of the N, S, B, L
// widths, 48 of M; a count below its 32- / 64-channel LDS row is zero-filled by out-of-range DMA lanes - at most 64 output channels; bf16): forward and,
// with flipped taps on dy, the data gradient.  Reference: the Bottleneck 3x3 convs of the first C2f stages (nn/modules/block.py:327-342
// through conv.py:120-122).
//
// These layers fell between the kernels: the resident-halo kernels (conv3x3_wide / conv3x3_tile) compute 128-output-channel tiles
// - half of the MFMA work is padding at 64 channels, and 32-channel inputs are below their 64-channel K slab - and the generic
// implicit GEMM gathers every input pixel nine times through L2 (32 -> 32 at 160x160: 82 / 101 us forward / data gradient for
// 105 MB of tensors).  Here ALL the weights are resident in LDS (9 taps x Cout x Cin: 18-74 KB), a persistent workgroup walks
// 8 x 16 pixel tiles, the (8 + 2) x (16 + 2) halo of a tile arrives by LDS-DMA (double buffered: the next tile in flight under the
// MFMAs and the stores of this one; the counted s_waitcnt leaves exactly the stores of a wave outstanding), all nine taps read
// their shifted windows out of it.  BatchNorm partial sums stay per lane over all tiles of a workgroup and are folded once
// (row = workgroup index: y3d_conv2d_stat_rows returns the workgroup count for the shapes this kernel takes).
#include "common.h"

namespace {

struct SmP {
  const bf16_t* x;
  const bf16_t* w;   // [Cout][Ktot], K index = tap * Cin + ci (forward packing, or the dgrad packing of the transposed conv)
  bf16_t* y;
  float* part;       // [rows][Cout][2] or null
  const float *scale, *shift;  // eval: y = act(conv * scale[c] + shift[c]) (+ res), no partial sums
  const bf16_t* res;           // optional residual added after the activation (Bottleneck shortcut, block.py:342), pixel stride rsw
  int act, rsw;
  int xsb, xsh, xsw, ysw;
  int B, H, W, Cin, Cout, Ktot, flip, rows;  // H, W: OUTPUT map
  int Hi, Wi;                                 // input map (= H, W at stride 1)
  int nty, ntx, ntiles;
  unsigned xbytes, ybytes;
};

template <int N> __device__ __forceinline__ void sm_wvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((ext_vector_type(2))) unsigned sm_u32x2;

// CB: bytes of one pixel's (or one weight row's) K slab in LDS: 64 (Cin = 32) or 128 (Cin = 64).  Rows of 16-byte chunks, XOR-swizzled
// so that 16 lanes reading 16 consecutive rows at one chunk index cover all banks: by (row & 7) for 128-byte rows, by ((row >> 2) & 3)
// for 64-byte rows.
template <int CB> __device__ __forceinline__ int sm_swz(int row) { return CB == 128 ? (row & 7) : ((row >> 2) & 3); }

// ST = 2 (round 3): the stride-2 forward of the first stages (32 -> 64 at 320x320: 315 MB of tensors took 146 us on the generic implicit
// GEMM).  The (2 TH + 1) x 33 input halo of an 8 x 16 output tile is laid out in LDS as FOUR PARITY PLANES of 9 x 17 pixels (plane =
// (row & 1, column & 1) of the halo coordinate): tap (r, q) of output pixel (y, x) reads halo (2y + r, 2x + q) = plane (r & 1, q & 1),
// position (y + (r >> 1), x + (q >> 1)) - sixteen consecutive lanes read sixteen consecutive pixels of one plane row, exactly the
// stride-1 access pattern, with the same swizzle.  Only the slot -> source-pixel map of the DMA plan and the tap offsets differ.
// NW = 8 (round 4, NCT even): eight waves share the 8 x 16 tile as 4 row pairs x 2 channel halves.  With four waves a SIMD holds ONE wave,
// which reads its fragments, waits, multiplies, 18 times per tile: 12 700 cycles per tile at 64 -> 64 against 2 300 of MFMA work and 3 400 of
// LDS reads.  Two waves per SIMD overlap one's LDS latency with the other's MFMAs (a wave then reads 2 + 2 fragments for 4 MFMAs instead of
// 4 + 2 for 8: a third more LDS bytes, which is not the bound).  Same accumulation order per output: bit-identical results.
template <int CB, int NCT, bool AFF, int ST, int NW>  // NCT: 16-channel output tiles (Cout <= 16 NCT); AFF: folded BatchNorm + SiLU (+ residual) epilogue
__global__ __launch_bounds__(NW * 64) void conv3x3_small_kernel(SmP p) {
  static_assert(NW == 4 || (NW == 8 && NCT % 2 == 0), "eight waves split the channel tiles in two halves");
  constexpr int NT = NW * 64;                                   // threads
  constexpr int NCW = NW == 8 ? NCT / 2 : NCT;                  // channel tiles per wave
  constexpr int TH = 8;
  constexpr int HW = ST == 1 ? 18 : 17;                         // pixels per row of a halo plane
  constexpr int PP = (ST == 1 ? TH + 2 : TH + 1) * HW;          // pixel slots per plane: 180 / 153
  constexpr int NPIX = ST == 1 ? PP : 4 * PP;                   // 180 halo pixels / 612 slots (561 real ones)
  constexpr int SPI = 1024 / CB;                                // pixel slots per DMA instruction (8 or 16)
  constexpr int NINST = (NPIX + SPI - 1) / SPI;                 // 23 / 12
  constexpr int NI = (NINST + NW - 1) / NW;                     // per wave (the last ones may be dummies into the slack)
  constexpr int TILE = NI * NW * 1024;                          // bytes per halo buffer
  constexpr int KS = CB / 64;                                   // 32-channel MFMA steps per tap
  constexpr int CPR = CB / 16;                                  // chunks per row
  constexpr int WB = 9 * NCT * 16 * CB;                         // resident weights [tap][co][CB]
  constexpr int NST = 2 * NCW;                                  // stores per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  char* sH = smem + WB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwk = gridDim.x, wk = blockIdx.x;
  const int Cin = p.Cin;  // <= CB / 2: the chunks of an LDS row past the real channels hold zeros (weights: below; halo: out-of-range DMA lanes)

  const int wr = wave & 3, wc = wave >> 2;  // row pair of the tile, channel half (NW = 8)
  for (int i = tid; i < 9 * NCT * 16 * CPR; i += NT) {
    const int c = i % CPR, r = (i / CPR) % (NCT * 16), tap = i / (CPR * NCT * 16);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < p.Cout && c * 8 < Cin) v = *(const uint4*)(p.w + (long)r * p.Ktot + (p.flip ? 8 - tap : tap) * Cin + c * 8);
    *(uint4*)(sW + (tap * NCT * 16 + r) * CB + ((c ^ sm_swz<CB>(r)) << 4)) = v;
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.ybytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  auto decode = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % p.ntx, t2 = t / p.ntx;
    x0 = tx * 16;
    y0 = (t2 % p.nty) * TH;
    b = t2 / p.nty;
  };
  // DMA plan of this lane, computed once: per instruction its halo pixel (row, column) and its element offset relative to the tile's
  // origin pixel (the chunk -> pixel maps cost more VALU time per tile than the bounds tests that remain)
  int p_off[NI], p_rc[NI];  // (row << 8) | column, row 255 = never valid
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int ii = wave * NI + n;
    const int P = ii * SPI + lane / CPR, cpos = lane % CPR;
    int row, col;
    bool pv = P < NPIX;
    if (ST == 1) {
      row = P / HW; col = P - row * HW;
    } else {
      const int pl = P / PP, rem = P - pl * PP, pr = rem / HW, pc = rem - pr * HW;
      row = 2 * pr + (pl >> 1); col = 2 * pc + (pl & 1);  // halo coordinate relative to input pixel (2 y0 - 1, 2 x0 - 1)
      pv = pv && row <= 2 * TH && col <= 32;
    }
    p_off[n] = (row - 1) * p.xsh + (col - 1) * p.xsw + ((cpos ^ sm_swz<CB>(P)) << 3);
    p_rc[n] = (((pv & ((cpos ^ sm_swz<CB>(P)) * 8 < Cin)) ? row : 255) << 8) | col;  // the chunk this lane FETCHES must hold real channels
  }
  auto issue = [&](int t, int buf) {
    int b, y0, x0;
    const bool live = t < p.ntiles;
    decode(live ? t : 0, b, y0, x0);
    const int base = b * p.xsb + ST * y0 * p.xsh + ST * x0 * p.xsw;
    const int hlim = live ? p.Hi : 0;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int r = p_rc[n] >> 8, c = p_rc[n] & 255;
      const bool ok = ((unsigned)(ST * y0 - 1 + r) < (unsigned)hlim) & ((unsigned)(ST * x0 - 1 + c) < (unsigned)p.Wi);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sH + buf * TILE + (wave * NI + n) * 1024), 16,
                                               ok ? (unsigned)(base + p_off[n]) * 2u : OOB, 0, 0, 0);
    }
  };
  if (wk < p.ntiles) issue(wk, 0);

  const int lp = lane & 15, lq = lane >> 4;
  float ssum[NCW][4], ssq[NCW][4];
#pragma unroll
  for (int a = 0; a < NCW; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }
  // eval epilogue constants of this lane's channels (16 a + 4 lq + j), loaded ONCE: a load inside the tile loop makes the compiler wait
  // vmcnt(0) there and drains the halo prefetch of the next tile every tile (58 us against 33 us for the training form at 64 -> 64 @80x80)
  float sv[AFF ? NCW : 1][4], hv[AFF ? NCW : 1][4];
  if (AFF) {
#pragma unroll
    for (int a = 0; a < NCW; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (wc * NCW + a) * 16 + 4 * lq + j;
        sv[a][j] = c < p.Cout ? p.scale[c] : 0.f;
        hv[a][j] = c < p.Cout ? p.shift[c] : 0.f;
      }
  }
  int buf = 0;
  bool first = true;
#pragma unroll 1
  for (int t = wk; t < p.ntiles; t += nwk) {
    if (first) { sm_wvm<0>(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); first = false; }
    else sm_wvm<NST>();
    __builtin_amdgcn_s_barrier();
    issue(t + nwk, buf ^ 1);
    int b, y0, x0;
    decode(t, b, y0, x0);
    const char* hb = sH + buf * TILE;
    f32x4_t acc[2][NCW];  // [row of the wave][output-channel tile of the wave]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int a = 0; a < NCW; ++a) acc[r][a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tr = tap / 3, tq = tap - tr * 3;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8_t fa[NCW], fb[2];
#pragma unroll
        for (int a = 0; a < NCW; ++a) {
          const int r = (wc * NCW + a) * 16 + lp;
          fa[a] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(sW + (tap * NCT * 16 + r) * CB + (((ks * 4 + lq) ^ sm_swz<CB>(r)) << 4)));
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int P = ST == 1 ? (wr * 2 + r + tr) * HW + tq + lp
                                : (((tr & 1) << 1) | (tq & 1)) * PP + (wr * 2 + r + (tr >> 1)) * HW + (tq >> 1) + lp;
          fb[r] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(hb + P * CB + (((ks * 4 + lq) ^ sm_swz<CB>(P)) << 4)));
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int a = 0; a < NCW; ++a) acc[r][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[r], acc[r][a], 0, 0, 0);
      }
    }
    // lane: pixel (y0 + 2 wave + r, x0 + lp), output channels 16 a + 4 lq .. + 3
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int yy = y0 + wr * 2 + r, xx = x0 + lp;
      const bool inb = (yy < p.H) & (xx < p.W);
      const unsigned pix = (unsigned)((b * p.H + yy) * p.W + xx) * (unsigned)p.ysw;
#pragma unroll
      for (int a = 0; a < NCW; ++a) {
        const int co = (wc * NCW + a) * 16 + 4 * lq;
        float v[4];
        if (AFF) {
          // (the residual is an ordinary load: the compiler waits vmcnt(0) for it, draining the halo prefetch - shortcut blocks only)
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.res && inb && co < p.Cout) {
            const uint2 rr = *(const uint2*)(p.res + (long)((b * p.H + yy) * p.W + xx) * p.rsw + co);
            rv[0] = __uint_as_float(rr.x << 16); rv[1] = __uint_as_float(rr.x & 0xffff0000u);
            rv[2] = __uint_as_float(rr.y << 16); rv[3] = __uint_as_float(rr.y & 0xffff0000u);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float u = acc[r][a][j] * sv[a][j] + hv[a][j];
            if (p.act) u = silu_f(u);
            v[j] = u + rv[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = bf2f(f2bf(acc[r][a][j]));
            if (inb) { ssum[a][j] += v[j]; ssq[a][j] += v[j] * v[j]; }
          }
        }
        const sm_u32x2 u = {(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
        __builtin_amdgcn_raw_buffer_store_b64(u, ry, (inb & (co < p.Cout)) ? (pix + co) * 2u : OOB, 0, 0);
      }
    }
    buf ^= 1;
  }
  sm_wvm<0>();
  if (p.part) {
    __syncthreads();
    float* red = (float*)sH;  // [4 row pairs][NCT * 16][2]
#pragma unroll
    for (int a = 0; a < NCW; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s = wave_xor_sum16(ssum[a][j]), q = wave_xor_sum16(ssq[a][j]);
        if (lp == 0) {
          red[(wr * NCT * 16 + (wc * NCW + a) * 16 + 4 * lq + j) * 2] = s;
          red[(wr * NCT * 16 + (wc * NCW + a) * 16 + 4 * lq + j) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < p.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += red[(w * NCT * 16 + tid) * 2]; q += red[(w * NCT * 16 + tid) * 2 + 1]; }
      float* dst = p.part + ((long)wk * p.Cout + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// (two channel tiles split over eight waves leave a wave one tile: 1 + 2 fragment reads for 2 MFMAs - measured slower, 32.8 against 30.0 us at
// 32 -> 32 @160x160; four tiles gain: 33.5 -> 27.9 us at 64 -> 64 @80x80 in the training step, 37.2 -> 29.7 us in eval)
constexpr int sm_waves(int nct) { return nct == 4 ? 8 : 4; }
constexpr size_t sm_lds(int cb, int nct, int st) {
  const int nw = sm_waves(nct);
  const int spi = 1024 / cb, ninst = ((st == 1 ? 180 : 612) + spi - 1) / spi, ni = (ninst + nw - 1) / nw;
  return (size_t)9 * nct * 16 * cb + 2 * (size_t)ni * nw * 1024;
}

template <int CB, int NCT, bool AFF, int ST = 1>
void sm_launch(const SmP& p, int grid, hipStream_t st) {
  constexpr int NW = sm_waves(NCT);
  const size_t lds = sm_lds(CB, NCT, ST);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_small_kernel<CB, NCT, AFF, ST, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_small_kernel<CB, NCT, AFF, ST, NW>), dim3(grid), dim3(NW * 64), lds, st, p);
}

}  // namespace

extern "C" int y3d_get_stream1x1(void);

int y3d_conv3x3_small_ok(int dtype, int B, int H, int W, int Cin, int Cout, int rows) {
  if (!y3d_get_stream1x1() || dtype != Y3D_BF16 || Cin < 8 || Cin > 64 || Cin % 8 != 0 || Cout > 64 || Cout < 8 || Cout % 4 != 0) return 0;
  if (H < 4 || W < 8 || rows < 1) return 0;
  return 1;
}

// workgroups (= rows of BatchNorm partials) for this shape
int y3d_conv3x3_small_rows(int B, int H, int W, int Cin, int Cout) {
  const int nct = cdiv(Cout, 16), cb = Cin <= 32 ? 64 : 128;  // bytes of an LDS row
  const size_t lds = sm_lds(cb, nct, 1);
  const long ntiles = (long)B * cdiv(H, 8) * cdiv(W, 16);
  const long grid = 256 * (lds <= 80 * 1024 ? 2 : 1);
  return (int)(grid < ntiles ? grid : ntiles);
}

int y3d_conv3x3_small_launch(const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cin, int Cout, const void* w, int Ktot, void* y,
                             long ysw, float* part, int rows, int flip, const float* scale, const float* shift, int act, const void* res, long rsw,
                             void* stream) {
  Y3D_CHECK(((uintptr_t)x & 15) == 0 && xsb % 8 == 0 && xsh % 8 == 0 && xsw % 8 == 0 && ((uintptr_t)w & 15) == 0 && Ktot % 8 == 0 &&
                ((uintptr_t)y & 7) == 0 && ysw % 4 == 0, "conv3x3_small: operand alignment");
  const long xext = ((long)(B - 1) * xsb + (long)(H - 1) * xsh + (long)(W - 1) * xsw + Cin) * 2;
  const long yext = (((long)B * H * W - 1) * ysw + Cout) * 2;
  Y3D_CHECK(xext < (1L << 32) - 64 && yext < (1L << 32) - 64 && (long)B * xsb < (1L << 31) && (long)B * H * W * ysw < (1L << 31),
            "conv3x3_small: tensors beyond 32-bit byte offsets");
  SmP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.part = part;
  p.scale = scale; p.shift = shift; p.act = act; p.res = (const bf16_t*)res; p.rsw = (int)rsw;
  Y3D_CHECK(!(scale && part) && (scale || !res), "conv3x3_small: the affine epilogue carries no BatchNorm partials; a residual needs the affine form");
  Y3D_CHECK(!res || (((uintptr_t)res & 7) == 0 && rsw % 4 == 0 && (long)B * H * W * rsw < (1L << 31)), "conv3x3_small: residual alignment");
  p.xsb = (int)xsb; p.xsh = (int)xsh; p.xsw = (int)xsw; p.ysw = (int)ysw;
  p.B = B; p.H = H; p.W = W; p.Hi = H; p.Wi = W; p.Cin = Cin; p.Cout = Cout; p.Ktot = Ktot; p.flip = flip; p.rows = rows;
  p.nty = cdiv(H, 8); p.ntx = cdiv(W, 16); p.ntiles = B * p.nty * p.ntx;
  p.xbytes = (unsigned)xext; p.ybytes = (unsigned)yext;
  const int nct = cdiv(Cout, 16);
  const int cb = Cin <= 32 ? 64 : 128;
  const int grid = y3d_conv3x3_small_rows(B, H, W, Cin, Cout);
  Y3D_CHECK(!part || rows == grid, "conv3x3_small: the partial buffer must have y3d_conv2d_stat_rows rows (%d given, %d written)", rows, grid);
  hipStream_t st = (hipStream_t)stream;
#define SM_GO(CB, AFF)                                                   \
  switch (nct) {                                                         \
    case 1: sm_launch<CB, 1, AFF>(p, grid, st); break;                   \
    case 2: sm_launch<CB, 2, AFF>(p, grid, st); break;                   \
    case 3: sm_launch<CB, 3, AFF>(p, grid, st); break;                   \
    default: sm_launch<CB, 4, AFF>(p, grid, st); break;                  \
  }
  if (scale) { if (cb == 128) { SM_GO(128, true) } else { SM_GO(64, true) } }
  else { if (cb == 128) { SM_GO(128, false) } else { SM_GO(64, false) } }
#undef SM_GO
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

// ---- stride 2 (forward only): input (B, H, W, Cin <= 32), output (B, Ho, Wo, Cout <= 64), 3x3, pad 1 --------------------------------
int y3d_conv3x3_small_s2_ok(int dtype, int B, int H, int W, int Cin, int Cout) {
  if (!y3d_get_stream1x1() || dtype != Y3D_BF16 || Cin < 8 || Cin > 32 || Cin % 8 != 0 || Cout > 64 || Cout < 8 || Cout % 4 != 0) return 0;
  if (H < 8 || W < 16) return 0;
  return 1;
}

int y3d_conv3x3_small_s2_rows(int B, int H, int W, int Cin, int Cout) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long ntiles = (long)B * cdiv(Ho, 8) * cdiv(Wo, 16);
  const long grid = 256 * (sm_lds(64, cdiv(Cout, 16), 2) <= 80 * 1024 ? 2 : 1);
  return (int)(grid < ntiles ? grid : ntiles);
}

int y3d_conv3x3_small_s2_launch(const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cin, int Cout, const void* w, int Ktot, void* y,
                                long ysw, float* part, int rows, const float* scale, const float* shift, int act, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  Y3D_CHECK(((uintptr_t)x & 15) == 0 && xsb % 8 == 0 && xsh % 8 == 0 && xsw % 8 == 0 && ((uintptr_t)w & 15) == 0 && Ktot % 8 == 0 &&
                ((uintptr_t)y & 7) == 0 && ysw % 4 == 0, "conv3x3_small_s2: operand alignment");
  const long xext = ((long)(B - 1) * xsb + (long)(H - 1) * xsh + (long)(W - 1) * xsw + Cin) * 2;
  const long yext = (((long)B * Ho * Wo - 1) * ysw + Cout) * 2;
  Y3D_CHECK(xext < (1L << 32) - 64 && yext < (1L << 32) - 64 && (long)B * xsb < (1L << 31) && (long)B * Ho * Wo * ysw < (1L << 31),
            "conv3x3_small_s2: tensors beyond 32-bit byte offsets");
  Y3D_CHECK(!(scale && part), "conv3x3_small_s2: the affine epilogue carries no BatchNorm partials");
  SmP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.part = part;
  p.scale = scale; p.shift = shift; p.act = act; p.res = nullptr; p.rsw = 0;
  p.xsb = (int)xsb; p.xsh = (int)xsh; p.xsw = (int)xsw; p.ysw = (int)ysw;
  p.B = B; p.H = Ho; p.W = Wo; p.Hi = H; p.Wi = W; p.Cin = Cin; p.Cout = Cout; p.Ktot = Ktot; p.flip = 0; p.rows = rows;
  p.nty = cdiv(Ho, 8); p.ntx = cdiv(Wo, 16); p.ntiles = B * p.nty * p.ntx;
  p.xbytes = (unsigned)xext; p.ybytes = (unsigned)yext;
  const int nct = cdiv(Cout, 16);
  const int grid = y3d_conv3x3_small_s2_rows(B, H, W, Cin, Cout);
  Y3D_CHECK(!part || rows == grid, "conv3x3_small_s2: the partial buffer must have y3d_conv2d_stat_rows rows (%d given, %d written)", rows, grid);
  hipStream_t st = (hipStream_t)stream;
#define SM_GO2(AFF)                                                      \
  switch (nct) {                                                         \
    case 1: sm_launch<64, 1, AFF, 2>(p, grid, st); break;                \
    case 2: sm_launch<64, 2, AFF, 2>(p, grid, st); break;                \
    case 3: sm_launch<64, 3, AFF, 2>(p, grid, st); break;                \
    default: sm_launch<64, 4, AFF, 2>(p, grid, st); break;               \
  }
  if (scale) { SM_GO2(true) } else { SM_GO2(false) }
#undef SM_GO2
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}
