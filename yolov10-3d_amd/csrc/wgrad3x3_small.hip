// Weight gradient of the NARROW 3x3 stride-1 "same" convs (32 or 64 input channels - or 40 / 48 / 56 in zero-padded 64-channel rows: the M
// widths - 32 / 48 / 64 output channels; bf16) - the partner of
// conv3x3_small.hip:
//
//     slab[split][co][tap * Cin + ci] = sum over the split's pixels p of dy[p][co] * x[p + tap][ci]
//
// conv3x3_wgrad_tile owns 128 output x 64 input channels per workgroup (half or three quarters of its MFMAs are padding on these
// layers: 300 TFLOP/s at 64 -> 64, and 32 input channels fall to the generic split-K kernel: 130 TFLOP/s).  Here one workgroup
// (4 waves) owns ALL of dW (9 x Cout x Cin fp32 in registers: a wave keeps the nine taps of its input-channel tile for all of its
// output-channel tiles, 36 accumulator tiles at 64 -> 64) and walks 8 x 16 pixel tiles: the dy tile and the (8 + 2) x (16 + 2) x halo
// arrive by LDS-DMA (double buffered, padded rows: the padding chunks are lanes with an out-of-range source), fragments come from
// the transposed LDS read (inline asm, see conv3x3_wgrad_tile.hip) - dy^T once per 32-pixel step for all nine taps, the shifted x
// windows per filter row, software-pipelined one group of reads ahead of the MFMAs with counted lgkmcnt waits.
#include "common.h"
#include <type_traits>

namespace {

struct WsP {
  const bf16_t* x;
  const bf16_t* dy;
  float* slab;  // [gridDim.x][Cout][9 * Cin]
  int xsb, xsh, xsw, dsw;
  int B, H, W, Cout, Cin;  // Cin <= CB / 2 real input channels (the rest of an LDS row stays zero)
  int nty, ntx, ntiles;
  unsigned xbytes, dbytes;
};

template <int OFF>
__device__ __forceinline__ s16x4_t ws_tr(unsigned a) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
// all but the N youngest LDS reads have landed; the fragments of the group about to be used are in/out operands so that their uses
// stay behind the wait
template <int N, int CT>
__device__ __forceinline__ void ws_wait(s16x4_t (&b)[3][2], s16x4_t (&a)[CT][2]) {
  if constexpr (CT == 1)
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(a[0][0]), "+v"(a[0][1]) : "n"(N));
  else if constexpr (CT == 2)
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(a[0][0]), "+v"(a[0][1]),
                 "+v"(a[1][0]), "+v"(a[1][1]) : "n"(N));
  else if constexpr (CT == 3)
    asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(a[0][0]), "+v"(a[0][1]),
                 "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]) : "n"(N));
  else {
    static_assert(CT == 4, "1..4 output-channel tiles per wave");
    asm volatile("s_waitcnt lgkmcnt(%14)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(a[0][0]), "+v"(a[0][1]),
                 "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]) : "n"(N));
  }
}

// CB: bytes of a pixel's input channels (64: Cin = 32, 128: Cin = 64); NCT: 16-channel output tiles (Cout = 16 NCT)
template <int CB, int NCT>
__global__ __launch_bounds__(256) void wgrad3x3_small_kernel(WsP p) {
  constexpr int CIN = CB / 2, NCIT = CIN / 16;
  constexpr int WCI = NCIT >= 4 ? 4 : NCIT;   // waves along the input-channel tiles (one tile each)
  constexpr int WCO = 4 / WCI;                // waves along the output-channel tiles
  constexpr int CT = NCT / WCO;               // output-channel tiles per wave
  static_assert(NCT % WCO == 0 && CT >= 1 && CT <= 4, "output-channel tiles must split over the waves");
  constexpr int HW = 18, NPIX = 10 * HW;
  constexpr int CPD = NCT * 2 + 2, PD = CPD * 16;     // dy rows: Cout channels + 32 bytes of padding
  constexpr int CPX = CB / 16 + 2, PX = CPX * 16;     // halo rows
  constexpr int ID = 2 * CPD;                          // DMA instructions of the dy tile (128 rows x CPD chunks / 64 lanes)
  constexpr int IX = (NPIX * CPX + 63) / 64;
  constexpr int NI = (ID + IX + 3) / 4;                // DMA instructions per wave per tile (the last ones may be dummies)
  constexpr int DBY = ID * 1024, BUF = (ID + IX) * 1024;
  constexpr int NBUF = 3;  // two tiles in flight under the one being multiplied: with one, every tile waited ~1 us for its data (1 wave per SIMD)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NBUF][dy tile | halo] + 1 KB dump
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave % WCI, wco = wave / WCI;
  const int nwk = gridDim.x, wk = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;  // LDS byte address (integers from here on)
  auto decode = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % p.ntx, t2 = t / p.ntx;
    x0 = tx * 16;
    y0 = (t2 % p.nty) * 8;
    b = t2 / p.nty;
  };
  // DMA plan of this lane, computed ONCE: instruction gi = wave * NI + n of a tile is a piece of the dy tile (gi < ID), of the halo, or
  // a dummy; per instruction the lane's pixel (row, column relative to the tile origin) and its element offset relative to the
  // origin pixel.  Per tile that leaves two compares, one add and a select per instruction (recomputing the chunk -> pixel maps cost
  // ~1000 instructions per tile and wave, more than the MFMAs they feed).
  int p_off[NI], p_rc[NI];  // element offset; (row << 8) | column, row 255 = never valid
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int gi = wave * NI + n;
    if (gi < ID) {
      const int q = gi * 64 + lane, m = q / CPD, col = q - m * CPD;
      const bool ok = (col < NCT * 2) & (col * 8 < p.Cout);
      p_off[n] = ((m >> 4) * p.W + (m & 15)) * p.dsw + col * 8;
      p_rc[n] = ((ok ? (m >> 4) : 255) << 8) | (m & 15);
    } else {
      const int q = (gi - ID) * 64 + lane, P = q / CPX, col = q - P * CPX;
      const int hr = P / HW, hc = P - hr * HW;
      const bool ok = (gi < ID + IX) & (P < NPIX) & (col < CB / 16) & (col * 8 < p.Cin);
      p_off[n] = (hr - 1) * p.xsh + (hc - 1) * p.xsw + col * 8;
      p_rc[n] = ((ok ? hr : 255) << 8) | hc;  // halo coordinates: image row y0 - 1 + hr, column x0 - 1 + hc
    }
  }
  auto issue = [&](int t, int buf) {
    int b, y0, x0;
    const bool live = t < p.ntiles;
    decode(live ? t : 0, b, y0, x0);
    const int baseD = ((b * p.H + y0) * p.W + x0) * p.dsw, baseX = b * p.xsb + y0 * p.xsh + x0 * p.xsw;
    const int hlim = live ? p.H : 0;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int r = p_rc[n] >> 8, c = p_rc[n] & 255;
      const int gi = wave * NI + n;  // uniform
      if (gi < ID) {
        const bool ok = (y0 + r < hlim) & (x0 + c < p.W);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(size_t)(sbase + buf * BUF + gi * 1024), 16,
                                                 ok ? (unsigned)(baseD + p_off[n]) * 2u : OOB, 0, 0, 0);
      } else {
        const bool ok = ((unsigned)(y0 - 1 + r) < (unsigned)hlim) & ((unsigned)(x0 - 1 + c) < (unsigned)p.W);
        const unsigned dst = gi < ID + IX ? sbase + buf * BUF + gi * 1024 : sbase + NBUF * BUF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(size_t)dst, 16, ok ? (unsigned)(baseX + p_off[n]) * 2u : OOB, 0, 0, 0);
      }
    }
  };
  issue(wk, 0);
  issue(wk + nwk, 1);

  f32x4_t acc[9][CT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < CT; ++a) acc[t][a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, li = lane & 15;
  // transposed-read addresses of this lane inside a 32-pixel step: pixel rows 4g + (li >> 2) (+ 16), columns 4 (li & 3) of a 16-wide tile
  const unsigned offA = (4 * g + (li >> 2)) * PD + (wco * CT * 16 + 4 * (li & 3)) * 2;
  const unsigned offB = DBY + (4 * g + (li >> 2)) * PX + (wci * 16 + 4 * (li & 3)) * 2;
  int buf = 0;
#pragma unroll 1
  for (int t = wk; t < p.ntiles; t += nwk) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");  // this tile has landed; the next one's rounds stay in flight (no stores in the loop)
    __builtin_amdgcn_s_barrier();
    issue(t + 2 * nwk, buf >= 1 ? buf - 1 : NBUF - 1);  // the buffer of the tile before this one
    const unsigned tb = sbase + buf * BUF;
    const unsigned aA = tb + offA, aB = tb + offB;
    // 12 groups (4 steps of 32 pixels x 3 filter rows); group (ks, r): the x windows of taps (r, 0..2); dy^T rides with r = 0
    s16x4_t fb[2][3][2], fa[2][CT][2];
    auto reads = [&](auto KS, auto R, auto SETC) {
      constexpr int ks = decltype(KS)::value, r = decltype(R)::value, set = decltype(SETC)::value;
      if constexpr (r == 0) {
        constexpr int kb = ks & 1;
        fa[kb][0][0] = ws_tr<(32 * ks) * PD>(aA);
        fa[kb][0][1] = ws_tr<(32 * ks + 16) * PD>(aA);
        if constexpr (CT > 1) { fa[kb][1][0] = ws_tr<(32 * ks) * PD + 32>(aA); fa[kb][1][1] = ws_tr<(32 * ks + 16) * PD + 32>(aA); }
        if constexpr (CT > 2) { fa[kb][2][0] = ws_tr<(32 * ks) * PD + 64>(aA); fa[kb][2][1] = ws_tr<(32 * ks + 16) * PD + 64>(aA); }
        if constexpr (CT > 3) { fa[kb][3][0] = ws_tr<(32 * ks) * PD + 96>(aA); fa[kb][3][1] = ws_tr<(32 * ks + 16) * PD + 96>(aA); }
      }
      // pixel (row 2 ks + hi, col) of the tile under tap (r, q) is halo slot (2 ks + hi + r) * 18 + col + q
      fb[set][0][0] = ws_tr<((2 * ks + r) * HW + 0) * PX>(aB);
      fb[set][0][1] = ws_tr<((2 * ks + 1 + r) * HW + 0) * PX>(aB);
      fb[set][1][0] = ws_tr<((2 * ks + r) * HW + 1) * PX>(aB);
      fb[set][1][1] = ws_tr<((2 * ks + 1 + r) * HW + 1) * PX>(aB);
      fb[set][2][0] = ws_tr<((2 * ks + r) * HW + 2) * PX>(aB);
      fb[set][2][1] = ws_tr<((2 * ks + 1 + r) * HW + 2) * PX>(aB);
    };
    auto mfmas = [&](auto KS, auto R, auto SETC) {
      constexpr int ks = decltype(KS)::value, r = decltype(R)::value, set = decltype(SETC)::value;
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(fb[set][q][0], fb[set][q][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int a = 0; a < CT; ++a) {
          const bf16x8_t av = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(fa[ks & 1][a][0], fa[ks & 1][a][1], 0, 1, 2, 3, 4, 5, 6, 7));
          acc[r * 3 + q][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, acc[r * 3 + q][a], 0, 0, 0);
        }
      }
    };
#define WS_I(n) std::integral_constant<int, n>()
#define WS_STEP(KS, R, NKS, NR, SET, NEXTN)                                     \
    reads(WS_I(NKS), WS_I(NR), WS_I((SET) ^ 1));                                \
    ws_wait<NEXTN, CT>(fb[SET], fa[(KS) & 1]);                                  \
    mfmas(WS_I(KS), WS_I(R), WS_I(SET));
    reads(WS_I(0), WS_I(0), WS_I(0));
    // NEXTN: reads issued for the NEXT group (6, plus 2 CT when it brings the next step's dy^T)
    WS_STEP(0, 0, 0, 1, 0, 6) WS_STEP(0, 1, 0, 2, 1, 6) WS_STEP(0, 2, 1, 0, 0, 6 + 2 * CT)
    WS_STEP(1, 0, 1, 1, 1, 6) WS_STEP(1, 1, 1, 2, 0, 6) WS_STEP(1, 2, 2, 0, 1, 6 + 2 * CT)
    WS_STEP(2, 0, 2, 1, 0, 6) WS_STEP(2, 1, 2, 2, 1, 6) WS_STEP(2, 2, 3, 0, 0, 6 + 2 * CT)
    WS_STEP(3, 0, 3, 1, 1, 6) WS_STEP(3, 1, 3, 2, 0, 6)
    ws_wait<0, CT>(fb[1], fa[1]);
    mfmas(WS_I(3), WS_I(2), WS_I(1));
#undef WS_STEP
#undef WS_I
    if (++buf == NBUF) buf = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // D[row = 4g + r -> output channel of tile a][column li -> input channel of this wave's tile]
  const int cin = p.Cin;
  float* slab = p.slab + (long)wk * p.Cout * (9 * cin);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = (wco * CT + a) * 16 + 4 * g + r;
        if (co < p.Cout && wci * 16 + li < cin) slab[(long)co * (9 * cin) + t * cin + wci * 16 + li] = acc[t][a][r];
      }
}

template <int CB, int NCT>
void ws_launch(const WsP& p, int grid, hipStream_t st) {
  constexpr int CPD = NCT * 2 + 2, CPX = CB / 16 + 2, ID = 2 * CPD, IX = (180 * CPX + 63) / 64;
  const size_t lds = (size_t)3 * (ID + IX) * 1024 + 1024;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad3x3_small_kernel<CB, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((wgrad3x3_small_kernel<CB, NCT>), dim3(grid), dim3(256), lds, st, p);
}

}  // namespace

extern "C" int y3d_get_stream1x1(void);

int y3d_wgrad3x3_small_ok(int dtype, int B, int H, int W, int Cin, int Cout) {
  if (!y3d_get_stream1x1() || dtype != Y3D_BF16 || (Cin != 32 && (Cin <= 32 || Cin > 64 || Cin % 8 != 0)) || Cout % 16 != 0 || Cout < 16 || Cout > 64) return 0;
  if (Cin == 32 && (Cout / 16) % 2 != 0) return 0;  // two waves share the output-channel tiles
  if (H < 4 || W < 8) return 0;
  return 1;
}

int y3d_wgrad3x3_small_splits(int B, int H, int W) {
  const long ntiles = (long)B * cdiv(H, 8) * cdiv(W, 16);
  // one workgroup per CU (up to 148 KB of LDS), one slab each.  (Two sets of four waves per workgroup, each with its own slab, doubled
  // the MFMA issue slots AND the slab traffic: 64 -> 64 at 80x80 went from 45 to 56 us per layer - the split-K epilogue is the bound.)
  return (int)(ntiles < 256 ? ntiles : 256);
}

int y3d_wgrad3x3_small_launch(const void* x, long xsb, long xsh, long xsw, const void* dy, long dsw, int B, int H, int W, int Cin, int Cout,
                              float* slab, int nsplit, void* stream) {
  Y3D_CHECK(((uintptr_t)x & 15) == 0 && xsb % 8 == 0 && xsh % 8 == 0 && xsw % 8 == 0 && ((uintptr_t)dy & 15) == 0 && dsw % 8 == 0,
            "wgrad3x3_small: operand alignment");
  Y3D_CHECK(nsplit == y3d_wgrad3x3_small_splits(B, H, W), "wgrad3x3_small: nsplit must come from y3d_conv2d_wgrad_plan");
  const long xext = ((long)(B - 1) * xsb + (long)(H - 1) * xsh + (long)(W - 1) * xsw + Cin) * 2;
  const long dext = (((long)B * H * W - 1) * dsw + Cout) * 2;
  Y3D_CHECK(xext < (1L << 32) - 64 && dext < (1L << 32) - 64 && (long)B * xsb < (1L << 31) && (long)B * H * W * dsw < (1L << 31),
            "wgrad3x3_small: tensors beyond 32-bit byte offsets");
  WsP p;
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.slab = slab;
  p.xsb = (int)xsb; p.xsh = (int)xsh; p.xsw = (int)xsw; p.dsw = (int)dsw;
  p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.Cin = Cin;
  p.nty = cdiv(H, 8); p.ntx = cdiv(W, 16); p.ntiles = B * p.nty * p.ntx;
  p.xbytes = (unsigned)xext; p.dbytes = (unsigned)dext;
  hipStream_t st = (hipStream_t)stream;
  const int nct = Cout / 16;
  if (Cin > 32) {
    switch (nct) {
      case 1: ws_launch<128, 1>(p, nsplit, st); break;
      case 2: ws_launch<128, 2>(p, nsplit, st); break;
      case 3: ws_launch<128, 3>(p, nsplit, st); break;
      default: ws_launch<128, 4>(p, nsplit, st); break;
    }
  } else {
    if (nct == 2) ws_launch<64, 2>(p, nsplit, st); else ws_launch<64, 4>(p, nsplit, st);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}
