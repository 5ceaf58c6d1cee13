// Weight gradient of a 3x3 stride-1 "same" convolution with the dy tile and the x halo tile resident in LDS (bf16).
//
//   dW[co][tap][ci] = sum_p dy[p][co] * x[p + tap][ci]
//
// One workgroup owns 128 output channels x one 64-channel input slab of one group and walks a range of TH x 16 pixel tiles.
// Per tile the dy tile (pixels x 128 co) and the (TH+2) x 18 x 64 x-halo slab arrive by LDS-DMA ONCE and feed all nine taps:
// 9 x (128 x 64) fp32 accumulators live in registers across the whole pixel range (144 VGPRs per lane), so the global->LDS
// traffic is ~343 FLOP/B instead of the 64 FLOP/B of the generic split-K kernel (conv_gemm.hip), which is L1-fill bound.
// Both MFMA operands are pixel-major in memory (NHWC) and the reduction runs over pixels, so fragments are fetched with the
// gfx950 transposed LDS read; the 16-byte chunks are XOR-swizzled on the DMA *source* side so that the eight rows a 32-lane
// half touches per read sit in eight different 32-byte bank groups (conflict-free for both images).
// Split-K over pixel-tile ranges into fp32 slabs; the existing wgrad_reduce kernel sums them into the OIHW gradient.
#include "common.h"
#include <cstdlib>

namespace {

__device__ uint4 y3d_wg_zero_page[4];

struct WG3P {
  const void* x;
  const void* dy;
  float* slab;     // [nsplit][G*Cn][9*Cg]
  long xsb, xsh, xsw, dsw;
  int B, H, W;
  int Cg, Cn, G;
  int ntx, nty;    // tiles per row / per image column
  int ntiles, tiles_per_split, nslab;
};

typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// Transposed LDS read as inline asm.  Through the builtin, hipcc (ROCm 7.2) orders every such read behind ALL outstanding LDS-DMA
// (`s_waitcnt vmcnt(0)` before the first read of each K step), i.e. the next tile's DMA could never overlap this tile's MFMAs.
// The asm form is invisible to that dependency tracking; the DMA'd buffer is only read after the kernel's own vmcnt(0) + barrier.
// The result is not tracked either: `lds_wait` (below) must sit between the reads and their first use.
template <int OFF>  // OFF: immediate byte offset (the row stride multiples of the fragment loop: no VALU address math per read)
__device__ __forceinline__ s16x4_t ds_tr16(unsigned a) {
  s16x4_t v;
#ifdef Y3D_WGP_NOLDS
  asm volatile("" : "=v"(v) : "v"(a));
  return v;
#endif
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
// waits for every LDS read issued so far; the fragments are in/out operands so that their uses stay behind the wait
// (N newest LDS operations may stay in flight: the next tap's fragments are issued before the current tap's are awaited)
template <int N>
__device__ __forceinline__ void lds_wait(s16x4_t& a, s16x4_t& b, s16x4_t& c, s16x4_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

template <int N>
__device__ __forceinline__ void lds_wait8(s16x4_t& a, s16x4_t& b, s16x4_t& c, s16x4_t& d, s16x4_t& e, s16x4_t& f, s16x4_t& g, s16x4_t& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}

template <int TH>
__global__ __launch_bounds__(512) void conv3x3_wgrad_tile_kernel(WG3P p) {
  typedef bf16_t T;
  constexpr int NT = 512;
  constexpr int NPX = TH * 16;
  constexpr int HWD = 18;
  constexpr int NPIXH = (TH + 2) * HWD;
  constexpr int DR = NPX * 16 / NT;                    // DMA rounds of the dy tile (256-byte rows)
  constexpr int HCH = NPIXH * 8;
  constexpr int HR = (HCH + NT - 1) / NT;              // DMA rounds of the halo slab (128-byte rows)
  constexpr int DBYTES = NPX * 256, HBYTES = HR * NT * 16, SBYTES = DBYTES + HBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;             // 32-channel co block, 32-channel ci block
  const int g = blockIdx.z, split = blockIdx.y;
  const int slab_i = blockIdx.x % p.nslab, cot = blockIdx.x / p.nslab;
  const int c0 = cot * 128, ci0 = slab_i * 64;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ D = (const T*)p.dy;
  const T* zero = (const T*)y3d_wg_zero_page;

  // static part of the DMA addressing
  int d_row[DR], d_px[DR], d_c[DR];
#pragma unroll
  for (int rd = 0; rd < DR; ++rd) {
    int chunk = rd * NT + tid;
    int pp = chunk >> 4, s = chunk & 15;
    d_row[rd] = pp >> 4;
    d_px[rd] = pp & 15;
    d_c[rd] = (s ^ ((pp & 7) << 1)) * 8;               // channel offset of the chunk this lane fetches
  }
  int h_y[HR], h_x[HR], h_c[HR];
#pragma unroll
  for (int rd = 0; rd < HR; ++rd) {
    int chunk = rd * NT + tid;
    int P = chunk >> 3, s = chunk & 7;
    h_y[rd] = chunk < HCH ? P / HWD : -100000;
    h_x[rd] = P - (P / HWD) * HWD;
    h_c[rd] = (s ^ (((h_x[rd] >> 1) & 3) << 1)) * 8;  // swizzle by the pixel's x inside the halo row: the same for every row
  }
  auto issue = [&](int tile, int buf) {
    int tx = tile % p.ntx;
    int t2 = tile / p.ntx;
    int ty = t2 % p.nty;
    int b = t2 / p.nty;
    const int x0 = tx * 16, y0 = ty * TH;
    char* db = smem + buf * SBYTES;
    char* hb = db + DBYTES;
#pragma unroll
    for (int rd = 0; rd < DR; ++rd) {
      int xx = x0 + d_px[rd];
      bool ok = xx < p.W && c0 + d_c[rd] < p.Cn;
      const T* src = ok ? D + (((long)b * p.H + y0 + d_row[rd]) * p.W + xx) * p.dsw + (long)g * p.Cn + c0 + d_c[rd] : zero;
#ifndef Y3D_WGP_NODMA
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(db + (rd * NT + wave * 64) * 16), 16, 0, 0);
#else
      asm volatile("" ::"v"(src));
#endif
    }
#pragma unroll
    for (int rd = 0; rd < HR; ++rd) {
      if (rd * NT + tid < HCH) {
        int yy = y0 + h_y[rd] - 1, xx = x0 + h_x[rd] - 1;
        bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && ci0 + h_c[rd] < p.Cg;  // Cg % 64 == 32: the last slab is half zeros
        const T* src = ok ? X + (long)b * p.xsb + (long)yy * p.xsh + (long)xx * p.xsw + (long)g * p.Cg + ci0 + h_c[rd] : zero;
#ifndef Y3D_WGP_NODMA
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(hb + (rd * NT + wave * 64) * 16), 16, 0, 0);
#else
        asm volatile("" ::"v"(src));
#endif
      }
    }
  };

  f32x4_t acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[t][a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int t_beg = split * p.tiles_per_split;
  const int t_end = min(p.ntiles, t_beg + p.tiles_per_split);
  const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp4 = li & 3;

  // per-lane LDS byte offsets of the fragment reads, computed ONCE: with them recomputed per read (~9 integer ops x 28 reads per K
  // chunk and wave) the loop was VALU-bound behind its own address arithmetic (ablation: no reads + no addresses = +25 %).
  // dy tile: pixel (2ks + hi) * 16 + 4 grp + qq -> (2ks + hi) * 4096 + aoff[mt]; the swizzle only sees the pixel's low three bits.
  // halo:    pixel (2ks + R) * 18 + x, x = 4 grp + qq + q -> (2ks + R) * 2304 + boff[q][nt]; the swizzle only sees x.
  unsigned aoff[2], boff[3][2];
  {
    const int px8 = 4 * grp + qq;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int cch = ((wi * 32 + mt * 16) >> 3) + (pp4 >> 1);
      aoff[mt] = px8 * 256 + ((cch ^ ((px8 & 7) << 1)) << 4) + (pp4 & 1) * 8;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int x = px8 + q;
        const int cch = ((wj * 32 + nt * 16) >> 3) + (pp4 >> 1);
        boff[q][nt] = x * 128 + ((cch ^ (((x >> 1) & 3) << 1)) << 4) + (pp4 & 1) * 8;
      }
  }
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;

  if (t_beg < t_end) issue(t_beg, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int cur = (tile - t_beg) & 1;
    if (tile + 1 < t_end) issue(tile + 1, cur ^ 1);
#pragma unroll  // the addresses are immediates now: unrolled, the next K chunk's reads overlap this one's MFMAs (+3 %)
    for (int ks = 0; ks < NPX / 32; ++ks) {
      // MFMA k = 8*grp + j  <->  pixel (row 2*ks + (j>>2), x 4*grp + (j&3)): a 32-lane half reads 8 consecutive pixels per instruction
      bf16x8_t fa[2];
      s16x4_t va[2][2];
      const unsigned dks = smem_base + cur * SBYTES + ks * 8192;               // dy rows 2ks, 2ks + 1
      const unsigned hks = smem_base + cur * SBYTES + DBYTES + ks * (2 * HWD * 128);  // halo rows 2ks ..
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        va[mt][0] = ds_tr16<0>(dks + aoff[mt]);
        va[mt][1] = ds_tr16<4096>(dks + aoff[mt]);
      }
      // The loop is LDS-read bound (a wave's 32 x 32 output tile reuses an x fragment for only two MFMAs), so the x rows are read
      // ONCE per column shift q and shared by the three filter rows: tap (r, q) pairs halo rows (2ks + r, 2ks + r + 1), i.e. rows
      // R0..R3 serve r = 0, 1, 2 as (R0,R1), (R1,R2), (R2,R3) -- 8 transposed reads per q instead of 12 (28 per K chunk instead of 40).
      // Software pipeline over q: the next shift's eight reads are issued before the current one's are awaited.
      s16x4_t vr[2][2][4];
      auto issue_q = [&](int q, int buf) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const unsigned a = hks + boff[q][nt];
          vr[buf][nt][0] = ds_tr16<0>(a);
          vr[buf][nt][1] = ds_tr16<HWD * 128>(a);
          vr[buf][nt][2] = ds_tr16<2 * HWD * 128>(a);
          vr[buf][nt][3] = ds_tr16<3 * HWD * 128>(a);
        }
      };
      issue_q(0, 0);
      lds_wait<8>(va[0][0], va[0][1], va[1][0], va[1][1]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(va[mt][0], va[mt][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int cb = q & 1;
        if (q < 2) {
          issue_q(q + 1, cb ^ 1);
          lds_wait8<8>(vr[cb][0][0], vr[cb][0][1], vr[cb][0][2], vr[cb][0][3], vr[cb][1][0], vr[cb][1][1], vr[cb][1][2], vr[cb][1][3]);
        } else {
          lds_wait8<0>(vr[cb][0][0], vr[cb][0][1], vr[cb][0][2], vr[cb][0][3], vr[cb][1][0], vr[cb][1][1], vr[cb][1][2], vr[cb][1][3]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          bf16x8_t fb[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            fb[nt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(vr[cb][nt][r], vr[cb][nt][r + 1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#ifndef Y3D_WGP_NOMFMA
              acc[r * 3 + q][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[r * 3 + q][mt][nt], 0, 0, 0);
#else
              asm volatile("" ::"v"(fa[mt]), "v"(fb[nt]));
#endif
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // D[i][j]: row i = co = (lane>>4)*4 + reg, col j = ci = lane&15
  const int Ktot = 9 * p.Cg;
  float* slab = p.slab + ((long)split * p.G * p.Cn + (long)g * p.Cn) * Ktot;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        int ci = ci0 + wj * 32 + nt * 16 + (lane & 15);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          int co = c0 + wi * 32 + mt * 16 + (lane >> 4) * 4 + rg;
          if (co < p.Cn && ci < p.Cg) slab[(long)co * Ktot + tap * p.Cg + ci] = acc[tap][mt][nt][rg];
        }
      }
}

template <int TH>
int launch_wg(const WG3P& p, int nsplit, hipStream_t st) {
  constexpr int NPX = TH * 16;
  constexpr int HCH = (TH + 2) * 18 * 8;
  constexpr int HR = (HCH + 511) / 512;
  size_t sm = 2 * ((size_t)NPX * 256 + (size_t)HR * 512 * 16);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_tile_kernel<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  dim3 grid(p.nslab * cdiv(p.Cn, 128), nsplit, p.G);
  hipLaunchKernelGGL((conv3x3_wgrad_tile_kernel<TH>), grid, dim3(512), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

// tile height of the resident wgrad kernel for this geometry (bf16 only), 0 -> generic split-K kernel
int y3d_wgrad_tile_height(int dtype, int H, int W, int Cg, int Cn, int kh, int kw, int stride, int pad) {
  if (dtype != Y3D_BF16 || kh != 3 || kw != 3 || stride != 1 || pad != 1) return 0;
  // 96 / 160 / ... channels: the last 64-channel slab is half empty; 80 (X widths): a quarter full - 62 % useful MFMAs, against
  // 266 TFLOP/s for the generic split-K kernel on 80 -> 80 at 160x160 (the DMA zero-fills per 8-channel chunk: `ci0 + h_c < Cg`)
  if (Cg % 16 != 0 || Cg < 64 || Cn % 8 != 0 || W < 8) return 0;
  if (H % 8 == 0) return 8;
  if (H % 4 == 0) return 4;
  return 0;
}

int y3d_wgrad_tile_splits(int th, int B, int H, int W, int Cg, int Cn, int G) {
  long ntiles = (long)B * (H / th) * cdiv(W, 16);
  long blocks = (long)cdiv(Cg, 64) * cdiv(Cn, 128) * G;
  // one round of workgroups: TH = 8 leaves room for one workgroup per CU (112 KB LDS), TH = 4 for two.  More splits only add slab
  // traffic (each workgroup writes its 9 x 128 x 64 fp32 accumulators) and a ragged second round: 512 -> 256 measured +2 % on the
  // head layers, +12 % at 40x40, +35-40 % on 96- / 192-channel body layers
  const int target8 = 256;
  // rounded DOWN: 27 channel blocks (192 -> 1152, the M head) x 10 splits = 270 workgroups ran as a full round plus a 14-workgroup tail
  // of the same length (733 TFLOP/s against 1 110 for the S head, whose 32 blocks x 8 fill the 256 CUs exactly); 9 splits = 243 fit one
  const long tgt = th == 8 ? target8 : 2 * target8;
  long want = tgt / blocks;
  if (want > ntiles) want = ntiles;
  long slab_bytes = (long)G * Cn * 9 * Cg * 4;
  long cap = (64L << 20) / slab_bytes;  // keep the fp32 partial slabs of one layer under ~64 MB
  if (cap < 16) cap = 16;
  if (cap > 256) cap = 256;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

int y3d_conv3x3_wgrad_tile_launch(int th, const void* x, long xsb, long xsh, long xsw, const void* dy, long dsw, int B, int H, int W, int Cg,
                                  int Cn, int G, float* slab, int nsplit, void* stream) {
  WG3P p;
  p.x = x; p.dy = dy; p.slab = slab;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.dsw = dsw;
  p.B = B; p.H = H; p.W = W; p.Cg = Cg; p.Cn = Cn; p.G = G;
  p.ntx = cdiv(W, 16); p.nty = H / th;
  p.ntiles = B * p.nty * p.ntx;
  p.tiles_per_split = cdiv(p.ntiles, nsplit);
  p.nslab = cdiv(Cg, 64);
  hipStream_t st = (hipStream_t)stream;
  if (th == 8) return launch_wg<8>(p, nsplit, st);
  return launch_wg<4>(p, nsplit, st);
}
