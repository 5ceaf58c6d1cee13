// fp8 (OCP e4m3) 3x3 stride-1 "same" convolution FORWARD on the block-scaled matrix-core instruction of gfx950,
// v_mfma_scale_f32_16x16x128_f8f6f4 - BASELINE.json configs[4] ("fp8 MFMA weights"); no reference counterpart (the reference has no
// 8-bit path).  Operands follow the OCP MX convention the instruction implements in hardware:
//   activations  q[pixel][c] e4m3 codes + one E8M0 scale byte per (pixel, 32-channel block): x ~ value(q) * 2^(s - 127)
//                (y3d_fp8_quantize_act: scale = smallest power of two with amax(block) / scale <= 448 - nothing saturates);
//   weights      e4m3 codes packed [G][Cn][9][Cg] + one E8M0 byte per output channel (the fp8w quantiser's power-of-two row scale).
// The instruction multiplies fp8 x fp8 exactly, applies both scales to each 32-deep partial sum and accumulates in fp32; it retires
// 4x the K of v_mfma_f32_16x16x32_bf16 in 2x its cycles.  dgrad / wgrad stay on the bf16 kernels (straight-through).
//
// Operand layout (tools/probe/mfma_fp8_layout.cpp, profiles/r04_mfma_fp8_layout.txt - found with exact integer data, it is NOT "32
// consecutive k per lane"): lane l, byte j of the 32-byte operand holds K = 64 * (j >> 4) + 16 * (l >> 4) + (j & 15) of row / column
// l & 15; the scale byte of lane l scales the K block [32 * (l >> 4), +32).  So the FIRST 16 bytes of all lanes are K 0..63 and the
// SECOND 16 bytes K 64..127: a K = 128 step here is two "half-steps" of 64 channels (one filter tap of a 64-channel slab each), every
// lane reads its 16-byte piece l >> 4 of half-step A, then of half-step B - the bf16 kernel's fragment read, twice.  Scale lanes: group
// 0 = (A, channels 0..31), 1 = (A, 32..63), 2 = (B, 0..31), 3 = (B, 32..63).
//
// Structure: conv3x3_wide3.hip's (persistent workgroups, resident halo in LDS, weights streamed through a ring by LDS-DMA, the two
// waves of a SIMD half a phase apart, counted vmcnt waits across raw barriers) with the SAME 64-byte-per-pixel LDS image - a slab is
// 64 fp8 channels instead of 32 bf16 ones - so the bank-conflict-free swizzles carry over: the first read of a fragment takes pieces
// conflict-free swizzles carry over unchanged.
// Tile: 4 images x (8 x 16) pixels x 128 output channels.  K loop per slab: taps paired (0,1) (2,3) (4,5) (6,7) (8,-): five stages of 32
// MFMAs (2 phases x 4 channel tiles x 4 pixel rows); the missing half of the fifth stage is zero registers (10 % of the matrix slots
// idle; pairing tap 8 with the next slab's tap 0 needs a third halo buffer: not in 160 KB at this tile).
#include <type_traits>

#include "common.h"
#include "fp8_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

struct F8P {
  const unsigned char* x;    // e4m3 activations, NHWC bytes
  const unsigned char* xs;   // E8M0 block scales [pixel][CS]
  const unsigned char* w;    // packed e4m3 weights [G][Cn][9][Cg]
  const unsigned char* ws;   // E8M0 per output channel [G * Cn]
  bf16_t* y;
  float* part;               // optional BN partials [B * nty * ntx][G * Cn][2]
  const float* scale;
  const float* shift;
  int act;
  long xsb, xsh, xsw, ysw;   // x strides in bytes (= elements), y pixel stride in elements
  int CS;                    // scale bytes per pixel (all channels of the tensor / 32)
  int B, H, W;
  int Cg, Cn, G;
  int Ktot;
  int ntx, nty, ntc, nbt;
  unsigned xbytes, wbytes, sbytes;
};

template <int N> __device__ __forceinline__ void f8_wvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void f8_lgk0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint4 f8_ld(const char* p) { return *(const uint4*)p; }

// acc += A x B with the A scale taken from byte CT of `SA`, the B scale from byte 0 of `SB`.  Inline asm with the accumulator TIED
// ("+v"): through the builtin hipcc (ROCm 7.2) gave most of these MFMAs a destination different from their C operand and spilled 900
// dwords per lane around them.  Hazards the assembler does not pad inside asm: operands come from LDS behind s_waitcnt lgkmcnt(0) + a
// barrier; the epilogue reads the accumulators behind f8_mfma_settle() below.
#define F8_MFMA_ASM(ACC, FA, FB, SA, SB, LO, HI)                                                                                          \
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[" #LO ",0,0] op_sel_hi:[" #HI ",0,0]"               \
               : "+v"(ACC) : "v"(FA), "v"(FB), "v"(SA), "v"(SB))
#define F8_MFMA0(ACC, FA, FB, SA, SB) F8_MFMA_ASM(ACC, FA, FB, SA, SB, 0, 0)
#define F8_MFMA1(ACC, FA, FB, SA, SB) F8_MFMA_ASM(ACC, FA, FB, SA, SB, 1, 0)
#define F8_MFMA2(ACC, FA, FB, SA, SB) F8_MFMA_ASM(ACC, FA, FB, SA, SB, 0, 1)
#define F8_MFMA3(ACC, FA, FB, SA, SB) F8_MFMA_ASM(ACC, FA, FB, SA, SB, 1, 1)
__device__ __forceinline__ void f8_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }  // XDL write -> VALU read of the accumulators

template <int EPI, bool HROLE>
__device__ __forceinline__ void f8_body(const F8P& p) {
  constexpr int TH = 8, NB = 4, HWD = 18;
  constexpr int NPIX = (TH + 2) * HWD;     // 180 halo pixels per image
  constexpr int IPI = (NPIX + 15) / 16;    // 12 data instructions per image (16 pixels x 64 B; the last covers 4 pixels)
  constexpr int NI = NB * IPI;             // 48 per slab = 12 rounds of the 4 halo waves
  constexpr int HRD = NI / 4;              // data rounds
  constexpr int SPI = (NPIX + 63) / 64;    // 3 scale instructions per image (64 pixels x 4 B: the aligned dword that holds the slab's two E8M0 bytes)
  constexpr int HR = HRD + NB * SPI / 4;   // 15 rounds per slab, all full
  constexpr int HBYTES = NB * NPIX * 64;   // 46 080
  constexpr int SBYTES = 3072;             // NB * NPIX * 4 = 2 880, padded
  constexpr int RD = 6, WB = 8192;         // weight ring: six half-step tiles (128 rows x 64 B)
  constexpr int NST = 16;
  static_assert(NI % 4 == 0 && (NB * SPI) % 4 == 0 && HR <= 15, "round plan");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;                             // [2][HBYTES]
  char* sS = smem + 2 * HBYTES;                // [2][SBYTES]
  char* sW = sS + 2 * SBYTES;                  // [RD][WB]
  float* red = (float*)(sW + RD * WB);         // [4][128][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wp = wave >> 1;
  const int wl = wave & 3;
  const int wimg = wp, wrow0 = 0;
  const int nslab = p.Cg >> 6;                 // >= 2, Cg % 64 == 0 (launcher)

  const int ntiles = p.G * p.nbt * p.nty * p.ntx * p.ntc;
  int tile, tile_end, tile_step;
  {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tile_step = gridDim.x >> 3;
    const int lo = (int)((long)ntiles * xcd / 8);
    tile_end = (int)((long)ntiles * (xcd + 1) / 8);
    tile = lo + slot;
  }
  if (tile >= tile_end) return;

  struct TileC { int g, b0, y0, x0, c0, ty, tx, live; };
  auto decode = [&](int t, bool live) {
    TileC c;
    int tc = t % p.ntc; t /= p.ntc;
    c.tx = t % p.ntx; t /= p.ntx;
    c.ty = t % p.nty; t /= p.nty;
    int bt = t % p.nbt; c.g = t / p.nbt;
    c.b0 = bt * NB; c.x0 = c.tx * 16; c.y0 = c.ty * TH; c.c0 = tc * 128; c.live = live;
    return c;
  };

  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xs, 0, (int)p.sbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  // ---- halo DMA (waves 0..3): round rd < HRD: data instruction i = rd * 4 + wl = 16 consecutive halo pixels of one image, 64 B each as
  // four 16-byte pieces, piece ^= 2 * bit2(hx) (conv3x3_wide3.hip); round rd >= HRD: scale instruction = 64 consecutive halo pixels of one
  // image, one DWORD each - the 4-byte aligned word around the slab's two E8M0 bytes (a 2-byte LDS-DMA still advances the LDS address by
  // four bytes per lane: the first form, which assumed a packed 2-byte image, read every other pixel's scale as 2^-127)
  const int xsw1 = (int)p.xsw, xsh1 = (int)p.xsh, xsb1 = (int)p.xsb;
  const int lj0 = lane >> 2, ls16 = (lane & 3) << 4;
  auto issue_h = [&](const TileC& c, int slab, int bufsel, int rd) {
    if (rd < HRD) {
      const int i = rd * 4 + wl;
      const int img = (i >= IPI) + (i >= 2 * IPI) + (i >= 3 * IPI);
      const int q = i - img * IPI;
      const int hy0 = (q * 16 * 3641) >> 16;
      const int hx0 = q * 16 - hy0 * HWD;
      const int bb = c.b0 + img, yb = c.y0 + hy0 - 1;
      const bool y0ok = (unsigned)yb < (unsigned)p.H, y1ok = (unsigned)(yb + 1) < (unsigned)p.H;
      const bool iok = (c.live != 0) & (bb < p.B);
      const int sbase = bb * xsb1 + yb * xsh1 + (c.x0 - 1) * xsw1 + c.g * p.Cg + slab * 64;
      int lj = lj0;
      asm volatile("" : "+v"(lj));
      const int t = lj + hx0;
      const bool wrap = t >= HWD;
      const int hx = t - (wrap ? HWD : 0);
      const bool xok = (unsigned)(c.x0 - 1 + hx) < (unsigned)p.W;
      const int pc = ls16 ^ ((hx & 4) << 3);
      const bool ok = iok & xok & (wrap ? y1ok : y0ok);
      const unsigned off = (unsigned)(sbase + __mul24(hx, xsw1) + (wrap ? xsh1 : 0) + pc);
      char* dst = sH + bufsel * HBYTES + (img * NPIX + q * 16) * 64;
      if (q == IPI - 1) {
        if (lj < NPIX - (IPI - 1) * 16)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : OOB, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : OOB, 0, 0, 0);
      }
    } else {
      const int si = (rd - HRD) * 4 + wl;
      const int img = (si >= SPI) + (si >= 2 * SPI) + (si >= 3 * SPI);
      const int part = si - img * SPI;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int pi = part * 64 + ln;
      const int hy = (pi * 3641) >> 16, hx = pi - hy * HWD;
      const int bb = c.b0 + img, yy = c.y0 + hy - 1, xx = c.x0 - 1 + hx;
      const bool ok = (c.live != 0) & (bb < p.B) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
      const unsigned off = (unsigned)((((bb * p.H + yy) * p.W + xx) * p.CS) + ((c.g * (p.Cg >> 5) + slab * 2) & ~3));
      char* dst = sS + bufsel * SBYTES + (img * NPIX + part * 64) * 4;
      if (part == SPI - 1) {
        if (pi < NPIX) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 4, ok ? off : OOB, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 4, ok ? off : OOB, 0, 0, 0);
      }
    }
  };
  // ---- weight DMA (waves 4..7): two instructions per half-step tile (128 rows x 64 B); rows swizzled by piece ^= 2 * bit4(n)
  const int ltid = tid & 255;
  int wrel0, wrel1;
  {
    const int n0 = ltid >> 2, s = ltid & 3;
    const int wch = (s ^ (((n0 >> 4) & 1) << 1)) << 4;
    wrel0 = n0 * p.Ktot + wch;
    wrel1 = (n0 + 64) * p.Ktot + wch;
  }
  auto issue_w = [&](const TileC& c, int slab, int tap, int slot, int rd) {
    const unsigned base = (unsigned)((c.g * p.Cn + c.c0) * p.Ktot + tap * p.Cg + slab * 64);
    const int n = rd * 64 + (ltid >> 2);
    const bool ok = (c.live != 0) & (c.c0 + n < p.Cn);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sW + slot * WB + (rd * 256 + wl * 64) * 16), 16,
                                             ok ? base + (unsigned)(rd ? wrel1 : wrel0) : OOB, 0, 0, 0);
  };

  // ---- fragment addressing (every lane: piece kb of half-step A, then piece kb of half-step B) -------------------------------------
  const int lp = lane & 15, kb = lane >> 4;
  const bool scB = lane >= 32;                  // this lane's SCALE byte belongs to half-step B; its 32-channel block is kb & 1
  const int arow = wc * 64 + (lp >> 2) * 8 + (lp & 3);
  const int ao = arow * 64 + ((kb ^ (((arow >> 4) & 1) << 1)) << 4);
  const int pixl = wimg * NPIX + wrow0 * HWD + lp;  // + r * 18 + q: halo pixel of pixel row 0
  int bo[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) bo[q] = ((pixl + q) << 6) + ((kb ^ ((((q + lp) >> 2) & 1) << 1)) << 4);
  const int so0 = (pixl << 2) + (kb & 1);
  i32x8_t fa[4], fb[4];
  int sb[4];

  // A-operand scales: byte ct of `sa` = E8M0 of weight row arow + (ct >> 1) * 32 + (ct & 1) * 4 of the tile (OPSEL = ct)
  auto load_sa = [&](const TileC& c) {
    unsigned v = 0x7f7f7f7fu;
    if (c.live) {
      v = 0;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int row = c.c0 + arow + (ct >> 1) * 32 + (ct & 1) * 4;
        const unsigned b = row < p.Cn ? (unsigned)p.ws[c.g * p.Cn + row] : 0x7fu;
        v |= b << (8 * ct);
      }
    }
    return (int)v;
  };

  TileC cur = decode(tile, true);
  TileC nx = decode(tile + tile_step < tile_end ? tile + tile_step : tile, tile + tile_step < tile_end);
  int sa = load_sa(cur);
  if (HROLE) {
#pragma unroll
    for (int rd = 0; rd < HR; ++rd) issue_h(cur, 0, 0, rd);
    f8_wvm<0>();
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) { issue_w(cur, 0, t, t, 0); issue_w(cur, 0, t, t, 1); }
    f8_wvm<4>();  // taps 0, 1
  }
  __builtin_amdgcn_s_barrier();

  int gs = 0;        // slabs retired: half-step (slab, tap) sits in ring slot (9 gs + tap) % 6 = (3 (gs & 1) + tap) % 6
  int hsel = 0;      // halo / scale buffer of the current slab
  int tolerate = 0;
#pragma unroll 1
  for (; tile < tile_end; tile += tile_step) {
    f32x4_t acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nvr = min(8, max(0, p.H - (cur.y0 + wrow0)));
    const int st_issued = cur.b0 + wimg < p.B ? (cur.c0 + wc * 64 + 32 < p.Cn ? 2 * nvr : cur.c0 + wc * 64 < p.Cn ? nvr : 0) : 0;
    const int st_wave = st_issued >= NST ? NST : (st_issued >= NST / 2 ? NST / 2 : 0);

    if (!HROLE) __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int k = 0; k < nslab; ++k, ++gs) {
      const bool last = k == nslab - 1;
      const int odd = gs & 1;
      const int hb = hsel * HBYTES, sbo = hsel * SBYTES;
      const int ssub = (cur.g * (p.Cg >> 5) + k * 2) & 2;  // where the slab's two scale bytes sit inside the aligned dword
      const int htgt = hsel ^ 1;
      hsel ^= 1;
      const bool hin = k + 1 < nslab;
      const int hs = hin ? k + 1 : 0;
      const TileC htile = hin ? cur : nx;
      const TileC wtile = last ? nx : cur;
      const int wslab = last ? 0 : k + 1;
      auto stage = [&](auto JC) {
        constexpr int j = decltype(JC)::value;
        constexpr int tA = 2 * j, tB = 2 * j + 1;       // tB == 9: no second half-step (zeros)
        constexpr int rA = tA / 3, qA = tA % 3, rB = (tB < 9 ? tB : tA) / 3, qB = (tB < 9 ? tB : tA) % 3;
        const int slA = odd ? (tA + 3) % 6 : tA % 6;
        const int slB = odd ? (tB + 3) % 6 : tB % 6;
        const int so = sbo + ssub + so0 + (scB ? (rB * HWD + qB) * 4 : (rA * HWD + qA) * 4);
        auto load_a = [&]() {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const uint4 lo = f8_ld(sW + slA * WB + ao + (ct >> 1) * 2048 + (ct & 1) * 256);
            uint4 hi = make_uint4(0, 0, 0, 0);
            if constexpr (tB < 9) hi = f8_ld(sW + slB * WB + ao + (ct >> 1) * 2048 + (ct & 1) * 256);
            fa[ct] = (i32x8_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
          }
        };
        auto load_b = [&](int half) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint4 lo = f8_ld(sH + hb + bo[qA] + (half * 4 + i + rA) * (HWD * 64));
            uint4 hi = make_uint4(0, 0, 0, 0);
            if constexpr (tB < 9) hi = f8_ld(sH + hb + bo[qB] + (half * 4 + i + rB) * (HWD * 64));
            fb[i] = (i32x8_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            sb[i] = *(const unsigned char*)(sS + so + (half * 4 + i) * (HWD * 4));
          }
        };
        // weights to issue during this stage: half-steps 2j + 4, 2j + 5 of the linear (slab, tap) sequence (j == 4: one)
        constexpr int iA = 2 * j + 4, iB = 2 * j + 5;
        auto issue_half = [&](int idx, int rd) {
          const int slot = odd ? (idx + 3) % 6 : idx % 6;
          if (idx < 9) issue_w(cur, k, idx, slot, rd); else issue_w(wtile, wslab, idx - 9, slot, rd);
        };
        // ================= phase 0: pixel rows 0..3 =================
        load_a();
        load_b(0);
        __builtin_amdgcn_sched_barrier(0);
        if (HROLE) {  // 15 rounds in the first eight phases of the slab; the fifth stage issues nothing and waits
          if (4 * j < HR) issue_h(htile, hs, htgt, 4 * j);
          if (4 * j + 1 < HR) issue_h(htile, hs, htgt, 4 * j + 1);
        } else {
          issue_half(iA, 0);
          issue_half(iA, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        f8_lgk0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          F8_MFMA0(acc[0][i], fa[0], fb[i], sa, sb[i]);
          F8_MFMA1(acc[1][i], fa[1], fb[i], sa, sb[i]);
          F8_MFMA2(acc[2][i], fa[2], fb[i], sa, sb[i]);
          F8_MFMA3(acc[3][i], fa[3], fb[i], sa, sb[i]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 1: pixel rows 4..7 =================
        load_b(1);
        __builtin_amdgcn_sched_barrier(0);
        if (HROLE) {
          if (4 * j + 2 < HR) issue_h(htile, hs, htgt, 4 * j + 2);
          if (4 * j + 3 < HR) issue_h(htile, hs, htgt, 4 * j + 3);
          if (j == 4) f8_wvm<0>();  // the next slab's halo + scales are read from the next stage on
        } else {
          if (j < 4) { issue_half(iB, 0); issue_half(iB, 1); }
          // the next stage's half-steps must have landed; what this stage issued (j == 3: and tap 0 of the next slab, which the one-tap
          // stage 4 does not read) stays in flight.  In a tile's first stage the previous epilogue's stores sit in front of this stage's loads.
          if (k == 0 && j == 0 && tolerate == NST) f8_wvm<4 + NST>();
          else if (k == 0 && j == 0 && tolerate == NST / 2) f8_wvm<4 + NST / 2>();
          else if (j == 3) f8_wvm<6>();
          else f8_wvm<4>();
        }
        __builtin_amdgcn_sched_barrier(0);
        f8_lgk0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          F8_MFMA0(acc[0][4 + i], fa[0], fb[i], sa, sb[i]);
          F8_MFMA1(acc[1][4 + i], fa[1], fb[i], sa, sb[i]);
          F8_MFMA2(acc[2][4 + i], fa[2], fb[i], sa, sb[i]);
          F8_MFMA3(acc[3][4 + i], fa[3], fb[i], sa, sb[i]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      };
      stage(std::integral_constant<int, 0>{});
      stage(std::integral_constant<int, 1>{});
      stage(std::integral_constant<int, 2>{});
      stage(std::integral_constant<int, 3>{});
      stage(std::integral_constant<int, 4>{});
    }
    if (HROLE) __builtin_amdgcn_s_barrier();
    f8_mfma_settle();

    // ---- epilogue (conv3x3_wide3.hip's: the C/D layout of the 16x16 forms does not depend on the operand type) ----------------------
    int el = ltid;
    asm volatile("" : "+v"(el));
    const int e_lp = el & 15, e_lq = (el >> 4) & 3;
    const int e_tid = (HROLE ? 0 : 256) + el;
    const int bb = cur.b0 + wimg;
    const bool xok = cur.x0 + e_lp < p.W;
    const int cl0 = wc * 64 + e_lq * 8;
    // the next tile's weight-row scales: ordinary loads, placed HERE - the vmcnt(0) the compiler puts in front of their use drains only
    // the prefetch of the next tile's first stage, which must land before that stage anyway - not after the stores below
    const int sa_next = load_sa(nx);
    bool cok[2];
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    f2_t s2[2][4], q2[2][4];
    float sv[2][8], hv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cok[h] = cur.c0 + cl0 + h * 32 < p.Cn && bb < p.B;
#pragma unroll
      for (int i = 0; i < 4; ++i) { s2[h][i] = (f2_t){0.f, 0.f}; q2[h][i] = (f2_t){0.f, 0.f}; }
      if (EPI == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sv[h][i] = cok[h] ? p.scale[cur.g * p.Cn + cur.c0 + cl0 + h * 32 + i] : 1.f;
          hv[h][i] = cok[h] ? p.shift[cur.g * p.Cn + cur.c0 + cl0 + h * 32 + i] : 0.f;
        }
      }
    }
    if (EPI == 0) {
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) {
        const int yy = cur.y0 + wrow0 + pt;
        const bool pok = xok & (yy < p.H);
        const unsigned keep = pok ? 0xffffffffu : 0u;
        bf16_t* dst = p.y + (((long)bb * p.H + yy) * p.W + cur.x0 + e_lp) * p.ysw + (long)cur.g * p.Cn + cur.c0 + cl0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned pk[4];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const f2_t a = {acc[2 * h + c2][pt][2 * jj], acc[2 * h + c2][pt][2 * jj + 1]};
              const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(a, b2_t)) & keep;
              pk[c2 * 2 + jj] = u;
              const f2_t r = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
              s2[h][c2 * 2 + jj] += r;
              q2[h][c2 * 2 + jj] = __builtin_elementwise_fma(r, r, q2[h][c2 * 2 + jj]);
            }
          if (pok && cok[h]) *(uint4*)(dst + h * 32) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
      }
    } else {
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) {
        const int yy = cur.y0 + wrow0 + pt;
        const bool pok = xok & (yy < p.H);
        bf16_t* dst = p.y + (((long)bb * p.H + yy) * p.W + cur.x0 + e_lp) * p.ysw + (long)cur.g * p.Cn + cur.c0 + cl0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              float u = acc[2 * h + c2][pt][jj];
              u = u * sv[h][c2 * 4 + jj] + hv[h][c2 * 4 + jj];
              if (p.act) u = silu_f(u);
              v[c2 * 4 + jj] = u;
            }
          if (pok && cok[h]) *(uint4*)(dst + h * 32) = Chunk<bf16_t>::pack(v);
        }
      }
    }
    if (EPI == 0 && p.part) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float s = wave_xor_sum16(s2[h][i >> 1][i & 1]);
          float qq = wave_xor_sum16(q2[h][i >> 1][i & 1]);
          if (e_lp == i) {
            red[(wp * 128 + cl0 + h * 32 + i) * 2 + 0] = s;
            red[(wp * 128 + cl0 + h * 32 + i) * 2 + 1] = qq;
          }
        }
      f8_lgk0();
      __builtin_amdgcn_s_barrier();
      if (e_tid < 128 * NB) {
        const int img = e_tid >> 7, ch = e_tid & 127;
        if (cur.c0 + ch < p.Cn && cur.b0 + img < p.B) {
          const float s = red[(img * 128 + ch) * 2], q = red[(img * 128 + ch) * 2 + 1];
          const long row = ((long)(cur.b0 + img) * p.nty + cur.ty) * p.ntx + cur.tx;
          float* dst = p.part + (row * (p.G * p.Cn) + cur.g * p.Cn + cur.c0 + ch) * 2;
          *(float2*)dst = make_float2(s, q);
        }
      }
    }
    tolerate = st_wave;
    sa = sa_next;
    cur = nx;
    {
      const int t2 = tile + 2 * tile_step;
      nx = decode(t2 < tile_end ? t2 : tile, t2 < tile_end);
    }
  }
  f8_wvm<0>();
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void conv3x3_fp8_kernel(F8P p) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) f8_body<EPI, true>(p);
  else f8_body<EPI, false>(p);
}

int f8_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int EPI>
int launch_f8(const F8P& p, hipStream_t st) {
  const size_t sm = 2 * 46080 + 2 * 3072 + 6 * 8192 + 4 * 128 * 2 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_fp8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  long ntiles = (long)p.G * p.nbt * p.nty * p.ntx * p.ntc;
  long nwg = (long)f8_cu_count() / 8 * 8;
  if (nwg < 8) nwg = 8;
  while (nwg > 8 && nwg / 8 > (ntiles + 7) / 8) nwg -= 8;
  hipLaunchKernelGGL((conv3x3_fp8_kernel<EPI>), dim3((unsigned)nwg), dim3(512), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

// ---- activation quantiser: bf16 NHWC (pixel-dense, pixel stride xsw elements) -> e4m3 codes [M][C] + E8M0 block scales [M][C / 32] -------
// One thread per 8 channels (16 bytes in, 8 out); the four threads of a 32-channel block fold their maxima with two quad-permute DPP ops.
__global__ __launch_bounds__(256) void fp8_act_quant_kernel(const bf16_t* __restrict__ x, long xsw, unsigned char* __restrict__ q, unsigned char* __restrict__ s,
                                                            long M, int C) {
  const int cpr = C >> 3, SP = fp8_scale_pitch(C);
  const long total = M * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ((total + 255) & ~255L); i += (long)gridDim.x * 256) {
    const bool live = i < total;
    const long pix = live ? i / cpr : 0;
    const int ch = live ? (int)(i - pix * cpr) : 0;
    float f[8];
    uint4 u = make_uint4(0, 0, 0, 0);
    if (live) u = *(const uint4*)(x + pix * xsw + ch * 8);
    Chunk<bf16_t>::unpack(u, f);
    unsigned lo, hi;
    const int sbyte = mx_quantize8(f, lo, hi);
    if (live) {
      *(uint2*)(q + pix * C + ch * 8) = make_uint2(lo, hi);
      if ((ch & 3) == 0) s[pix * SP + (ch >> 2)] = (unsigned char)sbyte;
    }
  }
}

// ---- weight packer: fp8w codes (rows, Cg * 9) in OIHW order (k = ci * 9 + tap) -> [row][tap][ci] bytes; row scales fp32 (powers of two) -> E8M0
__global__ __launch_bounds__(256) void fp8_pack_w_kernel(const unsigned char* __restrict__ codes, const float* __restrict__ scale, unsigned char* __restrict__ wq,
                                                         unsigned char* __restrict__ ws, int rows, int Cg) {
  const long total = (long)rows * 9 * Cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % Cg);
    const long r = i / Cg;
    const int tap = (int)(r % 9);
    const long row = r / 9;
    wq[i] = codes[(row * Cg + ci) * 9 + tap];
    if (ci == 0 && tap == 0) ws[row] = (unsigned char)((__float_as_uint(scale[row]) >> 23) & 255);
  }
}

}  // namespace

extern "C" {

int y3d_fp8_quantize_act(const void* x, int64_t xsw, int64_t M, int C, uint8_t* q, uint8_t* s, void* stream) {
  Y3D_CHECK(x && q && s && M >= 1 && C >= 32 && C % 32 == 0 && xsw >= C && xsw % 8 == 0, "fp8_quantize_act: C = %d must be a multiple of 32, rows 16-byte aligned", C);
  const long total = M * (C >> 3);
  long nb = (total + 255) / 256;
  hipLaunchKernelGGL(fp8_act_quant_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)xsw, q, s, (long)M, C);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_fp8_pack_weight_fwd(const uint8_t* codes, const float* scale, int rows, int Cg, uint8_t* wq, uint8_t* ws, void* stream) {
  Y3D_CHECK(codes && scale && wq && ws && rows >= 1 && Cg >= 1, "fp8_pack_weight_fwd: bad arguments");
  const long total = (long)rows * 9 * Cg;
  long nb = (total + 255) / 256;
  hipLaunchKernelGGL(fp8_pack_w_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, codes, scale, wq, ws, rows, Cg);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_conv3x3_fp8_ok(int B, int H, int W, int Cin, int Cout, int groups) {
  if (groups < 1 || Cin % groups || Cout % groups) return 0;
  const int Cg = Cin / groups, Cn = Cout / groups;
  return Cg % 64 == 0 && Cg >= 128 && Cn % 16 == 0 && W >= 8 && H >= 4 && B >= 1;
}

int y3d_conv3x3_fp8_stat_rows(int B, int H, int W) { return B * cdiv(H, 8) * cdiv(W, 16); }

int y3d_fp8_scale_pitch(int C) { return fp8_scale_pitch(C); }

int y3d_conv3x3_fp8_fwd(const uint8_t* xq, const uint8_t* xs, int B, int H, int W, int Cin, const uint8_t* wq, const uint8_t* ws, void* y, int64_t ysw,
                        int Cout, int groups, float* stat_partials, const float* scale, const float* shift, int act, void* stream) {
  Y3D_CHECK(xq && xs && wq && ws && y, "conv3x3_fp8_fwd: null argument");
  Y3D_CHECK(y3d_conv3x3_fp8_ok(B, H, W, Cin, Cout, groups), "conv3x3_fp8_fwd: geometry B=%d H=%d W=%d Cin=%d Cout=%d groups=%d is not served (Cin / groups a multiple of 64, >= 128)", B, H, W, Cin, Cout, groups);
  Y3D_CHECK((scale == nullptr) == (shift == nullptr) && !(scale && stat_partials), "conv3x3_fp8_fwd: either BatchNorm partials (training) or the affine epilogue (eval)");
  F8P p;
  p.x = xq; p.xs = xs; p.w = wq; p.ws = ws; p.y = (bf16_t*)y; p.part = stat_partials; p.scale = scale; p.shift = shift; p.act = act;
  p.xsw = Cin; p.xsh = (long)W * Cin; p.xsb = (long)H * W * Cin; p.ysw = ysw; p.CS = fp8_scale_pitch(Cin);
  p.B = B; p.H = H; p.W = W; p.G = groups; p.Cg = Cin / groups; p.Cn = Cout / groups; p.Ktot = 9 * p.Cg;
  p.ntx = cdiv(W, 16); p.nty = cdiv(H, 8); p.ntc = cdiv(p.Cn, 128); p.nbt = cdiv(B, 4);
  const unsigned long xb = (unsigned long)B * H * W * Cin, wb = (unsigned long)Cout * p.Ktot, sb = (unsigned long)B * H * W * p.CS;
  Y3D_CHECK(xb < 0xfffffff0ul && wb < 0xfffffff0ul, "conv3x3_fp8_fwd: operand larger than 4 GB");
  Y3D_CHECK(Cin < (1 << 22), "conv3x3_fp8_fwd: pixel stride beyond the 24-bit address multiply");
  p.xbytes = (unsigned)xb; p.wbytes = (unsigned)wb; p.sbytes = (unsigned)sb;
  hipStream_t st = (hipStream_t)stream;
  return scale ? launch_f8<1>(p, st) : launch_f8<0>(p, st);
}

}  // extern "C"
