// bf16 3x3 stride-1 "same" convolution, resident-halo implicit GEMM, PERSISTENT workgroups - the kernel that carries the YOLOv10-3D
// head (83 % of S-3D forward FLOPs are 3x3 convs at 128 channels per group) and, with the taps flipped, its data gradient.
//
// Tile: NB images x (TH x 16) pixels x 128 output channels of one group, NB * TH = 32 (TH = 16: two images, TH = 8: four) - 512 pixels
// share every weight byte brought on chip, and batching over IMAGES instead of widening the spatial tile keeps the tiling exact for
// 80x80 / 40x40 maps.  8 waves = 2 channel halves x 4 pixel groups; a wave owns 64 channels x 128 pixels (8 rows of one image): 128
// accumulator registers.  K runs over (32-channel slab, filter tap): one stage = one tap of one slab = one 16x16x32 MFMA k-step.
//   * the (TH+2) x 18 halo of the NB images of a slab (64-byte rows, pieces swizzled by the pixel's column) is brought into LDS once and
//     serves all nine taps; three halo buffers at TH = 16: the halo of slab s+2 streams in while slab s is computed (HBM latency under
//     load is ~2 us, one slab is ~8 us of MFMA work);
//   * the 128 x 32 weight tile of a tap (8 KB, L2-resident) streams through a 4-slot ring, three taps in flight;
//   * both by LDS-DMA (buffer_load ... lds, 16 B per lane: no VGPR staging, no ds_write) with the bank swizzle applied on the per-lane
//     SOURCE offset (the DMA destination is lane-linear); zero padding = the buffer descriptor's range check;
//   * the nine stages of a slab are unrolled: LDS offsets are immediates, wait counts compile-time constants; a stage ends with a
//     COUNTED s_waitcnt vmcnt(N) + raw s_barrier, the younger loads stay in flight across it;
//   * the two streams have different latencies and vmcnt retires in order, so they are issued by DIFFERENT waves: waves 0..3 (one per
//     SIMD) stream halos, waves 4..7 weights, each group with its own wait counts.
// The workgroup is persistent (one per CU): while the last slabs of a tile are computed the first halos and taps of the NEXT tile
// stream in, and the epilogue's stores drain behind the next tile's MFMAs (the first wait of a tile tolerates them in the vmcnt FIFO).
// XCD x walks a contiguous run of tiles, its workgroups interleaved, so the tiles in flight share weights and halos in that XCD's L2.
// Output channels are permuted inside the MFMA row index (row r of channel tile ct is channel (ct>>1)*32 + (r>>2)*8 + (ct&1)*4 + (r&3)),
// so a lane ends with two runs of 8 consecutive channels and one store instruction writes 64 contiguous bytes per pixel (with 16-byte
// pieces at a 32-byte stride the L2 wrote 1.8x the tensor: PMC WRITE_SIZE).  The epilogue also emits the per-tile BatchNorm partial sums
// (sum, sum of squares of the ROUNDED outputs) or applies the eval-mode affine + SiLU.
// Register budget: the build must stay spill-free - a scratch reload in the loop makes the compiler wait vmcnt(0) and drains the DMA
// pipeline (seen: -25 %); DMA addresses are recomputed at the issue point from tile scalars (an opaque asm keeps LICM from hoisting them).
//
// Round 3 - the two waves of a SIMD HALF A PHASE APART ("ping-pong").  Round 2 measured: with both waves of a SIMD on the same schedule
// (DMA issue, fragment reads, 32 MFMAs, wait, barrier) the matrix pipe idles while both read / issue / wait and both compete for it
// while both compute: a stage took 1 430-1 630 cycles for 1 024 cycles of matrix work (738 TFLOP/s for the first resident-halo kernel ->
// 870 unrolled taps + counted waits -> 1 069 in the training step), and no re-placement inside that structure got past 0.50 of the peak.
// Here a stage is two phases, each  L: [fragment reads + one DMA instruction + waits] | barrier | M: [16 MFMAs] | barrier,  and waves
// 4..7 execute ONE barrier more than waves 0..3 before the K loop of a tile (waves 0..3 one more after it): whenever waves 0..3 are in
// an M segment their SIMD partners are in an L segment and vice versa.  The fragments of a phase are read right before its MFMAs (no
// double buffer: 16 fewer registers), behind the partner's matrix work.  The halo address arithmetic had to shrink for that (an L
// segment lasts 16 MFMAs = 256 cycles): a halo DMA instruction covers 16 CONSECUTIVE halo pixels of one image, so image, halo row and
// first column are wave-uniform (SALU) and a lane adds at most one row wrap: ~14 VALU instructions per instruction instead of ~35.
// (The round-2 kernel, conv3x3_wide.hip, was retired in round 4; tools/probe keeps the probe copy of this one.)
#include "common.h"

namespace {


struct W3P {
  const bf16_t* x;
  const bf16_t* w;  // packed [G][Cn][9][Cg] (forward) or the dgrad packing; row pitch Ktot
  bf16_t* y;
  float* part;      // optional BN partials [B*nty*ntx][G*Cn][2]
  const float* scale;
  const float* shift;
  int act;
  long xsb, xsh, xsw, ysw;
  int B, H, W;
  int Cg, Cn, G;
  int Ktot;
  int ntx, nty, ntc, nbt;
  int flip;
  unsigned xbytes, wbytes;  // buffer extents for the hardware range check
};

template <int N> __device__ __forceinline__ void wvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void wvm_n(int n) {  // count known after unrolling: the switch folds to the one s_waitcnt
  switch (n) {
    case 0: wvm<0>(); break; case 1: wvm<1>(); break; case 2: wvm<2>(); break; case 3: wvm<3>(); break;
    case 4: wvm<4>(); break; case 5: wvm<5>(); break; case 6: wvm<6>(); break; case 7: wvm<7>(); break;
    case 8: wvm<8>(); break; case 9: wvm<9>(); break; case 10: wvm<10>(); break; case 11: wvm<11>(); break;
    case 12: wvm<12>(); break; case 13: wvm<13>(); break; case 14: wvm<14>(); break; case 15: wvm<15>(); break;
    case 16: wvm<16>(); break; case 17: wvm<17>(); break; case 18: wvm<18>(); break; case 19: wvm<19>(); break;
    case 20: wvm<20>(); break; case 21: wvm<21>(); break; case 22: wvm<22>(); break; case 23: wvm<23>(); break;
    case 24: wvm<24>(); break; case 25: wvm<25>(); break; case 26: wvm<26>(); break; case 27: wvm<27>(); break;
    case 28: wvm<28>(); break;
    default: wvm<0>(); break;
  }
}

__device__ __forceinline__ bf16x8_t ldf(const char* p) { return __builtin_bit_cast(bf16x8_t, *(const uint4*)p); }
__device__ __forceinline__ void lgk0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// HROLE: this wave streams the halo (waves 0..3, the leading group), else the weights (waves 4..7, one barrier behind)
template <int TH, int EPI, bool HROLE>
__device__ __forceinline__ void wide3_body(const W3P& p) {
  constexpr int NB = 32 / TH;              // images per tile
  constexpr int WPI = TH / 8;              // pixel-group waves per image
  constexpr int HWD = 18;
  constexpr int NPIX = (TH + 2) * HWD;     // halo pixels per image
  constexpr int IPI = (NPIX + 15) / 16;    // halo DMA instructions per image (16 pixels x 64 B each; the last one is partial)
  constexpr int NI = NB * IPI;             // ... per slab
  constexpr int HR = (NI + 3) / 4;         // rounds: instruction rd * 4 + w goes to halo wave w
  constexpr int HFULL = NI / 4;            // rounds in which every halo wave issues (what the wait counts may rely on)
  constexpr int HBYTES = NB * NPIX * 64;
  constexpr int NHB = TH == 16 ? 3 : 2;    // halo buffers: the halo of slab s + NHB - 1 streams in while slab s is computed
  constexpr int HD = NHB - 1;
  constexpr int RD = 4, D = 3;             // weight ring slots, taps in flight
  constexpr int WB = 8192;                 // bytes of one weight tile (128 rows x 64 B)
  constexpr int NST = 16;                  // epilogue store instructions of a wave that owns valid channels
  static_assert(HR <= 14, "one halo round per phase, all issued by stage 6");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;                         // [NHB][HBYTES]
  char* sW = smem + NHB * HBYTES;          // [RD][WB]
  float* red = (float*)(sW + RD * WB);     // [4][128][2]

  // `wave` and everything derived from it is wave-uniform: scalar registers (the compiler cannot prove threadIdx.x >> 6 uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wp = wave >> 1;
  const int wl = wave & 3;                 // index inside the loader group
  const int wimg = wp / WPI, wrow0 = (wp % WPI) * 8;
  const int nslab = (p.Cg + 31) >> 5;      // >= 2 (launcher); Cg % 32 != 0 (80 channels of the X widths): the last slab's chunks past Cg are
                                           // zeros on both operands (out-of-range DMA lanes)

  // ---- persistent schedule ------------------------------------------------------------------------------------------------------
  const int ntiles = p.G * p.nbt * p.nty * p.ntx * p.ntc;
  int tile, tile_end, tile_step;
  {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tile_step = gridDim.x >> 3;
    const int lo = (int)((long)ntiles * xcd / 8);
    tile_end = (int)((long)ntiles * (xcd + 1) / 8);
    tile = lo + slot;
  }
  if (tile >= tile_end) return;  // uniform per workgroup
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
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  // ---- halo DMA: instruction i = rd * 4 + wl of a slab covers pixels q * 16 .. q * 16 + 15 (q = i % IPI) of image i / IPI, 64 B per
  // pixel as four 16-byte pieces (lane = 4 * pixel + piece).  Scalars: image, first halo row hy0 and column hx0; a lane's pixel is at
  // most one row further (hx0 + j >= 18).  Pieces are swizzled by their column: piece ^= 2 * bit2(hx) (conflict-free fragment reads).
  // Byte offsets are 32-bit and may wrap for out-of-image lanes, which take the out-of-range offset anyway.
  const int xsw2 = (int)p.xsw * 2, xsh2 = (int)p.xsh * 2, xsb2 = (int)p.xsb * 2;  // tensors are < 4 GB - 16 (launcher)
  const int lj0 = lane >> 2, ls16 = (lane & 3) << 4;  // two resident registers; everything else is recomputed at the issue point
  auto issue_h = [&](const TileC& c, int slab, int bufo, int rd) {
    const int i = rd * 4 + wl;
    if (rd >= HFULL && i >= NI) return;  // scalar: the last round is short by some waves
    const int img = NB == 2 ? (i >= IPI) : (i >= IPI) + (i >= 2 * IPI) + (i >= 3 * IPI);
    const int q = i - img * IPI;
    const int hy0 = (q * 16 * 3641) >> 16;  // q * 16 / 18, exact below 8192
    const int hx0 = q * 16 - hy0 * HWD;
    const int bb = c.b0 + img, yb = c.y0 + hy0 - 1;
    const bool y0ok = (unsigned)yb < (unsigned)p.H, y1ok = (unsigned)(yb + 1) < (unsigned)p.H;
    const bool iok = (c.live != 0) & (bb < p.B);
    const int sbase = bb * xsb2 + yb * xsh2 + (c.x0 - 1) * xsw2 + (c.g * p.Cg + slab * 32) * 2;
    int lj = lj0;
    asm volatile("" : "+v"(lj));  // opaque: the per-lane part below depends only on (round, lane), and LICM would keep all rounds' worth resident
    const int t = lj + hx0;
    const bool wrap = t >= HWD;
    const int hx = t - (wrap ? HWD : 0);
    const bool xok = (unsigned)(c.x0 - 1 + hx) < (unsigned)p.W;
    const int pc = ls16 ^ ((hx & 4) << 3);           // 16 x the piece this lane fetches = 2 x its first channel inside the slab
    const bool ok = iok & xok & (wrap ? y1ok : y0ok) & (pc < (p.Cg - slab * 32) * 2);
    const unsigned off = (unsigned)(sbase + __mul24(hx, xsw2) + (wrap ? xsh2 : 0) + pc);
    char* dst = sH + bufo + (img * NPIX + q * 16) * 64;
    if (q == IPI - 1 && NPIX % 16 != 0) {  // the partial instruction of an image: the lanes past its last pixel must not write LDS
      if (lj < NPIX - (IPI - 1) * 16)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : OOB, 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : OOB, 0, 0, 0);
    }
  };
  // ---- weight DMA: two instructions per tap (128 rows x 64 B); lane -> row n = rd * 64 + (ltid >> 2) of the tile, piece s; rows
  // swizzled by piece ^= 2 * bit4(n).  Per-lane byte offsets inside the tap computed once.
  const int ltid = tid & 255;
  int wrel0, wrel1, wch;  // wch: first channel (inside a slab) of the piece this lane fetches - the same for both rows (bit 4 of n and n + 64 agree)
  {
    const int n0 = ltid >> 2, s = ltid & 3;
    wch = (s ^ (((n0 >> 4) & 1) << 1)) << 3;
    wrel0 = (n0 * p.Ktot + wch) * 2;
    wrel1 = ((n0 + 64) * p.Ktot + wch) * 2;
  }
  auto issue_w = [&](const TileC& c, int slab, int tap, int slot, int rd) {
    const unsigned base = (unsigned)((c.g * p.Cn + c.c0) * p.Ktot + (p.flip ? 8 - tap : tap) * p.Cg + slab * 32) * 2u;
    const int n = rd * 64 + (ltid >> 2);
    const bool ok = (c.live != 0) & (c.c0 + n < p.Cn) & (wch < p.Cg - slab * 32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sW + slot * WB + (rd * 256 + wl * 64) * 16), 16,
                                             ok ? base + (unsigned)(rd ? wrel1 : wrel0) : OOB, 0, 0, 0);
  };

  // ---- fragment addressing -------------------------------------------------------------------------------------------------------
  const int lp = lane & 15, lq = lane >> 4;
  const int arow = wc * 64 + (lp >> 2) * 8 + (lp & 3);  // + (ct >> 1) * 32 + (ct & 1) * 4  (bit 4 of the row does not depend on ct)
  const int ao = arow * 64 + ((lq ^ (((arow >> 4) & 1) << 1)) << 4);
  int bo[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) bo[q] = ((wimg * NPIX + wrow0 * HWD + q + lp) << 6) + ((lq ^ ((((q + lp) >> 2) & 1) << 1)) << 4);
  bf16x8_t fa[4], fb[4];
  auto load_b = [&](int bufo, int tap, int half) {
    const int r = tap / 3, q = tap - r * 3;
    const char* hb = sH + bufo + bo[q];
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = ldf(hb + (half * 4 + i + r) * (HWD * 64));
  };
  auto load_a = [&](int slot) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) fa[ct] = ldf(sW + slot * WB + ao + (ct >> 1) * 2048 + (ct & 1) * 256);
  };

  // ---- prologue: halo of the first HD slabs, taps 0 .. D-1 -----------------------------------------------------------------------
  TileC cur = decode(tile, true);
  TileC nx = decode(tile + tile_step < tile_end ? tile + tile_step : tile, tile + tile_step < tile_end);
  if (HROLE) {
#pragma unroll
    for (int h = 0; h < HD; ++h)
#pragma unroll
      for (int rd = 0; rd < HR; ++rd) issue_h(cur, h, h * HBYTES, rd);  // nslab >= 2 >= HD
    wvm_n((HD - 1) * HFULL);  // slab 0 landed; the full rounds of slab 1 may still be in flight
  } else {
#pragma unroll
    for (int t = 0; t < D; ++t) { issue_w(cur, 0, t, t, 0); issue_w(cur, 0, t, t, 1); }
    wvm<2 * (D - 1)>();  // tap 0
  }
  __builtin_amdgcn_s_barrier();

  int gs = 0;   // slabs retired: ring slot of (slab, tap) = (9 gs + tap) % 4 = (gs + tap) % 4
  // halo buffers as rotating byte offsets: current slab, next slab, the one being streamed into (== next when NHB == 2)
  int ho_cur = 0, ho_nxt = HBYTES, ho_tgt = (NHB - 1) * HBYTES;
  int tolerate = 0;  // epilogue stores of the previous tile that this wave put into the vmcnt FIFO ahead of this tile's loads (0, 8 or 16)
#pragma unroll 1
  for (; tile < tile_end; tile += tile_step) {
    f32x4_t acc[4][8];  // [channel tile][pixel row]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // store instructions of `cur`'s epilogue with at least one active lane: 8 per valid 32-channel pass (Cn % 64 may be 16 or 32; the
    // compiler branches around a store whose EXEC is empty).  Rows past a ragged map issue no store either: a LOWER bound of the
    // issued stores, rounded down to the two counts the wait knows.
    const int nvr = min(8, max(0, p.H - (cur.y0 + wrow0)));
    const int st_issued = cur.b0 + wimg < p.B ? (cur.c0 + wc * 64 + 32 < p.Cn ? 2 * nvr : cur.c0 + wc * 64 < p.Cn ? nvr : 0) : 0;
    const int st_wave = st_issued >= NST ? NST : (st_issued >= NST / 2 ? NST / 2 : 0);

    if (!HROLE) __builtin_amdgcn_s_barrier();  // the trailing group falls one barrier behind: its L segments meet the leaders' M segments

#pragma unroll 1
    for (int k = 0; k < nslab; ++k, ++gs) {
      const bool last = k == nslab - 1;
      const int kr = gs & 3;
      const int hb_cur = ho_cur, hb_nxt = ho_nxt, hb_tgt = ho_tgt;
      if (NHB == 3) { ho_cur = hb_nxt; ho_nxt = hb_tgt; ho_tgt = hb_cur; } else { ho_cur = hb_nxt; ho_nxt = hb_cur; ho_tgt = hb_cur; }
      const bool hin = k + HD < nslab;                      // the slab streamed in belongs to this tile (else to the next one)
      const int hs = hin ? k + HD : k + HD - nslab;
      const TileC htile = hin ? cur : nx;                   // scalar selects, once per slab: the issue code below is straight-line
      const TileC wtile = last ? nx : cur;
      const int wslab = last ? 0 : k + 1;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int t2 = t + D;
        // ================= phase 0: weights of this tap + pixel rows 0..3 =================
        load_a((kr + t) & 3);
        load_b(hb_cur, t, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (HROLE) {
          if (2 * t < HR) issue_h(htile, hs, hb_tgt, 2 * t);
        } else {
          if (t2 < 9) issue_w(cur, k, t2, (kr + t2) & 3, 0); else issue_w(wtile, wslab, t2 - 9, (kr + t2) & 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        lgk0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ct][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[i], acc[ct][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 1: pixel rows 4..7 =================
        load_b(hb_cur, t, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (HROLE) {
          if (2 * t + 1 < HR) issue_h(htile, hs, hb_tgt, 2 * t + 1);
          // the next slab's halo (issued one slab ago, or earlier in this one when NHB == 2) is read from the next stage on; what this
          // slab issued for the slab after it stays in flight
          if (t == 8) { if (NHB == 3) wvm_n(HFULL); else wvm<0>(); }
        } else {
          if (t2 < 9) issue_w(cur, k, t2, (kr + t2) & 3, 1); else issue_w(wtile, wslab, t2 - 9, (kr + t2) & 3, 1);
          // tap t+1 is read by the leading group right after the next barrier: taps t+2 and t+3 (four instructions) may stay in flight.
          // In a tile's first two stages the previous epilogue's stores sit between tap t+2 and tap t+3 in the FIFO (in order)
          if (k == 0 && t < 2 && tolerate == NST) wvm<2 * (D - 1) + NST>();
          else if (k == 0 && t < 2 && tolerate == NST / 2) wvm<2 * (D - 1) + NST / 2>();
          else wvm<2 * (D - 1)>();
        }
        __builtin_amdgcn_sched_barrier(0);
        lgk0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ct][4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[i], acc[ct][4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    if (HROLE) __builtin_amdgcn_s_barrier();  // the leading group waits here for the trailing group's last M segment: aligned again

    // ---- epilogue: this lane holds channels cl(h) .. cl(h)+7, h = 0 / 1, of pixels (wrow0 + pt, lp) of image b0 + wimg.  The two
    // 64-byte halves of a pixel's 128-byte line are stored back to back ---------------------------------------------------------------
    // lane-derived epilogue values are recomputed HERE from an opaque copy of the lane id: derived before the K loop (where the compiler
    // would hoist them, the tile's coordinates being known there) they are carried through it in registers the loop does not have
    int el = ltid;
    asm volatile("" : "+v"(el));
    const int e_lp = el & 15, e_lq = (el >> 4) & 3;
    const int e_tid = (HROLE ? 0 : 256) + el;
    const int bb = cur.b0 + wimg;
    const bool xok = cur.x0 + e_lp < p.W;
    const int cl0 = wc * 64 + e_lq * 8;
    bool cok[2];
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    f2_t s2[2][4], q2[2][4];  // BatchNorm sums of this lane's 2 x 8 channels, as register pairs (packed fp32 adds / FMAs)
    float sv[2][8], hv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cok[h] = cur.c0 + cl0 + h * 32 < p.Cn && bb < p.B;  // Cn % 16 == 0
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
      // training epilogue, ~3 VALU instructions per output (the first form took ~7): one v_cvt_pk_bf16_f32 per channel pair IS the store
      // operand; pixels past a ragged map are cleared by ONE and on the packed pair; the rounded values come back by a shift / mask for
      // the BatchNorm sums, which run as packed fp32 adds / FMAs (two channels per instruction)
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) {
        const int yy = cur.y0 + wrow0 + pt;
        const bool pok = xok & (yy < p.H);  // rows past a ragged map (H % TH != 0) are computed and dropped
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
            for (int j = 0; j < 4; ++j) {
              float u = acc[2 * h + c2][pt][j];
              u = u * sv[h][c2 * 4 + j] + hv[h][c2 * 4 + j];
              if (p.act) u = silu_f(u);
              v[c2 * 4 + j] = u;
            }
          if (pok && cok[h]) *(uint4*)(dst + h * 32) = Chunk<bf16_t>::pack(v);
        }
      }
    }
    if (EPI == 0 && p.part) {
      // `red` was last read a whole tile ago
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
      lgk0();  // not __syncthreads(): its fence would drain the DMA prefetch of the next tile
      __builtin_amdgcn_s_barrier();
      if (e_tid < 128 * NB) {
        const int img = e_tid >> 7, ch = e_tid & 127;
        if (cur.c0 + ch < p.Cn && cur.b0 + img < p.B) {
          float s = 0.f, q2 = 0.f;
#pragma unroll
          for (int w = 0; w < WPI; ++w) { s += red[((img * WPI + w) * 128 + ch) * 2]; q2 += red[((img * WPI + w) * 128 + ch) * 2 + 1]; }
          const long row = ((long)(cur.b0 + img) * p.nty + cur.ty) * p.ntx + cur.tx;
          float* dst = p.part + (row * (p.G * p.Cn) + cur.g * p.Cn + cur.c0 + ch) * 2;
          *(float2*)dst = make_float2(s, q2);
        }
      }
    }
    tolerate = st_wave;
    cur = nx;
    {
      const int t2 = tile + 2 * tile_step;
      nx = decode(t2 < tile_end ? t2 : tile, t2 < tile_end);
    }
  }
  wvm<0>();  // the dead loads past the last tile must land before the workgroup retires
}

template <int TH, int EPI>
__global__ __launch_bounds__(512, 1) void conv3x3_wide3_kernel(W3P p) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) wide3_body<TH, EPI, true>(p);
  else wide3_body<TH, EPI, false>(p);
}

int wide3_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int TH, int EPI>
int launch_wide3(const W3P& p, hipStream_t st) {
  constexpr int NB = 32 / TH;
  constexpr int HBYTES = NB * (TH + 2) * 18 * 64;
  size_t sm = (size_t)(TH == 16 ? 3 : 2) * HBYTES + 4 * 8192 + 4 * 128 * 2 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_wide3_kernel<TH, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  long ntiles = (long)p.G * p.nbt * p.nty * p.ntx * p.ntc;
  long nwg = (long)wide3_cu_count() / 8 * 8;  // persistent: one workgroup per CU (LDS-limited), a multiple of the 8 XCDs
  if (nwg < 8) nwg = 8;
  while (nwg > 8 && nwg / 8 > (ntiles + 7) / 8) nwg -= 8;
  hipLaunchKernelGGL((conv3x3_wide3_kernel<TH, EPI>), dim3((unsigned)nwg), dim3(512), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

int y3d_conv3x3_wide3_launch(int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G, const void* w,
                             int Ktot, void* y, long ysw, float* part, int flip, const float* scale, const float* shift, int act, void* stream) {
  W3P p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.part = part; p.scale = scale; p.shift = shift; p.act = act;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.H = H; p.W = W; p.Cg = Cg; p.Cn = Cn; p.G = G; p.Ktot = Ktot;
  p.ntx = cdiv(W, 16); p.nty = cdiv(H, th); p.ntc = cdiv(Cn, 128); p.nbt = cdiv(B, 32 / th); p.flip = flip;
  // extents in bytes (last addressable element + 1) of the input view and the packed weights; both must stay below 4 GB - 16
  const unsigned long xb = ((unsigned long)(B - 1) * xsb + (unsigned long)(H - 1) * xsh + (unsigned long)(W - 1) * xsw + (unsigned long)G * Cg) * 2;
  const unsigned long wb = (unsigned long)G * Cn * Ktot * 2;
  Y3D_CHECK(xb < 0xfffffff0ul && wb < 0xfffffff0ul, "conv3x3_wide: operand larger than 4 GB");
  Y3D_CHECK(2 * xsw < (1L << 23) && Ktot < (1 << 22), "conv3x3_wide: pixel stride beyond the 24-bit address multiply");
  Y3D_CHECK(Cg % 8 == 0 && Cg >= 40 && Cn % 16 == 0, "conv3x3_wide: Cg = %d must be a multiple of 8, at least 40 (two K slabs); Cn = %d a multiple of 16", Cg, Cn);
  p.xbytes = (unsigned)xb; p.wbytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  if (th == 16) return scale ? launch_wide3<16, 1>(p, st) : launch_wide3<16, 0>(p, st);
  return scale ? launch_wide3<8, 1>(p, st) : launch_wide3<8, 0>(p, st);
}
