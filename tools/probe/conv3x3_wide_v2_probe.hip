// ARCHIVED round-2 form of csrc/conv3x3_wide.hip with its -DY3D_PROBE_* / -DY3D_STAGGER ablation branches (tools/probe/build.sh);
// the product source no longer carries them.
// bf16 3x3 stride-1 "same" convolution, resident-halo implicit GEMM, PERSISTENT workgroups — the kernel that carries the
// YOLOv10-3D head (83 % of S-3D forward FLOPs are 3x3 convs at 128 channels per group) and its data gradient.
//
// Tile: NB images x (TH x 16) pixels x 128 output channels of one group, NB * TH = 32 (TH = 16: two images, TH = 8: four) —
// 512 pixels share every weight byte that is brought on chip, and batching over IMAGES instead of widening the spatial tile keeps
// the tiling exact for 80x80 / 40x40 maps.  8 waves = 2 channel halves x 4 pixel groups; a wave owns 64 channels x 128 pixels
// (8 rows of one image): 128 accumulator registers, 12 ds_read_b128 per 32 MFMAs.
//
// K runs over (32-channel slab, filter tap): one stage = one tap of one slab = one 16x16x32 MFMA k-step, 32 MFMAs per wave.
//   * the (TH+2) x 18 halo of the NB images of a slab (64-byte rows) is brought into LDS once and serves all nine taps; three
//     halo buffers at TH = 16: the halo of slab s+2 streams in while slab s is computed (HBM latency under load is ~2 us, one
//     slab is ~8 us of MFMA work, and the bytes in flight per CU are what bounds a latency-bound stream);
//   * the 128 x 32 weight tile of a tap (8 KB, L2-resident) streams through a 4-slot ring, three taps in flight;
//   * both by LDS-DMA (buffer_load ... lds, 16 B per lane: no VGPR staging, no ds_write) with the bank swizzle applied on the
//     per-lane SOURCE offset (the DMA destination is lane-linear); the conv's zero padding is the hardware range check of the
//     buffer descriptor (out-of-image lanes get an out-of-range offset and read zeros);
//   * the nine stages of a slab are unrolled: all LDS offsets are immediates, all wait counts compile-time constants; a stage
//     ends with a COUNTED s_waitcnt vmcnt(N) + raw s_barrier, the younger loads stay in flight across it;
//   * the two streams have different latencies and vmcnt retires in order, so they are issued by DIFFERENT waves: waves 0..3
//     (one per SIMD) stream halos, waves 4..7 weights, each group with its own wait counts (mixed in one FIFO every halo load
//     would have to land inside the weights' one-stage window).  The issuing wave is held for 250-450 cycles per DMA
//     instruction when the memory pipeline is busy, so the two waves of a SIMD issue at different points of the stage (halo:
//     top; weights: between the two halves) and the partner's MFMAs fill the matrix pipe meanwhile.  Measured on the head
//     shape (16 groups of 128->128 @80x80, B=32): 738 TFLOP/s for the first resident-halo kernel -> 870 (unrolled taps, counted
//     waits) -> 893 in the probe / 1 069 in the training step for this one (profiles/, tools/probe).
// The workgroup is persistent (one per CU): while the last slabs of a tile are computed, the first halos and taps of the NEXT
// tile are already streaming in, and the epilogue's stores drain behind the next tile's MFMAs (the first wait of a tile
// tolerates them in the vmcnt FIFO).  XCD x walks a contiguous run of tiles, its workgroups interleaved, so the tiles in flight
// share weights and halos in that XCD's L2.
//
// Output channels are permuted inside the MFMA row index (row r of channel tile ct is channel (ct>>1)*32 + (r>>2)*8 + (ct&1)*4 +
// (r&3)), so a lane ends with two runs of 8 consecutive channels and one store instruction writes 64 contiguous bytes per pixel
// (with 16-byte pieces at a 32-byte stride the L2 wrote 1.8x the tensor: PMC WRITE_SIZE).  The epilogue also emits the per-tile BatchNorm partial
// sums (sum, sum of squares of the ROUNDED outputs) or applies the eval-mode affine + SiLU.
// The data gradient of such a conv is the same kernel on dy with the taps flipped (`flip`).
// Register budget: 128 accumulators + 48 fragment registers leave ~60 for everything else at two waves per SIMD; the DMA
// addresses are therefore recomputed at the issue point from tile scalars (an opaque asm keeps LICM from hoisting them into
// resident registers) and the weight fragments are refreshed in place.  The build must stay spill-free: a scratch reload in the
// loop makes the compiler wait vmcnt(0) and drains the DMA pipeline (seen: -25 %).
#include "../../yolov10-3d_amd/csrc/common.h"

namespace {

#ifdef Y3D_PROBE_TRACE
__device__ unsigned y3d_probe_trace[2 * 18 * 6];
#endif
#ifdef Y3D_PROBE_STAMP
__device__ unsigned long long y3d_probe_stamps[4096 * 4];
#define Y3D_WSTAMP(i, v) if (threadIdx.x == 0) y3d_probe_stamps[blockIdx.x * 4 + (i)] v
#else
#define Y3D_WSTAMP(i, v)
#endif

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

// count known after unrolling: the switch folds to the one s_waitcnt
__device__ __forceinline__ void wvm_n(int n) {
  switch (n) {
    case 0: wvm<0>(); break; case 1: wvm<1>(); break; case 2: wvm<2>(); break; case 3: wvm<3>(); break;
    case 4: wvm<4>(); break; case 5: wvm<5>(); break; case 6: wvm<6>(); break; case 7: wvm<7>(); break;
    case 8: wvm<8>(); break; case 9: wvm<9>(); break; case 10: wvm<10>(); break; case 11: wvm<11>(); break;
    case 12: wvm<12>(); break; case 13: wvm<13>(); break; case 14: wvm<14>(); break; case 15: wvm<15>(); break;
    case 16: wvm<16>(); break; case 17: wvm<17>(); break; case 18: wvm<18>(); break;
    case 19: wvm<19>(); break; case 20: wvm<20>(); break; case 21: wvm<21>(); break; case 22: wvm<22>(); break;
    case 23: wvm<23>(); break; case 24: wvm<24>(); break; case 25: wvm<25>(); break; case 26: wvm<26>(); break;
    default: wvm<0>(); break;
  }
}

#ifdef Y3D_PROBE_NOLDS
__device__ __forceinline__ bf16x8_t ldf(const char* p) { unsigned a = (unsigned)(size_t)p; return __builtin_bit_cast(bf16x8_t, make_uint4(a, a, a, a)); }
#else
__device__ __forceinline__ bf16x8_t ldf(const char* p) { return __builtin_bit_cast(bf16x8_t, *(const uint4*)p); }
#endif

// HROLE: this wave streams the halo (waves 0..3), else the weights (waves 4..7).  The two roles run separate, branch-free copies
// of the loop (same barriers, same MFMAs) so that each keeps compile-time wait counts on its own vmcnt FIFO.
template <int TH, int EPI, bool HROLE>
__device__ __forceinline__ void wide_body(const W3P& p) {
  constexpr int NB = 32 / TH;              // images per tile
  constexpr int WPI = TH / 8;              // pixel-group waves per image
  constexpr int HWD = 18;
  constexpr int NPIX = (TH + 2) * HWD;     // halo pixels per image
  constexpr int HCH = NB * NPIX * 4;       // 16-byte chunks of one halo slab (64-byte rows)
  constexpr int LH = 256;                  // lanes of a loader group (4 waves)
  constexpr int HR = (HCH + LH - 1) / LH;  // DMA rounds per halo slab
  constexpr int HFULL = HCH / LH;          // rounds every halo wave takes part in (what the wait counts may rely on)
  constexpr int HBYTES = HCH * 16;
  constexpr int NHB = TH == 16 ? 3 : 2;    // halo buffers: the halo of slab s + NHB - 1 streams in while slab s is computed
  constexpr int HD = NHB - 1;
  constexpr int RD = 4, D = 3;             // weight ring slots, taps in flight
  constexpr int WB = 8192;                 // bytes of one weight tile (128 rows x 64 B)
  constexpr int NST = 16;                  // epilogue store instructions of a wave that owns valid channels
  static_assert(HR <= 14, "two halo rounds per stage, all issued by stage 6");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;                         // [NHB][HBYTES]
  char* sW = smem + NHB * HBYTES;          // [RD][WB]
  float* red = (float*)(sW + RD * WB);     // [4][128][2]
#ifdef Y3D_PROBE_TRACE
  unsigned* trc = (unsigned*)(red + 4 * 128 * 2);  // [2 waves][18 stages][6]
  int trc_tile = 0;
#define TRC(i) if (blockIdx.x == 0 && trc_tile == 2 && k < 2 && (wave & 3) == 0 && lane == 0) trc[((wave >> 2) * 18 + k * 9 + t) * 6 + (i)] = (unsigned)__builtin_amdgcn_s_memtime()
#else
#define TRC(i)
#endif

  // `wave` and everything derived from it is wave-uniform: scalar registers (the compiler cannot prove threadIdx.x >> 6 uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wp = wave >> 1;
  const int slot = wave & 3;  // DMA issue point of this wave inside its half of the stage (Y3D_STAGGER)
  const int ltid = tid & (LH - 1);
  const int wimg = wp / WPI, wrow0 = (wp % WPI) * 8;
  const int nslab = p.Cg >> 5;             // >= 2 (launcher)
  Y3D_WSTAMP(0, = __builtin_amdgcn_s_memtime());

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

  // ---- DMA issue: buffer loads (32-bit byte offset per lane, hardware range check -> out-of-range lanes read zeros, which is
  // the conv's zero padding); the offsets are derived on the fly from the tile's scalars (no per-tile tables, no resident
  // pointers, no 64-bit lane arithmetic).  Tensors are < 4 GB - 16 (launcher).
  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  // halo: chunk = rd * LH + ltid -> pixel P = chunk / 4 of the NB x (TH+2) x 18 halo, 16-byte piece s of its 64-byte slab row;
  // rows are swizzled by their column: piece ^= 2 * bit2(hx)  (conflict-free 16-pixel fragment reads)
  auto issue_h = [&](const TileC& c, int slab, int bufo, int rd) {  // bufo: byte offset of the halo buffer
    int l = ltid;
    asm volatile("" : "+v"(l));  // opaque: keeps the address arithmetic at the issue point (LICM would hoist all rounds into registers)
    l &= LH - 1;
    // straight-line on purpose (bitwise conditions, multiply-shift division): this runs between MFMAs every stage
    const int chunk = rd * LH + l;
    const int P = chunk >> 2, s = chunk & 3;
    int img = P >= NPIX;
    if (NB > 2) img += (P >= 2 * NPIX) + (P >= 3 * NPIX);
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate and this runs ~20 times per slab between the MFMAs): every factor is
    // far below 2^23 (pixel strides: the launcher checks), the image term needs no multiply at all
    const int pp = P - __mul24(img, NPIX);
    const int hy = __mul24(pp, 3641) >> 16;  // pp / 18, exact for pp < 8192
    const int hx = pp - __mul24(hy, HWD);
    const int bb = c.b0 + img, yy = c.y0 + hy - 1, xx = c.x0 + hx - 1;
    const bool inb = (c.live != 0) & (bb < p.B) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
    const int xsb = (int)p.xsb;
    const int imgoff = NB > 2 ? ((img & 1) ? xsb : 0) + ((img & 2) ? 2 * xsb : 0) : (img ? xsb : 0);
    const unsigned off = (unsigned)(c.b0 * xsb + imgoff + __mul24(yy, (int)p.xsh) + __mul24(xx, (int)p.xsw) + c.g * p.Cg + slab * 32 +
                                    ((s ^ (((hx >> 2) & 1) << 1)) << 3)) * 2u;
#ifdef Y3D_PROBE_CHEAPADDR  // upper bound of what cheaper halo address arithmetic could give (wrong data)
    if (rd < HFULL || chunk < HCH)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sH + bufo + (rd * LH + wave * 64) * 16), 16,
                                               (unsigned)(l * 16 + rd * 4096 + slab * 64), 0, 0, 0);
    return;
#endif
#ifdef Y3D_PROBE_CONTIG  // the full address arithmetic, kept live, but a contiguous (wrong) address in the load: isolates memory locality
    {
      unsigned keep = inb ? off : OOB;
      asm volatile("" ::"v"(keep));
      if (rd < HFULL || chunk < HCH)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sH + bufo + (rd * LH + wave * 64) * 16), 16,
                                                 (unsigned)(l * 16 + rd * 4096 + slab * 64), 0, 0, 0);
      return;
    }
#endif
    if (rd < HFULL || chunk < HCH)  // lanes past the end of the last (partial) round must not write LDS
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sH + bufo + (rd * LH + wave * 64) * 16), 16,
                                               inb ? off : OOB, 0, 0, 0);
  };
  // weights: chunk = rd * LH + ltid -> tile row n = chunk / 4 (output channel), piece s; rows swizzled by piece ^= 2 * bit4(n).
  // Two DMA instructions per tap, always (the wait counts depend on it).
  auto issue_w = [&](const TileC& c, int slab, int tap, int slot) {
    const unsigned base = (unsigned)((c.g * p.Cn + c.c0) * p.Ktot + (p.flip ? 8 - tap : tap) * p.Cg + slab * 32) * 2u;
    int l = ltid;
    asm volatile("" : "+v"(l));
    l &= LH - 1;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int chunk = rd * LH + l, n = chunk >> 2, s = chunk & 3;
      const unsigned wrel = (unsigned)(__mul24(n, p.Ktot) + ((s ^ (((n >> 4) & 1) << 1)) << 3)) * 2u;  // recomputed: no resident registers
      const bool ok = (c.live != 0) & (c.c0 + n < p.Cn);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sW + slot * WB + (rd * LH + (wave - 4) * 64) * 16), 16,
                                               ok ? base + wrel : OOB, 0, 0, 0);
    }
  };
  // halo rounds of stage t: two per stage, all issued by stage 5
  auto halo_stage = [&](const TileC& c, int slab, int bufo, int t) {
    if (2 * t < HR) issue_h(c, slab, bufo, 2 * t);
    if (2 * t + 1 < HR) issue_h(c, slab, bufo, 2 * t + 1);
  };
  constexpr int HY7 = HFULL;  // full rounds issued in stages 0..7 of a slab: all of them

  // ---- fragment addressing -------------------------------------------------------------------------------------------------------
  const int lp = lane & 15, lq = lane >> 4;
  const int arow = wc * 64 + (lp >> 2) * 8 + (lp & 3);  // + (ct >> 1) * 32 + (ct & 1) * 4  (bit 4 of the row does not depend on ct)
  const int ao = arow * 64 + ((lq ^ (((arow >> 4) & 1) << 1)) << 4);
  int bo[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) bo[q] = ((wimg * NPIX + wrow0 * HWD + q + lp) << 6) + ((lq ^ ((((q + lp) >> 2) & 1) << 1)) << 4);
  bf16x8_t fa[4], fb[2][4];
  auto load_b = [&](bf16x8_t* dst, int bufo, int tap, int half) {
    const int r = tap / 3, q = tap - r * 3;
    const char* hb = sH + bufo + bo[q];
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = ldf(hb + (half * 4 + i + r) * (HWD * 64));
  };
  auto load_a1 = [&](int slot, int ct) { return ldf(sW + slot * WB + ao + (ct >> 1) * 2048 + (ct & 1) * 256); };

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
    for (int t = 0; t < D; ++t) issue_w(cur, 0, t, t);
    wvm<2 * (D - 2)>();  // taps 0, 1
  }
  __builtin_amdgcn_s_barrier();
  Y3D_WSTAMP(1, = __builtin_amdgcn_s_memtime());

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
    // first fragments of the tile (not carried across the previous epilogue: that would pin registers there)
    load_b(fb[0], ho_cur, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) fa[ct] = load_a1(gs & 3, ct);
    // store instructions of `cur`'s epilogue with at least one active lane: 8 per valid 32-channel pass (Cn % 64 may be 16 or 32; the
    // compiler branches around a store whose EXEC is empty).  Counting only those is the safe side for the wait below.
    // Rows past a ragged map issue no store either: a LOWER bound of the issued stores, rounded down to the two counts the wait knows.
    const int nvr = min(8, max(0, p.H - (cur.y0 + wrow0)));
    const int st_issued = cur.b0 + wimg < p.B ? (cur.c0 + wc * 64 + 32 < p.Cn ? 2 * nvr : cur.c0 + wc * 64 < p.Cn ? nvr : 0) : 0;
    const int st_wave = st_issued >= NST ? NST : (st_issued >= NST / 2 ? NST / 2 : 0);

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
        TRC(0);
        // ---- issue (halo role): at the top of the stage.  The LDS-DMA engine of a CU moves ~1 KB per ~40 cycles and the issuing
        // wave is held meanwhile (measured: 250-450 cycles per instruction when all eight waves issue together), so the two waves
        // of a SIMD issue at DIFFERENT times: the halo wave here, while its partner runs half 0's MFMAs; the weight wave between
        // the halves, while this one computes --------------------------------------------------------------------------------------
#if !defined(Y3D_PROBE_NODMA) && !defined(Y3D_STAGGER)
        if (HROLE) {
#ifndef Y3D_PROBE_NOHALO
          __builtin_amdgcn_sched_barrier(0);  // confine the address arithmetic to the top of the stage, where the live set is smallest
          halo_stage(htile, hs, hb_tgt, t);
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
#endif
        TRC(1);
        // ---- half 0: pixel rows 0..3 while rows 4..7 of this tap are fetched --------------------------------------------------------
        load_b(fb[1], hb_cur, t, 1);
#ifdef Y3D_STAGGER
        // The four halo waves (one per SIMD) issue their DMA instructions at FOUR different points of half 0 - before MFMA group
        // `slot` = wave & 3 - and the four weight waves likewise inside half 1: eight instructions arriving together queue behind
        // each other in the CU's one LDS-DMA path and hold every issuing wave for 450-570 cycles of a 1500-cycle stage (probe trace);
        // spread over the stage each finds the path nearly free.
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          if (HROLE && ct == slot) {
            __builtin_amdgcn_sched_barrier(0);
            halo_stage(htile, hs, hb_tgt, t);
            __builtin_amdgcn_sched_barrier(0);
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ct][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[0][i], acc[ct][i], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
#else
        __builtin_amdgcn_s_setprio(1);  // keeps the MFMA cluster together and ahead of the partner wave's VALU / DMA issue (+2 %)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#ifdef Y3D_PROBE_NOMFMA
            asm volatile("" ::"v"(fa[ct]), "v"(fb[0][i]));
#else
            acc[ct][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[0][i], acc[ct][i], 0, 0, 0);
#endif
          }
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- half 1: rows 4..7 while the first fragments of the next stage are fetched (its tap was published one barrier ago);
        // the weight fragments are refreshed IN PLACE, each right behind the last MFMA that reads it (no second buffer: the
        // 128 accumulators leave no room for one) ------------------------------------------------------------------------------------
        TRC(2);
        __builtin_amdgcn_sched_barrier(0);  // keep half 1's fragment loads out of half 0: both sets live at once would spill
#if !defined(Y3D_PROBE_NODMA) && !defined(Y3D_STAGGER)
        if (!HROLE) {
#ifndef Y3D_PROBE_NOWEIGHT
          __builtin_amdgcn_sched_barrier(0);
          const int t2 = t + D;
          if (t2 < 9) issue_w(cur, k, t2, (kr + t2) & 3);
          else issue_w(wtile, wslab, t2 - 9, (kr + t2) & 3);
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
#endif
        const bool pre = t < 8 || !last;
        if (t < 8) load_b(fb[0], hb_cur, t + 1, 0); else if (!last) load_b(fb[0], hb_nxt, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#ifdef Y3D_STAGGER
          if (!HROLE && ct == slot) {
            __builtin_amdgcn_sched_barrier(0);
            const int t2 = t + D;
            if (t2 < 9) issue_w(cur, k, t2, (kr + t2) & 3);
            else issue_w(wtile, wslab, t2 - 9, (kr + t2) & 3);
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#ifdef Y3D_PROBE_NOMFMA
            asm volatile("" ::"v"(fa[ct]), "v"(fb[1][i]));
#else
            acc[ct][4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[1][i], acc[ct][4 + i], 0, 0, 0);
#endif
          }
          if (pre) fa[ct] = load_a1((kr + t + 1) & 3, ct);
        }
        TRC(3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- retire ----------------------------------------------------------------------------------------------------------------
        if (HROLE) {
          // the next slab's halo (issued one slab ago, or earlier in this one when NHB == 2) is read from stage 8 on; what this
          // slab issued so far for the slab after it stays in flight
          if (t == 7) { if (NHB == 3) wvm_n(HY7); else wvm<0>(); }
        } else {
          // tap t+2 was issued in stage t-1: only this stage's two loads are younger.  In a tile's first stage the previous
          // epilogue's stores are younger than the target too (vmcnt retires in order).
          if (t == 0 && k == 0 && tolerate == NST) wvm<2 + NST>();
          else if (t == 0 && k == 0 && tolerate == NST / 2) wvm<2 + NST / 2>();
          else wvm<2>();
        }
        TRC(4);
        __builtin_amdgcn_s_barrier();
        TRC(5);
      }
    }

    // ---- epilogue: this lane holds channels cl(h) .. cl(h)+7, h = 0 / 1, of pixels (wrow0 + pt, lp) of image b0 + wimg.  The two
    // 64-byte halves of a pixel's 128-byte line are stored back to back: written a pass apart (8 stores later) the L2 had evicted
    // the half-dirty line in between and re-filled it for the second half (PMC: WRITE_SIZE 1.36x the tensor, FETCH_SIZE up too) ------
    // Lane-derived epilogue values are recomputed HERE from the one lane id the DMA issue keeps live anyway (opaque copy): carried
    // through the K loop they were spilled to scratch, and a scratch reload is followed by s_waitcnt vmcnt(0) - the drain of the next
    // tile's DMA prefetch that this kernel is built to avoid
    int el = ltid;
    asm volatile("" : "+v"(el));
    const int e_lp = el & 15, e_lq = (el >> 4) & 3;
    const int e_tid = (HROLE ? 0 : LH) + el;
    const int bb = cur.b0 + wimg;
    const bool xok = cur.x0 + e_lp < p.W;
    const int cl0 = wc * 64 + e_lq * 8;
    bool cok[2];
    float ssum[2][8], ssq[2][8], sv[2][8], hv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cok[h] = cur.c0 + cl0 + h * 32 < p.Cn && bb < p.B;  // Cn % 16 == 0
#pragma unroll
      for (int i = 0; i < 8; ++i) { ssum[h][i] = 0.f; ssq[h][i] = 0.f; }
      if (EPI == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sv[h][i] = cok[h] ? p.scale[cur.g * p.Cn + cur.c0 + cl0 + h * 32 + i] : 1.f;
          hv[h][i] = cok[h] ? p.shift[cur.g * p.Cn + cur.c0 + cl0 + h * 32 + i] : 0.f;
        }
      }
    }
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
      const int yy = cur.y0 + wrow0 + pt;
      const bool pok = xok & (yy < p.H);  // rows past a ragged map (H % TH != 0) are computed and dropped
      bf16_t* dst = p.y + (((long)bb * p.H + yy) * p.W + cur.x0 + e_lp) * p.ysw + (long)cur.g * p.Cn + cur.c0 + cl0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float u = acc[2 * h + c2][pt][j];
            if (EPI == 1) { u = u * sv[h][c2 * 4 + j] + hv[h][c2 * 4 + j]; if (p.act) u = silu_f(u); }
            u = pok ? bf2f(f2bf(u)) : 0.f;
            v[c2 * 4 + j] = u;
            if (EPI == 0) { ssum[h][c2 * 4 + j] += u; ssq[h][c2 * 4 + j] += u * u; }
          }
#ifdef Y3D_PROBE_NOEPI
        if (pok && cok[h] && v[0] == 123.456f) {
#else
        if (pok && cok[h]) {
#endif
          *(uint4*)(dst + h * 32) = Chunk<bf16_t>::pack(v);
        }
      }
    }
    if (EPI == 0 && p.part) {
      // `red` was last read a whole tile (>= 18 barriers) ago
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float s = wave_xor_sum16(ssum[h][i]);
          float q2 = wave_xor_sum16(ssq[h][i]);
          if (e_lp == i) {
            red[(wp * 128 + cl0 + h * 32 + i) * 2 + 0] = s;
            red[(wp * 128 + cl0 + h * 32 + i) * 2 + 1] = q2;
          }
        }
    }
    if (EPI == 0 && p.part) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // not __syncthreads(): its fence would drain the DMA prefetch of the next tile
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
#ifdef Y3D_PROBE_TRACE
    ++trc_tile;
#endif
    tolerate = st_wave;
    cur = nx;
    {
      const int t2 = tile + 2 * tile_step;
      nx = decode(t2 < tile_end ? t2 : tile, t2 < tile_end);
    }
  }
  wvm<0>();  // the dead loads past the last tile must land before the workgroup retires
#ifdef Y3D_PROBE_TRACE
  __syncthreads();
  if (blockIdx.x == 0 && tid < 2 * 18 * 6) y3d_probe_trace[tid] = trc[tid];
#endif
  Y3D_WSTAMP(3, = __builtin_amdgcn_s_memtime());
}

template <int TH, int EPI>
__global__ __launch_bounds__(512, 1) void conv3x3_wide_kernel(W3P p) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) wide_body<TH, EPI, true>(p);
  else wide_body<TH, EPI, false>(p);
}

int wide_cu_count() {
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
int launch_wide(const W3P& p, hipStream_t st) {
  constexpr int NB = 32 / TH;
  constexpr int HCH = NB * (TH + 2) * 18 * 4;
  size_t sm = (size_t)(TH == 16 ? 3 : 2) * HCH * 16 + 4 * 8192 + 4 * 128 * 2 * 4 + 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_wide_kernel<TH, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    attr_set = true;
  }
  long ntiles = (long)p.G * p.nbt * p.nty * p.ntx * p.ntc;
  long nwg = (long)wide_cu_count() / 8 * 8;  // persistent: one workgroup per CU (LDS-limited), a multiple of the 8 XCDs
  if (nwg < 8) nwg = 8;
  while (nwg > 8 && nwg / 8 > (ntiles + 7) / 8) nwg -= 8;
  hipLaunchKernelGGL((conv3x3_wide_kernel<TH, EPI>), dim3((unsigned)nwg), dim3(512), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

int y3d_conv3x3_wide_ok(int H, int W, int Cg, int Cn) { return H >= 8 && W >= 8 && Cg % 32 == 0 && Cg >= 64 && Cn % 16 == 0; }

int y3d_conv3x3_wide_launch(int th, const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G, const void* w,
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
  Y3D_CHECK(xsh < (1L << 23) && xsw < (1L << 23) && Ktot < (1 << 23), "conv3x3_wide: pixel strides beyond the 24-bit address multiplies");
  p.xbytes = (unsigned)xb; p.wbytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  if (th == 16) return scale ? launch_wide<16, 1>(p, st) : launch_wide<16, 0>(p, st);
  return scale ? launch_wide<8, 1>(p, st) : launch_wide<8, 0>(p, st);
}
