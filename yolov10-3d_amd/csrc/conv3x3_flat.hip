// bf16 3x3 stride-1 "same" convolution on a FLATTENED padded pixel space - conv3x3_wide3.hip's pipeline (persistent workgroups, resident
// halo + weight ring by LDS-DMA, SIMD partners half a phase apart) with a tile geometry that does not depend on the map's width / height.
//
// Why: wide3's tile is NB images x TH x 16 pixels.  On the stride-16 / stride-32 head levels (40x40, 20x20) 16-pixel tile rows keep 83 % /
// 62 % useful columns and the 20-row maps 83 % useful rows: the P5 head layers ran at 435-742 TFLOP/s against 1 200 at P3 (VERDICT round 3,
// item 3).  Here every image is taken as its zero-padded (H + 2) x (W + 2) map, the maps of the batch are laid end to end, and a tile is 512
// CONSECUTIVE positions f of that flat space, whatever rows / images they span: a filter tap (r, q) of output position j is input position
// j + (r - 1) (W + 2) + (q - 1) - a constant shift - and the padding ring supplies the zeros (and keeps neighbouring images apart).
// Outputs that fall on padding positions are computed and dropped: (H + 2)(W + 2) / (H W) = 1.21 at 20x20, 1.10 at 40x40 (wide3: 1.92 /
// 1.20).  The LDS halo of a tile is the 512 + 2 (W + 3) positions around it (rounded to 64), 64 bytes each as in wide3; its piece swizzle is
// keyed on bit 2 of the LDS pixel index, which keeps 16 consecutive pixels conflict-free for ANY start offset (wide3 keys on the column of a
// 2-D halo, which is why it could not shift by arbitrary taps).  The DMA maps a flat position to (image, row, column) once per slab by
// division and then walks it incrementally (64 positions per round); positions on the padding ring / past the batch take the
// out-of-range offset and read zeros.  One BatchNorm partial row per tile.
#include "common.h"

namespace {


struct WFP {
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
  int ntf, ntc;             // flat pixel tiles, channel tiles
  int flip;
  unsigned xbytes, wbytes;  // buffer extents for the hardware range check
  int HP, WP, IMG;          // padded map: H + 2, W + 2, their product
  int HPIX;                 // halo positions per tile: 512 + 2 (WP + 1) rounded up to a multiple of 64
  float inv_img, inv_wp;    // reciprocals for the flat -> (image, row, column) divisions
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
template <int NHB, int EPI, bool HROLE>
__device__ __forceinline__ void flat_body(const WFP& p) {
  constexpr int HD = NHB - 1;              // halo slabs in flight ahead of the one being computed
  constexpr int RD = 4, D = 3;             // weight ring slots, taps in flight
  constexpr int WB = 8192;                 // bytes of one weight tile (128 rows x 64 B)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HBYTES = p.HPIX * 64;
  const int NI = p.HPIX >> 4;              // halo DMA instructions per slab (16 positions x 64 B each), a multiple of 4
  const int HR = NI >> 2;                  // rounds: instruction rd * 4 + w goes to halo wave w; every round is full (HR <= 12)
  char* sH = smem;                         // [NHB][HBYTES]
  char* sW = smem + NHB * HBYTES;          // [RD][WB]
  float* red = (float*)(sW + RD * WB);     // [4][128][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wp = wave >> 1;
  const int wl = wave & 3;
  const int nslab = (p.Cg + 31) >> 5;      // >= 2 (launcher)

  // ---- persistent schedule ------------------------------------------------------------------------------------------------------
  const int ntiles = p.G * p.ntf * p.ntc;
  int tile, tile_end, tile_step;
  {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tile_step = gridDim.x >> 3;
    const int lo = (int)((long)ntiles * xcd / 8);
    tile_end = (int)((long)ntiles * (xcd + 1) / 8);
    tile = lo + slot;
  }
  if (tile >= tile_end) return;  // uniform per workgroup

  struct TileC { int g, f0, c0, tf, live; };
  auto decode = [&](int t, bool live) {
    TileC c;
    int tc = t % p.ntc; t /= p.ntc;
    c.tf = t % p.ntf; c.g = t / p.ntf;
    c.f0 = c.tf * 512; c.c0 = tc * 128; c.live = live;
    return c;
  };

  constexpr unsigned OOB = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
  // ---- halo DMA: instruction i = rd * 4 + wl of a slab covers halo positions 16 i .. 16 i + 15 = flat positions f0 - (WP + 1) + 16 i + lj,
  // 64 B each as four 16-byte pieces (lane = 4 * position + piece), piece ^= 2 * bit2(LDS position) = 2 * bit2(lj): a lane constant.
  // (image, padded row, padded column) of a lane's position: by division in round 0 of a slab, then +64 positions per round.
  const int xsw2 = (int)p.xsw * 2, xsh2 = (int)p.xsh * 2, xsb2 = (int)p.xsb * 2;
  const int lj0 = lane >> 2;
  const int pc0 = ((lane & 3) << 4) ^ ((lj0 & 4) << 3);
  const int a64 = 64 / p.WP, c64 = 64 - a64 * p.WP;
  int hb = 0, hy = 0, hx = 0;  // walking state of this lane (halo waves)
  auto issue_h = [&](const TileC& c, int slab, int bufo, int rd) {
    if (rd == 0) {
      // f = f0 - (WP + 1) + 16 wl + lj, shifted by one image so that it is never negative
      int lj = lj0;
      asm volatile("" : "+v"(lj));
      const int fs = c.f0 - (p.WP + 1) + 16 * wl + lj + p.IMG;
      int q = (int)((float)fs * p.inv_img);
      int r = fs - q * p.IMG;
      if (r < 0) { r += p.IMG; --q; }
      if (r >= p.IMG) { r -= p.IMG; ++q; }
      int y = (int)((float)r * p.inv_wp);
      int x = r - y * p.WP;
      if (x < 0) { x += p.WP; --y; }
      if (x >= p.WP) { x -= p.WP; ++y; }
      hb = q - 1; hy = y; hx = x;
    } else {
      hx += c64; hy += a64;
      if (hx >= p.WP) { hx -= p.WP; ++hy; }
      if (hy >= p.HP) { hy -= p.HP; ++hb; }
    }
    const bool ok = (c.live != 0) & ((unsigned)hb < (unsigned)p.B) & ((unsigned)(hy - 1) < (unsigned)p.H) & ((unsigned)(hx - 1) < (unsigned)p.W) &
                    (pc0 < (p.Cg - slab * 32) * 2);
    const unsigned off = (unsigned)(hb * xsb2 + (hy - 1) * xsh2 + __mul24(hx - 1, xsw2) + (c.g * p.Cg + slab * 32) * 2 + pc0);
    char* dst = sH + bufo + (rd * 4 + wl) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, ok ? off : OOB, 0, 0, 0);
  };
  // ---- weight DMA: as conv3x3_wide3.hip
  const int ltid = tid & 255;
  int wrel0, wrel1, wch;
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
  const int arow = wc * 64 + (lp >> 2) * 8 + (lp & 3);
  const int ao = arow * 64 + ((lq ^ (((arow >> 4) & 1) << 1)) << 4);
  const int pb0 = wp * 128 + lp;  // + r * WP + q: halo position of this lane's pixel in chunk 0 for tap (r, q)
  bf16x8_t fa[4], fb[4];
  int bo;  // byte offset of the current tap's chunk-0 fragment inside a halo buffer
  auto set_tap = [&](int tap) {
    const int r = tap / 3, q = tap - r * 3;
    const int pb = pb0 + r * p.WP + q;
    bo = (pb << 6) + ((lq ^ ((pb >> 1) & 2)) << 4);
  };
  auto load_b = [&](int bufo, int half) {
    const char* hbp = sH + bufo + bo;
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = ldf(hbp + (half * 4 + i) * 1024);
  };
  auto load_a = [&](int slot) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) fa[ct] = ldf(sW + slot * WB + ao + (ct >> 1) * 2048 + (ct & 1) * 256);
  };

  // ---- prologue: halo of the first HD slabs, taps 0 .. D-1 -----------------------------------------------------------------------
  TileC cur = decode(tile, true);
  TileC nx = decode(tile + tile_step < tile_end ? tile + tile_step : tile, tile + tile_step < tile_end);
  if (HROLE) {
#pragma unroll 1
    for (int h = 0; h < HD; ++h)
#pragma unroll 1
      for (int rd = 0; rd < HR; ++rd) issue_h(cur, h, h * HBYTES, rd);  // nslab >= 2 >= HD
    wvm_n((HD - 1) * HR);  // slab 0 landed; slab 1 may still be in flight
  } else {
#pragma unroll
    for (int t = 0; t < D; ++t) { issue_w(cur, 0, t, t, 0); issue_w(cur, 0, t, t, 1); }
    wvm<2 * (D - 1)>();  // tap 0
  }
  __builtin_amdgcn_s_barrier();

  int gs = 0;
  int ho_cur = 0, ho_nxt = HBYTES, ho_tgt = (NHB - 1) * HBYTES;
  int tolerate = 0;  // epilogue stores of the previous tile that this wave put into the vmcnt FIFO ahead of this tile's loads (exact count)
#pragma unroll 1
  for (; tile < tile_end; tile += tile_step) {
    f32x4_t acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (!HROLE) __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int k = 0; k < nslab; ++k, ++gs) {
      const bool last = k == nslab - 1;
      const int kr = gs & 3;
      const int hb_cur = ho_cur, hb_nxt = ho_nxt, hb_tgt = ho_tgt;
      if (NHB == 3) { ho_cur = hb_nxt; ho_nxt = hb_tgt; ho_tgt = hb_cur; } else { ho_cur = hb_nxt; ho_nxt = hb_cur; ho_tgt = hb_cur; }
      const bool hin = k + HD < nslab;
      const int hs = hin ? k + HD : k + HD - nslab;
      const TileC htile = hin ? cur : nx;
      const TileC wtile = last ? nx : cur;
      const int wslab = last ? 0 : k + 1;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int t2 = t + D;
        // ================= phase 0: weights of this tap + pixel chunks 0..3 =================
        set_tap(t);
        load_a((kr + t) & 3);
        load_b(hb_cur, 0);
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
        // ================= phase 1: pixel chunks 4..7 =================
        load_b(hb_cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (HROLE) {
          if (2 * t + 1 < HR) issue_h(htile, hs, hb_tgt, 2 * t + 1);
          if (t == 8) { if (NHB == 3) wvm_n(HR); else wvm<0>(); }
        } else {
          if (t2 < 9) issue_w(cur, k, t2, (kr + t2) & 3, 1); else issue_w(wtile, wslab, t2 - 9, (kr + t2) & 3, 1);
          if (k == 0 && t < 2) wvm_n(2 * (D - 1) + tolerate);
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
    if (HROLE) __builtin_amdgcn_s_barrier();

    // ---- epilogue: this lane holds channels cl(h) .. cl(h)+7, h = 0 / 1, of flat positions f0 + 128 wp + 16 pt + lp, pt = 0..7 -----------
    int el = ltid;
    asm volatile("" : "+v"(el));
    const int e_lp = el & 15, e_lq = (el >> 4) & 3;
    const int e_tid = (HROLE ? 0 : 256) + el;
    const int cl0 = wc * 64 + e_lq * 8;
    const int a16 = 16 / p.WP, c16 = 16 - a16 * p.WP;
    int eb, ey, ex;
    {
      const int f = cur.f0 + wp * 128 + e_lp;
      int q = (int)((float)f * p.inv_img);
      int r = f - q * p.IMG;
      if (r < 0) { r += p.IMG; --q; }
      if (r >= p.IMG) { r -= p.IMG; ++q; }
      int y = (int)((float)r * p.inv_wp);
      int x = r - y * p.WP;
      if (x < 0) { x += p.WP; --y; }
      if (x >= p.WP) { x -= p.WP; ++y; }
      eb = q; ey = y; ex = x;
    }
    bool cok[2];
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    f2_t s2[2][4], q2[2][4];
    float sv[2][8], hv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cok[h] = cur.c0 + cl0 + h * 32 < p.Cn;  // Cn % 16 == 0
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
    int issued = 0;  // store instructions with at least one active lane (what the next tile's first waits may leave in flight)
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
      const bool pok = ((unsigned)eb < (unsigned)p.B) & ((unsigned)(ey - 1) < (unsigned)p.H) & ((unsigned)(ex - 1) < (unsigned)p.W);
      const unsigned keep = pok ? 0xffffffffu : 0u;
      bf16_t* dst = p.y + (((long)eb * p.H + (ey - 1)) * p.W + (ex - 1)) * p.ysw + (long)cur.g * p.Cn + cur.c0 + cl0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (EPI == 0) {
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
        } else {
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
        issued += __builtin_amdgcn_ballot_w64(pok && cok[h]) != 0;
      }
      ex += c16; ey += a16;
      if (ex >= p.WP) { ex -= p.WP; ++ey; }
      if (ey >= p.HP) { ey -= p.HP; ++eb; }
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
      lgk0();
      __builtin_amdgcn_s_barrier();
      if (e_tid < 128) {
        const int ch = e_tid;
        if (cur.c0 + ch < p.Cn) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) { s += red[(w * 128 + ch) * 2]; q += red[(w * 128 + ch) * 2 + 1]; }
          float* dst = p.part + ((long)cur.tf * (p.G * p.Cn) + cur.g * p.Cn + cur.c0 + ch) * 2;
          *(float2*)dst = make_float2(s, q);
        }
      }
    }
    tolerate = __builtin_amdgcn_readfirstlane(issued);
    cur = nx;
    {
      const int t2 = tile + 2 * tile_step;
      nx = decode(t2 < tile_end ? t2 : tile, t2 < tile_end);
    }
  }
  wvm<0>();
}

template <int NHB, int EPI>
__global__ __launch_bounds__(512, 1) void conv3x3_flat_kernel(WFP p) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) flat_body<NHB, EPI, true>(p);
  else flat_body<NHB, EPI, false>(p);
}

int flat_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int NHB, int EPI>
int launch_flat(const WFP& p, hipStream_t st) {
  size_t sm = (size_t)NHB * p.HPIX * 64 + 4 * 8192 + 4 * 128 * 2 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_flat_kernel<NHB, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  long ntiles = (long)p.G * p.ntf * p.ntc;
  long nwg = (long)flat_cu_count() / 8 * 8;
  if (nwg < 8) nwg = 8;
  while (nwg > 8 && nwg / 8 > (ntiles + 7) / 8) nwg -= 8;
  hipLaunchKernelGGL((conv3x3_flat_kernel<NHB, EPI>), dim3((unsigned)nwg), dim3(512), sm, st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

// halo positions per tile of the flat kernel, and whether its halo buffers fit
static int flat_hpix(int W) { return (512 + 2 * (W + 3) + 63) / 64 * 64; }

// number of 512-position tiles of the flat space (= BatchNorm partial rows the kernel writes)
int y3d_conv3x3_flat_tiles(int B, int H, int W) { return cdiv((long)B * (H + 2) * (W + 2), 512); }

// does the flat kernel take this geometry (bf16 3x3 s1 p1; same operand conditions as conv3x3_wide3), and is it the better tiling?
int y3d_conv3x3_flat_ok(int B, int H, int W, int Cg, int Cn, int G) {
  if (Cg % 8 != 0 || Cg < 40 || Cn % 16 != 0 || W < 8 || W > 126 || H < 8) return 0;  // (H + 2 > 64 / (W + 2) + 1: one row wrap per walk step)
  if ((size_t)2 * flat_hpix(W) * 64 + 4 * 8192 + 4096 > 160 * 1024) return 0;
  if ((long)B * (H + 2) * (W + 2) >= (1L << 23)) return 0;  // float divisions of flat positions
  return 1;
}

int y3d_conv3x3_flat_launch(const void* x, long xsb, long xsh, long xsw, int B, int H, int W, int Cg, int Cn, int G, const void* w, int Ktot, void* y,
                            long ysw, float* part, int flip, const float* scale, const float* shift, int act, void* stream) {
  WFP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.part = part; p.scale = scale; p.shift = shift; p.act = act;
  p.xsb = xsb; p.xsh = xsh; p.xsw = xsw; p.ysw = ysw;
  p.B = B; p.H = H; p.W = W; p.Cg = Cg; p.Cn = Cn; p.G = G; p.Ktot = Ktot;
  p.HP = H + 2; p.WP = W + 2; p.IMG = p.HP * p.WP; p.HPIX = flat_hpix(W);
  p.inv_img = 1.0f / (float)p.IMG; p.inv_wp = 1.0f / (float)p.WP;
  p.ntf = y3d_conv3x3_flat_tiles(B, H, W); p.ntc = cdiv(Cn, 128); p.flip = flip;
  const unsigned long xb = ((unsigned long)(B - 1) * xsb + (unsigned long)(H - 1) * xsh + (unsigned long)(W - 1) * xsw + (unsigned long)G * Cg) * 2;
  const unsigned long wb = (unsigned long)G * Cn * Ktot * 2;
  Y3D_CHECK(xb < 0xfffffff0ul && wb < 0xfffffff0ul, "conv3x3_flat: operand larger than 4 GB");
  Y3D_CHECK(2 * xsw < (1L << 23) && Ktot < (1 << 22), "conv3x3_flat: pixel stride beyond the 24-bit address multiply");
  Y3D_CHECK(y3d_conv3x3_flat_ok(B, H, W, Cg, Cn, G), "conv3x3_flat: geometry not served");
  p.xbytes = (unsigned)xb; p.wbytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  const bool three = (size_t)3 * p.HPIX * 64 + 4 * 8192 + 4096 <= 160 * 1024;
  if (three) return scale ? launch_flat<3, 1>(p, st) : launch_flat<3, 0>(p, st);
  return scale ? launch_flat<2, 1>(p, st) : launch_flat<2, 0>(p, st);
}
