// Weight gradient of a 1x1 stride-1 convolution as a streaming split-K GEMM (bf16, fp32 accumulate):
//
//     slab[split][co][ci] = sum over the split's pixels m of dy[m][co] * x[m][ci]
//
// Reference: autograd of nn/modules/conv.py:120-122 for k = 1 (C2f / SCDown / PSA / SPPF cv1, cv2).  The generic kernel
// (conv_gemm.hip, conv_wgrad_kernel) stages each 64-pixel step through registers, one step ahead: its pixel loop is bound by the
// latency of that one prefetch (35 us per launch on the S-3D body, 125-290 TFLOP/s).  Same tiles, same split-K plan, same slab format
// here (wgrad_reduce_kernel folds the slabs), but the two operand tiles of a step (64 pixels x WD output channels, 64 pixels x WX
// input channels) arrive by LDS-DMA into a ring of four stages, three in flight, with one counted s_waitcnt + raw barrier per stage.
// Both operands are pixel-major (NHWC) and the reduction runs over pixels: the MFMA fragments are fetched with the transposed LDS
// read (ds_read_b64_tr_b16) out of rows padded by 32 bytes (the DMA writes LDS lane-linearly, so the padding chunks are lanes with
// an out-of-range source; out-of-range pixels and channels read zeros the same way).  The transposed reads are inline asm: through
// the builtin hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of them while an LDS-DMA is outstanding, which would serialise the ring.
// The pixel <-> MFMA-k assignment is conv_wgrad_kernel's (k = 8g + j  <->  row 16 (j >> 2) + 4g + (j & 3) of a 32-pixel sub-step) and so
// is the order of the accumulation: the slabs are bit-identical to the generic kernel's.
#include "common.h"

namespace {

struct Wg1P {
  const bf16_t* x;
  const bf16_t* dy;
  float* slab;  // [nsplit][Cn][Cg]
  int xsw, dsw;
  int M, Cg, Cn, chunk_px;
  unsigned xbytes, dbytes;
};

template <int N> __device__ __forceinline__ void wg1_wvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ s16x4_t wg1_tr(unsigned a) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
// every transposed read issued so far has landed; the fragments are in/out operands so that their uses stay behind the wait
template <int NF>
__device__ __forceinline__ void wg1_wait(s16x4_t (&f)[NF][2]) {
  if constexpr (NF == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]));
  else if constexpr (NF == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]));
  else if constexpr (NF == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]));
  else {
    static_assert(NF == 6, "2, 3, 4 or 6 fragments");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]),
                 "+v"(f[4][0]), "+v"(f[4][1]), "+v"(f[5][0]), "+v"(f[5][1]));
  }
}

// WD x WX output tile (output channels x input channels), 512 threads = WDW x (8 / WDW) waves
template <int WD, int WX, int WDW>
__global__ __launch_bounds__(512) void wgrad1x1_stream_kernel(Wg1P p) {
  constexpr int NS = 4, BPK = 64;
  constexpr int WXW = 8 / WDW;
  constexpr int TA = WD / WDW / 16, TB = WX / WXW / 16;
  static_assert(TA >= 1 && TB >= 1, "wave tile");
  constexpr int CD = WD / 8 + 2, CX = WX / 8 + 2;      // 16-byte chunks per LDS row (two of padding)
  constexpr int PD = CD * 16, PX = CX * 16;            // row pitches: 8 x odd dwords
  constexpr int ID = CD, IX = CX;                      // DMA instructions per tile: 64 rows x C chunks / 64 lanes
  constexpr int NI = (ID + IX + 7) / 8;                // per wave per stage (the last ones may be dummies)
  constexpr int STAGE = BPK * (PD + PX);
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NS][dy tile | x tile] + 1 KB dump for the dummy instructions
  char* dump = smem + NS * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WXW, wj = wave % WXW;
  const int split = blockIdx.z;
  const int j0 = blockIdx.x * WX, i0 = blockIdx.y * WD;
  const int mbeg = split * p.chunk_px;
  const int mend = min(p.M, mbeg + p.chunk_px);
  const int nsteps = mend > mbeg ? (mend - mbeg + BPK - 1) / BPK : 0;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;

  // instruction ii of a stage: ii < ID -> rows of the dy tile, ii < ID + IX -> rows of the x tile, else a dummy; wave w issues ii = w, w + 8, ...
  int is_ = 0, i_slot = 0;
  auto issue = [&]() {
    const bool live = is_ < nsteps;
    const int m0 = mbeg + is_ * BPK;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int ii = wave + 8 * n;  // uniform
      if (ii < ID) {
        const int q = ii * 64 + lane, row = q / CD, col = q - row * CD;
        const int m = m0 + row, c = i0 + col * 8;
        const bool ok = live & (m < mend) & (col < WD / 8) & (c < p.Cn);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(smem + i_slot * STAGE + ii * 1024), 16,
                                                 ok ? (unsigned)(m * p.dsw + c) * 2u : OOB, 0, 0, 0);
      } else if (ii < ID + IX) {
        const int q = (ii - ID) * 64 + lane, row = q / CX, col = q - row * CX;
        const int m = m0 + row, c = j0 + col * 8;
        const bool ok = live & (m < mend) & (col < WX / 8) & (c < p.Cg);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(smem + i_slot * STAGE + BPK * PD + (ii - ID) * 1024), 16,
                                                 ok ? (unsigned)(m * p.xsw + c) * 2u : OOB, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dump, 16, OOB, 0, 0, 0);
      }
    }
    ++is_;
    if (++i_slot == NS) i_slot = 0;
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue();

  f32x4_t acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // transposed-read address of this lane inside a 32-pixel sub-step: rows 4g + (li >> 2) (+16 for the second half), columns 4 (li & 3)
  const int grp = lane >> 4, li = lane & 15;
  const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned offD = (4 * grp + (li >> 2)) * PD + (wi * (WD / WDW) + 4 * (li & 3)) * 2;
  const unsigned offX = BPK * PD + (4 * grp + (li >> 2)) * PX + (wj * (WX / WXW) + 4 * (li & 3)) * 2;
  int c_slot = 0;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    wg1_wvm<NI * (NS - 2)>();
    __builtin_amdgcn_s_barrier();
    issue();
    const unsigned tb = sbase + c_slot * STAGE;
#pragma unroll
    for (int k0 = 0; k0 < BPK; k0 += 32) {
      s16x4_t f[TA + TB][2];
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        f[a][0] = wg1_tr(tb + offD + k0 * PD + a * 32);
        f[a][1] = wg1_tr(tb + offD + (k0 + 16) * PD + a * 32);
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        f[TA + b][0] = wg1_tr(tb + offX + k0 * PX + b * 32);
        f[TA + b][1] = wg1_tr(tb + offX + (k0 + 16) * PX + b * 32);
      }
      wg1_wait<TA + TB>(f);
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      bf16x8_t fr[TA + TB];
#pragma unroll
      for (int i = 0; i < TA + TB; ++i) fr[i] = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(f[i][0], f[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[a], fr[TA + b], acc[a][b], 0, 0, 0);
    }
    if (++c_slot == NS) c_slot = 0;
  }
  wg1_wvm<0>();  // trailing dummy rounds
  // D[i][j]: row i = co = 4 (lane >> 4) + reg, column j = ci = lane & 15
  float* slab = p.slab + (long)split * p.Cn * p.Cg;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int k = j0 + wj * (WX / WXW) + b * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = i0 + wi * (WD / WDW) + a * 16 + grp * 4 + r;
        if (co < p.Cn && k < p.Cg) slab[(long)co * p.Cg + k] = acc[a][b][r];
      }
    }
}

template <int WD, int WX, int WDW>
void wg1_launch(const Wg1P& p, dim3 grid, hipStream_t st) {
  constexpr int CD = WD / 8 + 2, CX = WX / 8 + 2;
  const size_t lds = (size_t)4 * 64 * (CD + CX) * 16 + 1024;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad1x1_stream_kernel<WD, WX, WDW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((wgrad1x1_stream_kernel<WD, WX, WDW>), grid, dim3(512), lds, st, p);
}

}  // namespace

extern "C" int y3d_get_stream1x1(void);

// tile widths as conv_gemm.hip's wgrad_tile_w: 64 or 128 here (narrower operands stay on the generic kernel)
int y3d_wgrad1x1_stream_ok(int dtype, long M, int Cg, int Cn, long xsw, long dsw) {
  if (!y3d_get_stream1x1() || dtype != Y3D_BF16 || Cg <= 32 || Cn <= 32 || Cg % 8 || Cn % 8) return 0;
  if ((M * xsw + Cg) * 2 >= (1L << 32) - 64 || (M * dsw + Cn) * 2 >= (1L << 32) - 64) return 0;
  return 1;
}

int y3d_wgrad1x1_stream_launch(const void* x, long xsw, const void* dy, long dsw, long M, int Cg, int Cn, float* slab, int nsplit, int chunk_px,
                               void* stream) {
  Wg1P p;
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.slab = slab; p.xsw = (int)xsw; p.dsw = (int)dsw;
  p.M = (int)M; p.Cg = Cg; p.Cn = Cn; p.chunk_px = chunk_px;
  p.xbytes = (unsigned)(((M - 1) * xsw + Cg) * 2);
  p.dbytes = (unsigned)(((M - 1) * dsw + Cn) * 2);
  const int wd = Cn <= 64 ? 64 : 128, wx = Cg <= 64 ? 64 : 128;
  dim3 grid(cdiv(Cg, wx), cdiv(Cn, wd), nsplit);
  hipStream_t st = (hipStream_t)stream;
  if (wd == 128 && wx == 128) wg1_launch<128, 128, 2>(p, grid, st);
  else if (wd == 64 && wx == 128) wg1_launch<64, 128, 2>(p, grid, st);
  else if (wd == 128 && wx == 64) wg1_launch<128, 64, 4>(p, grid, st);
  else wg1_launch<64, 64, 2>(p, grid, st);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}
