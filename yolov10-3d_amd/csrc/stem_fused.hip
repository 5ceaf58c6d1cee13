// Stem of the eval forward - Conv(3, c, 3, 2) + folded BatchNorm + SiLU (nn/tasks.py yaml row 0, nn/modules/conv.py:120-122 with
// running statistics) - in ONE pass over the image: HBM bound (read the image once, write the activation once).
//
// The training path keeps the two-step form (y3d_stem_im2col + a dense K = 32 conv: its [B][Ho][Wo][32] column tensor is also the
// operand of the weight gradient).  In eval that form moved the column tensor out and back in (2 x 210 MB for a batch of 32 at 640x640,
// 297 us for im2col + GEMM); here a wave gathers the 27 window samples of 16 output pixels straight into the B operand of one
// v_mfma_f32_16x16x32_bf16 (K = 32 = 27 taps x channels + 5 zeros: the whole reduction is ONE instruction per 16 pixels x 16 channels),
// the A operand - the weights of 16 output channels - stays in registers for the life of the wave, and the epilogue applies
// scale / shift / SiLU and stores 16-byte runs of 8 channels (two 16-channel blocks exchanged across lane rows with
// v_permlane16_swap), 1 KB contiguous per store instruction when the map has 32 channels.
//
// Arithmetic is that of the two-step form: samples rounded to bf16 (RNE), fp32 accumulation, one rounding of the activation.
// Algorithmic bytes per output pixel: 4 * 3 * 4 (the fp32 samples of a 2x2 input patch) + 2 * Cout.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2_t f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

struct StemP {
  const void* x;        // IN 0: (B, 3, H, W) fp32;  1: (B, 3, H, W) uint8;  2: (B, H, W, 3) uint8   (uint8: value / 255, exact division)
  const float* wcol;    // [Cout][32] fp32, column (r*3 + q)*3 + ci, columns 27..31 zero
  const float* scale;   // [Cout] folded BatchNorm
  const float* shift;   // [Cout]
  bf16_t* y;            // (B, Ho, Wo) pixels x Cout channels, pixel pitch ysw elements
  long ysw;
  bf16_t* xcol;         // TRAIN: the column tensor [B * Ho * Wo][32] (operand of the weight gradient, y3d_conv2d_bwd_weight with a 1x1 kernel)
  float* part;          // TRAIN: BatchNorm partials [waves][Cout][2] (sum, sum of squares of the bf16-rounded outputs)
  int B, H, W, Ho, Wo, Cout, act;
  int nxb;              // 64-pixel column blocks per output row
};

// NB16 = Cout / 16 channel blocks; IN = input format; TRAIN: raw conv output + BatchNorm partials + column tensor instead of the
// folded-BatchNorm activation
template <int NB16, int IN, bool TRAIN>
__global__ __launch_bounds__(256) void stem_fused_kernel(StemP p) {
  __shared__ float lut[256];  // u8 -> u8 / 255 with the reference's fp32 division (data/datasets/kitti.py:204-205)
  if (IN != 0) {
    lut[threadIdx.x] = (float)threadIdx.x / 255.f;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 15, kg = lane >> 4;  // this lane's pixel inside a 16-pixel block and its 8 reduction columns 8 kg .. 8 kg + 7
  // window coordinates of the lane's columns: k = (r * 3 + q) * 3 + ci; pc = the column's channel offset inside one image
  int rr[8], qq[8], pc[8];
  bool kok[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kg * 8 + j, tap = k / 3, ci = k - tap * 3;
    rr[j] = tap / 3;
    qq[j] = tap - rr[j] * 3;
    kok[j] = k < 27;
    pc[j] = IN == 2 ? ci : ci * p.H * p.W;
  }
  // A operand: row = channel (lane & 15) of block nb, the lane's 8 columns; folded BatchNorm of the 4 channels the lane gets back
  bf16x8_t wa[NB16];
  float sc[NB16][4], sh[NB16][4];  // eval: folded BatchNorm;  TRAIN: running sum / sum of squares of the lane's 4 channels
#pragma unroll
  for (int nb = 0; nb < NB16; ++nb) {
    const float* wr = p.wcol + (nb * 16 + px) * 32 + kg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) wa[nb][j] = (__bf16)wr[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sc[nb][i] = TRAIN ? 0.f : p.scale[nb * 16 + kg * 4 + i];
      sh[nb][i] = TRAIN ? 0.f : p.shift[nb * 16 + kg * 4 + i];
    }
  }
  const long items = (long)p.B * p.Ho * p.nxb;
  const long nwaves = (long)gridDim.x * 4;
  for (long it = (long)blockIdx.x * 4 + wave; it < items; it += nwaves) {
    const int xb = (int)(it % p.nxb);
    long t = it / p.nxb;
    const int oy = (int)(t % p.Ho), b = (int)(t / p.Ho);
    const int iy0 = 2 * oy - 1;
    const long ib = (long)b * 3 * p.H * p.W;
    float v[4][8];
    // The kernel is VALU-bound, not HBM-bound, when every sample pays for its own address and padding test (first form: 228 us for the
    // batch of 32, 1.6 TB/s).  Interior items - all of them but the top row and partial / odd-sized right edges - compute 8 addresses
    // per lane (block 0) and reach blocks 1..3 through the instruction's immediate offset; the one sample that can fall into the left
    // padding (column block 0, pixel 0, q = 0: a valid address, one element before the row) is zeroed under a wave-uniform branch.
    const bool fast = iy0 >= 0 && iy0 + 2 < p.H && xb * 128 + 128 <= p.W && xb * 64 + 64 <= p.Wo;
    if (fast) {
      const int ix0 = 2 * (xb * 64 + px) - 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long a = ib + (IN == 2 ? ((long)(iy0 + rr[j]) * p.W + ix0 + qq[j]) * 3 + pc[j] : (long)pc[j] + (long)(iy0 + rr[j]) * p.W + ix0 + qq[j]);
        const long ac = kok[j] ? a : ib;  // columns 27..31: any valid address, zero weight AND zero sample below
        if (IN == 0) {
          const float* q = (const float*)p.x + ac;
#pragma unroll
          for (int s = 0; s < 4; ++s) v[s][j] = q[s * 32];
        } else {
          const unsigned char* q = (const unsigned char*)p.x + ac;
#pragma unroll
          for (int s = 0; s < 4; ++s) v[s][j] = lut[q[s * (IN == 2 ? 96 : 32)]];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!kok[j]) {
#pragma unroll
          for (int s = 0; s < 4; ++s) v[s][j] = 0.f;
        }
      }
      if (xb == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[0][j] = (px == 0 && qq[j] == 0) ? 0.f : v[0][j];
      }
    } else {
      // border items: every sample from a clamped (always valid) address, zeroed afterwards when it lies in the padding
      int ro[8];
      bool rok[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int iy = iy0 + rr[j];
        rok[j] = kok[j] && (unsigned)iy < (unsigned)p.H;
        const int iyc = min(max(iy, 0), p.H - 1);
        ro[j] = IN == 2 ? iyc * p.W * 3 + pc[j] : pc[j] + iyc * p.W;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ox = xb * 64 + s * 16 + px;
        const int ix0 = 2 * ox - 1;
        const bool pv = ox < p.Wo;  // pixels past the row end: zero samples -> zero outputs (they must not reach the BatchNorm partials)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ix = ix0 + qq[j];
          const int ixc = min(max(ix, 0), p.W - 1);
          const long a = ib + ro[j] + (IN == 2 ? ixc * 3 : ixc);
          const float raw = IN == 0 ? ((const float*)p.x)[a] : lut[((const unsigned char*)p.x)[a]];
          v[s][j] = (pv && rok[j] && (unsigned)ix < (unsigned)p.W) ? raw : 0.f;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ox0 = xb * 64 + s * 16;
      if (ox0 >= p.Wo) break;  // wave-uniform
      f32x8_t xf;
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = v[s][j];
      const bf16x8_t xb8 = __builtin_convertvector(xf, bf16x8_t);  // v_cvt_pk_bf16_f32 pairs
      const bool pok = ox0 + px < p.Wo;
      const long pix = ((long)b * p.Ho + oy) * p.Wo + ox0 + px;
      if (TRAIN && pok) *(uint4*)(p.xcol + pix * 32 + kg * 8) = __builtin_bit_cast(uint4, xb8);  // the B operand IS the column tensor
      // D[channel][pixel]: lane -> pixel px, channels 16 nb + 4 kg .. + 3
      unsigned pk[NB16][2];
#pragma unroll
      for (int nb = 0; nb < NB16; ++nb) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb], xb8, acc, 0, 0, 0);
        float z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (TRAIN) {
            z[i] = acc[i];
          } else {
            const float u = acc[i] * sc[nb][i] + sh[nb][i];
            z[i] = p.act ? silu_f(u) : u;
          }
        }
        pk[nb][0] = pack2(z[0], z[1]);
        pk[nb][1] = pack2(z[2], z[3]);
        if (TRAIN) {  // statistics of the values as stored (bf16), as every conv epilogue; pixels past the row end are exact zeros
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float r = __uint_as_float(i & 1 ? pk[nb][i >> 1] & 0xffff0000u : pk[nb][i >> 1] << 16);
            sc[nb][i] += r;
            sh[nb][i] += r * r;
          }
        }
      }
      bf16_t* yp = p.y + pix * p.ysw;
      // pairs of 16-channel blocks: after the row exchange lane row g holds 8 consecutive channels at 32 t + (g & 1) * 16 + (g >> 1) * 8
#pragma unroll
      for (int t2 = 0; t2 + 1 < NB16; t2 += 2) {
        auto e0 = __builtin_amdgcn_permlane16_swap(pk[t2][0], pk[t2 + 1][0], false, false);
        auto e1 = __builtin_amdgcn_permlane16_swap(pk[t2][1], pk[t2 + 1][1], false, false);
        if (pok) *(uint4*)(yp + t2 * 16 + (kg & 1) * 16 + (kg >> 1) * 8) = make_uint4(e0[0], e1[0], e0[1], e1[1]);
      }
      if (NB16 & 1) {
        if (pok) *(uint2*)(yp + (NB16 - 1) * 16 + kg * 4) = make_uint2(pk[NB16 - 1][0], pk[NB16 - 1][1]);
      }
    }
  }
  if (TRAIN) {  // one partial row per wave (zeros when the wave had no item)
    float* dst = p.part + ((long)blockIdx.x * 4 + wave) * p.Cout * 2;
#pragma unroll
    for (int nb = 0; nb < NB16; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float su = wave_xor_sum16(sc[nb][i]), sq = wave_xor_sum16(sh[nb][i]);
        if (px == 0) *(float2*)(dst + (nb * 16 + kg * 4 + i) * 2) = make_float2(su, sq);
      }
  }
}

int stem_grid(long items) {
  long g = (items + 3) / 4;
  return (int)(g > 2048 ? 2048 : g);
}

template <int NB16, bool TRAIN>
void stem_launch(const StemP& p, int in_mode, hipStream_t st) {
  const unsigned g = (unsigned)stem_grid((long)p.B * p.Ho * p.nxb);
  if (in_mode == 0) hipLaunchKernelGGL((stem_fused_kernel<NB16, 0, TRAIN>), dim3(g), dim3(256), 0, st, p);
  else if (in_mode == 1) hipLaunchKernelGGL((stem_fused_kernel<NB16, 1, TRAIN>), dim3(g), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((stem_fused_kernel<NB16, 2, TRAIN>), dim3(g), dim3(256), 0, st, p);
}

template <bool TRAIN>
int stem_dispatch(StemP& p, int in_mode, void* stream) {
  p.Ho = (p.H - 1) / 2 + 1; p.Wo = (p.W - 1) / 2 + 1;
  p.nxb = cdiv(p.Wo, 64);
  hipStream_t st = (hipStream_t)stream;
  switch (p.Cout / 16) {
    case 1: stem_launch<1, TRAIN>(p, in_mode, st); break;
    case 2: stem_launch<2, TRAIN>(p, in_mode, st); break;
    case 3: stem_launch<3, TRAIN>(p, in_mode, st); break;
    case 4: stem_launch<4, TRAIN>(p, in_mode, st); break;
    default: stem_launch<5, TRAIN>(p, in_mode, st); break;
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // namespace

extern "C" {

#define STEM_ARG_CHECKS(name)                                                                                                                   \
  Y3D_CHECK(in_mode >= 0 && in_mode <= 2, name ": in_mode 0 (fp32 NCHW), 1 (uint8 NCHW) or 2 (uint8 NHWC)");                                   \
  Y3D_CHECK(B >= 1 && H >= 1 && W >= 1 && Cout >= 16 && Cout <= 80 && Cout % 16 == 0, name ": Cout = %d must be 16, 32, 48, 64 or 80", Cout); \
  Y3D_CHECK(ysw >= Cout && ysw % 8 == 0 && ((uintptr_t)y & 15) == 0, name ": output pitch / alignment");                                       \
  Y3D_CHECK((long)B * 3 * H * W < (1L << 31), name ": image batch beyond 32-bit element offsets")

int y3d_stem_conv_eval(const void* x, int in_mode, const float* wcol, const float* scale, const float* shift, int act, void* y, int64_t ysw,
                       int B, int H, int W, int Cout, void* stream) {
  Y3D_CHECK(x && wcol && scale && shift && y, "stem_conv_eval: null argument");
  STEM_ARG_CHECKS("stem_conv_eval");
  StemP p;
  p.x = x; p.wcol = wcol; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y; p.ysw = ysw; p.xcol = nullptr; p.part = nullptr;
  p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.act = act;
  return stem_dispatch<false>(p, in_mode, stream);
}

int y3d_stem_conv_train_rows(int B, int H, int W) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  return stem_grid((long)B * Ho * cdiv(Wo, 64)) * 4;
}

int y3d_stem_conv_train(const void* x, int in_mode, const float* wcol, void* y, int64_t ysw, void* xcol, float* part, int B, int H, int W,
                        int Cout, void* stream) {
  Y3D_CHECK(x && wcol && y && xcol && part, "stem_conv_train: null argument");
  STEM_ARG_CHECKS("stem_conv_train");
  Y3D_CHECK(((uintptr_t)xcol & 15) == 0, "stem_conv_train: column tensor alignment");
  StemP p;
  p.x = x; p.wcol = wcol; p.scale = nullptr; p.shift = nullptr; p.y = (bf16_t*)y; p.ysw = ysw; p.xcol = (bf16_t*)xcol; p.part = part;
  p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.act = 0;
  return stem_dispatch<true>(p, in_mode, stream);
}

}  // extern "C"
