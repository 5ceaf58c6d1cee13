// BatchNorm backward of the second head layer WITHOUT the 839 MB gradient tensor in between (bf16).
//
// Training head, per level (modules.v10Detect3d.forward_train_fused): y2 = grouped 3x3 conv (16 branches x `cin` channels), out =
// proj_b(SiLU(BN(y2)))  (proj_group.hip: projg_fwd_bn).  Backward so far: projg_bwd_data wrote dz2 = dout . W (839 MB at P3),
// bn_act_bwd_reduce read y2 + dz2, bn_act_bwd_apply read y2 + dz2 and wrote dy2: 5.0 GB of traffic, 1.3 ms per step at 4.6 TB/s.
// dz2[px][ch] = sum_o dout[px][o] * W[o][ch] has K = cout <= 24: ONE 16x16x32 MFMA per 16 pixels x 16 channels.  Both BatchNorm
// passes recompute it from the 31 MB `dout` instead of reading it:
//   MODE 0 (reduce): partial sums of g = dz2 * act'(u) and g * xhat per channel  (same partial-slab format as bn_act.hip)
//   MODE 1 (apply):  dy2 = scale * (g - mean(g) - xhat * mean(g * xhat))
// 2.5 GB instead of 5.0 GB; dz2 is also never rounded to bf16.
// One workgroup = one branch x a run of pixels, 4 waves x 16 pixels per step.  The MFMA result (lane = 4 pixels x 1 channel) goes
// through a per-wave LDS tile to the row layout (lane = 1 pixel x 8 channels) in which y2 is loaded and dy2 stored as coalesced
// 16-byte chunks and the per-channel constants / sums are lane-stationary.
#include "common.h"

namespace {

constexpr int PB_MAXB = 16;

struct PBP {
  const bf16_t* y;     // pre-BatchNorm conv output, pixel stride ysw
  const bf16_t* dout;  // gradient of the projected map, pixel stride dsw
  bf16_t* dy;          // MODE 1: gradient wrt y, pixel stride dysw
  float* part;         // MODE 0: [gridDim.x][C][2]
  const float* w[PB_MAXB];  // [cout][cin] fp32
  int xoff[PB_MAXB], ooff[PB_MAXB], cout[PB_MAXB];
  const float *scale, *shift, *mean, *invstd, *mg, *mgx;
  long ysw, dsw, dysw, P;
  int cin, C, act, px_per_block;
};

template <int MODE, int NT>
__global__ __launch_bounds__(256) void projg_bn_bwd_kernel(PBP p) {
  constexpr int CIN = NT * 16, LDW = CIN + 4;  // LDS row pitch (floats): +4 keeps rows 16-byte aligned and spreads the four row groups over the banks
  constexpr int CH = CIN / 8;                  // 16-byte chunks per pixel row of this branch
  constexpr int PPP = 64 / CH;                 // pixels per pass of a wave in the row layout
  constexpr int NPASS = 16 / PPP;
  static_assert(NPASS >= 1 && 64 % CH == 0 && 16 % PPP == 0, "cin must be 32, 64 or 128");
  extern __shared__ __attribute__((aligned(16))) float psm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, lp = lane & 15;
  const int br = blockIdx.y, cout = p.cout[br];
  float* tile = psm + wave * 16 * LDW;

  // B operand: W[o][ch], lane (k = 8q + j -> o, column lp -> channel of tile t); rows o >= cout are zero
  bf16x8_t wf[NT];
  {
    const float* w = p.w[br];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = 8 * q + j;
        wf[t][j] = (__bf16)(o < cout ? w[o * CIN + t * 16 + lp] : 0.f);
      }
  }
  // row layout: this lane owns chunk cc (8 channels) of pixel row (lane / CH) of every pass
  const int cc = lane % CH, pr = lane / CH;
  const int c0 = p.xoff[br] + cc * 8;
  float sc[8], sf[8], mu[8], is[8], m1[8], m2[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = p.scale[c0 + j]; sf[j] = p.shift[c0 + j]; mu[j] = p.mean[c0 + j]; is[j] = p.invstd[c0 + j];
    if (MODE == 1) { m1[j] = p.mg[c0 + j]; m2[j] = p.mgx[c0 + j]; }
    s1[j] = 0.f; s2[j] = 0.f;
  }
  const long pbeg = (long)blockIdx.x * p.px_per_block;
  const long pend = pbeg + p.px_per_block < p.P ? pbeg + p.px_per_block : p.P;
  const bf16_t* dop = p.dout + p.ooff[br];
  for (long pb = pbeg; pb < pend; pb += 64) {  // uniform trip count: the barriers below are taken by all four waves
    const long p0 = pb + wave * 16;
    // y of this wave's 16 pixels, row layout, issued first (the longest latency)
    uint4 yv[NPASS];
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      const long px = p0 + s * PPP + pr;
      yv[s] = px < pend ? *(const uint4*)(p.y + px * p.ysw + c0) : make_uint4(0, 0, 0, 0);
    }
    // A operand: dout, lane (row lp -> pixel, k = 8q + j -> o)
    bf16x8_t a;
    {
      const long px = p0 + lp;
      const bool pok = px < pend;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = 8 * q + j;
        unsigned short raw = 0;
        if (pok && o < cout) raw = dop[px * p.dsw + o];
        a[j] = __builtin_bit_cast(__bf16, raw);
      }
    }
    // dz tile -> LDS: D[row = 4q + r -> pixel][column lp -> channel]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[t], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(4 * q + r) * LDW + t * 16 + lp] = d[r];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      const int pl = s * PPP + pr;
      const long px = p0 + pl;
      const float4 z0 = *(const float4*)(tile + pl * LDW + cc * 8), z1 = *(const float4*)(tile + pl * LDW + cc * 8 + 4);
      const float dz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
      float v[8], o[8];
      Chunk<bf16_t>::unpack(yv[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float g = dz[j];
        if (p.act) g *= silu_grad_f(v[j] * sc[j] + sf[j]);
        const float xh = (v[j] - mu[j]) * is[j];
        if (MODE == 0) { s1[j] += g; s2[j] += g * (v[j] - mu[j]) * is[j]; }
        else o[j] = sc[j] * (g - m1[j] - xh * m2[j]);
      }
      if (MODE == 1 && px < pend) *(uint4*)(p.dy + px * p.dysw + c0) = Chunk<bf16_t>::pack(o);
    }
    __syncthreads();  // the tile is rewritten by the next step
  }
  if (MODE == 0) {
    // fold the 4 waves x PPP pixel rows that share a chunk; rows past `pend` contributed exact zeros (zero dout rows)
    float* red = psm;  // [4 * PPP][CIN][2]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[((wave * PPP + pr) * CIN + cc * 8 + j) * 2 + 0] = s1[j];
      red[((wave * PPP + pr) * CIN + cc * 8 + j) * 2 + 1] = s2[j];
    }
    __syncthreads();
    if (tid < CIN) {
      float a0 = 0.f, a1 = 0.f;
      for (int r = 0; r < 4 * PPP; ++r) { a0 += red[(r * CIN + tid) * 2]; a1 += red[(r * CIN + tid) * 2 + 1]; }
      float* dst = p.part + ((long)blockIdx.x * p.C + p.xoff[br] + tid) * 2;
      dst[0] = a0;
      dst[1] = a1;
    }
  }
}

// Forward of the same stage: out[px][ooff + o] = sum_ch act[px][ch] * W[o][ch] + b[o], act = SiLU(BN(y)) formed in the row layout
// (coalesced 16-byte loads, lane-stationary scale / shift), handed to the matrix cores through a per-wave bf16 LDS tile:
// K = cin (2-4 MFMA k-steps), N = cout padded to 16 / 32.  The VALU form (proj_group.hip: projg_fwd_kernel) spent 24 x cin
// multiply-adds per pixel and branch on outputs that mostly do not exist (cout = 1..3 for 12 of the 16 branches): 0.57 ms at P3.
struct PFP {
  const bf16_t* y;
  bf16_t* out;
  const float* w[PB_MAXB];
  const float* b[PB_MAXB];
  int xoff[PB_MAXB], ooff[PB_MAXB], cout[PB_MAXB];
  const float *scale, *shift;
  long ysw, osw, P;
  int act, px_per_block;
};

template <int NT>
__global__ __launch_bounds__(256) void projg_fwd_bn_mfma_kernel(PFP p) {
  constexpr int CIN = NT * 16, KS = CIN / 32;
  constexpr int LDB = CIN * 2 + 16;  // LDS row pitch in BYTES (bf16 rows + one 16-byte slot: rows land on different bank groups)
  constexpr int CH = CIN / 8, PPP = 64 / CH, NPASS = 16 / PPP;
  extern __shared__ __attribute__((aligned(16))) float psm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, lp = lane & 15;
  const int br = blockIdx.y, cout = p.cout[br];
  char* tile = (char*)psm + wave * 16 * LDB;
  // B operand: W^T, lane (k = 8q + j -> channel 32 ks + 8q + j, column lp -> output 16 n + lp)
  bf16x8_t wf[2][KS];
  float bias[2];
  {
    const float* w = p.w[br];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int o = 16 * n + lp;
      bias[n] = o < cout ? p.b[br][o] : 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[n][ks][j] = (__bf16)(o < cout ? w[o * CIN + ks * 32 + 8 * q + j] : 0.f);
    }
  }
  const int cc = lane % CH, pr = lane / CH;
  const int c0 = p.xoff[br] + cc * 8;
  float sc[8], sf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = p.scale[c0 + j]; sf[j] = p.shift[c0 + j]; }
  const long pbeg = (long)blockIdx.x * p.px_per_block;
  const long pend = pbeg + p.px_per_block < p.P ? pbeg + p.px_per_block : p.P;
  bf16_t* op = p.out + p.ooff[br];
  const bool two = cout > 16;  // uniform per workgroup
  for (long pb = pbeg; pb < pend; pb += 64) {
    const long p0 = pb + wave * 16;
    uint4 yv[NPASS];
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      const long px = p0 + s * PPP + pr;
      yv[s] = px < pend ? *(const uint4*)(p.y + px * p.ysw + c0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      float v[8];
      Chunk<bf16_t>::unpack(yv[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = v[j] * sc[j] + sf[j];
        v[j] = p.act ? silu_f(u) : u;
      }
      *(uint4*)(tile + (s * PPP + pr) * LDB + cc * 16) = Chunk<bf16_t>::pack(v);  // rounded as the materialised activation would be
    }
    __syncthreads();
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *(const uint4*)(tile + lp * LDB + (ks * 4 + q) * 16));
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[0][ks], acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[1][ks], acc1, 0, 0, 0);
    }
    // D[row = 4q + r -> pixel][column lp -> output]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long px = p0 + 4 * q + r;
      if (px < pend) {
        if (lp < cout) op[px * p.osw + lp] = f2bf(acc0[r] + bias[0]);
        if (two && 16 + lp < cout) op[px * p.osw + 16 + lp] = f2bf(acc1[r] + bias[1]);
      }
    }
    __syncthreads();
  }
}

// Weight / bias gradient of the projections: dW[o][ch] = sum_px dout[px][o] * act[px][ch], db[o] = sum_px dout[px][o] - a contraction
// over PIXELS.  Per wave and step of 32 pixels: act = SiLU(BN(y)) is formed in the row layout and parked in a bf16 LDS tile; the B
// operand (k = pixel, column = channel) is read back with the transposed LDS read (ds_read_tr16_b64: two 4 x 16 blocks per
// fragment), the A operand is dout^T (row = output, k = pixel: eight 2-byte loads per lane, only the cout <= 24 live rows load).
// Both use the same pixel <-> k permutation: k = 8g + j  <->  row 16 (j >> 2) + 4g + (j & 3) of the step.
struct PWP {
  const bf16_t* y;
  const bf16_t* dout;
  float* slab;   // [gridDim.x][ctot][cin]
  float* bslab;  // [gridDim.x][ctot]
  float* dw[PB_MAXB];
  float* db[PB_MAXB];
  int xoff[PB_MAXB], ooff[PB_MAXB], cout[PB_MAXB];
  const float *scale, *shift;
  long ysw, dsw, P;
  int act, px_per_block, ctot;
};

template <int NT>
__global__ __launch_bounds__(256) void projg_bwd_weight_bn_mfma_kernel(PWP p) {
  constexpr int CIN = NT * 16;
  constexpr int LDB = CIN * 2 + 32;  // tile row pitch in bytes (8 x odd dwords: the 8 rows of a transposed read cover all banks once)
  constexpr int CH = CIN / 8, PPP = 64 / CH, NPASS = 32 / PPP;
  extern __shared__ __attribute__((aligned(16))) float psm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lp = lane & 15;
  const int br = blockIdx.y, cout = p.cout[br];
  const bool two = cout > 16;  // uniform
  char* tile = (char*)psm + wave * 32 * LDB;
  const int cc = lane % CH, pr = lane / CH;
  const int c0 = p.xoff[br] + cc * 8;
  float sc[8], sf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = p.scale[c0 + j]; sf[j] = p.shift[c0 + j]; }
  f32x4_t acc[2][NT];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[n][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  const long pbeg = (long)blockIdx.x * p.px_per_block;
  const long pend = pbeg + p.px_per_block < p.P ? pbeg + p.px_per_block : p.P;
  const bf16_t* dop = p.dout + p.ooff[br];
  // transposed-read address of this lane inside a 4-row block: lane 4q' + p' of a 16-lane group supplies row q', columns 4p' .. 4p'+3
  const int trow = 4 * g + (lp >> 2), tcol = (lp & 3) * 4;
  for (long pb = pbeg; pb < pend; pb += 128) {
    const long p0 = pb + wave * 32;
    uint4 yv[NPASS];
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      const long px = p0 + s * PPP + pr;
      yv[s] = px < pend ? *(const uint4*)(p.y + px * p.ysw + c0) : make_uint4(0, 0, 0, 0);
    }
    // A operands: dout^T rows 0..15 (and 16..31 for the 24-output branch)
    bf16x8_t a[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long px = p0 + 16 * (j >> 2) + 4 * g + (j & 3);
        const int o = 16 * n + lp;
        unsigned short raw = 0;
        if ((n == 0 || two) && o < cout && px < pend) raw = dop[px * p.dsw + o];
        a[n][j] = __builtin_bit_cast(__bf16, raw);
        bsum[n] += bf2f(raw);
      }
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      float v[8];
      Chunk<bf16_t>::unpack(yv[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = v[j] * sc[j] + sf[j];
        v[j] = p.act ? silu_f(u) : u;
      }
      // pixels past the end hold act(shift) != 0, but their dout rows are zero: they add nothing
      *(uint4*)(tile + (s * PPP + pr) * LDB + cc * 16) = Chunk<bf16_t>::pack(v);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const char* a0 = tile + trow * LDB + (t * 16 + tcol) * 2;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 16 * LDB));
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      const bf16x8_t b = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b, acc[0][t], 0, 0, 0);
      if (two) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b, acc[1][t], 0, 0, 0);
    }
    __syncthreads();
  }
  // fold the four waves: red[wave][o 0..31][CIN]; D[row = 4g + r -> output][column lp -> channel of tile t]
  float* red = psm;
  const int NO = two ? 2 : 1;
  for (int n = 0; n < NO; ++n)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * n + 4 * g + r) * CIN + t * 16 + lp] = (n == 0 ? acc[0][t][r] : acc[1][t][r]);
  // bias sums: lanes (lp = o, g) -> fold the four k groups of the wave, then the waves
  __shared__ float bred[4][32];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float v = bsum[n];
    v = lane_xor32_sum(lane_xor16_sum(v));
    if (g == 0) bred[wave][16 * n + lp] = v;
  }
  __syncthreads();
  for (int i = tid; i < cout * CIN; i += 256) {
    const int o = i / CIN, ch = i - o * CIN;
    p.slab[((long)blockIdx.x * p.ctot + p.ooff[br] + o) * CIN + ch] =
        red[(0 * 32 + o) * CIN + ch] + red[(1 * 32 + o) * CIN + ch] + red[(2 * 32 + o) * CIN + ch] + red[(3 * 32 + o) * CIN + ch];
  }
  if (tid < cout) p.bslab[(long)blockIdx.x * p.ctot + p.ooff[br] + tid] = bred[0][tid] + bred[1][tid] + bred[2][tid] + bred[3][tid];
}

// dw[br][o][ch] = sum_blk slab[blk][ooff + o][ch];  db[br][o] = sum_blk bslab[blk][ooff + o]
__global__ void projg_wslab_reduce_kernel(PWP p, int nblk, int cin) {
  const int br = blockIdx.y, cout = p.cout[br];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < cout * cin) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += p.slab[((long)b * p.ctot + p.ooff[br]) * cin + idx];
    p.dw[br][idx] = s;
  }
  if (idx < cout) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += p.bslab[(long)b * p.ctot + p.ooff[br] + idx];
    p.db[br][idx] = s;
  }
}

}  // namespace

extern "C" {

int y3d_proj_group_bwd_weight_bn_mfma_blocks(int64_t P) {
  long n = (P + 1023) / 1024;
  return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
}

int y3d_proj_group_bwd_weight_bn_mfma(int nb, int cin, const void* y_pre, int64_t ysw, const int* xoff, const void* dout, int64_t dsw,
                                      const int* couts, const float* scale, const float* shift, int act, float* slab, float* bslab,
                                      float* const* dw, float* const* db, int64_t P, void* stream) {
  Y3D_CHECK(nb >= 1 && nb <= PB_MAXB && (cin == 64 || cin == 128), "proj_group_bwd_weight_bn_mfma: 1..16 branches of 64 or 128 channels");
  Y3D_CHECK(y_pre && dout && scale && shift && slab && bslab && dw && db && P >= 1 && ysw % 8 == 0 && ((uintptr_t)y_pre & 15) == 0,
            "proj_group_bwd_weight_bn_mfma: null / misaligned argument");
  PWP p;
  p.y = (const bf16_t*)y_pre; p.dout = (const bf16_t*)dout; p.slab = slab; p.bslab = bslab;
  int off = 0;
  for (int i = 0; i < nb; ++i) {
    Y3D_CHECK(couts[i] >= 1 && couts[i] <= 32 && xoff[i] % 8 == 0, "proj_group_bwd_weight_bn_mfma: cout in 1..32, channel slices 16-byte aligned");
    p.dw[i] = dw[i]; p.db[i] = db[i]; p.xoff[i] = xoff[i]; p.ooff[i] = off; p.cout[i] = couts[i];
    off += couts[i];
  }
  Y3D_CHECK(dsw >= off, "proj_group_bwd_weight_bn_mfma: dout pixel stride smaller than the projected channels");
  p.scale = scale; p.shift = shift; p.ysw = ysw; p.dsw = dsw; p.P = P; p.act = act; p.ctot = off;
  const int nrun = y3d_proj_group_bwd_weight_bn_mfma_blocks(P);
  p.px_per_block = (int)(((P + nrun - 1) / nrun + 127) / 128 * 128);
  hipStream_t st = (hipStream_t)stream;
  const size_t sm = (size_t)4 * 32 * cin * sizeof(float);  // the cross-wave fold (>= the four bf16 tiles)
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)projg_bwd_weight_bn_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 128 * 4);
    (void)hipFuncSetAttribute((const void*)projg_bwd_weight_bn_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 64 * 4);
    attr = true;
  }
  if (cin == 128) hipLaunchKernelGGL((projg_bwd_weight_bn_mfma_kernel<8>), dim3(nrun, nb), dim3(256), sm, st, p);
  else hipLaunchKernelGGL((projg_bwd_weight_bn_mfma_kernel<4>), dim3(nrun, nb), dim3(256), sm, st, p);
  hipLaunchKernelGGL(projg_wslab_reduce_kernel, dim3((32 * cin + 255) / 256, nb), dim3(256), 0, st, p, nrun, cin);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_group_fwd_bn_mfma(int nb, int cin, const void* y_pre, int64_t ysw, const int* xoff, const float* const* w, const float* const* b,
                               const int* couts, const float* scale, const float* shift, int act, void* out, int64_t osw, int64_t P,
                               void* stream) {
  Y3D_CHECK(nb >= 1 && nb <= PB_MAXB && (cin == 64 || cin == 128), "proj_group_fwd_bn_mfma: 1..16 branches of 64 or 128 channels");
  Y3D_CHECK(y_pre && out && scale && shift && P >= 1 && ysw % 8 == 0 && ((uintptr_t)y_pre & 15) == 0, "proj_group_fwd_bn_mfma: null / misaligned argument");
  PFP p;
  p.y = (const bf16_t*)y_pre; p.out = (bf16_t*)out;
  int off = 0;
  for (int i = 0; i < nb; ++i) {
    Y3D_CHECK(couts[i] >= 1 && couts[i] <= 32 && xoff[i] % 8 == 0, "proj_group_fwd_bn_mfma: cout in 1..32, channel slices 16-byte aligned");
    p.w[i] = w[i]; p.b[i] = b[i]; p.xoff[i] = xoff[i]; p.ooff[i] = off; p.cout[i] = couts[i];
    off += couts[i];
  }
  Y3D_CHECK(osw >= off, "proj_group_fwd_bn_mfma: output pixel stride smaller than the projected channels");
  p.scale = scale; p.shift = shift; p.ysw = ysw; p.osw = osw; p.P = P; p.act = act;
  long nrun = (P + 63) / 64;
  if (nrun > 128) nrun = 128;
  p.px_per_block = (int)(((P + nrun - 1) / nrun + 63) / 64 * 64);
  dim3 grid((unsigned)nrun, nb);
  hipStream_t st = (hipStream_t)stream;
  if (cin == 128) hipLaunchKernelGGL((projg_fwd_bn_mfma_kernel<8>), grid, dim3(256), (size_t)4 * 16 * (128 * 2 + 16), st, p);
  else hipLaunchKernelGGL((projg_fwd_bn_mfma_kernel<4>), grid, dim3(256), (size_t)4 * 16 * (64 * 2 + 16), st, p);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_group_bn_bwd_blocks(int64_t P) {
  long n = (P + 63) / 64;
  return (int)(n < 1 ? 1 : (n > 128 ? 128 : n));
}

int y3d_proj_group_bn_bwd(int mode, int nb, int cin, const void* y_pre, int64_t ysw, const int* xoff, const void* dout, int64_t dsw,
                          const float* const* w, const int* couts, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const float* mean_g, const float* mean_gx, int act, float* partials, int nblk, void* dy,
                          int64_t dysw, int64_t P, int C, void* stream) {
  Y3D_CHECK(mode == 0 || mode == 1, "proj_group_bn_bwd: mode 0 (reduce) or 1 (apply)");
  Y3D_CHECK(nb >= 1 && nb <= PB_MAXB && (cin == 64 || cin == 128), "proj_group_bn_bwd: 1..16 branches of 64 or 128 channels");
  Y3D_CHECK(y_pre && dout && scale && shift && mean && invstd && P >= 1, "proj_group_bn_bwd: null argument");
  Y3D_CHECK(mode == 0 ? (partials != nullptr && nblk == y3d_proj_group_bn_bwd_blocks(P)) : (dy && mean_g && mean_gx && dysw >= cin),
            "proj_group_bn_bwd: partials / nblk (reduce) or dy / mean_g / mean_gx (apply) missing");
  Y3D_CHECK(ysw % 8 == 0 && ((uintptr_t)y_pre & 15) == 0 && (mode == 0 || (dysw % 8 == 0 && ((uintptr_t)dy & 15) == 0)), "proj_group_bn_bwd: 16-byte alignment");
  PBP p;
  p.y = (const bf16_t*)y_pre; p.dout = (const bf16_t*)dout; p.dy = (bf16_t*)dy; p.part = partials;
  int off = 0;
  for (int i = 0; i < nb; ++i) {
    Y3D_CHECK(couts[i] >= 1 && couts[i] <= 32 && xoff[i] % 8 == 0 && xoff[i] + cin <= C, "proj_group_bn_bwd: cout in 1..32, channel slices 16-byte aligned inside C");
    p.w[i] = w[i]; p.xoff[i] = xoff[i]; p.ooff[i] = off; p.cout[i] = couts[i];
    off += couts[i];
  }
  Y3D_CHECK(dsw >= off, "proj_group_bn_bwd: dout pixel stride smaller than the projected channels");
  p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.mg = mean_g; p.mgx = mean_gx;
  p.ysw = ysw; p.dsw = dsw; p.dysw = dysw; p.P = P; p.cin = cin; p.C = C; p.act = act;
  const int nrun = y3d_proj_group_bn_bwd_blocks(P);
  p.px_per_block = (int)(((P + nrun - 1) / nrun + 63) / 64 * 64);
  dim3 grid(nrun, nb);
  hipStream_t st = (hipStream_t)stream;
  const size_t sm128 = (size_t)4 * 16 * (128 + 4) * sizeof(float), sm64 = (size_t)4 * 16 * (64 + 4) * sizeof(float);
  if (cin == 128) {
    if (mode == 0) hipLaunchKernelGGL((projg_bn_bwd_kernel<0, 8>), grid, dim3(256), sm128, st, p);
    else hipLaunchKernelGGL((projg_bn_bwd_kernel<1, 8>), grid, dim3(256), sm128, st, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((projg_bn_bwd_kernel<0, 4>), grid, dim3(256), sm64, st, p);
    else hipLaunchKernelGGL((projg_bn_bwd_kernel<1, 4>), grid, dim3(256), sm64, st, p);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
