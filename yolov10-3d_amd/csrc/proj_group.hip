// The 16 final projections of one head level (8 branches x 2 head sets: nn.Conv2d(mid, out, 1) + bias, head.py:637, followed by the
// torch.cat of head.py:742) as ONE launch per direction: blockIdx.z walks the branches, each reading its channel slice of the
// stacked feature tensor and writing its channel slice of the (B, sum(out), H, W) map.  Small-N work: VALU, HBM-bound.
#include "common.h"

namespace {

constexpr int MAXB = 16;

struct ProjG {
  const float* w[MAXB];   // [cout][cin] fp32
  const float* b[MAXB];   // [cout]
  float* dw[MAXB];
  float* db[MAXB];
  int xoff[MAXB];         // channel offset of the branch's input slice inside x
  int ooff[MAXB];         // channel offset of the branch's outputs inside y / dy
  int cout[MAXB];
  int nb, cin;
};

// BN: the per-channel constants of the BatchNorm that precedes the projections, indexed by the channel of the STACKED tensor.
// When given, x is the PRE-BatchNorm conv output and act(x * scale + shift) is formed on the fly: the activation tensor is
// never materialised (forward) and neither is its gradient (backward): 6 passes over the 839 MB head tensors less per step.
struct BNP {
  const float* scale; const float* shift; const float* mean; const float* invstd; const float* mg; const float* mgx;
  int act;
};

template <typename T>
__global__ __launch_bounds__(256) void projg_fwd_kernel(ProjG g, const T* __restrict__ x, long xsw, T* __restrict__ y, long ysw, long P, BNP bn) {
  constexpr int CE = TT<T>::CE;
  constexpr int CN = 24;
  constexpr int PXB = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sw = (float*)smem;             // [24][Cin]
  float* sbn = sw + 24 * g.cin;         // [2][Cin] scale, shift of this branch's channels
  const int br = blockIdx.z, Cin = g.cin, Cout = g.cout[br];
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = g.w[br][i];
  if (bn.scale)
    for (int i = threadIdx.x; i < Cin; i += 256) { sbn[i] = bn.scale[g.xoff[br] + i]; sbn[Cin + i] = bn.shift[g.xoff[br] + i]; }
  __syncthreads();
  const int sub = threadIdx.x & 7;
  const int cpr = Cin / CE;
  const T* xp = x + g.xoff[br];
  // PXB pixels per thread (a block covers 32 * PXB pixels): the block's weight + BatchNorm slab is loaded once for all of them
  for (int it = 0; it < PXB; ++it) {
    const long px = ((long)blockIdx.x * PXB + it) * 32 + (threadIdx.x >> 3);
    float acc[CN];
#pragma unroll
    for (int o = 0; o < CN; ++o) acc[o] = 0.f;
    if (px < P) {
      for (int ch = sub; ch < cpr; ch += 8) {
        float v[CE];
        Chunk<T>::unpack(*(const uint4*)(xp + px * xsw + ch * CE), v);
        if (bn.scale) {
#pragma unroll
          for (int j = 0; j < CE; ++j) {
            float u = v[j] * sbn[ch * CE + j] + sbn[Cin + ch * CE + j];
            v[j] = TT<T>::rnd(bn.act ? silu_f(u) : u);  // rounded as the materialised activation tensor would be
          }
        }
#pragma unroll
        for (int o = 0; o < CN; ++o)
          if (o < Cout) {
            const float* wr = sw + o * Cin + ch * CE;
#pragma unroll
            for (int j = 0; j < CE; ++j) acc[o] += v[j] * wr[j];
          }
      }
    }
#pragma unroll
    for (int o = 0; o < CN; ++o)
      if (o < Cout) {  // uniform
        acc[o] += __shfl_xor(acc[o], 1);
        acc[o] += __shfl_xor(acc[o], 2);
        acc[o] += __shfl_xor(acc[o], 4);
      }
    if (px < P && sub == 0) {
      T* yp = y + px * ysw + g.ooff[br];
#pragma unroll
      for (int o = 0; o < CN; ++o)
        if (o < Cout) TT<T>::st(yp + o, acc[o] + g.b[br][o]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void projg_bwd_data_kernel(ProjG g, const T* __restrict__ dy, long dsw, T* __restrict__ dx, long xsw, long P) {
  constexpr int CE = TT<T>::CE;
  constexpr int CN = 24;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sw = (float*)smem;
  const int br = blockIdx.z, Cin = g.cin, Cout = g.cout[br];
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) sw[i] = g.w[br][i];
  __syncthreads();
  const int cpr = Cin / CE;
  long total = P * cpr;
  const T* dp = dy + g.ooff[br];
  T* xp = dx + g.xoff[br];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long px = idx / cpr;
    int c = (int)(idx - px * cpr) * CE;
    float acc[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) acc[j] = 0.f;
#pragma unroll
    for (int o = 0; o < CN; ++o)
      if (o < Cout) {
        float d = TT<T>::ld(dp + px * dsw + o);
        const float* wr = sw + o * Cin + c;
#pragma unroll
        for (int j = 0; j < CE; ++j) acc[j] += d * wr[j];
      }
    *(uint4*)(xp + px * xsw + c) = Chunk<T>::pack(acc);
  }
}

// slab[blk][ooff + co][ci] / bslab[blk][ooff + co]: per-block partial sums over the block's pixel range
template <typename T>
__global__ __launch_bounds__(256, 2) void projg_bwd_weight_kernel(ProjG g, const T* __restrict__ x, long xsw, const T* __restrict__ dy, long dsw,
                                                               float* __restrict__ slab, float* __restrict__ bslab, long P, int ctot,
                                                               int px_per_block, BNP bn) {
  constexpr int CE = TT<T>::CE;
  constexpr int CT = 64 / CE;
  constexpr int PT = 256 / CT;
  constexpr int OG = 8;  // outputs per pass over the pixel range (the 24-output heading branch: 3 passes)
  __shared__ float sh[4][OG * 64];
  __shared__ float shb[4][OG];
  const int br = blockIdx.z, Cin = g.cin, Cout = g.cout[br];
  const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
  const int cs = blockIdx.y * 64;
  const int c = cs + ct * CE;
  const T* xp = x + g.xoff[br];
  const T* dp = dy + g.ooff[br];
  long pbeg = (long)blockIdx.x * px_per_block;
  long pend = pbeg + px_per_block < P ? pbeg + px_per_block : P;
  for (int o0 = 0; o0 < Cout; o0 += OG) {
    float acc[OG][CE], bacc[OG];
#pragma unroll
    for (int o = 0; o < OG; ++o) {
      bacc[o] = 0.f;
#pragma unroll
      for (int j = 0; j < CE; ++j) acc[o][j] = 0.f;
    }
    if (c < Cin) {
      const int no = Cout - o0 < OG ? Cout - o0 : OG;
      float bsc[CE], bsh[CE];
      if (bn.scale) {
#pragma unroll
        for (int j = 0; j < CE; ++j) { bsc[j] = bn.scale[g.xoff[br] + c + j]; bsh[j] = bn.shift[g.xoff[br] + c + j]; }
      }
      // dy values of one pixel for this pass (clamped index: lanes past the branch's last output re-read it, never past the row)
      // pixel indices are block-relative 32-bit ints on scalar base pointers (64-bit per-lane addresses for four pixels in flight
      // pushed the kernel to 12 B of scratch)
      const T* dpb = dp + pbeg * dsw;
      const T* xpb = xp + pbeg * xsw + c;
      const int dswi = (int)dsw, xswi = (int)xsw, npx = (int)(pend - pbeg);
      auto dload = [&](int px, float* d) {
        const T* drow = dpb + px * dswi + o0;
#pragma unroll
        for (int o = 0; o < OG; ++o) d[o] = TT<T>::ld(drow + (o < no ? o : no - 1));
      };
      auto one = [&](const uint4& xv, const float* d) {
        float v[CE];
        Chunk<T>::unpack(xv, v);
        if (bn.scale) {  // x is the pre-BatchNorm tensor: rebuild the activation the forward projected
#pragma unroll
          for (int j = 0; j < CE; ++j) {
            const float u = v[j] * bsc[j] + bsh[j];
            v[j] = TT<T>::rnd(bn.act ? silu_f(u) : u);
          }
        }
#pragma unroll
        for (int o = 0; o < OG; ++o)
          if (o < no) {
            bacc[o] += d[o];
#pragma unroll
            for (int j = 0; j < CE; ++j) acc[o][j] += d[o] * v[j];
          }
      };
      int px = pt;
      for (; px + 3 * PT < npx; px += 4 * PT) {  // four pixels in flight, x chunks AND dy values: the loop is latency-bound otherwise
        const uint4 x0 = *(const uint4*)(xpb + px * xswi), x1 = *(const uint4*)(xpb + (px + PT) * xswi);
        const uint4 x2 = *(const uint4*)(xpb + (px + 2 * PT) * xswi), x3 = *(const uint4*)(xpb + (px + 3 * PT) * xswi);
        float d0[OG], d1[OG], d2[OG], d3[OG];
        dload(px, d0); dload(px + PT, d1); dload(px + 2 * PT, d2); dload(px + 3 * PT, d3);
        one(x0, d0);
        one(x1, d1);
        one(x2, d2);
        one(x3, d3);
      }
      for (; px < npx; px += PT) {
        float d0[OG];
        dload(px, d0);
        one(*(const uint4*)(xpb + px * xswi), d0);
      }
    }
    // fold the block's 32 pixel rows: the 8 rows of a wave by VALU lane swaps (bf16 layout: lane = (row % 8) * 8 + chunk), the 4 waves
    // through LDS once per pass (it was 2-4 barriers and a 32-step LDS loop per output)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // the previous pass's readers are done with sh
#pragma unroll
    for (int o = 0; o < OG; ++o) {
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        float v = acc[o][j];
        if (CT == 8) v = lane_xor8_sum(v);
        v = lane_xor32_sum(lane_xor16_sum(v));
        if (lane < CT) sh[wave][o * 64 + ct * CE + j] = v;
      }
      float bv = bacc[o];
      if (CT == 8) bv = lane_xor8_sum(bv);
      bv = lane_xor32_sum(lane_xor16_sum(bv));
      if (lane == 0) shb[wave][o] = bv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < OG * 64; i += 256) {
      const int o = i >> 6, cc = i & 63;
      if (o0 + o < Cout && cs + cc < Cin)
        slab[((long)blockIdx.x * ctot + g.ooff[br] + o0 + o) * Cin + cs + cc] = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
    }
    if (blockIdx.y == 0 && threadIdx.x < OG && o0 + threadIdx.x < Cout)
      bslab[(long)blockIdx.x * ctot + g.ooff[br] + o0 + threadIdx.x] = shb[0][threadIdx.x] + shb[1][threadIdx.x] + shb[2][threadIdx.x] + shb[3][threadIdx.x];
  }
}

// per branch: dw[br][co][ci] = sum_blk slab[blk][ooff+co][ci];  db[br][co] = sum_blk bslab[blk][ooff+co]
__global__ void projg_reduce_kernel(ProjG g, const float* __restrict__ slab, const float* __restrict__ bslab, int nblk, int ctot) {
  const int br = blockIdx.y, Cin = g.cin, Cout = g.cout[br];
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < Cout * Cin) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += slab[((long)b * ctot + g.ooff[br]) * Cin + idx];
    g.dw[br][idx] = s;
  }
  if (idx < Cout) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += bslab[(long)b * ctot + g.ooff[br] + idx];
    g.db[br][idx] = s;
  }
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

int fill(ProjG& g, int nb, int cin, const float* const* w, const float* const* b, float* const* dw, float* const* db, const int* xoff,
         const int* couts) {
  Y3D_CHECK(nb >= 1 && nb <= MAXB, "proj_group: 1..%d branches", MAXB);
  Y3D_CHECK((size_t)24 * cin * 4 <= 64 * 1024, "proj_group: cin too large for the LDS weight slab");
  int off = 0;
  for (int i = 0; i < nb; ++i) {
    Y3D_CHECK(couts[i] >= 1 && couts[i] <= 24, "proj_group: cout in 1..24");
    g.w[i] = w[i]; g.b[i] = b ? b[i] : nullptr; g.dw[i] = dw ? dw[i] : nullptr; g.db[i] = db ? db[i] : nullptr;
    g.xoff[i] = xoff[i]; g.ooff[i] = off; g.cout[i] = couts[i];
    off += couts[i];
  }
  g.nb = nb; g.cin = cin;
  return off;
}

}  // namespace

extern "C" {

static int proj_group_fwd_impl(int dtype, int nb, int cin, const void* x, int64_t xsw, const int* xoff, const float* const* w,
                               const float* const* b, const int* couts, void* y, int64_t ysw, int64_t P, BNP bn, void* stream) {
  ProjG g;
  int ctot = fill(g, nb, cin, w, b, nullptr, nullptr, xoff, couts);
  if (ctot < 0) return ctot;
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "proj_group_fwd: bad dtype");
  Y3D_CHECK(cin % (dtype == Y3D_BF16 ? 8 : 4) == 0 && ysw >= ctot, "proj_group_fwd: channel alignment");
  dim3 grid(cdiv(P, 32 * 4), 1, nb);  // PXB = 4 pixels per thread
  size_t sm = (size_t)26 * cin * 4;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(projg_fwd_kernel<bf16_t>, grid, dim3(256), sm, (hipStream_t)stream, g, (const bf16_t*)x, (long)xsw, (bf16_t*)y, (long)ysw, (long)P, bn);
  else hipLaunchKernelGGL(projg_fwd_kernel<float>, grid, dim3(256), sm, (hipStream_t)stream, g, (const float*)x, (long)xsw, (float*)y, (long)ysw, (long)P, bn);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_group_fwd(int dtype, int nb, int cin, const void* x, int64_t xsw, const int* xoff, const float* const* w,
                       const float* const* b, const int* couts, void* y, int64_t ysw, int64_t P, void* stream) {
  BNP bn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return proj_group_fwd_impl(dtype, nb, cin, x, xsw, xoff, w, b, couts, y, ysw, P, bn, stream);
}

int y3d_proj_group_fwd_bn(int dtype, int nb, int cin, const void* y_pre, int64_t xsw, const int* xoff, const float* const* w,
                          const float* const* b, const int* couts, const float* scale, const float* shift, int act, void* out,
                          int64_t ysw, int64_t P, void* stream) {
  Y3D_CHECK(scale && shift, "proj_group_fwd_bn: scale / shift missing");
  BNP bn{scale, shift, nullptr, nullptr, nullptr, nullptr, act};
  return proj_group_fwd_impl(dtype, nb, cin, y_pre, xsw, xoff, w, b, couts, out, ysw, P, bn, stream);
}

int y3d_proj_group_bwd_data(int dtype, int nb, int cin, const void* dy, int64_t dsw, const int* xoff, const float* const* w,
                            const int* couts, void* dx, int64_t xsw, int64_t P, void* stream) {
  ProjG g;
  int ctot = fill(g, nb, cin, w, nullptr, nullptr, nullptr, xoff, couts);
  if (ctot < 0) return ctot;
  long total = P * (cin / (dtype == Y3D_BF16 ? 8 : 4));
  dim3 grid(ew_grid(total), 1, nb);
  size_t sm = (size_t)24 * cin * 4;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(projg_bwd_data_kernel<bf16_t>, grid, dim3(256), sm, (hipStream_t)stream, g, (const bf16_t*)dy, (long)dsw, (bf16_t*)dx, (long)xsw, (long)P);
  else hipLaunchKernelGGL(projg_bwd_data_kernel<float>, grid, dim3(256), sm, (hipStream_t)stream, g, (const float*)dy, (long)dsw, (float*)dx, (long)xsw, (long)P);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_group_blocks(int64_t P) {
  long n = (P + 1023) / 1024;
  return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
}

/* slab: blocks * ctot * cin floats, bslab: blocks * ctot floats */
static int proj_group_bwd_weight_impl(int dtype, int nb, int cin, const void* x, int64_t xsw, const int* xoff, const void* dy, int64_t dsw,
                                      const int* couts, float* slab, float* bslab, float* const* dw, float* const* db, int64_t P, BNP bn,
                                      void* stream) {
  ProjG g;
  const float* dummy[MAXB] = {nullptr};
  int ctot = fill(g, nb, cin, dummy, nullptr, dw, db, xoff, couts);
  if (ctot < 0) return ctot;
  int nblk = y3d_proj_group_blocks(P);
  int ppb = (int)((P + nblk - 1) / nblk);
  Y3D_CHECK((long)ppb * (xsw > dsw ? xsw : dsw) < (1L << 31), "proj_group_bwd_weight: a block's pixel range exceeds 32-bit element offsets");
  dim3 grid(nblk, cdiv(cin, 64), nb);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16) hipLaunchKernelGGL(projg_bwd_weight_kernel<bf16_t>, grid, dim3(256), 0, st, g, (const bf16_t*)x, (long)xsw, (const bf16_t*)dy, (long)dsw, slab, bslab, (long)P, ctot, ppb, bn);
  else hipLaunchKernelGGL(projg_bwd_weight_kernel<float>, grid, dim3(256), 0, st, g, (const float*)x, (long)xsw, (const float*)dy, (long)dsw, slab, bslab, (long)P, ctot, ppb, bn);
  hipLaunchKernelGGL(projg_reduce_kernel, dim3(cdiv(24 * cin, 256), nb), dim3(256), 0, st, g, slab, bslab, nblk, ctot);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_proj_group_bwd_weight(int dtype, int nb, int cin, const void* x, int64_t xsw, const int* xoff, const void* dy, int64_t dsw,
                              const int* couts, float* slab, float* bslab, float* const* dw, float* const* db, int64_t P, void* stream) {
  BNP bn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return proj_group_bwd_weight_impl(dtype, nb, cin, x, xsw, xoff, dy, dsw, couts, slab, bslab, dw, db, P, bn, stream);
}

int y3d_proj_group_bwd_weight_bn(int dtype, int nb, int cin, const void* y_pre, int64_t xsw, const int* xoff, const void* dy, int64_t dsw,
                                 const int* couts, const float* scale, const float* shift, int act, float* slab, float* bslab,
                                 float* const* dw, float* const* db, int64_t P, void* stream) {
  Y3D_CHECK(scale && shift, "proj_group_bwd_weight_bn: scale / shift missing");
  BNP bn{scale, shift, nullptr, nullptr, nullptr, nullptr, act};
  return proj_group_bwd_weight_impl(dtype, nb, cin, y_pre, xsw, xoff, dy, dsw, couts, slab, bslab, dw, db, P, bn, stream);
}

}  // extern "C"
