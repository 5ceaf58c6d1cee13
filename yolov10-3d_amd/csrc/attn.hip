// PSA attention core (reference nn/modules/block.py:785-797): per (image, head)
//   attn = softmax_j( scale * q_i . k_j ),  out_i = sum_j attn_ij v_j
// on the NHWC qkv tensor whose per-head channel block is [q(kd) | k(kd) | v(hd)].
// Two implementations: flash-style fp32 VALU kernels (online softmax, one query or key per lane, the other operand streamed
// through LDS in 64-row tiles; any dtype, the exact-fp32 parity mode) and bf16 MFMA kernels for kd = 32 / hd = 64 (forward and
// backward, further down).  N is small on this path (400 @640^2, 1600 @1280^2).
#include "common.h"

extern "C" int y3d_get_tile_kernels(void);

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
constexpr int TK = 64;

struct AttnP {
  const void* qkv;   // [B][N][nh*(2kd+hd)]
  long qsw;          // pixel stride of qkv
  void* out;         // [B][N][nh*hd]
  long osw;
  float* lse;        // [B][nh][N]
  int B, N, nh, kd, hd;
  float scale;
};

template <typename T, int KD, int HD>
__global__ __launch_bounds__(128) void attn_fwd_kernel(AttnP p) {
  __shared__ float sK[TK][KD];
  __shared__ float sV[TK][HD];
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const int i = blockIdx.x * 128 + threadIdx.x;
  const int hoff = h * (2 * KD + HD);
  const T* base = (const T*)p.qkv + (long)b * p.N * p.qsw + hoff;
  float q[KD], o[HD];
  const bool valid = i < p.N;
#pragma unroll
  for (int d = 0; d < KD; ++d) q[d] = valid ? TT<T>::ld(base + (long)i * p.qsw + d) * p.scale : 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < p.N; j0 += TK) {
    __syncthreads();
    for (int e = threadIdx.x; e < TK * KD; e += 128) {
      int r = e / KD, d = e % KD;
      sK[r][d] = (j0 + r < p.N) ? TT<T>::ld(base + (long)(j0 + r) * p.qsw + KD + d) : 0.f;
    }
    for (int e = threadIdx.x; e < TK * HD; e += 128) {
      int r = e / HD, d = e % HD;
      sV[r][d] = (j0 + r < p.N) ? TT<T>::ld(base + (long)(j0 + r) * p.qsw + 2 * KD + d) : 0.f;
    }
    __syncthreads();
    const int jn = min(TK, p.N - j0);
    for (int j = 0; j < jn; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < KD; ++d) s += q[d] * sK[j][d];
      if (s > m) {
        float corr = __expf(m - s);
        l *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= corr;
        m = s;
      }
      float pj = __expf(s - m);
      l += pj;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] += pj * sV[j][d];
    }
  }
  if (valid) {
    float inv = 1.f / l;
    T* dst = (T*)p.out + ((long)b * p.N + i) * p.osw + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) TT<T>::st(dst + d, o[d] * inv);
    p.lse[((long)b * p.nh + h) * p.N + i] = m + __logf(l);
  }
}

struct AttnBP {
  const void* qkv; long qsw;
  const void* out; long osw;     // forward output (for delta)
  const void* dout; long dsw;    // grad wrt out, [B][N][nh*hd]
  const void* dv_extra; long esw;  // optional extra grad for v (from the `pe` branch), [B][N][nh*hd]
  void* dqkv; long gsw;          // [B][N][nh*(2kd+hd)]
  const float* lse;
  float* delta;                  // [B][nh][N]
  int B, N, nh, kd, hd;
  float scale;
  int chunk;                     // MFMA kernels: rows of the LDS image per pass (multiple of 32); N > chunk walks the image in passes
};

// dq_i = scale * sum_j ds_ij k_j,   ds_ij = p_ij (do_i . v_j - delta_i)
template <typename T, int KD, int HD>
__global__ __launch_bounds__(128) void attn_bwd_q_kernel(AttnBP p) {
  __shared__ float sK[TK][KD];
  __shared__ float sV[TK][HD];
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const int i = blockIdx.x * 128 + threadIdx.x;
  const int hoff = h * (2 * KD + HD);
  const T* base = (const T*)p.qkv + (long)b * p.N * p.qsw + hoff;
  const bool valid = i < p.N;
  float q[KD], dq[KD], dO[HD];
  float delta = 0.f, lse = 0.f;
#pragma unroll
  for (int d = 0; d < KD; ++d) { q[d] = valid ? TT<T>::ld(base + (long)i * p.qsw + d) * p.scale : 0.f; dq[d] = 0.f; }
  if (valid) {
    const T* op = (const T*)p.out + ((long)b * p.N + i) * p.osw + h * HD;
    const T* dp = (const T*)p.dout + ((long)b * p.N + i) * p.dsw + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { dO[d] = TT<T>::ld(dp + d); delta += dO[d] * TT<T>::ld(op + d); }
    lse = p.lse[((long)b * p.nh + h) * p.N + i];
    p.delta[((long)b * p.nh + h) * p.N + i] = delta;
  } else {
#pragma unroll
    for (int d = 0; d < HD; ++d) dO[d] = 0.f;
  }
  for (int j0 = 0; j0 < p.N; j0 += TK) {
    __syncthreads();
    for (int e = threadIdx.x; e < TK * KD; e += 128) {
      int r = e / KD, d = e % KD;
      sK[r][d] = (j0 + r < p.N) ? TT<T>::ld(base + (long)(j0 + r) * p.qsw + KD + d) : 0.f;
    }
    for (int e = threadIdx.x; e < TK * HD; e += 128) {
      int r = e / HD, d = e % HD;
      sV[r][d] = (j0 + r < p.N) ? TT<T>::ld(base + (long)(j0 + r) * p.qsw + 2 * KD + d) : 0.f;
    }
    __syncthreads();
    const int jn = min(TK, p.N - j0);
    for (int j = 0; j < jn; ++j) {
      float s = 0.f, dpv = 0.f;
#pragma unroll
      for (int d = 0; d < KD; ++d) s += q[d] * sK[j][d];
#pragma unroll
      for (int d = 0; d < HD; ++d) dpv += dO[d] * sV[j][d];
      float ds = __expf(s - lse) * (dpv - delta);
#pragma unroll
      for (int d = 0; d < KD; ++d) dq[d] += ds * sK[j][d];
    }
  }
  if (valid) {
    T* dst = (T*)p.dqkv + ((long)b * p.N + i) * p.gsw + hoff;
#pragma unroll
    for (int d = 0; d < KD; ++d) TT<T>::st(dst + d, dq[d] * p.scale);
  }
}

// dk_j = scale * sum_i ds_ij q_i ;  dv_j = sum_i p_ij do_i (+ dv_extra_j)
template <typename T, int KD, int HD>
__global__ __launch_bounds__(128) void attn_bwd_kv_kernel(AttnBP p) {
  __shared__ float sQ[TK][KD];
  __shared__ float sD[TK][HD];
  __shared__ float sL[TK], sDel[TK];
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const int j = blockIdx.x * 128 + threadIdx.x;
  const int hoff = h * (2 * KD + HD);
  const T* base = (const T*)p.qkv + (long)b * p.N * p.qsw + hoff;
  const bool valid = j < p.N;
  float k[KD], v[HD], dk[KD], dv[HD];
#pragma unroll
  for (int d = 0; d < KD; ++d) { k[d] = valid ? TT<T>::ld(base + (long)j * p.qsw + KD + d) : 0.f; dk[d] = 0.f; }
#pragma unroll
  for (int d = 0; d < HD; ++d) { v[d] = valid ? TT<T>::ld(base + (long)j * p.qsw + 2 * KD + d) : 0.f; dv[d] = 0.f; }
  for (int i0 = 0; i0 < p.N; i0 += TK) {
    __syncthreads();
    for (int e = threadIdx.x; e < TK * KD; e += 128) {
      int r = e / KD, d = e % KD;
      sQ[r][d] = (i0 + r < p.N) ? TT<T>::ld(base + (long)(i0 + r) * p.qsw + d) * p.scale : 0.f;
    }
    for (int e = threadIdx.x; e < TK * HD; e += 128) {
      int r = e / HD, d = e % HD;
      sD[r][d] = (i0 + r < p.N) ? TT<T>::ld((const T*)p.dout + ((long)b * p.N + i0 + r) * p.dsw + h * HD + d) : 0.f;
    }
    if (threadIdx.x < TK) {
      int i = i0 + threadIdx.x;
      sL[threadIdx.x] = i < p.N ? p.lse[((long)b * p.nh + h) * p.N + i] : 0.f;
      sDel[threadIdx.x] = i < p.N ? p.delta[((long)b * p.nh + h) * p.N + i] : 0.f;
    }
    __syncthreads();
    const int in = min(TK, p.N - i0);
    for (int i = 0; i < in; ++i) {
      float s = 0.f, dpv = 0.f;
#pragma unroll
      for (int d = 0; d < KD; ++d) s += sQ[i][d] * k[d];
#pragma unroll
      for (int d = 0; d < HD; ++d) dpv += sD[i][d] * v[d];
      float pij = __expf(s - sL[i]);
      float ds = pij * (dpv - sDel[i]);
#pragma unroll
      for (int d = 0; d < KD; ++d) dk[d] += ds * sQ[i][d];   // sQ already carries `scale`
#pragma unroll
      for (int d = 0; d < HD; ++d) dv[d] += pij * sD[i][d];
    }
  }
  if (valid) {
    T* dst = (T*)p.dqkv + ((long)b * p.N + j) * p.gsw + hoff;
#pragma unroll
    for (int d = 0; d < KD; ++d) TT<T>::st(dst + KD + d, dk[d]);
    const T* ex = p.dv_extra ? (const T*)p.dv_extra + ((long)b * p.N + j) * p.esw + h * HD : nullptr;
#pragma unroll
    for (int d = 0; d < HD; ++d) TT<T>::st(dst + 2 * KD + d, dv[d] + (ex ? TT<T>::ld(ex + d) : 0.f));
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// MFMA forward (bf16; kd = 32 / hd = 64, and kd = 36 / hd = 72 of the M widths): S^T = K Q^T and O^T = V^T P^T on the matrix cores.
// One wave owns 16 queries.  S^T tiles come out of `mfma(A = K rows, B = Q rows)` with the query on the lane and four keys in
// the registers - exactly the B-operand layout the second product needs (`An accumulator tile as the next MFMA's operand`), so
// P never leaves the registers.  The reduction index of O^T = V^T P^T is the key; its MFMA-k <-> key assignment is permuted
// (k = 8g+j <-> key 32*ks + 16*(j>>2) + 4g + (j&3)) to match the S^T accumulator layout, and V^T fragments are fetched from a
// swizzled LDS image of V with the transposed read.  Softmax is two-pass (row max first: QK^T is one MFMA per 16 x 16 tile, so
// recomputing it is cheaper than rescaling O), fp32 throughout.
// Head dims that are not multiples of 32 / 16 are zero-padded IN THE FRAGMENTS (kd = 36: a second K step with four live elements;
// hd = 72: a fifth, half-empty output tile and a third K step for the products that contract over hd).  The q | k | v blocks of a
// head then start at 8-byte, not 16-byte, offsets (k at element 36): global 16-byte loads only need dword alignment; the element
// mask is applied per 8-element chunk (`ld8m`).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int KC = 512;  // keys per LDS chunk of V

__device__ __forceinline__ bf16x8_t ld8(const bf16_t* p) { return __builtin_bit_cast(bf16x8_t, *(const uint4*)p); }
__device__ __forceinline__ bf16x8_t zero8() { return (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; }
// elements e0 .. e0 + 7 of a row whose live length is lim (a multiple of 4): zeros past lim
__device__ __forceinline__ bf16x8_t ld8m(const bf16_t* row, int e0, int lim) {
  if (e0 + 8 <= lim) return ld8(row + e0);
  if (e0 >= lim) return zero8();
  const uint2 u = *(const uint2*)(row + e0);  // four live elements
  return __builtin_bit_cast(bf16x8_t, make_uint4(u.x, u.y, 0u, 0u));
}
__device__ __forceinline__ uint4 ld8mu(const bf16_t* row, int e0, int lim) { return __builtin_bit_cast(uint4, ld8m(row, e0, lim)); }

// transposed A fragment out of an LDS image with ROWB-byte rows whose 32-byte pieces are XOR-swizzled by the row:
// element block [rows 32*ks + 16*hi + 4*grp + 0..3][columns col0 + 16-wide tile], lane li = 4*qq + pp4
template <int ROWB>
__device__ __forceinline__ int img_swz(int P) { return ((P >> (ROWB == 128 ? 1 : 0)) & (ROWB / 32 - 1)) << 1; }
template <int ROWB>
__device__ __forceinline__ bf16x8_t frag_t(const char* img, int ks, int col0, int grp, int qq, int pp4) {
  s16x4_t v[2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int P = ks * 32 + 16 * hi + 4 * grp + qq;
    const int cch = (col0 >> 3) + (pp4 >> 1);  // 16-byte chunk of the row
    const char* a = img + P * ROWB + ((cch ^ img_swz<ROWB>(P)) << 4) + (pp4 & 1) * 8;
    v[hi] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)a);
  }
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int KD, int HD>
struct AttnDims {
  static constexpr int KSK = (KD + 31) / 32, NTK = (KD + 15) / 16;   // K steps / 16-wide tiles over the key dimension
  static constexpr int KSH = (HD + 31) / 32, NTH = (HD + 15) / 16;   // ... over the head dimension
  static constexpr int VROW = HD <= 64 ? 128 : 256;                  // V image row (forward)
  static constexpr int KROW = 128;                                   // K image row (bwd_q): up to 64 elements
  static constexpr int HDI = KSH * 32, KDI = NTK * 16 <= 32 ? 32 : 64;
  static constexpr int IROW = (HDI + KDI) * 2 <= 256 ? 256 : 512;    // dO | Q image row (bwd_kv)
};

template <int KD, int HD>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnP p) {
  typedef bf16_t T;
  typedef AttnDims<KD, HD> D;
  constexpr int VROW = D::VROW, CPV = VROW / 16;
  extern __shared__ __attribute__((aligned(16))) char sV[];  // [KC][VROW], 16-byte chunks XOR-swizzled by the row (img_swz)
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp4 = li & 3;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int hoff = h * (2 * KD + HD);
  const T* base = (const T*)p.qkv + (long)b * p.N * p.qsw + hoff;
  // Q fragments (B operand): query q0 + li, elements 32 k2 + 8 grp .. + 7
  bf16x8_t fq[D::KSK];
#pragma unroll
  for (int k2 = 0; k2 < D::KSK; ++k2) fq[k2] = q0 + li < p.N ? ld8m(base + (long)(q0 + li) * p.qsw, 32 * k2 + 8 * grp, KD) : zero8();
  const int ntile = (p.N + 15) / 16;
  auto score_tile = [&](int t) -> f32x4_t {  // S^T tile: reg r <-> key 16t + 4*grp + r, column <-> query q0 + li
    const int key = 16 * t + li;
    f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k2 = 0; k2 < D::KSK; ++k2) {
      const bf16x8_t fk = key < p.N ? ld8m(base + (long)key * p.qsw + KD, 32 * k2 + 8 * grp, KD) : zero8();
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[k2], s, 0, 0, 0);
    }
    return s;
  };
  // pass 1: row maxima
  float m = -INFINITY;
  for (int t = 0; t < ntile; ++t) {
    f32x4_t s = score_tile(t);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * t + 4 * grp + r < p.N) m = fmaxf(m, s[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  const float ms = m * p.scale;
  // pass 2
  f32x4_t o[D::NTH];
#pragma unroll
  for (int dt = 0; dt < D::NTH; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int c0 = 0; c0 < p.N; c0 += KC) {
    const int nk = min(KC, p.N - c0);
    const int nk32 = (nk + 31) & ~31;
    __syncthreads();
    for (int id = tid; id < nk32 * CPV; id += 256) {
      const int P = id / CPV, s8 = id % CPV;
      const int src = s8 ^ img_swz<VROW>(P);  // position s8 of the row holds source chunk src
      uint4 v = make_uint4(0, 0, 0, 0);
      if (P < nk) v = ld8mu(base + (long)(c0 + P) * p.qsw + 2 * KD, src * 8, HD);
      *(uint4*)(sV + id * 16) = v;
    }
    __syncthreads();
    for (int ks = 0; ks < nk32 / 32; ++ks) {
      float pv[8];
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        int t = (c0 >> 4) + 2 * ks + hi;
        f32x4_t s = score_tile(t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = (16 * t + 4 * grp + r < p.N) ? __expf(s[r] * p.scale - ms) : 0.f;
          pv[hi * 4 + r] = e;
          l += e;
        }
      }
      bf16x8_t fp;
#pragma unroll
      for (int j = 0; j < 8; ++j) fp[j] = (__bf16)pv[j];
#pragma unroll
      for (int dt = 0; dt < D::NTH; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_t<VROW>(sV, ks, dt * 16, grp, qq, pp4), fp, o[dt], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (q0 + li < p.N) {
    const float inv = 1.f / l;
    T* dst = (T*)p.out + ((long)b * p.N + q0 + li) * p.osw + h * HD;
#pragma unroll
    for (int dt = 0; dt < D::NTH; ++dt) {
      if (dt * 16 + 4 * grp >= HD) continue;
      uint2 u;
      u.x = (unsigned)f2bf(o[dt][0] * inv) | ((unsigned)f2bf(o[dt][1] * inv) << 16);
      u.y = (unsigned)f2bf(o[dt][2] * inv) | ((unsigned)f2bf(o[dt][3] * inv) << 16);
      *(uint2*)(dst + dt * 16 + 4 * grp) = u;
    }
    if (grp == 0) p.lse[((long)b * p.nh + h) * p.N + q0 + li] = ms + __logf(l);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// MFMA backward (bf16), same register choreography as the forward:
//   attn_bwd_q_mfma : one wave owns 16 queries (the lane's column).  Per 32 keys: S^T = K Q^T and dP^T = V dO^T tiles come out
//     with four keys in the registers, dS^T = P^T o (dP^T - delta) becomes the B operand of dQ^T += K^T dS^T as it is; the K^T
//     fragments come from a swizzled LDS image of K with the transposed read.  Also writes delta_i = dO_i . O_i.
//   attn_bwd_kv_mfma: one wave owns 16 keys.  Per 32 queries: S = Q K^T and dP = dO V^T tiles with four queries in the registers;
//     P and dS are the B operands of dV^T += dO^T P and dK^T += Q^T dS, whose A fragments come from an LDS image
//     [query][dO | Q] with the transposed read.
// The reduction index <-> MFMA-k assignment is permuted as in the forward (k = 8g+j <-> row 32*ks + 16*(j>>2) + 4g + (j&3)).
// ---------------------------------------------------------------------------------------------------------------------------
template <int KD, int HD>
__global__ __launch_bounds__(256) void attn_bwd_q_mfma_kernel(AttnBP p) {
  typedef bf16_t T;
  typedef AttnDims<KD, HD> D;
  constexpr int KROW = D::KROW, NCK = D::NTK * 2;  // chunks of a key's K image row that the dQ tiles read (zero-padded past kd)
  extern __shared__ __attribute__((aligned(16))) char sK[];  // [chunk keys][128 B]
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp4 = li & 3;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int hoff = h * (2 * KD + HD);
  const T* base = (const T*)p.qkv + (long)b * p.N * p.qsw + hoff;
  const int n32 = (p.N + 31) & ~31;
  const bool qv = q0 + li < p.N;
  bf16x8_t fq[D::KSK], fdo[D::KSH];
#pragma unroll
  for (int k2 = 0; k2 < D::KSK; ++k2) fq[k2] = qv ? ld8m(base + (long)(q0 + li) * p.qsw, 32 * k2 + 8 * grp, KD) : zero8();
  float delta = 0.f, lse = 0.f;
  {
    const T* dp = (const T*)p.dout + ((long)b * p.N + q0 + li) * p.dsw + h * HD;
    const T* op = (const T*)p.out + ((long)b * p.N + q0 + li) * p.osw + h * HD;
#pragma unroll
    for (int k2 = 0; k2 < D::KSH; ++k2) {
      fdo[k2] = qv ? ld8m(dp, 32 * k2 + 8 * grp, HD) : zero8();
      if (qv) {
        const bf16x8_t fo = ld8m(op, 32 * k2 + 8 * grp, HD);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)fdo[k2][j] * (float)fo[j];
      }
    }
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
    if (qv) {
      lse = p.lse[((long)b * p.nh + h) * p.N + q0 + li];
      if (grp == 0) p.delta[((long)b * p.nh + h) * p.N + q0 + li] = delta;
    }
  }
  f32x4_t dq[D::NTK];
#pragma unroll
  for (int dt = 0; dt < D::NTK; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int cb = 0; cb < n32; cb += p.chunk) {  // the K image holds p.chunk keys: long sequences (hi-res maps) walk it in passes
  const int cn = min(p.chunk, n32 - cb);
  if (cb) __syncthreads();  // the previous pass's readers are done with the image
  for (int id = tid; id < cn * NCK; id += 256) {  // K image: NCK chunks of 16 B per key, zeros past kd
    const int Pl = id / NCK, c = id % NCK, P = cb + Pl;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (P < p.N) v = ld8mu(base + (long)P * p.qsw + KD, c * 8, KD);
    *(uint4*)(sK + Pl * KROW + ((c ^ img_swz<KROW>(Pl)) << 4)) = v;
  }
  __syncthreads();
  for (int ks = 0; ks < cn / 32; ++ks) {
    float dsv[8];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      const int key = cb + 32 * ks + 16 * hi + li;  // this lane's row of the A operands
      const bool kv = key < p.N;
      const T* kp = base + (long)key * p.qsw;
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < D::KSK; ++k2) {
        const bf16x8_t fk = kv ? ld8m(kp + KD, 32 * k2 + 8 * grp, KD) : zero8();
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[k2], s, 0, 0, 0);
      }
      f32x4_t dpt = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < D::KSH; ++k2) {
        const bf16x8_t fv = kv ? ld8m(kp + 2 * KD, 32 * k2 + 8 * grp, HD) : zero8();
        dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fdo[k2], dpt, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = qv && cb + 32 * ks + 16 * hi + 4 * grp + r < p.N;
        dsv[hi * 4 + r] = ok ? __expf(s[r] * p.scale - lse) * (dpt[r] - delta) : 0.f;
      }
    }
    bf16x8_t fds;
#pragma unroll
    for (int j = 0; j < 8; ++j) fds[j] = (__bf16)dsv[j];
#pragma unroll
    for (int dt = 0; dt < D::NTK; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_t<KROW>(sK, ks, dt * 16, grp, qq, pp4), fds, dq[dt], 0, 0, 0);
  }
  }
  if (qv) {
    T* dst = (T*)p.dqkv + ((long)b * p.N + q0 + li) * p.gsw + hoff;
#pragma unroll
    for (int dt = 0; dt < D::NTK; ++dt) {
      if (dt * 16 + 4 * grp >= KD) continue;
      uint2 u;
      u.x = (unsigned)f2bf(dq[dt][0] * p.scale) | ((unsigned)f2bf(dq[dt][1] * p.scale) << 16);
      u.y = (unsigned)f2bf(dq[dt][2] * p.scale) | ((unsigned)f2bf(dq[dt][3] * p.scale) << 16);
      *(uint2*)(dst + dt * 16 + 4 * grp) = u;
    }
  }
}

template <int KD, int HD>
__global__ __launch_bounds__(256) void attn_bwd_kv_mfma_kernel(AttnBP p) {
  typedef bf16_t T;
  typedef AttnDims<KD, HD> D;
  constexpr int IROW = D::IROW, HDI = D::HDI;
  constexpr int NCD = D::KSH * 4, NCQ = D::KSK * 4, NCI = NCD + NCQ;  // chunks per query: dO then Q, whole 32-element K steps (zeros past hd / kd)
  extern __shared__ __attribute__((aligned(16))) char sI[];  // [chunk queries][IROW]: dO at column 0, Q at column HDI
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp4 = li & 3;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const int hoff = h * (2 * KD + HD);
  const T* base = (const T*)p.qkv + (long)b * p.N * p.qsw + hoff;
  const T* dob = (const T*)p.dout + (long)b * p.N * p.dsw + h * HD;
  const int n32 = (p.N + 31) & ~31;
  const bool kv = k0 + li < p.N;
  const T* kp = base + (long)(k0 + li) * p.qsw;
  bf16x8_t fk[D::KSK], fv[D::KSH];  // B operands: this lane's key
#pragma unroll
  for (int k2 = 0; k2 < D::KSK; ++k2) fk[k2] = kv ? ld8m(kp + KD, 32 * k2 + 8 * grp, KD) : zero8();
#pragma unroll
  for (int k2 = 0; k2 < D::KSH; ++k2) fv[k2] = kv ? ld8m(kp + 2 * KD, 32 * k2 + 8 * grp, HD) : zero8();
  const float* lsep = p.lse + ((long)b * p.nh + h) * p.N;
  const float* delp = p.delta + ((long)b * p.nh + h) * p.N;
  f32x4_t dv[D::NTH], dk[D::NTK];
#pragma unroll
  for (int i = 0; i < D::NTH; ++i) dv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < D::NTK; ++i) dk[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int cb = 0; cb < n32; cb += p.chunk) {  // the dO | Q image holds p.chunk queries per pass
  const int cn = min(p.chunk, n32 - cb);
  if (cb) __syncthreads();
  for (int id = tid; id < cn * NCI; id += 256) {
    const int Pl = id / NCI, c = id - Pl * NCI, P = cb + Pl;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (P < p.N) v = c < NCD ? ld8mu(dob + (long)P * p.dsw, c * 8, HD) : ld8mu(base + (long)P * p.qsw, (c - NCD) * 8, KD);
    const int pos = c < NCD ? c : HDI / 8 + (c - NCD);  // chunk position inside the row
    *(uint4*)(sI + Pl * IROW + ((pos ^ img_swz<IROW>(Pl)) << 4)) = v;
  }
  // log-sum-exp and delta of the pass's queries next to the image: the inner loop reads nothing from global memory (its Q / dO
  // fragments come out of the image as plain 16-byte rows; with per-step global loads the loop ran at their latency: 172 us at N = 400)
  float* sLse = (float*)(sI + (size_t)p.chunk * IROW);
  float* sDel = sLse + p.chunk;
  for (int id = tid; id < cn; id += 256) {
    const int P = cb + id;
    sLse[id] = P < p.N ? lsep[P] : 0.f;
    sDel[id] = P < p.N ? delp[P] : 0.f;
  }
  __syncthreads();
  for (int ks = 0; ks < cn / 32; ++ks) {
    float pv[8], dsv[8];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      const int ql = 32 * ks + 16 * hi + li;  // this lane's row of the A operands (image row; rows past N are zeros)
      const char* row = sI + ql * IROW;
      const int sz = img_swz<IROW>(ql);
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < D::KSK; ++k2) {
        const bf16x8_t fq = __builtin_bit_cast(bf16x8_t, *(const uint4*)(row + (((HDI / 8 + 4 * k2 + grp) ^ sz) << 4)));
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq, fk[k2], s, 0, 0, 0);
      }
      f32x4_t dpt = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < D::KSH; ++k2) {
        const bf16x8_t fd = __builtin_bit_cast(bf16x8_t, *(const uint4*)(row + (((4 * k2 + grp) ^ sz) << 4)));
        dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd, fv[k2], dpt, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrl = 32 * ks + 16 * hi + 4 * grp + r;  // the query of register r (row of this pass)
        const bool ok = kv && cb + qrl < p.N;
        const float pe = ok ? __expf(s[r] * p.scale - sLse[qrl]) : 0.f;
        pv[hi * 4 + r] = pe;
        dsv[hi * 4 + r] = ok ? pe * (dpt[r] - sDel[qrl]) : 0.f;
      }
    }
    bf16x8_t fp, fds;
#pragma unroll
    for (int j = 0; j < 8; ++j) { fp[j] = (__bf16)pv[j]; fds[j] = (__bf16)dsv[j]; }
#pragma unroll
    for (int ct = 0; ct < D::NTH; ++ct) dv[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_t<IROW>(sI, ks, ct * 16, grp, qq, pp4), fp, dv[ct], 0, 0, 0);
#pragma unroll
    for (int dt = 0; dt < D::NTK; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_t<IROW>(sI, ks, HDI + dt * 16, grp, qq, pp4), fds, dk[dt], 0, 0, 0);
  }
  }
  if (kv) {
    T* dst = (T*)p.dqkv + ((long)b * p.N + k0 + li) * p.gsw + hoff;
#pragma unroll
    for (int dt = 0; dt < D::NTK; ++dt) {
      if (dt * 16 + 4 * grp >= KD) continue;
      uint2 u;
      u.x = (unsigned)f2bf(dk[dt][0] * p.scale) | ((unsigned)f2bf(dk[dt][1] * p.scale) << 16);
      u.y = (unsigned)f2bf(dk[dt][2] * p.scale) | ((unsigned)f2bf(dk[dt][3] * p.scale) << 16);
      *(uint2*)(dst + KD + dt * 16 + 4 * grp) = u;
    }
    const T* ex = p.dv_extra ? (const T*)p.dv_extra + ((long)b * p.N + k0 + li) * p.esw + h * HD : nullptr;
#pragma unroll
    for (int ct = 0; ct < D::NTH; ++ct) {
      if (ct * 16 + 4 * grp >= HD) continue;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (ex) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = bf2f(ex[ct * 16 + 4 * grp + r]);
      }
      uint2 u;
      u.x = (unsigned)f2bf(dv[ct][0] + e[0]) | ((unsigned)f2bf(dv[ct][1] + e[1]) << 16);
      u.y = (unsigned)f2bf(dv[ct][2] + e[2]) | ((unsigned)f2bf(dv[ct][3] + e[3]) << 16);
      *(uint2*)(dst + 2 * KD + ct * 16 + 4 * grp) = u;
    }
  }
}

}  // namespace

extern "C" {

int y3d_attn_fwd(int dtype, const void* qkv, int64_t qsw, void* out, int64_t osw, float* lse, int B, int N, int nh, int kd, int hd,
                 float scale, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "attn_fwd: bad dtype");
  Y3D_CHECK((kd == 32 && hd == 64) || (kd == 36 && hd == 72), "attn_fwd: unsupported head dims kd=%d hd=%d (32/64 and 36/72 built)", kd, hd);
  AttnP p{qkv, (long)qsw, out, (long)osw, lse, B, N, nh, kd, hd, scale};
  dim3 grid(cdiv(N, 128), B * nh), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == Y3D_BF16 && y3d_get_tile_kernels() && qsw % 4 == 0 && osw % 4 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 7) == 0) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, KC * 128);
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<36, 72>, hipFuncAttributeMaxDynamicSharedMemorySize, KC * 256);
      attr_set = true;
    }
    if (kd == 32) hipLaunchKernelGGL((attn_fwd_mfma_kernel<32, 64>), dim3(cdiv(N, 64), B * nh), dim3(256), KC * 128, st, p);
    else hipLaunchKernelGGL((attn_fwd_mfma_kernel<36, 72>), dim3(cdiv(N, 64), B * nh), dim3(256), KC * 256, st, p);
    Y3D_LAUNCH_CHECK();
    return Y3D_OK;
  }
  if (dtype == Y3D_BF16) {
    if (kd == 32) hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 32, 64>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 36, 72>), grid, block, 0, st, p);
  } else {
    if (kd == 32) hipLaunchKernelGGL((attn_fwd_kernel<float, 32, 64>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<float, 36, 72>), grid, block, 0, st, p);
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_attn_bwd(int dtype, const void* qkv, int64_t qsw, const void* out, int64_t osw, const void* dout, int64_t dsw,
                 const void* dv_extra, int64_t esw, const float* lse, float* delta, void* dqkv, int64_t gsw, int B, int N, int nh,
                 int kd, int hd, float scale, void* stream) {
  Y3D_CHECK(dtype == Y3D_BF16 || dtype == Y3D_F32, "attn_bwd: bad dtype");
  Y3D_CHECK((kd == 32 && hd == 64) || (kd == 36 && hd == 72), "attn_bwd: unsupported head dims kd=%d hd=%d", kd, hd);
  AttnBP p{qkv, (long)qsw, out, (long)osw, dout, (long)dsw, dv_extra, (long)esw, dqkv, (long)gsw, lse, delta, B, N, nh, kd, hd, scale, 0};
  dim3 grid(cdiv(N, 128), B * nh), block(128);
  hipStream_t st = (hipStream_t)stream;
  const int n32 = (N + 31) & ~31;
  if (dtype == Y3D_BF16 && y3d_get_tile_kernels() && qsw % 4 == 0 && osw % 4 == 0 && dsw % 4 == 0 && gsw % 4 == 0 && (dv_extra == nullptr || esw % 1 == 0) &&
      ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)dout & 7) == 0 && ((uintptr_t)dqkv & 7) == 0) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_q_mfma_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_bwd_kv_mfma_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_bwd_q_mfma_kernel<36, 72>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_bwd_kv_mfma_kernel<36, 72>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    // LDS images of at most 128 KB: 1024 keys (128-byte rows); 512 queries (256-byte rows) or 256 (512-byte rows) per pass
    const dim3 g2(cdiv(N, 64), B * nh);
    p.chunk = n32 < 1024 ? n32 : 1024;
    if (kd == 32) hipLaunchKernelGGL((attn_bwd_q_mfma_kernel<32, 64>), g2, dim3(256), (size_t)p.chunk * 128, st, p);
    else hipLaunchKernelGGL((attn_bwd_q_mfma_kernel<36, 72>), g2, dim3(256), (size_t)p.chunk * 128, st, p);
    if (kd == 32) {
      p.chunk = n32 < 512 ? n32 : 512;
      hipLaunchKernelGGL((attn_bwd_kv_mfma_kernel<32, 64>), g2, dim3(256), (size_t)p.chunk * (256 + 8), st, p);
    } else {
      p.chunk = n32 < 256 ? n32 : 256;
      hipLaunchKernelGGL((attn_bwd_kv_mfma_kernel<36, 72>), g2, dim3(256), (size_t)p.chunk * (512 + 8), st, p);
    }
    Y3D_LAUNCH_CHECK();
    return Y3D_OK;
  }
#define ATT_BWD(T, KD, HD)                                                              \
  hipLaunchKernelGGL((attn_bwd_q_kernel<T, KD, HD>), grid, block, 0, st, p);           \
  hipLaunchKernelGGL((attn_bwd_kv_kernel<T, KD, HD>), grid, block, 0, st, p)
  if (dtype == Y3D_BF16) {
    if (kd == 32) { ATT_BWD(bf16_t, 32, 64); } else { ATT_BWD(bf16_t, 36, 72); }
  } else {
    if (kd == 32) { ATT_BWD(float, 32, 64); } else { ATT_BWD(float, 36, 72); }
  }
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
