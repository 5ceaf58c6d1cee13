// Optimizer side of the training step as three multi-tensor launches:
//   clip_grad_norm_(max_norm) (engine/trainer.py:570, torch.nn.utils.clip_grad_norm_) + SGD with Nesterov momentum and weight decay
//   (trainer.py:571, build_optimizer :734-790; torch.optim.SGD semantics, dampening 0).
// The ~570 parameter tensors are addressed through device tables (pointer, size) and a chunk table (tensor id, offset): one workgroup
// per 16 K-element chunk, 16-byte accesses.  HBM-bound: 16 B read + 8 B written per element for the update, 4 B read for the norm.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void mt_sqnorm_kernel(const long* __restrict__ gptr, const long* __restrict__ sizes, const int* __restrict__ ctensor,
                                                        const int* __restrict__ coff, int chunk, float* __restrict__ partials) {
  __shared__ float sh[256];
  const int t = ctensor[blockIdx.x];
  const long off = (long)coff[blockIdx.x] * chunk;
  const float* g = (const float*)gptr[t] + off;
  long n = sizes[t] - off;
  if (n > chunk) n = chunk;
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {
    long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 v = ((const float4*)g)[i];
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  } else {
    for (long i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// out[0] = total L2 norm, out[1] = min(1, max_norm / (norm + 1e-6)), out[2] = 1 if the norm is finite else 0 (the update kernels
// are no-ops for that step: GradScaler.step's skip, engine/trainer.py:569-572), out[3] += out[2] (number of steps applied so far),
// out[4] += 1 - out[2] (number of steps skipped)
__global__ __launch_bounds__(256) void mt_clip_kernel(const float* __restrict__ partials, int n, float max_norm, float* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float norm = (float)sqrt(sh[0]);
    float c = max_norm / (norm + 1e-6f);
    const bool ok = isfinite(norm);
    out[0] = norm;
    out[1] = c < 1.f ? c : 1.f;
    out[2] = ok ? 1.f : 0.f;
    out[3] += ok ? 1.f : 0.f;
    out[4] += ok ? 0.f : 1.f;
  }
}

__global__ __launch_bounds__(256) void mt_sgd_kernel(const long* __restrict__ pptr, const long* __restrict__ gptr, const long* __restrict__ bptr,
                                                     const long* __restrict__ sizes, const float* __restrict__ lr, const float* __restrict__ wd,
                                                     const int* __restrict__ ctensor, const int* __restrict__ coff, int chunk, float momentum,
                                                     int nesterov, int first, const float* __restrict__ clip) {
  const int t = ctensor[blockIdx.x];
  const long off = (long)coff[blockIdx.x] * chunk;
  float* p = (float*)pptr[t] + off;
  const float* g = (const float*)gptr[t] + off;
  float* b = (float*)bptr[t] + off;
  long n = sizes[t] - off;
  if (n > chunk) n = chunk;
  if (clip && clip[2] == 0.f) return;  // non-finite gradient norm: this step is skipped on every rank (the norm is that of the reduced buffer)
  const float c = clip ? clip[1] : 1.f, l = lr[t], w = wd[t];
  for (long i = threadIdx.x; i < n; i += 256) {
    float pv = p[i];
    float gv = g[i] * c;
    if (w != 0.f) gv += w * pv;
    float bv = first ? gv : momentum * b[i] + gv;
    b[i] = bv;
    float step = nesterov ? gv + momentum * bv : bv;
    p[i] = pv - l * step;
  }
}

// dst_t[i] = src_t[i] * scale for every tensor t of the table: gathers the step's gradient tensors into the flat all-reduce buffer
__global__ __launch_bounds__(256) void mt_copy_kernel(const long* __restrict__ sptr, const long* __restrict__ dptr, const long* __restrict__ sizes,
                                                      const int* __restrict__ ctensor, const int* __restrict__ coff, int chunk, float scale) {
  const int t = ctensor[blockIdx.x];
  const long off = (long)coff[blockIdx.x] * chunk;
  const float* s = (const float*)sptr[t] + off;
  float* d = (float*)dptr[t] + off;
  long n = sizes[t] - off;
  if (n > chunk) n = chunk;
  if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 v = ((const float4*)s)[i];
      ((float4*)d)[i] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) d[i] = s[i] * scale;
  } else {
    for (long i = threadIdx.x; i < n; i += 256) d[i] = s[i] * scale;
  }
}


// torch.optim.AdamW (amsgrad off, maximize off), one parameter chunk per workgroup.  bc1 = 1 - beta1^step, bc2s = sqrt(1 - beta2^step).
__global__ __launch_bounds__(256) void mt_adamw_kernel(const long* __restrict__ pptr, const long* __restrict__ gptr, const long* __restrict__ mptr,
                                                       const long* __restrict__ vptr, const long* __restrict__ sizes, const float* __restrict__ lr,
                                                       const float* __restrict__ wd, const int* __restrict__ ctensor, const int* __restrict__ coff,
                                                       int chunk, float beta1, float beta2, float eps, float bc1, float bc2s,
                                                       const float* __restrict__ clip) {
  const int t = ctensor[blockIdx.x];
  const long off = (long)coff[blockIdx.x] * chunk;
  float* p = (float*)pptr[t] + off;
  const float* g = (const float*)gptr[t] + off;
  float* m = (float*)mptr[t] + off;
  float* v = (float*)vptr[t] + off;
  long n = sizes[t] - off;
  if (n > chunk) n = chunk;
  if (clip) {
    if (clip[2] == 0.f) return;  // skipped step (non-finite gradient norm): state and step count stay
    // the step count of the bias corrections is the number of APPLIED steps, which only the device knows
    const double tstep = (double)clip[3];
    bc1 = (float)(1.0 - pow((double)beta1, tstep));
    bc2s = (float)sqrt(1.0 - pow((double)beta2, tstep));
  }
  const float c = clip ? clip[1] : 1.f, l = lr[t], w = wd[t];
  const float step_size = l / bc1, decay = 1.f - l * w, w1 = 1.f - beta1, w2 = 1.f - beta2;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float gv = g[i] * c;
    const float pv = p[i] * decay;
    const float mv = m[i] + w1 * (gv - m[i]);  // exp_avg.lerp_(grad, 1 - beta1)
    const float vv = v[i] * beta2 + w2 * gv * gv;
    m[i] = mv;
    v[i] = vv;
    p[i] = pv - step_size * (mv / (sqrtf(vv) / bc2s + eps));
  }
}

// ModelEMA.update (utils/torch_utils.py:431-443): e = e * d + (1 - d) * m, applied reps[t] times to tensor t (an EMA tensor that the
// reference's state_dict lists under several keys -- the aliased one-to-one head branches -- is updated once per key)
__global__ __launch_bounds__(256) void mt_ema_kernel(const long* __restrict__ eptr, const long* __restrict__ mptr, const long* __restrict__ sizes,
                                                     const int* __restrict__ reps, const int* __restrict__ ctensor, const int* __restrict__ coff,
                                                     int chunk, float d, float omd, const float* __restrict__ guard) {
  if (guard && guard[2] == 0.f) return;  // the optimizer step this update follows was skipped: the model did not move
  const int t = ctensor[blockIdx.x];
  const long off = (long)coff[blockIdx.x] * chunk;
  float* e = (float*)eptr[t] + off;
  const float* m = (const float*)mptr[t] + off;
  long n = sizes[t] - off;
  if (n > chunk) n = chunk;
  const int r = reps[t];
  for (long i = threadIdx.x; i < n; i += 256) {
    float ev = e[i];
    const float mv = omd * m[i];
    for (int k = 0; k < r; ++k) ev = ev * d + mv;
    e[i] = ev;
  }
}

}  // namespace

extern "C" {

int y3d_mt_copy(const int64_t* src_ptrs, const int64_t* dst_ptrs, const int64_t* sizes, const int* chunk_tensor, const int* chunk_off,
                int nchunks, int chunk, float scale, void* stream) {
  Y3D_CHECK(nchunks >= 1 && chunk >= 256, "mt_copy: empty chunk table");
  hipLaunchKernelGGL(mt_copy_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)src_ptrs, (const long*)dst_ptrs,
                     (const long*)sizes, chunk_tensor, chunk_off, chunk, scale);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}


int y3d_mt_sqnorm(const int64_t* grad_ptrs, const int64_t* sizes, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk,
                  float* partials, void* stream) {
  Y3D_CHECK(nchunks >= 1 && chunk >= 256, "mt_sqnorm: empty chunk table");
  hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)grad_ptrs, (const long*)sizes, chunk_tensor,
                     chunk_off, chunk, partials);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_mt_clip_coef(const float* partials, int nchunks, float max_norm, float* out_norm_clip, void* stream) {
  hipLaunchKernelGGL(mt_clip_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nchunks, max_norm, out_norm_clip);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_mt_sgd(const int64_t* param_ptrs, const int64_t* grad_ptrs, const int64_t* buf_ptrs, const int64_t* sizes, const float* lr,
               const float* wd, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk, float momentum, int nesterov,
               int first_step, const float* norm_clip, void* stream) {
  Y3D_CHECK(nchunks >= 1 && chunk >= 256, "mt_sgd: empty chunk table");
  hipLaunchKernelGGL(mt_sgd_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)param_ptrs, (const long*)grad_ptrs,
                     (const long*)buf_ptrs, (const long*)sizes, lr, wd, chunk_tensor, chunk_off, chunk, momentum, nesterov, first_step, norm_clip);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_mt_adamw(const int64_t* param_ptrs, const int64_t* grad_ptrs, const int64_t* m_ptrs, const int64_t* v_ptrs, const int64_t* sizes,
                 const float* lr, const float* wd, const int* chunk_tensor, const int* chunk_off, int nchunks, int chunk, float beta1, float beta2,
                 float eps, float bias_corr1, float bias_corr2_sqrt, const float* norm_clip, void* stream) {
  Y3D_CHECK(nchunks >= 1 && chunk >= 256, "mt_adamw: empty chunk table");
  Y3D_CHECK(bias_corr1 > 0.f && bias_corr2_sqrt > 0.f, "mt_adamw: bias corrections must be positive (step >= 1)");
  hipLaunchKernelGGL(mt_adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)param_ptrs, (const long*)grad_ptrs,
                     (const long*)m_ptrs, (const long*)v_ptrs, (const long*)sizes, lr, wd, chunk_tensor, chunk_off, chunk, beta1, beta2, eps,
                     bias_corr1, bias_corr2_sqrt, norm_clip);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

int y3d_mt_ema(const int64_t* ema_ptrs, const int64_t* model_ptrs, const int64_t* sizes, const int* reps, const int* chunk_tensor,
               const int* chunk_off, int nchunks, int chunk, float decay, float one_minus_decay, const float* guard, void* stream) {
  Y3D_CHECK(nchunks >= 1 && chunk >= 256, "mt_ema: empty chunk table");
  hipLaunchKernelGGL(mt_ema_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)ema_ptrs, (const long*)model_ptrs,
                     (const long*)sizes, reps, chunk_tensor, chunk_off, chunk, decay, one_minus_decay, guard);
  Y3D_LAUNCH_CHECK();
  return Y3D_OK;
}

}  // extern "C"
