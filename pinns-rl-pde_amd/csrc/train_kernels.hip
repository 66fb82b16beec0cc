// Small kernels of the captured training step (include/pinn_jet.h, "training step" section): the point-wise loss
// terms of PDEBase.compute_loss on the boundary / initial points, and gradient clipping + Adam over ONE flat
// parameter buffer.  Everything else of a step is the jet engine's residual launch; with these two the whole step is
// free of autograd and runs as a handful of launches inside a HIP graph.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/pinn_jet.h"

namespace {

struct PointTerms {
  int n_terms;
  int lo[PINN_MAX_POINT_TERMS], hi[PINN_MAX_POINT_TERMS];
  const float* target[PINN_MAX_POINT_TERMS];
  float weight[PINN_MAX_POINT_TERMS];
  int loss;
  float huber_delta;
};

__device__ __forceinline__ float loss_val(int kind, float d, float r, float* dl) {
  if (kind == PINN_LOSS_MAE) {
    *dl = r > 0.0f ? 1.0f : (r < 0.0f ? -1.0f : 0.0f);
    return fabsf(r);
  }
  if (kind == PINN_LOSS_HUBER) {
    const float a = fabsf(r);
    if (a < d) {
      *dl = r;
      return 0.5f * r * r;
    }
    *dl = r > 0.0f ? d : -d;
    return d * (a - 0.5f * d);
  }
  *dl = 2.0f * r;
  return r * r;
}

// one workgroup; term k: losses[k] = mean_{n in [lo,hi)} l(u[n] - target_k[n]); cot[n] += weight_k l'(.) / (hi - lo)
__global__ __launch_bounds__(256) void point_loss_kernel(const float* u, PointTerms p, int n_total, float* losses, float* cot,
                                                         const float* residual_sum, float residual_scale, float residual_weight,
                                                         int n_boundary_terms, float* summary4) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  for (int n = tid; n < n_total; n += 256) cot[n] = 0.0f;
  __syncthreads();
  for (int k = 0; k < p.n_terms; ++k) {
    const int cnt = p.hi[k] - p.lo[k];
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.0f;
    float acc = 0.0f;
    for (int n = p.lo[k] + tid; n < p.hi[k]; n += 256) {
      float dl;
      acc += loss_val(p.loss, p.huber_delta, u[n] - p.target[k][n - p.lo[k]], &dl);
      cot[n] += p.weight[k] * dl * inv;  // terms are processed one after another: no two threads share n within a term
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) losses[k] = red[0] * inv;
    __syncthreads();
  }
  if (tid == 0 && summary4) {  // {residual, boundary, initial, total} of PDEBase.compute_loss (pde_base.py:1168-1235)
    const float res = residual_sum ? residual_sum[0] * residual_scale : 0.0f;
    float bnd = 0.0f, ini = 0.0f, tot = residual_weight * res;
    for (int k = 0; k < p.n_terms; ++k) {
      if (k < n_boundary_terms) bnd += losses[k];
      else ini += losses[k];
      tot += p.weight[k] * losses[k];
    }
    summary4[0] = res;
    summary4[1] = bnd;
    summary4[2] = ini;
    summary4[3] = tot;
  }
}

// The general form: jets J[stream][n] (K x n_total), term k on stream `stream[k]`, either against a target array or —
// pair[k] != 0 — as the difference of two point ranges, d_i = J[s][lo + i] - J[s][lo + pair + i] (periodic boundary
// conditions: value and d/dx at paired wall points, heat_equation.py:420-445).  cot is K x n_total, overwritten.
struct JetTerms {
  int n_terms;
  int lo[PINN_MAX_POINT_TERMS], hi[PINN_MAX_POINT_TERMS], stream[PINN_MAX_POINT_TERMS], pair[PINN_MAX_POINT_TERMS];
  const float* target[PINN_MAX_POINT_TERMS];
  float weight[PINN_MAX_POINT_TERMS];
  int loss;
  float huber_delta;
};

__global__ __launch_bounds__(256) void jet_loss_kernel(const float* J, int K, JetTerms p, int n_total, float* losses, float* cot,
                                                       const float* residual_sum, float residual_scale, float residual_weight,
                                                       int n_boundary_terms, float* summary4) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  for (int n = tid; n < K * n_total; n += 256) cot[n] = 0.0f;
  __syncthreads();
  for (int k = 0; k < p.n_terms; ++k) {
    const int cnt = p.hi[k] - p.lo[k];
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.0f;
    const float* Js = J + (long long)p.stream[k] * n_total;
    float* cs = cot + (long long)p.stream[k] * n_total;
    float acc = 0.0f;
    for (int n = p.lo[k] + tid; n < p.hi[k]; n += 256) {
      float dl;
      const float other = p.pair[k] ? Js[n + p.pair[k]] : p.target[k][n - p.lo[k]];
      acc += loss_val(p.loss, p.huber_delta, Js[n] - other, &dl);
      const float c = p.weight[k] * dl * inv;
      cs[n] += c;  // terms run one after another and a term's two ranges are disjoint: no two threads share an address
      if (p.pair[k]) cs[n + p.pair[k]] -= c;
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) losses[k] = red[0] * inv;
    __syncthreads();
  }
  if (tid == 0 && summary4) {
    const float res = residual_sum ? residual_sum[0] * residual_scale : 0.0f;
    float bnd = 0.0f, ini = 0.0f, tot = residual_weight * res;
    for (int k = 0; k < p.n_terms; ++k) {
      if (k < n_boundary_terms) bnd += losses[k];
      else ini += losses[k];
      tot += p.weight[k] * losses[k];
    }
    summary4[0] = res;
    summary4[1] = bnd;
    summary4[2] = ini;
    summary4[3] = tot;
  }
}

constexpr int kNormBlocks = 64;

// partial sums of squares in a fixed order (deterministic): block b sums elements b*256+tid, +64*256, ...
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long long n, float* partial) {
  __shared__ float red[256];
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)kNormBlocks * 256) acc = fmaf(g[i], g[i], acc);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// torch.nn.utils.clip_grad_norm_(max_norm) followed by torch.optim.Adam(lr, betas, eps, weight_decay).step()
// (pinnrl/training/trainer.py:686-698) over one flat buffer.  `step` holds the number of steps taken so far.
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n, const float* lr,
                                                   float beta1, float beta2, float eps, float wd, float max_norm,
                                                   float* step, const float* partial, float* norm_out) {
  float tot = 0.0f;
  for (int i = 0; i < kNormBlocks; ++i) tot += partial[i];  // every thread, same order
  const float norm = sqrtf(tot);
  float coef = 1.0f;
  if (max_norm > 0.0f) {
    coef = max_norm / (norm + 1e-6f);
    coef = coef < 1.0f ? coef : 1.0f;
  }
  const float t = step[0] + 1.0f;
  const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
  const float step_size = lr[0] / bc1, rs2 = rsqrtf(bc2);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    const float pi = p[i];
    if (wd != 0.0f) gi = fmaf(wd, pi, gi);
    const float mi = fmaf(1.0f - beta1, gi - m[i], m[i]);           // torch: m.lerp_(g, 1 - beta1)
    const float vi = fmaf(beta2, v[i], (1.0f - beta2) * gi * gi);   // torch: v.mul_(beta2).addcmul_(g, g, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * rs2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
}

__global__ void step_inc_kernel(float* step) { step[0] += 1.0f; }

}  // namespace

extern "C" int pinn_internal_fail(int code, const char* msg);  // pinn_abi.hip: sets pinn_last_error()

static int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PINN_OK;
  char msg[256];
  snprintf(msg, sizeof(msg), "HIP error %d: %s (%s)", (int)e, hipGetErrorString(e), what);
  return pinn_internal_fail(PINN_ERR_HIP, msg);
}

extern "C" {

int pinn_point_losses(const float* u, int32_t n_total, int32_t n_terms, const int32_t* lo, const int32_t* hi,
                      const float* const* targets, const float* weights, int32_t loss, float huber_delta,
                      float* term_losses, float* cotangent, const float* residual_sum, float residual_scale,
                      float residual_weight, int32_t n_boundary_terms, float* summary4, void* stream) {
  if (!u || !lo || !hi || !targets || !weights || !term_losses || !cotangent) return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_point_losses: null argument");
  if (n_terms < 0 || n_terms > PINN_MAX_POINT_TERMS || n_total < 0) return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_point_losses: term count / point count out of range");
  PointTerms p;
  p.n_terms = n_terms;
  for (int k = 0; k < n_terms; ++k) {
    if (lo[k] < 0 || hi[k] < lo[k] || hi[k] > n_total || !targets[k]) return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_point_losses: bad term range or null target");
    p.lo[k] = lo[k];
    p.hi[k] = hi[k];
    p.target[k] = targets[k];
    p.weight[k] = weights[k];
  }
  p.loss = loss;
  p.huber_delta = huber_delta;
  hipLaunchKernelGGL(point_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), u, p, n_total, term_losses, cotangent,
                     residual_sum, residual_scale, residual_weight, n_boundary_terms, summary4);
  return launched("point_loss_kernel");
}

int pinn_jet_losses(const float* jets, int32_t n_streams, int32_t n_total, int32_t n_terms, const int32_t* lo, const int32_t* hi,
                    const int32_t* stream_of, const int32_t* pair_offset, const float* const* targets, const float* weights,
                    int32_t loss, float huber_delta, float* term_losses, float* cotangent, const float* residual_sum,
                    float residual_scale, float residual_weight, int32_t n_boundary_terms, float* summary4, void* stream) {
  if (!jets || !lo || !hi || !stream_of || !pair_offset || !targets || !weights || !term_losses || !cotangent)
    return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_jet_losses: null argument");
  if (n_terms < 0 || n_terms > PINN_MAX_POINT_TERMS || n_total < 0 || n_streams < 1 || n_streams > PINN_MAX_STREAMS)
    return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_jet_losses: term / stream / point count out of range");
  JetTerms p;
  p.n_terms = n_terms;
  for (int k = 0; k < n_terms; ++k) {
    if (lo[k] < 0 || hi[k] < lo[k] || hi[k] > n_total || stream_of[k] < 0 || stream_of[k] >= n_streams)
      return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_jet_losses: bad term range or stream");
    if (pair_offset[k]) {  // the partner range must lie inside the jets and must not overlap the term's own range
      if (pair_offset[k] < hi[k] - lo[k] || hi[k] + pair_offset[k] > n_total)
        return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_jet_losses: paired range overlaps its partner or leaves the jets");
    } else if (!targets[k]) {
      return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_jet_losses: null target of an unpaired term");
    }
    p.lo[k] = lo[k];
    p.hi[k] = hi[k];
    p.stream[k] = stream_of[k];
    p.pair[k] = pair_offset[k];
    p.target[k] = targets[k];
    p.weight[k] = weights[k];
  }
  p.loss = loss;
  p.huber_delta = huber_delta;
  hipLaunchKernelGGL(jet_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), jets, n_streams, p, n_total, term_losses,
                     cotangent, residual_sum, residual_scale, residual_weight, n_boundary_terms, summary4);
  return launched("jet_loss_kernel");
}

int pinn_adam_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr,
                        float beta1, float beta2, float eps, float weight_decay, float max_norm, float* step,
                        float* scratch64, float* grad_norm_out, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !lr || !step || !scratch64 || n <= 0)
    return pinn_internal_fail(PINN_ERR_BAD_DESC, "pinn_adam_clip_step: null argument or n <= 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, st, grads, (long long)n, scratch64);
  int rc = launched("sumsq_kernel");
  if (rc) return rc;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (long long)n, lr, beta1,
                     beta2, eps, weight_decay, max_norm, step, scratch64, grad_norm_out);
  if ((rc = launched("adam_kernel"))) return rc;
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step);
  return launched("step_inc_kernel");
}

}  // extern "C"
