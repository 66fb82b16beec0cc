// Fused jet kernel for the ResNet architecture (pinnrl/neural_networks/resnet.py:9-142), gfx950.
//
//   h0 = act(W_in (x,t) + b_in)
//   block b:  y1 = LN1(W1 h + b1);  a1 = act(y1);  y2 = LN2(W2 a1 + b2);  h <- act(h + y2)
//   u  = w_out . h + b_out
//
// Same stream-serial structure as jet_kernel.h (activations of all K streams in the accumulator layout, one
// stream at a time published in LDS for the MFMA GEMMs, widths up to 256), plus what LayerNorm needs:
// a LayerNorm over the feature axis couples all features of a point, and its jets couple the streams
// (tests/jet_model.py::ln_fwd): per point, the means of the K streams, then var and the first/second
// derivative moments  v1_d = 2 mean(c0 c1_d),  v2_d = 2 mean(c1_d^2 + c0 c2_d)  per direction d.  Features are
// split over the 4 waves, so each LayerNorm costs two cross-wave reductions (in-lane over the owned rows,
// one shuffle across lane halves, then 4 partials through LDS).  The reverse sweep mirrors it with two more
// reductions (rbar / r1bar / r2bar, then the re-centring means) and per-feature sums for dgamma / dbeta.
// Derivative orders above 2 through a LayerNorm are not supported (the BASELINE ResNet config is
// Allen-Cahn: NT = 1, NX = 2).
#pragma once
#include "jet_kernel.h"

namespace pinn {

// layer table convention for PINN_ARCH_RESNET: layer[2b] = block b's first Linear (+ LN1), layer[2b+1] = its
// second Linear (+ LN2); the input Linear is the ENC_LINEAR encoding; `ln_*` pointers live in LayerDev.

constexpr int kMaxMom = 8;  // moments reduced at once (K means, or 1 + first + second order moments)

// all-features sum of per-lane partials q[0..NQ): after this every lane of every wave holds the sums for ITS point
template <int NQ>
__device__ __forceinline__ void cross_wave_sum(float (&q)[NQ], float* red, const Lane& L) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) q[i] += __shfl_xor(q[i], 32);
  if (L.lh == 0) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) red[(L.wave * kMaxMom + i) * kT + L.ln] = q[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += red[(w * kMaxMom + i) * kT + L.ln];
    q[i] = s;
  }
}

// per-point LayerNorm statistics of the K streams (needed by both sweeps)
template <int NT, int NX>
struct LnStats {
  float mu[1 + NT + NX];
  float r;                       // rsqrt(var + eps)
  float v1t, v2t, v1x, v2x;      // derivative moments per direction (orders 1 and 2)
};

// centre v in place (v <- c = z - mean) and return the statistics; two cross-wave reductions
template <int NT, int NX, int NTILE>
__device__ __forceinline__ void ln_stats(f32x16 (&v)[NTILE][1 + NT + NX], int dim, float eps, float* red0, float* red1,
                                         LnStats<NT, NX>& S, const Lane& L) {
  constexpr int K = 1 + NT + NX;
  static_assert(NT <= 2 && NX <= 2, "LayerNorm jets are implemented up to second order");
  const float invH = 1.0f / (float)dim;
  float q[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    float p = 0.0f;
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt)
      if ((L.wave + kWaves * jt) * 32 < dim) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p += v[jt][s][r];
      }
    q[s] = p;
  }
  cross_wave_sum<K>(q, red0, L);
#pragma unroll
  for (int s = 0; s < K; ++s) {
    S.mu[s] = q[s] * invH;
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt)
      if ((L.wave + kWaves * jt) * 32 < dim) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[jt][s][r] -= S.mu[s];
      }
  }
  float m[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // sum c0^2 | c0 c1t | c1t^2 + c0 c2t | c0 c1x | c1x^2 + c0 c2x
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt)
    if ((L.wave + kWaves * jt) * 32 < dim) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float c0 = v[jt][0][r];
        m[0] = fmaf(c0, c0, m[0]);
        if constexpr (NT >= 1) m[1] = fmaf(c0, v[jt][1][r], m[1]);
        if constexpr (NT >= 2) m[2] += fmaf(v[jt][1][r], v[jt][1][r], c0 * v[jt][2][r]);
        if constexpr (NX >= 1) m[3] = fmaf(c0, v[jt][1 + NT][r], m[3]);
        if constexpr (NX >= 2) m[4] += fmaf(v[jt][1 + NT][r], v[jt][1 + NT][r], c0 * v[jt][2 + NT][r]);
      }
    }
  cross_wave_sum<5>(m, red1, L);
  S.r = rsqrtf(m[0] * invH + eps);
  S.v1t = 2.0f * m[1] * invH;
  S.v2t = 2.0f * m[2] * invH;
  S.v1x = 2.0f * m[3] * invH;
  S.v2x = 2.0f * m[4] * invH;
}

// normalised jets from centred streams: yhat_0 = c0 r; yhat_1 = c1 r + c0 r1; yhat_2 = c2 r + 2 c1 r1 + c0 r2
template <int M>
__device__ __forceinline__ void ln_dir_fwd(float c0, const float* c, float r, float v1, float v2, float* y) {
  const float r3 = r * r * r;
  const float r1 = -0.5f * r3 * v1;
  if constexpr (M >= 1) y[0] = c[0] * r + c0 * r1;
  if constexpr (M >= 2) {
    const float r2 = 0.75f * r3 * r * r * v1 * v1 - 0.5f * r3 * v2;
    y[1] = c[1] * r + 2.0f * c[0] * r1 + c0 * r2;
  }
}

// v (centred) -> gamma * yhat + beta, in place
template <int NT, int NX, int NTILE>
__device__ __forceinline__ void ln_apply(f32x16 (&v)[NTILE][1 + NT + NX], int dim, const LnStats<NT, NX>& S,
                                         const float* gamma, const float* beta, const Lane& L) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int ft = L.wave + kWaves * jt;
    if (ft * 32 < dim) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = ft * 32 + acc_row(r, L.lh);
        const float g = gamma[f], b = beta[f];
        float c[K], y[K];
#pragma unroll
        for (int s = 0; s < K; ++s) c[s] = v[jt][s][r];
        y[0] = c[0] * S.r;
        ln_dir_fwd<NT>(c[0], c + 1, S.r, S.v1t, S.v2t, y + 1);
        ln_dir_fwd<NX>(c[0], c + 1 + NT, S.r, S.v1x, S.v2x, y + 1 + NT);
        v[jt][0][r] = fmaf(g, y[0], b);
#pragma unroll
        for (int s = 1; s < K; ++s) v[jt][s][r] = g * y[s];
      }
    }
  }
}

// Reverse of LayerNorm.  In: c = centred pre-LN jets (from ln_stats on the taped z), yb = cotangent of the LN
// output.  Out: yb <- zbar (cotangent of the pre-LN jets); gsum/bsum <- per-element dgamma / dbeta contributions
// (summed over points by the caller).  Two cross-wave reductions.
template <int NT, int NX, int NTILE>
__device__ __forceinline__ void ln_backward(f32x16 (&c)[NTILE][1 + NT + NX], f32x16 (&yb)[NTILE][1 + NT + NX],
                                            f32x16 (&gsum)[NTILE], f32x16 (&bsum)[NTILE], int dim,
                                            const LnStats<NT, NX>& S, const float* gamma, float* red0, float* red1,
                                            const Lane& L) {
  constexpr int K = 1 + NT + NX;
  const float invH = 1.0f / (float)dim;
  const float r = S.r, r2_ = r * r, r3 = r2_ * r, r5 = r3 * r2_;
  const float r1t = -0.5f * r3 * S.v1t, r1x = -0.5f * r3 * S.v1x;
  const float r2t = 0.75f * r5 * S.v1t * S.v1t - 0.5f * r3 * S.v2t;
  const float r2x = 0.75f * r5 * S.v1x * S.v1x - 0.5f * r3 * S.v2x;
  // phase A: yhat-bar = gamma * ybar; dgamma/dbeta contributions; sums rbar, r1bar_d, r2bar_d
  float q[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // rbar | r1bar_t | r2bar_t | r1bar_x | r2bar_x
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int ft = L.wave + kWaves * jt;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      gsum[jt][rr] = 0.0f;
      bsum[jt][rr] = 0.0f;
    }
    if (ft * 32 < dim) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int f = ft * 32 + acc_row(rr, L.lh);
        const float g = gamma[f];
        float cc[K], y[K], hb[K];
#pragma unroll
        for (int s = 0; s < K; ++s) cc[s] = c[jt][s][rr];
        y[0] = cc[0] * r;
        ln_dir_fwd<NT>(cc[0], cc + 1, r, S.v1t, S.v2t, y + 1);
        ln_dir_fwd<NX>(cc[0], cc + 1 + NT, r, S.v1x, S.v2x, y + 1 + NT);
        float gs = 0.0f;
        bsum[jt][rr] = yb[jt][0][rr];  // dbeta contribution = ybar_0
#pragma unroll
        for (int s = 0; s < K; ++s) {
          gs = fmaf(yb[jt][s][rr], y[s], gs);
          hb[s] = g * yb[jt][s][rr];
          yb[jt][s][rr] = hb[s];  // now holds yhat-bar
        }
        gsum[jt][rr] = gs;
        q[0] = fmaf(hb[0], cc[0], q[0]);
        if constexpr (NT >= 1) { q[0] = fmaf(hb[1], cc[1], q[0]); q[1] = fmaf(hb[1], cc[0], q[1]); }
        if constexpr (NT >= 2) { q[0] = fmaf(hb[2], cc[2], q[0]); q[1] = fmaf(2.0f * hb[2], cc[1], q[1]); q[2] = fmaf(hb[2], cc[0], q[2]); }
        if constexpr (NX >= 1) { q[0] = fmaf(hb[1 + NT], cc[1 + NT], q[0]); q[3] = fmaf(hb[1 + NT], cc[0], q[3]); }
        if constexpr (NX >= 2) { q[0] = fmaf(hb[2 + NT], cc[2 + NT], q[0]); q[3] = fmaf(2.0f * hb[2 + NT], cc[1 + NT], q[3]); q[4] = fmaf(hb[2 + NT], cc[0], q[4]); }
      }
    }
  }
  cross_wave_sum<5>(q, red0, L);
  // r = v^-1/2, r1 = -1/2 r^3 v1, r2 = 3/4 r^5 v1^2 - 1/2 r^3 v2
  const float v2bt = -0.5f * r3 * q[2], v2bx = -0.5f * r3 * q[4];
  const float v1bt = -0.5f * r3 * q[1] + 1.5f * r5 * S.v1t * q[2];
  const float v1bx = -0.5f * r3 * q[3] + 1.5f * r5 * S.v1x * q[4];
  const float rtot = q[0] + q[1] * (-1.5f * r2_ * S.v1t) + q[2] * (3.75f * r2_ * r2_ * S.v1t * S.v1t - 1.5f * r2_ * S.v2t) +
                     q[3] * (-1.5f * r2_ * S.v1x) + q[4] * (3.75f * r2_ * r2_ * S.v1x * S.v1x - 1.5f * r2_ * S.v2x);
  const float vb = rtot * (-0.5f * r3);
  const float k2 = 2.0f * invH;
  // phase B: cbar, then re-centre (zbar = cbar - mean(cbar))
  float mq[K];
#pragma unroll
  for (int s = 0; s < K; ++s) mq[s] = 0.0f;
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int ft = L.wave + kWaves * jt;
    if (ft * 32 < dim) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        float cc[K], hb[K], cb[K];
#pragma unroll
        for (int s = 0; s < K; ++s) {
          cc[s] = c[jt][s][rr];
          hb[s] = yb[jt][s][rr];
        }
        cb[0] = hb[0] * r + vb * k2 * cc[0];
        if constexpr (NT >= 1) {
          cb[0] += hb[1] * r1t + v1bt * k2 * cc[1];
          cb[1] = hb[1] * r + v1bt * k2 * cc[0];
        }
        if constexpr (NT >= 2) {
          cb[0] += hb[2] * r2t + v2bt * k2 * cc[2];
          cb[1] += 2.0f * hb[2] * r1t + 2.0f * v2bt * k2 * cc[1];
          cb[2] = hb[2] * r + v2bt * k2 * cc[0];
        }
        if constexpr (NX >= 1) {
          cb[0] += hb[1 + NT] * r1x + v1bx * k2 * cc[1 + NT];
          cb[1 + NT] = hb[1 + NT] * r + v1bx * k2 * cc[0];
        }
        if constexpr (NX >= 2) {
          cb[0] += hb[2 + NT] * r2x + v2bx * k2 * cc[2 + NT];
          cb[1 + NT] += 2.0f * hb[2 + NT] * r1x + 2.0f * v2bx * k2 * cc[1 + NT];
          cb[2 + NT] = hb[2 + NT] * r + v2bx * k2 * cc[0];
        }
#pragma unroll
        for (int s = 0; s < K; ++s) {
          yb[jt][s][rr] = cb[s];
          mq[s] += cb[s];
        }
      }
    }
  }
  cross_wave_sum<K>(mq, red1, L);
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt)
    if ((L.wave + kWaves * jt) * 32 < dim) {
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) yb[jt][s][rr] -= mq[s] * invH;
    }
}

// ---------------------------------------------------------------------------
// building blocks of the sweep
// ---------------------------------------------------------------------------
// out[jt][s] = W[own rows] . in_s (+ bias on the value stream): one stream step per stream
// ACCUM: start from the values already in `out` (chunked accumulation) instead of zero; a null bias is skipped
template <int K, int NTILE, bool ACCUM = false>
__device__ __forceinline__ void linear_forward(f32x16 (&out)[NTILE][K], const f32x16 (&in)[NTILE][K], const LayerDev& Ly,
                                               float* SB, int sbuf, int& c, const Lane& L) {
#pragma unroll
  for (int s = 0; s < K; ++s) {
    float* S = SB + (c & 1) * sbuf;
    {
      f32x16 tmp[NTILE];
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) tmp[jt] = in[jt][s];
      stage_one<NTILE>(tmp, S, Ly.in_dim, L);
    }
    __syncthreads();
    f32x16 accs[NTILE];
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) {
      const int ft = L.wave + kWaves * jt;
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[jt][r] = ACCUM ? out[jt][s][r] : 0.0f;
      if (s == 0 && Ly.b && ft * 32 < Ly.out_dim) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(Ly.b + ft * 32 + 8 * q + 4 * L.lh);
#pragma unroll
          for (int i = 0; i < 4; ++i) accs[jt][4 * q + i] += bq[i];
        }
      }
    }
    gemm_rows<NTILE>(accs, Ly, S, L);
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) out[jt][s] = accs[jt];
    ++c;
  }
}

// reverse of a Linear: dW += zb a^T, db += sum zb_0, and (returned in abn) W^T zb — one stream step per stream
template <int K, int NTILE>
__device__ __forceinline__ void linear_backward(f32x16 (&abn)[NTILE][K], const f32x16 (&zb)[NTILE][K],
                                                const f32x16 (&ain)[NTILE][K], const LayerDev& Ly, float* SB, int sbuf,
                                                int& c, int tid, const Lane& L) {
  constexpr int NKT = 4 * NTILE;
#pragma unroll
  for (int jt0 = 0; jt0 < NTILE; ++jt0) {  // one owned output tile at a time keeps the dW accumulators at 64 * NTILE VGPRs
    f32x16 dacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
    const int ft0 = L.wave + kWaves * jt0;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      float* Z = SB + (c & 1) * sbuf;
      float* A2 = SB + (2 + (c & 1)) * sbuf;
      {
        f32x16 tz[NTILE], ta[NTILE];
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          tz[jt] = zb[jt][s];
          ta[jt] = ain[jt][s];
        }
        stage_one<NTILE>(tz, Z, Ly.out_dim, L);
        stage_one<NTILE>(ta, A2, Ly.in_dim, L);
      }
      __syncthreads();
      if (jt0 == 0 && s == 0 && Ly.db && tid < Ly.out_dim) atomicAdd(Ly.db + tid, row_sum(Z + tid * kTP));
      if (Ly.dW && ft0 * 32 < Ly.out_dim) gemm_outer<NKT>(dacc, ft0, Ly.in_dim, Z, A2, L);
      if (jt0 == 0) {
        f32x16 accs[NTILE];
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[jt][r] = 0.0f;
        gemm_cols<NTILE>(accs, Ly, Z, L);
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) abn[jt][s] = accs[jt];
      }
      ++c;
    }
    if (Ly.dW && ft0 * 32 < Ly.out_dim) {
      float* base = Ly.dW + (long long)(ft0 * 32 + 4 * L.lh) * Ly.ld + L.ln;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt * 32 < Ly.in_dim) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * Ly.ld + kt * 32, dacc[kt][r]);
        }
      }
    }
  }
}

// per-feature sums over the tile's points of two accumulator-layout quantities -> atomics into dgamma / dbeta
template <int NTILE>
__device__ __forceinline__ void ln_param_grads(const f32x16 (&gsum)[NTILE], const f32x16 (&bsum)[NTILE], int dim,
                                               float* dg, float* db, float* SB, int sbuf, int tid, const Lane& L) {
  float* G = SB + 2 * sbuf;  // the A2 pair is idle between stream loops; the callers' reductions fence its readers
  float* B = SB + 3 * sbuf;
  stage_one<NTILE>(gsum, G, dim, L);
  stage_one<NTILE>(bsum, B, dim, L);
  __syncthreads();
  if (tid < dim) {
    if (dg) atomicAdd(dg + tid, row_sum(G + tid * kTP));
    if (db) atomicAdd(db + tid, row_sum(B + tid * kTP));
  }
  if (dim > kThreads) {  // never (width <= 256)
  }
  __syncthreads();
}

template <int ACT, int NT, int NX, int NTILE, bool BWD>
__global__ __launch_bounds__(kThreads, 1) void jet_kernel_resnet(const KernelArgs a) {
  constexpr int K = 1 + NT + NX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDev& net = a.net;
  const int hmax = net.hmax;
  const int sbuf = hmax * kTP;
  float* SB = smem;                              // (BWD ? 4 : 2) stream buffers
  float* RED = SB + (BWD ? 4 : 2) * sbuf;        // output-layer reduction: kWaves * K * kT
  float* U = RED + kWaves * K * kT;
  float* UB = U + K * kT;
  float* xin = UB + K * kT;                      // kMaxDin * kT
  float* red0 = xin + kMaxDin * kT;              // LayerNorm reductions: 2 x kWaves * kMaxMom * kT
  float* red1 = red0 + kWaves * kMaxMom * kT;

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int tid = L.tid;
  const int din = net.din;
  const int H = net.enc_out;  // constant width of a ResNet
  const int nb = net.n_layers >> 1;
  const float eps = net.ln_eps;
  const long long ntiles = (a.N + kT - 1) / kT;
  float* tape = BWD ? a.tape + (long long)blockIdx.x * a.tape_stride : nullptr;
  int c = 0;

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long p0 = tile * kT;
    __syncthreads();
    if (tid < kT) {
      const long long p = p0 + tid;
      const bool ok = p < a.N;
      for (int cc = 0; cc < din - 1; ++cc) xin[cc * kT + tid] = ok ? a.x[p * (din - 1) + cc] : 0.0f;
      xin[(din - 1) * kT + tid] = ok ? a.t[p] : 0.0f;
    }
    __syncthreads();

    f32x16 v[NTILE][K];
    encode_regs<ACT, NT, NX, NTILE>(net, xin, v, L);
    if constexpr (BWD) {
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
        if ((L.wave + kWaves * jt) * 32 < H) tape_put<K, NTILE>(v[jt], tape, 0, jt, tid);
    }

    // ---- residual blocks ----
    for (int b = 0; b < nb; ++b) {
      const LayerDev L1 = uniform_layer(net.layer[2 * b]);
      const LayerDev L2 = uniform_layer(net.layer[2 * b + 1]);
      const int slot = 1 + 4 * b;
      f32x16 y[NTILE][K];
      linear_forward<K, NTILE>(y, v, L1, SB, sbuf, c, L);
      if constexpr (BWD) {
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
          if ((L.wave + kWaves * jt) * 32 < H) tape_put<K, NTILE>(y[jt], tape, slot + 0, jt, tid);
      }
      LnStats<NT, NX> S1;
      ln_stats<NT, NX, NTILE>(y, H, eps, red0, red1, S1, L);
      ln_apply<NT, NX, NTILE>(y, H, S1, L1.ln_g, L1.ln_b, L);
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
        if ((L.wave + kWaves * jt) * 32 < H) ew_forward<ACT, NT, NX, NTILE, BWD>(y[jt], L1.act_param, tape, slot + 1, jt, tid);
      f32x16 y2[NTILE][K];
      linear_forward<K, NTILE>(y2, y, L2, SB, sbuf, c, L);
      if constexpr (BWD) {
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
          if ((L.wave + kWaves * jt) * 32 < H) tape_put<K, NTILE>(y2[jt], tape, slot + 2, jt, tid);
      }
      LnStats<NT, NX> S2;
      ln_stats<NT, NX, NTILE>(y2, H, eps, red0, red1, S2, L);
      ln_apply<NT, NX, NTILE>(y2, H, S2, L2.ln_g, L2.ln_b, L);
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
        for (int s = 0; s < K; ++s) v[jt][s] += y2[jt][s];  // skip connection: q = h + y2
        if ((L.wave + kWaves * jt) * 32 < H) ew_forward<ACT, NT, NX, NTILE, BWD>(v[jt], L1.act_param, tape, slot + 3, jt, tid);
      }
    }

    // ---- output layer ----
#pragma unroll
    for (int s = 0; s < K; ++s) {
      float p = 0.0f;
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int ft = L.wave + kWaves * jt;
        if (ft * 32 < net.h_last) {
#pragma unroll
          for (int r = 0; r < 16; ++r) p = fmaf(net.w_out[ft * 32 + acc_row(r, L.lh)], v[jt][s][r], p);
        }
      }
      p += __shfl_xor(p, 32);
      if (L.lh == 0) RED[(L.wave * K + s) * kT + L.ln] = p;
    }
    __syncthreads();
    if (tid < K * kT) {
      const int s = tid / kT;
      float u = (s == 0) ? net.b_out[0] : 0.0f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) u += RED[w * K * kT + tid];
      U[tid] = u;
    }
    __syncthreads();

    // ---- epilogue ----
    if (tid < kT) {
      const long long p = p0 + tid;
      const bool ok = p < a.N;
      float j[K];
#pragma unroll
      for (int s = 0; s < K; ++s) j[s] = U[s * kT + tid];
      if (a.mode == MODE_JETS) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
          if (ok && a.jets_out[s]) a.jets_out[s][p] = j[s];
          if constexpr (BWD) UB[s * kT + tid] = (ok && a.jets_bar[s]) ? a.jets_bar[s][p] : 0.0f;
        }
      } else {
        float d[K];
        const float r = pde_residual<NT, NX>(a.pde, j, xin[tid], d);
        float dl;
        float lt = loss_term(a.pde, r, &dl);
        if (!ok) {
          lt = 0.0f;
          dl = 0.0f;
        }
        if (ok && a.residual_out) a.residual_out[p] = r;
        if (a.loss_sum) {
          float sacc = lt;
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
          if (tid == 0) atomicAdd(a.loss_sum, sacc);
        }
        if constexpr (BWD) {
          const float rb = a.res_bar ? (ok ? a.res_bar[p] : 0.0f) : a.grad_scale * dl;
#pragma unroll
          for (int s = 0; s < K; ++s) UB[s * kT + tid] = rb * d[s];
        }
      }
    }

    if constexpr (BWD) {
      __syncthreads();
      // ---- output layer reverse ----
      f32x16 ab[NTILE][K];
      {
        float ub[K];
#pragma unroll
        for (int s = 0; s < K; ++s) ub[s] = UB[s * kT + L.ln];
        float* S = SB + (c & 1) * sbuf;
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = L.wave + kWaves * jt;
          const bool on = ft * 32 < net.h_last;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = ft * 32 + acc_row(r, L.lh);
            float pr = 0.0f;
#pragma unroll
            for (int s = 0; s < K; ++s) pr = fmaf(ub[s], v[jt][s][r], pr);
            if (on) S[f * kTP + L.ln] = pr;
            const float wv = on ? net.w_out[f] : 0.0f;
#pragma unroll
            for (int s = 0; s < K; ++s) ab[jt][s][r] = wv * ub[s];
          }
        }
        __syncthreads();
        if (tid < net.h_last && net.dw_out) atomicAdd(net.dw_out + tid, row_sum(S + tid * kTP));
        if (L.wave == 3 && net.db_out) {
          float g = L.lh == 0 ? UB[L.ln] : 0.0f;
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) g += __shfl_xor(g, o);
          if ((tid & 63) == 0) atomicAdd(net.db_out, g);
        }
        ++c;
      }

      for (int b = nb - 1; b >= 0; --b) {
        const LayerDev L1 = uniform_layer(net.layer[2 * b]);
        const LayerDev L2 = uniform_layer(net.layer[2 * b + 1]);
        const int slot = 1 + 4 * b;
        // qbar = act_bwd(tape q, hbar)
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
          if ((L.wave + kWaves * jt) * 32 < H) ew_backward<ACT, NT, NX, NTILE>(ab[jt], L1.act_param, tape, slot + 3, jt, tid);
        f32x16 skipb[NTILE][K];
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
          for (int s = 0; s < K; ++s) skipb[jt][s] = ab[jt][s];
        f32x16 cz[NTILE][K], gsum[NTILE], bsum[NTILE], ap[NTILE][K], abn[NTILE][K];
        // LN2 reverse (pre-LN jets z2 from the tape)
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) cz[jt][s][r] = 0.0f;
          if ((L.wave + kWaves * jt) * 32 < H) tape_get<K, NTILE>(cz[jt], tape, slot + 2, jt, tid);
        }
        LnStats<NT, NX> S2;
        ln_stats<NT, NX, NTILE>(cz, H, eps, red0, red1, S2, L);
        ln_backward<NT, NX, NTILE>(cz, ab, gsum, bsum, H, S2, L2.ln_g, red0, red1, L);
        ln_param_grads<NTILE>(gsum, bsum, H, L2.d_ln_g, L2.d_ln_b, SB, sbuf, tid, L);
        // second Linear reverse: a1 replayed from its activation tape
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) ap[jt][s][r] = 0.0f;
          if ((L.wave + kWaves * jt) * 32 < H) ew_replay<ACT, NT, NX, NTILE>(ap[jt], L1.act_param, tape, slot + 1, jt, tid);
        }
        linear_backward<K, NTILE>(abn, ab, ap, L2, SB, sbuf, c, tid, L);
        // y1bar = act_bwd(tape y1, a1bar)
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
          for (int s = 0; s < K; ++s) ab[jt][s] = abn[jt][s];
          if ((L.wave + kWaves * jt) * 32 < H) ew_backward<ACT, NT, NX, NTILE>(ab[jt], L1.act_param, tape, slot + 1, jt, tid);
        }
        // LN1 reverse
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) cz[jt][s][r] = 0.0f;
          if ((L.wave + kWaves * jt) * 32 < H) tape_get<K, NTILE>(cz[jt], tape, slot + 0, jt, tid);
        }
        LnStats<NT, NX> S1;
        ln_stats<NT, NX, NTILE>(cz, H, eps, red0, red1, S1, L);
        ln_backward<NT, NX, NTILE>(cz, ab, gsum, bsum, H, S1, L1.ln_g, red0, red1, L);
        ln_param_grads<NTILE>(gsum, bsum, H, L1.d_ln_g, L1.d_ln_b, SB, sbuf, tid, L);
        // first Linear reverse: its input h = previous block's output (replayed) or the encoding (taped)
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) ap[jt][s][r] = 0.0f;
          if ((L.wave + kWaves * jt) * 32 < H) {
            if (b > 0) {
              ew_replay<ACT, NT, NX, NTILE>(ap[jt], L1.act_param, tape, slot - 1, jt, tid);  // block b-1's outer activation tape
            } else {
              tape_get<K, NTILE>(ap[jt], tape, 0, jt, tid);
            }
          }
        }
        linear_backward<K, NTILE>(abn, ab, ap, L1, SB, sbuf, c, tid, L);
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
          for (int s = 0; s < K; ++s) ab[jt][s] = abn[jt][s] + skipb[jt][s];
      }

      // ---- input Linear reverse ----
      if (net.d_encW) {
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = L.wave + kWaves * jt;
          if (ft * 32 < H) ew_enc_backward<ACT, NT, NX>(ab[jt], net, xin, ft, L);
        }
        float* S0 = SB + 0 * sbuf;
        float* S1 = SB + 1 * sbuf;
        float* S2 = BWD ? SB + 2 * sbuf : SB;
        __syncthreads();
        {
          f32x16 t0[NTILE], t1[NTILE], t2[NTILE];
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
            t0[jt] = ab[jt][0];
            t1[jt] = ab[jt][NT >= 1 ? 1 : 0];
            t2[jt] = ab[jt][NX >= 1 ? 1 + NT : 0];
          }
          stage_one<NTILE>(t0, S0, H, L);
          if constexpr (NT >= 1) stage_one<NTILE>(t1, S1, H, L);
          if constexpr (NX >= 1) stage_one<NTILE>(t2, S2, H, L);
        }
        __syncthreads();
        if (tid < H) {
          float gb = 0.0f, gt = 0.0f, gx = 0.0f;
          float gw[kMaxDin] = {0.0f, 0.0f, 0.0f, 0.0f};
          for (int n = 0; n < kT; ++n) {
            const float vv = S0[tid * kTP + n];
            gb += vv;
#pragma unroll
            for (int cc = 0; cc < kMaxDin; ++cc)
              if (cc < din) gw[cc] = fmaf(vv, xin[cc * kT + n], gw[cc]);
            if constexpr (NT >= 1) gt += S1[tid * kTP + n];
            if constexpr (NX >= 1) gx += S2[tid * kTP + n];
          }
#pragma unroll
          for (int cc = 0; cc < kMaxDin; ++cc)
            if (cc < din)
              atomicAdd(net.d_encW + tid * din + cc, gw[cc] + (cc == din - 1 ? gt : 0.0f) + (cc == 0 ? gx : 0.0f));
          if (net.d_encb) atomicAdd(net.d_encb + tid, gb);
        }
      }
    }
  }
}

inline size_t jet_resnet_lds_bytes(int K, int hmax, bool bwd) {
  return sizeof(float) * ((size_t)(bwd ? 4 : 2) * hmax * kTP + (size_t)kWaves * K * kT + 2 * K * kT + kMaxDin * kT +
                          2 * (size_t)kWaves * kMaxMom * kT);
}

inline long long jet_resnet_tape_floats_per_wg(int K, int n_blocks, int ntile) {
  return (long long)(1 + 4 * n_blocks) * ntile * K * 16 * kThreads;
}

template <int NT, int NX>
hipError_t launch_jet_resnet(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  constexpr int K = 1 + NT + NX;
  if constexpr (NT > 2 || NX > 2) {
    return hipErrorNotSupported;  // LayerNorm jets are implemented up to second order
  } else {
    const int ntile = a.net.hmax > 128 ? 2 : 1;
    const size_t lds = jet_resnet_lds_bytes(K, a.net.hmax, bwd);
    const int act = a.net.enc_act;
    hipError_t e = hipSuccess;
#define PINN_RLAUNCH1(ACT_, NTILE_, BWD_)                                                                    \
  do {                                                                                                       \
    auto kern = jet_kernel_resnet<ACT_, NT, NX, NTILE_, BWD_>;                                               \
    e = allow_full_lds(reinterpret_cast<const void*>(kern));                                                 \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, a);                                    \
  } while (0)
#define PINN_RLAUNCH(ACT_)                                                               \
  if (ntile == 1) { if (bwd) PINN_RLAUNCH1(ACT_, 1, true); else PINN_RLAUNCH1(ACT_, 1, false); } \
  else { if (bwd) PINN_RLAUNCH1(ACT_, 2, true); else PINN_RLAUNCH1(ACT_, 2, false); }
    switch (act) {  // resnet.py builds its blocks from relu | leaky_relu | tanh | sigmoid | gelu
      case PINN_ACT_TANH: PINN_RLAUNCH(PINN_ACT_TANH) break;
#ifndef PINN_DEV
      case PINN_ACT_GELU: PINN_RLAUNCH(PINN_ACT_GELU) break;
      case PINN_ACT_SIGMOID: PINN_RLAUNCH(PINN_ACT_SIGMOID) break;
      default: PINN_RLAUNCH(PINN_ACT_RELU) break;
#else
      default: return hipErrorInvalidValue;
#endif
    }
#undef PINN_RLAUNCH
#undef PINN_RLAUNCH1
    return hipGetLastError();
  }
}

}  // namespace pinn
