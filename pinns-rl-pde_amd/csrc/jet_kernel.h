// Fused jet kernel for the plain-MLP family (feedforward / fourier / SIREN), gfx950.
//
// One 256-thread workgroup (4 waves) owns a TILE of T = 32 collocation points and
// pushes all K = 1 + NT + NX derivative streams of those points through the whole
// network without leaving the CU:
//
//   LDS       X[s][f][n]  (K x Hmax x 36 floats, row = one feature of one stream, 32 points + 4 pad)
//   MFMA      v_mfma_f32_32x32x2_f32, exact fp32.  D[f][n] = sum_k W[f][k] * a_s[k][n]:
//             A operand = weight rows, streamed from L2 as 16-byte loads (one load feeds
//             4 k-steps x K streams = 4K MFMAs); B operand = ds_read_b32 along the point axis.
//             Wave w owns output-feature tiles {w, w+4}: the accumulator (feature rows in
//             registers, point columns on lanes) is exactly the layout the activation jets
//             and the next layer's B operand want, so nothing is transposed.
//   reverse   (BWD) per tile, straight after the forward and the PDE epilogue:
//             zbar -> LDS Z, a_{l-1} (recomputed from the tape) -> LDS A2, then
//             dW[j][k] = sum_{s,n} Z_s[j][n] A2_s[k][n]  (both operands ds_read_b128 along n)
//             flushed with 2x128-byte-row float atomics, and abar_{l-1} = W^T zbar (W columns
//             read as 128-byte coalesced dwords).  Three GEMMs per layer, equal MFMA counts.
//   tape      pre-activation jets of every hidden layer, in accumulator layout, private to the
//             workgroup (written and re-read by the same lanes; L2/MALL resident).
//
// Algorithmic FLOPs per point: K * 2 * sum(in*out) forward, 3x that with the reverse sweep
// (SURVEY.md §8d).  tests/jet_model.py is the executable specification.
#pragma once
#include <hip/hip_runtime.h>

#include "jet_device.h"

namespace pinn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 32;        // points per tile
constexpr int kTP = 36;       // padded LDS row (floats): conflict-free b32 column and b128 row reads
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kMaxLayers = PINN_MAX_LINEAR;
constexpr int kMaxDin = 4;

enum { ENC_LINEAR = 0, ENC_FOURIER = 1 };
enum { MODE_JETS = 0, MODE_PDE = 1 };

struct LayerDev {
  const float* W;  // (out_dim, in_dim) row-major — torch.nn.Linear.weight
  const float* b;  // (out_dim)
  float* dW;       // nullable
  float* db;       // nullable
  int in_dim;      // multiple of 8
  int out_dim;     // multiple of 32
  int act;
  float act_param;
};

struct NetDev {
  int enc;          // ENC_LINEAR: first Linear (din -> enc_out) + activation; ENC_FOURIER: [sin, cos](inp @ B)
  int din;          // input_dim (time = last column)
  int enc_out;      // features after the encoding (multiple of 8)
  const float* encW;  // ENC_LINEAR: (enc_out, din); ENC_FOURIER: B (din, enc_out / 2)
  const float* encb;
  float* d_encW;
  float* d_encb;
  int enc_act;
  float enc_param;
  int n_layers;  // MFMA layers
  LayerDev layer[kMaxLayers];
  const float* w_out;  // (1, H_last)
  const float* b_out;  // (1)
  float* dw_out;
  float* db_out;
  int h_last;
  int hmax;  // max feature count over encoding + layers, rounded up to 32
};

struct KernelArgs {
  NetDev net;
  PdeDev pde;
  const float* x;  // (N, din-1)
  const float* t;  // (N, 1)
  long long N;
  int mode;
  float grad_scale;                          // MODE_PDE backward: cotangent of sum_n l(r_n)
  float* jets_out[PINN_MAX_STREAMS];         // MODE_JETS
  const float* jets_bar[PINN_MAX_STREAMS];   // MODE_JETS backward
  float* residual_out;                       // MODE_PDE, nullable
  float* loss_sum;                           // MODE_PDE, nullable
  const float* res_bar;                      // MODE_PDE backward, nullable: external cotangent of r (N floats)
  float* tape;                               // BWD workspace
  long long tape_stride;                     // floats per workgroup
};

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// ---------------------------------------------------------------------------
// Encoding layer: (x, t) -> K streams of enc_out features, written to dst[s][f][n].
// With KEEP_Z (backward of ENC_LINEAR) the pre-activation jets are what the caller needs,
// so this is also used in "z only" form by enc_backward below.
// ---------------------------------------------------------------------------
template <int NT, int NX>
__device__ __forceinline__ void encode(const NetDev& net, const float* xin, float* dst, int hmax, int tid) {
  constexpr int K = 1 + NT + NX;
  const int n = tid & 31;
  const int din = net.din;
  if (net.enc == ENC_FOURIER) {
    const int M = net.enc_out >> 1;
    for (int m = tid >> 5; m < M; m += 8) {
      float z[K];
#pragma unroll
      for (int s = 0; s < K; ++s) z[s] = 0.0f;
      float v = 0.0f;
      for (int c = 0; c < din; ++c) v = fmaf(xin[c * kT + n], net.encW[c * M + m], v);
      z[0] = v;
      if constexpr (NT >= 1) z[1] = net.encW[(din - 1) * M + m];
      if constexpr (NX >= 1) z[1 + NT] = net.encW[m];
      float a[K];
      act_fwd<PINN_ACT_SIN, NT, NX>(1.0f, z, a);
#pragma unroll
      for (int s = 0; s < K; ++s) dst[(s * hmax + m) * kTP + n] = a[s];
      // cos(z) = sin(z + pi/2) has the same derivative recursion: evaluate with exact cos/sin instead of shifting z
      float sn, cs;
      sincosf(v, &sn, &cs);
      float f[6] = {cs, -sn, -cs, sn, cs, -sn};
      float b[K];
      b[0] = f[0];
      dir_fwd<NT>(f, z + 1, b + 1);
      dir_fwd<NX>(f, z + 1 + NT, b + 1 + NT);
#pragma unroll
      for (int s = 0; s < K; ++s) dst[(s * hmax + M + m) * kTP + n] = b[s];
    }
  } else {
    const int H = net.enc_out;
    for (int f0 = tid >> 5; f0 < H; f0 += 8) {
      float z[K];
#pragma unroll
      for (int s = 0; s < K; ++s) z[s] = 0.0f;
      float v = net.encb[f0];
      for (int c = 0; c < din; ++c) v = fmaf(xin[c * kT + n], net.encW[f0 * din + c], v);
      z[0] = v;
      if constexpr (NT >= 1) z[1] = net.encW[f0 * din + din - 1];
      if constexpr (NX >= 1) z[1 + NT] = net.encW[f0 * din];
      float a[K];
      PINN_ACT_SWITCH(net.enc_act, act_fwd<ACT, NT, NX>(net.enc_param, z, a);)
#pragma unroll
      for (int s = 0; s < K; ++s) dst[(s * hmax + f0) * kTP + n] = a[s];
    }
  }
}


// ---------------------------------------------------------------------------
// Element-wise stages on one accumulator tile (32 features x 32 points x K streams per wave)
// ---------------------------------------------------------------------------
__device__ __forceinline__ long long tape_idx(int l, int jt, int ntile, int K, int s, int r, int tid) {
  return ((((long long)(l * ntile + jt) * K + s) * 16 + r) * kThreads) + tid;
}

// forward: z = acc -> (tape) -> y = act jets -> dst[s][f][n]
template <int ACT, int NT, int NX, int NTILE, bool TAPE>
__device__ __forceinline__ void ew_forward(const f32x16 (&acc)[1 + NT + NX], float w, float* dst, int hmax, int ft,
                                           int ln, int lh, float* tape, int l, int jt, int tid) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float z[K], y[K];
#pragma unroll
    for (int s = 0; s < K; ++s) z[s] = acc[s][r];
    if constexpr (TAPE) {
#pragma unroll
      for (int s = 0; s < K; ++s) tape[tape_idx(l, jt, NTILE, K, s, r, tid)] = z[s];
    }
    act_fwd<ACT, NT, NX>(w, z, y);
    const int f = ft * 32 + acc_row(r, lh);
#pragma unroll
    for (int s = 0; s < K; ++s) dst[(s * hmax + f) * kTP + ln] = y[s];
  }
}

// reverse S1: ab <- act_bwd(z from tape, ab)
template <int ACT, int NT, int NX, int NTILE>
__device__ __forceinline__ void ew_backward(f32x16 (&ab)[1 + NT + NX], float w, const float* tape, int l, int jt,
                                            int tid) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float z[K], abv[K], zb[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      z[s] = tape[tape_idx(l, jt, NTILE, K, s, r, tid)];
      abv[s] = ab[s][r];
    }
    act_bwd<ACT, NT, NX>(w, z, abv, zb);
#pragma unroll
    for (int s = 0; s < K; ++s) ab[s][r] = zb[s];
  }
}

// reverse S2: dst[s][f][n] <- act jets of the taped pre-activation of layer l
template <int ACT, int NT, int NX, int NTILE>
__device__ __forceinline__ void ew_replay(float w, const float* tape, int l, int jt, int tid, float* dst, int hmax,
                                          int ft, int ln, int lh) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float z[K], y[K];
#pragma unroll
    for (int s = 0; s < K; ++s) z[s] = tape[tape_idx(l, jt, NTILE, K, s, r, tid)];
    act_fwd<ACT, NT, NX>(w, z, y);
    const int f = ft * 32 + acc_row(r, lh);
#pragma unroll
    for (int s = 0; s < K; ++s) dst[(s * hmax + f) * kTP + ln] = y[s];
  }
}

// reverse, first Linear (din -> H): recompute z from the coordinates, zbar -> dst
template <int ACT, int NT, int NX>
__device__ __forceinline__ void ew_enc_backward(const f32x16 (&ab)[1 + NT + NX], const NetDev& net, const float* xin,
                                                float* dst, int hmax, int ft, int ln, int lh) {
  constexpr int K = 1 + NT + NX;
  const int din = net.din;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = ft * 32 + acc_row(r, lh);
    float z[K], abv[K], zb[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      z[s] = 0.0f;
      abv[s] = ab[s][r];
    }
    float v = net.encb[f];
    for (int c = 0; c < din; ++c) v = fmaf(xin[c * kT + ln], net.encW[f * din + c], v);
    z[0] = v;
    if constexpr (NT >= 1) z[1] = net.encW[f * din + din - 1];
    if constexpr (NX >= 1) z[1 + NT] = net.encW[f * din];
    act_bwd<ACT, NT, NX>(net.enc_param, z, abv, zb);
#pragma unroll
    for (int s = 0; s < K; ++s) dst[(s * hmax + f) * kTP + ln] = zb[s];
  }
}

// ---------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------
template <int NT, int NX, int NTILE, bool BWD>
__global__ __launch_bounds__(kThreads) void jet_kernel(const KernelArgs a) {
  constexpr int K = 1 + NT + NX;
  constexpr int NKT = 4 * NTILE;  // k-tiles (of 32 input features) a dW accumulator row can span
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDev& net = a.net;
  const int hmax = net.hmax;
  float* X = smem;                                   // K * hmax * kTP  (forward activations; Z = zbar in the reverse sweep)
  float* A2 = X + (BWD ? K * hmax * kTP : 0);        // K * hmax * kTP  (reverse sweep: a_{l-1})
  float* U = A2 + K * hmax * kTP;                    // K * kT          (output jets)
  float* UB = U + K * kT;                            // K * kT          (their cotangents)
  float* xin = UB + K * kT;                          // kMaxDin * kT

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, lh = lane >> 5;
  const int din = net.din;
  const long long ntiles = (a.N + kT - 1) / kT;
  float* tape = BWD ? a.tape + (long long)blockIdx.x * a.tape_stride : nullptr;

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long p0 = tile * kT;
    // ---- stage the tile's coordinates (zeros beyond N: finite values, masked cotangents) ----
    if (tid < kT) {
      const long long p = p0 + tid;
      const bool ok = p < a.N;
      for (int c = 0; c < din - 1; ++c) xin[c * kT + tid] = ok ? a.x[p * (din - 1) + c] : 0.0f;
      xin[(din - 1) * kT + tid] = ok ? a.t[p] : 0.0f;
    }
    __syncthreads();
    encode<NT, NX>(net, xin, X, hmax, tid);
    __syncthreads();

    // ---- hidden layers ----
    for (int l = 0; l < net.n_layers; ++l) {
      const LayerDev& L = net.layer[l];
      f32x16 acc[NTILE][K];
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int ft = wave + kWaves * jt;
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[jt][s][r] = 0.0f;
        if (ft * 32 < L.out_dim) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[jt][0][r] = L.b[ft * 32 + acc_row(r, lh)];
        }
      }
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int ft = wave + kWaves * jt;
        if (ft * 32 < L.out_dim) {
          const float* wrow = L.W + (long long)(ft * 32 + ln) * L.in_dim + 4 * lh;
          const float* xcol = X + (4 * lh) * kTP + ln;
          for (int g = 0; g < L.in_dim; g += 8) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
              for (int s = 0; s < K; ++s) {
                const float b = xcol[(s * hmax + g + i) * kTP];
                acc[jt][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i], b, acc[jt][s], 0, 0, 0);
              }
            }
          }
        }
      }
      __syncthreads();  // every wave has finished reading X: overwrite in place
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int ft = wave + kWaves * jt;
        if (ft * 32 < L.out_dim) {
          PINN_ACT_SWITCH(L.act, ew_forward<ACT, NT, NX, NTILE, BWD>(acc[jt], L.act_param, X, hmax, ft, ln, lh, tape,
                                                                     l, jt, tid);)
        }
      }
      __syncthreads();
    }

    // ---- output layer (H_last -> 1): wave w reduces streams w, w+4 ----
    {
      const int H = net.h_last, half = H >> 1;
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        const int s = wave + kWaves * si;
        if (s < K) {
          float p = 0.0f;
          const float* col = X + (s * hmax + lh * half) * kTP + ln;
          const float* wv = net.w_out + lh * half;
          for (int k = 0; k < half; ++k) p = fmaf(wv[k], col[k * kTP], p);
          p += __shfl_xor(p, 32);
          if (lh == 0) U[s * kT + ln] = p + (s == 0 ? net.b_out[0] : 0.0f);
        }
      }
    }
    __syncthreads();

    // ---- epilogue: jets out, or PDE residual + loss; cotangents of the jets for the reverse sweep ----
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
        if (!ok) { lt = 0.0f; dl = 0.0f; }
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
      // ---- B0: output layer.  dw_out[k] = sum_{s,n} ub_s[n] a_s[k][n];  abar = w_out (x) ub ----
      if (tid < net.h_last && net.dw_out) {
        float g = 0.0f;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const float* row = X + (s * hmax + tid) * kTP;
          for (int n = 0; n < kT; n += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + n);
            const f32x4 ub = *reinterpret_cast<const f32x4*>(UB + s * kT + n);
            g += v[0] * ub[0] + v[1] * ub[1] + v[2] * ub[2] + v[3] * ub[3];
          }
        }
        atomicAdd(net.dw_out + tid, g);
      }
      if (wave == 3 && net.db_out) {
        float g = lh == 0 ? UB[ln] : 0.0f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) g += __shfl_xor(g, o);
        if (lane == 0) atomicAdd(net.db_out, g);
      }
      f32x16 ab[NTILE][K];  // cotangent of the current layer's activation jets, accumulator layout
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int ft = wave + kWaves * jt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = ft * 32 + acc_row(r, lh);
          const float wv = f < net.h_last ? net.w_out[f] : 0.0f;
#pragma unroll
          for (int s = 0; s < K; ++s) ab[jt][s][r] = wv * UB[s * kT + ln];
        }
      }

      for (int l = net.n_layers - 1; l >= 0; --l) {
        const LayerDev& L = net.layer[l];
        // S1: zbar = act_bwd(z from tape, abar)  (registers only)
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = wave + kWaves * jt;
          if (ft * 32 < L.out_dim) {
            PINN_ACT_SWITCH(L.act, ew_backward<ACT, NT, NX, NTILE>(ab[jt], L.act_param, tape, l, jt, tid);)
          }
        }
        __syncthreads();  // previous readers of X(/Z) and A2 are done
        // S2: Z <- zbar;  A2 <- a_{l-1}
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = wave + kWaves * jt;
          if (ft * 32 < L.out_dim) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = ft * 32 + acc_row(r, lh);
#pragma unroll
              for (int s = 0; s < K; ++s) X[(s * hmax + f) * kTP + ln] = ab[jt][s][r];
            }
          }
        }
        if (l > 0) {
          const LayerDev& P = net.layer[l - 1];
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
            const int ft = wave + kWaves * jt;
            if (ft * 32 < P.out_dim) {
              PINN_ACT_SWITCH(P.act, ew_replay<ACT, NT, NX, NTILE>(P.act_param, tape, l - 1, jt, tid, A2, hmax, ft, ln,
                                                                    lh);)
            }
          }
        } else {
          encode<NT, NX>(net, xin, A2, hmax, tid);
        }
        __syncthreads();
        // S3: dW[j][k] += sum_{s,n} Z_s[j][n] * A2_s[k][n];  db[j] += sum_n Z_0[j][n]
        if (L.dW) {
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
            const int ft = wave + kWaves * jt;
            if (ft * 32 < L.out_dim) {
              f32x16 dacc[NKT];
#pragma unroll
              for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
#pragma unroll
              for (int s = 0; s < K; ++s) {
                const float* zrow = X + (s * hmax + ft * 32 + ln) * kTP + 4 * lh;
                const float* arow = A2 + (s * hmax + ln) * kTP + 4 * lh;
#pragma unroll
                for (int g = 0; g < kT; g += 8) {
                  const f32x4 zv = *reinterpret_cast<const f32x4*>(zrow + g);
#pragma unroll
                  for (int kt = 0; kt < NKT; ++kt) {
                    if (kt * 32 < L.in_dim) {
                      const f32x4 av = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP + g);
#pragma unroll
                      for (int i = 0; i < 4; ++i)
                        dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[i], av[i], dacc[kt], 0, 0, 0);
                    }
                  }
                }
              }
#pragma unroll
              for (int kt = 0; kt < NKT; ++kt) {
                if (kt * 32 < L.in_dim) {
#pragma unroll
                  for (int r = 0; r < 16; ++r)
                    atomicAdd(L.dW + (long long)(ft * 32 + acc_row(r, lh)) * L.in_dim + kt * 32 + ln, dacc[kt][r]);
                }
              }
            }
          }
        }
        if (L.db && tid < L.out_dim) {
          const float* row = X + tid * kTP;
          float g = 0.0f;
          for (int n = 0; n < kT; n += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + n);
            g += v[0] + v[1] + v[2] + v[3];
          }
          atomicAdd(L.db + tid, g);
        }
        // S4: abar_{l-1}[k][n] = sum_j W[j][k] zbar[j][n]   (needed unless the encoding below has no parameters)
        if (l > 0 || net.enc == ENC_LINEAR) {
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
            const int kt = wave + kWaves * jt;
#pragma unroll
            for (int s = 0; s < K; ++s)
#pragma unroll
              for (int r = 0; r < 16; ++r) ab[jt][s][r] = 0.0f;
            if (kt * 32 < L.in_dim) {
              const float* wcol = L.W + (long long)(4 * lh) * L.in_dim + kt * 32 + ln;
              const float* zcol = X + (4 * lh) * kTP + ln;
              for (int g = 0; g < L.out_dim; g += 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float wv = wcol[(long long)(g + i) * L.in_dim];
#pragma unroll
                  for (int s = 0; s < K; ++s) {
                    const float b = zcol[(s * hmax + g + i) * kTP];
                    ab[jt][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, b, ab[jt][s], 0, 0, 0);
                  }
                }
              }
            }
          }
        }
      }

      // ---- encoding backward (first Linear of feedforward / SIREN); the Fourier matrix B is a buffer ----
      if (net.enc == ENC_LINEAR && net.d_encW) {
        __syncthreads();
        const int H = net.enc_out;
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = wave + kWaves * jt;
          if (ft * 32 < H) {
            PINN_ACT_SWITCH(net.enc_act, ew_enc_backward<ACT, NT, NX>(ab[jt], net, xin, X, hmax, ft, ln, lh);)
          }
        }
        __syncthreads();
        if (tid < H) {
          float gb = 0.0f, gt = 0.0f, gx = 0.0f;
          float gw[kMaxDin] = {0.0f, 0.0f, 0.0f, 0.0f};
          const float* r0 = X + tid * kTP;
          for (int n = 0; n < kT; ++n) {
            const float v = r0[n];
            gb += v;
#pragma unroll
            for (int c = 0; c < kMaxDin; ++c)
              if (c < din) gw[c] = fmaf(v, xin[c * kT + n], gw[c]);
            if constexpr (NT >= 1) gt += X[(1 * hmax + tid) * kTP + n];
            if constexpr (NX >= 1) gx += X[((1 + NT) * hmax + tid) * kTP + n];
          }
#pragma unroll
          for (int c = 0; c < kMaxDin; ++c)
            if (c < din) atomicAdd(net.d_encW + tid * din + c, gw[c] + (c == din - 1 ? gt : 0.0f) + (c == 0 ? gx : 0.0f));
          if (net.d_encb) atomicAdd(net.d_encb + tid, gb);
        }
      }
    }
    __syncthreads();  // the next tile overwrites xin / X / U
  }
}

// ---------------------------------------------------------------------------
// Host-side launch helper, one instantiation per (NT, NX) lives in its own translation unit
// ---------------------------------------------------------------------------
inline size_t jet_lds_bytes(int K, int hmax, bool bwd) {
  return sizeof(float) * ((size_t)(bwd ? 2 : 1) * K * hmax * kTP + 2 * K * kT + kMaxDin * kT);
}

inline long long jet_tape_floats_per_wg(int K, int n_layers, int ntile) {
  return (long long)n_layers * ntile * K * 16 * kThreads;
}

template <int NT, int NX>
hipError_t launch_jet(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  constexpr int K = 1 + NT + NX;
  const int ntile = a.net.hmax > 128 ? 2 : 1;
  const size_t lds = jet_lds_bytes(K, a.net.hmax, bwd);
  hipError_t e = hipSuccess;
#define PINN_LAUNCH(NTILE_, BWD_)                                                                          \
  do {                                                                                                     \
    auto kern = jet_kernel<NT, NX, NTILE_, BWD_>;                                                          \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                     \
    if (e != hipSuccess) return e;                                                                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, a);                                  \
  } while (0)
  if (ntile == 1) {
    if (bwd) PINN_LAUNCH(1, true); else PINN_LAUNCH(1, false);
  } else {
    if (bwd) PINN_LAUNCH(2, true); else PINN_LAUNCH(2, false);
  }
#undef PINN_LAUNCH
  return hipGetLastError();
}

}  // namespace pinn
