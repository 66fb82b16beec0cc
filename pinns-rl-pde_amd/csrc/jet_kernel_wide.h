// "Wide" variant of the fused jet kernel: ALL K streams of a tile are resident in LDS at once.
//
// Used when K * Hmax * 36 * 4 B (twice that with the reverse sweep) fits the 160 KB LDS of a CU — e.g. the
// headline Burgers / fourier 4x128 workload (K = 4).  Compared with the stream-serial kernel (jet_kernel.h),
// which scales to any K and to width 256, this layout
//   * reuses every weight fragment for all K streams of a k-step (one A operand, K MFMAs),
//   * needs two barriers per layer instead of K,
//   * leaves the registers free for PERSISTENT weight-gradient accumulators: a workgroup keeps the dW tile rows
//     of up to kPersist layers in registers across ALL its tiles and issues its float atomics once, at the end
//     (6x fewer atomic bytes on 49 729 points; the per-tile flush ran at the memory-side atomic rate with every
//     wave stalled on it — 27 % of the kernel).
// Same arithmetic, same tape and argument structures as jet_kernel.h; tests run both variants.
#pragma once
#include "jet_kernel.h"

namespace pinn {

constexpr int kPersist = 3;  // layers whose dW accumulators stay in registers (3 x 64 VGPRs)

// encoding straight into an LDS activation image dst[s][f][n]; every thread takes (feature, point) pairs
template <int ACT, int NT, int NX>
__device__ __forceinline__ void encode_lds(const NetDev& net, const float* xin, float* dst, int hmax, int tid) {
  constexpr int K = 1 + NT + NX;
  const int n = tid & 31;
  if (net.enc == ENC_FOURIER) {
    const int M = net.enc_out >> 1;
    for (int m = tid >> 5; m < M; m += 8) {
      float z[K], ys[K], yc[K];
      enc_preact<NT, NX>(net, xin, m, n, z);
      float sn, cs;
      fast_sincosf(z[0], &sn, &cs);
      const float fs[6] = {sn, cs, -sn, -cs, sn, cs};
      const float fc[6] = {cs, -sn, -cs, sn, cs, -sn};
      ys[0] = sn;
      yc[0] = cs;
      dir_fwd<NT>(fs, z + 1, ys + 1);
      dir_fwd<NX>(fs, z + 1 + NT, ys + 1 + NT);
      dir_fwd<NT>(fc, z + 1, yc + 1);
      dir_fwd<NX>(fc, z + 1 + NT, yc + 1 + NT);
#pragma unroll
      for (int s = 0; s < K; ++s) {
        dst[(s * hmax + m) * kTP + n] = ys[s];
        dst[(s * hmax + M + m) * kTP + n] = yc[s];
      }
    }
  } else {
    const int H = net.enc_out;
    for (int f = tid >> 5; f < H; f += 8) {
      float z[K], y[K];
      enc_preact<NT, NX>(net, xin, f, n, z);
      act_fwd<ACT, NT, NX>(net.enc_param, z, y);
#pragma unroll
      for (int s = 0; s < K; ++s) dst[(s * hmax + f) * kTP + n] = y[s];
    }
  }
}

// acc[s] += frag . X[s][:][n] for all K streams; one weight fragment element feeds K MFMAs.
// B operands are fetched two k-groups (8 ds_read_b32 per stream) ahead of the MFMAs that consume them.
template <int K, int NG>
__device__ __forceinline__ void gemm_frag_wide_n(f32x16 (&acc)[K], const WFrag& wf, const float* X, int hmax,
                                                 const Lane& L) {
  const float* col = X + (4 * L.lh) * kTP + L.ln;
  float bc[2][4][K], bn[2][4][K];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < K; ++s) bc[g][i][s] = (g < NG) ? col[(s * hmax + 8 * g + i) * kTP] : 0.0f;
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += 2) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s)
          bn[g][i][s] = (g0 + 2 + g < NG) ? col[(s * hmax + 8 * (g0 + 2 + g) + i) * kTP] : 0.0f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g0 + g < NG) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < K; ++s)
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.g[g0 + g][i], bc[g][i][s], acc[s], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) bc[g][i][s] = bn[g][i][s];
  }
}

template <int K>
__device__ __forceinline__ void gemm_frag_wide(f32x16 (&acc)[K], const WFrag& wf, int depth, const float* X, int hmax,
                                               const Lane& L) {
  switch (depth >> 3) {  // wave-uniform
    case 16: gemm_frag_wide_n<K, 16>(acc, wf, X, hmax, L); break;
    case 12: gemm_frag_wide_n<K, 12>(acc, wf, X, hmax, L); break;
    case 8: gemm_frag_wide_n<K, 8>(acc, wf, X, hmax, L); break;
    case 4: gemm_frag_wide_n<K, 4>(acc, wf, X, hmax, L); break;
    default: {
      const float* col = X + (4 * L.lh) * kTP + L.ln;
#pragma unroll
      for (int g = 0; g < kMaxG; ++g) {
        if (g * 8 < depth) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < K; ++s)
              acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.g[g][i], col[(s * hmax + 8 * g + i) * kTP], acc[s], 0, 0, 0);
        }
      }
    }
  }
}

// dacc[kt] += sum_s Z_s[own rows][n] A_s[rows of tile kt][n]^T over all K streams
template <int K, int NA>
__device__ __forceinline__ void gemm_outer_wide_n(f32x16 (&dacc)[4], int ft, const float* Z, const float* A, int hmax,
                                                  const Lane& L) {
  const float* zrow = Z + (ft * 32 + L.ln) * kTP + 4 * L.lh;
  const float* arow = A + L.ln * kTP + 4 * L.lh;
#pragma unroll
  for (int s = 0; s < K; ++s) {
    f32x4 zv[4], av[4][NA];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      zv[g] = *reinterpret_cast<const f32x4*>(zrow + s * hmax * kTP + 8 * g);
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) av[g][kt] = *reinterpret_cast<const f32x4*>(arow + (s * hmax + kt * 32) * kTP + 8 * g);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kt = 0; kt < NA; ++kt)
          dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[g][i], av[g][kt][i], dacc[kt], 0, 0, 0);
  }
}

template <int K>
__device__ __forceinline__ void gemm_outer_wide(f32x16 (&dacc)[4], int ft, int in_dim, const float* Z, const float* A,
                                                int hmax, const Lane& L) {
  switch ((in_dim + 31) >> 5) {  // wave-uniform
    case 4: gemm_outer_wide_n<K, 4>(dacc, ft, Z, A, hmax, L); break;
    case 3: gemm_outer_wide_n<K, 3>(dacc, ft, Z, A, hmax, L); break;
    case 2: gemm_outer_wide_n<K, 2>(dacc, ft, Z, A, hmax, L); break;
    default: gemm_outer_wide_n<K, 1>(dacc, ft, Z, A, hmax, L); break;
  }
}

__device__ __forceinline__ void load_bias(f32x4 (&bias)[4], const LayerDev& Ly, int ft, const Lane& L) {
  const int row0 = (ft * 32 < Ly.out_dim ? ft : 0) * 32 + 4 * L.lh;
#pragma unroll
  for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const f32x4*>(Ly.b + row0 + 8 * q);
}

__device__ __forceinline__ void flush_rows(const f32x16 (&dacc)[4], const LayerDev& Ly, const Lane& L) {
  if (!Ly.dW || L.wave * 32 >= Ly.out_dim) return;
  float* base = Ly.dW + (long long)(L.wave * 32 + 4 * L.lh) * Ly.ld + L.ln;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt * 32 < Ly.in_dim) {
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * Ly.ld + kt * 32, dacc[kt][r]);
    }
  }
}

template <int ACT, int NT, int NX, bool BWD>
__global__ __launch_bounds__(kThreads, 1) void jet_kernel_wide(const KernelArgs a) {
  constexpr int K = 1 + NT + NX;
  constexpr int NTILE = 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDev& net = a.net;
  const int hmax = net.hmax;
  const int img = K * hmax * kTP;        // floats per activation image
  float* X = smem;                        // forward activations; zbar in the reverse sweep
  float* A2 = X + (BWD ? img : 0);        // reverse sweep: a_{l-1}
  float* U = A2 + img;                    // K * kT
  float* UB = U + K * kT;                 // K * kT
  float* xin = UB + K * kT;               // kMaxDin * kT
  float* wout_s = xin + kMaxDin * kT;     // hmax: the output layer's weight row, staged once per kernel

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int tid = L.tid;
  const int din = net.din;
  const int ft = L.wave;  // this wave's feature tile in every layer
  const long long ntiles = (a.N + kT - 1) / kT;
  float* tape = BWD ? a.tape + (long long)blockIdx.x * a.tape_stride : nullptr;
  PINN_STAMP_DECL

  // output-layer weights: LDS copy for the forward dot products, per-lane registers (accumulator layout) for abar
  for (int k = tid; k < hmax; k += kThreads) wout_s[k] = k < net.h_last ? net.w_out[k] : 0.0f;
  float wout_r[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = ft * 32 + acc_row(r, L.lh);
    wout_r[r] = (BWD && f < net.h_last) ? net.w_out[f] : 0.0f;
  }
  const float b_out0 = net.b_out[0];

  // persistent weight-gradient accumulators (reverse sweep): layer l < kPersist -> pacc[l]
  f32x16 pacc[kPersist][4];
  float pw_out = 0.0f;  // dw_out[tid]
  if constexpr (BWD) {
#pragma unroll
    for (int p = 0; p < kPersist; ++p)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[p][kt][r] = 0.0f;
  }

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long p0 = tile * kT;
    __syncthreads();  // previous tile's readers are done
    if (tid < kT) {
      const long long p = p0 + tid;
      const bool ok = p < a.N;
      for (int cc = 0; cc < din - 1; ++cc) xin[cc * kT + tid] = ok ? a.x[p * (din - 1) + cc] : 0.0f;
      xin[(din - 1) * kT + tid] = ok ? a.t[p] : 0.0f;
    }
    WFrag wf;
    f32x4 bias[4];  // this lane's 16 bias values of the NEXT layer to run (rows 8q + 4h .. + 3 are contiguous)
    if (net.n_layers > 0) {  // latency hides under the encoding
      const LayerDev L0 = uniform_layer(net.layer[0]);
      load_wrows(wf, L0, ft, L);
      load_bias(bias, L0, ft, L);
    }
    __syncthreads();
    PINN_STAMP(ST_STAGE);
    encode_lds<ACT, NT, NX>(net, xin, X, hmax, tid);
    __syncthreads();
    PINN_STAMP(ST_ENCODE);

    // ---- hidden layers ----
    for (int l = 0; l < net.n_layers; ++l) {
      const LayerDev Ly = uniform_layer(net.layer[l]);
      const bool on = ft * 32 < Ly.out_dim;
      f32x16 acc[K];
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
      if (on) {
        gemm_frag_wide<K>(acc, wf, Ly.in_dim, X, hmax, L);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[0][4 * q + i] += bias[q][i];
      }
      PINN_STAMP(ST_FWD_GEMM);
      if (l + 1 < net.n_layers) {  // next layer's fragment + bias: latency hides under the activation jets
        const LayerDev Ln = uniform_layer(net.layer[l + 1]);
        load_wrows(wf, Ln, ft, L);
        load_bias(bias, Ln, ft, L);
      }
      __syncthreads();  // every wave has finished reading X: overwrite in place
      if (on) {
        ew_forward<ACT, NT, NX, NTILE, BWD>(acc, Ly.act_param, tape, l, 0, tid);
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) X[(s * hmax + ft * 32 + acc_row(r, L.lh)) * kTP + L.ln] = acc[s][r];
      }
      __syncthreads();
      PINN_STAMP(ST_FWD_EW);
    }

    // ---- output layer (H_last -> 1): wave w reduces streams w, w+4 ----
    {
      const int half = net.h_last >> 1;
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        const int s = L.wave + kWaves * si;
        if (s < K) {
          float p = 0.0f;
          const float* col = X + (s * hmax + L.lh * half) * kTP + L.ln;
          const float* wv = wout_s + L.lh * half;
#pragma unroll 8
          for (int k = 0; k < half; ++k) p = fmaf(wv[k], col[k * kTP], p);
          p += __shfl_xor(p, 32);
          if (L.lh == 0) U[s * kT + L.ln] = p + (s == 0 ? b_out0 : 0.0f);
        }
      }
    }
    __syncthreads();
    PINN_STAMP(ST_OUT);

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
    PINN_STAMP(ST_EPI);

    if constexpr (BWD) {
      __syncthreads();
      // ---- B0: output layer.  dw_out[k] += sum_{s,n} ub_s[n] a_s[k][n] (kept per thread);  abar = w_out (x) ub ----
      if (tid < net.h_last) {
        float g = 0.0f;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const float* row = X + (s * hmax + tid) * kTP;
#pragma unroll
          for (int n = 0; n < kT; n += 4) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(row + n);
            const f32x4 ub = *reinterpret_cast<const f32x4*>(UB + s * kT + n);
            g += v4[0] * ub[0] + v4[1] * ub[1] + v4[2] * ub[2] + v4[3] * ub[3];
          }
        }
        pw_out += g;
      }
      if (L.wave == 3 && net.db_out) {
        float g = L.lh == 0 ? UB[L.ln] : 0.0f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) g += __shfl_xor(g, o);
        if ((tid & 63) == 0) atomicAdd(net.db_out, g);
      }
      f32x16 ab[K];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int s = 0; s < K; ++s) ab[s][r] = wout_r[r] * UB[s * kT + L.ln];
      }
      PINN_STAMP(ST_B0);

      for (int l = net.n_layers - 1; l >= 0; --l) {
        const LayerDev Ly = uniform_layer(net.layer[l]);
        const bool on = ft * 32 < Ly.out_dim;
        const bool need_abar = l > 0 || net.enc == ENC_LINEAR;
        const bool kon = ft * 32 < Ly.in_dim;  // this wave owns an input-feature tile of the layer
        if (need_abar) load_wcols(wf, Ly, ft, L);  // W^T slice; latency hides under the jets below
        if (on) ew_backward<ACT, NT, NX, NTILE>(ab, Ly.act_param, tape, l, 0, tid);
        __syncthreads();  // previous readers of X (zbar image) and A2 are done
        if (on) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[(s * hmax + ft * 32 + acc_row(r, L.lh)) * kTP + L.ln] = ab[s][r];
        }
        if (l > 0) {
          const LayerDev P = uniform_layer(net.layer[l - 1]);
          if (ft * 32 < P.out_dim) {
            f32x16 ap[K];
            ew_replay<ACT, NT, NX, NTILE>(ap, P.act_param, tape, l - 1, 0, tid);
#pragma unroll
            for (int s = 0; s < K; ++s)
#pragma unroll
              for (int r = 0; r < 16; ++r) A2[(s * hmax + ft * 32 + acc_row(r, L.lh)) * kTP + L.ln] = ap[s][r];
          }
        } else {
          encode_lds<ACT, NT, NX>(net, xin, A2, hmax, tid);
        }
        __syncthreads();
        PINN_STAMP(ST_BWD_EW);
        // dW (persistent for the first kPersist layers) and db
        if (Ly.db && tid < Ly.out_dim) atomicAdd(Ly.db + tid, row_sum(X + tid * kTP));
        if (on && Ly.dW) {
          if (l == 0) {
            gemm_outer_wide<K>(pacc[0], ft, Ly.in_dim, X, A2, hmax, L);
          } else if (l == 1) {
            gemm_outer_wide<K>(pacc[1], ft, Ly.in_dim, X, A2, hmax, L);
          } else if (l == 2) {
            gemm_outer_wide<K>(pacc[2], ft, Ly.in_dim, X, A2, hmax, L);
          } else {
            f32x16 dacc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
              for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
            gemm_outer_wide<K>(dacc, ft, Ly.in_dim, X, A2, hmax, L);
            flush_rows(dacc, Ly, L);
          }
        }
        // abar_{l-1} = W^T zbar for all streams
        if (need_abar) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) ab[s][r] = 0.0f;
          if (kon) gemm_frag_wide<K>(ab, wf, Ly.out_dim, X, hmax, L);
        }
        PINN_STAMP(ST_BWD_STREAM);
      }

      // ---- encoding backward (first Linear of feedforward / SIREN) ----
      if (net.enc == ENC_LINEAR && net.d_encW) {
        const int H = net.enc_out;
        __syncthreads();
        if (ft * 32 < H) {
          ew_enc_backward<ACT, NT, NX>(ab, net, xin, ft, L);
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[(s * hmax + ft * 32 + acc_row(r, L.lh)) * kTP + L.ln] = ab[s][r];
        }
        __syncthreads();
        if (tid < H) {
          float gb = 0.0f, gt = 0.0f, gx = 0.0f;
          float gw[kMaxDin] = {0.0f, 0.0f, 0.0f, 0.0f};
          for (int n = 0; n < kT; ++n) {
            const float vv = X[tid * kTP + n];
            gb += vv;
#pragma unroll
            for (int cc = 0; cc < kMaxDin; ++cc)
              if (cc < din) gw[cc] = fmaf(vv, xin[cc * kT + n], gw[cc]);
            if constexpr (NT >= 1) gt += X[(1 * hmax + tid) * kTP + n];
            if constexpr (NX >= 1) gx += X[((1 + NT) * hmax + tid) * kTP + n];
          }
#pragma unroll
          for (int cc = 0; cc < kMaxDin; ++cc)
            if (cc < din)
              atomicAdd(net.d_encW + tid * din + cc, gw[cc] + (cc == din - 1 ? gt : 0.0f) + (cc == 0 ? gx : 0.0f));
          if (net.d_encb) atomicAdd(net.d_encb + tid, gb);
        }
      }
      PINN_STAMP(ST_ENC_BWD);
    }
  }

  if constexpr (BWD) {  // one flush per workgroup
    if (net.dw_out && tid < net.h_last) atomicAdd(net.dw_out + tid, pw_out);
#pragma unroll
    for (int p = 0; p < kPersist; ++p) {
      if (p < net.n_layers) flush_rows(pacc[p], uniform_layer(net.layer[p]), L);
    }
    PINN_STAMP(ST_BWD_FLUSH);
  }
#ifdef PINN_STAMPS
  if (a.stamps && (tid & 63) == 0) {
    st_acc[ST_TOTAL] = pinn_now() - st_begin;
    for (int i = 0; i < kNumStamps; ++i) a.stamps[((long long)blockIdx.x * kWaves + L.wave) * kNumStamps + i] = st_acc[i];
  }
#endif
}

inline size_t jet_wide_lds_bytes(int K, int hmax, bool bwd) {
  return sizeof(float) * ((size_t)(bwd ? 2 : 1) * K * hmax * kTP + 2 * K * kT + kMaxDin * kT + hmax);
}

// true if the wide kernel can run this problem (width <= 128, all K streams fit in LDS)
inline bool jet_wide_fits(int K, int hmax, bool bwd) { return hmax <= 128 && jet_wide_lds_bytes(K, hmax, bwd) <= 160 * 1024; }

template <int NT, int NX>
hipError_t launch_jet_wide(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  constexpr int K = 1 + NT + NX;
  const size_t lds = jet_wide_lds_bytes(K, a.net.hmax, bwd);
  const int act = a.net.n_layers > 0 ? a.net.layer[0].act : a.net.enc_act;
  hipError_t e = hipSuccess;
#define PINN_WLAUNCH1(ACT_, BWD_)                                                                            \
  do {                                                                                                       \
    auto kern = jet_kernel_wide<ACT_, NT, NX, BWD_>;                                                         \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                       \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, a);                                    \
  } while (0)
#ifdef PINN_DEV
#define PINN_WLAUNCH(BWD_) \
  if (act == PINN_ACT_TANH) PINN_WLAUNCH1(PINN_ACT_TANH, BWD_); else return hipErrorInvalidValue;
#else
#define PINN_WLAUNCH(BWD_)                                                  \
  switch (act) {                                                            \
    case PINN_ACT_TANH: PINN_WLAUNCH1(PINN_ACT_TANH, BWD_); break;          \
    case PINN_ACT_SIN: PINN_WLAUNCH1(PINN_ACT_SIN, BWD_); break;            \
    case PINN_ACT_GELU: PINN_WLAUNCH1(PINN_ACT_GELU, BWD_); break;          \
    case PINN_ACT_SIGMOID: PINN_WLAUNCH1(PINN_ACT_SIGMOID, BWD_); break;    \
    default: PINN_WLAUNCH1(PINN_ACT_RELU, BWD_); break;                     \
  }
#endif
  if (bwd) { PINN_WLAUNCH(true) } else { PINN_WLAUNCH(false) }
#undef PINN_WLAUNCH
#undef PINN_WLAUNCH1
  return hipGetLastError();
}

}  // namespace pinn
