// Fused jet kernel for the "attention" architecture (pinnrl/neural_networks/attention.py:11-183), gfx950.
//
// The reference applies multi-head self-attention to a sequence of length ONE (attention.py:50-52), so the
// softmax is identically 1 and the block reduces to  h <- LN_a(W_p (W_v h + b_v) + b_p + h); the query/key
// projections are mathematically dead (zero gradient).  Then the feed-forward block
// h <- LN_f(h + W_2 gelu(W_1 h + b_1) + b_2) with a 4x expansion.  The expansion (4H = 512 at H = 128) is
// processed in four H-wide chunks — z_c = W_1[c] h, g_c = gelu(z_c), out += W_2[:, c] g_c — so every GEMM
// stays an H x H problem for the stream-serial machinery and the LDS footprint stays that of width H.
//
// layer table convention for PINN_ARCH_ATTENTION (4 entries per attention layer):
//   layer[4l+0] = value (H -> H)          layer[4l+1] = proj (H -> H) + LN_a
//   layer[4l+2] = net.0 (H -> 4H)         layer[4l+3] = net.3 (4H -> H) + LN_f
// LayerNorm jets, reductions and limits (derivative orders <= 2) as in jet_kernel_resnet.h.
#pragma once
#include "jet_kernel_resnet.h"

namespace pinn {

constexpr int kAttnSlots = 9;  // tape records per attention layer: h, v, za, h1, z1[0..3], zf

// an H x H view of rows [c*H, (c+1)*H) of W (chunk of net.0) or of columns [c*H, (c+1)*H) of W (chunk of net.3)
__device__ __forceinline__ LayerDev chunk_rows(const LayerDev& F, int c, int H) {
  LayerDev v = F;
  v.W = F.W + (long long)c * H * F.ld;
  v.b = F.b + c * H;
  v.dW = F.dW ? F.dW + (long long)c * H * F.ld : nullptr;
  v.db = F.db ? F.db + c * H : nullptr;
  v.out_dim = H;
  return v;
}
__device__ __forceinline__ LayerDev chunk_cols(const LayerDev& F, int c, int H, bool with_bias) {
  LayerDev v = F;
  v.W = F.W + c * H;
  v.dW = F.dW ? F.dW + c * H : nullptr;
  v.in_dim = H;
  if (!with_bias) {
    v.b = nullptr;
    v.db = nullptr;
  }
  return v;
}

template <int K, int NTILE>
__device__ __forceinline__ void tile_zero(f32x16 (&v)[NTILE][K]) {
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[jt][s][r] = 0.0f;
}

template <int K, int NTILE>
__device__ __forceinline__ void tile_put(const f32x16 (&v)[NTILE][K], float* tape, int slot, int H, int tid, const Lane& L) {
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt)
    if ((L.wave + kWaves * jt) * 32 < H) tape_put<K, NTILE>(v[jt], tape, slot, jt, tid);
}

template <int K, int NTILE>
__device__ __forceinline__ void tile_get(f32x16 (&v)[NTILE][K], const float* tape, int slot, int H, int tid, const Lane& L) {
  tile_zero<K, NTILE>(v);
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt)
    if ((L.wave + kWaves * jt) * 32 < H) tape_get<K, NTILE>(v[jt], tape, slot, jt, tid);
}

template <int ACT, int NT, int NX, int NTILE, bool BWD>
__global__ __launch_bounds__(kThreads, 1) void jet_kernel_attn(const KernelArgs a) {
  constexpr int K = 1 + NT + NX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDev& net = a.net;
  const int hmax = net.hmax;
  const int sbuf = hmax * kTP;
  float* SB = smem;
  float* RED = SB + (BWD ? 4 : 2) * sbuf;
  float* U = RED + kWaves * K * kT;
  float* UB = U + K * kT;
  float* xin = UB + K * kT;
  float* red0 = xin + kMaxDin * kT;
  float* red1 = red0 + kWaves * kMaxMom * kT;

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int tid = L.tid;
  const int din = net.din;
  const int H = net.enc_out;
  const int nl = net.n_layers >> 2;
  const float eps = net.ln_eps;
  const long long ntiles = (a.N + kT - 1) / kT;
  float* tape = BWD ? a.tape + (long long)blockIdx.x * a.tape_stride : nullptr;
  int c = 0;
  PINN_STAMP_DECL

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
    PINN_STAMP(ST_ENCODE);

    for (int l = 0; l < nl; ++l) {
      const LayerDev LV = uniform_layer(net.layer[4 * l + 0]);
      const LayerDev LP = uniform_layer(net.layer[4 * l + 1]);
      const LayerDev F1 = uniform_layer(net.layer[4 * l + 2]);
      const LayerDev F2 = uniform_layer(net.layer[4 * l + 3]);
      const int slot = kAttnSlots * l;
      if constexpr (BWD) tile_put<K, NTILE>(v, tape, slot + 0, H, tid, L);  // h
      f32x16 y[NTILE][K], y2[NTILE][K];
      linear_forward<K, NTILE, false>(y, v, LV, SB, sbuf, c, L);             // v = W_v h + b_v
      if constexpr (BWD) tile_put<K, NTILE>(y, tape, slot + 1, H, tid, L);
      linear_forward<K, NTILE, false>(y2, y, LP, SB, sbuf, c, L);            // p = W_p v + b_p
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
        for (int s = 0; s < K; ++s) y2[jt][s] += v[jt][s];                   // za = p + h
      if constexpr (BWD) tile_put<K, NTILE>(y2, tape, slot + 2, H, tid, L);
      PINN_STAMP(ST_FWD_GEMM);  // value + projection Linears (and their tape stores)
      LnStats<NT, NX> Sa;
      ln_stats<NT, NX, NTILE>(y2, H, eps, red0, red1, Sa, L);
      ln_apply<NT, NX, NTILE>(y2, H, Sa, LP.ln_g, LP.ln_b, L);               // h1
      if constexpr (BWD) tile_put<K, NTILE>(y2, tape, slot + 3, H, tid, L);
      PINN_STAMP(ST_FWD_EW);  // LayerNorm
      // feed-forward block, expansion processed in H-wide chunks; v <- zf = h1 + b_2 + sum_c W_2[:, c] gelu(W_1[c] h1 + b_1[c])
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
        for (int s = 0; s < K; ++s) v[jt][s] = y2[jt][s];
      const int nchunk = F1.out_dim / H;
      for (int ch = 0; ch < nchunk; ++ch) {
        const LayerDev C1 = chunk_rows(F1, ch, H);
        const LayerDev C2 = chunk_cols(F2, ch, H, ch == 0);
        linear_forward<K, NTILE, false>(y, y2, C1, SB, sbuf, c, L);          // z1_c
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
          if ((L.wave + kWaves * jt) * 32 < H)
            ew_forward<PINN_ACT_GELU, NT, NX, NTILE, BWD>(y[jt], 0.0f, tape, slot + 4 + ch, jt, tid);  // g_c (tape: z1_c)
        linear_forward<K, NTILE, true>(v, y, C2, SB, sbuf, c, L);            // zf += W_2[:, c] g_c (+ b_2 once)
      }
      if constexpr (BWD) tile_put<K, NTILE>(v, tape, slot + 8, H, tid, L);
      PINN_STAMP(ST_OUT);  // forward feed-forward chunks (8 Linear passes + GELU jets)
      LnStats<NT, NX> Sf;
      ln_stats<NT, NX, NTILE>(v, H, eps, red0, red1, Sf, L);
      ln_apply<NT, NX, NTILE>(v, H, Sf, F2.ln_g, F2.ln_b, L);                // h2
      PINN_STAMP(ST_FWD_EW);
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

    PINN_STAMP(ST_EPI);
    if constexpr (BWD) {
      __syncthreads();
      f32x16 ab[NTILE][K];
      {  // output layer reverse
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

      PINN_STAMP(ST_B0);
      for (int l = nl - 1; l >= 0; --l) {
        const LayerDev LV = uniform_layer(net.layer[4 * l + 0]);
        const LayerDev LP = uniform_layer(net.layer[4 * l + 1]);
        const LayerDev F1 = uniform_layer(net.layer[4 * l + 2]);
        const LayerDev F2 = uniform_layer(net.layer[4 * l + 3]);
        const int slot = kAttnSlots * l;
        f32x16 cz[NTILE][K], gsum[NTILE], bsum[NTILE], ap[NTILE][K], abn[NTILE][K], acc1[NTILE][K], zb[NTILE][K];
        // LN_f reverse: ab (= h2bar) -> zfbar
        tile_get<K, NTILE>(cz, tape, slot + 8, H, tid, L);
        LnStats<NT, NX> Sf;
        ln_stats<NT, NX, NTILE>(cz, H, eps, red0, red1, Sf, L);
        ln_backward<NT, NX, NTILE>(cz, ab, gsum, bsum, H, Sf, F2.ln_g, red0, red1, L);
        ln_param_grads<NTILE>(gsum, bsum, H, F2.d_ln_g, F2.d_ln_b, SB, sbuf, tid, L);
        // feed-forward chunks: h1bar = zfbar (skip) + sum_c W_1[c]^T gelu'(z1_c) * (W_2[:, c]^T zfbar)
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
          for (int s = 0; s < K; ++s) acc1[jt][s] = ab[jt][s];
        tile_get<K, NTILE>(cz, tape, slot + 3, H, tid, L);  // h1, the input of every W_1 chunk
        PINN_STAMP(ST_BWD_EW);  // LayerNorm reverse (+ parameter gradients)
        const int nchunk = F1.out_dim / H;
        for (int ch = 0; ch < nchunk; ++ch) {
          const LayerDev C1 = chunk_rows(F1, ch, H);
          const LayerDev C2 = chunk_cols(F2, ch, H, ch == 0);
          tile_zero<K, NTILE>(ap);
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt)
            if ((L.wave + kWaves * jt) * 32 < H) ew_replay<PINN_ACT_GELU, NT, NX, NTILE>(ap[jt], 0.0f, tape, slot + 4 + ch, jt, tid);
          linear_backward<K, NTILE>(abn, ab, ap, C2, SB, sbuf, c, tid, L);   // dW_2[:, c], (db_2), g_c bar
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
            for (int s = 0; s < K; ++s) zb[jt][s] = abn[jt][s];
            if ((L.wave + kWaves * jt) * 32 < H) ew_backward<PINN_ACT_GELU, NT, NX, NTILE>(zb[jt], 0.0f, tape, slot + 4 + ch, jt, tid);
          }
          linear_backward<K, NTILE>(abn, zb, cz, C1, SB, sbuf, c, tid, L);   // dW_1[c], db_1[c], contribution to h1bar
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
            for (int s = 0; s < K; ++s) acc1[jt][s] += abn[jt][s];
        }
        PINN_STAMP(ST_BWD_STREAM);  // reverse feed-forward chunks (8 linear_backward passes + GELU replay / adjoint)
        // LN_a reverse: acc1 (= h1bar) -> zabar
        tile_get<K, NTILE>(cz, tape, slot + 2, H, tid, L);
        LnStats<NT, NX> Sa;
        ln_stats<NT, NX, NTILE>(cz, H, eps, red0, red1, Sa, L);
        ln_backward<NT, NX, NTILE>(cz, acc1, gsum, bsum, H, Sa, LP.ln_g, red0, red1, L);
        ln_param_grads<NTILE>(gsum, bsum, H, LP.d_ln_g, LP.d_ln_b, SB, sbuf, tid, L);
        PINN_STAMP(ST_BWD_EW);
        // proj and value reverse; hbar = zabar (skip) + W_v^T W_p^T zabar
        tile_get<K, NTILE>(ap, tape, slot + 1, H, tid, L);  // v
        linear_backward<K, NTILE>(abn, acc1, ap, LP, SB, sbuf, c, tid, L);
        tile_get<K, NTILE>(ap, tape, slot + 0, H, tid, L);  // h
        linear_backward<K, NTILE>(zb, abn, ap, LV, SB, sbuf, c, tid, L);
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
          for (int s = 0; s < K; ++s) ab[jt][s] = zb[jt][s] + acc1[jt][s];
        PINN_STAMP(ST_BWD_DX);  // projection + value linear_backward
      }

      // ---- input projection reverse ----
      if (net.d_encW) {
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = L.wave + kWaves * jt;
          if (ft * 32 < H) ew_enc_backward<ACT, NT, NX>(ab[jt], net, xin, ft, L);
        }
        float* S0 = SB + 0 * sbuf;
        float* S1 = SB + 1 * sbuf;
        float* S2 = SB + 2 * sbuf;
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
#ifdef PINN_STAMPS
  if (a.stamps && (tid & 63) == 0) {
    st_acc[ST_TOTAL] = pinn_now() - st_begin;
    for (int i = 0; i < kNumStamps; ++i) a.stamps[((long long)blockIdx.x * kWaves + L.wave) * kNumStamps + i] = st_acc[i];
  }
#endif
}

inline long long jet_attn_tape_floats_per_wg(int K, int n_attn_layers, int ntile) {
  return (long long)kAttnSlots * n_attn_layers * ntile * K * 16 * kThreads;
}

template <int NT, int NX>
hipError_t launch_jet_attn(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  constexpr int K = 1 + NT + NX;
  if constexpr (NT > 2 || NX > 2) {
    return hipErrorNotSupported;
  } else {
    const int ntile = a.net.hmax > 128 ? 2 : 1;
    const size_t lds = jet_resnet_lds_bytes(K, a.net.hmax, bwd);
    const int act = a.net.enc_act;
    hipError_t e = hipSuccess;
#define PINN_ALAUNCH1(ACT_, NTILE_, BWD_)                                                                    \
  do {                                                                                                       \
    auto kern = jet_kernel_attn<ACT_, NT, NX, NTILE_, BWD_>;                                                 \
    e = allow_full_lds(reinterpret_cast<const void*>(kern));                                                 \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, a);                                    \
  } while (0)
#define PINN_ALAUNCH(ACT_)                                                                          \
  if (ntile == 1) { if (bwd) PINN_ALAUNCH1(ACT_, 1, true); else PINN_ALAUNCH1(ACT_, 1, false); }   \
  else { if (bwd) PINN_ALAUNCH1(ACT_, 2, true); else PINN_ALAUNCH1(ACT_, 2, false); }
    switch (act) {  // activation of the input projection (config default "gelu"); the feed-forward block is always GELU
      case PINN_ACT_GELU: PINN_ALAUNCH(PINN_ACT_GELU) break;
#ifndef PINN_DEV
      case PINN_ACT_TANH: PINN_ALAUNCH(PINN_ACT_TANH) break;
      case PINN_ACT_SIGMOID: PINN_ALAUNCH(PINN_ACT_SIGMOID) break;
      default: PINN_ALAUNCH(PINN_ACT_RELU) break;
#else
      default: return hipErrorInvalidValue;
#endif
    }
#undef PINN_ALAUNCH
#undef PINN_ALAUNCH1
    return hipGetLastError();
  }
}

}  // namespace pinn
