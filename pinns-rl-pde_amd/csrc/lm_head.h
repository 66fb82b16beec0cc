// Output layer + PDE epilogue of the layer-major engine (see lm_common.h).  gfx950.
//
// u_s = w_out . V_s (+ b_out on the value stream) for the K streams of a tile, then — per point — either the jets are
// written out (MODE_JETS) or the PDE residual, its loss term and the cotangents ubar_s = l'(r) dr/du_s (MODE_PDE,
// jet_device.h::pde_residual / loss_term).  With the reverse sweep the kernel also leaves the seed of the reverse
// chain: the (K x 32) cotangent block U of every tile (lm_ew_bwd forms Vbar = w_out (x) ubar from it on the fly) and
// the gradients of w_out / b_out.  Same thread map as lm_ew.h.
#pragma once
#include "lm_ew.h"

namespace pinn {
namespace lm {

struct HeadArgs {
  int H, Hp, G;
  long long ntiles, N, p_base;
  const float* V;       // head prologue output record
  const float* w_out;   // packed [Hp]
  const float* b_out;   // [1]
  PdeDev pde;
  int mode;             // MODE_JETS | MODE_PDE
  int bwd;
  float grad_scale;
  const float* x;       // (N, din - 1): first spatial coordinate feeds the Black-Scholes coefficients
  int din;
  float* jets_out[PINN_MAX_STREAMS];
  const float* jets_bar[PINN_MAX_STREAMS];
  float* residual_out;
  float* loss_sum;
  const float* res_bar;
  float* U;             // [tile][K][32]
  float* dw_out;        // packed [Hp]
  float* db_out;        // packed [1]
  float* det_partial;   // deterministic mode: per-workgroup partials [grid][1028] = dw_out (1024) | loss | db_out | dcoef (2)
};

template <int NT, int NX, int FPT>
__global__ __launch_bounds__(1024) void lm_head(const HeadArgs a) {
  constexpr int K = 1 + NT + NX;
  __shared__ float red[2 * kMaxWavesEw * kRedQ * kPT];
  __shared__ float pacc[1024];
  const int tid = threadIdx.x, n = tid & (kPT - 1), g = tid >> 4;
  const int wave = tid >> 6, nwaves = (a.G + 3) >> 2;
  const int nthreads = kPT * a.G;
  const unsigned voff = static_cast<unsigned>(g * kT + n) * 4u, goff = static_cast<unsigned>(g) * 4u;
  int slot = 0;
  for (int i = tid; i < 1024; i += nthreads) pacc[i] = 0.0f;
  __syncthreads();
  float w[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) w[i] = vec_ld(a.w_out, a.G * i, goff);  // zero on padding features
  const float b0 = a.b_out[0];
  float ploss = 0.0f, pdb = 0.0f, pdc0 = 0.0f, pdc1 = 0.0f;
  for (long long uu = 2LL * blockIdx.x; uu < 2 * a.ntiles; uu += (uu & 1) ? 2LL * gridDim.x - 1 : 1) {
    // both 16-point halves of a tile back to back in the SAME workgroup: a 128-byte record row is then fetched from HBM
    // once (the second half hits this CU's caches) and its two 64-byte stores merge; with the halves on neighbouring
    // workgroups (different XCDs under round-robin dispatch) every line crossed the fabric twice
    const long long unit = uu;
    const long long p = a.p_base + unit * kPT + n;
    const bool ok = p < a.N;
    const long long rec_off = (unit >> 1) * (long long)K * a.Hp * kT + (unit & 1) * kPT;
    float v[FPT][K];
    {
      const float* base = a.V + rec_off;
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) v[i][s] = rec_ld(base, s * a.Hp + a.G * i, voff);
    }
    float j[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < FPT; ++i) q = fmaf(w[i], v[i][s], q);
      j[s] = q;
    }
    block_sum<K>(j, red, slot, nwaves, wave, tid, n);
    j[0] += b0;
    float ub[K];
    if (a.mode == MODE_JETS) {
#pragma unroll
      for (int s = 0; s < K; ++s) {
        if (g == 0 && ok && a.jets_out[s]) a.jets_out[s][p] = j[s];
        ub[s] = (a.bwd && ok && a.jets_bar[s]) ? a.jets_bar[s][p] : 0.0f;
      }
    } else {
      const float x0 = (ok && a.din > 1) ? a.x[p * (a.din - 1)] : 0.0f;
      float d[K];
      const float r = pde_residual<NT, NX>(a.pde, j, x0, d);
      float dl;
      float lt = loss_term(a.pde, r, &dl);
      if (!ok) {
        lt = 0.0f;
        dl = 0.0f;
      }
      if (g == 0) {
        if (ok && a.residual_out) a.residual_out[p] = r;
        ploss += lt;
      }
      const float rb = !a.bwd ? 0.0f : (a.res_bar ? (ok ? a.res_bar[p] : 0.0f) : a.grad_scale * dl);
#pragma unroll
      for (int s = 0; s < K; ++s) ub[s] = rb * d[s];
      if (a.bwd && a.pde.dcoef && g == 0) {  // inverse problems: rbar dr/dc_k
        float dc0, dc1;
        pde_coef_grads<NT, NX>(a.pde, j, x0, dc0, dc1);
        pdc0 += rb * dc0;
        pdc1 += rb * dc1;
      }
    }
    if (a.bwd) {
      if (g == 0) {
#pragma unroll
        for (int s = 0; s < K; ++s) a.U[((unit >> 1) * K + s) * kT + (unit & 1) * kPT + n] = ub[s];
        pdb += ub[0];
      }
      if (a.dw_out) {
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
          float q = 0.0f;
#pragma unroll
          for (int s = 0; s < K; ++s) q = fmaf(ub[s], v[i][s], q);
          q = pt_sum(q);
          if (n == 0) pacc[g + a.G * i] += q;
        }
      }
    }
  }
  __syncthreads();
  if (a.det_partial) {
    float* P = a.det_partial + (long long)blockIdx.x * 1028;
    for (int i = tid; i < 1024; i += nthreads) P[i] = pacc[i];
    if (g == 0) {
      const float ls = pt_sum(ploss), ds = pt_sum(pdb), c0s = pt_sum(pdc0), c1s = pt_sum(pdc1);
      if (n == 0) {
        P[1024] = ls;
        P[1025] = ds;
        P[1026] = c0s;
        P[1027] = c1s;
      }
    }
    return;
  }
  if (a.bwd && a.dw_out)
    for (int f = tid; f < a.H; f += nthreads) atomicAdd(a.dw_out + f, pacc[f]);
  if (g == 0) {  // the 16 lanes of group 0 hold the per-point partials
    const float ls = pt_sum(ploss), ds = pt_sum(pdb);
    if (n == 0) {
      if (a.mode == MODE_PDE && a.loss_sum) atomicAdd(a.loss_sum, ls);
      if (a.bwd && a.db_out) atomicAdd(a.db_out, ds);
    }
    if (a.bwd && a.pde.dcoef && a.mode == MODE_PDE) {
      const float c0s = pt_sum(pdc0), c1s = pt_sum(pdc1);
      if (n == 0) {
        atomicAdd(a.pde.dcoef, c0s);
        atomicAdd(a.pde.dcoef + 1, c1s);
      }
    }
  }
}

}  // namespace lm
}  // namespace pinn
