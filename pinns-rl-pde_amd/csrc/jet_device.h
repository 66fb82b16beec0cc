// Per-element jet arithmetic for the fused PINN kernels (gfx950).
//
// A "jet" is the tuple of K = 1 + NT + NX streams
//   [value, d/dt .. d^NT/dt^NT, d/dx .. d^NX/dx^NX]
// of one scalar quantity at one collocation point.  Linear layers act on every
// stream with the same weights (MFMA, see jet_kernel.h); activations act per
// element through Faa di Bruno, and the reverse sweep uses the hand-derived
// adjoints below.  tests/jet_model.py is the executable specification of this
// file (checked against the autograd oracle in fp64).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pinn_jet.h"

namespace pinn {

constexpr int kMaxOrd = 4;

// ---------------------------------------------------------------------------
// sin and cos together, ~1 ulp for |x| < 2^13: Cody-Waite reduction by pi/2 in three exact pieces, then the
// Cephes single-precision minimax polynomials on [-pi/4, pi/4].  The Fourier features evaluate 2*M of these per
// point and SIREN one per hidden unit; the library sincosf (Payne-Hanek capable, ~150 instructions) made the
// encoding the slowest phase of a tile.  Larger arguments take the library path.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void fast_sincosf(float x, float* sn, float* cs) {
  if (fabsf(x) > 8192.0f) {
    sincosf(x, sn, cs);
    return;
  }
  const float k = rintf(x * 0.63661977236758134308f);  // x * 2/pi
  float r = fmaf(k, -1.5703125f, x);
  r = fmaf(k, -4.837512969970703125e-4f, r);
  r = fmaf(k, -7.54978995489188216e-8f, r);
  const float z = r * r;
  float s = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  s = fmaf(s * z, r, r);
  float c = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  c = fmaf(c * z, z, fmaf(-0.5f, z, 1.0f));
  const int q = static_cast<int>(k) & 3;
  const float s1 = (q & 1) ? c : s;
  const float c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// ---------------------------------------------------------------------------
// tanh, branch-free, ~1-2 ulp: |z| < 0.625 -> the odd minimax polynomial the ROCm device library uses;
// otherwise 1 - 2 / (exp(2|z|) + 1) on v_exp_f32 / v_rcp_f32 (the argument error of the hardware exp2 is damped
// by 2e/(e+1)^2 <= 0.35 in this range).  The library tanhf diverges on the 0.625 threshold, so a wave pays
// for both of its branches (~45 instructions); this is ~20.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fast_tanhf(float z) {
  const float a = fabsf(z);
  const float t = z * z;
  float p = fmaf(-0.005700020585209131f, t, 0.02063407190144062f);
  p = fmaf(p, t, -0.053737930953502655f);
  p = fmaf(p, t, 0.13331416249275208f);
  p = fmaf(p, t, -0.3333328068256378f);
  const float small = fmaf(z * t, p, z);
  const float e = __builtin_amdgcn_exp2f(a * 2.8853900817779268f);  // exp(2a); +inf for large a -> big = 1
  const float big = copysignf(fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f), z);
  return a < 0.625f ? small : big;
}

// ---------------------------------------------------------------------------
// f[k] = k-th derivative of the activation at z, k = 0..ORD (ORD <= 5)
// ---------------------------------------------------------------------------
template <int ACT, int ORD>
__device__ __forceinline__ void act_derivs(float z, float w, float (&f)[6]) {
  if constexpr (ACT == PINN_ACT_TANH) {
    const float y = fast_tanhf(z);
    const float y2 = y * y;
    const float f1 = 1.0f - y2;
    f[0] = y;
    f[1] = f1;
    if constexpr (ORD >= 2) f[2] = -2.0f * y * f1;
    if constexpr (ORD >= 3) f[3] = f1 * (6.0f * y2 - 2.0f);
    if constexpr (ORD >= 4) f[4] = 8.0f * y * f1 * (2.0f - 3.0f * y2);
    if constexpr (ORD >= 5) f[5] = 8.0f * f1 * (2.0f - 15.0f * y2 + 15.0f * y2 * y2);
  } else if constexpr (ACT == PINN_ACT_SIN) {
    float s, c;
    fast_sincosf(w * z, &s, &c);
    const float w2 = w * w;
    f[0] = s;
    f[1] = w * c;
    if constexpr (ORD >= 2) f[2] = -w2 * s;
    if constexpr (ORD >= 3) f[3] = -w2 * w * c;
    if constexpr (ORD >= 4) f[4] = w2 * w2 * s;
    if constexpr (ORD >= 5) f[5] = w2 * w2 * w * c;
  } else if constexpr (ACT == PINN_ACT_GELU) {
    const float z2 = z * z;
    const float phi = expf(-0.5f * z2) * 0.3989422804014327f;
    const float Phi = 0.5f * (1.0f + erff(z * 0.7071067811865476f));
    f[0] = z * Phi;
    f[1] = Phi + z * phi;
    if constexpr (ORD >= 2) f[2] = phi * (2.0f - z2);
    if constexpr (ORD >= 3) f[3] = phi * z * (z2 - 4.0f);
    if constexpr (ORD >= 4) f[4] = phi * (-z2 * z2 + 7.0f * z2 - 4.0f);
    if constexpr (ORD >= 5) f[5] = phi * z * (z2 * z2 - 11.0f * z2 + 18.0f);
  } else if constexpr (ACT == PINN_ACT_SIGMOID) {
    const float s = 1.0f / (1.0f + expf(-z));
    const float f1 = s * (1.0f - s);
    f[0] = s;
    f[1] = f1;
    if constexpr (ORD >= 2) f[2] = f1 * (1.0f - 2.0f * s);
    if constexpr (ORD >= 3) f[3] = f1 * (1.0f - 6.0f * f1);
    if constexpr (ORD >= 4) f[4] = f[2] * (1.0f - 12.0f * f1);
    if constexpr (ORD >= 5) f[5] = f[3] * (1.0f - 12.0f * f1) - 12.0f * f[2] * f[2];
  } else {  // piecewise linear: relu (w = 0), leaky_relu (w = 0.01), identity (w = 1)
    const float m = z > 0.0f ? 1.0f : w;
    f[0] = z * m;
    f[1] = m;
    f[2] = f[3] = f[4] = f[5] = 0.0f;
  }
}

// one direction, forward: z[0..M) = derivative streams of the pre-activation, y = of the activation
template <int M>
__device__ __forceinline__ void dir_fwd(const float (&f)[6], const float* z, float* y) {
  if constexpr (M >= 1) y[0] = f[1] * z[0];
  if constexpr (M >= 2) y[1] = f[2] * z[0] * z[0] + f[1] * z[1];
  if constexpr (M >= 3) y[2] = f[3] * z[0] * z[0] * z[0] + 3.0f * f[2] * z[0] * z[1] + f[1] * z[2];
  if constexpr (M >= 4)
    y[3] = f[4] * z[0] * z[0] * z[0] * z[0] + 6.0f * f[3] * z[0] * z[0] * z[1] + 3.0f * f[2] * z[1] * z[1] +
           4.0f * f[2] * z[0] * z[2] + f[1] * z[3];
}

// one direction, adjoint: returns the contribution to zbar_value, writes zb[0..M)
template <int M>
__device__ __forceinline__ float dir_bwd(const float (&f)[6], const float* z, const float* ab, float* zb) {
  float z0b = 0.0f;
  if constexpr (M >= 1) {
    z0b += f[2] * z[0] * ab[0];
    zb[0] = f[1] * ab[0];
  }
  if constexpr (M >= 2) {
    z0b += (f[3] * z[0] * z[0] + f[2] * z[1]) * ab[1];
    zb[0] += 2.0f * f[2] * z[0] * ab[1];
    zb[1] = f[1] * ab[1];
  }
  if constexpr (M >= 3) {
    z0b += (f[4] * z[0] * z[0] * z[0] + 3.0f * f[3] * z[0] * z[1] + f[2] * z[2]) * ab[2];
    zb[0] += (3.0f * f[3] * z[0] * z[0] + 3.0f * f[2] * z[1]) * ab[2];
    zb[1] += 3.0f * f[2] * z[0] * ab[2];
    zb[2] = f[1] * ab[2];
  }
  if constexpr (M >= 4) {
    z0b += (f[5] * z[0] * z[0] * z[0] * z[0] + 6.0f * f[4] * z[0] * z[0] * z[1] + 3.0f * f[3] * z[1] * z[1] +
            4.0f * f[3] * z[0] * z[2] + f[2] * z[3]) *
           ab[3];
    zb[0] += (4.0f * f[4] * z[0] * z[0] * z[0] + 12.0f * f[3] * z[0] * z[1] + 4.0f * f[2] * z[2]) * ab[3];
    zb[1] += (6.0f * f[3] * z[0] * z[0] + 6.0f * f[2] * z[1]) * ab[3];
    zb[2] += 4.0f * f[2] * z[0] * ab[3];
    zb[3] = f[1] * ab[3];
  }
  return z0b;
}

template <int A, int B>
struct MaxOf {
  static constexpr int v = A > B ? A : B;
};

// a[0..K) = jets of act(z) given z[0..K)
template <int ACT, int NT, int NX>
__device__ __forceinline__ void act_fwd(float w, const float* z, float* a) {
  float f[6];
  act_derivs<ACT, MaxOf<MaxOf<NT, NX>::v, 1>::v>(z[0], w, f);
  a[0] = f[0];
  dir_fwd<NT>(f, z + 1, a + 1);
  dir_fwd<NX>(f, z + 1 + NT, a + 1 + NT);
}

// zb[0..K) = adjoint of act_fwd given the cotangent ab[0..K) of the activation's jets
template <int ACT, int NT, int NX>
__device__ __forceinline__ void act_bwd(float w, const float* z, const float* ab, float* zb) {
  float f[6];
  act_derivs<ACT, MaxOf<NT, NX>::v + 1>(z[0], w, f);
  float z0b = f[1] * ab[0];
  z0b += dir_bwd<NT>(f, z + 1, ab + 1, zb + 1);
  z0b += dir_bwd<NX>(f, z + 1 + NT, ab + 1 + NT, zb + 1 + NT);
  zb[0] = z0b;
}

// ---------------------------------------------------------------------------
// Tape form.  The reverse sweep needs f'(z), f''(z), ... at every hidden pre-activation.  For tanh
// and sigmoid these are polynomials of the activation VALUE, so the tape keeps y = f(z) in slot 0 and
// the reverse sweep evaluates no transcendental; for the others slot 0 keeps z.
// ---------------------------------------------------------------------------
template <int ACT>
struct ActTape {
  static constexpr bool value_is_output = (ACT == PINN_ACT_TANH || ACT == PINN_ACT_SIGMOID);
};

template <int ACT, int ORD>
__device__ __forceinline__ void act_derivs_tape(float t0, float w, float (&f)[6]) {
  if constexpr (ACT == PINN_ACT_TANH) {
    const float y = t0, y2 = y * y, f1 = 1.0f - y2;
    f[0] = y;
    f[1] = f1;
    if constexpr (ORD >= 2) f[2] = -2.0f * y * f1;
    if constexpr (ORD >= 3) f[3] = f1 * (6.0f * y2 - 2.0f);
    if constexpr (ORD >= 4) f[4] = 8.0f * y * f1 * (2.0f - 3.0f * y2);
    if constexpr (ORD >= 5) f[5] = 8.0f * f1 * (2.0f - 15.0f * y2 + 15.0f * y2 * y2);
  } else if constexpr (ACT == PINN_ACT_SIGMOID) {
    const float s = t0, f1 = s * (1.0f - s);
    f[0] = s;
    f[1] = f1;
    if constexpr (ORD >= 2) f[2] = f1 * (1.0f - 2.0f * s);
    if constexpr (ORD >= 3) f[3] = f1 * (1.0f - 6.0f * f1);
    if constexpr (ORD >= 4) f[4] = f[2] * (1.0f - 12.0f * f1);
    if constexpr (ORD >= 5) f[5] = f[3] * (1.0f - 12.0f * f1) - 12.0f * f[2] * f[2];
  } else {
    act_derivs<ACT, ORD>(t0, w, f);
  }
}

// activation jets replayed from a tape record [t0, z_1 .. z_{K-1}]
template <int ACT, int NT, int NX>
__device__ __forceinline__ void act_fwd_tape(float w, const float* z, float* a) {
  float f[6];
  act_derivs_tape<ACT, MaxOf<MaxOf<NT, NX>::v, 1>::v>(z[0], w, f);
  a[0] = f[0];
  dir_fwd<NT>(f, z + 1, a + 1);
  dir_fwd<NX>(f, z + 1 + NT, a + 1 + NT);
}

// adjoint of the activation jets from a tape record
template <int ACT, int NT, int NX>
__device__ __forceinline__ void act_bwd_tape(float w, const float* z, const float* ab, float* zb) {
  float f[6];
  act_derivs_tape<ACT, MaxOf<NT, NX>::v + 1>(z[0], w, f);
  float z0b = f[1] * ab[0];
  z0b += dir_bwd<NT>(f, z + 1, ab + 1, zb + 1);
  z0b += dir_bwd<NX>(f, z + 1 + NT, ab + 1 + NT, zb + 1 + NT);
  zb[0] = z0b;
}

// Runtime activation id -> compile-time tag (the id is uniform across the workgroup).
// The body sees `ACT` as a constant expression; variadic so template commas survive.
#define PINN_ACT_SWITCH(act_id, ...)                                           \
  switch (act_id) {                                                            \
    case PINN_ACT_TANH: { constexpr int ACT = PINN_ACT_TANH; __VA_ARGS__ } break;       \
    case PINN_ACT_SIN: { constexpr int ACT = PINN_ACT_SIN; __VA_ARGS__ } break;         \
    case PINN_ACT_GELU: { constexpr int ACT = PINN_ACT_GELU; __VA_ARGS__ } break;       \
    case PINN_ACT_SIGMOID: { constexpr int ACT = PINN_ACT_SIGMOID; __VA_ARGS__ } break; \
    default: { constexpr int ACT = PINN_ACT_RELU; __VA_ARGS__ } break;                  \
  }

// ---------------------------------------------------------------------------
// PDE epilogue: residual r(jets) and dr/d(jet_s); tests/jet_model.py::pde_residual
// ---------------------------------------------------------------------------
struct PdeDev {
  int kind;
  int dimension;
  int loss;
  float c0, c1, c2, c3;
  float huber_delta;
  float* dcoef;  // reverse sweeps, nullable: dcoef[k] += sum_n rbar_n dr_n/dc_k (inverse problems: trainable coefficients)
};

template <int NT, int NX>
__device__ __forceinline__ float pde_residual(const PdeDev& p, const float* j, float x0, float* d) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int s = 0; s < K; ++s) d[s] = 0.0f;
  const float u = j[0];
  // helpers: T(k) = j[k], X(k) = j[NT + k]
  if (p.dimension > 1) {
    // >= 2-D: every spatial term of the reference differentiates w.r.t. a fresh slice and vanishes
    if constexpr (NT >= 2) {
      if (p.kind == PINN_PDE_WAVE) { d[2] = 1.0f; return j[2]; }
      if (p.kind == PINN_PDE_PENDULUM) { d[0] = p.c0 * cosf(u); d[2] = 1.0f; return j[2] + p.c0 * sinf(u); }
    }
    if constexpr (NT >= 1) {
      if (p.kind == PINN_PDE_ALLEN_CAHN) { d[0] = 3.0f * u * u - 1.0f; d[1] = 1.0f; return j[1] - u + u * u * u; }
      if (p.kind == PINN_PDE_BLACK_SCHOLES) { d[0] = -p.c1; d[1] = 1.0f; return j[1] - p.c1 * u; }  // black_scholes.py:84-91: -rV survives
      d[1] = 1.0f;
      return j[1];
    }
    return 0.0f;
  }
  switch (p.kind) {
    case PINN_PDE_BURGERS:
      if constexpr (NT >= 1 && NX >= 2) {
        d[0] = j[NT + 1]; d[1] = 1.0f; d[NT + 1] = u; d[NT + 2] = -p.c0;
        return j[1] + u * j[NT + 1] - p.c0 * j[NT + 2];
      }
      break;
    case PINN_PDE_HEAT:
      if constexpr (NT >= 1 && NX >= 1) {
        d[1] = 1.0f; d[NT + 1] = -p.c0;
        return j[1] - p.c0 * j[NT + 1];
      }
      break;
    case PINN_PDE_HEAT_LAPLACIAN:
      if constexpr (NT >= 1 && NX >= 2) {
        d[1] = 1.0f; d[NT + 2] = -p.c0;
        return j[1] - p.c0 * j[NT + 2];
      }
      break;
    case PINN_PDE_ALLEN_CAHN:
      if constexpr (NT >= 1 && NX >= 2) {
        const float e2 = p.c0 * p.c0;
        d[0] = 3.0f * u * u - 1.0f; d[1] = 1.0f; d[NT + 2] = -e2;
        return j[1] - e2 * j[NT + 2] - u + u * u * u;
      }
      break;
    case PINN_PDE_KDV:
      if constexpr (NT >= 1 && NX >= 3) {
        d[0] = 6.0f * j[NT + 1]; d[1] = 1.0f; d[NT + 1] = 6.0f * u; d[NT + 3] = 1.0f;
        return j[1] + 6.0f * u * j[NT + 1] + j[NT + 3];
      }
      break;
    case PINN_PDE_CAHN_HILLIARD:
      if constexpr (NT >= 1 && NX >= 4) {
        const float e2 = p.c0 * p.c0;
        const float m = (u >= -10.0f && u <= 10.0f) ? 1.0f : 0.0f;
        const float c = fminf(fmaxf(u, -10.0f), 10.0f);
        const float ux = j[NT + 1], uxx = j[NT + 2];
        d[0] = -m * (6.0f * ux * ux + 6.0f * c * uxx);
        d[1] = 1.0f;
        d[NT + 1] = -m * 12.0f * c * ux;
        d[NT + 2] = -m * (3.0f * c * c - 1.0f);
        d[NT + 4] = e2;
        return j[1] + e2 * j[NT + 4] - m * (6.0f * c * ux * ux + (3.0f * c * c - 1.0f) * uxx);
      }
      break;
    case PINN_PDE_WAVE:
      if constexpr (NT >= 2 && NX >= 2) {
        const float c2 = p.c0 * p.c0;
        d[2] = 1.0f; d[NT + 2] = -c2;
        return j[2] - c2 * j[NT + 2];
      }
      break;
    case PINN_PDE_CONVECTION:
      if constexpr (NT >= 1 && NX >= 1) {
        d[1] = 1.0f; d[NT + 1] = p.c0;
        return j[1] + p.c0 * j[NT + 1];
      }
      break;
    case PINN_PDE_BLACK_SCHOLES:
      if constexpr (NT >= 1 && NX >= 2) {
        const float hs = 0.5f * p.c0 * p.c0 * x0 * x0;
        d[0] = -p.c1; d[1] = 1.0f; d[NT + 1] = p.c1 * x0; d[NT + 2] = hs;
        return j[1] + hs * j[NT + 2] + p.c1 * x0 * j[NT + 1] - p.c1 * u;
      }
      break;
    case PINN_PDE_PENDULUM:
      if constexpr (NT >= 2) {
        d[0] = p.c0 * cosf(u); d[2] = 1.0f;
        return j[2] + p.c0 * sinf(u);
      }
      break;
    default:
      break;
  }
  return 0.0f;
}

// dr/dc_0 and dr/dc_1 of the residual above (pde_base.py:246-279: a trainable coefficient is a live nn.Parameter inside the
// residual; the reference gets these by autograd).  Coefficients c_2, c_3 are unused by every PDE of pinnrl/pdes/.
template <int NT, int NX>
__device__ __forceinline__ void pde_coef_grads(const PdeDev& p, const float* j, float x0, float& dc0, float& dc1) {
  dc0 = dc1 = 0.0f;
  const float u = j[0];
  // the streams the formulas read (absent streams of a smaller compiled set: the PDE kind cannot occur with it)
  const float ux = NX >= 1 ? j[NT + (NX >= 1 ? 1 : 0)] : 0.0f;
  const float uxx = NX >= 2 ? j[NT + (NX >= 2 ? 2 : 0)] : 0.0f;
  const float uxxxx = NX >= 4 ? j[NT + (NX >= 4 ? 4 : 0)] : 0.0f;
  const int k = p.kind;
  if (p.dimension > 1) {  // only terms that survive the reference's >= 2-D behaviour carry a coefficient
    dc0 = k == PINN_PDE_PENDULUM ? sinf(u) : 0.0f;
    dc1 = k == PINN_PDE_BLACK_SCHOLES ? -u : 0.0f;
    return;
  }
  if (k == PINN_PDE_BURGERS || k == PINN_PDE_HEAT_LAPLACIAN) dc0 = -uxx;
  else if (k == PINN_PDE_HEAT) dc0 = -ux;
  else if (k == PINN_PDE_ALLEN_CAHN || k == PINN_PDE_WAVE) dc0 = -2.0f * p.c0 * uxx;
  else if (k == PINN_PDE_CAHN_HILLIARD) dc0 = 2.0f * p.c0 * uxxxx;
  else if (k == PINN_PDE_CONVECTION) dc0 = ux;
  else if (k == PINN_PDE_PENDULUM) dc0 = sinf(u);
  else if (k == PINN_PDE_BLACK_SCHOLES) {
    dc0 = p.c0 * x0 * x0 * uxx;
    dc1 = x0 * ux - u;
  }
}

// l(r) and l'(r) for PDEBase._apply_loss_fn (pinnrl/pdes/pde_base.py:309-326), per sample, before the mean
__device__ __forceinline__ float loss_term(const PdeDev& p, float r, float* dl) {
  if (p.loss == PINN_LOSS_MAE) {
    *dl = r > 0.0f ? 1.0f : (r < 0.0f ? -1.0f : 0.0f);
    return fabsf(r);
  }
  if (p.loss == PINN_LOSS_HUBER) {
    const float a = fabsf(r), dlt = p.huber_delta;
    if (a < dlt) { *dl = r; return 0.5f * r * r; }  // torch: |r| < delta uses the quadratic branch
    *dl = r > 0.0f ? dlt : -dlt;
    return dlt * (a - 0.5f * dlt);
  }
  *dl = 2.0f * r;
  return r * r;
}

}  // namespace pinn
