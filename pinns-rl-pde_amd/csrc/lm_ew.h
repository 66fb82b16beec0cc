// Element-wise ("prologue") kernels of the layer-major engine (see lm_common.h).  gfx950.
//
// A prologue maps stored records to the input jets V of the next GEMM:
//
//     z  = source            record | first Linear evaluated from the coordinates | Fourier features of the coordinates
//     p  = LayerNorm(z)      optional; jets to 4th order (tests/jet_model.py::ln_fwd_gen / ln_bwd_gen)
//     p += skip record       optional (ResNet's h + LN2(..))
//     V  = act(p)            optional; Faa di Bruno jets (jet_device.h)
//
// and the reverse kernel is its adjoint: Vbar -> cotangent of the source record (or the encoder's weight gradient),
// the skip cotangent, and the LayerNorm parameter gradients.
//
// Thread map: a workgroup owns one 16-point HALF tile at a time (unit = 2 * tile + half); thread (n = tid & 15,
// g = tid >> 4) owns point n and the FPT features g, g + G, g + 2G, ... (G = Hp / FPT groups, 16 G <= 1024 threads),
// all K streams of each in registers — so a record is read once and written once, every global access of a quarter
// wave is one 64-byte row segment, and at width 256 a thread holds 4 features x K streams (no scratch at four waves
// per SIMD; with whole 32-point tiles and 8 features per thread the reverse kernel spilled 150-1600 VGPRs).  The
// per-point feature reductions LayerNorm needs (K means + K moments forward — kept per point in a small record so that
// the reverse launch re-reads instead of re-reducing them — and K + K more in reverse) go through LDS;
// per-feature sums over points (dgamma, dbeta, encoder gradient) are quarter-wave reductions accumulated in LDS
// across all units of the workgroup and flushed once.
#pragma once
#include "lm_common.h"

namespace pinn {
namespace lm {

enum { SRC_REC = 0, SRC_COORDS_LINEAR = 1, SRC_COORDS_FOURIER = 2 };

constexpr int kRedQ = 8;      // values per block reduction (K <= 7)
constexpr int kMaxWavesEw = 16;
constexpr int kPT = 16;       // points per unit (half a record tile)

struct EwArgs {
  int H, Hp, G;         // logical / padded feature count, feature groups (threads = 16 G)
  long long ntiles;     // tiles of this launch
  long long N;          // points of the call
  long long p_base;     // first point of this launch's tile 0
  int src_kind;
  const float* srcA;    // SRC_REC
  const float* encW;    // SRC_COORDS_LINEAR: packed [Hp][4];  SRC_COORDS_FOURIER: packed B^T [Mp][4]
  const float* encb;    // [Hp]
  int din, M;
  const float* x;       // (N, din - 1)
  const float* t;       // (N)
  const float* ln_g;    // packed [Hp] (zero beyond H) or null
  const float* ln_b;
  float eps;
  const float* skip;    // record or null
  int has_act;
  float act_param;
  float* V;             // forward output record
  // reverse
  const float* Vbar;    // cotangent record of V, or null when the cotangent comes from the head
  const float* U;       // head mode: [tile][K][32] cotangents of the output jets
  const float* w_out;   // head mode: packed [Hp]
  float* Zbar;          // cotangent of the source record (SRC_REC) or null
  float* Pbar;          // cotangent of the skip record or null
  float* d_ln_g;        // packed gradients (accumulated)
  float* d_ln_b;
  float* d_encW;        // [Hp][4]
  float* d_encb;
  float* det_partial;   // deterministic mode: per-workgroup partials [grid][7][1024] instead of float atomics
  float* stats;         // LayerNorm prologues: [tile][2 K][32] per-point sums (K stream sums, then K second-moment sums)
                        // written by the forward launch and re-read by the reverse one (saves two block reductions)
};

// Record access: element (row, point n) of this thread's feature group lives at tile_base + row * 32 floats + tid * 4
// bytes — a wave-uniform 64-bit base (SGPRs, scalar adds per row) plus ONE 32-bit lane offset shared by every access.
// Written as per-element 64-bit pointers, the optimizer hoists ~60 loop-invariant address pairs out of the tile loop
// and spills them (190-1600 VGPRs of scratch).
__device__ __forceinline__ float rec_ld(const float* tile_base, int row, unsigned voff) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(tile_base + (long long)row * kT) + voff);
}
__device__ __forceinline__ void rec_st(float* tile_base, int row, unsigned voff, float v) {
#ifdef PINN_FEXP_NOSTORE
  if (v != 12345.678f) return;  // timing experiment (tools/micro/fused_bench.hip): the store never happens, its operand stays live
#endif
  *reinterpret_cast<float*>(reinterpret_cast<char*>(tile_base + (long long)row * kT) + voff) = v;
}
// A kernel-argument pointer the optimizer may not reason about across loop iterations.  The element-wise kernels carry
// several rarely-taken source / head paths whose per-lane addresses (argument pointer + lane offset + row) are loop
// invariant; hoisted out of the unit loop they cost two VGPRs each for the whole kernel and pushed the LayerNorm adjoint
// 45 registers into scratch.  Laundered inside the loop body, the address is rebuilt where it is used.
template <typename T>
__device__ __forceinline__ T* in_loop(T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

__device__ __forceinline__ int in_loop_i(int v) {  // the same for a uniform integer (row strides: their multiples are addresses too)
  asm volatile("" : "+s"(v));
  return v;
}

// per-feature parameter vector: element f = g + G i  ->  base + G i floats (uniform) + 4 g bytes (lane)
__device__ __forceinline__ float vec_ld(const float* base, int row, unsigned goff) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + row) + goff);
}

// binomial coefficients for n <= 4 (branch-only, so that unrolled loops fold them to literals)
__host__ __device__ constexpr int binom(int n, int k) {
  return (k == 0 || k == n) ? 1 : (n == 2 ? 2 : (n == 3 ? 3 : (k == 2 ? 6 : 4)));
}

// sum over all threads that share this thread's point n (all feature groups); every thread gets the totals.
// `slot` alternates between two LDS areas so that one barrier per reduction suffices.
// Barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. an LDS-DMA prefetch in flight; the
// kernels that keep one across their reductions synchronise with this instead (all waves reach every barrier).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NQ, bool RAW = false>
__device__ __forceinline__ void block_sum(float (&q)[NQ], float* red, int& slot, int nwaves, int wave, int tid, int n) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    q[i] += __shfl_xor(q[i], 16);
    q[i] += __shfl_xor(q[i], 32);
  }
  float* area = red + slot * (kMaxWavesEw * kRedQ * kPT);
  slot ^= 1;
  if ((tid & 63) < kPT) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) area[(wave * kRedQ + i) * kPT + n] = q[i];
  }
  if constexpr (RAW) lds_barrier();
  else __syncthreads();
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    float s = 0.0f;
    for (int w = 0; w < nwaves; ++w) s += area[(w * kRedQ + i) * kPT + n];
    q[i] = s;
  }
}

// sum over the 16 lanes (points) of a quarter wave
__device__ __forceinline__ float pt_sum(float v) {
#pragma unroll
  for (int o = kPT / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// g(v) = v^(-1/2) and derivatives
__device__ __forceinline__ void rsqrt_derivs(float v, float (&g)[6]) {
  const float r = rsqrtf(v), w = r * r;
  g[0] = r;
  g[1] = -0.5f * r * w;
  g[2] = 0.75f * r * w * w;
  g[3] = -1.875f * r * w * w * w;
  g[4] = 6.5625f * r * w * w * w * w;
  g[5] = -29.53125f * r * w * w * w * w * w;
}

// per-point LayerNorm statistics of one direction: moments v_1..v_M and r_k = d^k (v^-1/2), k = 1..M
template <int M>
struct LnDir {
  float v[M > 0 ? M : 1];
  float r[M > 0 ? M : 1];
};

template <int NT, int NX>
struct LnPoint {
  float r0;
  float g[6];
  LnDir<NT> t;
  LnDir<NX> x;
};

// stream index of (direction offset lo, order j): j == 0 is the value stream
__device__ __forceinline__ constexpr int sidx(int lo, int j) { return j == 0 ? 0 : lo + j - 1; }

// second-moment sums -> per-point statistics (shared by the forward pass and the reverse pass that re-reads the sums)
template <int NT, int NX>
__device__ __forceinline__ void ln_point_from_moments(const float (&m)[1 + NT + NX], float invH, float eps, LnPoint<NT, NX>& S) {
  rsqrt_derivs(m[0] * invH + eps, S.g);
  S.r0 = S.g[0];
  if constexpr (NT > 0) {
#pragma unroll
    for (int k = 0; k < NT; ++k) S.t.v[k] = m[1 + k] * invH;
    dir_fwd<NT>(S.g, S.t.v, S.t.r);
  }
  if constexpr (NX > 0) {
#pragma unroll
    for (int k = 0; k < NX; ++k) S.x.v[k] = m[1 + NT + k] * invH;
    dir_fwd<NX>(S.g, S.x.v, S.x.r);
  }
}

// The reverse pass: the sums the forward launch kept -> centred streams and statistics, no reduction.
template <int NT, int NX, int FPT>
__device__ __forceinline__ void ln_stats_restore(float (&c)[FPT][1 + NT + NX], const bool (&valid)[FPT], int H, float eps,
                                                 LnPoint<NT, NX>& S, const float* stats_in, int ln) {
  constexpr int K = 1 + NT + NX;
  const float invH = 1.0f / (float)H;
  float q[K], m[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    q[s] = stats_in[s * kT + ln];
    m[s] = stats_in[(K + s) * kT + ln];
  }
#pragma unroll
  for (int i = 0; i < FPT; ++i)
#pragma unroll
    for (int s = 0; s < K; ++s) c[i][s] = valid[i] ? c[i][s] - q[s] * invH : 0.0f;
  ln_point_from_moments<NT, NX>(m, invH, eps, S);
}

// c (centred, zero on padding features) -> statistics.  Two block reductions.
template <int NT, int NX, int FPT, bool RAW = false>
__device__ __forceinline__ void ln_stats(float (&c)[FPT][1 + NT + NX], const bool (&valid)[FPT], int H, float eps,
                                         LnPoint<NT, NX>& S, float* red, int& slot, int nwaves, int wave, int tid, int ln,
                                         float* stats_out) {
  constexpr int K = 1 + NT + NX;
  const float invH = 1.0f / (float)H;
  float q[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < FPT; ++i) p += valid[i] ? c[i][s] : 0.0f;
    q[s] = p;
  }
  block_sum<K, RAW>(q, red, slot, nwaves, wave, tid, ln);
  if (stats_out && tid < kPT) {  // one lane per point keeps the sums for the reverse sweep
#pragma unroll
    for (int s = 0; s < K; ++s) stats_out[s * kT + ln] = q[s];
  }
#pragma unroll
  for (int i = 0; i < FPT; ++i)
#pragma unroll
    for (int s = 0; s < K; ++s) c[i][s] = valid[i] ? c[i][s] - q[s] * invH : 0.0f;
  float m[K];  // m[0] = sum c0^2 ; m[sidx(lo, k)] = sum_j C(k, j) sum c_j c_{k-j}
#pragma unroll
  for (int s = 0; s < K; ++s) m[s] = 0.0f;
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    m[0] = fmaf(c[i][0], c[i][0], m[0]);
#pragma unroll
    for (int k = 1; k <= NT; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) m[sidx(1, k)] = fmaf((float)binom(k, j) * c[i][sidx(1, j)], c[i][sidx(1, k - j)], m[sidx(1, k)]);
#pragma unroll
    for (int k = 1; k <= NX; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j)
        m[sidx(1 + NT, k)] = fmaf((float)binom(k, j) * c[i][sidx(1 + NT, j)], c[i][sidx(1 + NT, k - j)], m[sidx(1 + NT, k)]);
  }
  block_sum<K, RAW>(m, red, slot, nwaves, wave, tid, ln);
  if (stats_out && tid < kPT) {
#pragma unroll
    for (int s = 0; s < K; ++s) stats_out[(K + s) * kT + ln] = m[s];
  }
  ln_point_from_moments<NT, NX>(m, invH, eps, S);
}

// yhat jets of one element from its centred streams
template <int NT, int NX>
__device__ __forceinline__ void ln_yhat(const float (&c)[1 + NT + NX], const LnPoint<NT, NX>& S, float (&y)[1 + NT + NX]) {
  y[0] = c[0] * S.r0;
#pragma unroll
  for (int k = 1; k <= NT; ++k) {
    float a = c[sidx(1, k)] * S.r0;
#pragma unroll
    for (int j = 0; j < k; ++j) a = fmaf((float)binom(k, j) * c[sidx(1, j)], S.t.r[k - j - 1], a);
    y[sidx(1, k)] = a;
  }
#pragma unroll
  for (int k = 1; k <= NX; ++k) {
    float a = c[sidx(1 + NT, k)] * S.r0;
#pragma unroll
    for (int j = 0; j < k; ++j) a = fmaf((float)binom(k, j) * c[sidx(1 + NT, j)], S.x.r[k - j - 1], a);
    y[sidx(1 + NT, k)] = a;
  }
}

// Adjoint of LayerNorm for this thread's elements.  In: c (centred), pb = cotangent of gamma * yhat + beta.
// Out: pb <- cotangent of the pre-LayerNorm jets; dgamma / dbeta contributions per element in dg / dbt.
template <int NT, int NX, int FPT, bool RAW = false>
__device__ __forceinline__ void ln_backward(const float (&c)[FPT][1 + NT + NX], float (&pb)[FPT][1 + NT + NX],
                                            const bool (&valid)[FPT], const float (&gamv)[FPT], int G, float* pacc_g,
                                            float* pacc_b, int g, int H, const LnPoint<NT, NX>& S, float* red, int& slot,
                                            int nwaves, int wave, int tid, int ln) {
  constexpr int K = 1 + NT + NX;
  const float invH = 1.0f / (float)H;
  // rb[0] = rbar_0 ; rb[sidx(lo, j)] = rbar_j of that direction (j >= 1)
  float rb[K];
#pragma unroll
  for (int s = 0; s < K; ++s) rb[s] = 0.0f;
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    float y[K];
    ln_yhat<NT, NX>(c[i], S, y);
    float gsum = 0.0f;
#pragma unroll
    for (int s = 0; s < K; ++s) gsum = fmaf(pb[i][s], y[s], gsum);
    {  // dgamma / dbeta: sums over the 16 points of the unit, accumulated in this group's own LDS slots
      const float s0 = pt_sum(valid[i] ? gsum : 0.0f), s1 = pt_sum(valid[i] ? pb[i][0] : 0.0f);
      if (ln == 0) {
        pacc_g[g + G * i] += s0;
        pacc_b[g + G * i] += s1;
      }
    }
#pragma unroll
    for (int s = 0; s < K; ++s) pb[i][s] = valid[i] ? gamv[i] * pb[i][s] : 0.0f;  // now yhat-bar
    rb[0] = fmaf(c[i][0], pb[i][0], rb[0]);
#pragma unroll
    for (int k = 1; k <= NT; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) rb[sidx(1, k - j)] = fmaf((float)binom(k, j) * c[i][sidx(1, j)], pb[i][sidx(1, k)], rb[sidx(1, k - j)]);
#pragma unroll
    for (int k = 1; k <= NX; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j)
        rb[sidx(1 + NT, k - j)] = fmaf((float)binom(k, j) * c[i][sidx(1 + NT, j)], pb[i][sidx(1 + NT, k)], rb[sidx(1 + NT, k - j)]);
  }
  block_sum<K, RAW>(rb, red, slot, nwaves, wave, tid, ln);
  // per-point scalars: vbar_k per direction, vbar_0
  float v0b = S.g[1] * rb[0];
  float vbt[NT > 0 ? NT : 1], vbx[NX > 0 ? NX : 1];
  if constexpr (NT > 0) {
    float rbd[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) rbd[k] = rb[1 + k];
    v0b += dir_bwd<NT>(S.g, S.t.v, rbd, vbt);
  }
  if constexpr (NX > 0) {
    float rbd[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) rbd[k] = rb[1 + NT + k];
    v0b += dir_bwd<NX>(S.g, S.x.v, rbd, vbx);
  }
  const float k2 = 2.0f * invH;
  float mq[K];
#pragma unroll
  for (int s = 0; s < K; ++s) mq[s] = 0.0f;
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    float cb[K];
    cb[0] = fmaf(S.r0, pb[i][0], v0b * k2 * c[i][0]);
#pragma unroll
    for (int s = 1; s < K; ++s) cb[s] = 0.0f;
#pragma unroll
    for (int k = 1; k <= NT; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) {
        const float rr = (k - j == 0) ? S.r0 : S.t.r[k - j - 1];
        cb[sidx(1, j)] += (float)binom(k, j) * (rr * pb[i][sidx(1, k)] + vbt[k - 1] * k2 * c[i][sidx(1, k - j)]);
      }
#pragma unroll
    for (int k = 1; k <= NX; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) {
        const float rr = (k - j == 0) ? S.r0 : S.x.r[k - j - 1];
        cb[sidx(1 + NT, j)] += (float)binom(k, j) * (rr * pb[i][sidx(1 + NT, k)] + vbx[k - 1] * k2 * c[i][sidx(1 + NT, k - j)]);
      }
#pragma unroll
    for (int s = 0; s < K; ++s) {
      pb[i][s] = valid[i] ? cb[s] : 0.0f;
      mq[s] += pb[i][s];
    }
  }
  block_sum<K, RAW>(mq, red, slot, nwaves, wave, tid, ln);
#pragma unroll
  for (int i = 0; i < FPT; ++i)
#pragma unroll
    for (int s = 0; s < K; ++s) pb[i][s] = valid[i] ? pb[i][s] - mq[s] * invH : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// source jets of this thread's elements
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int NX, int FPT>
__device__ __forceinline__ void load_source(const EwArgs& a, long long rec_off, unsigned voff, unsigned goff, const float (&xin)[4],
                                            float (&z)[FPT][1 + NT + NX]) {
  constexpr int K = 1 + NT + NX;
  if (a.src_kind == SRC_REC) {
    const float* base = in_loop(a.srcA) + rec_off;
#pragma unroll
    for (int i = 0; i < FPT; ++i)
#pragma unroll
      for (int s = 0; s < K; ++s) z[i][s] = rec_ld(base, s * a.Hp + a.G * i, voff);
  } else if (a.src_kind == SRC_COORDS_LINEAR) {
    const float* encW = in_loop(a.encW);
    const float* encb = in_loop(a.encb);
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(encW + 4 * a.G * i) + 4u * goff);  // zero beyond din / H
      float v = vec_ld(encb, a.G * i, goff);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) v = fmaf(xin[cc], w[cc], v);
#pragma unroll
      for (int s = 0; s < K; ++s) z[i][s] = 0.0f;
      z[i][0] = v;
      float wt = w[0];
#pragma unroll
      for (int cc = 1; cc < 4; ++cc) wt = (cc == a.din - 1) ? w[cc] : wt;
      if constexpr (NT >= 1) z[i][1] = wt;
      if constexpr (NX >= 1) z[i][1 + NT] = w[0];
    }
  }
}

// Lanes beyond the batch (the ragged end of the last tile) take the LAST point's coordinates, not zeros: their cotangents
// are zero (lm_head.h seeds nothing for them), so they contribute 0 x finite to every reduction.  With zero coordinates and
// a zero-initialised first bias every feature of such a lane is equal, a LayerNorm there divides by sqrt(eps), its 4th-order
// jets reach 1e20 after a few layers, overflow fp32, and 0 x inf = NaN went into every weight gradient
// (tools/fuzz_parity.py: attention x Cahn-Hilliard 1-D, two layers, any width).
__device__ __forceinline__ void load_coords(const EwArgs& a, long long unit, int n, float (&xin)[4], bool& ok) {
  long long p = a.p_base + unit * kPT + n;
  ok = p < a.N;
  p = ok ? p : a.N - 1;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) xin[cc] = 0.0f;
  if (a.src_kind != SRC_REC) {
    const float* xs = in_loop(a.x);
    const float* ts = in_loop(a.t);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      if (cc < a.din - 1) xin[cc] = xs[p * (a.din - 1) + cc];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      if (cc == a.din - 1) xin[cc] = ts[p];
  }
}

// pre-activation jets p of element i: zc holds the source jets, or — with a LayerNorm — their centred streams
template <int NT, int NX, int FPT, bool LN>
__device__ __forceinline__ void elem_pre(const EwArgs& a, const float* skip_base, int row, unsigned voff, float gam, float bet,
                                         const float (&zc)[1 + NT + NX], const LnPoint<NT, NX>& S, float (&p)[1 + NT + NX]) {
  constexpr int K = 1 + NT + NX;
  if constexpr (LN) {
    float y[K];
    ln_yhat<NT, NX>(zc, S, y);
    p[0] = fmaf(gam, y[0], bet);
#pragma unroll
    for (int s = 1; s < K; ++s) p[s] = gam * y[s];
  } else {
#pragma unroll
    for (int s = 0; s < K; ++s) p[s] = zc[s];
  }
  if (skip_base) {
#pragma unroll
    for (int s = 0; s < K; ++s) p[s] += rec_ld(skip_base, s * a.Hp + row, voff);
  }
}


// Fourier features of the coordinates (fourier.py:12-16): V[f] = sin(x B)_f for f < M, cos(x B)_{f-M} for M <= f < 2M,
// jets included.  A forward-only prologue of its own: B is a buffer, nothing is differentiated through it.
template <int NT, int NX, int FPT>
__global__ __launch_bounds__(1024) void lm_fourier_fwd(const EwArgs a) {
  constexpr int K = 1 + NT + NX;
  const int tid = threadIdx.x, n = tid & (kPT - 1), g = tid >> 4;
  const unsigned voff = static_cast<unsigned>(g * kT + n) * 4u;
  for (long long uu = 2LL * blockIdx.x; uu < 2 * a.ntiles; uu += (uu & 1) ? 2LL * gridDim.x - 1 : 1) {
    // both 16-point halves of a tile back to back in the SAME workgroup: a 128-byte record row is then fetched from HBM
    // once (the second half hits this CU's caches) and its two 64-byte stores merge; with the halves on neighbouring
    // workgroups (different XCDs under round-robin dispatch) every line crossed the fabric twice
    const long long unit = uu;
    float xin[4];
    bool ok;
    load_coords(a, unit, n, xin, ok);
    const long long rec_off = (unit >> 1) * (long long)K * a.Hp * kT + (unit & 1) * kPT;
    float* out = a.V + rec_off;
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      const int f = g + a.G * i;
      const bool on = f < 2 * a.M;
      const int m = f < a.M ? f : f - a.M;
      const f32x4 w = on ? *reinterpret_cast<const f32x4*>(a.encW + 4 * m) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      float v = 0.0f;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) v = fmaf(xin[cc], w[cc], v);
      float wt = w[0];
#pragma unroll
      for (int cc = 1; cc < 4; ++cc) wt = (cc == a.din - 1) ? w[cc] : wt;
      float sn, cs;
      fast_sincosf(v, &sn, &cs);
      float fd[6];
      fd[0] = f < a.M ? sn : cs;
      fd[1] = f < a.M ? cs : -sn;
      fd[2] = -fd[0];
      fd[3] = -fd[1];
      fd[4] = fd[0];
      fd[5] = fd[1];
      float zz[K], yy[K];
#pragma unroll
      for (int s = 0; s < K; ++s) zz[s] = 0.0f;
      if constexpr (NT >= 1) zz[1] = wt;
      if constexpr (NX >= 1) zz[1 + NT] = w[0];
      yy[0] = fd[0];
      dir_fwd<NT>(fd, zz + 1, yy + 1);
      dir_fwd<NX>(fd, zz + 1 + NT, yy + 1 + NT);
#pragma unroll
      for (int s = 0; s < K; ++s) rec_st(out, s * a.Hp + a.G * i, voff, on ? yy[s] : 0.0f);
    }
  }
}

template <int ACT, int NT, int NX, int FPT, bool LN>
__global__ __launch_bounds__(1024) void lm_ew_fwd(const EwArgs a) {
  constexpr int K = 1 + NT + NX;
  __shared__ float red[LN ? 2 * kMaxWavesEw * kRedQ * kPT : 1];
  const int tid = threadIdx.x, n = tid & (kPT - 1), g = tid >> 4;
  const int wave = tid >> 6, nwaves = (a.G + 3) >> 2;
  int slot = 0;
  const unsigned voff = static_cast<unsigned>(g * kT + n) * 4u, goff = static_cast<unsigned>(g) * 4u;
  bool valid[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) valid[i] = g + a.G * i < a.H;
  // LayerNorm scale / shift of this thread's features: loop-invariant, kept in registers.  Re-loading them per unit
  // looked cheaper, but the compiler hoists the per-lane 64-bit ADDRESSES out of the loop instead (two registers per
  // value) and spilled them: 36 scratch loads per thread and unit in the LayerNorm adjoint.
  float gamv[FPT], betv[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    gamv[i] = LN ? vec_ld(a.ln_g, a.G * i, goff) : 0.0f;
    betv[i] = LN ? vec_ld(a.ln_b, a.G * i, goff) : 0.0f;
  }
  for (long long uu = 2LL * blockIdx.x; uu < 2 * a.ntiles; uu += (uu & 1) ? 2LL * gridDim.x - 1 : 1) {
    // both 16-point halves of a tile back to back in the SAME workgroup: a 128-byte record row is then fetched from HBM
    // once (the second half hits this CU's caches) and its two 64-byte stores merge; with the halves on neighbouring
    // workgroups (different XCDs under round-robin dispatch) every line crossed the fabric twice
    const long long unit = uu;
    EwArgs al = a;  // row strides laundered per unit: their multiples are addresses, and hoisted addresses are registers
    al.Hp = in_loop_i(a.Hp);
    al.G = in_loop_i(a.G);
    float xin[4];
    bool ok;
    load_coords(al, unit, n, xin, ok);
    const long long rec_off = (unit >> 1) * (long long)K * al.Hp * kT + (unit & 1) * kPT;
    float zc[FPT][K];
    LnPoint<NT, NX> S;
    load_source<NT, NX, FPT>(al, rec_off, voff, goff, xin, zc);
    if constexpr (LN)
      ln_stats<NT, NX, FPT>(zc, valid, al.H, al.eps, S, red, slot, nwaves, wave, tid, n,
                            al.stats ? al.stats + (unit >> 1) * (2LL * K * kT) + (unit & 1) * kPT : nullptr);
    float* out = in_loop(al.V) + rec_off;
    const float* skip_base = al.skip ? in_loop(al.skip) + rec_off : nullptr;
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      float p[K], v[K];
      elem_pre<NT, NX, FPT, LN>(al, skip_base, al.G * i, voff, gamv[i], betv[i], zc[i], S, p);
      if (al.has_act) {
        act_fwd<ACT, NT, NX>(al.act_param, p, v);
      } else {
#pragma unroll
        for (int s = 0; s < K; ++s) v[s] = p[s];
      }
#pragma unroll
      for (int s = 0; s < K; ++s) rec_st(out, s * al.Hp + al.G * i, voff, valid[i] ? v[s] : 0.0f);
    }
  }
}

template <int ACT, int NT, int NX, int FPT, bool LN>
__global__ __launch_bounds__(1024) void lm_ew_bwd(const EwArgs a) {
  constexpr int K = 1 + NT + NX;
  constexpr int kAcc = 7;  // per-feature accumulators: dgamma, dbeta | encoder: 4 weight columns + bias
  __shared__ float red[LN ? 2 * kMaxWavesEw * kRedQ * kPT : 1];
  __shared__ float pacc[kAcc * 1024];
  const int tid = threadIdx.x, n = tid & (kPT - 1), g = tid >> 4;
  const int wave = tid >> 6, nwaves = (a.G + 3) >> 2;
  const int nthreads = kPT * a.G;
  int slot = 0;
  const unsigned voff = static_cast<unsigned>(g * kT + n) * 4u, goff = static_cast<unsigned>(g) * 4u;
  bool valid[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) valid[i] = g + a.G * i < a.H;
  // LayerNorm scale / shift of this thread's features: loop-invariant, kept in registers.  Re-loading them per unit
  // looked cheaper, but the compiler hoists the per-lane 64-bit ADDRESSES out of the loop instead (two registers per
  // value) and spilled them: 36 scratch loads per thread and unit in the LayerNorm adjoint.
  float gamv[FPT], betv[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    gamv[i] = LN ? vec_ld(a.ln_g, a.G * i, goff) : 0.0f;
    betv[i] = LN ? vec_ld(a.ln_b, a.G * i, goff) : 0.0f;
  }
  const bool enc_grad = a.src_kind == SRC_COORDS_LINEAR && a.d_encW;
  if (LN || enc_grad) {
    for (int i = tid; i < kAcc * 1024; i += nthreads) pacc[i] = 0.0f;
    __syncthreads();
  }
  for (long long uu = 2LL * blockIdx.x; uu < 2 * a.ntiles; uu += (uu & 1) ? 2LL * gridDim.x - 1 : 1) {
    // both 16-point halves of a tile back to back in the SAME workgroup: a 128-byte record row is then fetched from HBM
    // once (the second half hits this CU's caches) and its two 64-byte stores merge; with the halves on neighbouring
    // workgroups (different XCDs under round-robin dispatch) every line crossed the fabric twice
    const long long unit = uu;
    EwArgs al = a;  // row strides laundered per unit (see lm_ew_fwd)
    al.Hp = in_loop_i(a.Hp);
    al.G = in_loop_i(a.G);
    float xin[4];
    bool ok;
    load_coords(al, unit, n, xin, ok);
    const long long rec_off = (unit >> 1) * (long long)K * al.Hp * kT + (unit & 1) * kPT;
    float zc[FPT][K], pb[FPT][K];
    LnPoint<NT, NX> S;
    load_source<NT, NX, FPT>(al, rec_off, voff, goff, xin, zc);
    // cotangent of V
    if (al.Vbar) {
      const float* base = in_loop(al.Vbar) + rec_off;
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) pb[i][s] = rec_ld(base, s * al.Hp + al.G * i, voff);
    } else {
      float ub[K];
      const float* U = in_loop(al.U);
      const float* w_out = in_loop(al.w_out);
#pragma unroll
      for (int s = 0; s < K; ++s) ub[s] = U[((unit >> 1) * K + s) * kT + (unit & 1) * kPT + n];
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const float w = vec_ld(w_out, al.G * i, goff);
#pragma unroll
        for (int s = 0; s < K; ++s) pb[i][s] = w * ub[s];
      }
    }
    if constexpr (LN)  // the engine always gives the reverse launch the sums its forward launch kept
      ln_stats_restore<NT, NX, FPT>(zc, valid, al.H, al.eps, S, al.stats + (unit >> 1) * (2LL * K * kT) + (unit & 1) * kPT, n);
    const float* skip_base = al.skip ? in_loop(al.skip) + rec_off : nullptr;
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      float p[K];
      elem_pre<NT, NX, FPT, LN>(al, skip_base, al.G * i, voff, gamv[i], betv[i], zc[i], S, p);
      if (al.has_act) {
        float zb[K];
        act_bwd<ACT, NT, NX>(al.act_param, p, pb[i], zb);
#pragma unroll
        for (int s = 0; s < K; ++s) pb[i][s] = zb[s];
      }
#pragma unroll
      for (int s = 0; s < K; ++s) pb[i][s] = valid[i] ? pb[i][s] : 0.0f;
      __builtin_amdgcn_sched_barrier(0);  // one element at a time: interleaving the four raises the register peak into scratch
    }
    if (al.Pbar) {
      float* out = in_loop(al.Pbar) + rec_off;
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) rec_st(out, s * al.Hp + al.G * i, voff, pb[i][s]);
    }
    if constexpr (LN)
      ln_backward<NT, NX, FPT>(zc, pb, valid, gamv, al.G, pacc, pacc + 1024, g, al.H, S, red, slot, nwaves, wave, tid, n);
    if (al.src_kind == SRC_REC) {
      if (al.Zbar) {
        float* out = in_loop(al.Zbar) + rec_off;
#pragma unroll
        for (int i = 0; i < FPT; ++i)
#pragma unroll
          for (int s = 0; s < K; ++s) rec_st(out, s * al.Hp + al.G * i, voff, pb[i][s]);
      }
    } else if (enc_grad) {
      // first Linear: dW[f][c] += sum_n zb_0 coord_c (+ zb_t1 for the time column, + zb_x1 for column 0); db[f] += sum_n zb_0
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        float gw[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) gw[cc] = pb[i][0] * xin[cc];
        if constexpr (NT >= 1) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) gw[cc] += (cc == al.din - 1) ? pb[i][1] : 0.0f;
        }
        if constexpr (NX >= 1) gw[0] += pb[i][1 + NT];
        float gb = pb[i][0];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) gw[cc] = pt_sum(gw[cc]);
        gb = pt_sum(gb);
        if (n == 0) {  // this thread group is the only writer of its features' slots
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) pacc[(2 + cc) * 1024 + g + al.G * i] += gw[cc];
          pacc[6 * 1024 + g + al.G * i] += gb;
        }
      }
    }
  }
  if ((LN || enc_grad) && a.det_partial) {  // plain stores; lm_reduce_slots adds them up in workgroup order
    __syncthreads();
    float* P = a.det_partial + (long long)blockIdx.x * (kAcc * 1024);
    for (int i = tid; i < kAcc * 1024; i += nthreads) P[i] = pacc[i];
  } else if (LN || enc_grad) {
    __syncthreads();
    for (int f = tid; f < a.H; f += nthreads) {
      if (LN && a.d_ln_g) {
        atomicAdd(a.d_ln_g + f, pacc[f]);
        atomicAdd(a.d_ln_b + f, pacc[1024 + f]);
      }
      if (enc_grad) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) atomicAdd(a.d_encW + 4 * f + cc, pacc[(2 + cc) * 1024 + f]);
        if (a.d_encb) atomicAdd(a.d_encb + f, pacc[6 * 1024 + f]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// LayerNorm prologue adjoint with the NEXT unit prefetched by LDS-DMA.
//
// lm_ew_bwd holds a unit in registers (one 1024-thread workgroup per CU), so nothing overlaps its load, reduce and store
// phases: 3.2 TB/s where the bare access pattern reaches 4.7.  Here the source and cotangent jets of
// unit u + 1 (and its saved LayerNorm sums) travel HBM -> LDS by global_load_lds_dwordx4 while unit u is being
// reduced: 2 K pieces of 1 KB per wave, each 16 half rows of 64 bytes gathered through the per-lane source address
// into a linear [row][16 points] image.  Barriers inside the unit are LDS-only (lds_barrier), so the DMA stays in
// flight across them; it is retired with vmcnt(0) just before the unit's final stores, which are then never waited
// for.  The skip record (ResNet's second prologue only) is still read into registers at the top of a unit.
// (The same prefetch in the FORWARD prologue measured no gain — 294 vs 287 us: two reductions are too little work to
// cover a unit's transfer — so the forward kernel stays register-staged.)
// Requirements (checked by the launcher): record source, cotangent record, LayerNorm, K * Hp <= 1024 rows.
// Dynamic LDS: 2 images of 64 K Hp bytes + 16 half rows of sums + 8 KB of gamma/beta accumulators + 16 KB of partials.
// ---------------------------------------------------------------------------------------------------------------
template <int ACT, int NT, int NX, int FPT>
__global__ __launch_bounds__(1024) void lm_ew_bwd_dma(const EwArgs a) {
  constexpr int K = 1 + NT + NX;
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int R = K * a.Hp;                      // rows of one record per unit
  float* zimg = dsm;                           // [R][16]
  float* pimg = zimg + R * kPT;                // [R][16]
  float* simg = pimg + R * kPT;                // [16 half rows][16]: the first 2 K are this unit's saved sums
  float* pacc = simg + 16 * kPT;               // [2][1024]
  float* red = pacc + 2 * 1024;                // [2][16 waves][8][16]
  const int tid = threadIdx.x, n = tid & (kPT - 1), g = tid >> 4, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (a.G + 3) >> 2;
  const int nthreads = kPT * a.G;
  int slot = 0;
  const unsigned voff = static_cast<unsigned>(g * kT + n) * 4u, goff = static_cast<unsigned>(g) * 4u;
  bool valid[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) valid[i] = g + a.G * i < a.H;
  float gamv[FPT], betv[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    gamv[i] = vec_ld(a.ln_g, a.G * i, goff);
    betv[i] = vec_ld(a.ln_b, a.G * i, goff);
  }
  for (int i = tid; i < 2 * 1024; i += nthreads) pacc[i] = 0.0f;

  // DMA of one unit: pieces 0 .. R/16 - 1 of the source record, the same of the cotangent record, one of the sums
  const int ppr = R >> 4;  // pieces per record
  const unsigned lsrc = static_cast<unsigned>((lane >> 2) * kT + (lane & 3) * 4) * 4u;  // half row (lane >> 2), quarter (lane & 3)
  auto issue = [&](long long unit) {
    const long long rec_off = (unit >> 1) * (long long)K * a.Hp * kT + (unit & 1) * kPT;
    for (int p = wave; p < 2 * ppr; p += nwaves) {
      const bool isz = p < ppr;
      const int j = isz ? p : p - ppr;
      const float* base = uniform_ptr((isz ? a.srcA : a.Vbar) + rec_off + (long long)j * 16 * kT);
      float* dst = (isz ? zimg : pimg) + j * 256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + lsrc),
                                       (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    }
    if (wave == 0) {  // [tile][2 K][32] sums: 2 K half rows (<= 14), gathered like the others; rows beyond are never read
      const float* base = uniform_ptr(a.stats + (unit >> 1) * (2LL * K * kT) + (unit & 1) * kPT);
      const unsigned ls = static_cast<unsigned>(((lane >> 2) < 2 * K ? (lane >> 2) : 2 * K - 1) * kT + (lane & 3) * 4) * 4u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + ls),
                                       (__attribute__((address_space(3))) void*)(simg), 16, 0, 0);
    }
  };

  // 16-point units dealt as contiguous runs (as in lm_fused.h): 3 125 tiles on 256 workgroups are 13 tiles for the slowest dealt
  // tile by tile, 12.5 dealt by unit; contiguous, so that the two halves of a tile stay with one workgroup, back to back
  const long long n_units = 2 * a.ntiles;
  const long long u_q = n_units / gridDim.x, u_r = n_units % gridDim.x;
  const long long first = (long long)blockIdx.x * u_q + ((long long)blockIdx.x < u_r ? blockIdx.x : u_r);
  const long long last = first + u_q + ((long long)blockIdx.x < u_r ? 1 : 0);
  auto next_of = [&](long long uu) { return uu + 1; };
  if (first < last) issue(first);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (long long uu = first; uu < last; uu = next_of(uu)) {
    const long long unit = uu;
    const long long rec_off = (unit >> 1) * (long long)K * a.Hp * kT + (unit & 1) * kPT;
    lds_barrier();  // every wave has retired its pieces of this unit (and zeroed its share of pacc the first time round)
    const int Hp = in_loop_i(a.Hp), G = in_loop_i(a.G);  // 32 image addresses are NOT loop invariants worth 32 registers
    float zc[FPT][K], pb[FPT][K];
    {
      const float* zl = zimg + g * kPT + n;
      const float* pl = pimg + g * kPT + n;
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) {
          zc[i][s] = zl[(s * Hp + G * i) * kPT];
          pb[i][s] = pl[(s * Hp + G * i) * kPT];
        }
    }
    LnPoint<NT, NX> S;
    {
      const float invH = 1.0f / (float)a.H;
      float q[K], m[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        q[s] = simg[s * kPT + n];
        m[s] = simg[(K + s) * kPT + n];
      }
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) zc[i][s] = valid[i] ? zc[i][s] - q[s] * invH : 0.0f;
      ln_point_from_moments<NT, NX>(m, invH, a.eps, S);
    }
    // the skip record's jets (ordinary loads: requested and consumed BEFORE the next DMA goes out, so that their wait
    // does not drain it): p = LN(z) gamma + beta + skip, then the activation adjoint
    const bool has_skip = a.skip != nullptr;
    {
      const float* sb = has_skip ? in_loop(a.skip) + rec_off : nullptr;
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        float p[K];
        elem_pre<NT, NX, FPT, true>(a, sb, G * i, voff, gamv[i], betv[i], zc[i], S, p);
        if (a.has_act) {
          float zb[K];
          act_bwd<ACT, NT, NX>(a.act_param, p, pb[i], zb);
#pragma unroll
          for (int s = 0; s < K; ++s) pb[i][s] = zb[s];
        }
#pragma unroll
        for (int s = 0; s < K; ++s) pb[i][s] = valid[i] ? pb[i][s] : 0.0f;
      }
    }
    lds_barrier();  // all waves have taken their elements out of the images: the next unit may land
    const long long nxt = next_of(uu);
    if (nxt < last) issue(nxt);
    if (a.Pbar) {
      float* out = in_loop(a.Pbar) + rec_off;
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) rec_st(out, s * Hp + G * i, voff, pb[i][s]);
    }
    ln_backward<NT, NX, FPT, true>(zc, pb, valid, gamv, G, pacc, pacc + 1024, g, a.H, S, red, slot, nwaves, wave, tid, n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetch (and the Pbar stores) retired; the stores below are not waited for
    {
      float* out = in_loop(a.Zbar) + rec_off;
#pragma unroll
      for (int i = 0; i < FPT; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) rec_st(out, s * Hp + G * i, voff, pb[i][s]);
    }
  }
  __syncthreads();
  if (a.det_partial) {  // plain stores; lm_reduce_slots adds them up in workgroup order
    float* P = a.det_partial + (long long)blockIdx.x * (7 * 1024);
    for (int i = tid; i < 2 * 1024; i += nthreads) P[i] = pacc[i];
  } else if (a.d_ln_g) {
    for (int f = tid; f < a.H; f += nthreads) {
      atomicAdd(a.d_ln_g + f, pacc[f]);
      atomicAdd(a.d_ln_b + f, pacc[1024 + f]);
    }
  }
}

inline size_t lm_ew_bwd_dma_lds_bytes(int K, int Hp) { return sizeof(float) * ((size_t)2 * K * Hp * kPT + 16 * kPT + 2 * 1024 + 2 * kMaxWavesEw * kRedQ * kPT); }
inline bool lm_ew_bwd_dma_ok(const EwArgs& a, int K) {
  return a.ln_g && a.src_kind == SRC_REC && a.Vbar && a.Zbar && a.stats && K * a.Hp <= 1024 && (a.Hp % 16) == 0;
}

}  // namespace lm
}  // namespace pinn
