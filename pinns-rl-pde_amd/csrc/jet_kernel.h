// Fused jet kernel for the plain-MLP family (feedforward / fourier / SIREN), gfx950.
//
// One 256-thread workgroup (4 waves) owns a TILE of T = 32 collocation points and pushes all
// K = 1 + NT + NX derivative streams of those points through the whole network — forward jets,
// PDE epilogue and (BWD) the reverse sweep — without leaving the CU.
//
//   registers  Wave w owns output-feature tiles {w, w+4} of every layer.  Its activations live in
//              the MFMA accumulator layout (32 feature rows in 16 registers x 2 lane halves, 32
//              point columns on lanes) for ALL K streams: exactly what the per-element activation
//              jets need, so nothing is transposed between the GEMM and the activation.
//   LDS        "stream-serial" staging: one stream at a time is published as S[f][n] (Hmax rows of
//              32 points + 4 pad, double-buffered, one barrier per stream step), so the LDS
//              footprint is independent of K: 2*Hmax*36*4 B forward (36 KB at width 128), twice
//              that with the reverse sweep — two workgroups per CU overlap each other's
//              activation/barrier phases with MFMA.
//   MFMA       v_mfma_f32_32x32x2_f32, exact fp32 (157 TFLOP/s peak).  Three GEMMs per layer with
//              equal MFMA counts:
//                z_s   = W a_s          A = weight rows   (global/L2, 16 B per lane, prefetched)
//                                       B = S[k][n]       (ds_read_b32 along the point axis)
//                abar_s= W^T zbar_s     A = weight columns(global, 128 B coalesced dwords)
//                dW   += zbar_s a_s^T   A = Z[j][n], B = A[k][n] (both ds_read_b128 along n)
//              dW tiles are flushed with float atomics whose wave footprint is 2 x 128-byte rows.
//   tape       what the reverse sweep re-reads, in accumulator layout, private to the workgroup
//              (same lanes write and read it; L2/MALL resident): the activation VALUE (tanh,
//              sigmoid: derivatives are polynomials of it) or the pre-activation (sin, gelu,
//              piecewise linear) plus the K-1 pre-activation derivative streams.
//
// Algorithmic FLOPs per point: K * 2 * sum(in*out) forward, 3x that with the reverse sweep
// (SURVEY.md §8d).  tests/jet_model.py is the executable specification of the arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <utility>

#include "jet_device.h"

namespace pinn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 32;   // points per tile
constexpr int kTP = 36;  // padded LDS row (floats): conflict-free b32 column reads and b128 row reads
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
  int ld;          // row stride of W / dW in floats (= in_dim unless this is a column-chunk view of a wider matrix)
  int act;
  float act_param;
  // LayerNorm that follows this Linear (ResNet / attention); null for the plain-MLP family
  const float* ln_g;
  const float* ln_b;
  float* d_ln_g;
  float* d_ln_b;
};

struct NetDev {
  int enc;            // ENC_LINEAR: first Linear (din -> enc_out) + activation; ENC_FOURIER: [sin, cos](inp @ B)
  int din;            // input_dim (time = last column)
  int enc_out;        // features after the encoding
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
  int arch;    // PinnArch (selects the kernel family)
  float ln_eps;
};

struct KernelArgs {
  NetDev net;
  PdeDev pde;
  const float* x;  // (N, din-1)
  const float* t;  // (N, 1)
  long long N;
  int mode;
  float grad_scale;                         // MODE_PDE backward: cotangent of sum_n l(r_n)
  float* jets_out[PINN_MAX_STREAMS];        // MODE_JETS
  const float* jets_bar[PINN_MAX_STREAMS];  // MODE_JETS backward
  float* residual_out;                      // MODE_PDE, nullable
  float* loss_sum;                          // MODE_PDE, nullable
  const float* res_bar;                     // MODE_PDE backward, nullable: external cotangent of r (N floats)
  float* tape;                              // BWD workspace
  long long tape_stride;                    // floats per workgroup
  unsigned long long* stamps;               // diagnostic builds (-DPINN_STAMPS) only: [grid][4 waves][kNumStamps] cycles
};

// Every kernel of this family may use the whole 160 KB LDS of a CU as dynamic shared memory.  The attribute is set
// ONCE per (kernel, device) — not per launch: it is a driver call on the launch path, and it is not allowed while
// the stream is being captured into a HIP graph.
inline hipError_t allow_full_lds(const void* kern) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> guard(mu);
  if (done.count({kern, dev})) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.insert({kern, dev});
  return e;
}

// In-kernel phase timing for diagnostic builds; in normal builds these expand to nothing.
constexpr int kNumStamps = 16;
enum { ST_STAGE = 0, ST_ENCODE, ST_FWD_GEMM, ST_FWD_EW, ST_OUT, ST_EPI, ST_B0, ST_BWD_EW, ST_BWD_STREAM, ST_BWD_FLUSH,
       ST_ENC_BWD, ST_TOTAL, ST_BWD_DX, ST_BWD_PUT };
#ifdef PINN_STAMPS
#define PINN_STAMP_DECL unsigned long long st_acc[kNumStamps] = {}; unsigned long long st_prev = pinn_now(); const unsigned long long st_begin = st_prev;
#define PINN_STAMP(idx) do { const unsigned long long st_now = pinn_now(); st_acc[idx] += st_now - st_prev; st_prev = st_now; } while (0)
__device__ __forceinline__ unsigned long long pinn_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#else
#define PINN_STAMP_DECL
#define PINN_STAMP(idx)
#endif

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// Tape layout: record (slot l, tile jt, stream s) is 16 registers x 256 threads, stored so that the four
// consecutive accumulator registers 4q .. 4q+3 of a thread are one aligned 16-byte word ([q][tid][i]): the
// reverse sweep moves it with global_load/store_dwordx4 (4x fewer VMEM instructions than dword accesses).
// The slot / stream / group part of the address is wave-uniform (scalar base), the thread part a 32-bit byte
// offset: global_load/store_dwordx4 v, v_off, s[base:base+1] — no per-access 64-bit address VGPR pairs.
__device__ __forceinline__ long long tape_qbase(int l, int jt, int ntile, int K, int s, int q) {
  return ((((long long)(l * ntile + jt) * K + s) * 4 + q) * kThreads) * 4;
}
__device__ __forceinline__ f32x4 tape_ld4(const float* tape, int l, int jt, int ntile, int K, int s, int q, int tid) {
  const char* base = reinterpret_cast<const char*>(tape + tape_qbase(l, jt, ntile, K, s, q));
  return *reinterpret_cast<const f32x4*>(base + static_cast<unsigned>(tid) * 16u);
}
__device__ __forceinline__ void tape_st4(float* tape, int l, int jt, int ntile, int K, int s, int q, int tid, f32x4 v) {
  char* base = reinterpret_cast<char*>(tape + tape_qbase(l, jt, ntile, K, s, q));
  *reinterpret_cast<f32x4*>(base + static_cast<unsigned>(tid) * 16u) = v;
}

struct Lane {
  int tid, wave, ln, lh;
};

// The layer table is indexed with a run-time layer number, which makes the compiler keep it in private
// memory (per-lane loads).  Re-uniformise each field so that sizes and base pointers live in SGPRs.
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  // rebuild the pointer IN THE GLOBAL ADDRESS SPACE, otherwise every access through it becomes a flat_* op
  typedef T __attribute__((address_space(1))) * global_ptr;
  return (T*)(global_ptr)((static_cast<unsigned long long>(hi) << 32) | lo);
}

__device__ __forceinline__ LayerDev uniform_layer(const LayerDev& s) {
  LayerDev u;
  u.W = uniform_ptr(s.W);
  u.b = uniform_ptr(s.b);
  u.dW = uniform_ptr(s.dW);
  u.db = uniform_ptr(s.db);
  u.in_dim = __builtin_amdgcn_readfirstlane(s.in_dim);
  u.ld = __builtin_amdgcn_readfirstlane(s.ld);
  u.out_dim = __builtin_amdgcn_readfirstlane(s.out_dim);
  u.ln_g = uniform_ptr(s.ln_g);
  u.ln_b = uniform_ptr(s.ln_b);
  u.d_ln_g = uniform_ptr(s.d_ln_g);
  u.d_ln_b = uniform_ptr(s.d_ln_b);
  u.act = __builtin_amdgcn_readfirstlane(s.act);
  u.act_param = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s.act_param)));
  return u;
}

// Publish ONE stream of a register tile set into an LDS stream buffer (rows = features, cols = points).
template <int NTILE>
__device__ __forceinline__ void stage_one(const f32x16 (&v)[NTILE], float* buf, int dim, const Lane& L) {
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int ft = L.wave + kWaves * jt;
    if (ft * 32 < dim) {
#pragma unroll
      for (int r = 0; r < 16; ++r) buf[(ft * 32 + acc_row(r, L.lh)) * kTP + L.ln] = v[jt][r];
    }
  }
}

// acc[jt] += W[own rows of tile jt][:] . S[:][n]   (one stream).  Loads run one k-group ahead of the MFMAs.
template <int NTILE>
__device__ __forceinline__ void gemm_rows(f32x16 (&acc)[NTILE], const LayerDev& Ly, const float* S, const Lane& L) {
  const int in = Ly.in_dim, ld = Ly.ld;
  const float* xcol = S + (4 * L.lh) * kTP + L.ln;
  const float* wrow[NTILE];
  bool on[NTILE];
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int ft = L.wave + kWaves * jt;
    on[jt] = ft * 32 < Ly.out_dim;
    wrow[jt] = Ly.W + (long long)((on[jt] ? ft : 0) * 32 + L.ln) * ld + 4 * L.lh;
  }
  if (!on[0]) return;
  f32x4 wc[NTILE], wn[NTILE];
  float bc[4], bn[4];
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) wc[jt] = *reinterpret_cast<const f32x4*>(wrow[jt]);
#pragma unroll
  for (int i = 0; i < 4; ++i) bc[i] = xcol[i * kTP];
  for (int g = 0; g < in; g += 8) {
    const int gn = g + 8 < in ? g + 8 : g;  // the last iteration re-loads its own group (harmless, branch-free)
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) wn[jt] = *reinterpret_cast<const f32x4*>(wrow[jt] + gn);
#pragma unroll
    for (int i = 0; i < 4; ++i) bn[i] = xcol[(gn + i) * kTP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
        if (on[jt]) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[jt][i], bc[i], acc[jt], 0, 0, 0);
    }
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) wc[jt] = wn[jt];
#pragma unroll
    for (int i = 0; i < 4; ++i) bc[i] = bn[i];
  }
}

// acc[jt] += W[:][own columns of tile jt]^T . Z[:][n]   (one stream; delta-propagation)
template <int NTILE>
__device__ __forceinline__ void gemm_cols(f32x16 (&acc)[NTILE], const LayerDev& Ly, const float* Z, const Lane& L) {
  const int in = Ly.in_dim, out = Ly.out_dim, ld = Ly.ld;
  const float* zcol = Z + (4 * L.lh) * kTP + L.ln;
  const float* wcol[NTILE];
  bool on[NTILE];
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int kt = L.wave + kWaves * jt;
    on[jt] = kt * 32 < in;
    wcol[jt] = Ly.W + (long long)(4 * L.lh) * ld + (on[jt] ? kt : 0) * 32 + L.ln;
  }
  if (!on[0]) return;
  float wc[NTILE][4], wn[NTILE][4], bc[4], bn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) wc[jt][i] = wcol[jt][(long long)i * ld];
    bc[i] = zcol[i * kTP];
  }
  for (int g = 0; g < out; g += 8) {
    const int gn = g + 8 < out ? g + 8 : g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) wn[jt][i] = wcol[jt][(long long)(gn + i) * ld];
      bn[i] = zcol[(gn + i) * kTP];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
        if (on[jt]) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[jt][i], bc[i], acc[jt], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) wc[jt][i] = wn[jt][i];
      bc[i] = bn[i];
    }
  }
}

// ---------------------------------------------------------------------------
// Register-resident weight fragments (widths <= 128): a wave's slice of one layer's weights is 32 rows
// (or columns) x <= 128 = 64 VGPRs per lane.  It is loaded ONCE per layer and tile and reused by all K
// stream steps, so the stream loop issues no global loads at all (L2 latency is paid once, ahead of the
// activation phase, instead of once per k-group).
// ---------------------------------------------------------------------------
constexpr int kMaxG = 16;  // k-groups of 8 input features

struct WFrag {
  f32x4 g[kMaxG];
};

// lane (j = ln, h) <- W[32 ft + j][8 g + 4 h .. + 3]
__device__ __forceinline__ void load_wrows(WFrag& wf, const LayerDev& Ly, int ft, const Lane& L) {
  const bool on = ft * 32 < Ly.out_dim;
  const float* wrow = Ly.W + (long long)((on ? ft : 0) * 32 + L.ln) * Ly.ld + 4 * L.lh;
#pragma unroll
  for (int g = 0; g < kMaxG; ++g)
    if (g * 8 < Ly.in_dim) wf.g[g] = *reinterpret_cast<const f32x4*>(wrow + 8 * g);
}

// lane (k = ln, h) <- W[8 g + 4 h + i][32 kt + k], i = 0..3
__device__ __forceinline__ void load_wcols(WFrag& wf, const LayerDev& Ly, int kt, const Lane& L) {
  const bool on = kt * 32 < Ly.in_dim;
  const float* wcol = Ly.W + (long long)(4 * L.lh) * Ly.ld + (on ? kt : 0) * 32 + L.ln;
#pragma unroll
  for (int g = 0; g < kMaxG; ++g) {
    if (g * 8 < Ly.out_dim) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wf.g[g][i] = wcol[(long long)(8 * g + i) * Ly.ld];
    }
  }
}

// acc += (fragment of depth 8*NG) . S[:][n]  — used for both W a (depth = in_dim) and W^T zbar (depth = out_dim).
// Straight-line code: all B-operand reads of a half are in flight before the first MFMA needs one, so the LDS
// latency is paid once per half instead of once per MFMA pair (run-time guards inside this loop cost 2x).
template <int NG>
__device__ __forceinline__ void gemm_frag_n(f32x16& acc, const WFrag& wf, const float* S, const Lane& L) {
  const float* col = S + (4 * L.lh) * kTP + L.ln;
  // B operands run two k-groups (8 ds_read_b32) ahead of the 4 MFMAs that consume them; sched_barrier pins that
  // order (left alone, the scheduler sinks each read to just before its MFMA and every MFMA waits out the LDS latency)
  float b[3][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) b[g][i] = (g < NG) ? col[(8 * g + i) * kTP] : 0.0f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 2 < NG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b[(g + 2) % 3][i] = col[(8 * (g + 2) + i) * kTP];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.g[g][i], b[g % 3][i], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void gemm_frag(f32x16& acc, const WFrag& wf, int depth, const float* S, const Lane& L) {
  switch (depth >> 3) {  // wave-uniform
    case 16: gemm_frag_n<16>(acc, wf, S, L); break;
    case 12: gemm_frag_n<12>(acc, wf, S, L); break;
    case 8: gemm_frag_n<8>(acc, wf, S, L); break;
    case 4: gemm_frag_n<4>(acc, wf, S, L); break;
    default: {
      const float* col = S + (4 * L.lh) * kTP + L.ln;
#pragma unroll
      for (int g = 0; g < kMaxG; ++g) {
        if (g * 8 < depth) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.g[g][i], col[(8 * g + i) * kTP], acc, 0, 0, 0);
        }
      }
    }
  }
}

// dacc[kt] += Z[own rows of tile ft][n] * A[rows of tile kt][n]^T  (one stream; weight gradient)
template <int NKT, int NA>  // NA = active k-tiles (in_dim / 32), compile-time so that the body is straight-line
__device__ __forceinline__ void gemm_outer_n(f32x16 (&dacc)[NKT], int ft, const float* Z, const float* A, const Lane& L) {
  const float* zrow = Z + (ft * 32 + L.ln) * kTP + 4 * L.lh;
  const float* arow = A + L.ln * kTP + 4 * L.lh;
  // operands of point group g+1 are requested before the 4 NA MFMAs of group g (pinned, see gemm_frag_n)
  f32x4 zc, zn, ac[NA], an[NA];
  zc = *reinterpret_cast<const f32x4*>(zrow);
#pragma unroll
  for (int kt = 0; kt < NA; ++kt) ac[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g + 1 < 4) {
      zn = *reinterpret_cast<const f32x4*>(zrow + 8 * (g + 1));
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) an[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP + 8 * (g + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kt = 0; kt < NA; ++kt)
        dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[i], ac[kt][i], dacc[kt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < 4) {
      zc = zn;
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) ac[kt] = an[kt];
    }
  }
}

template <int NKT>
__device__ __forceinline__ void gemm_outer(f32x16 (&dacc)[NKT], int ft, int in_dim, const float* Z, const float* A,
                                           const Lane& L) {
  const int na = (in_dim + 31) >> 5;  // wave-uniform
  if constexpr (NKT >= 4) {
    if (na == 4) { gemm_outer_n<NKT, 4>(dacc, ft, Z, A, L); return; }
    if (na == 3) { gemm_outer_n<NKT, 3>(dacc, ft, Z, A, L); return; }
  }
  if (na == 2) { gemm_outer_n<NKT, 2>(dacc, ft, Z, A, L); return; }
  if (na == 1) { gemm_outer_n<NKT, 1>(dacc, ft, Z, A, L); return; }
  const float* zrow = Z + (ft * 32 + L.ln) * kTP + 4 * L.lh;
  const float* arow = A + L.ln * kTP + 4 * L.lh;
#pragma unroll
  for (int g = 0; g < kT; g += 8) {
    const f32x4 zv = *reinterpret_cast<const f32x4*>(zrow + g);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt * 32 < in_dim) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP + g);
#pragma unroll
        for (int i = 0; i < 4; ++i) dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[i], av[i], dacc[kt], 0, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Encoding layer in registers: lane (n, h) of the wave owning tile ft evaluates its 16 features.
// ---------------------------------------------------------------------------
template <int NT, int NX>
__device__ __forceinline__ void enc_preact(const NetDev& net, const float* xin, int f, int n, float* z) {
  constexpr int K = 1 + NT + NX;
  const int din = net.din;
#pragma unroll
  for (int s = 0; s < K; ++s) z[s] = 0.0f;
  if (net.enc == ENC_FOURIER) {
    const int M = net.enc_out >> 1;
    const int m = f < M ? f : f - M;
    float v = 0.0f;
    for (int c = 0; c < din; ++c) v = fmaf(xin[c * kT + n], net.encW[c * M + m], v);
    z[0] = v;
    if constexpr (NT >= 1) z[1] = net.encW[(din - 1) * M + m];
    if constexpr (NX >= 1) z[1 + NT] = net.encW[m];
  } else {
    float v = net.encb[f];
    for (int c = 0; c < din; ++c) v = fmaf(xin[c * kT + n], net.encW[f * din + c], v);
    z[0] = v;
    if constexpr (NT >= 1) z[1] = net.encW[f * din + din - 1];
    if constexpr (NX >= 1) z[1 + NT] = net.encW[f * din];
  }
}

template <int NT, int NX>
__device__ __forceinline__ void encode_tile_fourier(const NetDev& net, const float* xin, f32x16 (&a)[1 + NT + NX], int ft,
                                                    const Lane& L) {
  constexpr int K = 1 + NT + NX;
  const int M = net.enc_out >> 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = ft * 32 + acc_row(r, L.lh);
    float z[K], y[K];
    enc_preact<NT, NX>(net, xin, f, L.ln, z);
    float sn, cs;
    fast_sincosf(z[0], &sn, &cs);
    // derivative ladder of sin: s, c, -s, -c, s, c ; of cos: c, -s, -c, s, c, -s
    const bool is_sin = f < M;
    float fd[6];
    fd[0] = is_sin ? sn : cs;
    fd[1] = is_sin ? cs : -sn;
    fd[2] = -fd[0];
    fd[3] = -fd[1];
    fd[4] = fd[0];
    fd[5] = fd[1];
    y[0] = fd[0];
    dir_fwd<NT>(fd, z + 1, y + 1);
    dir_fwd<NX>(fd, z + 1 + NT, y + 1 + NT);
#pragma unroll
    for (int s = 0; s < K; ++s) a[s][r] = y[s];
  }
}

template <int ACT, int NT, int NX>
__device__ __forceinline__ void encode_tile_linear(const NetDev& net, const float* xin, f32x16 (&a)[1 + NT + NX], int ft,
                                                   const Lane& L) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = ft * 32 + acc_row(r, L.lh);
    float z[K], y[K];
    enc_preact<NT, NX>(net, xin, f, L.ln, z);
    act_fwd<ACT, NT, NX>(net.enc_param, z, y);
#pragma unroll
    for (int s = 0; s < K; ++s) a[s][r] = y[s];
  }
}

template <int ACT, int NT, int NX, int NTILE>
__device__ __forceinline__ void encode_regs(const NetDev& net, const float* xin, f32x16 (&a)[NTILE][1 + NT + NX],
                                            const Lane& L) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int jt = 0; jt < NTILE; ++jt) {
    const int ft = L.wave + kWaves * jt;
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[jt][s][r] = 0.0f;
    if (ft * 32 < net.enc_out) {
      if (net.enc == ENC_FOURIER) {
        encode_tile_fourier<NT, NX>(net, xin, a[jt], ft, L);
      } else {
        encode_tile_linear<ACT, NT, NX>(net, xin, a[jt], ft, L);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Element-wise stages on one accumulator tile
// ---------------------------------------------------------------------------
// forward: v holds z (pre-activation jets) on entry, activation jets on exit; the tape gets what backward needs
template <int ACT, int NT, int NX, int NTILE, bool TAPE>
__device__ __forceinline__ void ew_forward(f32x16 (&v)[1 + NT + NX], float w, float* tape, int l, int jt, int tid) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 rec[K];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      float z[K], y[K];
#pragma unroll
      for (int s = 0; s < K; ++s) z[s] = v[s][r];
      act_fwd<ACT, NT, NX>(w, z, y);
      rec[0][i] = ActTape<ACT>::value_is_output ? y[0] : z[0];
#pragma unroll
      for (int s = 1; s < K; ++s) rec[s][i] = z[s];
#pragma unroll
      for (int s = 0; s < K; ++s) v[s][r] = y[s];
    }
    if constexpr (TAPE) {
#pragma unroll
      for (int s = 0; s < K; ++s) tape_st4(tape, l, jt, NTILE, K, s, q, tid, rec[s]);
    }
  }
}

// reverse S1: ab <- act_bwd(tape, ab)
template <int ACT, int NT, int NX, int NTILE>
__device__ __forceinline__ void ew_backward(f32x16 (&ab)[1 + NT + NX], float w, const float* tape, int l, int jt,
                                            int tid) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 rec[K];
#pragma unroll
    for (int s = 0; s < K; ++s) rec[s] = tape_ld4(tape, l, jt, NTILE, K, s, q, tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      float z[K], abv[K], zb[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        z[s] = rec[s][i];
        abv[s] = ab[s][r];
      }
      act_bwd_tape<ACT, NT, NX>(w, z, abv, zb);
#pragma unroll
      for (int s = 0; s < K; ++s) ab[s][r] = zb[s];
    }
  }
}

// reverse: a <- activation jets of layer l replayed from its tape
template <int ACT, int NT, int NX, int NTILE>
__device__ __forceinline__ void ew_replay(f32x16 (&a)[1 + NT + NX], float w, const float* tape, int l, int jt, int tid) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 rec[K];
#pragma unroll
    for (int s = 0; s < K; ++s) rec[s] = tape_ld4(tape, l, jt, NTILE, K, s, q, tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float z[K], y[K];
#pragma unroll
      for (int s = 0; s < K; ++s) z[s] = rec[s][i];
      act_fwd_tape<ACT, NT, NX>(w, z, y);
#pragma unroll
      for (int s = 0; s < K; ++s) a[s][4 * q + i] = y[s];
    }
  }
}

// reverse, first Linear (din -> H): recompute z from the coordinates, ab <- zbar
template <int ACT, int NT, int NX>
__device__ __forceinline__ void ew_enc_backward(f32x16 (&ab)[1 + NT + NX], const NetDev& net, const float* xin, int ft,
                                                const Lane& L) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = ft * 32 + acc_row(r, L.lh);
    float z[K], abv[K], zb[K];
    enc_preact<NT, NX>(net, xin, f, L.ln, z);
#pragma unroll
    for (int s = 0; s < K; ++s) abv[s] = ab[s][r];
    act_bwd<ACT, NT, NX>(net.enc_param, z, abv, zb);
#pragma unroll
    for (int s = 0; s < K; ++s) ab[s][r] = zb[s];
  }
}

// raw copy of one accumulator tile (all streams) to / from a tape slot
template <int K, int NTILE>
__device__ __forceinline__ void tape_put(const f32x16 (&v)[K], float* tape, int slot, int jt, int tid) {
#pragma unroll
  for (int s = 0; s < K; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w4 = {v[s][4 * q], v[s][4 * q + 1], v[s][4 * q + 2], v[s][4 * q + 3]};
      tape_st4(tape, slot, jt, NTILE, K, s, q, tid, w4);
    }
}
template <int K, int NTILE>
__device__ __forceinline__ void tape_get(f32x16 (&v)[K], const float* tape, int slot, int jt, int tid) {
#pragma unroll
  for (int s = 0; s < K; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w4 = tape_ld4(tape, slot, jt, NTILE, K, s, q, tid);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[s][4 * q + i] = w4[i];
    }
}

// sum over the 32 points of one LDS row (thread per feature)
__device__ __forceinline__ float row_sum(const float* row) {
  float g = 0.0f;
#pragma unroll
  for (int n = 0; n < kT; n += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + n);
    g += (v[0] + v[1]) + (v[2] + v[3]);
  }
  return g;
}

// ---------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------
template <int ACT, int NT, int NX, int NTILE, bool BWD, int OCC>
__global__ __launch_bounds__(kThreads, OCC) void jet_kernel(const KernelArgs a) {
  constexpr int K = 1 + NT + NX;
  constexpr int NKT = 4 * NTILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDev& net = a.net;
  const int hmax = net.hmax;
  const int sbuf = hmax * kTP;              // floats per stream buffer
  float* SB = smem;                         // (BWD ? 4 : 2) stream buffers
  float* RED = SB + (BWD ? 4 : 2) * sbuf;   // kWaves * K * kT  (cross-wave reduction of the output layer)
  float* U = RED + kWaves * K * kT;         // K * kT (output jets)
  float* UB = U + K * kT;                   // K * kT (their cotangents)
  float* xin = UB + K * kT;                 // kMaxDin * kT

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);  // wave-uniform => tile-ownership tests are scalar branches
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int tid = L.tid;
  const int din = net.din;
  const long long ntiles = (a.N + kT - 1) / kT;
  float* tape = BWD ? a.tape + (long long)blockIdx.x * a.tape_stride : nullptr;
  int c = 0;  // stream-step counter: buffer parity (a buffer is rewritten two steps after it was last read)
  PINN_STAMP_DECL

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long p0 = tile * kT;
    __syncthreads();  // previous tile's readers of xin / U / UB / stream buffers are done
    if (tid < kT) {
      const long long p = p0 + tid;
      const bool ok = p < a.N;
      for (int cc = 0; cc < din - 1; ++cc) xin[cc * kT + tid] = ok ? a.x[p * (din - 1) + cc] : 0.0f;
      xin[(din - 1) * kT + tid] = ok ? a.t[p] : 0.0f;
    }
    __syncthreads();

    PINN_STAMP(ST_STAGE);
    f32x16 v[NTILE][K];  // this wave's activations (all streams) in accumulator layout
    WFrag wf;            // this wave's weight slice of the current layer (NTILE == 1 only)
    if constexpr (NTILE == 1) {
      if (net.n_layers > 0) load_wrows(wf, uniform_layer(net.layer[0]), L.wave, L);  // latency hides under the encoding
    }
    encode_regs<ACT, NT, NX, NTILE>(net, xin, v, L);
    if constexpr (BWD) {  // the reverse sweep re-reads the encoding's jets instead of re-evaluating sin/cos
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt)
        if ((L.wave + kWaves * jt) * 32 < net.enc_out) tape_put<K, NTILE>(v[jt], tape, net.n_layers, jt, tid);
    }
    PINN_STAMP(ST_ENCODE);

    // ---- hidden layers ----
    for (int l = 0; l < net.n_layers; ++l) {
      const LayerDev Ly = uniform_layer(net.layer[l]);
      f32x16 acc[NTILE][K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        float* S = SB + (c & 1) * sbuf;
        {
          f32x16 tmp[NTILE];
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) tmp[jt] = v[jt][s];
          stage_one<NTILE>(tmp, S, Ly.in_dim, L);
        }
        __syncthreads();
        f32x16 accs[NTILE];
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = L.wave + kWaves * jt;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[jt][r] = 0.0f;
          if (s == 0 && ft * 32 < Ly.out_dim) {  // the value stream starts from the bias (rows 8q+4h .. +3 are contiguous)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 bq = *reinterpret_cast<const f32x4*>(Ly.b + ft * 32 + 8 * q + 4 * L.lh);
#pragma unroll
              for (int i = 0; i < 4; ++i) accs[jt][4 * q + i] = bq[i];
            }
          }
        }
        if constexpr (NTILE == 1) {
          if (L.wave * 32 < Ly.out_dim) gemm_frag(accs[0], wf, Ly.in_dim, S, L);
        } else {
          gemm_rows<NTILE>(accs, Ly, S, L);
        }
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) acc[jt][s] = accs[jt];
        ++c;
      }
      PINN_STAMP(ST_FWD_GEMM);
      if constexpr (NTILE == 1) {
        if (l + 1 < net.n_layers) load_wrows(wf, uniform_layer(net.layer[l + 1]), L.wave, L);  // hides under ew_forward
      }
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int ft = L.wave + kWaves * jt;
        if (ft * 32 < Ly.out_dim) {
          ew_forward<ACT, NT, NX, NTILE, BWD>(acc[jt], Ly.act_param, tape, l, jt, tid);
        }
#pragma unroll
        for (int s = 0; s < K; ++s) v[jt][s] = acc[jt][s];
      }
      PINN_STAMP(ST_FWD_EW);
    }

    // ---- output layer (H_last -> 1): per-lane partial dot over own features, then across halves and waves ----
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

    PINN_STAMP(ST_OUT);
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
      // ---- B0: output layer.  dw_out[f] = sum_{s,n} ub_s[n] a_s[f][n];  abar = w_out (x) ub ----
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
      PINN_STAMP(ST_B0);

      for (int l = net.n_layers - 1; l >= 0; --l) {
        const LayerDev Ly = uniform_layer(net.layer[l]);
        const bool need_abar = l > 0 || net.enc == ENC_LINEAR;
        if constexpr (NTILE == 1) {
          if (need_abar) load_wcols(wf, Ly, L.wave, L);  // W^T slice for delta-propagation; hides under the jets below
        }
        // zbar = act_bwd(tape_l, abar) in place;  ap = a_{l-1} replayed from tape_{l-1} (or the encoding)
        f32x16 ap[NTILE][K];
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = L.wave + kWaves * jt;
          if (ft * 32 < Ly.out_dim) {
            ew_backward<ACT, NT, NX, NTILE>(ab[jt], Ly.act_param, tape, l, jt, tid);
          }
        }
        if (l > 0) {
          const LayerDev P = uniform_layer(net.layer[l - 1]);
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
            const int ft = L.wave + kWaves * jt;
#pragma unroll
            for (int s = 0; s < K; ++s)
#pragma unroll
              for (int r = 0; r < 16; ++r) ap[jt][s][r] = 0.0f;
            if (ft * 32 < P.out_dim) {
              ew_replay<ACT, NT, NX, NTILE>(ap[jt], P.act_param, tape, l - 1, jt, tid);
            }
          }
        } else {
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
#pragma unroll
            for (int s = 0; s < K; ++s)
#pragma unroll
              for (int r = 0; r < 16; ++r) ap[jt][s][r] = 0.0f;
            if ((L.wave + kWaves * jt) * 32 < net.enc_out) tape_get<K, NTILE>(ap[jt], tape, net.n_layers, jt, tid);
          }
        }
        PINN_STAMP(ST_BWD_EW);
        f32x16 abn[NTILE][K];
        f32x16 dacc[NTILE][NKT];
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[jt][kt][r] = 0.0f;
        // stream loop: publish zbar_s and a_{l-1,s}; dW += Z A^T; abar_{l-1,s} = W^T zbar_s
#pragma unroll
        for (int s = 0; s < K; ++s) {
          float* Z = SB + (c & 1) * sbuf;
          float* A2 = SB + (2 + (c & 1)) * sbuf;
          {
            f32x16 tz[NTILE], ta[NTILE];
#pragma unroll
            for (int jt = 0; jt < NTILE; ++jt) {
              tz[jt] = ab[jt][s];
              ta[jt] = ap[jt][s];
            }
            stage_one<NTILE>(tz, Z, Ly.out_dim, L);
            stage_one<NTILE>(ta, A2, Ly.in_dim, L);
          }
          __syncthreads();
          if (s == 0 && Ly.db && tid < Ly.out_dim) atomicAdd(Ly.db + tid, row_sum(Z + tid * kTP));
          if (Ly.dW) {
#pragma unroll
            for (int jt = 0; jt < NTILE; ++jt) {
              const int ft = L.wave + kWaves * jt;
              if (ft * 32 < Ly.out_dim) gemm_outer<NKT>(dacc[jt], ft, Ly.in_dim, Z, A2, L);
            }
          }
          if (need_abar) {
            f32x16 accs[NTILE];
#pragma unroll
            for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
              for (int r = 0; r < 16; ++r) accs[jt][r] = 0.0f;
            if constexpr (NTILE == 1) {
              if (L.wave * 32 < Ly.in_dim) gemm_frag(accs[0], wf, Ly.out_dim, Z, L);
            } else {
              gemm_cols<NTILE>(accs, Ly, Z, L);
            }
#pragma unroll
            for (int jt = 0; jt < NTILE; ++jt) abn[jt][s] = accs[jt];
          }
          ++c;
        }
        PINN_STAMP(ST_BWD_STREAM);
        // dW tile rows: float atomics, a wave's footprint is 2 x 128-byte rows per instruction.  (Measured: draining
        // them behind later MFMA work does not help — atomics share the in-order vmcnt queue with the tape / weight
        // loads that follow; the real fix is fewer atomics, i.e. the wide kernel's persistent accumulators.)
        if (Ly.dW) {
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt) {
            const int ft = L.wave + kWaves * jt;
            if (ft * 32 < Ly.out_dim) {
#pragma unroll
              for (int kt = 0; kt < NKT; ++kt) {
                if (kt * 32 < Ly.in_dim) {
#pragma unroll
                  for (int r = 0; r < 16; ++r)
                    if (kt * 32 + L.ln < Ly.in_dim)  // in_dim may end inside a k-tile (e.g. 24 Fourier features)
                      atomicAdd(Ly.dW + (long long)(ft * 32 + acc_row(r, L.lh)) * Ly.ld + kt * 32 + L.ln,
                                dacc[jt][kt][r]);
                }
              }
            }
          }
        }
        if (need_abar) {
#pragma unroll
          for (int jt = 0; jt < NTILE; ++jt)
#pragma unroll
            for (int s = 0; s < K; ++s) ab[jt][s] = abn[jt][s];
        }
        PINN_STAMP(ST_BWD_FLUSH);
      }

      // ---- encoding backward (first Linear of feedforward / SIREN); the Fourier matrix B is a buffer ----
      if (net.enc == ENC_LINEAR && net.d_encW) {
        const int H = net.enc_out;
#pragma unroll
        for (int jt = 0; jt < NTILE; ++jt) {
          const int ft = L.wave + kWaves * jt;
          if (ft * 32 < H) {
            ew_enc_backward<ACT, NT, NX>(ab[jt], net, xin, ft, L);
          }
        }
        // three row sums per feature: zbar_value (weighted by the coordinates), zbar_t, zbar_x
        float* S0 = SB + 0 * sbuf;
        float* S1 = SB + 1 * sbuf;
        float* S2 = SB + 2 * sbuf;
        __syncthreads();  // readers of the last stream step are done
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
      PINN_STAMP(ST_ENC_BWD);
    }
  }
#ifdef PINN_STAMPS
  if (a.stamps && (tid & 63) == 0) {
    st_acc[ST_TOTAL] = pinn_now() - st_begin;
    for (int i = 0; i < kNumStamps; ++i) a.stamps[((long long)blockIdx.x * kWaves + L.wave) * kNumStamps + i] = st_acc[i];
  }
#endif
}

// ---------------------------------------------------------------------------
// Host-side launch helper, one instantiation per (NT, NX) lives in its own translation unit
// ---------------------------------------------------------------------------
inline size_t jet_lds_bytes(int K, int hmax, bool bwd) {
  return sizeof(float) * ((size_t)(bwd ? 4 : 2) * hmax * kTP + (size_t)kWaves * K * kT + 2 * K * kT + kMaxDin * kT);
}

inline long long jet_tape_floats_per_wg(int K, int n_layers, int ntile) {
  return (long long)(n_layers + 1) * ntile * K * 16 * kThreads;  // + 1: the encoding's output jets (slot n_layers)
}

// occ = workgroups per CU the kernel is register-budgeted for (2 => <= 256 VGPR+AGPR per lane)
template <int NT, int NX>
hipError_t launch_jet(const KernelArgs& a, bool bwd, int grid, int occ, hipStream_t stream) {
  constexpr int K = 1 + NT + NX;
  (void)occ;  // the launch is sized by the caller; the register budget follows from (NTILE, BWD) below
  const int ntile = a.net.hmax > 128 ? 2 : 1;
  const size_t lds = jet_lds_bytes(K, a.net.hmax, bwd);
  hipError_t e = hipSuccess;
#define PINN_LAUNCH1(ACT_, NTILE_, BWD_, OCC_)                                                               \
  do {                                                                                                       \
    auto kern = jet_kernel<ACT_, NT, NX, NTILE_, BWD_, OCC_>;                                                          \
    e = allow_full_lds(reinterpret_cast<const void*>(kern));                                                 \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, a);                                    \
  } while (0)
#ifdef PINN_DEV /* fast-compiling developer build: ONE activation (default tanh) and ONE tile count (default 1) */
#ifndef PINN_DEV_ACT
#define PINN_DEV_ACT PINN_ACT_TANH
#endif
#ifndef PINN_DEV_NTILE
#define PINN_DEV_NTILE 1
#endif
#define PINN_LAUNCH(NTILE_, BWD_, OCC_)                                                        \
  if constexpr (NTILE_ == PINN_DEV_NTILE) {                                                    \
    if (act == PINN_DEV_ACT) PINN_LAUNCH1(PINN_DEV_ACT, NTILE_, BWD_, OCC_); else return hipErrorInvalidValue; \
  } else return hipErrorInvalidValue;
#else
#define PINN_LAUNCH(NTILE_, BWD_, OCC_)                                                        \
  switch (act) {                                                                               \
    case PINN_ACT_TANH: PINN_LAUNCH1(PINN_ACT_TANH, NTILE_, BWD_, OCC_); break;                \
    case PINN_ACT_SIN: PINN_LAUNCH1(PINN_ACT_SIN, NTILE_, BWD_, OCC_); break;                  \
    case PINN_ACT_GELU: PINN_LAUNCH1(PINN_ACT_GELU, NTILE_, BWD_, OCC_); break;                \
    case PINN_ACT_SIGMOID: PINN_LAUNCH1(PINN_ACT_SIGMOID, NTILE_, BWD_, OCC_); break;          \
    default: PINN_LAUNCH1(PINN_ACT_RELU, NTILE_, BWD_, OCC_); break; /* piecewise linear */    \
  }
#endif
  // every hidden layer of the supported architectures shares one activation (ENC_LINEAR's included)
  const int act = a.net.n_layers > 0 ? a.net.layer[0].act : a.net.enc_act;
  if (ntile == 1) {
    if (bwd) PINN_LAUNCH(1, true, 1)  // one workgroup per CU: the 256-register budget of two spills ~2000 VGPRs
    else PINN_LAUNCH(1, false, 2)
  } else {
    if (bwd) PINN_LAUNCH(2, true, 1) else PINN_LAUNCH(2, false, 1)
  }
#undef PINN_LAUNCH
#undef PINN_LAUNCH1
  return hipGetLastError();
}

}  // namespace pinn
