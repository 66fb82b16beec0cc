// Shared device definitions of the jet kernels (gfx950): network / layer descriptors as the kernels see them, the
// v_mfma_f32_32x32x2_f32 accumulator layout (32 feature rows in 16 registers x 2 lane halves, 32 point columns on
// lanes: exactly what the per-element activation jets need, so nothing is transposed between a GEMM and its
// activation), the 16-byte-word tape layout of the fused tile-major kernel (jet_kernel_wide.h) and the in-kernel
// phase stamps of the diagnostic build.  Both engines include this file: the fused kernel for plain MLPs of width
// <= 128, and the layer-major engine (lm_*.h) for everything else.
//
// Algorithmic FLOPs per point: K * 2 * sum(in*out) forward, 3x that with the reverse sweep.
// tests/jet_model.py is the executable specification of the arithmetic.
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
  long long det_stride;                     // deterministic mode: the gradient / loss pointers above point into row 0 of a
                                            // [grid][det_stride] slab and workgroup b adds (plainly) into row b; 0 = atomics
  int det_mask;                             // two-level flush (det_stride < 0): rows - 1, a power of two minus one
  int flush_store;                          // det_stride > 0 only: the end-of-kernel flush writes this workgroup's row with plain
                                            // STORES (every address exactly once; no memset of the slab beforehand)
};

// Gradient / loss accumulation at the end of a workgroup (or per tile for layers beyond the persistent ones): float
// atomics on the caller's tensors, or - deterministic mode - on this workgroup's own row of a [grid][det_stride] slab
// (the pointers then point into row 0): nobody else touches that row, every address gets its adds in program order
// (at most two per flush, which commute exactly from a zero start), and a fixed-order reduction sums the rows
// afterwards (pinn_abi.hip).  One code path: the row offset is simply 0 when the mode is off.
__device__ __forceinline__ long long det_row_offset(const KernelArgs& a) {
  // det_stride > 0: one row per workgroup (deterministic mode); < 0: det_mask + 1 shared rows of -det_stride floats
  // (two-level flush: fewer workgroups contend for an address); 0: the caller's tensors directly
  return a.det_stride >= 0 ? (long long)blockIdx.x * a.det_stride : (long long)(blockIdx.x & a.det_mask) * -a.det_stride;
}
__device__ __forceinline__ void grad_add(float* p, float v, long long off) { atomicAdd(p + off, v); }
// end-of-kernel flush: `store` (uniform) = this workgroup owns the row and writes each address once
__device__ __forceinline__ void grad_put(float* p, float v, long long off, bool store) {
  if (store) p[off] = v;
  else atomicAdd(p + off, v);
}

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


inline long long jet_tape_floats_per_wg(int K, int n_layers, int ntile) {
  return (long long)(n_layers + 1) * ntile * K * 16 * kThreads;  // + 1: the encoding's output jets (slot n_layers)
}


}  // namespace pinn
