// One stream set (PINN_NT, PINN_NX) of the layer-major engine's element-wise and head kernels.
#include <cstdlib>

#include "lm_engine.h"

#ifndef PINN_NT
#error "compile with -DPINN_NT=<time order> -DPINN_NX=<space order>"
#endif

namespace pinn {
namespace lm {

#define PINN_CAT2(a, b, c, d) a##b##c##_##d
#define PINN_NAME(a, b, c) PINN_CAT2(launch_lm_, a, b, c)

static bool ew_dma_on() {  // PINN_LM_EWDMA=0: the register-staged LayerNorm adjoint everywhere (experiments), read once
  static const bool v = [] {
    const char* e = getenv("PINN_LM_EWDMA");
    return !(e && atoi(e) == 0);
  }();
  return v;
}

template <int ACT, int FPT, bool LN>
static hipError_t launch_ew(const EwArgs& a, bool bwd, int grid, hipStream_t st) {
  const int threads = kPT * a.G;
  if constexpr (LN && FPT == 4) {
    constexpr int K = 1 + PINN_NT + PINN_NX;
    if (bwd && ew_dma_on() && lm_ew_bwd_dma_ok(a, K)) {  // next unit prefetched by LDS-DMA (lm_ew.h)
      auto kern = lm_ew_bwd_dma<ACT, PINN_NT, PINN_NX, FPT>;
      const hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lm_ew_bwd_dma_lds_bytes(K, a.Hp), st, a);
      return hipGetLastError();
    }
  }
  if (bwd) hipLaunchKernelGGL((lm_ew_bwd<ACT, PINN_NT, PINN_NX, FPT, LN>), dim3(grid), dim3(threads), 0, st, a);
  else hipLaunchKernelGGL((lm_ew_fwd<ACT, PINN_NT, PINN_NX, FPT, LN>), dim3(grid), dim3(threads), 0, st, a);
  return hipGetLastError();
}

template <int ACT, int FPT>
static hipError_t launch_ew_ln(const EwArgs& a, bool bwd, int grid, hipStream_t st) {
  return a.ln_g ? launch_ew<ACT, FPT, true>(a, bwd, grid, st) : launch_ew<ACT, FPT, false>(a, bwd, grid, st);
}

template <int ACT>
static hipError_t launch_ew_fpt(const EwArgs& a, bool bwd, int fpt, int grid, hipStream_t st) {
  switch (fpt) {
    case 1: return launch_ew_ln<ACT, 1>(a, bwd, grid, st);
    case 4: return launch_ew_ln<ACT, 4>(a, bwd, grid, st);
    case 8: return launch_ew_ln<ACT, 8>(a, bwd, grid, st);
    case 16: return launch_ew_ln<ACT, 16>(a, bwd, grid, st);
    default: return hipErrorInvalidValue;
  }
}

static hipError_t launch_fourier(const EwArgs& a, int fpt, int grid, hipStream_t st) {
  const int threads = kPT * a.G;
  switch (fpt) {
    case 1: hipLaunchKernelGGL((lm_fourier_fwd<PINN_NT, PINN_NX, 1>), dim3(grid), dim3(threads), 0, st, a); break;
    case 4: hipLaunchKernelGGL((lm_fourier_fwd<PINN_NT, PINN_NX, 4>), dim3(grid), dim3(threads), 0, st, a); break;
    case 8: hipLaunchKernelGGL((lm_fourier_fwd<PINN_NT, PINN_NX, 8>), dim3(grid), dim3(threads), 0, st, a); break;
    case 16: hipLaunchKernelGGL((lm_fourier_fwd<PINN_NT, PINN_NX, 16>), dim3(grid), dim3(threads), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// act: PinnAct of the prologue, -1 = no activation, -2 = Fourier features of the coordinates (forward only)
hipError_t PINN_NAME(ew_, PINN_NT, PINN_NX)(const EwArgs& a, bool bwd, int act, int fpt, int grid, hipStream_t st) {
  if (act == -2) return bwd ? hipErrorInvalidValue : launch_fourier(a, fpt, grid, st);
  switch (act) {
    case PINN_ACT_TANH: return launch_ew_fpt<PINN_ACT_TANH>(a, bwd, fpt, grid, st);
    case PINN_ACT_SIN: return launch_ew_fpt<PINN_ACT_SIN>(a, bwd, fpt, grid, st);
    case PINN_ACT_GELU: return launch_ew_fpt<PINN_ACT_GELU>(a, bwd, fpt, grid, st);
    case PINN_ACT_SIGMOID: return launch_ew_fpt<PINN_ACT_SIGMOID>(a, bwd, fpt, grid, st);
    default: return launch_ew_fpt<PINN_ACT_RELU>(a, bwd, fpt, grid, st);  // piecewise linear (slope in act_param), also "no activation"
  }
}

hipError_t PINN_NAME(head_, PINN_NT, PINN_NX)(const HeadArgs& a, int fpt, int grid, hipStream_t st) {
  const int threads = kPT * a.G;
  switch (fpt) {
    case 1: hipLaunchKernelGGL((lm_head<PINN_NT, PINN_NX, 1>), dim3(grid), dim3(threads), 0, st, a); break;
    case 4: hipLaunchKernelGGL((lm_head<PINN_NT, PINN_NX, 4>), dim3(grid), dim3(threads), 0, st, a); break;
    case 8: hipLaunchKernelGGL((lm_head<PINN_NT, PINN_NX, 8>), dim3(grid), dim3(threads), 0, st, a); break;
    case 16: hipLaunchKernelGGL((lm_head<PINN_NT, PINN_NX, 16>), dim3(grid), dim3(threads), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace lm
}  // namespace pinn
