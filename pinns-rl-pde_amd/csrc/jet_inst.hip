// One translation unit per (time_order, space_order) stream set; the Makefile compiles this file
// several times with -DPINN_NT=.. -DPINN_NX=.. so the instantiations build in parallel.
#if defined(PINN_DEV_STREAM)
#include "jet_kernel.h"
#elif !defined(PINN_DEV_WIDE)
#include "jet_kernel_attn.h"
#endif
#ifndef PINN_DEV_STREAM
#include "jet_kernel_wide.h"
#endif

#ifndef PINN_NT
#error "compile with -DPINN_NT=<0..2> -DPINN_NX=<0..4>"
#endif

#define PINN_CAT2(a, b, c) a##b##_##c
#define PINN_CAT(a, b, c) PINN_CAT2(a, b, c)

namespace pinn {
#ifdef PINN_DEV_WIDE /* make dev WIDE=1: only the wide kernel, for quick iteration on it */
hipError_t PINN_CAT(launch_jet_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jetr_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jeta_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
#else
// stream-serial kernel: any K, widths up to 256
hipError_t PINN_CAT(launch_jet_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, int occ, hipStream_t stream) {
  return launch_jet<PINN_NT, PINN_NX>(a, bwd, grid, occ, stream);
}
#endif
#ifdef PINN_DEV_STREAM /* make dev STREAM=1: only the stream-serial kernel */
hipError_t PINN_CAT(launch_jetw_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jetr_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jeta_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
#else
// wide kernel: all K streams LDS-resident, persistent dW accumulators (K * Hmax small enough)
hipError_t PINN_CAT(launch_jetw_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_wide<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
#endif
#if !defined(PINN_DEV_WIDE) && !defined(PINN_DEV_STREAM)
// ResNet kernel (LayerNorm jets; derivative orders <= 2)
hipError_t PINN_CAT(launch_jetr_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_resnet<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
// attention-as-MLP kernel (LayerNorm + chunked 4x feed-forward)
hipError_t PINN_CAT(launch_jeta_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_attn<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
#endif
}  // namespace pinn
