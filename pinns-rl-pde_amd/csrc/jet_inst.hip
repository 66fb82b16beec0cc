// One translation unit per (time_order, space_order) stream set and kernel group; the Makefile compiles this file
// several times with -DPINN_NT=.. -DPINN_NX=.. [-DPINN_TU_WIDE | -DPINN_TU_NOWIDE] so that the instantiations build in
// parallel.  The wide kernel gets its own objects because it is compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (see the
// Makefile): with the default AGPR-form MFMAs its persistent accumulator tiles plus the activation phases' VGPRs
// spill (~150 VGPRs, and every scratch reload queues behind the in-flight tape loads); in VGPR form it spills nothing.
//   PINN_TU_WIDE    only the wide kernel, with -DPINN_WIDE_ACT=<0..4> one activation family per unit
//   PINN_TU_NOWIDE  launch_jet_ / launch_jetr_ / launch_jeta_ (stream-serial, ResNet, attention)
//   neither         everything (developer builds; PINN_DEV_WIDE / PINN_DEV_STREAM restrict those further)
#if defined(PINN_TU_WIDE) || defined(PINN_DEV_WIDE)
#define PINN_WITH_WIDE 1
#define PINN_WITH_REST 0
#elif defined(PINN_TU_NOWIDE) || defined(PINN_DEV_STREAM)
#define PINN_WITH_WIDE 0
#define PINN_WITH_REST 1
#else
#define PINN_WITH_WIDE 1
#define PINN_WITH_REST 1
#endif

#if PINN_WITH_REST
#ifdef PINN_DEV_STREAM
#include "jet_kernel.h"
#else
#include "jet_kernel_attn.h"
#endif
#endif
#if PINN_WITH_WIDE
#include "jet_kernel_wide.h"
#endif

#ifndef PINN_NT
#error "compile with -DPINN_NT=<0..2> -DPINN_NX=<0..4>"
#endif

#define PINN_CAT2(a, b, c) a##b##_##c
#define PINN_CAT(a, b, c) PINN_CAT2(a, b, c)

namespace pinn {
#if PINN_WITH_REST
// stream-serial kernel: any K, widths up to 256
hipError_t PINN_CAT(launch_jet_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, int occ, hipStream_t stream) {
  return launch_jet<PINN_NT, PINN_NX>(a, bwd, grid, occ, stream);
}
#ifndef PINN_DEV_STREAM
// ResNet kernel (LayerNorm jets; derivative orders <= 2)
hipError_t PINN_CAT(launch_jetr_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_resnet<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
// attention-as-MLP kernel (LayerNorm + chunked 4x feed-forward)
hipError_t PINN_CAT(launch_jeta_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_attn<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
#else
hipError_t PINN_CAT(launch_jetr_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jeta_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
#endif
#elif defined(PINN_DEV_WIDE) /* make dev WIDE=1: only the wide kernel, stubs for the rest */
hipError_t PINN_CAT(launch_jet_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jetr_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t PINN_CAT(launch_jeta_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
#endif

#if PINN_WITH_WIDE && defined(PINN_WIDE_ACT)
// wide kernel, ONE activation family per translation unit: launch_jetw_<NT>_<NX>_a<ACT>; the unit of activation 0
// also holds the dispatcher launch_jetw_<NT>_<NX>
#define PINN_CAT4(a, b, c, d) a##b##_##c##_a##d
#define PINN_CATA(a, b, c, d) PINN_CAT4(a, b, c, d)
hipError_t PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, PINN_WIDE_ACT)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_wide_act<PINN_WIDE_ACT, PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
#if PINN_WIDE_ACT == 0
hipError_t PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 1)(const KernelArgs&, bool, int, hipStream_t);
hipError_t PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 2)(const KernelArgs&, bool, int, hipStream_t);
hipError_t PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 3)(const KernelArgs&, bool, int, hipStream_t);
hipError_t PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 4)(const KernelArgs&, bool, int, hipStream_t);
hipError_t PINN_CAT(launch_jetw_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  switch (jet_wide_act_family(a)) {
    case 0: return PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 0)(a, bwd, grid, stream);
    case 1: return PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 1)(a, bwd, grid, stream);
    case 2: return PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 2)(a, bwd, grid, stream);
    case 3: return PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 3)(a, bwd, grid, stream);
    default: return PINN_CATA(launch_jetw_, PINN_NT, PINN_NX, 4)(a, bwd, grid, stream);
  }
}
#endif
#elif PINN_WITH_WIDE
// wide kernel: all K streams LDS-resident, persistent dW accumulators (K * Hmax small enough)
hipError_t PINN_CAT(launch_jetw_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_wide<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
#elif defined(PINN_DEV_STREAM) /* make dev STREAM=1: only the stream-serial kernel */
hipError_t PINN_CAT(launch_jetw_, PINN_NT, PINN_NX)(const KernelArgs&, bool, int, hipStream_t) { return hipErrorInvalidValue; }
#endif
}  // namespace pinn
