// Instantiations of the fused tile-major kernel (jet_kernel_wide.h).  The Makefile compiles this file once per
// (time_order, space_order) stream set AND activation family — -DPINN_NT=.. -DPINN_NX=.. -DPINN_WIDE_ACT=<0..4> — so
// that the 40 units build in parallel and each one can be compiled with -mllvm -amdgpu-mfma-vgpr-form=1 on its own:
// with the default AGPR-form MFMAs the kernel's persistent accumulator tiles plus the activation phases' VGPRs spill
// (~150 VGPRs, and every scratch reload queues behind the in-flight tape loads); in VGPR form it spills nothing,
// but the option crashes this LLVM on some units, which are then rebuilt in the default form (build/*.fallback,
// pinn_build_info()).
//   -DPINN_WIDE_ACT=a   launch_jetw_<NT>_<NX>_a<a>; the unit of activation 0 also holds the dispatcher launch_jetw_<NT>_<NX>
//   (none, make dev)    launch_jetw_<NT>_<NX> with every activation the build flags leave in (PINN_DEV: tanh only)
#include "jet_kernel_wide.h"

#ifndef PINN_NT
#error "compile with -DPINN_NT=<0..2> -DPINN_NX=<0..4>"
#endif

#define PINN_CAT2(a, b, c) a##b##_##c
#define PINN_CAT(a, b, c) PINN_CAT2(a, b, c)

namespace pinn {
#ifdef PINN_WIDE_ACT
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
#else
hipError_t PINN_CAT(launch_jetw_, PINN_NT, PINN_NX)(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  return launch_jet_wide<PINN_NT, PINN_NX>(a, bwd, grid, stream);
}
#endif
}  // namespace pinn
