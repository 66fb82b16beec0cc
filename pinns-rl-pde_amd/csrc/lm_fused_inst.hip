// One (stream set, activation family) unit of the fused GEMM + prologue kernels (lm_fused.h):
//   -DPINN_NT=<time order> -DPINN_NX=<space order> -DPINN_FACT=<PinnAct family: 0 tanh, 1 sin, 2 gelu, 3 sigmoid, 4 relu>
#include "lm_fused.h"

#ifndef PINN_FACT
#error "compile with -DPINN_NT= -DPINN_NX= -DPINN_FACT="
#endif

namespace pinn {
namespace lm {

#define PINN_FCAT(a, b, c) launch_lm_fused_##a##_##b##_##c
#define PINN_FNAME(a, b, c) PINN_FCAT(a, b, c)

template <int NCH, int RT, bool LN, bool BWD, bool AUX>
static hipError_t launch_aux(const FusedArgs& a, int gx, int gy, hipStream_t st) {
  auto kern = lm_fused<NCH, RT, PINN_NT, PINN_NX, PINN_FACT, LN, BWD, AUX>;
  const hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(kFThreads), lm_fused_lds_bytes(NCH, RT, 1 + PINN_NT + PINN_NX, BWD, AUX), st, a);
  return hipGetLastError();
}

// AUX: forward — a skip or an add record (never both: lm_engine.hip::plan_fusion); reverse — an add record
template <int NCH, int RT, bool LN, bool BWD>
static hipError_t launch_one(const FusedArgs& a, int gx, int gy, hipStream_t st) {
  const bool aux = BWD ? a.add0 != nullptr : (a.add0 != nullptr || a.skip != nullptr);
  return aux ? launch_aux<NCH, RT, LN, BWD, true>(a, gx, gy, st) : launch_aux<NCH, RT, LN, BWD, false>(a, gx, gy, st);
}

template <int NCH, int RT>
static hipError_t launch_shape(const FusedArgs& a, bool bwd, bool ln, int gx, int gy, hipStream_t st) {
  if (ln) return bwd ? launch_one<NCH, RT, true, true>(a, gx, gy, st) : launch_one<NCH, RT, true, false>(a, gx, gy, st);
  return bwd ? launch_one<NCH, RT, false, true>(a, gx, gy, st) : launch_one<NCH, RT, false, false>(a, gx, gy, st);
}

// shapes: (nch, rt) = (8, 8) depth 256 x 256-row blocks, (4, 8) depth 128 x 256-row blocks, (4, 4) depth 128 x 128 rows
hipError_t PINN_FNAME(PINN_NT, PINN_NX, PINN_FACT)(const FusedArgs& a, bool bwd, bool ln, int nch, int rt, int gx, int gy,
                                                   hipStream_t st) {
  if (nch == 8 && rt == 8) return launch_shape<8, 8>(a, bwd, ln, gx, gy, st);
  if (nch == 4 && rt == 8) return launch_shape<4, 8>(a, bwd, ln, gx, gy, st);
  if (nch == 4 && rt == 4) return launch_shape<4, 4>(a, bwd, ln, gx, gy, st);
  return hipErrorInvalidValue;
}

}  // namespace lm
}  // namespace pinn
