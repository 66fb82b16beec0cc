// Host side of the layer-major engine: PinnNetDesc -> node program -> launch list.  See lm_common.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../../include/pinn_jet.h"
#include "lm_ew.h"
#include "lm_head.h"

namespace pinn {
namespace lm {

struct CallArgs {
  const PinnNetDesc* net;
  const float* const* weights;
  float* const* grads;  // null for forward-only calls
  int num_tensors;
  PdeDev pde;
  const float* x;
  const float* t;
  long long N;
  int nt, nx;
  int mode;  // MODE_JETS | MODE_PDE
  float grad_scale;
  float* const* jets_out;
  const float* const* jets_bar;
  float* residual_out;
  float* loss_sum;
  const float* res_bar;
  void* workspace;
  size_t ws_bytes;
  bool bwd;
  bool deterministic;
  hipStream_t stream;
};

// tensors the reference's state_dict holds for this descriptor (the length every weight table must have)
int lm_expected_tensors(const PinnNetDesc* d);
// 0 if the engine can run this descriptor, else a PinnStatus with a message in err
int lm_check(const PinnNetDesc* d, char* err, size_t errlen);
size_t lm_workspace_bytes(const PinnNetDesc* d, long long N, int nt, int nx, bool bwd, bool deterministic);
int lm_run(const CallArgs& c, char* err, size_t errlen);

// per-stream-set translation units (lm_inst.hip); `act` = PinnAct, -1 = none, -2 = Fourier features (forward only)
#define PINN_LM_DECL(nt, nx)                                                                              \
  hipError_t launch_lm_ew_##nt##_##nx(const EwArgs&, bool bwd, int act, int fpt, int grid, hipStream_t); \
  hipError_t launch_lm_head_##nt##_##nx(const HeadArgs&, int fpt, int grid, hipStream_t);
PINN_LM_DECL(0, 0)
PINN_LM_DECL(1, 0)
PINN_LM_DECL(1, 1)
PINN_LM_DECL(1, 2)
PINN_LM_DECL(1, 3)
PINN_LM_DECL(1, 4)
PINN_LM_DECL(2, 0)
PINN_LM_DECL(2, 2)
#undef PINN_LM_DECL

}  // namespace lm
}  // namespace pinn
