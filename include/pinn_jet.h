/*
 * pinn_jet.h — C ABI of the MI355X (gfx950) PINN jet engine.
 *
 * The reference (pinnrl 0.3.1) has no FFI: its boundary for this path is the
 * Python object API.  Each entry point below replaces the ATen op sequence that
 * one reference call site generates; the Python host mirror binds them with
 * ctypes (see INTEGRATION.md for the stub a pinnrl maintainer would add).
 *
 *   pinn_jet_forward         <- PINNModel.forward (pinnrl/neural_networks/__init__.py:144-154)
 *                               + PDEBase.compute_derivatives (pinnrl/pdes/pde_base.py:590-794):
 *                               u and its input derivatives ("jets") in one launch.
 *   pinn_jet_backward        <- loss.backward() through that graph (pinnrl/training/trainer.py:689):
 *                               d(sum_s <cotangent_s, jet_s>)/d(theta), accumulated into weight_grads.
 *   pinn_residual_forward    <- XxxEquation.compute_residual (pinnrl/pdes/burgers_equation.py:40-75,
 *                               heat_equation.py:54-110, allen_cahn.py:39-111, kdv_equation.py:38-92,
 *                               cahn_hilliard.py:39-160, wave_equation.py:38-119,
 *                               convection_equation.py:43-78, black_scholes.py:44-93,
 *                               pendulum_equation.py:51-94) + PDEBase._apply_loss_fn
 *                               (pde_base.py:309-326) as a fused per-point epilogue.
 *   pinn_residual_backward   <- loss.backward() through a residual tensor (trainer.py:689, :607-626).
 *   pinn_residual_loss_grad  <- the metric's unit of work: compute_residual -> mean loss -> backward,
 *                               one launch (pde_base.py:1098-1099 + trainer.py:689).
 *
 * Conventions: fp32, contiguous.  x:(N,dim) row-major, t:(N,1); the network input is
 * cat([x,t],1) — spatial columns first, time LAST (pde_base.py:640).  All pointers are
 * device pointers owned by the caller (PyTorch); the library never allocates, frees or
 * retains device memory and launches on the stream it is given without synchronising.
 * Return value: 0 = ok, negative = PinnStatus; pinn_last_error() describes the failure
 * (thread-local).  `weights` / `weight_grads` follow the reference's state_dict order
 * for the architecture (buffers included, e.g. fourier: B, W0, b0, ..., W_out, b_out);
 * a NULL entry in weight_grads skips that tensor.
 */
#ifndef PINN_JET_H
#define PINN_JET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PINN_ABI_VERSION 2
#define PINN_MAX_LINEAR 24
#define PINN_MAX_STREAMS 7 /* value + up to 2 time + up to 4 space derivatives */

/* PinnNetDesc.flags */
#define PINN_FLAG_LAYER_NORM 1    /* feedforward: a LayerNorm follows every hidden Linear (feedforward.py:43-45) */
#define PINN_FLAG_DETERMINISTIC 2 /* weight gradients (and, on calls WITH a reverse sweep, the loss sum) reduced in a fixed
                                     order: two launches on the same inputs give bit-identical results (reference anchor:
                                     tests/unit_tests/test_benchmarks.py:61-64).  Both engines: per-workgroup slab rows +
                                     ordered row sum; the workspace grows by grid x parameter count floats.  Forward-only
                                     calls: the fused tile-major kernel honours the flag for the loss sum too, the
                                     layer-major engine accumulates it with float atomics (per-point outputs are
                                     bit-reproducible either way) */
#define PINN_FLAG_LAYER_MAJOR 4   /* engine hint: run the layer-major engine even where the fused tile-major kernel
                                     applies (same results to rounding; tests run both) */

typedef enum PinnStatus {
  PINN_OK = 0,
  PINN_ERR_BAD_DESC = -1,
  PINN_ERR_UNSUPPORTED = -2,
  PINN_ERR_MISALIGNED = -3,
  PINN_ERR_WORKSPACE = -4,
  PINN_ERR_HIP = -5,
  PINN_ERR_BAD_ORDER = -6
} PinnStatus;

typedef enum PinnArch {
  PINN_ARCH_FEEDFORWARD = 0, /* feedforward.py:9-73 (PINN_FLAG_LAYER_NORM: Linear, LayerNorm, act per hidden layer) */
  PINN_ARCH_FOURIER = 1,     /* fourier.py:65-124 */
  PINN_ARCH_SIREN = 2,       /* siren.py:49-90 */
  PINN_ARCH_RESNET = 3,      /* resnet.py:68-142 */
  PINN_ARCH_ATTENTION = 4    /* attention.py:110-183 (sequence length 1 => an MLP with LayerNorm) */
} PinnArch;

typedef enum PinnAct { /* base_network.py:91-104, plus SIREN's sin(omega_0 z) */
  PINN_ACT_TANH = 0,
  PINN_ACT_SIN = 1,
  PINN_ACT_GELU = 2,
  PINN_ACT_SIGMOID = 3,
  PINN_ACT_RELU = 4,
  PINN_ACT_LEAKY_RELU = 5,
  PINN_ACT_IDENTITY = 6
} PinnAct;

typedef enum PinnPde { /* as-reference residuals; see DESIGN.md for the quirks kept */
  PINN_PDE_BURGERS = 0,       /* u_t + u u_x - c0 u_xx              c0 = nu        */
  PINN_PDE_HEAT = 1,          /* u_t - c0 u_x   (reference: "dx2" is a FIRST derivative) */
  PINN_PDE_ALLEN_CAHN = 2,    /* u_t - c0^2 u_xx - u + u^3          c0 = epsilon   */
  PINN_PDE_KDV = 3,           /* u_t + 6 u u_x + u_xxx                             */
  PINN_PDE_CAHN_HILLIARD = 4, /* u_t - d_xx(-c0^2 u_xx + c^3 - c), c = clamp(u,+-10) */
  PINN_PDE_WAVE = 5,          /* u_tt - c0^2 u_xx                                  */
  PINN_PDE_CONVECTION = 6,    /* u_t + c0 u_x                                      */
  PINN_PDE_BLACK_SCHOLES = 7, /* u_t + .5 c0^2 x^2 u_xx + c1 x u_x - c1 u  (sigma, r) */
  PINN_PDE_PENDULUM = 8,      /* u_tt + c0 sin(u)                   c0 = g/L       */
  PINN_PDE_HEAT_LAPLACIAN = 9 /* u_t - c0 u_xx  (the intended heat equation; never the parity path) */
} PinnPde;

typedef enum PinnLoss { PINN_LOSS_MSE = 0, PINN_LOSS_MAE = 1, PINN_LOSS_HUBER = 2 } PinnLoss;

typedef struct PinnNetDesc {
  int32_t arch;                    /* PinnArch */
  int32_t activation;              /* PinnAct of the hidden layers */
  int32_t input_dim;               /* spatial dimension + 1 */
  int32_t num_linear;              /* Linear layers including the output layer */
  int32_t widths[PINN_MAX_LINEAR]; /* out_features of each Linear (1 .. 1024, any value); widths[num_linear-1] must be 1 */
  int32_t mapping_size;            /* fourier: columns of B (features = 2 * mapping_size) */
  float act_param;                 /* omega_0 for PINN_ACT_SIN */
  float ln_eps;                    /* LayerNorm epsilon (resnet / attention) */
  int32_t num_blocks;              /* resnet blocks / attention layers */
  int32_t flags;                   /* PINN_FLAG_* */
} PinnNetDesc;

typedef struct PinnPdeDesc {
  int32_t kind;      /* PinnPde */
  int32_t dimension; /* spatial dimension; >= 2 keeps only the terms the reference keeps */
  int32_t loss;      /* PinnLoss */
  float coef[4];
  float huber_delta;
} PinnPdeDesc;

int pinn_abi_version(void);
const char* pinn_last_error(void);

/* How the library was built: one line per kernel translation unit that did not build in its preferred form
 * (see csrc/Makefile), or "" — so that a degraded build is visible to callers, tests and the bench line. */
const char* pinn_build_info(void);

/* Number of tensors the reference's state_dict holds for this descriptor — the length every `weights` /
 * `weight_grads` table passed below must have (`num_tensors`).  Negative PinnStatus on a bad descriptor. */
int pinn_num_tensors(const PinnNetDesc* net);

/* (time_order, space_order) of the jet streams a PDE's residual consumes; K = 1 + nt + nx. */
int pinn_pde_streams(const PinnPdeDesc* pde, int32_t* time_order, int32_t* space_order);

/* Bytes of scratch a call on N points needs (`backward` = 0 for the forward-only entry points, 1 otherwise).
 * Zero is a valid answer (small networks run from registers and LDS alone); 0 is also returned for a descriptor the
 * library cannot run — the compute entry points then report why. */
size_t pinn_workspace_bytes(const PinnNetDesc* net, int64_t N, int32_t time_order, int32_t space_order, int32_t backward);

/* Every compute entry point: `weights` (and `weight_grads`) are tables of `num_tensors` device pointers in the
 * reference's state_dict order; the count is validated against the descriptor BEFORE any entry is read.
 * `workspace` must hold pinn_workspace_bytes(...) bytes, 16-byte aligned (may be NULL when that is 0).
 * Weight tensors: contiguous fp32, hidden-layer weight matrices 16-byte aligned where the descriptor takes the fused
 * tile-major kernel (plain MLP family, widths multiples of 32 up to 128: it reads them in place with 16-byte loads) —
 * a misaligned view there is refused with PINN_ERR_MISALIGNED rather than silently re-routed to the engine
 * pinn_workspace_bytes did not size for; PINN_FLAG_LAYER_MAJOR selects the packing engine, which takes any alignment. */

/* jets_out[s] : N floats each, stream order [u, d/dt.., d/dx..]; all K = 1+nt+nx entries required. */
int pinn_jet_forward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors, const float* x,
                     const float* t, int64_t N, int32_t time_order, int32_t space_order, float* const* jets_out,
                     void* workspace, size_t ws_bytes, void* stream);

/* Recomputes the forward, then accumulates
 * d(sum_s sum_n jet_cotangents[s][n] * jet_s[n]) / d(weights) into weight_grads (+=). */
int pinn_jet_backward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors, const float* x,
                      const float* t, int64_t N, int32_t time_order, int32_t space_order,
                      const float* const* jet_cotangents, float* const* weight_grads, void* workspace, size_t ws_bytes,
                      void* stream);

/* residual_out: N floats or NULL.  loss_sum_out: 1 float or NULL; receives += sum_n l(r_n)
 * (l = r^2 | |r| | huber), i.e. the UNnormalised loss — the caller divides by the global N. */
int pinn_residual_forward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                          const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, float* residual_out,
                          float* loss_sum_out, void* workspace, size_t ws_bytes, void* stream);

/* One call: residual, loss sum, and weight_grads += grad_scale * d(sum_n l(r_n))/d(weights).
 * For mean-squared loss over N_total points use grad_scale = upstream / N_total. */
int pinn_residual_loss_grad(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                            const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, float grad_scale,
                            float* residual_out, float* loss_sum_out, float* const* weight_grads, void* workspace,
                            size_t ws_bytes, void* stream);

/* The same with trainable PDE coefficients (inverse problems, pinnrl/pdes/pde_base.py:246-279: get_parameter returns a live
 * nn.Parameter that sits inside the residual): additionally coef_grads[k] += grad_scale * d(sum_n l(r_n))/d(pde->coef[k])
 * for k = 0, 1 (device pointer to >= 2 floats, nullable = pinn_residual_loss_grad; coefficients 2 and 3 are unused by the nine
 * PDEs).  One extra per-point reduction in the residual epilogue of the layer-major engine's head kernel; no second pass.
 * Descriptors that would take the fused tile-major kernel must carry PINN_FLAG_LAYER_MAJOR for such a call (and for the
 * pinn_workspace_bytes query that sizes its workspace): PINN_ERR_UNSUPPORTED otherwise. */
int pinn_residual_loss_grad_coef(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                                 const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, float grad_scale,
                                 float* residual_out, float* loss_sum_out, float* const* weight_grads, float* coef_grads,
                                 void* workspace, size_t ws_bytes, void* stream);

/* weight_grads += d(sum_n residual_cotangent[n] * r_n)/d(weights): the backward of pinn_residual_forward for an
 * arbitrary downstream graph (loss.backward() through `residual`, trainer.py:689; LRW's per-component
 * backward passes, trainer.py:607-626). */
int pinn_residual_backward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                           const PinnPdeDesc* pde, const float* x, const float* t, int64_t N,
                           const float* residual_cotangent, float* const* weight_grads, void* workspace,
                           size_t ws_bytes, void* stream);

/* ---- training step (pinnrl/training/trainer.py:686-698, pinnrl/pdes/pde_base.py:1101-1165) -------------------------
 * With these two entry points a whole optimiser step is a handful of launches with no autograd in it (and can be
 * captured in a HIP graph): pinn_residual_loss_grad for the residual term, pinn_jet_forward / pinn_jet_backward
 * (orders 0, 0) for the network values on the boundary / initial points, pinn_point_losses for their loss terms,
 * pinn_adam_clip_step for clip_grad_norm_ + Adam. */
#define PINN_MAX_POINT_TERMS 8

/* Term k (k < n_terms) covers points [lo[k], hi[k]) of u and has its own target array of hi[k] - lo[k] floats:
 * term_losses[k] = mean l(u - target_k) with l = PinnLoss `loss` (pde_base.py:309-326), and
 * cotangent[n] = sum_k weights[k] * l'(u[n] - target_k[n]) / (hi[k] - lo[k])  (n_total floats, overwritten).
 * lo / hi / targets / weights are HOST arrays (read before the call returns); u, targets[k], outputs: device.
 * summary4 (nullable, device): {residual, boundary, initial, total} of compute_loss — residual = residual_sum[0] *
 * residual_scale (the residual launch's loss sum / N), boundary = sum of the first n_boundary_terms term losses,
 * initial = the rest, total = residual_weight * residual + sum_k weights[k] * term_losses[k]. */
int pinn_point_losses(const float* u, int32_t n_total, int32_t n_terms, const int32_t* lo, const int32_t* hi,
                      const float* const* targets, const float* weights, int32_t loss, float huber_delta,
                      float* term_losses, float* cotangent, const float* residual_sum, float residual_scale,
                      float residual_weight, int32_t n_boundary_terms, float* summary4, void* stream);

/* The general form of pinn_point_losses, for PDEs whose compute_loss reads more than the value stream on its boundary
 * points (HeatEquation.compute_loss, pinnrl/pdes/heat_equation.py:375-623: periodic boundary conditions on u AND du/dx at
 * paired wall points): jets is the (n_streams x n_total) output of pinn_jet_forward; term k reads stream stream_of[k] and is
 * either  l(J[n] - targets[k][n - lo[k]])  or — pair_offset[k] != 0, targets[k] may be null —  l(J[n] - J[n + pair_offset[k]])
 * for n in [lo[k], hi[k]) (the partner range must not overlap it).  cotangent: (n_streams x n_total), overwritten: the
 * jet_cotangents of pinn_jet_backward.  Everything else as pinn_point_losses. */
int pinn_jet_losses(const float* jets, int32_t n_streams, int32_t n_total, int32_t n_terms, const int32_t* lo, const int32_t* hi,
                    const int32_t* stream_of, const int32_t* pair_offset, const float* const* targets, const float* weights,
                    int32_t loss, float huber_delta, float* term_losses, float* cotangent, const float* residual_sum,
                    float residual_scale, float residual_weight, int32_t n_boundary_terms, float* summary4, void* stream);

/* torch.nn.utils.clip_grad_norm_(params, max_norm) (skipped when max_norm <= 0) followed by
 * torch.optim.Adam(lr, (beta1, beta2), eps, weight_decay).step() on ONE flat fp32 buffer of n elements.
 * lr and step are DEVICE scalars (step = number of steps taken so far, incremented by the call) so that a captured
 * graph follows a learning-rate schedule; scratch64: 64 floats; grad_norm_out: nullable, receives the norm before
 * clipping.  The norm is reduced in a fixed order: deterministic. */
int pinn_adam_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr,
                        float beta1, float beta2, float eps, float weight_decay, float max_norm, float* step,
                        float* scratch64, float* grad_norm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PINN_JET_H */
