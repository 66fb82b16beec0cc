// C ABI of the jet engine (include/pinn_jet.h): descriptor validation, pointer plumbing, launches.
// No device memory is allocated or retained here; every launch goes to the caller's stream.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "jet_kernel_attn.h"
#include "jet_kernel_wide.h"

namespace pinn {
#define PINN_DECL(nt, nx)                                                               \
  hipError_t launch_jet_##nt##_##nx(const KernelArgs&, bool, int, int, hipStream_t); \
  hipError_t launch_jetw_##nt##_##nx(const KernelArgs&, bool, int, hipStream_t);      \
  hipError_t launch_jetr_##nt##_##nx(const KernelArgs&, bool, int, hipStream_t);      \
  hipError_t launch_jeta_##nt##_##nx(const KernelArgs&, bool, int, hipStream_t);
#ifdef PINN_DEV /* developer build: ONE stream set, -DPINN_DEV_NT / -DPINN_DEV_NX (default 1, 2) */
#ifndef PINN_DEV_NT
#define PINN_DEV_NT 1
#define PINN_DEV_NX 2
#endif
#define PINN_DECL2(a, b) PINN_DECL(a, b)
PINN_DECL2(PINN_DEV_NT, PINN_DEV_NX)
#else
PINN_DECL(0, 0)
PINN_DECL(1, 0)
PINN_DECL(1, 1)
PINN_DECL(1, 2)
PINN_DECL(1, 3)
PINN_DECL(1, 4)
PINN_DECL(2, 0)
PINN_DECL(2, 2)
#endif
#undef PINN_DECL

static unsigned long long* g_stamps = nullptr;  // diagnostic builds: device buffer for in-kernel phase timing

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int num_cus() {
  static int cached = 0;
  if (cached) return cached;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
    cached = cus;
    return cus;
  }
  (void)hipGetLastError();
  return 256;  // MI355X; used only for sizing when no device is visible (CPU-side workspace queries)
}

constexpr size_t kLdsLimit = 160 * 1024;

static float act_param_of(int act, float user) {
  switch (act) {
    case PINN_ACT_SIN: return user;
    case PINN_ACT_RELU: return 0.0f;
    case PINN_ACT_LEAKY_RELU: return 0.01f;  // nn.LeakyReLU() default slope
    case PINN_ACT_IDENTITY: return 1.0f;
    default: return 0.0f;
  }
}

// PinnNetDesc + state_dict-ordered pointers -> the layer program the kernel executes.
static int build_net(const PinnNetDesc* d, const float* const* w, float* const* g, NetDev* out) {
  if (!d || !w) return fail(PINN_ERR_BAD_DESC, "null descriptor or weights");
  if (d->num_linear < 2 || d->num_linear > PINN_MAX_LINEAR)
    return fail(PINN_ERR_BAD_DESC, "num_linear=%d outside [2,%d]", d->num_linear, PINN_MAX_LINEAR);
  if (d->input_dim < 1 || d->input_dim > kMaxDin) return fail(PINN_ERR_UNSUPPORTED, "input_dim=%d (max %d)", d->input_dim, kMaxDin);
  if (d->widths[d->num_linear - 1] != 1) return fail(PINN_ERR_UNSUPPORTED, "output_dim must be 1");
  if (d->activation < PINN_ACT_TANH || d->activation > PINN_ACT_IDENTITY)
    return fail(PINN_ERR_BAD_DESC, "unknown activation id %d", d->activation);
  NetDev n;
  memset(&n, 0, sizeof(n));
  n.din = d->input_dim;
  const int act = d->arch == PINN_ARCH_SIREN ? PINN_ACT_SIN : d->activation;
  const float par = act_param_of(act, d->act_param);
  int first_mfma;  // index of the first Linear executed on MFMA
  int wbase;       // index of Linear 0's weight in the pointer arrays
  if (d->arch == PINN_ARCH_FOURIER) {
    if (d->mapping_size <= 0 || (2 * d->mapping_size) % 8) return fail(PINN_ERR_UNSUPPORTED, "fourier mapping_size=%d must be a multiple of 4", d->mapping_size);
    n.enc = ENC_FOURIER;
    n.enc_out = 2 * d->mapping_size;
    n.encW = w[0];
    first_mfma = 0;
    wbase = 1;
  } else if (d->arch == PINN_ARCH_FEEDFORWARD || d->arch == PINN_ARCH_SIREN) {
    n.enc = ENC_LINEAR;
    n.enc_out = d->widths[0];
    n.encW = w[0];
    n.encb = w[1];
    n.d_encW = g ? g[0] : nullptr;
    n.d_encb = g ? g[1] : nullptr;
    n.enc_act = act;
    n.enc_param = par;
    first_mfma = 1;
    wbase = 0;
  } else if (d->arch == PINN_ARCH_RESNET) {
    // state_dict order (resnet.py:114-127): input_layer.{weight,bias}; per block layers.0 (Linear), layers.1 (LN1),
    // layers.4 (Linear), layers.5 (LN2) — weight then bias each; output_layer.{weight,bias}
    const int nb = d->num_blocks;
    if (nb < 1 || 2 * nb > kMaxLayers || d->num_linear != 2 * nb + 2) return fail(PINN_ERR_BAD_DESC, "resnet: num_blocks=%d / num_linear=%d", nb, d->num_linear);
    const int H = d->widths[0];
    if (H % 32 || H <= 0 || H > 256) return fail(PINN_ERR_UNSUPPORTED, "resnet width %d must be a multiple of 32 in [32,256]", H);
    n.arch = PINN_ARCH_RESNET;
    n.ln_eps = d->ln_eps > 0.0f ? d->ln_eps : 1e-5f;
    n.enc = ENC_LINEAR;
    n.enc_out = H;
    n.encW = w[0];
    n.encb = w[1];
    n.d_encW = g ? g[0] : nullptr;
    n.d_encb = g ? g[1] : nullptr;
    n.enc_act = act;
    n.enc_param = par;
    n.n_layers = 2 * nb;
    for (int b = 0; b < nb; ++b) {
      for (int half = 0; half < 2; ++half) {
        const int base = 2 + 8 * b + 4 * half;
        LayerDev& L = n.layer[2 * b + half];
        L.W = w[base];
        L.b = w[base + 1];
        L.ln_g = w[base + 2];
        L.ln_b = w[base + 3];
        L.dW = g ? g[base] : nullptr;
        L.db = g ? g[base + 1] : nullptr;
        L.d_ln_g = g ? g[base + 2] : nullptr;
        L.d_ln_b = g ? g[base + 3] : nullptr;
        L.in_dim = H;
        L.ld = H;
        L.out_dim = H;
        L.act = act;
        L.act_param = par;
        if (!L.W || !L.b || !L.ln_g || !L.ln_b) return fail(PINN_ERR_BAD_DESC, "null weight pointer in resnet block %d", b);
        if ((reinterpret_cast<uintptr_t>(L.W) & 15) || (reinterpret_cast<uintptr_t>(L.b) & 15))
          return fail(PINN_ERR_MISALIGNED, "resnet block %d weights are not 16-byte aligned", b);
      }
    }
    const int io = 2 + 8 * nb;
    n.w_out = w[io];
    n.b_out = w[io + 1];
    n.dw_out = g ? g[io] : nullptr;
    n.db_out = g ? g[io + 1] : nullptr;
    n.h_last = H;
    n.hmax = H;
    if (!n.w_out || !n.b_out || !n.encW || !n.encb) return fail(PINN_ERR_BAD_DESC, "null weight pointer");
    *out = n;
    return PINN_OK;
  } else if (d->arch == PINN_ARCH_ATTENTION) {
    // state_dict order (attention.py:136-156): input_proj.{w,b}; per layer: query, key, value, proj (w,b each),
    // layer_norm (LN_a), net.0, net.3, layer_norm (LN_f); output_proj.{w,b}.  query/key are dead (sequence length 1).
    const int nl = d->num_blocks;
    if (nl < 1 || 4 * nl > kMaxLayers || d->num_linear != 2) return fail(PINN_ERR_BAD_DESC, "attention: num_blocks=%d / num_linear=%d", nl, d->num_linear);
    const int H = d->widths[0];
    if (H % 32 || H <= 0 || H > 256) return fail(PINN_ERR_UNSUPPORTED, "attention width %d must be a multiple of 32 in [32,256]", H);
    n.arch = PINN_ARCH_ATTENTION;
    n.ln_eps = d->ln_eps > 0.0f ? d->ln_eps : 1e-5f;
    n.enc = ENC_LINEAR;
    n.enc_out = H;
    n.encW = w[0];
    n.encb = w[1];
    n.d_encW = g ? g[0] : nullptr;
    n.d_encb = g ? g[1] : nullptr;
    n.enc_act = act;
    n.enc_param = par;
    n.n_layers = 4 * nl;
    for (int l = 0; l < nl; ++l) {
      const int base = 2 + 16 * l;
      struct { int wi, lni, in, out; } e[4] = {{base + 4, -1, H, H}, {base + 6, base + 8, H, H},
                                                {base + 10, -1, H, 4 * H}, {base + 12, base + 14, 4 * H, H}};
      for (int q = 0; q < 4; ++q) {
        LayerDev& L = n.layer[4 * l + q];
        L.W = w[e[q].wi];
        L.b = w[e[q].wi + 1];
        L.dW = g ? g[e[q].wi] : nullptr;
        L.db = g ? g[e[q].wi + 1] : nullptr;
        if (e[q].lni >= 0) {
          L.ln_g = w[e[q].lni];
          L.ln_b = w[e[q].lni + 1];
          L.d_ln_g = g ? g[e[q].lni] : nullptr;
          L.d_ln_b = g ? g[e[q].lni + 1] : nullptr;
        }
        L.in_dim = e[q].in;
        L.out_dim = e[q].out;
        L.ld = e[q].in;
        L.act = PINN_ACT_GELU;
        L.act_param = 0.0f;
        if (!L.W || !L.b) return fail(PINN_ERR_BAD_DESC, "null weight pointer in attention layer %d", l);
        if ((reinterpret_cast<uintptr_t>(L.W) & 15) || (reinterpret_cast<uintptr_t>(L.b) & 15))
          return fail(PINN_ERR_MISALIGNED, "attention layer %d weights are not 16-byte aligned", l);
      }
    }
    const int io = 2 + 16 * nl;
    n.w_out = w[io];
    n.b_out = w[io + 1];
    n.dw_out = g ? g[io] : nullptr;
    n.db_out = g ? g[io + 1] : nullptr;
    n.h_last = H;
    n.hmax = H;
    if (!n.w_out || !n.b_out || !n.encW || !n.encb) return fail(PINN_ERR_BAD_DESC, "null weight pointer");
    *out = n;
    return PINN_OK;
  } else {
    return fail(PINN_ERR_UNSUPPORTED, "architecture id %d has no fused kernel yet", d->arch);
  }
  if (n.enc_out % 32 && n.enc == ENC_LINEAR) return fail(PINN_ERR_UNSUPPORTED, "first layer width %d must be a multiple of 32", n.enc_out);
  int prev = n.enc_out, hmax = (n.enc_out + 31) / 32 * 32;
  n.n_layers = 0;
  for (int i = first_mfma; i < d->num_linear - 1; ++i) {
    const int wd = d->widths[i];
    if (wd % 32 || wd <= 0 || wd > 256) return fail(PINN_ERR_UNSUPPORTED, "hidden width %d must be a multiple of 32 in [32,256]", wd);
    if (prev % 8) return fail(PINN_ERR_UNSUPPORTED, "layer input width %d must be a multiple of 8", prev);
    LayerDev& L = n.layer[n.n_layers++];
    L.W = w[wbase + 2 * i];
    L.b = w[wbase + 2 * i + 1];
    L.dW = g ? g[wbase + 2 * i] : nullptr;
    L.db = g ? g[wbase + 2 * i + 1] : nullptr;
    L.in_dim = prev;
    L.ld = prev;
    L.out_dim = wd;
    L.act = act;
    L.act_param = par;
    if (!L.W || !L.b) return fail(PINN_ERR_BAD_DESC, "null weight pointer for Linear %d", i);
    if ((reinterpret_cast<uintptr_t>(L.W) & 15)) return fail(PINN_ERR_MISALIGNED, "Linear %d weight is not 16-byte aligned", i);
    prev = wd;
    if (wd > hmax) hmax = wd;
  }
  const int io = d->num_linear - 1;
  n.w_out = w[wbase + 2 * io];
  n.b_out = w[wbase + 2 * io + 1];
  n.dw_out = g ? g[wbase + 2 * io] : nullptr;
  n.db_out = g ? g[wbase + 2 * io + 1] : nullptr;
  n.h_last = prev;
  if (prev % 32) return fail(PINN_ERR_UNSUPPORTED, "last hidden width %d must be a multiple of 32", prev);
  if (!n.w_out || !n.b_out || !n.encW) return fail(PINN_ERR_BAD_DESC, "null weight pointer");
  n.hmax = hmax;
  *out = n;
  return PINN_OK;
}

static long long tape_floats_per_wg(const NetDev& n, int K, int ntile) {
  if (n.arch == PINN_ARCH_RESNET) return jet_resnet_tape_floats_per_wg(K, n.n_layers / 2, ntile);
  if (n.arch == PINN_ARCH_ATTENTION) return jet_attn_tape_floats_per_wg(K, n.n_layers / 4, ntile);
  return jet_tape_floats_per_wg(K, n.n_layers, ntile);
}

static int check_orders(int nt, int nx) {
  if (nt < 0 || nt > 2) return fail(PINN_ERR_BAD_ORDER, "Temporal derivative order %d is not supported. Maximum order is 2.", nt);
  if (nx < 0 || nx > 4) return fail(PINN_ERR_BAD_ORDER, "Spatial derivative order %d is not supported. Maximum order is 4.", nx);
  return PINN_OK;
}

// smallest compiled stream set that contains (nt, nx); extra streams are computed and ignored
static bool pick_streams(int nt, int nx, int* knt, int* knx) {
  static const int sets[][2] = {{0, 0}, {1, 0}, {1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 0}, {2, 2}};
  for (auto& s : sets)
    if (s[0] == nt && s[1] == nx) { *knt = nt; *knx = nx; return true; }
  return false;
}

static hipError_t dispatch(int nt, int nx, const KernelArgs& a, bool bwd, int grid, int occ, bool wide, hipStream_t st) {
#define PINN_CASE(NT_, NX_)                                                                    \
  if (nt == NT_ && nx == NX_) {                                                                \
    if (a.net.arch == PINN_ARCH_RESNET) return launch_jetr_##NT_##_##NX_(a, bwd, grid, st);    \
    if (a.net.arch == PINN_ARCH_ATTENTION) return launch_jeta_##NT_##_##NX_(a, bwd, grid, st); \
    return wide ? launch_jetw_##NT_##_##NX_(a, bwd, grid, st) : launch_jet_##NT_##_##NX_(a, bwd, grid, occ, st); \
  }
#ifdef PINN_DEV
#define PINN_CASE2(a, b) PINN_CASE(a, b)
  PINN_CASE2(PINN_DEV_NT, PINN_DEV_NX)
#else
  PINN_CASE(0, 0) PINN_CASE(1, 0) PINN_CASE(1, 1) PINN_CASE(1, 2) PINN_CASE(1, 3) PINN_CASE(1, 4) PINN_CASE(2, 0) PINN_CASE(2, 2)
#endif
#undef PINN_CASE
  return hipErrorInvalidValue;
}

// Workgroups per CU a launch is sized (and register-budgeted) for.  PINN_OCC=1 forces one (experiments).
static int occupancy_for(const NetDev& n, size_t lds, bool bwd) {
  // Forward-only launches run two workgroups per CU (256 VGPRs each).  With the reverse sweep the 256-register
  // budget spills ~2000 VGPRs and one workgroup per CU is 30 % faster (KdV / siren 4x128, K = 5: 6.4 vs 8.3 ms).
  int occ = (!bwd && n.hmax <= 128 && (int)(kLdsLimit / lds) >= 2) ? 2 : 1;
  if (const char* e = getenv("PINN_OCC")) {
    const int v = atoi(e);
    if (v == 1) occ = 1;
  }
  return occ;
}

// Kernel variant: "wide" (all K streams LDS-resident, persistent dW accumulators) whenever it fits, else the
// stream-serial kernel.  PINN_KERNEL=stream forces the stream-serial kernel (tests run both variants).
static bool use_wide(const NetDev& n, int K, bool bwd) {
  const bool fits = jet_wide_fits(K, n.hmax, bwd, n.n_layers);
  if (const char* e = getenv("PINN_KERNEL")) {
    if (!strcmp(e, "stream")) return false;
  }
  return fits;
}

static int grid_for(const NetDev& n, int K, long long N, bool bwd, size_t* lds_out, int* occ_out = nullptr,
                    bool* wide_out = nullptr) {
  const bool resnet = n.arch == PINN_ARCH_RESNET || n.arch == PINN_ARCH_ATTENTION;  // LayerNorm kernels
  const bool wide = !resnet && use_wide(n, K, bwd);
  if (wide_out) *wide_out = wide;
  const size_t lds = resnet ? jet_resnet_lds_bytes(K, n.hmax, bwd)
                            : (wide ? jet_wide_lds_bytes(K, jet_wide_hmax(n.hmax), bwd, n.n_layers) : jet_lds_bytes(K, n.hmax, bwd));
  if (lds_out) *lds_out = lds;
  if (lds > kLdsLimit) return 0;
  const long long ntiles = (N + kT - 1) / kT;
  const int per_cu = (wide || resnet) ? 1 : occupancy_for(n, lds, bwd);
  if (occ_out) *occ_out = per_cu;
  long long g = (long long)num_cus() * per_cu;
  if (g > ntiles) g = ntiles;
  return (int)(g < 1 ? 1 : g);
}

static int run(const PinnNetDesc* net, const float* const* weights, float* const* grads, const PinnPdeDesc* pde,
               const float* x, const float* t, int64_t N, int nt, int nx, int mode, float grad_scale,
               float* const* jets_out, const float* const* jets_bar, float* residual_out, float* loss_sum,
               void* workspace, size_t ws_bytes, bool bwd, void* stream, const float* res_bar = nullptr) {
  if (N <= 0) return PINN_OK;
  if (!x && net && net->input_dim > 1) return fail(PINN_ERR_BAD_DESC, "x is null");
  if (!t) return fail(PINN_ERR_BAD_DESC, "t is null");
  int rc = check_orders(nt, nx);
  if (rc) return rc;
  int knt, knx;
  if (!pick_streams(nt, nx, &knt, &knx)) return fail(PINN_ERR_UNSUPPORTED, "stream set (nt=%d, nx=%d) is not compiled", nt, nx);
  KernelArgs a;
  memset(&a, 0, sizeof(a));
  rc = build_net(net, weights, grads, &a.net);
  if (rc) return rc;
  const int K = 1 + knt + knx;
  size_t lds = 0;
  int occ = 1;
  bool wide = false;
  const int grid = grid_for(a.net, K, N, bwd, &lds, &occ, &wide);
  if (grid == 0)
    return fail(PINN_ERR_UNSUPPORTED, "LDS need %zu B > %zu B (K=%d streams, width %d%s)", lds, kLdsLimit, K, a.net.hmax, bwd ? ", reverse sweep" : "");
  if (pde) {
    a.pde.kind = pde->kind;
    a.pde.dimension = pde->dimension;
    a.pde.loss = pde->loss;
    a.pde.c0 = pde->coef[0];
    a.pde.c1 = pde->coef[1];
    a.pde.c2 = pde->coef[2];
    a.pde.c3 = pde->coef[3];
    a.pde.huber_delta = pde->huber_delta;
  }
  a.x = x;
  a.t = t;
  a.N = N;
  a.mode = mode;
  a.grad_scale = grad_scale;
  for (int s = 0; s < K; ++s) {
    a.jets_out[s] = jets_out ? jets_out[s] : nullptr;
    a.jets_bar[s] = jets_bar ? jets_bar[s] : nullptr;
  }
  a.residual_out = residual_out;
  a.loss_sum = loss_sum;
  a.res_bar = res_bar;
  a.stamps = g_stamps;
  if (bwd) {
    const int ntile = a.net.hmax > 128 ? 2 : 1;
    a.tape_stride = tape_floats_per_wg(a.net, K, ntile);
    const size_t need = (size_t)a.tape_stride * sizeof(float) * grid;
    if (need > 0 && (!workspace || ws_bytes < need))
      return fail(PINN_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    if ((reinterpret_cast<uintptr_t>(workspace) & 15)) return fail(PINN_ERR_MISALIGNED, "workspace is not 16-byte aligned");
    a.tape = static_cast<float*>(workspace);
  }
  const hipError_t e = dispatch(knt, knx, a, bwd, grid, occ, wide, static_cast<hipStream_t>(stream));
  if (e == hipErrorNotSupported) return fail(PINN_ERR_UNSUPPORTED, "derivative orders above 2 through LayerNorm (nt=%d, nx=%d)", knt, knx);
  if (e != hipSuccess) return fail(PINN_ERR_HIP, "HIP error %d: %s", (int)e, hipGetErrorString(e));
  return PINN_OK;
}

}  // namespace pinn

using namespace pinn;

extern "C" {

int pinn_abi_version(void) { return PINN_ABI_VERSION; }

const char* pinn_last_error(void) { return g_err; }

int pinn_pde_streams(const PinnPdeDesc* pde, int32_t* time_order, int32_t* space_order) {
  if (!pde || !time_order || !space_order) return fail(PINN_ERR_BAD_DESC, "null argument");
  int nt = 1, nx = 0;
  if (pde->dimension > 1) {
    nt = (pde->kind == PINN_PDE_WAVE || pde->kind == PINN_PDE_PENDULUM) ? 2 : 1;
  } else {
    switch (pde->kind) {
      case PINN_PDE_BURGERS: case PINN_PDE_ALLEN_CAHN: case PINN_PDE_BLACK_SCHOLES: case PINN_PDE_HEAT_LAPLACIAN: nt = 1; nx = 2; break;
      case PINN_PDE_HEAT: case PINN_PDE_CONVECTION: nt = 1; nx = 1; break;
      case PINN_PDE_KDV: nt = 1; nx = 3; break;
      case PINN_PDE_CAHN_HILLIARD: nt = 1; nx = 4; break;
      case PINN_PDE_WAVE: nt = 2; nx = 2; break;
      case PINN_PDE_PENDULUM: nt = 2; nx = 0; break;
      default: return fail(PINN_ERR_BAD_DESC, "unknown pde kind %d", pde->kind);
    }
  }
  *time_order = nt;
  *space_order = nx;
  return PINN_OK;
}

size_t pinn_workspace_bytes(const PinnNetDesc* net, int64_t N, int32_t time_order, int32_t space_order) {
  if (!net || N <= 0) return 0;
  // pointers are irrelevant for sizing: build the program with a dummy table
  // sized for the widest weight table: attention has 16 tensors per layer (q, k, v, proj, two LayerNorms, two
  // feed-forward Linears) — a table of 2 * PINN_MAX_LINEAR + 2 entries was overrun by a 4-layer attention network
  const float* dummy[16 * PINN_MAX_LINEAR + 8];
  static float sixteen_aligned[4] __attribute__((aligned(16)));
  for (auto& p : dummy) p = sixteen_aligned;
  NetDev n;
  if (build_net(net, dummy, nullptr, &n) != PINN_OK) return 0;
  const int K = 1 + time_order + space_order;
  const int grid = grid_for(n, K, N, true, nullptr);
  const int ntile = n.hmax > 128 ? 2 : 1;
  return (size_t)tape_floats_per_wg(n, K, ntile) * sizeof(float) * (size_t)grid;
}

int pinn_jet_forward(const PinnNetDesc* net, const float* const* weights, const float* x, const float* t, int64_t N,
                     int32_t time_order, int32_t space_order, float* const* jets_out, void* stream) {
  if (!jets_out) return fail(PINN_ERR_BAD_DESC, "jets_out is null");
  return run(net, weights, nullptr, nullptr, x, t, N, time_order, space_order, MODE_JETS, 0.0f, jets_out, nullptr,
             nullptr, nullptr, nullptr, 0, false, stream);
}

int pinn_jet_backward(const PinnNetDesc* net, const float* const* weights, const float* x, const float* t, int64_t N,
                      int32_t time_order, int32_t space_order, const float* const* jet_cotangents,
                      float* const* weight_grads, void* workspace, size_t ws_bytes, void* stream) {
  if (!jet_cotangents || !weight_grads) return fail(PINN_ERR_BAD_DESC, "null cotangents or weight_grads");
  return run(net, weights, weight_grads, nullptr, x, t, N, time_order, space_order, MODE_JETS, 0.0f, nullptr,
             jet_cotangents, nullptr, nullptr, workspace, ws_bytes, true, stream);
}

int pinn_residual_forward(const PinnNetDesc* net, const float* const* weights, const PinnPdeDesc* pde, const float* x,
                          const float* t, int64_t N, float* residual_out, float* loss_sum_out, void* stream) {
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, nullptr, pde, x, t, N, nt, nx, MODE_PDE, 0.0f, nullptr, nullptr, residual_out,
             loss_sum_out, nullptr, 0, false, stream);
}

int pinn_residual_loss_grad(const PinnNetDesc* net, const float* const* weights, const PinnPdeDesc* pde,
                            const float* x, const float* t, int64_t N, float grad_scale, float* residual_out,
                            float* loss_sum_out, float* const* weight_grads, void* workspace, size_t ws_bytes,
                            void* stream) {
  if (!weight_grads) return fail(PINN_ERR_BAD_DESC, "weight_grads is null");
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, weight_grads, pde, x, t, N, nt, nx, MODE_PDE, grad_scale, nullptr, nullptr, residual_out,
             loss_sum_out, workspace, ws_bytes, true, stream);
}

int pinn_residual_backward(const PinnNetDesc* net, const float* const* weights, const PinnPdeDesc* pde, const float* x,
                           const float* t, int64_t N, const float* residual_cotangent, float* const* weight_grads,
                           void* workspace, size_t ws_bytes, void* stream) {
  if (!weight_grads || !residual_cotangent) return fail(PINN_ERR_BAD_DESC, "null cotangent or weight_grads");
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, weight_grads, pde, x, t, N, nt, nx, MODE_PDE, 0.0f, nullptr, nullptr, nullptr, nullptr,
             workspace, ws_bytes, true, stream, residual_cotangent);
}

/* Diagnostic hook (not part of the documented ABI): device buffer of grid*4*16 uint64 for -DPINN_STAMPS builds. */
void pinn_debug_set_stamps(void* device_buffer) { g_stamps = static_cast<unsigned long long*>(device_buffer); }

}  // extern "C"
