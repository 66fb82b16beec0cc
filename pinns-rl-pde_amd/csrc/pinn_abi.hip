// C ABI of the jet engine (include/pinn_jet.h): descriptor validation, engine choice, pointer plumbing.
// No device memory is allocated or retained here; every launch goes to the caller's stream.
//
// Two engines sit behind the same entry points:
//   * the fused tile-major "wide" kernel (jet_kernel_wide.h): plain MLP family (feedforward without LayerNorm,
//     fourier, siren), hidden widths multiples of 32 up to 128, all K streams of a tile resident in LDS — the
//     headline Burgers / fourier 4x128 configuration;
//   * the layer-major engine (lm_*.h): everything else — LayerNorm architectures (ResNet, attention, feedforward with
//     layer_norm), widths up to 1024 that need not be multiples of 32, any derivative order the ABI admits.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "jet_kernel_wide.h"
#include "lm_engine.h"

namespace pinn {
#define PINN_DECL(nt, nx) hipError_t launch_jetw_##nt##_##nx(const KernelArgs&, bool, int, hipStream_t);
PINN_DECL(0, 0)
PINN_DECL(1, 0)
PINN_DECL(1, 1)
PINN_DECL(1, 2)
PINN_DECL(1, 3)
PINN_DECL(1, 4)
PINN_DECL(2, 0)
PINN_DECL(2, 2)
#undef PINN_DECL

#ifdef PINN_STAMPS
static unsigned long long* g_stamps = nullptr;  // diagnostic builds only: device buffer for in-kernel phase timing
#endif

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace pinn

// the training-step entry points (train_kernels.hip) leave their messages in the same thread-local string
extern "C" int pinn_internal_fail(int code, const char* msg) { return pinn::fail(code, "%s", msg); }

namespace pinn {

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


// PINN_KERNEL=lm routes every architecture through the layer-major engine (tests run both engines on the MLP
// family).  Read once per process: the ABI itself carries no mutable state.
static bool force_lm() {
  static const bool v = [] {
    const char* e = getenv("PINN_KERNEL");
    return e && !strcmp(e, "lm");
  }();
  return v;
}

static float act_param_of(int act, float user) {
  switch (act) {
    case PINN_ACT_SIN: return user;
    case PINN_ACT_RELU: return 0.0f;
    case PINN_ACT_LEAKY_RELU: return 0.01f;  // nn.LeakyReLU() default slope
    case PINN_ACT_IDENTITY: return 1.0f;
    default: return 0.0f;
  }
}

// Wide-kernel layer program of the plain-MLP family, or false when the descriptor is outside what that kernel runs
// (the caller then takes the layer-major engine).  `w` / `g` may be null (sizing queries); when given they hold
// `num_tensors` entries, already validated against the descriptor.
static bool build_wide(const PinnNetDesc* d, const float* const* w, float* const* g, NetDev* out, int* misaligned = nullptr) {
  if (d->arch != PINN_ARCH_FOURIER && d->arch != PINN_ARCH_FEEDFORWARD && d->arch != PINN_ARCH_SIREN) return false;
  if (d->flags & (PINN_FLAG_LAYER_NORM | PINN_FLAG_LAYER_MAJOR)) return false;
  NetDev n;
  memset(&n, 0, sizeof(n));
  n.din = d->input_dim;
  const int act = d->arch == PINN_ARCH_SIREN ? PINN_ACT_SIN : d->activation;
  const float par = act_param_of(act, d->act_param);
  static float sixteen_aligned[4] __attribute__((aligned(16)));
  auto W = [&](int i) -> const float* { return w ? w[i] : sixteen_aligned; };
  auto G = [&](int i) -> float* { return g ? g[i] : nullptr; };
  int first_mfma, wbase;
  if (d->arch == PINN_ARCH_FOURIER) {
    if (d->mapping_size <= 0 || (2 * d->mapping_size) % 8) return false;
    n.enc = ENC_FOURIER;
    n.enc_out = 2 * d->mapping_size;
    n.encW = W(0);
    first_mfma = 0;
    wbase = 1;
  } else {
    n.enc = ENC_LINEAR;
    n.enc_out = d->widths[0];
    n.encW = W(0);
    n.encb = W(1);
    n.d_encW = G(0);
    n.d_encb = G(1);
    n.enc_act = act;
    n.enc_param = par;
    first_mfma = 1;
    wbase = 0;
    if (n.enc_out % 32) return false;
  }
  int prev = n.enc_out, hmax = (n.enc_out + 31) / 32 * 32;
  n.n_layers = 0;
  for (int i = first_mfma; i < d->num_linear - 1; ++i) {
    const int wd = d->widths[i];
    if (wd % 32 || wd <= 0 || wd > 128 || prev % 8) return false;
    LayerDev& L = n.layer[n.n_layers++];
    L.W = W(wbase + 2 * i);
    L.b = W(wbase + 2 * i + 1);
    L.dW = G(wbase + 2 * i);
    L.db = G(wbase + 2 * i + 1);
    L.in_dim = prev;
    L.ld = prev;
    L.out_dim = wd;
    L.act = act;
    L.act_param = par;
    if (reinterpret_cast<uintptr_t>(L.W) & 15) {  // 16-byte weight loads
      if (misaligned) *misaligned = wbase + 2 * i;
      return false;
    }
    prev = wd;
    if (wd > hmax) hmax = wd;
  }
  const int io = d->num_linear - 1;
  n.w_out = W(wbase + 2 * io);
  n.b_out = W(wbase + 2 * io + 1);
  n.dw_out = G(wbase + 2 * io);
  n.db_out = G(wbase + 2 * io + 1);
  n.h_last = prev;
  if (prev % 32 || hmax > 128) return false;
  n.hmax = hmax;
  *out = n;
  return true;
}

static bool wide_store_flush_on() {  // PINN_WIDE_STORE_FLUSH=0: the two-level atomic flush everywhere (experiments), read once
  static const bool v = [] {
    const char* e = getenv("PINN_WIDE_STORE_FLUSH");
    return !(e && atoi(e) == 0);
  }();
  return v;
}

// ---- deterministic mode of the fused kernel -------------------------------------------------------------------------
// Every gradient / loss pointer of the NetDev is redirected into row 0 of a [grid][stride] slab in the workspace;
// workgroup b adds into row b with plain (non-atomic) adds, and this kernel then sums the rows of every element in
// workgroup order and adds the total to the caller's tensor: two launches on the same inputs give identical bits.
constexpr int kFlushRows = 8;  // shared slab rows of the default (non-deterministic) two-level flush

struct DetTable {
  int n;
  float* user[2 * PINN_MAX_LINEAR + 8];
  unsigned off[2 * PINN_MAX_LINEAR + 8], cnt[2 * PINN_MAX_LINEAR + 8];
  unsigned stride;
};

__global__ void wide_det_reduce(const DetTable tab, const float* slab, int rows) {
  const unsigned it = blockIdx.y;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < tab.cnt[it]; e += gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int b = 0; b < rows; ++b) s += slab[(size_t)b * tab.stride + tab.off[it] + e];
    tab.user[it][e] += s;
  }
}

// Store flush (KernelArgs::flush_store): every workgroup has WRITTEN its row; user[e] += sum over rows in row order.
// A block sums 128 consecutive elements (32 threads x 16 bytes) over 8 interleaved row groups and combines them through
// LDS in a fixed order: 42 MB of rows at ~4 TB/s instead of 256 dependent loads per thread.
__global__ __launch_bounds__(256) void wide_rows_reduce(const DetTable tab, const float* slab, int rows) {
  __shared__ float part[8][128];
  const unsigned it = blockIdx.y;
  const unsigned cnt4 = (tab.cnt[it] + 3u) & ~3u;  // rows are padded to 4 floats per target (det_slot)
  const int el = threadIdx.x & 31, rg = threadIdx.x >> 5;
  for (unsigned e0 = blockIdx.x * 128u; e0 < cnt4; e0 += gridDim.x * 128u) {
    const unsigned e = e0 + 4u * el;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < cnt4) {
      const float* src = slab + tab.off[it] + e;
#pragma unroll 8
      for (int b = rg; b < rows; b += 8) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)b * tab.stride);
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
      }
    }
    part[rg][4 * el + 0] = s0;
    part[rg][4 * el + 1] = s1;
    part[rg][4 * el + 2] = s2;
    part[rg][4 * el + 3] = s3;
    __syncthreads();
    if (threadIdx.x < 128 && e0 + threadIdx.x < tab.cnt[it]) {
      float t = 0.0f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
      tab.user[it][e0 + threadIdx.x] += t;
    }
    __syncthreads();
  }
}

// redirect one accumulation target into the slab; returns the slab pointer (row 0)
static float* det_slot(DetTable& t, float* slab, float* user, unsigned count) {
  if (!user) return nullptr;
  const int i = t.n++;
  t.user[i] = user;
  t.off[i] = t.stride;
  t.cnt[i] = count;
  t.stride += (count + 3u) & ~3u;
  return slab + t.off[i];
}

// all targets of a NetDev (and the loss sum) -> slab; `slab` may be null for sizing (only t.stride is then meaningful)
static void det_redirect(NetDev& n, float*& loss_sum, float*& dcoef, float* slab, DetTable& t) {
  memset(&t, 0, sizeof(t));
  static float dummy;
  float* base = slab ? slab : &dummy;
  auto slot = [&](float*& p, unsigned count, bool sizing_always) {
    if (slab) p = det_slot(t, base, p, count);
    else if (p || sizing_always) t.stride += (count + 3u) & ~3u;
  };
  // sizing queries have no pointers: count every target the descriptor can have
  const bool sizing = slab == nullptr;
  if (n.enc == ENC_LINEAR) {
    slot(n.d_encW, (unsigned)(n.enc_out * n.din), sizing);
    slot(n.d_encb, (unsigned)n.enc_out, sizing);
  }
  for (int l = 0; l < n.n_layers; ++l) {
    slot(n.layer[l].dW, (unsigned)(n.layer[l].out_dim * n.layer[l].ld), sizing);
    slot(n.layer[l].db, (unsigned)n.layer[l].out_dim, sizing);
  }
  slot(n.dw_out, (unsigned)n.h_last, sizing);
  slot(n.db_out, 1u, sizing);
  slot(loss_sum, 1u, sizing);
  (void)dcoef;
  if (t.stride == 0) t.stride = 4;
}

static int check_orders(int nt, int nx) {
  if (nt < 0 || nt > 2) return fail(PINN_ERR_BAD_ORDER, "Temporal derivative order %d is not supported. Maximum order is 2.", nt);
  if (nx < 0 || nx > 4) return fail(PINN_ERR_BAD_ORDER, "Spatial derivative order %d is not supported. Maximum order is 4.", nx);
  return PINN_OK;
}

static bool stream_set_compiled(int nt, int nx) {
  static const int sets[][2] = {{0, 0}, {1, 0}, {1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 0}, {2, 2}};
  for (auto& s : sets)
    if (s[0] == nt && s[1] == nx) return true;
  return false;
}

static hipError_t dispatch_wide(int nt, int nx, const KernelArgs& a, bool bwd, int grid, hipStream_t st) {
#define PINN_CASE2(NT_, NX_) \
  if (nt == NT_ && nx == NX_) return launch_jetw_##NT_##_##NX_(a, bwd, grid, st);
#define PINN_CASE(NT_, NX_) PINN_CASE2(NT_, NX_)
#ifdef PINN_DEV /* make dev: one stream set */
  PINN_CASE(PINN_DEV_NT, PINN_DEV_NX)
#else
  PINN_CASE(0, 0) PINN_CASE(1, 0) PINN_CASE(1, 1) PINN_CASE(1, 2) PINN_CASE(1, 3) PINN_CASE(1, 4) PINN_CASE(2, 0) PINN_CASE(2, 2)
#endif
#undef PINN_CASE
#undef PINN_CASE2
  return hipErrorInvalidValue;
}

// the wide kernel runs this problem: program built, all K streams fit the LDS, engine not overridden
static bool use_wide(const PinnNetDesc* d, const float* const* w, float* const* g, int K, bool bwd, NetDev* n, int* misaligned = nullptr) {
  if (force_lm()) return false;
  if (!build_wide(d, w, g, n, misaligned)) return false;
  return jet_wide_fits(K, n->hmax, bwd, n->n_layers);
}

static int wide_grid(const NetDev& n, int K, long long N, bool bwd) {
  (void)n;
  (void)K;
  (void)bwd;
  const long long ntiles = (N + kT - 1) / kT;
  long long g = num_cus();
  if (g > ntiles) g = ntiles;
  return (int)(g < 1 ? 1 : g);
}

static int validate_table(const PinnNetDesc* net, const void* table, int num_tensors, const char* what) {
  if (!net) return fail(PINN_ERR_BAD_DESC, "null descriptor");
  const int want = lm::lm_expected_tensors(net);
  if (want < 0) return fail(PINN_ERR_UNSUPPORTED, "architecture id %d has no kernel", net->arch);
  if (!table) return fail(PINN_ERR_BAD_DESC, "%s table is null", what);
  if (num_tensors != want)
    return fail(PINN_ERR_BAD_DESC, "%s table has %d entries; the state_dict of this architecture has %d", what, num_tensors, want);
  return PINN_OK;
}

static int run(const PinnNetDesc* net, const float* const* weights, float* const* grads, int num_tensors,
               const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, int nt, int nx, int mode,
               float grad_scale, float* const* jets_out, const float* const* jets_bar, float* residual_out,
               float* loss_sum, void* workspace, size_t ws_bytes, bool bwd, void* stream, const float* res_bar = nullptr,
               float* coef_grads = nullptr) {
  int rc = validate_table(net, weights, num_tensors, "weights");
  if (rc) return rc;
  if (bwd && (rc = validate_table(net, grads, num_tensors, "weight_grads"))) return rc;
  if (N <= 0) return PINN_OK;
  if (!x && net->input_dim > 1) return fail(PINN_ERR_BAD_DESC, "x is null");
  if (!t) return fail(PINN_ERR_BAD_DESC, "t is null");
  if ((rc = check_orders(nt, nx))) return rc;
  if (!stream_set_compiled(nt, nx)) return fail(PINN_ERR_UNSUPPORTED, "stream set (nt=%d, nx=%d) is not compiled", nt, nx);
  char lerr[256] = "";
  if ((rc = lm::lm_check(net, lerr, sizeof(lerr)))) return fail(rc, "%s", lerr);
  const int K = 1 + nt + nx;
  PdeDev pd;
  memset(&pd, 0, sizeof(pd));
  if (pde) {
    pd.kind = pde->kind;
    pd.dimension = pde->dimension;
    pd.loss = pde->loss;
    pd.c0 = pde->coef[0];
    pd.c1 = pde->coef[1];
    pd.c2 = pde->coef[2];
    pd.c3 = pde->coef[3];
    pd.huber_delta = pde->huber_delta;
    pd.dcoef = bwd ? coef_grads : nullptr;
  }
  KernelArgs a;
  memset(&a, 0, sizeof(a));
  int misaligned = -1;
  // Coefficient cotangents (inverse problems) are a reduction of the layer-major engine's head kernel only: in the fused
  // tile-major kernel the extra per-tile code cost the headline configuration scratch (SGPR spills to memory) even with the
  // feature off, so such calls take the layer-major engine — which the caller selects (PINN_FLAG_LAYER_MAJOR) so that
  // pinn_workspace_bytes sizes the workspace for it.
  if (coef_grads && !(net->flags & PINN_FLAG_LAYER_MAJOR) && !force_lm() && use_wide(net, nullptr, nullptr, K, bwd, &a.net))
    return fail(PINN_ERR_UNSUPPORTED, "coefficient gradients run on the layer-major engine: set PINN_FLAG_LAYER_MAJOR in the "
                "descriptor for this call and for its pinn_workspace_bytes query");
  const bool wide = use_wide(net, weights, grads, K, bwd, &a.net, &misaligned);
  if (!wide && misaligned >= 0)  // pinn_workspace_bytes sized this descriptor for the fused kernel: say what is wrong instead of "workspace too small"
    return fail(PINN_ERR_MISALIGNED, "weight tensor %d is not 16-byte aligned: the fused kernel of this descriptor reads hidden-layer "
                "weights with 16-byte loads (pass aligned tensors, or set PINN_FLAG_LAYER_MAJOR to take the packing engine)", misaligned);
  if (wide) {
    const int grid = wide_grid(a.net, K, N, bwd);
    a.pde = pd;
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
#ifdef PINN_STAMPS
    a.stamps = g_stamps;
#endif
    // Default mode of a reverse launch: TWO-LEVEL flush.  Workgroup b adds (atomically) into row b mod 8 of an 8-row
    // slab and a small launch sums the rows into the caller's tensors: 32 instead of 256 workgroups contend for an
    // address.  Measured on the headline launch (kernel + memset + row sum, events around the call): direct atomics
    // 0.554 ms, 2 rows 0.553, 4 rows 0.549, 8 rows 0.543, 16 rows 0.545, 32 rows 0.550, 64 rows 0.567.
    // Networks whose MFMA layers all keep their weight-gradient tiles in registers (n_layers <= kPersist: nothing is
    // flushed inside the tile loop) take the STORE flush instead, in both modes: one slab row per workgroup, written
    // with plain stores, then wide_rows_reduce.  The last workgroups of a launch no longer spend 32 us adding 160 KB
    // at the memory-side atomic rate, and the slab needs no memset (jet_kernel_wide.h, end of the kernel).
    const int shared_rows = kFlushRows;
    const bool det_flag = (net->flags & PINN_FLAG_DETERMINISTIC) != 0;
    const bool store_flush = bwd && a.net.n_layers <= kPersist && wide_store_flush_on();
    const bool two_level = !det_flag && bwd && grid > shared_rows && !store_flush;
    const bool det = det_flag || two_level || store_flush;
    const int slab_rows = two_level ? shared_rows : grid;
    const size_t tape_floats = bwd ? (size_t)jet_tape_floats_per_wg(K, a.net.n_layers, 1) * grid : 0;
    size_t need = tape_floats * sizeof(float);
    DetTable dt;
    dt.n = 0;
    if (det) {
      NetDev probe = a.net;
      float* lprobe = a.loss_sum;
      float* cprobe = a.pde.dcoef;
      det_redirect(probe, lprobe, cprobe, nullptr, dt);  // sizing pass: the same stride pinn_workspace_bytes reports
      need += (size_t)dt.stride * slab_rows * sizeof(float);
    }
    if (need > 0 && (!workspace || ws_bytes < need))
      return fail(PINN_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    if (need > 0 && (reinterpret_cast<uintptr_t>(workspace) & 15)) return fail(PINN_ERR_MISALIGNED, "workspace is not 16-byte aligned");
    if (bwd) {
      a.tape_stride = jet_tape_floats_per_wg(K, a.net.n_layers, 1);
      a.tape = static_cast<float*>(workspace);
    }
    float* slab = nullptr;
    if (det) {
      const unsigned stride = dt.stride;
      slab = static_cast<float*>(workspace) + tape_floats;
      det_redirect(a.net, a.loss_sum, a.pde.dcoef, slab, dt);
      dt.stride = stride;  // rows are as wide as the sizing pass said, whatever subset of targets this call has
      a.det_stride = two_level ? -(long long)stride : (long long)stride;
      a.det_mask = slab_rows - 1;
      a.flush_store = store_flush ? 1 : 0;
      if (!store_flush) {
        const hipError_t em = hipMemsetAsync(slab, 0, (size_t)stride * slab_rows * sizeof(float), static_cast<hipStream_t>(stream));
        if (em != hipSuccess) return fail(PINN_ERR_HIP, "HIP error %d: %s", (int)em, hipGetErrorString(em));
      }
    }
    hipError_t e = dispatch_wide(nt, nx, a, bwd, grid, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(PINN_ERR_HIP, "HIP error %d: %s", (int)e, hipGetErrorString(e));
    if (det && dt.n > 0 && store_flush) {
      hipLaunchKernelGGL(wide_rows_reduce, dim3(136, dt.n), dim3(256), 0, static_cast<hipStream_t>(stream), dt, slab, slab_rows);
      e = hipGetLastError();
      if (e != hipSuccess) return fail(PINN_ERR_HIP, "HIP error %d: %s", (int)e, hipGetErrorString(e));
    } else if (det && dt.n > 0) {
      hipLaunchKernelGGL(wide_det_reduce, dim3(32, dt.n), dim3(256), 0, static_cast<hipStream_t>(stream), dt, slab, slab_rows);
      e = hipGetLastError();
      if (e != hipSuccess) return fail(PINN_ERR_HIP, "HIP error %d: %s", (int)e, hipGetErrorString(e));
    }
    return PINN_OK;
  }
  lm::CallArgs c;
  memset(&c, 0, sizeof(c));
  c.net = net;
  c.weights = weights;
  c.grads = grads;
  c.num_tensors = num_tensors;
  c.pde = pd;
  c.x = x;
  c.t = t;
  c.N = N;
  c.nt = nt;
  c.nx = nx;
  c.mode = mode;
  c.grad_scale = grad_scale;
  c.jets_out = jets_out;
  c.jets_bar = jets_bar;
  c.residual_out = residual_out;
  c.loss_sum = loss_sum;
  c.res_bar = res_bar;
  c.workspace = workspace;
  c.ws_bytes = ws_bytes;
  c.bwd = bwd;
  c.deterministic = (net->flags & PINN_FLAG_DETERMINISTIC) != 0;
  c.stream = static_cast<hipStream_t>(stream);
  rc = lm::lm_run(c, lerr, sizeof(lerr));
  if (rc) return fail(rc, "%s", lerr);
  return PINN_OK;
}

}  // namespace pinn

using namespace pinn;

extern "C" {

int pinn_abi_version(void) { return PINN_ABI_VERSION; }

const char* pinn_last_error(void) { return g_err; }

int pinn_num_tensors(const PinnNetDesc* net) {
  if (!net) return fail(PINN_ERR_BAD_DESC, "null descriptor");
  const int n = lm::lm_expected_tensors(net);
  if (n < 0) return fail(PINN_ERR_UNSUPPORTED, "architecture id %d has no kernel", net->arch);
  char lerr[256] = "";
  const int rc = lm::lm_check(net, lerr, sizeof(lerr));
  if (rc) return fail(rc, "%s", lerr);
  return n;
}

int pinn_pde_streams(const PinnPdeDesc* pde, int32_t* time_order, int32_t* space_order) {
  if (!pde || !time_order || !space_order) return fail(PINN_ERR_BAD_DESC, "null argument");
  int nt = 1, nx = 0;
  if (pde->dimension > 1 && pde->kind == PINN_PDE_CONVECTION)  // convection_equation.py:66-76 differentiates u w.r.t. a SLICE of x: torch raises
    return fail(PINN_ERR_UNSUPPORTED, "convection with dimension %d: the reference's residual raises for dimension > 1", pde->dimension);
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

size_t pinn_workspace_bytes(const PinnNetDesc* net, int64_t N, int32_t time_order, int32_t space_order, int32_t backward) {
  if (!net || N <= 0) return 0;
  if (time_order < 0 || time_order > 2 || space_order < 0 || space_order > 4) return 0;
  char lerr[64];
  if (lm::lm_check(net, lerr, sizeof(lerr)) != PINN_OK) return 0;
  const int K = 1 + time_order + space_order;
  const bool bwd = backward != 0;
  NetDev n;
  if (use_wide(net, nullptr, nullptr, K, bwd, &n)) {
    const size_t grid = (size_t)wide_grid(n, K, N, bwd);
    size_t bytes = bwd ? (size_t)jet_tape_floats_per_wg(K, n.n_layers, 1) * sizeof(float) * grid : 0;
    if ((net->flags & PINN_FLAG_DETERMINISTIC) || (bwd && n.n_layers <= kPersist && wide_store_flush_on())) {
      DetTable dt;
      float* lprobe = nullptr;
      float* cprobe = nullptr;
      det_redirect(n, lprobe, cprobe, nullptr, dt);
      bytes += (size_t)dt.stride * grid * sizeof(float);
    } else if (bwd && grid > (size_t)kFlushRows) {  // the two-level flush's shared rows
      DetTable dt;
      float* lprobe = nullptr;
      float* cprobe = nullptr;
      det_redirect(n, lprobe, cprobe, nullptr, dt);
      bytes += (size_t)dt.stride * kFlushRows * sizeof(float);
    }
    return bytes;
  }
  return lm::lm_workspace_bytes(net, N, time_order, space_order, bwd, (net->flags & PINN_FLAG_DETERMINISTIC) != 0);
}

int pinn_jet_forward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors, const float* x,
                     const float* t, int64_t N, int32_t time_order, int32_t space_order, float* const* jets_out,
                     void* workspace, size_t ws_bytes, void* stream) {
  if (!jets_out) return fail(PINN_ERR_BAD_DESC, "jets_out is null");
  return run(net, weights, nullptr, num_tensors, nullptr, x, t, N, time_order, space_order, MODE_JETS, 0.0f, jets_out,
             nullptr, nullptr, nullptr, workspace, ws_bytes, false, stream);
}

int pinn_jet_backward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors, const float* x,
                      const float* t, int64_t N, int32_t time_order, int32_t space_order,
                      const float* const* jet_cotangents, float* const* weight_grads, void* workspace, size_t ws_bytes,
                      void* stream) {
  if (!jet_cotangents || !weight_grads) return fail(PINN_ERR_BAD_DESC, "null cotangents or weight_grads");
  return run(net, weights, weight_grads, num_tensors, nullptr, x, t, N, time_order, space_order, MODE_JETS, 0.0f, nullptr,
             jet_cotangents, nullptr, nullptr, workspace, ws_bytes, true, stream);
}

int pinn_residual_forward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                          const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, float* residual_out,
                          float* loss_sum_out, void* workspace, size_t ws_bytes, void* stream) {
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, nullptr, num_tensors, pde, x, t, N, nt, nx, MODE_PDE, 0.0f, nullptr, nullptr, residual_out,
             loss_sum_out, workspace, ws_bytes, false, stream);
}

int pinn_residual_loss_grad(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                            const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, float grad_scale,
                            float* residual_out, float* loss_sum_out, float* const* weight_grads, void* workspace,
                            size_t ws_bytes, void* stream) {
  if (!weight_grads) return fail(PINN_ERR_BAD_DESC, "weight_grads is null");
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, weight_grads, num_tensors, pde, x, t, N, nt, nx, MODE_PDE, grad_scale, nullptr, nullptr,
             residual_out, loss_sum_out, workspace, ws_bytes, true, stream);
}

int pinn_residual_loss_grad_coef(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                                 const PinnPdeDesc* pde, const float* x, const float* t, int64_t N, float grad_scale,
                                 float* residual_out, float* loss_sum_out, float* const* weight_grads, float* coef_grads,
                                 void* workspace, size_t ws_bytes, void* stream) {
  if (!weight_grads) return fail(PINN_ERR_BAD_DESC, "weight_grads is null");
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, weight_grads, num_tensors, pde, x, t, N, nt, nx, MODE_PDE, grad_scale, nullptr, nullptr,
             residual_out, loss_sum_out, workspace, ws_bytes, true, stream, nullptr, coef_grads);
}

int pinn_residual_backward(const PinnNetDesc* net, const float* const* weights, int32_t num_tensors,
                           const PinnPdeDesc* pde, const float* x, const float* t, int64_t N,
                           const float* residual_cotangent, float* const* weight_grads, void* workspace,
                           size_t ws_bytes, void* stream) {
  if (!weight_grads || !residual_cotangent) return fail(PINN_ERR_BAD_DESC, "null cotangent or weight_grads");
  int32_t nt, nx;
  int rc = pinn_pde_streams(pde, &nt, &nx);
  if (rc) return rc;
  return run(net, weights, weight_grads, num_tensors, pde, x, t, N, nt, nx, MODE_PDE, 0.0f, nullptr, nullptr, nullptr,
             nullptr, workspace, ws_bytes, true, stream, residual_cotangent);
}

#ifdef PINN_STAMPS
/* Diagnostic builds only (make dev STAMPS=1): device buffer of grid*4*16 uint64 for the in-kernel phase timers. */
void pinn_debug_set_stamps(void* device_buffer) { g_stamps = static_cast<unsigned long long*>(device_buffer); }
#endif

}  // extern "C"
