// Host side of the layer-major engine: PinnNetDesc -> node program -> launch list (see lm_common.h).
// tests/jet_model.py::net_program / program_forward / program_backward is the executable specification of this file.
#include "lm_engine.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <vector>

#include "lm_fused.h"
#include "lm_gemm.h"

namespace pinn {
namespace lm {

namespace {

__global__ void lm_pack_kernel(const PackTable tab, float* packed) {
  const PackItem it = tab.item[blockIdx.y];
  if (!it.src) return;
  const int total = it.rows_p * it.cols_p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / it.cols_p, c = i - r * it.cols_p;
    float v = 0.0f;
    long long dst = i;
    if (it.transpose == 1 || it.transpose == 3 || it.transpose == 5) {
      if (c < it.rows && r < it.cols) v = it.src[c * it.cols + r];
    } else {
      if (r < it.rows && c < it.cols) v = it.src[r * it.cols + c];
    }
    if (it.transpose >= 4) dst = frag16_index(r, c, it.cols_p >> 4);
    else if (it.transpose >= 2) dst = frag_index(r, c, it.cols_p >> 5);
    packed[it.off + dst] = v;
  }
}

// user_grad += packed_grad (logical window only)
__global__ void lm_unpack_kernel(const PackTable tab, const float* packed_grad) {
  const PackItem it = tab.item[blockIdx.y];
  if (!it.user_grad) return;
  const int total = it.rows * it.cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / it.cols, c = i - r * it.cols;
    const float g = it.transpose ? packed_grad[it.off + c * it.cols_p + r] : packed_grad[it.off + r * it.cols_p + c];
    it.user_grad[i] += g;
  }
}

// Attention with sequence length 1: softmax == 1, so the attention branch is W_p (W_v h + b_v) + b_p.  The engine runs
// it as ONE GEMM with W_pv = W_p W_v, b_pv = W_p b_v + b_p (computed here per call from the packed, zero-padded
// H_p x H_p parameters) and maps the merged gradient back afterwards:
//   dW_p = G W_v^T + g b_v^T,  dW_v = W_p^T G,  db_v = W_p^T g,  db_p = g        (G = dL/dW_pv, g = dL/db_pv)
// One thread per output element, fixed summation order: deterministic.  Two of the twelve GEMM launches per layer and
// one record round trip go away (C5: 113 -> 105 ms).
struct MergeItem {
  unsigned wp, bp, wv, bv, wm, bm;  // offsets in the packed block
  int Hp;
};
struct MergeTable {
  int n;
  MergeItem item[kMaxNodes / 3 + 1];
};

__global__ void lm_merge_pv_kernel(const MergeTable tab, float* packed) {
  const MergeItem it = tab.item[blockIdx.y];
  const int Hp = it.Hp;
  const float* Wp = packed + it.wp;
  const float* Wv = packed + it.wv;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Hp * Hp; idx += gridDim.x * blockDim.x) {
    const int i = idx / Hp, j = idx - i * Hp;
    float s = 0.0f;
    for (int k = 0; k < Hp; ++k) s = fmaf(Wp[i * Hp + k], Wv[k * Hp + j], s);
    packed[it.wm + idx] = s;
    if (j == 0) {
      float b = packed[it.bp + i];
      for (int k = 0; k < Hp; ++k) b = fmaf(Wp[i * Hp + k], packed[it.bv + k], b);
      packed[it.bm + i] = b;
    }
  }
}

__global__ void lm_unmerge_pv_kernel(const MergeTable tab, const float* packed, float* grads) {
  const MergeItem it = tab.item[blockIdx.y];
  const int Hp = it.Hp;
  const float* Wp = packed + it.wp;
  const float* Wv = packed + it.wv;
  const float* G = grads + it.wm;
  const float* g = grads + it.bm;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < Hp * Hp; idx += gridDim.x * blockDim.x) {
    const int a = idx / Hp, b = idx - a * Hp;
    float sp = g[a] * packed[it.bv + b], sv = 0.0f;
    for (int k = 0; k < Hp; ++k) {
      sp = fmaf(G[a * Hp + k], Wv[b * Hp + k], sp);  // dW_p[a][b] = sum_j G[a][j] W_v[b][j] + g[a] b_v[b]
      sv = fmaf(Wp[k * Hp + a], G[k * Hp + b], sv);  // dW_v[a][b] = sum_i W_p[i][a] G[i][b]
    }
    grads[it.wp + idx] += sp;
    grads[it.wv + idx] += sv;
    if (b == 0) {
      float sb = 0.0f;
      for (int k = 0; k < Hp; ++k) sb = fmaf(Wp[k * Hp + a], g[k], sb);  // db_v[a] = sum_i W_p[i][a] g[i]
      grads[it.bv + a] += sb;
      grads[it.bp + a] += g[a];
    }
  }
}

// deterministic mode: dst_k[f * mul_k + add_k] += sum over workgroups (fixed order) of partial[b][slot_k][f]
struct SlotReduce {
  int n;               // destinations
  float* dst[8];
  int slot[8], mul[8], add[8], len[8];
  int row;             // floats per workgroup row of the partial buffer
};
__global__ void lm_reduce_slots(const float* partial, int nblocks, SlotReduce r) {
  for (int k = 0; k < r.n; ++k) {
    if (!r.dst[k]) continue;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < r.len[k]; f += gridDim.x * blockDim.x) {
      float s = 0.0f;
      for (int b = 0; b < nblocks; ++b) s += partial[(long long)b * r.row + r.slot[k] + f];
      r.dst[k][f * r.mul[k] + r.add[k]] += s;
    }
  }
}

int failf(char* err, size_t n, int code, const char* fmt, ...) {
  if (err && n) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, n, fmt, ap);
    va_end(ap);
  }
  return code;
}

int num_cus() {
  static int cached = 0;
  if (cached) return cached;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      cus > 0) {
    cached = cus;
    return cus;
  }
  (void)hipGetLastError();
  return 256;
}

// Bytes of one record per chunk the sizing aims at; PINN_LM_RECORD_MB overrides it, read once.  Measured on MI355X
// (tools/bench_configs.py, C3 / C4 / C5): bigger is faster until every launch has >= ~25 units of work per workgroup —
// 96 MB 41.5 / 46.0 / 177 ms, 512 MB 33.8 / 36.8 / 143 ms, then flat for C3 / C4 and 138 -> 134 -> 132 -> 131 ms for C5
// at 512 / 1024 / 2048 / 4096 MB: what counts is full waves of work per launch and few one-off weight-gradient
// flushes, not Infinity Cache residency of the records.  1 GB keeps a 10^6-point batch of the widest supported
// network within ~60 GB of the 288.
size_t record_target_bytes() {
  static size_t v = 0;
  if (!v) {
    const char* e = getenv("PINN_LM_RECORD_MB");
    const long mb = e ? atol(e) : 0;
    v = (size_t)(mb > 0 ? mb : 1024) << 20;
  }
  return v;
}

float act_param_of(int act, float user) {
  switch (act) {
    case PINN_ACT_SIN: return user;
    case PINN_ACT_RELU: return 0.0f;
    case PINN_ACT_LEAKY_RELU: return 0.01f;
    case PINN_ACT_IDENTITY: return 1.0f;
    default: return 0.0f;
  }
}

// ----------------------------------------------------------------------------------------------------------------
// program
// ----------------------------------------------------------------------------------------------------------------
struct Prologue {
  int src_kind = SRC_REC;
  int src_node = -1;   // SRC_REC: node whose Y record is the source
  int enc_w = -1, enc_b = -1;  // tensor indices: first Linear (COORDS_LINEAR) or Fourier B (COORDS_FOURIER)
  int ln_g = -1, ln_b = -1;    // tensor indices of the LayerNorm applied to the source
  int skip_node = -1;  // node whose V record is added before the activation
  int act = -1;        // PinnAct or -1
  float act_param = 0.0f;
  int H = 0;           // features of this prologue (= in-features of the GEMM that follows)
  int M = 0;           // Fourier mapping size
  bool identity() const { return src_kind == SRC_REC && ln_g < 0 && skip_node < 0 && act < 0; }
};

struct Node {
  Prologue pro;
  int w = -1, b = -1;  // tensor indices of the Linear
  int Hin = 0, Hout = 0;
  int add_node = -1;   // node whose V record is added to Y
};

struct Program {
  int din = 0;
  int n_nodes = 0;
  Node node[kMaxNodes];
  Prologue head;
  int w_out = -1, b_out = -1;
  int n_tensors = 0;
  int rows[kMaxPack], cols[kMaxPack];  // logical shapes of the tensors the program uses (0 rows = unused)
  bool transpose[kMaxPack];
  bool enc_cols4[kMaxPack];            // (H x din) first-Linear weights are packed with 4 columns
  bool matrix[kMaxPack];               // a GEMM weight: its packed rows are padded to 32 even when it has ONE row (a 1-wide
                                       // layer); vectors (biases, LayerNorm parameters, the output layer) stay 1 x cols_p
  float ln_eps = 1e-5f;
  // merged attention branches (see lm_merge_pv_kernel): derived tensors live behind the fragment copies in the pack
  // table, at index derived_base + 2 j (weight) and + 1 (bias)
  int n_derived = 0, derived_base = 0;
  struct Derived {
    int wp, bp, wv, bv, H;
  } derived[kMaxNodes / 3 + 1];
  // fused GEMM + prologue launches (lm_fused.h), planned per call by plan_fusion()
  struct Fuse {
    bool on;
    int nch, rt, gy;
  } fuse_fwd[kMaxNodes], fuse_bwd[kMaxNodes];
};

void use_tensor(Program& P, int idx, int rows, int cols, bool enc4 = false, bool transpose = false, bool matrix = false) {
  P.matrix[idx] = matrix;
  P.rows[idx] = rows;
  P.cols[idx] = cols;
  P.enc_cols4[idx] = enc4;
  P.transpose[idx] = transpose;
}

int expected_tensors(const PinnNetDesc* d) {
  switch (d->arch) {
    case PINN_ARCH_FOURIER: return 1 + 2 * d->num_linear;
    case PINN_ARCH_FEEDFORWARD: return (d->flags & PINN_FLAG_LAYER_NORM) ? 4 * (d->num_linear - 1) + 2 : 2 * d->num_linear;
    case PINN_ARCH_SIREN: return 2 * d->num_linear;
    case PINN_ARCH_RESNET: return 4 + 8 * d->num_blocks;
    case PINN_ARCH_ATTENTION: return 4 + 16 * d->num_blocks;
    default: return -1;
  }
}

int build_program(const PinnNetDesc* d, Program& P, char* err, size_t en) {
  memset(&P, 0, sizeof(P));
  for (int i = 0; i < kMaxPack; ++i) P.rows[i] = 0;
  if (!d) return failf(err, en, PINN_ERR_BAD_DESC, "null descriptor");
  if (d->num_linear < 2 || d->num_linear > PINN_MAX_LINEAR) return failf(err, en, PINN_ERR_BAD_DESC, "num_linear=%d outside [2,%d]", d->num_linear, PINN_MAX_LINEAR);
  if (d->input_dim < 1 || d->input_dim > 4) return failf(err, en, PINN_ERR_UNSUPPORTED, "input_dim=%d (max 4)", d->input_dim);
  if (d->widths[d->num_linear - 1] != 1) return failf(err, en, PINN_ERR_UNSUPPORTED, "output_dim must be 1");
  if (d->activation < PINN_ACT_TANH || d->activation > PINN_ACT_IDENTITY) return failf(err, en, PINN_ERR_BAD_DESC, "unknown activation id %d", d->activation);
  P.din = d->input_dim;
  P.n_tensors = expected_tensors(d);
  if (P.n_tensors < 0) return failf(err, en, PINN_ERR_UNSUPPORTED, "architecture id %d has no kernel", d->arch);
  if (P.n_tensors + 2 * kMaxNodes > kMaxPack) return failf(err, en, PINN_ERR_UNSUPPORTED, "%d tensors exceed the pack table (%d)", P.n_tensors, kMaxPack - 2 * kMaxNodes);
  P.ln_eps = d->ln_eps > 0.0f ? d->ln_eps : 1e-5f;
  const int act = d->arch == PINN_ARCH_SIREN ? PINN_ACT_SIN : d->activation;
  const float par = act_param_of(act, d->act_param);
  auto check_w = [&](int wd) { return wd >= 1 && wd <= 1024; };
  auto add_node = [&](const Prologue& pro, int w, int b, int Hin, int Hout, int add) -> int {
    Node& nd = P.node[P.n_nodes];
    nd.pro = pro;
    nd.pro.H = Hin;
    nd.w = w;
    nd.b = b;
    nd.Hin = Hin;
    nd.Hout = Hout;
    nd.add_node = add;
    use_tensor(P, w, Hout, Hin, false, false, true);
    use_tensor(P, b, 1, Hout);
    return P.n_nodes++;
  };
  if (d->arch == PINN_ARCH_FOURIER) {
    const int M = d->mapping_size;
    if (M < 1 || 2 * M > 1024) return failf(err, en, PINN_ERR_UNSUPPORTED, "fourier mapping_size=%d", M);
    use_tensor(P, 0, d->input_dim, M, false, true);
    Prologue pro;
    pro.src_kind = SRC_COORDS_FOURIER;
    pro.enc_w = 0;
    pro.M = M;
    int prev = 2 * M;
    for (int i = 0; i < d->num_linear - 1; ++i) {
      const int wd = d->widths[i];
      if (!check_w(wd)) return failf(err, en, PINN_ERR_UNSUPPORTED, "hidden width %d outside [1,1024]", wd);
      if (P.n_nodes >= kMaxNodes) return failf(err, en, PINN_ERR_UNSUPPORTED, "too many layers");
      const int m = add_node(pro, 1 + 2 * i, 2 + 2 * i, prev, wd, -1);
      pro = Prologue();
      pro.src_node = m;
      pro.act = act;
      pro.act_param = par;
      prev = wd;
    }
    P.head = pro;
    P.head.H = prev;
    P.w_out = 1 + 2 * (d->num_linear - 1);
    P.b_out = P.w_out + 1;
  } else if (d->arch == PINN_ARCH_FEEDFORWARD || d->arch == PINN_ARCH_SIREN) {
    const bool ln = d->arch == PINN_ARCH_FEEDFORWARD && (d->flags & PINN_FLAG_LAYER_NORM);
    const int step = ln ? 4 : 2;
    const int nh = d->num_linear - 1;  // hidden Linears
    if (!check_w(d->widths[0])) return failf(err, en, PINN_ERR_UNSUPPORTED, "hidden width %d outside [1,1024]", d->widths[0]);
    use_tensor(P, 0, d->widths[0], d->input_dim, true);
    use_tensor(P, 1, 1, d->widths[0]);
    Prologue pro;
    pro.src_kind = SRC_COORDS_LINEAR;
    pro.enc_w = 0;
    pro.enc_b = 1;
    pro.act = act;
    pro.act_param = par;
    if (ln) {
      pro.ln_g = 2;
      pro.ln_b = 3;
      use_tensor(P, 2, 1, d->widths[0]);
      use_tensor(P, 3, 1, d->widths[0]);
    }
    int prev = d->widths[0];
    for (int i = 1; i < nh; ++i) {
      const int wd = d->widths[i];
      if (!check_w(wd)) return failf(err, en, PINN_ERR_UNSUPPORTED, "hidden width %d outside [1,1024]", wd);
      if (P.n_nodes >= kMaxNodes) return failf(err, en, PINN_ERR_UNSUPPORTED, "too many layers");
      const int m = add_node(pro, step * i, step * i + 1, prev, wd, -1);
      pro = Prologue();
      pro.src_node = m;
      pro.act = act;
      pro.act_param = par;
      if (ln) {
        pro.ln_g = step * i + 2;
        pro.ln_b = step * i + 3;
        use_tensor(P, pro.ln_g, 1, wd);
        use_tensor(P, pro.ln_b, 1, wd);
      }
      prev = wd;
    }
    P.head = pro;
    P.head.H = prev;
    P.w_out = step * nh;
    P.b_out = P.w_out + 1;
  } else if (d->arch == PINN_ARCH_RESNET) {
    const int nb = d->num_blocks, H = d->widths[0];
    if (nb < 1 || 2 * nb > kMaxNodes || d->num_linear != 2 * nb + 2) return failf(err, en, PINN_ERR_BAD_DESC, "resnet: num_blocks=%d / num_linear=%d", nb, d->num_linear);
    if (!check_w(H)) return failf(err, en, PINN_ERR_UNSUPPORTED, "resnet width %d outside [1,1024]", H);
    use_tensor(P, 0, H, d->input_dim, true);
    use_tensor(P, 1, 1, H);
    Prologue pro;
    pro.src_kind = SRC_COORDS_LINEAR;
    pro.enc_w = 0;
    pro.enc_b = 1;
    pro.act = act;
    pro.act_param = par;
    for (int b = 0; b < nb; ++b) {
      const int base = 2 + 8 * b;
      const int n1 = add_node(pro, base, base + 1, H, H, -1);
      Prologue p2;
      p2.src_node = n1;
      p2.ln_g = base + 2;
      p2.ln_b = base + 3;
      p2.act = act;
      p2.act_param = par;
      use_tensor(P, base + 2, 1, H);
      use_tensor(P, base + 3, 1, H);
      const int n2 = add_node(p2, base + 4, base + 5, H, H, -1);
      pro = Prologue();  // q_b = LN2(z2_b) + V(n1_b);  h_b = act(q_b)
      pro.src_node = n2;
      pro.ln_g = base + 6;
      pro.ln_b = base + 7;
      pro.skip_node = n1;
      pro.act = act;
      pro.act_param = par;
      use_tensor(P, base + 6, 1, H);
      use_tensor(P, base + 7, 1, H);
    }
    P.head = pro;
    P.head.H = H;
    P.w_out = 2 + 8 * nb;
    P.b_out = P.w_out + 1;
  } else if (d->arch == PINN_ARCH_ATTENTION) {
    const int nl = d->num_blocks, H = d->widths[0];
    if (nl < 1 || 3 * nl > kMaxNodes || d->num_linear != 2) return failf(err, en, PINN_ERR_BAD_DESC, "attention: num_blocks=%d / num_linear=%d", nl, d->num_linear);
    if (!check_w(H) || 4 * H > 1024) return failf(err, en, PINN_ERR_UNSUPPORTED, "attention width %d outside [1,256]", H);
    use_tensor(P, 0, H, d->input_dim, true);
    use_tensor(P, 1, 1, H);
    Prologue pro;
    pro.src_kind = SRC_COORDS_LINEAR;
    pro.enc_w = 0;
    pro.enc_b = 1;
    pro.act = act;
    pro.act_param = par;
    P.derived_base = P.n_tensors + 2 * (3 * nl);  // three GEMM nodes per layer, two fragment copies each
    if (P.derived_base + 2 * nl > kMaxPack) return failf(err, en, PINN_ERR_UNSUPPORTED, "attention: %d layers exceed the pack table", nl);
    for (int l = 0; l < nl; ++l) {
      const int base = 2 + 16 * l;  // q(0,1) k(2,3) value(4,5) proj(6,7) LN_a(8,9) net.0(10,11) net.3(12,13) LN_f(14,15)
      // value and projection as ONE GEMM: za = (W_p W_v) h + (W_p b_v + b_p) + h
      use_tensor(P, base + 4, H, H, false, false, true);  // lm_merge_pv_kernel reads both as H_p x H_p
      use_tensor(P, base + 5, 1, H);
      use_tensor(P, base + 6, H, H, false, false, true);
      use_tensor(P, base + 7, 1, H);
      Program::Derived& dv = P.derived[P.n_derived];
      dv.wv = base + 4;
      dv.bv = base + 5;
      dv.wp = base + 6;
      dv.bp = base + 7;
      dv.H = H;
      const int wm = P.derived_base + 2 * P.n_derived, bm = wm + 1;
      ++P.n_derived;
      const int np = add_node(pro, wm, bm, H, H, P.n_nodes);  // add_node = itself: + its own input record h
      Prologue p1;
      p1.src_node = np;
      p1.ln_g = base + 8;
      p1.ln_b = base + 9;
      use_tensor(P, base + 8, 1, H);
      use_tensor(P, base + 9, 1, H);
      const int n1 = add_node(p1, base + 10, base + 11, H, 4 * H, -1);
      Prologue p2;
      p2.src_node = n1;
      p2.act = PINN_ACT_GELU;
      const int n2 = add_node(p2, base + 12, base + 13, 4 * H, H, n1);  // zf = W_2 gelu(z1) + b_2 + h1
      pro = Prologue();
      pro.src_node = n2;
      pro.ln_g = base + 14;
      pro.ln_b = base + 15;
      use_tensor(P, base + 14, 1, H);
      use_tensor(P, base + 15, 1, H);
    }
    P.head = pro;
    P.head.H = H;
    P.w_out = 2 + 16 * nl;
    P.b_out = P.w_out + 1;
  } else {
    return failf(err, en, PINN_ERR_UNSUPPORTED, "architecture id %d has no kernel", d->arch);
  }
  use_tensor(P, P.w_out, 1, P.head.H);
  use_tensor(P, P.b_out, 1, 1);
  return PINN_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// fusion plan: which GEMM nodes run the consumer's prologue (forward) / their own prologue's adjoint (reverse) in
// their epilogue (lm_fused.h).  PINN_LM_FUSED=0 keeps every node on the unfused kernels (experiments, second
// implementation for the parity tests), read once.
// ----------------------------------------------------------------------------------------------------------------
bool fused_off() {
  static const bool v = [] {
    const char* e = getenv("PINN_LM_FUSED");
    return e && atoi(e) == 0;
  }();
  return v;
}

bool fused_set_built(int nt, int nx) { return !(nt == 2); }  // time order 2 (wave, pendulum) stays unfused: see the Makefile

// Which node kinds take the fused kernels.  Measured on MI355X (tools/micro/fused_bench.hip, tools/bench_configs.py): without
// LayerNorm the fused launch beats GEMM + element-wise launch in both directions (C4 30.8 -> 26.1 ms); with LayerNorm the
// forward launch wins (C3 25.2 -> 23.4 ms with bits 1 | 2 | 4), and the reverse launch — whose epilogue (two block
// reductions, ~1 200 VALU instructions per wave and unit) runs with all eight waves in the same phase and nothing
// overlapping it — only where the weight slice leaves the epilogue its registers (depth 128; at depth 256 it spills
// ~80 VGPRs and loses to GEMM + lm_ew_bwd_dma: 25.3 ms with all bits).  PINN_LM_FUSED_LN=<bits> overrides: 1 forward,
// 2 forward with a skip record, 4 reverse at depth 128, 8 reverse at depth 256.
int fused_ln_mask() {
  static const int v = [] {
    const char* e = getenv("PINN_LM_FUSED_LN");
    return e ? atoi(e) : (1 | 2 | 4);
  }();
  return v;
}

// Unfused GEMMs of reduction depth 384 / 512 take lm_gemm_wres16, whose weight operand is in frag16 order as well
// (PINN_LM_WRES16=0: lm_gemm, experiments), read once.
bool wres16_shape(int depth_p) {
  static const bool off = [] {
    const char* e = getenv("PINN_LM_WRES16");
    const char* w = getenv("PINN_LM_WRES");
    return (e && atoi(e) == 0) || (w && atoi(w) == 0);
  }();
  return !off && (depth_p == 384 || depth_p == 512);
}

bool fuse_shape(int rows, int depth, bool ln, Program::Fuse& f) {
  const int rp = round32(rows), dp = round32(depth);
  f.on = false;
  if (dp != 128 && dp != 256) return false;
  if (rp == 128 && dp == 128) {
    f = {true, 4, 4, 1};
    return true;
  }
  if (rp % 256 == 0 && (!ln || rp == 256)) {
    f = {true, dp / 32, 8, rp / 256};
    return true;
  }
  return false;
}

const Prologue& consumer_of(const Program& P, int m) { return m + 1 < P.n_nodes ? P.node[m + 1].pro : P.head; }

void plan_fusion(Program& P, int nt, int nx) {
  for (int m = 0; m < P.n_nodes; ++m) {
    P.fuse_fwd[m].on = P.fuse_bwd[m].on = false;
    if (fused_off() || !fused_set_built(nt, nx)) continue;
    const Node& nd = P.node[m];
    const Prologue& cp = consumer_of(P, m);
    const int lnm = fused_ln_mask();
    if (cp.src_kind == SRC_REC && cp.src_node == m && !cp.identity() && cp.H == nd.Hout && !(nd.add_node >= 0 && cp.skip_node >= 0) &&
        (cp.ln_g < 0 || (lnm & (cp.skip_node >= 0 ? 2 : 1))))
      fuse_shape(nd.Hout, nd.Hin, cp.ln_g >= 0, P.fuse_fwd[m]);
    const Prologue& pro = nd.pro;
    if (pro.src_kind == SRC_REC && !pro.identity() && (pro.ln_g < 0 || (lnm & (round32(nd.Hout) <= 128 ? 4 : 8))))
      fuse_shape(nd.Hin, nd.Hout, pro.ln_g >= 0, P.fuse_bwd[m]);
  }
}

#define PINN_LM_FDECL(nt, nx, act) \
  hipError_t launch_lm_fused_##nt##_##nx##_##act(const FusedArgs&, bool bwd, bool ln, int nch, int rt, int gx, int gy, hipStream_t);
#define PINN_LM_FDECL5(nt, nx) PINN_LM_FDECL(nt, nx, 0) PINN_LM_FDECL(nt, nx, 1) PINN_LM_FDECL(nt, nx, 2) PINN_LM_FDECL(nt, nx, 3) PINN_LM_FDECL(nt, nx, 4)
}  // namespace
PINN_LM_FDECL5(0, 0) PINN_LM_FDECL5(1, 0) PINN_LM_FDECL5(1, 1) PINN_LM_FDECL5(1, 2) PINN_LM_FDECL5(1, 3) PINN_LM_FDECL5(1, 4)
#undef PINN_LM_FDECL5
#undef PINN_LM_FDECL
namespace {

hipError_t launch_fused(int nt, int nx, int act, const FusedArgs& a, bool bwd, bool ln, const Program::Fuse& f, int gx, hipStream_t st) {
  const int fam = (act == PINN_ACT_TANH || act == PINN_ACT_SIN || act == PINN_ACT_GELU || act == PINN_ACT_SIGMOID) ? act : PINN_ACT_RELU;
#define PINN_LM_FCASE(NT_, NX_, A_) \
  if (nt == NT_ && nx == NX_ && fam == A_) return launch_lm_fused_##NT_##_##NX_##_##A_(a, bwd, ln, f.nch, f.rt, gx, f.gy, st);
#define PINN_LM_FCASE5(NT_, NX_) PINN_LM_FCASE(NT_, NX_, 0) PINN_LM_FCASE(NT_, NX_, 1) PINN_LM_FCASE(NT_, NX_, 2) PINN_LM_FCASE(NT_, NX_, 3) PINN_LM_FCASE(NT_, NX_, 4)
  PINN_LM_FCASE5(0, 0) PINN_LM_FCASE5(1, 0) PINN_LM_FCASE5(1, 1) PINN_LM_FCASE5(1, 2) PINN_LM_FCASE5(1, 3) PINN_LM_FCASE5(1, 4)
#undef PINN_LM_FCASE5
#undef PINN_LM_FCASE
  return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------------------------------------------
// workspace layout
// ----------------------------------------------------------------------------------------------------------------
struct Layout {
  PackTable tab;
  size_t n_packed = 0;       // floats of one packed block (parameters; gradients have the same layout)
  long long ct = 0;          // tiles per chunk
  int K = 0;
  // offsets in floats from the workspace start
  size_t params = 0, grads = 0, U = 0;
  size_t V[kMaxNodes], Y[kMaxNodes], Vh = 0;
  size_t Zbar[kMaxNodes], Vbar[kMaxNodes], Pbar[kMaxNodes + 1];  // Pbar[m]: skip cotangent produced by node m's prologue (kMaxNodes = head)
  size_t Stats[kMaxNodes + 1];  // LayerNorm prologues: per-point sums kept by the forward launch ([tile][2 K][32]); 0 = none
  size_t partial = 0, partial_floats = 0, det = 0;
  size_t total = 0;          // floats
};

size_t rec_floats(long long ct, int K, int H) { return (size_t)ct * K * round32(H) * kT; }

void make_layout(const Program& P, long long N, int K, bool bwd, bool deterministic, Layout& L) {
  memset(&L.tab, 0, sizeof(L.tab));
  L.K = K;
  size_t off = 0;
  L.tab.n = P.n_tensors;
  for (int i = 0; i < P.n_tensors; ++i) {
    PackItem& it = L.tab.item[i];
    it.src = nullptr;
    it.user_grad = nullptr;
    it.off = (unsigned)off;
    it.rows = P.rows[i];
    it.cols = P.cols[i];
    it.transpose = P.transpose[i] ? 1 : 0;
    if (P.rows[i] == 0) continue;
    if (P.transpose[i]) {  // Fourier B (din x M) -> [round32(M)][4]
      it.rows_p = round32(P.cols[i]);
      it.cols_p = 4;
    } else if (P.enc_cols4[i]) {  // first Linear (H x din) -> [round32(H)][4]
      it.rows_p = round32(P.rows[i]);
      it.cols_p = 4;
    } else {
      // A 1 x H GEMM weight (a layer of width 1) gets its 32 padded rows like any other: the weight-gradient kernels own the
      // whole z_rows x v_rows block behind dW.  Packed as a vector, the block ran over the NEXT items of the gradient
      // twin; harmless with float atomics (+= 0), but the deterministic reduction's plain "+= 0" on those addresses raced
      // with the "+= db" of the bias that lives there and lost it about half the time (tools/fuzz_parity.py).
      it.rows_p = (P.rows[i] == 1 && !P.matrix[i]) ? 1 : round32(P.rows[i]);
      it.cols_p = round32(P.cols[i]);
    }
    off += (size_t)it.rows_p * it.cols_p;
  }
  // a transposed copy of every GEMM weight: Vbar = W^T Zbar then runs as the rows form too (16-byte loads along the
  // reduction axis instead of 16 dword loads per 32-deep chunk: 154 -> 116 us per launch at width 256)
  for (int m = 0; m < P.n_nodes; ++m) {
    for (int tr = 0; tr < 2; ++tr) {  // [0]: W in fragment order (forward GEMM), [1]: W^T in fragment order (reverse GEMM)
      PackItem& it = L.tab.item[P.n_tensors + 2 * m + tr];
      it.src = nullptr;
      it.user_grad = nullptr;
      it.off = (unsigned)off;
      it.rows = P.node[m].Hout;
      it.cols = P.node[m].Hin;
      it.rows_p = round32(tr ? P.node[m].Hin : P.node[m].Hout);
      it.cols_p = round32(tr ? P.node[m].Hout : P.node[m].Hin);
      // fused kernels and lm_gemm_wres16: 16 x 16 x 4 operand order
      it.transpose = (((tr ? P.fuse_bwd[m].on : P.fuse_fwd[m].on) || wres16_shape(it.cols_p)) ? 4 : 2) + tr;
      off += (size_t)it.rows_p * it.cols_p;
    }
  }
  L.tab.n = P.n_tensors + 2 * P.n_nodes;
  for (int j = 0; j < P.n_derived; ++j) {  // merged attention weights / biases: filled by lm_merge_pv_kernel, no user tensor
    const int Hp = round32(P.derived[j].H);
    for (int t = 0; t < 2; ++t) {
      PackItem& it = L.tab.item[P.derived_base + 2 * j + t];
      it.src = nullptr;
      it.user_grad = nullptr;
      it.off = (unsigned)off;
      it.rows = t ? 1 : Hp;
      it.cols = Hp;
      it.rows_p = it.rows;
      it.cols_p = Hp;
      it.transpose = 0;
      off += (size_t)it.rows_p * it.cols_p;
    }
    L.tab.n = P.derived_base + 2 * P.n_derived;
  }
  L.n_packed = off;
  // chunk size: the widest record of a chunk stays near the target
  int hmax = P.head.H;
  for (int m = 0; m < P.n_nodes; ++m) {
    if (P.node[m].Hin > hmax) hmax = P.node[m].Hin;
    if (P.node[m].Hout > hmax) hmax = P.node[m].Hout;
  }
  const long long ntiles = (N + kT - 1) / kT;
  long long ct = (long long)(record_target_bytes() / ((size_t)K * round32(hmax) * kT * sizeof(float)));
  if (ct < 64) ct = 64;
  if (ct > ntiles) ct = ntiles;
  L.ct = ct;
  size_t o = 0;
  L.params = o;
  o += L.n_packed;
  L.grads = o;
  if (bwd) o += L.n_packed;
  L.U = o;
  o += (size_t)ct * K * kT;
  for (int m = 0; m < P.n_nodes; ++m) {
    const Node& nd = P.node[m];
    if (nd.pro.identity()) {
      L.V[m] = L.Y[nd.pro.src_node];
    } else {
      L.V[m] = o;
      o += rec_floats(ct, K, nd.Hin);
    }
    L.Y[m] = o;
    o += rec_floats(ct, K, nd.Hout);
  }
  L.Vh = o;
  o += rec_floats(ct, K, P.head.H);
  for (int m = 0; m <= kMaxNodes; ++m) L.Stats[m] = 0;
  if (bwd) {
    o = (o + 3) & ~(size_t)3;
    for (int m = 0; m < P.n_nodes; ++m)
      if (P.node[m].pro.ln_g >= 0) {
        L.Stats[m] = o;
        o += (size_t)ct * 2 * K * kT;
      }
    if (P.head.ln_g >= 0) {
      L.Stats[kMaxNodes] = o;
      o += (size_t)ct * 2 * K * kT;
    }
    for (int m = 0; m < P.n_nodes; ++m) {
      L.Zbar[m] = o;
      o += rec_floats(ct, K, P.node[m].Hout);
    }
    for (int m = 0; m < P.n_nodes; ++m) {
      const Node& nd = P.node[m];
      if (nd.pro.identity()) {
        L.Vbar[m] = L.Zbar[nd.pro.src_node];  // the cotangent of V IS the cotangent of the source record
      } else {
        L.Vbar[m] = o;
        o += rec_floats(ct, K, nd.Hin);
      }
      L.Pbar[m] = 0;
      if (nd.pro.skip_node >= 0) {
        L.Pbar[m] = o;
        o += rec_floats(ct, K, nd.Hin);
      }
    }
    L.Pbar[kMaxNodes] = 0;
    if (P.head.skip_node >= 0) {
      L.Pbar[kMaxNodes] = o;
      o += rec_floats(ct, K, P.head.H);
    }
    L.partial = o;
    L.partial_floats = 0;
    L.det = o;
    if (deterministic) {
      L.det = o;
      o += (size_t)1024 * 7 * 1024;  // per-workgroup partials of the element-wise / head kernels (<= 1024 workgroups)
      L.partial = o;
      size_t big = 0;
      for (int m = 0; m < P.n_nodes; ++m) {
        const size_t e = (size_t)round32(P.node[m].Hout) * round32(P.node[m].Hin) + round32(P.node[m].Hout);
        if (e > big) big = e;
      }
      L.partial_floats = big * 512;  // up to 512 splits per weight-gradient launch
      o += L.partial_floats;
    }
  }
  L.total = o;
}

int fpt_for(int Hp) { return Hp <= 64 ? 1 : (Hp <= 256 ? 4 : (Hp <= 512 ? 8 : 16)); }

hipError_t allow_lds(const void* kern, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> guard(mu);
  if (done.count({kern, dev})) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.insert({kern, dev});
  (void)bytes;
  return e;
}

hipError_t launch_ew(int nt, int nx, const EwArgs& a, bool bwd, int act, int fpt, int grid, hipStream_t st) {
#define PINN_LM_CASE(NT_, NX_) \
  if (nt == NT_ && nx == NX_) return launch_lm_ew_##NT_##_##NX_(a, bwd, act, fpt, grid, st);
  PINN_LM_CASE(0, 0) PINN_LM_CASE(1, 0) PINN_LM_CASE(1, 1) PINN_LM_CASE(1, 2) PINN_LM_CASE(1, 3) PINN_LM_CASE(1, 4)
  PINN_LM_CASE(2, 0) PINN_LM_CASE(2, 2)
#undef PINN_LM_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_head(int nt, int nx, const HeadArgs& a, int fpt, int grid, hipStream_t st) {
#define PINN_LM_CASE(NT_, NX_) \
  if (nt == NT_ && nx == NX_) return launch_lm_head_##NT_##_##NX_(a, fpt, grid, st);
  PINN_LM_CASE(0, 0) PINN_LM_CASE(1, 0) PINN_LM_CASE(1, 1) PINN_LM_CASE(1, 2) PINN_LM_CASE(1, 3) PINN_LM_CASE(1, 4)
  PINN_LM_CASE(2, 0) PINN_LM_CASE(2, 2)
#undef PINN_LM_CASE
  return hipErrorInvalidValue;
}

int gemm_prefetch_mode() {  // PINN_LM_PREFETCH=0|1 (experiments), read once
  static const int v = [] {
    const char* e = getenv("PINN_LM_PREFETCH");
    return e ? atoi(e) : 0;
  }();
  return v;
}

bool gemm_wres_off() {  // PINN_LM_WRES=0 (experiments: the streaming kernel everywhere), read once
  static const bool v = [] {
    const char* e = getenv("PINN_LM_WRES");
    return e && atoi(e) == 0;
  }();
  return v;
}

template <bool COLS>
hipError_t launch_gemm(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  g.prefetch = gemm_prefetch_mode();
  const int out_rows = COLS ? g.w_cols : g.w_rows;
  const int depth = COLS ? g.w_rows : g.w_cols;
  const int items = (g.ncb + kCB - 1) / kCB;
  const size_t lds = lm_gemm_lds_bytes(depth);
  hipError_t e;
  if constexpr (!COLS) {
    // weight-resident kernel where the wave's slice fits the register file and the double-buffered stage the LDS
    const int nch = depth >> 5;
    const int rt = out_rows > 128 ? 8 : 4;
#define PINN_WRES16(NCH_)                                                                            \
  if (nch == NCH_ && wres16_shape(depth)) {                                                          \
    auto kern = lm_gemm_wres16<NCH_>;                                                                \
    const size_t wl = lm_gemm_wres16_lds_bytes(NCH_);                                                \
    if ((e = allow_lds(reinterpret_cast<const void*>(kern), wl)) != hipSuccess) return e;            \
    const int gy = (out_rows + 127) / 128;                                                           \
    int gx = num_cus() / gy;                                                                         \
    if (gx < 1) gx = 1;                                                                              \
    if (gx > g.ncb) gx = g.ncb;                                                                      \
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(kWresThreads), wl, st, g);                           \
    return hipGetLastError();                                                                        \
  }
    PINN_WRES16(12) PINN_WRES16(16)
#undef PINN_WRES16
#define PINN_WRES(NCH_, RT_)                                                                         \
  if (nch == NCH_ && rt == RT_ && !gemm_wres_off()) {                                               \
    auto kern = lm_gemm_wres<NCH_, RT_>;                                                             \
    const size_t wl = lm_gemm_wres_lds_bytes(NCH_, RT_);                                             \
    if ((e = allow_lds(reinterpret_cast<const void*>(kern), wl)) != hipSuccess) return e;            \
    const int gy = (out_rows + 32 * RT_ - 1) / (32 * RT_);                                           \
    const int wi = (g.ncb + 2 * (8 / RT_) - 1) / (2 * (8 / RT_));                                    \
    int gx = num_cus() / gy;                                                                         \
    if (gx < 1) gx = 1;                                                                              \
    if (gx > wi) gx = wi;                                                                            \
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(kWresThreads), wl, st, g);                           \
    return hipGetLastError();                                                                        \
  }
    PINN_WRES(8, 8) PINN_WRES(4, 8) PINN_WRES(4, 4)
#undef PINN_WRES
  }
  if (out_rows > 128) {
    auto kern = lm_gemm<COLS, 2>;
    if ((e = allow_lds(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return e;
    const int gy = (out_rows + 255) / 256;
    int gx = 2 * num_cus() / gy;
    if (gx < 1) gx = 1;
    if (gx > items) gx = items;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(kThreads), lds, st, g);
  } else {
    auto kern = lm_gemm<COLS, 1>;
    if ((e = allow_lds(reinterpret_cast<const void*>(kern), lds)) != hipSuccess) return e;
    int gx = 2 * num_cus();
    if (gx > items) gx = items;
    hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(kThreads), lds, st, g);
  }
  return hipGetLastError();
}

}  // namespace

int lm_expected_tensors(const PinnNetDesc* d) { return d ? expected_tensors(d) : -1; }

int lm_check(const PinnNetDesc* d, char* err, size_t errlen) {
  Program P;
  return build_program(d, P, err, errlen);
}

size_t lm_workspace_bytes(const PinnNetDesc* d, long long N, int nt, int nx, bool bwd, bool deterministic) {
  if (N <= 0) return 0;
  Program P;
  char err[64];
  if (build_program(d, P, err, sizeof(err)) != PINN_OK) return 0;
  plan_fusion(P, nt, nx);
  Layout L;
  make_layout(P, N, 1 + nt + nx, bwd, deterministic, L);
  return (L.total * sizeof(float) + 255) & ~(size_t)255;
}

int lm_run(const CallArgs& c, char* err, size_t en) {
  Program P;
  int rc = build_program(c.net, P, err, en);
  if (rc) return rc;
  if (c.num_tensors != P.n_tensors)
    return failf(err, en, PINN_ERR_BAD_DESC, "weight table has %d entries, this architecture's state_dict has %d", c.num_tensors, P.n_tensors);
  const int K = 1 + c.nt + c.nx;
  plan_fusion(P, c.nt, c.nx);
  static thread_local Layout L;  // 10 KB of offsets: keep it off the stack
  make_layout(P, c.N, K, c.bwd, c.deterministic, L);
  const size_t need = (L.total * sizeof(float) + 255) & ~(size_t)255;
  if (!c.workspace || c.ws_bytes < need) return failf(err, en, PINN_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, c.ws_bytes);
  if (reinterpret_cast<uintptr_t>(c.workspace) & 15) return failf(err, en, PINN_ERR_MISALIGNED, "workspace is not 16-byte aligned");
  float* ws = static_cast<float*>(c.workspace);
  hipStream_t st = c.stream;
  hipError_t e = hipSuccess;
#define LM_CHECK(call)                                                                                     \
  do {                                                                                                     \
    e = (call);                                                                                            \
    if (e != hipSuccess) return failf(err, en, PINN_ERR_HIP, "HIP error %d: %s (%s)", (int)e, hipGetErrorString(e), #call); \
  } while (0)

  // ---- pack parameters (and zero the packed gradients) ----
  for (int i = 0; i < P.n_tensors; ++i) {
    if (P.rows[i] == 0) continue;
    if (!c.weights[i]) return failf(err, en, PINN_ERR_BAD_DESC, "weight pointer %d is null", i);
    L.tab.item[i].src = c.weights[i];
    L.tab.item[i].user_grad = (c.bwd && c.grads && !P.transpose[i]) ? c.grads[i] : nullptr;
  }
  float* params = ws + L.params;
  float* grads = ws + L.grads;
  static thread_local PackTable second;  // fragment copies of merged weights: packed after lm_merge_pv_kernel has run
  second = L.tab;
  for (int i = 0; i < second.n; ++i) second.item[i].src = nullptr;
  for (int m = 0; m < P.n_nodes; ++m) {
    const int w = P.node[m].w;
    for (int tr = 0; tr < 2; ++tr) {
      const int fi = P.n_tensors + 2 * m + tr;
      if (tr && !c.bwd) {
        L.tab.item[fi].src = nullptr;
      } else if (w < P.n_tensors) {
        L.tab.item[fi].src = c.weights[w];
      } else {  // merged weight: the source is its packed (padded) image inside the workspace
        L.tab.item[fi].src = nullptr;
        second.item[fi].src = params + L.tab.item[w].off;
        second.item[fi].rows = L.tab.item[w].rows_p;
        second.item[fi].cols = L.tab.item[w].cols_p;
      }
    }
  }
  hipLaunchKernelGGL(lm_pack_kernel, dim3(8, L.tab.n), dim3(256), 0, st, L.tab, params);
  LM_CHECK(hipGetLastError());
  MergeTable merge;
  merge.n = P.n_derived;
  for (int j = 0; j < P.n_derived; ++j) {
    MergeItem& mi = merge.item[j];
    mi.wp = L.tab.item[P.derived[j].wp].off;
    mi.bp = L.tab.item[P.derived[j].bp].off;
    mi.wv = L.tab.item[P.derived[j].wv].off;
    mi.bv = L.tab.item[P.derived[j].bv].off;
    mi.wm = L.tab.item[P.derived_base + 2 * j].off;
    mi.bm = L.tab.item[P.derived_base + 2 * j + 1].off;
    mi.Hp = round32(P.derived[j].H);
  }
  if (P.n_derived > 0) {
    hipLaunchKernelGGL(lm_merge_pv_kernel, dim3(16, P.n_derived), dim3(256), 0, st, merge, params);
    LM_CHECK(hipGetLastError());
    hipLaunchKernelGGL(lm_pack_kernel, dim3(8, second.n), dim3(256), 0, st, second, params);
    LM_CHECK(hipGetLastError());
  }
  if (c.bwd) LM_CHECK(hipMemsetAsync(grads, 0, L.n_packed * sizeof(float), st));
  auto pp = [&](int idx) -> const float* { return idx >= 0 ? params + L.tab.item[idx].off : nullptr; };
  auto gp = [&](int idx) -> float* {  // packed gradient slot, or null when nobody wants it (merged tensors: always wanted)
    if (idx < 0 || !c.bwd || !c.grads) return nullptr;
    if (idx >= P.n_tensors) return grads + L.tab.item[idx].off;
    return c.grads[idx] ? grads + L.tab.item[idx].off : nullptr;
  };

  const long long ntiles = (c.N + kT - 1) / kT;
  const int cus = num_cus();

  auto fill_ew = [&](EwArgs& a, const Prologue& pro, long long ct, long long p_base, int self) {  // self: node index, kMaxNodes = head
    memset(&a, 0, sizeof(a));
    a.stats = (c.bwd && L.Stats[self]) ? ws + L.Stats[self] : nullptr;
    a.H = pro.H;
    a.Hp = round32(pro.H);
    const int fpt = fpt_for(a.Hp);
    a.G = a.Hp / fpt;
    a.ntiles = ct;
    a.N = c.N;
    a.p_base = p_base;
    a.src_kind = pro.src_kind;
    a.din = P.din;
    a.M = pro.M;
    a.x = c.x;
    a.t = c.t;
    a.eps = P.ln_eps;
    a.has_act = pro.act >= 0;
    a.act_param = pro.act_param;
    if (pro.src_kind == SRC_REC) a.srcA = ws + L.Y[pro.src_node];
    else {
      a.encW = pp(pro.enc_w);
      a.encb = pp(pro.enc_b);
    }
    a.ln_g = pp(pro.ln_g);
    a.ln_b = pp(pro.ln_b);
    a.skip = pro.skip_node >= 0 ? ws + L.V[pro.skip_node] : nullptr;
    return fpt;
  };
  // Element-wise / head launches: one workgroup per tile (both of its 16-point halves), as many workgroups as fit the
  // chip at once.  A 1024-thread workgroup of these kernels (65-128 VGPRs) owns a CU; 512-thread ones (widths <= 128)
  // share it in pairs.  More workgroups than that only repeat the per-workgroup set-up and flush (measured: C3 26.8 ->
  // 26.7 ms with the exact count, C5 101.7 -> 105.0 ms when its 512-thread launches lose their second workgroup).
  auto ew_grid = [&](long long ct, int G) {
    long long g = ct;
    const long long cap = (kPT * G > 512 ? 1LL : 2LL) * cus;
    return (int)(g < cap ? g : cap);
  };

  for (long long t0 = 0; t0 < ntiles; t0 += L.ct) {
    const long long ct = ntiles - t0 < L.ct ? ntiles - t0 : L.ct;
    const long long p_base = t0 * kT;
    const int ncb = (int)(ct * K);
    // ---------------- forward ----------------
    bool v_ready[kMaxNodes + 1];  // V record already written by the producing node's fused epilogue
    for (int m = 0; m <= kMaxNodes; ++m) v_ready[m] = false;
    auto fused_grid = [&](const Program::Fuse& f) {
      long long g = cus / f.gy;
      if (g < 1) g = 1;
      const long long work = f.rt == 8 ? 2 * ct : ct;  // 16-point units / whole tiles (lm_fused.h)
      return (int)(g < work ? g : work);
    };
    for (int m = 0; m < P.n_nodes; ++m) {
      const Node& nd = P.node[m];
      if (!nd.pro.identity() && !v_ready[m]) {
        EwArgs a;
        const int fpt = fill_ew(a, nd.pro, ct, p_base, m);
        a.V = ws + L.V[m];
        if (nd.pro.src_kind == SRC_COORDS_FOURIER) LM_CHECK(launch_ew(c.nt, c.nx, a, false, -2, fpt, ew_grid(ct, a.G), st));
        else LM_CHECK(launch_ew(c.nt, c.nx, a, false, nd.pro.act, fpt, ew_grid(ct, a.G), st));
      }
      if (P.fuse_fwd[m].on) {  // Y[m] = W V + b (+ add) and the consumer's prologue in one launch
        const bool to_head = m + 1 >= P.n_nodes;
        const Prologue& cp = consumer_of(P, m);
        const int self = to_head ? kMaxNodes : m + 1;
        FusedArgs f;
        memset(&f, 0, sizeof(f));
        f.W = params + L.tab.item[P.n_tensors + 2 * m].off;
        f.bias = pp(nd.b);
        f.X = ws + L.V[m];
        f.rows_p = round32(nd.Hout);
        f.rows = nd.Hout;
        f.ntiles = ct;
        f.add0 = nd.add_node >= 0 ? ws + L.V[nd.add_node] : nullptr;
        f.Y = ws + L.Y[m];
        f.ln_g = pp(cp.ln_g);
        f.ln_b = pp(cp.ln_b);
        f.eps = P.ln_eps;
        f.skip = cp.skip_node >= 0 ? ws + L.V[cp.skip_node] : nullptr;
        f.has_act = cp.act >= 0;
        f.act_param = cp.act_param;
        f.V = ws + (to_head ? L.Vh : L.V[m + 1]);
        f.stats = (c.bwd && L.Stats[self]) ? ws + L.Stats[self] : nullptr;
        LM_CHECK(launch_fused(c.nt, c.nx, cp.act, f, false, cp.ln_g >= 0, P.fuse_fwd[m], fused_grid(P.fuse_fwd[m]), st));
        v_ready[self] = true;
        continue;
      }
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.W = params + L.tab.item[P.n_tensors + 2 * m].off;  // W in fragment order
      g.bias = pp(nd.b);
      g.X = ws + L.V[m];
      g.Y = ws + L.Y[m];
      g.add0 = nd.add_node >= 0 ? ws + L.V[nd.add_node] : nullptr;
      g.w_rows = round32(nd.Hout);
      g.w_cols = round32(nd.Hin);
      g.ncb = ncb;
      g.K = K;
      LM_CHECK(launch_gemm<false>(g, st));
    }
    {
      EwArgs a;
      const int fpt = fill_ew(a, P.head, ct, p_base, kMaxNodes);
      a.V = ws + L.Vh;
      if (v_ready[kMaxNodes]) {
      } else if (P.head.src_kind == SRC_COORDS_FOURIER) LM_CHECK(launch_ew(c.nt, c.nx, a, false, -2, fpt, ew_grid(ct, a.G), st));
      else LM_CHECK(launch_ew(c.nt, c.nx, a, false, P.head.act, fpt, ew_grid(ct, a.G), st));
      HeadArgs h;
      memset(&h, 0, sizeof(h));
      h.H = a.H;
      h.Hp = a.Hp;
      h.G = a.G;
      h.ntiles = ct;
      h.N = c.N;
      h.p_base = p_base;
      h.V = ws + L.Vh;
      h.w_out = pp(P.w_out);
      h.b_out = c.weights[P.b_out];
      h.pde = c.pde;
      h.mode = c.mode;
      h.bwd = c.bwd ? 1 : 0;
      h.grad_scale = c.grad_scale;
      h.x = c.x;
      h.din = P.din;
      for (int s = 0; s < K; ++s) {
        h.jets_out[s] = c.jets_out ? c.jets_out[s] : nullptr;
        h.jets_bar[s] = c.jets_bar ? c.jets_bar[s] : nullptr;
      }
      h.residual_out = c.residual_out;
      h.loss_sum = c.loss_sum;
      h.res_bar = c.res_bar;
      h.U = ws + L.U;
      h.dw_out = gp(P.w_out);
      h.db_out = gp(P.b_out);
      const bool det = c.deterministic && c.bwd;
      if (det) h.det_partial = ws + L.det;
      const int hgrid = ew_grid(ct, h.G);
      LM_CHECK(launch_head(c.nt, c.nx, h, fpt, hgrid, st));
      if (det) {
        SlotReduce r;
        memset(&r, 0, sizeof(r));
        r.row = 1028;
        r.n = 5;
        r.dst[3] = (c.mode == MODE_PDE) ? c.pde.dcoef : nullptr; r.slot[3] = 1026; r.mul[3] = 1; r.len[3] = 1;
        r.dst[4] = (c.mode == MODE_PDE && c.pde.dcoef) ? c.pde.dcoef + 1 : nullptr; r.slot[4] = 1027; r.mul[4] = 1; r.len[4] = 1;
        r.dst[0] = h.dw_out; r.slot[0] = 0; r.mul[0] = 1; r.len[0] = h.H;
        r.dst[1] = (c.mode == MODE_PDE) ? c.loss_sum : nullptr; r.slot[1] = 1024; r.mul[1] = 1; r.len[1] = 1;
        r.dst[2] = h.db_out; r.slot[2] = 1025; r.mul[2] = 1; r.len[2] = 1;
        hipLaunchKernelGGL(lm_reduce_slots, dim3(4), dim3(256), 0, st, ws + L.det, hgrid, r);
        LM_CHECK(hipGetLastError());
      }
    }
    if (!c.bwd) continue;
    // ---------------- reverse ----------------
    // extra cotangents of V records: skip connections (Pbar of the consumer's prologue) and epilogue adds (Zbar of the adding node)
    const float* extra[kMaxNodes][2];
    for (int m = 0; m < P.n_nodes; ++m) extra[m][0] = extra[m][1] = nullptr;
    auto add_extra = [&](int m, const float* rec) {
      if (!extra[m][0]) extra[m][0] = rec;
      else extra[m][1] = rec;
    };
    auto run_ew_bwd = [&](const Prologue& pro, int self, const float* vbar) -> int {
      // self = node index (kMaxNodes for the head); vbar = cotangent record of its V, or null for the head's (U, w_out) form
      const bool needs = pro.src_kind == SRC_REC || (pro.src_kind == SRC_COORDS_LINEAR) || pro.ln_g >= 0 || pro.skip_node >= 0;
      if (!needs) return PINN_OK;  // Fourier features straight from the coordinates: nothing upstream to differentiate
      EwArgs a;
      const int fpt = fill_ew(a, pro, ct, p_base, self);
      a.Vbar = vbar;
      if (!vbar) {
        a.U = ws + L.U;
        a.w_out = pp(P.w_out);
      }
      if (pro.src_kind == SRC_REC) a.Zbar = ws + L.Zbar[pro.src_node];
      if (pro.skip_node >= 0) {
        a.Pbar = ws + L.Pbar[self];
        add_extra(pro.skip_node, a.Pbar);
      }
      a.d_ln_g = gp(pro.ln_g);
      a.d_ln_b = gp(pro.ln_b);
      if (pro.src_kind == SRC_COORDS_LINEAR) {
        a.d_encW = gp(pro.enc_w);
        a.d_encb = gp(pro.enc_b);
        if (!a.d_encW && pro.ln_g < 0 && pro.skip_node < 0) return PINN_OK;
      }
      const bool has_sums = pro.ln_g >= 0 || (pro.src_kind == SRC_COORDS_LINEAR && a.d_encW);
      if (c.deterministic && has_sums) a.det_partial = ws + L.det;
      const int egrid = ew_grid(ct, a.G);
      e = launch_ew(c.nt, c.nx, a, true, pro.act, fpt, egrid, st);
      if (e != hipSuccess) return failf(err, en, PINN_ERR_HIP, "HIP error %d: %s (lm_ew_bwd)", (int)e, hipGetErrorString(e));
      if (a.det_partial) {
        SlotReduce r;
        memset(&r, 0, sizeof(r));
        r.row = 7 * 1024;
        r.n = 7;
        r.dst[0] = pro.ln_g >= 0 ? a.d_ln_g : nullptr; r.slot[0] = 0; r.mul[0] = 1;
        r.dst[1] = pro.ln_g >= 0 ? a.d_ln_b : nullptr; r.slot[1] = 1024; r.mul[1] = 1;
        for (int cc = 0; cc < 4; ++cc) {
          r.dst[2 + cc] = a.d_encW; r.slot[2 + cc] = (2 + cc) * 1024; r.mul[2 + cc] = 4; r.add[2 + cc] = cc;
        }
        r.dst[6] = a.d_encb; r.slot[6] = 6 * 1024; r.mul[6] = 1;
        for (int k = 0; k < 7; ++k) r.len[k] = a.H;
        hipLaunchKernelGGL(lm_reduce_slots, dim3(4), dim3(256), 0, st, ws + L.det, egrid, r);
        e = hipGetLastError();
        if (e != hipSuccess) return failf(err, en, PINN_ERR_HIP, "HIP error %d: %s (lm_reduce_slots)", (int)e, hipGetErrorString(e));
      }
      return PINN_OK;
    };
    if ((rc = run_ew_bwd(P.head, kMaxNodes, nullptr)) != PINN_OK) return rc;
    // 256-multiple weight gradients are collected and run as one launch behind the sweep (lm_gemm_nt8d_batch;
    // PINN_LM_NT_BATCH=0: one launch per layer, as the deterministic mode always does)
    static const bool nt_batch_on = [] { const char* e = getenv("PINN_LM_NT_BATCH"); return !(e && atoi(e) == 0); }();
    static thread_local GemmNtBatch batches[4];  // [0]: 256 x 256 blocks (nt8d); [1..3]: ntd<256,128>, <128,256>, <128,128>
    for (GemmNtBatch& b : batches) {
      b.n = 0;
      b.ncb = ncb;
      b.K = K;
    }
    auto flush_batch = [&](int which) -> hipError_t {
      GemmNtBatch& batch = batches[which];
      if (batch.n == 0) return hipSuccess;
      const void* kern = which == 0   ? reinterpret_cast<const void*>(lm_gemm_nt8d_batch)
                         : which == 1 ? reinterpret_cast<const void*>(lm_gemm_ntd_batch<256, 128>)
                         : which == 2 ? reinterpret_cast<const void*>(lm_gemm_ntd_batch<128, 256>)
                                      : reinterpret_cast<const void*>(lm_gemm_ntd_batch<128, 128>);
      const size_t lds = which == 0 ? lm_gemm_nt8d_lds_bytes() : lm_gemm_ntd_lds_bytes(which == 1 ? 256 : 128, which == 2 ? 256 : 128);
      hipError_t e = allow_lds(kern, lds);
      if (e != hipSuccess) return e;
      int gx = (lds * 2 <= 160 * 1024 ? 2 : 1) * cus / batch.n;
      if (gx > ncb / 12) gx = ncb / 12;
      if (gx < 1) gx = 1;
      const dim3 grid(gx, batch.n);
      if (which == 0) hipLaunchKernelGGL(lm_gemm_nt8d_batch, grid, dim3(512), lds, st, batch);
      else if (which == 1) hipLaunchKernelGGL((lm_gemm_ntd_batch<256, 128>), grid, dim3(512), lds, st, batch);
      else if (which == 2) hipLaunchKernelGGL((lm_gemm_ntd_batch<128, 256>), grid, dim3(512), lds, st, batch);
      else hipLaunchKernelGGL((lm_gemm_ntd_batch<128, 128>), grid, dim3(512), lds, st, batch);
      batch.n = 0;
      return hipGetLastError();
    };
    auto push_jobs = [&](int which, const GemmNtArgs& g, int zr, int vr) -> hipError_t {
      const int gy = g.z_rows / zr, gz = g.v_rows / vr;
      GemmNtBatch& batch = batches[which];
      if (batch.n + gy * gz > kMaxNtJobs) {
        const hipError_t e = flush_batch(which);
        if (e != hipSuccess) return e;
      }
      for (int by = 0; by < gy; ++by)
        for (int bz = 0; bz < gz; ++bz) {
          GemmNtBatch::Job& j = batch.job[batch.n++];
          j.Z = g.Z;
          j.V = g.V;
          j.dW = g.dW;
          j.db = bz == 0 ? g.db : nullptr;
          j.z_rows = g.z_rows;
          j.v_rows = g.v_rows;
          j.zr0 = by * zr;
          j.vc0 = bz * vr;
        }
      return hipSuccess;
    };
    for (int m = P.n_nodes - 1; m >= 0; --m) {
      const Node& nd = P.node[m];
      const float* zbar = ws + L.Zbar[m];
      if (nd.add_node >= 0) add_extra(nd.add_node, zbar);
      // weight gradient
      if (gp(nd.w) || gp(nd.b)) {
        GemmNtArgs g;
        memset(&g, 0, sizeof(g));
        g.Z = zbar;
        g.V = ws + L.V[m];
        g.dW = grads + L.tab.item[nd.w].off;
        g.db = gp(nd.b);
        g.z_rows = round32(nd.Hout);
        g.v_rows = round32(nd.Hin);
        g.ncb = ncb;
        g.K = K;
        const bool big = g.z_rows >= 256 && g.v_rows >= 128;  // 256-row blocks on eight waves (x up to 256 columns)
        const int gy = big ? (g.z_rows + kNt8 - 1) / kNt8 : (g.z_rows + 127) / 128;
        const int gz = big ? (g.v_rows + kNt8 - 1) / kNt8 : (g.v_rows + kNtCols - 1) / kNtCols;
        // splits: fill the chip, but leave every workgroup >= 12 column blocks to amortise its one-off flush
        // (a 128 x 128 block is 64 KB of float atomics; at 3 column blocks per workgroup the flush was 3x the GEMM)
        int gx = (big ? cus : 2 * cus) / (gy * gz);
        if (gx > ncb / 12) gx = ncb / 12;
        if (gx < 1) gx = 1;
        if (gx > 512) gx = 512;
        static const bool dma_on = [] { const char* e = getenv("PINN_LM_NT8D"); return !(e && atoi(e) == 0); }();  // 0: register-staged nt8
        const bool dma = dma_on && big && g.z_rows % kNt8 == 0 && g.v_rows % kNt8 == 0;  // the DMA kernel takes complete blocks only
        // width-128 shapes: the LDS-DMA kernel on 256 x 128 / 128 x 256 / 128 x 128 blocks (complete blocks only)
        static const bool ntd_on = [] { const char* e = getenv("PINN_LM_NTD"); return !(e && atoi(e) == 0); }();
        int zr = 0, vr = 0;
        if (ntd_on && !(big && dma) && g.z_rows % 128 == 0 && g.v_rows % 128 == 0) {
          zr = g.z_rows % 256 == 0 ? 256 : 128;
          vr = (g.v_rows % 256 == 0 && zr == 128) ? 256 : 128;
        }
        const bool can_batch = nt_batch_on && !c.deterministic;
        if (can_batch && dma && gy * gz <= kMaxNtJobs) {
          LM_CHECK(push_jobs(0, g, kNt8, kNt8));
        } else if (can_batch && zr && (g.z_rows / zr) * (g.v_rows / vr) <= kMaxNtJobs) {
          LM_CHECK(push_jobs(zr == 256 ? 1 : (vr == 256 ? 2 : 3), g, zr, vr));
        } else {
        const size_t lds = big ? (dma ? lm_gemm_nt8d_lds_bytes() : lm_gemm_nt8_lds_bytes()) : lm_gemm_nt_lds_bytes();
        const void* kern = big ? (dma ? reinterpret_cast<const void*>(lm_gemm_nt8d) : reinterpret_cast<const void*>(lm_gemm_nt8))
                               : reinterpret_cast<const void*>(lm_gemm_nt);
        LM_CHECK(allow_lds(kern, lds));
        const size_t stride = (size_t)g.z_rows * g.v_rows + g.z_rows;
        if (c.deterministic) g.partial = ws + L.partial;  // every element of a split's block is stored by exactly one workgroup
        if (zr) {
          const int gy2 = g.z_rows / zr, gz2 = g.v_rows / vr;
          const size_t l2 = lm_gemm_ntd_lds_bytes(zr, vr);
          int gx2 = (l2 * 2 <= 160 * 1024 ? 2 : 1) * cus / (gy2 * gz2);  // 512-thread workgroups, 64-96 KB of LDS each
          if (gx2 > ncb / 12) gx2 = ncb / 12;
          if (gx2 < 1) gx2 = 1;
          if (gx2 > 512) gx2 = 512;
          const void* k2 = zr == 256 ? reinterpret_cast<const void*>(lm_gemm_ntd<256, 128>)
                           : (vr == 256 ? reinterpret_cast<const void*>(lm_gemm_ntd<128, 256>) : reinterpret_cast<const void*>(lm_gemm_ntd<128, 128>));
          LM_CHECK(allow_lds(k2, l2));
          if (zr == 256) hipLaunchKernelGGL((lm_gemm_ntd<256, 128>), dim3(gx2, gy2, gz2), dim3(512), l2, st, g);
          else if (vr == 256) hipLaunchKernelGGL((lm_gemm_ntd<128, 256>), dim3(gx2, gy2, gz2), dim3(512), l2, st, g);
          else hipLaunchKernelGGL((lm_gemm_ntd<128, 128>), dim3(gx2, gy2, gz2), dim3(512), l2, st, g);
          gx = gx2;  // the deterministic reduction below sums this many partial blocks
        } else if (big && dma) hipLaunchKernelGGL(lm_gemm_nt8d, dim3(gx, gy, gz), dim3(512), lds, st, g);
        else if (big) hipLaunchKernelGGL(lm_gemm_nt8, dim3(gx, gy, gz), dim3(512), lds, st, g);
        else hipLaunchKernelGGL(lm_gemm_nt, dim3(gx, gy, gz), dim3(kThreads), lds, st, g);
        LM_CHECK(hipGetLastError());
        if (c.deterministic) {
          hipLaunchKernelGGL(lm_reduce_partials, dim3(256), dim3(256), 0, st, g.partial, (long long)stride, gx, g.dW,
                             (long long)g.z_rows * g.v_rows, g.db, g.z_rows);
          LM_CHECK(hipGetLastError());
        }
        }
      }
      // cotangent of this node's GEMM input, then through its prologue
      const Prologue& pro = nd.pro;
      const bool upstream = pro.src_kind == SRC_REC || pro.src_kind == SRC_COORDS_LINEAR || pro.ln_g >= 0 || pro.skip_node >= 0;
      if (!upstream) continue;
      if (pro.src_kind == SRC_COORDS_LINEAR && !gp(pro.enc_w) && !gp(pro.enc_b) && pro.ln_g < 0 && pro.skip_node < 0) continue;
      if (P.fuse_bwd[m].on) {  // Vbar = W^T Zbar (+ extras) and the prologue's adjoint in one launch
        FusedArgs f;
        memset(&f, 0, sizeof(f));
        f.W = params + L.tab.item[P.n_tensors + 2 * m + 1].off;
        f.X = zbar;
        f.rows_p = round32(nd.Hin);
        f.rows = nd.Hin;
        f.ntiles = ct;
        f.add0 = extra[m][0];
        f.add1 = extra[m][1];
        f.ln_g = pp(pro.ln_g);
        f.ln_b = pp(pro.ln_b);
        f.eps = P.ln_eps;
        f.skip = pro.skip_node >= 0 ? ws + L.V[pro.skip_node] : nullptr;
        f.has_act = pro.act >= 0;
        f.act_param = pro.act_param;
        f.stats = L.Stats[m] ? ws + L.Stats[m] : nullptr;
        f.Zsrc = ws + L.Y[pro.src_node];
        f.Zbar = ws + L.Zbar[pro.src_node];
        if (pro.skip_node >= 0) {
          f.Pbar = ws + L.Pbar[m];
          add_extra(pro.skip_node, f.Pbar);
        }
        f.d_ln_g = gp(pro.ln_g);
        f.d_ln_b = gp(pro.ln_b);
        if (c.deterministic && pro.ln_g >= 0) f.det_partial = ws + L.det;
        const int fgrid = fused_grid(P.fuse_bwd[m]);
        LM_CHECK(launch_fused(c.nt, c.nx, pro.act, f, true, pro.ln_g >= 0, P.fuse_bwd[m], fgrid, st));
        if (f.det_partial) {
          SlotReduce r;
          memset(&r, 0, sizeof(r));
          r.row = 7 * 1024;
          r.n = 2;
          r.dst[0] = f.d_ln_g; r.slot[0] = 0; r.mul[0] = 1; r.len[0] = nd.Hin;
          r.dst[1] = f.d_ln_b; r.slot[1] = 1024; r.mul[1] = 1; r.len[1] = nd.Hin;
          hipLaunchKernelGGL(lm_reduce_slots, dim3(4), dim3(256), 0, st, ws + L.det, fgrid, r);
          LM_CHECK(hipGetLastError());
        }
        continue;
      }
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.W = params + L.tab.item[P.n_tensors + 2 * m + 1].off;  // W^T in fragment order (Hin_p x Hout_p)
      g.X = zbar;
      g.Y = ws + L.Vbar[m];
      g.add0 = extra[m][0];
      g.add1 = extra[m][1];
      g.w_rows = round32(nd.Hin);
      g.w_cols = round32(nd.Hout);
      g.ncb = ncb;
      g.K = K;
      LM_CHECK(launch_gemm<false>(g, st));
      if (!pro.identity() && (rc = run_ew_bwd(pro, m, ws + L.Vbar[m])) != PINN_OK) return rc;
    }
    for (int w = 0; w < 4; ++w) LM_CHECK(flush_batch(w));
  }
  if (c.bwd && P.n_derived > 0) {  // merged gradients back onto W_p, b_p, W_v, b_v (their packed slots), then the one unpack
    hipLaunchKernelGGL(lm_unmerge_pv_kernel, dim3(16, P.n_derived), dim3(256), 0, st, merge, params, grads);
    LM_CHECK(hipGetLastError());
  }
  if (c.bwd) {
    hipLaunchKernelGGL(lm_unpack_kernel, dim3(8, L.tab.n), dim3(256), 0, st, L.tab, grads);
    LM_CHECK(hipGetLastError());
  }
#undef LM_CHECK
  return PINN_OK;
}

}  // namespace lm
}  // namespace pinn
