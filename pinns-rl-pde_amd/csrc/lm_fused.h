// Fused GEMM + prologue kernels of the layer-major engine (see lm_common.h).  gfx950, v_mfma_f32_16x16x4_f32.
//
// The unfused engine runs a layer as  V = prologue(Y_prev)  (lm_ew.h: LayerNorm / skip / activation jets, pure HBM
// traffic, no MFMA)  and  Y = W V + b  (lm_gemm.h), and the reverse sweep as  Vbar = W^T Zbar  and
// Zbar_prev = prologue^T(Vbar): 30-37 % of the width-256 configurations' GPU time went to the element-wise launches
// (profiles/r02_C3.md).  Here the prologue of the CONSUMER runs in the epilogue of the producing GEMM:
//
//   forward   Y = W X + b (+ add record)  ->  store Y (the reverse sweep's source record)
//             V' = act(LayerNorm(Y) gamma + beta + skip)  ->  store V' (the next GEMM's input, and its dW operand)
//   reverse   Vbar = W^T Zbar (+ skip-path cotangents)  ->  never stored
//             Zbar_prev = prologue^T(Vbar; z = Y_prev, kept LayerNorm sums, skip)  ->  store (+ Pbar, dgamma, dbeta)
//
// What makes the epilogue possible is that a lane owns ALL K streams of its elements: a workgroup (8 waves) owns a
// 16-point UNIT (half a record tile) x up to 256 output rows, wave w the rows 32 w .. 32 w + 31 as two 16 x 16 MFMA
// blocks, K streams x 2 blocks x 4 accumulator registers = 8 K VGPRs (40 at K = 5; the 32 x 32 x 2 form needs 16 K).
// The accumulator layout of v_mfma_f32_16x16x4_f32 — row 4 (lane >> 4) + i, point lane & 15 — is the thread map of
// the element-wise kernels (point = tid & 15, lanes l, l ^ 16, l ^ 32 share it), so their jet / LayerNorm arithmetic
// and block reductions carry over.
//
//   * weights: the wave's 32 x depth slice in REGISTERS for the whole launch (16 NCH VGPRs; lm_common.h::frag16_index
//     packs them so that the start-up load is dwordx4);
//   * B operand: one (unit, stream) SLAB [depth][16 points] per stage, a ring of kFRing stages in LDS filled by
//     LDS-DMA (1 KB pieces = 16 half rows of 64 B gathered through the per-lane source address, two per wave and
//     stage) kFRing - 1 stages ahead; stream-major MFMA order (the weights are resident, so nothing is re-fetched),
//     one plain s_barrier per stage, VMEM waits counted by hand (a barrier that drained vmcnt would wait for the
//     epilogue's stores);
//   * B reads: lane l reads float l of every 64-float group of four reduction rows: ds_read2st64_b32 from ONE address.
//
// Shapes: depth 128 / 256 (NCH 4 / 8), 256 output rows per workgroup (RT 8; more rows = more workgroups in y, not
// with LayerNorm) or 128 (RT 4: the two wave groups take the two halves of a tile).  Everything else stays on the
// unfused kernels, which remain the second implementation the parity tests run (PINN_LM_FUSED=0).
#pragma once
#include "lm_ew.h"

namespace pinn {
namespace lm {

constexpr int kFThreads = 512;
constexpr int kFRing = 6;

struct FusedArgs {
  const float* W;       // (rows_p x depth) weight in frag16 order
  const float* bias;    // forward: rows_p floats (added on the value stream) or null
  const float* X;       // GEMM input record: [ntiles][K][depth][32]
  int rows_p, rows;     // padded / logical output rows = features of the epilogue
  long long ntiles;
  const float* add0;    // records added to the GEMM result (shape of the output) or null
  const float* add1;
  float* Y;             // forward: the GEMM result (source record of the consumer's prologue)
  const float* ln_g;    // prologue: LayerNorm scale / shift, packed [rows_p] (zero beyond rows)
  const float* ln_b;
  float eps;
  const float* skip;    // record added before the activation, or null
  int has_act;
  float act_param;
  float* V;             // forward: prologue output record
  float* stats;         // LayerNorm: [tile][2 K][32] per-point sums (written forward, read in reverse)
  const float* Zsrc;    // reverse: source record z of the prologue
  float* Zbar;          // reverse: cotangent of the source record
  float* Pbar;          // reverse: cotangent of the skip record or null
  float* d_ln_g;        // reverse: packed gradients (accumulated) or null
  float* d_ln_b;
  float* det_partial;   // deterministic mode: per-workgroup partials [grid][7][1024] (slots 0 / 1 used)
  unsigned long long* stamps;  // diagnostic builds (-DPINN_FSTAMPS, tools/micro/fused_bench.hip): [grid][8 waves][8] cycles
};

#ifdef PINN_FSTAMPS
#define PINN_FSTAMP_DECL unsigned long long fst_acc[8] = {}; unsigned long long fst_prev = pinn_fnow(); const unsigned long long fst_begin = fst_prev;
#define PINN_FSTAMP(idx) do { const unsigned long long fst_now = pinn_fnow(); fst_acc[idx] += fst_now - fst_prev; fst_prev = fst_now; } while (0)
__device__ __forceinline__ unsigned long long pinn_fnow() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#else
#define PINN_FSTAMP_DECL
#define PINN_FSTAMP(idx)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef PINN_FEXP_NOWAIT
  return;
#endif
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48: the lanes that share a point), every lane gets
// the total.  v_permlane16_swap / v_permlane32_swap (gfx950) with both operands holding x: [r0 r1 r2 r3] x 2 ->
// [r0 r0 r2 r2], [r1 r1 r3 r3]; their sum [a a b b] x 2 -> [a a a a], [b b b b].  No LDS crossbar round trip (__shfl_xor is
// ds_bpermute_b32).  Inline asm: the builtin form folds r[0] + r[1] into 2 r[0] when both operands are the same value.
__device__ __forceinline__ float rows4_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float y = a + b;
  float c = y, d = y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  return c + d;
}

// sum over the 16 lanes of a DPP row (the 16 points of a unit), every lane gets the total: xor-butterfly on quad_perm /
// row_half_mirror / row_mirror
__device__ __forceinline__ float pts16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// sum over the lanes / waves that share this lane's point (the four rows of each of the NW waves of a group); partials
// in LDS as [quantity][point][wave], so that a lane fetches all waves' partials of a quantity with one or two 16-byte reads
template <int NQ, int NW>
__device__ __forceinline__ void group_sum(float (&q)[NQ], float* red, int& slot, int widx, int lane, int p) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) q[i] = rows4_sum(q[i]);
  float* area = red + slot * (8 * kRedQ * kPT);
  slot ^= 1;
  if (lane < kPT) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) area[(i * kPT + p) * 8 + widx] = q[i];
  }
  lds_barrier();
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(area + (i * kPT + p) * 8);
    float s = (v0[0] + v0[1]) + (v0[2] + v0[3]);
    if constexpr (NW == 8) {
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(area + (i * kPT + p) * 8 + 4);
      s += (v1[0] + v1[1]) + (v1[2] + v1[3]);
    }
    q[i] = s;
  }
}

// LayerNorm statistics of the 8 elements x K streams a lane holds (lm_ew.h::ln_stats on this thread map)
template <int NT, int NX, int NW>
__device__ __forceinline__ void fused_ln_stats(float (&c)[8][1 + NT + NX], const bool (&valid)[8], int H, float eps, LnPoint<NT, NX>& S,
                                               float* red, int& slot, int widx, int lane, int p, float* stats_out) {
  constexpr int K = 1 + NT + NX;
  const float invH = 1.0f / (float)H;
  float q[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += valid[i] ? c[i][s] : 0.0f;
    q[s] = t;
  }
  group_sum<K, NW>(q, red, slot, widx, lane, p);
  if (stats_out && widx == 0 && lane < kPT) {
#pragma unroll
    for (int s = 0; s < K; ++s) stats_out[s * kT + p] = q[s];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int s = 0; s < K; ++s) c[i][s] = valid[i] ? c[i][s] - q[s] * invH : 0.0f;
  float m[K];
#pragma unroll
  for (int s = 0; s < K; ++s) m[s] = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m[0] = fmaf(c[i][0], c[i][0], m[0]);
#pragma unroll
    for (int k = 1; k <= NT; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j) m[sidx(1, k)] = fmaf((float)binom(k, j) * c[i][sidx(1, j)], c[i][sidx(1, k - j)], m[sidx(1, k)]);
#pragma unroll
    for (int k = 1; k <= NX; ++k)
#pragma unroll
      for (int j = 0; j <= k; ++j)
        m[sidx(1 + NT, k)] = fmaf((float)binom(k, j) * c[i][sidx(1 + NT, j)], c[i][sidx(1 + NT, k - j)], m[sidx(1 + NT, k)]);
  }
  group_sum<K, NW>(m, red, slot, widx, lane, p);
  if (stats_out && widx == 0 && lane < kPT) {
#pragma unroll
    for (int s = 0; s < K; ++s) stats_out[(K + s) * kT + p] = m[s];
  }
  ln_point_from_moments<NT, NX>(m, invH, eps, S);
}

// Ring depth.  Without a landing zone: kFRing stages.  With one — the reverse kernels' source record z, the forward
// kernels' skip / add record: K x 16 KB of LDS, every wave's own 32 rows x 16 points x K streams, DMA'd one unit ahead —
// the ring takes what is left.
__host__ __device__ constexpr int fused_fixed_lds_floats(int cg) { return 3 * 256 + 2 * 512 + cg * 2 * 8 * kRedQ * kPT; }
__host__ __device__ constexpr int fused_ring(int K, bool land, int cg, int nch) {
  const int avail = 160 * 1024 - 4 * fused_fixed_lds_floats(cg) - (land ? 8 * K * 512 * 4 : 0);
  const int r = avail / (cg * 32 * nch * kPT * 4);
  return !land ? kFRing : (r > 4 ? 4 : r);
}

// VMEM operations every epilogue issues at least (forward: the Y and V stores; reverse: the Zbar stores and the next
// unit's z pieces): what the hand-counted vmcnt of the stage loop may assume to be younger than a DMA piece
template <int K>
__host__ __device__ constexpr int fused_epi_ops(bool bwd) { return bwd ? 10 * K : 16 * K; }

// younger-than-the-awaited-pieces VMEM operations at stage s of a steady-state iteration (see the kernel)
template <int K, int PP, bool BWD, int R>
__host__ __device__ constexpr int fused_vm_wait(int s) {
  int cnt = 0;
  for (int t = 1; t <= R - 1; ++t)
    if ((((s - t) % K) + K) % K == K - 1) ++cnt;
  const int n = PP * (R - 2) + fused_epi_ops<K>(BWD) * cnt;
  return n > 63 ? 63 : n;
}

// AUX: forward — a skip or add record exists and travels through the landing zone; reverse — an add record (skip-path
// cotangent) exists and is requested into registers before the last stage's MFMAs.
template <int NCH, int RT, int NT, int NX, int ACT, bool LN, bool BWD, bool AUX>
__global__ __launch_bounds__(kFThreads, 1) void lm_fused(const FusedArgs a) {
  constexpr int K = 1 + NT + NX;
  constexpr bool LAND = BWD || AUX;
  constexpr int CG = 8 / RT;
  constexpr int DEPTH = 32 * NCH;
  constexpr int NJ = DEPTH / 4;             // k-steps of 4
  constexpr int kStage = CG * DEPTH * kPT;  // floats per ring stage
  constexpr int PP = CG * DEPTH / 16 / 8;   // DMA pieces per wave and stage
  constexpr int R = fused_ring(K, LAND, CG, NCH);
  static_assert(PP >= 1 && (CG * DEPTH / 16) % 8 == 0, "a stage must divide over the eight waves");
  static_assert(R >= 2, "the ring needs a slot to fill while one is read");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;                  // [R][kStage]
  float* sbias = ring + R * kStage;    // [256] bias of this block's rows (forward)
  float* sgam = sbias + 256;           // [256] LayerNorm scale / shift of this block's rows
  float* sbet = sgam + 256;
  float* paccb = sbet + 256;           // [CG][2][256] dgamma / dbeta partial sums of each wave group (reverse)
  float* redb = paccb + 2 * 512;       // [CG][2][8][kRedQ][16] block reductions
  float* land = redb + CG * (2 * 8 * kRedQ * kPT);  // reverse: [8 waves][K][2][16 rows][16 points] source-record jets

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave % RT, cg = wave / RT;
  const int p = lane & 15, kk = lane >> 4;
  const int blk_row0 = (int)blockIdx.y * (32 * RT);
  const int row0 = blk_row0 + 32 * rt;  // first output row of this wave
  // Work items: whole tiles for two wave groups (one half each), dealt round-robin; 16-point units for one, dealt as
  // CONTIGUOUS runs (a launch of 3 125 tiles on 256 CUs is 12.2 tiles each: as tiles the slowest workgroup takes 13, as
  // units 12.5; contiguous, so that the two halves of a tile — the two 64-byte halves of every 128-byte line of its
  // records — are read by the same workgroup back to back: dealt round-robin they went to two workgroups at the same
  // moment, every line was fetched twice, and the HBM-bound attention launches lost 2 %).
  const long long n_work = CG == 1 ? 2 * a.ntiles : a.ntiles;
  const long long w_q = n_work / gridDim.x, w_r = n_work % gridDim.x;
  const long long w_start = CG == 1 ? (long long)blockIdx.x * w_q + ((long long)blockIdx.x < w_r ? blockIdx.x : w_r) : blockIdx.x;
  const int n_iters = CG == 1 ? (int)(w_q + ((long long)blockIdx.x < w_r ? 1 : 0))
                              : (int)(n_work > (long long)blockIdx.x ? (n_work - 1 - blockIdx.x) / gridDim.x + 1 : 0);
  if (n_iters == 0) return;
  float* red = redb + cg * (2 * 8 * kRedQ * kPT);
  float* pacc = paccb + cg * 512;  // the two groups of RT = 4 own the same features (of different half tiles)
  float* zl = land + wave * (K * 512);

  // per-block parameter vectors
  for (int i = tid; i < 32 * RT; i += kFThreads) {
    const int row = blk_row0 + i;
    const bool in = row < a.rows_p;
    sbias[i] = (!BWD && a.bias && in) ? a.bias[row] : 0.0f;
    sgam[i] = (LN && in) ? a.ln_g[row] : 0.0f;
    sbet[i] = (LN && in) ? a.ln_b[row] : 0.0f;
    paccb[i] = paccb[256 + i] = paccb[512 + i] = paccb[768 + i] = 0.0f;
  }

  // the wave's weight slice: w[mb][j] = W[row0 + 16 mb + (lane & 15)][4 j + (lane >> 4)].  Resident for the whole launch,
  // except where the epilogue needs the registers (WRELOAD: the LayerNorm adjoint at depth 256 spilled 80 of them): there
  // the slice is dead during the epilogue's arithmetic and re-read from L2 (1 KB per wave-instruction, fragment order)
  // just before the epilogue's stores — BEFORE them, so that the first MFMA's wait does not include their round trip.
  constexpr bool WRELOAD = LN && BWD && NCH == 8;
  float w[2][NJ];
  auto load_w = [&]() {
    const float* wt = in_loop(a.W) + (long long)(row0 >> 5) * (2 * (NJ / 4) * 256);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int j4 = 0; j4 < NJ / 4; ++j4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wt + ((mb * (NJ / 4) + j4) * 64 + lane) * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) w[mb][4 * j4 + jj] = v[jj];
      }
  };
  load_w();

  auto work_of = [&](int it) -> long long { return CG == 1 ? w_start + it : w_start + (long long)it * gridDim.x; };
  auto tile_of = [&](int it) -> long long { return CG == 1 ? work_of(it) >> 1 : work_of(it); };
  auto half_of = [&](int it, int g) -> int { return CG == 1 ? (int)(work_of(it) & 1) : g; };

  // DMA of stage (it, s) into ring slot `buf`: piece pc = 8 u + wave covers 16 reduction rows of wave group pc / (DEPTH / 16)
  const unsigned lsrc = static_cast<unsigned>((lane >> 2) * kT + (lane & 3) * 4) * 4u;
  auto issue = [&](int it, int s, int buf) {
    if (it >= n_iters) it = n_iters - 1;  // past the end: re-read the last unit (never consumed); keeps the VMEM count static
    const long long tile = tile_of(it);
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const int pc = 8 * u + wave;
      const int g = pc / (DEPTH / 16), rg = pc % (DEPTH / 16);
      const float* base = uniform_ptr(a.X + ((tile * K + s) * DEPTH + 16 * rg) * kT + kPT * half_of(it, g));
      float* dst = ring + buf * kStage + pc * 256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + lsrc),
                                       (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    }
  };
  // this wave's rows of unit `it` of the landing record (reverse: the source record z; forward: the skip / add record),
  // 2 K pieces of 16 rows x 16 points, into its landing zone
  // Landing zone: the 16 rows of a piece land with bits 0 and 2 of the row index exchanged (a per-lane SOURCE permutation; the
  // DMA writes linearly), so that the rows 4 kk + i a half wave reads (kk in {0, 1} or {2, 3}, fixed i) sit on different halves
  // of the 32 ds_read_b32 banks.  In row order they shared one half: every landing-zone access was 2-way conflicted
  // (rocprofv3: SQ_LDS_BANK_CONFLICT 9-25 % of SQ_LDS_IDX_ACTIVE in the kernels with a zone).
  const unsigned lsrc_z = static_cast<unsigned>((((lane >> 2) & 10) | (((lane >> 2) & 1) << 2) | ((lane >> 4) & 1)) * kT + (lane & 3) * 4) * 4u;
  const float* land_src = BWD ? a.Zsrc : (a.skip ? a.skip : a.add0);
  auto issue_z = [&](int it) {
    if (it >= n_iters) it = n_iters - 1;
    const float* zb = land_src + tile_of(it) * (long long)K * a.rows_p * kT + kPT * half_of(it, cg);
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float* base = uniform_ptr(zb + ((long long)s * a.rows_p + row0 + 16 * mb) * kT);
        float* dst = zl + (s * 2 + mb) * 256;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + lsrc_z),
                                         (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
      }
  };

  // ring bookkeeping: stage counter st = it K + s; reads from slot st % R, stage st + R - 1 is issued into slot (st - 1) % R
  int i_it = 0, i_s = 0, wbuf = 0;
  auto issue_next = [&]() {
    issue(i_it, i_s, wbuf);
    wbuf = wbuf + 1 == R ? 0 : wbuf + 1;
    if (++i_s == K) {
      i_s = 0;
      ++i_it;
    }
  };
  if constexpr (LAND) issue_z(0);
#pragma unroll
  for (int q = 0; q < R - 1; ++q) issue_next();
  int rbuf = 0;

  // per-lane constants of the epilogue
  const unsigned voff = static_cast<unsigned>(4 * kk * kT + p) * 4u;  // rows 4 kk + ii of a 16-row block, point p
  const int frow = 32 * rt + 4 * kk;                                    // element e = 4 mb + ii: block row frow + 16 mb + ii
  bool valid[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) valid[e] = blk_row0 + frow + 16 * (e >> 2) + (e & 3) < a.rows;
  int slot = 0;
  constexpr int kWarmIters = (R - 1 + K - 1) / K + 1;  // iterations whose stages may have fewer epilogues behind them than the steady state
  const int rows_p = a.rows_p;
  auto rrow = [&](int s, int e) { return s * rows_p + row0 + 16 * (e >> 2) + (e & 3); };  // record row of (stream, element)

  PINN_FSTAMP_DECL
  for (int it = 0; it < n_iters; ++it) {
    const long long tile = tile_of(it);
    const int half = half_of(it, cg);
    const long long tile_off = tile * (long long)K * rows_p * kT + kPT * half;
    f32x4 acc[K][2];
    // Records the epilogue reads besides z travel in REGISTERS and are requested before the last stage's MFMAs: a load
    // issued inside the epilogue would expose an HBM round trip per unit (and one issued after the epilogue's first store
    // would wait for that store's round trip too: vmcnt is an in-order counter)
    float ad0[AUX && BWD ? 8 : 1][K];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) acc[s][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < K; ++s) {
      // this wave's pieces of the stage have landed: everything but the N youngest VMEM operations is complete, and at
      // least N are younger than those pieces (the R - 2 later stages' pieces, plus — steady state — the stores of the
      // epilogues in between)
      if (it < kWarmIters) {
        wait_vm<PP*(R - 2)>();
      } else {  // `s` is a literal after unrolling: exactly one of these survives
#define PINN_FUSED_WAIT(S_) if (s == S_) wait_vm<fused_vm_wait<K, PP, BWD, R>(S_ < K ? S_ : 0)>();
        PINN_FUSED_WAIT(0) PINN_FUSED_WAIT(1) PINN_FUSED_WAIT(2) PINN_FUSED_WAIT(3) PINN_FUSED_WAIT(4) PINN_FUSED_WAIT(5) PINN_FUSED_WAIT(6)
#undef PINN_FUSED_WAIT
      }
      PINN_FSTAMP(6);
      lds_barrier();  // all pieces of the stage visible; every wave is done with the previous stage's slot
      PINN_FSTAMP(0);
      issue_next();
      if constexpr (AUX && BWD) {
        if (s == K - 1) {
          const float* base = in_loop(a.add0) + tile_off;
#pragma unroll
          for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int ss = 0; ss < K; ++ss) ad0[e][ss] = rec_ld(base, rrow(ss, e), voff);
        }
      }
      const float* col = ring + rbuf * kStage + cg * (DEPTH * kPT) + lane;
      rbuf = rbuf + 1 == R ? 0 : rbuf + 1;
      float bc[4], bn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bc[i] = col[i * 64];
#pragma unroll
      for (int j4 = 0; j4 < NJ / 4; ++j4) {
        if (j4 + 1 < NJ / 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bn[i] = col[(4 * (j4 + 1) + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc[s][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[mb][4 * j4 + i], bc[i], acc[s][mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (j4 + 1 < NJ / 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bc[i] = bn[i];
        }
      }
      PINN_FSTAMP(1);
    }

    // ---------------------------------------------------------------- epilogue: this wave's 32 rows x 16 points x K streams
    // landing zone, element (mb, ii) of stream s: zr[zoff(s, e)]; row 4 kk + ii of a piece sits at position
    // (kk & 1) + 2 (ii >> 1) + 4 (ii & 1) + 8 (kk >> 1)
    [[maybe_unused]] const float* zr = zl + ((kk & 1) + 8 * (kk >> 1)) * kPT + p;
    auto zoff = [&](int s, int e) { return ((2 * s + (e >> 2)) * 16 + 2 * ((e >> 1) & 1) + 4 * (e & 1)) * kPT; };
    if constexpr (LAND) wait_vm<(K * PP > 63 ? 63 : K * PP)>();  // landed: pieces issued a unit ago, >= K PP ring pieces are younger
    if constexpr (!BWD) {
      float zc[8][K];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + frow);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sbias + frow + 16);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) zc[e][s] = acc[s][e >> 2][e & 3] + (s == 0 ? (e < 4 ? b0[e & 3] : b1[e & 3]) : 0.0f);
      }
      if constexpr (AUX) {
        if (a.add0) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int s = 0; s < K; ++s) zc[e][s] += zr[zoff(s, e)];
        }
      }
      {
        float* out = in_loop(a.Y) + tile_off;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) rec_st(out, rrow(s, e), voff, zc[e][s]);
      }
      PINN_FSTAMP(5);
      LnPoint<NT, NX> S;
      if constexpr (LN)
        fused_ln_stats<NT, NX, RT>(zc, valid, a.rows, a.eps, S, red, slot, rt, lane, p,
                               a.stats ? in_loop(a.stats) + tile * (2LL * K * kT) + kPT * half : nullptr);
      PINN_FSTAMP(3);
      f32x4 g4[2], be4[2];
      if constexpr (LN) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          g4[mb] = *reinterpret_cast<const f32x4*>(sgam + frow + 16 * mb);
          be4[mb] = *reinterpret_cast<const f32x4*>(sbet + frow + 16 * mb);
        }
      }
      float* out = in_loop(a.V) + tile_off;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float pj[K], v[K];
        if constexpr (LN) {
          float y[K];
          ln_yhat<NT, NX>(zc[e], S, y);
          const float gam = g4[e >> 2][e & 3], bet = be4[e >> 2][e & 3];
          pj[0] = fmaf(gam, y[0], bet);
#pragma unroll
          for (int s = 1; s < K; ++s) pj[s] = gam * y[s];
        } else {
#pragma unroll
          for (int s = 0; s < K; ++s) pj[s] = zc[e][s];
        }
        if constexpr (AUX) {
          if (a.skip) {
#pragma unroll
            for (int s = 0; s < K; ++s) pj[s] += zr[zoff(s, e)];
          }
        }
        if (a.has_act) {
          act_fwd<ACT, NT, NX>(a.act_param, pj, v);
        } else {
#pragma unroll
          for (int s = 0; s < K; ++s) v[s] = pj[s];
        }
#pragma unroll
        for (int s = 0; s < K; ++s) rec_st(out, rrow(s, e), voff, valid[e] ? v[s] : 0.0f);
      }
      if constexpr (AUX) {  // the landing zone is free again: request the next unit's skip / add record
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_z(it + 1);
      }
    } else {
      float pb[8][K];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int s = 0; s < K; ++s) pb[e][s] = acc[s][e >> 2][e & 3];
      if constexpr (AUX) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) pb[e][s] += ad0[e][s];
      }
      float skp[8][K];
      if (a.skip) {  // requested together with the LayerNorm sums below: one exposed round trip per unit
        const float* base = in_loop(a.skip) + tile_off;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) skp[e][s] = rec_ld(base, rrow(s, e), voff);
      }
      if (a.add1) {  // second skip-path cotangent (rare): ordinary loads, consumed on the spot
        const float* base = in_loop(a.add1) + tile_off;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) pb[e][s] += rec_ld(base, rrow(s, e), voff);
      }
      LnPoint<NT, NX> S;
      const float invH = 1.0f / (float)a.rows;
      if constexpr (LN) {
        const float* st = in_loop(a.stats) + tile * (2LL * K * kT) + kPT * half;
        float q[K], m[K];
#pragma unroll
        for (int s = 0; s < K; ++s) {
          q[s] = st[s * kT + p];
          m[s] = st[(K + s) * kT + p];
        }
        // centred jets written back in place: the two passes of the adjoint re-read them instead of holding 8 K registers
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) {
            float* zz = const_cast<float*>(zr) + zoff(s, e);
            *zz = valid[e] ? *zz - q[s] * invH : 0.0f;
          }
        ln_point_from_moments<NT, NX>(m, invH, a.eps, S);
      }
      PINN_FSTAMP(4);
      float rb[K];
#pragma unroll
      for (int s = 0; s < K; ++s) rb[s] = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float c[K], pj[K];
#pragma unroll
        for (int s = 0; s < K; ++s) c[s] = zr[zoff(s, e)];
        float y[K];
        if constexpr (LN) {
          ln_yhat<NT, NX>(c, S, y);
          const float bet = sbet[frow + 16 * (e >> 2) + (e & 3)];
          const float gam = sgam[frow + 16 * (e >> 2) + (e & 3)];
          pj[0] = fmaf(gam, y[0], bet);
#pragma unroll
          for (int s = 1; s < K; ++s) pj[s] = gam * y[s];
        } else {
#pragma unroll
          for (int s = 0; s < K; ++s) pj[s] = c[s];
        }
        if (a.skip) {
#pragma unroll
          for (int s = 0; s < K; ++s) pj[s] += skp[e][s];
        }
        if (a.has_act) {
          float zb[K];
          act_bwd<ACT, NT, NX>(a.act_param, pj, pb[e], zb);
#pragma unroll
          for (int s = 0; s < K; ++s) pb[e][s] = zb[s];
        }
#pragma unroll
        for (int s = 0; s < K; ++s) pb[e][s] = valid[e] ? pb[e][s] : 0.0f;
        if constexpr (LN) {  // first half of the LayerNorm adjoint (lm_ew.h::ln_backward): dgamma / dbeta, rbar partial sums
          float gsum = 0.0f;
#pragma unroll
          for (int s = 0; s < K; ++s) gsum = fmaf(pb[e][s], y[s], gsum);
          const float s0 = pts16_sum(valid[e] ? gsum : 0.0f), s1 = pts16_sum(valid[e] ? pb[e][0] : 0.0f);
          if (p == 0) {  // this 16-lane row is the only writer of its feature's slots
            pacc[frow + 16 * (e >> 2) + (e & 3)] += s0;
            pacc[256 + frow + 16 * (e >> 2) + (e & 3)] += s1;
          }
          if (a.Pbar) rec_st(in_loop(a.Pbar) + tile_off, rrow(0, e), voff, pb[e][0]);
          if (a.Pbar) {
#pragma unroll
            for (int s = 1; s < K; ++s) rec_st(in_loop(a.Pbar) + tile_off, rrow(s, e), voff, pb[e][s]);
          }
#pragma unroll
          for (int s = 0; s < K; ++s) pb[e][s] = valid[e] ? sgam[frow + 16 * (e >> 2) + (e & 3)] * pb[e][s] : 0.0f;  // now yhat-bar
          rb[0] = fmaf(c[0], pb[e][0], rb[0]);
#pragma unroll
          for (int k = 1; k <= NT; ++k)
#pragma unroll
            for (int j = 0; j <= k; ++j) rb[sidx(1, k - j)] = fmaf((float)binom(k, j) * c[sidx(1, j)], pb[e][sidx(1, k)], rb[sidx(1, k - j)]);
#pragma unroll
          for (int k = 1; k <= NX; ++k)
#pragma unroll
            for (int j = 0; j <= k; ++j)
              rb[sidx(1 + NT, k - j)] = fmaf((float)binom(k, j) * c[sidx(1 + NT, j)], pb[e][sidx(1 + NT, k)], rb[sidx(1 + NT, k - j)]);
        } else if (a.Pbar) {
#pragma unroll
          for (int s = 0; s < K; ++s) rec_st(in_loop(a.Pbar) + tile_off, rrow(s, e), voff, pb[e][s]);
        }
        if constexpr (LN) __builtin_amdgcn_sched_barrier(0);  // one element at a time: interleaved, the eight raise the register peak into scratch
      }
      if constexpr (LN) {
        group_sum<K, RT>(rb, red, slot, rt, lane, p);
        float v0b = S.g[1] * rb[0];
        float vbt[NT > 0 ? NT : 1], vbx[NX > 0 ? NX : 1];
        if constexpr (NT > 0) {
          float rbd[NT];
#pragma unroll
          for (int k = 0; k < NT; ++k) rbd[k] = rb[1 + k];
          v0b += dir_bwd<NT>(S.g, S.t.v, rbd, vbt);
        }
        if constexpr (NX > 0) {
          float rbd[NX];
#pragma unroll
          for (int k = 0; k < NX; ++k) rbd[k] = rb[1 + NT + k];
          v0b += dir_bwd<NX>(S.g, S.x.v, rbd, vbx);
        }
        const float k2 = 2.0f * invH;
        float mq[K];
#pragma unroll
        for (int s = 0; s < K; ++s) mq[s] = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float c[K], cb[K];
#pragma unroll
          for (int s = 0; s < K; ++s) c[s] = zr[zoff(s, e)];
          cb[0] = fmaf(S.r0, pb[e][0], v0b * k2 * c[0]);
#pragma unroll
          for (int s = 1; s < K; ++s) cb[s] = 0.0f;
#pragma unroll
          for (int k = 1; k <= NT; ++k)
#pragma unroll
            for (int j = 0; j <= k; ++j) {
              const float rr = (k - j == 0) ? S.r0 : S.t.r[k - j - 1];
              cb[sidx(1, j)] += (float)binom(k, j) * (rr * pb[e][sidx(1, k)] + vbt[k - 1] * k2 * c[sidx(1, k - j)]);
            }
#pragma unroll
          for (int k = 1; k <= NX; ++k)
#pragma unroll
            for (int j = 0; j <= k; ++j) {
              const float rr = (k - j == 0) ? S.r0 : S.x.r[k - j - 1];
              cb[sidx(1 + NT, j)] += (float)binom(k, j) * (rr * pb[e][sidx(1 + NT, k)] + vbx[k - 1] * k2 * c[sidx(1 + NT, k - j)]);
            }
#pragma unroll
          for (int s = 0; s < K; ++s) {
            pb[e][s] = valid[e] ? cb[s] : 0.0f;
            mq[s] += pb[e][s];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        group_sum<K, RT>(mq, red, slot, rt, lane, p);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) pb[e][s] = valid[e] ? pb[e][s] - mq[s] * invH : 0.0f;
      }
      PINN_FSTAMP(3);
      // the landing zone is free again (this wave's own reads are done): request the next unit's z, then the stores
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_z(it + 1);
      if constexpr (WRELOAD) load_w();
      {
        float* out = in_loop(a.Zbar) + tile_off;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int s = 0; s < K; ++s) rec_st(out, rrow(s, e), voff, pb[e][s]);
      }
    }
    PINN_FSTAMP(2);
  }
#ifdef PINN_FSTAMPS
  if (a.stamps && lane == 0) {
    fst_acc[7] = pinn_fnow() - fst_begin;
    for (int i = 0; i < 8; ++i) a.stamps[((long long)blockIdx.x * 8 + wave) * 8 + i] += fst_acc[i];
  }
#endif
  // drain the run-ahead DMA before the workgroup's LDS is released, then flush the LayerNorm parameter gradients
  wait_vm<0>();
  if constexpr (BWD && LN) {
    __syncthreads();
    if (a.det_partial) {
      float* P = a.det_partial + (long long)blockIdx.x * (7 * 1024);
      for (int i = tid; i < 32 * RT; i += kFThreads) {
        P[i] = paccb[i] + (CG > 1 ? paccb[512 + i] : 0.0f);
        P[1024 + i] = paccb[256 + i] + (CG > 1 ? paccb[768 + i] : 0.0f);
      }
    } else if (a.d_ln_g) {
      for (int f = tid; f < a.rows; f += kFThreads) {
        atomicAdd(a.d_ln_g + f, paccb[f] + (CG > 1 ? paccb[512 + f] : 0.0f));
        atomicAdd(a.d_ln_b + f, paccb[256 + f] + (CG > 1 ? paccb[768 + f] : 0.0f));
      }
    }
  }
}

inline size_t lm_fused_lds_bytes(int nch, int rt, int K, bool bwd, bool aux) {
  const int cg = 8 / rt;
  const bool land = bwd || aux;
  return sizeof(float) * ((size_t)fused_ring(K, land, cg, nch) * cg * 32 * nch * kPT + fused_fixed_lds_floats(cg) +
                          (land ? (size_t)8 * K * 512 : 0));
}

}  // namespace lm
}  // namespace pinn
