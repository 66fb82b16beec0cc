// Dense fp32 MFMA GEMMs of the layer-major engine (see lm_common.h).  gfx950, v_mfma_f32_32x32x2_f32.
//
// The operands are plain matrices: a packed zero-padded weight (rows x cols, both multiples of 32) and records whose
// column blocks are (rows_p x 32) slabs.  Nothing here knows about jets — a stream is just more columns.
//
//   lm_gemm_wres<NCH, RT> Y[cb] = W  X[cb] (+ bias on value-stream blocks) (+ add records), depth <= 256: the weight
//                         slice of every wave stays in registers for the whole launch (the engine's default; W^T has its
//                         own fragment-order copy, so the reverse GEMM is the same kernel)
//   lm_gemm<COLS=false>   the same for any depth, weight slice streamed from L2                   "rows" form
//   lm_gemm<COLS=true>    Y[cb] = W^T X[cb] (+ add records) from the untransposed weight          "cols" form (unused now)
//   lm_gemm_nt / _nt8     dW += sum_cb Z[cb] V[cb]^T ,  db += sum over value-stream blocks of Z[cb] 1
//
// lm_gemm: a 256-thread workgroup stages CB = 2 column blocks of the input (up to 256 reduction rows, 72 KB of LDS:
// two workgroups per CU overlap each other's staging / store phases with MFMA), every wave owns 32-row output tiles,
// the weight operand streams from L2 one 32-deep chunk ahead of its MFMAs (rows form: 16-byte loads along k; cols
// form: 128-byte coalesced dword loads of weight columns, no transposed copy), the B operand is read from LDS one
// k-group ahead (sched_barrier-pinned, see jet_kernel_wide.h for why).
// lm_gemm_nt: a workgroup accumulates a 128 x (<= 256) block of dW in registers over ALL column blocks of its
// split (128 accumulator registers per lane) and flushes once: no atomics inside the loop.
#pragma once
#include "lm_common.h"

namespace pinn {
namespace lm {

constexpr int kCB = 2;     // column blocks per staged image
constexpr int kKC = 256;   // reduction rows per staged chunk

struct GemmArgs {
  const float* W;     // packed weight (w_rows x w_cols) in MFMA fragment order (lm_common.h::frag_index; lm_gemm_wres16: frag16_index)
  const float* bias;  // rows form: w_rows floats or null
  const float* X;     // input record: ncb blocks of (x_rows x 32)
  float* Y;           // output record: ncb blocks of (y_rows x 32)
  const float* add0;  // optional records added to Y (shape of Y)
  const float* add1;
  int w_rows, w_cols;
  int ncb;            // column blocks = tiles * K
  int K;              // streams per tile: bias goes to blocks with cb % K == 0
  int prefetch;       // 0: next stage requested after the MFMAs (under the epilogue); 1: before them
};

struct WFrag16 {
  f32x4 g[4];
};

// W: this wave's row tile in fragment order (1 KB per (chunk, group)); lane_off = 16 * lane
__device__ __forceinline__ void ld_rows(WFrag16& w, const float* W, unsigned lane_off, int ch) {
  const char* base = reinterpret_cast<const char*>(W + 1024 * ch);
#pragma unroll
  for (int g = 0; g < 4; ++g) w.g[g] = *reinterpret_cast<const f32x4*>(base + lane_off + 1024u * g);
}

__device__ __forceinline__ void ld_cols(WFrag16& w, const float* W, int ld, unsigned lane_off, int ch) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* base = reinterpret_cast<const char*>(W + (long long)(32 * ch + 8 * g + i) * ld);
      w.g[g][i] = *reinterpret_cast<const float*>(base + lane_off);
    }
}

// Bias and add records around an accumulator tile.  The first version's epilogue did load-add-store per element; a
// load issued after a store makes s_waitcnt vmcnt wait for that store's round trip (the counter is in order), 16
// times per tile: 165 of 579 us per launch (rocprofv3 ablation, C3).  Now the bias is the tile's start value, and the
// add records (skip-path cotangents: large next to the products, so they are still added LAST, to the finished sum)
// are loaded for all tiles of a wave before its first store.
__device__ __forceinline__ void acc_start(f32x16& acc, const float* bias_rows, int lh) {
  if (bias_rows) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_rows[acc_row(r, lh)];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  }
}

template <int NT>
__device__ __forceinline__ void acc_add_records(f32x16 (&acc)[NT], const float* add, const long long (&base)[NT], const bool (&live)[NT],
                                                int lh) {
  float t[NT][16];
#pragma unroll
  for (int c = 0; c < NT; ++c)
    if (live[c]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) t[c][r] = add[base[c] + acc_row(r, lh) * kT];
    }
#pragma unroll
  for (int c = 0; c < NT; ++c)
    if (live[c]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] += t[c][r];
    }
}

// acc[c] += Wslice(32 x 32 nc) . img[c][0 .. 32 nc)[n]      img: [kCB][kc_rows][kTP]
template <bool COLS>
__device__ __forceinline__ void gemm_core(f32x16 (&acc)[kCB], const float* W, int ld, unsigned lane_off, int nc,
                                          const float* img, int kc_rows, const Lane& L) {
  const float* col = img + (4 * L.lh) * kTP + L.ln;
  float bc[4][kCB], bn[4][kCB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < kCB; ++c) bc[i][c] = col[(c * kc_rows + i) * kTP];
  WFrag16 cur, nxt;
  if constexpr (COLS) ld_cols(cur, W, ld, lane_off, 0);
  else ld_rows(cur, W, lane_off, 0);
#pragma unroll 1
  for (int ch = 0; ch < nc; ++ch) {
    const int chn = ch + 1 < nc ? ch + 1 : ch;  // the last chunk re-requests itself (branch-free, harmless)
    if constexpr (COLS) ld_cols(nxt, W, ld, lane_off, chn);
    else ld_rows(nxt, W, lane_off, chn);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int kn = 32 * ch + 8 * (g + 1);
      if (kn >= 32 * nc) kn = 32 * ch + 8 * g;  // last group of the last chunk: re-read its own rows
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < kCB; ++c) bn[i][c] = col[(c * kc_rows + kn + i) * kTP];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < kCB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.g[g][i], bc[i][c], acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < kCB; ++c) bc[i][c] = bn[i][c];
    }
    cur = nxt;
  }
}

// NTILE: 32-row output tiles per wave (1: 128-row workgroup blocks, 2: 256-row blocks)
//
// The loop runs over STAGES (item, reduction chunk).  The rows of stage q + 1 are requested into registers before the
// MFMAs of stage q and written to LDS after them: without that, every workgroup of the chip staged, multiplied and
// stored in lockstep (identical work per item, simultaneous start), HBM saw 32 MB bursts separated by idle gaps and
// the matrix pipe sat at 53 % (rocprofv3: SQ_WAIT_ANY 36 % with both waves of a SIMD parked together).
template <bool COLS, int NTILE>
__global__ __launch_bounds__(kThreads, 2) void lm_gemm(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int out_rows = COLS ? a.w_cols : a.w_rows;
  const int depth = COLS ? a.w_rows : a.w_cols;
  const int x_rows = depth, y_rows = out_rows;
  const int row_blk = blockIdx.y * (128 * NTILE);
  const int items = (a.ncb + kCB - 1) / kCB;
  const int kc_rows = depth < kKC ? depth : kKC;
  const int nk = (depth + kKC - 1) / kKC;                       // reduction chunks per item
  const int my_items = items > (int)blockIdx.x ? (items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nstages = my_items * nk;

  constexpr int kPre = kCB * kKC * 8 / kThreads;  // float4 per thread for one stage (16)
  f32x4 pre[kPre];
  auto stage_item = [&](int q) { return (int)blockIdx.x + (q / nk) * (int)gridDim.x; };
  auto fetch = [&](int q) {
    const int cb0 = stage_item(q) * kCB, k0 = (q % nk) * kKC;
    const int kn = depth - k0 < kKC ? depth - k0 : kKC;
    const int quads = kn * 8;
#pragma unroll
    for (int c = 0; c < kCB; ++c) {
      // unconditional, clamped loads (see lm_gemm_nt8): a column block past the end re-reads the last one and is never stored
      const float* src = a.X + ((long long)(cb0 + c < a.ncb ? cb0 + c : a.ncb - 1) * x_rows + k0) * kT;
#pragma unroll
      for (int u = 0; u < kPre / kCB; ++u) {
        int q4 = u * kThreads + L.tid;
        q4 = q4 < quads ? q4 : quads - 1;
        pre[c * (kPre / kCB) + u] = *reinterpret_cast<const f32x4*>(src + 4 * q4);
      }
    }
  };
  auto stash = [&](int q) {
    const int k0 = (q % nk) * kKC;
    const int kn = depth - k0 < kKC ? depth - k0 : kKC;
    const int quads = kn * 8;
#pragma unroll
    for (int c = 0; c < kCB; ++c)
#pragma unroll
      for (int u = 0; u < kPre / kCB; ++u) {
        const int q4 = u * kThreads + L.tid;
        if (q4 < quads) *reinterpret_cast<f32x4*>(smem + (c * kc_rows + (q4 >> 3)) * kTP + 4 * (q4 & 7)) = pre[c * (kPre / kCB) + u];
      }
  };

  f32x16 acc[NTILE][kCB];
  auto start_tiles = [&](int q) {  // accumulators of the item that begins at stage q
    const int cb0 = stage_item(q) * kCB;
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) {
      const int row0 = row_blk + (L.wave + kWaves * jt) * 32;
#pragma unroll
      for (int c = 0; c < kCB; ++c) {
        const int cb = cb0 + c;
        const bool with_bias = row0 < out_rows && cb < a.ncb && !COLS && a.bias && (cb % a.K) == 0;
        acc_start(acc[jt][c], with_bias ? a.bias + row0 : nullptr, L.lh);
      }
    }
  };
  if (nstages > 0) {
    fetch(0);
    stash(0);
    start_tiles(0);
  }
  __syncthreads();
  for (int q = 0; q < nstages; ++q) {
    // vmcnt is an in-order queue: requested BEFORE the MFMAs, the next stage's rows sit in front of every weight
    // chunk this stage streams (each chunk then waits for them once); requested AFTER, they overlap only the
    // epilogue stores and the barrier.  Both forms are kept; the engine picks per launch (a.prefetch).
    if (a.prefetch && q + 1 < nstages) fetch(q + 1);
    const int k0 = (q % nk) * kKC;
    const int kn = depth - k0 < kKC ? depth - k0 : kKC;
#pragma unroll
    for (int jt = 0; jt < NTILE; ++jt) {
      const int row0 = row_blk + (L.wave + kWaves * jt) * 32;
      if (row0 < out_rows) {
        if constexpr (COLS) {
          const unsigned lane_off = static_cast<unsigned>(4 * L.lh * a.w_cols + row0 + L.ln) * 4u;
          gemm_core<true>(acc[jt], a.W + (long long)k0 * a.w_cols, a.w_cols, lane_off, kn >> 5, smem, kc_rows, L);
        } else {
          const unsigned lane_off = static_cast<unsigned>(L.tid & 63) * 16u;
          const float* wt = a.W + ((long long)(row0 >> 5) * (a.w_cols >> 5) + (k0 >> 5)) * 1024;  // row tile, first chunk of this stage
          gemm_core<false>(acc[jt], wt, a.w_cols, lane_off, kn >> 5, smem, kc_rows, L);
        }
      }
    }
    if (!a.prefetch && q + 1 < nstages) fetch(q + 1);
    if ((q % nk) == nk - 1) {
      // epilogue: add records (all loads first), then stores only; 128-byte row segments per half wave
      const int cb0 = stage_item(q) * kCB;
#pragma unroll
      for (int jt = 0; jt < NTILE; ++jt) {
        const int row0 = row_blk + (L.wave + kWaves * jt) * 32;
        if (row0 < out_rows) {
          long long base[kCB];
          bool live[kCB];
#pragma unroll
          for (int c = 0; c < kCB; ++c) {
            live[c] = cb0 + c < a.ncb;
            base[c] = ((long long)(cb0 + c) * y_rows + row0) * kT + L.ln;
          }
          if (a.add0) acc_add_records<kCB>(acc[jt], a.add0, base, live, L.lh);
          if (a.add1) acc_add_records<kCB>(acc[jt], a.add1, base, live, L.lh);
#pragma unroll
          for (int c = 0; c < kCB; ++c)
            if (live[c]) {
#pragma unroll
              for (int r = 0; r < 16; ++r) a.Y[base[c] + acc_row(r, L.lh) * kT] = acc[jt][c][r];
            }
        }
      }
      if (q + 1 < nstages) start_tiles(q + 1);
    }
    __syncthreads();  // every wave has finished reading this stage's image
    if (q + 1 < nstages) stash(q + 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-resident rows-form GEMM.  Y[cb] = W X[cb] with reduction depth 32 NCH <= 256.
//
// The weights of a layer are tiny next to its batch, yet lm_gemm re-streams a wave's 32 x depth slice from L2 for
// every staged image (and exposes the first chunk's latency each time): its matrix pipe sits at 58-66 % (rocprofv3,
// profiles/r02_C3.md).  Here a 512-thread workgroup (8 waves, one per CU) keeps each wave's slice IN REGISTERS for the
// whole launch (16 NCH VGPRs, 128 at depth 256), double-buffers the staged column blocks in LDS (one barrier per
// stage; the next image's rows are in flight during the MFMAs), and the inner loop is nothing but ds_read_b32 with
// immediate offsets + MFMA.  Waves: RT row tiles x CG = 8 / RT column groups, each wave 2 column blocks per stage.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWresThreads = 512;

template <int NCH, int RT>
__global__ __launch_bounds__(kWresThreads, 1) void lm_gemm_wres(const GemmArgs a) {
  constexpr int CG = 8 / RT;
  constexpr int SB = 2 * CG;                      // column blocks per stage
  constexpr int DEPTH = 32 * NCH;
  constexpr int kBlk = DEPTH * kT;                // floats per staged block: unpadded rows, the image of an LDS-DMA copy
  constexpr int kPieces = SB * DEPTH / 8;         // 1 KB DMA pieces (8 rows) per stage
  constexpr int kPpw = kPieces / 8;               // per wave (8 at depth 256 / SB 2)
  static_assert(kPieces % 8 == 0, "stage must divide over the eight waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][SB][DEPTH][32]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = tid & 31, lh = (tid >> 5) & 1;
  const int rt = wave % RT, cg = wave / RT;
  const int row0 = (int)blockIdx.y * (32 * RT) + 32 * rt;
  const bool row_ok = row0 < a.w_rows;
  const int y_rows = a.w_rows;
  const int items = (a.ncb + SB - 1) / SB;
  const int my_items = items > (int)blockIdx.x ? (items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;

  WFrag16 w[NCH];
  {
    const unsigned lane_off = static_cast<unsigned>(tid & 63) * 16u;
    const float* wt = a.W + (long long)((row_ok ? row0 : 0) >> 5) * NCH * 1024;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) ld_rows(w[ch], wt, lane_off, ch);
  }

  // Staging by LDS-DMA (global_load_lds_dwordx4): a column block is DEPTH x 32 contiguous floats in the record and in
  // the image alike, so a stage is a plain copy in 1 KB pieces, wave w taking pieces w, w + 8, ... — no prefetch
  // registers, no ds_write pass.  Each half wave of a B-operand read covers one whole 128-byte row: no bank conflicts
  // without padding.
  const unsigned loff = static_cast<unsigned>(tid & 63) * 16u;
  auto stage_piece = [&](int q, int buf, int u) {
    const int cb0 = ((int)blockIdx.x + q * (int)gridDim.x) * SB;
    const int p = 8 * u + wave;
    const int c = p / (DEPTH / 8), rg = p % (DEPTH / 8);
    const int cb = cb0 + c < a.ncb ? cb0 + c : a.ncb - 1;  // clamped, unconditional (never stored past the end)
    const float* base = uniform_ptr(a.X + (long long)cb * kT * DEPTH + rg * 256);
    float* dst = smem + buf * (SB * kBlk) + c * kBlk + rg * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + loff),
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
  };

  if (my_items > 0) {
#pragma unroll
    for (int u = 0; u < kPpw; ++u) stage_piece(0, 0, u);
  }
  float* sbias = smem + 2 * SB * kBlk;  // this block's 32 RT bias values
  if (tid < 32 * RT) {
    const int row = (int)blockIdx.y * (32 * RT) + tid;
    sbias[tid] = a.bias && row < a.w_rows ? a.bias[row] : 0.0f;
  }
  __syncthreads();
  f32x16 prev[2];
  float* prev_base[2] = {a.Y, a.Y};  // wave-uniform block bases; the lane part of every address is `voff`
  bool prev_live[2] = {false, false};
  const unsigned voff = static_cast<unsigned>(4 * lh * kT + ln) * 4u;
  auto row_ptr = [&](float* base, int r) {  // accumulator register r -> its row of the block (immediate offset)
    return reinterpret_cast<float*>(reinterpret_cast<char*>(base + ((r & 3) + 8 * (r >> 2)) * kT) + voff);
  };
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[c][r] = 0.0f;
  for (int q = 0; q < my_items; ++q) {
    f32x16 acc[2];
    {
      const int cb0 = ((int)blockIdx.x + q * (int)gridDim.x) * SB + 2 * cg;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int cb = cb0 + c;
        acc_start(acc[c], row_ok && cb < a.ncb && a.bias && (cb % a.K) == 0 ? sbias + 32 * rt : nullptr, lh);
      }
    }
    const float* col = smem + (q & 1) * (SB * kBlk) + (2 * cg) * kBlk + (4 * lh) * kT + ln;
    float bc[4][2], bn[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) bc[i][c] = col[c * kBlk + i * kT];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        constexpr int last = 32 * NCH - 8;
        const int kn = (32 * ch + 8 * g) < last ? 32 * ch + 8 * (g + 1) : last;  // the final group re-reads itself
        // the PREVIOUS stage's tiles leave during the first half of this stage's MFMAs (kSt stores per k-group)
        constexpr int kGroupsHalf = 2 * NCH;
        constexpr int kSt = 32 / kGroupsHalf;
        const int gi = 4 * ch + g;
        if (gi < kGroupsHalf) {
#pragma unroll
          for (int e = 0; e < kSt; ++e) {
            const int idx = gi * kSt + e, c = idx >> 4, r = idx & 15;
            if (prev_live[c]) *row_ptr(prev_base[c], r) = prev[c][r];
          }
        } else {  // second half: the next stage's DMA pieces, one per k-group, behind the stores (lands before the barrier)
          const int u = gi - kGroupsHalf;
          if (u < kPpw && q + 1 < my_items) stage_piece(q + 1, (q + 1) & 1, u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) bn[i][c] = col[c * kBlk + (kn + i) * kT];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ch].g[g][i], bc[i][c], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) bc[i][c] = bn[i][c];
      }
    }
    {  // hand the finished tiles over: add records now (all loads before any store), stores during the next stage
      const int cb0 = ((int)blockIdx.x + q * (int)gridDim.x) * SB + 2 * cg;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        prev_live[c] = row_ok && cb0 + c < a.ncb;
        prev_base[c] = uniform_ptr(a.Y + ((long long)(cb0 + c) * y_rows + row0) * kT);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {  // one tile at a time: 16 temporaries
        if (!prev_live[c]) continue;
        const long long blk = prev_base[c] - a.Y;
        if (a.add0) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = *row_ptr(const_cast<float*>(a.add0) + blk, r);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][r] += t[r];
        }
        if (a.add1) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = *row_ptr(const_cast<float*>(a.add1) + blk, r);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][r] += t[r];
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) prev[c] = acc[c];
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)  // the last stage's tiles
    if (prev_live[c]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) *row_ptr(prev_base[c], r) = prev[c][r];
    }
}

inline size_t lm_gemm_wres_lds_bytes(int nch, int rt) { return sizeof(float) * (2u * (2u * (8 / rt)) * (32u * nch) * kT + 32u * rt); }

// ---------------------------------------------------------------------------------------------------------------
// Weight-resident rows-form GEMM for reduction depths 257 .. 512 (attention's 4 H -> H Linear and its transpose on the
// way back, width-512 networks): v_mfma_f32_16x16x4_f32, so that a wave's slice is 16 rows x depth = 128 VGPRs at depth
// 512 (the 32 x 32 x 2 form needs 32 rows: 256).  A 512-thread workgroup owns 128 output rows (wave w: rows 16 w ..
// 16 w + 15) and ONE column block per stage, 64 KB at depth 512, double-buffered by LDS-DMA; per k-step a lane reads the
// two 16-point halves of reduction row 4 j + (lane >> 4) (ds_read_b32 x 2, immediate offsets; a half wave covers banks
// 0-15 and 32-47: conflict-free without padding) and issues two MFMAs.  Weight in lm_common.h::frag16_index order.
// lm_gemm on these shapes: 68 % MFMA busy (profiles/r03_C5.md: the weight slice re-streamed from L2 per staged image,
// two 72 KB workgroups per CU in near-lockstep).
// ---------------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(kWresThreads, 1) void lm_gemm_wres16(const GemmArgs a) {
  constexpr int DEPTH = 32 * NCH;
  constexpr int NJ = DEPTH / 4;      // k-steps
  constexpr int kBlk = DEPTH * kT;   // floats per column block
  constexpr int kPpw = DEPTH / 64;   // 1 KB DMA pieces (8 reduction rows) per wave and stage
  static_assert(NCH % 4 == 0 && NCH <= 16, "depth in 128-row steps up to 512");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][DEPTH][32] + 128 bias values
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, kk = lane >> 4;
  const int row0 = (int)blockIdx.y * 128 + 16 * wave;
  const bool row_ok = row0 < a.w_rows;
  const int y_rows = a.w_rows;
  const int my_items = a.ncb > (int)blockIdx.x ? (a.ncb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (my_items == 0) return;

  float w[NJ];  // w[j] = W[row0 + (lane & 15)][4 j + (lane >> 4)]
  {
    const int r0 = row_ok ? row0 : 0;
    const float* wt = a.W + (long long)((r0 >> 5) * 2 + ((r0 >> 4) & 1)) * (NJ / 4) * 256;
#pragma unroll
    for (int j4 = 0; j4 < NJ / 4; ++j4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wt + (j4 * 64 + lane) * 4);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) w[4 * j4 + jj] = v[jj];
    }
  }

  // A DMA piece lands linearly (lane l at base + 16 l: 8 reduction rows x 8 chunks of 4 points); the SOURCE chunk of a lane is
  // XOR-swizzled so that ODD rows hold their two 16-point halves swapped: ds_read_b32 banks are (address / 4) mod 32 per
  // 32-lane half, and a half wave of a B read is rows 4 j + {0, 1} (or {2, 3}) x 16 points of one half — unswizzled, both
  // rows on banks 0-15: every read 2-way conflicted (rocprofv3: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).
  const unsigned loff = static_cast<unsigned>((lane & ~7) | ((lane & 7) ^ (((lane >> 3) & 1) * 4))) * 16u;
  auto stage = [&](int q, int buf) {  // a column block is one contiguous slab in the record and in LDS alike
    const long long cb = (long long)blockIdx.x + (long long)q * gridDim.x;
#pragma unroll
    for (int u = 0; u < kPpw; ++u) {
      const int pc = 8 * u + wave;
      const float* base = uniform_ptr(a.X + cb * kBlk + pc * 256);
      float* dst = smem + buf * kBlk + pc * 256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + loff),
                                       (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    }
  };
  stage(0, 0);
  float* sbias = smem + 2 * kBlk;
  if (tid < 128) {
    const int row = (int)blockIdx.y * 128 + tid;
    sbias[tid] = a.bias && row < a.w_rows ? a.bias[row] : 0.0f;
  }
  __syncthreads();  // drains vmcnt: stage 0 has landed

  // accumulator register (nb, i): row row0 + 4 kk + i, point 16 nb + p
  const unsigned voff = static_cast<unsigned>(4 * kk * kT + p) * 4u;
  auto elem = [&](float* blk, int nb, int i) { return reinterpret_cast<float*>(reinterpret_cast<char*>(blk + i * kT + 16 * nb) + voff); };
  f32x4 prev[2];
  float* prev_blk = a.Y;
  bool prev_live = false;
  auto stage_piece = [&](int q, int buf, int u) {
    const long long cb = (long long)blockIdx.x + (long long)q * gridDim.x;
    const int pc = 8 * u + wave;
    const float* base = uniform_ptr(a.X + cb * kBlk + pc * 256);
    float* dst = smem + buf * kBlk + pc * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + loff),
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
  };
  for (int q = 0; q < my_items; ++q) {
    const long long cb = (long long)blockIdx.x + (long long)q * gridDim.x;
    f32x4 acc[2];
    {
      const bool bias_blk = a.bias && (cb % a.K) == 0;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + 16 * wave + 4 * kk);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[nb][i] = bias_blk ? b4[i] : 0.0f;
    }
    float* blk = uniform_ptr(a.Y + (cb * y_rows + row0) * kT);
    const long long blk_off = blk - a.Y;
    const bool more = q + 1 < my_items;
    float t0[8], t1[8];
    const float* col0 = smem + (q & 1) * kBlk + kk * kT + 16 * (kk & 1) + p;        // points p of rows 4 j + kk
    const float* col1 = smem + (q & 1) * kBlk + kk * kT + 16 * ((kk & 1) ^ 1) + p;  // points 16 + p
    // B operand two k-groups ahead of its MFMAs (LDS latency under the DMA's writes); the VMEM work of a stage is spread
    // over its k-groups, one or two instructions each, so that the matrix pipe starts right after the barrier: the
    // previous stage's tile leaves (groups 0-7), then the next stage is requested (its slot was read last in iteration
    // q - 1, which every wave has left), then this tile's add records (needed after the last group, before any store).
    float bq[3][4][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bq[g][i][0] = col0[(4 * g + i) * (4 * kT)];
        bq[g][i][1] = col1[(4 * g + i) * (4 * kT)];
      }
#pragma unroll
    for (int j4 = 0; j4 < NJ / 4; ++j4) {
      if (j4 + 2 < NJ / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bq[(j4 + 2) % 3][i][0] = col0[(4 * (j4 + 2) + i) * (4 * kT)];
          bq[(j4 + 2) % 3][i][1] = col1[(4 * (j4 + 2) + i) * (4 * kT)];
        }
      }
      if (j4 < 8) {
        if (prev_live) *elem(prev_blk, j4 & 1, j4 >> 1) = prev[j4 & 1][j4 >> 1];  // the two 64-byte halves of a row back to back
      } else if (j4 < 8 + kPpw) {
        if (more) stage_piece(q + 1, (q + 1) & 1, j4 - 8);
      } else if (j4 < 16 + kPpw) {
        const int e = j4 - 8 - kPpw;
        if (row_ok && a.add0) t0[e] = *elem(const_cast<float*>(a.add0) + blk_off, e >> 2, e & 3);
        if (row_ok && a.add1) t1[e] = *elem(const_cast<float*>(a.add1) + blk_off, e >> 2, e & 3);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * j4 + i], bq[j4 % 3][i][nb], acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (row_ok) {
      if (a.add0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e >> 2][e & 3] += t0[e];
      }
      if (a.add1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e >> 2][e & 3] += t1[e];
      }
    }
    prev[0] = acc[0];
    prev[1] = acc[1];
    prev_blk = blk;
    prev_live = row_ok;
    __syncthreads();  // the next stage has landed (vmcnt drained); this slot is free
  }
  if (prev_live) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) *elem(prev_blk, nb, i) = prev[nb][i];
  }
}

inline size_t lm_gemm_wres16_lds_bytes(int nch) { return sizeof(float) * (2u * (32u * nch) * kT + 128u); }

inline size_t lm_gemm_lds_bytes(int depth) { return sizeof(float) * (size_t)kCB * (depth < kKC ? depth : kKC) * kTP; }

// ---------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------
struct GemmNtArgs {
  const float* Z;   // cotangent record: ncb blocks of (z_rows x 32)        z_rows = out features (padded)
  const float* V;   // GEMM-input record: ncb blocks of (v_rows x 32)       v_rows = in features (padded)
  float* dW;        // packed gradient (z_rows x v_rows), accumulated with float atomics at the end
  float* db;        // packed (z_rows) or null
  float* partial;   // deterministic mode: per-split partial blocks [(split)][z_rows x v_rows (+ z_rows)] or null
  int z_rows, v_rows;
  int ncb, K;
};

constexpr int kNtCols = 128;  // dW columns per workgroup (64 accumulator registers per lane)

template <int NA>
__device__ __forceinline__ void outer_block(f32x16 (&dacc)[kNtCols / 32], const float* zrow, const float* arow) {
  f32x4 zc, zn, ac[NA], an[NA];
  zc = *reinterpret_cast<const f32x4*>(zrow);
#pragma unroll
  for (int kt = 0; kt < NA; ++kt) ac[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g + 1 < 4) {
      zn = *reinterpret_cast<const f32x4*>(zrow + 8 * (g + 1));
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) an[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP + 8 * (g + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[i], ac[kt][i], dacc[kt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < 4) {
      zc = zn;
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) ac[kt] = an[kt];
    }
  }
}

// grid = (splits, ceil(z_rows / 128), ceil(v_rows / kNtCols))
__global__ __launch_bounds__(kThreads, 2) void lm_gemm_nt(const GemmNtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zi = smem;                // [128][kTP]
  float* Vi = smem + 128 * kTP;    // [kNtCols][kTP]
  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int zr0 = blockIdx.y * 128, vc0 = blockIdx.z * kNtCols;
  const int zn = a.z_rows - zr0 < 128 ? a.z_rows - zr0 : 128;          // rows of Z staged (multiple of 32)
  const int vn = a.v_rows - vc0 < kNtCols ? a.v_rows - vc0 : kNtCols;  // rows of V staged
  const int na = vn >> 5;
  const bool on = L.wave * 32 < zn;
  constexpr int NKT = kNtCols / 32;
  f32x16 dacc[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
  float dbacc = 0.0f;
  const bool do_db = a.db && blockIdx.z == 0;

  for (int cb = blockIdx.x; cb < a.ncb; cb += gridDim.x) {
    __syncthreads();
    {
      const float* zs = a.Z + ((long long)cb * a.z_rows + zr0) * kT;
      const float* vs = a.V + ((long long)cb * a.v_rows + vc0) * kT;
      const int zq = zn * 8, vq = vn * 8;
      for (int q0 = 0; q0 < zq + vq; q0 += 4 * kThreads) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * kThreads + L.tid;
          if (q < zq) v[u] = *reinterpret_cast<const f32x4*>(zs + 4 * q);
          else if (q < zq + vq) v[u] = *reinterpret_cast<const f32x4*>(vs + 4 * (q - zq));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * kThreads + L.tid;
          if (q < zq) *reinterpret_cast<f32x4*>(Zi + (q >> 3) * kTP + 4 * (q & 7)) = v[u];
          else if (q < zq + vq) *reinterpret_cast<f32x4*>(Vi + ((q - zq) >> 3) * kTP + 4 * ((q - zq) & 7)) = v[u];
        }
      }
    }
    __syncthreads();
    if (do_db && (cb % a.K) == 0 && L.tid < zn) dbacc += row_sum(Zi + L.tid * kTP);
    if (on) {
      const float* zrow = Zi + (L.wave * 32 + L.ln) * kTP + 4 * L.lh;
      const float* arow = Vi + L.ln * kTP + 4 * L.lh;
      switch (na) {  // wave-uniform
        case 1: outer_block<1>(dacc, zrow, arow); break;
        case 2: outer_block<2>(dacc, zrow, arow); break;
        case 3: outer_block<3>(dacc, zrow, arow); break;
        default: outer_block<4>(dacc, zrow, arow); break;
      }
    }
  }
  // one flush per workgroup
  if (a.partial) {  // deterministic mode: plain stores of this split's partial; lm_reduce_partials sums them in order
    float* P = a.partial + (long long)blockIdx.x * ((long long)a.z_rows * a.v_rows + a.z_rows);
    if (on) {
      float* base = P + (long long)(zr0 + L.wave * 32 + 4 * L.lh) * a.v_rows + vc0 + L.ln;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        if (kt < na) {
#pragma unroll
          for (int r = 0; r < 16; ++r) base[(long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32] = dacc[kt][r];
        }
    }
    if (do_db && L.tid < zn) P[(long long)a.z_rows * a.v_rows + zr0 + L.tid] = dbacc;
    return;
  }
  if (on) {
    float* base = a.dW + (long long)(zr0 + L.wave * 32 + 4 * L.lh) * a.v_rows + vc0 + L.ln;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt < na) {
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32, dacc[kt][r]);
      }
  }
  if (do_db && L.tid < zn) atomicAdd(a.db + zr0 + L.tid, dbacc);
}

// Width >= 256: eight waves own a 256 x 256 block of dW (128 accumulator registers per lane at two waves per SIMD),
// so every staged byte of Z and V feeds twice the MFMAs of the 128 x 128 form (64 instead of 32 FLOP per byte —
// the 128-wide blocks were bound by record traffic, not by the matrix pipe).  The next column block's rows travel in
// registers while the current one is multiplied (one barrier per column block).
// grid = (splits, ceil(z_rows / 256), ceil(v_rows / 256)), 512 threads.
constexpr int kNt8 = 256;

__global__ __launch_bounds__(512, 2) void lm_gemm_nt8(const GemmNtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kImg = 2 * kNt8 * kTP;  // floats per buffer: Z rows then V rows
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = tid & 31, lh = (tid >> 5) & 1;
  const int zr0 = blockIdx.y * kNt8, vc0 = blockIdx.z * kNt8;
  const int zn = a.z_rows - zr0 < kNt8 ? a.z_rows - zr0 : kNt8;
  const int vn = a.v_rows - vc0 < kNt8 ? a.v_rows - vc0 : kNt8;
  const int na = vn >> 5;
  const bool on = wave * 32 < zn;
  f32x16 dacc[8];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
  float dbacc = 0.0f;
  const bool do_db = a.db && blockIdx.z == 0;
  const int zq = zn * 8, vq = vn * 8;  // float4 per staged block
  f32x4 pre[8];
  auto fetch = [&](int cb) {
    const float* zs = a.Z + ((long long)cb * a.z_rows + zr0) * kT;
    const float* vs = a.V + ((long long)cb * a.v_rows + vc0) * kT;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // ONE UNCONDITIONAL load per register (address clamped, surplus lanes re-read the last quad): predicated loads
      // into a register make the compiler drain vmcnt before each of them, which serialises the whole prefetch
      int q = u * 512 + tid;
      q = q < zq + vq ? q : zq + vq - 1;
      const float* src = q < zq ? zs + 4 * q : vs + 4 * (q - zq);
      pre[u] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto stash = [&](float* img) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = u * 512 + tid;
      if (q < zq) *reinterpret_cast<f32x4*>(img + (q >> 3) * kTP + 4 * (q & 7)) = pre[u];
      else if (q < zq + vq) *reinterpret_cast<f32x4*>(img + (kNt8 + ((q - zq) >> 3)) * kTP + 4 * ((q - zq) & 7)) = pre[u];
    }
  };
  int buf = 0;
  int cb = blockIdx.x;
  if (cb < a.ncb) {
    fetch(cb);
    stash(smem);
  }
  __syncthreads();
  for (; cb < a.ncb; cb += gridDim.x) {
    const int nxt = cb + gridDim.x;
    if (nxt < a.ncb) fetch(nxt);  // in flight under the MFMAs below
    const float* img = smem + buf * kImg;
    if (do_db && (cb % a.K) == 0 && tid < zn) dbacc += row_sum(img + tid * kTP);
    if (on) {
      const float* zrow = img + (wave * 32 + ln) * kTP + 4 * lh;
      const float* arow = img + (kNt8 + ln) * kTP + 4 * lh;
      // operands of point group g: one row segment of Z and eight of V per lane; the reads of group g + 1 are issued
      // behind the first half of group g's MFMAs (a full second set of eight would push the prefetch registers of
      // the next column block into scratch, and a spilled prefetch waits for its loads on the spot)
      f32x4 zc, ac[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        zc = *reinterpret_cast<const f32x4*>(zrow + 8 * g);
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          if (kt < na) ac[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP + 8 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kt = 0; kt < 8; ++kt)
            if (kt < na) dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[i], ac[kt][i], dacc[kt], 0, 0, 0);
      }
    }
    if (nxt < a.ncb) stash(smem + (buf ^ 1) * kImg);
    __syncthreads();
    buf ^= 1;
  }
  if (a.partial) {
    float* P = a.partial + (long long)blockIdx.x * ((long long)a.z_rows * a.v_rows + a.z_rows);
    if (on) {
      float* base = P + (long long)(zr0 + wave * 32 + 4 * lh) * a.v_rows + vc0 + ln;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        if (kt < na) {
#pragma unroll
          for (int r = 0; r < 16; ++r) base[(long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32] = dacc[kt][r];
        }
    }
    if (do_db && tid < zn) P[(long long)a.z_rows * a.v_rows + zr0 + tid] = dbacc;
    return;
  }
  if (on) {
    float* base = a.dW + (long long)(zr0 + wave * 32 + 4 * lh) * a.v_rows + vc0 + ln;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
      if (kt < na) {
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32, dacc[kt][r]);
      }
  }
  if (do_db && tid < zn) atomicAdd(a.db + zr0 + tid, dbacc);
}

// ---------------------------------------------------------------------------------------------------------------
// lm_gemm_nt8 with LDS-DMA staging (global_load_lds_dwordx4): the staged Z and V blocks go from HBM straight into
// LDS — no prefetch registers, no ds_write pass — which leaves room to double-buffer the MFMA operands (the
// register-staged kernel spends 32 VGPRs on the in-flight block: its MFMA loop alone runs at 80 %, staging adds
// 20 % on top).  A DMA writes 1 KB per wave-instruction LINEARLY (lane l lands at base + 16 l), so the image is
// unpadded [row][32 points]; bank conflicts of the row-segment reads are avoided by an XOR swizzle of the 16-byte chunk
// index with (row >> 1) & 7 — applied to the per-lane SOURCE address of the DMA and to the reads alike (an
// involution; 16 consecutive rows then cover all 64 banks).
// ---------------------------------------------------------------------------------------------------------------
// One 256 x 256 block of dW (rows zr0.., columns vc0..) accumulated over the column blocks split, split + nsplits, ...
__device__ __forceinline__ void nt8d_block(const GemmNtArgs& a, const int zr0, const int vc0, const bool do_db, const int split,
                                           const int nsplits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kBlkF = kNt8 * kT;      // floats per staged block (256 rows x 32 points)
  constexpr int kImg = 2 * kBlkF;       // Z block then V block
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, ln = tid & 31, lh = (tid >> 5) & 1;
  f32x16 dacc[8];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
  float dbacc = 0.0f;

  // DMA piece p = 8 u + wave (u = 0..7): pieces 0..31 are 8-row groups of Z, 32..63 of V; lane l -> row 8 p' + (l >> 3),
  // physical chunk l & 7 = logical chunk ^ ((row >> 1) & 7).  (row >> 1) & 7 depends on the piece only through its
  // parity = the wave's, so the lane part of every source address is ONE 32-bit offset beside a uniform base.
  const int rloc = lane >> 3, pc = lane & 7;
  const unsigned loff = static_cast<unsigned>(rloc * kT + 4 * (pc ^ ((4 * (wave & 1) + (rloc >> 1)) & 7))) * 4u;
  auto stage = [&](int cb, float* img) {
    const float* zs = a.Z + ((long long)cb * a.z_rows + zr0) * kT;
    const float* vs = a.V + ((long long)cb * a.v_rows + vc0) * kT;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int p = 8 * u + wave;
      const float* base = uniform_ptr((u < 4 ? zs : vs) + (64 * (u & 3) + 8 * wave) * kT);  // piece p' = 8 (u & 3) + wave
      float* dst = img + p * 256;  // 1 KB per piece, wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + loff),
                                       (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    }
  };

  const int sw = (ln >> 1) & 7;  // read-side swizzle of this lane's rows (Z row wave * 32 + ln, V rows kt * 32 + ln)
  int buf = 0;
  int cb = split;
  if (cb < a.ncb) stage(cb, smem);
  __syncthreads();
  for (; cb < a.ncb; cb += nsplits) {
    const int nxt = cb + nsplits;
    const float* img = smem + buf * kImg;
    if (nxt < a.ncb) stage(nxt, smem + (buf ^ 1) * kImg);  // lands under the MFMAs below
    if (do_db && (cb % a.K) == 0 && tid < kNt8) {
      const f32x4* row = reinterpret_cast<const f32x4*>(img + tid * kT);
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v = row[q ^ ((tid >> 1) & 7)];  // any order sums the row; this one spreads a 16-lane group over all 16 slots
        s += (v[0] + v[1]) + (v[2] + v[3]);
      }
      dbacc += s;
    }
    {
      const float* zrow = img + (wave * 32 + ln) * kT;
      const float* arow = img + kBlkF + ln * kT;
      f32x4 zc, zn4, ac[8], an[8];
      auto load_ops = [&](int g, f32x4& z, f32x4 (&av)[8]) {
        const int off = ((2 * g + lh) ^ sw) * 4;
        z = *reinterpret_cast<const f32x4*>(zrow + off);
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) av[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kT + off);
      };
      load_ops(0, zc, ac);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) load_ops(g + 1, zn4, an);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kt = 0; kt < 8; ++kt) dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[i], ac[kt][i], dacc[kt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 4) {
          zc = zn4;
#pragma unroll
          for (int kt = 0; kt < 8; ++kt) ac[kt] = an[kt];
        }
      }
    }
    __syncthreads();  // drains this stage's DMA (vmcnt) and frees the buffer just read
    buf ^= 1;
  }
  if (a.partial) {
    float* P = a.partial + (long long)split * ((long long)a.z_rows * a.v_rows + a.z_rows);
    float* base = P + (long long)(zr0 + wave * 32 + 4 * lh) * a.v_rows + vc0 + ln;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32] = dacc[kt][r];
    if (do_db && tid < kNt8) P[(long long)a.z_rows * a.v_rows + zr0 + tid] = dbacc;
    return;
  }
  float* base = a.dW + (long long)(zr0 + wave * 32 + 4 * lh) * a.v_rows + vc0 + ln;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32, dacc[kt][r]);
  if (do_db && tid < kNt8) atomicAdd(a.db + zr0 + tid, dbacc);
}


__global__ __launch_bounds__(512, 2) void lm_gemm_nt8d(const GemmNtArgs a) {  // z_rows and v_rows multiples of 256 only
  nt8d_block(a, blockIdx.y * kNt8, blockIdx.z * kNt8, a.db && blockIdx.z == 0, blockIdx.x, gridDim.x);
}

// All 256-multiple weight gradients of a reverse sweep in ONE launch, after the sweep (every Zbar / V record of the chunk is
// still in the workspace): grid (splits, jobs), a job = one 256 x 256 block of one layer's dW.  What it saves is the
// flush: a workgroup's 256 KB block goes out as 65 536 float atomics, 64 MB per launch chip-wide, 41 of the 449 us of a
// per-layer launch on C3 (rocprofv3, kernel without its flush) — paid once per chunk instead of once per layer — and the
// per-launch ramp.
constexpr int kMaxNtJobs = 48;
struct GemmNtBatch {
  int n, ncb, K;
  struct Job {
    const float* Z;
    const float* V;
    float* dW;
    float* db;  // null unless this block owns the bias gradient of its rows
    int z_rows, v_rows, zr0, vc0;
  } job[kMaxNtJobs];
};

__global__ __launch_bounds__(512, 2) void lm_gemm_nt8d_batch(const GemmNtBatch b) {
  const GemmNtBatch::Job& j = b.job[blockIdx.y];
  GemmNtArgs a;
  a.Z = j.Z;
  a.V = j.V;
  a.dW = j.dW;
  a.db = j.db;
  a.partial = nullptr;
  a.z_rows = j.z_rows;
  a.v_rows = j.v_rows;
  a.ncb = b.ncb;
  a.K = b.K;
  nt8d_block(a, j.zr0, j.vc0, j.db != nullptr, blockIdx.x, gridDim.x);
}

// The same kernel for blocks of width 128 (attention: dW of 512 x 128 and 128 x 512 Linears, 128 x 128 of the merged
// value / projection; any width-128 network): ZR x VR block of dW per workgroup, ZR, VR in {128, 256}.  ZR = 256: wave w owns
// row tile w and all VR columns (VR / 32 accumulator tiles); ZR = 128: waves = 4 row tiles x 2 column halves (VR / 64 tiles
// each).  Round 2 ran these shapes on the register-staged kernels (lm_gemm_nt 54 %, lm_gemm_nt8 61 % MFMA busy on C5).
template <int ZR, int VR>
__device__ __forceinline__ void ntd_block(const GemmNtArgs& a, const int zr0, const int vc0, const bool do_db, const int split,
                                          const int nsplits) {  // z_rows % ZR == 0 and v_rows % VR == 0
  static_assert((ZR == 128 || ZR == 256) && (VR == 128 || VR == 256), "block shapes");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kZF = ZR * kT, kVF = VR * kT, kImg = kZF + kVF;  // floats: Z block then V block
  constexpr int kPieces = (ZR + VR) / 8, kPpw = kPieces / 8;     // 1 KB pieces (8 rows) per stage / per wave
  constexpr int NA = ZR == 256 ? VR / 32 : VR / 64;              // accumulator tiles per wave
  static_assert(kPieces % 8 == 0, "a stage must divide over the eight waves");
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, ln = tid & 31, lh = (tid >> 5) & 1;
  const int rtile = ZR == 256 ? wave : (wave & 3);
  const int cbase = ZR == 256 ? 0 : (wave >> 2) * (VR / 2);  // first dW column of this wave inside the block
  f32x16 dacc[NA];
#pragma unroll
  for (int kt = 0; kt < NA; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
  float dbacc = 0.0f;
  const int rloc = lane >> 3, pc = lane & 7;
  const unsigned loff = static_cast<unsigned>(rloc * kT + 4 * (pc ^ ((4 * (wave & 1) + (rloc >> 1)) & 7))) * 4u;
  auto stage = [&](int cb, float* img) {
    const float* zs = a.Z + ((long long)cb * a.z_rows + zr0) * kT;
    const float* vs = a.V + ((long long)cb * a.v_rows + vc0) * kT;
#pragma unroll
    for (int u = 0; u < kPpw; ++u) {
      const int p = 8 * u + wave;  // piece: 8 rows of Z (p < ZR / 8) or of V
      const bool isz = p < ZR / 8;
      const float* base = uniform_ptr(isz ? zs + (8 * p) * kT : vs + (8 * (p - ZR / 8)) * kT);
      float* dst = img + p * 256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + loff),
                                       (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    }
  };
  const int sw = (ln >> 1) & 7;
  int buf = 0;
  int cb = split;
  if (cb < a.ncb) stage(cb, smem);
  __syncthreads();
  for (; cb < a.ncb; cb += nsplits) {
    const int nxt = cb + nsplits;
    const float* img = smem + buf * kImg;
    if (nxt < a.ncb) stage(nxt, smem + (buf ^ 1) * kImg);
    if (do_db && (cb % a.K) == 0 && tid < ZR) {
      const f32x4* row = reinterpret_cast<const f32x4*>(img + tid * kT);
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v = row[q ^ ((tid >> 1) & 7)];  // any order sums the row; this one spreads a 16-lane group over all 16 slots
        s += (v[0] + v[1]) + (v[2] + v[3]);
      }
      dbacc += s;
    }
    {
      const float* zrow = img + (rtile * 32 + ln) * kT;
      const float* arow = img + kZF + (cbase + ln) * kT;
      f32x4 zc, zn4, ac[NA], an[NA];
      auto load_ops = [&](int g, f32x4& z, f32x4 (&av)[NA]) {
        const int off = ((2 * g + lh) ^ sw) * 4;
        z = *reinterpret_cast<const f32x4*>(zrow + off);
#pragma unroll
        for (int kt = 0; kt < NA; ++kt) av[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kT + off);
      };
      load_ops(0, zc, ac);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) load_ops(g + 1, zn4, an);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kt = 0; kt < NA; ++kt) dacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[i], ac[kt][i], dacc[kt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 4) {
          zc = zn4;
#pragma unroll
          for (int kt = 0; kt < NA; ++kt) ac[kt] = an[kt];
        }
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  const long long row = zr0 + rtile * 32 + 4 * lh;
  if (a.partial) {
    float* P = a.partial + (long long)split * ((long long)a.z_rows * a.v_rows + a.z_rows);
    float* base = P + row * a.v_rows + vc0 + cbase + ln;
#pragma unroll
    for (int kt = 0; kt < NA; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32] = dacc[kt][r];
    if (do_db && tid < ZR) P[(long long)a.z_rows * a.v_rows + zr0 + tid] = dbacc;
    return;
  }
  float* base = a.dW + row * a.v_rows + vc0 + cbase + ln;
#pragma unroll
  for (int kt = 0; kt < NA; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(base + (long long)((r & 3) + 8 * (r >> 2)) * a.v_rows + kt * 32, dacc[kt][r]);
  if (do_db && tid < ZR) atomicAdd(a.db + zr0 + tid, dbacc);
}

template <int ZR, int VR>
__global__ __launch_bounds__(512, 2) void lm_gemm_ntd(const GemmNtArgs a) {
  ntd_block<ZR, VR>(a, blockIdx.y * ZR, blockIdx.z * VR, a.db && blockIdx.z == 0, blockIdx.x, gridDim.x);
}

// every ZR x VR block of a reverse sweep's weight gradients in one launch (see lm_gemm_nt8d_batch): grid (splits, jobs)
template <int ZR, int VR>
__global__ __launch_bounds__(512, 2) void lm_gemm_ntd_batch(const GemmNtBatch b) {
  const GemmNtBatch::Job& j = b.job[blockIdx.y];
  GemmNtArgs a;
  a.Z = j.Z;
  a.V = j.V;
  a.dW = j.dW;
  a.db = j.db;
  a.partial = nullptr;
  a.z_rows = j.z_rows;
  a.v_rows = j.v_rows;
  a.ncb = b.ncb;
  a.K = b.K;
  ntd_block<ZR, VR>(a, j.zr0, j.vc0, j.db != nullptr, blockIdx.x, gridDim.x);
}

inline size_t lm_gemm_ntd_lds_bytes(int zr, int vr) { return sizeof(float) * (size_t)2 * (zr + vr) * kT; }

inline size_t lm_gemm_nt8d_lds_bytes() { return sizeof(float) * (size_t)2 * 2 * kNt8 * kT; }

inline size_t lm_gemm_nt8_lds_bytes() { return sizeof(float) * (size_t)2 * 2 * kNt8 * kTP; }

// deterministic mode: out[i] += sum over splits (fixed order) of partial[s][i]
__global__ void lm_reduce_partials(const float* partial, long long stride, int splits, float* dW, long long nW, float* db,
                                   int nb) {
  const long long total = nW + (db ? nb : 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int k = 0; k < splits; ++k) s += partial[(long long)k * stride + i];
    if (i < nW) dW[i] += s;
    else db[i - nW] += s;
  }
}

inline size_t lm_gemm_nt_lds_bytes() { return sizeof(float) * (size_t)(128 + kNtCols) * kTP; }

}  // namespace lm
}  // namespace pinn
