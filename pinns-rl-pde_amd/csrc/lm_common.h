// Layer-major ("LM") engine — common definitions.  gfx950.
//
// The fused tile-major kernels (jet_kernel_wide.h) keep a whole network's state of a 32-point tile on one CU.  That
// stops working where the state outgrows the CU: width 256 with 4-5 streams needs 750 KB for one layer's reverse
// step against 672 KB of registers + LDS, LayerNorm couples all features of a point, and widths such as 124 or 512
// fit no 32-row MFMA tiling of the live weights.  This engine runs the same arithmetic LAYER BY LAYER instead:
// every quantity that crosses a layer boundary is a *record* in HBM, every layer is a handful of plain launches
//
//     V = prologue(sources)            lm_ew.h    element-wise jets: [LayerNorm] -> [+ skip record] -> [activation]
//     Y = W V + b [+ add record]       lm_gemm.h  dense fp32 MFMA GEMM over (stream, point) columns
//     ...                              lm_head.h  H -> 1 output layer + PDE residual + loss + cotangent seed
//     Vbar = W^T Zbar [+ adds]         lm_gemm.h
//     dW  += Zbar V^T, db += Zbar 1    lm_gemm.h  (accumulated over all tiles of a launch in registers)
//     Zbar_prev = prologue^T(Vbar)     lm_ew.h    (+ LayerNorm / encoder parameter gradients)
//
// and a batch is processed in chunks of ~1 GB per record (bounded workspace; measured: bigger chunks are faster, the
// records are streamed from HBM, not held in the Infinity Cache).  The host side (lm_engine.hip) turns a PinnNetDesc
// into that launch list; tests/jet_model.py::net_program is its executable specification.
//
// Record layout: R[tile][stream s][feature f][32 points], f < Hp = features rounded up to 32 (padding rows are
// written as zeros).  One (tile, stream) pair is a *column block*: a contiguous Hp x 32 fp32 slab, which is at the
// same time one B-operand block of the GEMMs and 128-byte-row coalesced for the element-wise kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "jet_kernel.h"  // f32x16 / f32x4, kT, kTP, acc_row, Lane, PdeDev, activation jets

namespace pinn {
namespace lm {

constexpr int kMaxNodes = 40;   // GEMM nodes of one network (attention: 4 per layer)
constexpr int kMaxPack = 184;   // tensors in one pack / unpack launch (state_dict entries + one transposed copy per GEMM weight)

__host__ __device__ inline int round32(int v) { return (v + 31) & ~31; }

// ---------------------------------------------------------------------------------------------------------------
// parameter packing: every tensor the LM kernels read is a zero-padded, 16-byte aligned copy in the workspace
// (rows x cols -> rows_p x cols_p), refreshed at the start of each call; gradients are accumulated in a packed
// twin and added to the caller's tensors by one unpack launch at the end.
// ---------------------------------------------------------------------------------------------------------------
struct PackItem {
  const float* src;  // caller's tensor (rows x cols, row-major), null = skip
  float* user_grad;  // caller's gradient tensor (same shape), null = none
  unsigned off;      // offset in floats inside the packed block
  int rows, cols;    // logical shape
  int cols_p;        // padded row length (rows_p is implied by the next offset)
  int rows_p;
  int transpose;     // 0: packed[r][c] = src[r][c];  1: packed[c][r] = src[r][c] (Fourier B (din, M) -> [M][4]);
                     // 2 / 3: MFMA A-fragment order of src / of src^T (see frag_index): what lm_gemm streams
                     // 4 / 5: the same in the 16 x 16 x 4 order of the fused kernels (frag16_index)
};

struct PackTable {
  int n;
  PackItem item[kMaxPack];
};

// MFMA A-operand ("fragment") order of a padded (R_p x C_p) matrix: the 16 bytes lane (ln, lh) of a wave needs for
// row tile rt, 32-deep reduction chunk ch, group g — A[32 rt + ln][32 ch + 8 g + 4 lh + 0..3] — are stored at
// ((((rt * nch + ch) * 4 + g) * 64 + 32 lh + ln) * 4 floats, so one wave-instruction reads 1 KB of CONTIGUOUS memory
// (8 cache lines, all bytes used) instead of 32 bytes from each of 32 rows (32 lines, a quarter of each used).
__host__ __device__ inline long long frag_index(int r, int c, int nch) {
  const int rt = r >> 5, ln = r & 31, ch = c >> 5, g = (c & 31) >> 3, lh = (c & 7) >> 2, i = c & 3;
  return ((((long long)(rt * nch + ch) * 4 + g) * 64 + lh * 32 + ln) * 4 + i);
}

// A-operand order of v_mfma_f32_16x16x4_f32 for the fused kernels (lm_fused.h): lane l = 16 kk + lr of the wave that owns
// row tile rt needs, for 16-row block mb and k-step j, A[32 rt + 16 mb + lr][4 j + kk]; the four k-steps 4 j4 .. 4 j4 + 3
// of a lane are one 16-byte word, the 64 lanes of a (block, j4) pair 1 KB of consecutive memory.  nj4 = columns / 16.
__host__ __device__ inline long long frag16_index(int r, int c, int nj4) {
  const int rt = r >> 5, mb = (r >> 4) & 1, lr = r & 15, j = c >> 2, kk = c & 3;
  return ((((long long)(rt * 2 + mb) * nj4 + (j >> 2)) * 64 + kk * 16 + lr) * 4 + (j & 3));
}

}  // namespace lm
}  // namespace pinn
