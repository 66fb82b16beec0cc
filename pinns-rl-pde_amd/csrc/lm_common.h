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
// and a batch is processed in chunks so that the short-lived records stay in the 256 MB Infinity Cache.  The host
// side (lm_engine.hip) turns a PinnNetDesc into that launch list; tests/jet_model.py::net_program is its executable
// specification.
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
constexpr int kMaxPack = 144;   // tensors in one pack / unpack launch (state_dict entries + one transposed copy per GEMM weight)

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
  int transpose;     // 1: packed[c][r] = src[r][c]  (Fourier B is (din, M); packed as [M][4])
};

struct PackTable {
  int n;
  PackItem item[kMaxPack];
};

}  // namespace lm
}  // namespace pinn
