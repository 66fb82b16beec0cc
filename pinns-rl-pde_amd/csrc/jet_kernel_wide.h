// "Wide" variant of the fused jet kernel: ALL K streams of a tile are resident in LDS at once.
//
// Used when K * HMAX * 36 * 4 B (twice that with the reverse sweep) fits the 160 KB LDS of a CU — e.g. the
// headline Burgers / fourier 4x128 workload (K = 4).  Compared with a stream-serial staging (one stream in LDS at a time),
// which scales to any K and to width 256, this layout
//   * reuses every weight element for all K streams of a k-step (one A operand, K MFMAs) and streams the weight
//     operand from L2 in 32-k chunks one chunk ahead of its MFMAs (16 VGPRs in flight instead of a 64-VGPR fragment),
//   * needs two barriers per layer instead of K,
//   * keeps the weight-gradient tiles of the first three layers in registers across ALL tiles of a workgroup
//     (one flat array of MFMA accumulator tiles) and accumulates db, dw_out, db_out, the first Linear's gradient
//     and the loss per lane / in LDS, so the tile loop issues NO global atomics: everything is flushed once per
//     workgroup (the per-tile flush ran at the memory-side atomic rate with every wave stalled on it — 27 %),
//   * evaluates the H_last -> 1 output layer, the PDE epilogue (redundantly in all 8 lanes that own a point), the
//     output layer's weight gradient and abar = w_out (x) ubar from registers: no LDS image of the last hidden layer,
//     whose tape record is parked in LDS (inside the wave's own rows of the idle a_{l-1} image) instead of the slab,
//   * has a compile-time image height (HMAX) so that every LDS access is base register + immediate offset, and
//     pins each GEMM's operand prefetch one group ahead of its MFMAs with sched_barrier.
// Shared descriptors, accumulator layout and tape words: jet_kernel.h; the layer-major engine (lm_*.h) computes the same arithmetic and tests run both.
#pragma once
#include "jet_kernel.h"

namespace pinn {

constexpr int kPersist = 3;  // layers whose dW accumulators stay in registers

// Encoding parameters staged in LDS once per kernel: ep[c * HMAX + j] = B[c][j] (Fourier, j < M) or W[j][c]
// (first Linear, j < H); ep[kMaxDin * HMAX + j] = bias[j].  Plain 32-bit LDS addressing instead of per-lane
// 64-bit global address arithmetic that the optimizer hoists out of the tile loop and then spills.
template <int HMAX>
__device__ __forceinline__ void stage_enc_params(const NetDev& net, float* ep, int tid) {
  const int din = net.din;
  for (int j = tid; j < HMAX; j += kThreads) {
    if (net.enc == ENC_FOURIER) {
      const int M = net.enc_out >> 1;
      for (int c = 0; c < kMaxDin; ++c) ep[c * HMAX + j] = (c < din && j < M) ? net.encW[c * M + j] : 0.0f;
      ep[kMaxDin * HMAX + j] = 0.0f;
    } else {
      for (int c = 0; c < kMaxDin; ++c) ep[c * HMAX + j] = (c < din && j < net.enc_out) ? net.encW[j * din + c] : 0.0f;
      ep[kMaxDin * HMAX + j] = j < net.enc_out ? net.encb[j] : 0.0f;
    }
  }
}

// pre-activation jets of encoding feature j at point n: value, d/dt (last input column), d/dx (first column)
template <int NT, int NX, int HMAX>
__device__ __forceinline__ void enc_preact_lds(const float* ep, int din, const float* xin, int j, int n, float* z) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int s = 0; s < K; ++s) z[s] = 0.0f;
  float v = ep[kMaxDin * HMAX + j];
  float wt = 0.0f;
#pragma unroll
  for (int c = 0; c < kMaxDin; ++c) {
    const float w = ep[c * HMAX + j];  // zero beyond din
    v = fmaf(xin[c * kT + n], w, v);
    wt = c == din - 1 ? w : wt;
  }
  z[0] = v;
  if constexpr (NT >= 1) z[1] = wt;
  if constexpr (NX >= 1) z[1 + NT] = ep[j];
}

// encoding straight into an LDS activation image dst[s][f][n]; every thread takes (feature, point) pairs
template <int ACT, int NT, int NX, int HMAX>
__device__ __forceinline__ void encode_lds(const NetDev& net, const float* ep, const float* xin, float* dst, int tid) {
  constexpr int K = 1 + NT + NX;
  const int n = tid & 31;
  const int din = net.din;
  if (net.enc == ENC_FOURIER) {
    const int M = net.enc_out >> 1;
#pragma unroll 1
    for (int m = tid >> 5; m < M; m += 8) {
      float z[K], ys[K], yc[K];
      enc_preact_lds<NT, NX, HMAX>(ep, din, xin, m, n, z);
      float sn, cs;
      fast_sincosf(z[0], &sn, &cs);
      const float fs[6] = {sn, cs, -sn, -cs, sn, cs};
      const float fc[6] = {cs, -sn, -cs, sn, cs, -sn};
      ys[0] = sn;
      yc[0] = cs;
      dir_fwd<NT>(fs, z + 1, ys + 1);
      dir_fwd<NX>(fs, z + 1 + NT, ys + 1 + NT);
      dir_fwd<NT>(fc, z + 1, yc + 1);
      dir_fwd<NX>(fc, z + 1 + NT, yc + 1 + NT);
#pragma unroll
      for (int s = 0; s < K; ++s) {
        dst[(s * HMAX + m) * kTP + n] = ys[s];
        dst[(s * HMAX + M + m) * kTP + n] = yc[s];
      }
    }
  } else {
    const int H = net.enc_out;
#pragma unroll 1
    for (int f = tid >> 5; f < H; f += 8) {
      float z[K], y[K];
      enc_preact_lds<NT, NX, HMAX>(ep, din, xin, f, n, z);
      act_fwd<ACT, NT, NX>(net.enc_param, z, y);
#pragma unroll
      for (int s = 0; s < K; ++s) dst[(s * HMAX + f) * kTP + n] = y[s];
    }
  }
}

// reverse, first Linear (din -> H): recompute z from the coordinates, ab <- zbar
template <int ACT, int NT, int NX, int HMAX>
__device__ __forceinline__ void ew_enc_backward_lds(f32x16 (&ab)[1 + NT + NX], const NetDev& net, const float* ep,
                                                    const float* xin, int ft, const Lane& L) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = ft * 32 + acc_row(r, L.lh);
    float z[K], abv[K], zb[K];
    enc_preact_lds<NT, NX, HMAX>(ep, net.din, xin, f, L.ln, z);
#pragma unroll
    for (int s = 0; s < K; ++s) abv[s] = ab[s][r];
    act_bwd<ACT, NT, NX>(net.enc_param, z, abv, zb);
#pragma unroll
    for (int s = 0; s < K; ++s) ab[s][r] = zb[s];
  }
}

// ---------------------------------------------------------------------------
// Layer GEMMs with a STREAMED weight operand.  A wave's slice of the weights (32 rows or columns x depth) is
// fetched from L2 in chunks of 32 k (16 VGPRs), one chunk ahead of the 16 K MFMAs that consume it; only chunk 0
// is requested early (before the activation phase / barrier that precedes the GEMM).  Holding the whole 64-VGPR
// fragment across those phases, as the stream-serial kernel does, pushed this kernel's allocation into scratch.
// ---------------------------------------------------------------------------
struct WChunk {
  f32x4 g[4];
};

// rows form (z = W a): lane (j = ln, h) <- W[32 ft + j][32 c + 8 g + 4 h .. + 3]; lane_off = byte offset of
// W[32 ft + j][4 h], the chunk / group part of the address is wave-uniform
__device__ __forceinline__ unsigned wrows_lane_off(const LayerDev& Ly, int ft, const Lane& L) {
  const int row = (ft * 32 < Ly.out_dim ? ft : 0) * 32 + L.ln;
  return static_cast<unsigned>(row * Ly.ld + 4 * L.lh) * 4u;
}
__device__ __forceinline__ void load_chunk_rows(WChunk& w, const float* W, unsigned lane_off, int c) {
  const char* base = reinterpret_cast<const char*>(W + 32 * c);
#pragma unroll
  for (int g = 0; g < 4; ++g) w.g[g] = *reinterpret_cast<const f32x4*>(base + lane_off + 32u * g);
}

// columns form (abar = W^T zbar): lane (k = ln, h) <- W[32 c + 8 g + 4 h + i][32 kt + k]
__device__ __forceinline__ unsigned wcols_lane_off(const LayerDev& Ly, int kt, const Lane& L) {
  const int col = (kt * 32 < Ly.in_dim ? kt : 0) * 32 + L.ln;
  return static_cast<unsigned>(4 * L.lh * Ly.ld + col) * 4u;
}
__device__ __forceinline__ void load_chunk_cols(WChunk& w, const float* W, int ld, unsigned lane_off, int c) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* base = reinterpret_cast<const char*>(W + (long long)(32 * c + 8 * g + i) * ld);
      w.g[g][i] = *reinterpret_cast<const float*>(base + lane_off);
    }
}

// acc[s] += Wslice . X[s][:][n] for all K streams, depth = 32 NC; one weight element feeds K MFMAs.
// B operands are fetched one k-group (4 ds_read_b32 per stream) ahead of the 4K MFMAs that consume them.
template <int K, int NC, bool COLS>
__device__ __forceinline__ void gemm_wide_n(f32x16 (&acc)[K], const float* W, int ld, unsigned lane_off, const WChunk& w0,
                                            const float* X, int hmax, const Lane& L) {
  const float* col = X + (4 * L.lh) * kTP + L.ln;
  float bc[4][K], bn[4][K];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < K; ++s) bc[i][s] = col[(s * hmax + i) * kTP];
  WChunk cur = w0, nxt;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c + 1 < NC) {
      if constexpr (COLS) load_chunk_cols(nxt, W, ld, lane_off, c + 1);
      else load_chunk_rows(nxt, W, lane_off, c + 1);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int gg = 4 * c + g;
      if (gg + 1 < 4 * NC) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < K; ++s) bn[i][s] = col[(s * hmax + 8 * (gg + 1) + i) * kTP];
      }
      // keep the operand requests of group gg+1 AHEAD of the MFMAs of group gg: left alone, the scheduler sinks
      // each ds_read to just before its MFMA (shorter live ranges) and every MFMA then waits out the LDS latency
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < K; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.g[g][i], bc[i][s], acc[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (gg + 1 < 4 * NC) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < K; ++s) bc[i][s] = bn[i][s];
      }
    }
    if (c + 1 < NC) cur = nxt;
  }
}

// depth = number of k (input features for rows, output features for columns), a multiple of 8
template <int K, bool COLS>
__device__ __forceinline__ void gemm_wide(f32x16 (&acc)[K], const float* W, int ld, unsigned lane_off, const WChunk& w0,
                                          int depth, const float* X, int hmax, const Lane& L) {
  if ((depth & 31) == 0) {
    switch (depth >> 5) {  // wave-uniform
      case 4: gemm_wide_n<K, 4, COLS>(acc, W, ld, lane_off, w0, X, hmax, L); return;
      case 3: gemm_wide_n<K, 3, COLS>(acc, W, ld, lane_off, w0, X, hmax, L); return;
      case 2: gemm_wide_n<K, 2, COLS>(acc, W, ld, lane_off, w0, X, hmax, L); return;
      case 1: gemm_wide_n<K, 1, COLS>(acc, W, ld, lane_off, w0, X, hmax, L); return;
      default: break;
    }
  }
  // odd depths (e.g. 24 Fourier features): plain loop, operands fetched where they are used
  const float* col = X + (4 * L.lh) * kTP + L.ln;
#pragma unroll 1
  for (int g = 0; g < (depth >> 3); ++g) {
    f32x4 w4;
    if constexpr (COLS) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        w4[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(W + (long long)(8 * g + i) * ld) + lane_off);
    } else {
      w4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(W + 8 * g) + lane_off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < K; ++s)
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i], col[(s * hmax + 8 * g + i) * kTP], acc[s], 0, 0, 0);
  }
}

// pt[OFF + kt] += sum_s Z_s[own rows][n] A_s[rows of tile kt][n]^T over all K streams, kt < NA.
// The weight-gradient tiles of all persistent layers are ONE flat register array (pt, NPT tiles) so that a narrow
// first layer does not reserve accumulators it never touches.
template <int K, int NA, int OFF, int NPT>
__device__ __forceinline__ void gemm_outer_wide_n(f32x16 (&pt)[NPT], int ft, const float* Z, const float* A, int hmax,
                                                  const Lane& L) {
  const float* zrow = Z + (ft * 32 + L.ln) * kTP + 4 * L.lh;
  const float* arow = A + L.ln * kTP + 4 * L.lh;
  f32x4 zc, zn, ac[NA], an[NA];
  zc = *reinterpret_cast<const f32x4*>(zrow);
#pragma unroll
  for (int kt = 0; kt < NA; ++kt) ac[kt] = *reinterpret_cast<const f32x4*>(arow + kt * 32 * kTP);
#pragma unroll
  for (int st = 0; st < 4 * K; ++st) {  // step = (stream s, point group g)
    if (st + 1 < 4 * K) {
      const int s = (st + 1) >> 2, g = (st + 1) & 3;
      zn = *reinterpret_cast<const f32x4*>(zrow + s * hmax * kTP + 8 * g);
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) an[kt] = *reinterpret_cast<const f32x4*>(arow + (s * hmax + kt * 32) * kTP + 8 * g);
    }
    __builtin_amdgcn_sched_barrier(0);  // operand requests of step st+1 stay ahead of the MFMAs of step st
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kt = 0; kt < NA; ++kt)
        pt[OFF + kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zc[i], ac[kt][i], pt[OFF + kt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (st + 1 < 4 * K) {
      zc = zn;
#pragma unroll
      for (int kt = 0; kt < NA; ++kt) ac[kt] = an[kt];
    }
  }
}

template <int K, int OFF, int NMAX, int NPT>
__device__ __forceinline__ void gemm_outer_wide(f32x16 (&pt)[NPT], int ft, int in_dim, const float* Z, const float* A,
                                                int hmax, const Lane& L) {
  const int na = (in_dim + 31) >> 5;  // wave-uniform
  if constexpr (NMAX >= 4) {
    if (na >= 4) { gemm_outer_wide_n<K, 4, OFF, NPT>(pt, ft, Z, A, hmax, L); return; }
  }
  if constexpr (NMAX >= 3) {
    if (na == 3) { gemm_outer_wide_n<K, 3, OFF, NPT>(pt, ft, Z, A, hmax, L); return; }
  }
  if constexpr (NMAX >= 2) {
    if (na >= 2) { gemm_outer_wide_n<K, 2, OFF, NPT>(pt, ft, Z, A, hmax, L); return; }
  }
  gemm_outer_wide_n<K, 1, OFF, NPT>(pt, ft, Z, A, hmax, L);
}

// activation jets of tape slot `slot` replayed straight into an LDS image (own feature rows); record groups are
// streamed one ahead, group 0 (`rp0`) was requested by the caller a phase earlier
template <int ACT, int NT, int NX>
__device__ __forceinline__ void ew_replay_lds(const float* tape, int slot, float w, const f32x4 (&rp0)[1 + NT + NX],
                                              float* img, int hmax, int ft, const Lane& L) {
  constexpr int K = 1 + NT + NX;
  f32x4 cur[K], nxt[K];
#pragma unroll
  for (int s = 0; s < K; ++s) cur[s] = rp0[s];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q + 1 < 4) {
#pragma unroll
      for (int s = 0; s < K; ++s) nxt[s] = tape_ld4(tape, slot, 0, 1, K, s, q + 1, L.tid);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float z[K], y[K];
#pragma unroll
      for (int s = 0; s < K; ++s) z[s] = cur[s][i];
      act_fwd_tape<ACT, NT, NX>(w, z, y);
#pragma unroll
      for (int s = 0; s < K; ++s) img[(s * hmax + ft * 32 + acc_row(4 * q + i, L.lh)) * kTP + L.ln] = y[s];
    }
    if (q + 1 < 4) {
#pragma unroll
      for (int s = 0; s < K; ++s) cur[s] = nxt[s];
    }
  }
}

template <int K>
__device__ __forceinline__ void put_tile(float* img, const f32x16 (&v)[K], int hmax, int ft, const Lane& L) {
#pragma unroll
  for (int s = 0; s < K; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) img[(s * hmax + ft * 32 + acc_row(r, L.lh)) * kTP + L.ln] = v[s][r];
}

// The LAST hidden layer's tape record never goes to global memory: it is parked in LDS, inside this wave's OWN feature
// rows of the (still unused) a_{l-1} image — 16 registers x 64 lanes = 4 KB per stream, the rows hold 4.6 KB — as
// 16-byte words [q][lane].  The output-layer step of the reverse sweep reads it back ~100 cycles away instead of
// ~1 us, and the wave's own replay overwrites those rows only after it has consumed the record.
__device__ __forceinline__ float* rec_park(float* img, int hmax, int ft, int s, int q, int tid) {
  return img + (s * hmax + ft * 32) * kTP + (q * 64 + (tid & 63)) * 4;
}

template <int ACT, int NT, int NX>
__device__ __forceinline__ void ew_forward_park(f32x16 (&v)[1 + NT + NX], float w, float* img, int hmax, int ft, int tid) {
  constexpr int K = 1 + NT + NX;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 rec[K];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      float z[K], y[K];
#pragma unroll
      for (int s = 0; s < K; ++s) z[s] = v[s][r];
      act_fwd<ACT, NT, NX>(w, z, y);
      rec[0][i] = ActTape<ACT>::value_is_output ? y[0] : z[0];
#pragma unroll
      for (int s = 1; s < K; ++s) rec[s][i] = z[s];
#pragma unroll
      for (int s = 0; s < K; ++s) v[s][r] = y[s];
    }
#pragma unroll
    for (int s = 0; s < K; ++s) *reinterpret_cast<f32x4*>(rec_park(img, hmax, ft, s, q, tid)) = rec[s];
  }
}

// adjoint of the activation jets of accumulator register r = 4q + i, record group q given as rq[s]
template <int ACT, int NT, int NX>
__device__ __forceinline__ void ew_backward_q1(f32x16 (&ab)[1 + NT + NX], float w, const f32x4 (&rq)[1 + NT + NX], int r) {
  constexpr int K = 1 + NT + NX;
  float z[K], abv[K], zb[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    z[s] = rq[s][r & 3];
    abv[s] = ab[s][r];
  }
  act_bwd_tape<ACT, NT, NX>(w, z, abv, zb);
#pragma unroll
  for (int s = 0; s < K; ++s) ab[s][r] = zb[s];
}

// activation adjoint of tape slot `slot` with the records streamed one 16-byte group ahead of their use;
// rq0 = group 0, requested by the caller a phase earlier (under the preceding GEMM)
template <int ACT, int NT, int NX>
__device__ __forceinline__ void ew_backward_stream(f32x16 (&ab)[1 + NT + NX], float w, const float* tape, int slot,
                                                   const f32x4 (&rq0)[1 + NT + NX], int tid) {
  constexpr int K = 1 + NT + NX;
  f32x4 cur[K], nxt[K];
#pragma unroll
  for (int s = 0; s < K; ++s) cur[s] = rq0[s];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q + 1 < 4) {
#pragma unroll
      for (int s = 0; s < K; ++s) nxt[s] = tape_ld4(tape, slot, 0, 1, K, s, q + 1, tid);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ew_backward_q1<ACT, NT, NX>(ab, w, cur, 4 * q + i);
    if (q + 1 < 4) {
#pragma unroll
      for (int s = 0; s < K; ++s) cur[s] = nxt[s];
    }
  }
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the total): xor-butterfly on quad_perm /
// row_half_mirror / row_mirror, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

template <int OFF, int NMAX, int NPT>
__device__ __forceinline__ void flush_rows(const f32x16 (&pt)[NPT], const LayerDev& Ly, const Lane& L, long long doff, bool store = false) {
  if (!Ly.dW || L.wave * 32 >= Ly.out_dim) return;
  float* base = Ly.dW + (long long)(L.wave * 32 + 4 * L.lh) * Ly.ld + L.ln;
#pragma unroll
  for (int kt = 0; kt < NMAX; ++kt) {
    if (kt * 32 + L.ln < Ly.in_dim) {  // in_dim may end inside a k-tile (e.g. 24 Fourier features)
#pragma unroll
      for (int r = 0; r < 16; ++r) grad_put(base + (long long)((r & 3) + 8 * (r >> 2)) * Ly.ld + kt * 32, pt[OFF + kt][r], doff, store);
    }
  }
}

// HMAX: LDS image height (64 or 128), compile-time so that every LDS access of the straight-line GEMM / activation
// code is base register + immediate offset.  With a run-time height each of the several hundred distinct offsets
// became a loop-invariant address VGPR, and the register allocator spilled them (2.3 KB of scratch per lane).
// NA0: k-tiles (32 input features) of the FIRST MFMA layer's weight-gradient accumulators — 2 for the headline
// network (64 Fourier features -> 128), HMAX / 32 in general.
template <int ACT, int NT, int NX, bool BWD, int HMAX, int NA0>
__global__ __launch_bounds__(kThreads, 1) void jet_kernel_wide(const KernelArgs a) {
  constexpr int K = 1 + NT + NX;
  constexpr int NKT = HMAX / 32;                       // k-tiles of a full-width layer
  constexpr int NPT = BWD ? NA0 + (kPersist - 1) * NKT : 1;  // persistent weight-gradient tiles
  static_assert(kPersist == 3, "the layer -> tile-offset table below is written for three persistent layers");
  constexpr int NTILE = 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDev& net = a.net;
  constexpr int hmax = HMAX;
  constexpr int img = K * hmax * kTP;    // floats per activation image
  float* X = smem;                        // forward activations; zbar in the reverse sweep
  float* A2 = X + (BWD ? img : 0);        // reverse sweep: a_{l-1}
  float* UP = A2 + img;                   // kWaves * K * kT: per-wave partial sums of the output layer
  float* xin = UP + kWaves * K * kT;      // kMaxDin * kT
  float* wb = xin + kMaxDin * kT;         // (1 + n_layers) * hmax: w_out, then the hidden-layer biases
  float* dwo = wb + (1 + net.n_layers) * hmax;  // kWaves * 4 rows * 16: dw_out partials of each 16-lane row
  float* ep = dwo + kThreads;             // (kMaxDin + 1) * hmax: encoding parameters
  float* pl = ep + (kMaxDin + 1) * hmax;  // (kPersist + kMaxDin + 1) * hmax: per-feature db / first-Linear partials (thread tid < hmax owns feature tid)

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.ln = L.tid & 31;
  L.lh = (L.tid >> 5) & 1;
  const int tid = L.tid;
  const int din = net.din;
  const int ft = L.wave;  // this wave's feature tile in every layer
  const int nl = net.n_layers;
  const long long ntiles = (a.N + kT - 1) / kT;
  float* tape = BWD ? a.tape + (long long)blockIdx.x * a.tape_stride : nullptr;
  PINN_STAMP_DECL

  // w_out and the biases are staged once per kernel; a lane's 16 accumulator rows are four 16-byte LDS words
  for (int k = tid; k < hmax; k += kThreads) {
    wb[k] = k < net.h_last ? net.w_out[k] : 0.0f;
    for (int l = 0; l < nl; ++l) wb[(1 + l) * hmax + k] = k < net.layer[l].out_dim ? net.layer[l].b[k] : 0.0f;
  }
  dwo[tid] = 0.0f;
  stage_enc_params<HMAX>(net, ep, tid);
  const float b_out0 = net.b_out[0];
  float* dwo_row = dwo + (tid >> 4) * 16;  // this lane's 16-lane row: [r]
  const bool row_lead = (tid & 15) == 0;
  const int row4 = ft * 32 + 4 * L.lh;  // rows 8q + 4h + i of this lane's tile: wb[.. + row4 + 8q + i]

  // Accumulators that live across ALL tiles of the workgroup and are flushed once: the weight gradients of the
  // first kPersist layers (MFMA accumulator tiles), and per-LANE partial sums of dw_out, db, db_out and the loss
  // whose cross-lane reduction is deferred to the end — no atomics inside the tile loop.
  f32x16 pt[NPT];
  float pdb_out = 0.0f, ploss = 0.0f;
  // per-thread running sums that are touched once per tile live in LDS (own slot per thread), not in VGPRs:
  // pl[p * 256 + tid] = db_p[tid] (p < kPersist), pl[(kPersist + c) * 256 + tid] = first-Linear gradient column c / bias
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < kPersist + kMaxDin + 1; ++i)
      if (tid < hmax) pl[i * hmax + tid] = 0.0f;
  }
  if constexpr (BWD) {
#pragma unroll
    for (int kt = 0; kt < NPT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) pt[kt][r] = 0.0f;
  }

  // coordinates of the NEXT tile travel in registers while the current one is processed
  float xr[kMaxDin] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto fetch_coords = [&](long long tile) {
    if (tid < kT) {
      const long long p = tile * kT + tid;
      const bool ok = tile < ntiles && p < a.N;
#pragma unroll
      for (int cc = 0; cc < kMaxDin; ++cc) {
        if (cc < din - 1) xr[cc] = ok ? a.x[p * (din - 1) + cc] : 0.0f;
        if (cc == din - 1) xr[cc] = ok ? a.t[p] : 0.0f;
      }
    }
  };
  fetch_coords(blockIdx.x);

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long p0 = tile * kT;
    __syncthreads();  // previous tile's readers of xin / X / A2 / UP are done
    if (tid < kT) {
#pragma unroll
      for (int cc = 0; cc < kMaxDin; ++cc) xin[cc * kT + tid] = xr[cc];  // rows >= din are zero (read with zero weights)
    }
    fetch_coords(tile + gridDim.x);
    WChunk w0;  // first weight chunk of the NEXT GEMM to run, requested a phase ahead
    if (nl > 0) {
      const LayerDev L0 = uniform_layer(net.layer[0]);
      load_chunk_rows(w0, L0.W, wrows_lane_off(L0, ft, L), 0);  // latency hides under the encoding
    }
    __syncthreads();
    PINN_STAMP(ST_STAGE);
    encode_lds<ACT, NT, NX, HMAX>(net, ep, xin, X, tid);
    __syncthreads();
    PINN_STAMP(ST_ENCODE);

    // ---- hidden layers ----
    f32x16 acc[K];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
    if (nl == 0 && ft * 32 < net.enc_out) {
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = X[(s * hmax + ft * 32 + acc_row(r, L.lh)) * kTP + L.ln];
    }
    for (int l = 0; l < nl; ++l) {
      const LayerDev Ly = uniform_layer(net.layer[l]);
      const bool on = ft * 32 < Ly.out_dim;
      const bool last = l + 1 == nl;
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
      if (on) {
        gemm_wide<K, false>(acc, Ly.W, Ly.ld, wrows_lane_off(Ly, ft, L), w0, Ly.in_dim, X, hmax, L);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(wb + (1 + l) * hmax + row4 + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[0][4 * q + i] += b4[i];
        }
      }
      PINN_STAMP(ST_FWD_GEMM);
      if (!last) {
        {
          const LayerDev Ln = uniform_layer(net.layer[l + 1]);
          load_chunk_rows(w0, Ln.W, wrows_lane_off(Ln, ft, L), 0);  // latency hides under the jets
        }
        __syncthreads();  // every wave has finished reading X: overwrite in place
        if (on) {
          ew_forward<ACT, NT, NX, NTILE, BWD>(acc, Ly.act_param, tape, l, 0, tid);
          put_tile<K>(X, acc, hmax, ft, L);
        }
        __syncthreads();
      } else if (on) {  // last hidden layer: its activations stay in registers (acc) for the H_last -> 1 layer
        if constexpr (BWD) ew_forward_park<ACT, NT, NX>(acc, Ly.act_param, A2, hmax, ft, tid);
        else ew_forward<ACT, NT, NX, NTILE, false>(acc, Ly.act_param, tape, l, 0, tid);
      }
      PINN_STAMP(ST_FWD_EW);
    }

    // ---- output layer (H_last -> 1) from registers: lane partials -> wave partials in LDS -> every lane sums ----
    {
      float po[K];
#pragma unroll
      for (int s = 0; s < K; ++s) po[s] = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wb + row4 + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < K; ++s) po[s] = fmaf(w4[i], acc[s][4 * q + i], po[s]);
      }
#pragma unroll
      for (int s = 0; s < K; ++s) po[s] += __shfl_xor(po[s], 32);
      if (L.lh == 0) {
#pragma unroll
        for (int s = 0; s < K; ++s) UP[(L.wave * K + s) * kT + L.ln] = po[s];
      }
    }
    __syncthreads();
    PINN_STAMP(ST_OUT);

    f32x4 rp0[K];  // first tape record group of the next replay, requested a phase ahead (reverse sweep)

    // ---- epilogue, evaluated by EVERY lane for its point column n = ln (8 lanes per point, same values) ----
    float ub[K];
    {
      const long long p = p0 + L.ln;
      const bool ok = p < a.N;
      const bool writer = tid < kT;
      float j[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        float v = s == 0 ? b_out0 : 0.0f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += UP[(w * K + s) * kT + L.ln];
        j[s] = v;
      }
      if (a.mode == MODE_JETS) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
          if (writer && ok && a.jets_out[s]) a.jets_out[s][p] = j[s];
          ub[s] = (BWD && ok && a.jets_bar[s]) ? a.jets_bar[s][p] : 0.0f;
        }
      } else {
        float d[K];
        const float r = pde_residual<NT, NX>(a.pde, j, xin[L.ln], d);
        float dl;
        float lt = loss_term(a.pde, r, &dl);
        if (!ok) {
          lt = 0.0f;
          dl = 0.0f;
        }
        if (writer && ok && a.residual_out) a.residual_out[p] = r;
        if (writer) ploss += lt;
        const float rb = !BWD ? 0.0f : (a.res_bar ? (ok ? a.res_bar[p] : 0.0f) : a.grad_scale * dl);
#pragma unroll
        for (int s = 0; s < K; ++s) ub[s] = rb * d[s];
      }
    }
    PINN_STAMP(ST_EPI);

    if constexpr (BWD) {
      // ---- B0: output layer, all in registers.  dw_out partials stay per lane; abar = w_out (x) ub ----
      if (tid < kT) pdb_out += ub[0];
      f32x16 ab[K];
      {
        // a_{L-1} is replayed from the last layer's record (acc is not kept alive across the epilogue), which is
        // parked in LDS (rec_park)
        const bool e_on = nl > 0 && ft * 32 < net.layer[nl - 1].out_dim;
        const float w_last = nl > 0 ? uniform_layer(net.layer[nl - 1]).act_param : 0.0f;
        float gr[16];
        if (e_on) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 cur[K];
#pragma unroll
            for (int s = 0; s < K; ++s) cur[s] = *reinterpret_cast<const f32x4*>(rec_park(A2, hmax, ft, s, q, tid));
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wb + row4 + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * q + i;
              float z[K], y[K];
#pragma unroll
              for (int s = 0; s < K; ++s) z[s] = cur[s][i];
              act_fwd_tape<ACT, NT, NX>(w_last, z, y);
              float g = 0.0f;
#pragma unroll
              for (int s = 0; s < K; ++s) {
                g = fmaf(ub[s], y[s], g);
                ab[s][r] = w4[i] * ub[s];
              }
              gr[r] = g;
              ew_backward_q1<ACT, NT, NX>(ab, w_last, cur, r);
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wb + row4 + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * q + i;
              float g = 0.0f;
#pragma unroll
              for (int s = 0; s < K; ++s) {
                g = fmaf(ub[s], nl == 0 ? acc[s][r] : 0.0f, g);
                ab[s][r] = w4[i] * ub[s];
              }
              gr[r] = g;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gr[r] = row16_sum(gr[r]);
        if (row_lead) {  // ds_add_f32: this lane is the only writer of its row's 16 slots
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(dwo_row + r, gr[r]);
        }
      }
      if (nl >= 2) {
#pragma unroll
        for (int s = 0; s < K; ++s) rp0[s] = tape_ld4(tape, nl - 2, 0, 1, K, s, 0, tid);
      }
      PINN_STAMP(ST_B0);

      for (int l = nl - 1; l >= 0; --l) {
        const LayerDev Ly = uniform_layer(net.layer[l]);
        const bool on = ft * 32 < Ly.out_dim;
        const bool need_abar = l > 0 || net.enc == ENC_LINEAR;
        const bool kon = ft * 32 < Ly.in_dim;  // this wave owns an input-feature tile of the layer
        float pw = 0.0f;                        // act_param of layer l-1
        // the GEMMs of layer l+1 have finished reading X (zbar) and A2; with a single MFMA layer the barrier keeps
        // encode_lds (all rows of A2, below) from overwriting another wave's parked record before its B0 has read it
        if (l + 1 < nl || nl == 1) __syncthreads();
        PINN_STAMP(ST_BWD_PUT);
        if (on) put_tile<K>(X, ab, hmax, ft, L);
        if (l > 0) {
          const LayerDev P = uniform_layer(net.layer[l - 1]);
          pw = P.act_param;
          if (kon) ew_replay_lds<ACT, NT, NX>(tape, l - 1, pw, rp0, A2, hmax, ft, L);  // kon: owns a tile of layer l-1
        } else {
          encode_lds<ACT, NT, NX, HMAX>(net, ep, xin, A2, tid);
        }
        if (need_abar) load_chunk_cols(w0, Ly.W, Ly.ld, wcols_lane_off(Ly, ft, L), 0);  // hides under the barrier
        __syncthreads();
        PINN_STAMP(ST_BWD_EW);
        if (Ly.db && tid < Ly.out_dim) {
          const float g = row_sum(X + tid * kTP);
          if (l < kPersist) pl[l * hmax + tid] += g;
          else grad_add(Ly.db + tid, g, det_row_offset(a));
        }
        // abar_{l-1} = W^T zbar for all streams
        f32x4 rq0[K];
        if (l > 0 && kon) {  // first record group of layer l-1's adjoint: in flight under the GEMM
#pragma unroll
          for (int s = 0; s < K; ++s) rq0[s] = tape_ld4(tape, l - 1, 0, 1, K, s, 0, tid);
        }
        if (need_abar) {
#pragma unroll
          for (int s = 0; s < K; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) ab[s][r] = 0.0f;
          if (kon) gemm_wide<K, true>(ab, Ly.W, Ly.ld, wcols_lane_off(Ly, ft, L), w0, Ly.out_dim, X, hmax, L);
        }
        PINN_STAMP(ST_BWD_DX);
        if (l >= 2) {  // first record group of the NEXT step's replay (layer l-2): in flight under the dW GEMM
#pragma unroll
          for (int s = 0; s < K; ++s) rp0[s] = tape_ld4(tape, l - 2, 0, 1, K, s, 0, tid);
        }
        // dW (persistent for the first kPersist layers), then the activation adjoint of layer l-1
        {
          if (on && Ly.dW) {
            if (l == 0) {
              gemm_outer_wide<K, 0, NA0, NPT>(pt, ft, Ly.in_dim, X, A2, hmax, L);
            } else if (l == 1) {
              gemm_outer_wide<K, NA0, NKT, NPT>(pt, ft, Ly.in_dim, X, A2, hmax, L);
            } else if (l == 2) {
              gemm_outer_wide<K, NA0 + NKT, NKT, NPT>(pt, ft, Ly.in_dim, X, A2, hmax, L);
            } else {
              f32x16 dacc[NKT];
#pragma unroll
              for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[kt][r] = 0.0f;
              gemm_outer_wide<K, 0, NKT, NKT>(dacc, ft, Ly.in_dim, X, A2, hmax, L);
              flush_rows<0, NKT, NKT>(dacc, Ly, L, det_row_offset(a));
            }
          }
          if (l > 0 && kon) ew_backward_stream<ACT, NT, NX>(ab, pw, tape, l - 1, rq0, tid);
        }
        PINN_STAMP(ST_BWD_STREAM);
      }

      // ---- encoding backward (first Linear of feedforward / SIREN) ----
      if (net.enc == ENC_LINEAR && net.d_encW) {
        const int H = net.enc_out;
        __syncthreads();
        if (ft * 32 < H) {
          ew_enc_backward_lds<ACT, NT, NX, HMAX>(ab, net, ep, xin, ft, L);
          put_tile<K>(X, ab, hmax, ft, L);
        }
        __syncthreads();
        if (tid < H) {
          float gb = 0.0f, gt = 0.0f, gx = 0.0f;
          float gw[kMaxDin] = {0.0f, 0.0f, 0.0f, 0.0f};
          for (int n = 0; n < kT; ++n) {
            const float vv = X[tid * kTP + n];
            gb += vv;
#pragma unroll
            for (int cc = 0; cc < kMaxDin; ++cc)
              if (cc < din) gw[cc] = fmaf(vv, xin[cc * kT + n], gw[cc]);
            if constexpr (NT >= 1) gt += X[(1 * hmax + tid) * kTP + n];
            if constexpr (NX >= 1) gx += X[((1 + NT) * hmax + tid) * kTP + n];
          }
#pragma unroll
          for (int cc = 0; cc < kMaxDin; ++cc)
            pl[(kPersist + cc) * hmax + tid] += gw[cc] + (cc == din - 1 ? gt : 0.0f) + (cc == 0 ? gx : 0.0f);
          pl[(kPersist + kMaxDin) * hmax + tid] += gb;
        }
      }
      PINN_STAMP(ST_ENC_BWD);
    }
  }

  // ---- one flush per workgroup ----
  const long long doff = det_row_offset(a);  // deterministic mode: this workgroup's own slab row
  // Store flush (a.flush_store; networks whose MFMA layers are all persistent): the workgroup owns row blockIdx.x of a
  // [grid][stride] slab and WRITES it — 160 KB of gradient tiles at the store rate instead of the memory-side atomic
  // rate (one 256-byte wave-instruction per ~50 ns per CU: 32 us on the critical path of the last workgroups), no memset
  // of the slab; a fixed-order row sum follows (pinn_abi.hip::wide_rows_reduce).
  const bool st = a.flush_store != 0;
  if (a.mode == MODE_PDE && a.loss_sum && L.wave == 0) {
    float sacc = L.lh == 0 ? ploss : 0.0f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    if (tid == 0) grad_put(a.loss_sum, sacc, doff, st);
  }
  if constexpr (BWD) {
    if (net.db_out && L.wave == 0) {
      float g = L.lh == 0 ? pdb_out : 0.0f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) g += __shfl_xor(g, o);
      if (tid == 0) grad_put(net.db_out, g, doff, st);
    }
    if (net.enc == ENC_LINEAR && net.d_encW && tid < net.enc_out) {
#pragma unroll
      for (int cc = 0; cc < kMaxDin; ++cc)
        if (cc < din) grad_put(net.d_encW + tid * din + cc, pl[(kPersist + cc) * hmax + tid], doff, st);
      if (net.d_encb) grad_put(net.d_encb + tid, pl[(kPersist + kMaxDin) * hmax + tid], doff, st);
    }
    if (net.dw_out) {  // dwo[wave][row][r]: rows 0,1 of a wave are the two point halves of lh = 0; 2,3 of lh = 1
      const int f = ft * 32 + acc_row(tid & 15, L.lh);
      if (st) {  // the two lanes (point halves) of a feature: one of them stores the pair's sum
        if (f < net.h_last && (tid & 16) == 0) net.dw_out[doff + f] = dwo[tid] + dwo[tid ^ 16];
      } else if (f < net.h_last) {
        grad_add(net.dw_out + f, dwo[tid], doff);  // two lanes (point halves) per feature: a + b == b + a
      }
    }
#pragma unroll
    for (int p = 0; p < kPersist; ++p) {
      if (p < nl) {
        const LayerDev Lp = uniform_layer(net.layer[p]);
        if (Lp.db && tid < Lp.out_dim) grad_put(Lp.db + tid, pl[p * hmax + tid], doff, st);
        if (p == 0) flush_rows<0, NA0, NPT>(pt, Lp, L, doff, st);
        else if (p == 1) flush_rows<NA0, NKT, NPT>(pt, Lp, L, doff, st);
        else flush_rows<NA0 + NKT, NKT, NPT>(pt, Lp, L, doff, st);
      }
    }
    PINN_STAMP(ST_BWD_FLUSH);
  }
#ifdef PINN_STAMPS
  if (a.stamps && (tid & 63) == 0) {
    st_acc[ST_TOTAL] = pinn_now() - st_begin;
    for (int i = 0; i < kNumStamps; ++i) a.stamps[((long long)blockIdx.x * kWaves + L.wave) * kNumStamps + i] = st_acc[i];
  }
#endif
}

inline size_t jet_wide_lds_bytes(int K, int hmax, bool bwd, int n_layers) {
  return sizeof(float) * ((size_t)(bwd ? 2 : 1) * K * hmax * kTP + kWaves * K * kT + kMaxDin * kT + (1 + n_layers) * hmax + kThreads +
                          (kMaxDin + 1) * hmax + (kPersist + kMaxDin + 1) * hmax);
}

// true if the wide kernel can run this problem (width <= 128, all K streams fit in LDS)
inline int jet_wide_hmax(int hmax) { return hmax <= 64 ? 64 : 128; }  // the compiled image heights

inline bool jet_wide_fits(int K, int hmax, bool bwd, int n_layers) {
  return hmax <= 128 && jet_wide_lds_bytes(K, jet_wide_hmax(hmax), bwd, n_layers) <= 160 * 1024;
}

// one activation's kernels (the library builds one translation unit per stream set AND activation, see the Makefile)
template <int ACT, int NT, int NX>
hipError_t launch_jet_wide_act(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  constexpr int K = 1 + NT + NX;
  const int hm = jet_wide_hmax(a.net.hmax);
  const int na0 = a.net.n_layers > 0 ? (a.net.layer[0].in_dim + 31) / 32 : 4;  // k-tiles of the first MFMA layer
  const size_t lds = jet_wide_lds_bytes(K, hm, bwd, a.net.n_layers);
  hipError_t e = hipSuccess;
#define PINN_WLAUNCH1(BWD_)                                                                                  \
  do {                                                                                                       \
    auto kern = hm == 64 ? jet_kernel_wide<ACT, NT, NX, BWD_, 64, 2>                                         \
                : (BWD_ && na0 <= 2) ? jet_kernel_wide<ACT, NT, NX, BWD_, 128, BWD_ ? 2 : 4>                 \
                                     : jet_kernel_wide<ACT, NT, NX, BWD_, 128, 4>;                           \
    e = allow_full_lds(reinterpret_cast<const void*>(kern));                                                 \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, a);                                    \
  } while (0)
  if (bwd) PINN_WLAUNCH1(true); else PINN_WLAUNCH1(false);
#undef PINN_WLAUNCH1
  return hipGetLastError();
}

// activation id of the network -> index of the compiled activation family (piecewise-linear ones share RELU's kernels)
inline int jet_wide_act_family(const KernelArgs& a) {
  const int act = a.net.n_layers > 0 ? a.net.layer[0].act : a.net.enc_act;
  return (act == PINN_ACT_TANH || act == PINN_ACT_SIN || act == PINN_ACT_GELU || act == PINN_ACT_SIGMOID) ? act : PINN_ACT_RELU;
}

// all activations from one translation unit (developer builds: tanh only)
template <int NT, int NX>
hipError_t launch_jet_wide(const KernelArgs& a, bool bwd, int grid, hipStream_t stream) {
  switch (jet_wide_act_family(a)) {
    case PINN_ACT_TANH: return launch_jet_wide_act<PINN_ACT_TANH, NT, NX>(a, bwd, grid, stream);
#ifndef PINN_DEV
    case PINN_ACT_SIN: return launch_jet_wide_act<PINN_ACT_SIN, NT, NX>(a, bwd, grid, stream);
    case PINN_ACT_GELU: return launch_jet_wide_act<PINN_ACT_GELU, NT, NX>(a, bwd, grid, stream);
    case PINN_ACT_SIGMOID: return launch_jet_wide_act<PINN_ACT_SIGMOID, NT, NX>(a, bwd, grid, stream);
    case PINN_ACT_RELU: return launch_jet_wide_act<PINN_ACT_RELU, NT, NX>(a, bwd, grid, stream);
#endif
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pinn
