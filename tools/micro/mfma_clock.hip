// Sustained fp32-MFMA rate and shader clock of the device under a pure matrix-pipe load (no memory traffic):
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
// Prints TFLOP/s of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 with 8 waves per CU (the occupancy of this repository's
// GEMM kernels) and the clock64() / wall_clock64() ratio = shader clock in units of the 100 MHz constant clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512, 1) void spin(float* out, long long* clk, int iters) {
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
  const long long c0 = clock64(), w0 = wall_clock64();
  float r = 0.0f;
  if constexpr (KIND == 0) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    for (int k = 0; k < 4; ++k) r += acc[k][0];
  } else {
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k)
      for (int i = 0; i < 4; ++i) acc[k][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    for (int k = 0; k < 8; ++k) r += acc[k][0];
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = w1 - w0;
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out;
  long long* clk;
  (void)hipMalloc(&out, sizeof(float) * cus * 512);
  (void)hipMalloc(&clk, sizeof(long long) * cus * 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 4; ++rep) {
      const int iters = rep < 2 ? 2000 : 40000;  // ~2.6 ms / ~52 ms at 2 GHz
      (void)hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(spin<0>, dim3(cus), dim3(512), 0, 0, out, clk, iters);
      else hipLaunchKernelGGL(spin<1>, dim3(cus), dim3(512), 0, 0, out, clk, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      long long h[2];
      (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
      const double flop = (double)cus * 8 /*waves*/ * iters * (kind == 0 ? 32.0 * 4096 : 64.0 * 2048);
      printf("%s iters %6d: %8.3f ms  %7.1f TFLOP/s  clock64/wall_clock64 = %.3f (shader clock %.0f MHz if the constant clock is 100 MHz)\n",
             kind == 0 ? "32x32x2" : "16x16x4", iters, ms, flop / ms * 1e-9, (double)h[0] / (double)h[1], 100.0 * h[0] / h[1]);
    }
  }
  printf("CUs %d, clockRate %d kHz\n", cus, p.clockRate);
  return 0;
}
