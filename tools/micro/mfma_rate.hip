// Sustained v_mfma_f32_32x32x2_f32 rate on this GPU: waves per SIMD x independent accumulators, with and without an
// LDS operand read per MFMA.  Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(512, 1) void burn(float* out, int iters) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = 1e-3f * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  const float* col = sm + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (LDS) y = col[(u * 36 + it) & 4095];
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(int threads, const char* name) {
  float* out;
  hipMalloc(&out, 256 * 8 * 512 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {256, 512}) {
    burn<NACC, LDS><<<blocks, threads>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burn<NACC, LDS><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 16.0 * NACC * 4096.0;
    printf("%-28s threads %4d blocks %3d: %.3f ms  %.1f TFLOP/s\n", name, threads, blocks, ms, flops / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  run<2, false>(256, "2 acc, regs only");
  run<4, false>(256, "4 acc, regs only");
  run<2, false>(512, "2 acc, regs only");
  run<4, false>(512, "4 acc, regs only");
  run<2, true>(512, "2 acc, 1 ds_read / 2 mfma");
  run<4, true>(256, "4 acc, 1 ds_read / 4 mfma");
  run<1, false>(512, "1 acc (dependent chain)");
  run<1, false>(1024, "1 acc, 4 waves/SIMD");
  return 0;
}
