// Stand-alone timing harness for the fused GEMM + prologue kernels (csrc/lm_fused.h): random records, one template
// instance chosen at compile time, HIP-event timing, optional in-kernel phase stamps (-DPINN_FSTAMPS).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I pinns-rl-pde_amd/csrc \
//     -DB_NCH=8 -DB_RT=8 -DB_NT=1 -DB_NX=2 -DB_ACT=0 -DB_LN=1 -DB_BWD=0 [-DPINN_FSTAMPS] tools/micro/fused_bench.hip -o tools/micro/fused_bench
//   tools/micro/fused_bench <points> [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lm_fused.h"

using namespace pinn;
using namespace pinn::lm;

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = scale * ((float)(s >> 8) / 8388608.0f - 1.0f);
  }
  float* d = nullptr;
  if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const long long N = argc > 1 ? atoll(argv[1]) : 100000;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  constexpr int NCH = B_NCH, RT = B_RT, NT = B_NT, NX = B_NX, ACT = B_ACT;
  constexpr bool LN = B_LN, BWD = B_BWD;
  constexpr int K = 1 + NT + NX, DEPTH = 32 * NCH, ROWS = 32 * RT;
  const long long ntiles = (N + 31) / 32;
  const size_t rec_in = (size_t)ntiles * K * DEPTH * 32, rec_out = (size_t)ntiles * K * ROWS * 32;
  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  a.W = dev_rand((size_t)ROWS * DEPTH, 0.06f, 1);
  a.bias = dev_rand(ROWS, 0.1f, 2);
  a.X = dev_rand(rec_in, 1.0f, 3);
  a.rows_p = a.rows = ROWS;
  a.ntiles = ntiles;
  a.ln_g = dev_rand(ROWS, 1.0f, 4);
  a.ln_b = dev_rand(ROWS, 0.1f, 5);
  a.eps = 1e-5f;
  a.has_act = 1;
  a.act_param = 1.0f;
  a.stats = dev_rand((size_t)ntiles * 2 * K * 32, 1.0f, 6);
  if (!BWD) {
    a.Y = dev_rand(rec_out, 0.0f, 7);
    a.V = dev_rand(rec_out, 0.0f, 8);
#ifdef B_SKIP
    a.skip = dev_rand(rec_out, 1.0f, 9);
#endif
  } else {
    a.Zsrc = dev_rand(rec_out, 1.0f, 7);
    a.Zbar = dev_rand(rec_out, 0.0f, 8);
#ifdef B_SKIP
    a.skip = dev_rand(rec_out, 1.0f, 9);
    a.Pbar = dev_rand(rec_out, 0.0f, 10);
    a.add0 = dev_rand(rec_out, 1.0f, 11);
#endif
    a.d_ln_g = dev_rand(ROWS, 0.0f, 12);
    a.d_ln_b = dev_rand(ROWS, 0.0f, 13);
  }
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int gx = (int)(ntiles < cus ? ntiles : cus);
#ifdef PINN_FSTAMPS
  unsigned long long* stamps = nullptr;
  CK(hipMalloc(&stamps, (size_t)gx * 8 * 8 * sizeof(unsigned long long)));
  CK(hipMemset(stamps, 0, (size_t)gx * 8 * 8 * sizeof(unsigned long long)));
  a.stamps = stamps;
#endif
  #ifdef B_SKIP
  constexpr bool AUX = true;
#else
  constexpr bool AUX = false;
#endif
  auto kern = lm_fused<NCH, RT, NT, NX, ACT, LN, BWD, AUX>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t lds = lm_fused_lds_bytes(NCH, RT, K, BWD, AUX);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(kFThreads), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(kFThreads), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0.0f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps;
  const double flop = 2.0 * ROWS * DEPTH * 32.0 * K * ntiles;
  printf("lm_fused<%d,%d,%d,%d,%d,%d,%d> N=%lld tiles=%lld grid=%d lds=%zu: %.1f us  %.1f TFLOP/s (%.3f of 157.3)\n", NCH, RT, NT, NX,
         ACT, (int)LN, (int)BWD, N, ntiles, gx, lds, us, flop / us / 1e6, flop / us / 1e6 / 157.3);
#ifdef PINN_FSTAMPS
  std::vector<unsigned long long> h((size_t)gx * 8 * 8);
  CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double acc[8] = {0};
  for (size_t i = 0; i < h.size(); ++i) acc[i % 8] += (double)h[i];
  const double launches = 3 + reps, waves = (double)gx * 8;
  const char* names[8] = {"wait+barrier", "mfma", "epilogue", "ln_reduce", "epi_loads", "epi_stores", "wait_vm", "total"};
  for (int i = 0; i < 8; ++i) printf("  %-14s %10.0f cycles per wave and launch (%5.1f %%)\n", names[i], acc[i] / launches / waves, 100.0 * acc[i] / acc[7]);
#endif
  return 0;
}
