// What read+write bandwidth do the element-wise kernels' access patterns get on this GPU?  Records of
// [tile][row 0..R)[32 points] fp32 (R = K * Hp = 1024 rows per tile here), NR read records + NW written records.
//   A  flat float4 per lane, grid-stride (the textbook copy)
//   B  16-point half-tile units, thread = (n = tid & 15, g = tid >> 4), rows g + 64 i: dword per lane, 64-byte segments
//   C  32-point tile units, thread = (n = tid & 31, g = tid >> 5): dword per lane, 128-byte segments
//   D  32-point tile units, thread = (quad = tid & 7, g = tid >> 3): float4 per lane, 128-byte rows
// All loads of a unit are issued before its first store, 1024-thread workgroups, 2 per CU, persistent loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int R = 1024;

template <int NR, int NW>
__global__ __launch_bounds__(1024, 2) void pat_a(const float4* const* in, float4* const* out, long long n4) {
  for (long long i = blockIdx.x * 1024LL + threadIdx.x; i < n4; i += gridDim.x * 1024LL) {
    float4 s = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int k = 0; k < NR; ++k) { const float4 v = in[k][i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    for (int k = 0; k < NW; ++k) out[k][i] = s;
    if (NW == 0 && s.x == 12345.678f) out[0][0] = s;
  }
}

template <int NR, int NW, int PTS>  // PTS = 16 (B) or 32 (C)
__global__ __launch_bounds__(1024, 2) void pat_bc(const float* const* in, float* const* out, int ntiles) {
  constexpr int G = 1024 / PTS, FPT = R / G / (32 / PTS) * (32 / PTS);  // rows per thread per unit
  constexpr int RPT = R / G;                                            // 16 (B) or 32 (C)
  const int n = threadIdx.x & (PTS - 1), g = threadIdx.x / PTS;
  const int units = ntiles * (32 / PTS);
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const long long base = (long long)(u / (32 / PTS)) * R * 32 + (u % (32 / PTS)) * PTS + n;
    float acc[RPT];
    for (int i = 0; i < RPT; ++i) acc[i] = 0.f;
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int i = 0; i < RPT; ++i) acc[i] += in[k][base + (long long)(g + G * i) * 32];
    for (int k = 0; k < NW; ++k)
#pragma unroll
      for (int i = 0; i < RPT; ++i) out[k][base + (long long)(g + G * i) * 32] = acc[i];
    if (NW == 0 && acc[0] == 12345.678f) out[0][0] = acc[1];
  }
}

template <int NR, int NW>
__global__ __launch_bounds__(1024, 2) void pat_d(const float* const* in, float* const* out, int ntiles) {
  constexpr int G = 128, RPT = R / G;  // 8 rows x float4 per thread
  const int qd = threadIdx.x & 7, g = threadIdx.x >> 3;
  for (int u = blockIdx.x; u < ntiles; u += gridDim.x) {
    const long long base = (long long)u * R * 32 + 4 * qd;
    float4 acc[RPT];
    for (int i = 0; i < RPT; ++i) acc[i] = make_float4(0, 0, 0, 0);
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(in[k] + base + (long long)(g + G * i) * 32);
        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
      }
    for (int k = 0; k < NW; ++k)
#pragma unroll
      for (int i = 0; i < RPT; ++i) *reinterpret_cast<float4*>(out[k] + base + (long long)(g + G * i) * 32) = acc[i];
    if (NW == 0 && acc[0].x == 12345.678f) out[0][0] = acc[1].y;
  }
}

int main() {
  const int ntiles = 3125;  // 100 000 points: 410 MB per record
  const size_t bytes = (size_t)ntiles * R * 32 * 4;
  std::vector<float*> bufs(6);
  for (auto& b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
  const float** din; float** dout;
  hipMalloc(&din, 3 * sizeof(float*)); hipMalloc(&dout, 3 * sizeof(float*));
  hipMemcpy(din, bufs.data(), 3 * sizeof(float*), hipMemcpyHostToDevice);
  hipMemcpy(dout, bufs.data() + 3, 3 * sizeof(float*), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto report = [&](const char* name, int nr, int nw, float ms) {
    printf("%-34s %d reads + %d writes: %.1f us  %.2f TB/s\n", name, nr, nw, 1e3 * ms, (nr + nw) * (double)bytes / ms / 1e9);
  };
#define TIME(NAME, NR_, NW_, LAUNCH)                                  \
  { LAUNCH; hipDeviceSynchronize(); hipEventRecord(e0); for (int it = 0; it < 5; ++it) { LAUNCH; } hipEventRecord(e1);  \
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); report(NAME, NR_, NW_, ms / 5); }
#define ALL(NR_, NW_)                                                                                                   \
  TIME("A flat float4", NR_, NW_, (pat_a<NR_, NW_><<<512, 1024>>>((const float4* const*)din, (float4* const*)dout, bytes / 16)))      \
  TIME("B 16-pt units, dword, 64 B", NR_, NW_, (pat_bc<NR_, NW_, 16><<<512, 1024>>>(din, dout, ntiles)))              \
  TIME("C 32-pt units, dword, 128 B", NR_, NW_, (pat_bc<NR_, NW_, 32><<<512, 1024>>>(din, dout, ntiles)))             \
  TIME("D 32-pt units, float4", NR_, NW_, (pat_d<NR_, NW_><<<512, 1024>>>(din, dout, ntiles)))
  ALL(1, 1)
  ALL(2, 1)
  ALL(3, 2)
  ALL(1, 0)
  ALL(0, 1)
  return 0;
}
