// Diagnostic (not product): what does the SHAPE of the row-major action read cost, with no arithmetic around it?
// A pendulum-shaped trajectory launch (per env-step: one fp32 action read, five fp32 values written lane-major, four
// environments per lane, 16-byte stores) whose actions come
//   mode 0: from a lane-major array [K][B]                       (16-byte loads, 1 KiB runs per wave and step)
//   mode W: from a row-major array [B][K] in windows of W bytes per environment, each lane fetching the window of each of its own
//           environments into registers (W = 64: one 64-byte sector per request — what the product's LDS windows ask the memory
//           system for; W = 128: both sectors of a 128-byte line; W = 32: half a sector, the acrobot's window)
// DESIGN.md §4.1b claims the 15 ... 45 % the small models lose with row-major actions is this request shape, not residency.
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 -o /tmp/rmw tools/microbench/rowmajor_windows.hip && /tmp/rmw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int V = 4, NS = 5;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                      \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

// W = 0: lane-major actions; else window bytes per environment
template <int W>
__global__ void __launch_bounds__(256) traj(const float* __restrict__ act, float* __restrict__ out, int64_t B, int K) {
  const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (e0 >= B) return;
  float s[V] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (W == 0) {
    for (int k = 0; k < K; ++k) {
      const v4 a = *reinterpret_cast<const v4*>(act + (int64_t)k * B + e0);
#pragma unroll
      for (int v = 0; v < V; ++v) s[v] = s[v] * 0.5f + a[v];
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        v4 o;
#pragma unroll
        for (int v = 0; v < V; ++v) o[v] = s[v] + (float)j;
        __builtin_nontemporal_store(o, reinterpret_cast<v4*>(out + ((int64_t)k * NS + j) * B + e0));
      }
    }
  } else {
    constexpr int NP = W / 16, NR = W / 4;  // pieces and rows per window
    for (int k0 = 0; k0 < K; k0 += NR) {
      v4 win[V][NP];
#pragma unroll
      for (int v = 0; v < V; ++v)
#pragma unroll
        for (int i = 0; i < NP; ++i) win[v][i] = *reinterpret_cast<const v4*>(act + (e0 + v) * (int64_t)K + k0 + 4 * i);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = s[v] * 0.5f + win[v][r / 4][r % 4];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          v4 o;
#pragma unroll
          for (int v = 0; v < V; ++v) o[v] = s[v] + (float)j;
          __builtin_nontemporal_store(o, reinterpret_cast<v4*>(out + ((int64_t)(k0 + r) * NS + j) * B + e0));
        }
      }
    }
  }
}

template <int W> static float run(const float* act, float* out, int64_t B, int K, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const dim3 grid((unsigned)((B / V + 255) / 256)), block(256);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(traj<W>, grid, block, 0, 0, act, out, B, K);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(traj<W>, grid, block, 0, 0, act, out, B, K);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t B = (int64_t)1 << (argc > 1 ? std::atoi(argv[1]) : 22);
  const int K = argc > 2 ? std::atoi(argv[2]) : 128;  // rows of K * 4 bytes: a multiple of 256 keeps every window aligned
  if (K <= 0 || K % 32 != 0) {  // the window loops assume whole windows per row (and never read or write past a row)
    std::printf("K must be a positive multiple of 32 (rows of whole 128-byte windows)\n");
    return 2;
  }
  float *act, *out;
  CK(hipMalloc(&act, sizeof(float) * B * K));
  CK(hipMalloc(&out, sizeof(float) * B * K * NS));
  CK(hipMemset(act, 0, sizeof(float) * B * K));
  const double bytes = (double)B * K * 4.0 * (1 + NS);
  std::printf("B = 2^%d, K = %d: %.2f GB per launch (%d B per env-step), action share %.0f %%\n", argc > 1 ? std::atoi(argv[1]) : 22, K,
              bytes / 1e9, 4 * (1 + NS), 100.0 / (1 + NS));
  for (int round = 0; round < 2; ++round) {
    const float t0 = run<0>(act, out, B, K, 10), t32 = run<32>(act, out, B, K, 10), t64 = run<64>(act, out, B, K, 10), t128 = run<128>(act, out, B, K, 10);
    std::printf("round %d  lane-major %.3f ms (%.0f GB/s) | 32-byte windows %.3f (%.2f x) | 64-byte %.3f (%.2f x) | 128-byte %.3f (%.2f x)\n", round, t0,
                bytes / t0 / 1e6, t32, t32 / t0, t64, t64 / t0, t128, t128 / t0);
  }
  CK(hipFree(act));
  CK(hipFree(out));
  return 0;
}
