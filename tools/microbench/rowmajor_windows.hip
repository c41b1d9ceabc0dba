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

// The same launch with the product's mechanism for the 64-byte windows: LDS-direct loads (four adjacent lanes fetch the four pieces
// of one environment's sector, 16 environments per instruction, the hardware puts lane t's 16 bytes at M0 + 16 t), then every lane
// reads its own window from LDS, one 16-byte piece per four rows. Synchronous like the register form above (fill, wait, use).
__global__ void __launch_bounds__(256) traj_dma(const float* __restrict__ act, float* __restrict__ out, int64_t B, int K) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (e0 >= B) return;  // B is a multiple of 64 * V here: whole waves
  const unsigned wave = threadIdx.x / 64u, lane = threadIdx.x % 64u;
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (V * 4 * 1024));
  const int64_t wave_e0 = ((int64_t)blockIdx.x * 256 + wave * 64) * V;
  float s[V] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* src = act + (wave_e0 + (int64_t)(i * 16 + lane / 4) * V + v) * (int64_t)K + k0 + 4 * (lane % 4);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(wave_lds + (unsigned)(v * 4 + i) * 1024u) : "memory", "m0");
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      v4 piece[V];
#pragma unroll
      for (int v = 0; v < V; ++v)
        piece[v] = *reinterpret_cast<const v4*>(smem + wave * (V * 4 * 1024) + (unsigned)(v * 4 + lane / 16) * 1024u + (lane % 16) * 64u + p * 16u);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = s[v] * 0.5f + piece[v][r];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          v4 o;
#pragma unroll
          for (int v = 0; v < V; ++v) o[v] = s[v] + (float)j;
          __builtin_nontemporal_store(o, reinterpret_cast<v4*>(out + ((int64_t)(k0 + 4 * p + r) * NS + j) * B + e0));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the windows' LDS is re-filled at the top of the loop
  }
}
static float run_dma(const float* act, float* out, int64_t B, int K, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const dim3 grid((unsigned)((B / V + 255) / 256)), block(256);
  const size_t lds = 4 * V * 4 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(traj_dma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(traj_dma, grid, block, lds, 0, act, out, B, K);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(traj_dma, grid, block, lds, 0, act, out, B, K);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
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
  if (B < 1024 || K <= 0 || K % 32 != 0) {  // the window loops assume whole windows per row (and never read or write past a row)
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
    const float td = (B % (64 * V) == 0) ? run_dma(act, out, B, K, 10) : 0.f;
    std::printf("round %d  lane-major %.3f ms (%.0f GB/s) | 32-byte windows %.3f (%.2f x) | 64-byte %.3f (%.2f x) | 128-byte %.3f (%.2f x) | "
                "64-byte windows by LDS-direct loads %.3f (%.2f x)\n", round, t0, bytes / t0 / 1e6, t32, t32 / t0, t64, t64 / t0, t128, t128 / t0, td, td / t0);
  }
  CK(hipFree(act));
  CK(hipFree(out));
  return 0;
}
