// Diagnostic (not product): does wave lock-step / tile-sequential writing raise the ceiling of the persistent-lane pattern?
// NR=2 loads + NW=15 stores of float4 per lane per step; variants: block size, __syncthreads per step, tiled layout.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int NR = 2, NW = 15;

template <int BS, bool SYNC, bool TILED>
__global__ void __launch_bounds__(BS) pattern(const float* __restrict__ in, float* __restrict__ out, int64_t B, int K) {
  const int64_t tile = (int64_t)BS * 4;                       // envs per workgroup
  const int64_t blk0 = (int64_t)blockIdx.x * tile;
  const unsigned lane = threadIdx.x * 4;
  // lane-major: row r of stream s at (k*N + s)*B + env ; tiled: wg base + (k*N + s)*tile + lane
  const float* ib = TILED ? in + (int64_t)blockIdx.x * K * NR * tile : in + blk0;
  float* ob = TILED ? out + (int64_t)blockIdx.x * K * NW * tile : out + blk0;
  const int64_t rs = TILED ? tile : B;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  v4 nxt[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(ib + (int64_t)r * rs + lane);
  for (int k = 0; k < K; ++k) {
    v4 cur[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) cur[r] = nxt[r];
    const int kn = (k + 1 < K) ? k + 1 : k;
#pragma unroll
    for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(ib + ((int64_t)kn * NR + r) * rs + lane);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += cur[r];
    if (SYNC) __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      v4 v = acc + (float)w;
      __builtin_nontemporal_store(v, (v4*)(ob + ((int64_t)k * NW + w) * rs + lane));
    }
  }
}

template <int BS, bool SYNC, bool TILED> void run(const char* name, const float* in, float* out, int64_t B, int K) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  dim3 grid((unsigned)(B / (BS * 4))), block(BS);
  hipLaunchKernelGGL((pattern<BS, SYNC, TILED>), grid, block, 0, 0, in, out, B, K);
  (void)hipEventRecord(a);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pattern<BS, SYNC, TILED>), grid, block, 0, 0, in, out, B, K);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-34s : %.0f GB/s\n", name, (double)(NR + NW) * 4.0 * B * K * reps / (ms * 1e-3) / 1e9);
}

int main() {
  const int64_t B = 1 << 22; const int K = 100;
  float *in, *out;
  (void)hipMalloc(&in, (size_t)NR * 4 * B * K);
  (void)hipMalloc(&out, (size_t)NW * 4 * B * K);
  (void)hipMemset(in, 0, (size_t)NR * 4 * B * K);
  run<256, false, false>("lane-major  bs256", in, out, B, K);
  run<256, true, false>("lane-major  bs256 sync", in, out, B, K);
  run<1024, false, false>("lane-major  bs1024", in, out, B, K);
  run<1024, true, false>("lane-major  bs1024 sync", in, out, B, K);
  run<256, false, true>("tiled       bs256", in, out, B, K);
  run<256, true, true>("tiled       bs256 sync", in, out, B, K);
  run<1024, false, true>("tiled       bs1024", in, out, B, K);
  run<1024, true, true>("tiled       bs1024 sync", in, out, B, K);
  run<64, false, true>("tiled       bs64", in, out, B, K);
  run<64, false, false>("lane-major  bs64", in, out, B, K);
  return 0;
}
