// Diagnostic (not product): ceiling of the sim_ahead access pattern without any arithmetic — per "step" each lane
// loads NR float4 (one per input stream) and stores NW float4 (one per output stream), streams are [K][B] rows like
// the lane-major trajectories. Build: hipcc -O3 --offload-arch=gfx950 stream_pattern.hip -o stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) pattern(const float* __restrict__ in, float* __restrict__ out, int64_t B, int K) {
  const int64_t blk0 = (int64_t)blockIdx.x * 1024;
  const unsigned lane = threadIdx.x * 4;
  if (blk0 + lane >= B) return;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  v4 nxt[NR > 0 ? NR : 1];
#pragma unroll
  for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(in + (int64_t)r * B + blk0 + lane);
  for (int k = 0; k < K; ++k) {
    v4 cur[NR > 0 ? NR : 1];
#pragma unroll
    for (int r = 0; r < NR; ++r) cur[r] = nxt[r];
    const int kn = (k + 1 < K) ? k + 1 : k;
#pragma unroll
    for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(in + ((int64_t)kn * NR + r) * B + blk0 + lane);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += cur[r];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      v4 v = acc + (float)w;
      float* p = out + ((int64_t)k * NW + w) * B + blk0 + lane;
      if (NT) __builtin_nontemporal_store(v, (v4*)p); else *(v4*)p = v;
    }
  }
}

template <int NR, int NW, bool NT> double run(const float* in, float* out, int64_t B, int K, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  dim3 grid((unsigned)(B / 1024)), block(256);
  hipLaunchKernelGGL((pattern<NR, NW, NT>), grid, block, 0, 0, in, out, B, K);
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pattern<NR, NW, NT>), grid, block, 0, 0, in, out, B, K);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double bytes = (double)(NR + NW) * 4.0 * B * K * reps;
  return bytes / (ms * 1e-3) / 1e9;
}

int main() {
  const int64_t B = 1 << 22; const int K = 100;
  float *in, *out;
  (void)hipMalloc(&in, (size_t)2 * 4 * B * K);
  (void)hipMalloc(&out, (size_t)15 * 4 * B * K);
  (void)hipMemset(in, 0, (size_t)2 * 4 * B * K);
  printf("pattern NR=2 NW=15 plain : %.0f GB/s\n", run<2, 15, false>(in, out, B, K, 5));
  printf("pattern NR=2 NW=15 nt    : %.0f GB/s\n", run<2, 15, true>(in, out, B, K, 5));
  printf("pattern NR=0 NW=15 nt    : %.0f GB/s\n", run<0, 15, true>(in, out, B, K, 5));
  printf("pattern NR=2 NW=1  nt    : %.0f GB/s\n", run<2, 1, true>(in, out, B, K, 5));
  printf("pattern NR=0 NW=1  nt    : %.0f GB/s\n", run<0, 1, true>(in, out, B, K, 5));
  printf("pattern NR=0 NW=4  nt    : %.0f GB/s\n", run<0, 4, true>(in, out, B, K, 5));
  printf("pattern NR=1 NW=4  nt    : %.0f GB/s\n", run<1, 4, true>(in, out, B, K, 5));
  printf("pattern NR=0 NW=15 plain : %.0f GB/s\n", run<0, 15, false>(in, out, B, K, 5));
  return 0;
}
