// Diagnostic (not product): does a non-power-of-two row stride (padding) raise the ceiling of the persistent-lane pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int NR = 2, NW = 15;

__global__ void __launch_bounds__(256) pattern(const float* __restrict__ in, float* __restrict__ out, int64_t B, int64_t RS, int K) {
  const int64_t blk0 = (int64_t)blockIdx.x * 1024;
  const unsigned lane = threadIdx.x * 4;
  if (blk0 + lane >= B) return;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  v4 nxt[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(in + (int64_t)r * RS + blk0 + lane);
  for (int k = 0; k < K; ++k) {
    v4 cur[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) cur[r] = nxt[r];
    const int kn = (k + 1 < K) ? k + 1 : k;
#pragma unroll
    for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(in + ((int64_t)kn * NR + r) * RS + blk0 + lane);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += cur[r];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      v4 v = acc + (float)w;
      __builtin_nontemporal_store(v, (v4*)(out + ((int64_t)k * NW + w) * RS + blk0 + lane));
    }
  }
}

int main() {
  const int64_t B = 1 << 22; const int K = 100;
  const int64_t maxpad = 1 << 16;
  float *in, *out;
  (void)hipMalloc(&in, (size_t)NR * 4 * (B + maxpad) * K);
  (void)hipMalloc(&out, (size_t)NW * 4 * (B + maxpad) * K);
  (void)hipMemset(in, 0, (size_t)NR * 4 * (B + maxpad) * K);
  const int64_t pads[] = {0, 64, 256, 1024, 4096, 4096 + 256, 16384 + 1024, 65536 - 1024, 0};
  for (int64_t pad : pads) {
    const int64_t RS = B + pad;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    dim3 grid((unsigned)(B / 1024)), block(256);
    hipLaunchKernelGGL(pattern, grid, block, 0, 0, in, out, B, RS, K);
    (void)hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pattern, grid, block, 0, 0, in, out, B, RS, K);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("row stride B + %-6lld floats : %.0f GB/s\n", (long long)pad, (double)(NR + NW) * 4.0 * B * K * reps / (ms * 1e-3) / 1e9);
  }
  return 0;
}
