// Diagnostic (not product): access-pattern ceiling of an in-kernel time-tiled env-major writer, no arithmetic.
// One wave (64 threads) owns 64 envs; per step each lane produces OW obs words + S state words into LDS; every TK steps the
// wave flushes: obs rows [B][N][OW] get per-env runs of TK*OW floats (16 B per lane), state leaves [B][N] runs of TK floats.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int OW = 8, S = 7, A = 2;

template <int TK>
__global__ void __launch_bounds__(64) pattern(const float* __restrict__ act, float* __restrict__ obs, float* __restrict__ st, int64_t B, int N) {
  constexpr int LDO = TK * OW + 1, LDS_ = TK + 1;
  __shared__ float tobs[64 * LDO];
  __shared__ float tst[S][64 * LDS_];
  const int lane = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * 64;
  float acc = (float)lane;
  for (int n0 = 0; n0 < N; n0 += TK) {
    // action tile: per env TK*A contiguous floats -> 4 lanes x float4 per env (TK=8,A=2: 16 floats)
    {
      constexpr int PER = TK * A / 4;  // float4 per env
      for (int i = lane; i < 64 * PER; i += 64) {
        const int e = i / PER, j = i % PER;
        v4 v = *(const v4*)(act + ((e0 + e) * N + n0) * A + 4 * j);
        acc += v.x + v.y + v.z + v.w;
      }
    }
    for (int t = 0; t < TK; ++t) {
#pragma unroll
      for (int c = 0; c < OW; ++c) tobs[lane * LDO + t * OW + c] = acc + c;
#pragma unroll
      for (int s = 0; s < S; ++s) tst[s][lane * LDS_ + t] = acc - s;
      acc += 1.0f;
    }
    __syncthreads();
    // flush obs: per env TK*OW floats contiguous; 16 lanes x float4 (TK=8) per env
    {
      constexpr int PER = TK * OW / 4;
      for (int i = lane; i < 64 * PER; i += 64) {
        const int e = i / PER, j = i % PER;
        v4 v;
        v.x = tobs[e * LDO + 4 * j]; v.y = tobs[e * LDO + 4 * j + 1]; v.z = tobs[e * LDO + 4 * j + 2]; v.w = tobs[e * LDO + 4 * j + 3];
        *(v4*)(obs + ((e0 + e) * N + n0) * OW + 4 * j) = v;
      }
    }
    // flush states: per leaf per env TK floats; scalar stores (4-byte alignment only in the real layout)
#pragma unroll
    for (int s = 0; s < S; ++s)
      for (int i = lane; i < 64 * TK; i += 64) {
        const int e = i / TK, t = i % TK;
        st[(int64_t)s * B * N + (e0 + e) * N + n0 + t] = tst[s][e * LDS_ + t];
      }
    __syncthreads();
  }
}

template <int TK> void run(const float* act, float* obs, float* st, int64_t B, int N) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  dim3 grid((unsigned)(B / 64)), block(64);
  hipLaunchKernelGGL((pattern<TK>), grid, block, 0, 0, act, obs, st, B, N);
  (void)hipEventRecord(a);
  const int reps = 3;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pattern<TK>), grid, block, 0, 0, act, obs, st, B, N);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("env-major LDS time-tile TK=%-2d : %.0f GB/s  (%.2f ms per launch, %.2e env-steps/s)\n", TK,
         (double)(A + OW + S) * 4.0 * B * N * reps / (ms * 1e-3) / 1e9, ms / reps, (double)B * N * reps / (ms * 1e-3));
}

int main() {
  const int64_t B = 1 << 22; const int N = 96;
  float *act, *obs, *st;
  (void)hipMalloc(&act, (size_t)A * 4 * B * N);
  (void)hipMalloc(&obs, (size_t)OW * 4 * B * N);
  (void)hipMalloc(&st, (size_t)S * 4 * B * N);
  (void)hipMemset(act, 0, (size_t)A * 4 * B * N);
  run<4>(act, obs, st, B, N);
  run<8>(act, obs, st, B, N);
  run<16>(act, obs, st, B, N);
  run<32>(act, obs, st, B, N);
  return 0;
}
