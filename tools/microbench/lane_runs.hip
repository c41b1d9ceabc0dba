// Diagnostic (not product): the write pattern of an env-major trajectory kernel whose lanes each own one environment and write
// that environment's aligned 128-byte runs themselves (8 consecutive 16-byte stores per lane and leaf, one window of 32 steps),
// observations as 1 KiB per environment and window. No arithmetic. Environments of a wave are P apart (P = 1: neighbours; P = 32:
// the stride that gives all lanes of a wave the same window phase for rows of 101 elements). LDS padding limits the waves per CU.
//   hipcc -O3 --offload-arch=gfx950 lane_runs.hip -o lane_runs && ./lane_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void __launch_bounds__(64) lane_runs(float* obs, float* const* leaf, int64_t B, int R, int P, int obs_too) {
  extern __shared__ float pad[];
  const int64_t w = blockIdx.x;
  const int64_t e = (w / P) * (64 * (int64_t)P) + (w % P) + (int64_t)P * threadIdx.x;
  if (e >= B) return;
  v4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  if (threadIdx.x == 1000) pad[0] = 1.f;
  for (int win = 0; win < R / 32; ++win) {
    const int64_t row = e * R + win * 32;
    if (obs_too) {
      v4* o = (v4*)(obs + row * 8);
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        if (NT) __builtin_nontemporal_store(v, o + q);
        else o[q] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      v4* l = (v4*)(leaf[j] + row);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (NT) __builtin_nontemporal_store(v, l + q);
        else l[q] = v;
      }
    }
  }
}

int main() {
  const int64_t B = (int64_t)1 << 22;
  const int R = 96;
  float* obs;
  std::vector<float*> leaves(7);
  if (hipMalloc(&obs, (size_t)B * R * 8 * 4) != hipSuccess) return 1;
  for (auto& l : leaves)
    if (hipMalloc(&l, (size_t)B * R * 4) != hipSuccess) return 1;
  float** dl;
  (void)hipMalloc(&dl, 7 * sizeof(float*));
  (void)hipMemcpy(dl, leaves.data(), 7 * sizeof(float*), hipMemcpyHostToDevice);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int obs_too : {1, 0})
    for (int nt : {0, 1})
      for (int P : {1, 32})
        for (int lds_kb : {8, 20, 40}) {
          const size_t lds = (size_t)lds_kb << 10;
          auto go = [&] {
            if (nt) hipLaunchKernelGGL(lane_runs<1>, dim3(B / 64), dim3(64), lds, 0, obs, dl, B, R, P, obs_too);
            else hipLaunchKernelGGL(lane_runs<0>, dim3(B / 64), dim3(64), lds, 0, obs, dl, B, R, P, obs_too);
          };
          go();
          (void)hipEventRecord(a);
          go();
          (void)hipEventRecord(b);
          (void)hipEventSynchronize(b);
          float ms;
          (void)hipEventElapsedTime(&ms, a, b);
          const double bytes = (double)B * R * 4 * (7 + (obs_too ? 8 : 0));
          printf("%s %s P=%2d lds %2d KB/wave: %7.3f ms %6.0f GB/s\n", obs_too ? "obs+leaves" : "leaves only", nt ? "nt   " : "plain", P, lds_kb,
                 ms, bytes / ms / 1e6);
          fflush(stdout);
        }
  return 0;
}
