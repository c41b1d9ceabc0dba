// Diagnostic (not product): what does HBM / L2 make of scattered aligned runs? The env-major state leaves leave the fused kernel
// as 32-byte runs 404 bytes apart (one run per environment, leaf and 8-step window). Here: a buffer of G bytes is written exactly
// once, in pieces of P bytes (16 B per lane, P/16 adjacent lanes per piece), piece i of the launch going to position
// (i * STRIDE) mod n_pieces — STRIDE = 1 is a dense fill, STRIDE = 13 puts consecutive pieces ~13 pieces apart like consecutive
// environments' runs. Prints GB/s per (P, STRIDE).
//   hipcc -O3 --offload-arch=gfx950 scatter_runs.hip -o scatter_runs && ./scatter_runs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) scatter(float* out, int64_t n_pieces, int lanes_per_piece, int64_t stride, int nt) {
  const int64_t lane_global = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total_lanes = (int64_t)gridDim.x * 256;
  v4 v = {1.f, 2.f, 3.f, 4.f};
  for (int64_t l = lane_global; l < n_pieces * lanes_per_piece; l += total_lanes) {
    const int64_t piece = l / lanes_per_piece, sub = l % lanes_per_piece;
    const int64_t pos = (piece * stride) % n_pieces;
    v4* p = (v4*)out + pos * lanes_per_piece + sub;
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
  }
}

int main() {
  const size_t G = (size_t)8 << 30;
  float* buf;
  if (hipMalloc(&buf, G) != hipSuccess) return 1;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int nt = 0; nt < 2; ++nt)
    for (int P : {16, 32, 64, 128, 256, 1024}) {
      for (int64_t stride : {(int64_t)1, (int64_t)13, (int64_t)100003}) {
        const int lpp = P / 16;
        const int64_t n_pieces = (int64_t)(G / P);
        hipLaunchKernelGGL(scatter, dim3(256 * 8), dim3(256), 0, 0, buf, n_pieces, lpp, stride, nt);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(scatter, dim3(256 * 8), dim3(256), 0, 0, buf, n_pieces, lpp, stride, nt);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%s pieces of %4d B, stride %6lld pieces: %6.0f GB/s\n", nt ? "nt   " : "plain", P, (long long)stride, (double)G / ms / 1e6);
        fflush(stdout);
      }
    }
  return 0;
}
