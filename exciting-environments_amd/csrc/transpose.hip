// LDS-tiled matrix transpose out[n][m] = in[m][n], used to convert between the reference's env-major
// (row-major [B][K][C]) trajectories and the kernels' lane-major ([K][C][B]) layout when a caller insists on
// env-major buffers: time is the contiguous axis there while parallelism runs across environments, so a
// transposition is inherent — done here once, coalesced on both sides, instead of as scattered 4-byte accesses.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/excenv.h"

namespace excenv {

constexpr int TT = 64;  // tile edge; 256 threads move a 64x64 tile, 16 elements each

template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t M, int64_t N,
                                                        unsigned gx) {
  __shared__ T tile[TT][TT + 1];  // +1: column reads hit distinct banks
  const int64_t n0 = (int64_t)(blockIdx.x % gx) * TT;  // along the contiguous axis of `in`
  const int64_t m0 = (int64_t)(blockIdx.x / gx) * TT;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int r = ty; r < TT; r += 4) {
    const int64_t m = m0 + r, n = n0 + tx;
    if (m < M && n < N) tile[r][tx] = in[m * N + n];
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < TT; r += 4) {
    const int64_t n = n0 + r, m = m0 + tx;
    if (n < N && m < M) out[n * M + m] = tile[tx][r];
  }
}

int launch_transpose(int dtype, int64_t M, int64_t N, const void* in, void* out, hipStream_t stream) {
  if (M <= 0 || N <= 0) return EXCENV_OK;
  const int64_t gx = (N + TT - 1) / TT, gy = (M + TT - 1) / TT;
  if (gx * gy >= ((int64_t)1 << 31)) return EXCENV_EINVAL;
  const dim3 grid((unsigned)(gx * gy)), block(256);
  if (dtype == EXCENV_F32)
    hipLaunchKernelGGL((transpose_kernel<float>), grid, block, 0, stream, (const float*)in, (float*)out, M, N, (unsigned)gx);
  else
    hipLaunchKernelGGL((transpose_kernel<double>), grid, block, 0, stream, (const double*)in, (double*)out, M, N, (unsigned)gx);
  return hipGetLastError() == hipSuccess ? EXCENV_OK : EXCENV_EHIP;
}

}  // namespace excenv
