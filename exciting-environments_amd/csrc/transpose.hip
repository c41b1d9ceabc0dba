// LDS-tiled matrix transpose out[n][m] = in[m][n], used to convert between the reference's env-major
// (row-major [B][K][C]) trajectories and the kernels' lane-major ([K][C][B]) layout when a caller insists on
// env-major buffers: time is the contiguous axis there while parallelism runs across environments, so a
// transposition is inherent — done here once, coalesced on both sides, instead of as scattered 4-byte accesses.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/excenv.h"

namespace excenv {

// Tall tiles: TN columns of the input (its contiguous axis) x TM rows, TM as large as 64 KiB of LDS allows, so each
// output row receives a long contiguous run (the whole row when M is small, e.g. the [K+1]-long state leaves).
constexpr int LDS_BUDGET = 64 * 1024;

template <typename T, int TN>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t M, int64_t N,
                                                        int TM, unsigned gx) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* tile = reinterpret_cast<T*>(smem);  // [TM][TN + 1]
  constexpr int LD = TN + 1;             // odd leading dimension: column reads hit distinct banks
  const int64_t n0 = (int64_t)(blockIdx.x % gx) * TN;
  const int64_t m0 = (int64_t)(blockIdx.x / gx) * TM;
  const int tm = (int)((M - m0 < TM) ? (M - m0) : TM);
  const int tn = (int)((N - n0 < TN) ? (N - n0) : TN);
  const int t = threadIdx.x;
  {  // load: lanes run along the input's contiguous axis
    const int c = t % TN;
    const T* src = in + m0 * N + n0 + c;
    if (c < tn)
      for (int r = t / TN; r < tm; r += 256 / TN) tile[r * LD + c] = src[(int64_t)r * N];
  }
  __syncthreads();
  {  // store: lanes run along the output's contiguous axis (m); one wave per output row at a time
    const int lane = t & 63, wave = t >> 6;
    for (int c = wave; c < tn; c += 4) {
      T* dst = out + (n0 + c) * M + m0;
      for (int r = lane; r < tm; r += 64) dst[r] = tile[r * LD + c];
    }
  }
}

template <typename T> static int launch_t(int64_t M, int64_t N, const T* in, T* out, hipStream_t stream) {
  // short output rows (M small): 64 input columns x the whole row; otherwise 32 columns x up to ~500 rows
  const bool wide = (int64_t)M * 65 * (int64_t)sizeof(T) <= LDS_BUDGET;
  const int TN = wide ? 64 : 32;
  const int max_tm = LDS_BUDGET / ((TN + 1) * (int)sizeof(T));
  const int64_t parts = (M + max_tm - 1) / max_tm;
  const int TM = (int)((M + parts - 1) / parts);
  const int64_t gx = (N + TN - 1) / TN, gy = (M + TM - 1) / TM;
  if (gx * gy >= ((int64_t)1 << 31)) return EXCENV_EINVAL;
  const dim3 grid((unsigned)(gx * gy)), block(256);
  const size_t lds = (size_t)TM * (TN + 1) * sizeof(T);
  if (wide) hipLaunchKernelGGL((transpose_kernel<T, 64>), grid, block, lds, stream, in, out, M, N, TM, (unsigned)gx);
  else hipLaunchKernelGGL((transpose_kernel<T, 32>), grid, block, lds, stream, in, out, M, N, TM, (unsigned)gx);
  return hipGetLastError() == hipSuccess ? EXCENV_OK : EXCENV_EHIP;
}

int launch_transpose(int dtype, int64_t M, int64_t N, const void* in, void* out, hipStream_t stream) {
  if (M <= 0 || N <= 0) return EXCENV_OK;
  return dtype == EXCENV_F32 ? launch_t<float>(M, N, (const float*)in, (float*)out, stream)
                             : launch_t<double>(M, N, (const double*)in, (double*)out, stream);
}

}  // namespace excenv
