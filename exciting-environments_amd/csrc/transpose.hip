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

template <typename T, int W> struct VecT;
template <> struct VecT<float, 1> { using type = float; };
template <> struct VecT<float, 4> { using type = float4; };
template <> struct VecT<double, 1> { using type = double; };
template <> struct VecT<double, 2> { using type = double2; };

// VL / VS: elements per global load / store (16 bytes when the row strides and bases allow it, else 1).
template <typename T, int TN, int VL, int VS>
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
  {  // load: lanes run along the input's contiguous axis, VL elements each
    constexpr int CPR = TN / VL;  // vector columns per row
    const int c = (t % CPR) * VL;
    const T* src = in + m0 * N + n0 + c;
    if (c < tn) {
      for (int r = t / CPR; r < tm; r += 256 / CPR) {
        if constexpr (VL == 1) {
          tile[r * LD + c] = src[(int64_t)r * N];
        } else {
          using V = typename VecT<T, VL>::type;
          const V v = *reinterpret_cast<const V*>(src + (int64_t)r * N);  // tn is a multiple of VL on this path
          const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
          for (int j = 0; j < VL; ++j) tile[r * LD + c + j] = e[j];
        }
      }
    }
  }
  __syncthreads();
  {  // store: lanes run along the output's contiguous axis (m), VS elements each; one wave per output row at a time
    const int lane = t & 63, wave = t >> 6;
    for (int c = wave; c < tn; c += 4) {
      T* dst = out + (n0 + c) * M + m0;
      for (int r = lane * VS; r < tm; r += 64 * VS) {
        if constexpr (VS == 1) {
          dst[r] = tile[r * LD + c];
        } else {
          using V = typename VecT<T, VS>::type;
          V v;
          T* e = reinterpret_cast<T*>(&v);
#pragma unroll
          for (int j = 0; j < VS; ++j) e[j] = tile[(r + j) * LD + c];  // tm is a multiple of VS on this path
          *reinterpret_cast<V*>(dst + r) = v;
        }
      }
    }
  }
}

// Short input rows (N * sizeof(T) <= 2 KiB: the [K*A]-long action rows of a trajectory): a tile is TM WHOLE rows, i.e. one
// contiguous chunk of the input — streamed in with 16-byte loads — and leaves as N runs of TM elements, 16 consecutive lanes per
// run. The general kernel reads such rows as 128-byte pieces 800 bytes apart (3.2 TB/s for the actions of the headline launch).
template <typename T, int TM>
__global__ void __launch_bounds__(256) transpose_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t M, int N) {
  constexpr int VW = 16 / (int)sizeof(T);
  extern __shared__ __align__(16) unsigned char smem[];
  T* tile = reinterpret_cast<T*>(smem);  // [TM][N + 1]
  const int LD = N + 1;
  const int64_t m0 = (int64_t)blockIdx.x * TM;
  const int tm = (int)((M - m0 < TM) ? (M - m0) : TM);  // a multiple of VW (host: M % VW == 0)
  const int t = threadIdx.x;
  const T* src = in + m0 * N;
  const int nvec = tm * N / VW;  // N % VW == 0 (host)
  using V = typename VecT<T, VW>::type;
  for (int i = t; i < nvec; i += 256) {
    const V v = *reinterpret_cast<const V*>(src + (int64_t)i * VW);
    const T* e = reinterpret_cast<const T*>(&v);
    const int r = (i * VW) / N, c = (i * VW) % N;  // a vector never straddles rows (N % VW == 0)
#pragma unroll
    for (int j = 0; j < VW; ++j) tile[r * LD + c + j] = e[j];
  }
  __syncthreads();
  const int rv = tm / VW;  // 16-byte pieces per output run
  for (int i = t; i < N * rv; i += 256) {
    const int c = i / rv, r = (i % rv) * VW;
    V v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < VW; ++j) e[j] = tile[(r + j) * LD + c];
    *reinterpret_cast<V*>(out + (int64_t)c * M + m0 + r) = v;
  }
}

template <typename T> static int launch_t(int64_t M, int64_t N, const T* in, T* out, hipStream_t stream) {
  // short output rows (M small): 64 input columns x the whole row; otherwise 32 columns x up to ~500 rows
  {
    constexpr int VW = 16 / (int)sizeof(T);
    constexpr int TMR = 64;
    const bool al = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    if (al && N * (int64_t)sizeof(T) <= 2048 && (N % VW) == 0 && (M % VW) == 0 && M >= 4096 &&
        (N + 1) * TMR * (int64_t)sizeof(T) <= LDS_BUDGET && (M + TMR - 1) / TMR < ((int64_t)1 << 31)) {
      const dim3 grid((unsigned)((M + TMR - 1) / TMR)), block(256);
      hipLaunchKernelGGL((transpose_rows_kernel<T, TMR>), grid, block, (size_t)TMR * (N + 1) * sizeof(T), stream, in, out, M, (int)N);
      return hipGetLastError() == hipSuccess ? EXCENV_OK : EXCENV_EHIP;
    }
  }
  const bool wide = (int64_t)M * 65 * (int64_t)sizeof(T) <= LDS_BUDGET;
  const int TN = wide ? 64 : 32;
  constexpr int VMAX = 16 / (int)sizeof(T);
  const int max_tm = LDS_BUDGET / ((TN + 1) * (int)sizeof(T));
  const int64_t parts = (M + max_tm - 1) / max_tm;
  int TM = (int)((M + parts - 1) / parts);
  const bool al_in = (reinterpret_cast<uintptr_t>(in) & 15u) == 0, al_out = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  const bool vl = al_in && (N % VMAX) == 0;                   // every input row start and tile edge stays 16-byte aligned
  bool vs = al_out && (M % VMAX) == 0;
  if (vs) {                                                    // tile height must keep the alignment along m
    TM = (TM + VMAX - 1) / VMAX * VMAX;
    if (TM > max_tm) TM -= VMAX;
    if (TM < VMAX) vs = false;
  }
  const int64_t gx = (N + TN - 1) / TN, gy = (M + TM - 1) / TM;
  if (gx * gy >= ((int64_t)1 << 31)) return EXCENV_EINVAL;
  const dim3 grid((unsigned)(gx * gy)), block(256);
  const size_t lds = (size_t)TM * (TN + 1) * sizeof(T);
#define EXCENV_TR(TNN, VLL, VSS) hipLaunchKernelGGL((transpose_kernel<T, TNN, VLL, VSS>), grid, block, lds, stream, in, out, M, N, TM, (unsigned)gx)
  if (wide) {
    if (vl && vs) EXCENV_TR(64, VMAX, VMAX); else if (vl) EXCENV_TR(64, VMAX, 1); else if (vs) EXCENV_TR(64, 1, VMAX); else EXCENV_TR(64, 1, 1);
  } else {
    if (vl && vs) EXCENV_TR(32, VMAX, VMAX); else if (vl) EXCENV_TR(32, VMAX, 1); else if (vs) EXCENV_TR(32, 1, VMAX); else EXCENV_TR(32, 1, 1);
  }
#undef EXCENV_TR
  return hipGetLastError() == hipSuccess ? EXCENV_OK : EXCENV_EHIP;
}

int launch_transpose(int dtype, int64_t M, int64_t N, const void* in, void* out, hipStream_t stream) {
  if (M <= 0 || N <= 0) return EXCENV_OK;
  return dtype == EXCENV_F32 ? launch_t<float>(M, N, (const float*)in, (float*)out, stream)
                             : launch_t<double>(M, N, (const double*)in, (double*)out, stream);
}

}  // namespace excenv
