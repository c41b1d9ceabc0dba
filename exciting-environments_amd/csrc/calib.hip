// Same-run calibration of the trajectory kernels' memory access shape, without arithmetic (excenv_stream_pattern).
//
// sim_ahead_kernel keeps one lane per environment for the whole trajectory: per saved row every workgroup reads A and
// writes O + S pieces of 4 KiB, one piece per stream, and every stream advances by its own row stride. How fast HBM takes
// that shape depends on where the driver placed the buffers in physical memory (write traffic that falls into ONE
// physical region of the device memory runs ~15-19 % below traffic spread over two or more — the same holds for a plain
// sequential fill, so it is a property of the platform, not of the kernel). This kernel moves the same bytes through the same
// addresses in the same order as a trajectory launch would and nothing else, so that
//   * bench.py can report, in the same run and over the very buffers it timed, the no-arithmetic ceiling of the placement
//     (roofline.same_run_pattern_gbs), and
//   * the Python mirror can decide at allocation time whether a set of trajectory buffers is a slow placement
//     (core_env.py: trajectory sets) before any trajectory is written into it.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/excenv.h"

namespace excenv {
void set_error(const char* fmt, ...);

constexpr int PAT_MAX_READ = 4, PAT_MAX_WRITE = 32;
typedef float pat_v4 __attribute__((ext_vector_type(4)));

struct PatternArgs {
  int32_t n_read, n_write;
  int64_t row_bytes, rows;
  const char* rd[PAT_MAX_READ];
  int64_t rd_rs[PAT_MAX_READ];
  char* wr[PAT_MAX_WRITE];
  int64_t wr_rs[PAT_MAX_WRITE];
};

// NWC: compile-time bound of the write streams (the loop over them is unrolled like the trajectory kernels' save-row code, so
// that as many stores are in flight per wave; every lane keeps one running 64-bit address per stream in registers)
template <bool NT, int NWC> __global__ void __launch_bounds__(256) stream_pattern_kernel(const PatternArgs a) {
  const int64_t off = (int64_t)blockIdx.x * 4096 + threadIdx.x * 16;
  if (off >= a.row_bytes) return;
  pat_v4 acc = {0.f, 0.f, 0.f, 0.f};
  pat_v4 nxt[PAT_MAX_READ];
#pragma unroll
  for (int r = 0; r < PAT_MAX_READ; ++r) {
    nxt[r] = acc;
    if (r < a.n_read) nxt[r] = *reinterpret_cast<const pat_v4*>(a.rd[r] + off);
  }
  char* wp[NWC];
#pragma unroll
  for (int q = 0; q < NWC; ++q) wp[q] = (q < a.n_write) ? a.wr[q] + off : nullptr;
  for (int64_t n = 0; n < a.rows; ++n) {
    // like the trajectory kernels: the read of row n + 1 is requested before row n is written (clamped, unconditional)
    const int64_t n1 = (n + 1 < a.rows) ? n + 1 : n;
#pragma unroll
    for (int r = 0; r < PAT_MAX_READ; ++r) {
      acc += nxt[r];
      if (r < a.n_read) nxt[r] = *reinterpret_cast<const pat_v4*>(a.rd[r] + n1 * a.rd_rs[r] + off);
    }
#pragma unroll
    for (int q = 0; q < NWC; ++q) {
      if (q < a.n_write) {
        pat_v4* p = reinterpret_cast<pat_v4*>(wp[q]);
        const pat_v4 v = acc + (float)q;
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
        wp[q] += a.wr_rs[q];
      }
    }
  }
}

template <bool NT> static void launch_pattern(const PatternArgs& a, dim3 grid, hipStream_t st) {
  const dim3 block(256);
  if (a.n_write <= 4) hipLaunchKernelGGL((stream_pattern_kernel<NT, 4>), grid, block, 0, st, a);
  else if (a.n_write <= 8) hipLaunchKernelGGL((stream_pattern_kernel<NT, 8>), grid, block, 0, st, a);
  else if (a.n_write <= 16) hipLaunchKernelGGL((stream_pattern_kernel<NT, 16>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((stream_pattern_kernel<NT, PAT_MAX_WRITE>), grid, block, 0, st, a);
}

}  // namespace excenv

extern "C" int excenv_stream_pattern(int32_t n_read, const void* const* read_base, const int64_t* read_row_stride_bytes,
                                     int32_t n_write, void* const* write_base, const int64_t* write_row_stride_bytes,
                                     int64_t row_bytes, int64_t rows, int32_t nontemporal, void* stream) {
  using namespace excenv;
  if (n_read < 0 || n_read > PAT_MAX_READ || n_write < 0 || n_write > PAT_MAX_WRITE || row_bytes < 0 || rows < 0 ||
      row_bytes % 16 != 0) {
    set_error("excenv_stream_pattern: bad argument (n_read %d <= %d, n_write %d <= %d, row_bytes %lld a multiple of 16, rows %lld)",
              n_read, PAT_MAX_READ, n_write, PAT_MAX_WRITE, (long long)row_bytes, (long long)rows);
    return EXCENV_EINVAL;
  }
  if ((n_read > 0 && (!read_base || !read_row_stride_bytes)) || (n_write > 0 && (!write_base || !write_row_stride_bytes))) {
    set_error("excenv_stream_pattern: NULL argument");
    return EXCENV_ENULL;
  }
  PatternArgs a{};
  a.n_read = n_read;
  a.n_write = n_write;
  a.row_bytes = row_bytes;
  a.rows = rows;
  for (int r = 0; r < n_read; ++r) {
    if (!read_base[r] || ((uintptr_t)read_base[r] & 15) || (read_row_stride_bytes[r] & 15)) {
      set_error("excenv_stream_pattern: read stream %d must be non-NULL and 16-byte aligned (base and row stride)", r);
      return EXCENV_EINVAL;
    }
    a.rd[r] = (const char*)read_base[r];
    a.rd_rs[r] = read_row_stride_bytes[r];
  }
  for (int q = 0; q < n_write; ++q) {
    if (!write_base[q] || ((uintptr_t)write_base[q] & 15) || (write_row_stride_bytes[q] & 15)) {
      set_error("excenv_stream_pattern: write stream %d must be non-NULL and 16-byte aligned (base and row stride)", q);
      return EXCENV_EINVAL;
    }
    a.wr[q] = (char*)write_base[q];
    a.wr_rs[q] = write_row_stride_bytes[q];
  }
  if (row_bytes == 0 || rows == 0) return EXCENV_OK;
  const dim3 grid((unsigned)((row_bytes + 4095) / 4096));
  if (nontemporal) launch_pattern<true>(a, grid, (hipStream_t)stream);
  else launch_pattern<false>(a, grid, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("excenv_stream_pattern: HIP launch failed: %s", hipGetErrorString(e));
    return EXCENV_EHIP;
  }
  return EXCENV_OK;
}
