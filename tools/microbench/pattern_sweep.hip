// Diagnostic (not product): the CEILING of the trajectory kernel's access pattern. The headline launch (PMSM Euler fp32, 2^22
// environments x 100 steps) runs at 98 % of a no-arithmetic copy of its own reads and writes (2 action streams read, 8 observation
// components + 7 state leaves written, 16 B per lane and row) — about 5.8 TB/s where a plain fill of the same bytes does 6.85. This
// program holds the bytes fixed (in a placement of the fast kind: observations and state block 24 GiB apart in one arena) and
// varies the SHAPE of the traffic:
//   wg      threads per workgroup (256 / 512 / 1024: 4 / 8 / 16 KiB contiguous per stream and row)
//   vpl     16-byte vectors per lane and row (1 / 2; a wave's vectors are adjacent 1 KiB runs)
//   rows    rows stored per stream before moving to the next stream (1 / 2 / 4: what staging rows in LDS would give)
//   sync    a workgroup barrier per row group (waves of a workgroup store a stream's run together)
//   occ     workgroups per CU, limited through dynamic LDS (0 = what registers allow)
//   order   stream order inside a row: 0 observations then states, 1 alternating between the two buffers
//   nw      number of equal write streams over the same bytes (1 ... 30), half of them in either buffer
//   tiled   every workgroup's streams of a row adjacent in memory ([K][B/1024][C][1024] instead of [K][C][B])
//   ro / wo reads alone / writes alone
// One line per variant: GB/s (median of 5 launches after 2 warm ones), min / max, the fill rate of the same ranges next to it.
//   hipcc -O3 --offload-arch=gfx950 pattern_sweep.hip -o pattern_sweep && ./pattern_sweep [--only NAME] [--reps N]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int MAXR = 4, MAXW = 32;

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

struct Streams {
  const float* rd[MAXR];
  int64_t rd_rs[MAXR];  // row stride in floats
  float* wr[MAXW];
  int64_t wr_rs[MAXW];
  int nr, nw;
};

// Every workgroup owns TH * 4 * VPL consecutive environments for all K rows (the product kernel's persistent lanes).
template <int TH, int VPL, int ROWS, bool SYNC>
__global__ void __launch_bounds__(TH) pat(const Streams s, int64_t B, int K) {
  extern __shared__ float occ_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t base = (int64_t)blockIdx.x * (TH * 4 * VPL) + (int64_t)wave * (256 * VPL) + lane * 4;
  if (base >= B) return;
  v4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = v4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += ROWS) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (k + r >= K) break;
      for (int q = 0; q < s.nr; ++q)
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] += *(const v4*)(s.rd[q] + (int64_t)(k + r) * s.rd_rs[q] + base + v * 256);
    }
    if (SYNC) __syncthreads();
    for (int q = 0; q < s.nw; ++q) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        if (k + r >= K) break;
#pragma unroll
        for (int v = 0; v < VPL; ++v) *(v4*)(s.wr[q] + (int64_t)(k + r) * s.wr_rs[q] + base + v * 256) = acc[v] + (float)(q + r);
      }
    }
  }
  if (occ_lds[0] == 123.456f && B < 0) s.wr[0][0] = 1.f;  // keep the dynamic LDS referenced
}

// tiled: streams of one workgroup and row adjacent: buffer q0 holds [K][nwg][n0][1024], buffer q1 [K][nwg][n1][1024]
__global__ void __launch_bounds__(256) pat_tiled(const float* a0, const float* a1, float* b0, int n0, float* b1, int n1, int64_t B, int K) {
  const int64_t nwg = gridDim.x, w = blockIdx.x;
  const int64_t off = w * 1024 + threadIdx.x * 4;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    acc += *(const v4*)(a0 + (int64_t)k * B + off);
    acc += *(const v4*)(a1 + (int64_t)k * B + off);
    float* t0 = b0 + ((int64_t)k * nwg + w) * n0 * 1024 + threadIdx.x * 4;
    float* t1 = b1 + ((int64_t)k * nwg + w) * n1 * 1024 + threadIdx.x * 4;
    for (int q = 0; q < n0; ++q) *(v4*)(t0 + q * 1024) = acc + (float)q;
    for (int q = 0; q < n1; ++q) *(v4*)(t1 + q * 1024) = acc + (float)q;
  }
}

__global__ void __launch_bounds__(256) fill(float* p, int64_t n_v4) {
  v4 v = {1.f, 2.f, 3.f, 4.f};
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n_v4; i += (int64_t)gridDim.x * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n_v4) ((v4*)p)[i + u * 256] = v;
  }
}

__global__ void __launch_bounds__(256) fill2(float* p0, int64_t n0, float* p1, int64_t n1) {
  v4 v = {1.f, 2.f, 3.f, 4.f};
  const bool hi = blockIdx.x & 1;
  float* p = hi ? p1 : p0;
  const int64_t n = hi ? n1 : n0, half = gridDim.x / 2;
  for (int64_t i = (int64_t)(blockIdx.x >> 1) * 1024 + threadIdx.x; i < n; i += half * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n) ((v4*)p)[i + u * 256] = v;
  }
}

struct Variant {
  std::string name;
  int wg = 256, vpl = 1, rows = 1, sync = 0, occ = 0, tiled = 0;
  Streams s;
  double bytes;
};

template <int TH, int VPL>
static void launch_rows(const Variant& v, int64_t B, int K, size_t lds) {
  dim3 g((unsigned)(B / (TH * 4 * VPL))), b(TH);
#define L(R, S) hipLaunchKernelGGL((pat<TH, VPL, R, S>), g, b, lds, 0, v.s, B, K)
  if (v.rows == 1 && !v.sync) L(1, false);
  else if (v.rows == 1) L(1, true);
  else if (v.rows == 2 && !v.sync) L(2, false);
  else if (v.rows == 2) L(2, true);
  else if (v.rows == 4 && !v.sync) L(4, false);
  else L(4, true);
#undef L
}

static float *g_t0, *g_t1;
static const float *g_a0, *g_a1;
static void launch(const Variant& v, int64_t B, int K) {
  if (v.tiled) {
    hipLaunchKernelGGL(pat_tiled, dim3((unsigned)(B / 1024)), dim3(256), 0, 0, g_a0, g_a1, g_t0, 8, g_t1, 7, B, K);
    return;
  }
  // occupancy limit: LDS per workgroup so that only `occ` of them fit the CU's 160 KB
  size_t lds = v.occ ? (size_t)(160 * 1024 / v.occ) - 1024 : 0;
  if (lds > 65536) lds = 65536 + ((lds - 65536) & ~(size_t)1023);
  if (v.wg == 256 && v.vpl == 1) launch_rows<256, 1>(v, B, K, lds);
  else if (v.wg == 256) launch_rows<256, 2>(v, B, K, lds);
  else if (v.wg == 512 && v.vpl == 1) launch_rows<512, 1>(v, B, K, lds);
  else if (v.wg == 512) launch_rows<512, 2>(v, B, K, lds);
  else if (v.vpl == 1) launch_rows<1024, 1>(v, B, K, lds);
  else launch_rows<1024, 2>(v, B, K, lds);
}

int main(int argc, char** argv) {
  const char* only = nullptr;
  int reps = 5;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
    if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
  }
  const int64_t B = 1 << 22;
  const int K = 100, ROWS = K + 1;
  const int64_t rb = B * 4;                        // bytes of one row of one stream
  const int64_t obs_bytes = (int64_t)ROWS * 8 * rb;  // [K+1][8][B]
  const int64_t st_bytes = (int64_t)ROWS * 7 * rb;   // 7 x [K+1][B]
  const int64_t act_bytes = (int64_t)K * 2 * rb;
  const int64_t GAP = (int64_t)24 << 30;  // states start this far above the observations (the fast placement level)
  char* arena;
  const int64_t HI = (int64_t)13 << 30;  // room above the gap (the stream-count sweep needs 12.6 GB there)
  CK(hipMalloc(&arena, GAP + HI));
  char* actp;
  CK(hipMalloc(&actp, act_bytes));
  CK(hipMemset(actp, 0, act_bytes));
  float* obs = (float*)arena;
  float* st = (float*)(arena + GAP);
  float* act = (float*)actp;
  // opt in to large dynamic LDS for the occupancy variants
  for (int pass = 0; pass < 1; ++pass) {
#define ATTR(TH, VPL, R, S) CK(hipFuncSetAttribute((const void*)pat<TH, VPL, R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024))
    ATTR(256, 1, 1, false);
    ATTR(1024, 1, 1, false);
    ATTR(512, 1, 1, false);
    ATTR(512, 1, 1, true);
#undef ATTR
  }
  g_a0 = act; g_a1 = act + (int64_t)K * B; g_t0 = obs; g_t1 = st;

  // pad: extra floats per row of every stream (row pitch B + pad): the streams' rows are then no longer 16 MiB multiples apart
  auto headline = [&](int order, int layout = 0, int64_t pad = 0) {  // layout 1: observations as [8][K+1][B] (every component its own stream)
    Streams s{};
    s.nr = 2;
    for (int r = 0; r < 2; ++r) { s.rd[r] = act + (int64_t)r * K * B; s.rd_rs[r] = B; }  // lane-major actions [A][K][B]-like: two streams
    s.nw = 15;
    float* w[15]; int64_t rs[15];
    const int64_t BP = B + pad;
    for (int c = 0; c < 8; ++c) { w[c] = layout ? obs + (int64_t)c * ROWS * BP : obs + (int64_t)c * BP; rs[c] = layout ? BP : 8 * BP; }
    for (int j = 0; j < 7; ++j) { w[8 + j] = st + (int64_t)j * ROWS * BP; rs[8 + j] = BP; }
    if (order == 0) for (int q = 0; q < 15; ++q) { s.wr[q] = w[q]; s.wr_rs[q] = rs[q]; }
    else {
      int o = 0, j = 8, q = 0;
      while (q < 15) { if (o < 8) { s.wr[q] = w[o]; s.wr_rs[q++] = rs[o++]; } if (j < 15 && q < 15) { s.wr[q] = w[j]; s.wr_rs[q++] = rs[j++]; } }
    }
    return s;
  };
  const double HB = (double)(2 + 15) * rb * K;
  std::vector<Variant> vs;
  auto add = [&](const char* n, int wg, int vpl, int rows, int sync, int occ, Streams s, double bytes) {
    Variant v; v.name = n; v.wg = wg; v.vpl = vpl; v.rows = rows; v.sync = sync; v.occ = occ; v.s = s; v.bytes = bytes; vs.push_back(v);
  };
  add("base wg256 vpl1 rows1 (the product kernel's shape)", 256, 1, 1, 0, 0, headline(0), HB);
  add("order: observations and states alternating", 256, 1, 1, 0, 0, headline(1), HB);
  add("rows2", 256, 1, 2, 0, 0, headline(0), HB);
  add("rows4", 256, 1, 4, 0, 0, headline(0), HB);
  add("rows4 sync", 256, 1, 4, 1, 0, headline(0), HB);
  add("sync per row", 256, 1, 1, 1, 0, headline(0), HB);
  add("vpl2 (2 KiB per wave, stream and row)", 256, 2, 1, 0, 0, headline(0), HB);
  add("vpl2 rows2", 256, 2, 2, 0, 0, headline(0), HB);
  add("wg512", 512, 1, 1, 0, 0, headline(0), HB);
  add("wg1024", 1024, 1, 1, 0, 0, headline(0), HB);
  add("wg1024 sync per row", 1024, 1, 1, 1, 0, headline(0), HB);
  add("wg1024 vpl2", 1024, 2, 1, 0, 0, headline(0), HB);
  add("occ1 wg256 (4 waves per CU)", 256, 1, 1, 0, 1, headline(0), HB);
  add("occ2 wg256 (8 waves per CU)", 256, 1, 1, 0, 2, headline(0), HB);
  add("occ4 wg256 (16 waves per CU)", 256, 1, 1, 0, 4, headline(0), HB);
  add("occ1 wg1024 (16 waves per CU)", 1024, 1, 1, 0, 1, headline(0), HB);
  add("row pitch + 4 KiB", 256, 1, 1, 0, 0, headline(0, 0, 1024), HB);
  add("row pitch + 68 KiB", 256, 1, 1, 0, 0, headline(0, 0, 17 * 1024), HB);
  add("row pitch + 1 MiB + 4 KiB", 256, 1, 1, 0, 0, headline(0, 0, 262144 + 1024), HB);
  add("row pitch + 2 MiB", 256, 1, 1, 0, 0, headline(0, 0, 524288), HB);
  add("wg512 sync per row", 512, 1, 1, 1, 0, headline(0), HB);
  add("vpl2 sync per row", 256, 2, 1, 1, 0, headline(0), HB);
  add("obs [8][K+1][B] (component-major observations)", 256, 1, 1, 0, 0, headline(0, 1), HB);
  add("obs [8][K+1][B] order alternating", 256, 1, 1, 0, 0, headline(1, 1), HB);
  add("obs [8][K+1][B] sync per row", 256, 1, 1, 1, 0, headline(0, 1), HB);
  add("obs [8][K+1][B] vpl2", 256, 2, 1, 0, 0, headline(0, 1), HB);
  add("obs [8][K+1][B] wg1024 sync per row", 1024, 1, 1, 1, 0, headline(0, 1), HB);
  add("obs [8][K+1][B] wg512 sync per row", 512, 1, 1, 1, 0, headline(0, 1), HB);
  { Streams s = headline(0, 1); s.nr = 0; add("obs [8][K+1][B] writes only", 256, 1, 1, 0, 0, s, 15.0 * rb * K); }
  { Streams s = headline(0); s.nr = 0; add("writes only", 256, 1, 1, 0, 0, s, 15.0 * rb * K); }
  { Streams s = headline(0); s.nw = 0; add("reads only (2 streams)", 256, 1, 1, 0, 0, s, 2.0 * rb * K); }
  // n equal write streams over the same bytes: stream i has 15 K / n rows of B; even i in the observation buffer, odd in the states'
  for (int n : {1, 2, 4, 6, 10, 15, 30}) {
    Streams s{};
    s.nr = 0; s.nw = n;
    const int64_t rows_each = 15 * K / n;
    int64_t lo = 0, hi = 0;
    for (int i = 0; i < n; ++i) {
      const bool up = (i & 1) && n > 1;
      float* basep = up ? st : obs;
      int64_t& used = up ? hi : lo;
      s.wr[i] = basep + used; s.wr_rs[i] = B; used += rows_each * B;
    }
    if (lo * 4 > GAP || hi * 4 > HI) { fprintf(stderr, "stream sweep does not fit\n"); continue; }
    char nm[96]; snprintf(nm, sizeof nm, "nw%-2d equal write streams (writes only, %lld rows each)", n, (long long)rows_each);
    Variant v; v.name = nm; v.s = s; v.bytes = (double)n * rows_each * rb; v.rows = 1;
    vs.push_back(v);
    // K for this variant differs: stash in occ as a negative marker is ugly — use tiled=2 + bytes to carry rows
    vs.back().tiled = 0; vs.back().occ = 0; vs.back().sync = 0; vs.back().wg = 256; vs.back().vpl = 1;
    vs.back().rows = 1;
    vs.back().s.nr = -(int)rows_each;  // negative nr carries the row count (no read streams)
  }
  { Variant v; v.name = "tiled [K][B/1024][C][1024] (streams of a workgroup adjacent)"; v.tiled = 1; v.s = headline(0); v.bytes = HB; vs.push_back(v); }

  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea));
  CK(hipEventCreate(&eb));
  auto timeit = [&](auto&& fn) {
    fn(); fn();
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(ea, 0)); fn(); CK(hipEventRecord(eb, 0)); CK(hipEventSynchronize(eb));
      float ms; CK(hipEventElapsedTime(&ms, ea, eb)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t;
  };
  // the fill rate of the same write ranges (one launch per buffer)
  auto tf = timeit([&] {
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, obs, obs_bytes / 16);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, st, st_bytes / 16);
  });
  printf("fill of the write ranges: %.0f GB/s (ms %.3f)\n", (obs_bytes + st_bytes) / tf[tf.size() / 2] / 1e6, tf[tf.size() / 2]);
  auto tf2 = timeit([&] { hipLaunchKernelGGL(fill2, dim3(4096), dim3(256), 0, 0, obs, obs_bytes / 16, st, st_bytes / 16); });
  printf("fill of both ranges at once (even workgroups one buffer, odd the other): %.0f GB/s (ms %.3f)\n", (obs_bytes + st_bytes) / tf2[tf2.size() / 2] / 1e6, tf2[tf2.size() / 2]);
  for (auto& v : vs) {
    if (only && v.name.find(only) == std::string::npos) continue;
    int k = K;
    Variant run = v;
    if (v.s.nr < 0) { k = -v.s.nr; run.s.nr = 0; }
    auto t = timeit([&] { launch(run, B, k); });
    CK(hipGetLastError());
    const double med = t[t.size() / 2];
    printf("%-66s %6.0f GB/s  ms med %.3f min %.3f max %.3f\n", v.name.c_str(), v.bytes / med / 1e6, med, t.front(), t.back());
    fflush(stdout);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
