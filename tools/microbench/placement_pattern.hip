// Diagnostic (not product): what makes the persistent-lane store pattern of sim_ahead_kernel slow on some buffer placements?
// The headline's traffic shape without arithmetic: NR = 2 action streams read, NW = 15 trajectory streams written
// (8 observation components inside ONE [K+1][8][B] buffer + 7 state leaves [K+1][B]), 16 B per lane, 4 KiB per workgroup and
// stream per step. Streams are described by a table (base, row stride), so the same kernel runs over
//   * separate hipMalloc's per buffer (what torch's allocator does for large tensors),
//   * one arena with the buffers carved back to back (the consistently slow case of profiles/r02_placement_sensitivity.md),
//   * the arena with per-stream skews / padded component strides / a different workgroup -> environment mapping,
// and a plain fill (one sequential stream per workgroup, grid-stride) over the same bytes tells whether the platform's write
// ceiling moved with the placement too.
//   hipcc -O3 --offload-arch=gfx950 placement_pattern.hip -o placement_pattern && ./placement_pattern
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int NR = 2, NW = 15;

struct Streams {
  const float* rd[NR];
  int64_t rd_rs[NR];
  float* wr[NW];
  int64_t wr_rs[NW];
};

// MAP 0: workgroup w owns envs [1024 w, 1024 (w+1));  MAP 1: XCD-contiguous (w % 8 picks an eighth of the batch)
// CHUNK: rows staged per stream before they are stored (1 = the product kernel's behaviour)
// FL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1, 6 sc0   (gfx950 cache-policy bits of global_store)
template <int FL> __device__ __forceinline__ void store_fl(v4* p, v4 v) {
  if (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (FL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (FL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int FL>
__global__ void __launch_bounds__(256) pattern_fl(const Streams s, int64_t B, int K, int nwg) {
  const int64_t off = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  if (off >= B) return;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  v4 nxt[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(s.rd[r] + off);
  for (int k = 0; k < K; ++k) {
    v4 cur[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) cur[r] = nxt[r];
    const int kn = (k + 1 < K) ? k + 1 : k;
#pragma unroll
    for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(s.rd[r] + (int64_t)kn * s.rd_rs[r] + off);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += cur[r];
#pragma unroll
    for (int q = 0; q < NW; ++q) store_fl<FL>((v4*)(s.wr[q] + (int64_t)k * s.wr_rs[q] + off), acc + (float)q);
  }
}

template <int MAP, bool NT>
__global__ void __launch_bounds__(256) pattern(const Streams s, int64_t B, int K, int nwg) {
  int64_t w = blockIdx.x;
  if (MAP == 1) w = (int64_t)(blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
  const int64_t off = w * 1024 + threadIdx.x * 4;
  if (off >= B) return;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  v4 nxt[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(s.rd[r] + off);
  for (int k = 0; k < K; ++k) {
    v4 cur[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) cur[r] = nxt[r];
    const int kn = (k + 1 < K) ? k + 1 : k;
#pragma unroll
    for (int r = 0; r < NR; ++r) nxt[r] = *(const v4*)(s.rd[r] + (int64_t)kn * s.rd_rs[r] + off);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += cur[r];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      v4 v = acc + (float)q;
      v4* p = (v4*)(s.wr[q] + (int64_t)k * s.wr_rs[q] + off);
      if (NT) __builtin_nontemporal_store(v, p);
      else *p = v;
    }
  }
}

// plain fill: every workgroup writes 4 KiB pieces of ONE sequential range, grid-stride
__global__ void __launch_bounds__(256) fill(float* p, int64_t n_v4) {
  v4 v = {1.f, 2.f, 3.f, 4.f};
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n_v4; i += (int64_t)gridDim.x * 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n_v4) ((v4*)p)[i + u * 256] = v;
  }
}

static double time_ms(hipStream_t st, int reps, void (*launch)(void*), void* ctx, std::vector<float>* all = nullptr) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  launch(ctx);
  launch(ctx);
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) {
    (void)hipEventRecord(a, st);
    launch(ctx);
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  if (all) *all = t;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return t[t.size() / 2];
}

struct Ctx {
  Streams s;
  int64_t B;
  int K, map, nt;
};
static void launch_pattern(void* p) {
  Ctx* c = (Ctx*)p;
  const int nwg = (int)(c->B / 1024);
  dim3 g(nwg), b(256);
  if (c->map >= 100) {
    switch (c->map - 100) {
      case 0: hipLaunchKernelGGL((pattern_fl<0>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
      case 1: hipLaunchKernelGGL((pattern_fl<1>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
      case 2: hipLaunchKernelGGL((pattern_fl<2>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
      case 3: hipLaunchKernelGGL((pattern_fl<3>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
      case 4: hipLaunchKernelGGL((pattern_fl<4>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
      case 5: hipLaunchKernelGGL((pattern_fl<5>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
      default: hipLaunchKernelGGL((pattern_fl<6>), g, b, 0, 0, c->s, c->B, c->K, nwg); break;
    }
    return;
  }
  if (c->map == 0 && c->nt) hipLaunchKernelGGL((pattern<0, true>), g, b, 0, 0, c->s, c->B, c->K, nwg);
  else if (c->map == 0) hipLaunchKernelGGL((pattern<0, false>), g, b, 0, 0, c->s, c->B, c->K, nwg);
  else if (c->nt) hipLaunchKernelGGL((pattern<1, true>), g, b, 0, 0, c->s, c->B, c->K, nwg);
  else hipLaunchKernelGGL((pattern<1, false>), g, b, 0, 0, c->s, c->B, c->K, nwg);
}
struct FillCtx {
  std::vector<std::pair<float*, int64_t>> ranges;  // (base, bytes)
};
static void launch_fill(void* p) {
  FillCtx* c = (FillCtx*)p;
  for (auto& r : c->ranges) hipLaunchKernelGGL(fill, dim3(256 * 64), dim3(256), 0, 0, r.first, r.second / 16);
}

static const int64_t B = 1 << 22;
static const int K = 100;
static const double BYTES = (double)(NR + NW) * 4.0 * B * K;

// the product's buffer set carved out of `base` (bytes): actions [K][2][B], obs [K+1][8][B (+pad)], 7 x state [K+1][B];
// skew = extra bytes between consecutive buffers, cpad = extra floats per component row of the obs / action buffers
static size_t carve(char* base, Streams& s, int64_t skew, int64_t cpad, int64_t spad, FillCtx* fc) {
  size_t off = 0;
  const int64_t cs = B + cpad;
  for (int r = 0; r < NR; ++r) {
    s.rd[r] = (const float*)(base + off) + (int64_t)r * cs;
    s.rd_rs[r] = NR * cs;
  }
  off += (size_t)K * NR * cs * 4 + skew;
  off = (off + 4095) & ~(size_t)4095;
  if (fc) fc->ranges.clear();
  for (int q = 0; q < 8; ++q) {
    s.wr[q] = (float*)(base + off) + (int64_t)q * cs;
    s.wr_rs[q] = 8 * cs;
  }
  if (fc) fc->ranges.push_back({(float*)(base + off), (int64_t)K * 8 * cs * 4});
  off += (size_t)(K + 1) * 8 * cs * 4 + skew;
  off = (off + 4095) & ~(size_t)4095;
  const int64_t ss = B + spad;
  for (int j = 0; j < 7; ++j) {
    s.wr[8 + j] = (float*)(base + off);
    s.wr_rs[8 + j] = ss;
    if (fc) fc->ranges.push_back({(float*)(base + off), (int64_t)K * ss * 4});
    off += (size_t)(K + 1) * ss * 4 + skew * (j + 2);
    off = (off + 4095) & ~(size_t)4095;
  }
  return off;
}

static void report(const char* name, Ctx& c, FillCtx* fc) {
  std::vector<float> all;
  const double ms = time_ms(0, 9, launch_pattern, &c, &all);
  printf("%-58s pattern %6.0f GB/s  (ms min/med/max %.2f %.2f %.2f)", name, BYTES / ms / 1e6, all.front(), ms, all.back());
  if (fc) {
    double fb = 0;
    for (auto& r : fc->ranges) fb += (double)r.second;
    const double fms = time_ms(0, 5, launch_fill, fc);
    printf("   fill of the same write ranges %6.0f GB/s", fb / fms / 1e6);
  }
  printf("\n");
  fflush(stdout);
}

// ---- experiment set "wide": one very large arena; does the ABSOLUTE position of a compact buffer set matter, and does
// spreading the buffers (or the 15 streams as 15 separate [K+1][B] buffers) over the whole arena help? ----
static int wide_main(size_t arena_gb) {
  char* arena;
  const size_t arena_bytes = arena_gb << 30;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena %zu GB: alloc failed\n", arena_gb); return 1; }
  printf("arena %zu GB at %p\n", arena_gb, (void*)arena);
  const size_t set_bytes = (size_t)30 << 30;
  for (size_t off_gb = 0; (off_gb << 30) + set_bytes <= arena_bytes; off_gb += 32) {
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    FillCtx fc;
    carve(arena + (off_gb << 30), c.s, 0, 0, 0, &fc);
    (void)hipMemset((void*)c.s.rd[0], 0, (size_t)K * NR * B * 4);
    char name[128];
    snprintf(name, sizeof name, "W compact set at arena + %zu GB", off_gb);
    report(name, c, &fc);
  }
  // 15 + 2 independent streams (every stream its own [K+1][B] buffer, 1.7 GB each), compact vs spread over the arena
  for (int spread = 0; spread < 4; ++spread) {
    const size_t one = (size_t)(K + 1) * B * 4;
    size_t pitch = one;
    if (spread == 1) pitch = (arena_bytes / (NR + NW)) & ~(size_t)((2 << 20) - 1);
    if (spread == 2) pitch = one + ((size_t)1 << 30);                      // 1.58 + 1 GiB
    if (spread == 3) pitch = (size_t)4 << 30;                              // exactly 4 GiB apart
    if (pitch * (NR + NW) > arena_bytes) continue;
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    FillCtx fc;
    for (int r = 0; r < NR; ++r) { c.s.rd[r] = (const float*)(arena + pitch * r); c.s.rd_rs[r] = B; (void)hipMemset((void*)c.s.rd[r], 0, one); }
    for (int q = 0; q < NW; ++q) {
      c.s.wr[q] = (float*)(arena + pitch * (NR + q));
      c.s.wr_rs[q] = B;
      fc.ranges.push_back({c.s.wr[q], (int64_t)K * B * 4});
    }
    char name[128];
    snprintf(name, sizeof name, "W 17 separate stream buffers, pitch %.2f GiB", (double)pitch / (1 << 30));
    report(name, c, &fc);
  }
  // the product's 9 buffers spread evenly over the arena
  {
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    FillCtx fc;
    const size_t pitch = (arena_bytes / 9) & ~(size_t)((2 << 20) - 1);
    if (pitch >= (size_t)(K + 1) * 8 * B * 4) {
      for (int r = 0; r < NR; ++r) { c.s.rd[r] = (const float*)arena + (int64_t)r * B; c.s.rd_rs[r] = NR * B; }
      (void)hipMemset(arena, 0, (size_t)K * NR * B * 4);
      float* obs = (float*)(arena + pitch);
      for (int q = 0; q < 8; ++q) { c.s.wr[q] = obs + (int64_t)q * B; c.s.wr_rs[q] = 8 * B; }
      fc.ranges.push_back({obs, (int64_t)K * 8 * B * 4});
      for (int j = 0; j < 7; ++j) {
        c.s.wr[8 + j] = (float*)(arena + pitch * (2 + j));
        c.s.wr_rs[8 + j] = B;
        fc.ranges.push_back({c.s.wr[8 + j], (int64_t)K * B * 4});
      }
      report("W product's 9 buffers spread evenly over the arena", c, &fc);
    }
  }
  // sequential fill ceiling by position: 8 GiB windows
  for (size_t off_gb = 0; off_gb + 8 <= arena_gb; off_gb += 24) {
    FillCtx fc;
    fc.ranges.push_back({(float*)(arena + (off_gb << 30)), (int64_t)8 << 30});
    const double fms = time_ms(0, 5, launch_fill, &fc);
    printf("W fill 8 GiB at arena + %3zu GB: %6.0f GB/s\n", off_gb, (double)((int64_t)8 << 30) / fms / 1e6);
  }
  (void)hipFree(arena);
  return 0;
}

// ---- experiment set "map": structure of the placement effect over one large arena ----
static int map_main(size_t arena_gb) {
  char* arena;
  const size_t arena_bytes = arena_gb << 30;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena %zu GB: alloc failed\n", arena_gb); return 1; }
  printf("arena %zu GB at %p\n", arena_gb, (void*)arena);
  printf("M fill GB/s per 4 GiB window:");
  for (size_t off_gb = 0; off_gb + 4 <= arena_gb; off_gb += 4) {
    FillCtx fc;
    fc.ranges.push_back({(float*)(arena + (off_gb << 30)), (int64_t)4 << 30});
    const double fms = time_ms(0, 5, launch_fill, &fc);
    if (off_gb % 32 == 0) printf("\n  +%3zu GB:", off_gb);
    printf(" %5.0f", (double)((int64_t)4 << 30) / fms / 1e6);
  }
  printf("\nM pattern GB/s, compact 30 GB set starting every 8 GB:");
  const size_t set_bytes = (size_t)30 << 30;
  std::vector<double> rate;
  for (size_t off_gb = 0; (off_gb << 30) + set_bytes <= arena_bytes; off_gb += 8) {
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    carve(arena + (off_gb << 30), c.s, 0, 0, 0, nullptr);
    const double ms = time_ms(0, 5, launch_pattern, &c);
    if (off_gb % 64 == 0) printf("\n  +%3zu GB:", off_gb);
    printf(" %5.0f", BYTES / ms / 1e6);
    rate.push_back(BYTES / ms / 1e6);
  }
  printf("\n");
  // streams split between two windows: actions + obs in window a, the seven state leaves in window b
  auto split = [&](size_t a_gb, size_t b_gb) {
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    Streams sa, sb;
    carve(arena + (a_gb << 30), sa, 0, 0, 0, nullptr);
    carve(arena + (b_gb << 30), sb, 0, 0, 0, nullptr);
    c.s = sa;
    for (int j = 0; j < 7; ++j) { c.s.wr[8 + j] = sb.wr[8 + j]; c.s.wr_rs[8 + j] = sb.wr_rs[8 + j]; }
    const double ms = time_ms(0, 5, launch_pattern, &c);
    printf("M split: actions+obs in window +%zu GB, state leaves in window +%zu GB: %5.0f GB/s\n", a_gb, b_gb, BYTES / ms / 1e6);
  };
  split(0, 0);
  split(0, 32);
  split(0, 64);
  split(32, 96);
  if (arena_gb >= 224) { split(0, 128); split(0, 192); split(128, 192); split(160, 160); split(64, 160); }
  // write streams only / read streams only in a slow and a fast window
  (void)hipFree(arena);
  return 0;
}

// ---- experiment set "flavour": cache policy of the trajectory stores, in a slow and in a fast window of one arena ----
static int flavour_main(size_t arena_gb) {
  char* arena;
  const size_t arena_bytes = arena_gb << 30;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena %zu GB: alloc failed\n", arena_gb); return 1; }
  const char* names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"};
  for (size_t off_gb : {(size_t)0, (size_t)16, (size_t)32, (size_t)56}) {
    if (((off_gb + 30) << 30) > arena_bytes) continue;
    Ctx c;
    c.B = B; c.K = K; c.nt = 1;
    carve(arena + (off_gb << 30), c.s, 0, 0, 0, nullptr);
    (void)hipMemset((void*)c.s.rd[0], 0, (size_t)K * NR * B * 4);
    printf("F window +%2zu GB:", off_gb);
    for (int f = 0; f < 7; ++f) {
      c.map = 100 + f;
      const double ms = time_ms(0, 5, launch_pattern, &c);
      printf("  %s %5.0f", names[f], BYTES / ms / 1e6);
    }
    c.map = 0;
    const double ms = time_ms(0, 5, launch_pattern, &c);
    printf("  (builtin nt %5.0f)\n", BYTES / ms / 1e6);
    fflush(stdout);
  }
  (void)hipFree(arena);
  return 0;
}

// ---- experiment set "shift": in a slow window, which relative displacement of which buffers removes the slowdown? ----
static int shift_main(size_t arena_gb) {
  char* arena;
  const size_t arena_bytes = arena_gb << 30;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena %zu GB: alloc failed\n", arena_gb); return 1; }
  (void)hipMemset(arena, 0, (size_t)8 << 30);
  auto run = [&](const char* what, Ctx& c) {
    const double ms = time_ms(0, 5, launch_pattern, &c);
    printf("S %-70s %5.0f GB/s\n", what, BYTES / ms / 1e6);
    fflush(stdout);
  };
  Ctx base;
  base.B = B; base.K = K; base.map = 0; base.nt = 1;
  carve(arena, base.s, 0, 0, 0, nullptr);
  run("compact set at arena + 0", base);
  char name[160];
  const size_t MiB = (size_t)1 << 20;
  for (size_t sh : {16 * MiB, 32 * MiB, 64 * MiB, 128 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB, 2048 * MiB, 4096 * MiB, 8192 * MiB,
                    16384 * MiB, 32768 * MiB, 24 * MiB, 1000 * MiB, 3 * 1024 * MiB + 48 * MiB}) {
    Ctx c = base;
    for (int j = 0; j < 7; ++j) c.s.wr[8 + j] = (float*)((char*)base.s.wr[8 + j] + sh);
    snprintf(name, sizeof name, "all 7 state leaves displaced by %zu MiB", sh / MiB);
    run(name, c);
  }
  for (size_t sk : {16 * MiB, 48 * MiB, 80 * MiB, 272 * MiB, 1040 * MiB}) {
    Ctx c = base;
    for (int j = 0; j < 7; ++j) c.s.wr[8 + j] = (float*)((char*)base.s.wr[8 + j] + sk * (j + 1));
    snprintf(name, sizeof name, "state leaf j displaced by (j+1) x %zu MiB", sk / MiB);
    run(name, c);
  }
  {  // only the action reads displaced
    Ctx c = base;
    for (int r = 0; r < NR; ++r) c.s.rd[r] = (const float*)((const char*)base.s.rd[r] + ((size_t)40 << 30));
    (void)hipMemset((void*)c.s.rd[0], 0, (size_t)K * NR * B * 4);
    run("only the action reads displaced by 40 GiB", c);
  }
  {  // only the observations displaced
    Ctx c = base;
    for (int q = 0; q < 8; ++q) c.s.wr[q] = (float*)((char*)base.s.wr[q] + ((size_t)32 << 30));
    run("only the observation buffer displaced by 32 GiB", c);
  }
  for (int only = 0; only < 7; only += 3) {
    Ctx c = base;
    c.s.wr[8 + only] = (float*)((char*)base.s.wr[8 + only] + ((size_t)32 << 30));
    snprintf(name, sizeof name, "only state leaf %d displaced by 32 GiB", only);
    run(name, c);
  }
  {  // states 0..3 displaced, 4..6 in place
    Ctx c = base;
    for (int j = 0; j < 4; ++j) c.s.wr[8 + j] = (float*)((char*)base.s.wr[8 + j] + ((size_t)32 << 30));
    run("state leaves 0..3 displaced by 32 GiB, 4..6 in place", c);
  }
  {  // write streams all pointing into the observation buffer's rows only (8 streams + 7 further components of a wider row)
    Ctx c = base;
    for (int q = 0; q < NW; ++q) { c.s.wr[q] = base.s.wr[0] + (int64_t)q * B; c.s.wr_rs[q] = 15 * B; }
    run("one [K][15][B] buffer (15 component streams of one array)", c);
  }
  run("compact set at arena + 0 (again)", base);
  (void)hipFree(arena);
  return 0;
}

// ---- experiment set "pair": two output sets A and B in one arena, each set's observations and state leaves >= 32 GiB apart:
//   [obsA | statesB | pad to 32 GiB][obsB | statesA]   — at several phases of the arena, against the compact placement ----
static int pair_main(size_t arena_gb) {
  char* arena;
  const size_t arena_bytes = arena_gb << 30;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena %zu GB: alloc failed\n", arena_gb); return 1; }
  const size_t GiB = (size_t)1 << 30;
  const size_t obs_bytes = (size_t)(K + 1) * 8 * B * 4, st_bytes = (size_t)(K + 1) * B * 4;
  float* act = (float*)(arena + arena_bytes - 4 * GiB);
  (void)hipMemset(act, 0, (size_t)K * NR * B * 4);
  auto make = [&](char* obs, char* st0, Ctx& c) {
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    for (int r = 0; r < NR; ++r) { c.s.rd[r] = act + (int64_t)r * B; c.s.rd_rs[r] = NR * B; }
    for (int q = 0; q < 8; ++q) { c.s.wr[q] = (float*)obs + (int64_t)q * B; c.s.wr_rs[q] = 8 * B; }
    for (int j = 0; j < 7; ++j) { c.s.wr[8 + j] = (float*)(st0 + j * st_bytes); c.s.wr_rs[8 + j] = B; }
  };
  printf("P phase GiB | compact | set A (obs at a, states at a+32+obs) | set B (obs at a+32, states at a+obs) | obs at a, states at a+32\n");
  for (size_t a_gb = 0; a_gb + 64 + 4 <= arena_gb; a_gb += 4) {
    char* a = arena + a_gb * GiB;
    Ctx cc, ca, cb, cd;
    make(a, a + obs_bytes, cc);
    make(a, a + 32 * GiB + obs_bytes, ca);
    make(a + 32 * GiB, a + obs_bytes, cb);
    make(a, a + 32 * GiB, cd);
    printf("P %3zu | %5.0f | %5.0f | %5.0f | %5.0f\n", a_gb, BYTES / time_ms(0, 5, launch_pattern, &cc) / 1e6,
           BYTES / time_ms(0, 5, launch_pattern, &ca) / 1e6, BYTES / time_ms(0, 5, launch_pattern, &cb) / 1e6,
           BYTES / time_ms(0, 5, launch_pattern, &cd) / 1e6);
    fflush(stdout);
  }
  (void)hipFree(arena);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "pair")) return pair_main(argc > 2 ? (size_t)atol(argv[2]) : 132);
  if (argc > 1 && !strcmp(argv[1], "shift")) return shift_main(argc > 2 ? (size_t)atol(argv[2]) : 96);
  if (argc > 1 && !strcmp(argv[1], "flavour")) return flavour_main(argc > 2 ? (size_t)atol(argv[2]) : 96);
  if (argc > 1 && !strcmp(argv[1], "map")) return map_main(argc > 2 ? (size_t)atol(argv[2]) : 224);
  if (argc > 1 && !strcmp(argv[1], "wide")) return wide_main(argc > 2 ? (size_t)atol(argv[2]) : 224);
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  // ---- A: separate allocations, several placements (a dummy allocation of varying size shifts everything) ----
  const size_t dummies_mb[] = {0, 129, 1500, 4097, 9000};
  for (size_t dm : dummies_mb) {
    void* dummy = nullptr;
    if (dm) (void)hipMalloc(&dummy, dm << 20);
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    FillCtx fc;
    float *act, *obs, *st[7];
    if (hipMalloc(&act, (size_t)K * NR * B * 4) != hipSuccess) return 1;
    (void)hipMemset(act, 0, (size_t)K * NR * B * 4);
    if (hipMalloc(&obs, (size_t)(K + 1) * 8 * B * 4) != hipSuccess) return 1;
    for (int j = 0; j < 7; ++j)
      if (hipMalloc(&st[j], (size_t)(K + 1) * B * 4) != hipSuccess) return 1;
    for (int r = 0; r < NR; ++r) { c.s.rd[r] = act + (int64_t)r * B; c.s.rd_rs[r] = NR * B; }
    for (int q = 0; q < 8; ++q) { c.s.wr[q] = obs + (int64_t)q * B; c.s.wr_rs[q] = 8 * B; }
    fc.ranges.push_back({obs, (int64_t)K * 8 * B * 4});
    for (int j = 0; j < 7; ++j) { c.s.wr[8 + j] = st[j]; c.s.wr_rs[8 + j] = B; fc.ranges.push_back({st[j], (int64_t)K * B * 4}); }
    char name[128];
    snprintf(name, sizeof name, "A separate hipMalloc per buffer, dummy %zu MB", dm);
    report(name, c, &fc);
    if (dm == 0) {
      c.map = 1; report("A   same, XCD-contiguous workgroup map", c, nullptr); c.map = 0;
      c.nt = 0; report("A   same, plain (cacheable) stores", c, nullptr); c.nt = 1;
    }
    (void)hipFree(act); (void)hipFree(obs);
    for (int j = 0; j < 7; ++j) (void)hipFree(st[j]);
    if (dummy) (void)hipFree(dummy);
    if (quick) break;
  }
  // ---- B: one arena ----
  char* arena;
  const size_t arena_bytes = (size_t)40 << 30;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) { printf("arena alloc failed\n"); return 1; }
  (void)hipMemset(arena, 0, (size_t)K * NR * (B + 65536 * 4) * 4 + (64 << 20));
  struct Var { const char* name; int64_t skew, cpad, spad; int map, nt; };
  const Var vars[] = {
      {"B arena, buffers back to back", 0, 0, 0, 0, 1},
      {"B arena, back to back, XCD-contiguous map", 0, 0, 0, 1, 1},
      {"B arena, back to back, plain stores", 0, 0, 0, 0, 0},
      {"B arena, buffer skew 4 KiB", 4096, 0, 0, 0, 1},
      {"B arena, buffer skew 68 KiB", 68 << 10, 0, 0, 0, 1},
      {"B arena, buffer skew 1 MiB + 4 KiB", (1 << 20) + 4096, 0, 0, 0, 1},
      {"B arena, buffer skew 2 MiB + 68 KiB", (2 << 20) + (68 << 10), 0, 0, 0, 1},
      {"B arena, buffer skew 5 MiB + 324 KiB", (5 << 20) + (324 << 10), 0, 0, 0, 1},
      {"B arena, component pad 1024 floats (4 KiB)", 0, 1024, 0, 0, 1},
      {"B arena, component pad 17408 floats (68 KiB)", 0, 17408, 0, 0, 1},
      {"B arena, component pad 2^18+1024 floats (1 MiB + 4 KiB)", 0, (1 << 18) + 1024, 0, 0, 1},
      {"B arena, component+state pad 17408 floats", 0, 17408, 17408, 0, 1},
      {"B arena, component+state pad 2^18+17408 floats", 0, (1 << 18) + 17408, (1 << 18) + 17408, 0, 1},
      {"B arena, pad 17408 + skew 2 MiB + 68 KiB", (2 << 20) + (68 << 10), 17408, 17408, 0, 1},
      {"B arena, back to back (again)", 0, 0, 0, 0, 1},
  };
  for (const Var& v : vars) {
    Ctx c;
    c.B = B; c.K = K; c.map = v.map; c.nt = v.nt;
    FillCtx fc;
    const size_t used = carve(arena, c.s, v.skew, v.cpad, v.spad, &fc);
    if (used > arena_bytes) { printf("%s: does not fit\n", v.name); continue; }
    report(v.name, c, (v.map == 0 && v.nt == 1) ? &fc : nullptr);
  }
  // ---- C: the same arena carve at other arena offsets (is it the relative or the absolute placement?) ----
  for (size_t shift_mb : {(size_t)3, (size_t)257, (size_t)2049, (size_t)6000}) {
    Ctx c;
    c.B = B; c.K = K; c.map = 0; c.nt = 1;
    FillCtx fc;
    const size_t used = carve(arena + (shift_mb << 20), c.s, 0, 0, 0, &fc);
    if (used + (shift_mb << 20) > arena_bytes) continue;
    (void)hipMemset((void*)c.s.rd[0], 0, (size_t)K * NR * B * 4);
    char name[128];
    snprintf(name, sizeof name, "C arena + %zu MB, back to back", shift_mb);
    report(name, c, &fc);
  }
  (void)hipFree(arena);
  return 0;
}
