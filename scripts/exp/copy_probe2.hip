// Developer experiment, round 2: why does a streaming copy on this pool's MI355X stop at 5.4-5.7 TB/s when the
// guide quotes 6.29 for a float4 copy?  Sweeps footprint, store/load cache policy (inline asm modifiers),
// workgroup -> tile mappings (interleaved vs one contiguous span per XCD), persistent grids, block sizes and the
// relative placement of source and destination.
// build: hipcc -O3 --offload-arch=gfx950 copy_probe2.hip -o copy_probe2 ; run: ./copy_probe2 [GiB] [section...]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- loads / stores with explicit cache-policy modifiers
enum Pol { P_DEF = 0, P_NT = 1, P_SC1 = 2, P_SC0SC1 = 3, P_NTSC1 = 4, P_NTSC0SC1 = 5, P_SC0 = 6 };
static const char* pol_name(int p) {
  static const char* n[] = {"default", "nt", "sc1", "sc0 sc1", "nt sc1", "nt sc0 sc1", "sc0"};
  return n[p];
}
template <int P> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
  u32x4 v;
  if (P == P_DEF) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (P == P_NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (P == P_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (P == P_SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (P == P_NTSC1) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (P == P_NTSC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (P == P_SC0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int P> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if (P == P_DEF) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  if (P == P_NT) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  if (P == P_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  if (P == P_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
  if (P == P_NTSC1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
  if (P == P_NTSC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
  if (P == P_SC0) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- workgroup -> tile mappings
enum Map { M_LINEAR = 0, M_XCD_SPAN = 1, M_XCD_SPAN_REV = 2 };
__device__ __forceinline__ size_t map_tile(int map, size_t bid, size_t ntiles) {
  if (map == M_LINEAR) return bid;
  // workgroups are dealt to the 8 XCDs round-robin: XCD x sees bid = x, x + 8, ...; give it ONE contiguous eighth
  const size_t per = (ntiles + 7) / 8;
  const size_t x = bid & 7, k = bid >> 3;
  size_t t = x * per + k;
  if (map == M_XCD_SPAN_REV) t = x * per + (per - 1 - k);
  return (k < per && t < ntiles) ? t : (size_t)-1;
}

// one tile per workgroup (or a persistent loop over tiles when gridDim.x < ntiles)
template <int U, int PL, int PS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_tiles(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n,
                                                    int tile_vecs, size_t ntiles, int map) {
  const size_t padded = map == M_LINEAR ? ntiles : ((ntiles + 7) / 8) * 8;
  for (size_t bid = blockIdx.x; bid < padded; bid += gridDim.x) {
    const size_t t = map_tile(map, bid, ntiles);
    if (t == (size_t)-1) continue;
    const size_t base = t * (size_t)tile_vecs;
    for (int k = threadIdx.x; k < tile_vecs; k += BLOCK * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const size_t j = base + k + u * BLOCK; if (j < n && k + u * BLOCK < tile_vecs) v[u] = ld<PL>(src + j); }
      wait_loads();
#pragma unroll
      for (int u = 0; u < U; ++u) { const size_t j = base + k + u * BLOCK; if (j < n && k + u * BLOCK < tile_vecs) st<PS>(dst + j, v[u]); }
    }
  }
}

template <int U, int PL>
__global__ __launch_bounds__(256) void read_tiles(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n, int tile_vecs,
                                                  size_t ntiles, int map) {
  const size_t t = map_tile(map, blockIdx.x, ntiles);
  if (t == (size_t)-1) return;
  const size_t base = t * (size_t)tile_vecs;
  u32x4 acc = {0, 0, 0, 0};
  for (int k = threadIdx.x; k < tile_vecs; k += 256 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = base + k + u * 256; if (j < n && k + u * 256 < tile_vecs) v[u] = ld<PL>(src + j); else v[u] = acc; }
    wait_loads();
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if (acc.x == 0x12345678u) dst[0] = acc;
}

template <int PS>
__global__ __launch_bounds__(256) void write_tiles(u32x4* __restrict__ dst, size_t n, int tile_vecs, size_t ntiles, int map) {
  const size_t t = map_tile(map, blockIdx.x, ntiles);
  if (t == (size_t)-1) return;
  const size_t base = t * (size_t)tile_vecs;
  const u32x4 v = {1, 2, 3, 4};
  for (int k = threadIdx.x; k < tile_vecs; k += 256) { const size_t j = base + k; if (j < n) st<PS>(dst + j, v); }
}

template <typename F> static float timeit(F f, int iters = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) {
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  hipEventDestroy(a); hipEventDestroy(b);
  return ts[ts.size() / 2];
}

static void rep(const char* name, float ms, double moved) { printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms, moved / ms / 1e9); fflush(stdout); }

template <int U, int PL, int PS, int BLOCK = 256>
static float run_copy(const u32x4* src, u32x4* dst, size_t n, int tile_kib, int map, unsigned grid_cap = 0) {
  const int tile_vecs = tile_kib * 1024 / 16;
  const size_t ntiles = (n + tile_vecs - 1) / tile_vecs;
  const size_t padded = map == M_LINEAR ? ntiles : ((ntiles + 7) / 8) * 8;
  const unsigned grid = grid_cap ? grid_cap : (unsigned)padded;
  return timeit([&] { hipLaunchKernelGGL((copy_tiles<U, PL, PS, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, src, dst, n, tile_vecs, ntiles, map); });
}

static bool want(int argc, char** argv, const char* sec) {
  if (argc <= 2) return true;
  for (int i = 2; i < argc; ++i) if (!strcmp(argv[i], sec)) return true;
  return false;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  const size_t bytes = (size_t)(gib * (1ull << 30));
  const size_t slack = 64ull << 20;
  u32x4 *src, *dst_base;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst_base, bytes + slack) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, bytes); hipMemset(dst_base, 0, bytes + slack);
  u32x4* dst = dst_base;
  const size_t n = bytes / 16;
  printf("buffers %.2f GiB each; src %p dst %p\n", gib, (void*)src, (void*)dst);
  char nm[160];

  if (want(argc, argv, "foot")) {
    printf("--- footprint (tile 64 KiB, U=4, nt/nt, linear)\n");
    for (double g : {0.0625, 0.25, 1.0, 4.0, 16.0}) {
      if (g > gib) break;
      const size_t nn = (size_t)(g * (1ull << 30)) / 16;
      snprintf(nm, sizeof nm, "copy %.4g GiB", g);
      rep(nm, run_copy<4, P_NT, P_NT>(src, dst, nn, 64, M_LINEAR), 2.0 * nn * 16);
    }
  }
  if (want(argc, argv, "policy")) {
    printf("--- cache policy (tile 64 KiB, U=4, linear): load policy x store policy\n");
#define POL(PL, PS) { snprintf(nm, sizeof nm, "load %-10s store %-10s", pol_name(PL), pol_name(PS)); rep(nm, run_copy<4, PL, PS>(src, dst, n, 64, M_LINEAR), 2.0 * bytes); }
    POL(P_DEF, P_DEF) POL(P_NT, P_NT) POL(P_NT, P_DEF) POL(P_DEF, P_NT)
    POL(P_NT, P_SC1) POL(P_NT, P_SC0SC1) POL(P_NT, P_NTSC1) POL(P_NT, P_NTSC0SC1) POL(P_NT, P_SC0)
    POL(P_SC1, P_NT) POL(P_SC0SC1, P_NT) POL(P_NTSC1, P_NT) POL(P_NTSC0SC1, P_NT) POL(P_NTSC0SC1, P_NTSC0SC1)
    POL(P_SC1, P_SC1) POL(P_SC0SC1, P_SC0SC1)
  }
  if (want(argc, argv, "map")) {
    printf("--- workgroup -> tile mapping (U=4, nt/nt)\n");
    for (int tile : {16, 64, 256, 1024}) {
      for (int map : {M_LINEAR, M_XCD_SPAN, M_XCD_SPAN_REV}) {
        snprintf(nm, sizeof nm, "tile %4d KiB  %s", tile, map == M_LINEAR ? "linear (XCDs interleaved)" : map == M_XCD_SPAN ? "one contiguous span per XCD" : "span per XCD, walked backwards");
        rep(nm, run_copy<4, P_NT, P_NT>(src, dst, n, tile, map), 2.0 * bytes);
      }
    }
    printf("--- persistent grids (tile 64 KiB, U=4, nt/nt)\n");
    for (unsigned g : {256u, 512u, 1024u, 2048u, 4096u, 8192u}) {
      for (int map : {M_LINEAR, M_XCD_SPAN}) {
        snprintf(nm, sizeof nm, "grid %5u  %s", g, map == M_LINEAR ? "linear" : "span per XCD");
        rep(nm, run_copy<4, P_NT, P_NT>(src, dst, n, 64, map, g), 2.0 * bytes);
      }
    }
  }
  if (want(argc, argv, "small")) {
    printf("--- small tiles: one load batch + one store batch per workgroup (nt/nt)\n");
#define SM(U, BLK, KIB) for (int map : {M_LINEAR, M_XCD_SPAN}) { snprintf(nm, sizeof nm, "tile %3d KiB = block %4d x U=%d  %s", KIB, BLK, U, map == M_LINEAR ? "linear" : "span per XCD"); \
      rep(nm, run_copy<U, P_NT, P_NT, BLK>(src, dst, n, KIB, map), 2.0 * bytes); }
    SM(4, 64, 4) SM(2, 128, 4) SM(1, 256, 4)
    SM(8, 64, 8) SM(4, 128, 8) SM(2, 256, 8) SM(1, 512, 8)
    SM(8, 128, 16) SM(4, 256, 16) SM(2, 512, 16) SM(1, 1024, 16)
    SM(8, 256, 32) SM(4, 512, 32) SM(2, 1024, 32)
    SM(4, 1024, 64)
    printf("--- 16 KiB tiles, block 256, U=4: policies with the span mapping\n");
#define POLS(PL, PS) { snprintf(nm, sizeof nm, "span: load %-10s store %-10s", pol_name(PL), pol_name(PS)); rep(nm, run_copy<4, PL, PS>(src, dst, n, 16, M_XCD_SPAN), 2.0 * bytes); }
    POLS(P_DEF, P_DEF) POLS(P_NT, P_NT) POLS(P_NT, P_DEF) POLS(P_DEF, P_NT) POLS(P_NT, P_NTSC0SC1) POLS(P_NTSC0SC1, P_NTSC0SC1)
  }
  if (want(argc, argv, "shape")) {
    printf("--- unroll / block size (tile 64 KiB, nt/nt, linear)\n");
    rep("U=1 block 256", run_copy<1, P_NT, P_NT>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=2 block 256", run_copy<2, P_NT, P_NT>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=4 block 256", run_copy<4, P_NT, P_NT>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=8 block 256", run_copy<8, P_NT, P_NT>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=16 block 256", run_copy<16, P_NT, P_NT>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=4 block 64", run_copy<4, P_NT, P_NT, 64>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=4 block 128", run_copy<4, P_NT, P_NT, 128>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=4 block 512", run_copy<4, P_NT, P_NT, 512>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=4 block 1024", run_copy<4, P_NT, P_NT, 1024>(src, dst, n, 64, M_LINEAR), 2.0 * bytes);
    rep("U=8 block 64  (tile 16 KiB)", run_copy<8, P_NT, P_NT, 64>(src, dst, n, 16, M_LINEAR), 2.0 * bytes);
  }
  if (want(argc, argv, "offset")) {
    printf("--- placement of dst relative to src (tile 64 KiB, U=4, nt/nt, linear)\n");
    for (size_t off : {(size_t)0, (size_t)4096, (size_t)65536, (size_t)(1 << 20), (size_t)(2 << 20) + 4096, (size_t)(32 << 20) + 128 * 1024}) {
      snprintf(nm, sizeof nm, "dst + %zu bytes", off);
      rep(nm, run_copy<4, P_NT, P_NT>(src, (u32x4*)((char*)dst_base + off), n, 64, M_LINEAR), 2.0 * bytes);
    }
  }
  if (want(argc, argv, "rw")) {
    printf("--- one direction only (tile 64 KiB)\n");
    const int tile_vecs = 64 * 1024 / 16;
    const size_t ntiles = (n + tile_vecs - 1) / tile_vecs;
    const unsigned grid = (unsigned)(((ntiles + 7) / 8) * 8);
    for (int map : {M_LINEAR, M_XCD_SPAN}) {
      const char* mn = map == M_LINEAR ? "linear" : "span per XCD";
#define RD(PL) { snprintf(nm, sizeof nm, "read-only  U=8 %-10s %s", pol_name(PL), mn); rep(nm, timeit([&] { hipLaunchKernelGGL((read_tiles<8, PL>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs, ntiles, map); }), 1.0 * bytes); }
#define WR(PS) { snprintf(nm, sizeof nm, "write-only     %-10s %s", pol_name(PS), mn); rep(nm, timeit([&] { hipLaunchKernelGGL((write_tiles<PS>), dim3(grid), dim3(256), 0, 0, dst, n, tile_vecs, ntiles, map); }), 1.0 * bytes); }
      RD(P_DEF) RD(P_NT) RD(P_SC1) RD(P_NTSC0SC1)
      WR(P_DEF) WR(P_NT) WR(P_SC1) WR(P_SC0SC1) WR(P_NTSC0SC1)
    }
  }
  return 0;
}
