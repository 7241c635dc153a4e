// Developer experiment: what does a plain streaming copy reach on this box, as a function of launch
// geometry / unroll / cache policy?  (Sets the ceiling the row mover is compared against.)
// build: hipcc -O3 --offload-arch=gfx950 copy_probe.hip -o copy_probe ; run: ./copy_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_gs(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// tile-per-block variant (like the mover: each block owns a contiguous 256 KiB tile)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_tile2(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n, int tile_vecs) {
  size_t base = (size_t)blockIdx.x * tile_vecs;
  for (int k = threadIdx.x; k < tile_vecs; k += 256 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t j = base + k + u * 256; if (j < n && k + u * 256 < tile_vecs) v[u] = NTL ? __builtin_nontemporal_load(src + j) : src[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t j = base + k + u * 256; if (j < n && k + u * 256 < tile_vecs) { if (NTS) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; } }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_tile(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n, int tile_vecs) {
  size_t base = (size_t)blockIdx.x * tile_vecs;
  for (int k = threadIdx.x; k < tile_vecs; k += 256 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t j = base + k + u * 256; if (j < n && k + u * 256 < tile_vecs) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t j = base + k + u * 256; if (j < n && k + u * 256 < tile_vecs) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; } }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_gs(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if (acc.x == 0x12345678u) dst[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_gs(u32x4* __restrict__ dst, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  u32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

template <typename F> float timeit(F f, int iters = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  double gib = argc > 1 ? atof(argv[1]) : 16.0;
  size_t bytes = (size_t)(gib * (1ull << 30));
  size_t n = bytes / 16;
  u32x4 *src, *dst;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  printf("buffer %.1f GiB each\n", gib);
  auto rep = [&](const char* name, float ms, double moved) { printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, moved / ms / 1e9); };
  for (int bpc : {2, 4, 8, 16, 32}) {
    int grid = 256 * bpc;
    char nm[128];
#define RUN(U, NT) { snprintf(nm, sizeof nm, "copy grid-stride %d blk/CU U=%d %s", bpc, U, NT ? "nt" : "  "); rep(nm, timeit([&] { hipLaunchKernelGGL((copy_gs<U, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n); }), 2.0 * bytes); }
    RUN(4, false) RUN(4, true) RUN(8, true)
  }
  for (int tile_kib : {64, 256, 1024}) {
    int tile_vecs = tile_kib * 1024 / 16;
    unsigned grid = (unsigned)((n + tile_vecs - 1) / tile_vecs);
    char nm[128];
    snprintf(nm, sizeof nm, "copy tile-per-block %d KiB U=4 nt", tile_kib);
    rep(nm, timeit([&] { hipLaunchKernelGGL((copy_tile<4, true>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs); }), 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy tile-per-block %d KiB U=4   ", tile_kib);
    rep(nm, timeit([&] { hipLaunchKernelGGL((copy_tile<4, false>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs); }), 2.0 * bytes);
  }
  for (int tile_kib : {64, 256}) {
    int tile_vecs = tile_kib * 1024 / 16;
    unsigned grid = (unsigned)((n + tile_vecs - 1) / tile_vecs);
    char nm[128];
    snprintf(nm, sizeof nm, "copy tile %d KiB U=4 nt-load only", tile_kib);
    rep(nm, timeit([&] { hipLaunchKernelGGL((copy_tile2<4, true, false>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs); }), 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy tile %d KiB U=4 nt-store only", tile_kib);
    rep(nm, timeit([&] { hipLaunchKernelGGL((copy_tile2<4, false, true>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs); }), 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy tile %d KiB U=8 nt-load only", tile_kib);
    rep(nm, timeit([&] { hipLaunchKernelGGL((copy_tile2<8, true, false>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs); }), 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy tile %d KiB U=2 nt both", tile_kib);
    rep(nm, timeit([&] { hipLaunchKernelGGL((copy_tile2<2, true, true>), dim3(grid), dim3(256), 0, 0, src, dst, n, tile_vecs); }), 2.0 * bytes);
  }
  rep("hipMemcpyDtoD", timeit([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }), 2.0 * bytes);
  for (int bpc : {8, 16}) {
    int grid = 256 * bpc; char nm[128];
    snprintf(nm, sizeof nm, "read-only grid-stride %d blk/CU U=8 nt", bpc);
    rep(nm, timeit([&] { hipLaunchKernelGGL((read_gs<8, true>), dim3(grid), dim3(256), 0, 0, src, dst, n); }), 1.0 * bytes);
    snprintf(nm, sizeof nm, "read-only grid-stride %d blk/CU U=8   ", bpc);
    rep(nm, timeit([&] { hipLaunchKernelGGL((read_gs<8, false>), dim3(grid), dim3(256), 0, 0, src, dst, n); }), 1.0 * bytes);
    snprintf(nm, sizeof nm, "write-only grid-stride %d blk/CU nt", bpc);
    rep(nm, timeit([&] { hipLaunchKernelGGL((write_gs<true>), dim3(grid), dim3(256), 0, 0, dst, n); }), 1.0 * bytes);
    snprintf(nm, sizeof nm, "write-only grid-stride %d blk/CU   ", bpc);
    rep(nm, timeit([&] { hipLaunchKernelGGL((write_gs<false>), dim3(grid), dim3(256), 0, 0, dst, n); }), 1.0 * bytes);
  }
  return 0;
}
