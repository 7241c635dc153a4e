// Developer experiment (round 2, last day): do the "classes" of DESIGN.md 4.1a belong to PHYSICAL chunks, and does a
// buffer striped over chunks of different classes move at the fast rate whatever its partner is?
//   1. one hipMalloc'ed source S; N physical chunks (hipMemCreate), each mapped alone: copy S -> chunk, time it;
//   2. destinations assembled (hipMemMap) from chunks of ONE class, and striped over all classes at 2 MiB pieces;
//   3. the same 16 KiB-tile copy kernel as copy_probe2 (span per XCD) between S and each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_tiles(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n_vec,
                                                  size_t tiles_per_xcd) {
  const size_t tile = (size_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
  const size_t base = tile * 1024;                    // 16 KiB tiles = 1024 vectors
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const size_t j = base + threadIdx.x + u * 256; if (j < n_vec) v[u] = __builtin_nontemporal_load(src + j); }
#pragma unroll
  for (int u = 0; u < 4; ++u) { const size_t j = base + threadIdx.x + u * 256; if (j < n_vec) __builtin_nontemporal_store(v[u], dst + j); }
}

static float copy_ms(const void* src, void* dst, size_t bytes, int reps = 5) {
  const size_t n_vec = bytes / 16, tiles = (n_vec + 1023) / 1024, per_xcd = (tiles + 7) / 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipEventRecord(e0));
    copy_tiles<<<(unsigned)(per_xcd * 8), 256>>>((const u32x4*)src, (u32x4*)dst, n_vec, per_xcd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const size_t GiB = 1ull << 30;
  const size_t chunk = 2 * GiB;
  const int n_chunks = argc > 1 ? atoi(argv[1]) : 40;         // 80 GiB of physical chunks
  const size_t big = 16 * GiB;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu KiB\n", gran >> 10);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;

  char* S; CK(hipMalloc(&S, big)); CK(hipMemset(S, 1, big));
  char* S2; CK(hipMalloc(&S2, big)); CK(hipMemset(S2, 2, big));   // a second plain source, probably another class

  std::vector<hipMemGenericAllocationHandle_t> h(n_chunks);
  for (int i = 0; i < n_chunks; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));

  // 1. every chunk alone, against the first 2 GiB of S and of S2
  std::vector<float> t1(n_chunks), t2(n_chunks);
  void* va; CK(hipMemAddressReserve(&va, chunk, 0, nullptr, 0));
  for (int i = 0; i < n_chunks; ++i) {
    CK(hipMemMap(va, chunk, 0, h[i], 0)); CK(hipMemSetAccess(va, chunk, &acc, 1));
    t1[i] = copy_ms(S, va, chunk, 3);
    t2[i] = copy_ms(S2, va, chunk, 3);
    CK(hipMemUnmap(va, chunk));
  }
  printf("chunk  S->chunk ms (TB/s)   S2->chunk ms\n");
  for (int i = 0; i < n_chunks; ++i) printf("%4d   %.4f (%.2f)   %.4f (%.2f)\n", i, t1[i], 2.0 * chunk / t1[i] / 1e9, t2[i], 2.0 * chunk / t2[i] / 1e9);

  // 2. order the chunks by their time against S: the slow ones share S's class
  std::vector<int> order(n_chunks);
  for (int i = 0; i < n_chunks; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return t1[a] < t1[b]; });
  const int per = (int)(big / chunk);                          // 8 chunks make a 16 GiB buffer
  auto assemble = [&](const std::vector<int>& ids, size_t piece) {
    void* p; CK(hipMemAddressReserve(&p, big, 0, nullptr, 0));
    // consecutive `piece`-byte pieces of the buffer come from ids[0], ids[1], ... round-robin
    const size_t pieces_per_chunk = chunk / piece;
    size_t off = 0;
    std::vector<size_t> used(ids.size(), 0);
    for (size_t k = 0; off < big; ++k, off += piece) {
      const int w = (int)(k % ids.size());
      CK(hipMemMap((char*)p + off, piece, used[w] * piece, h[ids[w]], 0));
      ++used[w];
      if (used[w] > pieces_per_chunk) { printf("chunk overflow\n"); exit(1); }
    }
    CK(hipMemSetAccess(p, big, &acc, 1));
    return p;
  };
  std::vector<int> fast(order.begin(), order.begin() + per), slow(order.end() - per, order.end());
  std::vector<int> mixed;
  for (int i = 0; i < per / 2; ++i) { mixed.push_back(order[per + i]); mixed.push_back(order[n_chunks - 1 - per - i]); }
  void* d_fast = assemble(fast, chunk);
  void* d_slow = assemble(slow, chunk);
  printf("16 GiB copies (32 GiB of traffic each):\n");
  printf("  S  -> 8 fastest chunks, whole chunks in a row   %.3f ms\n", copy_ms(S, d_fast, big));
  printf("  S  -> 8 slowest chunks, whole chunks in a row   %.3f ms\n", copy_ms(S, d_slow, big));
  printf("  S2 -> 8 fastest-for-S chunks                    %.3f ms\n", copy_ms(S2, d_fast, big));
  printf("  S2 -> 8 slowest-for-S chunks                    %.3f ms\n", copy_ms(S2, d_slow, big));
  {
    // hipMemMap takes whole handles only (a sub-range of a handle is "invalid argument"), so the stripe is a chunk
    void* d_mix = assemble(mixed, chunk);
    printf("  S  -> 4 fast + 4 slow chunks, alternating by 2 GiB   %.3f ms\n", copy_ms(S, d_mix, big));
    printf("  S2 -> the same                                       %.3f ms\n", copy_ms(S2, d_mix, big));
    printf("  4 fast + 4 slow alternating -> 8 fastest             %.3f ms\n", copy_ms(d_mix, d_fast, big));
    printf("  4 fast + 4 slow alternating -> 8 slowest             %.3f ms\n", copy_ms(d_mix, d_slow, big));
    printf("  8 fastest -> 8 slowest                               %.3f ms\n", copy_ms(d_fast, d_slow, big));
    printf("  8 slowest -> 8 fastest                               %.3f ms\n", copy_ms(d_slow, d_fast, big));
  }
  printf("  S  -> S2 (two hipMalloc buffers)                  %.3f ms\n", copy_ms(S, S2, big));
  return 0;
}
