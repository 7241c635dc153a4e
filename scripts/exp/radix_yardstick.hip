// Developer yardstick (not product code): what rocPRIM's device radix sort takes for the bucketing problem of
// rua_index_buckets — M (key, row) pairs, keys < S — so that the hand-written sort has a number to stand against.
//   hipcc --offload-arch=gfx950 -O3 -o radix_yardstick radix_yardstick.hip && ./radix_yardstick [M] [S]
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill(uint32_t* k, uint32_t* v, uint64_t* w, int64_t* i64, long long M, unsigned S, int row_bits) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  unsigned x = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  unsigned key = x % S;
  k[i] = key; v[i] = (uint32_t)i; w[i] = ((uint64_t)key << row_bits) | (uint64_t)i; i64[i] = key;
}

int main(int argc, char** argv) {
  long long M = argc > 1 ? atoll(argv[1]) : 17046960;
  unsigned S = argc > 2 ? (unsigned)atoi(argv[2]) : 65536;
  int row_bits = 1; while ((1ll << row_bits) < M) ++row_bits;
  int key_bits = 1; while ((1u << key_bits) < S) ++key_bits;
  uint32_t *k, *v, *k2, *v2; uint64_t *w, *w2; int64_t* i64;
  CK(hipMalloc(&k, M * 4)); CK(hipMalloc(&v, M * 4)); CK(hipMalloc(&k2, M * 4)); CK(hipMalloc(&v2, M * 4));
  CK(hipMalloc(&w, M * 8)); CK(hipMalloc(&w2, M * 8)); CK(hipMalloc(&i64, M * 8));
  fill<<<(unsigned)((M + 255) / 256), 256>>>(k, v, w, i64, M, S, row_bits);
  CK(hipDeviceSynchronize());
  size_t tmp_bytes = 0, tmp2 = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k, k2, v, v2, (size_t)M, 0, key_bits));
  CK(rocprim::radix_sort_keys(nullptr, tmp2, w, w2, (size_t)M, row_bits, row_bits + key_bits));
  void* tmp; CK(hipMalloc(&tmp, tmp_bytes > tmp2 ? tmp_bytes : tmp2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 8; ++it) {
      CK(hipEventRecord(e0));
      if (mode == 0) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k, k2, v, v2, (size_t)M, 0, key_bits));
      else CK(rocprim::radix_sort_keys(tmp, tmp2, w, w2, (size_t)M, row_bits, row_bits + key_bits));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%s: M=%lld S=%u key_bits=%d  best %.1f us\n", mode == 0 ? "rocprim pairs (u32 key, u32 row)" : "rocprim keys (packed u64 words, bits [row_bits, +key_bits))", M, S, key_bits, best * 1e3f);
  }
  return 0;
}
