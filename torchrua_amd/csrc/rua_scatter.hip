// rua_scatter.hip — bucket an arbitrary destination index so scatter_* (reference
// reduce.py:6-31: torch.index_reduce / index_add) becomes a segmented reduce with a row
// indirection (rua_segment_reduce(perm=...)): no float atomics on the data path.
//
// The bucketing is a STABLE least-significant-digit radix sort of packed (destination << row_bits | source row)
// words on the destination only, so every destination's source rows come out in ascending order whatever the
// fan-in: the summation order is a fixed function of the inputs and results are bitwise reproducible.
// Digits are up to 9 bits wide (ceil(bits(S) / passes)), so 65 536 destinations (+ the "ignored" key S for
// out-of-range indices) take two passes.  counts / offsets come from binary searches in the sorted words —
// no histogram atomics.  gfx950, wave64; one wave per 2 048-element block, so that "order inside the block"
// is program order; the block's elements are fetched 8 chunks at a time to keep loads in flight.
#include "rua_dev.h"

namespace rua {

constexpr int RADIX_BITS_MAX = 9;
constexpr int RADIX_MAX = 1 << RADIX_BITS_MAX;
constexpr int SORT_CHUNKS = 32;                       // 64-element chunks per block
constexpr int SORT_BLOCK = RUA_WAVE * SORT_CHUNKS;    // 2 048 elements per (one-wave) workgroup
constexpr int SORT_GROUP = 8;                         // chunks fetched together

// pass 0 reads the raw index (out-of-range -> key S, sorted to the end and never referenced);
// later passes read the packed words written by the previous pass
__device__ __forceinline__ int64_t load_word(const int64_t* __restrict__ in, int64_t i, int first, int64_t S,
                                             int row_bits) {
  const int64_t v = in[i];
  if (!first) return v;
  const int64_t k = (v >= 0 && v < S) ? v : S;
  return (k << row_bits) | i;
}

// step 1 of a pass: per-block digit histogram, stored digit-major so that ONE exclusive scan over the whole
// [radix][n_blocks] table yields the global base of every (digit, block)
__global__ __launch_bounds__(RUA_WAVE) void radix_hist_kernel(const int64_t* __restrict__ in, int64_t M, int64_t S,
                                                              int first, int row_bits, int shift, int width,
                                                              int64_t n_blocks, int64_t* __restrict__ table) {
  __shared__ unsigned int h[RADIX_MAX];
  const int lane = threadIdx.x;
  const int radix = 1 << width;
  for (int d = lane; d < radix; d += RUA_WAVE) h[d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * SORT_BLOCK;
  for (int c0 = 0; c0 < SORT_CHUNKS; c0 += SORT_GROUP) {
    int64_t w[SORT_GROUP];
#pragma unroll
    for (int u = 0; u < SORT_GROUP; ++u) {
      const int64_t i = base + (int64_t)(c0 + u) * RUA_WAVE + lane;
      w[u] = i < M ? load_word(in, i, first, S, row_bits) : -1;
    }
#pragma unroll
    for (int u = 0; u < SORT_GROUP; ++u)
      if (w[u] >= 0) atomicAdd(&h[((w[u] >> row_bits) >> shift) & (radix - 1)], 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < radix; d += RUA_WAVE) table[(int64_t)d * n_blocks + blockIdx.x] = h[d];
}

// step 2: stable scatter.  The block walks its elements 64 at a time in order; inside a chunk the lanes that
// share a digit are found with `width` ballots, a lane's position among them is a popcount of the lower lanes,
// and a running per-digit cursor in LDS carries the order across chunks.
__global__ __launch_bounds__(RUA_WAVE) void radix_scatter_kernel(const int64_t* __restrict__ in, int64_t M, int64_t S,
                                                                 int first, int row_bits, int shift, int width,
                                                                 int64_t n_blocks, const int64_t* __restrict__ table,
                                                                 int64_t* __restrict__ out) {
  __shared__ int64_t cursor[RADIX_MAX];
  const int lane = threadIdx.x;
  const int radix = 1 << width;
  for (int d = lane; d < radix; d += RUA_WAVE) cursor[d] = table[(int64_t)d * n_blocks + blockIdx.x];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * SORT_BLOCK;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int c0 = 0; c0 < SORT_CHUNKS; c0 += SORT_GROUP) {
    int64_t w[SORT_GROUP];
#pragma unroll
    for (int u = 0; u < SORT_GROUP; ++u) {
      const int64_t i = base + (int64_t)(c0 + u) * RUA_WAVE + lane;
      w[u] = i < M ? load_word(in, i, first, S, row_bits) : -1;
    }
#pragma unroll
    for (int u = 0; u < SORT_GROUP; ++u) {
      const bool live = w[u] >= 0;
      const int digit = (int)(((w[u] >> row_bits) >> shift) & (radix - 1));
      unsigned long long peers = __ballot(live);
      for (int b = 0; b < width; ++b) {
        const unsigned long long m = __ballot(live && ((digit >> b) & 1));
        peers &= ((digit >> b) & 1) ? m : ~m;
      }
      const int rank = __popcll(peers & lt_mask);
      const int count = __popcll(peers);
      int64_t dst = 0;
      if (live) dst = cursor[digit] + rank;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (live && rank == 0) cursor[digit] += count;   // one lane per digit advances the running cursor
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (live) out[dst] = w[u];
    }
  }
}

// counts / exclusive offsets of every destination by binary search in the sorted words
__global__ __launch_bounds__(RUA_BLOCK) void bucket_bounds_kernel(const int64_t* __restrict__ words, int64_t M,
                                                                  int64_t S, int row_bits,
                                                                  int64_t* __restrict__ counts,
                                                                  int64_t* __restrict__ off) {
  const int64_t s = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (s >= S) return;
  int64_t bound[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {   // first position whose key >= s + k
    int64_t lo = 0, hi = M;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((words[mid] >> row_bits) < s + k) lo = mid + 1; else hi = mid;
    }
    bound[k] = lo;
  }
  off[s] = bound[0];
  counts[s] = bound[1] - bound[0];
}

__global__ __launch_bounds__(RUA_BLOCK) void bucket_rows_kernel(const int64_t* __restrict__ words, int64_t M,
                                                                int row_bits, int64_t* __restrict__ perm) {
  const int64_t j = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (j < M) perm[j] = words[j] & ((1ll << row_bits) - 1);
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RUA_BLOCK - 1) / RUA_BLOCK); }
static inline int bits_of(int64_t v) { int b = 1; while (b < 63 && (v >> b) != 0) ++b; return b; }

}  // namespace rua

using namespace rua;

extern "C" {

// ws layout (int64 words): [scan scratch for the table] [table RADIX_MAX * nb] [words A: M] [words B: M]
int64_t rua_bucket_ws_elems(int64_t M, int64_t S) {
  if (M < 0 || S < 0) return 0;
  const int64_t nb = (M + SORT_BLOCK - 1) / SORT_BLOCK;
  const int64_t tab = (int64_t)RADIX_MAX * (nb > 0 ? nb : 1);
  return rua_scan_ws_elems(tab) + tab + 2 * M + 8;
}

int rua_index_buckets(const int64_t* index, int64_t M, int64_t S, int64_t* counts, int64_t* off, int64_t* perm,
                      int64_t* ws, void* stream) {
  if (M < 0 || S < 0) return RUA_EINVAL;
  if (S == 0) return 0;
  if (!counts || !off || !ws || (M > 0 && (!index || !perm))) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)S, s);
    if (e == hipSuccess) e = hipMemsetAsync(off, 0, sizeof(int64_t) * (size_t)S, s);
    return (int)e;
  }
  const int64_t nb = (M + SORT_BLOCK - 1) / SORT_BLOCK;
  if (nb > 0x7fffffffLL) return RUA_ERANGE;
  const int row_bits = bits_of(M - 1 > 0 ? M - 1 : 1);
  const int key_bits = bits_of(S);                      // keys 0..S (S = "ignored")
  if (row_bits + key_bits > 62) return RUA_ERANGE;      // the packed word must stay a non-negative int64
  const int passes = (key_bits + RADIX_BITS_MAX - 1) / RADIX_BITS_MAX;
  const int width = (key_bits + passes - 1) / passes;   // <= 9 bits per pass
  const int64_t tab = ((int64_t)1 << width) * nb;
  int64_t* scan_ws = ws;
  int64_t* table = ws + rua_scan_ws_elems((int64_t)RADIX_MAX * nb);
  int64_t* buf[2] = {table + (int64_t)RADIX_MAX * nb, table + (int64_t)RADIX_MAX * nb + M};

  const int64_t* in = index;
  for (int p = 0; p < passes; ++p) {
    int64_t* out = buf[p & 1];
    const int first = p == 0 ? 1 : 0;
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nb), dim3(RUA_WAVE), 0, s, in, M, S, first, row_bits, p * width,
                       width, nb, table);
    int r = rua_exclusive_scan_i64(table, table, nullptr, tab, scan_ws, stream);
    if (r != 0) return r;
    hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nb), dim3(RUA_WAVE), 0, s, in, M, S, first, row_bits,
                       p * width, width, nb, (const int64_t*)table, out);
    in = out;
  }
  hipLaunchKernelGGL(bucket_bounds_kernel, dim3(grid_for(S)), dim3(RUA_BLOCK), 0, s, in, M, S, row_bits, counts, off);
  hipLaunchKernelGGL(bucket_rows_kernel, dim3(grid_for(M)), dim3(RUA_BLOCK), 0, s, in, M, row_bits, perm);
  return (int)hipGetLastError();
}

}  // extern "C"
