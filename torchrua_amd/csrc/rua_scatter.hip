// rua_scatter.hip — bucket an arbitrary destination index so scatter_* (reference
// reduce.py:6-31: torch.index_reduce / index_add) becomes a segmented reduce with a row
// indirection (rua_segment_reduce(perm=...)): no float atomics on the data path.
//
// The bucketing is a STABLE least-significant-digit radix sort of (destination, source row) on the
// destination only (8-bit digits, ceil(log2(S) / 8) passes), so every destination's source rows come out in
// ascending order whatever the fan-in: the summation order is a fixed function of the inputs and results are
// bitwise reproducible.  gfx950, wave64; one wave per 2 048-element block so that "order inside the block"
// is simply program order.
#include "rua_dev.h"

namespace rua {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_CHUNKS = 32;                       // 64-element chunks per block
constexpr int SORT_BLOCK = RUA_WAVE * SORT_CHUNKS;    // 2 048 elements per (one-wave) workgroup

__global__ __launch_bounds__(RUA_BLOCK) void bucket_count_kernel(const int64_t* __restrict__ index, int64_t M,
                                                                 int64_t S, unsigned long long* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (i >= M) return;
  const int64_t s = index[i];
  if (s >= 0 && s < S) atomicAdd(&counts[s], 1ull);
}

// out-of-range destinations sort to the end (key S) and are never referenced by counts/off
__device__ __forceinline__ int64_t clamp_key(int64_t k, int64_t S) { return (k >= 0 && k < S) ? k : S; }

// pass, step 1: per-block digit histogram, stored digit-major so that ONE exclusive scan over the whole
// [RADIX][n_blocks] table yields the global base of every (digit, block)
__global__ __launch_bounds__(RUA_WAVE) void radix_hist_kernel(const int64_t* __restrict__ keys, int64_t M, int64_t S,
                                                              int shift, int clamp, int64_t n_blocks,
                                                              int64_t* __restrict__ table) {
  __shared__ unsigned int h[RADIX];
  const int lane = threadIdx.x;
  for (int d = lane; d < RADIX; d += RUA_WAVE) h[d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * SORT_BLOCK;
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t i = base + (int64_t)c * RUA_WAVE + lane;
    if (i < M) {
      const int64_t k = clamp ? clamp_key(keys[i], S) : keys[i];
      atomicAdd(&h[(k >> shift) & (RADIX - 1)], 1u);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < RADIX; d += RUA_WAVE) table[(int64_t)d * n_blocks + blockIdx.x] = h[d];
}

// pass, step 2: stable scatter.  The block walks its elements 64 at a time in order; inside a chunk the
// lanes that share a digit are found with 8 ballots, a lane's position among them is a popcount of the
// lower lanes, and a running per-digit counter in LDS carries the order across chunks.
__global__ __launch_bounds__(RUA_WAVE) void radix_scatter_kernel(const int64_t* __restrict__ keys_in,
                                                                 const int64_t* __restrict__ vals_in, int64_t M,
                                                                 int64_t S, int shift, int clamp, int64_t n_blocks,
                                                                 const int64_t* __restrict__ table,
                                                                 int64_t* __restrict__ keys_out,
                                                                 int64_t* __restrict__ vals_out) {
  __shared__ int64_t cursor[RADIX];
  const int lane = threadIdx.x;
  for (int d = lane; d < RADIX; d += RUA_WAVE) cursor[d] = table[(int64_t)d * n_blocks + blockIdx.x];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * SORT_BLOCK;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t i = base + (int64_t)c * RUA_WAVE + lane;
    const bool live = i < M;
    int64_t k = 0, v = 0;
    if (live) {
      k = clamp ? clamp_key(keys_in[i], S) : keys_in[i];
      v = vals_in ? vals_in[i] : i;                  // first pass: the value is the source row itself
    }
    const int digit = (int)((k >> shift) & (RADIX - 1));
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const unsigned long long m = __ballot(live && ((digit >> b) & 1));
      peers &= ((digit >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const int count = __popcll(peers);
    int64_t dst = 0;
    if (live) dst = cursor[digit] + rank;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live && rank == 0) cursor[digit] += count;   // one lane per digit advances the running counter
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live) {
      if (keys_out) keys_out[dst] = k;
      vals_out[dst] = v;
    }
  }
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RUA_BLOCK - 1) / RUA_BLOCK); }

}  // namespace rua

using namespace rua;

extern "C" {

// ws layout (int64 words): [scan scratch for max(S, RADIX*nb)] [table RADIX*nb] [keys A: M] [vals A: M] [keys B: M] [vals B: M]
int64_t rua_bucket_ws_elems(int64_t M, int64_t S) {
  if (M < 0 || S < 0) return 0;
  const int64_t nb = (M + SORT_BLOCK - 1) / SORT_BLOCK;
  const int64_t tab = (int64_t)RADIX * (nb > 0 ? nb : 1);
  const int64_t scan_n = tab > S ? tab : S;
  return rua_scan_ws_elems(scan_n) + tab + 4 * M + 8;
}

int rua_index_buckets(const int64_t* index, int64_t M, int64_t S, int64_t* counts, int64_t* off, int64_t* perm,
                      int64_t* ws, void* stream) {
  if (M < 0 || S < 0) return RUA_EINVAL;
  if (S == 0) return 0;
  if (!counts || !off || !ws || (M > 0 && (!index || !perm))) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = (M + SORT_BLOCK - 1) / SORT_BLOCK;
  if (nb > 0x7fffffffLL) return RUA_ERANGE;
  const int64_t tab = (int64_t)RADIX * (nb > 0 ? nb : 1);
  int64_t* scan_ws = ws;
  int64_t* table = ws + rua_scan_ws_elems(tab > S ? tab : S);
  int64_t* bufs[2][2] = {{table + tab, table + tab + M}, {table + tab + 2 * M, table + tab + 3 * M}};

  // counts / offsets of the destinations (order-independent integer atomics)
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)S, s);
  if (e != hipSuccess) return (int)e;
  if (M > 0)
    hipLaunchKernelGGL(bucket_count_kernel, dim3(grid_for(M)), dim3(RUA_BLOCK), 0, s, index, M, S,
                       (unsigned long long*)counts);
  int r = rua_exclusive_scan_i64(counts, off, nullptr, S, scan_ws, stream);
  if (r != 0 || M == 0) return r;

  // stable LSD radix sort of (destination, row) by destination; the key S (out of range) needs bits(S)
  int bits = 1;
  while (bits < 63 && (S >> bits) != 0) ++bits;
  const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;
  const int64_t* kin = index;
  const int64_t* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * RADIX_BITS;
    const bool last = p == passes - 1;
    int64_t* kout = last ? nullptr : bufs[p & 1][0];
    int64_t* vout = last ? perm : bufs[p & 1][1];
    const int clamp = p == 0 ? 1 : 0;      // later passes read keys that were clamped by the first
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nb), dim3(RUA_WAVE), 0, s, kin, M, S, shift, clamp, nb, table);
    r = rua_exclusive_scan_i64(table, table, nullptr, tab, scan_ws, stream);
    if (r != 0) return r;
    hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nb), dim3(RUA_WAVE), 0, s, kin, vin, M, S, shift, clamp, nb,
                       (const int64_t*)table, kout, vout);
    kin = kout;
    vin = vout;
  }
  return (int)hipGetLastError();
}

}  // extern "C"
