// rua_scatter.hip — bucket an arbitrary destination index so scatter_* (reference
// reduce.py:6-31: torch.index_reduce / index_add) becomes a segmented reduce with a row
// indirection (rua_segment_reduce(perm=...)): no float atomics on the data path.
//
// The bucketing is a STABLE least-significant-digit radix sort of packed (destination << row_bits | source row)
// words on the destination only, so every destination's source rows come out in ascending order whatever the
// fan-in: the summation order is a fixed function of the inputs and results are bitwise reproducible.
// Digits are up to 9 bits wide (ceil(bits(S) / passes)), so 65 536 destinations (+ the "ignored" key S for
// out-of-range indices) take two passes.  counts / offsets come from binary searches in the sorted words —
// no histogram atomics in global memory.
//
// gfx950, wave64.  A workgroup of 8 waves owns 8 192 consecutive words (wave w the w-th run of 1 024, 64 at a time,
// all 16 loads of a lane in flight at once).  Order inside the block = (wave, chunk, lane) = the input order.
//   * a lane's position among the earlier equal digits of its wave: `width` ballots give the lanes of the chunk
//     that share the digit, a per-(wave, digit) counter in LDS carries the count across the wave's chunks;
//   * one pass over the (wave, digit) counters turns them into the wave's start inside the digit, a workgroup scan
//     of the digit totals into the digit's start inside the block;
//   * the words are first put in block-sorted order in LDS and written from there: the ~16 words of a digit
//     leave as one 128-byte run instead of sixteen 8-byte stores into sixteen different lines.
// 17 M entries into 65 536 destinations: 0.89 ms with the first version of this file (one wave per 2 048 words, a
// wave barrier per 64 of them, 8-byte stores) -> 0.42 ms (hist 45 us + scan 19 us + scatter 113 / 146 us per pass,
// bounds 22 us).  Writing the runs to consecutive addresses instead would only save another 25 us per pass
// (measured), so what is left is the ranking itself and the size of the launch.
#include <stdlib.h>
#include "rua_dev.h"

namespace rua {

constexpr int RADIX_BITS_MAX = 9;
constexpr int RADIX_MAX = 1 << RADIX_BITS_MAX;
constexpr int SORT_WAVES = 8;
constexpr int SORT_THREADS = SORT_WAVES * RUA_WAVE;   // 512 >= RADIX_MAX: one thread per digit in the scans
constexpr int SORT_ITEMS = 16;                       // 64-word chunks per wave
constexpr int SORT_BLOCK = SORT_THREADS * SORT_ITEMS; // 8 192 words per workgroup
static_assert(SORT_THREADS >= RADIX_MAX, "one thread per digit");
static_assert(SORT_BLOCK <= 65536 && RUA_WAVE * SORT_ITEMS <= 65535, "16-bit positions inside a block");

// pass 0 reads the raw index (out-of-range -> key S, sorted to the end and never referenced);
// later passes read the packed words written by the previous pass
__device__ __forceinline__ int64_t load_word(const int64_t* __restrict__ in, int64_t i, int first, int64_t S,
                                             int row_bits) {
  const int64_t v = in[i];
  if (!first) return v;
  const int64_t k = (v >= 0 && v < S) ? v : S;
  return (k << row_bits) | i;
}

// workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); every XCD takes ONE contiguous span of blocks, so
// that the eight (digit, block) table entries of a 64-byte sector, and the neighbouring runs of a digit in the
// output, meet in one L2 instead of eight
__device__ __forceinline__ int64_t block_of_workgroup(int64_t per_xcd) {
  return (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
}

// the 16 words of this lane (-1 where the input has ended); wave `wave` of the block reads words
// [wave * 1024, (wave + 1) * 1024) of the block, chunk u = the u-th 64 of them
__device__ __forceinline__ void load_block_words(const int64_t* __restrict__ in, int64_t M, int64_t S, int first,
                                                 int row_bits, int64_t block, int wave, int lane,
                                                 int64_t (&w)[SORT_ITEMS]) {
  const int64_t base = block * SORT_BLOCK + (int64_t)wave * (RUA_WAVE * SORT_ITEMS) + lane;
#pragma unroll
  for (int u = 0; u < SORT_ITEMS; ++u) {
    const int64_t i = base + (int64_t)u * RUA_WAVE;
    w[u] = i < M ? load_word(in, i, first, S, row_bits) : -1;
  }
}

// step 1 of a pass: per-block digit histogram, stored digit-major so that ONE exclusive scan over the whole
// [radix][n_blocks] table yields the global base of every (digit, block)
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const int64_t* __restrict__ in, int64_t M, int64_t S,
                                                                  int first, int row_bits, int shift, int width,
                                                                  int64_t n_blocks, int64_t per_xcd,
                                                                  int64_t* __restrict__ table) {
  __shared__ unsigned int h[RADIX_MAX];
  const int tid = threadIdx.x, lane = tid & (RUA_WAVE - 1), wave = tid / RUA_WAVE;
  const int radix = 1 << width;
  const int64_t block = block_of_workgroup(per_xcd);
  if (block >= n_blocks) return;
  if (tid < radix) h[tid] = 0;
  __syncthreads();
  int64_t w[SORT_ITEMS];
  load_block_words(in, M, S, first, row_bits, block, wave, lane, w);
#pragma unroll
  for (int u = 0; u < SORT_ITEMS; ++u) {
    const int digit = (int)(((w[u] >> row_bits) >> shift) & (radix - 1));
    // 64 words of one digit (an index that is sorted, or the high digit of one that is clustered): one add of 64,
    // not 64 adds to one LDS address one after the other
    const int d0 = __builtin_amdgcn_readfirstlane(digit);
    if (__all(w[u] >= 0 && digit == d0)) {
      if (lane == 0) atomicAdd(&h[d0], (unsigned int)RUA_WAVE);
    } else if (w[u] >= 0) {
      atomicAdd(&h[digit], 1u);
    }
  }
  __syncthreads();
  if (tid < radix) table[(int64_t)tid * n_blocks + block] = h[tid];
}

// step 2: stable scatter (see the head of the file).  LAST: also write the source rows (the low bits of the words)
// to `perm` — the bucketed row order the reducers read.
template <bool LAST>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const int64_t* __restrict__ in, int64_t M, int64_t S,
                                                                     int first, int row_bits, int shift, int width,
                                                                     int64_t n_blocks, int64_t per_xcd,
                                                                     const int64_t* __restrict__ table,
                                                                     int64_t* __restrict__ out, int64_t* __restrict__ perm) {
  __shared__ int64_t stage[SORT_BLOCK];                    // the block's words in block-sorted order
  __shared__ unsigned short wcnt[SORT_WAVES][RADIX_MAX];   // (wave, digit): count, then start inside the digit
  __shared__ unsigned short dstart[RADIX_MAX];             // digit: start inside the block (16 bits: two
                                                           // workgroups' worth of LDS fit a CU)
  __shared__ int64_t delta[RADIX_MAX];                     // digit: global base - start inside the block
  __shared__ unsigned int wave_total[SORT_WAVES];
  const int tid = threadIdx.x, lane = tid & (RUA_WAVE - 1), wave = tid / RUA_WAVE;
  const int radix = 1 << width;
  const int64_t block = block_of_workgroup(per_xcd);
  if (block >= n_blocks) return;
  for (int i = tid; i < SORT_WAVES * RADIX_MAX / 2; i += SORT_THREADS) reinterpret_cast<unsigned int*>(&wcnt[0][0])[i] = 0;
  __syncthreads();

  int64_t w[SORT_ITEMS];
  load_block_words(in, M, S, first, row_bits, block, wave, lane, w);

  // (a) position of every word among the earlier words of the same digit IN ITS WAVE
  unsigned short before[SORT_ITEMS];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned short* __restrict__ mine = wcnt[wave];
#pragma unroll
  for (int u = 0; u < SORT_ITEMS; ++u) {
    const bool live = w[u] >= 0;
    const int digit = (int)(((w[u] >> row_bits) >> shift) & (radix - 1));
    unsigned long long peers = __ballot(live);
    for (int b = 0; b < width; ++b) {
      const unsigned long long m = __ballot(live && ((digit >> b) & 1));
      peers &= ((digit >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const unsigned int seen = live ? mine[digit] : 0u;       // every lane of the digit reads the counter ...
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live && rank == 0) mine[digit] = (unsigned short)(seen + (unsigned int)__popcll(peers));   // ... then its first lane moves it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    before[u] = (unsigned short)(seen + rank);
  }
  __syncthreads();

  // (b) one thread per digit: the waves' counts become the waves' starts inside the digit; the digit totals are
  //     scanned over the workgroup into the digits' starts inside the block
  unsigned int total = 0;
  if (tid < radix) {
#pragma unroll
    for (int v = 0; v < SORT_WAVES; ++v) {
      const unsigned int c = wcnt[v][tid];
      wcnt[v][tid] = (unsigned short)total;
      total += c;
    }
  }
  unsigned int incl = total;
#pragma unroll
  for (int s = 1; s < RUA_WAVE; s <<= 1) {
    const unsigned int up = __shfl_up(incl, s, RUA_WAVE);
    if (lane >= s) incl += up;
  }
  if (lane == RUA_WAVE - 1) wave_total[wave] = incl;
  __syncthreads();
  unsigned int carry = 0;
#pragma unroll
  for (int v = 0; v < SORT_WAVES; ++v) carry += v < wave ? wave_total[v] : 0u;
  if (tid < radix) {
    const unsigned int start = carry + incl - total;
    dstart[tid] = (unsigned short)start;
    delta[tid] = table[(int64_t)tid * n_blocks + block] - (int64_t)start;
  }
  __syncthreads();

  // (c) block-sorted order in LDS
#pragma unroll
  for (int u = 0; u < SORT_ITEMS; ++u) {
    if (w[u] < 0) continue;
    const int digit = (int)(((w[u] >> row_bits) >> shift) & (radix - 1));
    stage[dstart[digit] + wcnt[wave][digit] + before[u]] = w[u];
  }
  __syncthreads();

  // (d) out: consecutive threads write consecutive words of a digit's run
  const int64_t left = M - block * SORT_BLOCK;
  const int n_here = left < SORT_BLOCK ? (int)left : SORT_BLOCK;
  const int64_t row_mask = ((int64_t)1 << row_bits) - 1;
  for (int j = tid; j < n_here; j += SORT_THREADS) {
    const int64_t word = stage[j];
    const int digit = (int)(((word >> row_bits) >> shift) & (radix - 1));
    const int64_t dst = delta[digit] + j;
    out[dst] = word;
    if (LAST) perm[dst] = word & row_mask;
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

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RUA_BLOCK - 1) / RUA_BLOCK); }
static inline int bits_of(int64_t v) { int b = 1; while (b < 63 && (v >> b) != 0) ++b; return b; }

// rua_bucket_msd.hip: the two-level most-significant-digit-first builder (up to 262 144 destinations, M < 2^31)
int64_t bucket_msd_ws_bytes(int64_t M);
bool bucket_msd_applies(int64_t M, int64_t S);
int bucket_msd(const int64_t* index, int64_t M, int64_t S, int64_t* counts, int64_t* off, int64_t* perm, void* ws,
               hipStream_t s);

}  // namespace rua

using namespace rua;

extern "C" {

// ws layout (int64 words): [scan scratch for the table] [table RADIX_MAX * nb] [words A: M] [words B: M]
int64_t rua_bucket_ws_elems(int64_t M, int64_t S) {
  if (M < 0 || S < 0) return 0;
  const int64_t nb = (M + SORT_BLOCK - 1) / SORT_BLOCK;
  const int64_t tab = (int64_t)RADIX_MAX * (nb > 0 ? nb : 1);
  const int64_t lsd = rua_scan_ws_elems(tab) + tab + 2 * M + 8;
  const int64_t msd = (bucket_msd_ws_bytes(M) + 7) / 8 + 2;          // whichever builder the call takes fits
  return lsd > msd ? lsd : msd;
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
  // 513 .. 262 144 destinations: most significant digit first, the second level local to a bin (rua_bucket_msd.hip)
  if (bucket_msd_applies(M, S)) return bucket_msd(index, M, S, counts, off, perm, (void*)ws, s);
  const int64_t nb = (M + SORT_BLOCK - 1) / SORT_BLOCK;
  if (nb > 0x7ffffff0LL) return RUA_ERANGE;
  const int64_t per_xcd = (nb + 7) / 8;
  const unsigned grid = (unsigned)(per_xcd * 8);
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
    hipLaunchKernelGGL(radix_hist_kernel, dim3(grid), dim3(SORT_THREADS), 0, s, in, M, S, first, row_bits,
                       p * width, width, nb, per_xcd, table);
    int r = rua_exclusive_scan_i64(table, table, nullptr, tab, scan_ws, stream);
    if (r != 0) return r;
    if (p + 1 == passes)
      hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(grid), dim3(SORT_THREADS), 0, s, in, M, S, first,
                         row_bits, p * width, width, nb, per_xcd, (const int64_t*)table, out, perm);
    else
      hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(grid), dim3(SORT_THREADS), 0, s, in, M, S, first,
                         row_bits, p * width, width, nb, per_xcd, (const int64_t*)table, out, (int64_t*)nullptr);
    in = out;
  }
  hipLaunchKernelGGL(bucket_bounds_kernel, dim3(grid_for(S)), dim3(RUA_BLOCK), 0, s, in, M, S, row_bits, counts, off);
  return (int)hipGetLastError();
}

}  // extern "C"
