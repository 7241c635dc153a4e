// rua_bucket_msd.hip — the bucket builder of scatter_* (reference reduce.py:6-31) for up to 262 144 destinations:
// a stable MOST-significant-digit-first radix sort in two levels, whose second level is LOCAL to a bin.
//
// rua_scatter.hip sorts (destination << row_bits | row) words least-significant digit first: every pass reads and
// writes 8 bytes per entry through HBM, pass 2 scatters each block's words over the whole output again, and the
// bucket bounds are one more kernel of binary searches (0.42 ms for 17 M entries into 65 536 destinations; rocPRIM's
// radix sort takes 0.27-0.37 ms for the same words: scripts/exp/radix_yardstick.hip).  Going from the top digit down
// changes what each level has to carry:
//   level 1  bins the entries by the HIGH digit of the destination (hist -> row scans -> stable scatter).  What it
//            writes is no longer the whole word: the high digit is the bin, so a word is (low digit << row_bits | row)
//            — 32 bits for up to 2^24..2^25 rows with 65 536 destinations — HALF the bytes written, and read twice by
//            level 2.  Entries whose destination is out of range are dropped here.
//   level 2  works bin by bin: a bin is cut into segments of 4 096 words, a workgroup counts its segment's low digits,
//            ONE workgroup per bin scans the bin's (digit, segment) table — which yields, on the side, counts[] and
//            off[] of the bin's destinations: no bounds kernel — and the segments scatter the rows to their final
//            places.  All traffic of a bin stays inside the bin's own range (266 KiB on average): stores that the L2
//            merges into whole lines, where the last pass of the LSD order wrote 128-byte runs across 136 MB.
// Six launches instead of eleven; bytes through HBM per entry: 8 + 8 (index, read by hist and scatter) + 4 written +
// 4 + 4 read + 8 written (perm) = 36 instead of 56.  Stable at both levels, so every destination's rows come out in
// ascending order: the summation order of scatter_* stays a fixed function of the inputs (bitwise reproducible).
//
// gfx950, wave64.  A workgroup of 8 waves owns 4 096 consecutive entries (8 per thread; 16 measured slower); ranking inside the block is the scheme of
// rua_scatter.hip (ballots find the lanes of a chunk that share a digit, (wave, digit) counters in LDS carry the count
// across the wave's chunks, the block's words are put in sorted order in LDS and leave as runs).
#include "rua_dev.h"

namespace rua {

constexpr int MSD_RADIX_BITS = 9;
constexpr int MSD_RADIX = 1 << MSD_RADIX_BITS;
constexpr int MSD_WAVES = 8;
constexpr int MSD_THREADS = MSD_WAVES * RUA_WAVE;        // 512 >= MSD_RADIX: one thread per digit / per bin
#ifndef RUA_MSD_ITEMS        // developer knob for A/B builds
#define RUA_MSD_ITEMS 8
#endif
constexpr int MSD_ITEMS = RUA_MSD_ITEMS;
constexpr int MSD_BLOCK = MSD_THREADS * MSD_ITEMS;       // 4 096 entries per workgroup / per segment
static_assert(MSD_THREADS >= MSD_RADIX, "one thread per digit");

struct MsdGeom {
  int64_t M, S;
  int lo_bits, row_bits, n_bins;        // destination = (bin << lo_bits) | low digit; n_bins <= 512
  int64_t n_blocks;                     // level-1 blocks of MSD_BLOCK index entries
  int64_t grid2;                        // level-2 workgroups launched: an upper bound of the number of segments
};

__device__ __forceinline__ int64_t msd_block(int64_t per_xcd) {        // one contiguous span of blocks per XCD
  return per_xcd > 0 ? (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
}

// exclusive scan of one value per thread over the workgroup (MSD_THREADS threads); `total` = the sum
__device__ __forceinline__ unsigned int msd_block_scan(unsigned int v, unsigned int* __restrict__ wave_tot, unsigned int& total) {
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  unsigned int incl = v;
#pragma unroll
  for (int s = 1; s < RUA_WAVE; s <<= 1) {
    const unsigned int up = __shfl_up(incl, s, RUA_WAVE);
    if (lane >= s) incl += up;
  }
  __syncthreads();                                    // (wave_tot may still be read from the previous call)
  if (lane == RUA_WAVE - 1) wave_tot[wave] = incl;
  __syncthreads();
  unsigned int carry = 0, all = 0;
#pragma unroll
  for (int w = 0; w < MSD_WAVES; ++w) {
    const unsigned int t = wave_tot[w];
    carry += w < wave ? t : 0u;
    all += t;
  }
  total = all;
  return carry + incl - v;
}

// The bins as every level-2 workgroup sees them, rebuilt from the level-1 row totals (<= 512 numbers): first word and
// first segment of every bin.  s_base[b] / s_seg[b] for b <= n_bins (the last entry = the totals).
__device__ __forceinline__ void msd_bins(const unsigned int* __restrict__ rowtot, int n_bins, unsigned int* __restrict__ s_base,
                                         unsigned int* __restrict__ s_seg, unsigned int* __restrict__ wave_tot) {
  const int tid = threadIdx.x;
  const unsigned int cnt = tid < n_bins ? rowtot[tid] : 0u;
  unsigned int tot_w, tot_s;
  const unsigned int base = msd_block_scan(cnt, wave_tot, tot_w);
  const unsigned int seg = msd_block_scan((cnt + MSD_BLOCK - 1) / MSD_BLOCK, wave_tot, tot_s);
  if (tid < n_bins) { s_base[tid] = base; s_seg[tid] = seg; }
  if (tid == 0) { s_base[n_bins] = tot_w; s_seg[n_bins] = tot_s; }
  __syncthreads();
}

// level-2 workgroup w -> (bin, segment of the bin, first word, number of words); false past the last segment
__device__ __forceinline__ bool msd_segment(const unsigned int* __restrict__ s_base, const unsigned int* __restrict__ s_seg,
                                            int n_bins, int64_t w, int& bin, unsigned int& j, unsigned int& nseg,
                                            int64_t& first, int& n_here) {
  if (w >= (int64_t)s_seg[n_bins]) return false;
  int lo = 0, hi = n_bins;                              // largest bin with s_seg[bin] <= w that owns a segment
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)s_seg[mid] <= w) lo = mid; else hi = mid;
  }
  bin = lo;                                             // (empty bins share their successor's s_seg: the search lands
  j = (unsigned int)(w - s_seg[bin]);                   //  on the LAST bin with s_seg <= w, which is the non-empty one)
  nseg = s_seg[bin + 1] - s_seg[bin];
  const unsigned int cnt = s_base[bin + 1] - s_base[bin];
  first = (int64_t)s_base[bin] + (int64_t)j * MSD_BLOCK;
  const unsigned int left = cnt - j * MSD_BLOCK;
  n_here = left < (unsigned int)MSD_BLOCK ? (int)left : MSD_BLOCK;
  return true;
}

// ---- histogram of a block's digits into LDS (h[radix]); `live` entries only
__device__ __forceinline__ void msd_count(unsigned int* __restrict__ h, const int (&digit)[MSD_ITEMS], const bool (&live)[MSD_ITEMS]) {
  const int lane = threadIdx.x & (RUA_WAVE - 1);
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) {
    // 64 entries of one digit (a sorted or clustered index): one add of 64 instead of 64 adds to one LDS address
    const int d0 = __builtin_amdgcn_readfirstlane(digit[u]);
    if (__all(live[u] && digit[u] == d0)) {
      if (lane == 0) atomicAdd(&h[d0], (unsigned int)RUA_WAVE);
    } else if (live[u]) {
      atomicAdd(&h[digit[u]], 1u);
    }
  }
}

// ---- level 1, step 1: per-block histogram of the HIGH digit, digit-major: table1[bin * n_blocks + block]
__global__ __launch_bounds__(MSD_THREADS) void msd_hist1_kernel(const int64_t* __restrict__ index, MsdGeom G, int64_t per_xcd,
                                                               unsigned int* __restrict__ table1) {
  __shared__ unsigned int h[MSD_RADIX];
  const int tid = threadIdx.x;
  const int64_t block = msd_block(per_xcd);
  if (block >= G.n_blocks) return;
  h[tid] = 0;
  __syncthreads();
  const int64_t base = block * MSD_BLOCK + (int64_t)(tid >> 6) * (RUA_WAVE * MSD_ITEMS) + (tid & (RUA_WAVE - 1));
  int digit[MSD_ITEMS];
  bool live[MSD_ITEMS];
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) {
    const int64_t i = base + (int64_t)u * RUA_WAVE;
    const int64_t v = i < G.M ? index[i] : -1;
    live[u] = v >= 0 && v < G.S;
    digit[u] = live[u] ? (int)(v >> G.lo_bits) : 0;
  }
  msd_count(h, digit, live);
  __syncthreads();
  if (tid < G.n_bins) table1[(int64_t)tid * G.n_blocks + block] = h[tid];
}

// ---- exclusive scan of every ROW of a [n_rows, row_len] uint32 table, in place; rowtot[r] = the row's sum.
// One workgroup per row (level 1: n_bins rows of n_blocks entries).
__global__ __launch_bounds__(RUA_BLOCK) void msd_scan_rows_kernel(unsigned int* __restrict__ table, int64_t row_len,
                                                                 unsigned int* __restrict__ rowtot) {
  __shared__ unsigned int wave_tot[RUA_WAVES_PER_BLOCK];
  unsigned int* __restrict__ row = table + (int64_t)blockIdx.x * row_len;
  const int tid = threadIdx.x, lane = tid & (RUA_WAVE - 1), wave = tid >> 6;
  unsigned int carry = 0;
  for (int64_t t0 = 0; t0 < row_len; t0 += RUA_BLOCK * 4) {
    const int64_t i0 = t0 + (int64_t)tid * 4;
    unsigned int v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i0 + q < row_len ? row[i0 + q] : 0u;
    const unsigned int mine = v[0] + v[1] + v[2] + v[3];
    unsigned int incl = mine;
#pragma unroll
    for (int s = 1; s < RUA_WAVE; s <<= 1) {
      const unsigned int up = __shfl_up(incl, s, RUA_WAVE);
      if (lane >= s) incl += up;
    }
    __syncthreads();
    if (lane == RUA_WAVE - 1) wave_tot[wave] = incl;
    __syncthreads();
    unsigned int before = carry, all = 0;
#pragma unroll
    for (int w = 0; w < RUA_WAVES_PER_BLOCK; ++w) {
      const unsigned int t = wave_tot[w];
      before += w < wave ? t : 0u;
      all += t;
    }
    unsigned int run = before + incl - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i0 + q < row_len) row[i0 + q] = run;
      run += v[q];
    }
    carry += all;
  }
  if (tid == 0) rowtot[blockIdx.x] = carry;
}

// ---- stable ranking of a block's entries by digit and staging in block-sorted order (the scheme of rua_scatter.hip).
// On return: `stage[q]` holds the q-th entry of the block in (digit, input order); dstart[d] = first slot of digit d.
// OUT: what is staged (the level-1 word, or the level-2 row).
// `side` (may be NULL): side[q] = digit >> 1 of the entry staged at slot q, for outputs that have no room for the digit.
template <typename OUT>
__device__ __forceinline__ void msd_rank_and_stage(const int (&digit)[MSD_ITEMS], const bool (&live)[MSD_ITEMS],
                                                   const OUT (&val)[MSD_ITEMS], int width, OUT* __restrict__ stage,
                                                   unsigned short (*__restrict__ wcnt)[MSD_RADIX],
                                                   unsigned short* __restrict__ dstart, unsigned int* __restrict__ wave_tot,
                                                   unsigned char* __restrict__ side = nullptr) {
  const int tid = threadIdx.x, lane = tid & (RUA_WAVE - 1), wave = tid >> 6;
  const int radix = 1 << width;
  for (int i = tid; i < MSD_WAVES * MSD_RADIX / 2; i += MSD_THREADS) reinterpret_cast<unsigned int*>(&wcnt[0][0])[i] = 0;
  __syncthreads();
  unsigned short before[MSD_ITEMS];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned short* __restrict__ mine = wcnt[wave];
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) {
    unsigned long long peers = __ballot(live[u]);
    for (int b = 0; b < width; ++b) {
      const unsigned long long m = __ballot(live[u] && ((digit[u] >> b) & 1));
      peers &= ((digit[u] >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const unsigned int seen = live[u] ? mine[digit[u]] : 0u;       // every lane of the digit reads the counter ...
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live[u] && rank == 0) mine[digit[u]] = (unsigned short)(seen + (unsigned int)__popcll(peers));   // ... then its first lane moves it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    before[u] = (unsigned short)(seen + rank);
  }
  __syncthreads();
  // one thread per digit: the waves' counts become the waves' starts inside the digit; digit totals -> digit starts
  unsigned int total = 0;
  if (tid < radix) {
#pragma unroll
    for (int v = 0; v < MSD_WAVES; ++v) {
      const unsigned int c = wcnt[v][tid];
      wcnt[v][tid] = (unsigned short)total;
      total += c;
    }
  }
  unsigned int all;
  const unsigned int start = msd_block_scan(total, wave_tot, all);
  if (tid < radix) dstart[tid] = (unsigned short)start;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u)
    if (live[u]) {
      const int q = dstart[digit[u]] + wcnt[wave][digit[u]] + before[u];
      stage[q] = val[u];
      if (side) side[q] = (unsigned char)(digit[u] >> 1);
    }
  __syncthreads();
}

// ---- level 1, step 3: stable scatter by the high digit; writes (low digit << row_bits | row) words
template <typename W>
__global__ __launch_bounds__(MSD_THREADS) void msd_scatter1_kernel(const int64_t* __restrict__ index, MsdGeom G, int64_t per_xcd,
                                                                  const unsigned int* __restrict__ table1,
                                                                  const unsigned int* __restrict__ rowtot,
                                                                  W* __restrict__ words) {
  __shared__ W stage[MSD_BLOCK];
  __shared__ unsigned short wcnt[MSD_WAVES][MSD_RADIX];
  __shared__ unsigned short dstart[MSD_RADIX];
  __shared__ unsigned int delta[MSD_RADIX];            // bin: global first slot - first slot inside the block
  __shared__ unsigned char sbin[MSD_BLOCK];            // bin >> 1 of every staged slot (a 32-bit word has no room for it)
  __shared__ unsigned int wave_tot[MSD_WAVES];
  const int tid = threadIdx.x;
  const int64_t block = msd_block(per_xcd);
  if (block >= G.n_blocks) return;
  int width = 0;
  while ((1 << width) < G.n_bins) ++width;
  // first output slot of every bin (exclusive scan of the row totals)
  unsigned int all_valid;
  const unsigned int bin_base = msd_block_scan(tid < G.n_bins ? rowtot[tid] : 0u, wave_tot, all_valid);

  const int64_t base = block * MSD_BLOCK + (int64_t)(tid >> 6) * (RUA_WAVE * MSD_ITEMS) + (tid & (RUA_WAVE - 1));
  const int64_t lo_mask = ((int64_t)1 << G.lo_bits) - 1;
  int digit[MSD_ITEMS];
  bool live[MSD_ITEMS];
  W val[MSD_ITEMS];
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) {
    const int64_t i = base + (int64_t)u * RUA_WAVE;
    const int64_t v = i < G.M ? index[i] : -1;
    live[u] = v >= 0 && v < G.S;
    digit[u] = live[u] ? (int)(v >> G.lo_bits) : 0;
    val[u] = (W)(((uint64_t)(v & lo_mask) << G.row_bits) | (uint64_t)i);
  }
  msd_rank_and_stage<W>(digit, live, val, width, stage, wcnt, dstart, wave_tot, sbin);
  if (tid < G.n_bins) delta[tid] = bin_base + table1[(int64_t)tid * G.n_blocks + block] - (unsigned int)dstart[tid];
  __syncthreads();
  // the block's live entries occupy slots [0, total); consecutive threads write consecutive words, and a slot's bin is
  // the last one whose first slot is <= the slot (bins without entries share their successor's first slot)
  __shared__ int s_total;
  if (tid == 0) s_total = 0;
  __syncthreads();
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) cnt += live[u] ? 1 : 0;
  // wave reduction, one atomic per wave
#pragma unroll
  for (int d = RUA_WAVE / 2; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, RUA_WAVE);
  if ((tid & (RUA_WAVE - 1)) == 0) atomicAdd(&s_total, cnt);
  __syncthreads();
  const int total = s_total;
  for (int q = tid; q < total; q += MSD_THREADS) {
    int bin = (int)sbin[q] << 1;                      // the even bin of the pair; the odd one starts at dstart[bin + 1]
    if (bin + 1 < G.n_bins && (int)dstart[bin + 1] <= q) ++bin;
    words[(int64_t)(unsigned int)(delta[bin] + (unsigned int)q)] = stage[q];     // (delta wraps: mod 2^32 arithmetic)
  }
}

// ---- level 2, step 1: per-segment histogram of the LOW digit into the bin's (digit, segment) table
template <typename W>
__global__ __launch_bounds__(MSD_THREADS) void msd_hist2_kernel(const W* __restrict__ words, MsdGeom G, int64_t per_xcd,
                                                               const unsigned int* __restrict__ rowtot,
                                                               unsigned int* __restrict__ table2) {
  __shared__ unsigned int h[MSD_RADIX];
  __shared__ unsigned int s_base[MSD_RADIX + 1], s_seg[MSD_RADIX + 1];
  __shared__ unsigned int wave_tot[MSD_WAVES];
  const int tid = threadIdx.x;
  const int64_t w = msd_block(per_xcd);
  if (w >= G.grid2) return;
  msd_bins(rowtot, G.n_bins, s_base, s_seg, wave_tot);
  const int radix = 1 << G.lo_bits;
  int bin, n_here;
  unsigned int j, nseg;
  int64_t first;
  if (!msd_segment(s_base, s_seg, G.n_bins, w, bin, j, nseg, first, n_here)) {
    // past the last segment: this workgroup's slice of the table lies behind everything the bins use — zero it, so
    // that the whole table is defined
    if (tid < radix) table2[(w << G.lo_bits) + tid] = 0u;
    return;
  }
  h[tid] = 0;
  __syncthreads();
  const int off = (tid >> 6) * (RUA_WAVE * MSD_ITEMS) + (tid & (RUA_WAVE - 1));
  int digit[MSD_ITEMS];
  bool live[MSD_ITEMS];
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) {
    const int q = off + u * RUA_WAVE;
    live[u] = q < n_here;
    const W wd = live[u] ? words[first + q] : (W)0;
    digit[u] = (int)((uint64_t)wd >> G.row_bits);
  }
  msd_count(h, digit, live);
  __syncthreads();
  if (tid < radix) table2[((int64_t)s_seg[bin] << G.lo_bits) + (int64_t)tid * nseg + j] = h[tid];
}

// ---- level 2, step 2: ONE workgroup per bin scans the bin's (digit, segment) table in place — entries become the
// final position of the first row of every (digit, segment) group — and writes counts[] / off[] of the bin's
// destinations (what rua_scatter.hip needs a kernel of binary searches for).
__global__ __launch_bounds__(MSD_THREADS) void msd_scan_bins_kernel(MsdGeom G, const unsigned int* __restrict__ rowtot,
                                                                   unsigned int* __restrict__ table2,
                                                                   int64_t* __restrict__ counts, int64_t* __restrict__ off) {
  __shared__ unsigned int s_base[MSD_RADIX + 1], s_seg[MSD_RADIX + 1];
  __shared__ unsigned int wave_tot[MSD_WAVES];
  const int tid = threadIdx.x;
  const int bin = blockIdx.x;
  msd_bins(rowtot, G.n_bins, s_base, s_seg, wave_tot);
  const unsigned int nseg = s_seg[bin + 1] - s_seg[bin];
  const unsigned int cnt = s_base[bin + 1] - s_base[bin];
  const int radix = 1 << G.lo_bits;
  unsigned int* __restrict__ t = table2 + ((int64_t)s_seg[bin] << G.lo_bits);
  const int64_t len = (int64_t)nseg << G.lo_bits;
  unsigned int carry = s_base[bin];
  for (int64_t t0 = 0; t0 < len; t0 += MSD_THREADS * 4) {
    const int64_t i0 = t0 + (int64_t)tid * 4;
    unsigned int v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i0 + q < len ? t[i0 + q] : 0u;
    const unsigned int mine = v[0] + v[1] + v[2] + v[3];
    unsigned int all;
    unsigned int run = carry + msd_block_scan(mine, wave_tot, all);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i0 + q < len) t[i0 + q] = run;
      run += v[q];
    }
    carry += all;
  }
  __threadfence_block();
  __syncthreads();
  // destinations of this bin: key = (bin << lo_bits) | d.  off = position of the digit's first row
  if (tid < radix) {
    const int64_t key = ((int64_t)bin << G.lo_bits) | tid;
    if (key < G.S) {
      const unsigned int end = s_base[bin] + cnt;
      // (written a moment ago by other waves of this workgroup: read past the L1, which may still hold the old line)
      const unsigned int a = nseg ? __hip_atomic_load(&t[(int64_t)tid * nseg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s_base[bin];
      const unsigned int b = (nseg && tid + 1 < radix)
                                 ? __hip_atomic_load(&t[(int64_t)(tid + 1) * nseg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : end;
      off[key] = (int64_t)a;
      counts[key] = (int64_t)(b - a);
    }
  }
}

// ---- level 2, step 3: stable scatter of a segment's rows by the low digit -> perm (int64 rows)
template <typename W>
__global__ __launch_bounds__(MSD_THREADS) void msd_scatter2_kernel(const W* __restrict__ words, MsdGeom G, int64_t per_xcd,
                                                                  const unsigned int* __restrict__ rowtot,
                                                                  const unsigned int* __restrict__ table2,
                                                                  int64_t* __restrict__ perm) {
  __shared__ W stage[MSD_BLOCK];                       // the words themselves: the digit rides in their high bits
  __shared__ unsigned short wcnt[MSD_WAVES][MSD_RADIX];
  __shared__ unsigned short dstart[MSD_RADIX];
  __shared__ unsigned int delta[MSD_RADIX];
  __shared__ unsigned int s_base[MSD_RADIX + 1], s_seg[MSD_RADIX + 1];
  __shared__ unsigned int wave_tot[MSD_WAVES];
  const int tid = threadIdx.x;
  const int64_t w = msd_block(per_xcd);
  if (w >= G.grid2) return;
  msd_bins(rowtot, G.n_bins, s_base, s_seg, wave_tot);
  int bin, n_here;
  unsigned int j, nseg;
  int64_t first;
  if (!msd_segment(s_base, s_seg, G.n_bins, w, bin, j, nseg, first, n_here)) return;
  const int radix = 1 << G.lo_bits;
  const uint64_t row_mask = ((uint64_t)1 << G.row_bits) - 1;
  const int off = (tid >> 6) * (RUA_WAVE * MSD_ITEMS) + (tid & (RUA_WAVE - 1));
  int digit[MSD_ITEMS];
  bool live[MSD_ITEMS];
  W val[MSD_ITEMS];
#pragma unroll
  for (int u = 0; u < MSD_ITEMS; ++u) {
    const int q = off + u * RUA_WAVE;
    live[u] = q < n_here;
    val[u] = live[u] ? words[first + q] : (W)0;
    digit[u] = (int)((uint64_t)val[u] >> G.row_bits);
  }
  msd_rank_and_stage<W>(digit, live, val, G.lo_bits, stage, wcnt, dstart, wave_tot);
  if (tid < radix) delta[tid] = table2[((int64_t)s_seg[bin] << G.lo_bits) + (int64_t)tid * nseg + j] - (unsigned int)dstart[tid];
  __syncthreads();
  for (int q = tid; q < n_here; q += MSD_THREADS) {
    const uint64_t wd = (uint64_t)stage[q];
    perm[(int64_t)(unsigned int)(delta[(int)(wd >> G.row_bits)] + (unsigned int)q)] = (int64_t)(wd & row_mask);
  }
}

static inline int msd_bits_of(int64_t v) { int b = 1; while (b < 63 && (v >> b) != 0) ++b; return b; }

// Workspace (bytes): table1 [512 x n_blocks] u32 | rowtot [512] u32 | table2 [(n_blocks + 512) x 512] u32 | words [M] u64
int64_t bucket_msd_ws_bytes(int64_t M) {
  const int64_t nb = (M + MSD_BLOCK - 1) / MSD_BLOCK;
  return 4 * ((int64_t)MSD_RADIX * nb + MSD_RADIX + (nb + MSD_RADIX) * MSD_RADIX) + 8 * M + 256;
}

bool bucket_msd_applies(int64_t M, int64_t S) {
  if (M <= 0 || S <= MSD_RADIX || M >= (1ll << 31)) return false;      // small S: one LSD pass does it
  return msd_bits_of(S - 1) <= 2 * MSD_RADIX_BITS;
}

int bucket_msd(const int64_t* index, int64_t M, int64_t S, int64_t* counts, int64_t* off, int64_t* perm, void* ws,
               hipStream_t s) {
  MsdGeom G;
  G.M = M;
  G.S = S;
  const int key_bits = msd_bits_of(S - 1);
  G.row_bits = msd_bits_of(M - 1 > 0 ? M - 1 : 1);
  G.lo_bits = key_bits / 2;
  // a level-1 word is (low digit << row_bits | row): give the low digit fewer bits if that makes the word fit 32
  // (17 M rows into 65 536 destinations: 25 row bits, 9 + 7 instead of 8 + 8)
  if (G.lo_bits + G.row_bits > 32 && G.row_bits < 32 && key_bits - (32 - G.row_bits) <= MSD_RADIX_BITS) G.lo_bits = 32 - G.row_bits;
  G.n_bins = (int)(((S - 1) >> G.lo_bits) + 1);
  if (G.n_bins > MSD_RADIX || G.lo_bits > MSD_RADIX_BITS) return RUA_ERANGE;
  G.n_blocks = (M + MSD_BLOCK - 1) / MSD_BLOCK;
  G.grid2 = G.n_blocks + G.n_bins;                         // every bin rounds its segment count up at most once
  unsigned int* table1 = (unsigned int*)ws;
  unsigned int* rowtot = table1 + (int64_t)MSD_RADIX * G.n_blocks;
  unsigned int* table2 = rowtot + MSD_RADIX;
  void* words = (void*)(((uintptr_t)(table2 + (G.n_blocks + MSD_RADIX) * (int64_t)MSD_RADIX) + 15) & ~(uintptr_t)15);
  const bool narrow = G.lo_bits + G.row_bits <= 32;

  const int64_t per1 = (G.n_blocks + 7) / 8, per2 = (G.grid2 + 7) / 8;
  const dim3 g1((unsigned)(per1 * 8)), g2((unsigned)(per2 * 8)), blk(MSD_THREADS);
  hipLaunchKernelGGL(msd_hist1_kernel, g1, blk, 0, s, index, G, per1, table1);
  hipLaunchKernelGGL(msd_scan_rows_kernel, dim3((unsigned)G.n_bins), dim3(RUA_BLOCK), 0, s, table1, G.n_blocks, rowtot);
  if (narrow) {
    hipLaunchKernelGGL(msd_scatter1_kernel<uint32_t>, g1, blk, 0, s, index, G, per1, (const unsigned int*)table1,
                       (const unsigned int*)rowtot, (uint32_t*)words);
    hipLaunchKernelGGL(msd_hist2_kernel<uint32_t>, g2, blk, 0, s, (const uint32_t*)words, G, per2,
                       (const unsigned int*)rowtot, table2);
  } else {
    hipLaunchKernelGGL(msd_scatter1_kernel<uint64_t>, g1, blk, 0, s, index, G, per1, (const unsigned int*)table1,
                       (const unsigned int*)rowtot, (uint64_t*)words);
    hipLaunchKernelGGL(msd_hist2_kernel<uint64_t>, g2, blk, 0, s, (const uint64_t*)words, G, per2,
                       (const unsigned int*)rowtot, table2);
  }
  hipLaunchKernelGGL(msd_scan_bins_kernel, dim3((unsigned)G.n_bins), blk, 0, s, G, (const unsigned int*)rowtot, table2,
                     counts, off);
  if (narrow)
    hipLaunchKernelGGL(msd_scatter2_kernel<uint32_t>, g2, blk, 0, s, (const uint32_t*)words, G, per2,
                       (const unsigned int*)rowtot, (const unsigned int*)table2, perm);
  else
    hipLaunchKernelGGL(msd_scatter2_kernel<uint64_t>, g2, blk, 0, s, (const uint64_t*)words, G, per2,
                       (const unsigned int*)rowtot, (const unsigned int*)table2, perm);
  return (int)hipGetLastError();
}

}  // namespace rua
