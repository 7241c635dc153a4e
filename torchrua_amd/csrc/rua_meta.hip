// rua_meta.hip — index-generation kernels (int64, bit-exact): prefix scan, PackedSequence
// metadata, row enumeration (ptr/idx) and masks.  gfx950, wave64.  See include/rua.h.
#include <stdlib.h>
#include "rua_dev.h"

namespace rua {

// ------------------------------------------------------------------ wave / block scan
__device__ __forceinline__ int64_t wave_inclusive_scan(int64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < RUA_WAVE; d <<= 1) {
    int64_t o = __shfl_up(v, d, RUA_WAVE);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ __forceinline__ int64_t wave_sum(int64_t v) {
#pragma unroll
  for (int d = RUA_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, RUA_WAVE);
  return v;
}

// exclusive scan of one value per thread across the 256-thread block; *block_total = sum
__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* block_total) {
  __shared__ int64_t s_wave[RUA_WAVES_PER_BLOCK];
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  int64_t inc = wave_inclusive_scan(v, lane);
  if (lane == RUA_WAVE - 1) s_wave[wave] = inc;
  __syncthreads();
  int64_t pre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RUA_WAVES_PER_BLOCK; ++w) {
    int64_t x = s_wave[w];
    if (w < wave) pre += x;
    tot += x;
  }
  __syncthreads();  // s_wave reusable by the caller's next call
  *block_total = tot;
  return pre + inc - v;
}

constexpr int SCAN_ITEMS = 8;                       // int64 per thread
constexpr int SCAN_TILE = RUA_BLOCK * SCAN_ITEMS;   // 2048 per block

// pass A: per-tile totals
__global__ __launch_bounds__(RUA_BLOCK) void scan_partials_kernel(const int64_t* __restrict__ in, int64_t n,
                                                                  int64_t* __restrict__ part) {
  __shared__ int64_t s_wave[RUA_WAVES_PER_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = base + (int64_t)k * RUA_BLOCK + threadIdx.x;  // coalesced
    if (i < n) s += in[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t t = 0;
    for (int w = 0; w < RUA_WAVES_PER_BLOCK; ++w) t += s_wave[w];
    part[blockIdx.x] = t;
  }
}

// pass B: one block turns the nb tile totals into exclusive prefixes (in place)
__global__ __launch_bounds__(RUA_BLOCK) void scan_spine_kernel(int64_t* __restrict__ part, int64_t nb,
                                                               int64_t* __restrict__ total) {
  int64_t carry = 0;
  for (int64_t c = 0; c < nb; c += RUA_BLOCK) {
    int64_t i = c + threadIdx.x;
    int64_t v = i < nb ? part[i] : 0;
    int64_t tot;
    int64_t ex = block_exclusive_scan(v, &tot);
    if (i < nb) part[i] = carry + ex;
    carry += tot;
  }
  // (total[1]: the apply pass counts the inputs that are <= 0 into it — see rua_exclusive_scan_i64)
  if (threadIdx.x == 0 && total) { total[0] = carry; total[1] = 0; }
}

// pass C: scan inside the tile, offset by the tile prefix.  Thread k owns SCAN_ITEMS
// consecutive elements so the per-thread partial order is the global order.
__global__ __launch_bounds__(RUA_BLOCK) void scan_apply_kernel(const int64_t* __restrict__ in, int64_t n,
                                                               const int64_t* __restrict__ part,
                                                               int64_t* __restrict__ out,
                                                               int64_t* __restrict__ total_single,
                                                               unsigned long long* __restrict__ nonpos) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t v[SCAN_ITEMS];
  int64_t s = 0;
  int np = 0;                                   // inputs <= 0 (a length vector: the EMPTY sequences)
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
    np += (base + k < n && v[k] <= 0) ? 1 : 0;
  }
  if (total_single || nonpos) {                  // (block-uniform)
    __shared__ int s_np;
    if (threadIdx.x == 0) s_np = 0;
    __syncthreads();
    if (np) atomicAdd(&s_np, np);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (total_single) total_single[1] = s_np;                      // one tile: this block is the whole input
      else if (s_np) atomicAdd(nonpos, (unsigned long long)s_np);    // many tiles: the spine pass zeroed the word
    }
  }
  int64_t tot;
  int64_t run = block_exclusive_scan(s, &tot) + (part ? part[blockIdx.x] : 0);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (total_single && threadIdx.x == 0) total_single[0] = tot;
}

// ------------------------------------------------------------------ PackedSequence metadata
__global__ __launch_bounds__(RUA_BLOCK) void pack_meta_kernel(const int64_t* __restrict__ lens,
                                                              const int64_t* __restrict__ sorted, int64_t B,
                                                              int64_t T, int64_t* __restrict__ unsorted,
                                                              int64_t* __restrict__ bsz) {
  const int64_t i = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (i < B && unsorted) unsorted[sorted[i]] = i;
  if (i < T && bsz) {
    // lens[sorted[r]] is non-increasing in r: count r with lens[sorted[r]] > i
    int64_t lo = 0, hi = B;  // first r in [lo, hi] with key <= i
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (lens[sorted[mid]] > i) lo = mid + 1; else hi = mid;
    }
    bsz[i] = lo;
  }
}

// Everything pack() needs in ONE launch when T and B are moderate (mid-size batches spend as long in the launch
// gaps of five tiny kernels as in their payload): block 0 computes batch_sizes AND their exclusive offsets (T <=
// one scan tile); blocks 1 .. nt scan one tile of the lengths each — a block first adds up everything IN FRONT of
// its tile by itself (B <= 64 tiles: at most 129 024 coalesced int64 from L2 per block, all blocks at once; a
// single looping block would walk the 32 tiles of the north-star batch one after the other, and the three-pass scan
// is three more launches); the other blocks invert the permutation.
constexpr int64_t PREP_T_MAX = SCAN_TILE;
constexpr int64_t PREP_B_MAX = 64 * SCAN_TILE;

__global__ __launch_bounds__(RUA_BLOCK) void pack_prepare_kernel(const int64_t* __restrict__ lens,
                                                                 const int64_t* __restrict__ sorted, int64_t B,
                                                                 int64_t T, int64_t* __restrict__ unsorted,
                                                                 int64_t* __restrict__ bsz,
                                                                 int64_t* __restrict__ boff,
                                                                 int64_t* __restrict__ off, int64_t n_off_tiles) {
  if (blockIdx.x == 0) {
    // batch_sizes: lens[sorted[r]] is non-increasing in r, bsz[t] = #{r : lens[sorted[r]] > t}.  A lane takes the
    // time steps tid, tid + 256, ... and runs their searches in LOCKSTEP (fixed trip count, SCAN_ITEMS independent
    // load pairs in flight per step): the chain is as long as one search, not SCAN_ITEMS of them.
    __shared__ int64_t s_bsz[SCAN_TILE];
    int64_t lo[SCAN_ITEMS], hi[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { lo[k] = 0; hi[k] = ((int64_t)threadIdx.x + k * RUA_BLOCK < T) ? B : 0; }
    for (int64_t span = B; span > 0; span >>= 1) {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; ++k) {
        const bool go = lo[k] < hi[k];
        const int64_t mid = (lo[k] + hi[k]) >> 1;
        int64_t b = go ? sorted[mid] : 0;
        if (b < 0 || b >= B) b = 0;                       // a corrupt order must not index out of range
        const int64_t v = go ? lens[b] : 0;
        if (go) { if (v > (int64_t)threadIdx.x + k * RUA_BLOCK) lo[k] = mid + 1; else hi[k] = mid; }
      }
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      const int64_t t = (int64_t)threadIdx.x + k * RUA_BLOCK;
      s_bsz[t] = lo[k];
      if (t < T) bsz[t] = lo[k];
    }
    __syncthreads();
    // their exclusive offsets: thread k owns SCAN_ITEMS consecutive time steps, so the per-thread order is the
    // global order
    const int64_t base = (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = s_bsz[base + k]; s += v[k]; }
    int64_t tot;
    int64_t run = block_exclusive_scan(s, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      if (base + k < T) boff[base + k] = run;
      run += v[k];
    }
  } else if ((int64_t)blockIdx.x <= n_off_tiles) {
    if (!off) return;
    const int64_t tile = ((int64_t)blockIdx.x - 1) * SCAN_TILE;
    // everything in front of this tile, added up by the whole block (8 independent loads in flight per thread)
    // (32 independent loads in flight per thread: the walk is a chain of L2 round trips, 31 of them at 8 loads a
    // round for the last tile of the north-star batch — 31 us measured — 8 at 32)
    int64_t pre = 0;
    constexpr int64_t STEP = (int64_t)RUA_BLOCK * SCAN_ITEMS;                   // one tile; `tile` is a multiple of it
    int64_t i0 = 0;
    for (; i0 + 4 * STEP <= tile; i0 += 4 * STEP) {
      int64_t v[4 * SCAN_ITEMS];
#pragma unroll
      for (int k = 0; k < 4 * SCAN_ITEMS; ++k) v[k] = lens[i0 + (int64_t)k * RUA_BLOCK + threadIdx.x];
#pragma unroll
      for (int k = 0; k < 4 * SCAN_ITEMS; ++k) pre += v[k];
    }
    for (; i0 < tile; i0 += STEP) {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; ++k) pre += lens[i0 + (int64_t)k * RUA_BLOCK + threadIdx.x];
    }
    int64_t carry;
    (void)block_exclusive_scan(pre, &carry);
    const int64_t base = tile + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < B) ? lens[base + k] : 0; s += v[k]; }
    int64_t tot;
    int64_t run = carry + block_exclusive_scan(s, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      if (base + k < B) off[base + k] = run;
      run += v[k];
    }
  } else {
    const int64_t i = ((int64_t)blockIdx.x - 1 - n_off_tiles) * RUA_BLOCK + threadIdx.x;
    if (i < B) {
      const int64_t b = sorted[i];
      if (b >= 0 && b < B) unsorted[b] = i;
    }
  }
}

__global__ __launch_bounds__(RUA_BLOCK) void lens_from_pack_kernel(const int64_t* __restrict__ bsz, int64_t T,
                                                                   const int64_t* __restrict__ unsorted,
                                                                   int64_t B, int64_t* __restrict__ lens) {
  const int64_t b = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (b >= B) return;
  const int64_t r = unsorted ? unsorted[b] : b;
  int64_t lo = 0, hi = T;  // bsz non-increasing: count t with bsz[t] > r
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (bsz[mid] > r) lo = mid + 1; else hi = mid;
  }
  lens[b] = lo;
}

// ------------------------------------------------------------------ ptr() / idx()
constexpr int ENUM_CHUNKS = 4;     // 64-token chunks one wave enumerates from ONE search + window
__global__ __launch_bounds__(RUA_BLOCK) void enum_rows_kernel(rua_layout L, int64_t n, int64_t* __restrict__ bp,
                                                              int64_t* __restrict__ tp,
                                                              int64_t* __restrict__ flat) {
  // a wave owns 256 consecutive tokens: ONE 64-ary search for the first of them and one 64-entry window
  // (coop_window), from which every lane reads the answers of its four tokens (coop_lookup); a token the window
  // does not reach (more than ~60 sequence starts inside the wave's tokens) searches by itself
  const int lane = threadIdx.x & (RUA_WAVE - 1);
  const int64_t wave_id = ((int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x) >> 6;
  const int64_t j0 = wave_id * (RUA_WAVE * ENUM_CHUNKS);
  if (j0 >= n) return;                                         // wave-uniform
  if (!bp && !tp && (L.kind == RUA_CAT || L.kind == RUA_PACK)) {   // C.idx() / P.idx(): storage order IS token order
#pragma unroll
    for (int c = 0; c < ENUM_CHUNKS; ++c) {
      const int64_t j = j0 + c * RUA_WAVE + lane;
      if (j < n && flat) flat[j] = j;
    }
    return;
  }
  const bool pack = L.kind == RUA_PACK;
  const int64_t n_total = pack ? L.T : L.B;
  const bool coop = pack ? (L.boff != nullptr) : (L.off != nullptr);
  int64_t lo = 0, W = 0;
  if (coop) {
    if (pack) coop_window([&](int64_t k) { return L.boff[k]; }, n_total, j0, lane, lo, W);
    else coop_window([&](int64_t k) { return cat_off(L, k); }, n_total, j0, lane, lo, W);
  }
#pragma unroll
  for (int c = 0; c < ENUM_CHUNKS; ++c) {
    const int64_t j = j0 + c * RUA_WAVE + lane;
    if (j0 + c * RUA_WAVE >= n) break;                         // wave-uniform
    const bool mine = j < n;
    int64_t k = 0, fk = 0;
    bool ok = false;
    if (coop) ok = coop_lookup(W, lo, n_total, j, k, fk);
    if (!mine) continue;
    int64_t b, t, row = j;
    if (pack) {
      if (!ok) { k = search_boff(L.boff, L.T, j); fk = L.boff[k]; }
      t = k;
      int64_t r = j - fk;
      if (r < 0 || r >= L.B) r = 0;            // a token count that does not match batch_sizes must not index out of range
      b = L.sorted ? L.sorted[r] : r;
    } else {  // CAT / LEFT / RIGHT enumerate tokens batch-major
      if (!ok) { k = search_cat(L, j); fk = cat_off(L, k); }
      b = k;
      t = j - fk;
      row = L.kind == RUA_CAT ? j : token_to_row(L, b, t, seq_len(L, b));
    }
    if (bp) bp[j] = b;
    if (tp) tp[j] = t;
    if (flat) flat[j] = row;
  }
}

// L.idx() / R.idx() (flat storage row of every token of a padded batch, layout/left.py:73-77, right.py:74-79) and the
// flat-only enumeration of a CattedSequence: a lane owns TWO CONSECUTIVE tokens (ENUM_PAIR) of each of its wave's four
// chunks — one cooperative lookup for the first, then a step along the offsets (consecutive tokens share a sequence
// until the next boundary) — and writes the pair as ONE 16-byte store, the lanes of a wave side by side (four and
// eight tokens per lane were measured and dropped: they stride the stores).  enum_rows_kernel resolved every token by itself: 60 us for the 136 MB of the
// north-star batch, a quarter of the rate of a plain store stream.
constexpr int ENUM_PAIR = 2;        // consecutive tokens per lane and chunk: one 16-byte store per output
constexpr int ENUM_NCH = 4;         // chunks of 64 * ENUM_PAIR tokens one wave enumerates from ONE search + window
__device__ __forceinline__ void store2(int64_t* __restrict__ dst, int64_t j, int64_t n, int64_t a, int64_t b) {
  if (j + 1 < n && ((uintptr_t)dst & 15) == 0) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const i64x2 v = {a, b};
    *reinterpret_cast<i64x2*>(dst + j) = v;      // lanes side by side: 1 KiB per wave instruction
  } else {
    if (j < n) dst[j] = a;
    if (j + 1 < n) dst[j + 1] = b;
  }
}

// the batch-major layouts (C / L / R enumerate their tokens sequence by sequence): ptr() and idx() — any of bp, tp, flat.
// A lane owns TWO CONSECUTIVE tokens of each of the wave's four chunks — one lookup for the first, a step along the
// offsets for the second — and writes them as ONE 16-byte store per output, the lanes of a wave side by side.
// (enum_rows_kernel resolved and stored every token by itself: 60 us for L.idx() at the north-star shape, a quarter of the
// rate of a plain store stream; 4 and 8 consecutive tokens per lane save lookups but stride the stores: 39 / 50 us.)
__global__ __launch_bounds__(RUA_BLOCK) void enum_flat_kernel(rua_layout L, int64_t n, int64_t* __restrict__ bp,
                                                              int64_t* __restrict__ tp, int64_t* __restrict__ flat) {
  constexpr int64_t BIG = 0x7fffffffffffffffLL;
  const int lane = threadIdx.x & (RUA_WAVE - 1);
  const int64_t wave_id = ((int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x) >> 6;
  const int64_t j0 = wave_id * (RUA_WAVE * ENUM_PAIR * ENUM_NCH);
  if (j0 >= n) return;                                         // wave-uniform
  int64_t lo, W;
  coop_window([&](int64_t q) { return cat_off(L, q); }, L.B, j0, lane, lo, W);
#pragma unroll
  for (int c = 0; c < ENUM_NCH; ++c) {
    const int64_t c0 = j0 + (int64_t)c * (RUA_WAVE * ENUM_PAIR);
    if (c0 >= n) break;                                        // wave-uniform
    const int64_t j = c0 + (int64_t)lane * ENUM_PAIR;
    int64_t k, fk;
    const bool ok = coop_lookup(W, lo, L.B, j < n ? j : n - 1, k, fk);
    if (j >= n) continue;
    if (!ok) { k = search_cat(L, j); fk = cat_off(L, k); }
    int64_t len = L.kind == RUA_RIGHT ? seq_len(L, k) : 0;
    const int64_t k0 = k, t0 = j - fk;
    const int64_t r0 = L.kind == RUA_CAT ? j : L.kind == RUA_LEFT ? k * L.T_phys + t0 : k * L.T_phys + (L.T_log - len) + t0;
    // the second token: the same sequence, or the next one that holds a token (zero-length ones are stepped over)
    int64_t next = k + 1 < L.B ? cat_off(L, k + 1) : BIG;
    while (j + 1 >= next) {
      ++k;
      fk = next;
      next = k + 1 < L.B ? cat_off(L, k + 1) : BIG;
      if (L.kind == RUA_RIGHT) len = seq_len(L, k);
    }
    const int64_t t1 = j + 1 - fk;
    const int64_t r1 = L.kind == RUA_CAT ? j + 1 : L.kind == RUA_LEFT ? k * L.T_phys + t1 : k * L.T_phys + (L.T_log - len) + t1;
    if (bp) store2(bp, j, n, k0, k);
    if (tp) store2(tp, j, n, t0, t1);
    if (flat) store2(flat, j, n, r0, r1);
  }
}

// The same enumeration SEQUENCE BY SEQUENCE, for batches whose sequences are long enough to fill wave instructions
// (average >= ENUM_SEQ_MIN_AVG_* tokens): a wave takes ENUM_SEQ_PER_WAVE consecutive sequences — their offsets and
// lengths arrive in ONE coalesced load, lane i holding sequence i's — and writes every sequence's tokens as a run of
// 16-byte stores (two tokens per lane, lanes side by side).  No search at all: the only dependent load in front of the
// stores is that one.  The runs start wherever the sequence starts, so the 16-byte stores sit on 8-byte boundaries
// (gfx950 takes a dwordx4 at any dword-aligned address).  L.idx() at the north-star shape: see DESIGN.md §4.2.
constexpr int ENUM_SEQ_PER_WAVE = 8;
// measured (gpurun_out/r4d/enum_ab.txt, MI355X): L.idx() 36.9 -> 24.2 us at the north-star shape (average 260 tokens),
// 50 -> 41 us at an average of 48, but 41 -> 72 us at an average of 20; with batch_ptr / token_ptr as well (two or
// three stores per token) the per-sequence form only draws level from ~130 tokens up (46.6 -> 44.4 us at 260)
constexpr int64_t ENUM_SEQ_MIN_AVG_FLAT = 48, ENUM_SEQ_MIN_AVG_PTR = 128;
constexpr int64_t ENUM_SEQ_MAX_SHARE_INV = 256;   // the longest sequence may hold at most this fraction^-1 of the tokens
typedef long long i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void store_run2(int64_t* __restrict__ dst, int64_t t, int64_t len, int64_t a, int64_t b) {
  if (t + 1 < len) {
    const i64x2_a8 v = {a, b};
    *reinterpret_cast<i64x2_a8*>(dst + t) = v;
  } else if (t < len) {
    dst[t] = a;
  }
}
__global__ __launch_bounds__(RUA_BLOCK) void enum_seq_kernel(rua_layout L, int64_t n, int64_t* __restrict__ bp,
                                                             int64_t* __restrict__ tp, int64_t* __restrict__ flat) {
  const int lane = threadIdx.x & (RUA_WAVE - 1);
  const int64_t wave_id = ((int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x) >> 6;
  const int64_t b0 = wave_id * ENUM_SEQ_PER_WAVE;
  if (b0 >= L.B) return;                                       // wave-uniform
  const int64_t bl = b0 + lane;
  const bool have = lane < ENUM_SEQ_PER_WAVE && bl < L.B;
  const int64_t my_off = have ? cat_off(L, bl) : 0;
  const int64_t my_len = have ? seq_len(L, bl) : 0;
#pragma unroll
  for (int i = 0; i < ENUM_SEQ_PER_WAVE; ++i) {
    const int64_t b = b0 + i;
    if (b >= L.B) break;                                       // wave-uniform
    const int64_t off = __shfl(my_off, i, RUA_WAVE);
    int64_t len = __shfl(my_len, i, RUA_WAVE);
    if (off + len > n) len = n - off;                          // lengths that overrun the output read as cut short
    if (len <= 0) continue;                                    // wave-uniform
    const int64_t r0 = L.kind == RUA_CAT ? off : L.kind == RUA_LEFT ? b * L.T_phys : b * L.T_phys + (L.T_log - seq_len(L, b));
    for (int64_t t = (int64_t)lane * 2; t < len; t += RUA_WAVE * 2) {
      if (bp) store_run2(bp + off, t, len, b, b);
      if (tp) store_run2(tp + off, t, len, t, t + 1);
      if (flat) store_run2(flat + off, t, len, r0 + t, r0 + t + 1);
    }
  }
}

// P.ptr(): tokens in storage order, (sorted[rank], t) — layout/pack.py:23-27 — the same way
__global__ __launch_bounds__(RUA_BLOCK) void enum_pack_kernel(rua_layout L, int64_t n, int64_t* __restrict__ bp,
                                                              int64_t* __restrict__ tp) {
  constexpr int64_t BIG = 0x7fffffffffffffffLL;
  const int lane = threadIdx.x & (RUA_WAVE - 1);
  const int64_t wave_id = ((int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x) >> 6;
  const int64_t j0 = wave_id * (RUA_WAVE * ENUM_PAIR * ENUM_NCH);
  if (j0 >= n) return;                                         // wave-uniform
  int64_t lo, W;
  coop_window([&](int64_t q) { return L.boff[q]; }, L.T, j0, lane, lo, W);
#pragma unroll
  for (int c = 0; c < ENUM_NCH; ++c) {
    const int64_t c0 = j0 + (int64_t)c * (RUA_WAVE * ENUM_PAIR);
    if (c0 >= n) break;                                        // wave-uniform
    const int64_t j = c0 + (int64_t)lane * ENUM_PAIR;
    int64_t t, bt;
    const bool ok = coop_lookup(W, lo, L.T, j < n ? j : n - 1, t, bt);
    if (j >= n) continue;
    if (!ok) { t = search_boff(L.boff, L.T, j); bt = L.boff[t]; }
    int64_t r = j - bt;
    if (r < 0 || r >= L.B) r = 0;              // a token count that does not match batch_sizes must not index out of range
    const int64_t b0 = L.sorted ? L.sorted[r] : r, t0 = t;
    int64_t next = t + 1 < L.T ? L.boff[t + 1] : BIG;
    while (j + 1 >= next) {
      ++t;
      bt = next;
      next = t + 1 < L.T ? L.boff[t + 1] : BIG;
    }
    r = j + 1 - bt;
    if (r < 0 || r >= L.B) r = 0;
    const int64_t b1 = L.sorted ? L.sorted[r] : r;
    if (bp) store2(bp, j, n, b0, b1);
    if (tp) store2(tp, j, n, t0, t);
  }
}

// ------------------------------------------------------------------ masks
// One lane writes 16 bytes (EPV elements) of the flat [B, T] grid: ONE integer division per 16-byte store locates
// the first element, the rest walk along the row (and into the next one where a vector straddles a row end).
// The B x T int64 grid of get_mask is 256 MiB at the north-star shape: a pure store stream.
// VPT: 16-byte stores per thread (a workgroup's u-th store instruction covers one contiguous 4 KiB).  A byte mask is
// 32 MiB at the north-star shape: with one store per thread the kernel was 8 192 short-lived workgroups, each waiting
// one L2 round trip for its `lens[b]` before its only store (3.5 TB/s); four per thread put four loads in flight first.
template <typename E, int VPT>
__global__ __launch_bounds__(RUA_BLOCK) void mask_kernel(const int64_t* __restrict__ lens, int64_t B, int64_t T,
                                                         E* __restrict__ out, E zero, E one, int64_t n,
                                                         int64_t head) {
  constexpr int EPV = 16 / sizeof(E);
  struct alignas(16) Vec { E v[EPV]; };
  // elements [0, head) in front of the first 16-byte boundary and the tail behind the last one go one by one
  const int64_t nvec = (n - head) / EPV;
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
  const int64_t k = ((int64_t)blockIdx.x * VPT + u) * RUA_BLOCK + threadIdx.x;
  if (k < nvec) {
    const int64_t e0 = head + k * EPV;
    // (a 64-bit division is ~100 instructions; for a byte mask that is one per 16 output BYTES and the kernel was bound
    // by it: grids below 2^32 cells divide in 32 bits)
    int64_t b = n <= 0xffffffffLL ? (int64_t)((uint32_t)e0 / (uint32_t)T) : e0 / T;
    int64_t t = e0 - b * T;
    int64_t len = lens[b];
    Vec v;
    if (sizeof(E) == 1 && t + EPV <= T) {
      // a bool / byte mask (bmask: 32 MiB at the north-star shape) whose 16 elements lie in one row: the first n1 bytes
      // are `one`, the rest `zero` — two 64-bit selects instead of sixteen compare-and-step rounds
      const int64_t left = len - t;
      const int n1 = left <= 0 ? 0 : left >= EPV ? EPV : (int)left;
      const uint64_t ones = 0x0101010101010101ull * (uint64_t)(uint8_t)one, zeros = 0x0101010101010101ull * (uint64_t)(uint8_t)zero;
      const int lo_n = n1 < 8 ? n1 : 8, hi_n = n1 > 8 ? n1 - 8 : 0;
      const uint64_t lo_m = lo_n >= 8 ? ~0ull : ((1ull << (8 * lo_n)) - 1ull);
      const uint64_t hi_m = hi_n >= 8 ? ~0ull : ((1ull << (8 * hi_n)) - 1ull);
      uint64_t w[2] = {(ones & lo_m) | (zeros & ~lo_m), (ones & hi_m) | (zeros & ~hi_m)};
      __builtin_memcpy(&v, w, sizeof(Vec));
    } else {
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        v.v[i] = t < len ? one : zero;
        if (++t == T) { t = 0; ++b; len = b < B ? lens[b] : 0; }
      }
    }
    *reinterpret_cast<Vec*>(out + e0) = v;
  } else {
    // the few unaligned elements: element `head - 1 - j` for j < head, then the tail
    const int64_t j = k - nvec;
    const int64_t tail0 = head + nvec * EPV;
    const int64_t e = j < head ? j : tail0 + (j - head);
    if (e < n) {
      const int64_t b = e / T, t = e - b * T;
      out[e] = t < lens[b] ? one : zero;
    }
  }
  }
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RUA_BLOCK - 1) / RUA_BLOCK); }

}  // namespace rua

using namespace rua;

extern "C" {

int64_t rua_scan_ws_elems(int64_t n) { return n <= 0 ? 1 : (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

int rua_exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t* total, int64_t n, int64_t* ws,
                           void* stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (total) return (int)hipMemsetAsync(total, 0, 2 * sizeof(int64_t), s);
    return 0;
  }
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb > 0x7fffffffLL) return RUA_ERANGE;
  if (nb == 1) {
    hipLaunchKernelGGL(scan_apply_kernel, dim3(1), dim3(RUA_BLOCK), 0, s, in, n, (const int64_t*)nullptr, out,
                       total, (unsigned long long*)nullptr);
    return (int)hipGetLastError();
  }
  if (!ws) return RUA_EINVAL;
  hipLaunchKernelGGL(scan_partials_kernel, dim3((unsigned)nb), dim3(RUA_BLOCK), 0, s, in, n, ws);
  hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(RUA_BLOCK), 0, s, ws, nb, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(RUA_BLOCK), 0, s, in, n, (const int64_t*)ws, out,
                     (int64_t*)nullptr, total ? (unsigned long long*)(total + 1) : (unsigned long long*)nullptr);
  return (int)hipGetLastError();
}

int rua_pack_meta(const int64_t* lens, const int64_t* sorted, int64_t B, int64_t T, int64_t* unsorted,
                  int64_t* bsz, void* stream) {
  if (B < 0 || T < 0 || (B > 0 && !sorted) || (T > 0 && !lens)) return RUA_EINVAL;
  const int64_t n = B > T ? B : T;
  if (n == 0) return 0;
  hipLaunchKernelGGL(pack_meta_kernel, dim3(grid_for(n)), dim3(RUA_BLOCK), 0, (hipStream_t)stream, lens, sorted, B,
                     T, unsorted, bsz);
  return (int)hipGetLastError();
}

int rua_pack_prepare(const int64_t* lens, const int64_t* sorted, int64_t B, int64_t T, int64_t* unsorted,
                     int64_t* bsz, int64_t* boff, int64_t* off, int64_t* ws, void* stream) {
  if (B < 0 || T < 0) return RUA_EINVAL;
  if (B == 0) return 0;
  if (!lens || !sorted || !unsorted || (T > 0 && (!bsz || !boff))) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (T <= PREP_T_MAX && B <= PREP_B_MAX) {
    const int64_t nt = (B + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(pack_prepare_kernel, dim3((unsigned)(1 + nt + grid_for(B))), dim3(RUA_BLOCK), 0, s, lens, sorted, B,
                       T, unsorted, bsz, boff, off, nt);
    return (int)hipGetLastError();
  }
  int r = rua_pack_meta(lens, sorted, B, T, unsorted, bsz, stream);
  if (r == 0 && T > 0) r = rua_exclusive_scan_i64(bsz, boff, nullptr, T, ws, stream);
  if (r == 0 && off) r = rua_exclusive_scan_i64(lens, off, nullptr, B, ws, stream);
  return r;
}

int rua_lens_from_pack(const int64_t* bsz, int64_t T, const int64_t* unsorted, int64_t B, int64_t* lens,
                       void* stream) {
  if (B < 0 || T < 0 || (B > 0 && !lens) || (T > 0 && !bsz)) return RUA_EINVAL;
  if (B == 0) return 0;
  hipLaunchKernelGGL(lens_from_pack_kernel, dim3(grid_for(B)), dim3(RUA_BLOCK), 0, (hipStream_t)stream, bsz, T,
                     unsorted, B, lens);
  return (int)hipGetLastError();
}

int rua_enum_rows(const rua_layout* lay, int64_t n_tokens, int64_t* batch_ptr, int64_t* token_ptr, int64_t* flat,
                  void* stream) {
  if (!lay || n_tokens < 0) return RUA_EINVAL;
  if (lay->kind == RUA_PACK) {
    if (n_tokens > 0 && (!lay->boff || lay->T <= 0)) return RUA_EINVAL;
  } else if (lay->kind == RUA_CAT || lay->kind == RUA_LEFT || lay->kind == RUA_RIGHT) {
    if (n_tokens > 0 && lay->B <= 0) return RUA_EINVAL;
    if (lay->lens && !lay->off) return RUA_EINVAL;
  } else {
    return RUA_EINVAL;
  }
  if (n_tokens == 0) return 0;
  if (lay->kind == RUA_PACK && (batch_ptr || token_ptr) && !flat) {      // P.ptr()
    const int64_t per_block = (int64_t)RUA_BLOCK * ENUM_PAIR * ENUM_NCH;
    hipLaunchKernelGGL(enum_pack_kernel, dim3((unsigned)((n_tokens + per_block - 1) / per_block)), dim3(RUA_BLOCK), 0,
                       (hipStream_t)stream, *lay, n_tokens, batch_ptr, token_ptr);
    return (int)hipGetLastError();
  }
  // the batch-major layouts with ragged lengths (C.idx() alone stays an iota)
  if (lay->kind != RUA_PACK && lay->off && lay->lens && (batch_ptr || token_ptr || lay->kind != RUA_CAT)) {
    const int64_t min_avg = (batch_ptr || token_ptr) ? ENUM_SEQ_MIN_AVG_PTR : ENUM_SEQ_MIN_AVG_FLAT;
    // ... sequence by sequence only when no sequence can be a large share of the launch: a wave walks its sequences
    // whole, so one multi-million-token sequence among short ones would be written by a single wave (milliseconds where
    // the token-balanced enum_flat_kernel takes tens of microseconds: ADVICE r4).  An upper bound on the lengths is the
    // storage's own T for the padded layouts and the caller's hint (rua_layout::T_log, 0 = unknown) for a CattedSequence.
    const int64_t longest = lay->kind == RUA_CAT ? lay->T_log : lay->T_phys;
    const bool by_seq = n_tokens >= lay->B * min_avg && longest > 0 && longest <= n_tokens / ENUM_SEQ_MAX_SHARE_INV;
    if (by_seq && lay->len_add == 0) {
      const int64_t waves = (lay->B + ENUM_SEQ_PER_WAVE - 1) / ENUM_SEQ_PER_WAVE;
      hipLaunchKernelGGL(enum_seq_kernel, dim3((unsigned)((waves + RUA_WAVES_PER_BLOCK - 1) / RUA_WAVES_PER_BLOCK)),
                         dim3(RUA_BLOCK), 0, (hipStream_t)stream, *lay, n_tokens, batch_ptr, token_ptr, flat);
      return (int)hipGetLastError();
    }
    const int64_t per_block = (int64_t)RUA_BLOCK * ENUM_PAIR * ENUM_NCH;
    hipLaunchKernelGGL(enum_flat_kernel, dim3((unsigned)((n_tokens + per_block - 1) / per_block)), dim3(RUA_BLOCK), 0,
                       (hipStream_t)stream, *lay, n_tokens, batch_ptr, token_ptr, flat);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(enum_rows_kernel, dim3(grid_for((n_tokens + ENUM_CHUNKS - 1) / ENUM_CHUNKS)), dim3(RUA_BLOCK), 0,
                     (hipStream_t)stream, *lay, n_tokens, batch_ptr, token_ptr, flat);
  return (int)hipGetLastError();
}

int rua_mask(const int64_t* lens, int64_t B, int64_t T, void* out, int32_t elem_bytes, uint64_t zero_bits,
             uint64_t one_bits, void* stream) {
  if (B < 0 || T < 0) return RUA_EINVAL;
  const int64_t n = B * T;
  if (n == 0) return 0;
  if (!lens || !out) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 blk(RUA_BLOCK);
#define RUA_MASK(E)                                                                                             \
  {                                                                                                             \
    constexpr int64_t EPV = 16 / sizeof(E);                                                                      \
    int64_t head = (int64_t)((16 - ((uintptr_t)out & 15)) & 15) / (int64_t)sizeof(E);                            \
    if (((uintptr_t)out % sizeof(E)) != 0) return RUA_EALIGN;                                                    \
    if (head > n) head = n;                                                                                      \
    const int64_t nvec = (n - head) / EPV;                                                                       \
    const int64_t threads = nvec + head + (n - head - nvec * EPV);                                               \
    if ((threads + RUA_BLOCK - 1) / RUA_BLOCK > 0x7fffffffLL) return RUA_ERANGE;                                 \
    constexpr int VPT = sizeof(E) <= 2 ? 4 : 1;                                                                  \
    hipLaunchKernelGGL((mask_kernel<E, VPT>), dim3(grid_for((threads + VPT - 1) / VPT)), blk, 0, s, lens, B, T,   \
                       (E*)out, (E)zero_bits, (E)one_bits, n, head);                                             \
  }
  switch (elem_bytes) {
    case 1: RUA_MASK(uint8_t); break;
    case 2: RUA_MASK(uint16_t); break;
    case 4: RUA_MASK(uint32_t); break;
    case 8: RUA_MASK(uint64_t); break;
    default: return RUA_EINVAL;
  }
#undef RUA_MASK
  return (int)hipGetLastError();
}

int rua_abi_version(void) { return RUA_ABI_VERSION; }
const char* rua_build_target(void) { return "gfx950"; }

}  // extern "C"
