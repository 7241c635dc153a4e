// rua_move.hip — the row mover: every layout conversion / select of the hot path as ONE
// two-phase kernel (gfx950, wave64, HBM-bound; no MFMA — this is byte movement).
//
//   phase 1 (index generation, one lane per row): for the tile's 256 destination rows, turn
//            row j -> token (b,t) -> source row with the closed forms of rua_dev.h; the
//            (load_row, store_row) pairs are staged in LDS.  Integer work happens once per ROW.
//   phase 2 (payload, one lane per 16 bytes): the 4 waves stream the tile, each wave
//            instruction moving 1 KiB (64 lanes x dwordx4), 4 loads in flight per lane,
//            addresses = LDS row index * row_bytes + column.  Padding rows store the fill
//            pattern without loading, so padded outputs are written exactly once.
//
// Tiles are contiguous in the DESTINATION storage, so stores are perfectly coalesced and every
// tile carries the same number of bytes (no ragged load imbalance); loads are row-granular
// gathers (>= 128 B lines, 1 KiB at the north-star shape).
#include <stdlib.h>
#include "rua_dev.h"

namespace rua {

#ifndef RUA_MOVE_BLOCK        // developer knobs for A/B builds (scripts/pack_ab.py)
#define RUA_MOVE_BLOCK 256
#endif
#ifndef RUA_MOVE_UNROLL
#define RUA_MOVE_UNROLL 4
#endif
constexpr int MOVE_BLOCK = RUA_MOVE_BLOCK;          // threads per workgroup of the generic mover
constexpr int MOVE_TILE = MOVE_BLOCK;               // one lane per row in phase 1
constexpr int UNROLL = RUA_MOVE_UNROLL;             // row groups in flight per wave in phase 2
constexpr int64_t MOVE_TILE_BYTES = 16 << 10;       // destination bytes one workgroup takes (rows: a power of two, 4..256)
constexpr int64_t MOVE_SPAN_MIN_TILES = 2048;       // launches at least this large give every XCD one contiguous span ...
constexpr int64_t MOVE_SPAN_MIN_TILES_DENSE = 1 << 19;   // ... when the destination is padded (two thirds of a pad's
// traffic is stores, and a span per XCD is worth 11-13 % to them at every size measured); a gather into a dense
// destination (C / P) only draws level from ~8 GB up and LOSES 2-5 % below (cfg2's 0.5 GB: 183 -> 173 us with plain
// blockIdx order; gpurun_out/r4j/midsize_ab.txt, r4k/span_ab.txt -> profiles/r04_span_ab.txt)
// The (rank x time) tiles of narrow rows have their own rule (profiles/r04_narrow_span_ab.txt): neighbouring tiles share
// the 128-byte lines their 512-byte runs straddle, and only a span keeps the two on one XCD's L2.  Pack and the roll
// tiles gain at every size (16-byte rows: 3.0 -> 4.5 and 3.7 -> 4.4-4.6 TB/s; 32: 3.7 -> 4.9 and 4.5 -> 5.0); P.cat,
// whose stores are whole 1-KiB runs either way, gains from ~8 GB (4-9 %) and loses 2-8 % at 1-3 GB.
constexpr int64_t TILE_SPAN_MIN_TILES_FROM_PACK = 1 << 18;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));     // a 16-byte piece that only sits on an 8-byte boundary
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));     // ... or on a 4-byte one (gfx950 takes a dwordx4 at any dword-aligned address)

template <int VEC> struct vec_of;
template <> struct vec_of<16> { using type = u32x4; };
template <> struct vec_of<8>  { using type = u32x2; };
template <> struct vec_of<4>  { using type = uint32_t; };
template <> struct vec_of<2>  { using type = uint16_t; };
template <> struct vec_of<1>  { using type = uint8_t; };

template <int VEC> __device__ __forceinline__ typename vec_of<VEC>::type fill_of(uint4 p);
template <> __device__ __forceinline__ u32x4    fill_of<16>(uint4 p) { u32x4 v = {p.x, p.y, p.z, p.w}; return v; }
template <> __device__ __forceinline__ u32x2    fill_of<8>(uint4 p)  { u32x2 v = {p.x, p.y}; return v; }
template <> __device__ __forceinline__ uint32_t fill_of<4>(uint4 p)  { return p.x; }
template <> __device__ __forceinline__ uint16_t fill_of<2>(uint4 p)  { return (uint16_t)p.x; }
template <> __device__ __forceinline__ uint8_t  fill_of<1>(uint4 p)  { return (uint8_t)p.x; }

// NT: non-temporal (streaming) accesses for payloads far larger than the 256 MiB Infinity Cache —
// every byte is touched exactly once, so keeping it in L2/MALL only evicts the index vectors.
// Measured at the north-star shape: 6.53 -> 6.26 ms (+4 %).  Small payloads keep the default policy
// so that a following kernel can still find them in cache.
template <typename V, bool NT> __device__ __forceinline__ V ld_row(const char* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
  return *reinterpret_cast<const V*>(p);
}
template <typename V, bool NT> __device__ __forceinline__ void st_row(char* p, V v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
  else *reinterpret_cast<V*>(p) = v;
}

// Rows that are a multiple of 8 but not of 16 bytes (H = 500 in bf16: 1 000-byte rows): every other row starts on an
// 8-byte boundary only.  With 8-byte lanes such a row is two wave instructions of 512 and 488 bytes (measured 4.2 TB/s
// where 1 024-byte rows move at 6.1); gfx950 takes a dwordx4 at any dword-aligned address, so the row goes as 62
// sixteen-byte lanes and one 8-byte lane instead — ONE wave instruction again (TAIL8).
template <bool NT> __device__ __forceinline__ u32x4 ld_row_a8(const char* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4_a8*>(p));
  return *reinterpret_cast<const u32x4_a8*>(p);
}
template <bool NT> __device__ __forceinline__ void st_row_a8(char* p, u32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_a8*>(p));
  else *reinterpret_cast<u32x4_a8*>(p) = v;
}

// The other side of the wave's enumerated rows: lane i holds enumerated row jw0 + i of `D` (i < nw; ALL 64 lanes must
// call) and gets the row of `S` it maps to — the source row of a gather, the destination row of a scatter — or -1 (fill /
// skip), or `pad_row` for a padding row of a padded destination.  The wave resolves its rows TOGETHER (coop_resolve).
// `win` (may be NULL): a window of D's offsets computed ONCE for rows at or before jw0 (wave_window) — the narrow-row
// kernel resolves 64 * CH consecutive rows per wave from one 64-ary search instead of CH.
struct WaveWindow { int64_t lo, W; };
__device__ __forceinline__ bool wave_window(const rua_layout& D, int64_t jw0, int lane1, WaveWindow& w) {
  if (D.kind == RUA_PACK && D.T > 0 && D.boff) { coop_window([&](int64_t k) { return D.boff[k]; }, D.T, jw0, lane1, w.lo, w.W); return true; }
  if (D.kind == RUA_CAT && D.off && D.B > 0) { coop_window([&](int64_t k) { return cat_off(D, k); }, D.B, jw0, lane1, w.lo, w.W); return true; }
  return false;
}
template <bool SCATTER>
__device__ __forceinline__ int64_t resolve_wave_rows(const rua_layout& D, const rua_layout& S, int32_t tmap, int64_t targ,
                                                     int64_t pad_row, int64_t jw0, int nw, int lane1, bool same_pack,
                                                     const WaveWindow* win = nullptr) {
  const int64_t j = jw0 + lane1;
  const bool mine = lane1 < nw;
  int64_t b = 0, t = 0, other = -1;
  bool token = false;
  bool resolved = false;
  bool done = false;
  if (D.kind == RUA_PACK && D.T > 0 && D.boff) {
    int64_t bt;
    const bool ok = win ? coop_lookup(win->W, win->lo, D.T, j, t, bt)
                        : coop_resolve([&](int64_t k) { return D.boff[k]; }, D.T, jw0, nw, lane1, t, bt);
    if (same_pack) {
      // roll / rev inside ONE PackedSequence: the rank r of a row is its own source rank, and its length is
      // #{t : bsz[t] > r} — read off batch_sizes instead of two random gathers per row.  The wave looks its rows'
      // lengths up TOGETHER (a 10-step binary search per row was the longest dependent chain of the tile: cfg4's
      // P.roll, 2-KiB rows): -bsz is non-decreasing, len = 1 + the largest k with -bsz[k] <= -(r + 1); the rows
      // of a wave are neighbouring ranks, whose lengths sit within one 64-entry window of the smallest of them.
      constexpr int64_t BIG = 0x7fffffffffffffffLL;
      const bool have = mine && ok && j - bt >= 0 && j - bt < D.bsz[0];
      const int64_t r = have ? j - bt : 0;
      const int64_t x = have ? -(r + 1) : BIG;
      int64_t base = x;
#pragma unroll
      for (int d = RUA_WAVE / 2; d > 0; d >>= 1) {
        const int64_t o = __shfl_xor(base, d, RUA_WAVE);
        base = o < base ? o : base;
      }
      int64_t wlo = 0, W = BIG, k = 0, fk = 0;
      bool hit = false;
      if (base != BIG) {                                          // wave-uniform
        coop_window([&](int64_t q) { return -D.bsz[q]; }, D.T, base, lane1, wlo, W);
        hit = coop_lookup(W, wlo, D.T, x, k, fk);
      }
      if (mine && ok) {
        resolved = true;
        int64_t len = 0;
        if (have) {
          if (hit && fk <= x) {
            len = k + 1;
          } else {                                                // the window did not reach: search alone
            int64_t lo = 0, hi = D.T;
            while (lo < hi) {
              const int64_t mid = (lo + hi) >> 1;
              if (D.bsz[mid] > r) lo = mid + 1; else hi = mid;
            }
            len = lo;
          }
        }
        const int64_t ts = apply_tmap(tmap, targ, t, len, len);
        if (have && ts >= 0 && ts < len) other = S.boff[ts] + r;
        if (other >= S.n_rows) other = -1;
        done = true;
      }
    } else if (mine && ok) {
      resolved = true;
      const int64_t r = j - bt;
      if (r >= 0 && r < D.B) {
        b = D.sorted ? D.sorted[r] : r;
        token = b >= 0 && b < D.B;
      }
    }
  } else if (D.kind == RUA_CAT && D.off && D.B > 0) {
    int64_t ob;
    const bool ok = win ? coop_lookup(win->W, win->lo, D.B, j, b, ob)
                        : coop_resolve([&](int64_t k) { return cat_off(D, k); }, D.B, jw0, nw, lane1, b, ob);
    if (mine && ok) { resolved = true; token = true; t = j - ob; }
  }
  if (mine && !done) {
    if (!resolved) token = row_to_token(D, j, b, t);
    if (token) {
      // caller-supplied (batch_ptr, token_ptr) pairs are range-checked: a bad pair yields the fill /
      // is skipped instead of faulting the GPU (the reference raises an IndexError there)
      if (D.kind == RUA_LIST) {
        if (!D.bptr) { if (t < 0) t += S.n_rows; }        // a flat row list wraps negatives like torch's indexing
        else if (b < 0 || b >= S.B) { b = 0; t = -1; }
      }
      const int64_t slen = seq_len(S, b);
      const int64_t dlen = D.kind == RUA_LIST ? slen : seq_len(D, b);
      const int64_t ts = apply_tmap(tmap, targ, t, slen, dlen);
      if (ts >= 0 && ts < slen) other = token_to_row(S, b, ts, slen);
      // metadata that does not match the storage (lengths summing past the payload, a corrupt
      // PackedSequence) must not become an out-of-bounds access: such rows read as padding
      if (other >= S.n_rows) other = -1;
    } else {
      other = pad_row;  // padding row: fill (-1) or a copy of one fixed source row
    }
  }
  (void)SCATTER;
  return mine ? other : -1;
}

// lpr      : lanes (VEC-byte columns) per row = ceil(row_bytes / VEC)
// lp_log2  : log2 of lanes a wave gives one row per instruction (<= 6); rows narrower than
//            1 KiB share a wave instruction (64 >> lp_log2 rows at a time)
// cpr      : 64-lane column chunks per row (1 unless row_bytes > 64*VEC)
// RPT      : destination rows per lane in phase 1 (1; 4 = the narrow-row variant of roll / rev inside one
//            PackedSequence, see below)
// TROWS    : destination rows per workgroup tile (<= MOVE_BLOCK; RPT == 1).  Small tiles in launch order make the
//            chip sweep the destination almost sequentially (see tile_of below and DESIGN.md §4).
// tiles_per_xcd > 0: workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); give every XCD ONE contiguous
//            span of tiles_per_xcd tiles instead of every eighth tile.
// TAIL8    : VEC == 16, rows end in an 8-byte piece and start on 8-byte boundaries (see above)
template <int VEC, bool SCATTER, bool NT, int RPT = 1, int TROWS = MOVE_TILE, int BLOCK = MOVE_BLOCK, int UNR = UNROLL,
          bool TAIL8 = false>
__global__ __launch_bounds__(BLOCK) void move_rows_kernel(rua_layout D, rua_layout S, int32_t tmap,
                                                              int64_t targ, char* __restrict__ dst,
                                                              const char* __restrict__ src, int64_t row_bytes,
                                                              int64_t lpr, int lp_log2, int cpr, uint4 fillpat,
                                                              int64_t pad_row, int64_t tiles_per_xcd) {
  using V = typename vec_of<VEC>::type;
  constexpr int TILE = TROWS * RPT;
  __shared__ int64_t s_ld[TILE];
  __shared__ int64_t s_st[TILE];

  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
  const int64_t tile0 = tile * TILE;
  const int64_t left = D.n_rows - tile0;
  if (left <= 0 || (tiles_per_xcd > 0 && (int64_t)(blockIdx.x >> 3) >= tiles_per_xcd)) return;
  const int nrows = left < TILE ? (int)left : TILE;

  // ---- phase 1: one lane per destination row
  const bool same_pack = !SCATTER && D.kind == RUA_PACK && S.kind == RUA_PACK && D.bsz && D.boff == S.boff &&
                         D.sorted == S.sorted && D.len_add == 0 && S.len_add == 0 && D.T == S.T;
  if (RPT > 1) {
    // Rows of at most 32 B inside one PackedSequence (the host only picks this variant when `same_pack` holds): a
    // 256-row tile is 8 KiB, and a workgroup spends its life in the two dependent-load searches of phase 1
    // (measured 3.1 TB/s at 32-byte rows).  Each lane therefore takes RPT rows and runs their searches in
    // LOCKSTEP — fixed trip counts, RPT independent loads in flight per step — so the tile is RPT times larger
    // for the same latency.
    int64_t j[RPT], lo[RPT], hi[RPT], t[RPT], r[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) { j[k] = tile0 + threadIdx.x + k * BLOCK; lo[k] = 0; hi[k] = D.T; }
    for (int64_t span = D.T; span > 1; span = (span + 1) >> 1) {     // largest t with boff[t] <= j
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int64_t mid = (lo[k] + hi[k]) >> 1;
        const bool go = hi[k] - lo[k] > 1;
        const int64_t v = go ? D.boff[mid] : 0;
        if (go) { if (v <= j[k]) lo[k] = mid; else hi[k] = mid; }
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) { t[k] = lo[k]; r[k] = j[k] - D.boff[t[k]]; lo[k] = 0; hi[k] = D.T; }
    for (int64_t span = D.T; span > 0; span >>= 1) {                  // len = #{t : bsz[t] > r}
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const bool go = lo[k] < hi[k];
        const int64_t mid = (lo[k] + hi[k]) >> 1;
        const int64_t v = go ? D.bsz[mid] : 0;
        if (go) { if (v > r[k]) lo[k] = mid + 1; else hi[k] = mid; }
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int i = threadIdx.x + k * BLOCK;
      if (i < nrows) {
        const int64_t len = lo[k];
        const int64_t ts = apply_tmap(tmap, targ, t[k], len, len);
        int64_t other = -1;
        if (ts >= 0 && ts < len) other = S.boff[ts] + r[k];
        if (other >= S.n_rows) other = -1;
        s_ld[i] = other;
        s_st[i] = j[k];
      }
    }
  } else {
    // every wave resolves the (up to) 64 tile rows that sit in its lanes
    const int i = threadIdx.x;
    const int lane1 = threadIdx.x & (RUA_WAVE - 1);
    const int w0 = i - lane1;                                     // first tile row of this wave
    const int nw = nrows - w0 < RUA_WAVE ? nrows - w0 : RUA_WAVE; // rows in this wave (<= 0: none)
    if (nw > 0) {                                                 // wave-uniform
      const int64_t other = resolve_wave_rows<SCATTER>(D, S, tmap, targ, pad_row, tile0 + w0, nw, lane1, same_pack);
      if (i < nrows) {
        const int64_t j = tile0 + i;
        if (SCATTER) { s_ld[i] = j; s_st[i] = other; }   // enumerated rows are the source
        else         { s_ld[i] = other; s_st[i] = j; }   // enumerated rows are the destination
      }
    }
  }
  __syncthreads();

  // ---- phase 2: stream the payload
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  const int rpw = RUA_WAVE >> lp_log2;           // rows per wave instruction
  const int rsub = lane >> lp_log2;              // which of those rows this lane serves
  const int64_t col0 = lane & ((1 << lp_log2) - 1);
  const V fillv = fill_of<VEC>(fillpat);

  for (int g0 = wave; g0 * rpw < nrows; g0 += (BLOCK / RUA_WAVE) * UNR) {
    for (int c = 0; c < cpr; ++c) {
      const int64_t col = col0 + (int64_t)c * RUA_WAVE;
      const bool colok = col < lpr;
      V val[UNR];
      int64_t st[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = (g0 + u * (BLOCK / RUA_WAVE)) * rpw + rsub;
        st[u] = -1;
        val[u] = fillv;
        if (colok && r < nrows) {
          const int64_t ld = s_ld[r];
          st[u] = s_st[r];
          if (ld >= 0 && st[u] >= 0) {
            if constexpr (TAIL8) {
              const char* p = src + ld * row_bytes + col * VEC;
              if (col == lpr - 1) {                       // the row's last, 8-byte piece
                const u32x2 h = ld_row<u32x2, NT>(p);
                const u32x4 v2 = {h.x, h.y, 0u, 0u};
                val[u] = v2;
              } else {
                val[u] = ld_row_a8<NT>(p);
              }
            } else {
              val[u] = ld_row<V, NT>(src + ld * row_bytes + col * VEC);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (st[u] >= 0) {
          if constexpr (TAIL8) {
            char* p = dst + st[u] * row_bytes + col * VEC;
            if (col == lpr - 1) { u32x2 h = {val[u].x, val[u].y}; st_row<u32x2, NT>(p, h); }
            else st_row_a8<NT>(p, val[u]);
          } else {
            st_row<V, NT>(dst + st[u] * row_bytes + col * VEC, val[u]);
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// [r5] GATHERS of rows that are a multiple of 4 but not of 16 bytes (H = 500 in bf16: 1 000-byte rows; 500-byte rows in
// fp32 ...).  Such rows start off 16-byte boundaries, and a row-granular mover moves them with misaligned 16-byte lanes on
// both sides: 4.0 TB/s for the pack at 1 000-byte rows, 4.3 even for a streaming copy, where 1 024-byte rows move at
// 6.0 - 6.3.  But a tile of consecutive DESTINATION rows is ONE contiguous span of bytes that starts on a 16-byte boundary
// (the tile holds a multiple of 16 rows).  So the tile goes through LDS: every source row is fetched with ALIGNED 16-byte
// loads — the aligned vectors that overlap it, its first and last vector holding a few bytes of its neighbours — and laid
// down in LDS at its own offset in dword pieces; then the whole span is stored with aligned 16-byte lanes.  Fill rows are
// laid down as the pattern.  Both sides of HBM see aligned full lanes; what is off-boundary happens inside LDS.
constexpr int SPAN_TILE_BYTES = 16 << 10;
template <bool NT>
__global__ __launch_bounds__(RUA_BLOCK) void move_rows_span_kernel(rua_layout D, rua_layout S, int32_t tmap, int64_t targ,
                                                                   char* __restrict__ dst, const char* __restrict__ src,
                                                                   int rb, int tile_rows, uint4 fillpat, int64_t pad_row,
                                                                   int64_t tiles_per_xcd) {
  __shared__ int64_t s_ld[RUA_BLOCK];
  __shared__ __attribute__((aligned(16))) uint32_t stage[SPAN_TILE_BYTES / 4];
  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
  const int64_t tile0 = tile * tile_rows;
  const int64_t left = D.n_rows - tile0;
  if (left <= 0 || (tiles_per_xcd > 0 && (int64_t)(blockIdx.x >> 3) >= tiles_per_xcd)) return;
  const int nrows = left < tile_rows ? (int)left : tile_rows;
  const bool same_pack = D.kind == RUA_PACK && S.kind == RUA_PACK && D.bsz && D.boff == S.boff && D.sorted == S.sorted &&
                         D.len_add == 0 && S.len_add == 0 && D.T == S.T;
  {
    const int i = threadIdx.x, lane1 = threadIdx.x & (RUA_WAVE - 1), w0 = i - lane1;
    const int nw = nrows - w0 < RUA_WAVE ? nrows - w0 : RUA_WAVE;
    if (nw > 0) {
      const int64_t other = resolve_wave_rows<false>(D, S, tmap, targ, pad_row, tile0 + w0, nw, lane1, same_pack);
      if (i < nrows) s_ld[i] = other;
    }
  }
  __syncthreads();
  // ---- the rows into LDS: item = (row, aligned 16-byte vector l of the source that overlaps it)
  const int vpr = ((rb + 12) >> 4) + 1;                     // vectors that can overlap a row that starts 0 .. 12 bytes in
  const int n_items = nrows * vpr;
  const uint32_t pat[4] = {fillpat.x, fillpat.y, fillpat.z, fillpat.w};
  constexpr int UN = 4;
  for (int it0 = threadIdx.x; it0 < n_items; it0 += RUA_BLOCK * UN) {
    u32x4 x[UN];
    int k0[UN], base[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = it0 + u * RUA_BLOCK;
      k0[u] = -1000000;                                     // (nothing to lay down)
      if (idx >= n_items) continue;
      const int row = idx / vpr, l = idx - row * vpr;
      const int64_t ld = s_ld[row];
      base[u] = row * rb;
      if (ld < 0) {                                         // a fill row: the pattern, phase = the row's own offset
        if ((l << 4) < rb) { k0[u] = l << 4; x[u] = u32x4{pat[0], pat[1], pat[2], pat[3]}; }
        continue;
      }
      const int64_t A = ld * (int64_t)rb;
      const int shift = (int)(A & 15);
      const int kk = (l << 4) - shift;                      // the vector holds bytes kk .. kk + 15 of the row
      if (kk >= rb || kk + 16 <= 0) continue;
      const char* p = src + (A - shift) + ((int64_t)l << 4);
      if (kk + 16 <= rb || ld + 1 < S.n_rows) {             // (the bytes behind the row belong to the next row: readable)
        x[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
      } else {                                              // the storage's last row: stop at its end
        uint32_t w[4] = {0, 0, 0, 0};
        for (int q = 0; q < 4; ++q) if (kk + 4 * q < rb) w[q] = reinterpret_cast<const uint32_t*>(p)[q];
        x[u] = u32x4{w[0], w[1], w[2], w[3]};
      }
      k0[u] = kk;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (k0[u] < -16) continue;
      const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
      if (((k0[u] | rb | base[u]) & 15) == 0) {             // whole vectors (rows of a multiple of 16 bytes: always)
        if (k0[u] >= 0 && k0[u] < rb) *reinterpret_cast<u32x4*>(&stage[(base[u] + k0[u]) >> 2]) = x[u];
      } else if (((k0[u] | rb | base[u]) & 7) == 0) {       // 8-byte pieces (rows of 8 mod 16 bytes: always)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int k = k0[u] + 8 * q;
          if (k >= 0 && k < rb) *reinterpret_cast<u32x2*>(&stage[(base[u] + k) >> 2]) = u32x2{w[2 * q], w[2 * q + 1]};
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = k0[u] + 4 * q;
          if (k >= 0 && k < rb) stage[(base[u] + k) >> 2] = w[q];
        }
      }
    }
  }
  __syncthreads();
  // ---- the span out: aligned 16-byte lanes
  const int nbytes = nrows * rb;
  char* d = dst + tile0 * (int64_t)rb;
  const int nvec = nbytes >> 4;
  for (int v = threadIdx.x; v < nvec; v += RUA_BLOCK) {
    const u32x4 o = *reinterpret_cast<const u32x4*>(&stage[v << 2]);
    if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(d + (v << 4)));
    else *reinterpret_cast<u32x4*>(d + (v << 4)) = o;
  }
  const int tail = (nbytes & 15) >> 2;
  if ((int)threadIdx.x < tail) reinterpret_cast<uint32_t*>(d + (nvec << 4))[threadIdx.x] = stage[(nvec << 2) + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// [r5] Rows of ONE vector — 1, 2, 4, 8 or 16 bytes: 1-D payloads (int64 token ids, fp32 scalars, bool masks), the
// commonest thing `C.new([...]).left()` is called on.  The generic kernel gives such a row a lane in phase 1 and a lane in
// phase 2 and a workgroup 256 rows: 2 KiB per workgroup at 8-byte rows behind a chain of dependent index loads — 5.8 ms
// for 500 M rows (0.09 T rows/s, 1.4 TB/s at 8 bytes, 0.17 at 1 byte), whatever the width.  Here a lane keeps its row
// from resolution to store (no LDS, no barrier) and has CH rows in flight: a wave takes 64 * CH consecutive enumerated
// rows, chunk by chunk (every chunk one contiguous run of 64 rows on the enumerated side), all CH loads are issued before
// the first store.  Same row maps as the generic kernel (resolve_wave_rows): every layout pair, gather and scatter,
// pad_row.  A gather between a PackedSequence and a batch-major layout still reads one row per cache line this way — that
// pair belongs to the (rank x time) tiles below, which this kernel only replaces when the host withholds the tile table.
template <int VEC, bool SCATTER, bool NT, int CH>
__global__ __launch_bounds__(RUA_BLOCK) void move_rows_narrow_kernel(rua_layout D, rua_layout S, int32_t tmap, int64_t targ,
                                                                     char* __restrict__ dst, const char* __restrict__ src,
                                                                     uint4 fillpat, int64_t pad_row,
                                                                     int64_t tiles_per_xcd) {
  using V = typename vec_of<VEC>::type;
  constexpr int TILE = RUA_BLOCK * CH;
  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
  const int64_t tile0 = tile * TILE;
  if (tile0 >= D.n_rows || (tiles_per_xcd > 0 && (int64_t)(blockIdx.x >> 3) >= tiles_per_xcd)) return;
  const bool same_pack = !SCATTER && D.kind == RUA_PACK && S.kind == RUA_PACK && D.bsz && D.boff == S.boff &&
                         D.sorted == S.sorted && D.len_add == 0 && S.len_add == 0 && D.T == S.T;
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  const int64_t jw = tile0 + (int64_t)wave * (RUA_WAVE * CH);
  const V fillv = fill_of<VEC>(fillpat);
  V val[CH];
  int64_t st[CH];
  WaveWindow ww;
  const bool have_win = jw < D.n_rows && wave_window(D, jw, lane, ww);      // (wave-uniform)
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int64_t jw0 = jw + c * RUA_WAVE;
    const int64_t left = D.n_rows - jw0;
    st[c] = -1;
    val[c] = fillv;
    if (left <= 0) continue;                                          // wave-uniform
    const int nw = left < RUA_WAVE ? (int)left : RUA_WAVE;
    const int64_t other = resolve_wave_rows<SCATTER>(D, S, tmap, targ, pad_row, jw0, nw, lane, same_pack, have_win ? &ww : nullptr);
    if (lane < nw) {
      const int64_t j = jw0 + lane;
      const int64_t ld = SCATTER ? j : other;
      st[c] = SCATTER ? other : j;
      if (ld >= 0 && st[c] >= 0) val[c] = ld_row<V, NT>(src + ld * VEC);
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (st[c] >= 0) st_row<V, NT>(dst + st[c] * VEC, val[c]);
}

// ---------------------------------------------------------------------------------------------
// [r5] Narrow rows between two BATCH-MAJOR layouts (C.left(), L.cat(), C.right(), R.left(), C.roll(s), trunc / head of a
// padded batch ...): a sequence's tokens are one contiguous run of bytes on both sides, so the cast is a SEGMENTED
// MEMCPY — per sequence at most two runs to copy (a roll wraps once) and the rest of the destination's slots to fill —
// and rows stop mattering: a lane moves 16 bytes whatever the row width.  The row kernels give every ROW a lane and its
// own index arithmetic: 6.3 ms for C.left() of 500 M 8-byte rows (1.9 TB/s), 4.2 ms for C.roll.  Here a wave takes
// SEQ_PER_WAVE consecutive sequences (their lengths and offsets arrive in one coalesced load, lane i holding sequence
// i's), and streams every run with 16-byte lanes at whatever dword-aligned address it starts (gfx950 takes a dwordx4
// there), four vectors in flight per lane; runs whose two ends are not a multiple of 4 bytes apart (rows of 1 or 2
// bytes: bool masks, int16) go through a byte shift in registers (copy_bytes_any).  The launcher takes it when no
// sequence can be a large share of the launch (one wave walks a whole sequence).
constexpr int SEQ_PER_WAVE = 8;        // at narrow rows; rows from 256 bytes up (odd widths) give a wave ONE sequence
// A run of bytes copied by NTHR threads with 16-byte lanes, WHATEVER the two addresses' alignment: the destination is
// brought to a dword boundary by a few single bytes, then every lane stores an aligned-to-4 dwordx4 assembled from FIVE
// aligned source dwords shifted by the two addresses' distance mod 4 (v_alignbyte: rows of 1 or 2 bytes, 6, 18 ...),
// or loaded as it is when that distance is zero (rows that are a multiple of 4 bytes).  (Aligning BOTH sides to 16 bytes —
// two aligned loads per lane and a shift — was measured and is slower: C.left() of 8-byte rows 2.3 -> 3.6 ms, r5n.)
template <bool NT, int NTHR>
__device__ __forceinline__ void copy_bytes_any(char* __restrict__ d, const char* __restrict__ s, int64_t nbytes, int tid) {
  if (nbytes <= 0) return;
  int head = (int)((4 - ((uintptr_t)d & 3)) & 3);
  if (head > nbytes) head = (int)nbytes;
  if (tid < head) d[tid] = s[tid];
  d += head; s += head; nbytes -= head;
  const int m = (int)((uintptr_t)s & 3);
  const char* sa = s - m;                                          // dword-aligned; sa + m = s
  const int64_t nvec = nbytes >> 4;
  constexpr int UN = NTHR == RUA_WAVE ? 8 : 4;          // vectors in flight per lane
  for (int64_t v0 = 0; v0 < nvec; v0 += NTHR * UN) {
    u32x4 x[UN];
    uint32_t e[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t v = v0 + u * NTHR + tid;
      e[u] = 0;
      if (v < nvec) {
        x[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(sa + (v << 4)))
                  : *reinterpret_cast<const u32x4_a4*>(sa + (v << 4));
        if (m) e[u] = *reinterpret_cast<const uint32_t*>(sa + (v << 4) + 16);      // (holds bytes of this very run)
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t v = v0 + u * NTHR + tid;
      if (v < nvec) {
        u32x4 o = x[u];
        if (m) {
          o.x = __builtin_amdgcn_alignbyte(x[u].y, x[u].x, m);
          o.y = __builtin_amdgcn_alignbyte(x[u].z, x[u].y, m);
          o.z = __builtin_amdgcn_alignbyte(x[u].w, x[u].z, m);
          o.w = __builtin_amdgcn_alignbyte(e[u], x[u].w, m);
        }
        if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_a4*>(d + (v << 4)));
        else *reinterpret_cast<u32x4_a4*>(d + (v << 4)) = o;
      }
    }
  }
  const int tail = (int)(nbytes & 15);
  if (tid < tail) d[(nvec << 4) + tid] = s[(nvec << 4) + tid];
}
template <bool NT>
__device__ __forceinline__ void wave_copy_bytes(char* __restrict__ d, const char* __restrict__ s, int64_t nbytes, int lane) {
  copy_bytes_any<NT, RUA_WAVE>(d, s, nbytes, lane);
}
// ... and filled: the pattern repeats every element (1, 2, 4, 8 or 16 bytes) and a run starts on an element boundary, so
// a dword-aligned vector of it is the pattern itself (elements wider than a dword: the run's start is a multiple of the
// element, hence the vector's phase is a multiple of 4 inside the element — rotate by it), single bytes are indexed by
// their address.
template <bool NT>
__device__ __forceinline__ void wave_fill_bytes(char* __restrict__ d, int64_t nbytes, u32x4 pat, int lane) {
  if (nbytes <= 0) return;
  // (byte k of the run holds pattern byte k mod 16: the run starts on an element boundary and 16 is a multiple of it)
  auto pat_byte = [&](int k) -> char {
    const uint32_t w = (k & 8) ? ((k & 4) ? pat.w : pat.z) : ((k & 4) ? pat.y : pat.x);
    return (char)(w >> (8 * (k & 3)));
  };
  int head = (int)((4 - ((uintptr_t)d & 3)) & 3);
  if (head > nbytes) head = (int)nbytes;
  if (lane < head) d[lane] = pat_byte(lane & 15);
  char* d4 = d + head;
  nbytes -= head;
  u32x4 rot = pat;                                                 // the pattern as seen from byte `head` of the run
  if (head) {
    const uint32_t w[4] = {pat.x, pat.y, pat.z, pat.w};
    rot.x = __builtin_amdgcn_alignbyte(w[1], w[0], head);
    rot.y = __builtin_amdgcn_alignbyte(w[2], w[1], head);
    rot.z = __builtin_amdgcn_alignbyte(w[3], w[2], head);
    rot.w = __builtin_amdgcn_alignbyte(w[0], w[3], head);
  }
  const int64_t nvec = nbytes >> 4;
  for (int64_t v = lane; v < nvec; v += RUA_WAVE) {
    if (NT) __builtin_nontemporal_store(rot, reinterpret_cast<u32x4_a4*>(d4 + (v << 4)));
    else *reinterpret_cast<u32x4_a4*>(d4 + (v << 4)) = rot;
  }
  const int tail = (int)(nbytes & 15);
  if (lane < tail) d4[(nvec << 4) + lane] = pat_byte((head + lane) & 15);
}

template <bool NT>
__global__ __launch_bounds__(RUA_BLOCK) void seq_copy_kernel(rua_layout D, rua_layout S, int32_t tmap, int64_t targ,
                                                             char* __restrict__ dst, const char* __restrict__ src,
                                                             int64_t rb, uint4 fillpat, int spw) {
  const int lane = threadIdx.x & (RUA_WAVE - 1);
  const int64_t wave_id = ((int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x) >> 6;
  const int64_t b0 = wave_id * spw;
  if (b0 >= D.B) return;                                            // wave-uniform
  const int64_t bl = b0 + lane;
  const bool have = lane < spw && bl < D.B;
  int64_t my_dlen = 0, my_slen = 0, my_dbase = 0, my_sbase = 0;
  if (have) {
    my_dlen = seq_len(D, bl);
    my_slen = bl < S.B ? seq_len(S, bl) : 0;
    if (my_dlen < 0) my_dlen = 0;
    if (my_slen < 0) my_slen = 0;
    my_dbase = token_to_row(D, bl, 0, my_dlen);
    my_sbase = bl < S.B ? token_to_row(S, bl, 0, my_slen) : 0;
  }
  const u32x4 pat = fill_of<16>(fillpat);
  const bool padded = D.kind == RUA_LEFT || D.kind == RUA_RIGHT;
#pragma unroll 1
  for (int i = 0; i < spw; ++i) {
    const int64_t b = b0 + i;
    if (b >= D.B) break;                                            // wave-uniform
    int64_t dlen = __shfl(my_dlen, i, RUA_WAVE), slen = __shfl(my_slen, i, RUA_WAVE);
    const int64_t dbase = __shfl(my_dbase, i, RUA_WAVE), sbase = __shfl(my_sbase, i, RUA_WAVE);
    // metadata that does not match the storage must not become an out-of-bounds access
    if (dbase < 0 || dbase > D.n_rows) continue;
    if (dbase + dlen > D.n_rows) dlen = D.n_rows - dbase;
    if (sbase < 0 || sbase > S.n_rows) slen = 0;
    else if (sbase + slen > S.n_rows) slen = S.n_rows - sbase;
    // destination tokens [lo, hi) come from the source tokens that start at `from`; a roll adds the wrapped run [0, lo)
    int64_t lo = 0, hi = 0, from = 0, wrap_from = -1;
    if (tmap == RUA_T_SHIFT) {
      lo = targ < 0 ? -targ : 0;
      hi = slen - targ < dlen ? slen - targ : dlen;
      if (lo > dlen) lo = dlen;
      if (hi < lo) hi = lo;
      from = lo + targ;
    } else {                                                        // RUA_T_ROLL (the launcher checked dlen == slen)
      const int64_t n = dlen < slen ? dlen : slen;
      int64_t k = 0;
      if (n > 0) { k = targ % n; if (k < 0) k += n; }
      lo = k; hi = n; from = 0; wrap_from = n - k;                  // dst [k, n) <- src [0, n - k);  dst [0, k) <- src [n - k, n)
      dlen = n;
    }
    char* drow = dst + dbase * rb;
    const char* srow = src + sbase * rb;
    if (wrap_from >= 0) { if (lo > 0) wave_copy_bytes<NT>(drow, srow + wrap_from * rb, lo * rb, lane); }
    else if (lo > 0) wave_fill_bytes<NT>(drow, lo * rb, pat, lane);
    if (hi > lo) wave_copy_bytes<NT>(drow + lo * rb, srow + from * rb, (hi - lo) * rb, lane);
    if (dlen > hi) wave_fill_bytes<NT>(drow + hi * rb, (dlen - hi) * rb, pat, lane);
    if (padded) {                                                   // the slots of the sequence that hold no token
      const int64_t slot0 = b * D.T_phys;
      const int64_t dl = __shfl(my_dlen, i, RUA_WAVE);              // (the unclipped length: where the tokens sit)
      int64_t front = dbase - slot0, back0 = dbase + dl;            // LEFT: front = 0; RIGHT: front = T_log - len
      if (front > 0) wave_fill_bytes<NT>(dst + slot0 * rb, front * rb, pat, lane);
      int64_t back = slot0 + D.T_phys - back0;
      if (back0 + back > D.n_rows) back = D.n_rows - back0;
      if (back > 0) wave_fill_bytes<NT>(dst + back0 * rb, back * rb, pat, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Narrow rows (<= 128 B): C/L/R <-> P is a transpose of the (rank, time) plane — C is contiguous along
// time, P along rank — and a destination-row-major tile makes the other side a one-row-granular gather:
// a 32-byte row out of every 128-byte line, i.e. 4x over-fetch (measured 1.3 TB/s at 32-byte rows).
// Here a tile is TR ranks x TT time steps, so BOTH sides move 16-row runs, and phase 2 walks it in 4x4
// sub-tiles.  (At >= 1 KiB rows the same tiling changes nothing — a row already fills whole lines — so
// wide rows keep the generic kernel.)  Phase 1 needs no search: rank r is live at time t iff r < bsz[t].
constexpr int TR_MAX = 128;       // a tile is (1 << TRL) ranks x (1 << TTL) time steps, TRL = 4 .. 7, TTL = 4 .. 7 (rua_layout::tile_t_log2)
constexpr int TT_MAX = 128;
constexpr int64_t TILE_MAX_ROW_BYTES = 64;       // rows up to this take the tile kernel (pack_tile_lds_kernel)

// [r4] A tile's life is one chain of dependent loads in front of its payload, and eight resident workgroups per CU
// cannot feed the HBM through it when the chain carries 8 KiB (32-byte rows, 16 x 16 tiles: 4.3 TB/s for the pack, 3.6
// for P.cat).  Two changes.  (1) Phase 1 no longer resolves every (rank, time) cell by itself (five loads per cell,
// three of them dependent): a cell's rows are AFFINE in the tile — batch-major row = base[rank] + time, PackedSequence
// row = boff[time] + rank — so sixteen lanes fetch their rank's sequence (sorted -> offset / length) and TT lanes their
// time step's (boff, bsz), and phase 2 computes every row from those two small tables.  (2) The tile grows ALONG TIME
// as rows get narrower (1 << TTL steps, chosen by the host: 64 at <= 16 B, 32 at <= 32 B, 16 at 64 B — 16 KiB of payload
// per chain, eight workgroups per CU), which also makes the batch-major side's runs 1 KiB instead of 512 / 256 B.
// Measured (8 GB payloads, profiles/r04_tile_ab.txt, r03_width_sweep.txt): pack 4.0-4.3 -> 4.9 TB/s at 32-byte rows,
// 3.3 -> 4.9 at 16; P.cat 3.3-3.6 -> 3.9, 3.1 -> 4.2; 64-byte rows unchanged (5.2 / 4.4).  32 KiB tiles and 32 ranks
// per tile: slower.
// (3) [r4, late] LINE-ALIGNED RUNS ON THE BATCH-MAJOR SIDE (P.cat only: see launch_pack_tiles).  A rank's run in a tile
// is TT rows = 1 KiB of one sequence, at whatever 32-byte offset the sequence happens to start: both ends of every run
// were partial 128-byte lines, shared with the neighbouring time chunk's tile — which runs much later, on another CU —
// i.e. over-fetched by the pack and written as partial lines by P.cat.  Measured with all-equal lengths (scripts/exp/aligned_pcat.py, 32-byte rows):
// everything aligned 6.0 / 5.6 TB/s (pack / P.cat); only the PackedSequence's runs misaligned 5.6 / 5.35; only the
// batch-major runs misaligned 5.1 / 4.35; ragged lengths 4.5 / 3.8.  So every rank gets its OWN time origin: with
// R = 128 / row_bytes rows per line and s = (first row of the sequence) mod R, the rank's windows are
// [c TT - s, (c + 1) TT - s) — whole lines on the batch-major side, except where a sequence begins and ends.  The
// PackedSequence side pays: a tile touches TT + R - 1 time steps, and at the R - 1 steps on either edge only some of
// the sixteen ranks (the cheap kind of misalignment, above).  The host builds the tile table for the shifted windows
// (chunk c needs the ranks alive at step c TT - (R - 1)) and says so in rua_layout::tile_t_log2 bits 16..23.
constexpr int TILE_SHIFT_MAX = 8;                // R <= 8: rows of 16 bytes
struct TileTables {
  int64_t obase[TR_MAX];  // batch-major storage row of the rank's window start: first row of the sequence - shift + t0
  int64_t olen[TR_MAX];   // the sequence's length (0: no such sequence)
  int shift[TR_MAX];      // the rank's time shift s in [0, R)
  int64_t ofill[TR_MAX];  // padded destination (full-grid tiles): storage row of the sequence's slot 0, -1: no such sequence
  int nlive_t[TR_MAX];    // [r5] steps of the tile the rank's sequence holds a token at: clamp(len - t0, 0, TT), clipped to the storage
  int nlive_r[TT_MAX + TILE_SHIFT_MAX];   // ranks of the tile alive at the step: clamp(bsz - r0, 0, TR), clipped to the storage
  int64_t pboff[TT_MAX + TILE_SHIFT_MAX];  // first PackedSequence row of time step t0 - (R - 1) + k
  int64_t pbsz[TT_MAX + TILE_SHIFT_MAX];   // sequences alive at that step (0 before 0 and past T)
};
constexpr int TILE_FULL_GRID = 1 << 24;   // rua_layout::tile_t_log2 bit 24: tiles cover the whole (sequence x step) grid
constexpr int TILE_STEP_ROWS = 1 << 25;   // ... bit 25: tiles of ONE time step x (1 << bits 8-15) ranks (pack_roll_steps_kernel)

template <int TTL, int TRL>
__device__ __forceinline__ void tile_tables(const rua_layout& Pk, const rua_layout& Ot, int64_t tile, TileTables& tb,
                                            int64_t& r0_out, int64_t& t0_out, int R = 1, int64_t phase = 0) {
  constexpr int TT = 1 << TTL, TR = 1 << TRL;
  int64_t lo = 0;
  int64_t r0;
  if (Pk.tile_t_log2 & TILE_FULL_GRID) {
    // [r5] a PADDED destination (P.left() / P.right()): the tiles cover the whole (sequence x step) grid, rank groups
    // fastest — no table: every chunk holds ceil(B / TR) of them — and a cell that holds no token is written as fill
    const int64_t groups = (Pk.B + TR - 1) >> TRL;
    lo = tile / groups;
    r0 = (tile - lo * groups) << TRL;
  } else {
    // which time chunk does this tile belong to?  largest c with tile_start[c] <= tile.  Up to 64 chunks
    // every lane loads one entry and a ballot counts them: ONE load instead of a six-step chain of dependent ones
    int64_t hi = Pk.n_tchunks;
    if (Pk.n_tchunks <= RUA_WAVE) {
      const int lane = threadIdx.x & (RUA_WAVE - 1);
      const int64_t v = lane < Pk.n_tchunks ? Pk.tile_start[lane] : 0x7fffffffffffffffLL;
      lo = (int64_t)__popcll(__ballot(v <= tile)) - 1;
      if (lo < 0) lo = 0;
    } else {
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (Pk.tile_start[mid] <= tile) lo = mid; else hi = mid;
      }
    }
    r0 = (tile - Pk.tile_start[lo]) * TR;
  }
  const int64_t t0 = lo * TT;
  r0_out = r0;
  t0_out = t0;
  for (int i = threadIdx.x; i < TR; i += RUA_BLOCK) {          // the ranks: sequence -> first row, length
    const int64_t r = r0 + i;
    int64_t base = 0, len = 0, slot0 = -1;
    if (r < Pk.B) {
      const int64_t b = Pk.sorted ? Pk.sorted[r] : r;
      if (b >= 0 && b < Ot.B) {
        len = seq_len(Ot, b);
        base = token_to_row(Ot, b, 0, len);
        if (Ot.kind == RUA_LEFT || Ot.kind == RUA_RIGHT) slot0 = b * Ot.T_phys;
      }
    }
    const int s = R > 1 ? (int)((base + phase) & (int64_t)(R - 1)) : 0;
    tb.shift[i] = s;
    tb.obase[i] = base - s + t0;                  // batch-major row of the rank's window start (a whole-line boundary)
    tb.olen[i] = len;
    tb.ofill[i] = slot0;
    int64_t nl = len - t0;                          // (plain windows only: the callers that shift do not read these)
    if (Ot.n_rows - (base + t0) < nl) nl = Ot.n_rows - (base + t0);
    tb.nlive_t[i] = nl < 0 ? 0 : nl > TT ? TT : (int)nl;
  }
  // the time steps t0 - (R - 1) .. t0 + TT - 1 (threads from the far end: the first ones hold a rank already)
  for (int k = RUA_BLOCK - 1 - (int)threadIdx.x; k < TT + R - 1; k += RUA_BLOCK) {
    const int64_t t = t0 - (R - 1) + k;
    const bool ok = t >= 0 && t < Pk.T;
    const int64_t first = ok ? Pk.boff[t] : 0;
    tb.pboff[k] = first;
    tb.pbsz[k] = ok ? Pk.bsz[t] : 0;
    int64_t nl = tb.pbsz[k] - r0;
    if (Pk.n_rows - (first + r0) < nl) nl = Pk.n_rows - (first + r0);
    tb.nlive_r[k] = nl < 0 ? 0 : nl > TR ? TR : (int)nl;
  }
}

// The tile goes through LDS: read in the source's contiguous order, written in the destination's (measured at 32-byte
// rows: pack 3.5 -> 4.0 TB/s against a 4x4 sub-tile walk without staging).  Rows of 128 B and more take the generic
// mover: since round 2 (16 KiB tiles, cooperative row resolution) it is the faster one there — 5.5 / 5.8 TB/s for
// C->P / P->C at 128 B against 5.3 / 5.2 for the tile kernels; at 64 B the tiles win 4.8 to 3.2, at 32 B 3.9 to 1.6
// (a round-2 A/B, profiles/r02_width_sweep.txt).
template <int VEC, bool TO_PACK, int TTL, int TRL>
__global__ __launch_bounds__(RUA_BLOCK) void pack_tile_lds_kernel(rua_layout Pk, rua_layout Ot, char* __restrict__ dst,
                                                              const char* __restrict__ src, int64_t row_bytes,
                                                              int64_t lpr, int64_t tiles_per_xcd, int R, int64_t phase,
                                                              uint4 fillpat) {
  using V = typename vec_of<VEC>::type;
  constexpr int TT = 1 << TTL, TR = 1 << TRL, TILE = TR * TT;
  static_assert(TT <= TT_MAX && TR <= TR_MAX, "TileTables holds the tile's ranks and time steps");
  __shared__ TileTables tb;

  int64_t tile = blockIdx.x;                  // (block-uniform) one contiguous span of tiles per XCD, as in the row mover
  if (tiles_per_xcd > 0) {
    tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= tiles_per_xcd || tile >= Pk.n_tiles) return;
  }
  int64_t r0, t0;
  tile_tables<TTL, TRL>(Pk, Ot, tile, tb, r0, t0, R, phase);
  __syncthreads();

  // ---- phase 2: the tile goes through LDS.  It is read in the SOURCE's contiguous order (consecutive lanes =
  // consecutive 16-byte pieces of consecutive rows of one run) and written in the DESTINATION's contiguous order,
  // so both sides see whole runs per wave instruction (16 ranks on the PackedSequence's side, TT time steps on the
  // batch-major side).  Cell (rank, j), j = the slot in the rank's OWN window (see TileTables): time t = t0 + j - shift[rank],
  // batch-major row obase[rank] + j, PackedSequence row pboff[k] + r0 + rank with k = j - shift[rank] + R - 1 (the step's
  // slot in the tile's table); live iff the sequence has a token there (0 <= t < len) and the rank is alive at the step.
  extern __shared__ __attribute__((aligned(16))) unsigned char s_stage_raw[];
  V* stage = reinterpret_cast<V*>(s_stage_raw);
  // one padding ROW (lpr slots) per rank: the transposed order strides over TT * lpr slots, which would otherwise land
  // the sixteen ranks of a time step on the same LDS banks; with (TT + 1) * lpr the sixteen lanes of a 128-bit LDS
  // pass — ranks x pieces — fall on sixteen different bank groups (a single slot of padding left piece 1 of rank r
  // on the banks of piece 0 of rank r + 1)
#define RUA_SLOT(rank, j, piece) (((((rank) << TTL) | (j)) + (rank)) * (int)lpr + (piece))
#define RUA_CELL(rank, j, k, orow, prow, live)                                                                     \
  const int64_t t_ = t0 + (j) - tb.shift[rank];                                                                    \
  const int64_t orow = tb.obase[rank] + (j), prow = tb.pboff[k] + r0 + (rank);                                     \
  const bool live = t_ >= 0 && t_ < tb.olen[rank] && r0 + (rank) < tb.pbsz[k] && orow < Ot.n_rows && prow < Pk.n_rows
  const int n_major = TILE * (int)lpr;                        // batch-major order: (rank, j), j fastest
  const int n_packed = (TT + R - 1) * TR * (int)lpr;          // PackedSequence order: (k, rank), rank fastest
  if (TO_PACK) {
#pragma unroll 4
    for (int idx = threadIdx.x; idx < n_major; idx += RUA_BLOCK) {
      const int pos = idx / (int)lpr, piece = idx - pos * (int)lpr;
      const int rank = pos >> TTL, j = pos & (TT - 1), k = j - tb.shift[rank] + R - 1;
      RUA_CELL(rank, j, k, orow, prow, live);
      (void)prow;
      if (live) stage[RUA_SLOT(rank, j, piece)] = ld_row<V, false>(src + orow * row_bytes + (int64_t)piece * VEC);
    }
    __syncthreads();
#pragma unroll 4
    for (int idx = threadIdx.x; idx < n_packed; idx += RUA_BLOCK) {
      const int pos = idx / (int)lpr, piece = idx - pos * (int)lpr;
      const int rank = pos & (TR - 1), k = pos >> TRL, j = k - (R - 1) + tb.shift[rank];
      if (j < 0 || j >= TT) continue;
      RUA_CELL(rank, j, k, orow, prow, live);
      (void)orow;
      if (live) st_row<V, false>(dst + prow * row_bytes + (int64_t)piece * VEC, stage[RUA_SLOT(rank, j, piece)]);
    }
  } else {
#pragma unroll 4
    for (int idx = threadIdx.x; idx < n_packed; idx += RUA_BLOCK) {
      const int pos = idx / (int)lpr, piece = idx - pos * (int)lpr;
      const int rank = pos & (TR - 1), k = pos >> TRL, j = k - (R - 1) + tb.shift[rank];
      if (j < 0 || j >= TT) continue;
      RUA_CELL(rank, j, k, orow, prow, live);
      (void)orow;
      if (live) stage[RUA_SLOT(rank, j, piece)] = ld_row<V, false>(src + prow * row_bytes + (int64_t)piece * VEC);
    }
    __syncthreads();
    const bool padded = (Pk.tile_t_log2 & TILE_FULL_GRID) != 0;      // (R == 1 then: the host asks for plain windows)
    const V fillv = fill_of<VEC>(fillpat);
#pragma unroll 4
    for (int idx = threadIdx.x; idx < n_major; idx += RUA_BLOCK) {
      const int pos = idx / (int)lpr, piece = idx - pos * (int)lpr;
      const int rank = pos >> TTL, j = pos & (TT - 1), k = j - tb.shift[rank] + R - 1;
      RUA_CELL(rank, j, k, orow, prow, live);
      (void)prow;
      if (live) st_row<V, false>(dst + orow * row_bytes + (int64_t)piece * VEC, stage[RUA_SLOT(rank, j, piece)]);
      else if (padded && tb.ofill[rank] >= 0 && t_ >= tb.olen[rank] && t_ < Ot.T_phys) {
        // a slot of the padded destination that holds no token: LEFT keeps its tokens in front (slot = step), RIGHT
        // at the end of the logical width (the dead steps len .. T_log - 1 are the slots 0 .. T_log - len - 1 in front)
        const int64_t slot = (Ot.kind == RUA_RIGHT && t_ < Ot.T_log) ? t_ - tb.olen[rank] : t_;
        st_row<V, false>(dst + (tb.ofill[rank] + slot) * row_bytes + (int64_t)piece * VEC, fillv);
      }
    }
  }
#undef RUA_CELL
#undef RUA_SLOT
}

// [r5] The same transpose for rows of ONE 8- or 4-byte vector (1-D int64 / fp32 payloads), SIXTEEN BYTES PER LANE.  At these
// widths the kernel above is bound by instructions, not bytes: a lane moves one row per trip (a global load, an LDS write,
// an LDS read, a global store and ~60 instructions of cell arithmetic for 8 bytes) — 2.5 ms for 500 M 8-byte rows,
// 2.0 ms for as many 4-byte rows.  Here a lane takes CPL = 16 / RB consecutive cells along the side's CONTIGUOUS axis — steps
// of one rank on the batch-major side, ranks of one step on the PackedSequence's — as one 16-byte global access (any
// dword-aligned address will do on gfx950) and CPL row-sized LDS accesses; a group that straddles the end of a sequence
// or of a time step falls back to its cells one by one.  Tiles: 32 ranks x 64 steps (8-byte rows), 64 x 64 (4-byte).
template <int RB, bool TO_PACK, int TTL, int TRL>
__global__ __launch_bounds__(RUA_BLOCK) void pack_tile_vec_kernel(rua_layout Pk, rua_layout Ot, char* __restrict__ dst,
                                                                  const char* __restrict__ src, int64_t tiles_per_xcd,
                                                                  uint4 fillpat) {
  using E = typename vec_of<RB>::type;
  constexpr int CPL = 16 / RB, TT = 1 << TTL, TR = 1 << TRL;
  constexpr int CPL_LOG2 = RB == 16 ? 0 : RB == 8 ? 1 : 2;
  constexpr int JG_LOG2 = TTL - CPL_LOG2, RG_LOG2 = TRL - CPL_LOG2;     // groups per rank / per step
  static_assert(RB == 16 || RB == 8 || RB == 4, "rows of one 16-, 8- or 4-byte vector");
  union Vec { u32x4 v; E e[CPL]; };
  __shared__ TileTables tb;
  __shared__ E stage[TR * (TT + 1)];             // one padding cell per rank: the transposed walk strides over TT + 1 cells
  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) {
    tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= tiles_per_xcd || tile >= Pk.n_tiles) return;
  }
  int64_t r0, t0;
  tile_tables<TTL, TRL>(Pk, Ot, tile, tb, r0, t0);
  __syncthreads();
  const bool padded = (Pk.tile_t_log2 & TILE_FULL_GRID) != 0;
  Vec fillv;
  fillv.v = fill_of<16>(fillpat);
  // cell (rank, j): step t0 + j of the rank's sequence; batch-major row obase[rank] + j, PackedSequence row pboff[j] + r0 + rank
#define RUA_LIVE(rank, j) ((j) < tb.nlive_t[rank] && (rank) < tb.nlive_r[j])
#define RUA_STG(rank, j) stage[(rank) * (TT + 1) + (j)]
  // ---- the batch-major side: a group = CPL consecutive steps of one rank
  auto major_pass = [&](auto&& both, auto&& one) {
    for (int idx = threadIdx.x; idx < (TR << JG_LOG2); idx += RUA_BLOCK) {
      const int rank = idx >> JG_LOG2, j = (idx & ((1 << JG_LOG2) - 1)) * CPL;
      bool all = true;
#pragma unroll
      for (int c = 0; c < CPL; ++c) all = all && RUA_LIVE(rank, j + c);
      if (all) both(rank, j);
      else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) one(rank, j + c, RUA_LIVE(rank, j + c));
      }
    }
  };
  // ---- the PackedSequence's side: a group = CPL consecutive ranks of one step
  auto packed_pass = [&](auto&& both, auto&& one) {
    for (int idx = threadIdx.x; idx < (TT << RG_LOG2); idx += RUA_BLOCK) {
      const int j = idx >> RG_LOG2, rank = (idx & ((1 << RG_LOG2) - 1)) * CPL;
      bool all = true;
#pragma unroll
      for (int c = 0; c < CPL; ++c) all = all && RUA_LIVE(rank + c, j);
      if (all) both(rank, j);
      else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) one(rank + c, j, RUA_LIVE(rank + c, j));
      }
    }
  };
  if (TO_PACK) {
    major_pass(
        [&](int rank, int j) {
          Vec v;
          v.v = *reinterpret_cast<const u32x4_a4*>(src + (tb.obase[rank] + j) * RB);
#pragma unroll
          for (int c = 0; c < CPL; ++c) RUA_STG(rank, j + c) = v.e[c];
        },
        [&](int rank, int j, bool live) {
          if (live) RUA_STG(rank, j) = *reinterpret_cast<const E*>(src + (tb.obase[rank] + j) * RB);
        });
    __syncthreads();
    packed_pass(
        [&](int rank, int j) {
          Vec v;
#pragma unroll
          for (int c = 0; c < CPL; ++c) v.e[c] = RUA_STG(rank + c, j);
          *reinterpret_cast<u32x4_a4*>(dst + (tb.pboff[j] + r0 + rank) * RB) = v.v;
        },
        [&](int rank, int j, bool live) {
          if (live) *reinterpret_cast<E*>(dst + (tb.pboff[j] + r0 + rank) * RB) = RUA_STG(rank, j);
        });
  } else {
    packed_pass(
        [&](int rank, int j) {
          Vec v;
          v.v = *reinterpret_cast<const u32x4_a4*>(src + (tb.pboff[j] + r0 + rank) * RB);
#pragma unroll
          for (int c = 0; c < CPL; ++c) RUA_STG(rank + c, j) = v.e[c];
        },
        [&](int rank, int j, bool live) {
          if (live) RUA_STG(rank, j) = *reinterpret_cast<const E*>(src + (tb.pboff[j] + r0 + rank) * RB);
        });
    __syncthreads();
    major_pass(
        [&](int rank, int j) {
          Vec v;
#pragma unroll
          for (int c = 0; c < CPL; ++c) v.e[c] = RUA_STG(rank, j + c);
          *reinterpret_cast<u32x4_a4*>(dst + (tb.obase[rank] + j) * RB) = v.v;
        },
        [&](int rank, int j, bool live) {
          if (live) {
            *reinterpret_cast<E*>(dst + (tb.obase[rank] + j) * RB) = RUA_STG(rank, j);
          } else if (padded && tb.ofill[rank] >= 0 && t0 + j >= tb.olen[rank] && t0 + j < Ot.T_phys) {
            // a slot of the padded destination that holds no token (see pack_tile_lds_kernel)
            const int64_t t_ = t0 + j;
            const int64_t slot = (Ot.kind == RUA_RIGHT && t_ < Ot.T_log) ? t_ - tb.olen[rank] : t_;
            *reinterpret_cast<E*>(dst + (tb.ofill[rank] + slot) * RB) = fillv.e[0];
          }
        });
  }
#undef RUA_STG
#undef RUA_LIVE
}

// roll / rev INSIDE one PackedSequence at rows of <= 32 bytes, on the same (rank x time) tiles: both sides are in the
// PackedSequence's own order (runs of sixteen ranks), so nothing is transposed and nothing is staged — a cell's source
// row is boff[t'] + rank with t' = the token map of t under the rank's length.  The generic mover's variant for these
// rows (four rows per lane, two 10-step binary searches in lockstep) spends its life in ~20 dependent loads per tile;
// here the chain is tile_start -> (sorted -> lens | boff, bsz of the tile's steps) -> boff[t'] -> payload.
template <int VEC, int TTL>
__global__ __launch_bounds__(RUA_BLOCK) void pack_roll_tile_kernel(rua_layout Pk, int32_t tmap, int64_t targ,
                                                               char* __restrict__ dst, const char* __restrict__ src,
                                                               int64_t row_bytes, int64_t lpr, uint4 fillpat,
                                                               int64_t tiles_per_xcd) {
  using V = typename vec_of<VEC>::type;
  constexpr int TT = 1 << TTL, TR = 16, TILE = TR * TT;
  __shared__ TileTables tb;
  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) {
    tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= tiles_per_xcd || tile >= Pk.n_tiles) return;
  }
  int64_t r0, t0;                                  // (no shift here: both sides are the PackedSequence's own runs, so
  tile_tables<TTL, 4>(Pk, Pk, tile, tb, r0, t0);   //  pboff / pbsz [time] are those of step t0 + time, olen[rank] the length)
  __syncthreads();
  const V fillv = fill_of<VEC>(fillpat);
  const int n_pieces = TILE * (int)lpr;
#pragma unroll 4
  for (int idx = threadIdx.x; idx < n_pieces; idx += RUA_BLOCK) {
    const int pos = idx / (int)lpr, piece = idx - pos * (int)lpr;
    const int rank = pos & (TR - 1), time = pos >> 4;
    const int64_t len = tb.olen[rank];
    if (t0 + time < len && r0 + rank < tb.pbsz[time]) {
      const int64_t t = t0 + time;
      const int64_t ts = apply_tmap(tmap, targ, t, len, len);
      const int64_t drow = tb.pboff[time] + r0 + rank;
      V val = fillv;
      if (ts >= 0 && ts < len) {
        const int64_t srow = Pk.boff[ts] + r0 + rank;
        if (srow < Pk.n_rows) val = ld_row<V, false>(src + srow * row_bytes + (int64_t)piece * VEC);
      }
      if (drow < Pk.n_rows) st_row<V, false>(dst + drow * row_bytes + (int64_t)piece * VEC, val);
    }
  }
}

// [r5] roll INSIDE one PackedSequence at narrow rows, time step by time step.  out[boff[t] + r] = in[boff[ts] + r] with
// ts = (t - s) mod len_r: for every rank that does not wrap at step t, ts = t - s is the SAME step, and those ranks are a
// prefix [0, bsz[max(t, ts)]) — so a step's rows are ONE contiguous run copied from ONE contiguous run, plus the few
// ranks that wrap (those whose sequence ends between the two steps; all ranks of the first |s| steps), which are
// gathered row by row.  A tile = one time step x (16 KiB / row bytes) ranks (the host's table: rua_layout::tile_start
// with TILE_STEP_ROWS set), moved with 16 bytes per lane whatever the row width.  The row kernels resolve every row by
// itself: 5.2 ms for 500 M 8-byte rows (1.5 TB/s).
template <bool NT>
__global__ __launch_bounds__(RUA_BLOCK) void pack_roll_steps_kernel(rua_layout Pk, int64_t shift, char* __restrict__ dst,
                                                                    const char* __restrict__ src, int64_t rb, int trl,
                                                                    int64_t tiles_per_xcd) {
  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) {
    tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= tiles_per_xcd) return;
  }
  if (tile >= Pk.n_tiles) return;
  int64_t t = 0, hi = Pk.n_tchunks;                       // largest t with tile_start[t] <= tile (block-uniform)
  while (hi - t > 1) {
    const int64_t mid = (t + hi) >> 1;
    if (Pk.tile_start[mid] <= tile) t = mid; else hi = mid;
  }
  const int64_t r0 = (tile - Pk.tile_start[t]) << trl;
  int64_t n = Pk.bsz[t] - r0;
  if (n > ((int64_t)1 << trl)) n = (int64_t)1 << trl;
  const int64_t drow0 = Pk.boff[t] + r0;
  if (n <= 0 || drow0 + n > Pk.n_rows) return;
  // the ranks that do not wrap: alive at step ts = t - shift as well (a prefix of the ranks)
  const int64_t ts = t - shift;
  int64_t nd = 0, srow0 = 0;
  if (ts >= 0 && ts < Pk.T) {
    nd = Pk.bsz[ts] - r0;
    if (nd > n) nd = n;
    if (nd < 0) nd = 0;
    srow0 = Pk.boff[ts] + r0;
    if (srow0 + nd > Pk.n_rows) nd = 0;
  }
  if (nd > 0) copy_bytes_any<NT, RUA_BLOCK>(dst + drow0 * rb, src + srow0 * rb, nd * rb, threadIdx.x);
  // the ranks that wrap at this step, one row per thread
  const bool dw4 = (rb & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0;
  for (int64_t i = nd + threadIdx.x; i < n; i += RUA_BLOCK) {
    const int64_t r = r0 + i;
    int64_t lo = 0, hh = Pk.T;                             // len = #{t : bsz[t] > r}
    while (lo < hh) {
      const int64_t mid = (lo + hh) >> 1;
      if (Pk.bsz[mid] > r) lo = mid + 1; else hh = mid;
    }
    const int64_t tsr = apply_tmap(RUA_T_ROLL, shift, t, lo, lo);
    if (tsr < 0 || tsr >= lo) continue;
    const int64_t srow = Pk.boff[tsr] + r;
    if (srow >= Pk.n_rows) continue;
    const char* sp = src + srow * rb;
    char* dp = dst + (drow0 + i) * rb;
    if (dw4) { for (int w = 0; w < (int)(rb >> 2); ++w) reinterpret_cast<uint32_t*>(dp)[w] = reinterpret_cast<const uint32_t*>(sp)[w]; }
    else { for (int w = 0; w < (int)rb; ++w) dp[w] = sp[w]; }
  }
}

static int launch_roll_tiles(int vec, hipStream_t s, const rua_layout& Pk, int32_t tmap, int64_t targ, char* dst,
                             const char* src, int64_t row_bytes, uint4 fp, bool xcd_span) {
  const int64_t per_xcd = xcd_span ? (Pk.n_tiles + 7) / 8 : 0;
  const int64_t grid = xcd_span ? per_xcd * 8 : Pk.n_tiles;
  if (grid > 0x7fffffffLL) return RUA_ERANGE;
  const int64_t lpr = (row_bytes + vec - 1) / vec;
  const int ttl = (Pk.tile_t_log2 & 0xff) == 0 ? 4 : (Pk.tile_t_log2 & 0xff);
  if (ttl < 4 || ttl > 6 || vec != 16) return RUA_EINVAL;
  const dim3 g((unsigned)grid), b(RUA_BLOCK);
  switch (ttl) {
    case 4: hipLaunchKernelGGL((pack_roll_tile_kernel<16, 4>), g, b, 0, s, Pk, tmap, targ, dst, src, row_bytes, lpr, fp, per_xcd); break;
    case 5: hipLaunchKernelGGL((pack_roll_tile_kernel<16, 5>), g, b, 0, s, Pk, tmap, targ, dst, src, row_bytes, lpr, fp, per_xcd); break;
    default: hipLaunchKernelGGL((pack_roll_tile_kernel<16, 6>), g, b, 0, s, Pk, tmap, targ, dst, src, row_bytes, lpr, fp, per_xcd); break;
  }
  return (int)hipGetLastError();
}

// the tile shapes beyond 16 ranks: rows of ONE vector below 16 bytes (see launch_pack_tiles)
static inline bool tile_shape_narrow(int vec, int64_t lpr, int ttl, int trl) {
  return lpr == 1 && ((vec == 8 && trl == 5 && ttl == 6) || (vec == 4 && trl == 6 && ttl == 6) ||
                      (vec == 8 && trl == 4 && ttl == 7) || (vec == 4 && trl == 5 && ttl == 7) ||   // out of a PackedSequence
                      (vec == 2 && trl == 6 && ttl == 7) || (vec == 1 && trl == 7 && ttl == 7));
}
static inline bool tile_shape_exists(int vec, int64_t row_bytes, int code) {
  const int ttl = (code & 0xff) == 0 ? 4 : (code & 0xff), trl = ((code >> 8) & 0xff) == 0 ? 4 : ((code >> 8) & 0xff);
  return (trl == 4 && ttl >= 4 && ttl <= 6) || tile_shape_narrow(vec, (row_bytes + vec - 1) / vec, ttl, trl);
}

template <bool TO_PACK>
static int launch_pack_tiles(int vec, hipStream_t s, const rua_layout& Pk, const rua_layout& Ot, char* dst,
                             const char* src, int64_t row_bytes, bool xcd_span, uint4 fp) {
  const int64_t per_xcd = xcd_span ? (Pk.n_tiles + 7) / 8 : 0;
  const int64_t grid = xcd_span ? per_xcd * 8 : Pk.n_tiles;
  if (grid > 0x7fffffffLL) return RUA_ERANGE;
  const int64_t lpr = (row_bytes + vec - 1) / vec;
  const dim3 g((unsigned)grid), b(RUA_BLOCK);
  const int ttl = (Pk.tile_t_log2 & 0xff) == 0 ? 4 : (Pk.tile_t_log2 & 0xff);      // (0: a caller of ABI <= 3, 16 x 16 tiles)
  const int trl = ((Pk.tile_t_log2 >> 8) & 0xff) == 0 ? 4 : ((Pk.tile_t_log2 >> 8) & 0xff);
  // tile shapes that exist: 16 ranks x 16 / 32 / 64 steps for any vector width, and — [r5] rows of ONE vector below 16
  // bytes — 32 x 64 (8-byte rows), 64 x 64 (4), 64 x 128 (2), 128 x 128 (1): 16 KiB of payload per tile at every width,
  // runs of 128 .. 512 bytes on both sides
  const bool shape16 = trl == 4 && ttl >= 4 && ttl <= 6;
  const bool shape_narrow = tile_shape_narrow(vec, lpr, ttl, trl);
  if (!shape16 && !shape_narrow) return RUA_EINVAL;
  const bool full_grid = (Pk.tile_t_log2 & TILE_FULL_GRID) != 0;
  if (full_grid && (TO_PACK || (Ot.kind != RUA_LEFT && Ot.kind != RUA_RIGHT))) return RUA_EINVAL;
  // the batch-major side's runs start on 128-byte lines (TileTables): R rows per line, if the host built the table for
  // it, the rows divide a line and the payload's own address does not spoil it (its offset inside a line, in rows)
  int R = (int)((Pk.tile_t_log2 >> 16) & 0xff);
  const uintptr_t major = TO_PACK ? (uintptr_t)src : (uintptr_t)dst;
  int64_t phase = 0;
  // ... and only where the batch-major side is the one WRITTEN (P.cat): what costs is a partial-line STORE — the shifted
  // windows trade whole lines on the batch-major side for fragments on the PackedSequence's, so they gain 2-20 % for
  // P.cat and LOSE 6-15 % for the pack, whose packed-side fragments would then be the stores (same-box A/B of both
  // directions at 16 / 32 / 64-byte rows: profiles/r04_tile_shift_ab.txt)
  if (TO_PACK || full_grid || R < 2 || R > TILE_SHIFT_MAX || (R & (R - 1)) != 0 || R * row_bytes != 128 || (major & 127) % (uintptr_t)row_bytes != 0) R = 1;
  else phase = (int64_t)((major & 127) / (uintptr_t)row_bytes);
#ifdef RUA_TILE_NO_SHIFT          // developer A/B build
  R = 1; phase = 0;
#endif
  const size_t lds = (size_t)((((int64_t)1 << (ttl + trl)) + ((int64_t)1 << trl)) * lpr) * vec;   // the staged tile + one padding row per rank
  if (lds > (48u << 10)) return RUA_EINVAL;                            // (the host picks the tile by row width: 32 KiB staged at most)
#define RUA_LAUNCH_T(VEC, TTLV, TRLV) \
  hipLaunchKernelGGL((pack_tile_lds_kernel<VEC, TO_PACK, TTLV, TRLV>), g, b, lds, s, Pk, Ot, dst, src, row_bytes, lpr, per_xcd, R, phase, fp)
#define RUA_LAUNCH(VEC)                                                             \
  switch (ttl) {             /* (32 ranks per tile were measured too at >= 16-byte rows: slower, not instantiated) */ \
    case 4: RUA_LAUNCH_T(VEC, 4, 4); break;                                         \
    case 5: RUA_LAUNCH_T(VEC, 5, 4); break;                                         \
    default: RUA_LAUNCH_T(VEC, 6, 4); break;                                        \
  }
  // 16-byte rows INTO a PackedSequence take the lane-group kernel too (one cell per lane, relative liveness tables: +4 %
  // over the staged kernel, r5r); out of one the staged kernel's line-aligned windows are worth more (-5 % without them)
  if (TO_PACK && vec == 16 && lpr == 1 && trl == 4 && ttl == 6 && !full_grid) {
    hipLaunchKernelGGL((pack_tile_vec_kernel<16, TO_PACK, 6, 4>), g, b, 0, s, Pk, Ot, dst, src, per_xcd, fp);
    return (int)hipGetLastError();
  }
  if (shape_narrow) {
    switch (vec) {
      // (the taller tiles — 16 x 128 at 8 bytes, 32 x 128 at 4 — are what the host asks for OUT of a PackedSequence: the
      // batch-major side is then the one written, and its runs are 1 KiB / 512 B instead of 512 / 256)
      case 8: if (ttl == 7) hipLaunchKernelGGL((pack_tile_vec_kernel<8, TO_PACK, 7, 4>), g, b, 0, s, Pk, Ot, dst, src, per_xcd, fp);
              else hipLaunchKernelGGL((pack_tile_vec_kernel<8, TO_PACK, 6, 5>), g, b, 0, s, Pk, Ot, dst, src, per_xcd, fp);
              break;
      case 4: if (ttl == 7) hipLaunchKernelGGL((pack_tile_vec_kernel<4, TO_PACK, 7, 5>), g, b, 0, s, Pk, Ot, dst, src, per_xcd, fp);
              else hipLaunchKernelGGL((pack_tile_vec_kernel<4, TO_PACK, 6, 6>), g, b, 0, s, Pk, Ot, dst, src, per_xcd, fp);
              break;
      case 2: RUA_LAUNCH_T(2, 7, 6); break;
      default: RUA_LAUNCH_T(1, 7, 7); break;
    }
    return (int)hipGetLastError();
  }
  switch (vec) {
    case 16: RUA_LAUNCH(16); break;
    case 8:  RUA_LAUNCH(8); break;
    case 4:  RUA_LAUNCH(4); break;
    case 2:  RUA_LAUNCH(2); break;
    default: RUA_LAUNCH(1); break;
  }
#undef RUA_LAUNCH
#undef RUA_LAUNCH_T
  return (int)hipGetLastError();
}

static int check_layout(const rua_layout* L, bool is_dst) {
  if (!L) return RUA_EINVAL;
  if (L->B < 0 || L->n_rows < 0) return RUA_EINVAL;
  switch (L->kind) {
    case RUA_CAT:
      if (L->lens && !L->off) return RUA_EINVAL;
      return 0;
    case RUA_LEFT:
    case RUA_RIGHT:
      return L->T_phys >= 0 ? 0 : RUA_EINVAL;
    case RUA_PACK:
      if (L->T > 0 && !L->boff) return RUA_EINVAL;
      return 0;
    case RUA_LIST:
      if (!is_dst) return RUA_EINVAL;
      if (L->n_rows > 0 && !L->tptr) return RUA_EINVAL;
      return 0;
  }
  return RUA_EINVAL;
}

constexpr int NARROW_RPT = 4;
constexpr int NARROW_CH = 8;         // rows in flight per lane of move_rows_narrow_kernel

template <bool SCATTER, bool NT>
static int launch_narrow(int vec, int64_t n_rows, hipStream_t s, const rua_layout& D, const rua_layout& S, int32_t tmap,
                         int64_t targ, char* dst, const char* src, uint4 fp, int64_t pad_row, bool xcd_span) {
  const int64_t per_tile = (int64_t)RUA_BLOCK * NARROW_CH;
  const int64_t ntiles = (n_rows + per_tile - 1) / per_tile;
  const int64_t per_xcd = xcd_span ? (ntiles + 7) / 8 : 0;
  const int64_t grid = xcd_span ? per_xcd * 8 : ntiles;
  if (grid > 0x7fffffffLL) return RUA_ERANGE;
  const dim3 g((unsigned)grid), b(RUA_BLOCK);
#define RUA_LAUNCH_N(VEC) \
  hipLaunchKernelGGL((move_rows_narrow_kernel<VEC, SCATTER, NT, NARROW_CH>), g, b, 0, s, D, S, tmap, targ, dst, src, fp, pad_row, per_xcd)
  switch (vec) {
    case 16: RUA_LAUNCH_N(16); break;
    case 8:  RUA_LAUNCH_N(8); break;
    case 4:  RUA_LAUNCH_N(4); break;
    case 2:  RUA_LAUNCH_N(2); break;
    default: RUA_LAUNCH_N(1); break;
  }
#undef RUA_LAUNCH_N
  return (int)hipGetLastError();
}

template <bool SCATTER, bool NT>
static int launch_move(int vec, int64_t n_rows, hipStream_t s, const rua_layout& D, const rua_layout& S, int32_t tmap,
                       int64_t targ, char* dst, const char* src, int64_t row_bytes, uint4 fp, int64_t pad_row,
                       bool narrow_same_pack, int tile_rows, bool xcd_span, bool tail8 = false) {
  if (tail8) vec = 16;
  const int64_t lpr = (row_bytes + vec - 1) / vec;
  int lp_log2 = 0;
  while ((1 << lp_log2) < lpr && lp_log2 < 6) ++lp_log2;
  const int cpr = (int)((lpr + RUA_WAVE - 1) / RUA_WAVE);
  const dim3 b(MOVE_BLOCK);
  if (narrow_same_pack) {   // vec == 16, gather: see move_rows_kernel<..., RPT>
    const int64_t nt4 = (n_rows + MOVE_TILE * NARROW_RPT - 1) / (MOVE_TILE * NARROW_RPT);
    hipLaunchKernelGGL((move_rows_kernel<16, false, NT, NARROW_RPT>), dim3((unsigned)nt4), b, 0, s, D, S, tmap, targ, dst,
                       src, row_bytes, lpr, lp_log2, cpr, fp, pad_row, (int64_t)0);
    return (int)hipGetLastError();
  }
  const int64_t ntiles = (n_rows + tile_rows - 1) / tile_rows;
  const int64_t per_xcd = xcd_span ? (ntiles + 7) / 8 : 0;
  const int64_t grid = xcd_span ? per_xcd * 8 : ntiles;
  if (grid > 0x7fffffffLL) return RUA_ERANGE;
  const dim3 g((unsigned)grid);
#define RUA_LAUNCH_T(VEC, TR) \
  hipLaunchKernelGGL((move_rows_kernel<VEC, SCATTER, NT, 1, TR>), g, b, 0, s, D, S, tmap, targ, dst, src, row_bytes, lpr, lp_log2, cpr, fp, pad_row, per_xcd)
  // 16-byte rows get every tile size; the narrower vector widths (odd row sizes) a coarser choice
#define RUA_LAUNCH16()                         \
  switch (tile_rows) {                         \
    case 4: RUA_LAUNCH_T(16, 4); break;        \
    case 8: RUA_LAUNCH_T(16, 8); break;        \
    case 16: RUA_LAUNCH_T(16, 16); break;      \
    case 32: RUA_LAUNCH_T(16, 32); break;      \
    case 64: RUA_LAUNCH_T(16, 64); break;      \
    case 128: RUA_LAUNCH_T(16, 128); break;    \
    default: RUA_LAUNCH_T(16, MOVE_TILE); break; \
  }
#define RUA_LAUNCH(VEC)                        \
  switch (tile_rows) {                         \
    case 16: RUA_LAUNCH_T(VEC, 16); break;     \
    case 64: RUA_LAUNCH_T(VEC, 64); break;     \
    default: RUA_LAUNCH_T(VEC, MOVE_TILE); break; \
  }
#define RUA_LAUNCH_TAIL(TR) \
  hipLaunchKernelGGL((move_rows_kernel<16, SCATTER, NT, 1, TR, MOVE_BLOCK, UNROLL, true>), g, b, 0, s, D, S, tmap, targ, dst, src, row_bytes, lpr, lp_log2, cpr, fp, pad_row, per_xcd)
  if (tail8) {
    switch (tile_rows) {
      case 16: RUA_LAUNCH_TAIL(16); break;
      case 64: RUA_LAUNCH_TAIL(64); break;
      default: RUA_LAUNCH_TAIL(MOVE_TILE); break;
    }
    return (int)hipGetLastError();
  }
#undef RUA_LAUNCH_TAIL
  switch (vec) {
    case 16: RUA_LAUNCH16(); break;
    case 8:  RUA_LAUNCH(8); break;
    case 4:  RUA_LAUNCH(4); break;
    case 2:  RUA_LAUNCH(2); break;
    default: RUA_LAUNCH(1); break;
  }
#undef RUA_LAUNCH
#undef RUA_LAUNCH16
#undef RUA_LAUNCH_T
  return (int)hipGetLastError();
}

}  // namespace rua

using namespace rua;

extern "C" int rua_move_rows(const rua_layout* dst, const rua_layout* src, int32_t tmap, int64_t tmap_arg,
                             void* dst_data, const void* src_data, int64_t row_bytes, const void* fill16,
                             int64_t pad_row, int32_t flags, void* stream) {
  int e;
  if ((e = check_layout(dst, true)) != 0) return e;
  if ((e = check_layout(src, false)) != 0) return e;
  if (tmap < RUA_T_SHIFT || tmap > RUA_T_ZERO || row_bytes < 0) return RUA_EINVAL;
  if (dst->n_rows == 0 || row_bytes == 0) return 0;
  if (!dst_data || !src_data) return RUA_EINVAL;
  if (pad_row < -1 || pad_row >= src->n_rows) return RUA_EINVAL;

  // widest power-of-two access that divides the row size and both base addresses
  const uint64_t mix = (uint64_t)row_bytes | (uint64_t)(uintptr_t)dst_data | (uint64_t)(uintptr_t)src_data | 16u;
  const int vec = (int)(mix & (~mix + 1));

  uint4 fp = make_uint4(0, 0, 0, 0);
  if (fill16) {
    const uint32_t* f = (const uint32_t*)fill16;
    fp = make_uint4(f[0], f[1], f[2], f[3]);
  }
  hipStream_t s = (hipStream_t)stream;
  // narrow rows between a PackedSequence and a batch-major layout: (rank x time) tiles
  const int span_flags = RUA_MOVE_XCD_SPAN_ON | RUA_MOVE_XCD_SPAN_OFF | RUA_MOVE_NO_TAIL8 | RUA_MOVE_NO_NARROW;
  if ((flags & ~span_flags) == 0 && tmap == RUA_T_SHIFT && tmap_arg == 0 && row_bytes <= TILE_MAX_ROW_BYTES) {
    const bool to_pack = dst->kind == RUA_PACK && (src->kind == RUA_CAT || src->kind == RUA_LEFT || src->kind == RUA_RIGHT);
    // [r5] a padded destination takes the tiles too when the caller asks for the FULL (sequence x step) grid
    // (rua_layout::tile_t_log2 bit 24): every slot is then written, tokens and fill, in the one pass
    const bool full_grid = src->kind == RUA_PACK && (src->tile_t_log2 & TILE_FULL_GRID) != 0;
    const bool from_pack = src->kind == RUA_PACK && (dst->kind == RUA_CAT || (full_grid && pad_row < 0 &&
                                                     (dst->kind == RUA_LEFT || dst->kind == RUA_RIGHT)));
    const rua_layout* pk = to_pack ? dst : src;
    // (a table built for a tile shape this vector width has no kernel for — rows of 8 bytes at a base that is only
    // 4-byte aligned, say — is simply not used)
    if ((to_pack || from_pack) && (pk->tile_start || full_grid) && pk->bsz && pk->n_tiles > 0 && pk->boff &&
        (!full_grid || (!to_pack && dst->kind != RUA_CAT)) && !(pk->tile_t_log2 & TILE_STEP_ROWS) &&
        tile_shape_exists(vec, row_bytes, pk->tile_t_log2)) {
      bool span = pk->n_tiles >= (to_pack || full_grid ? MOVE_SPAN_MIN_TILES : TILE_SPAN_MIN_TILES_FROM_PACK);
      if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
      if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
      return to_pack ? launch_pack_tiles<true>(vec, s, *dst, *src, (char*)dst_data, (const char*)src_data, row_bytes, span, fp)
                     : launch_pack_tiles<false>(vec, s, *src, *dst, (char*)dst_data, (const char*)src_data, row_bytes, span, fp);
    }
  }
  const bool big = (double)dst->n_rows * (double)row_bytes >= (double)(512ll << 20);
  const bool nt = (flags & RUA_MOVE_NT_ON) ? true : (flags & RUA_MOVE_NT_OFF) ? false : big;
  char* d = (char*)dst_data;
  const char* c = (const char*)src_data;
  // Launch geometry (profiles/r02_copy_probe.txt, r02_mover_geometry.txt): HBM rewards a launch whose
  // in-flight addresses form a small window sweeping the destination in order, so a workgroup takes only ~16 KiB
  // of destination rows (workgroups are dispatched in blockIdx order), and on big launches every XCD sweeps ONE
  // contiguous span of the destination instead of every eighth tile.  At the north-star shape (1 KiB rows):
  // 256-row tiles 5.6 TB/s -> 16-row tiles 6.15 TB/s for C->P, 5.3 -> 6.1 for P->C.
  int tile_rows = MOVE_TILE;
  for (int64_t tb = MOVE_TILE * row_bytes; tile_rows > 4 && tb > MOVE_TILE_BYTES; tb >>= 1) tile_rows >>= 1;
  const int tsel = (flags >> 4) & 0xf;               // developer override: RUA_MOVE_TILE_LOG2 / RUA_MOVE_XCD_SPAN_*
  if (tsel >= 2 && tsel <= 8) tile_rows = 1 << tsel;
  if (tile_rows > MOVE_BLOCK) tile_rows = MOVE_BLOCK;
  // rows that are a multiple of 8 but not of 16 bytes: 16-byte lanes at 8-byte-aligned addresses + an 8-byte tail
  // (RUA_MOVE_NO_TAIL8, a developer flag, keeps the 8-byte lanes for A/B runs)
  // Only rows that really END in an 8-byte piece: vec == 8 also comes from a base pointer that is only 8-byte aligned
  // under rows of a multiple of 16 bytes (a view at a storage offset), and those have no tail — they keep 8-byte lanes.
  const bool tail8 = vec == 8 && (row_bytes & 15) == 8 && row_bytes >= 24 && !(flags & RUA_MOVE_NO_TAIL8);
  if (vec != 16) tile_rows = tile_rows <= 16 ? 16 : tile_rows <= 64 ? 64 : MOVE_TILE;
  const int64_t nr = dst->n_rows;
  const bool padded_dst = dst->kind == RUA_LEFT || dst->kind == RUA_RIGHT;
  bool xcd_span = (nr + tile_rows - 1) / tile_rows >= (padded_dst ? MOVE_SPAN_MIN_TILES : MOVE_SPAN_MIN_TILES_DENSE);
  if (flags & RUA_MOVE_XCD_SPAN_ON) xcd_span = true;
  if (flags & RUA_MOVE_XCD_SPAN_OFF) xcd_span = false;
  // [r5] a roll inside one PackedSequence, time step by time step, when the caller handed over the table of one-step tiles
  if (dst->kind == RUA_PACK && src->kind == RUA_PACK && (dst->tile_t_log2 & TILE_STEP_ROWS) && dst->tile_start && dst->bsz &&
      dst->boff && dst->boff == src->boff && dst->sorted == src->sorted && dst->len_add == 0 && src->len_add == 0 &&
      dst->T == src->T && dst->T > 0 && dst->n_tchunks == dst->T && dst->n_tiles > 0 && tmap == RUA_T_ROLL && pad_row < 0 &&
      (flags & ~(RUA_MOVE_XCD_SPAN_ON | RUA_MOVE_XCD_SPAN_OFF)) == 0) {
    bool span = dst->n_tiles >= MOVE_SPAN_MIN_TILES_DENSE;
    if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
    if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
    const int64_t per_xcd = span ? (dst->n_tiles + 7) / 8 : 0;
    const int64_t grid = span ? per_xcd * 8 : dst->n_tiles;
    if (grid > 0x7fffffffLL) return RUA_ERANGE;
    const int trl = (dst->tile_t_log2 >> 8) & 0xff;
    if (nt) hipLaunchKernelGGL(pack_roll_steps_kernel<true>, dim3((unsigned)grid), dim3(RUA_BLOCK), 0, s, *dst, tmap_arg, d, c, row_bytes, trl, per_xcd);
    else hipLaunchKernelGGL(pack_roll_steps_kernel<false>, dim3((unsigned)grid), dim3(RUA_BLOCK), 0, s, *dst, tmap_arg, d, c, row_bytes, trl, per_xcd);
    return (int)hipGetLastError();
  }
  // [r5] narrow rows between two batch-major layouts: the segmented memcpy (seq_copy_kernel)
  {
    const bool major_d = dst->kind == RUA_CAT || dst->kind == RUA_LEFT || dst->kind == RUA_RIGHT;
    const bool major_s = src->kind == RUA_CAT || src->kind == RUA_LEFT || src->kind == RUA_RIGHT;
    const bool same_lens = dst->lens == src->lens && dst->len_add == src->len_add;
    // the longest sequence either side can hold: the storage's own T for the padded layouts, the caller's hint for a
    // CattedSequence (rua_layout::T_log, 0 = unknown).  One wave walks a sequence whole: none may be a large share
    const int64_t long_d = dst->kind == RUA_CAT ? dst->T_log : dst->T_phys;
    const int64_t long_s = src->kind == RUA_CAT ? src->T_log : src->T_phys;
    const int64_t longest = long_d > long_s ? long_d : long_s;
    const bool balanced = long_d > 0 && long_s > 0 && dst->B >= 4096 && longest <= nr / 1024;
    // (rows wider than 64 bytes that are not a multiple of 16 were tried here too: one wave per sequence streams at the
    // walk's 4.5 - 5.4 TB/s whatever the alignment, below what tiles in destination order give: move_rows_span_kernel)
    if (major_d && major_s && !(flags & (RUA_MOVE_SCATTER | RUA_MOVE_NO_NARROW)) && tsel == 0 && pad_row < 0 &&
        row_bytes <= 64 && dst->B == src->B && (tmap == RUA_T_SHIFT || (tmap == RUA_T_ROLL && same_lens)) && balanced &&
        (dst->kind != RUA_CAT || dst->off || !dst->lens) && (src->kind != RUA_CAT || src->off || !src->lens)) {
      // sequences per wave: about 16 KiB of tokens (8 at narrow rows, 1 from a few hundred bytes per row up)
      const int64_t seq_bytes = nr / (dst->B > 0 ? dst->B : 1) * row_bytes;
      const int spw = seq_bytes >= 8192 ? 1 : seq_bytes >= 4096 ? 2 : seq_bytes >= 2048 ? 4 : SEQ_PER_WAVE;
      const int64_t waves = (dst->B + spw - 1) / spw;
      const int64_t grid = (waves + RUA_WAVES_PER_BLOCK - 1) / RUA_WAVES_PER_BLOCK;
      if (grid > 0x7fffffffLL) return RUA_ERANGE;
      if (nt) hipLaunchKernelGGL(seq_copy_kernel<true>, dim3((unsigned)grid), dim3(RUA_BLOCK), 0, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, spw);
      else hipLaunchKernelGGL(seq_copy_kernel<false>, dim3((unsigned)grid), dim3(RUA_BLOCK), 0, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, spw);
      return (int)hipGetLastError();
    }
  }
  // [r5] rows of one vector (1-D payloads: 1 .. 16 bytes): a lane per row from resolution to store, eight rows in flight
  // per lane (move_rows_narrow_kernel).  The roll tiles below keep 16-byte rows inside one PackedSequence.
  const bool one_vec = row_bytes == vec && !(flags & RUA_MOVE_NO_NARROW) && tsel == 0;
  if (one_vec) {
    const int64_t per_tile = (int64_t)RUA_BLOCK * NARROW_CH;
    bool span = (nr + per_tile - 1) / per_tile >= (padded_dst ? MOVE_SPAN_MIN_TILES : MOVE_SPAN_MIN_TILES_DENSE);
    if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
    if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
    const bool roll_tiles = vec == 16 && dst->kind == RUA_PACK && src->kind == RUA_PACK && dst->tile_start && dst->bsz &&
                            !(dst->tile_t_log2 & TILE_STEP_ROWS) &&
                            dst->boff == src->boff && !(flags & RUA_MOVE_SCATTER);
    if (!roll_tiles) {
      if (flags & RUA_MOVE_SCATTER)
        return nt ? launch_narrow<true, true>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, fp, pad_row, span)
                  : launch_narrow<true, false>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, fp, pad_row, span);
      return nt ? launch_narrow<false, true>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, fp, pad_row, span)
                : launch_narrow<false, false>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, fp, pad_row, span);
    }
  }
  if (flags & RUA_MOVE_SCATTER)
    return nt ? launch_move<true, true>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, false, tile_rows, xcd_span, tail8)
              : launch_move<true, false>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, false, tile_rows, xcd_span, tail8);
  // roll / rev inside ONE PackedSequence with rows of at most 32 B (3.1 -> 3.9 TB/s; no gain at 64 B): RPT rows per lane (the kernel's `same_pack` test)
  const bool narrow_same_pack = vec == 16 && row_bytes <= 32 && dst->kind == RUA_PACK && src->kind == RUA_PACK &&
                                dst->bsz && dst->boff && dst->boff == src->boff && dst->sorted == src->sorted &&
                                dst->len_add == 0 && src->len_add == 0 && dst->T == src->T && dst->T > 0;
  // ... and on the (rank x time) tiles when the caller handed the tile table over (pack_roll_tile_kernel)
  if (narrow_same_pack && dst->tile_start && !(dst->tile_t_log2 & TILE_STEP_ROWS) && dst->n_tiles > 0 && dst->lens && dst->sorted &&
      pad_row < 0 && (flags & ~(RUA_MOVE_XCD_SPAN_ON | RUA_MOVE_XCD_SPAN_OFF)) == 0 &&
      (tmap == RUA_T_ROLL || tmap == RUA_T_REV_S || tmap == RUA_T_REV_D || tmap == RUA_T_SHIFT)) {
    bool span = dst->n_tiles >= MOVE_SPAN_MIN_TILES;
    if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
    if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
    return launch_roll_tiles(vec, s, *dst, tmap, tmap_arg, d, c, row_bytes, fp, span);
  }
  // [r5] a gather of rows that are a multiple of 4 but not of 16 bytes: the destination tile goes through LDS and leaves as
  // one aligned span (move_rows_span_kernel)
  // ([r5, late] and of rows that ARE a multiple of 16 bytes but not of a 128-byte line — 2 000-byte rows, H = 1 000 in
  // bf16: the row mover's 16-byte lanes are aligned there, but every wave instruction straddles one line more than it
  // fills and every row boundary is two partial lines; as a span the tile is stored in whole lines)
  const bool off16 = (row_bytes & 15) != 0 && (row_bytes & 3) == 0;
  // (from 1 KiB up: at 640 / 656-byte rows the row mover is level or ahead — 5.41 / 5.40 against 5.36 / 5.00 TB/s for the
  // pack, profiles/r05_span16_ab.txt)
  const bool off128 = (row_bytes & 15) == 0 && (row_bytes & 127) != 0 && row_bytes > 1024;
  if ((off16 || off128) && row_bytes >= 64 && row_bytes <= SPAN_TILE_BYTES / 2 &&
      ((uintptr_t)dst_data & 15) == 0 && ((uintptr_t)src_data & 15) == 0 && tsel == 0 && !(flags & RUA_MOVE_NO_TAIL8)) {
    // rows per tile: a multiple of 2 (rows of 8 mod 16 bytes) or 4 (4 / 12 mod 16), so that every tile starts on a
    // 16-byte boundary (any number of rows of whole vectors: only the two ends of a tile are partial lines then); as many
    // as fit 16 KiB of LDS (16 rows of 1 000 bytes, 8 of 2 000), one resolving lane each
    int step = (row_bytes & 7) ? 4 : 2;
    if (off128) step = 1;
    int trows = (int)(SPAN_TILE_BYTES / row_bytes) / step * step;
    if (trows > RUA_BLOCK) trows = RUA_BLOCK / step * step;
    if (trows > 0) {        // (a line-aligned tile of some widths would not fit the 16 KiB: the row mover takes those)
      const int64_t ntiles = (nr + trows - 1) / trows;
      bool span = ntiles >= (padded_dst ? MOVE_SPAN_MIN_TILES : MOVE_SPAN_MIN_TILES_DENSE);
      if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
      if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
      const int64_t per_xcd = span ? (ntiles + 7) / 8 : 0;
      const int64_t grid = span ? per_xcd * 8 : ntiles;
      if (grid > 0x7fffffffLL) return RUA_ERANGE;
      if (nt) hipLaunchKernelGGL(move_rows_span_kernel<true>, dim3((unsigned)grid), dim3(RUA_BLOCK), 0, s, *dst, *src, tmap, tmap_arg, d, c, (int)row_bytes, trows, fp, pad_row, per_xcd);
      else hipLaunchKernelGGL(move_rows_span_kernel<false>, dim3((unsigned)grid), dim3(RUA_BLOCK), 0, s, *dst, *src, tmap, tmap_arg, d, c, (int)row_bytes, trows, fp, pad_row, per_xcd);
      return (int)hipGetLastError();
    }
  }
  return nt ? launch_move<false, true>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, narrow_same_pack, tile_rows, xcd_span, tail8)
            : launch_move<false, false>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, narrow_same_pack, tile_rows, xcd_span, tail8);
}
