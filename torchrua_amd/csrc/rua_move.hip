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
      const int64_t j = tile0 + i;
      const bool mine = i < nrows;
      int64_t b = 0, t = 0, other = -1;
      bool token = false;
      bool resolved = false;
      bool done = false;
      if (D.kind == RUA_PACK && D.T > 0 && D.boff) {
        int64_t bt;
        const bool ok = coop_resolve([&](int64_t k) { return D.boff[k]; }, D.T, tile0 + w0, nw, lane1, t, bt);
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
            s_ld[i] = other;
            s_st[i] = j;
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
        const bool ok = coop_resolve([&](int64_t k) { return cat_off(D, k); }, D.B, tile0 + w0, nw, lane1, b, ob);
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
// Narrow rows (<= 128 B): C/L/R <-> P is a transpose of the (rank, time) plane — C is contiguous along
// time, P along rank — and a destination-row-major tile makes the other side a one-row-granular gather:
// a 32-byte row out of every 128-byte line, i.e. 4x over-fetch (measured 1.3 TB/s at 32-byte rows).
// Here a tile is TR ranks x TT time steps, so BOTH sides move 16-row runs, and phase 2 walks it in 4x4
// sub-tiles.  (At >= 1 KiB rows the same tiling changes nothing — a row already fills whole lines — so
// wide rows keep the generic kernel.)  Phase 1 needs no search: rank r is live at time t iff r < bsz[t].
constexpr int TR_MAX = 32;        // a tile is (1 << TRL) ranks x (1 << TTL) time steps, TRL = 4 .. 5, TTL = 4 .. 6 (rua_layout::tile_t_log2)
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
  int64_t pboff[64 + TILE_SHIFT_MAX];      // first PackedSequence row of time step t0 - (R - 1) + k
  int64_t pbsz[64 + TILE_SHIFT_MAX];       // sequences alive at that step (0 before 0 and past T)
};

template <int TTL, int TRL>
__device__ __forceinline__ void tile_tables(const rua_layout& Pk, const rua_layout& Ot, int64_t tile, TileTables& tb,
                                            int64_t& r0_out, int64_t& t0_out, int R = 1, int64_t phase = 0) {
  constexpr int TT = 1 << TTL, TR = 1 << TRL;
  // which time chunk does this tile belong to?  largest c with tile_start[c] <= tile.  Up to 64 chunks
  // every lane loads one entry and a ballot counts them: ONE load instead of a six-step chain of dependent ones
  int64_t lo = 0, hi = Pk.n_tchunks;
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
  const int64_t t0 = lo * TT;
  const int64_t r0 = (tile - Pk.tile_start[lo]) * TR;
  r0_out = r0;
  t0_out = t0;
  const int i = threadIdx.x;
  if (i < TR) {                                   // wave 0, lanes 0..15: the ranks
    const int64_t r = r0 + i;
    int64_t base = 0, len = 0;
    if (r < Pk.B) {
      const int64_t b = Pk.sorted ? Pk.sorted[r] : r;
      if (b >= 0 && b < Ot.B) {
        len = seq_len(Ot, b);
        base = token_to_row(Ot, b, 0, len);
      }
    }
    const int s = R > 1 ? (int)((base + phase) & (int64_t)(R - 1)) : 0;
    tb.shift[i] = s;
    tb.obase[i] = base - s + t0;                  // batch-major row of the rank's window start (a whole-line boundary)
    tb.olen[i] = len;
  } else if (i >= RUA_WAVE && i < RUA_WAVE + TT + R - 1) {   // waves 1..: the time steps t0 - (R - 1) .. t0 + TT - 1
    const int k = i - RUA_WAVE;
    const int64_t t = t0 - (R - 1) + k;
    const bool ok = t >= 0 && t < Pk.T;
    tb.pboff[k] = ok ? Pk.boff[t] : 0;
    tb.pbsz[k] = ok ? Pk.bsz[t] : 0;
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
                                                              int64_t lpr, int64_t tiles_per_xcd, int R, int64_t phase) {
  using V = typename vec_of<VEC>::type;
  constexpr int TT = 1 << TTL, TR = 1 << TRL, TILE = TR * TT;
  static_assert(TT <= 64 && TR <= TR_MAX && RUA_BLOCK >= RUA_WAVE + TT + TILE_SHIFT_MAX,
                "one lane per rank in wave 0, per time step in waves 1..");
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
#pragma unroll 4
    for (int idx = threadIdx.x; idx < n_major; idx += RUA_BLOCK) {
      const int pos = idx / (int)lpr, piece = idx - pos * (int)lpr;
      const int rank = pos >> TTL, j = pos & (TT - 1), k = j - tb.shift[rank] + R - 1;
      RUA_CELL(rank, j, k, orow, prow, live);
      (void)prow;
      if (live) st_row<V, false>(dst + orow * row_bytes + (int64_t)piece * VEC, stage[RUA_SLOT(rank, j, piece)]);
    }
  }
#undef RUA_CELL
#undef RUA_SLOT
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

template <bool TO_PACK>
static int launch_pack_tiles(int vec, hipStream_t s, const rua_layout& Pk, const rua_layout& Ot, char* dst,
                             const char* src, int64_t row_bytes, bool xcd_span) {
  const int64_t per_xcd = xcd_span ? (Pk.n_tiles + 7) / 8 : 0;
  const int64_t grid = xcd_span ? per_xcd * 8 : Pk.n_tiles;
  if (grid > 0x7fffffffLL) return RUA_ERANGE;
  const int64_t lpr = (row_bytes + vec - 1) / vec;
  const dim3 g((unsigned)grid), b(RUA_BLOCK);
  const int ttl = (Pk.tile_t_log2 & 0xff) == 0 ? 4 : (Pk.tile_t_log2 & 0xff);      // (0: a caller of ABI <= 3, 16 x 16 tiles)
  const int trl = ((Pk.tile_t_log2 >> 8) & 0xff) == 0 ? 4 : ((Pk.tile_t_log2 >> 8) & 0xff);
  if (ttl < 4 || ttl > 6 || trl != 4) return RUA_EINVAL;
  // the batch-major side's runs start on 128-byte lines (TileTables): R rows per line, if the host built the table for
  // it, the rows divide a line and the payload's own address does not spoil it (its offset inside a line, in rows)
  int R = (int)((Pk.tile_t_log2 >> 16) & 0xff);
  const uintptr_t major = TO_PACK ? (uintptr_t)src : (uintptr_t)dst;
  int64_t phase = 0;
  // ... and only where the batch-major side is the one WRITTEN (P.cat): what costs is a partial-line STORE — the shifted
  // windows trade whole lines on the batch-major side for fragments on the PackedSequence's, so they gain 2-20 % for
  // P.cat and LOSE 6-15 % for the pack, whose packed-side fragments would then be the stores (same-box A/B of both
  // directions at 16 / 32 / 64-byte rows: profiles/r04_tile_shift_ab.txt)
  if (TO_PACK || R < 2 || R > TILE_SHIFT_MAX || (R & (R - 1)) != 0 || R * row_bytes != 128 || (major & 127) % (uintptr_t)row_bytes != 0) R = 1;
  else phase = (int64_t)((major & 127) / (uintptr_t)row_bytes);
#ifdef RUA_TILE_NO_SHIFT          // developer A/B build
  R = 1; phase = 0;
#endif
  const size_t lds = (size_t)((((int64_t)1 << (ttl + trl)) + ((int64_t)1 << trl)) * lpr) * vec;   // the staged tile + one padding row per rank
  if (lds > (48u << 10)) return RUA_EINVAL;                            // (the host picks the tile by row width: 32 KiB staged at most)
#define RUA_LAUNCH_T(VEC, TTLV, TRLV) \
  hipLaunchKernelGGL((pack_tile_lds_kernel<VEC, TO_PACK, TTLV, TRLV>), g, b, lds, s, Pk, Ot, dst, src, row_bytes, lpr, per_xcd, R, phase)
#define RUA_LAUNCH(VEC)                                                             \
  switch (ttl) {             /* (32 ranks per tile were measured too: slower at every width, not instantiated) */ \
    case 4: RUA_LAUNCH_T(VEC, 4, 4); break;                                         \
    case 5: RUA_LAUNCH_T(VEC, 5, 4); break;                                         \
    default: RUA_LAUNCH_T(VEC, 6, 4); break;                                        \
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
  const int span_flags = RUA_MOVE_XCD_SPAN_ON | RUA_MOVE_XCD_SPAN_OFF | RUA_MOVE_NO_TAIL8;
  if ((flags & ~span_flags) == 0 && tmap == RUA_T_SHIFT && tmap_arg == 0 && row_bytes <= TILE_MAX_ROW_BYTES) {
    const bool to_pack = dst->kind == RUA_PACK && (src->kind == RUA_CAT || src->kind == RUA_LEFT || src->kind == RUA_RIGHT);
    const bool from_pack = src->kind == RUA_PACK && dst->kind == RUA_CAT;   // padded destinations need the fill pass
    const rua_layout* pk = to_pack ? dst : src;
    if ((to_pack || from_pack) && pk->tile_start && pk->bsz && pk->n_tiles > 0 && pk->boff) {
      bool span = pk->n_tiles >= (to_pack ? MOVE_SPAN_MIN_TILES : TILE_SPAN_MIN_TILES_FROM_PACK);
      if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
      if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
      return to_pack ? launch_pack_tiles<true>(vec, s, *dst, *src, (char*)dst_data, (const char*)src_data, row_bytes, span)
                     : launch_pack_tiles<false>(vec, s, *src, *dst, (char*)dst_data, (const char*)src_data, row_bytes, span);
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
  if (flags & RUA_MOVE_SCATTER)
    return nt ? launch_move<true, true>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, false, tile_rows, xcd_span, tail8)
              : launch_move<true, false>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, false, tile_rows, xcd_span, tail8);
  // roll / rev inside ONE PackedSequence with rows of at most 32 B (3.1 -> 3.9 TB/s; no gain at 64 B): RPT rows per lane (the kernel's `same_pack` test)
  const bool narrow_same_pack = vec == 16 && row_bytes <= 32 && dst->kind == RUA_PACK && src->kind == RUA_PACK &&
                                dst->bsz && dst->boff && dst->boff == src->boff && dst->sorted == src->sorted &&
                                dst->len_add == 0 && src->len_add == 0 && dst->T == src->T && dst->T > 0;
  // ... and on the (rank x time) tiles when the caller handed the tile table over (pack_roll_tile_kernel)
  if (narrow_same_pack && dst->tile_start && dst->n_tiles > 0 && dst->lens && dst->sorted &&
      pad_row < 0 && (flags & ~(RUA_MOVE_XCD_SPAN_ON | RUA_MOVE_XCD_SPAN_OFF)) == 0 &&
      (tmap == RUA_T_ROLL || tmap == RUA_T_REV_S || tmap == RUA_T_REV_D || tmap == RUA_T_SHIFT)) {
    bool span = dst->n_tiles >= MOVE_SPAN_MIN_TILES;
    if (flags & RUA_MOVE_XCD_SPAN_ON) span = true;
    if (flags & RUA_MOVE_XCD_SPAN_OFF) span = false;
    return launch_roll_tiles(vec, s, *dst, tmap, tmap_arg, d, c, row_bytes, fp, span);
  }
  return nt ? launch_move<false, true>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, narrow_same_pack, tile_rows, xcd_span, tail8)
            : launch_move<false, false>(vec, nr, s, *dst, *src, tmap, tmap_arg, d, c, row_bytes, fp, pad_row, narrow_same_pack, tile_rows, xcd_span, tail8);
}
