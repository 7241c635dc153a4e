// rua_dev.h — device-side helpers shared by the gfx950 kernels of librua_hip.so.
// Row-map closed forms follow SURVEY.md §3 / the reference's core/get.py; see include/rua.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rua.h"

#define RUA_WAVE 64          // gfx950 wavefront
#define RUA_BLOCK 256        // 4 waves per workgroup everywhere
#define RUA_WAVES_PER_BLOCK (RUA_BLOCK / RUA_WAVE)

namespace rua {

// The reducer's wave-team rule (seg_reduce_team_kernel), in ONE place: the launcher (rua_reduce_impl.h) and the host
// planner that prices long-sequence splitting (rua_reduce_team_waves -> _meta.reduce_split_rows) both call it.
// Vector path only (16-byte lanes, rows up to 1 KiB): a team of 2 or 4 waves shares a unit when an average unit holds
// at least 4 row groups per wave and there are at most `REDUCE_TEAM_MAX_UNITS` units.
constexpr int REDUCE_UNROLL_T = 8;                      // rows in flight per wave (RUA_UNROLL_T overrides it in A/B builds)
constexpr int64_t REDUCE_TEAM_MAX_UNITS = 16384;        // beyond this one wave per unit keeps the chip balanced by itself
inline int reduce_team_waves(int64_t n_rows, int64_t B, int lp_log2, int64_t units, int unroll_t = REDUCE_UNROLL_T) {
  if (units <= 0 || units > REDUCE_TEAM_MAX_UNITS) return 1;
  const int64_t rows_per_group = (int64_t)(RUA_WAVE >> lp_log2) * unroll_t;
  const int64_t groups = n_rows / (B > 0 ? B : 1) / rows_per_group;       // row groups of an average unit
  return groups >= 4 * 4 ? 4 : groups >= 4 * 2 ? 2 : 1;
}

__device__ __forceinline__ int64_t seq_len(const rua_layout& L, int64_t b) {
  return (L.lens ? L.lens[b] : 0) + L.len_add;
}

// exclusive offset of sequence b in a CAT-ordered enumeration
__device__ __forceinline__ int64_t cat_off(const rua_layout& L, int64_t b) {
  return (L.off ? L.off[b] : 0) + b * L.len_add;
}

// j / n for 0 <= j, 0 < n: a 64-bit division is ~150 VALU instructions on gfx950 (no hardware divider); row numbers and
// row counts below 2^32 — every realistic storage — divide in 32 bits (~25).  One lane per ROW computes these, so at
// narrow rows (a 1-D payload moves 8 bytes per row) the division was most of the kernel.
__device__ __forceinline__ int64_t div_rows(int64_t j, int64_t n) {
  if (((uint64_t)j | (uint64_t)n) >> 32) return j / n;
  return (int64_t)((uint32_t)j / (uint32_t)n);
}

// largest b in [0, B) with cat_off(b) <= j   (requires cat_off(0) <= j)
__device__ __forceinline__ int64_t search_cat(const rua_layout& L, int64_t j) {
  int64_t lo = 0, hi = L.B;
  if (!L.off) {  // constant length: closed form
    int64_t n = L.len_add;
    return n > 0 ? div_rows(j, n) : 0;
  }
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (cat_off(L, mid) <= j) lo = mid; else hi = mid;
  }
  return lo;
}

// largest t in [0, T) with boff[t] <= j
__device__ __forceinline__ int64_t search_boff(const int64_t* __restrict__ boff, int64_t T, int64_t j) {
  int64_t lo = 0, hi = T;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (boff[mid] <= j) lo = mid; else hi = mid;
  }
  return lo;
}

// storage row j of layout D -> token (b, t); returns false for a padding row
__device__ __forceinline__ bool row_to_token(const rua_layout& D, int64_t j, int64_t& b, int64_t& t) {
  switch (D.kind) {
    case RUA_CAT:
      b = search_cat(D, j);
      t = j - cat_off(D, b);
      return true;
    case RUA_PACK: {
      t = search_boff(D.boff, D.T, j);
      int64_t r = j - D.boff[t];
      if (r < 0 || r >= D.B) return false;          // batch_sizes inconsistent with the storage
      b = D.sorted ? D.sorted[r] : r;
      return b >= 0 && b < D.B;                     // a corrupt sorted_indices must not index out of range
    }
    case RUA_LEFT:
      b = div_rows(j, D.T_phys);
      t = j - b * D.T_phys;
      return t < seq_len(D, b);
    case RUA_RIGHT: {
      b = div_rows(j, D.T_phys);
      int64_t len = seq_len(D, b);
      t = (j - b * D.T_phys) - (D.T_log - len);
      return t >= 0 && t < len;
    }
    case RUA_LIST:
      b = D.bptr ? D.bptr[j] : 0;   // bptr == NULL: one sequence, tptr = flat rows
      t = D.tptr[j];
      return true;
  }
  return false;
}

// token (b, t) -> storage row of layout S
__device__ __forceinline__ int64_t token_to_row(const rua_layout& S, int64_t b, int64_t t, int64_t slen) {
  switch (S.kind) {
    case RUA_CAT:   return cat_off(S, b) + t;
    case RUA_PACK:  return S.boff[t] + (S.unsorted ? S.unsorted[b] : b);
    case RUA_LEFT:  return b * S.T_phys + t;
    case RUA_RIGHT: return b * S.T_phys + (S.T_log - slen) + t;
  }
  return -1;
}

__device__ __forceinline__ int64_t apply_tmap(int32_t tmap, int64_t arg, int64_t t, int64_t slen, int64_t dlen) {
  switch (tmap) {
    case RUA_T_SHIFT: return t + arg;
    case RUA_T_ROLL: {
      if (slen <= 0) return -1;
      int64_t m = t - arg;            // 0 <= t < slen: a shift shorter than the sequence wraps at most once
      if (m < 0) m += slen; else if (m >= slen) m -= slen;
      if (m < 0 || m >= slen) {       // |shift| >= length: the full remainder (C remainder: sign of dividend)
        m = (t - arg) % slen;
        if (m < 0) m += slen;
      }
      return m;
    }
    case RUA_T_REV_S: return slen - 1 - t;
    case RUA_T_REV_D: return dlen - 1 - t;
    case RUA_T_ZERO:  return 0;
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------
// Resolving a run of consecutive storage rows is a latency chain, not work: a plain binary search of `boff` / `off` is 9-17 DEPENDENT
// L2 loads (3-6 us) in front of ~2.5 us of payload traffic, and the workgroup's bytes are not in flight meanwhile.
// Here the wave searches TOGETHER: 64 samples per step (a 64-ary search: 2 dependent loads for T <= 1 024, 3 for
// B <= 65 536), and the last step is a contiguous 64-entry window from which every row of the wave reads its own
// answer (consecutive rows resolve to the same or the next few entries).
//   f        non-decreasing over [0, n_total), f(0) <= j0
//   returns  for lane i < nw: k = largest index with f(k) <= j0 + i, fk = f(k); false = the window ran out for
//            this lane (more than ~48 boundaries inside the wave's rows: runs of zero-length sequences) -> the
//            caller falls back to its own binary search.  ALL 64 lanes must call.
// part 1: the 64-ary search for row j0 and the window f(lo .. lo + 63) that starts at or just before its answer
template <typename F>
__device__ __forceinline__ void coop_window(F f, int64_t n_total, int64_t j0, int lane, int64_t& lo, int64_t& W) {
  constexpr int64_t BIG = 0x7fffffffffffffffLL;
  int64_t n = n_total;
  lo = 0;
  while (n > 16) {                                   // wave-uniform
    const int64_t step = (n + 63) >> 6;
    const int64_t at = (int64_t)lane * step;
    const int64_t v = at < n ? f(lo + at) : BIG;
    int c = __popcll(__ballot(v <= j0));
    if (c < 1) c = 1;
    const int64_t adv = (int64_t)(c - 1) * step;
    lo += adv;
    n = (n - adv) < step ? (n - adv) : step;
  }
  W = (lo + lane < n_total) ? f(lo + lane) : BIG;
}

// part 2: every lane counts the window entries <= its own row x (x >= j0): a 6-step binary search through the
// (sorted) window by lane shuffles — all 64 lanes search at once (a ballot per row would serialise the wave's rows).
// ALL 64 lanes must call.  false = the window ran out before x.
__device__ __forceinline__ bool coop_lookup(int64_t W, int64_t lo, int64_t n_total, int64_t x, int64_t& k, int64_t& fk) {
  int mine = 0;
#pragma unroll
  for (int s = RUA_WAVE / 2; s >= 1; s >>= 1) {
    const int64_t w = __shfl(W, mine + s - 1, RUA_WAVE);
    if (w <= x) mine += s;
  }
  const int64_t w_last = __shfl(W, RUA_WAVE - 1, RUA_WAVE);       // (unconditional: every lane takes part in a shuffle)
  if (mine == RUA_WAVE - 1 && w_last <= x) mine = RUA_WAVE;
  if (mine < 1) mine = 1;
  k = lo + mine - 1;
  fk = __shfl(W, mine - 1, RUA_WAVE);
  return !(mine == RUA_WAVE && lo + RUA_WAVE < n_total);
}

template <typename F>
__device__ __forceinline__ bool coop_resolve(F f, int64_t n_total, int64_t j0, int nw, int lane, int64_t& k,
                                             int64_t& fk) {
  (void)nw;
  int64_t lo, W;
  coop_window(f, n_total, j0, lane, lo, W);
  return coop_lookup(W, lo, n_total, j0 + lane, k, fk);
}

}  // namespace rua
