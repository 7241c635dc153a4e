// rua_reduce.hip — segmented reductions over the sequences of ANY layout (gfx950, wave64).
//
// One wave owns one (sequence, 64-lane column chunk): lanes are spread over the hidden
// dimension with 16-byte loads (8 x bf16 / 4 x f32 per lane), rows of the sequence are walked
// in t with UNROLL_T rows in flight, accumulation is fp32 (fp64 for f64) in registers, and the
// result is rounded ONCE on the way out.  Rows narrower than 1 KiB share a wave instruction
// (several t per instruction) and are combined with a butterfly at the end.  No atomics on the
// data path, no LDS traffic: the kernel is a pure HBM stream of N*H*e bytes + S*H*e out.
//
// Row addressing per layout (include/rua.h): CAT off[b]+t (contiguous), PACK boff[t]+rank[b]
// (stride varies with t), LEFT/RIGHT b*T+t(+pad), and CAT+perm for the bucketed scatter_*.
#pragma once
#include <stdlib.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "rua_dev.h"

namespace rua {

#ifndef RUA_UNROLL_T
#define RUA_UNROLL_T 8
#endif
constexpr int UNROLL_T = RUA_UNROLL_T;
constexpr int EXTREME_SLOTS = 1024;  // contention spreading for the global min/max tracker (fold_flags)

// ---------------------------------------------------------------- element conversion
template <typename T> struct elem;
template <> struct elem<float> {
  using acc = float;
  static __device__ __forceinline__ float up(float v) { return v; }
  static __device__ __forceinline__ float down(float v) { return v; }
};
template <> struct elem<double> {
  using acc = double;
  static __device__ __forceinline__ double up(double v) { return v; }
  static __device__ __forceinline__ double down(double v) { return v; }
};
template <> struct elem<__hip_bfloat16> {
  using acc = float;
  static __device__ __forceinline__ float up(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 down(float v) { return __float2bfloat16(v); }
};
template <> struct elem<__half> {
  using acc = float;
  static __device__ __forceinline__ float up(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half down(float v) { return __float2half(v); }
};

template <typename A> __device__ __forceinline__ A acc_inf();
template <> __device__ __forceinline__ float acc_inf<float>() { return __builtin_inff(); }
template <> __device__ __forceinline__ double acc_inf<double>() { return __builtin_inf(); }

// NaN-propagating max/min (torch.segment_reduce / index_reduce semantics)
// — IEEE-754-2019 maximum/minimum: ONE v_maximum3_f32 / v_minimum3_f32 on gfx950, so the hot loops need no NaN
// side-tracking
template <typename A> __device__ __forceinline__ A nmax(A a, A b) { return __builtin_elementwise_maximum(a, b); }
template <typename A> __device__ __forceinline__ A nmin(A a, A b) { return __builtin_elementwise_minimum(a, b); }

// NaN-ignoring max/min (one v_max_f32 / v_min_f32)
__device__ __forceinline__ float fmaxx(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double fmaxx(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float fminx(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double fminx(double a, double b) { return fmin(a, b); }

__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ double fexp(double x) { return exp(x); }
// exp(x - m) for a whole chunk of x against one m: the subtraction and __expf's scaling by log2(e) fold into ONE fma
// in front of v_exp_f32 (the online logsumexp is the one reduction whose ALU time shows next to its memory time).
// `ms` = exp_shift(m), computed once per chunk and column.
constexpr float RUA_LOG2E = 1.44269504088896340736f;
__device__ __forceinline__ float exp_shift(float m) { return -m * RUA_LOG2E; }
__device__ __forceinline__ double exp_shift(double m) { return m; }
__device__ __forceinline__ float exp_shifted(float x, float ms) { return __builtin_amdgcn_exp2f(__builtin_fmaf(x, RUA_LOG2E, ms)); }
__device__ __forceinline__ double exp_shifted(double x, double m) { return exp(x - m); }
__device__ __forceinline__ float flog(float x) { return logf(x); }
__device__ __forceinline__ double flog(double x) { return log(x); }

// order-preserving map float -> unsigned so integer atomics give float min/max
__device__ __forceinline__ uint64_t ordered_bits(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? (uint32_t)~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t ordered_bits(double f) {
  uint64_t u = (uint64_t)__double_as_longlong(f);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ float unordered_f32(uint64_t o) {
  uint32_t u = (uint32_t)o;
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ double unordered_f64(uint64_t u) {
  u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
  return __longlong_as_double((long long)u);
}

// ---------------------------------------------------------------- the forward kernels
// EPL: elements per lane per load (16 B / sizeof(T) on the vector path, 1 on the scalar path)
// COPY: every row that is read is ALSO stored to its row of a PackedSequence (layout CD, storage
// `copy`) — pack and reduce in one pass over the payload (rua_pack_reduce): N*H*e read + N*H*e written
// instead of 3*N*H*e for pack-then-reduce.  Sequences are then walked in CD's rank order.
// SPLIT: sequences longer than `split` rows are cut into parts of `split` rows; the wave that owns the
// sequence publishes the extra parts to a work list (consumed by seg_reduce_tail_kernel, launched next on
// the stream) and every part stores a raw fp32 partial; seg_reduce_combine_kernel folds the partials in
// part order (deterministic) and finalises.  Without it one wave streams a whole sequence (~4 GB/s).

// the addressing of one (sequence slot, column chunk) unit
template <typename T, int EPL>
struct Unit {
  int64_t q, b, chunk, col, len, base, tb, n_rows;
  const int64_t* tbl;
  int rpw, rsub, lp_log2;
  bool colok, live;   // live: false for the padding groups of a RANKS wave past the last sequence
};

// CPW: 64-lane column chunks one wave covers per row (1, or 4 for rows wider than 1 KiB so that a whole row
// — up to 4 KiB — is read back to back by ONE wave: DRAM pages are used in full instead of being shared by
// several waves that each take 1 KiB of it; measured 4.4 -> 6 TB/s at 4-KiB rows)
// RANKS (PackedSequence input, rows narrower than 1 KiB): the 64 >> lp_log2 row groups of a wave serve ADJACENT
// RANKS at the same time step instead of consecutive time steps of one sequence — adjacent ranks are adjacent
// rows of every time step, so a wave instruction reads one contiguous run (a 32-byte row no longer costs a
// whole 128-byte line).  Each group then owns its own sequence: no cross-group combine.
template <typename T, int EPL, bool COPY, int CPW = 1, bool RANKS = false>
__device__ __forceinline__ Unit<T, EPL> make_unit(const rua_layout& L, const rua_layout& CD, const int64_t* perm,
                                                   int64_t q, int64_t chunk, int64_t H, int lp_log2, int lane,
                                                   int glog = 0) {
  Unit<T, EPL> u;
  bool live = true;
  // RANKS with glog > 0 (a CattedSequence of narrow rows): a sequence's lane group is 1 << glog ROW SLOTS wide — it
  // reads that many consecutive rows of its sequence per instruction — so a wave serves 64 >> (lp_log2 + glog) sequences
  if (RANKS) {
    q = q * (RUA_WAVE >> (lp_log2 + glog)) + (lane >> (lp_log2 + glog));
    live = q < L.B;
    if (!live) q = L.B - 1;
  }
  u.q = q;
  u.chunk = chunk;
  // PACK is walked in rank order (longest first = LPT schedule; neighbouring workgroups read
  // neighbouring rows of every time step), everything else in batch order.
  u.b = COPY ? (CD.sorted ? CD.sorted[q] : q) : ((L.kind == RUA_PACK && L.sorted) ? L.sorted[q] : q);
  if (u.b < 0 || u.b >= L.B) u.b = q;   // a corrupt sorted_indices must not index out of range
  u.lp_log2 = lp_log2;
  u.rpw = RANKS ? (1 << glog) : (RUA_WAVE >> lp_log2);                       // rows of ONE sequence per wave instruction
  u.rsub = RANKS ? ((lane >> lp_log2) & ((1 << glog) - 1)) : (lane >> lp_log2);
  u.col = (chunk * CPW * RUA_WAVE + (lane & ((1 << lp_log2) - 1))) * EPL;   // sub-chunk c adds c * 64 * EPL
  u.colok = u.col < H && live;
  // rows that end in half a vector (dispatch_reduce_main: tail_ok): the last lane covers the row's last EPL elements
  if (EPL > 1 && CPW == 1 && u.col < H && u.col + EPL > H) u.col = H - EPL;
  u.live = live;
  u.len = live ? seq_len(L, u.b) : 0;
  u.n_rows = L.n_rows;
  u.base = 0;
  u.tb = 0;
  u.tbl = nullptr;   // row(t) = base + (tbl ? tbl[tb + t] : t)
  switch (L.kind) {
    case RUA_CAT:
      if (perm) { u.tbl = perm; u.tb = cat_off(L, u.b); } else u.base = cat_off(L, u.b);
      break;
    case RUA_PACK:  u.tbl = L.boff; u.base = L.sorted ? q : (L.unsorted ? L.unsorted[u.b] : u.b); break;
    case RUA_LEFT:  u.base = u.b * L.T_phys; break;
    case RUA_RIGHT: u.base = u.b * L.T_phys + (L.T_log - u.len); break;
  }
  return u;
}

// Internal ops: max / min that ALSO count how many elements equal the extreme (in `aux`).  The forward of a
// differentiable reduce_max/min uses them and stores the counts, so the backward is ONE walk (read x, write
// g / ties where x == out) instead of a counting walk plus an applying walk — two passes over the payload, not three.
constexpr int RUA_MAX_T = 6, RUA_MIN_T = 7;
constexpr bool op_is_max(int op) { return op == RUA_MAX || op == RUA_MAX_T; }
constexpr bool op_is_min(int op) { return op == RUA_MIN || op == RUA_MIN_T; }
constexpr bool op_counts(int op) { return op == RUA_MAX_T || op == RUA_MIN_T; }
constexpr bool op_uses_aux(int op) { return op == RUA_LOGSUMEXP || op_counts(op); }

// (extreme, count) <- merge with (x, c).  NaN is the extreme of extremes (torch: max/min propagate NaN, and the
// backward treats NaN elements as the hits of a NaN result), -inf/+inf start values carry count 0.
template <typename A, bool IS_MAX>
__device__ __forceinline__ void tie_update(A& acc, A& cnt, A x, A c) {
  const bool xn = x != x, an = acc != acc;
  const bool better = xn ? !an : (!an && (IS_MAX ? x > acc : x < acc));
  const bool equal = xn ? an : (x == acc);
  cnt = better ? c : (equal ? cnt + c : cnt);
  acc = better ? x : acc;
}

// running state of one lane
template <typename A, int EPL>
struct Fold {
  A acc[EPL], aux[EPL];   // aux: running sum for LOGSUMEXP (acc holds the running max)
  // [r5] max / min / logsumexp: the OPPOSITE extreme of everything this lane has folded (the minimum under max and
  // logsumexp, the maximum under min) — the reference's `initial` (reduce.py:35,40,57: tensor.min() / .max()) for the
  // segments that turn out empty.  One scalar per lane, one v_minimum3 / v_maximum3 per two elements in loops that wait
  // for HBM anyway (NaN-propagating like the fold itself: one instruction, and a NaN anywhere raises the poison flag,
  // which decides everything); fold_flags hands the wave's value to the scratch, so no second walk over the payload
  // is ever needed.
  A opp;
};

template <typename A, int EPL, int OP>
__device__ __forceinline__ void fold_init(Fold<A, EPL>& f) {
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    f.acc[e] = (OP == RUA_PROD) ? (A)1 : (op_is_max(OP) || OP == RUA_LOGSUMEXP) ? -acc_inf<A>()
             : op_is_min(OP) ? acc_inf<A>() : (A)0;
    f.aux[e] = (A)0;
  }
  f.opp = op_is_min(OP) ? -acc_inf<A>() : acc_inf<A>();
}

// fold rows [t_lo, t_hi) of the unit's sequence
template <typename T, int EPL, int OP, bool NT, bool COPY, int CPW = 1, bool RANKS = false>
__device__ __forceinline__ void fold_rows(const Unit<T, EPL>& U, int64_t t_lo, int64_t t_hi,
                                          const T* __restrict__ data, int64_t H,
                                          Fold<typename elem<T>::acc, EPL * CPW>& f, const rua_layout& CD,
                                          T* __restrict__ copy, int lane, int team_w = 0, int team_n = 1,
                                          bool track = true) {
  using A = typename elem<T>::acc;
  constexpr int UT = CPW == 1 ? UNROLL_T : UNROLL_T / 2;     // rows in flight (x CPW loads each)
  constexpr int CW = RUA_WAVE * EPL;                          // elements per 64-lane column chunk
  // (a 16-byte piece may sit on an 8-byte boundary only: rows of 8 (mod 16) bytes, see dispatch_reduce_main)
  struct alignas(sizeof(T) * EPL > 8 ? 8 : sizeof(T) * EPL) Pack { T v[EPL]; };
  typedef unsigned int RawV0 __attribute__((ext_vector_type(sizeof(T) * EPL >= 4 ? sizeof(T) * EPL / 4 : 1)));
  typedef RawV0 RawV __attribute__((aligned(sizeof(T) * EPL > 8 ? 8 : (sizeof(T) * EPL >= 4 ? sizeof(T) * EPL : 4))));
  const int rpw = U.rpw, rsub = U.rsub;
  const bool colok = U.colok;
  const int64_t col = U.col, base = U.base, tb = U.tb, L_rows = U.n_rows;
  const int64_t* __restrict__ tbl = U.tbl;

  // the row table (boff / perm) is fetched 64 entries at a time with one coalesced load and
  // handed to the lanes by ds_bpermute; the next block's entries are in flight while this
  // block's payload streams.
  int64_t tv = (tbl && t_lo + lane < t_hi) ? tbl[tb + t_lo + lane] : 0;
  int64_t cv = (COPY && t_lo + lane < t_hi) ? CD.boff[t_lo + lane] : 0;      // destination rows: boff[t] + rank
  for (int64_t tblk = t_lo; tblk < t_hi; tblk += RUA_WAVE) {
    const int64_t nxt = tblk + RUA_WAVE + lane;
    const int64_t tv_next = (tbl && nxt < t_hi) ? tbl[tb + nxt] : 0;
    const int64_t cv_next = (COPY && nxt < t_hi) ? CD.boff[nxt] : 0;
    const int nblk = (t_hi - tblk) < RUA_WAVE ? (int)(t_hi - tblk) : RUA_WAVE;
    for (int k = 0; k < nblk; k += rpw * UT) {       // (RANKS: rpw = the group's row slots, 1 for adjacent ranks)
      // a team of waves shares one sequence (seg_reduce_team_kernel): wave w takes every team_n-th row group
      if (team_n > 1 && (int)(((tblk - t_lo) / (rpw * UT) + k / (rpw * UT)) % team_n) != team_w) continue;
      int64_t row[UT];
      int64_t crow[UT];
      Pack p[UT][CPW];
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int tl = k + u * rpw + rsub;   // RANKS: every group walks the same time steps of its own sequence
        const int64_t tabv = __shfl(tv, tl & (RUA_WAVE - 1), RUA_WAVE);
        row[u] = -1;
        if (colok && tl < nblk && (!RANKS || tblk + tl < U.len)) row[u] = base + (tbl ? tabv : tblk + tl);
        if (row[u] >= L_rows) row[u] = -1;   // lengths that run past the storage read nothing
        if (COPY) crow[u] = __shfl(cv, tl & (RUA_WAVE - 1), RUA_WAVE) + U.q;
      }
#pragma unroll
      for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int c = 0; c < CPW; ++c)
          if (row[u] >= 0 && (CPW == 1 || col + c * CW < H)) {
            const T* src = data + row[u] * H + col + c * CW;
            if (NT && sizeof(Pack) >= 4) {   // streaming read of a payload that cannot stay in cache
              RawV raw = __builtin_nontemporal_load(reinterpret_cast<const RawV*>(src));
              __builtin_memcpy(&p[u][c], &raw, sizeof(Pack));
            } else {
              p[u][c] = *reinterpret_cast<const Pack*>(src);
            }
          }
      if (COPY) {
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
          for (int c = 0; c < CPW; ++c)
            if (row[u] >= 0 && crow[u] < CD.n_rows && (CPW == 1 || col + c * CW < H)) {
              T* dstp = copy + crow[u] * H + col + c * CW;
              if (NT && sizeof(Pack) >= 4) {
                RawV raw;
                __builtin_memcpy(&raw, &p[u][c], sizeof(Pack));
                __builtin_nontemporal_store(raw, reinterpret_cast<RawV*>(dstp));
              } else {
                *reinterpret_cast<Pack*>(dstp) = p[u][c];
              }
            }
      }
      if (OP == RUA_LOGSUMEXP) {
        // chunk-wise online logsumexp: the chunk's max first, ONE rescale of the running sum per
        // chunk, then one exp per element.  Rows that are not there are made -inf ONCE per 16-byte load (not
        // per element: the per-element selects used to cost as many issue slots as the arithmetic): they can
        // never raise the maximum and add exp(-inf - m) = 0.
        Pack ninf;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ninf.v[e] = elem<T>::down(-acc_inf<A>());
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
          for (int c = 0; c < CPW; ++c)
            if (!(row[u] >= 0 && (CPW == 1 || col + c * CW < H))) p[u][c] = ninf;
        bool all_there = true;                     // (the bulk of a sequence: every row of the chunk is there)
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
          for (int c = 0; c < CPW; ++c) all_there &= (row[u] >= 0 && (CPW == 1 || col + c * CW < H));
#pragma unroll
        for (int ce = 0; ce < EPL * CPW; ++ce) {
          const int c = ce / EPL, e = ce % EPL;
          A x[UT];
          A cm = -acc_inf<A>();
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            x[u] = elem<T>::up(p[u][c].v[e]);
            cm = nmax(cm, x[u]);
          }
          // the opposite extreme counts the rows that are there only (the others were made -inf above).  (A minimum per
          // column first and ONE link into f.opp — a shorter dependency chain — was measured and is slower: cfg2
          // reduce_logsumexp(p) 112 -> 125 us; this loop is the one reduction where the ALU shows, and the tracking costs
          // it 8-10 % at cfg3, where a wave lives for 32 rows: profiles/r05_initial_ab.txt)
          if (!track) {
          } else if (all_there) {
#pragma unroll
            for (int u = 0; u < UT; ++u) f.opp = nmin(f.opp, x[u]);
          } else {
#pragma unroll
            for (int u = 0; u < UT; ++u)
              if (row[u] >= 0 && (CPW == 1 || col + c * CW < H)) f.opp = nmin(f.opp, x[u]);
          }
          // NaN-propagating and sticky: once an element is NaN the running max stays NaN (that is how the
          // reference's NaN-poisoned `initial` is detected: fold_flags) and the sum turns NaN
          const A nm = nmax(f.acc[ce], cm);
          if (nm != f.acc[ce]) { f.aux[ce] *= fexp(f.acc[ce] - nm); f.acc[ce] = nm; }
          // a lane that has seen nothing above -inf yet subtracts 0, not -inf: -inf - -inf would turn the rows that
          // are not there into NaN.  (A sequence whose elements are ALL -inf is NaN in the reference — exp(-inf - -inf)
          // — and fold_store restores that from the final maximum.)
          const A m = (f.acc[ce] == -acc_inf<A>()) ? (A)0 : f.acc[ce];
          const A ms = exp_shift(m);
#pragma unroll
          for (int u = 0; u < UT; ++u) f.aux[ce] += exp_shifted(x[u], ms);   // NaN x -> NaN sum, as the reference
        }
      } else if (op_counts(OP)) {
        // max / min that also count the elements equal to the extreme, chunk-wise like logsumexp so that it stays
        // near the memory rate (~5 issue slots per element; a per-element (extreme, count) update is ~15 and made
        // the forward three times slower): (A) the chunk's extreme per column, (B) ONE reset of the count per
        // column if the extreme moved, (C) count += (x == extreme) per element.
        constexpr bool MX = OP == RUA_MAX_T;
        A m[EPL * CPW];
#pragma unroll
        for (int ce = 0; ce < EPL * CPW; ++ce) m[ce] = f.acc[ce];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          if (row[u] < 0) continue;
#pragma unroll
          for (int ce = 0; ce < EPL * CPW; ++ce) {
            const int c = ce / EPL, e = ce % EPL;
            if (CPW > 1 && col + c * CW >= H) continue;
            const A x = elem<T>::up(p[u][c].v[e]);
            m[ce] = MX ? nmax(m[ce], x) : nmin(m[ce], x);
            if (track) f.opp = MX ? nmin(f.opp, x) : nmax(f.opp, x);
          }
        }
        bool any_nan = false;
#pragma unroll
        for (int ce = 0; ce < EPL * CPW; ++ce) {
          const bool mn = m[ce] != m[ce];
          const bool same = (m[ce] == f.acc[ce]) || (mn && f.acc[ce] != f.acc[ce]);
          f.aux[ce] = same ? f.aux[ce] : (A)0;                      // the extreme moved (or turned NaN): start over
          any_nan |= mn;
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          if (row[u] < 0) continue;
#pragma unroll
          for (int ce = 0; ce < EPL * CPW; ++ce) {
            const int c = ce / EPL, e = ce % EPL;
            if (CPW > 1 && col + c * CW >= H) continue;
            const A x = elem<T>::up(p[u][c].v[e]);
            f.aux[ce] += (x == m[ce]) ? (A)1 : (A)0;
          }
        }
        if (any_nan) {
          // rare: a NaN extreme counts its NaN elements (torch's backward treats them as the hits of a NaN result)
#pragma unroll
          for (int ce = 0; ce < EPL * CPW; ++ce) {
            if (m[ce] == m[ce]) continue;
            const int c = ce / EPL, e = ce % EPL;
            A k = f.aux[ce];                                         // NaNs counted in earlier chunks (or 0)
#pragma unroll
            for (int u = 0; u < UT; ++u) {
              if (row[u] < 0 || (CPW > 1 && col + c * CW >= H)) continue;
              const A x = elem<T>::up(p[u][c].v[e]);
              k += (x != x) ? (A)1 : (A)0;
            }
            f.aux[ce] = k;
          }
        }
#pragma unroll
        for (int ce = 0; ce < EPL * CPW; ++ce) f.acc[ce] = m[ce];
      } else {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          if (row[u] < 0) continue;
#pragma unroll
          for (int ce = 0; ce < EPL * CPW; ++ce) {
            const int c = ce / EPL, e = ce % EPL;
            if (CPW > 1 && col + c * CW >= H) continue;
            const A x = elem<T>::up(p[u][c].v[e]);
            if (OP == RUA_SUM || OP == RUA_MEAN) f.acc[ce] += x;
            else if (OP == RUA_PROD) f.acc[ce] *= x;
            else if (OP == RUA_MAX) { f.acc[ce] = nmax(f.acc[ce], x); if (track) f.opp = nmin(f.opp, x); }   // v_maximum3_f32: NaN-propagating like torch
            else if (OP == RUA_MIN) { f.acc[ce] = nmin(f.acc[ce], x); if (track) f.opp = nmax(f.opp, x); }
          }
        }
      }
    }
    tv = tv_next;
    cv = cv_next;
  }
}

// fold the NaN flags in and combine the rpw row-groups of the wave (lanes that differ in the bits above
// lp_log2); afterwards every lane of a column holds the wave's value
template <typename A, int EPL, int OP, bool RANKS = false>
__device__ __forceinline__ void fold_wave(Fold<A, EPL>& f, int lp_log2, int glog = 0) {
  // RANKS: groups are separate sequences — only the row slots INSIDE a group (1 << glog of them) meet
  for (int d = 1 << lp_log2; d < (RANKS ? (1 << (lp_log2 + glog)) : RUA_WAVE); d <<= 1) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const A o = __shfl_xor(f.acc[e], d, RUA_WAVE);
      if (OP == RUA_SUM || OP == RUA_MEAN) f.acc[e] += o;
      else if (OP == RUA_PROD) f.acc[e] *= o;
      else if (OP == RUA_MAX) f.acc[e] = nmax(f.acc[e], o);
      else if (OP == RUA_MIN) f.acc[e] = nmin(f.acc[e], o);
      else if (op_counts(OP)) tie_update<A, OP == RUA_MAX_T>(f.acc[e], f.aux[e], o, __shfl_xor(f.aux[e], d, RUA_WAVE));
      else {
        const A os = __shfl_xor(f.aux[e], d, RUA_WAVE);
        const A m = nmax(f.acc[e], o);
        if (m == -acc_inf<A>()) { f.aux[e] = f.aux[e] + os; }       // both empty so far
        else { f.aux[e] = f.aux[e] * fexp(f.acc[e] - m) + os * fexp(o - m); }
        f.acc[e] = m;
      }
    }
  }
}

// merge another part's (already wave-folded) values, in part order
template <typename A, int EPL, int OP>
__device__ __forceinline__ void fold_merge(Fold<A, EPL>& f, const A* acc2, const A* aux2) {
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const A o = acc2[e];
    if (OP == RUA_SUM || OP == RUA_MEAN) f.acc[e] += o;
    else if (OP == RUA_PROD) f.acc[e] *= o;
    else if (OP == RUA_MAX) f.acc[e] = nmax(f.acc[e], o);
    else if (OP == RUA_MIN) f.acc[e] = nmin(f.acc[e], o);
    else if (op_counts(OP)) tie_update<A, OP == RUA_MAX_T>(f.acc[e], f.aux[e], o, aux2[e]);
    else {
      const A m = nmax(f.acc[e], o);
      if (m == -acc_inf<A>()) { f.aux[e] = f.aux[e] + aux2[e]; }
      else { f.aux[e] = f.aux[e] * fexp(f.acc[e] - m) + aux2[e] * fexp(o - m); }
      f.acc[e] = m;
    }
  }
}

// include_self: 0 = overwrite (empty sequence -> empty_val), 1 = fold the old out[b] in,
//               2 = leave out[b] untouched when the sequence is empty (index_reduce semantics)
template <typename T, int EPL, int OP, int CPW = 1, bool RANKS = false>
__device__ __forceinline__ void fold_store(const Unit<T, EPL>& U, Fold<typename elem<T>::acc, EPL * CPW>& f,
                                           T* __restrict__ out, int64_t H, int include_self, T empty_val,
                                           typename elem<T>::acc* __restrict__ ties = nullptr) {
  using A = typename elem<T>::acc;
  constexpr int CW = RUA_WAVE * EPL;
  const bool keep = include_self == 2 && U.len <= 0;
  const bool inc = include_self == 1;
  if (U.colok && U.rsub == 0 && !keep) {       // (RANKS: rsub = the row slot inside the sequence's group)
    const int64_t cnt = U.len + (inc ? 1 : 0);
#pragma unroll
    for (int ce = 0; ce < EPL * CPW; ++ce) {
      const int c = ce / EPL, e = ce % EPL;
      if (CPW > 1 && U.col + c * CW >= H) continue;
      T* o = out + U.b * H + U.col + c * CW;
      A r = f.acc[ce];
      if (OP == RUA_LOGSUMEXP) {
        if (inc) {  // fold exp(self) in
          const A x = elem<T>::up(o[e]);
          const A m = nmax(r, x);
          f.aux[ce] = f.aux[ce] * fexp(r - m) + fexp(x - m);
          r = m;
        }
        // every element -inf: exp(-inf - -inf) = NaN in the reference (reduce.py:56-61)
        r = (r == -acc_inf<A>() && cnt > 0) ? (A)__builtin_nanf("") : flog(f.aux[ce]) + r;
      } else if (inc) {
        const A x = elem<T>::up(o[e]);
        if (OP == RUA_SUM || OP == RUA_MEAN) r += x;
        else if (OP == RUA_PROD) r *= x;
        else if (OP == RUA_MAX) r = nmax(r, x);
        else if (OP == RUA_MIN) r = nmin(r, x);
        else if (op_counts(OP)) tie_update<A, OP == RUA_MAX_T>(r, f.aux[ce], x, (A)0);   // the old row's own tie: the caller's
      }
      if (OP == RUA_MEAN && cnt > 0) r = r / (A)cnt;
      o[e] = (cnt == 0) ? empty_val : elem<T>::down(r);
      if (op_counts(OP) && ties) ties[U.b * H + U.col + c * CW + e] = f.aux[ce];   // elements equal to the extreme
    }
  }
}

// The reference's `initial` for max / min / logsumexp is a GLOBAL extreme of the data (reduce.py:35,40: tensor.min()
// resp. tensor.max()).  It only shows in two rare cases.  After its fold every wave
//   * raises flags in extreme[EXTREME_SLOTS]:
//       bit 0  some element is NaN (the running max/min is NaN-propagating, so a NaN accumulator says so; for
//              logsumexp the accumulator is the running max, which inf - inf cannot turn NaN): `initial` is NaN and
//              poisons EVERY segment — NaN is written everywhere;
//       bit 1  some segment is empty: it takes the global extreme.
//     The flag word is read before the atomic: once it is set nobody touches it again.
//   * [r5] folds the opposite extreme of the rows IT read (Fold::opp) into one of 1 024 hashed slots with ONE
//     fire-and-forget atomic: the global extreme is complete when the reduce is, so nobody walks the payload a second
//     time (rounds 1-4: a lazy second walk, by the trailing launch's workgroups; and the empty rows of a mostly-empty
//     batch were then patched by ONE workgroup).  Zero-neutral in both directions (a scratch that arrives zeroed needs
//     no initialising launch): the maximum is kept as its ordered bits, the minimum as their complement, both under
//     atomicMax.
//     Why 1 024 slots, and why nothing cleverer: same-address atomics execute at the memory side, one per ~11 ns and
//     address at best, and a wave's slot on the CU is not free until its atomic is acknowledged.  With 64 slots cfg3's
//     16 384 waves queued 256 deep (segment_max 108 -> 117 us), 200 000 one-wave sequences 3 000 deep (0.34 ms of
//     queueing where the payload takes 0.05).  LOOKING at the slot first and sending only what beats it made it worse
//     every way it was tried: an agent-scope load is served at the memory side too (at the wave's end: 118 us; in front
//     of the row loop, where the in-order return of loads holds the first payload rows behind it: 158 us), a
//     non-temporal load of a line that atomics keep dropping from L2 250 us.  Neighbouring workgroups take slots on
//     different 128-byte lines (stride 17 words).  profiles/r05_initial_ab.txt.
// What patches the output afterwards: fill_empty_kernel (a trailing launch, every workgroup its share).
__device__ __forceinline__ int extreme_slot() {
  return (int)(((unsigned)blockIdx.x * 17u + (threadIdx.x >> 6) * 5u) & (unsigned)(EXTREME_SLOTS - 1));
}

// Does this launch need the global extreme at all?  Only if some sequence may be EMPTY.  The caller can rule that out
// (RUA_OP_NO_EMPTY: lengths it holds on the host; a PackedSequence, whose batch_sizes live there), and so can the device:
// a CAT layout may carry, in `bsz`, a pointer to the number of lengths <= 0 as rua_exclusive_scan_i64 counted them while
// it made `off` (total[1]) — one uniform load per wave, of a word an earlier launch wrote.  Then the hot loops leave
// Fold::opp alone and no wave sends its atomic: logsumexp, the one loop where the ALU shows, got 8-10 % back at cfg3,
// where a wave lives for 32 rows (profiles/r05_initial_ab.txt).  A NaN needs no extreme (the flag decides everything).
template <int OP>
__device__ __forceinline__ bool track_initial(const rua_layout& L, const unsigned long long* __restrict__ extreme,
                                              int no_empty) {
  if (!(op_is_max(OP) || op_is_min(OP) || OP == RUA_LOGSUMEXP) || !extreme || no_empty) return false;
  if (L.kind == RUA_CAT && L.bsz) return L.bsz[0] != 0;
  return true;
}

template <typename A, int EPL, int OP>
__device__ __forceinline__ void fold_flags(const Fold<A, EPL>& f, unsigned long long* __restrict__ extreme,
                                           int lane, bool empty_unit, bool track = true) {
  if (!(op_is_max(OP) || op_is_min(OP) || OP == RUA_LOGSUMEXP) || !extreme) return;
  bool nan = false;
#pragma unroll
  for (int e = 0; e < EPL; ++e) nan |= (f.acc[e] != f.acc[e]);
  A opp = f.opp;
#pragma unroll
  for (int d = RUA_WAVE / 2; d > 0; d >>= 1) {
    const A o = __shfl_xor(opp, d, RUA_WAVE);
    opp = op_is_min(OP) ? nmax(opp, o) : nmin(opp, o);
  }
  const unsigned long long want = (__any(nan) ? 1ull : 0ull) | (__any(empty_unit) ? 2ull : 0ull);
  if (lane != 0) return;
  // (a wave that read nothing still holds the start value: nothing to hand over; a NaN raised the poison flag and
  // decides everything)
  if (track && opp == opp && opp != (op_is_min(OP) ? -acc_inf<A>() : acc_inf<A>())) {
    const unsigned long long bits = (unsigned long long)ordered_bits(opp);
    atomicMax(&extreme[extreme_slot()], op_is_min(OP) ? bits : ~bits);
  }
  if (want == 0ull) return;
  const unsigned long long have = __atomic_load_n(&extreme[EXTREME_SLOTS], __ATOMIC_RELAXED);
  if ((have & want) != want) atomicOr(&extreme[EXTREME_SLOTS], want);
}

// ---- long-sequence splitting: workspace layout (int64 words unless noted)
//   forward: ctr[0] = (long units << 32) | published extra items (one atomic per long unit); backward: ctr[0] = items
//   long_list[max_u][4] = {q, chunk, nparts, pbase};  items[max_u][4] = {q, chunk, part, slot}
//   partials[2*max_u][2][64*EPL] of A
struct SplitWs {
  unsigned long long* ctr;
  int64_t* long_list;
  int64_t* items;
  void* partials;
  int64_t max_u;      // bound on extra items and on long units
  int64_t split;      // rows per part (0 = splitting off)
  void* ties;         // [B, H] accumulators of the tie-counting max/min (RUA_MAX_T / RUA_MIN_T), else NULL
};

template <typename A, int EPL, int OP>
__device__ __forceinline__ void store_partial(void* partials, int64_t slot, int lane, const Fold<A, EPL>& f) {
  A* p = reinterpret_cast<A*>(partials) + (slot * 2 * RUA_WAVE + lane) * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    p[e] = f.acc[e];
    if (op_uses_aux(OP)) p[RUA_WAVE * EPL + e] = f.aux[e];   // logsumexp's sum / the tie count; nothing else has one
  }
}

template <typename T, int EPL, int OP, bool NT, bool COPY, bool SPLIT, int CPW, int WPB = 1>
__global__ __launch_bounds__(RUA_WAVE * WPB) void seg_reduce_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                              const T* __restrict__ data, T* __restrict__ out,
                                                              int64_t H, int lp_log2, int64_t n_chunks,
                                                              int include_self, T empty_val,
                                                              unsigned long long* __restrict__ extreme,
                                                              rua_layout CD, T* __restrict__ copy, SplitWs W,
                                                              int no_empty) {
  using A = typename elem<T>::acc;
  // ONE wave per workgroup: sequences differ in length, and a multi-wave workgroup would hold
  // its CU slots until its longest sequence is done.  (WPB > 1: that many INDEPENDENT waves per workgroup, each with a
  // unit of its own — no barrier, no LDS: launch_reduce says where that pays.)
  const int lane = threadIdx.x & (RUA_WAVE - 1);
  const int64_t wid = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
  const int64_t q = wid / n_chunks;          // sequence slot
  if (q >= L.B) return;
  constexpr int NE = EPL * CPW;
  const Unit<T, EPL> U = make_unit<T, EPL, COPY, CPW>(L, CD, perm, q, wid - q * n_chunks, H, lp_log2, lane);
  Fold<A, NE> f;
  fold_init<A, NE, OP>(f);
  const bool track = track_initial<OP>(L, extreme, no_empty);

  if (SPLIT && U.len > W.split) {
    // long sequence: this wave takes part 0 and publishes the rest
    const int64_t nparts = (U.len + W.split - 1) / W.split;
    // ONE returning atomic per long unit (same-address atomics serialise in L2: ~20 ns each, and a batch of
    // moderately long sequences publishes thousands): low 32 bits count the extra parts, high 32 bits the long
    // units; the partial slots of a unit start at (extra parts before it) + (long units before it).
    int64_t pbase = 0, ibase = 0;
    if (lane == 0) {
      const unsigned long long old = atomicAdd(&W.ctr[0], (1ull << 32) | (unsigned long long)(nparts - 1));
      ibase = (int64_t)(old & 0xffffffffull);
      const int64_t li = (int64_t)(old >> 32);
      pbase = ibase + li;
      int64_t* e = W.long_list + li * 4;
      e[0] = q; e[1] = U.chunk; e[2] = nparts; e[3] = pbase;
    }
    pbase = __shfl(pbase, 0, RUA_WAVE);
    ibase = __shfl(ibase, 0, RUA_WAVE);
    for (int64_t p = 1 + lane; p < nparts; p += RUA_WAVE) {
      int64_t* e = W.items + (ibase + p - 1) * 4;
      e[0] = q; e[1] = U.chunk; e[2] = p; e[3] = pbase + p;
    }
    fold_rows<T, EPL, OP, NT, COPY, CPW>(U, 0, W.split, data, H, f, CD, copy, lane, 0, 1, track);
    fold_wave<A, NE, OP>(f, lp_log2);
    store_partial<A, NE, OP>(W.partials, pbase, lane, f);
  } else {
    fold_rows<T, EPL, OP, NT, COPY, CPW>(U, 0, U.len, data, H, f, CD, copy, lane, 0, 1, track);
    fold_wave<A, NE, OP>(f, lp_log2);
    fold_store<T, EPL, OP, CPW>(U, f, out, H, include_self, empty_val, (A*)W.ties);
  }
  fold_flags<A, NE, OP>(f, extreme, lane, U.len <= 0, track);
}

// Few but long sequences (units <= the wave slots of the chip, hundreds of rows each): with one wave per sequence
// everything starts at once, the short sequences leave early and the long ones finish on a half-empty chip.  Here a
// TEAM of waves (one workgroup) shares each (sequence, column chunk): wave w folds every TEAM-th group of rows —
// at any moment the team reads one contiguous run — the partials meet in LDS and wave 0 merges them in wave order
// (a fixed association: bitwise reproducible).  TEAM = blockDim.x / 64 (2 or 4).
constexpr int TEAM_MAX = 4;
template <typename T, int EPL, int OP, bool NT>
__global__ __launch_bounds__(RUA_WAVE * TEAM_MAX) void seg_reduce_team_kernel(
    rua_layout L, const int64_t* __restrict__ perm, const T* __restrict__ data, T* __restrict__ out, int64_t H,
    int lp_log2, int64_t n_chunks, int include_self, T empty_val, unsigned long long* __restrict__ extreme,
    typename elem<T>::acc* __restrict__ ties, int no_empty) {
  using A = typename elem<T>::acc;
  __shared__ A s_acc[TEAM_MAX][RUA_WAVE * EPL];
  __shared__ A s_aux[TEAM_MAX][RUA_WAVE * EPL];
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6, team = blockDim.x >> 6;
  const int64_t wid = blockIdx.x;
  const int64_t q = wid / n_chunks;
  if (q >= L.B) return;                                   // block-uniform
  const Unit<T, EPL> U = make_unit<T, EPL, false, 1>(L, L, perm, q, wid - q * n_chunks, H, lp_log2, lane);
  Fold<A, EPL> f;
  fold_init<A, EPL, OP>(f);
  const bool track = track_initial<OP>(L, extreme, no_empty);
  fold_rows<T, EPL, OP, NT, false, 1>(U, 0, U.len, data, H, f, L, nullptr, lane, wave, team, track);
  fold_wave<A, EPL, OP>(f, lp_log2);
  fold_flags<A, EPL, OP>(f, extreme, lane, wave == 0 && U.len <= 0, track);
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < EPL; ++k) { s_acc[wave][lane * EPL + k] = f.acc[k]; s_aux[wave][lane * EPL + k] = f.aux[k]; }
  }
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < team; ++w) {
      A a2[EPL], x2[EPL];
#pragma unroll
      for (int k = 0; k < EPL; ++k) { a2[k] = s_acc[w][lane * EPL + k]; x2[k] = s_aux[w][lane * EPL + k]; }
      fold_merge<A, EPL, OP>(f, a2, x2);
    }
    fold_store<T, EPL, OP, 1>(U, f, out, H, include_self, empty_val, ties);
  }
}

// reduce over a PackedSequence with narrow rows: adjacent ranks side by side (see make_unit)
// [r5] SPLIT (glog > 0 only): a sequence longer than W.split rows is cut into parts exactly as seg_reduce_kernel cuts it —
// this wave folds part 0 and publishes the rest to the work list of seg_reduce_tail_kernel / seg_reduce_combine_kernel,
// launched next — so lengths that live on the device only (the split is armed whenever nobody can vouch for them) keep
// the four-per-wave form at rows of <= 32 bytes instead of falling back to one wave per sequence (2.2 / 4.1 TB/s).
template <typename T, int EPL, int OP, bool NT, bool SPLIT = false>
__global__ __launch_bounds__(RUA_WAVE) void seg_reduce_ranks_kernel(rua_layout L, const T* __restrict__ data,
                                                                    T* __restrict__ out, int64_t H, int lp_log2,
                                                                    int include_self, T empty_val,
                                                                    unsigned long long* __restrict__ extreme,
                                                                    typename elem<T>::acc* __restrict__ ties, int glog,
                                                                    SplitWs W, int no_empty, int check) {
  using A = typename elem<T>::acc;
  const int lane = threadIdx.x;
  const Unit<T, EPL> U = make_unit<T, EPL, false, 1, true>(L, L, nullptr, blockIdx.x, 0, H, lp_log2, lane, glog);
  const bool track = track_initial<OP>(L, extreme, no_empty);
  int64_t t_hi = U.len;                       // the wave walks to its longest sequence
#pragma unroll
  for (int d = RUA_WAVE / 2; d > 0; d >>= 1) {
    const int64_t o = __shfl_xor(t_hi, d, RUA_WAVE);
    t_hi = o > t_hi ? o : t_hi;
  }
  if (glog > 0 || check) {
    // Several row slots per sequence (a CattedSequence at rows of <= 32 bytes, a few sequences per wave): the wave checks
    // its OWN lengths — the lengths may live on the device only, nobody vouched for them — and when they are far apart
    // (the longest beyond twice the wave's average + 64 rows: one giant sequence next to short ones would be walked by
    // a quarter of the lanes) it takes its sequences one after the other with every row slot, as seg_reduce_kernel does.
    // ([r5] `check`: the same for every row slot its own sequence — a batch of SHORT sequences, which the host can tell
    // from rows / sequences alone — when nobody vouched for the longest one: 1 M sequences of U(1,8) rows of 64 bytes with
    // device-only lengths 0.72 -> 0.09 ms, where one wave per sequence is bound by the rate of workgroup dispatch)
    const int per_wave = RUA_WAVE >> (lp_log2 + glog);
    int64_t total = (U.rsub == 0 && (lane & ((1 << lp_log2) - 1)) == 0 && U.live) ? U.len : 0;   // one lane per sequence
#pragma unroll
    for (int d = RUA_WAVE / 2; d > 0; d >>= 1) total += __shfl_xor(total, d, RUA_WAVE);
    if (t_hi * per_wave > 2 * total + (int64_t)64 * per_wave || (SPLIT && t_hi > W.split)) {        // (wave-uniform)
      for (int g = 0; g < per_wave; ++g) {
        const int64_t q = (int64_t)blockIdx.x * per_wave + g;
        if (q >= L.B) break;
        const Unit<T, EPL> V = make_unit<T, EPL, false, 1, false>(L, L, nullptr, q, 0, H, lp_log2, lane);
        Fold<A, EPL> fv;
        fold_init<A, EPL, OP>(fv);
        if (SPLIT && V.len > W.split) {                  // (as in seg_reduce_kernel: part 0 here, the rest published)
          const int64_t nparts = (V.len + W.split - 1) / W.split;
          int64_t pbase = 0, ibase = 0;
          if (lane == 0) {
            const unsigned long long old = atomicAdd(&W.ctr[0], (1ull << 32) | (unsigned long long)(nparts - 1));
            ibase = (int64_t)(old & 0xffffffffull);
            const int64_t li = (int64_t)(old >> 32);
            pbase = ibase + li;
            int64_t* e = W.long_list + li * 4;
            e[0] = q; e[1] = 0; e[2] = nparts; e[3] = pbase;
          }
          pbase = __shfl(pbase, 0, RUA_WAVE);
          ibase = __shfl(ibase, 0, RUA_WAVE);
          for (int64_t pp = 1 + lane; pp < nparts; pp += RUA_WAVE) {
            int64_t* e = W.items + (ibase + pp - 1) * 4;
            e[0] = q; e[1] = 0; e[2] = pp; e[3] = pbase + pp;
          }
          fold_rows<T, EPL, OP, NT, false, 1, false>(V, 0, W.split, data, H, fv, L, nullptr, lane, 0, 1, track);
          fold_wave<A, EPL, OP, false>(fv, lp_log2);
          store_partial<A, EPL, OP>(W.partials, pbase, lane, fv);
        } else {
          fold_rows<T, EPL, OP, NT, false, 1, false>(V, 0, V.len, data, H, fv, L, nullptr, lane, 0, 1, track);
          fold_wave<A, EPL, OP, false>(fv, lp_log2);
          fold_store<T, EPL, OP, 1, false>(V, fv, out, H, include_self, empty_val, ties);
        }
        fold_flags<A, EPL, OP>(fv, extreme, lane, V.len <= 0, track);
      }
      return;
    }
  }
  Fold<A, EPL> f;
  fold_init<A, EPL, OP>(f);
  fold_rows<T, EPL, OP, NT, false, 1, true>(U, 0, t_hi, data, H, f, L, nullptr, lane, 0, 1, track);
  fold_wave<A, EPL, OP, true>(f, lp_log2, glog);
  fold_store<T, EPL, OP, 1, true>(U, f, out, H, include_self, empty_val, ties);
  fold_flags<A, EPL, OP>(f, extreme, lane, U.live && U.len <= 0, track);
}

// the published parts 1.. of long sequences
template <typename T, int EPL, int OP, bool NT, bool COPY, int CPW>
__global__ __launch_bounds__(RUA_WAVE) void seg_reduce_tail_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                                   const T* __restrict__ data, int64_t H,
                                                                   int lp_log2,
                                                                   unsigned long long* __restrict__ extreme,
                                                                   rua_layout CD, T* __restrict__ copy, SplitWs W,
                                                                   int no_empty) {
  using A = typename elem<T>::acc;
  const int lane = threadIdx.x;
  constexpr int NE = EPL * CPW;
  const bool track = track_initial<OP>(L, extreme, no_empty);
  const int64_t n_items = (int64_t)(W.ctr[0] & 0xffffffffull);
  // the grid is capped (SPLIT_GRID_CAP): every workgroup strides over the published items
  for (int64_t i = blockIdx.x; i < n_items; i += gridDim.x) {
    const int64_t* e = W.items + i * 4;
    const Unit<T, EPL> U = make_unit<T, EPL, COPY, CPW>(L, CD, perm, e[0], e[1], H, lp_log2, lane);
    const int64_t t_lo = e[2] * W.split;
    const int64_t t_hi = (t_lo + W.split < U.len) ? t_lo + W.split : U.len;
    Fold<A, NE> f;
    fold_init<A, NE, OP>(f);
    fold_rows<T, EPL, OP, NT, COPY, CPW>(U, t_lo, t_hi, data, H, f, CD, copy, lane, 0, 1, track);
    fold_wave<A, NE, OP>(f, lp_log2);
    store_partial<A, NE, OP>(W.partials, e[3], lane, f);
    fold_flags<A, NE, OP>(f, extreme, lane, false, track);
  }
}

// fold the partials of every long unit and finalise.  The grid is small (COMBINE_GRID workgroups of 16 waves) and
// strides over the long units in two passes:
//   pass A  units with at most COMBINE_SOLO parts — the common case of a batch with many moderately long sequences —
//           are folded by ONE wave each, in part order (4 partials in flight); no LDS, no barrier;
//   pass B  units with more parts take the whole workgroup: wave w folds the contiguous range of parts
//           [w*per, (w+1)*per) in order, then wave 0 folds the 16 range results in wave order.
// Either association depends only on the part count, so the result is bitwise reproducible.
constexpr int COMBINE_WAVES_MAX = 16;
constexpr int REDUCE_HINT_NO_EMPTY = 1, REDUCE_HINT_SHORT_SEQS = 2;    // dispatch_reduce's `hints`
constexpr int64_t RANKS_MIN_WAVES = 4096;   // adjacent-rank waves (RANKS) only when B / ranks-per-wave still fills the chip
constexpr int64_t RANKS_MIN_WAVES_SHORT = 512;   // ... of SHORT sequences (dispatch_reduce_main)
constexpr int COMBINE_SOLO = 32;
constexpr int64_t COMBINE_GRID = 512;   // 2 workgroups per CU

template <typename A, int NE, int OP>
__device__ __forceinline__ void combine_range(Fold<A, NE>& f, const A* __restrict__ P, int64_t pbase, int64_t p_lo,
                                              int64_t p_hi, int lane) {
  constexpr bool LSE = op_uses_aux(OP);          // a second value travels with the partial
  constexpr int PF = LSE ? 4 : 8;               // partials in flight: the walk over one unit's parts is latency-bound
  for (int64_t p = p_lo; p < p_hi; p += PF) {
    A a2[PF][NE], x2[LSE ? PF : 1][NE];
    // unconditional loads (the index is clamped, the merge below is guarded): all PF loads are issued before
    // the first wait — guarded loads were serialised into PF round trips
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int64_t pu = (p + u < p_hi) ? p + u : p_hi - 1;
      const A* pp = P + ((pbase + pu) * 2 * RUA_WAVE + lane) * NE;
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        a2[u][k] = pp[k];
        if (LSE) x2[u][k] = pp[RUA_WAVE * NE + k];
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (p + u < p_hi) fold_merge<A, NE, OP>(f, a2[u], x2[LSE ? u : 0]);
  }
}

template <typename T, int EPL, int OP, int CPW>
__global__ __launch_bounds__(RUA_WAVE * (COMBINE_WAVES_MAX / CPW)) void seg_reduce_combine_kernel(
    rua_layout L, const int64_t* __restrict__ perm, T* __restrict__ out, int64_t H, int lp_log2, int include_self,
    T empty_val, rua_layout CD, int copy_mode, SplitWs W) {
  using A = typename elem<T>::acc;
  constexpr int NE = EPL * CPW;
  constexpr int COMBINE_WAVES = COMBINE_WAVES_MAX / CPW;   // LDS: 2 * waves * 64 * NE accumulators
  __shared__ A s_acc[COMBINE_WAVES][RUA_WAVE * NE];
  __shared__ A s_aux[COMBINE_WAVES][RUA_WAVE * NE];
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  const int64_t n_long = (int64_t)(W.ctr[0] >> 32);
  const A* P = reinterpret_cast<const A*>(W.partials);

  // pass A: one wave per unit with few parts
  for (int64_t j = (int64_t)blockIdx.x * COMBINE_WAVES + wave; j < n_long; j += (int64_t)gridDim.x * COMBINE_WAVES) {
    const int64_t* e = W.long_list + j * 4;
    const int64_t nparts = e[2], pbase = e[3];
    if (nparts > COMBINE_SOLO) continue;
    Fold<A, NE> f;
    fold_init<A, NE, OP>(f);
    combine_range<A, NE, OP>(f, P, pbase, 0, nparts, lane);
    const Unit<T, EPL> U = copy_mode ? make_unit<T, EPL, true, CPW>(L, CD, perm, e[0], e[1], H, lp_log2, lane)
                                     : make_unit<T, EPL, false, CPW>(L, CD, perm, e[0], e[1], H, lp_log2, lane);
    fold_store<T, EPL, OP, CPW>(U, f, out, H, include_self, empty_val, (A*)W.ties);
  }

  // pass B: the whole workgroup per unit with many parts (block-uniform loop and condition)
  for (int64_t j = blockIdx.x; j < n_long; j += gridDim.x) {
    const int64_t* e = W.long_list + j * 4;
    const int64_t nparts = e[2], pbase = e[3];
    if (nparts <= COMBINE_SOLO) continue;
    __syncthreads();                                             // s_acc / s_aux are reused per unit
    const int64_t per = (nparts + COMBINE_WAVES - 1) / COMBINE_WAVES;
    const int64_t p_lo = wave * per, p_hi = (p_lo + per < nparts) ? p_lo + per : nparts;
    Fold<A, NE> f;
    fold_init<A, NE, OP>(f);
    combine_range<A, NE, OP>(f, P, pbase, p_lo, p_hi, lane);
#pragma unroll
    for (int k = 0; k < NE; ++k) { s_acc[wave][lane * NE + k] = f.acc[k]; s_aux[wave][lane * NE + k] = f.aux[k]; }
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < COMBINE_WAVES; ++w) {
        if (w * per >= nparts) break;     // ranges beyond the last part are empty
        A a2[NE], x2[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) { a2[k] = s_acc[w][lane * NE + k]; x2[k] = s_aux[w][lane * NE + k]; }
        fold_merge<A, NE, OP>(f, a2, x2);
      }
      const Unit<T, EPL> U = copy_mode ? make_unit<T, EPL, true, CPW>(L, CD, perm, e[0], e[1], H, lp_log2, lane)
                                       : make_unit<T, EPL, false, CPW>(L, CD, perm, e[0], e[1], H, lp_log2, lane);
      fold_store<T, EPL, OP, CPW>(U, f, out, H, include_self, empty_val, (A*)W.ties);
    }
  }
}

// ---------------------------------------------------------------- backward of the reductions
// One wave per (sequence, column chunk), same row addressing as the forward.  grad_in[row] is
//   SUM  g            MEAN g / len           PROD g * out / x        LOGSUMEXP g * exp(x - out)
//   MAX/MIN  (x == out) ? g / ties : 0   — ties counted in a first walk over the sequence (the rows are
//   re-read from L2 by the second walk), so tied maxima share the gradient equally like torch's
//   segment_reduce backward.  Rows of padded layouts that hold no token are not written (the caller
//   zero-fills padded grads).
constexpr int UNROLL_B = 4;

// rows [t_lo, t_hi) of one unit.  MAX/MIN need the tie count of the WHOLE sequence:
//   TIES == 0  self-contained: a first walk over the whole sequence counts, a second writes (t_lo/t_hi = all);
//   TIES == 1  count only: this part's ties are added into ties[b, :] (integer-valued float atomics: exact and
//              order-independent), nothing is written to grad_in;
//   TIES == 2  apply only: ties[b, :] is complete (possibly including the old destination row of a
//              scatter_max/min with include_self); one walk writes the gradient.
//   RANKS: as in the forward (make_unit) — the wave's row groups are adjacent ranks of a PackedSequence walking the
//   same time steps; every group is its own sequence (per-lane len, no cross-group combine).
// `extra_count` of the backward kernels: bit 0 = the old destination row of a scatter_* took part (MEAN's divisor,
// PROD's factors); bit 1 = RUA_BWD_TIES_POSITIVE
constexpr int BWD_SELF_COUNTS = 1, BWD_TIES_POSITIVE = 2;

// the share of one of `c` tied extrema in the gradient g.  torch.segment_reduce's backward hands every tie the whole
// g and then divides only the entries that are > 0 (SegmentReduce.cpp: `if (grad_input > 0) grad_input /= counter`),
// so a negative or NaN g reaches every tie undivided; index_reduce's backward (scatter_max/min) divides always.
template <typename A>
__device__ __forceinline__ A tie_share(A g, A c, bool positive_only) {
  if (positive_only && !(g > (A)0)) return g;
  return g / (c > (A)0 ? c : (A)1);
}

template <typename T, int EPL, int OP, int TIES = 0, bool RANKS = false>
__device__ __forceinline__ void backward_unit(const Unit<T, EPL>& U, int64_t t_lo, int64_t t_hi,
                                              const T* __restrict__ data, const T* __restrict__ out,
                                              const T* __restrict__ gout, T* __restrict__ gin, int64_t H,
                                              int extra_count, int lane,
                                              typename elem<T>::acc* __restrict__ ties = nullptr,
                                              const T* __restrict__ self_in = nullptr) {
  using A = typename elem<T>::acc;
  struct alignas(sizeof(T) * EPL) Pack { T v[EPL]; };
  const int64_t b = U.b, col = U.col, len = U.len, base = U.base, tb = U.tb;
  const int64_t* __restrict__ tbl = U.tbl;
  const int rpw = U.rpw, rsub = U.rsub, lp_log2 = U.lp_log2;
  const bool colok = U.colok;

  A o[EPL], g[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) { o[e] = (A)0; g[e] = (A)0; }
  if (colok) {
    const Pack po = *reinterpret_cast<const Pack*>(out + b * H + col);
    const Pack pg = *reinterpret_cast<const Pack*>(gout + b * H + col);
#pragma unroll
    for (int e = 0; e < EPL; ++e) { o[e] = elem<T>::up(po.v[e]); g[e] = elem<T>::up(pg.v[e]); }
  }
  if (OP == RUA_MEAN) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) g[e] = g[e] / (A)(len + (extra_count & BWD_SELF_COUNTS));   // +1: the old row took part (scatter_mean, include_self)
  }

  // PROD with zeros: d/dx_i = g * prod_{j != i} x_j, which g*out/x cannot give when x_i == 0.  Whole-sequence
  // launches (t_lo == 0 && t_hi == len; PROD is never split by launch_backward) take a first walk that counts the
  // zeros of every column and multiplies the non-zero factors, exactly torch's special case.
  A zeros[EPL];   // PROD: zero factors per column (pass 0)
#pragma unroll
  for (int e = 0; e < EPL; ++e) zeros[e] = (A)0;
  const bool whole = RANKS || (t_lo == 0 && t_hi >= len);
  const bool two_pass = ((OP == RUA_MAX || OP == RUA_MIN) && TIES != 2) || (OP == RUA_PROD && whole);
  // scatter_* (perm) with the old destination row `self_in`: for MAX/MIN that row is one more candidate tie — torch
  // counts it even with include_self=False (index_reduce_backward: N = self_is_result + index_add(source_is_result));
  // for PROD with include_self it is one more factor of the product
  A sv[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) sv[e] = (A)1;
  const bool has_self = self_in != nullptr && colok;
  if (has_self) {
    const Pack ps = *reinterpret_cast<const Pack*>(self_in + b * H + col);
#pragma unroll
    for (int e = 0; e < EPL; ++e) sv[e] = elem<T>::up(ps.v[e]);
  }
  if (TIES == 2 && colok) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      A c = ties[b * H + col + e];
      if (has_self && ((sv[e] == o[e]) || (sv[e] != sv[e] && o[e] != o[e]))) c += (A)1;
      g[e] = tie_share(g[e], c, (extra_count & BWD_TIES_POSITIVE) != 0);
    }
  }
  A nz[EPL];      // PROD: product of the non-zero factors
#pragma unroll
  for (int e = 0; e < EPL; ++e) nz[e] = (A)1;
  for (int pass = two_pass ? 0 : 1; pass < (TIES == 1 ? 1 : 2); ++pass) {
    A cnt[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) cnt[e] = (A)0;
    int64_t tv = (tbl && t_lo + lane < t_hi) ? tbl[tb + t_lo + lane] : 0;
    for (int64_t tblk = t_lo; tblk < t_hi; tblk += RUA_WAVE) {
      const int64_t nxt = tblk + RUA_WAVE + lane;
      const int64_t tv_next = (tbl && nxt < t_hi) ? tbl[tb + nxt] : 0;
      const int nblk = (t_hi - tblk) < RUA_WAVE ? (int)(t_hi - tblk) : RUA_WAVE;
      for (int k = 0; k < nblk; k += (RANKS ? 1 : rpw) * UNROLL_B) {
        int64_t row[UNROLL_B];
        Pack p[UNROLL_B];
#pragma unroll
        for (int u = 0; u < UNROLL_B; ++u) {
          const int tl = RANKS ? k + u : k + u * rpw + rsub;
          const int64_t tabv = __shfl(tv, tl & (RUA_WAVE - 1), RUA_WAVE);
          row[u] = -1;
          if (colok && tl < nblk && (!RANKS || tblk + tl < len)) row[u] = base + (tbl ? tabv : tblk + tl);
          if (row[u] >= U.n_rows) row[u] = -1;
        }
        const bool need_x = (OP != RUA_SUM && OP != RUA_MEAN);
#pragma unroll
        for (int u = 0; u < UNROLL_B; ++u)
          if (need_x && row[u] >= 0) p[u] = *reinterpret_cast<const Pack*>(data + row[u] * H + col);
#pragma unroll
        for (int u = 0; u < UNROLL_B; ++u) {
          if (row[u] < 0) continue;
          Pack r;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const A x = need_x ? elem<T>::up(p[u].v[e]) : (A)0;
            A gi;
            if (OP == RUA_SUM || OP == RUA_MEAN) gi = g[e];
            else if (OP == RUA_PROD) {
              if (pass == 0) { if (x == (A)0) cnt[e] += (A)1; else nz[e] *= x; }
              if (!two_pass || zeros[e] == (A)0) gi = g[e] * o[e] / x;            // no zero in this column
              else if (zeros[e] == (A)1) gi = (x == (A)0) ? g[e] * nz[e] : (A)0;  // the single zero gets it all
              else gi = (A)0;                                                      // two zeros: every product is 0
            }
            else if (OP == RUA_LOGSUMEXP) gi = g[e] * fexp(x - o[e]);
            else {
              const bool hit = (x == o[e]) || (x != x && o[e] != o[e]);
              if (pass == 0) cnt[e] += hit ? (A)1 : (A)0;
              gi = hit ? g[e] : (A)0;
            }
            r.v[e] = elem<T>::down(gi);
          }
          if (pass == 1) *reinterpret_cast<Pack*>(gin + row[u] * H + col) = r;
        }
      }
      tv = tv_next;
    }
    if (pass == 0) {   // combine the row-groups of the wave
      for (int d = RANKS ? RUA_WAVE : (1 << lp_log2); d < RUA_WAVE; d <<= 1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          cnt[e] += __shfl_xor(cnt[e], d, RUA_WAVE);
          if (OP == RUA_PROD) nz[e] *= __shfl_xor(nz[e], d, RUA_WAVE);
        }
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        if (OP == RUA_PROD) {
          zeros[e] = cnt[e];
          if (has_self && (extra_count & BWD_SELF_COUNTS)) { if (sv[e] == (A)0) zeros[e] += (A)1; else nz[e] *= sv[e]; }
        }
        else if (TIES == 1) { if (colok && rsub == 0 && cnt[e] > (A)0) atomicAdd(&ties[b * H + col + e], cnt[e]); }
        else g[e] = tie_share(g[e], cnt[e], (extra_count & BWD_TIES_POSITIVE) != 0);   // MAX/MIN: ties share the gradient
      }
    }
  }
}

// SPLIT: long sequences are cut into parts like in the forward (the parts of a gradient are independent, so
// there is nothing to combine).  MAX/MIN split only in the phased form (TIES 1 then 2: the tie count spans the
// sequence, so it is accumulated in `ties` first); with TIES == 0 they keep whole sequences.
template <typename T, int EPL, int OP, bool SPLIT, int TIES>
__global__ __launch_bounds__(RUA_WAVE) void seg_backward_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                                const T* __restrict__ data,
                                                                const T* __restrict__ out,
                                                                const T* __restrict__ gout, T* __restrict__ gin,
                                                                int64_t H, int lp_log2, int64_t n_chunks,
                                                                int extra_count, SplitWs W,
                                                                typename elem<T>::acc* __restrict__ ties,
                                                                const T* __restrict__ self_in) {
  const int lane = threadIdx.x;
  const int64_t wid = blockIdx.x;
  const int64_t q = wid / n_chunks;
  if (q >= L.B) return;
  const Unit<T, EPL> U = make_unit<T, EPL, false>(L, L, perm, q, wid - q * n_chunks, H, lp_log2, lane);
  if (U.len <= 0) return;
  int64_t t_hi = U.len;
  if (SPLIT && (TIES != 0 || (OP != RUA_MAX && OP != RUA_MIN)) && U.len > W.split) {
    const int64_t nparts = (U.len + W.split - 1) / W.split;
    int64_t ibase = 0;
    if (lane == 0) ibase = (int64_t)atomicAdd(&W.ctr[0], (unsigned long long)(nparts - 1));
    ibase = __shfl(ibase, 0, RUA_WAVE);
    for (int64_t p = 1 + lane; p < nparts; p += RUA_WAVE) {
      int64_t* e = W.items + (ibase + p - 1) * 4;
      e[0] = q; e[1] = U.chunk; e[2] = p; e[3] = 0;
    }
    t_hi = W.split;
  }
  backward_unit<T, EPL, OP, TIES>(U, 0, t_hi, data, out, gout, gin, H, extra_count, lane, ties, self_in);
}

template <typename T, int EPL, int OP, int TIES>
__global__ __launch_bounds__(RUA_WAVE) void seg_backward_tail_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                                     const T* __restrict__ data,
                                                                     const T* __restrict__ out,
                                                                     const T* __restrict__ gout,
                                                                     T* __restrict__ gin, int64_t H, int lp_log2,
                                                                     int extra_count, SplitWs W,
                                                                     typename elem<T>::acc* __restrict__ ties,
                                                                     const T* __restrict__ self_in) {
  const int lane = threadIdx.x;
  const int64_t n_items = (int64_t)W.ctr[0];
  for (int64_t i = blockIdx.x; i < n_items; i += gridDim.x) {
    const int64_t* e = W.items + i * 4;
    const Unit<T, EPL> U = make_unit<T, EPL, false>(L, L, perm, e[0], e[1], H, lp_log2, lane);
    const int64_t t_lo = e[2] * W.split;
    const int64_t t_hi = (t_lo + W.split < U.len) ? t_lo + W.split : U.len;
    backward_unit<T, EPL, OP, TIES>(U, t_lo, t_hi, data, out, gout, gin, H, extra_count, lane, ties, self_in);
  }
}

// [r5] The walk, lean: SUM / MEAN / LOGSUMEXP and MAX / MIN with the forward's tie counts over whole sequences of any
// layout — no row indirection, no old destination row, no PROD, no parts.  backward_unit serves all of those at once and
// pays for it in registers (87 VGPRs for max / logsumexp: FIVE waves per SIMD, and plain stores); here the sequence's
// three rows become two register sets once (what a hit / every element receives, and out or its exp shift), rows are
// 64-bit element offsets, and payload accesses are non-temporal on big payloads like the mover's.  Over a
// PackedSequence at the north-star shape: see DESIGN.md §4.4.
template <typename T, int EPL, int OP, bool NT>
__global__ __launch_bounds__(RUA_WAVE) void seg_backward_walk_kernel(rua_layout L, const T* __restrict__ data,
                                                                     const T* __restrict__ out,
                                                                     const T* __restrict__ gout, T* __restrict__ gin,
                                                                     int64_t H, int lp_log2, int64_t n_chunks,
                                                                     int tie_rule,
                                                                     const typename elem<T>::acc* __restrict__ ties) {
  using A = typename elem<T>::acc;
  struct alignas(sizeof(T) * EPL) Pack { T v[EPL]; };
  typedef unsigned int RawV __attribute__((ext_vector_type(sizeof(T) * EPL >= 4 ? sizeof(T) * EPL / 4 : 1)));
  struct alignas(sizeof(A) * EPL >= 16 ? 16 : sizeof(A) * EPL) Cnt { A v[EPL]; };
  constexpr bool need_x = (OP != RUA_SUM && OP != RUA_MEAN);
  constexpr bool is_ext = (OP == RUA_MAX || OP == RUA_MIN);
  constexpr int UB = UNROLL_B;
  const int lane = threadIdx.x;
  const int64_t wid = blockIdx.x;
  const int64_t q = wid / n_chunks;
  if (q >= L.B) return;
  const Unit<T, EPL> U = make_unit<T, EPL, false>(L, L, nullptr, q, wid - q * n_chunks, H, lp_log2, lane);
  const int64_t len = U.len;
  if (len <= 0) return;
  A f[EPL], o[EPL];
  Pack rs;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { f[e] = (A)0; o[e] = (A)0; rs.v[e] = elem<T>::down((A)0); }
  if (U.colok) {
    const int64_t at = U.b * H + U.col;
    const Pack pg = *reinterpret_cast<const Pack*>(gout + at);
    if (need_x) {
      const Pack po = *reinterpret_cast<const Pack*>(out + at);
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = elem<T>::up(po.v[e]);
    }
    if (is_ext) {
      const Cnt pc = *reinterpret_cast<const Cnt*>(ties + at);
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] = tie_share(elem<T>::up(pg.v[e]), (A)pc.v[e], tie_rule != 0);
    } else {
      const A scale = OP == RUA_MEAN ? (A)1 / (A)len : (A)1;
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] = elem<T>::up(pg.v[e]) * scale;
    }
    if (OP == RUA_LOGSUMEXP) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = exp_shift(o[e]);
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) rs.v[e] = OP == RUA_SUM ? pg.v[e] : elem<T>::down(f[e]);
  }
  const int64_t* __restrict__ tbl = U.tbl;
  const int64_t tb = U.tb, base = U.base, n_rows = U.n_rows;
  const int rpw = U.rpw, rsub = U.rsub;
  const int64_t col = U.col;
  const bool colok = U.colok;
  int64_t tv = (tbl && lane < len) ? tbl[tb + lane] : 0;
  for (int64_t tblk = 0; tblk < len; tblk += RUA_WAVE) {
    const int64_t nxt = tblk + RUA_WAVE + lane;
    const int64_t tv_next = (tbl && nxt < len) ? tbl[tb + nxt] : 0;
    const int nblk = (len - tblk) < RUA_WAVE ? (int)(len - tblk) : RUA_WAVE;
    for (int k = 0; k < nblk; k += rpw * UB) {
      int64_t at[UB];
      Pack px[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int tl = k + u * rpw + rsub;
        const int64_t tabv = __shfl(tv, tl & (RUA_WAVE - 1), RUA_WAVE);
        int64_t row = -1;
        if (colok && tl < nblk) row = base + (tbl ? tabv : tblk + tl);
        at[u] = (row >= 0 && row < n_rows) ? row * H + col : -1;
        if (need_x && at[u] >= 0) {
          if (NT && sizeof(Pack) >= 4) {
            RawV raw = __builtin_nontemporal_load(reinterpret_cast<const RawV*>(data + at[u]));
            __builtin_memcpy(&px[u], &raw, sizeof(Pack));
          } else {
            px[u] = *reinterpret_cast<const Pack*>(data + at[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (at[u] < 0) continue;
        if (need_x) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const A x = elem<T>::up(px[u].v[e]);
            A gi;
            if (OP == RUA_LOGSUMEXP) gi = f[e] * exp_shifted(x, o[e]);
            else gi = ((x == o[e]) || (x != x && o[e] != o[e])) ? f[e] : (A)0;
            rs.v[e] = elem<T>::down(gi);
          }
        }
        if (NT && sizeof(Pack) >= 4) {
          RawV raw;
          __builtin_memcpy(&raw, &rs, sizeof(Pack));
          __builtin_nontemporal_store(raw, reinterpret_cast<RawV*>(gin + at[u]));
        } else {
          *reinterpret_cast<Pack*>(gin + at[u]) = rs;
        }
      }
    }
    tv = tv_next;
  }
}

// [r5] Measured and NOT kept: the same backward on (rank x time) tiles — sixteen neighbouring ranks x 8 / 32 / 64 time
// steps per workgroup, the sixteen sequences' rows fetched once into LDS, then runs of sixteen consecutive storage rows
// swept step by step — i.e. the PackedSequence in its own storage order.  In-process A/B at the north-star shape
// (profiles/r05_backward_ab_tiles.txt): sum 3.25-3.33 ms against the walk's 3.25, max 7.8-8.5 against 6.8, logsumexp
// 7.3-7.6 against 6.8, with or without a span of tiles per XCD.  Sweeping the storage row by row with the sequence's rows
// gathered per row (the mover's ZERO map) writes at 3.0 TB/s.  A contiguous span of RANKS per XCD (spans of equal row
// counts, so that an XCD writes one run of rows per time step): sum +2 %, max -1 %, logsumexp -2 %
// (profiles/r05_backward_ab_xcd_spans.txt).  The walk stays, ranks dealt round-robin.
// backward over a PackedSequence with narrow rows: adjacent ranks side by side (see seg_reduce_ranks_kernel)
template <typename T, int EPL, int OP, int TIES>
__global__ __launch_bounds__(RUA_WAVE) void seg_backward_ranks_kernel(rua_layout L, const T* __restrict__ data,
                                                                      const T* __restrict__ out,
                                                                      const T* __restrict__ gout,
                                                                      T* __restrict__ gin, int64_t H, int lp_log2,
                                                                      int tie_rule,
                                                                      typename elem<T>::acc* __restrict__ ties) {
  const int lane = threadIdx.x;
  const Unit<T, EPL> U = make_unit<T, EPL, false, 1, true>(L, L, nullptr, blockIdx.x, 0, H, lp_log2, lane);
  int64_t t_hi = U.len;                       // the wave walks to its longest sequence
#pragma unroll
  for (int d = RUA_WAVE / 2; d > 0; d >>= 1) {
    const int64_t o = __shfl_xor(t_hi, d, RUA_WAVE);
    t_hi = o > t_hi ? o : t_hi;
  }
  if (t_hi <= 0) return;
  backward_unit<T, EPL, OP, TIES, true>(U, 0, t_hi, data, out, gout, gin, H, tie_rule, lane, ties);
}

// ---------------------------------------------------------------- backward, one storage row at a time
// For SUM / MEAN / LOGSUMEXP and MAX / MIN with the forward's tie counts (TIES_FINAL) the gradient of a row depends
// on that row and on three per-sequence rows (out[b], grad_out[b], ties[b]) only — no walk over the sequence.  So the
// backward is laid out like the row mover: a workgroup takes ~16 KiB of consecutive storage rows of the gradient
// (tiles in launch order, one span of tiles per XCD on big launches), the wave resolves rows -> sequences
// cooperatively (coop_resolve), then streams x in and the gradient out; the per-sequence rows come from L1 / L2
// (every row of a sequence asks for the same ones).  No sequence-length imbalance, stores sweep the buffer in
// order, and rows of padded layouts that hold no token are written as zeros in the same pass (the caller need not
// pre-zero the gradient).  Walk-per-sequence form (seg_backward_kernel): max over C 4.7 TB/s at 1 KiB rows.
constexpr int BROWS_MAX = 256;
// (8 waves per SIMD, forced: max / min sit ON the 64-VGPR line and the allocator lands on 63 .. 66 from one edit to the
// next — 7 waves per SIMD cost 7-11 % at the north-star shape, two spilled dwords cost nothing measurable: r5u/bwd_ab.txt)
template <typename T, int EPL, int OP, bool NT, bool WIDE = false>
__global__ __launch_bounds__(RUA_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void seg_backward_rows_kernel(rua_layout L, const T* __restrict__ data,
                                                                      const T* __restrict__ out,
                                                                      const T* __restrict__ gout, T* __restrict__ gin,
                                                                      int64_t H, int lp_log2, int cpr, int tile_rows,
                                                                      int64_t tiles_per_xcd, int tie_rule,
                                                                      const typename elem<T>::acc* __restrict__ ties) {
  using A = typename elem<T>::acc;
  struct alignas(sizeof(T) * EPL) Pack { T v[EPL]; };
  typedef unsigned int RawV __attribute__((ext_vector_type(sizeof(T) * EPL >= 4 ? sizeof(T) * EPL / 4 : 1)));
  __shared__ int64_t s_b[BROWS_MAX];      // sequence of the row, -1: a padding row (zeros)
  __shared__ A s_scale[BROWS_MAX];        // MEAN: 1 / len

  int64_t tile = blockIdx.x;
  if (tiles_per_xcd > 0) {
    tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= tiles_per_xcd) return;
  }
  const int64_t tile0 = tile * tile_rows;
  const int64_t left = L.n_rows - tile0;
  if (left <= 0) return;
  const int nrows = left < tile_rows ? (int)left : tile_rows;

  // ---- phase 1: storage row -> sequence
  {
    const int i = threadIdx.x;
    const int lane = threadIdx.x & (RUA_WAVE - 1);
    const int w0 = i - lane;
    const int nw = nrows - w0 < RUA_WAVE ? nrows - w0 : RUA_WAVE;
    if (nw > 0) {                                                 // wave-uniform
      const int64_t j = tile0 + i;
      const bool mine = i < nrows;
      int64_t b = -1, t = 0;
      bool token = false, resolved = false;
      if (L.kind == RUA_PACK && L.T > 0 && L.boff) {
        int64_t bt;
        const bool ok = coop_resolve([&](int64_t k) { return L.boff[k]; }, L.T, tile0 + w0, nw, lane, t, bt);
        if (mine && ok) {
          resolved = true;
          const int64_t r = j - bt;
          if (r >= 0 && r < L.B) {
            b = L.sorted ? L.sorted[r] : r;
            token = b >= 0 && b < L.B;
          }
        }
      } else if (L.kind == RUA_CAT && L.off && L.B > 0) {
        int64_t ob;
        const bool ok = coop_resolve([&](int64_t k) { return cat_off(L, k); }, L.B, tile0 + w0, nw, lane, b, ob);
        if (mine && ok) { resolved = true; token = true; }
      }
      if (mine) {
        if (!resolved) token = row_to_token(L, j, b, t);
        s_b[i] = token ? b : -1;
        if (OP == RUA_MEAN) {
          const int64_t len = token ? seq_len(L, b) : 1;
          s_scale[i] = (A)1 / (A)(len > 0 ? len : 1);
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2 [r5].  A wave takes UB CONSECUTIVE row groups of the tile (4 KiB of contiguous stores at 1-KiB rows).
  // Round 4's body kept four rows' addresses as 64-bit pairs, fetched the sequence's rows inside the per-row loop and
  // divided by the tie count per ELEMENT and row: 103-123 VGPRs, i.e. FOUR waves per SIMD — half the loads in flight of
  // the mover, whose rate at that occupancy is this kernel's (8 -> 4 workgroups per CU takes C->P from
  // 6.2 to 5.6 TB/s) — 4.6 TB/s for max, 5.1 for logsumexp at the north-star shape with traffic = 1.000 x algorithmic.
  // Now: the rows of a wave almost always belong to ONE sequence (a wave-uniform test of its first and last row), so
  // the sequence's g / out / ties rows are fetched once through a scalar base, the tie share is divided once, and the
  // payload is addressed as tile base (scalar) + a 32-bit offset.  A wave that straddles a boundary, or holds padding
  // rows of L / R, walks its rows one at a time.
  constexpr int UB = 4;
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  const int rpw = RUA_WAVE >> lp_log2;
  const int rsub = lane >> lp_log2;
  const int col0 = (lane & ((1 << lp_log2) - 1)) * EPL;
  constexpr bool need_x = (OP != RUA_SUM && OP != RUA_MEAN);
  constexpr bool is_ext = (OP == RUA_MAX || OP == RUA_MIN);
  struct alignas(sizeof(A) * EPL >= 16 ? 16 : sizeof(A) * EPL) Cnt { A v[EPL]; };
  const T* __restrict__ xt = data + tile0 * H;          // (wave-uniform: scalar base + 32-bit offsets below)
  T* __restrict__ gt = gin + tile0 * H;
  const int Hi = (int)H;
  auto load_x = [&](unsigned off) -> Pack {
    Pack v;
    if (NT && sizeof(Pack) >= 4) {
      RawV raw = __builtin_nontemporal_load(reinterpret_cast<const RawV*>(xt + off));
      __builtin_memcpy(&v, &raw, sizeof(Pack));
    } else {
      v = *reinterpret_cast<const Pack*>(xt + off);
    }
    return v;
  };
  auto store_g = [&](unsigned off, const Pack& v) {
    if (NT && sizeof(Pack) >= 4) {
      RawV raw;
      __builtin_memcpy(&raw, &v, sizeof(Pack));
      __builtin_nontemporal_store(raw, reinterpret_cast<RawV*>(gt + off));
    } else {
      *reinterpret_cast<Pack*>(gt + off) = v;
    }
  };
  // a unit = (block of UB consecutive row groups, 64-lane column chunk): one per wave at rows up to 1 KiB; at wider rows
  // the waves of a workgroup take the CHUNKS of the same few rows (a 4-row tile of 4-KiB rows: wave w streams chunk w of
  // all four rows), so every wave still has UB loads in flight and holds ONE chunk of the sequence's rows
  const int nblocks = (nrows + UB * rpw - 1) / (UB * rpw);
  for (int unit = wave; unit < nblocks * cpr; unit += RUA_WAVES_PER_BLOCK) {
    const int blk = WIDE ? unit / cpr : unit, c = WIDE ? unit - blk * cpr : 0;      // (WIDE: rows wider than one chunk)
    const int g0 = blk * UB;
    const int rA = g0 * rpw;
    const int rB = (rA + UB * rpw < nrows ? rA + UB * rpw : nrows) - 1;
    const int64_t bA = s_b[rA], bB = s_b[rB];
    const int col = col0 + c * RUA_WAVE * EPL;
    const bool colok = col < Hi;
    if (bA == bB && bA >= 0) {                           // wave-uniform: every row of the block is a token of sequence bA
      const int64_t bs = ((int64_t)__builtin_amdgcn_readfirstlane((int)(bA >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)bA);
      const T* __restrict__ gs = gout + bs * H;
      const T* __restrict__ os = out + bs * H;
      A scale = (A)1;
      if (OP == RUA_MEAN) scale = s_scale[rA];
      Pack px[UB];
      unsigned off[UB];
      bool ok[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int r = (g0 + u) * rpw + rsub;
        ok[u] = colok && r < nrows;
        off[u] = (unsigned)(r * Hi + col);
        if (need_x && ok[u]) px[u] = load_x(off[u]);
      }
      if (!colok) continue;
      const Pack pg = *reinterpret_cast<const Pack*>(gs + col);
      A f[EPL], o[EPL];                                // f: what a hit (max / min), or every element, receives
      Pack po;                                         // (max / min keep `out` PACKED and widen it at the compare: the
      if (need_x) {                                    //  kernel sits on the 64-VGPR line — 8 waves per SIMD or 7)
        po = *reinterpret_cast<const Pack*>(os + col);
        if (!is_ext) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) o[e] = elem<T>::up(po.v[e]);
        }
      }
      if (is_ext) {
        const Cnt pc = *reinterpret_cast<const Cnt*>(ties + bs * H + col);
#pragma unroll
        for (int e = 0; e < EPL; ++e) f[e] = tie_share(elem<T>::up(pg.v[e]), (A)pc.v[e], tie_rule != 0);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) f[e] = elem<T>::up(pg.v[e]) * scale;
      }
      if (OP == RUA_LOGSUMEXP) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = exp_shift(o[e]);
      }
      Pack rs;
      if (!need_x) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) rs.v[e] = OP == RUA_SUM ? pg.v[e] : elem<T>::down(f[e]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (!ok[u]) continue;
        if (need_x) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const A x = elem<T>::up(px[u].v[e]);
            A gi;
            if (OP == RUA_LOGSUMEXP) gi = f[e] * exp_shifted(x, o[e]);
            else {
              const A oe = elem<T>::up(po.v[e]);
              gi = ((x == oe) || (x != x && oe != oe)) ? f[e] : (A)0;
            }
            rs.v[e] = elem<T>::down(gi);
          }
        }
        store_g(off[u], rs);
      }
      continue;
    }
    // a sequence boundary (or padding rows) inside the block's rows: one row group at a time
    if (!colok) continue;
#pragma unroll 1
    for (int u = 0; u < UB; ++u) {
      const int r = (g0 + u) * rpw + rsub;
      if (r >= nrows) continue;
      const int64_t b = s_b[r];
      const unsigned off = (unsigned)(r * Hi + col);
      Pack rs;
      if (b < 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) rs.v[e] = elem<T>::down((A)0);
      } else {
        Pack px;
        if (need_x) px = load_x(off);
        const Pack pg = *reinterpret_cast<const Pack*>(gout + b * H + col);
        Pack po;
        Cnt pc;
        if (need_x) po = *reinterpret_cast<const Pack*>(out + b * H + col);
        if (is_ext) pc = *reinterpret_cast<const Cnt*>(ties + b * H + col);
        A scale = (A)1;
        if (OP == RUA_MEAN) scale = s_scale[r];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const A g = elem<T>::up(pg.v[e]);
          A gi;
          if (OP == RUA_SUM) gi = g;
          else if (OP == RUA_MEAN) gi = g * scale;
          else {
            const A x = elem<T>::up(px.v[e]), o = elem<T>::up(po.v[e]);
            if (OP == RUA_LOGSUMEXP) gi = g * exp_shifted(x, exp_shift(o));
            else gi = ((x == o) || (x != x && o != o)) ? tie_share(g, (A)pc.v[e], tie_rule != 0) : (A)0;
          }
          rs.v[e] = elem<T>::down(gi);
        }
      }
      store_g(off, rs);
    }
  }
}

// ---- the reference's global `initial` written where it shows (see fold_flags)
// extreme scratch (RUA_EXTREME_WORDS = 1 027 words): [0..1023] hashed slots of the opposite extreme (zero-neutral),
// [1024] flags, [1025] fill_empty_kernel's reset ticket, [1026] spare.

// Waves `w0, w0 + nw, ...` of `nw` take 64 sequences each per step: only empty sequences (or everything, when a NaN
// poisoned `initial`) are written.  A lane inspects one sequence, then the wave writes the marked rows together, lanes
// side by side along H (coalesced stores; the poisoned case rewrites the whole [B, H] output).
// Rows of whole 16-byte pieces at 16-byte addresses are written as such (element-wide stores made a mostly-empty batch
// 14 GB/s per workgroup: profiles/r04_shape_cliffs.txt).
template <typename T>
__device__ __forceinline__ void fill_empty_body(const rua_layout& L, T* __restrict__ out, int64_t H,
                                                int want_max_of_data, const unsigned long long* __restrict__ ext,
                                                unsigned long long flags, int64_t w0, int64_t nw, int lane) {
  using A = typename elem<T>::acc;
  // decode the tracked extreme: lane i reads slots i, i + 64, ... (past this CU's L1: other workgroups' atomics wrote
  // them), then a 6-step butterfly
  unsigned long long best = 0ull;
#pragma unroll
  for (int k = 0; k < EXTREME_SLOTS / RUA_WAVE; ++k) {
    const unsigned long long o = __hip_atomic_load(&ext[k * RUA_WAVE + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    best = o > best ? o : best;
  }
#pragma unroll
  for (int d = RUA_WAVE / 2; d > 0; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d, RUA_WAVE);
    best = o > best ? o : best;
  }
  if (!want_max_of_data) best = ~best;
  const bool poison = (flags & 1ull) != 0ull;
  A val;
  if (sizeof(A) == 8) val = (A)unordered_f64(best); else val = (A)unordered_f32(best);
  if (poison) val = val - val + (A)__builtin_nanf("");
  const T tv = elem<T>::down(val);

  constexpr int VE = 16 / (int)sizeof(T);
  const bool wide = (H % VE) == 0 && ((uintptr_t)out & 15) == 0;
  struct alignas(16) Piece { T v[VE]; } piece;
#pragma unroll
  for (int e = 0; e < VE; ++e) piece.v[e] = tv;
  for (int64_t b0 = w0 * RUA_WAVE; b0 < L.B; b0 += nw * RUA_WAVE) {
    const int64_t b = b0 + lane;
    const bool mine = b < L.B && (poison || seq_len(L, b) <= 0);
    if (wide && H / VE <= 8) {           // narrow rows: every lane writes its own sequence's row
      if (mine) {
        Piece* o16 = reinterpret_cast<Piece*>(out + b * H);
        for (int64_t h = 0; h < H / VE; ++h) o16[h] = piece;
      }
      continue;
    }
    unsigned long long todo = __ballot(mine);
    while (todo) {
      const int k = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      T* o = out + (b0 + k) * H;
      if (wide) {
        Piece* o16 = reinterpret_cast<Piece*>(o);
        for (int64_t h = lane; h < H / VE; h += RUA_WAVE) o16[h] = piece;
      } else {
        for (int64_t h = lane; h < H; h += RUA_WAVE) o[h] = tv;
      }
    }
  }
}

// The trailing launch of max / min / logsumexp.  Common case: nothing was raised — every workgroup reads the flag word
// and leaves (the first one wipes the slots on its way: the reduce wrote them).  Otherwise EVERY workgroup patches its
// share of the batch (the global extreme is already in the slots), and the one that finishes last — a ticket — hands
// the scratch back zeroed (`reset`: the caller's persistent buffer, RUA_OP_SCRATCH_CLEAN).
template <typename T>
__global__ __launch_bounds__(RUA_BLOCK) void fill_empty_kernel(rua_layout L, T* __restrict__ out, int64_t H,
                                                               int want_max_of_data,
                                                               unsigned long long* __restrict__ ext, int reset) {
  const unsigned long long flags = ext[EXTREME_SLOTS];
  if (flags == 0ull) {
    // nothing to patch and nobody reads the slots: no ticket (every workgroup used to take one: B / 256 atomics on
    // ONE address, 0.6 ms after a reduce over 8 M short sequences: profiles/r04_cat_ranks_ab.txt)
    if (reset && blockIdx.x == 0)
      for (int i = threadIdx.x; i < EXTREME_SLOTS; i += RUA_BLOCK) ext[i] = 0ull;
    return;
  }
  fill_empty_body<T>(L, out, H, want_max_of_data, ext, flags, (int64_t)blockIdx.x * RUA_WAVES_PER_BLOCK + (threadIdx.x >> 6),
                     (int64_t)gridDim.x * RUA_WAVES_PER_BLOCK, threadIdx.x & (RUA_WAVE - 1));
  if (!reset) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long t = atomicAdd(&ext[EXTREME_SLOTS + 1], 1ull);
    if (t == (unsigned long long)gridDim.x - 1ull) {
      for (int i = 0; i < EXTREME_SLOTS + 3; ++i) ext[i] = 0ull;      // (rare path: one thread, 8 KB)
      __threadfence();
    }
  }
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RUA_BLOCK - 1) / RUA_BLOCK); }

// Gradient of scatter_* w.r.t. the destination `tensor` ([S, H], reduce.py:6-31 through torch's index_reduce /
// index_add backward): elementwise in (tensor, out, grad) plus the per-destination facts the other kernels already
// produced — the bucket size, MAX/MIN: the source rows' tie counts (the forward's ties_out), PROD: the product of
// the bucket's source rows.  One thread per element.
//   include_self:  SUM g | MEAN g/(n+1) | MAX/MIN (tensor == out) ? g/ties : 0 | PROD g*prod(sources) |
//                  LOGSUMEXP g*exp(tensor - out)
//   otherwise:     rows no index names keep `tensor` (index_reduce semantics): g there, 0 elsewhere
template <typename T>
__global__ __launch_bounds__(RUA_BLOCK) void scatter_self_grad_kernel(const int64_t* __restrict__ counts, int64_t S,
                                                                      int64_t H, const T* __restrict__ self_in,
                                                                      const T* __restrict__ out,
                                                                      const T* __restrict__ gout,
                                                                      const void* __restrict__ aux,
                                                                      T* __restrict__ gself, int op, int inc) {
  using A = typename elem<T>::acc;
  const int64_t i = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (i >= S * H) return;
  const int64_t n = counts[i / H];
  const A g = elem<T>::up(gout[i]);
  A r;
  if (!inc) {
    r = n == 0 ? g : (A)0;
  } else if (op == RUA_SUM) {
    r = g;
  } else if (op == RUA_MEAN) {
    r = g / (A)(n + 1);
  } else if (op == RUA_PROD) {
    r = g * elem<T>::up(reinterpret_cast<const T*>(aux)[i]);
  } else {
    const A x = elem<T>::up(self_in[i]), o = elem<T>::up(out[i]);
    if (op == RUA_LOGSUMEXP) {
      r = g * fexp(x - o);
    } else {
      const bool hit = (x == o) || (x != x && o != o);
      const A c = (aux ? reinterpret_cast<const A*>(aux)[i] : (A)0) + (A)1;
      r = hit ? g / c : (A)0;
    }
  }
  gself[i] = elem<T>::down(r);
}

// workspace carving for the long-sequence split (see SplitWs)
static inline int64_t split_max_extra(int64_t n_rows, int64_t split) { return split > 0 ? n_rows / split : 0; }
constexpr int64_t SPLIT_GRID_CAP = 16384;   // tail / combine grids: 2x the wave slots of the chip, then stride
static inline unsigned split_grid(int64_t max_u) { return (unsigned)(max_u < SPLIT_GRID_CAP ? max_u : SPLIT_GRID_CAP); }

template <typename A>
static SplitWs carve_ws(void* ws, int64_t max_u, int64_t split) {
  SplitWs W;
  char* p = (char*)ws;
  W.ctr = (unsigned long long*)p;            p += 4 * sizeof(unsigned long long);
  W.long_list = (int64_t*)p;                 p += (size_t)max_u * 4 * sizeof(int64_t);
  W.items = (int64_t*)p;                     p += (size_t)max_u * 4 * sizeof(int64_t);
  W.partials = (void*)p;
  W.max_u = max_u;
  W.split = split;
  return W;
}

template <typename T, int EPL, bool NT, bool COPY, int CPW>
static int launch_reduce(int op, unsigned grid, hipStream_t s, const rua_layout& L, const int64_t* perm,
                         const void* data, void* out, int64_t H, int lp_log2, int64_t n_chunks, int include_self,
                         uint64_t empty_bits, void* extreme, const rua_layout& CD, void* copy, int64_t split,
                         void* ws, void* ties, int no_empty) {
  using A = typename elem<T>::acc;
  T ev;
  __builtin_memcpy(&ev, &empty_bits, sizeof(T));
  const dim3 g(grid), b(RUA_WAVE);
  const int64_t max_u = split_max_extra(L.n_rows, split) * n_chunks;
  const bool do_split = split > 0 && ws && max_u > 0;
  SplitWs W = {};
  if (do_split) {
    if (max_u > 0x7fffffffLL) return RUA_ERANGE;
    W = carve_ws<A>(ws, max_u, split);
    hipError_t e = hipMemsetAsync(W.ctr, 0, 4 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return (int)e;
  }
  W.ties = ties;
  unsigned long long* ext = (unsigned long long*)extreme;
  // Two INDEPENDENT waves per workgroup over the batch-major layouts (C / L / R): neighbouring sequences are
  // neighbouring storage, and halving the number of workgroups is worth 6-9 % there (cfg3 segment_sum 108.8 -> 101.3 us,
  // north-star segment_sum(c) 2.86 -> 2.70 ms); four are no better, and over a PackedSequence — walked longest sequence
  // first, every rank its own slot — two LOSE 7 % (2.64 -> 2.84 ms): one wave per workgroup stays there
  // (profiles/r04_reduce_wpb_ab.txt, measured with a temporary environment knob).
  // Rows of at least 512 bytes only: at 16 / 32-byte rows (a whole short sequence per wave instruction) two waves per
  // workgroup lose 7-10 % (final width sweep of round 4: 2.36 -> 2.20, 4.40 -> 3.95 TB/s).
  const int wpb = (COPY || CPW != 1) ? 1 : ((L.kind != RUA_PACK && H * (int64_t)sizeof(T) >= 512) ? 2 : 1);
#define RUA_LAUNCH(OP)                                                                                              \
  if (do_split) {                                                                                                   \
    hipLaunchKernelGGL((seg_reduce_kernel<T, EPL, OP, NT, COPY, true, CPW>), g, b, 0, s, L, perm, (const T*)data,   \
                       (T*)out, H, lp_log2, n_chunks, include_self, ev, ext, CD, (T*)copy, W, no_empty);            \
    hipLaunchKernelGGL((seg_reduce_tail_kernel<T, EPL, OP, NT, COPY, CPW>), dim3(split_grid(max_u)), b, 0, s, L, perm, \
                       (const T*)data, H, lp_log2, ext, CD, (T*)copy, W, no_empty);                                 \
    hipLaunchKernelGGL((seg_reduce_combine_kernel<T, EPL, OP, CPW>),                                                \
                       dim3((unsigned)(max_u < COMBINE_GRID ? max_u : COMBINE_GRID)),                               \
                       dim3(RUA_WAVE * (COMBINE_WAVES_MAX / CPW)), 0, s, L, perm, (T*)out,                          \
                       H, lp_log2, include_self, ev, CD, COPY ? 1 : 0, W);                                          \
  } else if (wpb == 2) {                                                                                            \
    hipLaunchKernelGGL((seg_reduce_kernel<T, EPL, OP, NT, COPY, false, CPW, 2>), dim3((grid + 1) / 2), dim3(RUA_WAVE * 2), \
                       0, s, L, perm, (const T*)data, (T*)out, H, lp_log2, n_chunks, include_self, ev, ext, CD, (T*)copy, W, no_empty); \
  } else {                                                                                                          \
    hipLaunchKernelGGL((seg_reduce_kernel<T, EPL, OP, NT, COPY, false, CPW>), g, b, 0, s, L, perm, (const T*)data,  \
                       (T*)out, H, lp_log2, n_chunks, include_self, ev, ext, CD, (T*)copy, W, no_empty);            \
  }
  switch (op) {
    case RUA_SUM: RUA_LAUNCH(RUA_SUM); break;
    case RUA_MEAN: RUA_LAUNCH(RUA_MEAN); break;
    case RUA_MAX:
      if constexpr (!COPY) { if (ties) { RUA_LAUNCH(RUA_MAX_T); break; } }   // also count the ties (for the backward)
      RUA_LAUNCH(RUA_MAX); break;
    case RUA_MIN:
      if constexpr (!COPY) { if (ties) { RUA_LAUNCH(RUA_MIN_T); break; } }
      RUA_LAUNCH(RUA_MIN); break;
    case RUA_PROD: RUA_LAUNCH(RUA_PROD); break;
    case RUA_LOGSUMEXP: RUA_LAUNCH(RUA_LOGSUMEXP); break;
    default: return RUA_EINVAL;
  }
#undef RUA_LAUNCH
  return (int)hipGetLastError();
}

template <typename T>
static int dispatch_reduce_main(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data,
                                void* out, int64_t H, int include_self, uint64_t empty_bits, void* extreme,
                                int64_t split, void* ws, const rua_layout* CD = nullptr, void* copy = nullptr,
                                void* ties = nullptr, bool short_seqs = false, int no_empty = 0) {
  constexpr int FULL = 16 / sizeof(T);
  constexpr int HALF = FULL >= 4 ? FULL / 2 : 1;      // 8-byte loads: hidden sizes that are a multiple of 8 bytes only
  const uintptr_t fptrs = (uintptr_t)data | (uintptr_t)out | (uintptr_t)copy | (uintptr_t)ties;
  const bool aligned_ok = (H % FULL == 0) && (fptrs % 16 == 0);
  // Rows of 8 (mod 16) bytes (H = 500 in bf16): every other row starts on an 8-byte boundary only and the last lane of
  // a row would hold half a vector.  gfx950 takes a dwordx4 at any dword-aligned address, and the LAST lane simply
  // covers the last FULL elements of the row, overlapping its neighbour by half a vector: both lanes fold the same
  // elements in the same order and store the same results (make_unit clamps the column).  Round 2 used 8-byte lanes
  // there — two column chunks, two waves per row, 4.0 TB/s for segment_max over a CattedSequence at H = 500.
  // (Not with include_self == 1: the store then reads the old row, and two lanes would fold it in twice.)
  const bool tail_ok = !aligned_ok && FULL > 1 && H > FULL && (H + FULL - 1) / FULL <= RUA_WAVE &&
                       (H * (int64_t)sizeof(T)) % 8 == 0 && (fptrs % 8 == 0) && include_self != 1 && !copy;
  const bool vec_ok = aligned_ok || tail_ok;
  // (H = 300 or 650 in bf16 — GloVe vectors, PTB-sized LSTMs: the scalar path moves 128 B per wave instruction and
  // measured 2.9 TB/s; 8-byte lanes move 512 B)
  const bool half_ok = !vec_ok && HALF > 1 && (H % HALF == 0) && (fptrs % 8 == 0);
  const int epl = vec_ok ? FULL : half_ok ? HALF : 1;
  const int64_t lpr = (H + epl - 1) / epl;  // lanes per row
  int lp_log2 = 0;
  while ((1 << lp_log2) < lpr && lp_log2 < 6) ++lp_log2;
  // rows wider than one wave instruction (1 KiB): one wave owns 4 column chunks, i.e. up to 4 KiB of the row
  // (the same for 8-byte lanes: H = 500 in bf16 is 125 lanes — as two column chunks two waves each read every other
  // 512-byte half of the rows: reduce over P at H = 500 5.0 -> 5.7 TB/s
  // — over a PackedSequence only: over the batch-major layouts, whose sequences are contiguous, the two waves per row
  // were the better half of the bytes in flight: 4.0 -> 2.7 TB/s when tried)
  const bool wide = (vec_ok || (half_ok && L.kind == RUA_PACK)) && lpr > RUA_WAVE;
  const int cpw = wide ? 4 : 1;
  const int64_t n_chunks = (lpr + RUA_WAVE * cpw - 1) / (RUA_WAVE * cpw);
  const int64_t blocks = L.B * n_chunks;  // one wave per workgroup
  if (blocks > 0x7fffffffLL) return RUA_ERANGE;
  const bool nt = (double)L.n_rows * (double)H * sizeof(T) >= (double)(512ll << 20);
  const unsigned g = (unsigned)blocks;
  static const rua_layout none = {};
  const rua_layout& cd = copy ? *CD : none;
  if (copy && !aligned_ok) return RUA_EALIGN;   // fused pack + reduce: vector path only (caller falls back to two launches)
  // (only when that still leaves >= 4 waves per SIMD: with fewer sequences one wave per sequence fills the chip better)
  // (RUA_OP_SHORT_SEQS: the caller knows the longest sequence and vouches that none is far above the average — the
  // wave walks to the longest of its sequences, so ONE long sequence among short ones would be walked by one lane group)
  bool cat_ranks = L.kind == RUA_CAT && L.lens && L.len_add == 0 && lp_log2 < 6;
  int glog = 0;                                   // log2 of the row slots of one sequence's lane group (make_unit)
  if (cat_ranks) {
    const int64_t side = RUA_WAVE >> lp_log2;     // sequences side by side with one row slot each
    const int64_t short_avg = 4 * side < 16 ? 16 : (4 * side > 64 ? 64 : 4 * side);
    if (L.n_rows > short_avg * L.B) {             // longer than that on average:
      if (lp_log2 <= 1) glog = 4 - lp_log2;       // FOUR sequences per wave at rows of <= 32 bytes (16 / 8 rows of each
      else cat_ranks = false;                     // per instruction; the wave checks its own lengths), else one wave each
    }                                             // (short on average, no word about the longest: the waves check, `check`)
    // ([r5] a batch too small to fill the chip with every row slot its own sequence still fills it four to a wave at
    // rows of <= 32 bytes)
    if (cat_ranks && glog == 0 && lp_log2 <= 1 && (L.B >> (6 - lp_log2)) < RANKS_MIN_WAVES_SHORT) glog = 4 - lp_log2;
  }
  // How many side-by-side waves are enough: RANKS_MIN_WAVES when the sequences may be long (a wave then walks hundreds of
  // steps one after the other, and only plenty of them keep the chip busy); a few hundred when they are SHORT — a
  // CattedSequence that is short on average, a PackedSequence of at most 128 time steps — where the alternative is one
  // wave per sequence at one row per instruction: 200 000 x U(1,32) rows of 16 bytes are 3 125 waves of 64 sequences
  const bool short_form = (cat_ranks && glog == 0) || (L.kind == RUA_PACK && L.T > 0 && L.T <= 128);
  const int64_t ranks_min_waves = short_form ? RANKS_MIN_WAVES_SHORT : RANKS_MIN_WAVES;
  // ([r5] with the long-sequence split armed — lengths nobody vouches for — the four-per-wave form splits by itself)
  const int ranks_check = (cat_ranks && !short_seqs) ? 1 : 0;
  const bool ranks_split = split > 0 && ws && cat_ranks && (glog > 0 || ranks_check) && vec_ok && split_max_extra(L.n_rows, split) > 0;
  if (((L.kind == RUA_PACK && L.sorted) || cat_ranks) && !copy && !perm && lp_log2 < 6 && (!(split > 0 && ws) || ranks_split) &&
      (L.B >> (6 - lp_log2 - glog)) >= ranks_min_waves) {
    // narrow rows of a PackedSequence: adjacent ranks share a wave instruction
    // (tried for a CattedSequence with 32-byte rows too — groups = adjacent sequences: 4.0 -> 2.8 TB/s at U(8,512),
    // unsorted neighbours differ too much in length — so C keeps one wave per sequence, EXCEPT for batches of short
    // sequences, when the caller says so: there a wave per sequence is bound by the rate at which workgroups can be
    // dispatched at all — 4 M singletons: 2.98 ms, 1.3 workgroups per ns; 500 000 sequences of 16 rows on average:
    // 0.40 -> 0.06-0.07 ms at 16 / 32-byte rows, 0.43 -> 0.21 at 128: profiles/r04_cat_ranks_ab.txt.  And at rows of
    // <= 32 bytes a whole sequence of a few hundred rows is a handful of wave instructions behind a chain of dependent
    // loads — 2.2 / 4.1 TB/s at 16 / 32 bytes with U(8,512) lengths — so there FOUR sequences share a wave, sixteen /
    // eight rows of each per instruction.  Every row slot its own sequence only under the caller's word that no sequence
    // is far above the average, because the wave walks to the longest of its sequences; the four-per-wave form checks
    // that by itself, wave by wave (seg_reduce_ranks_kernel), so it also serves lengths that live on the device only)
    const int64_t rpw = RUA_WAVE >> (lp_log2 + glog);
    const int64_t nblk = (L.B + rpw - 1) / rpw;
    if (nblk > 0x7fffffffLL) return RUA_ERANGE;
    T ev;
    __builtin_memcpy(&ev, &empty_bits, sizeof(T));
    const dim3 gg((unsigned)nblk), bb(RUA_WAVE);
    unsigned long long* ext = (unsigned long long*)extreme;
    SplitWs W = {};
    const int64_t max_u = ranks_split ? split_max_extra(L.n_rows, split) : 0;      // (one column chunk per row here)
    if (ranks_split) {
      if (max_u > 0x7fffffffLL) return RUA_ERANGE;
      W = carve_ws<typename elem<T>::acc>(ws, max_u, split);
      const hipError_t e = hipMemsetAsync(W.ctr, 0, 4 * sizeof(unsigned long long), s);
      if (e != hipSuccess) return (int)e;
      W.ties = ties;
    }
#define RUA_RANKS(EPLV, NTV, OPV)                                                                                  \
  do { if (ranks_split) {                                                                                               \
    if constexpr (EPLV == FULL) {                                                                                  \
      hipLaunchKernelGGL((seg_reduce_ranks_kernel<T, EPLV, OPV, NTV, true>), gg, bb, 0, s, L, (const T*)data, (T*)out, H, \
                         lp_log2, include_self, ev, ext, (typename elem<T>::acc*)ties, glog, W, no_empty, ranks_check); \
      hipLaunchKernelGGL((seg_reduce_tail_kernel<T, EPLV, OPV, NTV, false, 1>), dim3(split_grid(max_u)), bb, 0, s, L, \
                         (const int64_t*)nullptr, (const T*)data, H, lp_log2, ext, L, (T*)nullptr, W, no_empty);   \
      hipLaunchKernelGGL((seg_reduce_combine_kernel<T, EPLV, OPV, 1>),                                             \
                         dim3((unsigned)(max_u < COMBINE_GRID ? max_u : COMBINE_GRID)),                            \
                         dim3(RUA_WAVE * COMBINE_WAVES_MAX), 0, s, L, (const int64_t*)nullptr, (T*)out, H, lp_log2, \
                         include_self, ev, L, 0, W);                                                               \
    }                                                                                                              \
  } else                                                                                                           \
  hipLaunchKernelGGL((seg_reduce_ranks_kernel<T, EPLV, OPV, NTV>), gg, bb, 0, s, L, (const T*)data, (T*)out, H,    \
                     lp_log2, include_self, ev, ext, (typename elem<T>::acc*)ties, glog, W, no_empty, ranks_check); } while (0)
#define RUA_RANKS_OP(EPLV, NTV)                                  \
  switch (op) {                                                  \
    case RUA_SUM: RUA_RANKS(EPLV, NTV, RUA_SUM); break;          \
    case RUA_MEAN: RUA_RANKS(EPLV, NTV, RUA_MEAN); break;        \
    case RUA_MAX: if (ties) RUA_RANKS(EPLV, NTV, RUA_MAX_T); else RUA_RANKS(EPLV, NTV, RUA_MAX); break; \
    case RUA_MIN: if (ties) RUA_RANKS(EPLV, NTV, RUA_MIN_T); else RUA_RANKS(EPLV, NTV, RUA_MIN); break; \
    case RUA_PROD: RUA_RANKS(EPLV, NTV, RUA_PROD); break;        \
    case RUA_LOGSUMEXP: RUA_RANKS(EPLV, NTV, RUA_LOGSUMEXP); break; \
    default: return RUA_EINVAL;                                  \
  }
    if (vec_ok) { if (nt) { RUA_RANKS_OP(FULL, true) } else { RUA_RANKS_OP(FULL, false) } }
    else if (half_ok) { RUA_RANKS_OP(HALF, false) }
    else { RUA_RANKS_OP(1, false) }
#undef RUA_RANKS_OP
#undef RUA_RANKS
    return (int)hipGetLastError();
  }
  // few-but-long units: a team of waves per unit (seg_reduce_team_kernel) — vector path, rows up to 1 KiB, no split
  if (vec_ok && !wide && !copy && !(split > 0 && ws)) {
    const int team = reduce_team_waves(L.n_rows, L.B, lp_log2, blocks, UNROLL_T);   // (rua_dev.h: the one rule)
    if (team > 1) {
      T ev;
      __builtin_memcpy(&ev, &empty_bits, sizeof(T));
      const dim3 gg((unsigned)blocks), bb(RUA_WAVE * team);
      unsigned long long* ext = (unsigned long long*)extreme;
      using A = typename elem<T>::acc;
#define RUA_TEAM(NTV, OPV)                                                                                           \
  hipLaunchKernelGGL((seg_reduce_team_kernel<T, FULL, OPV, NTV>), gg, bb, 0, s, L, perm, (const T*)data, (T*)out, H, \
                     lp_log2, n_chunks, include_self, ev, ext, (A*)ties, no_empty)
#define RUA_TEAM_OP(NTV)                                                                           \
  switch (op) {                                                                                    \
    case RUA_SUM: RUA_TEAM(NTV, RUA_SUM); break;                                                   \
    case RUA_MEAN: RUA_TEAM(NTV, RUA_MEAN); break;                                                 \
    case RUA_MAX: if (ties) RUA_TEAM(NTV, RUA_MAX_T); else RUA_TEAM(NTV, RUA_MAX); break;          \
    case RUA_MIN: if (ties) RUA_TEAM(NTV, RUA_MIN_T); else RUA_TEAM(NTV, RUA_MIN); break;          \
    case RUA_PROD: RUA_TEAM(NTV, RUA_PROD); break;                                                 \
    case RUA_LOGSUMEXP: RUA_TEAM(NTV, RUA_LOGSUMEXP); break;                                       \
    default: return RUA_EINVAL;                                                                    \
  }
      if (nt) { RUA_TEAM_OP(true) } else { RUA_TEAM_OP(false) }
#undef RUA_TEAM_OP
#undef RUA_TEAM
      return (int)hipGetLastError();
    }
  }
#define RUA_GO(EPLV, NTV, COPYV, CPWV)                                                                             \
  return launch_reduce<T, EPLV, NTV, COPYV, CPWV>(op, g, s, L, perm, data, out, H, lp_log2, n_chunks, include_self, \
                                                  empty_bits, extreme, cd, copy, split, ws, ties, no_empty)
  if (copy) {
    if (wide) { if (nt) RUA_GO(FULL, true, true, 4); else RUA_GO(FULL, false, true, 4); }      // (copy: vec_ok)
    if (nt) RUA_GO(FULL, true, true, 1); else RUA_GO(FULL, false, true, 1);
  }
  if (wide && vec_ok) { if (nt) RUA_GO(FULL, true, false, 4); else RUA_GO(FULL, false, false, 4); }
  if (wide && half_ok) RUA_GO(HALF, false, false, 4);
  if (vec_ok) { if (nt) RUA_GO(FULL, true, false, 1); else RUA_GO(FULL, false, false, 1); }
  if (half_ok) RUA_GO(HALF, false, false, 1);
  RUA_GO(1, false, false, 1);
#undef RUA_GO
}


// (rounds 1-4 launched a conditional second walk for the global extreme behind the reduce; the reduce now tracks it
// itself — fold_flags — so this is the reduce and nothing else)
template <typename T>
static int dispatch_reduce(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data,
                           void* out, int64_t H, int include_self, uint64_t empty_bits, void* extreme,
                           int64_t split, void* ws, const rua_layout* CD = nullptr, void* copy = nullptr,
                           void* ties = nullptr, int hints = 0) {
  return dispatch_reduce_main<T>(op, s, L, perm, data, out, H, include_self, empty_bits, extreme, split, ws, CD, copy,
                                 ties, (hints & REDUCE_HINT_SHORT_SEQS) != 0, (hints & REDUCE_HINT_NO_EMPTY) ? 1 : 0);
}

template <typename T, int EPL>
static int launch_backward(int op, unsigned grid, hipStream_t s, const rua_layout& L, const int64_t* perm,
                           const void* data, const void* out, const void* gout, void* gin, int64_t H, int lp_log2,
                           int64_t n_chunks, int extra_count, int64_t split, void* ws, void* ties,
                           bool ties_final, const void* self_in) {
  using A = typename elem<T>::acc;
  const dim3 g(grid), b(RUA_WAVE);
  const T* sp = (const T*)self_in;
  const bool extreme_op = op == RUA_MAX || op == RUA_MIN;
  const bool phased = extreme_op && ties != nullptr;        // count phase, then apply phase
  // ties_final: the forward already counted them (RUA_MAX_T / RUA_MIN_T) -> the apply phase alone, ONE walk
  const int64_t max_u = split_max_extra(L.n_rows, split) * n_chunks;
  // PROD keeps whole sequences: its zero-factor special case needs the zero count and the product of the other
  // factors of the whole sequence, and a product combined by atomics would not be reproducible
  const bool do_split = split > 0 && ws && max_u > 0 && (!extreme_op || phased) && op != RUA_PROD;
  SplitWs W = {};
  if (do_split) {
    if (max_u > 0x7fffffffLL) return RUA_ERANGE;
    W = carve_ws<A>(ws, max_u, split);
  }
  A* tp = (A*)ties;
#define RUA_PHASE(OP, TIESV)                                                                                       \
  {                                                                                                                \
    if (do_split) {                                                                                                \
      hipError_t e_ = hipMemsetAsync(W.ctr, 0, 4 * sizeof(unsigned long long), s);                                 \
      if (e_ != hipSuccess) return (int)e_;                                                                        \
      hipLaunchKernelGGL((seg_backward_kernel<T, EPL, OP, true, TIESV>), g, b, 0, s, L, perm, (const T*)data,      \
                         (const T*)out, (const T*)gout, (T*)gin, H, lp_log2, n_chunks, extra_count, W, tp, sp);    \
      hipLaunchKernelGGL((seg_backward_tail_kernel<T, EPL, OP, TIESV>), dim3(split_grid(max_u)), b, 0, s, L, perm, \
                         (const T*)data, (const T*)out, (const T*)gout, (T*)gin, H, lp_log2, extra_count, W, tp, sp); \
    } else {                                                                                                       \
      hipLaunchKernelGGL((seg_backward_kernel<T, EPL, OP, false, TIESV>), g, b, 0, s, L, perm, (const T*)data,     \
                         (const T*)out, (const T*)gout, (T*)gin, H, lp_log2, n_chunks, extra_count, W, tp, sp);    \
    }                                                                                                              \
  }
#define RUA_EXTREME(OP)                                       \
  if (phased && ties_final) RUA_PHASE(OP, 2)                  \
  else if (phased) { RUA_PHASE(OP, 1) RUA_PHASE(OP, 2) } else RUA_PHASE(OP, 0)
  switch (op) {
    case RUA_SUM: RUA_PHASE(RUA_SUM, 0); break;
    case RUA_MEAN: RUA_PHASE(RUA_MEAN, 0); break;
    case RUA_MAX: RUA_EXTREME(RUA_MAX); break;
    case RUA_MIN: RUA_EXTREME(RUA_MIN); break;
    case RUA_PROD: RUA_PHASE(RUA_PROD, 0); break;
    case RUA_LOGSUMEXP: RUA_PHASE(RUA_LOGSUMEXP, 0); break;
    default: return RUA_EINVAL;
  }
#undef RUA_EXTREME
#undef RUA_PHASE
  return (int)hipGetLastError();
}

template <typename T>
static int dispatch_backward(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data,
                             const void* out, const void* gout, void* gin, int64_t H, int extra_count,
                             int64_t split, void* ws, void* ties, bool ties_final, const void* self_in,
                             bool fill_padding) {
  constexpr int FULL = 16 / sizeof(T);
  const uintptr_t ptrs = (uintptr_t)data | (uintptr_t)out | (uintptr_t)gout | (uintptr_t)gin | (uintptr_t)ties |
                         (uintptr_t)self_in;
  constexpr int HALF = FULL >= 4 ? FULL / 2 : 1;
  const bool vec_ok = (H % FULL == 0) && (ptrs % 16 == 0);
  const bool half_ok = !vec_ok && HALF > 1 && (H % HALF == 0) && (ptrs % 8 == 0);
  const int epl = vec_ok ? FULL : half_ok ? HALF : 1;
  const int64_t lpr = (H + epl - 1) / epl;
  int lp_log2 = 0;
  while ((1 << lp_log2) < lpr && lp_log2 < 6) ++lp_log2;
  const int64_t n_chunks = (lpr + RUA_WAVE - 1) / RUA_WAVE;
  const int64_t blocks = L.B * n_chunks;
  if (blocks > 0x7fffffffLL) return RUA_ERANGE;
  // one storage row at a time (seg_backward_rows_kernel): every op whose row gradient needs no walk over the sequence
  const bool rows_op = op == RUA_SUM || op == RUA_MEAN || op == RUA_LOGSUMEXP ||
                       ((op == RUA_MAX || op == RUA_MIN) && ties && ties_final);
  // — for the BATCH-MAJOR layouts: consecutive storage rows there belong to one sequence and share its out / grad /
  // ties rows (L1 hits); consecutive rows of a PackedSequence belong to 16 different sequences, each with its own
  // three rows to fetch — measured 3.4 TB/s for sum over P against 6.0 for the walk, which loads them once per sequence
  // (ops that read x: rows up to 1 KiB — one wave instruction per row; wider rows leave a wave one row of a 4-row
  // tile and the walk's 4.4 TB/s beats 3.5)
  // ([r5] every width: at rows wider than 1 KiB the waves of a workgroup take the column chunks of the same few rows)
  const bool rows_width_ok = true;
  const int tie_rule = extra_count & BWD_TIES_POSITIVE;
  if (vec_ok && !perm && !self_in && !(extra_count & BWD_SELF_COUNTS) && rows_op && rows_width_ok && L.kind != RUA_PACK &&
      H * (int64_t)sizeof(T) >= 64) {
    const int64_t row_bytes = H * (int64_t)sizeof(T);
    int tile_rows = BROWS_MAX;
    for (int64_t tb = BROWS_MAX * row_bytes; tile_rows > 4 && tb > (16 << 10); tb >>= 1) tile_rows >>= 1;
    const int64_t ntiles = (L.n_rows + tile_rows - 1) / tile_rows;
    const bool span = ntiles >= 2048;
    const int64_t per_xcd = span ? (ntiles + 7) / 8 : 0;
    const int64_t grid = span ? per_xcd * 8 : ntiles;
    if (grid > 0x7fffffffLL) return RUA_ERANGE;
    const bool nt = (double)L.n_rows * (double)row_bytes >= (double)(512ll << 20);
    const dim3 gg((unsigned)grid), bb(RUA_BLOCK);
    using A = typename elem<T>::acc;
#define RUA_BROWS(OPV, NTV)                                                                                         \
  if (n_chunks == 1)                                                                                                \
    hipLaunchKernelGGL((seg_backward_rows_kernel<T, FULL, OPV, NTV, false>), gg, bb, 0, s, L, (const T*)data, (const T*)out,  \
                       (const T*)gout, (T*)gin, H, lp_log2, 1, tile_rows, per_xcd, tie_rule, (const A*)ties);       \
  else                                                                                                              \
    hipLaunchKernelGGL((seg_backward_rows_kernel<T, FULL, OPV, NTV, true>), gg, bb, 0, s, L, (const T*)data, (const T*)out,  \
                       (const T*)gout, (T*)gin, H, lp_log2, (int)n_chunks, tile_rows, per_xcd, tie_rule, (const A*)ties)
#define RUA_BROWS_OP(NTV)                                 \
  switch (op) {                                           \
    case RUA_SUM: RUA_BROWS(RUA_SUM, NTV); break;         \
    case RUA_MEAN: RUA_BROWS(RUA_MEAN, NTV); break;       \
    case RUA_MAX: RUA_BROWS(RUA_MAX, NTV); break;         \
    case RUA_MIN: RUA_BROWS(RUA_MIN, NTV); break;         \
    default: RUA_BROWS(RUA_LOGSUMEXP, NTV); break;        \
  }
    if (nt) { RUA_BROWS_OP(true) } else { RUA_BROWS_OP(false) }
#undef RUA_BROWS_OP
#undef RUA_BROWS
    return (int)hipGetLastError();
  }
  // the walk-per-sequence kernels write token rows only: zero the padding rows of a padded layout first when asked to
  if (fill_padding && (L.kind == RUA_LEFT || L.kind == RUA_RIGHT)) {
    const hipError_t e = hipMemsetAsync(gin, 0, (size_t)L.n_rows * (size_t)H * sizeof(T), s);
    if (e != hipSuccess) return (int)e;
  }
  if (L.kind == RUA_PACK && L.sorted && !perm && (!ties || ties_final) && lp_log2 < 6 && !(split > 0 && ws) &&
      !(extra_count & BWD_SELF_COUNTS) && !self_in && (L.B >> (6 - lp_log2)) >= RANKS_MIN_WAVES) {
    // narrow rows of a PackedSequence: adjacent ranks share a wave instruction
    const int64_t rpw = RUA_WAVE >> lp_log2;
    const int64_t nblk = (L.B + rpw - 1) / rpw;
    if (nblk > 0x7fffffffLL) return RUA_ERANGE;
    const dim3 gg((unsigned)nblk), bb(RUA_WAVE);
#define RUA_BRANKS(EPLV, OPV, TV)                                                                                   \
  hipLaunchKernelGGL((seg_backward_ranks_kernel<T, EPLV, OPV, TV>), gg, bb, 0, s, L, (const T*)data,              \
                     (const T*)out, (const T*)gout, (T*)gin, H, lp_log2, tie_rule, (typename elem<T>::acc*)ties)
#define RUA_BRANKS_OP(EPLV)                                     \
  switch (op) {                                                 \
    case RUA_SUM: RUA_BRANKS(EPLV, RUA_SUM, 0); break;          \
    case RUA_MEAN: RUA_BRANKS(EPLV, RUA_MEAN, 0); break;        \
    case RUA_MAX: if (ties) RUA_BRANKS(EPLV, RUA_MAX, 2); else RUA_BRANKS(EPLV, RUA_MAX, 0); break; \
    case RUA_MIN: if (ties) RUA_BRANKS(EPLV, RUA_MIN, 2); else RUA_BRANKS(EPLV, RUA_MIN, 0); break; \
    case RUA_PROD: RUA_BRANKS(EPLV, RUA_PROD, 0); break;        \
    case RUA_LOGSUMEXP: RUA_BRANKS(EPLV, RUA_LOGSUMEXP, 0); break; \
    default: return RUA_EINVAL;                                 \
  }
    if (vec_ok) { RUA_BRANKS_OP(FULL) } else if (half_ok) { RUA_BRANKS_OP(HALF) } else { RUA_BRANKS_OP(1) }
#undef RUA_BRANKS_OP
#undef RUA_BRANKS
    return (int)hipGetLastError();
  }
  // whole sequences, no indirection, an op whose row gradient needs no counting walk: the lean walk (one wave per
  // (sequence, column chunk) — a PackedSequence of wide rows, and x-reading ops over rows wider than 1 KiB)
  if (!perm && !self_in && !(extra_count & BWD_SELF_COUNTS) && rows_op && !(split > 0 && ws)) {
    const bool nt = (double)L.n_rows * (double)H * (double)sizeof(T) >= (double)(512ll << 20);
    const dim3 gg((unsigned)blocks), bb(RUA_WAVE);
    using A = typename elem<T>::acc;
#define RUA_BWALK(EPLV, OPV, NTV)                                                                                   \
  hipLaunchKernelGGL((seg_backward_walk_kernel<T, EPLV, OPV, NTV>), gg, bb, 0, s, L, (const T*)data, (const T*)out,  \
                     (const T*)gout, (T*)gin, H, lp_log2, n_chunks, tie_rule, (const A*)ties)
#define RUA_BWALK_OP(EPLV, NTV)                                 \
  switch (op) {                                                 \
    case RUA_SUM: RUA_BWALK(EPLV, RUA_SUM, NTV); break;         \
    case RUA_MEAN: RUA_BWALK(EPLV, RUA_MEAN, NTV); break;       \
    case RUA_MAX: RUA_BWALK(EPLV, RUA_MAX, NTV); break;         \
    case RUA_MIN: RUA_BWALK(EPLV, RUA_MIN, NTV); break;         \
    default: RUA_BWALK(EPLV, RUA_LOGSUMEXP, NTV); break;        \
  }
#define RUA_BWALK_NT(EPLV) if (nt) { RUA_BWALK_OP(EPLV, true) } else { RUA_BWALK_OP(EPLV, false) }
    if (vec_ok) { RUA_BWALK_NT(FULL) } else if (half_ok) { RUA_BWALK_NT(HALF) } else { RUA_BWALK_NT(1) }
#undef RUA_BWALK_NT
#undef RUA_BWALK_OP
#undef RUA_BWALK
    return (int)hipGetLastError();
  }
  if (vec_ok)
    return launch_backward<T, FULL>(op, (unsigned)blocks, s, L, perm, data, out, gout, gin, H, lp_log2, n_chunks,
                                    extra_count, split, ws, ties, ties_final, self_in);
  if (half_ok)
    return launch_backward<T, HALF>(op, (unsigned)blocks, s, L, perm, data, out, gout, gin, H, lp_log2, n_chunks,
                                    extra_count, split, ws, ties, ties_final, self_in);
  return launch_backward<T, 1>(op, (unsigned)blocks, s, L, perm, data, out, gout, gin, H, lp_log2, n_chunks, extra_count,
                               split, ws, ties, ties_final, self_in);
}

// ---- per-dtype entry points: each element type is compiled in its own translation unit
// (rua_reduce_<dtype>.hip) so the ~300 kernel instantiations build in parallel
#define RUA_DECLARE_REDUCE_DTYPE(NAME)                                                                              \
  int reduce_##NAME(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,  \
                    int64_t H, int include_self, uint64_t empty_bits, void* extreme, int64_t split, void* ws,     \
                    const rua_layout* CD, void* copy, void* ties, int hints);                                  \
  int backward_##NAME(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data,           \
                      const void* out, const void* gout, void* gin, int64_t H, int extra_count, int64_t split,    \
                      void* ws, void* ties, bool ties_final, const void* self_in, bool fill_padding);              \
  int fill_empty_##NAME(hipStream_t s, const rua_layout& L, void* out, int64_t H, int want_max, void* ext,        \
                        int reset);                                                                              \
  int self_grad_##NAME(hipStream_t s, const int64_t* counts, int64_t S, int64_t H, const void* self_in,            \
                       const void* out, const void* gout, const void* aux, void* gself, int op, int inc);
RUA_DECLARE_REDUCE_DTYPE(f32)
RUA_DECLARE_REDUCE_DTYPE(bf16)
RUA_DECLARE_REDUCE_DTYPE(f16)
RUA_DECLARE_REDUCE_DTYPE(f64)

#define RUA_DEFINE_REDUCE_DTYPE(NAME, T)                                                                            \
  namespace rua {                                                                                                   \
  int reduce_##NAME(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,  \
                    int64_t H, int include_self, uint64_t empty_bits, void* extreme, int64_t split, void* ws,     \
                    const rua_layout* CD, void* copy, void* ties, int hints) {                                 \
    return dispatch_reduce<T>(op, s, L, perm, data, out, H, include_self, empty_bits, extreme, split, ws, CD,      \
                              copy, ties, hints);                                                                  \
  }                                                                                                                 \
  int backward_##NAME(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data,           \
                      const void* out, const void* gout, void* gin, int64_t H, int extra_count, int64_t split,    \
                      void* ws, void* ties, bool ties_final, const void* self_in, bool fill_padding) {             \
    return dispatch_backward<T>(op, s, L, perm, data, out, gout, gin, H, extra_count, split, ws, ties, ties_final, \
                                self_in, fill_padding);                                                            \
  }                                                                                                                 \
  int fill_empty_##NAME(hipStream_t s, const rua_layout& L, void* out, int64_t H, int want_max, void* ext,        \
                        int reset) {                                                                             \
    /* (the body strides over the batch: a capped grid; in the common case every workgroup reads one flag word) */ \
    hipLaunchKernelGGL(fill_empty_kernel<T>, dim3(grid_for(L.B) < 2048u ? (grid_for(L.B) ? grid_for(L.B) : 1u) : 2048u), dim3(RUA_BLOCK), 0, s, L, (T*)out, H, want_max,  \
                       (unsigned long long*)ext, reset);                                                           \
    return (int)hipGetLastError();                                                                                  \
  }                                                                                                                 \
  int self_grad_##NAME(hipStream_t s, const int64_t* counts, int64_t S, int64_t H, const void* self_in,            \
                       const void* out, const void* gout, const void* aux, void* gself, int op, int inc) {         \
    if (S * H > 0x7fffffffLL * RUA_BLOCK) return RUA_ERANGE;                                                        \
    hipLaunchKernelGGL(scatter_self_grad_kernel<T>, dim3(grid_for(S * H)), dim3(RUA_BLOCK), 0, s, counts, S, H,    \
                       (const T*)self_in, (const T*)out, (const T*)gout, aux, (T*)gself, op, inc);                 \
    return (int)hipGetLastError();                                                                                  \
  }                                                                                                                 \
  }

}  // namespace rua
