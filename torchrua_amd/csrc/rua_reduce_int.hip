// rua_reduce_int.hip — scatter_* on INTEGER tensors (reference reduce.py:6-23: the reference hands any dtype to
// torch.index_reduce / torch.index_add; counting tokens per bucket with scatter_sum on a long tensor is the ordinary
// use).  Bit-exact integer work: sums and products wrap in the element type (two's complement, as ATen's do), max / min
// compare in it, and `mean` is ATen's div_(counts, "floor") — with `counts` a tensor OF THE RESULT'S DTYPE, so the
// count itself wraps in int8 / uint8 / int16 (256 entries of an int8 bucket count as 0, which ATen then replaces by 1).
//
// Same shape as the float reducer (rua_reduce_impl.h): one wave owns one (bucket, 64-lane column chunk), 16-byte lanes
// over the hidden dimension where the rows allow it (one element per lane otherwise), rows narrower than a wave
// instruction share it (64 >> lp_log2 rows at a time, merged by a butterfly), 4 row groups in flight.  Integer
// folds are associative and commutative, so no order needs keeping and no partial needs a wider type.  HBM-bound
// byte work, no MFMA.  Layout: CAT, with or without the bucket indirection `perm` (rua_index_buckets).
//
// Long buckets (a skewed histogram: the one token every sentence holds) would leave one wave streaming millions of rows.
// With a workspace (`split` rows per part > 0) the buckets longer than `split` are left out of the main launch and
// taken by POSITION instead: the bucket-ordered positions [0, P) are cut into ranges of `split`; a range overlaps at
// most two long buckets — the one that holds its first position and the one that holds its last — and folds its share
// of each into an int64 partial (int_part_kernel); the range in which a long bucket STARTS then folds that bucket's
// partials, one per range it reaches into, and finishes it like the main launch does (int_combine_kernel).  Integer
// folds are exact in any order, so the result does not depend on the cut.  Two small launches when nothing is long.
#include <limits.h>
#include "rua_dev.h"

namespace rua {

constexpr int INT_UNROLL = 4;

template <typename T> struct int_lim;
template <> struct int_lim<int64_t> { static constexpr int64_t lo = INT64_MIN, hi = INT64_MAX; };
template <> struct int_lim<int32_t> { static constexpr int64_t lo = INT32_MIN, hi = INT32_MAX; };
template <> struct int_lim<int16_t> { static constexpr int64_t lo = INT16_MIN, hi = INT16_MAX; };
template <> struct int_lim<int8_t>  { static constexpr int64_t lo = INT8_MIN,  hi = INT8_MAX; };
template <> struct int_lim<uint8_t> { static constexpr int64_t lo = 0,         hi = UINT8_MAX; };

template <typename T, int OP> __device__ __forceinline__ T int_identity() {
  if (OP == RUA_MAX) return (T)int_lim<T>::lo;
  if (OP == RUA_MIN) return (T)int_lim<T>::hi;
  if (OP == RUA_PROD) return (T)1;
  return (T)0;
}

// the low bits of a 64-bit sum / product are the element type's own wrapped result
template <typename T, int OP> __device__ __forceinline__ T int_fold(T a, T b) {
  if (OP == RUA_MAX) return a > b ? a : b;
  if (OP == RUA_MIN) return a < b ? a : b;
  if (OP == RUA_PROD) return (T)((uint64_t)(int64_t)a * (uint64_t)(int64_t)b);
  return (T)((uint64_t)(int64_t)a + (uint64_t)(int64_t)b);
}

template <typename T> __device__ __forceinline__ T int_shfl_xor(T v, int d) {
  return (T)__shfl_xor((long long)v, d, RUA_WAVE);
}

template <typename T, int EPL> struct int_vec { T v[EPL]; } __attribute__((aligned(sizeof(T) * EPL)));

// lane geometry of one (bucket or range, 64-lane column chunk) unit
struct IntLane {
  int rpw, rsub;
  int64_t col;        // in EPL-element columns
  bool colok;
};
__device__ __forceinline__ IntLane int_lane(int64_t chunk, int64_t lpr, int lp_log2) {
  IntLane G;
  const int lane = threadIdx.x;
  G.rpw = RUA_WAVE >> lp_log2;
  G.rsub = lane >> lp_log2;
  G.col = chunk * RUA_WAVE + (lane & ((1 << lp_log2) - 1));
  G.colok = G.col < lpr;
  return G;
}

// fold the rows at bucket-ordered positions [p0, p1) into acc, then let the row groups of the wave meet
template <typename T, int EPL, int OP>
__device__ __forceinline__ void int_fold_positions(const IntLane& G, const int64_t* __restrict__ perm,
                                                   const T* __restrict__ data, int64_t n_rows, int64_t H, int lp_log2,
                                                   int64_t p0, int64_t p1, T (&acc)[EPL]) {
  using V = int_vec<T, EPL>;
  for (int64_t t0 = p0 + G.rsub; t0 < p1; t0 += (int64_t)G.rpw * INT_UNROLL) {   // (wave-divergent trip counts are fine:
    V val[INT_UNROLL];                                                          //  no cross-lane traffic inside the loop)
    bool ok[INT_UNROLL];
#pragma unroll
    for (int u = 0; u < INT_UNROLL; ++u) {
      const int64_t t = t0 + (int64_t)u * G.rpw;
      ok[u] = G.colok && t < p1;
      int64_t row = -1;
      if (ok[u]) row = perm ? perm[t] : t;
      ok[u] = ok[u] && row >= 0 && row < n_rows;            // lengths / buckets that overrun the payload read as nothing
      if (ok[u]) val[u] = *reinterpret_cast<const V*>(data + row * H + G.col * EPL);
    }
#pragma unroll
    for (int u = 0; u < INT_UNROLL; ++u)
      if (ok[u]) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = int_fold<T, OP>(acc[e], val[u].v[e]);
      }
  }
  for (int d = RUA_WAVE / 2; d >= (1 << lp_log2); d >>= 1) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = int_fold<T, OP>(acc[e], int_shfl_xor(acc[e], d));
  }
}

// the folded bucket meets its old row (include_self == 1), `mean` divides, the row is stored
template <typename T, int EPL, int OP>
__device__ __forceinline__ void int_finish(const IntLane& G, const T (&acc)[EPL], int64_t b, int64_t len,
                                           T* __restrict__ out, int64_t H, int include_self) {
  using V = int_vec<T, EPL>;
  if (G.rsub != 0 || !G.colok) return;
  if (include_self == 2 && len == 0) return;                 // torch.index_reduce: rows no index names keep their value
  T* o = out + b * H + G.col * EPL;
  V old;
  if (include_self == 1) old = *reinterpret_cast<const V*>(o);
  V res;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    T r = acc[e];
    if (include_self == 1) r = int_fold<T, OP>(r, old.v[e]);
    if (OP == RUA_MEAN) {
      // counts = (include_self ? ones : zeros).index_add_(ones) in the RESULT'S dtype; zeros become one; floor division
      T c = (T)(uint64_t)(len + (include_self == 1 ? 1 : 0));
      if (c == 0) c = 1;
      const int64_t a = (int64_t)r, cc = (int64_t)c;
      int64_t q = a / cc;
      if (((a < 0) != (cc < 0)) && a % cc != 0) q -= 1;
      r = (T)q;
    }
    res.v[e] = r;
  }
  *reinterpret_cast<V*>(o) = res;
}

template <typename T, int EPL, int OP>
__global__ __launch_bounds__(RUA_WAVE) void int_reduce_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                              const T* __restrict__ data, T* __restrict__ out,
                                                              int64_t H, int64_t lpr, int lp_log2, int64_t n_chunks,
                                                              int include_self, int64_t split) {
  const int64_t unit = blockIdx.x;
  const int64_t b = unit / n_chunks, chunk = unit - b * n_chunks;
  const IntLane G = int_lane(chunk, lpr, lp_log2);
  const int64_t len = seq_len(L, b);
  if (split > 0 && len > split) return;                      // a long bucket: int_part_kernel / int_combine_kernel
  const int64_t base = cat_off(L, b);
  T acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = int_identity<T, OP>();
  int_fold_positions<T, EPL, OP>(G, perm, data, L.n_rows, H, lp_log2, base, base + len, acc);
  int_finish<T, EPL, OP>(G, acc, b, len, out, H, include_self);
}

// ---- long buckets, by position.  Range u = positions [u * split, (u + 1) * split) of [0, P), P = the sum of the lengths.
struct IntRange {
  int64_t p0, p1;            // the range, clipped to P (p0 >= p1: nothing there)
  int64_t bA, offA, lenA;    // the bucket that holds p0
  int64_t bB, offB, lenB;    // the bucket that holds p1 - 1
};
__device__ __forceinline__ IntRange int_range(const rua_layout& L, int64_t u, int64_t split) {
  IntRange R;
  const int64_t P = L.B > 0 ? cat_off(L, L.B - 1) + seq_len(L, L.B - 1) : 0;
  R.p0 = u * split;
  R.p1 = R.p0 + split < P ? R.p0 + split : P;
  R.bA = R.bB = 0; R.offA = R.offB = 0; R.lenA = R.lenB = 0;
  if (R.p0 >= R.p1) return R;
  R.bA = search_cat(L, R.p0);            // (the LAST bucket that starts at or before the position: never an empty one)
  R.offA = cat_off(L, R.bA);
  R.lenA = seq_len(L, R.bA);
  R.bB = search_cat(L, R.p1 - 1);
  R.offB = cat_off(L, R.bB);
  R.lenB = seq_len(L, R.bB);
  return R;
}

// partial[(u * 2 + slot) * cols + column]: slot 0 = the range's share of bucket A, slot 1 = of bucket B (when B != A)
template <typename T, int EPL, int OP>
__global__ __launch_bounds__(RUA_WAVE) void int_part_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                            const T* __restrict__ data, int64_t H, int64_t lpr,
                                                            int lp_log2, int64_t n_chunks, int64_t split,
                                                            int64_t* __restrict__ partial, int64_t cols) {
  const int64_t unit = blockIdx.x;
  const int64_t u = unit / n_chunks, chunk = unit - u * n_chunks;
  const IntLane G = int_lane(chunk, lpr, lp_log2);
  const IntRange R = int_range(L, u, split);
  if (R.p0 >= R.p1) return;
#pragma unroll
  for (int slot = 0; slot < 2; ++slot) {
    const bool second = slot == 1;
    if (second && R.bB == R.bA) break;
    const int64_t len = second ? R.lenB : R.lenA;
    if (len <= split) continue;                              // the main launch took it
    const int64_t from = second ? R.offB : R.p0;
    const int64_t end = second ? R.p1 : (R.offA + R.lenA < R.p1 ? R.offA + R.lenA : R.p1);
    T acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = int_identity<T, OP>();
    int_fold_positions<T, EPL, OP>(G, perm, data, L.n_rows, H, lp_log2, from, end, acc);
    if (G.rsub == 0 && G.colok) {
      int64_t* __restrict__ p = partial + (u * 2 + slot) * cols + G.col * EPL;
#pragma unroll
      for (int e = 0; e < EPL; ++e) p[e] = (int64_t)acc[e];
    }
  }
}

template <typename T, int EPL, int OP>
__global__ __launch_bounds__(RUA_WAVE) void int_combine_kernel(rua_layout L, T* __restrict__ out, int64_t H, int64_t lpr,
                                                               int lp_log2, int64_t n_chunks, int include_self,
                                                               int64_t split, const int64_t* __restrict__ partial,
                                                               int64_t cols) {
  const int64_t unit = blockIdx.x;
  const int64_t u = unit / n_chunks, chunk = unit - u * n_chunks;
  const IntLane G = int_lane(chunk, lpr, lp_log2);
  const IntRange R = int_range(L, u, split);
  if (R.p0 >= R.p1) return;
  // the long bucket that STARTS in this range, if any: it reaches past the range's end, so it is bucket B
  if (R.lenB <= split || R.offB < R.p0) return;
  if (G.rsub != 0 || !G.colok) return;
  const int64_t u1 = (R.offB + R.lenB - 1) / split;          // the last range it reaches into
  T acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = int_identity<T, OP>();
  for (int64_t v0 = u; v0 <= u1; v0 += INT_UNROLL) {          // INT_UNROLL partial rows in flight
    int64_t val[INT_UNROLL][EPL];
#pragma unroll
    for (int k = 0; k < INT_UNROLL; ++k) {
      const int64_t v = v0 + k;
      if (v <= u1) {
        const int slot = (v == u && R.bA != R.bB) ? 1 : 0;   // later ranges hold it as THEIR bucket A
        const int64_t* __restrict__ p = partial + (v * 2 + slot) * cols + G.col * EPL;
#pragma unroll
        for (int e = 0; e < EPL; ++e) val[k][e] = p[e];
      }
    }
#pragma unroll
    for (int k = 0; k < INT_UNROLL; ++k)
      if (v0 + k <= u1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = int_fold<T, OP>(acc[e], (T)val[k][e]);
      }
  }
  IntLane G0 = G;
  int_finish<T, EPL, OP>(G0, acc, R.bB, R.lenB, out, H, include_self);
}

template <typename T, int EPL>
static int launch_int(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,
                      int64_t H, int include_self, int64_t split, void* ws) {
  const int64_t lpr = H / EPL;
  int lp_log2 = 0;
  while ((1 << lp_log2) < lpr && lp_log2 < 6) ++lp_log2;
  const int64_t n_chunks = (lpr + RUA_WAVE - 1) / RUA_WAVE;
  const int64_t units = L.B * n_chunks;
  if (units > 0x7fffffffLL) return RUA_ERANGE;
  if (!ws || split <= 0 || L.n_rows <= split) split = 0;
  const int64_t n_ranges = split > 0 ? (L.n_rows + split - 1) / split : 0;      // positions <= rows of the payload
  if (n_ranges * n_chunks > 0x7fffffffLL) return RUA_ERANGE;
  const int64_t cols = n_chunks * RUA_WAVE * EPL;
  int64_t* partial = (int64_t*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
  const dim3 g((unsigned)units), g2((unsigned)(n_ranges * n_chunks)), b(RUA_WAVE);
#define RUA_INT(OPV)                                                                                              \
  hipLaunchKernelGGL((int_reduce_kernel<T, EPL, OPV>), g, b, 0, s, L, perm, (const T*)data, (T*)out, H, lpr,     \
                     lp_log2, n_chunks, include_self, split);                                                     \
  if (split > 0) {                                                                                                \
    hipLaunchKernelGGL((int_part_kernel<T, EPL, OPV>), g2, b, 0, s, L, perm, (const T*)data, H, lpr, lp_log2,    \
                       n_chunks, split, partial, cols);                                                           \
    hipLaunchKernelGGL((int_combine_kernel<T, EPL, OPV>), g2, b, 0, s, L, (T*)out, H, lpr, lp_log2, n_chunks,    \
                       include_self, split, (const int64_t*)partial, cols);                                       \
  }
  switch (op) {
    case RUA_SUM: RUA_INT(RUA_SUM); break;
    case RUA_MEAN: RUA_INT(RUA_MEAN); break;
    case RUA_MAX: RUA_INT(RUA_MAX); break;
    case RUA_MIN: RUA_INT(RUA_MIN); break;
    case RUA_PROD: RUA_INT(RUA_PROD); break;
    default: return RUA_EINVAL;                 // logsumexp of integers is a float computation (not offered)
  }
#undef RUA_INT
  return (int)hipGetLastError();
}

template <typename T>
static int dispatch_int(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,
                        int64_t H, int include_self, int64_t split, void* ws) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const bool vec = (H % VEC) == 0 && (((uintptr_t)data | (uintptr_t)out) & 15) == 0;
  return vec ? launch_int<T, VEC>(op, s, L, perm, data, out, H, include_self, split, ws)
             : launch_int<T, 1>(op, s, L, perm, data, out, H, include_self, split, ws);
}

// workspace of the long-bucket split: two int64 partial rows per range, as wide as the widest lane geometry
int64_t reduce_int_ws_bytes(int64_t n_rows, int64_t H, int64_t split) {
  if (split <= 0 || n_rows <= split || H <= 0) return 0;
  const int64_t n_ranges = (n_rows + split - 1) / split;
  return n_ranges * 2 * (H + RUA_WAVE * 16) * 8 + 256;
}

int reduce_int(int dtype, int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,
               int64_t H, int include_self, int64_t split, void* ws) {
  if (L.kind != RUA_CAT) return RUA_EINVAL;
  if (include_self < 0 || include_self > 2) return RUA_EINVAL;
  switch (dtype) {
    case RUA_I64: return dispatch_int<int64_t>(op, s, L, perm, data, out, H, include_self, split, ws);
    case RUA_I32: return dispatch_int<int32_t>(op, s, L, perm, data, out, H, include_self, split, ws);
    case RUA_I16: return dispatch_int<int16_t>(op, s, L, perm, data, out, H, include_self, split, ws);
    case RUA_I8:  return dispatch_int<int8_t>(op, s, L, perm, data, out, H, include_self, split, ws);
    case RUA_U8:  return dispatch_int<uint8_t>(op, s, L, perm, data, out, H, include_self, split, ws);
  }
  return RUA_EINVAL;
}

}  // namespace rua
