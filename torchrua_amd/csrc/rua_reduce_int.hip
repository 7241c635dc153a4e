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

template <typename T, int EPL, int OP>
__global__ __launch_bounds__(RUA_WAVE) void int_reduce_kernel(rua_layout L, const int64_t* __restrict__ perm,
                                                              const T* __restrict__ data, T* __restrict__ out,
                                                              int64_t H, int64_t lpr, int lp_log2, int64_t n_chunks,
                                                              int include_self) {
  using V = int_vec<T, EPL>;
  const int64_t unit = blockIdx.x;
  const int64_t b = unit / n_chunks, chunk = unit - b * n_chunks;
  const int lane = threadIdx.x;
  const int rpw = RUA_WAVE >> lp_log2;
  const int rsub = lane >> lp_log2;
  const int64_t col = chunk * RUA_WAVE + (lane & ((1 << lp_log2) - 1));       // in EPL-element columns
  const bool colok = col < lpr;
  const int64_t len = seq_len(L, b);
  const int64_t base = cat_off(L, b);

  T acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = int_identity<T, OP>();

  for (int64_t t0 = rsub; t0 < len; t0 += (int64_t)rpw * INT_UNROLL) {      // (wave-divergent trip counts are fine: no
    V val[INT_UNROLL];                                                     //  cross-lane traffic inside the loop)
    bool ok[INT_UNROLL];
#pragma unroll
    for (int u = 0; u < INT_UNROLL; ++u) {
      const int64_t t = t0 + (int64_t)u * rpw;
      ok[u] = colok && t < len;
      int64_t row = -1;
      if (ok[u]) row = perm ? perm[base + t] : base + t;
      ok[u] = ok[u] && row >= 0 && row < L.n_rows;          // lengths / buckets that overrun the payload read as nothing
      if (ok[u]) val[u] = *reinterpret_cast<const V*>(data + row * H + col * EPL);
    }
#pragma unroll
    for (int u = 0; u < INT_UNROLL; ++u)
      if (ok[u]) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = int_fold<T, OP>(acc[e], val[u].v[e]);
      }
  }
  // the row groups of the wave meet
  for (int d = RUA_WAVE / 2; d >= (1 << lp_log2); d >>= 1) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = int_fold<T, OP>(acc[e], int_shfl_xor(acc[e], d));
  }
  if (rsub != 0 || !colok) return;
  if (include_self == 2 && len == 0) return;                 // torch.index_reduce: rows no index names keep their value
  T* o = out + b * H + col * EPL;
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

template <typename T, int EPL>
static int launch_int(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,
                      int64_t H, int include_self) {
  const int64_t lpr = H / EPL;
  int lp_log2 = 0;
  while ((1 << lp_log2) < lpr && lp_log2 < 6) ++lp_log2;
  const int64_t n_chunks = (lpr + RUA_WAVE - 1) / RUA_WAVE;
  const int64_t units = L.B * n_chunks;
  if (units > 0x7fffffffLL) return RUA_ERANGE;
  const dim3 g((unsigned)units), b(RUA_WAVE);
#define RUA_INT(OPV)                                                                                              \
  hipLaunchKernelGGL((int_reduce_kernel<T, EPL, OPV>), g, b, 0, s, L, perm, (const T*)data, (T*)out, H, lpr,     \
                     lp_log2, n_chunks, include_self)
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
                        int64_t H, int include_self) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const bool vec = (H % VEC) == 0 && (((uintptr_t)data | (uintptr_t)out) & 15) == 0;
  return vec ? launch_int<T, VEC>(op, s, L, perm, data, out, H, include_self)
             : launch_int<T, 1>(op, s, L, perm, data, out, H, include_self);
}

int reduce_int(int dtype, int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,
               int64_t H, int include_self) {
  if (L.kind != RUA_CAT) return RUA_EINVAL;
  if (include_self < 0 || include_self > 2) return RUA_EINVAL;
  switch (dtype) {
    case RUA_I64: return dispatch_int<int64_t>(op, s, L, perm, data, out, H, include_self);
    case RUA_I32: return dispatch_int<int32_t>(op, s, L, perm, data, out, H, include_self);
    case RUA_I16: return dispatch_int<int16_t>(op, s, L, perm, data, out, H, include_self);
    case RUA_I8:  return dispatch_int<int8_t>(op, s, L, perm, data, out, H, include_self);
    case RUA_U8:  return dispatch_int<uint8_t>(op, s, L, perm, data, out, H, include_self);
  }
  return RUA_EINVAL;
}

}  // namespace rua
