// rua_reduce.hip — extern "C" entry points of the reductions; the kernels live in rua_reduce_impl.h and are
// instantiated per element type in rua_reduce_{f32,bf16,f16,f64}.hip.
#include <hip/hip_runtime.h>
#include "rua.h"
#include "rua_dev.h"

namespace rua {
constexpr int EXTREME_WORDS_ENTRY = RUA_EXTREME_WORDS;     // 1 024 slots, flags, the reset ticket, one spare
constexpr int BWD_TIES_POSITIVE = 2;      // bit 1 of the kernels' extra_count (rua_reduce_impl.h)
__global__ void extreme_init_entry_kernel(unsigned long long* ext, int want_max_of_data) {
  (void)want_max_of_data;                       // the slots are zero-neutral for the maximum and the minimum alike
  for (int i = threadIdx.x; i < EXTREME_WORDS_ENTRY; i += blockDim.x) ext[i] = 0ull;   // slots, flags, ticket
}
#define RUA_DECL(NAME)                                                                                              \
  int reduce_##NAME(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,  \
                    int64_t H, int include_self, uint64_t empty_bits, void* extreme, int64_t split, void* ws,     \
                    const rua_layout* CD, void* copy, void* ties, int hints);                                      \
  int backward_##NAME(int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data,           \
                      const void* out, const void* gout, void* gin, int64_t H, int extra_count, int64_t split,    \
                      void* ws, void* ties, bool ties_final, const void* self_in, bool fill_padding);              \
  int fill_empty_##NAME(hipStream_t s, const rua_layout& L, void* out, int64_t H, int want_max, void* ext,        \
                        int reset);                                                                              \
  int self_grad_##NAME(hipStream_t s, const int64_t* counts, int64_t S, int64_t H, const void* self_in,            \
                       const void* out, const void* gout, const void* aux, void* gself, int op, int inc);
RUA_DECL(f32) RUA_DECL(bf16) RUA_DECL(f16) RUA_DECL(f64)
#undef RUA_DECL
// integer element types (rua_reduce_int.hip)
int reduce_int(int dtype, int op, hipStream_t s, const rua_layout& L, const int64_t* perm, const void* data, void* out,
               int64_t H, int include_self, int64_t split, void* ws);
int64_t reduce_int_ws_bytes(int64_t n_rows, int64_t H, int64_t split);
}  // namespace rua

using namespace rua;

extern "C" {

int rua_reduce_team_waves(int64_t n_rows, int64_t B, int64_t row_bytes) {
  // what dispatch_reduce_main decides for an aligned payload whose rows are a multiple of 16 bytes — or of 8 bytes,
  // beyond one vector (16-byte lanes with an overlapping last lane)
  if (row_bytes <= 0 || row_bytes % 8 != 0 || (row_bytes % 16 != 0 && row_bytes <= 16) || row_bytes > 16 * RUA_WAVE) return 1;
  const int64_t lpr = (row_bytes + 15) / 16;
  int lp_log2 = 0;
  while ((1 << lp_log2) < lpr && lp_log2 < 6) ++lp_log2;
  return reduce_team_waves(n_rows, B, lp_log2, B);      // one column chunk per row: units = sequences
}

int rua_segment_reduce_backward(const rua_layout* lay, const int64_t* perm, const void* data, const void* out,
                                const void* grad_out, void* grad_in, int64_t H, int32_t dtype, int32_t op,
                                int32_t include_self, int64_t split_rows, void* ws, void* ties,
                                const void* self_in, void* stream) {
  if (!lay || H < 0 || lay->B < 0) return RUA_EINVAL;
  if (lay->kind != RUA_CAT && lay->kind != RUA_PACK && lay->kind != RUA_LEFT && lay->kind != RUA_RIGHT)
    return RUA_EINVAL;
  if (lay->kind == RUA_CAT && lay->lens && !lay->off) return RUA_EINVAL;
  if (lay->kind == RUA_PACK && lay->T > 0 && !lay->boff) return RUA_EINVAL;
  if (perm && lay->kind != RUA_CAT) return RUA_EINVAL;
  if (self_in && !perm) return RUA_EINVAL;       // the old destination row only exists for scatter_*
  if (lay->B == 0 || H == 0 || lay->n_rows == 0) return 0;
  if (!data || !out || !grad_out || !grad_in) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool fill = (include_self & RUA_BWD_FILL_PADDING) != 0;
  const int tie_rule = (include_self & RUA_BWD_TIES_POSITIVE) ? BWD_TIES_POSITIVE : 0;   // rides in extra_count
  include_self &= 0xff;
  const bool final = include_self == RUA_TIES_FINAL && ties != nullptr;   // the forward counted them (ties_out)
  switch (dtype) {
    case RUA_F32: return backward_f32(op, s, *lay, perm, data, out, grad_out, grad_in, H, (include_self == 1 ? 1 : 0) | tie_rule, split_rows, ws, ties, final, self_in, fill);
    case RUA_BF16: return backward_bf16(op, s, *lay, perm, data, out, grad_out, grad_in, H, (include_self == 1 ? 1 : 0) | tie_rule, split_rows, ws, ties, final, self_in, fill);
    case RUA_F16: return backward_f16(op, s, *lay, perm, data, out, grad_out, grad_in, H, (include_self == 1 ? 1 : 0) | tie_rule, split_rows, ws, ties, final, self_in, fill);
    case RUA_F64: return backward_f64(op, s, *lay, perm, data, out, grad_out, grad_in, H, (include_self == 1 ? 1 : 0) | tie_rule, split_rows, ws, ties, final, self_in, fill);
  }
  return RUA_EINVAL;
}

int rua_scatter_self_grad(const int64_t* counts, int64_t S, int64_t H, const void* self_in, const void* out,
                          const void* grad_out, const void* aux, void* grad_self, int32_t dtype, int32_t op,
                          int32_t include_self, void* stream) {
  if (S < 0 || H < 0 || op < RUA_SUM || op > RUA_LOGSUMEXP) return RUA_EINVAL;
  if (S == 0 || H == 0) return 0;
  if (!counts || !grad_out || !grad_self) return RUA_EINVAL;
  const bool reads_self = include_self && (op == RUA_MAX || op == RUA_MIN || op == RUA_LOGSUMEXP);
  if (reads_self && (!self_in || !out)) return RUA_EINVAL;
  if (include_self && op == RUA_PROD && !aux) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int inc = include_self ? 1 : 0;
  switch (dtype) {
    case RUA_F32: return self_grad_f32(s, counts, S, H, self_in, out, grad_out, aux, grad_self, op, inc);
    case RUA_BF16: return self_grad_bf16(s, counts, S, H, self_in, out, grad_out, aux, grad_self, op, inc);
    case RUA_F16: return self_grad_f16(s, counts, S, H, self_in, out, grad_out, aux, grad_self, op, inc);
    case RUA_F64: return self_grad_f64(s, counts, S, H, self_in, out, grad_out, aux, grad_self, op, inc);
  }
  return RUA_EINVAL;
}

int64_t rua_reduce_ws_bytes(int64_t n_rows, int64_t H, int32_t dtype, int64_t split_rows) {
  if (split_rows <= 0 || n_rows <= 0 || H <= 0) return 0;
  if (dtype >= RUA_I64 && dtype <= RUA_U8) return reduce_int_ws_bytes(n_rows, H, split_rows);
  const int64_t acc = dtype == RUA_F64 ? 8 : 4;
  const int64_t max_extra = n_rows / split_rows;
  const int64_t chunks_scalar = (H + RUA_WAVE - 1) / RUA_WAVE;              // worst case: one element per lane
  const int64_t max_u = max_extra * chunks_scalar;
  const int64_t cols = chunks_scalar * RUA_WAVE * (dtype == RUA_F64 ? 2 : 8);  // >= n_chunks * 64 * EPL on any path
  return 32 + max_u * 64 + 2 * max_extra * 2 * cols * acc + 256;
}

int rua_segment_reduce(const rua_layout* lay, const int64_t* perm, const void* data, void* out, int64_t H,
                       int32_t dtype, int32_t op, int32_t include_self, uint64_t empty_bits, void* extreme,
                       int64_t split_rows, void* ws, void* ties, void* stream) {
  if (!lay || H < 0 || lay->B < 0) return RUA_EINVAL;
  if (lay->kind != RUA_CAT && lay->kind != RUA_PACK && lay->kind != RUA_LEFT && lay->kind != RUA_RIGHT)
    return RUA_EINVAL;
  if (lay->kind == RUA_CAT && lay->lens && !lay->off) return RUA_EINVAL;
  if (lay->kind == RUA_PACK && lay->T > 0 && !lay->boff) return RUA_EINVAL;
  if (perm && lay->kind != RUA_CAT) return RUA_EINVAL;
  if (lay->B == 0 || H == 0) return 0;
  if (!out || (lay->n_rows > 0 && !data)) return RUA_EINVAL;
  const bool clean = (op & RUA_OP_SCRATCH_CLEAN) != 0;
  const int hints = ((op & RUA_OP_NO_EMPTY) ? 1 : 0) | ((op & RUA_OP_SHORT_SEQS) ? 2 : 0);     // dispatch_reduce's hints
  op &= 0xff;
  if (ties && op != RUA_MAX && op != RUA_MIN) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype >= RUA_I64 && dtype <= RUA_U8) {        // integer tensors: scatter_* only (reduce.py:6-23)
    if (ties || lay->kind != RUA_CAT) return RUA_EINVAL;
    return reduce_int(dtype, op, s, *lay, perm, data, out, H, include_self, split_rows, ws);
  }
  if (extreme && !clean && (op == RUA_MAX || op == RUA_MIN || op == RUA_LOGSUMEXP)) {
    hipLaunchKernelGGL(extreme_init_entry_kernel, dim3(1), dim3(256), 0, s, (unsigned long long*)extreme,
                       op == RUA_MIN ? 1 : 0);
  }
  switch (dtype) {
    case RUA_F32: return reduce_f32(op, s, *lay, perm, data, out, H, include_self, empty_bits, extreme, split_rows, ws, nullptr, nullptr, ties, hints);
    case RUA_BF16: return reduce_bf16(op, s, *lay, perm, data, out, H, include_self, empty_bits, extreme, split_rows, ws, nullptr, nullptr, ties, hints);
    case RUA_F16: return reduce_f16(op, s, *lay, perm, data, out, H, include_self, empty_bits, extreme, split_rows, ws, nullptr, nullptr, ties, hints);
    case RUA_F64: return reduce_f64(op, s, *lay, perm, data, out, H, include_self, empty_bits, extreme, split_rows, ws, nullptr, nullptr, ties, hints);
  }
  return RUA_EINVAL;
}

int rua_pack_reduce(const rua_layout* src, const rua_layout* pack, const void* data, void* pack_data, void* out,
                    int64_t H, int32_t dtype, int32_t op, uint64_t empty_bits, void* extreme, int64_t split_rows,
                    void* ws, void* stream) {
  if (!src || !pack || H < 0 || src->B < 0) return RUA_EINVAL;
  if (src->kind != RUA_CAT && src->kind != RUA_LEFT && src->kind != RUA_RIGHT) return RUA_EINVAL;
  if (src->kind == RUA_CAT && src->lens && !src->off) return RUA_EINVAL;
  if (pack->kind != RUA_PACK || pack->B != src->B || (pack->T > 0 && !pack->boff)) return RUA_EINVAL;
  if (src->B == 0 || H == 0) return 0;
  if (!out || !pack_data || !data) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool clean = (op & RUA_OP_SCRATCH_CLEAN) != 0;
  const int hints = (op & RUA_OP_NO_EMPTY) ? 1 : 0;
  op &= 0xff;
  if (extreme && !clean && (op == RUA_MAX || op == RUA_MIN || op == RUA_LOGSUMEXP))
    hipLaunchKernelGGL(extreme_init_entry_kernel, dim3(1), dim3(256), 0, s, (unsigned long long*)extreme,
                       op == RUA_MIN ? 1 : 0);
  switch (dtype) {
    case RUA_F32: return reduce_f32(op, s, *src, nullptr, data, out, H, 0, empty_bits, extreme, split_rows, ws, pack, pack_data, nullptr, hints);
    case RUA_BF16: return reduce_bf16(op, s, *src, nullptr, data, out, H, 0, empty_bits, extreme, split_rows, ws, pack, pack_data, nullptr, hints);
    case RUA_F16: return reduce_f16(op, s, *src, nullptr, data, out, H, 0, empty_bits, extreme, split_rows, ws, pack, pack_data, nullptr, hints);
    case RUA_F64: return reduce_f64(op, s, *src, nullptr, data, out, H, 0, empty_bits, extreme, split_rows, ws, pack, pack_data, nullptr, hints);
  }
  return RUA_EINVAL;
}

int rua_fill_empty(const rua_layout* lay, void* out, int64_t H, int32_t dtype, int32_t op, void* extreme,
                   void* stream) {
  if (!lay || H < 0 || !extreme) return RUA_EINVAL;
  const int reset = (op & RUA_OP_SCRATCH_CLEAN) ? 1 : 0;
  op &= 0xff;
  if (op != RUA_MAX && op != RUA_MIN && op != RUA_LOGSUMEXP) return RUA_EINVAL;
  const int64_t n = lay->B * H;
  if (n == 0) return 0;
  if (!out) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int wmax = op == RUA_MIN ? 1 : 0;
  switch (dtype) {
    case RUA_F32: return fill_empty_f32(s, *lay, out, H, wmax, extreme, reset);
    case RUA_BF16: return fill_empty_bf16(s, *lay, out, H, wmax, extreme, reset);
    case RUA_F16: return fill_empty_f16(s, *lay, out, H, wmax, extreme, reset);
    case RUA_F64: return fill_empty_f64(s, *lay, out, H, wmax, extreme, reset);
    default: return RUA_EINVAL;
  }
}

}  // extern "C"
