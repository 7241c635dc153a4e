// rua_scatter.hip — bucket an arbitrary destination index so scatter_* (reference
// reduce.py:6-31: torch.index_reduce / index_add) becomes a segmented reduce with a row
// indirection (rua_segment_reduce(perm=...)): no float atomics on the data path, and the
// summation order is a fixed function of the inputs (ascending source row per destination),
// so results are bitwise reproducible (for destinations receiving up to 1 024 rows).  gfx950, wave64.
#include "rua_dev.h"

namespace rua {

__global__ __launch_bounds__(RUA_BLOCK) void bucket_count_kernel(const int64_t* __restrict__ index, int64_t M,
                                                                 int64_t S, unsigned long long* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (i >= M) return;
  const int64_t s = index[i];
  if (s >= 0 && s < S) atomicAdd(&counts[s], 1ull);
}

__global__ __launch_bounds__(RUA_BLOCK) void bucket_place_kernel(const int64_t* __restrict__ index, int64_t M,
                                                                 int64_t S, unsigned long long* __restrict__ cursor,
                                                                 int64_t* __restrict__ tmp) {
  const int64_t i = (int64_t)blockIdx.x * RUA_BLOCK + threadIdx.x;
  if (i >= M) return;
  const int64_t s = index[i];
  if (s >= 0 && s < S) tmp[atomicAdd(&cursor[s], 1ull)] = i;
}

// one wave per bucket: order the bucket's source rows ascending (they are distinct)
constexpr int SORT_LDS = 1024;  // per-wave LDS staging (int64)
__global__ __launch_bounds__(RUA_BLOCK) void bucket_sort_kernel(const int64_t* __restrict__ off,
                                                                const int64_t* __restrict__ counts, int64_t S,
                                                                const int64_t* __restrict__ tmp,
                                                                int64_t* __restrict__ perm) {
  __shared__ int64_t s_buf[RUA_WAVES_PER_BLOCK][SORT_LDS];
  const int lane = threadIdx.x & (RUA_WAVE - 1), wave = threadIdx.x >> 6;
  const int64_t s = (int64_t)blockIdx.x * RUA_WAVES_PER_BLOCK + wave;
  if (s >= S) return;
  const int64_t n = counts[s], base = off[s];
  if (n <= 0) return;
  if (n <= RUA_WAVE) {
    // bitonic network across the 64 lanes; absent lanes hold +inf and end up last
    int64_t v = lane < n ? tmp[base + lane] : INT64_MAX;
#pragma unroll
    for (int k = 2; k <= RUA_WAVE; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int64_t o = __shfl_xor(v, j, RUA_WAVE);
        const bool up = (lane & k) == 0;          // ascending block
        const bool lower = (lane & j) == 0;       // lower index of the pair
        const bool take_min = up == lower;
        v = take_min ? (v < o ? v : o) : (v > o ? v : o);
      }
    }
    if (lane < n) perm[base + lane] = v;
    return;
  }
  if (n > SORT_LDS) {
    // very large fan-in: ordering would cost O(n^2 / 64) here; keep the atomic arrival order.  The
    // reduction is still correct; only its bitwise reproducibility is given up for this destination
    // (torch.index_add / index_reduce on a GPU make no such promise at any size).
    for (int64_t i = lane; i < n; i += RUA_WAVE) perm[base + i] = tmp[base + i];
    return;
  }
  // rank by counting (values distinct): rank(i) = #{j : v[j] < v[i]}, staged in LDS
  for (int64_t i = lane; i < n; i += RUA_WAVE) s_buf[wave][i] = tmp[base + i];
  // waves of a block do not share s_buf rows, and a wave executes in lockstep: a wave-level
  // fence is enough to make the staged values visible to the other lanes of this wave
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int64_t i = lane; i < n; i += RUA_WAVE) {
    const int64_t vi = s_buf[wave][i];
    int64_t rank = 0;
    for (int64_t j = 0; j < n; ++j) rank += s_buf[wave][j] < vi;
    perm[base + rank] = vi;
  }
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + RUA_BLOCK - 1) / RUA_BLOCK); }

}  // namespace rua

using namespace rua;

extern "C" int rua_index_buckets(const int64_t* index, int64_t M, int64_t S, int64_t* counts, int64_t* off,
                                 int64_t* perm, int64_t* ws, void* stream) {
  if (M < 0 || S < 0) return RUA_EINVAL;
  if (S == 0) return 0;
  if (!counts || !off || !ws || (M > 0 && (!index || !perm))) return RUA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // ws = [scan scratch | cursor[S] | tmp[M]]
  int64_t* scan_ws = ws;
  int64_t* cursor = ws + rua_scan_ws_elems(S);
  int64_t* tmp = cursor + S;
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)S, s);
  if (e != hipSuccess) return (int)e;
  if (M > 0)
    hipLaunchKernelGGL(bucket_count_kernel, dim3(grid_for(M)), dim3(RUA_BLOCK), 0, s, index, M, S,
                       (unsigned long long*)counts);
  int r = rua_exclusive_scan_i64(counts, off, nullptr, S, scan_ws, stream);
  if (r != 0) return r;
  if (M == 0) return 0;
  e = hipMemcpyAsync(cursor, off, sizeof(int64_t) * (size_t)S, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(bucket_place_kernel, dim3(grid_for(M)), dim3(RUA_BLOCK), 0, s, index, M, S,
                     (unsigned long long*)cursor, tmp);
  const int64_t blocks = (S + RUA_WAVES_PER_BLOCK - 1) / RUA_WAVES_PER_BLOCK;
  if (blocks > 0x7fffffffLL) return RUA_ERANGE;
  hipLaunchKernelGGL(bucket_sort_kernel, dim3((unsigned)blocks), dim3(RUA_BLOCK), 0, s, off, counts, S, tmp, perm);
  return (int)hipGetLastError();
}
