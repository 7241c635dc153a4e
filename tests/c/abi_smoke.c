/* Plain-C consumer of librua_hip.so: no Python, no torch.  Packs a tiny CattedSequence into a
 * PackedSequence layout and sums every sequence, through the C ABI only (include/rua.h), and checks the
 * result on the host.  Built and run by tests/test_c_abi.py on the GPU box:
 *   gcc -std=c99 abi_smoke.c -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -L <libdir> -lrua_hip -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rua.h"

#define CHECK(x) do { int rc_ = (int)(x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (line %d)\n", #x, rc_, __LINE__); return 1; } } while (0)

int main(void) {
  enum { B = 4, H = 4, N = 10, T = 4 };
  const int64_t lens[B] = {2, 4, 1, 3};
  const int64_t sorted[B] = {1, 3, 0, 2};                 /* descending lengths (host order) */
  float data[N * H];
  for (int i = 0; i < N * H; ++i) data[i] = (float)(i / H) + 0.25f * (float)(i % H);

  int64_t *d_lens, *d_off, *d_sorted, *d_unsorted, *d_bsz, *d_boff, *d_ws;
  float *d_data, *d_pack, *d_out;
  CHECK(hipMalloc((void**)&d_lens, sizeof lens));
  CHECK(hipMalloc((void**)&d_off, sizeof lens));
  CHECK(hipMalloc((void**)&d_sorted, sizeof sorted));
  CHECK(hipMalloc((void**)&d_unsorted, sizeof sorted));
  CHECK(hipMalloc((void**)&d_bsz, T * sizeof(int64_t)));
  CHECK(hipMalloc((void**)&d_boff, T * sizeof(int64_t)));
  CHECK(hipMalloc((void**)&d_ws, (size_t)rua_scan_ws_elems(B) * sizeof(int64_t)));
  CHECK(hipMalloc((void**)&d_data, sizeof data));
  CHECK(hipMalloc((void**)&d_pack, sizeof data));
  CHECK(hipMalloc((void**)&d_out, B * H * sizeof(float)));
  CHECK(hipMemcpy(d_lens, lens, sizeof lens, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_sorted, sorted, sizeof sorted, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_data, data, sizeof data, hipMemcpyHostToDevice));

  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  CHECK(rua_abi_version() != RUA_ABI_VERSION);
  CHECK(rua_exclusive_scan_i64(d_lens, d_off, NULL, B, d_ws, s));
  CHECK(rua_pack_meta(d_lens, d_sorted, B, T, d_unsorted, d_bsz, s));
  CHECK(rua_exclusive_scan_i64(d_bsz, d_boff, NULL, T, d_ws, s));

  rua_layout cat, pack;
  memset(&cat, 0, sizeof cat);
  memset(&pack, 0, sizeof pack);
  cat.kind = RUA_CAT; cat.n_rows = N; cat.B = B; cat.lens = d_lens; cat.off = d_off;
  pack.kind = RUA_PACK; pack.n_rows = N; pack.B = B; pack.lens = d_lens; pack.boff = d_boff; pack.T = T;
  pack.sorted = d_sorted; pack.unsorted = d_unsorted;
  CHECK(rua_move_rows(&pack, &cat, RUA_T_SHIFT, 0, d_pack, d_data, H * sizeof(float), NULL, -1, 0, s));
  CHECK(rua_segment_reduce(&pack, NULL, d_pack, d_out, H, RUA_F32, RUA_SUM, 0, 0, NULL, 0, NULL, NULL, s));
  CHECK(hipStreamSynchronize(s));

  /* the one-call form of the three metadata steps above must agree with them */
  int64_t *d_uns2, *d_bsz2, *d_boff2, *d_off2;
  CHECK(hipMalloc((void**)&d_uns2, sizeof sorted));
  CHECK(hipMalloc((void**)&d_bsz2, T * sizeof(int64_t)));
  CHECK(hipMalloc((void**)&d_boff2, T * sizeof(int64_t)));
  CHECK(hipMalloc((void**)&d_off2, sizeof lens));
  CHECK(rua_pack_prepare(d_lens, d_sorted, B, T, d_uns2, d_bsz2, d_boff2, d_off2, d_ws, s));
  CHECK(hipStreamSynchronize(s));
  int64_t a[B > T ? B : T], c2[B > T ? B : T];
  int bad = 0;
#define SAME(x, y, n)                                                   \
  CHECK(hipMemcpy(a, x, (n) * sizeof(int64_t), hipMemcpyDeviceToHost)); \
  CHECK(hipMemcpy(c2, y, (n) * sizeof(int64_t), hipMemcpyDeviceToHost)); \
  for (int i_ = 0; i_ < (n); ++i_) bad += a[i_] != c2[i_];
  SAME(d_unsorted, d_uns2, B) SAME(d_bsz, d_bsz2, T) SAME(d_boff, d_boff2, T) SAME(d_off, d_off2, B)

  float packed[N * H], out[B * H];
  int64_t bsz[T], unsorted[B];
  CHECK(hipMemcpy(packed, d_pack, sizeof packed, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(bsz, d_bsz, sizeof bsz, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(unsorted, d_unsorted, sizeof unsorted, hipMemcpyDeviceToHost));

  /* expectations from SURVEY.md §8c(i): rows of C are 0..9; P order = rows [2,7,0,6,3,8,1,4,9,5] */
  const int p_rows[N] = {2, 7, 0, 6, 3, 8, 1, 4, 9, 5};
  const int64_t e_bsz[T] = {4, 3, 2, 1}, e_uns[B] = {2, 0, 3, 1};
  for (int j = 0; j < N; ++j)
    for (int h = 0; h < H; ++h) bad += packed[j * H + h] != data[p_rows[j] * H + h];
  for (int t = 0; t < T; ++t) bad += bsz[t] != e_bsz[t];
  for (int b = 0; b < B; ++b) bad += unsorted[b] != e_uns[b];
  int row = 0;
  for (int b = 0; b < B; ++b) {
    for (int h = 0; h < H; ++h) {
      float ref = 0.0f;
      for (int t = 0; t < lens[b]; ++t) ref += data[(row + t) * H + h];
      bad += out[b * H + h] != ref;
    }
    row += (int)lens[b];
  }
  printf("abi_smoke: target %s, %d mismatches\n", rua_build_target(), bad);
  return bad ? 2 : 0;
}
