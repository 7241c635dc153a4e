/* Plain-C consumer of librua_hip.so, second program: the whole pack -> reduce pipeline on a ragged batch of a few
 * thousand sequences (some of them empty), host entry points included — the reference's host sort
 * (rua_host_sort_desc) and batch_sizes (rua_host_batch_sizes), then rua_pack_prepare, rua_move_rows,
 * rua_segment_reduce (sum; max with the reference's global `initial` through rua_fill_empty), the scatter_sum form
 * (rua_index_buckets + the row indirection), its integer twin (ABI 4: scatter_mean on int8), the host sort on the
 * helper thread, and the fused backward of max.  Everything is checked on the host with
 * loops written from the formulas in include/rua.h.  No Python, no torch in the process.
 * Built and run by tests/test_c_abi.py on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rua.h"

#define CHECK(x) do { int rc_ = (int)(x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (line %d)\n", #x, rc_, __LINE__); return 1; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static void* dmalloc(size_t n) { void* p = NULL; if (hipMalloc(&p, n ? n : 8) != hipSuccess) { fprintf(stderr, "hipMalloc(%zu)\n", n); exit(3); } return p; }
static void* upload(const void* h, size_t n) { void* d = dmalloc(n); if (hipMemcpy(d, h, n, hipMemcpyHostToDevice) != hipSuccess) exit(3); return d; }

int main(void) {
  enum { B = 3000, H = 24, TMAX = 70 };
  int64_t* lens = malloc(B * sizeof *lens);
  int64_t N = 0, T = 0;
  for (int b = 0; b < B; ++b) {
    lens[b] = (b % 17 == 3) ? 0 : (int64_t)(rng() % TMAX) + 1;      /* some empty sequences */
    N += lens[b];
    if (lens[b] > T) T = lens[b];
  }
  float* data = malloc((size_t)N * H * sizeof *data);
  for (int64_t i = 0; i < N * H; ++i) data[i] = (float)((int)(rng() % 2001) - 1000) / 64.0f;   /* exact in fp32 */

  /* ---- host side of pack(): the reference's sort order and batch_sizes */
  int64_t* sorted = malloc(B * sizeof *sorted);
  int64_t* bsz = malloc((size_t)T * sizeof *bsz);
  CHECK(rua_host_sort_desc(lens, B, sorted, 2));
  CHECK(rua_host_batch_sizes(lens, B, T, bsz));
  int bad = 0;
  char* seen = calloc(B, 1);
  for (int r = 0; r < B; ++r) {                       /* a permutation, lengths non-increasing */
    bad += sorted[r] < 0 || sorted[r] >= B || seen[sorted[r]]++;
    if (r) bad += lens[sorted[r - 1]] < lens[sorted[r]];
  }
  for (int64_t t = 0; t < T; ++t) {
    int64_t c = 0;
    for (int b = 0; b < B; ++b) c += lens[b] > t;
    bad += bsz[t] != c;
  }

  /* ---- device metadata */
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  int64_t* d_lens = upload(lens, B * sizeof *lens);
  int64_t* d_sorted = upload(sorted, B * sizeof *sorted);
  int64_t* d_uns = dmalloc(B * 8), *d_bsz = dmalloc((size_t)T * 8), *d_boff = dmalloc((size_t)T * 8), *d_off = dmalloc(B * 8);
  int64_t* d_ws = dmalloc((size_t)rua_scan_ws_elems(B > T ? B : T) * 8);
  float* d_data = upload(data, (size_t)N * H * sizeof *data);
  float* d_pack = dmalloc((size_t)N * H * 4), *d_sum = dmalloc((size_t)B * H * 4), *d_max = dmalloc((size_t)B * H * 4);
  float* d_ties = dmalloc((size_t)B * H * 4), *d_gin = dmalloc((size_t)N * H * 4), *d_gout = dmalloc((size_t)B * H * 4);
  uint64_t* d_ext = dmalloc(RUA_EXTREME_WORDS * 8);
  CHECK(rua_pack_prepare(d_lens, d_sorted, B, T, d_uns, d_bsz, d_boff, d_off, d_ws, s));

  rua_layout cat, pack;
  memset(&cat, 0, sizeof cat);
  memset(&pack, 0, sizeof pack);
  cat.kind = RUA_CAT; cat.n_rows = N; cat.B = B; cat.lens = d_lens; cat.off = d_off;
  pack.kind = RUA_PACK; pack.n_rows = N; pack.B = B; pack.lens = d_lens; pack.boff = d_boff; pack.T = T;
  pack.sorted = d_sorted; pack.unsorted = d_uns; pack.bsz = d_bsz;

  /* ---- pack, reduce over the PackedSequence, max over the CattedSequence with the reference's `initial` */
  CHECK(rua_move_rows(&pack, &cat, RUA_T_SHIFT, 0, d_pack, d_data, H * sizeof(float), NULL, -1, 0, s));
  CHECK(rua_segment_reduce(&pack, NULL, d_pack, d_sum, H, RUA_F32, RUA_SUM, 0, 0, NULL, 0, NULL, NULL, s));
  CHECK(rua_segment_reduce(&cat, NULL, d_data, d_max, H, RUA_F32, RUA_MAX, 0, 0, d_ext, 0, NULL, d_ties, s));
  CHECK(rua_fill_empty(&cat, d_max, H, RUA_F32, RUA_MAX, d_ext, s));

  /* ---- backward of max: cotangent b + 1 for sequence b, ties from the forward */
  float* gout = malloc((size_t)B * H * sizeof *gout);
  for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) gout[b * H + h] = (float)(b % 7) - 3.0f;
  CHECK(hipMemcpyAsync(d_gout, gout, (size_t)B * H * 4, hipMemcpyHostToDevice, s));
  CHECK(rua_segment_reduce_backward(&cat, NULL, d_data, d_max, d_gout, d_gin, H, RUA_F32, RUA_MAX,
                                    RUA_TIES_FINAL | RUA_BWD_TIES_POSITIVE, 0, NULL, d_ties, NULL, s));

  /* ---- scatter_sum form: rows shuffled, destination = the sequence of the row */
  int64_t* index = malloc((size_t)N * sizeof *index);
  int64_t* shuffle = malloc((size_t)N * sizeof *shuffle);
  {
    int64_t row = 0;
    for (int b = 0; b < B; ++b) for (int64_t t = 0; t < lens[b]; ++t) index[row++] = b;
    for (int64_t i = 0; i < N; ++i) shuffle[i] = i;
    for (int64_t i = N - 1; i > 0; --i) { int64_t j = (int64_t)(rng() % (uint64_t)(i + 1)), x = shuffle[i]; shuffle[i] = shuffle[j]; shuffle[j] = x; }
  }
  int64_t* sh_index = malloc((size_t)N * sizeof *sh_index);
  float* sh_data = malloc((size_t)N * H * sizeof *sh_data);
  for (int64_t i = 0; i < N; ++i) { sh_index[i] = index[shuffle[i]]; memcpy(sh_data + i * H, data + shuffle[i] * H, H * sizeof(float)); }
  int64_t* d_index = upload(sh_index, (size_t)N * 8);
  float* d_sh = upload(sh_data, (size_t)N * H * 4);
  int64_t* d_counts = dmalloc(B * 8), *d_boffs = dmalloc(B * 8), *d_perm = dmalloc((size_t)N * 8);
  int64_t* d_bws = dmalloc((size_t)rua_bucket_ws_elems(N, B) * 8);
  float* d_scat = dmalloc((size_t)B * H * 4);
  CHECK(rua_index_buckets(d_index, N, B, d_counts, d_boffs, d_perm, d_bws, s));
  rua_layout buckets;
  memset(&buckets, 0, sizeof buckets);
  buckets.kind = RUA_CAT; buckets.n_rows = N; buckets.B = B; buckets.lens = d_counts; buckets.off = d_boffs;
  CHECK(rua_segment_reduce(&buckets, d_perm, d_sh, d_scat, H, RUA_F32, RUA_SUM, 0, 0, NULL, 0, NULL, NULL, s));

  /* ---- ABI 4: the same buckets over an INTEGER payload (scatter_mean on int8 with include_self, reduce.py:18-19):
   * sums wrap in int8 and ATen divides by a count OF THAT TYPE, rounding towards minus infinity */
  enum { HI = 16 };
  int8_t* i_src = malloc((size_t)N * HI), *i_ten = malloc((size_t)B * HI);
  for (int64_t i = 0; i < N * HI; ++i) i_src[i] = (int8_t)((int)(rng() % 41) - 20);
  for (int64_t i = 0; i < (int64_t)B * HI; ++i) i_ten[i] = (int8_t)((int)(rng() % 255) - 127);
  int8_t* d_isrc = upload(i_src, (size_t)N * HI);
  int8_t* d_iout = upload(i_ten, (size_t)B * HI);            /* include_self: the call folds into the old rows */
  CHECK(rua_segment_reduce(&buckets, d_perm, d_isrc, d_iout, HI, RUA_I8, RUA_MEAN, 1, 0, NULL, 0, NULL, NULL, s));
  if (rua_segment_reduce(&buckets, d_perm, d_isrc, d_iout, HI, RUA_I8, RUA_LOGSUMEXP, 1, 0, NULL, 0, NULL, NULL, s) != RUA_EINVAL) return 1;

  /* ---- ABI 4: the host sort on the helper thread gives the same order; the two host scans next to it */
  int64_t* sorted2 = malloc(B * sizeof *sorted2), *h_boff = malloc((size_t)T * 8), *h_off = malloc(B * 8);
  CHECK(rua_host_sort_desc_begin(lens, B, sorted2, 3));
  CHECK(rua_host_pack_scans(lens, B, bsz, T, h_boff, h_off));
  CHECK(rua_host_sort_desc_end());
  CHECK(hipStreamSynchronize(s));

  /* ---- checks */
  float* packed = malloc((size_t)N * H * 4), *sum = malloc((size_t)B * H * 4), *mx = malloc((size_t)B * H * 4);
  float* gin = malloc((size_t)N * H * 4), *scat = malloc((size_t)B * H * 4);
  int64_t* uns = malloc(B * 8), *boff = malloc((size_t)T * 8), *perm = malloc((size_t)N * 8);
  CHECK(hipMemcpy(packed, d_pack, (size_t)N * H * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(sum, d_sum, (size_t)B * H * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(mx, d_max, (size_t)B * H * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(gin, d_gin, (size_t)N * H * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(scat, d_scat, (size_t)B * H * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(uns, d_uns, B * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(boff, d_boff, (size_t)T * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(perm, d_perm, (size_t)N * 8, hipMemcpyDeviceToHost));

  float gmin = INFINITY;                                   /* the reference's initial = tensor.min() (reduce.py:35) */
  for (int64_t i = 0; i < N * H; ++i) gmin = data[i] < gmin ? data[i] : gmin;
  int64_t row = 0, acc = 0;
  for (int64_t t = 0; t < T; ++t) { bad += boff[t] != acc; acc += bsz[t]; }
  for (int b = 0; b < B; ++b) {
    bad += sorted[uns[b]] != b;
    for (int h = 0; h < H; ++h) {
      double ref = 0.0, rabs = 0.0;
      float m = -INFINITY;
      int ties = 0;
      for (int64_t t = 0; t < lens[b]; ++t) {
        const float x = data[(row + t) * H + h];
        bad += packed[(boff[t] + uns[b]) * H + h] != x;                       /* PACK row = boff[t] + unsorted[b] */
        ref += x; rabs += fabs(x);
        if (x > m) { m = x; ties = 1; } else if (x == m) ++ties;
      }
      bad += fabs((double)sum[b * H + h] - ref) > 1e-5 * rabs + 1e-6;
      bad += fabs((double)scat[b * H + h] - ref) > 1e-5 * rabs + 1e-6;
      bad += mx[b * H + h] != (lens[b] ? m : gmin);
      const float g = gout[b * H + h];
      for (int64_t t = 0; t < lens[b]; ++t) {
        const float x = data[(row + t) * H + h];
        const float want = x == m ? (g > 0.0f ? g / (float)ties : g) : 0.0f;   /* torch.segment_reduce's tie rule */
        bad += gin[(row + t) * H + h] != want;
      }
    }
    row += lens[b];
  }
  /* buckets: rows of destination b in ascending order of their position in the shuffled input */
  {
    int64_t* counts = malloc(B * 8), *boffs = malloc(B * 8);
    CHECK(hipMemcpy(counts, d_counts, B * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(boffs, d_boffs, B * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
      bad += counts[b] != lens[b];
      for (int64_t k = 0; k < counts[b]; ++k) {
        const int64_t i = perm[boffs[b] + k];
        bad += i < 0 || i >= N || sh_index[i] != b || (k && perm[boffs[b] + k - 1] >= i);
      }
    }
  }
  /* integer scatter_mean (include_self) against ATen's steps written out */
  {
    int8_t* iout = malloc((size_t)B * HI);
    CHECK(hipMemcpy(iout, d_iout, (size_t)B * HI, hipMemcpyDeviceToHost));
    int8_t* acc8 = malloc((size_t)B * HI);
    memcpy(acc8, i_ten, (size_t)B * HI);
    for (int64_t i = 0; i < N; ++i)                        /* index_add in source order, wrapping */
      for (int h = 0; h < HI; ++h) {
        int8_t* a = acc8 + sh_index[i] * HI + h;
        *a = (int8_t)(uint8_t)((uint8_t)*a + (uint8_t)i_src[i * HI + h]);
      }
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < HI; ++h) {
        const int8_t accv = acc8[b * HI + h];
        int8_t cnt = (int8_t)(uint8_t)(uint64_t)(lens[b] + 1);
        if (cnt == 0) cnt = 1;
        int q = (int)accv / (int)cnt;
        if ((((int)accv < 0) != ((int)cnt < 0)) && (int)accv % (int)cnt != 0) --q;
        bad += iout[b * HI + h] != (int8_t)q;
      }
    for (int b = 0; b < B; ++b) bad += sorted2[b] != sorted[b];
    int64_t run = 0;
    for (int64_t t = 0; t < T; ++t) { bad += h_boff[t] != run; run += bsz[t]; }
    run = 0;
    for (int b = 0; b < B; ++b) { bad += h_off[b] != run; run += lens[b]; }
  }
  printf("abi_pipeline: target %s, %d sequences, %lld rows, %d mismatches\n", rua_build_target(), (int)B, (long long)N, bad);
  return bad ? 2 : 0;
}
