// Stress harness for rua_host_sort_desc (tests/test_host_sort.py builds it with -fsanitize=thread and with
// -fsanitize=address,undefined): three threads sort concurrently with varying thread counts, a fourth through the
// helper-thread form (rua_host_sort_desc_begin / _end); every result must equal
// std::sort over (key, index) pairs with the key-only comparator (what ATen runs for torch.sort on the CPU).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <algorithm>
extern "C" int rua_host_sort_desc(const int64_t*, int64_t, int64_t*, int32_t);
extern "C" int rua_host_batch_sizes(const int64_t*, int64_t, int64_t, int64_t*);
extern "C" int rua_host_sort_desc_begin(const int64_t*, int64_t, int64_t*, int32_t);
extern "C" int rua_host_sort_desc_end(void);
extern "C" int rua_host_pack_scans(const int64_t*, int64_t, const int64_t*, int64_t, int64_t*, int64_t*);
static void ref(const std::vector<int64_t>& k, std::vector<int64_t>& o) {
  std::vector<std::pair<int64_t,int64_t>> v(k.size());
  for (size_t i = 0; i < k.size(); ++i) v[i] = {k[i], (int64_t)i};
  std::sort(v.begin(), v.end(), [](auto& a, auto& b){ return a.first > b.first; });
  for (size_t i = 0; i < k.size(); ++i) o[i] = v[i].second;
}
int main() {
  int bad = 0;
  auto worker = [&](int seed) {
    srand(seed);
    for (int rep = 0; rep < 40; ++rep) {
      int64_t n = 1 + rand() % 70000;
      std::vector<int64_t> k(n), o(n), r(n);
      int range = 1 + rand() % 600;
      for (auto& x : k) x = rand() % range;
      rua_host_sort_desc(k.data(), n, o.data(), 1 + rep % 8);
      ref(k, r);
      if (o != r) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
    }
  };
  // a fourth thread drives the helper-thread form (begin / other host work / end) while the three above sort
  auto async_worker = [&](int seed) {
    unsigned st = (unsigned)seed;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return (int)(st >> 8); };
    for (int rep = 0; rep < 40; ++rep) {
      int64_t n = 1 + rnd() % 70000;
      std::vector<int64_t> k(n), o(n), r(n), off(n);
      int range = 1 + rnd() % 600;
      for (auto& x : k) x = rnd() % range;
      if (rua_host_sort_desc_begin(k.data(), n, o.data(), 1 + rep % 5) != 0) { __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED); continue; }
      if (rua_host_sort_desc_begin(k.data(), n, o.data(), 2) == 0) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);   // busy
      rua_host_pack_scans(k.data(), n, nullptr, 0, nullptr, off.data());
      ref(k, r);
      if (rua_host_sort_desc_end() != 0 || o != r) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
      int64_t run = 0;
      for (int64_t i = 0; i < n; ++i) { if (off[i] != run) { __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED); break; } run += k[i]; }
    }
  };
  std::thread a(worker, 1), b(worker, 2), c(worker, 3), d(async_worker, 4);
  a.join(); b.join(); c.join(); d.join();
  printf("mismatches %d\n", bad);
  return bad != 0;
}
