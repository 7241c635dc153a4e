// Developer tool (not product, not shipped in the .so): builds an input that drives the introsort of
// rua_host.cpp — and with it the C++ library sort behind torch.sort — through its depth budget into the heap-sort
// branch, with McIlroy's adversary ("A Killer Adversary for Quicksort", 1999): keys stay undecided ("gas") until a
// comparison needs them, and the element the sort seems to be using as a pivot is frozen to the next smallest
// value, so every partition peels off almost nothing.  Output: tests/golden/sort_killer.npy-style text (one key
// per line) on stdout; the committed fixture is checked against torch.sort in tests/test_host_sort.py.
//
//   g++ -O2 -std=c++17 scripts/exp/sort_killer.cpp -o /tmp/sort_killer && /tmp/sort_killer 3000 > keys.txt
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

static std::vector<int64_t> val;
static int64_t gas, nsolid, candidate;
static long heap_segments = 0;

// "a sorts before b" of the adversary
static bool adv_before(int64_t x, int64_t y) {
  if (val[x] == gas && val[y] == gas) {
    if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++;
  }
  if (val[x] == gas) candidate = x;
  else if (val[y] == gas) candidate = y;
  return val[x] < val[y];
}

// the same steps as rua_host.cpp, over element ids, with the adversary as comparator
using It = int64_t*;
static void median_to_first(It result, It a, It b, It c) {
  if (adv_before(*a, *b)) {
    if (adv_before(*b, *c)) std::swap(*result, *b);
    else if (adv_before(*a, *c)) std::swap(*result, *c);
    else std::swap(*result, *a);
  } else if (adv_before(*a, *c)) std::swap(*result, *a);
  else if (adv_before(*b, *c)) std::swap(*result, *c);
  else std::swap(*result, *b);
}
static It partition_pivot(It first, It last) {
  It mid = first + (last - first) / 2;
  median_to_first(first, first + 1, mid, last - 1);
  It lo = first + 1, hi = last;
  for (;;) {
    while (adv_before(*lo, *first)) ++lo;
    --hi;
    while (adv_before(*first, *hi)) --hi;
    if (!(lo < hi)) return lo;
    std::swap(*lo, *hi);
    ++lo;
  }
}
static void run(It first, It last, int64_t depth) {
  while (last - first > 16) {
    if (depth == 0) {
      ++heap_segments;
      std::make_heap(first, last, adv_before);
      std::sort_heap(first, last, adv_before);
      return;
    }
    --depth;
    It cut = partition_pivot(first, last);
    run(cut, last, depth);
    last = cut;
  }
  for (It i = first + 1; i < last; ++i) {
    int64_t v = *i;
    It j = i;
    while (j > first && adv_before(v, *(j - 1))) { *j = *(j - 1); --j; }
    *j = v;
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 3000;
  val.assign(n, 0);
  gas = n;                       // larger than every solid value
  for (auto& v : val) v = gas;
  nsolid = 0;
  candidate = 0;
  std::vector<int64_t> ids(n);
  for (int64_t i = 0; i < n; ++i) ids[i] = i;
  int64_t depth = 0;
  for (int64_t m = n; m > 1; m >>= 1) depth += 2;
  run(ids.data(), ids.data() + n, depth);
  fprintf(stderr, "n=%lld heap-sorted segments: %ld\n", (long long)n, heap_segments);
  // `before(a, b)` of the product is key[a] > key[b]: hand out keys that order like -val
  for (int64_t i = 0; i < n; ++i) printf("%lld\n", (long long)(n - val[i]));
  return heap_segments > 0 ? 0 : 1;
}
