// rua_host.cpp — the HOST side of pack(): the descending order of the lengths and batch_sizes.
//
// The reference computes `sorted_indices` with a host call, torch.sort(token_sizes.cpu(), descending=True)
// (core/view.py:48), whose tie order is not the stable one (SURVEY.md §8a note): ATen's CPU kernel runs the
// C++ library's introsort over (key, index) pairs with a comparator that looks at the key only.  That order is a
// deterministic function of the input, so it can be reproduced — and, unlike the library call, in parallel:
//
//   * quicksort levels: median-of-3 of (first+1, mid, last-1) moved to `first`, then the unguarded Hoare partition
//     of [first+1, last) around it; recurse on the right part, loop on the left, until a segment has <= 16 elements;
//     a depth budget of 2*floor(log2 n), after which a segment is heap-sorted instead;
//   * the closing insertion sort only ever moves an element inside its own <= 16-element segment (everything to
//     the left of a segment compares >= everything in it), so it is a stable sort of every leaf segment by itself.
//
// The two halves of a partition never touch each other again, so the right half goes to another thread.  The
// result is bit-identical to the sequential library sort on every input; torchrua_amd checks exactly that against
// torch.sort itself when the library is first used (torchrua_amd/_meta.py: _host_sort_selftest) and falls back to the
// reference's own call if the two ever disagree.  At B = 65 536 the order takes 0.3-0.5 ms instead of 1.9-3.3 ms,
// which is what a pack() with device-only lengths leaves the GPU idle for.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include <pthread.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "rua.h"

namespace {

// Two element types: (int64 key, int64 index), and — when every key fits 32 bits and n < 2^32, i.e. always for
// sequence lengths — (int32 key, uint32 index): the sort is bound by the bytes it moves, and half the bytes is 1.3x
// the speed.  Same comparisons, same swaps, same permutation.
struct KV16 { int64_t key, idx; };
struct KV8 { int32_t key; uint32_t idx; };
template <typename KV> inline bool before(const KV& a, const KV& b) { return a.key > b.key; }   // descending, key only

constexpr int64_t LEAF = 16;          // segments up to this size are left to the insertion sort
// right halves at least this long become tasks of their own (2 048, 1 024 and 512 measured slower on 4-16 threads:
// profiles/r04_host_sort_scaling.txt)
constexpr int64_t SPAWN_MIN = 4096;

inline int64_t floor_log2(int64_t n) { int64_t l = 0; while (n > 1) { n >>= 1; ++l; } return l; }

template <typename KV> inline void median_to_first(KV* result, KV* a, KV* b, KV* c) {
  if (before(*a, *b)) {
    if (before(*b, *c)) std::swap(*result, *b);
    else if (before(*a, *c)) std::swap(*result, *c);
    else std::swap(*result, *a);
  } else if (before(*a, *c)) std::swap(*result, *a);
  else if (before(*b, *c)) std::swap(*result, *c);
  else std::swap(*result, *b);
}

template <typename KV> inline KV* partition_pivot(KV* first, KV* last) {
  KV* mid = first + (last - first) / 2;
  median_to_first(first, first + 1, mid, last - 1);
  KV* lo = first + 1;
  KV* hi = last;
  const KV* pivot = first;
  for (;;) {
    while (before(*lo, *pivot)) ++lo;
    --hi;
    while (before(*pivot, *hi)) --hi;
    if (!(lo < hi)) return lo;
    std::swap(*lo, *hi);
    ++lo;
  }
}

// The same partition WITHOUT data-dependent branches.  On random keys the two scans above mispredict every other
// comparison (measured: ~3.6 ns per element and level; the whole sort of 65 536 lengths 1.8-2.8 ms on one core).
// What the loop does is fixed by the input alone: the left scan stops at the elements that are not before the pivot
// ("L-stops", key <= pivot), the right scan at those the pivot is not before ("R-stops", key >= pivot); the k-th L-stop
// from the left is swapped with the k-th R-stop from the right for as long as it lies to the left of it, and the
// scans never look at a swapped element before they cross.  So: list both kinds of stops in one branch-free pass,
// find how many pairs are in order (the predicate is monotone: binary search), swap them, and the cut is where the
// left scan ends — the next L-stop or the last swapped-in one, whichever comes first.  Same swaps, same cut, same
// permutation as partition_pivot (tests/test_host_sort.py compares the whole sort with torch.sort).
struct StopLists { std::vector<uint32_t> l, r; };
inline StopLists& stop_lists() {
  static thread_local StopLists s;           // grow, never shrink
  return s;
}

constexpr int64_t BRANCHLESS_MIN = 48;       // shorter segments keep the plain loop (less set-up than it saves)

// listing the stops: one branch-free pass (2 compares, 2 stores per element) ...
template <typename KV>
inline void list_stops(const KV* a, int64_t n, uint32_t* __restrict__ L, uint32_t* __restrict__ R, int64_t& nl_out,
                       int64_t& nr_out) {
  const auto pk = a[0].key;
  int64_t nl = 0, nr = 0;
  for (int64_t i = 1; i < n; ++i) {
    const auto k = a[i].key;
    L[nl] = (uint32_t)i;
    nl += (k <= pk);
    R[nr] = (uint32_t)i;
    nr += (k >= pk);
  }
  nl_out = nl;
  nr_out = nr;
}

// ... or, for the narrow elements on a host with AVX-512, 16 elements per step: the keys are the even dwords of two
// loads, two compares give the masks, and the positions are compressed IN A REGISTER and stored whole (a compressing
// store to memory is microcoded on some cores; the lists carry 16 elements of slack for the whole-register store).
#if defined(__x86_64__)
__attribute__((target("avx512f"))) inline void list_stops_avx512(const KV8* a, int64_t n, uint32_t* __restrict__ L,
                                                                 uint32_t* __restrict__ R, int64_t& nl_out, int64_t& nr_out) {
  const int32_t pk = a[0].key;
  const __m512i even = _mm512_set_epi32(30, 28, 26, 24, 22, 20, 18, 16, 14, 12, 10, 8, 6, 4, 2, 0);
  const __m512i pkv = _mm512_set1_epi32(pk);
  const __m512i step = _mm512_set1_epi32(16);
  __m512i pos = _mm512_set_epi32(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
  int64_t nl = 0, nr = 0, i = 1;
  for (; i + 16 <= n; i += 16) {
    const __m512i v0 = _mm512_loadu_si512((const void*)(a + i));
    const __m512i v1 = _mm512_loadu_si512((const void*)(a + i + 8));
    const __m512i keys = _mm512_permutex2var_epi32(v0, even, v1);
    const __mmask16 ml = _mm512_cmple_epi32_mask(keys, pkv);
    const __mmask16 mr = _mm512_cmpge_epi32_mask(keys, pkv);
    _mm512_storeu_si512((void*)(L + nl), _mm512_maskz_compress_epi32(ml, pos));
    _mm512_storeu_si512((void*)(R + nr), _mm512_maskz_compress_epi32(mr, pos));
    nl += __builtin_popcount((unsigned)ml);
    nr += __builtin_popcount((unsigned)mr);
    pos = _mm512_add_epi32(pos, step);
  }
  for (; i < n; ++i) {
    const int32_t k = a[i].key;
    L[nl] = (uint32_t)i;
    nl += (k <= pk);
    R[nr] = (uint32_t)i;
    nr += (k >= pk);
  }
  nl_out = nl;
  nr_out = nr;
}
const bool g_avx512 = [] {
  __builtin_cpu_init();        // (this runs as a static initialiser of a shared library: do not rely on constructor order)
  return __builtin_cpu_supports("avx512f") && std::getenv("RUA_HOST_SORT_SCALAR") == nullptr;
}();
#else
const bool g_avx512 = false;
#endif

inline void list_stops_fast(const KV8* a, int64_t n, uint32_t* L, uint32_t* R, int64_t& nl, int64_t& nr) {
#if defined(__x86_64__)
  if (g_avx512) { list_stops_avx512(a, n, L, R, nl, nr); return; }
#endif
  list_stops(a, n, L, R, nl, nr);
}
inline void list_stops_fast(const KV16* a, int64_t n, uint32_t* L, uint32_t* R, int64_t& nl, int64_t& nr) {
  list_stops(a, n, L, R, nl, nr);
}

template <typename KV>
inline KV* partition_pivot_lists(KV* first, KV* last) {
  const int64_t n = last - first;
  KV* mid = first + n / 2;
  median_to_first(first, first + 1, mid, last - 1);
  StopLists& S = stop_lists();
  if ((int64_t)S.l.size() < n + 16) { S.l.resize((size_t)n + 16); S.r.resize((size_t)n + 16); }
  uint32_t* __restrict__ L = S.l.data();
  uint32_t* __restrict__ R = S.r.data();
  int64_t nl = 0, nr = 0;
  list_stops_fast(first, n, L, R, nl, nr);
  // pairs (L[k], R[nr-1-k]) in order: a prefix
  int64_t lo = 0, hi = nl < nr ? nl : nr;
  while (lo < hi) {
    const int64_t m = (lo + hi) >> 1;
    if (L[m] < R[nr - 1 - m]) lo = m + 1; else hi = m;
  }
  const int64_t K = lo;
  for (int64_t k = 0; k < K; ++k) std::swap(first[L[k]], first[R[nr - 1 - k]]);
  int64_t cut = K < nl ? (int64_t)L[K] : n;
  if (K > 0 && (int64_t)R[nr - K] < cut) cut = R[nr - K];
  return first + cut;
}

// stable sort of one leaf segment (what the closing insertion sort makes of it), by counting: the place of an element
// is the number of elements before it in the order — those with a larger key, and those with an equal key to its left
template <typename KV>
inline void leaf_sort(KV* first, KV* last) {
  const int m = (int)(last - first);
  if (m < 2) return;
  KV tmp[LEAF];
  for (int i = 0; i < m; ++i) tmp[i] = first[i];
  for (int i = 0; i < m; ++i) {
    const auto ki = tmp[i].key;
    int r = 0;
    for (int j = 0; j < m; ++j) r += (int)(tmp[j].key > ki) | ((int)(tmp[j].key == ki) & (int)(j < i));
    first[r] = tmp[i];
  }
}

#if defined(__x86_64__)
// the same with AVX-512 for the narrow elements whose keys lie in [0, 2^27) (lengths always do): a key and its position
// in the leaf make ONE 32-bit word (key << 4 | 15 - position) — no two words are equal, a larger word comes earlier —
// so the place of an element is the count of larger words: one compare of the 16 words against a broadcast, one popcount.
__attribute__((target("avx512f"))) inline void leaf_sort_avx512(KV8* first, int m) {
  const __m512i even = _mm512_set_epi32(30, 28, 26, 24, 22, 20, 18, 16, 14, 12, 10, 8, 6, 4, 2, 0);
  const __m512i rev = _mm512_set_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  // masked loads: only the leaf's own elements are touched (the neighbouring segment may belong to another thread)
  const unsigned em = (1u << m) - 1u;
  const __m512i v0 = _mm512_maskz_loadu_epi64((__mmask8)(em & 0xffu), (const void*)first);
  const __m512i v1 = _mm512_maskz_loadu_epi64((__mmask8)(em >> 8), (const void*)(first + 8));
  const __m512i keys = _mm512_permutex2var_epi32(v0, even, v1);
  const __mmask16 valid = (__mmask16)((1u << m) - 1u);
  const __m512i words = _mm512_maskz_or_epi32(valid, _mm512_slli_epi32(keys, 4), rev);   // absent lanes: 0, never larger
  alignas(64) KV8 tmp[16];
  alignas(64) uint32_t w[16];
  _mm512_store_si512((void*)tmp, v0);
  _mm512_store_si512((void*)(tmp + 8), v1);
  _mm512_store_si512((void*)w, words);
  for (int i = 0; i < m; ++i) {
    const int r = __builtin_popcount((unsigned)_mm512_cmpgt_epu32_mask(words, _mm512_set1_epi32((int)w[i])));
    first[r] = tmp[i];
  }
}
#endif
inline void leaf_sort_fast(KV8* first, KV8* last, bool small_keys) {
#if defined(__x86_64__)
  if (g_avx512 && small_keys) { if (last - first >= 2) leaf_sort_avx512(first, (int)(last - first)); return; }
#endif
  leaf_sort(first, last);
}
inline void leaf_sort_fast(KV16* first, KV16* last, bool) { leaf_sort(first, last); }

struct Task { void* first; void* last; int64_t depth; bool narrow; bool small_keys; const void* base; int64_t* out; };

std::atomic<int64_t> g_heap_segments{0};   // diagnostics: segments that ran out of depth budget, all calls so far

class Pool {
 public:
  static Pool& get() {
    static Pool* p = new Pool();      // never destroyed: no join at process exit
    return *p;
  }

  // sort [first, last) with up to `threads` threads (the caller is one of them)
  // `out`: sorted_indices.  Every finished piece of the array — a leaf, a heap-sorted segment — writes its own slice
  // of it (its elements are in their final places, and the piece is hot in that thread's cache): the closing pass over
  // the whole array (65 536 elements: ~20 us on one thread, after everything else) is gone.
  template <typename KV>
  void sort(KV* first, KV* last, int threads, int64_t* out, bool small_keys = false) {
    const int64_t n = last - first;
    if (n < 2) {
      if (n == 1) out[0] = (int64_t)first->idx;
      return;
    }
    const int64_t depth = 2 * floor_log2(n);
    const Task root{first, last, depth, sizeof(KV) == sizeof(KV8), small_keys, first, out};
    if (threads <= 1 || n < 2 * SPAWN_MIN) {
      run(root, false);
      return;
    }
    std::lock_guard<std::mutex> serial(entry_);      // one parallel sort at a time; others wait their turn
    ensure_workers(threads - 1);
    {
      std::lock_guard<std::mutex> g(m_);
      pending_.store(1, std::memory_order_release);  // tasks queued or running; 0 = the job is over
      q_.push_back(root);
      queued_.store(1, std::memory_order_release);
      active_ = true;
      active_limit_ = threads - 1;
    }
    cv_.notify_all();
    work(-1);                                        // the caller works too, until the job is over
    std::lock_guard<std::mutex> g(m_);
    active_ = false;
    active_limit_ = 0;
  }

  void forget_workers_after_fork() {     // the child of a fork() has none of the parent's threads
    new (&m_) std::mutex();
    new (&entry_) std::mutex();
    new (&cv_) std::condition_variable();
    n_workers_ = 0;
    q_.clear();
    queued_.store(0);
    pending_.store(0);
    active_ = false;
    active_limit_ = 0;
  }

 private:
  Pool() { pthread_atfork(nullptr, nullptr, [] { Pool::get().forget_workers_after_fork(); }); }

  static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }

  void ensure_workers(int want) {
    std::lock_guard<std::mutex> g(m_);
    while (n_workers_ < want) {
      const int id = n_workers_++;
      std::thread([this, id] { for (;;) work(id); }).detach();
    }
  }

  // Take tasks until the job is over.  A worker (id >= 0) sleeps on the condition variable BETWEEN jobs (no CPU burnt
  // while the library is idle); within a job — it lasts a few hundred microseconds — whoever finds the queue empty spins
  // for the next task or the end instead of going to sleep (a futex wake-up per hand-off was a fifth of the sort).
  // All state lives in the pool (one job at a time: entry_), so a worker never looks at a finished caller's stack.
  void work(int id) {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(m_);
        if (id >= 0) cv_.wait(g, [&] { return active_ && id < active_limit_; });
        if (q_.empty()) {
          const bool over = pending_.load(std::memory_order_acquire) == 0;
          g.unlock();
          if (over) {
            if (id < 0) return;                     // the caller: done
            std::this_thread::yield();              // a worker: the caller is about to clear active_
            continue;
          }
          // (bounded: under a CPU quota or with oversubscribed ranks the spinners could starve the one thread that
          // holds the only task — after ~4 000 pauses, a few tens of microseconds, they give their core away instead)
          for (int spins = 0; queued_.load(std::memory_order_acquire) == 0 && pending_.load(std::memory_order_acquire) != 0; ++spins) {
            if (spins < 4096) cpu_relax(); else std::this_thread::yield();
          }
          continue;
        }
        t = q_.front();
        q_.pop_front();
        queued_.store((int64_t)q_.size(), std::memory_order_release);
      }
      run(t, true);
      pending_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }

  void push(const Task& t) {
    pending_.fetch_add(1, std::memory_order_acq_rel);
    std::lock_guard<std::mutex> g(m_);
    q_.push_back(t);
    queued_.store((int64_t)q_.size(), std::memory_order_release);
  }

  void run(const Task& t, bool spawn) {
    if (t.narrow) run_t<KV8>(t, spawn); else run_t<KV16>(t, spawn);
  }

  // the introsort loop of one segment; right halves large enough become tasks (spawn)
  template <typename KV>
  void run_t(const Task& t, bool spawn) {
    KV* first = (KV*)t.first;
    KV* last = (KV*)t.last;
    int64_t depth = t.depth;
    while (last - first > LEAF) {
      if (depth == 0) {                       // budget spent: heap sort of the segment
        g_heap_segments.fetch_add(1, std::memory_order_relaxed);
        std::make_heap(first, last, before<KV>);
        std::sort_heap(first, last, before<KV>);
        write_out(first, last, t);
        return;
      }
      --depth;
      KV* cut = last - first >= BRANCHLESS_MIN ? partition_pivot_lists(first, last) : partition_pivot(first, last);
      if (spawn && last - cut >= SPAWN_MIN) push(Task{cut, last, depth, t.narrow, t.small_keys, t.base, t.out});
      else run_t<KV>(Task{cut, last, depth, t.narrow, t.small_keys, t.base, t.out}, spawn);
      last = cut;
    }
    leaf_sort_fast(first, last, t.small_keys);
    write_out(first, last, t);
  }

  template <typename KV>
  static inline void write_out(const KV* first, const KV* last, const Task& t) {
    int64_t* __restrict__ o = t.out + (first - static_cast<const KV*>(t.base));
    for (const KV* p = first; p != last; ++p) *o++ = (int64_t)p->idx;
  }

  std::mutex m_, entry_;
  std::condition_variable cv_;
  std::deque<Task> q_;
  std::atomic<int64_t> queued_{0}, pending_{0};
  bool active_ = false;
  int n_workers_ = 0;
  int active_limit_ = 0;
};

std::vector<KV16>& scratch() {
  static thread_local std::vector<KV16> v;      // grows, never shrinks: no allocator traffic per call
  return v;
}

}  // namespace

extern "C" {

int rua_host_sort_desc(const int64_t* keys, int64_t n, int64_t* sorted_indices, int32_t n_threads) {
  if (n < 0 || (n > 0 && (!keys || !sorted_indices))) return RUA_EINVAL;
  if (n == 0) return 0;
  std::vector<KV16>& v = scratch();
  if ((int64_t)v.size() < n + 16) v.resize((size_t)n + 16);
  const int threads = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  // the narrow elements are filled optimistically while the key range is taken in the same pass (sequence lengths
  // always fit: a second, wide fill happens only for keys beyond 32 bits)
  if (n <= 0xffffffffLL) {
    KV8* a = reinterpret_cast<KV8*>(v.data());       // (the scratch is sized for the wide elements)
    int64_t lo = keys[0], hi = keys[0];
    for (int64_t i = 0; i < n; ++i) {
      const int64_t k = keys[i];
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
      a[i].key = (int32_t)k;
      a[i].idx = (uint32_t)i;
    }
    if (lo >= INT32_MIN && hi <= INT32_MAX) {
      Pool::get().sort(a, a + n, threads, sorted_indices, lo >= 0 && hi < (1 << 27));
      return 0;
    }
  }
  KV16* a = v.data();
  for (int64_t i = 0; i < n; ++i) { a[i].key = keys[i]; a[i].idx = i; }
  Pool::get().sort(a, a + n, threads, sorted_indices);
  return 0;
}

int64_t rua_host_sort_heap_segments(void) { return g_heap_segments.load(); }

}  // extern "C"

namespace {
// rua_host_sort_desc on a helper thread: pack() with device-only lengths leaves the GPU idle from the read-back of the
// lengths until the mover is launched, and the sort is half of that interval — so everything else the host has to do in
// it (batch_sizes, the offset scans, allocations, descriptors) runs on the calling thread WHILE the pool sorts.
// One job at a time; the helper sleeps between jobs.
class AsyncSort {
 public:
  static AsyncSort& get() {
    static AsyncSort* a = new AsyncSort();    // never destroyed: no join at process exit
    return *a;
  }
  int begin(const int64_t* keys, int64_t n, int64_t* out, int32_t threads) {
    std::unique_lock<std::mutex> g(m_);
    if (state_.load(std::memory_order_acquire) != IDLE) return RUA_EINVAL;     // a job is already posted
    if (!started_) {
      std::thread([this] { loop(); }).detach();
      started_ = true;
    }
    keys_ = keys; n_ = n; out_ = out; threads_ = threads;
    state_.store(POSTED, std::memory_order_release);
    cv_.notify_all();
    return 0;
  }
  int end() {
    if (state_.load(std::memory_order_acquire) == IDLE) return RUA_EINVAL;
    // the job lasts ~0.2 ms and the caller arrives towards its end: spin briefly, then give the core away
    for (int spins = 0; state_.load(std::memory_order_acquire) != DONE; ++spins) {
      if (spins < 20000) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
      } else {
        std::this_thread::yield();
      }
    }
    const int rc = rc_;
    state_.store(IDLE, std::memory_order_release);
    return rc;
  }
  void forget_after_fork() {
    new (&m_) std::mutex();
    new (&cv_) std::condition_variable();
    started_ = false;
    state_.store(IDLE);
  }

 private:
  enum { IDLE = 0, POSTED = 1, RUNNING = 2, DONE = 3 };
  AsyncSort() { pthread_atfork(nullptr, nullptr, [] { AsyncSort::get().forget_after_fork(); }); }
  void loop() {
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return state_.load(std::memory_order_acquire) == POSTED; });
        state_.store(RUNNING, std::memory_order_release);
      }
      rc_ = rua_host_sort_desc(keys_, n_, out_, threads_);
      state_.store(DONE, std::memory_order_release);
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<int> state_{IDLE};
  bool started_ = false;
  const int64_t* keys_ = nullptr;
  int64_t n_ = 0;
  int64_t* out_ = nullptr;
  int32_t threads_ = 1;
  int rc_ = 0;
};
}  // namespace

extern "C" {

int rua_host_sort_desc_begin(const int64_t* keys, int64_t n, int64_t* sorted_indices, int32_t n_threads) {
  if (n < 0 || (n > 0 && (!keys || !sorted_indices))) return RUA_EINVAL;
  return AsyncSort::get().begin(keys, n, sorted_indices, n_threads);
}

int rua_host_sort_desc_end(void) { return AsyncSort::get().end(); }

int rua_host_batch_sizes(const int64_t* lens, int64_t B, int64_t T, int64_t* batch_sizes) {
  if (B < 0 || T < 0 || (B > 0 && !lens) || (T > 0 && !batch_sizes)) return RUA_EINVAL;
  if (T == 0) return 0;
  // batch_sizes[t] = #{b : len[b] > t} = B - #{b : len[b] <= t}: a histogram and a running sum
  for (int64_t t = 0; t < T; ++t) batch_sizes[t] = 0;
  int64_t beyond = 0;                       // lengths > T - 1 never leave the count
  for (int64_t b = 0; b < B; ++b) {
    const int64_t len = lens[b];
    if (len < 0) return RUA_EINVAL;
    if (len < T) ++batch_sizes[len]; else ++beyond;
  }
  (void)beyond;
  int64_t at_most = 0;
  for (int64_t t = 0; t < T; ++t) {
    at_most += batch_sizes[t];
    batch_sizes[t] = B - at_most;
  }
  return 0;
}

int rua_host_pack_scans(const int64_t* lens, int64_t B, const int64_t* batch_sizes, int64_t T, int64_t* boff,
                        int64_t* off) {
  if (B < 0 || T < 0 || (B > 0 && !lens) || (T > 0 && !batch_sizes)) return RUA_EINVAL;
  int64_t run = 0;
  if (boff)
    for (int64_t t = 0; t < T; ++t) { boff[t] = run; run += batch_sizes[t]; }
  run = 0;
  if (off)
    for (int64_t b = 0; b < B; ++b) { off[b] = run; run += lens[b]; }
  return 0;
}

}  // extern "C"
