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
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include <pthread.h>

#include "rua.h"

namespace {

struct KV { int64_t key, idx; };
inline bool before(const KV& a, const KV& b) { return a.key > b.key; }   // descending, key only

constexpr int64_t LEAF = 16;          // segments up to this size are left to the insertion sort
constexpr int64_t SPAWN_MIN = 4096;   // right halves at least this long become tasks of their own

inline int64_t floor_log2(int64_t n) { int64_t l = 0; while (n > 1) { n >>= 1; ++l; } return l; }

inline void median_to_first(KV* result, KV* a, KV* b, KV* c) {
  if (before(*a, *b)) {
    if (before(*b, *c)) std::swap(*result, *b);
    else if (before(*a, *c)) std::swap(*result, *c);
    else std::swap(*result, *a);
  } else if (before(*a, *c)) std::swap(*result, *a);
  else if (before(*b, *c)) std::swap(*result, *c);
  else std::swap(*result, *b);
}

inline KV* partition_pivot(KV* first, KV* last) {
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

// stable insertion sort of one leaf segment
inline void leaf_sort(KV* first, KV* last) {
  for (KV* i = first + 1; i < last; ++i) {
    const KV v = *i;
    KV* j = i;
    while (j > first && before(v, *(j - 1))) { *j = *(j - 1); --j; }
    *j = v;
  }
}

struct Task { KV* first; KV* last; int64_t depth; };

std::atomic<int64_t> g_heap_segments{0};   // diagnostics: segments that ran out of depth budget, all calls so far

class Pool {
 public:
  static Pool& get() {
    static Pool* p = new Pool();      // never destroyed: no join at process exit
    return *p;
  }

  // sort [first, last) with up to `threads` threads (the caller is one of them)
  void sort(KV* first, KV* last, int threads) {
    const int64_t n = last - first;
    if (n < 2) return;
    const int64_t depth = 2 * floor_log2(n);
    if (threads <= 1 || n < 2 * SPAWN_MIN) {
      run(Task{first, last, depth}, nullptr);
      return;
    }
    std::lock_guard<std::mutex> serial(entry_);      // one parallel sort at a time; others wait their turn
    ensure_workers(threads - 1);
    Job job;
    job.pending.store(1);
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &job;
      active_limit_ = threads - 1;
      q_.push_back(Task{first, last, depth});
    }
    cv_.notify_all();
    // the caller works too
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(m_);
        if (q_.empty()) {
          if (job.pending.load() == 0) break;
          done_cv_.wait(g, [&] { return !q_.empty() || job.pending.load() == 0; });
          if (q_.empty()) { if (job.pending.load() == 0) break; else continue; }
        }
        t = q_.front();
        q_.pop_front();
      }
      run(t, &job);
      finish_one(job);
    }
    std::lock_guard<std::mutex> g(m_);
    job_ = nullptr;
  }

  void forget_workers_after_fork() {     // the child of a fork() has none of the parent's threads
    new (&m_) std::mutex();
    new (&entry_) std::mutex();
    new (&cv_) std::condition_variable();
    new (&done_cv_) std::condition_variable();
    n_workers_ = 0;
    q_.clear();
    job_ = nullptr;
  }

 private:
  struct Job { std::atomic<int64_t> pending{0}; };

  Pool() { pthread_atfork(nullptr, nullptr, [] { Pool::get().forget_workers_after_fork(); }); }

  void ensure_workers(int want) {
    std::lock_guard<std::mutex> g(m_);
    while (n_workers_ < want) {
      const int id = n_workers_++;
      std::thread([this, id] { worker(id); }).detach();
    }
  }

  void worker(int id) {
    for (;;) {
      Task t;
      Job* job;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !q_.empty() && id < active_limit_; });
        t = q_.front();
        q_.pop_front();
        job = job_;
      }
      run(t, job);
      finish_one(*job);
    }
  }

  void finish_one(Job& job) {
    if (job.pending.fetch_sub(1) == 1) {
      std::lock_guard<std::mutex> g(m_);
      done_cv_.notify_all();
    }
  }

  void push(const Task& t, Job* job) {
    job->pending.fetch_add(1);
    {
      std::lock_guard<std::mutex> g(m_);
      q_.push_back(t);
    }
    cv_.notify_all();
    done_cv_.notify_all();      // the caller may be waiting for work as well
  }

  // the introsort loop of one segment; right halves large enough become tasks (job != nullptr)
  void run(Task t, Job* job) {
    KV* first = t.first;
    KV* last = t.last;
    int64_t depth = t.depth;
    while (last - first > LEAF) {
      if (depth == 0) {                       // budget spent: heap sort of the segment
        g_heap_segments.fetch_add(1, std::memory_order_relaxed);
        std::make_heap(first, last, before);
        std::sort_heap(first, last, before);
        return;
      }
      --depth;
      KV* cut = partition_pivot(first, last);
      if (job && last - cut >= SPAWN_MIN) push(Task{cut, last, depth}, job);
      else run(Task{cut, last, depth}, job);
      last = cut;
    }
    leaf_sort(first, last);
  }

  std::mutex m_, entry_;
  std::condition_variable cv_, done_cv_;
  std::deque<Task> q_;
  Job* job_ = nullptr;
  int n_workers_ = 0;
  int active_limit_ = 0;
};

std::vector<KV>& scratch() {
  static thread_local std::vector<KV> v;      // grows, never shrinks: no allocator traffic per call
  return v;
}

}  // namespace

extern "C" {

int rua_host_sort_desc(const int64_t* keys, int64_t n, int64_t* sorted_indices, int32_t n_threads) {
  if (n < 0 || (n > 0 && (!keys || !sorted_indices))) return RUA_EINVAL;
  if (n == 0) return 0;
  std::vector<KV>& v = scratch();
  if ((int64_t)v.size() < n) v.resize((size_t)n);
  KV* a = v.data();
  for (int64_t i = 0; i < n; ++i) { a[i].key = keys[i]; a[i].idx = i; }
  Pool::get().sort(a, a + n, n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads));
  for (int64_t i = 0; i < n; ++i) sorted_indices[i] = a[i].idx;
  return 0;
}

int64_t rua_host_sort_heap_segments(void) { return g_heap_segments.load(); }

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

}  // extern "C"
