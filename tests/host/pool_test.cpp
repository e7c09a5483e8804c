// pool_test.cpp -- the context's worker pool (kogarashi_amd/csrc/worker_pool.h) on its own, built with -fsanitize=thread / address by
// tests/test_sanitizers.py: tasks that wait for other tasks of the same pool (a proof's assembly joins its MSMs' finishes), threads started
// on demand and reused, a refused thread start as an exception that leaves the pool usable, futures that are waited for on unwind.
#include "../../kogarashi_amd/csrc/worker_pool.h"
#include <atomic>
#include <chrono>
#include <cstdio>

using kg::WorkerPool;

static int fail(const char* what) { std::fprintf(stderr, "pool_test: %s\n", what); return 1; }

int main() {
  {                                                         // 1. many short tasks in turn: one thread serves them all
    WorkerPool p(64);
    long sum = 0;
    for (int i = 0; i < 2000; ++i) sum += p.submit([i] { return i; }).get();
    if (sum != 1999L * 2000 / 2) return fail("sum of task results");
    if (p.threads_started() != 1) return fail("sequential tasks must reuse one thread");
  }
  {                                                         // 2. nested: a task that submits and joins five more (the assembly's shape), eight at once
    WorkerPool p(64);
    std::atomic<int> done{0};
    std::future<int> outer[8];
    for (int k = 0; k < 8; ++k)
      outer[k] = p.submit([&p, &done, k] {
        std::future<int> inner[5];
        for (int j = 0; j < 5; ++j) inner[j] = p.submit([&done, j] { std::this_thread::sleep_for(std::chrono::microseconds(200)); ++done; return j; });
        int s = 0;
        for (auto& f : inner) s += f.get();
        return s + 100 * k;
      });
    for (int k = 0; k < 8; ++k)
      if (outer[k].get() != 10 + 100 * k) return fail("nested result");
    if (done != 40) return fail("nested task count");
    const int after_first = p.threads_started();
    if (after_first < 2 || after_first > 48) return fail("thread count of the nested round");
    for (int rep = 0; rep < 20; ++rep) {                    // the same load again and again: no growth beyond what was ever busy at once
      std::future<int> o2[8];
      for (int k = 0; k < 8; ++k) o2[k] = p.submit([&p] { return p.submit([] { return 7; }).get(); });
      for (auto& f : o2) if (f.get() != 7) return fail("repeat result");
    }
    if (p.threads_started() > 48) return fail("the pool must not grow with the number of calls");
  }
  {                                                         // 3. a refused start: exception, nothing queued, pool usable afterwards within its limit
    WorkerPool p(1);
    std::atomic<bool> release{false};
    std::future<int> busy = p.submit([&release] { while (!release) std::this_thread::yield(); return 1; });
    bool threw = false;
    try { p.submit([] { return 2; }); } catch (const std::system_error&) { threw = true; }
    if (!threw) return fail("a second thread beyond the limit must be refused");
    release = true;
    if (busy.get() != 1) return fail("the running task");
    if (p.submit([] { return 3; }).get() != 3) return fail("the pool after a refusal");
    WorkerPool none(0);
    threw = false;
    try { none.submit([] { return 0; }); } catch (const std::system_error&) { threw = true; }
    if (!threw || none.threads_started() != 0) return fail("a pool of zero threads");
  }
  {                                                         // 4. WaitAll: a frame whose tasks write into its locals waits for them when it unwinds
    WorkerPool p(8);
    int out[4] = {0, 0, 0, 0};
    try {
      std::future<int> f[4];
      kg::WaitAll guard{f, 4};
      for (int i = 0; i < 4; ++i) f[i] = p.submit([&out, i] { std::this_thread::sleep_for(std::chrono::milliseconds(2)); out[i] = i + 1; return 0; });
      throw 1;
    } catch (int) {}
    for (int i = 0; i < 4; ++i) if (out[i] != i + 1) return fail("WaitAll on unwind");
  }
  {                                                         // 5. destruction with idle and never-used workers
    WorkerPool p(16);
    std::future<int> f[6];
    for (auto& x : f) x = p.submit([] { return 1; });
    for (auto& x : f) x.get();
  }
  std::puts("pool_test ok");
  return 0;
}
