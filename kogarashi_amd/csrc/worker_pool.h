// worker_pool.h -- the context's host worker threads: host finishes of MSMs (the 255-step double-and-add), proof assemblies, the uploader
// of a host-scalar call, the per-context drivers of the sharded entries.
//
// Up to round 5 every such task was a std::async(std::launch::async, ...) or a std::thread of its own: a thread start (30-40 us) per MSM,
// per slice, per proof part -- on the critical path of short calls.  The pool starts a thread only when a task arrives and no worker is
// idle (tasks may wait for other tasks of the same pool: a proof's assembly joins its MSMs' finishes -- so a task never queues behind a
// busy worker; it gets an idle one or a new one), and the threads live until the context is destroyed: after the first calls no thread
// is created any more (tests/test_gpu_holes.py counts them).  A thread that cannot be started (std::system_error: RLIMIT_NPROC, or the
// KG_POOL_MAX_THREADS test knob) leaves the pool as it was and surfaces as an exception in submit() -- kg_guarded turns it into a status.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace kg {

class WorkerPool {
 public:
  explicit WorkerPool(int max_threads) : max_threads_(max_threads) {}
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : threads_) if (t.joinable()) t.join();
  }
  // runs fn on a worker thread; the future carries its status.  Throws std::system_error when the task needs a new thread and none can be started.
  std::future<int> submit(std::function<int()> fn) {
    Task task{std::move(fn), std::promise<int>()};
    std::future<int> fut = task.result.get_future();
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (queue_.size() + 1 > (size_t)(idle_ + spawning_)) {   // no idle (or just started) worker left for this task: one more thread
        if (max_threads_ >= 0 && (int)threads_.size() >= max_threads_) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "worker pool: thread limit");
        threads_.emplace_back([this] { run(); });           // (throws std::system_error when the OS refuses)
        ++started_;
        ++spawning_;
      }
      queue_.push_back(std::move(task));
    }
    cv_.notify_one();
    return fut;
  }
  int threads_started() const {
    std::lock_guard<std::mutex> lk(mu_);
    return started_;
  }

 private:
  struct Task { std::function<int()> fn; std::promise<int> result; };
  void run() {
    std::unique_lock<std::mutex> lk(mu_);
    --spawning_;
    ++idle_;
    for (;;) {
      cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) { --idle_; return; }              // stop_, nothing left
      --idle_;
      Task task = std::move(queue_.front());
      queue_.pop_front();
      lk.unlock();
      int rc = 0;
      std::exception_ptr err;
      try { rc = task.fn(); } catch (...) { err = std::current_exception(); }
      // idle again BEFORE the result is published: whoever has seen a task's result finds its worker counted as free (a caller that
      // submits, waits and submits again reuses the one thread instead of racing the worker back to its wait)
      lk.lock();
      ++idle_;
      lk.unlock();
      if (err) task.result.set_exception(err); else task.result.set_value(rc);
      lk.lock();
    }
  }
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Task> queue_;
  std::vector<std::thread> threads_;
  int idle_ = 0, spawning_ = 0, started_ = 0, max_threads_;
  bool stop_ = false;
};

// Futures of pool tasks do not wait in their destructors (std::async's did): a frame whose tasks refer to its locals waits for them
// through this guard, also when it unwinds (a later submit that throws).
struct WaitAll {
  std::future<int>* f; int n;
  ~WaitAll() { for (int i = 0; i < n; ++i) if (f[i].valid()) f[i].wait(); }
};

}  // namespace kg
