// kmx_pool.hpp -- the host-side task pool of the kmx driver: a fixed set of threads that run queued closures
// (the role of the reference's TaskPool, include/kmtricks/task_pool.hpp, for the host work around the GPU:
// parsing reads, reading count files, compressing and writing outputs), plus a bounded hand-over queue.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <vector>

namespace kmxio {

class Pool {
 public:
  explicit Pool(unsigned n) { for (unsigned i = 0; i < (n ? n : 1); i++) th_.emplace_back([this]() { run(); }); }
  ~Pool() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; } cv_.notify_all(); for (auto& t : th_) t.join(); }
  template <typename F> std::future<void> submit(F&& f) {
    auto task = std::make_shared<std::packaged_task<void()>>(std::forward<F>(f));
    std::future<void> fut = task->get_future();
    { std::lock_guard<std::mutex> lk(m_); q_.emplace_back([task]() { (*task)(); }); }
    cv_.notify_one();
    return fut;
  }
  // run fn(i) for i in [0, n) on the pool and wait; the first exception is rethrown
  template <typename F> void for_each(size_t n, F&& fn) {
    std::vector<std::future<void>> fs; fs.reserve(n);
    for (size_t i = 0; i < n; i++) fs.push_back(submit([&fn, i]() { fn(i); }));
    std::exception_ptr err;
    for (auto& f : fs) { try { f.get(); } catch (...) { if (!err) err = std::current_exception(); } }
    if (err) std::rethrow_exception(err);
  }
  unsigned size() const { return (unsigned)th_.size(); }
 private:
  void run() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [this]() { return stop_ || !q_.empty(); }); if (q_.empty()) return; f = std::move(q_.front()); q_.pop_front(); }
      f();
    }
  }
  std::vector<std::thread> th_; std::deque<std::function<void()>> q_; std::mutex m_; std::condition_variable cv_; bool stop_ = false;
};

// bounded multi-producer / multi-consumer queue; close() wakes the consumers once the producers are done
template <typename T> class Channel {
 public:
  explicit Channel(size_t cap) : cap_(cap ? cap : 1) {}
  void push(T&& v) { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [this]() { return q_.size() < cap_ || closed_; }); q_.push_back(std::move(v)); cv_.notify_all(); }
  bool pop(T& v) {
    std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [this]() { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return true;
  }
  bool try_pop(T& v) {      // what is there now, without waiting
    std::lock_guard<std::mutex> lk(m_);
    if (q_.empty()) return false;
    v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return true;
  }
  void close() { { std::lock_guard<std::mutex> lk(m_); closed_ = true; } cv_.notify_all(); }
 private:
  size_t cap_; std::deque<T> q_; std::mutex m_; std::condition_variable cv_; bool closed_ = false;
};

}  // namespace kmxio
