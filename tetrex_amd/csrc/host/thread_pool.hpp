// Host side: a small fork-join thread pool (product code).
// run(count, fn) calls fn(i, thread) for every i < count on the calling thread and the workers;
// indexes are handed out through one atomic counter, so uneven tasks balance themselves.  One job
// at a time, no nesting.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace tetrex {

class ThreadPool {
  public:
    using Body = std::function<void(size_t index, int thread)>;

    explicit ThreadPool(int threads) {  // the caller is thread 0; threads - 1 workers are started
        for (int t = 1; t < threads; ++t)
            workers_.emplace_back([this, t]() {
                size_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m_);
                    wake_.wait(lk, [&] { return stop_ || generation_ != seen; });
                    if (stop_) return;
                    seen = generation_;
                    const Body* fn = job_;
                    const size_t count = count_;
                    lk.unlock();
                    for (size_t i; (i = next_.fetch_add(1)) < count;) (*fn)(i, t);
                    lk.lock();
                    if (--running_ == 0) done_.notify_one();
                }
            });
    }
    ~ThreadPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& w : workers_) w.join();
    }
    ThreadPool(const ThreadPool&) = delete;
    ThreadPool& operator=(const ThreadPool&) = delete;

    int threads() const { return (int)workers_.size() + 1; }

    void run(size_t count, const Body& fn) {
        if (workers_.empty() || count < 2) {
            for (size_t i = 0; i < count; ++i) fn(i, 0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            count_ = count;
            next_.store(0);
            running_ = workers_.size();
            ++generation_;
        }
        wake_.notify_all();
        for (size_t i; (i = next_.fetch_add(1)) < count;) fn(i, 0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return running_ == 0; });
    }

  private:
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const Body* job_ = nullptr;
    size_t count_ = 0, generation_ = 0, running_ = 0;
    std::atomic<size_t> next_{0};
    bool stop_ = false;
};

}  // namespace tetrex
