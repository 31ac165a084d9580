#include "ss_pool.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace ss {
namespace {

class Pool {
    std::vector<std::thread> workers_;
    std::mutex m_;                 // protects the job fields below
    std::mutex callers_;           // one parallel_for at a time
    std::condition_variable work_, done_;
    const std::function<void(size_t)> *f_ = nullptr;
    size_t n_ = 0, helpers_wanted_ = 0, helpers_in_ = 0, helpers_out_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
    pid_t owner_ = getpid();  // a fork()ed child inherits this object but none of its threads

    void drain()
    {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) return;
            (*f_)(i);
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            work_.wait(lk, [&] { return stop_ || (gen_ != seen && helpers_in_ < helpers_wanted_); });
            if (stop_) return;
            seen = gen_;
            helpers_in_++;
            lk.unlock();
            drain();
            lk.lock();
            if (++helpers_out_ == helpers_wanted_) done_.notify_all();
        }
    }

public:
    Pool()
    {
        const unsigned n = effective_cpus();
        for (unsigned i = 1; i < n; i++) workers_.emplace_back([this] { worker(); });  // the caller is the n-th
    }
    ~Pool()
    {
        if (getpid() != owner_) { for (auto &t : workers_) t.detach(); return; }  // (the child of a fork: nothing to join)
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        work_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void run(size_t n, const std::function<void(size_t)> &f, size_t max_threads)
    {
        const size_t helpers = std::min(workers_.size(), std::min(n, max_threads) - 1);
        if (n <= 1 || max_threads <= 1 || helpers == 0 || getpid() != owner_) { for (size_t i = 0; i < n; i++) f(i); return; }
        std::lock_guard<std::mutex> one(callers_);
        {
            std::lock_guard<std::mutex> lk(m_);
            f_ = &f; n_ = n;
            next_.store(0, std::memory_order_relaxed);
            helpers_wanted_ = helpers; helpers_in_ = helpers_out_ = 0;
            gen_++;
        }
        work_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        // every helper that was asked for must have come and gone before the job's fields may change again
        done_.wait(lk, [&] { return helpers_out_ == helpers_wanted_; });
        f_ = nullptr;
    }
};

}  // namespace

void parallel_for(size_t n, const std::function<void(size_t)> &f, size_t max_threads)
{
    static Pool pool;  // constructed on first use; its threads end with the process
    if (max_threads == 0) max_threads = 1;
    pool.run(n, f, max_threads);
}

}  // namespace ss
