#include "ss_pool.h"

#include <atomic>
#include <condition_variable>
#include <exception>
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
    pid_t owner_ = getpid();  // a fork()ed child inherits this object but none of its threads
    std::exception_ptr error_;  // the first exception an item threw (under m_); rethrown on the calling thread

    void drain()
    {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) return;
            try {
                (*f_)(i);
            } catch (...) {  // e.g. bad_alloc in the host reader: fail the call, not the process
                std::lock_guard<std::mutex> lk(m_);
                if (!error_) error_ = std::current_exception();
                next_.store(n_, std::memory_order_relaxed);  // the remaining items are not started
            }
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            work_.wait(lk, [&] { return gen_ != seen && helpers_in_ < helpers_wanted_; });
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
    // No destructor runs: the pool is allocated once and never freed (parallel_for), so neither a normal exit nor the
    // exit of a fork()ed child -- which inherits the std::thread handles but none of the threads -- touches them.
    ~Pool() = delete;
    void run(size_t n, const std::function<void(size_t)> &f, size_t max_threads)
    {
        const size_t helpers = std::min(workers_.size(), std::min(n, max_threads) - 1);
        if (n <= 1 || max_threads <= 1 || helpers == 0 || getpid() != owner_) { for (size_t i = 0; i < n; i++) f(i); return; }
        std::lock_guard<std::mutex> one(callers_);
        {
            std::lock_guard<std::mutex> lk(m_);
            f_ = &f; n_ = n;
            error_ = nullptr;
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
        if (error_) {
            std::exception_ptr e = error_;
            error_ = nullptr;
            lk.unlock();
            std::rethrow_exception(e);
        }
    }
};

}  // namespace

void parallel_for(size_t n, const std::function<void(size_t)> &f, size_t max_threads)
{
    static Pool *const pool = new Pool();  // constructed on first use, never destroyed: its threads end with the process
    if (max_threads == 0) max_threads = 1;
    pool->run(n, f, max_threads);
}

}  // namespace ss
