// A small persistent worker pool for the host halves of the JPEG path (header parsing, scan cleaning, file
// reading): creating sixteen std::threads per call cost more than the work they did (~0.7 ms per parallel loop).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

namespace melf {

class WorkerPool {
public:
    explicit WorkerPool(int nthreads) : pid_(getpid())
    {
        for (int t = 0; t < nthreads; ++t) workers_.emplace_back([this]() { loop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return (int)workers_.size(); }

    // fn(i) for i in [0, n): indices are handed out in small blocks; the caller works too and returns when all are done.
    // fn must not throw: an exception on a worker would terminate the process, one on the caller would unwind past
    // workers that still dereference fn_.  Callers that can fail (allocation, I/O) catch inside fn and record a status.
    void run(int n, const std::function<void(int)>& fn)
    {
        if (n <= 0) return;
        // sequential when the loop is short, and in a forked child (the worker threads exist in the parent only)
        if (workers_.empty() || n < 32 || getpid() != pid_) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::lock_guard<std::mutex> one_caller(run_m_);  // contexts on different host threads share the pool
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            pending_ = (int)workers_.size();
            ++generation_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void drain()
    {
        const int block = 8;
        for (;;) {
            const int i0 = next_.fetch_add(block);
            if (i0 >= n_) break;
            const int i1 = i0 + block < n_ ? i0 + block : n_;
            for (int i = i0; i < i1; ++i) (*fn_)(i);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_m_;
    const pid_t pid_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    std::atomic<int> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// One pool PER DEVICE (round 4: get_meter_values drives every GPU of a node from one process, one host thread per device;
// with one pool per process the devices' parallel loops would queue behind each other), created on first use and
// deliberately never destroyed: a destructor that joins the workers would hang exit() in a forked child (the threads exist
// in the parent only), and at process exit the threads go away with the process anyway.  The entry points set the calling
// thread's device (pool_use_device) before their first parallel loop; the host's cores are shared out over the visible
// devices (pool_set_devices, called by melf_ctx_create).
inline thread_local int tl_pool_device = 0;
inline std::atomic<int> g_pool_devices{1};
inline void pool_use_device(int device) { tl_pool_device = device >= 0 ? device & 63 : 0; }
inline void pool_set_devices(int ndev)
{
    int cur = g_pool_devices.load();
    while (ndev > cur && !g_pool_devices.compare_exchange_weak(cur, ndev)) {}
}
inline WorkerPool& pool_of(WorkerPool** pools, std::mutex& m, const char* env, unsigned divisor, unsigned lo, unsigned hi)
{
    std::lock_guard<std::mutex> lk(m);
    WorkerPool*& p = pools[tl_pool_device];
    if (!p) {
        int n;
        if (getenv(env)) {
            n = std::max(0, atoi(getenv(env)) - 1);
        } else {
            const unsigned share = std::max<unsigned>(std::thread::hardware_concurrency(), 2u) / (divisor * (unsigned)std::max(1, g_pool_devices.load()));
            n = (int)std::min<unsigned>(std::max<unsigned>(share, lo + 1u) - 1u, hi);
        }
        p = new WorkerPool(n);
    }
    return *p;
}
inline WorkerPool& host_pool()
{
    static WorkerPool* pools[64] = {};
    static std::mutex m;
    return pool_of(pools, m, "MELF_HOST_THREADS", 1, 3, 15);   // the device's share of the cores, 4 .. 16 threads with the caller
}

// A second pool for the read stage of melf_jpeg_process_files: with several calls in flight the next call's files are read
// WHILE the current one is prepared and decoded, and one pool (one parallel loop at a time) made each wait for the other
// (read stage 0.9 -> 1.5 ms, the decode stage's host part 0.8 -> 2.2 ms).  Since round 4 the read stage is all the per-file
// host work there is (read() into the pinned arena, header parse, Huffman decode data, on the thread that read the file;
// the decode stage runs no parallel loop for such a call), so this pool gets the device's share of the cores: up to 12
// threads with the caller (6 / 8 / 12 / 16 threads: 0.35-0.40 / 0.48-0.50 / 0.54-0.55 / 0.50-0.55 M files/s on a 16-core box).
inline WorkerPool& io_pool()
{
    static WorkerPool* pools[64] = {};
    static std::mutex m;
    return pool_of(pools, m, "MELF_IO_THREADS", 1, 1, 11);
}

}  // namespace melf
