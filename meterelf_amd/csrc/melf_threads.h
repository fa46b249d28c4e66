// A small persistent worker pool for the host halves of the JPEG path (header parsing, scan cleaning, file
// reading): creating sixteen std::threads per call cost more than the work they did (~0.7 ms per parallel loop).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <sched.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace melf {

class WorkerPool {
public:
    // max_threads: the most workers the pool may ever have; they are STARTED when set_active first asks for them (ADVICE, round 5:
    // a pool that starts them all has its sleeping surplus woken by every loop's notify_all, only to go back to sleep)
    explicit WorkerPool(int max_threads, bool start_all = false) : pid_(getpid()), max_(max_threads < 0 ? 0 : max_threads), active_(0)
    {
        workers_.reserve((size_t)max_);
        if (start_all) set_active(max_);
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
    // workers that take part in a loop (the others sleep through it): the pool is created with the most threads a device's
    // share of the cores can ever be and uses as many of them as that share is NOW (pool_of)
    int size() const { return active_.load(); }
    void set_active(int n)
    {
        n = n < 0 ? 0 : (n > max_ ? max_ : n);
        if (n > (int)started_.load() && getpid() == pid_) {
            std::lock_guard<std::mutex> one_caller(run_m_);   // not while a loop is running: workers_ must not move under it
            while ((int)workers_.size() < n) {
                const int t = (int)workers_.size();
                workers_.emplace_back([this, t]() { loop(t); });
            }
            started_.store((int)workers_.size());
        }
        active_.store(n > (int)started_.load() ? (int)started_.load() : n);
    }

    // fn(i) for i in [0, n): indices are handed out in small blocks; the caller works too and returns when all are done.
    // fn must not throw: an exception on a worker would terminate the process, one on the caller would unwind past
    // workers that still dereference fn_.  Callers that can fail (allocation, I/O) catch inside fn and record a status.
    void run(int n, const std::function<void(int)>& fn)
    {
        if (n <= 0) return;
        // sequential when the loop is short, and in a forked child (the worker threads exist in the parent only)
        const int act = active_.load();
        if (act <= 0 || n < 32 || getpid() != pid_) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::lock_guard<std::mutex> one_caller(run_m_);  // contexts on different host threads share the pool
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            pending_ = act;
            run_active_ = act;
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
    void loop(int idx)
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                if (idx >= run_active_) continue;   // not this loop's worker
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
    int n_ = 0, pending_ = 0, run_active_ = 0;
    const int max_;
    std::atomic<int> active_;
    std::atomic<int> started_{0};
    std::atomic<int> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// One pool PER DEVICE (round 4: get_meter_values drives every GPU of a node from one process, one host thread per device;
// with one pool per process the devices' parallel loops would queue behind each other), created on first use and
// deliberately never destroyed: a destructor that joins the workers would hang exit() in a forked child (the threads exist
// in the parent only), and at process exit the threads go away with the process anyway.  The entry points set the calling
// thread's device (pool_use_device) before their first parallel loop.  The cores this process may run on (its affinity
// mask, not the machine's core count) are shared out over the devices the process has LIVE contexts on (pool_note_device /
// pool_forget_device, called by melf_ctx_create / melf_ctx_destroy) -- not over the visible devices: a one-GPU rank on an eight-GPU node keeps its whole share (round 4
// divided by the visible devices: 2 I/O threads instead of 12 there).  The share is re-read at every loop, so a process
// that opens its second device later narrows the first device's loops from then on.
inline thread_local int tl_pool_device = 0;
inline std::atomic<int> g_pool_ctx_count[64];   // contexts alive per device (melf_ctx_create / melf_ctx_destroy)
inline void pool_use_device(int device) { tl_pool_device = device >= 0 ? device & 63 : 0; }
inline void pool_note_device(int device) { g_pool_ctx_count[device & 63].fetch_add(1); }
inline void pool_forget_device(int device) { if (g_pool_ctx_count[device & 63].load() > 0) g_pool_ctx_count[device & 63].fetch_sub(1); }
inline int pool_devices()
{
    int n = 0;
    for (int d = 0; d < 64; ++d) n += g_pool_ctx_count[d].load() > 0 ? 1 : 0;
    return n < 1 ? 1 : n;
}
inline unsigned pool_cores()
{
    // the cores this process may USE: its affinity mask, cut down to the container's CPU quota (cgroup v2 cpu.max, v1
    // cfs_quota_us / cfs_period_us) -- the pool's boxes show 256 cores and allow 16; threads beyond the quota get the whole
    // process throttled for the rest of the scheduler period
    static const unsigned n = []() -> unsigned {
        unsigned cores = std::max<unsigned>(std::thread::hardware_concurrency(), 2u);
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) cores = (unsigned)CPU_COUNT(&set);
        long long quota = -1, period = 0;
        if (FILE* fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(fp, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
            fclose(fp);
        } else {
            if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
            if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fq, "%lld", &period) != 1) period = 0; fclose(fq); }
        }
        if (quota > 0 && period > 0) cores = std::min<unsigned>(cores, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        return std::max(cores, 1u);
    }();
    return n;
}
inline WorkerPool& pool_of(WorkerPool** pools, std::mutex& m, const char* env, unsigned lo, unsigned hi)
{
    std::lock_guard<std::mutex> lk(m);
    WorkerPool*& p = pools[tl_pool_device];
    const bool fixed = getenv(env) != nullptr;
    if (!p) p = fixed ? new WorkerPool(std::max(0, atoi(getenv(env)) - 1), true) : new WorkerPool((int)hi);
    if (!fixed) {
        const unsigned share = std::max<unsigned>(pool_cores(), 2u) / (unsigned)pool_devices();
        p->set_active((int)std::min<unsigned>(std::max<unsigned>(share, lo + 1u) - 1u, hi));
    }
    return *p;
}
inline WorkerPool& host_pool()
{
    static WorkerPool* pools[64] = {};
    static std::mutex m;
    return pool_of(pools, m, "MELF_HOST_THREADS", 3, 15);   // the device's share of the cores, 4 .. 16 threads with the caller
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
    return pool_of(pools, m, "MELF_IO_THREADS", 1, 11);
}

}  // namespace melf
