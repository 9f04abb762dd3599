// host_pool.h -- fork-join over persistent host threads: the K Fiat-Shamir transcripts of glp_prove_batch between two device stages,
// and the K transcripts + vanishing-polynomial checks of glp_verify_batch.  One pool per context, made on first use.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sched.h>
#include <thread>
#include "common.h"

namespace glp {

// fork-join over [0, count) on persistent host threads (the transcripts of a batch between two device stages)
class HostPool {
public:
    explicit HostPool(unsigned nthreads) {
        for (unsigned t = 1; t < nthreads; t++) workers.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto &w : workers) w.join();
    }
    void run(size_t count, const std::function<void(size_t)> &fn) {
        if (workers.empty() || count < 2) { for (size_t i = 0; i < count; i++) fn(i); return; }
        {
            std::lock_guard<std::mutex> l(m);
            job = &fn; total = count; next = 0; pending = workers.size(); gen++;
        }
        cv.notify_all();
        drain();
        std::unique_lock<std::mutex> l(m);
        done_cv.wait(l, [this] { return pending == 0; });
        job = nullptr;
    }
private:
    void drain() {
        for (;;) {
            size_t i;
            { std::lock_guard<std::mutex> l(m); if (!job || next >= total) return; i = next++; }
            (*job)(i);
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            drain();
            { std::lock_guard<std::mutex> l(m); if (--pending == 0) done_cv.notify_all(); }
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv, done_cv;
    const std::function<void(size_t)> *job = nullptr;
    size_t total = 0, next = 0, pending = 0;
    unsigned long long gen = 0;
    bool stop = false;
};

// Host threads for the transcripts of a batch: GLP_HOST_THREADS if set, else the cores this PROCESS may use -- the affinity
// mask capped by the cgroup CPU quota (a GPU box hands out one GPU's share of a large host; hardware_concurrency() reports
// the whole machine and every context has its own pool) -- at most 32.
inline unsigned usable_cores() {
    unsigned t = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) t = std::min(t ? t : (unsigned)k, (unsigned)k); }
    auto read_two = [](const char *path, long long &a, long long &b) -> int {
        FILE *f = fopen(path, "r");
        if (!f) return 0;
        char s0[32] = {0};
        const int got = fscanf(f, "%31s %lld", s0, &b);
        fclose(f);
        if (got < 1 || strcmp(s0, "max") == 0) return -1;
        a = atoll(s0);
        return got;
    };
    long long quota = 0, period = 0;
    if (read_two("/sys/fs/cgroup/cpu.max", quota, period) == 2 && quota > 0 && period > 0) {          // cgroup v2: "<quota> <period>" or "max <period>"
        t = std::min<unsigned>(t, (unsigned)std::max<long long>(1, (quota + period / 2) / period));
    } else {
        long long q = 0, per = 0, dummy = 0;                                                          // cgroup v1
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, dummy) >= 1 && read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", per, dummy) >= 1 && q > 0 && per > 0)
            t = std::min<unsigned>(t, (unsigned)std::max<long long>(1, (q + per / 2) / per));
    }
    return std::max(1u, t);
}
inline unsigned host_threads() {
    unsigned t = usable_cores();
    if (const char *e = getenv("GLP_HOST_THREADS")) { const int v = atoi(e); if (v > 0) t = (unsigned)v; }
    return std::max(1u, std::min(t, 32u));
}

inline HostPool &ctx_host_pool(glp_ctx *c) {
    if (!c->host_pool) { c->host_pool = new HostPool(host_threads()); c->host_pool_free = [](void *q) { delete static_cast<HostPool *>(q); }; }
    return *static_cast<HostPool *>(c->host_pool);       // persistent: thread start-up costs more than a small batch
}
}  // namespace glp
