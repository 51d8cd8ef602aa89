// bp_hostpool.hpp -- the host threads of a context (plain C++, no HIP: tests/cpp/hostpool_stress.cpp builds it with g++ -fsanitize=thread).
//   HostWorker  one parked helper thread for a job that runs beside the calling thread
//   HostPool    a few parked helpers for the host tail's independent Horner chains (bp_host_tail.hpp)
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// One helper thread per context for host work that can run beside the calling thread (the second serial tail of a paired
// MSM): started on first use, parked on a condition variable in between (spawning a std::thread per pair cost ~40 us each).
struct HostWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, quit = false, started = false;
    bool submit(std::function<void()> f) {
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            try { th = std::thread([this] { run(); }); } catch (...) { return false; }
            started = true;
        }
        job = std::move(f);
        has_job = true;
        done = false;
        cv.notify_all();
        return true;
    }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return done; }); }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return has_job || quit; });
            if (quit) return;
            std::function<void()> f = std::move(job);
            has_job = false;
            lk.unlock();
            f();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
    ~HostWorker() {
        if (!started) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        th.join();
    }
};

// A few helper threads per context for the host tail's independent Horner chains (bp_host_tail.hpp): run(njobs, fn) executes
// fn(0) .. fn(njobs - 1) on the helpers AND the calling thread and returns when all are done.  Threads start on first use and park
// on a condition variable in between.  One run at a time per pool (a context is used by one host thread).
//
// Claims are safe across consecutive runs: a helper is counted in `active` (under the mutex) from the moment it has seen a new
// epoch until it has left drain(), and run() does not touch next / njobs / fn while any helper is active.  Without that a helper
// holding a failed claim j >= njobs of the previous run could compare it against the NEXT run's larger njobs and execute job j a
// second time -- through a dangling fn, with one decrement of `pending` too many.  A helper that registers after
// the reset is simply a participant of the new run.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    const std::function<void(int)>* fn = nullptr;
    std::atomic<int> njobs{0};
    std::atomic<int> next{0};
    int pending = 0;
    int active = 0;                                            // helpers between "saw the epoch" and "left drain()"
    uint64_t epoch = 0;
    bool quit = false;
    static constexpr int kMaxHelpers = 15;
    void ensure(int helpers) {
        if (helpers > kMaxHelpers) helpers = kMaxHelpers;
        while ((int)th.size() < helpers) {
            try { th.emplace_back([this] { worker(); }); } catch (...) { return; }     // fewer helpers: the caller's thread does the rest
        }
    }
    void drain() {                                             // claim and run jobs until none is left
        for (;;) {
            const int j = next.fetch_add(1);
            if (j >= njobs.load()) return;
            (*fn)(j);
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }
    void run(int n, const std::function<void(int)>& f, int helpers) {
        if (n <= 0) return;
        if (n == 1 || helpers <= 0) { for (int j = 0; j < n; j++) f(j); return; }
        ensure(helpers < n - 1 ? helpers : n - 1);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [this] { return active == 0; });  // stragglers of the previous run have given up their (failed) claims
            fn = &f; njobs.store(n); pending = n; epoch++;
            next.store(0);                                     // last: a helper that claims a job sees fn / njobs of THIS run
        }
        cv_job.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return pending == 0; });
        njobs.store(0);                                        // late wakers find nothing to claim
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return quit || epoch != seen; });
                if (quit) return;
                seen = epoch;
                active++;
            }
            drain();
            std::lock_guard<std::mutex> lk(mu);
            if (--active == 0) cv_done.notify_all();
        }
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_job.notify_all();
        for (auto& t : th) t.join();
    }
};
