// tile_pool.hpp — the host threads of armon_hip_mgpu_cycle (multi_gpu.hip): one persistent thread per local tile.
//
// A cycle is a LIST OF STEPS; step s runs for tile k on thread k, and a barrier separates two steps, so that whatever a step
// needs from ANOTHER tile (its pack event, its send buffer) was produced by a step that is complete for every tile. A step
// that fails (non-zero status) stops the remaining steps of every tile — each thread still walks through every barrier — and
// the first failing tile's status and message are handed back; the pool is usable again afterwards.
// Pure C++ (no HIP): tests/native/tile_pool_test.cpp runs it under ThreadSanitizer on the CPU.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct tile_pool {
    using step_fn = std::function<int(size_t)>;

    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;
    const std::vector<step_fn>* steps = nullptr;
    std::vector<int> rc;                         // per tile
    std::vector<std::string> msg;
    std::function<std::string()> last_error;     // the message behind a non-zero status, read on the thread that got it
    std::atomic<int> failed{0};
    std::atomic<int> bar_count{0};
    std::atomic<int> bar_sense{0};
    int n = 0;

    // sense-reversing barrier: spin briefly (steps are tens of microseconds), then yield
    void barrier(int& local_sense)
    {
        local_sense ^= 1;
        if (bar_count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            bar_count.store(0, std::memory_order_relaxed);
            bar_sense.store(local_sense, std::memory_order_release);
        } else {
            int spins = 0;
            while (bar_sense.load(std::memory_order_acquire) != local_sense)
                if (++spins > 2000) std::this_thread::yield();
        }
    }

    void run_tile(size_t k, int& local_sense)
    {
        for (const auto& step : *steps) {
            if (!failed.load(std::memory_order_acquire)) {
                const int r = step(k);
                if (r != 0) {
                    rc[k] = r;
                    msg[k] = last_error ? last_error() : std::string();
                    failed.store(1, std::memory_order_release);
                }
            }
            barrier(local_sense);
        }
    }

    void worker(size_t k)
    {
        uint64_t seen = 0;
        int local_sense = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            run_tile(k, local_sense);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }

    tile_pool(int n_tiles, std::function<std::string()> last_error_) : rc(n_tiles, 0), msg(n_tiles), last_error(std::move(last_error_)), n(n_tiles)
    {
        for (int k = 0; k < n_tiles; k++) threads.emplace_back([this, k] { worker((size_t)k); });
    }

    ~tile_pool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& t : threads) t.join();
    }

    // run every step for every tile; 0, or the status of the first (lowest-numbered) failing tile with its number and message
    int run(const std::vector<step_fn>& s, int* failed_tile, std::string* message)
    {
        std::unique_lock<std::mutex> lk(m);
        steps = &s;
        failed.store(0);
        for (int k = 0; k < n; k++) rc[k] = 0;
        pending = n;
        generation++;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
        for (int k = 0; k < n; k++)
            if (rc[k] != 0) {
                if (failed_tile) *failed_tile = k;
                if (message) *message = msg[k];
                return rc[k];
            }
        return 0;
    }
};
