// csrc/rwlock.hpp on the CPU: readers that overlap back to back must not starve a writer (the reference's tokio RwLock is
// fair), readers still run side by side, and the lock is a mutual exclusion for writers.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <ctime>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

#include "../../vectorlite_amd/csrc/rwlock.hpp"

int main()
{
    vl::RwLock mu;
    std::atomic<bool> stop{false};
    std::atomic<int> readers_inside{0}, max_readers{0}, writers_inside{0};
    std::atomic<long> reads{0};
    long shared_value = 0;
    bool broken = false;
    std::vector<std::thread> th;
    for (int t = 0; t < 6; ++t)
        th.emplace_back([&]() {
            while (!stop.load()) {
                std::shared_lock<vl::RwLock> lk(mu);
                const int r = readers_inside.fetch_add(1) + 1;
                int m = max_readers.load();
                while (r > m && !max_readers.compare_exchange_weak(m, r)) {
                }
                if (writers_inside.load() != 0) broken = true;
                const long v = shared_value;
                std::this_thread::sleep_for(std::chrono::microseconds(200));  // overlapping readers: the lock is never free
                if (v != shared_value) broken = true;
                readers_inside.fetch_sub(1);
                reads.fetch_add(1);
            }
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    const auto t0 = std::chrono::steady_clock::now();
    double worst_ms = 0.0;
    for (int i = 0; i < 200; ++i) {
        const auto a = std::chrono::steady_clock::now();
        std::unique_lock<vl::RwLock> lk(mu);
        worst_ms = std::max(worst_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
        if (writers_inside.fetch_add(1) != 0 || readers_inside.load() != 0) broken = true;
        shared_value += 1;
        writers_inside.fetch_sub(1);
    }
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    for (auto& x : th) x.join();
    std::printf("200 writes beside 6 overlapping readers: %.1f ms in all, worst wait %.2f ms; %ld reads, up to %d readers inside at once\n",
                total_ms, worst_ms, reads.load(), max_readers.load());
    if (broken || shared_value != 200) {
        std::printf("FAILED: exclusion broken\n");
        return 1;
    }
    if (max_readers.load() < 2) {
        std::printf("FAILED: readers never ran side by side\n");
        return 1;
    }
    // Readers that stand back behind a waiting writer must SLEEP, not spin (advisor, round 3): a writer waits 300 ms behind
    // one long reader while 6 more readers arrive; together they may use a few ms of CPU, not 6 x 300 ms.
    {
        vl::RwLock mu2;
        std::atomic<bool> long_reader_in{false};
        std::thread long_reader([&]() {
            std::shared_lock<vl::RwLock> lk(mu2);
            long_reader_in.store(true);
            std::this_thread::sleep_for(std::chrono::milliseconds(300));
        });
        while (!long_reader_in.load()) std::this_thread::yield();
        std::thread writer([&]() { std::unique_lock<vl::RwLock> lk(mu2); });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));  // the writer is queued by now
        const std::clock_t c0 = std::clock();                       // process CPU time, all threads
        std::vector<std::thread> late;
        for (int t = 0; t < 6; ++t) late.emplace_back([&]() { std::shared_lock<vl::RwLock> lk(mu2); });
        for (auto& x : late) x.join();
        const double cpu_ms = 1000.0 * (double)(std::clock() - c0) / CLOCKS_PER_SEC;
        writer.join();
        long_reader.join();
        std::printf("6 readers behind a queued writer for ~280 ms used %.1f ms of CPU\n", cpu_ms);
        if (cpu_ms > 200.0) {
            std::printf("FAILED: waiting readers burn CPU (they should block)\n");
            return 1;
        }
    }
    if (total_ms > 2000.0) {  // 200 x (one reader's 0.2 ms to drain + the write): a starved writer takes far longer
        std::printf("FAILED: the writer was starved\n");
        return 1;
    }
    return 0;
}
