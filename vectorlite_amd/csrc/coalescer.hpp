// coalescer.hpp -- leader/follower grouping of concurrent single-query calls into shared device passes.
// The reference serves searches concurrently under RwLock::read (src/client.rs:398, src/server.rs:269), one
// full pass per caller.  Here a caller that finds no pass in flight becomes the leader, takes every queued
// request compatible with its own -- whatever piled up while the previous pass was on the GPU -- and answers
// them with ONE batched pass; the others sleep until their request is marked done.  No background thread;
// with window_us = 0 a lone caller pays nothing.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <vector>

namespace vl {

template <typename Req>  // Req needs a `bool done` member
class Coalescer {
public:
    void configure(int max_batch, int window_us, int hard_cap)
    {
        window_us_.store(window_us < 0 ? 0 : window_us);
        max_.store(max_batch < 0 ? 0 : (max_batch > hard_cap ? hard_cap : max_batch));
    }
    bool enabled() const { return max_.load(std::memory_order_relaxed) > 1; }
    void stats(uint64_t* batches, uint64_t* queries) const
    {
        if (batches) *batches = batches_.load();
        if (queries) *queries = queries_.load();
    }

    // Blocks until `r` has been answered.  same(a, b): may a and b share a pass?  exec(batch): answer every
    // request of the batch (batch[0] is the leader's own); it runs on the leader's thread, outside the lock.
    template <typename Same, typename Exec>
    void run(Req& r, Same same, Exec exec)
    {
        const size_t max_batch = (size_t)(max_.load() < 2 ? 2 : max_.load());
        std::unique_lock<std::mutex> lk(mu_);
        q_.push_back(&r);
        cv_.notify_all();  // a leader waiting in its window counts arrivals
        while (!r.done) {
            if (leader_) {
                cv_.wait(lk);
                continue;
            }
            leader_ = true;
            const int window = window_us_.load();
            if (window > 0 && q_.size() < max_batch)
                cv_.wait_for(lk, std::chrono::microseconds(window), [&] { return q_.size() >= max_batch; });
            std::vector<Req*> batch;
            batch.push_back(&r);
            for (auto it = q_.begin(); it != q_.end();) {
                Req* o = *it;
                if (o == &r) {
                    it = q_.erase(it);
                } else if (batch.size() < max_batch && same(r, *o)) {
                    batch.push_back(o);
                    it = q_.erase(it);
                } else {
                    ++it;
                }
            }
            // Whatever exec does -- return, or throw -- the batch is released: followers already taken off the
            // queue are marked done and the leader slot is freed, so nobody waits for a pass that will never come.
            // exec's contract is to answer every request (an rc per request) and not to throw; if it throws anyway
            // the requests it left unanswered keep their initial rc, which callers initialise to an error.
            struct Release {
                Coalescer* c;
                std::unique_lock<std::mutex>& lk;
                std::vector<Req*>& batch;
                ~Release()
                {
                    if (!lk.owns_lock()) lk.lock();
                    for (Req* o : batch) o->done = true;
                    c->leader_ = false;
                    c->cv_.notify_all();
                }
            };
            lk.unlock();
            batches_.fetch_add(1);
            queries_.fetch_add(batch.size());
            {
                Release rel{this, lk, batch};
                try {
                    exec(batch);
                } catch (...) {
                    // swallowed: no exception may cross the C ABI from a follower's pass either
                }
            }
        }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Req*> q_;
    bool leader_ = false;
    std::atomic<int> max_{0}, window_us_{0};
    std::atomic<uint64_t> batches_{0}, queries_{0};
};

}  // namespace vl
