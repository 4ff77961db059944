// coalescer.hpp -- leader/follower grouping of concurrent single-query calls into shared device passes.
// The reference serves searches concurrently under RwLock::read (src/client.rs:398, src/server.rs:269), one
// full pass per caller.  Here a caller that finds no pass in flight becomes the leader, takes every queued
// request compatible with its own -- whatever piled up while the previous pass was on the GPU -- and answers
// them with ONE batched pass; the others sleep until their request is marked done.  No background thread;
// with window_us = 0 a lone caller pays nothing.
//
// Adaptive gather (round 4, on by default; VL_COALESCE_ADAPTIVE=0 or set_adaptive(false) turns it off).  With window 0
// and N callers in a closed loop the passes fall into a 1 / N-1 rhythm: the first caller back from a pass of N-1 leads
// at once, alone, while the other N-1 are still on their way back (16 threads on the 10 M x 384 index: a lone pass of
// 2.23 ms, then a pass of 15 in 1.48 ms = 4.3 k QPS where passes of 16 would give 10 k).  So every pass leaves an
// estimate of the callers in the loop -- the requests it answered plus the compatible ones that queued up behind it --
// and a leader that finds fewer compatible requests queued than the larger of the last two estimates waits for them
// (they are all either queued or on their way back, since only one pass is in flight), but never longer than a quarter
// of the recent pass time, and never more than 400 us.  A lone caller's estimates are (1, 1): target 1, no wait.
// And only while waiting pays: requests must arrive faster during a gather than going at once would answer them (run()).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
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
    void set_adaptive(bool on) { adaptive_.store(on); }
    // passes whose leader waited for its peers, and the microseconds spent waiting
    void gather_stats(uint64_t* waits, uint64_t* waited_us) const
    {
        if (waits) *waits = waits_.load();
        if (waited_us) *waited_us = waited_us_.load();
    }
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
        // a leader waiting in its window / gather counts arrivals: it alone listens on cv_arrive_ (waking every sleeping
        // follower for every arrival was a T^2 storm of wake-ups on one mutex: 256 callers took 35 ms per pass)
        if (gathering_ && q_.size() >= gather_target_) cv_arrive_.notify_one();  // not before: the leader sleeps until its target is there (or its time is up)
        while (!r.done) {
            if (leader_) {
                cv_.wait(lk);
                continue;
            }
            leader_ = true;
            const int window = window_us_.load();
            gathering_ = true;
            gather_target_ = max_batch;
            if (window > 0 && q_.size() < max_batch)
                cv_arrive_.wait_for(lk, std::chrono::microseconds(window), [&] { return q_.size() >= max_batch; });
            else if (window == 0 && adaptive_.load(std::memory_order_relaxed)) {
                const size_t peers = std::min(max_batch, std::max(hist_[0], hist_[1]));
                const int64_t cap = std::min<int64_t>(pass_us_ / 4, 400);
                auto compatible = [&] {
                    size_t n = 0;
                    for (Req* o : q_) n += (o == &r || same(r, *o)) ? 1 : 0;
                    return n;
                };
                auto gathered = [&] { return q_.size() >= peers && compatible() >= peers; };
                const size_t n0 = compatible();
                // Is waiting worth it?  Going now answers n0 requests per pass time; waiting pays while requests arrive FASTER
                // than that (the rate the last gathers saw).  Few callers on an idle queue: n0 = 1, peers arrive within
                // microseconds -> wait.  Many more callers than cores: a long queue is there already and the peers are stuck
                // in the OS run queue -> go (256 threads on an HNSW handle: 123 k QPS waiting, 223 k not).  Every 16th
                // such pass waits anyway and measures again.
                const bool worth = arr_rate_ < 0.0 || arr_rate_ * (double)pass_us_ > (double)n0 || (++probe_ % 16u) == 0u;
                if (cap > 0 && n0 < peers && worth) {
                    // the more of the peers are here already, the less there is to wait for: the cap shrinks with the share missing
                    const int64_t cap_now = std::max<int64_t>(1, cap * (int64_t)(peers - n0) / (int64_t)peers);
                    const auto t0 = std::chrono::steady_clock::now();
                    gather_target_ = peers;
                    cv_arrive_.wait_for(lk, std::chrono::microseconds(cap_now), gathered);
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    const double rate = (double)(compatible() - n0) / (us > 1.0 ? us : 1.0);
                    arr_rate_ = arr_rate_ < 0.0 ? rate : 0.5 * (arr_rate_ + rate);
                    waits_.fetch_add(1);
                    waited_us_.fetch_add((uint64_t)us);
                }
            }
            gathering_ = false;
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
                Same& same;
                std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                ~Release()
                {
                    const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                    if (!lk.owns_lock()) lk.lock();
                    size_t behind = 0;  // compatible requests that arrived while this pass ran: callers of the same loop
                    for (Req* o : c->q_) behind += same(*batch[0], *o) ? 1 : 0;
                    for (Req* o : batch) o->done = true;
                    c->hist_[1] = c->hist_[0];
                    c->hist_[0] = batch.size() + behind;
                    c->pass_us_ = c->pass_us_ == 0 ? us : (3 * c->pass_us_ + us) / 4;
                    c->leader_ = false;
                    c->cv_.notify_all();
                }
            };
            lk.unlock();
            batches_.fetch_add(1);
            queries_.fetch_add(batch.size());
            {
                Release rel{this, lk, batch, same};
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
    std::condition_variable cv_;         // followers: "my request is done" / "the leader slot is free"
    std::condition_variable cv_arrive_;  // the gathering leader: "somebody queued up"
    std::deque<Req*> q_;
    bool leader_ = false;
    bool gathering_ = false;             // the leader is waiting for arrivals (under mu_)
    size_t gather_target_ = 0;           // ... for this many queued requests (under mu_)
    size_t hist_[2] = {0, 0};  // callers in the loop as the last two passes saw them: answered + queued behind (under mu_)
    int64_t pass_us_ = 0;      // recent pass time, exponentially averaged (under mu_)
    double arr_rate_ = -1.0;   // requests per microsecond that arrived while the last gathers waited; < 0: not measured yet (under mu_)
    uint32_t probe_ = 0;       // passes that skipped the gather since the last measurement (under mu_)
    std::atomic<bool> adaptive_{adaptive_default()};
    std::atomic<int> max_{0}, window_us_{0};
    std::atomic<uint64_t> batches_{0}, queries_{0}, waits_{0}, waited_us_{0};

    static bool adaptive_default()
    {
        const char* e = getenv("VL_COALESCE_ADAPTIVE");
        return !(e && e[0] == '0');
    }
};

}  // namespace vl
