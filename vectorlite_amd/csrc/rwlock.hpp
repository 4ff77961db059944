// rwlock.hpp -- the reader/writer lock of an index handle: search shared, add / delete exclusive.
//
// The reference guards its index with tokio::sync::RwLock (Collection.index, src/client.rs:245; read() for searches
// :398, write() for add / delete :333,383), which is FAIR: a waiting writer is not overtaken by readers that arrive
// after it.  std::shared_mutex on glibc prefers readers -- four threads searching back to back kept an add() of a
// three-part handle waiting 14 ms on average (tools/mutate_while_searching.py) and nothing bounds that wait.  This
// wrapper adds the writer's priority: readers that arrive while a writer waits stand back until it has had its turn.
// Inside the reference's own process the caller's RwLock already orders readers and writers; this matters to hosts that
// call the C ABI from their own threads without one.
//
// Readers that stand back BLOCK on a condition variable (round 4; they used to spin on yield(): a writer queued behind a
// 100 ms batch made every arriving reader thread burn a core until it drained -- advisor, round 3).
//
// A thread must not take the shared side twice (a reader that re-enters behind a waiting writer would wait for itself):
// the classes that use this lock hand their already-locked paths a *_locked variant instead.
#pragma once

#include <atomic>
#include <cassert>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#ifdef VL_RWLOCK_DEBUG
#include <vector>
#endif

namespace vl {

class RwLock {  // SharedLockable: works with std::shared_lock / std::unique_lock / std::lock_guard
public:
    void lock()
    {
        {
            std::lock_guard<std::mutex> g(gate_mu_);
            ++writers_waiting_;
        }
        mu_.lock();
        {
            std::lock_guard<std::mutex> g(gate_mu_);
            --writers_waiting_;
        }
        gate_cv_.notify_all();  // readers that stood back may queue on mu_ now (they get in when this writer unlocks)
    }
    bool try_lock() { return mu_.try_lock(); }
    void unlock() { mu_.unlock(); }
    void lock_shared()
    {
#ifdef VL_RWLOCK_DEBUG
        for (const RwLock* h : held()) assert(h != this && "RwLock: the shared side is not re-entrant (a waiting writer would deadlock the second lock_shared)");
#endif
        {
            std::unique_lock<std::mutex> g(gate_mu_);
            gate_cv_.wait(g, [this] { return writers_waiting_ == 0; });
        }
        // (a writer may arrive between the gate and the lock: it then waits for this reader like for any earlier one --
        // the priority is best effort by design, what matters is that readers arriving LATER stand back)
        mu_.lock_shared();
#ifdef VL_RWLOCK_DEBUG
        held().push_back(this);
#endif
    }
    bool try_lock_shared()
    {
        {
            std::lock_guard<std::mutex> g(gate_mu_);
            if (writers_waiting_ != 0) return false;
        }
        const bool ok = mu_.try_lock_shared();
#ifdef VL_RWLOCK_DEBUG
        if (ok) held().push_back(this);
#endif
        return ok;
    }
    void unlock_shared()
    {
#ifdef VL_RWLOCK_DEBUG
        for (size_t i = held().size(); i-- > 0;)
            if (held()[i] == this) {
                held().erase(held().begin() + (long)i);
                break;
            }
#endif
        mu_.unlock_shared();
    }

private:
#ifdef VL_RWLOCK_DEBUG  // diagnostic builds (tests/native/rwlock_test.cpp): the locks this thread holds shared
    static std::vector<const RwLock*>& held()
    {
        thread_local std::vector<const RwLock*> h;  // nested shared locks of DIFFERENT handles are legal (HNSW -> its row store)
        return h;
    }
#endif
    std::shared_mutex mu_;
    std::mutex gate_mu_;
    std::condition_variable gate_cv_;
    int writers_waiting_ = 0;  // under gate_mu_
};

}  // namespace vl
