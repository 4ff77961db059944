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
// A thread must not take the shared side twice (a reader that re-enters behind a waiting writer would wait for itself):
// the classes that use this lock hand their already-locked paths a *_locked variant instead.
#pragma once

#include <atomic>
#include <shared_mutex>
#include <thread>

namespace vl {

class RwLock {  // SharedLockable: works with std::shared_lock / std::unique_lock / std::lock_guard
public:
    void lock()
    {
        writers_waiting_.fetch_add(1, std::memory_order_acq_rel);
        mu_.lock();
        writers_waiting_.fetch_sub(1, std::memory_order_acq_rel);
    }
    bool try_lock() { return mu_.try_lock(); }
    void unlock() { mu_.unlock(); }
    void lock_shared()
    {
        while (writers_waiting_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
        mu_.lock_shared();
    }
    bool try_lock_shared() { return writers_waiting_.load(std::memory_order_acquire) == 0 && mu_.try_lock_shared(); }
    void unlock_shared() { mu_.unlock_shared(); }

private:
    std::shared_mutex mu_;
    std::atomic<int> writers_waiting_{0};
};

}  // namespace vl
