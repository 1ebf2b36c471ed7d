// Worker pool for the host transcript.
//
// Each sumcheck round needs one MiMC7 hash (~10 us) per sumcheck between two
// kernel launches, so wake-up latency matters more than fairness: while a
// session is open (inside one API call) the workers spin on a caller-supplied
// "try to find and do one unit of work" function; outside a session they sleep.
// The caller sizes the pool from the CPUs the process may really use (cgroup
// quota included): spinning on more threads than the quota allows gets the
// whole process throttled for the rest of a scheduler period.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#define GKR_CPU_RELAX() _mm_pause()
#else
#define GKR_CPU_RELAX() ((void)0)
#endif

namespace gkr {

// Process-wide board of host work that ANY thread of the library may take a piece of while it waits for its own
// GPU: contexts proving side by side (one calling thread each, gkr_amd.aggregate.ProvingStep) otherwise leave every
// thread that waits for a round's record spinning while another one works through its 64 hashes alone.
//   owner:   slot = post(&work); run `work` itself until it returns false; retire(slot)
//   helper:  help() -- runs one piece of one posted job, or returns false
// `work` claims and runs one piece per call (true) or reports that nothing is left to claim (false); it must be safe
// to call from many threads.  retire() returns once no helper is inside the job any more, so the job and everything
// it references may live on the owner's stack.
class HelpBoard {
   public:
    static HelpBoard& instance() {
        static HelpBoard board;
        return board;
    }
    // priority: how much work its owner still has ahead of it after this job (any unit, e.g. sumcheck rounds left in its
    // proof): helpers take pieces of the job whose owner is furthest from done -- the critical path of the whole step.
    // -1: no free slot (the owner then simply works alone)
    int post(const std::function<bool()>* work, int priority = 0) {
        for (int i = 0; i < kSlots; ++i) {
            const std::function<bool()>* expected = nullptr;
            if (slots_[i].job.compare_exchange_strong(expected, work, std::memory_order_seq_cst)) {
                slots_[i].priority.store(priority, std::memory_order_relaxed);   // (a helper that still reads the old value only picks differently)
                posted_.fetch_add(1, std::memory_order_seq_cst);
                return i;
            }
        }
        return -1;
    }
    void retire(int slot) {
        if (slot < 0) return;
        // store, then load another variable; the helper does the mirror image (count up, then read the job): both
        // sides sequentially consistent, as in SpinPool::run_now
        slots_[slot].job.store(nullptr, std::memory_order_seq_cst);
        posted_.fetch_sub(1, std::memory_order_seq_cst);
        while (slots_[slot].inside.load(std::memory_order_seq_cst) != 0) GKR_CPU_RELAX();
    }
    bool help() {
        if (posted_.load(std::memory_order_acquire) == 0) return false;
        const int start = next_.fetch_add(1, std::memory_order_relaxed);   // equal priorities: helpers spread over the jobs
        // posted jobs, highest priority first (at most kSlots rounds: a job whose pieces are all claimed returns false)
        uint64_t tried = 0;
        for (;;) {
            int best = -1, best_prio = 0;
            for (int n = 0; n < kSlots; ++n) {
                const int i = (start + n) % kSlots;
                if ((tried >> i) & 1u) continue;
                if (slots_[i].job.load(std::memory_order_acquire) == nullptr) continue;
                const int pr = slots_[i].priority.load(std::memory_order_relaxed);
                if (best < 0 || pr > best_prio) {
                    best = i;
                    best_prio = pr;
                }
            }
            if (best < 0) return false;
            tried |= (uint64_t)1 << best;
            Slot& sl = slots_[best];
            sl.inside.fetch_add(1, std::memory_order_seq_cst);
            const std::function<bool()>* job = sl.job.load(std::memory_order_seq_cst);
            const bool did = job && (*job)();
            sl.inside.fetch_sub(1, std::memory_order_seq_cst);
            if (did) return true;
        }
    }

    // Scope guard: the job is on the board for the lifetime of the guard
    class Posted {
       public:
        explicit Posted(const std::function<bool()>* work, int priority = 0) : slot_(HelpBoard::instance().post(work, priority)) {}
        ~Posted() { HelpBoard::instance().retire(slot_); }
        Posted(const Posted&) = delete;
        Posted& operator=(const Posted&) = delete;

       private:
        int slot_;
    };

   private:
    static constexpr int kSlots = 64;   // (the `tried` mask of help() is one 64-bit word)
    struct alignas(64) Slot {
        std::atomic<const std::function<bool()>*> job{nullptr};
        std::atomic<int> inside{0};
        std::atomic<int> priority{0};
    };
    Slot slots_[kSlots];
    std::atomic<int> posted_{0};
    std::atomic<unsigned> next_{0};
};

class SpinPool {
   public:
    explicit SpinPool(int workers) {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { worker(); });
    }
    ~SpinPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_.store(true);
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int workers() const { return (int)threads_.size(); }

    // try_work: returns true if it did something.  Must be safe to call from many threads.
    void begin_session(const std::function<bool()>* try_work) {
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = try_work;
            session_.fetch_add(1, std::memory_order_release);   // odd = open
        }
        cv_.notify_all();
    }
    // Within an open session: run `work` on the workers and the caller until it reports no work left,
    // then return once every participant has left it.  (A session opened with a null function idles
    // between run_now calls.)
    void run_now(const std::function<bool()>* work) {
        // job_busy_ is a balanced in/out count of workers inside the probe window: never reset it.
        // Retiring the job is a store followed by a load of ANOTHER variable, and a worker does the mirror
        // image (count up, then read the job): both sides must be sequentially consistent, or the caller can
        // read "no worker inside" before its own store is visible while a worker still picks up the old job and
        // calls it after the caller's stack frame is gone (seen as a crash with one worker and short rounds).
        job_.store(work, std::memory_order_seq_cst);
        while ((*work)()) {
        }
        job_.store(nullptr, std::memory_order_seq_cst);
        while (job_busy_.load(std::memory_order_seq_cst) != 0) GKR_CPU_RELAX();
    }

    // returns once no worker is inside try_work any more
    void end_session() {
        // same store-then-load-another-variable shape as run_now: sequentially consistent on both sides
        session_.fetch_add(1, std::memory_order_seq_cst);       // even = closed
        while (inside_.load(std::memory_order_seq_cst) != 0) GKR_CPU_RELAX();
    }

    // Scope guard for a session: whatever path leaves the scope (an early error return included), the
    // session is closed and no worker is left spinning on a function object of a dead stack frame.
    class Session {
       public:
        Session(SpinPool* pool, const std::function<bool()>* try_work) : pool_(pool) {
            if (pool_) pool_->begin_session(try_work);
        }
        ~Session() { close(); }
        Session(const Session&) = delete;
        Session& operator=(const Session&) = delete;
        void close() {
            if (pool_) pool_->end_session();
            pool_ = nullptr;
        }

       private:
        SpinPool* pool_;
    };

   private:
    void worker() {
        for (;;) {
            uint64_t s = session_.load(std::memory_order_acquire);
            if (!(s & 1)) {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return stop_.load() || (session_.load() & 1); });
                if (stop_.load()) return;
                continue;
            }
            inside_.fetch_add(1, std::memory_order_seq_cst);
            // re-check after announcing: end_session() may have closed in between
            if (session_.load(std::memory_order_seq_cst) == s) {
                const std::function<bool()>* fn = fn_;
                unsigned idle = 0;
                while (session_.load(std::memory_order_acquire) == s) {
                    bool did = false;
                    if (fn) {
                        did = (*fn)();
                    } else {
                        job_busy_.fetch_add(1, std::memory_order_seq_cst);
                        const std::function<bool()>* job = job_.load(std::memory_order_seq_cst);
                        if (job) did = (*job)();
                        job_busy_.fetch_sub(1, std::memory_order_seq_cst);
                    }
                    if (!did) did = HelpBoard::instance().help();   // nothing of this context's: another context's posted work
                    if (did) {
                        idle = 0;
                    } else {
                        GKR_CPU_RELAX();
                        if (++idle > 64) {   // back off a little: keeps the memory system quiet while the GPU works
                            for (int k = 0; k < 16; ++k) GKR_CPU_RELAX();
                        }
                    }
                }
            }
            inside_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> session_{0};
    std::atomic<int> inside_{0};
    std::atomic<bool> stop_{false};
    const std::function<bool()>* fn_ = nullptr;
    std::atomic<const std::function<bool()>*> job_{nullptr};
    std::atomic<int> job_busy_{0};
};

}  // namespace gkr
