// Stress of SpinPool::run_now and HelpBoard (gkr_amd/csrc/hostpool.h): many very short jobs whose state lives in the caller's
// stack frame.  A worker that still calls a job after run_now has returned (the retire-then-check race) finds the
// frame re-used: the job then sees a poisoned tag and the run fails.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

#include "hostpool.h"

static std::atomic<long> bad{0};

static void one_round(gkr::SpinPool& pool, int items, unsigned tag_value) {
    struct Frame {
        std::atomic<int> next{0};
        std::atomic<unsigned> tag{0};
        std::atomic<int> done{0};
    } f;
    f.tag.store(tag_value);
    const int n = items;
    const std::function<bool()> work = [&]() -> bool {
        if (f.tag.load(std::memory_order_relaxed) != tag_value) bad.fetch_add(1);
        const int i = f.next.fetch_add(1, std::memory_order_relaxed);
        if (i >= n) return false;
        f.done.fetch_add(1, std::memory_order_relaxed);
        return true;
    };
    pool.run_now(&work);
    if (f.done.load() != n) bad.fetch_add(1);
    f.tag.store(0xDEADBEEFu);   // the frame is dead from here on
}

int main(int argc, char** argv) {
    const int workers = argc > 1 ? atoi(argv[1]) : 1;
    const long rounds = argc > 2 ? atol(argv[2]) : 300000;
    gkr::SpinPool pool(workers);
    pool.begin_session(nullptr);
    for (long r = 0; r < rounds; ++r) one_round(pool, 2 + (int)(r % 3), 0x1000u + (unsigned)(r & 0xFFF));
    pool.end_session();
    // sessions opened and closed in quick succession, each with its own try_work function in the caller's frame
    // (the shape of the plain-sumcheck scheduler: workers spin on the function until the session closes)
    for (long sidx = 0; sidx < rounds / 100 + 10; ++sidx) {
        std::atomic<int> left{50};
        std::atomic<int> taken{0};
        const std::function<bool()> try_work = [&]() -> bool {
            int v = left.load(std::memory_order_relaxed);
            while (v > 0)
                if (left.compare_exchange_weak(v, v - 1)) {
                    taken.fetch_add(1, std::memory_order_relaxed);
                    return true;
                }
            return false;
        };
        {
            gkr::SpinPool::Session session(&pool, &try_work);
            while (left.load() > 0) try_work();
        }   // the guard closes the session: no worker may touch try_work / left / taken after this line
        if (taken.load() != 50) bad.fetch_add(1);
    }
    // HelpBoard: several owner threads post very short jobs from their own stack frames and help each other between
    // their own jobs (the shape of contexts proving side by side); a helper that still runs a job after its owner's
    // retire() has returned finds the frame poisoned
    {
        const int owners = workers + 2;
        const long jobs = rounds / 20 + 100;
        std::vector<std::thread> ts;
        for (int o = 0; o < owners; ++o)
            ts.emplace_back([o, jobs] {
                gkr::HelpBoard& board = gkr::HelpBoard::instance();
                for (long j = 0; j < jobs; ++j) {
                    struct Frame {
                        std::atomic<int> next{0};
                        std::atomic<unsigned> tag{0};
                        std::atomic<int> done{0};
                    } f;
                    const unsigned tag_value = 0x2000u + (unsigned)((o * 977 + j) & 0xFFF);
                    f.tag.store(tag_value);
                    const int n = 1 + (int)(j % 5);
                    const std::function<bool()> work = [&]() -> bool {
                        if (f.tag.load(std::memory_order_relaxed) != tag_value) bad.fetch_add(1);
                        const int i = f.next.fetch_add(1, std::memory_order_relaxed);
                        if (i >= n) return false;
                        f.done.fetch_add(1, std::memory_order_relaxed);
                        return true;
                    };
                    {
                        gkr::HelpBoard::Posted posted(&work);
                        while (work()) {
                        }
                    }   // retired: every claimed piece has been run to its end
                    if (f.done.load() != n) bad.fetch_add(1);
                    f.tag.store(0xDEADBEEFu);
                    for (int h = 0; h < 3; ++h) board.help();   // "waiting for the GPU": take pieces of the others' jobs
                }
            });
        for (auto& t : ts) t.join();
    }
    printf("workers=%d rounds=%ld bad=%ld\n", workers, rounds, bad.load());
    return bad.load() ? 1 : 0;
}
