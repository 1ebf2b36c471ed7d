// The host transcript's lazily built tables (MiMC7 constants in both layouts, the 52-bit IFMA tables) are first
// touched by up to 64 crew threads of gkr_prove_many at the same instant.  This driver makes that first touch happen
// from 32 threads at once through the C ABI's host-only entry points (no GPU needed) and is run under ThreadSanitizer
// against a -fsanitize=thread build of the library's host side (`make tsan`): no race report, and every thread must
// get the known answers.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/gkr_amd.h"

int main() {
    const int threads = 32;
    std::atomic<int> ready{0}, bad{0};
    std::atomic<bool> go{false};
    // multi_hash([12], 0) = 0x237c92644dbddb86d8a259e0e923aaab65a93f1ec5758b8799988894ac0958fd
    const uint64_t want[4] = {0x99988894ac0958fdull, 0x65a93f1ec5758b87ull, 0xd8a259e0e923aaabull, 0x237c92644dbddb86ull};
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t)
        ts.emplace_back([&, t] {
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) {
            }
            gkr_fr x = {{12, 0, 0, 0}}, key = {{0, 0, 0, 0}}, out = {{0, 0, 0, 0}};
            if (t % 3 == 0) {
                if (gkr_mimc7_multi_hash(&x, 1, &key, &out) != 0 || memcmp(out.l, want, 32) != 0) bad.fetch_add(1);
            } else if (t % 3 == 1) {
                gkr_fr vecs[24];
                uint32_t len[8];
                gkr_fr outs[8];
                memset(vecs, 0, sizeof vecs);
                for (int k = 0; k < 8; ++k) {
                    vecs[3 * k + 2].l[0] = 12;
                    len[k] = 1;
                }
                int used = -1;
                if (gkr_selftest_hash8(vecs, len, outs, &used) != 0) bad.fetch_add(1);
                for (int k = 0; k < 8; ++k)
                    if (memcmp(outs[k].l, want, 32) != 0) bad.fetch_add(1);
            } else {
                double a = 0, b = 0;
                if (gkr_ubench_host_hash(2, &a, &b) != 0) bad.fetch_add(1);
                if (gkr_mimc7_multi_hash(&x, 1, &key, &out) != 0 || memcmp(out.l, want, 32) != 0) bad.fetch_add(1);
            }
        });
    while (ready.load() < threads) {
    }
    go.store(true, std::memory_order_release);
    for (auto& t : ts) t.join();
    printf("first_use_race: bad=%d\n", bad.load());
    return bad.load() ? 1 : 0;
}
