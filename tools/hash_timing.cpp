// Host transcript hash cost: eight- and sixteen-lane AVX-512 IFMA MiMC7 against the scalar 4x64-bit code.
//   g++ -O3 -std=c++17 -Igkr_amd/csrc tools/hash_timing.cpp gkr_amd/csrc/mimc_ifma.o gkr_amd/csrc/keccak.o -o /tmp/hash_timing
#include "fr64.h"
#include "keccak.h"
#include "mimc7.h"
#include "mimc_ifma.h"
#include <chrono>
#include <cstdio>
#include <cstring>
using namespace gkr;

int main() {
    Fr cts32[91];
    mimc7_make_constants(cts32);
    h64::F cts64[91];
    memcpy(cts64, cts32, sizeof cts64);
    static uint64_t vec[16][3][4], out[16][4];
    uint32_t len[16];
    for (int k = 0; k < 16; ++k)
        for (int s = 0; s < 3; ++s)
            for (int j = 0; j < 4; ++j) vec[k][s][j] = (uint64_t)(k * 7 + s * 3 + j + 1) * 0x9e3779b97f4a7c15ull >> (j == 3 ? 4 : 0);
    const int N = 2000;
    uint64_t acc = 0;
    for (uint32_t l = 2; l <= 3; ++l) {
        for (int k = 0; k < 16; ++k) len[k] = l;
        if (gkr_ifma_available()) {
            uint64_t canon[91][4];
            for (int i = 0; i < 91; ++i) {
                h64::F c = h64::from_mont(cts64[i]);
                memcpy(canon[i], &c, 32);
            }
            gkr_ifma_init(canon);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) { vec[0][1][0] = i; gkr_ifma_multi_hash8(vec, len, 3, out); acc ^= out[0][0]; }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("len %u  ifma8 : %6.2f us per call = %.2f us per hash\n", l, us, us / 8);
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) { vec[0][1][0] = i; gkr_ifma_multi_hash16(vec, len, 3, out); acc ^= out[0][0]; }
            us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("len %u  ifma16: %6.2f us per call = %.2f us per hash\n", l, us, us / 16);
        }
        h64::F v[3];
        memcpy(v, vec[0], sizeof v);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) { v[1].l[0] = i; h64::F r = h64::mimc7_multi_hash(v + (3 - l), (int)l, cts64, nullptr); acc ^= r.l[0]; }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("len %u  scalar: %6.2f us per hash\n", l, us);
    }
    if (gkr_ifma_available()) {
        // the host's whole share of a 5-round pass for 16 sumchecks against its five 2-element hashes alone
        static uint64_t sums[16][32][4], c0[5][16][4], c1[5][16][4], r[5][16][4], w[16][32][4];
        static uint32_t ln[5][16];
        for (int k = 0; k < 16; ++k)
            for (int b = 0; b < 32; ++b)
                for (int j = 0; j < 4; ++j) sums[k][b][j] = (uint64_t)(k * 37 + b * 5 + j + 1) * 0x9e3779b97f4a7c15ull >> (j == 3 ? 4 : 0);
        for (int k = 0; k < 16; ++k) len[k] = 2;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            sums[0][0][0] = i;
            gkr_ifma_pass(&sums[0][0][0], 128, 16, 5, nullptr, c0, c1, r, ln, &w[0][0][0], 128);
            acc ^= r[4][0][0] ^ w[3][5][1];
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i)
            for (int t = 0; t < 5; ++t) { vec[0][1][0] = i + t; gkr_ifma_multi_hash16(vec, len, 3, out); acc ^= out[0][0]; }
        const double us_h = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("5-round pass, 16 lanes: %6.2f us whole (gkr_ifma_pass), %6.2f us its five hashes alone\n", us, us_h);
    }
    if (gkr_ifma_available()) {
        // the host's whole share of a 3-round PRODUCT pass of the layer sumcheck for 16 proofs (72-value records: the 8 x 8
        // cross sums and the 8 sub-block sums) against its three 3-element hashes alone
        static uint64_t recs[16][73][4], c2[3][16][4], lin[3][16][4], c0p[3][16][4], rr[3][16][4], wts[16][8][4];
        static uint32_t vl[3][16];
        for (int k = 0; k < 16; ++k)
            for (int b = 0; b < 72; ++b)
                for (int j = 0; j < 4; ++j) recs[k][b][j] = (uint64_t)(k * 91 + b * 7 + j + 1) * 0x9e3779b97f4a7c15ull >> (j == 3 ? 4 : 0);
        for (int t = 0; t < 3; ++t)
            for (int k = 0; k < 16; ++k) vl[t][k] = 3;
        for (int k = 0; k < 16; ++k) len[k] = 3;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            recs[0][0][0] = i;
            gkr_ifma_prod_pass(&recs[0][0][0], 73 * 4, 16, 3, vl, c2, lin, c0p, rr, &wts[0][0][0], 32);
            acc ^= rr[2][0][0] ^ wts[3][5][1];
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i)
            for (int t = 0; t < 3; ++t) { vec[0][1][0] = i + t; gkr_ifma_multi_hash16(vec, len, 3, out); acc ^= out[0][0]; }
        const double us_h = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("3-round product pass, 16 lanes: %6.2f us whole (gkr_ifma_prod_pass), %6.2f us its three hashes alone\n", us, us_h);
    }
    printf("(%llx)\n", (unsigned long long)acc);
    return 0;
}
