#include "keccak.h"

#include <string.h>

#include "mimc7.h"

namespace gkr {
namespace {

inline uint64_t rotl(uint64_t v, unsigned n) { return n ? (v << n) | (v >> (64 - n)) : v; }

// rho offsets indexed [x][y]
const unsigned kRho[5][5] = {
    {0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};

void permute(uint64_t a[5][5]) {
    uint64_t lfsr = 1;  // round constants from the degree-8 LFSR of the Keccak spec
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[5][5];
        for (int x = 0; x < 5; ++x) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) a[x][y] ^= d[x];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y][(2 * x + 3 * y) % 5] = rotl(a[x][y], kRho[x][y]);
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) a[x][y] = b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y]);
        uint64_t rc = 0;
        for (int j = 0; j < 7; ++j) {
            if (lfsr & 1) rc ^= 1ULL << ((1u << j) - 1);
            lfsr = (lfsr & 0x80) ? ((lfsr << 1) ^ 0x171) : (lfsr << 1);
        }
        a[0][0] ^= rc;
    }
}

}  // namespace

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint64_t a[5][5];
    memset(a, 0, sizeof a);
    uint8_t block[136];
    for (;;) {
        const bool last = len < rate;
        if (last) {
            memset(block, 0, rate);
            memcpy(block, data, len);
            block[len] ^= 0x01;
            block[rate - 1] ^= 0x80;
        } else {
            memcpy(block, data, rate);
        }
        for (size_t i = 0; i < rate / 8; ++i) {
            uint64_t w = 0;
            for (int j = 7; j >= 0; --j) w = (w << 8) | block[8 * i + j];
            a[i % 5][i / 5] ^= w;
        }
        permute(a);
        if (last) break;
        data += rate;
        len -= rate;
    }
    for (int i = 0; i < 4; ++i) {
        uint64_t w = a[i % 5][i / 5];
        for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(w >> (8 * j));
    }
}

void mimc7_make_constants(Fr* cts_mont) {
    uint8_t h[32];
    keccak256(reinterpret_cast<const uint8_t*>("mimc"), 4, h);
    cts_mont[0] = fr_zero();
    for (int i = 1; i < kMimcRounds; ++i) {
        uint8_t nh[32];
        keccak256(h, 32, nh);
        memcpy(h, nh, 32);
        Fr v;  // big-endian digest -> little-endian limbs
        for (int l = 0; l < 8; ++l) {
            const uint8_t* p = h + 4 * (7 - l);
            v.l[l] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        }
        for (int k = 0; k < 5; ++k) v = fr_reduce_once(v);  // 2^256 < 6 r
        cts_mont[i] = to_mont(v);
    }
}

}  // namespace gkr
