// gkr_r1cs_compile builds the constraints' trees on several threads through one arena (sharded interning tables behind
// spinlocks, per-thread id blocks and caches; gkr_amd/csrc/r1cs.cpp) and compiles its <= 20 groups in parallel.  Built with
// -fsanitize=thread (make -C gkr_amd/csrc tsan_r1cs): an R1CS shaped like a circom MiMC chain -- shared constants, every
// constraint reading its predecessor's wires, a few wide linear combinations -- compiled with 1, 3 and 8 threads must give
// the SAME circuits (node ids depend on the threads' timing; the output must not) and no race report.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/gkr_amd.h"

static std::vector<unsigned char> snapshot(const gkr_layered* L) {
    std::vector<unsigned char> out;
    auto put = [&](const void* p, size_t n) { out.insert(out.end(), (const unsigned char*)p, (const unsigned char*)p + n); };
    uint32_t count = 0;
    gkr_layered_count(L, &count);
    put(&count, 4);
    for (uint32_t j = 0; j < count; ++j) {
        gkr_circuit_desc d;
        gkr_layered_circuit(L, j, &d);
        put(&d.depth, 4);
        put(d.k, 4 * (d.depth + 1));
        for (uint32_t i = 0; i < d.depth; ++i) {
            const size_t g = (size_t)1 << d.k[i];
            put(d.gate_type[i], g);
            put(d.left[i], 4 * g);
            put(d.right[i], 4 * g);
        }
        const uint32_t* wire;
        const gkr_fr* constant;
        size_t slots;
        gkr_layered_input_layer(L, j, &wire, &constant, &slots);
        put(wire, 4 * slots);
        put(constant, 32 * slots);
    }
    return out;
}

int main(int argc, char** argv) {
    const uint32_t rounds = argc > 1 ? (uint32_t)atoi(argv[1]) : 3000;
    // wires: 0 = one, 1 = out, 2 = x, 3 = key, then four per round (t2, t4, t6, t7)
    const uint32_t n_wires = 4 + 4 * rounds;
    std::vector<uint32_t> counts, wires;
    std::vector<gkr_fr> coeffs;
    auto term = [&](uint32_t w, uint64_t c) {
        wires.push_back(w);
        coeffs.push_back(gkr_fr{{c, 0, 0, 0}});
    };
    uint32_t prev = 2;
    for (uint32_t i = 0; i < rounds; ++i) {
        const uint32_t t2 = 4 + 4 * i, t4 = t2 + 1, t6 = t2 + 2, t7 = t2 + 3;
        const uint64_t c = 1000 + (i % 91);           // 91 round constants, shared by the whole chain
        // (prev + key + c) * (prev + key + c) = t2
        term(prev, 1); term(3, 1); term(0, c); term(prev, 1); term(3, 1); term(0, c); term(t2, 1);
        counts.insert(counts.end(), {3, 3, 1});
        term(t2, 1); term(t2, 1); term(t4, 1);
        counts.insert(counts.end(), {1, 1, 1});
        term(t4, 1); term(t2, 1); term(t6, 1);
        counts.insert(counts.end(), {1, 1, 1});
        // t6 * (prev + key + c) = t7, with a wide C every 64th round
        term(t6, 1); term(prev, 1); term(3, 1); term(0, c); term(t7, 1);
        uint32_t extra = 0;
        if (i % 64 == 0)
            for (; extra < 5; ++extra) term(2 + extra % 2, 7 + extra);
        counts.insert(counts.end(), {1, 3, 1 + extra});
        prev = t7;
    }
    gkr_r1cs* r = nullptr;
    if (gkr_r1cs_build(n_wires, 1, 1, 1, counts.size() / 3, counts.data(), wires.data(), coeffs.data(), &r)) return 2;
    std::vector<unsigned char> want;
    int bad = 0;
    for (const char* threads : {"1", "3", "8", "8", "2"}) {
        setenv("GKR_COMPILE_THREADS", threads, 1);
        gkr_layered* L = nullptr;
        size_t bad_constraint = 0;
        if (gkr_r1cs_compile(r, &L, &bad_constraint)) return 3;
        const std::vector<unsigned char> got = snapshot(L);
        if (want.empty()) want = got;
        if (got != want) ++bad;
        gkr_layered_free(L);
    }
    gkr_r1cs_free(r);
    printf("constraints=%zu bytes=%zu bad=%d\n", counts.size() / 3, want.size(), bad);
    return bad ? 1 : 0;
}
