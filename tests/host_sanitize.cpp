// Host-only units of the product under AddressSanitizer + UBSan (gkr_amd/csrc/Makefile, targets `asan`): the R1CS /
// witness containers and the R1CS -> layered-circuit compiler (r1cs.cpp), the proof -> verifier.circom text
// (circom_input.cpp), keccak-256 and the MiMC7 constants (keccak.cpp), the 8/16-lane IFMA MiMC7 (mimc_ifma.cpp),
// driven through the C ABI with valid, ragged and hostile inputs.  Sanitizers cannot run on the GPU box's device
// code; this is the CPU side.  Exit status 0 and no sanitizer report = pass.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "../include/gkr_amd.h"
#include "../gkr_amd/csrc/fr64.h"
#include "../gkr_amd/csrc/keccak.h"
#include "../gkr_amd/csrc/mimc7.h"
#include "../gkr_amd/csrc/mimc_adx.h"
#include "../gkr_amd/csrc/mimc_ifma.h"

static int failures = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            printf("CHECK failed line %d: %s\n", __LINE__, #cond);      \
            ++failures;                                                  \
        }                                                                \
    } while (0)

static const uint64_t kMod[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static gkr_fr small(uint64_t v) { return gkr_fr{{v, 0, 0, 0}}; }
static gkr_fr minus_one() { return gkr_fr{{kMod[0] - 1, kMod[1], kMod[2], kMod[3]}}; }

static void r1cs_round_trips(std::mt19937_64& rng) {
    for (int trial = 0; trial < 40; ++trial) {
        const uint32_t n_wires = 2 + rng() % 40;
        const size_t n_cons = 1 + rng() % 60;
        std::vector<uint32_t> counts, wires;
        std::vector<gkr_fr> coeffs;
        for (size_t i = 0; i < n_cons; ++i)
            for (int j = 0; j < 3; ++j) {
                const uint32_t n = 1 + rng() % 5;
                counts.push_back(n);
                for (uint32_t t = 0; t < n; ++t) {
                    wires.push_back((uint32_t)(rng() % n_wires));
                    const int kind = rng() % 4;
                    coeffs.push_back(kind == 0 ? small(1) : kind == 1 ? minus_one() : kind == 2 ? small(0) : gkr_fr{{rng(), rng(), rng(), rng() >> 4}});
                }
            }
        gkr_r1cs* r = nullptr;
        CHECK(gkr_r1cs_build(n_wires, 1, 1, n_wires > 3 ? n_wires - 3 : 0, n_cons, counts.data(), wires.data(), coeffs.data(), &r) == GKR_OK);
        if (!r) continue;
        size_t need = 0;
        CHECK(gkr_r1cs_serialize(r, nullptr, 0, &need) == GKR_OK && need > 12);
        std::vector<uint8_t> image(need);
        CHECK(gkr_r1cs_serialize(r, image.data(), need - 1, &need) == GKR_ERR_NOMEM);
        CHECK(gkr_r1cs_serialize(r, image.data(), need, &need) == GKR_OK);
        gkr_r1cs* again = nullptr;
        CHECK(gkr_r1cs_parse(image.data(), image.size(), &again) == GKR_OK);
        gkr_r1cs_info_t a, b;
        CHECK(gkr_r1cs_info(r, &a) == GKR_OK && gkr_r1cs_info(again, &b) == GKR_OK && a.n_terms == b.n_terms && a.n_constraints == n_cons);
        // every truncation and a few corruptions of the image: a status, never an out-of-bounds read
        for (size_t cut = 0; cut < image.size(); cut += 1 + image.size() / 97) {
            gkr_r1cs* bad = nullptr;
            const int rc = gkr_r1cs_parse(image.data(), cut, &bad);
            CHECK(rc != GKR_OK || bad != nullptr);
            gkr_r1cs_free(bad);
        }
        for (int c = 0; c < 60; ++c) {
            std::vector<uint8_t> mut = image;
            mut[rng() % mut.size()] ^= (uint8_t)(1u << (rng() % 8));
            gkr_r1cs* bad = nullptr;
            (void)gkr_r1cs_parse(mut.data(), mut.size(), &bad);
            if (bad) {
                gkr_layered* L = nullptr;
                (void)gkr_r1cs_compile(bad, &L, nullptr);
                gkr_layered_free(L);
            }
            gkr_r1cs_free(bad);
        }
        gkr_layered* L = nullptr;
        size_t bad_idx = 0;
        CHECK(gkr_r1cs_compile(again, &L, &bad_idx) == GKR_OK);
        uint32_t n_circ = 0;
        CHECK(gkr_layered_count(L, &n_circ) == GKR_OK && n_circ >= 1 && n_circ <= 20);
        std::vector<gkr_fr> witness(n_wires);
        for (auto& w : witness) w = gkr_fr{{rng(), rng(), rng(), rng() >> 4}};
        for (uint32_t j = 0; j < n_circ; ++j) {
            gkr_circuit_desc d;
            CHECK(gkr_layered_circuit(L, j, &d) == GKR_OK);
            for (uint32_t i = 0; i < d.depth; ++i)
                for (size_t g = 0; g < ((size_t)1 << d.k[i]); ++g)
                    CHECK(d.gate_type[i][g] <= 1 && (d.left[i][g] >> d.k[i + 1]) == 0 && (d.right[i][g] >> d.k[i + 1]) == 0);
            size_t slots = 0;
            CHECK(gkr_layered_input_layer(L, j, nullptr, nullptr, &slots) == GKR_OK && slots == ((size_t)1 << d.k[d.depth]));
            std::vector<gkr_fr> vals(slots);
            CHECK(gkr_layered_input_values(L, j, witness.data(), witness.size(), vals.data()) == GKR_OK);
            CHECK(gkr_layered_input_values(L, j, witness.data(), 1, vals.data()) != GKR_OK || n_wires == 1);
        }
        CHECK(gkr_layered_circuit(L, n_circ, nullptr) == GKR_ERR_INVALID);
        gkr_layered_free(L);
        gkr_r1cs_free(again);
        gkr_r1cs_free(r);
    }
    // witness container
    std::vector<gkr_fr> vals(33);
    for (auto& v : vals) v = gkr_fr{{rng(), rng(), rng(), rng() >> 4}};
    size_t need = 0;
    CHECK(gkr_wtns_serialize(vals.data(), vals.size(), nullptr, 0, &need) == GKR_OK);
    std::vector<uint8_t> image(need);
    CHECK(gkr_wtns_serialize(vals.data(), vals.size(), image.data(), need, &need) == GKR_OK);
    std::vector<gkr_fr> back(33);
    size_t count = 0;
    CHECK(gkr_wtns_parse(image.data(), image.size(), back.data(), back.size(), &count) == GKR_OK && count == 33);
    CHECK(memcmp(back.data(), vals.data(), 33 * 32) == 0);
    CHECK(gkr_wtns_parse(image.data(), image.size(), back.data(), 32, &count) == GKR_ERR_NOMEM);
    for (size_t cut = 0; cut < image.size(); cut += 7) (void)gkr_wtns_parse(image.data(), cut, back.data(), back.size(), &count);
}

static void circom_text(std::mt19937_64& rng) {
    for (int trial = 0; trial < 30; ++trial) {
        const uint32_t L = 1 + rng() % 4;
        std::vector<uint32_t> k(L + 1);
        for (auto& x : k) x = rng() % 5;
        for (uint32_t i = 1; i <= L; ++i) k[i] = 1 + k[i] % 4;
        size_t rounds = 0, qs = 0, zs = 0;
        for (uint32_t i = 0; i < L; ++i) {
            rounds += 2 * k[i + 1];
            qs += k[i + 1] + 1;
        }
        for (uint32_t i = 0; i <= L; ++i) zs += k[i];
        auto rnd = [&](size_t n) {
            std::vector<gkr_fr> v(n ? n : 1);
            for (auto& x : v) x = (rng() % 3 == 0) ? small(0) : gkr_fr{{rng(), rng(), rng(), rng() >> 4}};
            return v;
        };
        std::vector<gkr_fr> sc = rnd(rounds * 3), sr = rnd(rounds), q = rnd(qs), z = rnd(zs), r = rnd(L), d = rnd((size_t)1 << k[0]),
                            in = rnd((size_t)1 << k[L]);
        std::vector<uint32_t> sl(rounds ? rounds : 1), ql(L);
        for (auto& x : sl) x = 1 + rng() % 3;
        for (uint32_t i = 0; i < L; ++i) ql[i] = 1 + rng() % (k[i + 1] + 1);
        gkr_circuit_desc desc{L, k.data(), nullptr, nullptr, nullptr};
        gkr_proof_buf p{sc.data(), sl.data(), sr.data(), q.data(), ql.data(), z.data(), r.data(), d.data(), in.data()};
        std::vector<uint32_t> meta(8 + L + 1);
        size_t count = 0;
        CHECK(gkr_circom_meta(&desc, &p, meta.data(), meta.size(), &count) == GKR_OK && count == 8 + L + 1);
        size_t need = 0;
        CHECK(gkr_circom_input_json(&desc, &p, trial, nullptr, 0, &need) == GKR_OK);
        std::string text(need, '\0');
        CHECK(gkr_circom_input_json(&desc, &p, trial, &text[0], need, &need) == GKR_OK && text[0] == '{');
        CHECK(gkr_circom_input_json(&desc, &p, trial, &text[0], need - 1, &need) == GKR_ERR_NOMEM);
        const size_t len = count;
        CHECK(gkr_circom_verifier_source(meta.data(), &len, 1, nullptr, 0, &need) == GKR_OK);
        std::string src(need, '\0');
        CHECK(gkr_circom_verifier_source(meta.data(), &len, 1, &src[0], need, &need) == GKR_OK);
        const char* circuit = "pragma circom 2.0.0;\ntemplate A(){\n    signal input a;\n}\ncomponent main = A();\n";
        CHECK(gkr_circom_inject(circuit, src.c_str(), nullptr, 0, &need) == GKR_OK);
        std::string outc(need, '\0');
        CHECK(gkr_circom_inject(circuit, src.c_str(), &outc[0], need, &need) == GKR_OK && outc.find("VerifyGKR(") != std::string::npos);
        sl[0] = 9;   // malformed length: refused
        if (rounds) CHECK(gkr_circom_meta(&desc, &p, meta.data(), meta.size(), &count) == GKR_ERR_INVALID);
    }
}

static void hashes(std::mt19937_64& rng) {
    uint8_t out[32];
    gkr::keccak256(reinterpret_cast<const uint8_t*>(""), 0, out);
    static const uint8_t empty[4] = {0xc5, 0xd2, 0x46, 0x01};
    CHECK(memcmp(out, empty, 4) == 0);
    std::vector<uint8_t> msg(500);
    for (size_t n = 0; n < msg.size(); n += 17) gkr::keccak256(msg.data(), n, out);
    gkr::Fr cts[gkr::kMimcRounds];
    gkr::mimc7_make_constants(cts);
    if (gkr::gkr_adx_available()) {
        // one transcript on mulx / adcx / adox against the portable 4 x 64-bit code, edge values included
        gkr::h64::F c64[gkr::kMimcRounds];
        memcpy(c64, cts, sizeof c64);
        for (int it = 0; it < 200; ++it) {
            gkr::h64::F v[3];
            for (auto& e : v) {
                e = gkr::h64::F{{rng(), rng(), rng(), rng() >> 4}};
                if (it % 7 == 0) e = gkr::h64::F{{0, 0, 0, 0}};
                if (it % 11 == 0) e = gkr::h64::F{{gkr::h64::kMod[0] - 1, gkr::h64::kMod[1], gkr::h64::kMod[2], gkr::h64::kMod[3]}};
            }
            for (int n = 0; n <= 3; ++n) {
                const gkr::h64::F want = gkr::h64::mimc7_multi_hash(v, n, c64, nullptr);
                gkr::h64::F got;
                gkr::gkr_adx_multi_hash(reinterpret_cast<const uint64_t(*)[4]>(v), n, reinterpret_cast<const uint64_t(*)[4]>(c64), got.l);
                CHECK(memcmp(&want, &got, 32) == 0);
            }
        }
    }
    if (gkr::gkr_ifma_available()) {
        static uint64_t canon[gkr::kMimcRounds][4];
        for (int i = 0; i < gkr::kMimcRounds; ++i) {
            gkr::h64::F m;
            memcpy(&m, &cts[i], 32);
            const gkr::h64::F c = gkr::h64::from_mont(m);
            memcpy(canon[i], &c, 32);
        }
        gkr::gkr_ifma_init(canon);
        uint64_t v[16][3][4], o8[8][4], o16[16][4];
        uint32_t len[16];
        for (int k = 0; k < 16; ++k) {
            len[k] = rng() % 4;
            for (int s = 0; s < 3; ++s) {
                v[k][s][0] = rng();
                v[k][s][1] = rng();
                v[k][s][2] = rng();
                v[k][s][3] = rng() >> 4;
            }
        }
        gkr::gkr_ifma_multi_hash8(v, len, 3, o8);
        gkr::gkr_ifma_multi_hash16(v, len, 3, o16);
        CHECK(memcmp(o8, o16, sizeof o8) == 0);   // the first eight lanes agree between the two forms
        // the whole-pass form: heap buffers of exactly the advertised size, every lane count, so that an out-of-range
        // gather or scatter of a partially filled lane group is an ASan report; round 0's challenge against the hash above
        for (int count = 1; count <= 16; ++count) {
            const int J = 1 + count % 5;
            std::vector<uint64_t> sums((size_t)count * 128), w((size_t)count * 128);
            for (size_t i = 0; i < sums.size(); ++i) sums[i] = (i % 4 == 3) ? rng() >> 4 : rng();
            std::vector<uint64_t> c0(5 * 16 * 4), c1(5 * 16 * 4), r(5 * 16 * 4);
            std::vector<uint32_t> ln(5 * 16);
            gkr::gkr_ifma_pass(sums.data(), 128, count, J, nullptr, reinterpret_cast<uint64_t(*)[16][4]>(c0.data()),
                               reinterpret_cast<uint64_t(*)[16][4]>(c1.data()), reinterpret_cast<uint64_t(*)[16][4]>(r.data()),
                               reinterpret_cast<uint32_t(*)[16]>(ln.data()), w.data(), 128);
            uint64_t hv[8][3][4], ho[8][4];
            uint32_t hl[8] = {};
            memset(hv, 0, sizeof hv);
            const int lanes = count < 8 ? count : 8;
            for (int k = 0; k < lanes; ++k) {
                memcpy(hv[k][1], &c1[(size_t)k * 4], 32);
                memcpy(hv[k][2], &c0[(size_t)k * 4], 32);
                hl[k] = ln[k];
            }
            gkr::gkr_ifma_multi_hash8(hv, hl, 3, ho);
            for (int k = 0; k < lanes; ++k) CHECK(memcmp(ho[k], &r[(size_t)k * 4], 32) == 0);
        }
        // the product pass of the layer sumcheck: records of exactly 72 values per lane, every lane count and depth
        for (int count = 1; count <= 16; ++count) {
            const int J = 1 + count % 3;
            std::vector<uint64_t> recs((size_t)count * 72 * 4), w((size_t)count * 32);
            for (size_t i = 0; i < recs.size(); ++i) recs[i] = (i % 4 == 3) ? rng() >> 4 : rng();
            std::vector<uint64_t> c2(3 * 16 * 4), lin(3 * 16 * 4), c0(3 * 16 * 4), r(3 * 16 * 4);
            uint32_t vl[3][16];
            for (int t = 0; t < 3; ++t)
                for (int k = 0; k < 16; ++k) vl[t][k] = 2 + (uint32_t)(rng() & 1);
            gkr::gkr_ifma_prod_pass(recs.data(), 72 * 4, count, J, vl, reinterpret_cast<uint64_t(*)[16][4]>(c2.data()),
                                    reinterpret_cast<uint64_t(*)[16][4]>(lin.data()), reinterpret_cast<uint64_t(*)[16][4]>(c0.data()),
                                    reinterpret_cast<uint64_t(*)[16][4]>(r.data()), w.data(), 32);
            // round 0's challenge against the plain lane hash of the same vector
            uint64_t hv[8][3][4], ho[8][4];
            uint32_t hl[8] = {};
            memset(hv, 0, sizeof hv);
            const int lanes = count < 8 ? count : 8;
            for (int k = 0; k < lanes; ++k) {
                memcpy(hv[k][0], &c2[(size_t)k * 4], 32);
                memcpy(hv[k][1], &lin[(size_t)k * 4], 32);
                memcpy(hv[k][2], &c0[(size_t)k * 4], 32);
                hl[k] = vl[0][k];
            }
            gkr::gkr_ifma_multi_hash8(hv, hl, 3, ho);
            for (int k = 0; k < lanes; ++k) CHECK(memcmp(ho[k], &r[(size_t)k * 4], 32) == 0);
        }
        // the host tail of the product passes on eight lanes: tables of EXACTLY 3 x 2^m entries, weights of exactly 2^jp, a record of
        // exactly 72 -- every fold depth and round count, against the scalar sums of products (fr64.h wide_mac / wide_reduce)
        for (uint32_t m = 1; m <= 9; ++m)
            for (uint32_t jp = 0; jp <= 3 && jp <= m; ++jp)
                for (uint32_t J = 1; J <= 3 && J + jp <= m; ++J) {
                    const size_t len = (size_t)1 << m;
                    std::vector<uint64_t> tabs(3 * len * 4), w((size_t)(1u << jp) * 4), rec(72 * 4, 0);
                    for (size_t i = 0; i < tabs.size(); ++i) tabs[i] = (i % 4 == 3) ? rng() >> 4 : rng();
                    for (size_t i = 0; i < w.size(); ++i) w[i] = (i % 4 == 3) ? rng() >> 4 : rng();
                    std::vector<uint64_t> want_tabs(tabs);
                    gkr::gkr_ifma_tail_pass(tabs.data(), len, m, jp, jp ? w.data() : nullptr, J, rec.data());
                    using gkr::h64::F;
                    F* T = reinterpret_cast<F*>(want_tabs.data());
                    const F* wf = reinterpret_cast<const F*>(w.data());
                    const uint32_t mf = m - jp, flen = 1u << mf, S = flen >> J;
                    for (int t = 0; t < 3 && jp; ++t)
                        for (uint32_t i = 0; i < flen; ++i) {
                            gkr::h64::Wide acc = gkr::h64::wide_zero();
                            for (uint32_t b = 0; b < (1u << jp); ++b) gkr::h64::wide_mac(acc, T[(size_t)t * len + ((size_t)b << mf) + i], wf[b]);
                            T[(size_t)t * len + i] = gkr::h64::wide_reduce(acc);
                        }
                    for (int t = 0; t < 3; ++t) CHECK(memcmp(&T[(size_t)t * len], &tabs[(size_t)t * len * 4], (size_t)flen * 32) == 0);
                    for (uint32_t a = 0; a < (1u << J); ++a)
                        for (uint32_t b = 0; b < (1u << J); ++b) {
                            gkr::h64::Wide acc = gkr::h64::wide_zero();
                            for (uint32_t i = 0; i < S; ++i) gkr::h64::wide_mac(acc, T[a * S + i], T[len + b * S + i]);
                            const F v = gkr::h64::wide_reduce(acc);
                            CHECK(memcmp(&v, &rec[(size_t)(a * 8 + b) * 4], 32) == 0);
                        }
                }
    } else {
        printf("(no AVX-512 IFMA on this CPU: lane hash skipped)\n");
    }
}

int main() {
    std::mt19937_64 rng(20261002);
    r1cs_round_trips(rng);
    circom_text(rng);
    hashes(rng);
    printf("host_sanitize: %s (%d failures)\n", failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}
