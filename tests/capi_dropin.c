/* A plain-C host driving the library the way a binding of prover::prove would (`prove_amd`): the toy circuit
 * of the reference's own test (python/test_gkr.py:7-112), handed over in the types prove(&GKRCircuit, &Input) receives --
 * per layer the 0/1 wire vectors `gate || left || right` (Layer.wire, rust/src/gkr.rs:35-51; here the very vectors
 * test_gkr.py's multlayerzero / multlayerone accept) and the input layer as a term list [coeff, e_1, e_2] (Input.w,
 * gkr.rs:21-33) -- proven with gkr_prove, checked with gkr_verify, printed as one JSON object with decimal strings
 * (file_utils.rs:20-28).  tests/test_gpu_capi_dropin.py builds this with gcc (no Python, no C++ in the host program), links
 * -lgkr_amd and compares the output with tests/golden/gkr_circuits.json[test_gkr_toy_z0_zero].
 *
 *   gcc -std=c99 -I include tests/capi_dropin.c -L gkr_amd/lib -lgkr_amd -Wl,-rpath,$PWD/gkr_amd/lib -o capi_dropin
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gkr_amd.h"

/* ---- decimal strings <-> gkr_fr (4 little-endian 64-bit limbs) ---- */
static int fr_from_decimal(const char *s, gkr_fr *out) {
    memset(out, 0, sizeof(*out));
    for (; *s; ++s) {
        if (*s < '0' || *s > '9') return -1;
        unsigned __int128 carry = (unsigned)(*s - '0');
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 t = (unsigned __int128)out->l[i] * 10 + carry;
            out->l[i] = (uint64_t)t;
            carry = t >> 64;
        }
        if (carry) return -1;
    }
    return 0;
}

static void fr_to_decimal(const gkr_fr *x, char *buf /* >= 80 bytes */) {
    uint64_t l[4] = {x->l[0], x->l[1], x->l[2], x->l[3]};
    char tmp[80];
    int n = 0;
    do {
        unsigned __int128 rem = 0;
        for (int i = 3; i >= 0; --i) {
            unsigned __int128 cur = (rem << 64) | l[i];
            l[i] = (uint64_t)(cur / 10);
            rem = cur % 10;
        }
        tmp[n++] = (char)('0' + (int)rem);
    } while (l[0] | l[1] | l[2] | l[3]);
    for (int i = 0; i < n; ++i) buf[i] = tmp[n - 1 - i];
    buf[n] = 0;
}

static void print_vec(const gkr_fr *v, size_t n) {
    char buf[80];
    printf("[");
    for (size_t i = 0; i < n; ++i) {
        fr_to_decimal(&v[i], buf);
        printf("%s\"%s\"", i ? ", " : "", buf);
    }
    printf("]");
}

static gkr_fr bit(int b) {
    gkr_fr x = {{(uint64_t)b, 0, 0, 0}};
    return x;
}

#define CHECK(call)                                                                               \
    do {                                                                                          \
        int rc_ = (call);                                                                         \
        if (rc_ != GKR_OK) {                                                                      \
            fprintf(stderr, "%s -> %s (%s)\n", #call, gkr_strerror(rc_), ctx ? gkr_last_error(ctx) : ""); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

int main(void) {
    gkr_ctx *ctx = NULL;
    /* GKRCircuit::get_k_list(): layer 0 has 2 gates over layer 1's 4 values, layer 1 has 4 gates over the 4 inputs */
    const uint32_t k[3] = {1, 2, 2};
    /* Layer.wire.1 (mult) of layers 0 and 1 -- every gate of the toy circuit multiplies; Layer.wire.0 (add) is empty */
    static const int mult0[2][5] = {{0, 0, 0, 0, 1}, {1, 1, 0, 1, 1}};
    static const int mult1[4][6] = {{0, 0, 0, 0, 0, 0}, {0, 1, 0, 1, 0, 1}, {1, 0, 0, 1, 1, 0}, {1, 1, 1, 1, 1, 1}};
    /* Input.w[2]: W2 = 3 - x2 - x1 x2 as get_multi_ext emits it (python/test_gkr.py W2func: 3, 2, 3, 1) */
    static const char *input_terms[3][3] = {
        {"3", "0", "0"},
        {"21888242871839275222246405745257275088548364400416034343698204186575808495616", "0", "1"},
        {"21888242871839275222246405745257275088548364400416034343698204186575808495616", "1", "1"}};

    gkr_fr w0[2 * 5], w1[4 * 6], terms[3 * 3], input_values[4];
    for (int g = 0; g < 2; ++g)
        for (int j = 0; j < 5; ++j) w0[g * 5 + j] = bit(mult0[g][j]);
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 6; ++j) w1[g * 6 + j] = bit(mult1[g][j]);
    for (int t = 0; t < 3; ++t)
        for (int j = 0; j < 3; ++j)
            if (fr_from_decimal(input_terms[t][j], &terms[t * 3 + j])) return 2;

    uint8_t type0[2], type1[4];
    uint32_t left0[2], right0[2], left1[4], right1[4];
    CHECK(gkr_layer_from_wires(1, 2, NULL, 0, w0, 2, type0, left0, right0));
    CHECK(gkr_layer_from_wires(2, 2, NULL, 0, w1, 4, type1, left1, right1));
    CHECK(gkr_values_from_terms(2, terms, 3, input_values));

    const uint8_t *types[2] = {type0, type1};
    const uint32_t *lefts[2] = {left0, left1}, *rights[2] = {right0, right1};
    gkr_circuit_desc circuit = {2, k, types, lefts, rights};
    gkr_proof_sizes_t sz;
    CHECK(gkr_proof_sizes(&circuit, &sz));

    gkr_proof_buf proof;
    uint32_t q_len[2];
    proof.sumcheck_coeffs = calloc(sz.rounds * 3, sizeof(gkr_fr));
    proof.sumcheck_len = calloc(sz.rounds, sizeof(uint32_t));
    proof.sumcheck_r = calloc(sz.rounds, sizeof(gkr_fr));
    proof.q = calloc(sz.q_slots, sizeof(gkr_fr));
    proof.q_len = q_len;
    proof.z = calloc(sz.z_values, sizeof(gkr_fr));
    proof.r = calloc(circuit.depth, sizeof(gkr_fr));
    proof.d_coeffs = calloc(sz.d_coeffs, sizeof(gkr_fr));
    proof.input_coeffs = calloc(sz.input_coeffs, sizeof(gkr_fr));

    CHECK(gkr_ctx_create(0, &ctx));
    /* require_zero_output = 0: the toy circuit's output 0 is 36 (the assertion of convert.rs:838 is the compiler's) */
    CHECK(gkr_prove(ctx, &circuit, input_values, 0, &proof));

    int accept = 0;
    uint32_t bad_layer = 0, bad_check = 0;
    CHECK(gkr_verify(&circuit, &proof, 1, &accept, &bad_layer, &bad_check));

    /* the same proof through the ONE-call form on the reference's types (gkr_prove_wires): must give the same bytes */
    {
        const gkr_fr *adds[2] = {NULL, NULL}, *mults[2] = {w0, w1};
        const size_t n_add[2] = {0, 0}, n_mult[2] = {2, 4};
        gkr_wire_circuit wc = {2, k, adds, n_add, mults, n_mult};
        gkr_proof_buf again;
        uint32_t q_len2[2];
        again.sumcheck_coeffs = calloc(sz.rounds * 3, sizeof(gkr_fr));
        again.sumcheck_len = calloc(sz.rounds, sizeof(uint32_t));
        again.sumcheck_r = calloc(sz.rounds, sizeof(gkr_fr));
        again.q = calloc(sz.q_slots, sizeof(gkr_fr));
        again.q_len = q_len2;
        again.z = calloc(sz.z_values, sizeof(gkr_fr));
        again.r = calloc(circuit.depth, sizeof(gkr_fr));
        again.d_coeffs = calloc(sz.d_coeffs, sizeof(gkr_fr));
        again.input_coeffs = calloc(sz.input_coeffs, sizeof(gkr_fr));
        CHECK(gkr_prove_wires(ctx, &wc, terms, 3, 0, &again));
        if (memcmp(again.sumcheck_coeffs, proof.sumcheck_coeffs, sz.rounds * 3 * sizeof(gkr_fr)) || memcmp(again.sumcheck_r, proof.sumcheck_r, sz.rounds * sizeof(gkr_fr)) ||
            memcmp(again.q, proof.q, sz.q_slots * sizeof(gkr_fr)) || memcmp(again.z, proof.z, sz.z_values * sizeof(gkr_fr)) ||
            memcmp(again.r, proof.r, circuit.depth * sizeof(gkr_fr)) || memcmp(again.d_coeffs, proof.d_coeffs, sz.d_coeffs * sizeof(gkr_fr)) ||
            memcmp(again.input_coeffs, proof.input_coeffs, sz.input_coeffs * sizeof(gkr_fr)) || q_len2[0] != q_len[0] || q_len2[1] != q_len[1]) {
            fprintf(stderr, "gkr_prove_wires and gkr_prove disagree\n");
            return 4;
        }
    }

    /* ---- the Proof of gkr.rs:7-19 as JSON ---- */
    printf("{\"input_values\": ");
    print_vec(input_values, 4);
    printf(", \"gates\": [");
    for (int i = 0; i < 2; ++i) {
        printf("%s{\"gate_type\": [", i ? ", " : "");
        for (uint32_t g = 0; g < (1u << k[i]); ++g) printf("%s%u", g ? ", " : "", types[i][g]);
        printf("], \"left\": [");
        for (uint32_t g = 0; g < (1u << k[i]); ++g) printf("%s%u", g ? ", " : "", lefts[i][g]);
        printf("], \"right\": [");
        for (uint32_t g = 0; g < (1u << k[i]); ++g) printf("%s%u", g ? ", " : "", rights[i][g]);
        printf("]}");
    }
    printf("], \"sumcheck_proofs\": [");
    size_t row = 0, qo = 0, zo = 0;
    for (uint32_t i = 0; i < circuit.depth; ++i) {
        printf("%s[", i ? ", " : "");
        for (uint32_t j = 0; j < 2 * k[i + 1]; ++j, ++row) {
            const uint32_t len = proof.sumcheck_len[row];
            printf("%s", j ? ", " : "");
            print_vec(proof.sumcheck_coeffs + row * 3 + (3 - len), len);   /* right-aligned: the last `len` slots */
        }
        printf("]");
    }
    printf("], \"sumcheck_r\": [");
    row = 0;
    for (uint32_t i = 0; i < circuit.depth; ++i) {
        printf("%s", i ? ", " : "");
        print_vec(proof.sumcheck_r + row, 2 * k[i + 1]);
        row += 2 * k[i + 1];
    }
    printf("], \"q\": [");
    for (uint32_t i = 0; i < circuit.depth; ++i) {
        printf("%s", i ? ", " : "");
        print_vec(proof.q + qo + (k[i + 1] + 1 - q_len[i]), q_len[i]);
        qo += k[i + 1] + 1;
    }
    printf("], \"z\": [");
    for (uint32_t i = 0; i <= circuit.depth; ++i) {
        printf("%s", i ? ", " : "");
        print_vec(proof.z + zo, k[i]);
        zo += k[i];
    }
    printf("], \"r\": ");
    print_vec(proof.r, circuit.depth);
    /* Proof.d and Proof.input_func: term lists of the non-zero monomials */
    for (int which = 0; which < 2; ++which) {
        const int kk = (int)(which ? k[circuit.depth] : k[0]);
        const gkr_fr *coeffs = which ? proof.input_coeffs : proof.d_coeffs;
        size_t n = 0;
        CHECK(gkr_terms_from_coeffs(kk, coeffs, NULL, 0, &n));
        gkr_fr *rows = calloc(n ? n * (size_t)(kk + 1) : 1, sizeof(gkr_fr));
        CHECK(gkr_terms_from_coeffs(kk, coeffs, rows, n, &n));
        printf(", \"%s\": [", which ? "input_func" : "d");
        for (size_t t = 0; t < n; ++t) {
            printf("%s", t ? ", " : "");
            print_vec(rows + t * (size_t)(kk + 1), (size_t)kk + 1);
        }
        printf("]");
        free(rows);
    }
    printf(", \"depth\": %u, \"k\": [%u, %u, %u], \"verifier_accepts\": %s, \"failed_check\": %u}\n", circuit.depth + 1, k[0], k[1], k[2],
           accept ? "true" : "false", bad_check);
    gkr_ctx_destroy(ctx);
    return accept ? 0 : 3;
}
