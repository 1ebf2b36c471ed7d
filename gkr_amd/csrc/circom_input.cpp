// The artefact right after the hot path: a proof as the input signals of verifier.circom.
//
// The reference turns every Proof into a CircomInputProof (aggregator.rs:20-82), after padding its ragged
// vectors to the dimensions the generated verifier component declares (get_meta aggregator.rs:92-146,
// modify_proof_for_circom :148-213, signal shapes :222-233), prints field elements as decimal strings
// (file_utils.rs:20-28) and merges the seven arrays into the circuit's input JSON under keys suffixed with
// the proof's index (file_utils.rs:49-67).  Host-only code: no device, no context.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/gkr_amd.h"

namespace {

// canonical 256-bit value -> decimal (stringify_fr: to_repr bytes -> BigInt -> radix 10)
std::string decimal(const gkr_fr& x) {
    uint64_t w[4] = {x.l[0], x.l[1], x.l[2], x.l[3]};
    if (!(w[0] | w[1] | w[2] | w[3])) return "0";
    std::string out;
    const uint64_t base = 10000000000000000000ull;   // 10^19
    while (w[0] | w[1] | w[2] | w[3]) {
        unsigned __int128 rem = 0;
        for (int i = 3; i >= 0; --i) {
            const unsigned __int128 cur = (rem << 64) | w[i];
            w[i] = (uint64_t)(cur / base);
            rem = cur % base;
        }
        uint64_t r = (uint64_t)rem;
        const bool last = !(w[0] | w[1] | w[2] | w[3]);
        for (int d = 0; d < 19 && (!last || r); ++d) {
            out.push_back((char)('0' + r % 10));
            r /= 10;
        }
    }
    return std::string(out.rbegin(), out.rend());
}

bool is_zero(const gkr_fr& x) { return !(x.l[0] | x.l[1] | x.l[2] | x.l[3]); }

struct View {
    uint32_t L = 0;
    std::vector<uint32_t> k;                  // L + 1
    const gkr_proof_buf* p = nullptr;
    std::vector<size_t> row0, q0, z0;         // per-layer offsets into the flat buffers
    bool ok = false;
};

View view(const gkr_circuit_desc* c, const gkr_proof_buf* p) {
    View v;
    if (!c || !p || !c->k || c->depth == 0 || c->depth > 4096) return v;
    if (!p->sumcheck_coeffs || !p->sumcheck_len || !p->sumcheck_r || !p->q || !p->q_len || !p->z || !p->r || !p->d_coeffs ||
        !p->input_coeffs)
        return v;
    v.L = c->depth;
    v.k.assign(c->k, c->k + v.L + 1);
    for (uint32_t ki : v.k)
        if (ki > 30) return v;
    size_t row = 0, q = 0, z = 0;
    for (uint32_t i = 0; i < v.L; ++i) {
        v.row0.push_back(row);
        v.q0.push_back(q);
        row += 2 * (size_t)v.k[i + 1];
        q += (size_t)v.k[i + 1] + 1;
        if (p->q_len[i] > v.k[i + 1] + 1) return v;
    }
    for (uint32_t i = 0; i <= v.L; ++i) {
        v.z0.push_back(z);
        z += v.k[i];
    }
    for (size_t r = 0; r < row; ++r)
        if (p->sumcheck_len[r] < 1 || p->sumcheck_len[r] > 3) return v;
    v.p = p;
    v.ok = true;
    return v;
}

size_t count_nonzero(const gkr_fr* a, size_t n) {
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += is_zero(a[i]) ? 0 : 1;
    return c;
}

// get_meta, aggregator.rs:92-146
std::vector<uint32_t> meta_of(const View& v) {
    std::vector<uint32_t> m;
    uint32_t largest_k = 0;
    for (uint32_t ki : v.k) largest_k = ki > largest_k ? ki : largest_k;
    uint32_t largest_round = 0, largest_q = 0;
    for (uint32_t i = 0; i < v.L; ++i) {
        for (size_t j = 0; j < 2 * (size_t)v.k[i + 1]; ++j) {
            const uint32_t len = v.p->sumcheck_len[v.row0[i] + j];
            largest_round = len > largest_round ? len : largest_round;
        }
        largest_q = v.p->q_len[i] > largest_q ? v.p->q_len[i] : largest_q;
    }
    m.push_back(v.L + 1);                                                          // [0] depth
    m.push_back(largest_k);                                                        // [1]
    m.push_back(v.k[0]);                                                           // [2] k_i(0)
    m.push_back((uint32_t)count_nonzero(v.p->d_coeffs, (size_t)1 << v.k[0]));      // [3] terms of D
    m.push_back(largest_round);                                                    // [4]
    m.push_back(largest_q);                                                        // [5]
    m.push_back((uint32_t)count_nonzero(v.p->input_coeffs, (size_t)1 << v.k[v.L])); // [6] terms of w_d
    m.push_back(v.k[v.L]);                                                         // [7] k_i(d - 1)
    for (uint32_t ki : v.k) m.push_back(ki);
    return m;
}

void put(std::string& s, const gkr_fr& x) {
    s.push_back('"');
    s += decimal(x);
    s.push_back('"');
}
void put_zero(std::string& s) { s += "\"0\""; }

// term list of a monomial-coefficient table: [coeff, e_1 .. e_k], variable 1 = most significant index bit
void put_terms(std::string& s, const gkr_fr* coeffs, uint32_t k) {
    s.push_back('[');
    bool first = true;
    for (size_t m = 0; m < ((size_t)1 << k); ++m) {
        if (is_zero(coeffs[m])) continue;
        if (!first) s.push_back(',');
        first = false;
        s.push_back('[');
        put(s, coeffs[m]);
        for (uint32_t j = 0; j < k; ++j) s += ((m >> (k - 1 - j)) & 1) ? ",\"1\"" : ",\"0\"";
        s.push_back(']');
    }
    s.push_back(']');
}

std::string json_of(const View& v, int index) {
    const std::vector<uint32_t> m = meta_of(v);
    const uint32_t largest_k = m[1], w_round = m[4], w_q = m[5];
    const std::string n = std::to_string(index);
    std::string s = "{";
    // sumcheckProof[d-1][2 largest_k][meta4]: round vectors left-padded, missing rounds zero vectors
    s += "\"sumcheckProof" + n + "\":[";
    for (uint32_t i = 0; i < v.L; ++i) {
        if (i) s.push_back(',');
        s.push_back('[');
        for (size_t j = 0; j < 2 * (size_t)largest_k; ++j) {
            if (j) s.push_back(',');
            s.push_back('[');
            const bool real = j < 2 * (size_t)v.k[i + 1];
            const uint32_t len = real ? v.p->sumcheck_len[v.row0[i] + j] : 0;
            for (uint32_t t = 0; t < w_round; ++t) {
                if (t) s.push_back(',');
                if (t < w_round - len)
                    put_zero(s);
                else   // the row holds its `len` coefficients right-aligned in 3 slots
                    put(s, v.p->sumcheck_coeffs[(v.row0[i] + j) * 3 + (3 - len) + (t - (w_round - len))]);
            }
            s.push_back(']');
        }
        s.push_back(']');
    }
    // sumcheckr[d-1][2 largest_k]: right-padded
    s += "],\"sumcheckr" + n + "\":[";
    for (uint32_t i = 0; i < v.L; ++i) {
        if (i) s.push_back(',');
        s.push_back('[');
        for (size_t j = 0; j < 2 * (size_t)largest_k; ++j) {
            if (j) s.push_back(',');
            if (j < 2 * (size_t)v.k[i + 1])
                put(s, v.p->sumcheck_r[v.row0[i] + j]);
            else
                put_zero(s);
        }
        s.push_back(']');
    }
    // q[d-1][meta5]: left-padded (the buffer holds q right-aligned in k+1 slots)
    s += "],\"q" + n + "\":[";
    for (uint32_t i = 0; i < v.L; ++i) {
        if (i) s.push_back(',');
        s.push_back('[');
        const uint32_t len = v.p->q_len[i], slots = v.k[i + 1] + 1;
        for (uint32_t t = 0; t < w_q; ++t) {
            if (t) s.push_back(',');
            if (t < w_q - len)
                put_zero(s);
            else
                put(s, v.p->q[v.q0[i] + (slots - len) + (t - (w_q - len))]);
        }
        s.push_back(']');
    }
    s += "],\"D" + n + "\":";
    put_terms(s, v.p->d_coeffs, v.k[0]);
    // z[d][largest_k]: right-padded
    s += ",\"z" + n + "\":[";
    for (uint32_t i = 0; i <= v.L; ++i) {
        if (i) s.push_back(',');
        s.push_back('[');
        for (uint32_t j = 0; j < largest_k; ++j) {
            if (j) s.push_back(',');
            if (j < v.k[i])
                put(s, v.p->z[v.z0[i] + j]);
            else
                put_zero(s);
        }
        s.push_back(']');
    }
    s += "],\"r" + n + "\":[";
    for (uint32_t i = 0; i < v.L; ++i) {
        if (i) s.push_back(',');
        put(s, v.p->r[i]);
    }
    s += "],\"inputFunc" + n + "\":";
    put_terms(s, v.p->input_coeffs, v.k[v.L]);
    s.push_back('}');
    return s;
}

// ---- the circom source the reference injects into the user's circuit (modify_circom_file, aggregator.rs:215-314) ----
// One `component verifier[total];` declaration, then per proof: the VerifyGKR instance, its seven input signals with
// the dimensions get_meta computed, and the loops that wire the circuit's new inputs to the component.  The text is
// built from two small tables (signals and their wiring loops); the reference renders the same text from a tera
// template (:219-269) -- `tests/test_circom_input.py` renders that template independently and compares.
struct SignalDim {
    const char* name;
    std::vector<std::string> dims;    // declaration: [dim]...
    std::vector<std::string> bounds;  // wiring loops, outermost first: for (var i = 0; i < bound; i++)
};

std::string verifier_block(const std::vector<uint32_t>& m, size_t num) {
    const std::string n = std::to_string(num);
    auto M = [&](int i) { return std::to_string(m[i]); };
    const std::string d = "d" + n, lk = "largest_k" + n, a = "a" + n;
    const std::vector<SignalDim> sig = {
        {"sumcheckProof", {d + " - 1", "2 * " + lk, M(4)}, {a, "2 * " + M(1), M(4)}},
        {"sumcheckr", {d + " - 1", "2 * " + lk}, {a, "2 * " + M(1)}},
        {"q", {d + " - 1", M(5)}, {a, M(5)}},
        {"D", {M(3), M(2) + " + 1"}, {M(3), M(2) + " + 1"}},
        {"z", {d, lk}, {a + " + 1", M(1)}},
        {"r", {d + " - 1"}, {a}},
        {"inputFunc", {M(6), M(7) + " + 1"}, {M(6), M(7) + " + 1"}},
    };
    std::string meta_dbg = "[";   // format!("{:?}", Vec<usize>)
    for (size_t i = 0; i < m.size(); ++i) meta_dbg += (i ? ", " : "") + std::to_string(m[i]);
    meta_dbg += "]";
    std::string s = "\n";
    const std::string ind = "    ";
    s += ind + "var " + d + " = " + M(0) + ";\n";
    s += ind + "var " + lk + " = " + M(1) + ";\n";
    for (const SignalDim& g : sig) {
        s += ind + "signal input " + g.name + n;
        for (const std::string& dim : g.dims) s += "[" + dim + "]";
        s += ";\n";
    }
    s += ind + "verifier[" + n + "] = VerifyGKR(" + meta_dbg + ");\n";
    s += ind + "var " + a + " = " + M(0) + " - 1;\n";
    const char* iv[3] = {"i", "j", "k"};
    for (const SignalDim& g : sig) {
        std::string idx;
        for (size_t l = 0; l < g.bounds.size(); ++l) {
            s += ind;
            for (size_t t = 0; t < l; ++t) s += ind;
            s += std::string("for (var ") + iv[l] + " = 0; " + iv[l] + " < " + g.bounds[l] + "; " + iv[l] + "++) {\n";
            idx += std::string("[") + iv[l] + "]";
        }
        s += ind;
        for (size_t t = 0; t < g.bounds.size(); ++t) s += ind;
        s += "verifier[" + n + "]." + g.name + idx + " <== " + g.name + n + idx + ";\n";
        for (size_t l = g.bounds.size(); l-- > 0;) {
            s += ind;
            for (size_t t = 0; t < l; ++t) s += ind;
            s += "}\n";
        }
    }
    s += ind;
    return s;
}

int copy_text(const std::string& s, char* out, size_t capacity, size_t* needed) {
    if (!needed) return GKR_ERR_INVALID;
    *needed = s.size() + 1;
    if (!out) return GKR_OK;
    if (capacity < s.size() + 1) return GKR_ERR_NOMEM;
    memcpy(out, s.c_str(), s.size() + 1);
    return GKR_OK;
}

}  // namespace

extern "C" {

// metas: the proofs' meta vectors one after the other, meta_len[i] entries each (gkr_circom_meta's output)
int gkr_circom_verifier_source(const uint32_t* metas, const size_t* meta_len, size_t proofs, char* out, size_t capacity, size_t* needed) {
    if ((!metas || !meta_len) && proofs) return GKR_ERR_INVALID;
    std::string v = "\n    component verifier[" + std::to_string(proofs) + "];\n    ";
    size_t off = 0;
    for (size_t i = 0; i < proofs; ++i) {
        if (meta_len[i] < 9) return GKR_ERR_INVALID;   // depth, ..., k_i(d - 1), and at least one k
        const std::vector<uint32_t> m(metas + off, metas + off + meta_len[i]);
        off += meta_len[i];
        v += "\n" + verifier_block(m, i);             // v = format!("{}\n{}", v, s), aggregator.rs:289
    }
    return copy_text(v, out, capacity, needed);
}

// The user's circuit with the verifier source injected (aggregator.rs:292-309): the exact line
// `pragma circom 2.0.0;` is followed by the include of verifier.circom (and whatever came before that line is
// dropped, as in the reference, which reassigns the text there), the FIRST line that is exactly `}` is preceded by
// the verifier source, every other line is kept.
int gkr_circom_inject(const char* circuit_text, const char* verifier_source, char* out, size_t capacity, size_t* needed) {
    if (!circuit_text || !verifier_source) return GKR_ERR_INVALID;
    std::string text(circuit_text), result;
    bool added = false;
    size_t pos = 0;
    while (pos < text.size()) {   // str::lines(): split at \n, a trailing \r is dropped, no empty last line
        size_t end = text.find('\n', pos);
        const bool last = end == std::string::npos;
        if (last) end = text.size();
        std::string line = text.substr(pos, end - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = last ? text.size() : end + 1;
        if (line == "pragma circom 2.0.0;") {
            result = line + "\ninclude \"../gkr-verifier-circuits/circom/circom/verifier.circom\";\n";
        } else if (line == "}" && !added) {
            result = result + "\n" + verifier_source + "\n}";
            added = true;
        } else {
            result += line + "\n";
        }
    }
    return copy_text(result, out, capacity, needed);
}

int gkr_circom_meta(const gkr_circuit_desc* circuit, const gkr_proof_buf* proof, uint32_t* meta, size_t capacity, size_t* count) {
    const View v = view(circuit, proof);
    if (!v.ok || !count) return GKR_ERR_INVALID;
    const std::vector<uint32_t> m = meta_of(v);
    *count = m.size();
    if (!meta || capacity < m.size()) return meta ? GKR_ERR_NOMEM : GKR_OK;
    memcpy(meta, m.data(), m.size() * sizeof(uint32_t));
    return GKR_OK;
}

int gkr_circom_input_json(const gkr_circuit_desc* circuit, const gkr_proof_buf* proof, int proof_index, char* out, size_t capacity,
                          size_t* needed) {
    const View v = view(circuit, proof);
    if (!v.ok || !needed || proof_index < 0) return GKR_ERR_INVALID;
    const std::string s = json_of(v, proof_index);
    *needed = s.size() + 1;
    if (!out) return GKR_OK;
    if (capacity < s.size() + 1) return GKR_ERR_NOMEM;
    memcpy(out, s.c_str(), s.size() + 1);
    return GKR_OK;
}

}  // extern "C"
