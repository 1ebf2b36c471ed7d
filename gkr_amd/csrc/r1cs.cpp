// The step in front of the hot path: R1CS + witness -> the <= 20 layered GKR circuits prover::prove is called
// on (rust/src/convert.rs).  Host-only code: no device, no context.
//
//   reference                                             here
//   ----------------------------------------------------  -------------------------------------------------
//   R1csFile::<32>::read / WtnsFile::<32>::read           gkr_r1cs_parse / gkr_wtns_parse (+ _build / _serialize:
//     (third-party r1cs-file / wtns-file crates,            there is no circom here, fixtures have to be written)
//      aggregator.rs:341,345,399,404)
//   convert_constraints_to_nodes  convert.rs:360-632      constraint_trees()
//   merge_nodes                   convert.rs:108-138      Arena::merge()
//   compile                       convert.rs:154-358      compile_groups()
//   get_k                         convert.rs:140-152      get_k()
//   input layer of calculate_input convert.rs:796-810     gkr_layered_input_values()
//
// The reference holds expression trees as boxed nodes, compares them by deep structural equality
// (IntermediateNode::eq, convert.rs:33-57) and finds a child's slot by a linear scan over the next layer
// (`next_nodes.contains`, :288-303).  Here every tree is interned in one arena (hash-consing): structurally equal
// trees ARE the same node id, so equality is an integer compare, the scan is a hash lookup, and the depth of a
// tree is stored with its node.  Same layering, slot order and gate lists as the reference's procedure.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <map>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/gkr_amd.h"
#include "host_cpus.h"
#include "options.h"

namespace {

constexpr uint64_t kMod[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
constexpr size_t kWidthLimit = 20;   // WIDTH_LIMIT, convert.rs:11

bool fr_eq(const gkr_fr& a, const gkr_fr& b) { return memcmp(&a, &b, sizeof a) == 0; }
bool fr_is_zero(const gkr_fr& a) { return !(a.l[0] | a.l[1] | a.l[2] | a.l[3]); }
bool fr_canonical(const gkr_fr& a) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] < kMod[i]) return true;
        if (a.l[i] > kMod[i]) return false;
    }
    return false;
}
gkr_fr fr_one() { return gkr_fr{{1, 0, 0, 0}}; }
gkr_fr fr_neg(const gkr_fr& a) {   // coeff * (0 - 1), convert.rs:517-518
    if (fr_is_zero(a)) return a;
    gkr_fr d;
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 t = (unsigned __int128)kMod[i] - a.l[i] - (uint64_t)borrow;
        d.l[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
    return d;
}

struct Term {
    gkr_fr coeff;
    uint32_t wire;
};
typedef std::array<std::vector<Term>, 3> Constraint;

// ------------------------------------------------------------------------------------------- little-endian I/O
struct Reader {
    const uint8_t* p;
    size_t len, off = 0;
    bool ok = true;
    bool need(size_t n) {
        if (!ok || n > len - off) ok = false;
        return ok;
    }
    uint32_t u32() {
        uint32_t v = 0;
        if (need(4)) {
            memcpy(&v, p + off, 4);
            off += 4;
        }
        return v;
    }
    uint64_t u64() {
        uint64_t v = 0;
        if (need(8)) {
            memcpy(&v, p + off, 8);
            off += 8;
        }
        return v;
    }
    gkr_fr fr() {
        gkr_fr v = {{0, 0, 0, 0}};
        if (need(32)) {
            memcpy(&v, p + off, 32);
            off += 32;
        }
        return v;
    }
};

struct Writer {
    std::vector<uint8_t> b;
    void u32(uint32_t v) { put(&v, 4); }
    void u64(uint64_t v) { put(&v, 8); }
    void fr(const gkr_fr& v) { put(&v, 32); }
    void put(const void* q, size_t n) {
        const uint8_t* c = static_cast<const uint8_t*>(q);
        b.insert(b.end(), c, c + n);
    }
};

// magic, version, then sections (type u32, size u64, data) in any order: first section of each type wins
bool split_sections(const uint8_t* bytes, size_t len, const char* magic, uint32_t version,
                    std::map<uint32_t, std::pair<const uint8_t*, size_t>>& out) {
    Reader r{bytes, len};
    if (len < 12 || memcmp(bytes, magic, 4) != 0) return false;
    r.off = 4;
    if (r.u32() != version) return false;
    const uint32_t n = r.u32();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t ty = r.u32();
        const uint64_t size = r.u64();
        if (!r.ok || size > len - r.off) return false;
        out.emplace(ty, std::make_pair(bytes + r.off, (size_t)size));
        r.off += size;
    }
    return r.ok;
}

bool field_header_ok(Reader& r) {
    if (r.u32() != 32) return false;   // the reference reads R1csFile::<32> / WtnsFile::<32>
    const gkr_fr prime = r.fr();
    return r.ok && memcmp(&prime, kMod, 32) == 0;
}

}  // namespace

struct gkr_r1cs {
    uint32_t n_wires = 0, n_pub_out = 0, n_pub_in = 0, n_prv_in = 0;
    uint64_t n_labels = 0;
    std::vector<Constraint> constraints;
    std::vector<uint64_t> wire2label;
};

namespace {

// ------------------------------------------------------------------------------------------- expression trees
// Interned nodes: id 0 is the constant 0 (zero_node, convert.rs:93-100).  Twelve bytes per node: the trees of a 262 144-
// constraint R1CS are 1.1 * 10^6 nodes that compile_groups reads in no particular order.
enum Kind : uint32_t { kValue = 0, kVariable = 1, kMult = 2, kAdd = 3 };
struct Node {
    uint32_t left, right;   // gates: child ids; variable: wire in `left`; constant: index into Arena::values in `left`
    uint32_t meta;          // depth << 2 | kind.  IntermediateNode::depth, convert.rs:86-90: a leaf has depth 1
    Kind kind() const { return (Kind)(meta & 3u); }
    uint32_t depth() const { return meta >> 2; }
};

// The arena: every node lives in ONE preallocated array (its capacity is an upper bound known from the R1CS; only what is
// used is ever touched), a node's id is its index.  Interning goes through kShards open-addressing tables of ids (shard = top
// bits of the node's hash), each behind its own spinlock, and every building thread keeps a small direct-mapped cache of
// ids in front of them (the hot nodes -- a circuit's constants, the variables of neighbouring constraints -- never reach a
// lock), so constraint_trees can build the trees of different constraints on different threads.  Round 6: the
// std::unordered_map of 56-byte keys this replaces took 0.9 s for those 1.1 * 10^6 nodes, on one thread.  Ids depend on the
// threads' timing; nothing downstream does: structurally equal trees still ARE one id, and compile_groups orders by
// position in its lists, never by id.
struct Arena {
    static constexpr uint32_t kShardBits = 8, kShards = 1u << kShardBits;
    static constexpr uint32_t kCacheSlots = 1u << 13;
    struct alignas(64) Shard {
        std::atomic_flag lock = ATOMIC_FLAG_INIT;
        std::vector<uint32_t> slots;   // id + 1, 0 = empty; capacity a power of two
        uint32_t used = 0;
    };
    // per building thread: the direct-mapped cache, and the block of fresh ids (and value slots) the thread fills -- ids are
    // handed out kIdBlock at a time, so two threads never write nodes of one cache line (unused ids at a block's end stay holes)
    static constexpr uint32_t kIdBlock = 1024;
    struct Cache {
        std::vector<uint32_t> id;
        uint32_t next_id = 0, end_id = 0, next_value = 0, end_value = 0;
        Cache() : id(kCacheSlots, UINT32_MAX) {}
    };
    std::unique_ptr<Node[]> nodes;
    std::unique_ptr<gkr_fr[]> values;
    size_t capacity = 0, value_capacity = 0;
    std::atomic<uint32_t> count{0}, n_values{0};
    std::unique_ptr<Shard[]> shards;

    Arena(size_t max_nodes, size_t max_values, size_t expected_nodes)
        : nodes(new Node[max_nodes + 1]), values(new gkr_fr[max_values + 1]), capacity(max_nodes + 1), value_capacity(max_values + 1),
          shards(new Shard[kShards]) {
        size_t per = 64;
        while (per * kShards < 2 * expected_nodes) per <<= 1;   // load factor below 1/2 for the expected count; a shard grows on its own
        for (uint32_t i = 0; i < kShards; ++i) shards[i].slots.assign(per, 0u);
        Cache c;
        value(gkr_fr{{0, 0, 0, 0}}, c);                         // id 0 = the constant 0
    }
    size_t size() const { return std::min<size_t>(count.load(std::memory_order_acquire), capacity); }
    static uint64_t mix(uint64_t h) {
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ULL;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBULL;
        return h ^ (h >> 31);
    }
    // kind, left, right identify a gate or a variable; a constant is identified by its value (v != nullptr)
    bool same(uint32_t id, Kind kind, uint32_t left, uint32_t right, const gkr_fr* v) const {
        const Node& n = nodes[id];
        if (n.kind() != kind) return false;
        return v ? fr_eq(values[n.left], *v) : (n.left == left && n.right == right);
    }
    // -> the id of the node structurally equal to the one described, inserting it if there is none; UINT32_MAX: the arena is full
    uint32_t intern(Kind kind, uint32_t left, uint32_t right, uint32_t depth, const gkr_fr* v, Cache& cache) {
        const uint64_t h = v ? mix(v->l[0] ^ mix(v->l[1] ^ mix(v->l[2] ^ mix(v->l[3] + 0x9E3779B97F4A7C15ULL))))
                             : mix((((uint64_t)left << 32) | right) + 0x9E3779B97F4A7C15ULL * ((uint64_t)kind + 1));
        uint32_t& cached = cache.id[(h >> 20) & (kCacheSlots - 1)];
        if (cached != UINT32_MAX && same(cached, kind, left, right, v)) return cached;
        Shard& sh = shards[h >> (64 - kShardBits)];
        while (sh.lock.test_and_set(std::memory_order_acquire)) {
        }
        uint32_t id = UINT32_MAX;
        size_t mask = sh.slots.size() - 1, i = (size_t)h & mask;
        for (;; i = (i + 1) & mask) {
            const uint32_t s = sh.slots[i];
            if (!s) break;
            if (same(s - 1, kind, left, right, v)) {
                id = s - 1;
                break;
            }
        }
        if (id == UINT32_MAX) {
            if (cache.next_id == cache.end_id) {
                cache.next_id = count.fetch_add(kIdBlock, std::memory_order_acq_rel);
                cache.end_id = cache.next_id + kIdBlock;
            }
            if (v && cache.next_value == cache.end_value) {
                cache.next_value = n_values.fetch_add(kIdBlock, std::memory_order_acq_rel);
                cache.end_value = cache.next_value + kIdBlock;
            }
            const uint32_t fresh = cache.next_id;
            const uint32_t vi = v ? cache.next_value : 0;
            if (fresh < capacity && vi < value_capacity) {
                ++cache.next_id;
                if (v) ++cache.next_value;
                if (v) values[vi] = *v;
                nodes[fresh] = Node{v ? vi : left, right, (depth << 2) | (uint32_t)kind};
                id = fresh;
                sh.slots[i] = id + 1;
                if (++sh.used * 2 > sh.slots.size()) {       // (more nodes than expected, or an uneven hash)
                    std::vector<uint32_t> bigger(sh.slots.size() * 2, 0u);
                    const size_t m2 = bigger.size() - 1;
                    for (uint32_t s : sh.slots)
                        if (s) {
                            const Node& n = nodes[s - 1];
                            const gkr_fr* nv = n.kind() == kValue ? &values[n.left] : nullptr;
                            const uint64_t hh = nv ? mix(nv->l[0] ^ mix(nv->l[1] ^ mix(nv->l[2] ^ mix(nv->l[3] + 0x9E3779B97F4A7C15ULL))))
                                                   : mix((((uint64_t)n.left << 32) | n.right) + 0x9E3779B97F4A7C15ULL * ((uint64_t)n.kind() + 1));
                            size_t j = (size_t)hh & m2;
                            while (bigger[j]) j = (j + 1) & m2;
                            bigger[j] = s;
                        }
                    sh.slots.swap(bigger);
                }
            }
        }
        sh.lock.clear(std::memory_order_release);
        if (id != UINT32_MAX) cached = id;
        return id;
    }
    uint32_t value(const gkr_fr& v, Cache& c) { return intern(kValue, 0, 0, 1, &v, c); }
    uint32_t variable(uint32_t wire, Cache& c) { return intern(kVariable, wire, 0, 1, nullptr, c); }
    uint32_t gate(Kind k, uint32_t l, uint32_t r, Cache& c) {
        if (l == UINT32_MAX || r == UINT32_MAX) return UINT32_MAX;
        return intern(k, l, r, std::max(nodes[l].depth(), nodes[r].depth()) + 1, nullptr, c);
    }
    bool is_leaf(uint32_t id) const { return nodes[id].kind() == kValue || nodes[id].kind() == kVariable; }
    // merge_nodes, convert.rs:108-138: pair neighbours with Add; with an odd count the pairs are merged first and
    // the last element is added on the right.  `v` must not be empty (the reference recurses without end there).
    uint32_t merge(const std::vector<uint32_t>& v, Cache& c) {
        if (v.size() == 1) return v[0];
        std::vector<uint32_t> pairs;
        pairs.reserve(v.size() / 2);
        for (size_t i = 0; i + 1 < v.size(); i += 2) pairs.push_back(gate(kAdd, v[i], v[i + 1], c));
        if (v.size() % 2 == 1) return gate(kAdd, merge(pairs, c), v.back(), c);
        return merge(pairs, c);
    }
};

// threads for the compiler's two parallel stages: GKR_COMPILE_THREADS (options.h), else the CPUs this process may use, at most 32
int compile_threads(size_t units) {
    int t = gkr::process_int("GKR_COMPILE_THREADS", 0);
    if (t <= 0) t = gkr::usable_cpus();
    if (t > 32) t = 32;
    if ((size_t)t > units) t = (int)units;
    return t < 1 ? 1 : t;
}

// count_mult, convert.rs:363-379
void count_mult(const std::vector<Term>& v, const gkr_fr& one, const gkr_fr& minus_one, int& a, int& b) {
    a = b = 0;
    for (const Term& t : v) {
        if (fr_eq(t.coeff, one)) {
            b += 1;
        } else if (fr_eq(t.coeff, minus_one)) {
            a += 1;
        } else {
            a += 1;
            b += 1;
        }
    }
}

// Variable(wire) when the coefficient is the unit of this reading (1, or -1 for the negated one), else
// Mult(Value(coeff or -coeff), Variable(wire)): convert.rs:512-542 (A), :554-566 (B), :578-610 (C)
uint32_t term_node(Arena& ar, Arena::Cache& cache, const Term& t, bool negated, const gkr_fr& one, const gkr_fr& minus_one) {
    if (fr_eq(t.coeff, negated ? minus_one : one)) return ar.variable(t.wire, cache);
    // (UINT32_MAX from a full arena passes through)
    return ar.gate(kMult, ar.value(negated ? fr_neg(t.coeff) : t.coeff, cache), ar.variable(t.wire, cache), cache);
}

// convert_constraints_to_nodes, convert.rs:360-632: one tree per constraint, A * B + (-C) or (-A) * B + C by the
// neg flag (:476-485), which picks the reading that needs fewer coefficient products.  The symbol-table shortcut
// (:487-511, :545-553) is dead in the reference -- its only insertion site is commented out (:576) -- so every
// constraint becomes one single-tree group (:625-631).  A constraint with an empty A, B or C sends the reference's
// merge_nodes into unbounded recursion (:619-622, :612): reported as GKR_ERR_UNSUPPORTED.
int constraint_trees(const gkr_r1cs& r, Arena& ar, std::vector<uint32_t>& roots, size_t* bad_constraint) {
    const gkr_fr one = fr_one(), minus_one = fr_neg(one);
    const size_t n = r.constraints.size();
    for (size_t i = 0; i < n; ++i) {
        const Constraint& c = r.constraints[i];
        if (c[0].empty() || c[1].empty() || c[2].empty()) {
            if (bad_constraint) *bad_constraint = i;
            return GKR_ERR_UNSUPPORTED;
        }
    }
    roots.assign(n, 0u);
    std::atomic<bool> full{false};
    auto build = [&](size_t first, size_t last) {
        Arena::Cache cache;
        std::vector<uint32_t> na, nb, nc;
        for (size_t i = first; i < last; ++i) {
            const Constraint& c = r.constraints[i];
            int a0, a1, b0, b1, c0, c1;
            count_mult(c[0], one, minus_one, a0, a1);
            count_mult(c[1], one, minus_one, b0, b1);
            count_mult(c[2], one, minus_one, c0, c1);
            const bool neg = (a0 + b0 + c1) > (a1 + b1 + c0);
            na.clear();
            nb.clear();
            nc.clear();
            for (const Term& t : c[0]) na.push_back(term_node(ar, cache, t, neg, one, minus_one));
            for (const Term& t : c[1]) nb.push_back(term_node(ar, cache, t, false, one, minus_one));   // B is never negated
            for (const Term& t : c[2]) nc.push_back(term_node(ar, cache, t, !neg, one, minus_one));
            const uint32_t a_times_b = ar.gate(kMult, ar.merge(na, cache), ar.merge(nb, cache), cache);
            roots[i] = ar.gate(kAdd, a_times_b, ar.merge(nc, cache), cache);
            if (roots[i] == UINT32_MAX) {
                full.store(true);
                return;
            }
        }
    };
    const int threads = compile_threads(n / 2048 + 1);
    if (threads <= 1) {
        build(0, n);
    } else {
        std::vector<std::thread> th;
        const size_t per = (n + threads - 1) / threads;
        for (int t = 1; t < threads; ++t) th.emplace_back(build, std::min(n, t * per), std::min(n, (t + 1) * per));
        build(0, std::min(n, per));
        for (auto& x : th) x.join();
    }
    if (full.load()) return GKR_ERR_NOMEM;   // (the capacity bound of gkr_r1cs_compile was wrong: never seen)
    return GKR_OK;
}

// get_k, convert.rs:140-152
uint32_t get_k(size_t n) {
    uint32_t k = 0;
    for (size_t m = n; m > 1; m >>= 1) ++k;
    return (n & (n - 1)) == 0 ? k : k + 1;
}

struct LayeredCircuit {
    std::vector<uint32_t> k;                          // L + 1
    std::vector<std::vector<uint8_t>> gate_type;      // L layers, 0 = Add, 1 = Mult (the C ABI's encoding)
    std::vector<std::vector<uint32_t>> left, right;
    std::vector<uint32_t> input_wire;                 // UINT32_MAX: a constant
    std::vector<gkr_fr> input_const;
    // pointer tables for gkr_circuit_desc
    std::vector<const uint8_t*> p_gate_type;
    std::vector<const uint32_t*> p_left, p_right;
};

// compile, convert.rs:154-358
int compile_groups(const Arena& ar, const std::vector<uint32_t>& roots, std::vector<LayeredCircuit>& out) {
    // Every constraint is a group of one tree (constraint_trees).  The groups are sorted by tree depth, stably (:164-169) -- a
    // counting sort over the depths -- and then neighbouring groups are concatenated until at most WIDTH_LIMIT remain, an odd
    // last one kept as it is (:171-186): a group is always a RANGE of the sorted list, so the merge works on range ends.
    uint32_t max_depth = 0;
    for (uint32_t id : roots) max_depth = std::max(max_depth, ar.nodes[id].depth());
    std::vector<uint32_t> sorted(roots.size());
    {
        std::vector<size_t> at(max_depth + 2, 0);
        for (uint32_t id : roots) ++at[ar.nodes[id].depth() + 1];
        for (uint32_t d = 0; d <= max_depth; ++d) at[d + 1] += at[d];
        for (uint32_t id : roots) sorted[at[ar.nodes[id].depth()]++] = id;
    }
    std::vector<size_t> ends(roots.size());        // group g = sorted[ends[g - 1] .. ends[g])
    for (size_t i = 0; i < roots.size(); ++i) ends[i] = i + 1;
    while (ends.size() > kWidthLimit) {
        std::vector<size_t> merged;
        merged.reserve(ends.size() / 2 + 1);
        for (size_t i = 0; i + 1 < ends.size(); i += 2) merged.push_back(ends[i + 1]);
        if (ends.size() % 2 == 1) merged.push_back(ends.back());
        ends.swap(merged);
    }
    std::vector<std::vector<uint32_t>> groups(ends.size());
    for (size_t g = 0; g < ends.size(); ++g) groups[g].assign(sorted.begin() + (g ? ends[g - 1] : 0), sorted.begin() + ends[g]);
    auto height_of = [&](const std::vector<uint32_t>& g) {
        uint32_t h = 0;
        for (uint32_t id : g) h = std::max(h, ar.nodes[id].depth());
        return h;
    };
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    // The groups compile independently (round 6: on the library's threads, one group at a time each).  A layer's two lookups
    // -- "which slot of the next layer holds this child" (`next_nodes.contains`, :288-303) and "was this leaf relayed already"
    // (`used`, :202) -- are arrays indexed by node id with a stamp per (group, layer) instead of hash maps: the maps' node
    // allocations were 2.2 s of a 262 144-constraint compile, the arrays cost 16 bytes per tree node and thread.
    const size_t n_nodes = ar.size();
    out.resize(groups.size());
    std::vector<int> status(groups.size(), GKR_OK);
    std::atomic<size_t> next_group{0};
    auto worker = [&]() {
        struct Seen {
            uint32_t slot_stamp, slot, relay_stamp, relay;
        };
        std::vector<Seen> seen(n_nodes, Seen{0, 0, 0, 0});
        uint32_t stamp = 0;
        for (;;) {
            const size_t gi = next_group.fetch_add(1);
            if (gi >= groups.size()) return;
            const std::vector<uint32_t>& one = groups[gi];
            const uint32_t height = height_of(one);
            if (height == 0) {   // an empty group (:197-199 returns an empty circuit list)
                status[gi] = GKR_ERR_INVALID;
                continue;
            }
            LayeredCircuit& lc = out[gi];
            std::vector<uint32_t> current = one, next;
            for (uint32_t d = 0; d <= height && status[gi] == GKR_OK; ++d) {
                const uint32_t k = get_k(current.size());
                current.resize((size_t)1 << k, 0u);   // pad with zero nodes (:209-214); node 0 is the constant 0
                lc.k.push_back(k);
                if (d == height) {   // the input layer (:215-221)
                    lc.input_wire.reserve(current.size());
                    lc.input_const.reserve(current.size());
                    for (uint32_t id : current) {
                        if (!ar.is_leaf(id)) {
                            status[gi] = GKR_ERR_INVALID;
                            break;
                        }
                        const Node& n = ar.nodes[id];
                        lc.input_wire.push_back(n.kind() == kVariable ? n.left : kNone);
                        lc.input_const.push_back(n.kind() == kValue ? ar.values[n.left] : gkr_fr{{0, 0, 0, 0}});
                    }
                    break;
                }
                ++stamp;
                std::vector<uint8_t> types;
                std::vector<uint32_t> lefts, rights;
                types.reserve(current.size());
                lefts.reserve(current.size());
                rights.reserve(current.size());
                next.clear();
                next.reserve(2 * current.size());
                uint32_t zero_index = kNone;
                auto place_child = [&](uint32_t child) {           // :285-303: the first slot holding an equal node, else a new one
                    Seen& e = seen[child];
                    if (e.slot_stamp == stamp) return e.slot;
                    next.push_back(child);
                    e.slot_stamp = stamp;
                    e.slot = (uint32_t)next.size() - 1;
                    return e.slot;
                };
                auto push_slot = [&](uint32_t node) {              // a push that does not look for an equal node first
                    next.push_back(node);
                    Seen& e = seen[node];
                    if (e.slot_stamp != stamp) {                   // (position() keeps the FIRST slot if one exists already)
                        e.slot_stamp = stamp;
                        e.slot = (uint32_t)next.size() - 1;
                    }
                    return (uint32_t)next.size() - 1;
                };
                for (uint32_t id : current) {
                    const Node& n = ar.nodes[id];
                    if (n.kind() == kMult || n.kind() == kAdd) {       // :280-306
                        if (d == height - 1) {                     // panic!("Unsupported"), :225-227
                            status[gi] = GKR_ERR_UNSUPPORTED;
                            break;
                        }
                        types.push_back(n.kind() == kMult ? 1 : 0);
                        const uint32_t l = place_child(n.left);
                        const uint32_t r = place_child(n.right);
                        lefts.push_back(l);
                        rights.push_back(r);
                        continue;
                    }
                    // a leaf above the input layer becomes the relay gate Add(slot of the leaf, zero slot) (:307-342, and
                    // the all-leaf case d == height - 1, :228-264)
                    types.push_back(0);
                    Seen& e = seen[id];
                    if (e.relay_stamp == stamp) {
                        lefts.push_back(e.relay);
                        rights.push_back(zero_index);
                        continue;
                    }
                    if (zero_index == kNone) zero_index = push_slot(0);   // allocated lazily at the current end (:314-317)
                    const uint32_t s = id == 0 ? zero_index : push_slot(id);   // the constant 0 maps to (zero, zero) (:321-325)
                    Seen& e2 = seen[id];
                    e2.relay_stamp = stamp;
                    e2.relay = s;
                    lefts.push_back(s);
                    rights.push_back(zero_index);
                }
                lc.gate_type.push_back(std::move(types));
                lc.left.push_back(std::move(lefts));
                lc.right.push_back(std::move(rights));
                current.swap(next);
            }
        }
    };
    {
        const int threads = compile_threads(groups.size());
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
    }
    for (int st : status)
        if (st != GKR_OK) return st;
    for (LayeredCircuit& lc : out) {
        lc.p_gate_type.clear();
        lc.p_left.clear();
        lc.p_right.clear();
        for (size_t i = 0; i < lc.gate_type.size(); ++i) {
            lc.p_gate_type.push_back(lc.gate_type[i].data());
            lc.p_left.push_back(lc.left[i].data());
            lc.p_right.push_back(lc.right[i].data());
        }
    }
    return GKR_OK;
}

}  // namespace

struct gkr_layered {
    std::vector<LayeredCircuit> circuits;
    size_t tree_nodes = 0;
};

extern "C" {

int gkr_r1cs_parse(const void* bytes, size_t len, gkr_r1cs** out) {
    if (!bytes || !out) return GKR_ERR_INVALID;
    *out = nullptr;
    std::map<uint32_t, std::pair<const uint8_t*, size_t>> sec;
    if (!split_sections(static_cast<const uint8_t*>(bytes), len, "r1cs", 1, sec)) return GKR_ERR_INVALID;
    if (!sec.count(1) || !sec.count(2)) return GKR_ERR_INVALID;
    std::unique_ptr<gkr_r1cs> r(new gkr_r1cs());
    Reader h{sec[1].first, sec[1].second};
    if (!field_header_ok(h)) return GKR_ERR_INVALID;
    r->n_wires = h.u32();
    r->n_pub_out = h.u32();
    r->n_pub_in = h.u32();
    r->n_prv_in = h.u32();
    r->n_labels = h.u64();
    const uint32_t n_constraints = h.u32();
    if (!h.ok) return GKR_ERR_INVALID;
    Reader b{sec[2].first, sec[2].second};
    if ((uint64_t)n_constraints * 12 > sec[2].second) return GKR_ERR_INVALID;
    r->constraints.resize(n_constraints);
    for (uint32_t i = 0; i < n_constraints; ++i)
        for (int j = 0; j < 3; ++j) {
            const uint32_t n = b.u32();
            if (!b.ok || (uint64_t)n * 36 > b.len - b.off) return GKR_ERR_INVALID;
            r->constraints[i][j].resize(n);
            for (uint32_t t = 0; t < n; ++t) {
                Term& term = r->constraints[i][j][t];
                term.wire = b.u32();   // on disk: wire id, then the coefficient
                term.coeff = b.fr();
                if (term.wire >= r->n_wires) return GKR_ERR_INVALID;
                if (!fr_canonical(term.coeff)) return GKR_ERR_NON_CANONICAL;   // from_repr(..).unwrap(), convert.rs:421
            }
        }
    if (!b.ok) return GKR_ERR_INVALID;
    if (sec.count(3)) {
        Reader w{sec[3].first, sec[3].second};
        for (size_t i = 0; i < sec[3].second / 8; ++i) r->wire2label.push_back(w.u64());
    }
    *out = r.release();
    return GKR_OK;
}

int gkr_r1cs_build(uint32_t n_wires, uint32_t n_pub_out, uint32_t n_pub_in, uint32_t n_prv_in, size_t n_constraints,
                   const uint32_t* term_counts, const uint32_t* wires, const gkr_fr* coeffs, gkr_r1cs** out) {
    if (!out || (n_constraints && (!term_counts || !wires || !coeffs))) return GKR_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<gkr_r1cs> r(new gkr_r1cs());
    r->n_wires = n_wires;
    r->n_pub_out = n_pub_out;
    r->n_pub_in = n_pub_in;
    r->n_prv_in = n_prv_in;
    r->n_labels = n_wires;
    r->constraints.resize(n_constraints);
    size_t pos = 0;
    for (size_t i = 0; i < n_constraints; ++i)
        for (int j = 0; j < 3; ++j)
            for (uint32_t t = 0; t < term_counts[3 * i + j]; ++t, ++pos) {
                if (wires[pos] >= n_wires) return GKR_ERR_INVALID;
                if (!fr_canonical(coeffs[pos])) return GKR_ERR_NON_CANONICAL;
                r->constraints[i][j].push_back(Term{coeffs[pos], wires[pos]});
            }
    for (uint32_t w = 0; w < n_wires; ++w) r->wire2label.push_back(w);
    *out = r.release();
    return GKR_OK;
}

int gkr_r1cs_info(const gkr_r1cs* r, gkr_r1cs_info_t* out) {
    if (!r || !out) return GKR_ERR_INVALID;
    out->n_wires = r->n_wires;
    out->n_pub_out = r->n_pub_out;
    out->n_pub_in = r->n_pub_in;
    out->n_prv_in = r->n_prv_in;
    out->n_labels = r->n_labels;
    out->n_constraints = r->constraints.size();
    out->n_terms = 0;
    for (const Constraint& c : r->constraints) out->n_terms += c[0].size() + c[1].size() + c[2].size();
    return GKR_OK;
}

int gkr_r1cs_export(const gkr_r1cs* r, uint32_t* term_counts, uint32_t* wires, gkr_fr* coeffs) {
    if (!r || !term_counts || (!wires && !coeffs)) return GKR_ERR_INVALID;
    size_t pos = 0;
    for (size_t i = 0; i < r->constraints.size(); ++i)
        for (int j = 0; j < 3; ++j) {
            term_counts[3 * i + j] = (uint32_t)r->constraints[i][j].size();
            for (const Term& t : r->constraints[i][j]) {
                if (wires) wires[pos] = t.wire;
                if (coeffs) coeffs[pos] = t.coeff;
                ++pos;
            }
        }
    return GKR_OK;
}

static int copy_out(const std::vector<uint8_t>& b, void* out, size_t capacity, size_t* needed) {
    if (needed) *needed = b.size();
    if (!out) return GKR_OK;
    if (capacity < b.size()) return GKR_ERR_NOMEM;
    memcpy(out, b.data(), b.size());
    return GKR_OK;
}

int gkr_r1cs_serialize(const gkr_r1cs* r, void* out, size_t capacity, size_t* needed) {
    if (!r || (!out && !needed)) return GKR_ERR_INVALID;
    Writer hdr, body, map, file;
    hdr.u32(32);
    hdr.put(kMod, 32);
    hdr.u32(r->n_wires);
    hdr.u32(r->n_pub_out);
    hdr.u32(r->n_pub_in);
    hdr.u32(r->n_prv_in);
    hdr.u64(r->n_labels);
    hdr.u32((uint32_t)r->constraints.size());
    for (const Constraint& c : r->constraints)
        for (int j = 0; j < 3; ++j) {
            body.u32((uint32_t)c[j].size());
            for (const Term& t : c[j]) {
                body.u32(t.wire);
                body.fr(t.coeff);
            }
        }
    for (uint64_t l : r->wire2label) map.u64(l);
    file.put("r1cs", 4);
    file.u32(1);
    file.u32(3);
    uint32_t ty = 1;
    for (const Writer* w : {&hdr, &body, &map}) {
        file.u32(ty++);
        file.u64(w->b.size());
        file.put(w->b.data(), w->b.size());
    }
    return copy_out(file.b, out, capacity, needed);
}

void gkr_r1cs_free(gkr_r1cs* r) { delete r; }

int gkr_wtns_parse(const void* bytes, size_t len, gkr_fr* out, size_t capacity, size_t* count) {
    if (!bytes || (!out && !count)) return GKR_ERR_INVALID;
    std::map<uint32_t, std::pair<const uint8_t*, size_t>> sec;
    if (!split_sections(static_cast<const uint8_t*>(bytes), len, "wtns", 2, sec)) return GKR_ERR_INVALID;
    if (!sec.count(1) || !sec.count(2)) return GKR_ERR_INVALID;
    Reader h{sec[1].first, sec[1].second};
    if (!field_header_ok(h)) return GKR_ERR_INVALID;
    const uint32_t n = h.u32();
    if (!h.ok || (uint64_t)n * 32 > sec[2].second) return GKR_ERR_INVALID;
    if (count) *count = n;
    if (!out) return GKR_OK;
    if (capacity < n) return GKR_ERR_NOMEM;
    memcpy(out, sec[2].first, (size_t)n * 32);
    for (uint32_t i = 0; i < n; ++i)
        if (!fr_canonical(out[i])) return GKR_ERR_NON_CANONICAL;   // from_repr(..).unwrap(), convert.rs:805
    return GKR_OK;
}

int gkr_wtns_serialize(const gkr_fr* values, size_t count, void* out, size_t capacity, size_t* needed) {
    if ((!values && count) || (!out && !needed) || count > 0xFFFFFFFFull) return GKR_ERR_INVALID;
    for (size_t i = 0; i < count; ++i)
        if (!fr_canonical(values[i])) return GKR_ERR_NON_CANONICAL;
    Writer hdr, file;
    hdr.u32(32);
    hdr.put(kMod, 32);
    hdr.u32((uint32_t)count);
    file.put("wtns", 4);
    file.u32(2);
    file.u32(2);
    file.u32(1);
    file.u64(hdr.b.size());
    file.put(hdr.b.data(), hdr.b.size());
    file.u32(2);
    file.u64(count * 32);
    file.put(values, count * 32);
    return copy_out(file.b, out, capacity, needed);
}

int gkr_r1cs_compile(const gkr_r1cs* r, gkr_layered** out, size_t* bad_constraint) {
    if (!r || !out) return GKR_ERR_INVALID;
    *out = nullptr;
    // an upper bound on the tree nodes: per term a constant, a variable and their product; a merge tree of m - 1 sums per
    // linear combination of m terms; two gates per constraint; the constants 0 and the two units
    size_t n_terms = 0;
    for (const Constraint& c : r->constraints) n_terms += c[0].size() + c[1].size() + c[2].size();
    if (4 * n_terms + 2 * r->constraints.size() + 8 > 0xFFF00000ull) return GKR_ERR_INVALID;
    // (expected: what a circom R1CS gives -- most variables and constants recur, roughly one new node per term)
    const size_t slack = 64 * (size_t)Arena::kIdBlock;     // (ids are handed to the building threads in blocks)
    Arena ar(4 * n_terms + 2 * r->constraints.size() + 8 + slack, n_terms + 8 + slack, n_terms + 2 * r->constraints.size() + 8);
    std::vector<uint32_t> groups;
    int rc = constraint_trees(*r, ar, groups, bad_constraint);
    if (rc) return rc;
    std::unique_ptr<gkr_layered> L(new gkr_layered());
    if (!groups.empty()) {
        rc = compile_groups(ar, groups, L->circuits);
        if (rc) return rc;
    }

    L->tree_nodes = ar.size();
    *out = L.release();
    return GKR_OK;
}

int gkr_layered_count(const gkr_layered* L, uint32_t* circuits) {
    if (!L || !circuits) return GKR_ERR_INVALID;
    *circuits = (uint32_t)L->circuits.size();
    return GKR_OK;
}

int gkr_layered_circuit(const gkr_layered* L, uint32_t index, gkr_circuit_desc* out) {
    if (!L || !out || index >= L->circuits.size()) return GKR_ERR_INVALID;
    const LayeredCircuit& c = L->circuits[index];
    out->depth = (uint32_t)c.gate_type.size();
    out->k = c.k.data();
    out->gate_type = c.p_gate_type.data();
    out->left = c.p_left.data();
    out->right = c.p_right.data();
    return GKR_OK;
}

int gkr_layered_input_layer(const gkr_layered* L, uint32_t index, const uint32_t** wire, const gkr_fr** constant, size_t* slots) {
    if (!L || index >= L->circuits.size()) return GKR_ERR_INVALID;
    const LayeredCircuit& c = L->circuits[index];
    if (wire) *wire = c.input_wire.data();
    if (constant) *constant = c.input_const.data();
    if (slots) *slots = c.input_wire.size();
    return GKR_OK;
}

int gkr_layered_input_values(const gkr_layered* L, uint32_t index, const gkr_fr* witness, size_t n_witness, gkr_fr* out_values) {
    if (!L || !witness || !out_values || index >= L->circuits.size()) return GKR_ERR_INVALID;
    const LayeredCircuit& c = L->circuits[index];
    for (size_t s = 0; s < c.input_wire.size(); ++s) {
        if (c.input_wire[s] == 0xFFFFFFFFu) {
            out_values[s] = c.input_const[s];
        } else {
            if (c.input_wire[s] >= n_witness) return GKR_ERR_INVALID;   // witness[var] out of bounds panics, convert.rs:804
            if (!fr_canonical(witness[c.input_wire[s]])) return GKR_ERR_NON_CANONICAL;
            out_values[s] = witness[c.input_wire[s]];
        }
    }
    return GKR_OK;
}

void gkr_layered_free(gkr_layered* L) { delete L; }

}  // extern "C"
