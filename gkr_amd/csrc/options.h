// Every switch of the library, in ONE table.  Nothing else in the library calls getenv.
//
//   * context options: a copy of the table's values lives in every gkr_ctx, filled when the context is created (the
//     table's default, or the value of the environment variable named beside it -- the variables are the way a test
//     or an A/B script reaches a library it does not call itself) and changed per context with
//     gkr_ctx_set_option(ctx, "name", value).  The code reads them through gkr::opt(OPT_x): the options of the context
//     whose entry point the calling thread is inside (OptionScope), the process defaults on a thread outside any.
//   * process switches: properties of the host CPU / the process as a whole (which hash code the CPU runs, how many
//     threads the process may use, debug output).  Read once, from the environment only.
//
// None of these changes a result: every value of every option is covered by the parity suite (tests/test_gpu_parity.py
// `test_*_variants`, tests/test_gpu_config_scale.py), which is what they are for.
#pragma once
#include <stdlib.h>
#include <string.h>

namespace gkr {

// X(id, "option name", "ENVIRONMENT VARIABLE", default, "what it does")
#define GKR_CONTEXT_OPTIONS(X)                                                                                                       \
    /* ---- plain sumcheck (prove_sumcheck, sumcheck.rs:158-214) ---- */                                                              \
    X(rounds_per_pass, "rounds_per_pass", "GKR_ROUNDS_PER_PASS", 0, "rounds a pass of the plain sumcheck covers (0: 5; 3 without the MFMA fold)") \
    X(no_mfma_fold, "no_mfma_fold", "GKR_NO_MFMA_FOLD", 0, "fold passes on v_mad_u64_u32 instead of int8 MFMA")                      \
    X(no_fused_reduce, "no_fused_reduce", "GKR_NO_FUSED_REDUCE", 0, "latency-bound passes publish through k_mle_sub_reduce instead of from their last block") \
    X(hash_chunk, "hash_chunk", "GKR_HASH_CHUNK", 0, "8 or 16: sumchecks per host hash call (0: 16 when host threads are scarce)")    \
    X(device_hash_percent, "device_hash_percent", "GKR_DEVICE_HASH_PERCENT", 0, "share of a batch of >= 64 plain sumchecks hashed on the device (0..90)") \
    X(group_size, "group_size", "GKR_GROUP_SIZE", 0, "sumchecks per scheduling group of a batch (0: groups of ~4 GiB of tables)")     \
    X(pass_queue_depth, "pass_queue_depth", "GKR_PASS_QUEUE_DEPTH", 0, "groups whose pass 0 is queued up front (0: two)")           \
    X(no_late_stream, "no_late_stream", "GKR_NO_LATE_STREAM", 0, "a group's small late passes stay on the main stream")              \
    X(plan_main, "plan_main", "GKR_PLAN_MAIN", 0, "the fold plan (digit matrix) on the main stream instead of the side stream")       \
    X(mle_per_round, "mle_per_round", "GKR_MLE_PER_ROUND", 0, "one variable per pass (the first schedule; what the device transcript uses)") \
    X(fold_blocks, "fold_blocks", "GKR_FOLD_BLOCKS", 0, "target block count of a fold pass (0: 65536)")                              \
    X(fold_min_chunk, "fold_min_chunk", "GKR_FOLD_MIN_CHUNK", 0, "smallest chunk of entries per fold block (0: 256)")                \
    X(items_per_block, "items_per_block", "GKR_ITEMS_PER_BLOCK", 0, "entries per block of the first pass (0: 1024)")                 \
    /* ---- layer sumcheck (prove_sumcheck_opt, sumcheck.rs:36-156) ---- */                                                            \
    X(predicate_atomics, "predicate_atomics", "GKR_PREDICATE_ATOMICS", 0, "dense predicate tables by widened-atomic scatter instead of the counting sort") \
    X(gate_groups_min_k, "gate_groups_min_k", "GKR_GATE_GROUPS_MIN_K", -1, "smallest k_next whose layers take the wide layers' item passes (-1: 13)") \
    X(no_fused_publish, "no_fused_publish", "GKR_NO_FUSED_PUBLISH", 0, "product passes publish through a second launch instead of from their last block") \
    X(gate_sort_global, "gate_sort_global", "GKR_GATE_SORT_GLOBAL", 0, "the gate lists' counting sort with global atomics where the LDS sort applies") \
    X(gate_segments_off, "gate_segments_off", "GKR_GATE_SEGMENTS_OFF", 0, "bucket kernels instead of the segment passes on large layers") \
    X(gate_segments_min_log2, "gate_segments_min_log2", "GKR_GATE_SEGMENTS_MIN_LOG2", 22, "smallest layer (log2 gates) that takes the segment passes") \
    X(gate_segment_log2, "gate_segment_log2", "GKR_GATE_SEGMENT_LOG2", 0, "log2 of the segments' mean length (0: 4)")              \
    X(gate_segments_no_lds, "gate_segments_no_lds", "GKR_GATE_SEGMENTS_NO_LDS", 0, "the segment pass gathers its table from L2 instead of LDS") \
    X(no_mfma_cross, "no_mfma_cross", "GKR_NO_MFMA_CROSS", 0, "wide layers' product passes over tables of 2^15 entries and more on v_mad_u64_u32 instead of int8 MFMA") \
    X(prod_cross_min_log2, "prod_cross_min_log2", "GKR_PROD_CROSS_MIN_LOG2", 0, "smallest tables (log2 entries, at least 13) whose cross sums of a product pass run on the matrix cores (0: 15)") \
    X(prod_fold_min_log2, "prod_fold_min_log2", "GKR_PROD_FOLD_MIN_LOG2", 0, "smallest tables (log2 entries) whose pending fold of a product pass runs on the matrix cores (0: 17)") \
    X(host_tail_log2, "host_tail_log2", "GKR_HOST_TAIL_LOG2", 0, "a phase's product passes over tables of 2^this entries and fewer run on the host (0: 6; -1: none; at most 12)") \
    X(host_tail_max_batch, "host_tail_max_batch", "GKR_HOST_TAIL_MAX_BATCH", 0, "largest batch (proofs advancing together) whose small product passes run on the host (0: 8)") \
    X(line_stepwise, "line_stepwise", "GKR_LINE_STEPWISE", 0, "q_i in its wide-layer form (one launch per variable) at every width")  \
    X(layer_no_fused, "layer_no_fused", "GKR_LAYER_NO_FUSED", 0, "device transcript: separate fold and sum kernels in the b-phase")   \
    /* ---- whole proofs (prover::prove and its par_iter, prover.rs:6-96, aggregator.rs:350-355) ---- */                              \
    X(no_circuit_cache, "no_circuit_cache", "GKR_NO_CIRCUIT_CACHE", 0, "do not keep proven circuits' gate arrays and lists on the device between calls") \
    X(prove_many_pieces, "prove_many_pieces", "GKR_PROVE_MANY_PIECES", 0, "gkr_prove_many: cut the costliest items in two until there are this many")      \
    X(prove_many_lockstep, "prove_many_lockstep", "GKR_PROVE_MANY_LOCKSTEP", 1, "gkr_prove_many: items whose circuits share a k list advance in lockstep, one launch per pass for the group (0: one chain per item)") \
    X(lockstep_max_proofs, "lockstep_max_proofs", "GKR_LOCKSTEP_MAX_PROOFS", 0, "most proofs one lockstep group may hold (0: 32 -- beyond that independent chains overlap better)")

enum OptionId : int {
#define GKR_OPT_ENUM(id, name, env, def, doc) OPT_##id,
    GKR_CONTEXT_OPTIONS(GKR_OPT_ENUM)
#undef GKR_OPT_ENUM
        OPT_COUNT
};

struct OptionInfo {
    const char* name;
    const char* env;
    long long def;
    const char* doc;
};

inline const OptionInfo* option_table() {
    static const OptionInfo table[OPT_COUNT] = {
#define GKR_OPT_ROW(id, name, env, def, doc) {name, env, def, doc},
        GKR_CONTEXT_OPTIONS(GKR_OPT_ROW)
#undef GKR_OPT_ROW
    };
    return table;
}

struct Options {
    long long v[OPT_COUNT];
    // the table's defaults, overridden by the environment as it is NOW (a context reads it when it is created)
    static Options from_environment() {
        Options o;
        const OptionInfo* t = option_table();
        for (int i = 0; i < OPT_COUNT; ++i) {
            o.v[i] = t[i].def;
            if (const char* e = getenv(t[i].env)) {
                char* end = nullptr;
                const long long val = strtoll(e, &end, 0);
                o.v[i] = end != e ? val : 1;   // (a switch set to anything that is not a number is "on")
            }
        }
        return o;
    }
};

inline int option_index(const char* name) {
    if (!name) return -1;
    const OptionInfo* t = option_table();
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(t[i].name, name) || !strcmp(t[i].env, name)) return i;
    return -1;
}

inline const Options*& current_options() {
    static thread_local const Options* cur = nullptr;
    return cur;
}
inline const Options& default_options() {
    static const Options d = Options::from_environment();
    return d;
}
inline long long opt(OptionId id) {
    const Options* o = current_options();
    return (o ? o : &default_options())->v[id];
}
// an entry point of the C ABI is inside one of these for its whole duration (threads the library starts for a context's
// work enter one themselves)
struct OptionScope {
    const Options* prev;
    explicit OptionScope(const Options* o) : prev(current_options()) { current_options() = o; }
    ~OptionScope() { current_options() = prev; }
    OptionScope(const OptionScope&) = delete;
    OptionScope& operator=(const OptionScope&) = delete;
};

// ---- process switches (environment only, read once) ------------------------------------------------------------------
//   GKR_DEBUG_TIMING   timers of the host loops on stderr
//   GKR_NO_IFMA        scalar host hashing even where AVX-512 IFMA is available
//   GKR_NO_ADX         a lone transcript's hash on the portable 4 x 64-bit code instead of mulx / adcx / adox
//   GKR_HOST_PASS_SCALAR  the plain sumcheck's host pass on the scalar code (A/B against the IFMA lanes)
//   GKR_NO_HELP        contexts proving side by side do not share their host work
//   GKR_HOST_THREADS   host threads a context's transcript may use (gkr_ctx_set_host_threads overrides per context)
//   GKR_COMPILE_THREADS  threads of gkr_r1cs_compile (0 / unset: the CPUs this process may use, at most 32)
//   LOCAL_WORLD_SIZE   ranks sharing this host's CPUs (set by torch.distributed.run)
inline bool process_switch(const char* env) { return getenv(env) != nullptr; }
inline int process_int(const char* env, int def) {
    const char* e = getenv(env);
    return e ? atoi(e) : def;
}
inline bool debug_timing() {
    static const bool on = process_switch("GKR_DEBUG_TIMING");
    return on;
}

}  // namespace gkr
