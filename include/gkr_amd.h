/* gkr_amd -- C ABI of the MI355X (gfx950) GKR sumcheck prover.
 *
 * Drop-in boundary for ONE path of jeong0982/gkr: the GKR prover's per-layer
 * sumcheck and what feeds it.  The reference has no FFI; its boundary is two
 * crate-internal Rust functions, and every entry point below names the one it
 * replaces.  INTEGRATION.md shows the Rust-side binding.
 *
 *   reference                                                    this library
 *   -----------------------------------------------------------  ----------------------------
 *   prover::prove(&GKRCircuit,&Input) -> Proof                    gkr_prove
 *       rust/src/gkr/prover.rs:6-96 (called aggregator.rs:354,415)
 *   sumcheck::prove_sumcheck_opt(add_wire,mult_wire,add_i,...)    gkr_sumcheck_layer
 *       rust/src/gkr/sumcheck.rs:36-156 (called prover.rs:52-60)
 *   sumcheck::prove_sumcheck(g, v)                                gkr_sumcheck_mle[_batch_device]
 *       rust/src/gkr/sumcheck.rs:158-214 (python/sumcheck.py:6-53)
 *   convert::calculate_input (forward step)                       gkr_layer_eval
 *       rust/src/convert.rs:812-831
 *   wiring predicates add_i/mult_i restricted to z                gkr_predicate_tables
 *       rust/src/convert.rs:715-767 + prover.rs:24-37 (poly.rs:28-62)
 *   poly::reduce_multiple_polynomial / l_function                 inside gkr_prove
 *       rust/src/gkr/poly.rs:469-500, 538-551
 *   Mimc7::new(91).multi_hash(v, &Fr::from(0))  (mimc-rs)         gkr_mimc7_multi_hash
 *       call sites sumcheck.rs:45,84,129,152; prover.rs:10,78
 *
 * Data: a field element is halo2curves bn256::Fr exchanged as its 32-byte
 * little-endian canonical repr (sumcheck.rs:10-22): gkr_fr, 4 LE u64 limbs,
 * value < r.  Never Montgomery at this ABI.
 *
 * Tables are dense evaluation tables over the boolean hypercube, index = the
 * variables' bit string with variable 1 most significant (poly.rs:507).
 *
 * Round vectors: the reference returns Vec<S> of varying length, highest degree
 * first (poly.rs:260-267).  Here every round has a fixed row of slots (2 for the
 * plain sumcheck, 3 for the layer sumcheck), RIGHT-aligned: the last slot is the
 * constant term, out_len says how many trailing slots form the reference's
 * vector, unused leading slots are zero.
 *
 * Ownership: the caller allocates every output; the library owns device memory
 * inside the context.  Errors: the reference panics; this ABI returns a status
 * and never aborts.  Threading: a context is single-owner (one HIP stream);
 * distinct contexts may be used concurrently (the reference calls prove from a
 * rayon par_iter, aggregator.rs:350-355).
 *
 * There is no CPU fallback: every compute entry point needs a gfx950 device and
 * returns GKR_ERR_NO_DEVICE / GKR_ERR_HIP otherwise.
 */
#ifndef GKR_AMD_H
#define GKR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } gkr_fr;

typedef struct gkr_ctx gkr_ctx;

/* ---- limits (every size limit of the library; tests/test_host_library.py checks this table against the code) ----
 * The reference proves any GKRCircuit its compiler emits (prover.rs:6-96; layers padded to whatever 2^k they need,
 * convert.rs:209-214).  The library's limits are those of 32-bit gate indices and device memory, not of a table form:
 *   GKR_MAX_K_NEXT   widest next layer of a layer sumcheck (2^k_next values W; k[i+1] of gkr_prove): U, V, W, the
 *                    c-phase row and eq(u, .) are 2^k_next-entry tables in HBM (9 x 32 B per value and proof);
 *   GKR_MAX_K_I      most gates of a layer, 2^k_i (also the output layer k[0]); sorted gate lists cost 16 B per gate;
 *   GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT   with GKR_TRANSCRIPT_DEVICE the layer sumcheck works on dense 2^(2 k_next)-entry
 *                    predicate tables (the host-free form is complete, not fast): k_next <= 14;
 *   GKR_MAX_MLE_N    plain sumcheck: tables of 2^n values, batch * 2^n <= 2^30 values (32 GiB) per call;
 *   GKR_MAX_BATCH    proofs per gkr_prove_batch call.
 * A call beyond a limit returns GKR_ERR_INVALID and names the limit in gkr_last_error. */
#define GKR_MAX_K_NEXT 24
#define GKR_MAX_K_I 28
#define GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT 14
#define GKR_MAX_MLE_N 30
#define GKR_MAX_BATCH 4096

enum {
    GKR_OK = 0,
    GKR_ERR_INVALID = 1,      /* bad sizes / null pointers / operand index out of range */
    GKR_ERR_NON_CANONICAL = 2,/* a field element >= r (the reference unwrap()s from_repr, sumcheck.rs:16,21) */
    GKR_ERR_NO_DEVICE = 3,
    GKR_ERR_HIP = 4,          /* a HIP runtime call failed; see gkr_last_error */
    GKR_ERR_NOMEM = 5,
    GKR_ERR_DEGENERATE = 6,   /* v == 0: the reference underflows (sumcheck.rs:49) */
    GKR_ERR_UNSUPPORTED = 7   /* an R1CS shape the reference's compiler cannot handle either (it panics or recurses
                                 without end: a constraint with an empty A, B or C, convert.rs:619-622) */
};

/* Transcript placement: where the per-round MiMC7 hash runs.  The hash is a
 * 728-deep serial chain of modular products per round vector with no data
 * parallelism inside it; all table work stays on the GPU in both modes. */
enum {
    GKR_TRANSCRIPT_DEVICE = 0, /* one GPU lane per sumcheck; no host involvement per round */
    GKR_TRANSCRIPT_HOST = 1    /* default: the device publishes the round sums to pinned memory, host
                                  cores hash (one sumcheck per thread), the next launch reads r_j */
};

const char *gkr_strerror(int status);
const char *gkr_version(void);

/* ---- context ---------------------------------------------------------- */
int  gkr_ctx_create(int device_id, gkr_ctx **out);
/* ONE host process driving several GPUs, as the reference is one process whose par_iter fans prover::prove out over the
 * (circuit, input) pairs of a step (aggregator.rs:350-355, 411-416): the context lives on device_ids[0]; gkr_prove_many
 * deals its items, by estimated cost, over child contexts created round-robin on ALL the listed devices (each with its
 * own thread, stream, workspaces and circuit cache; a device listed k times gets k child contexts per round -- which is
 * also how the dealing is tested on a box with one GPU).  Every other entry point runs on device_ids[0].
 * n_devices in [1, 64].  gkr_ctx_device_count: how many devices the context deals over (1 for gkr_ctx_create). */
int  gkr_ctx_create_multi(const int *device_ids, int n_devices, gkr_ctx **out);
int  gkr_ctx_device_count(const gkr_ctx *ctx);
void gkr_ctx_destroy(gkr_ctx *ctx);
const char *gkr_last_error(const gkr_ctx *ctx);
int  gkr_ctx_set_transcript(gkr_ctx *ctx, int mode);
/* Host threads this context may keep busy with the transcript (the calling thread included); 0 = the default:
 * GKR_HOST_THREADS, else (CPUs of the affinity mask and cgroup quota) / LOCAL_WORLD_SIZE - 2.  A process that drives
 * several contexts from several threads (the reference proves its <= 20 sub-circuits from a rayon par_iter,
 * aggregator.rs:350-355) gives each context its share; 1 = hash on the calling thread, no workers. */
int  gkr_ctx_set_host_threads(gkr_ctx *ctx, int threads);
/* The library's switches, per context.  gkr_amd/csrc/options.h holds the ONE table of them (name, default, what it does;
 * gkr_option_count / _name / _doc / _env enumerate it); a context takes the table's defaults -- or, for a test or an A/B
 * script that cannot call in, the value of the environment variable named in the table -- when it is created, and
 * gkr_ctx_set_option changes one value for this context only (cached per-circuit state built under the old value is
 * dropped; resident layers keep the layout they were created with, so set options first).  `name` is the option's name
 * ("rounds_per_pass") or its environment variable ("GKR_ROUNDS_PER_PASS").  No option changes a result: each is a choice
 * between schedules or kernel forms that the parity suite holds bit-exact against each other.  gkr_prove_many's child
 * contexts prove with the options of the context the call was made on. */
int  gkr_ctx_set_option(gkr_ctx *ctx, const char *name, long long value);
int  gkr_ctx_get_option(const gkr_ctx *ctx, const char *name, long long *value);
int  gkr_option_count(void);
const char *gkr_option_name(int index);
const char *gkr_option_doc(int index);
const char *gkr_option_env(int index);
int  gkr_ctx_device_name(const gkr_ctx *ctx, char *buf, size_t len);
/* Contexts proving side by side (one calling thread each -- the reference's par_iter over the (circuit, input) pairs,
 * aggregator.rs:350-355) share their host work: a thread that waits for its own GPU round takes pieces of another
 * context's posted work (a 16-lane hash call, a proof's line restriction).  A thread that has run out of proving
 * calls can lend itself to the others: gkr_host_help_while runs such pieces on the calling thread while *busy != 0
 * (the caller's count of threads still proving) and returns the number of pieces it ran.  GKR_NO_HELP=1 disables
 * the sharing. */
long gkr_host_help_while(const volatile int32_t *busy);
/* Where the proving threads' time goes, summed over every thread of the process while enabled (what GKR_DEBUG_TIMING prints
 * per call, as figures): enable = 1 clears the totals and starts counting, 0 stops.  gkr_host_accounting_read fills, in
 * microseconds: [0] a thread's own host pieces (hashing, incl. waiting for helpers to finish theirs), [1] pieces of other
 * contexts' work taken while waiting for its own GPU round, [2] spinning on the GPU with nothing to take, [3] the rest of
 * its gkr_prove / gkr_prove_batch calls (launches, set-up, copies), [4] pieces run by threads that had no proof left
 * (gkr_host_help_while), [5] their time with nothing to take, [6] the number of proving calls counted, and with
 * count >= 8 [7] the time gkr_prove_many's threads took to start on their items after the call woke them.  count >= 7.
 * With count >= 28 also the hashing PIECES of the layer sumchecks themselves, whoever ran them: [8] how many, [9] their time,
 * [10] the part of it inside the pass function (a piece's J hashes per transcript -- in IFMA lanes, or on the scalar code for
 * fewer than three transcripts -- and the field arithmetic between them; the rest of [9] is copying the round vectors out),
 * [10 + n] the number of pieces that carried n transcripts (n = 1 .. 16: the lanes of an IFMA call that were filled).
 * [0] + [1] + [4] - [9] is what the threads spent posting pieces, looking for them and waiting for the last one of a pass.
 * The reference has no counterpart (rayon hides its scheduling); bench.py reports these for its aggregated-proofs leg. */
int  gkr_host_accounting(int enable);
int  gkr_host_accounting_read(double *out_us, size_t count);

/* Per-kernel timing with HIP events on the stream each kernel is launched on (bench.py's
 * roofline leg).  enable: 0 off, 1 every kernel, 2 only the bandwidth-bound kernels ("mle_multifold",
 * "mle_sub_sums", "mle_fold_sum", "layer_round", ...): the small kernels on the round-trip path are
 * left alone so that the event records do not show in the wall time being measured.
 * kernel: "mle_multifold" (streaming fold passes on the main stream), "mle_multifold_late" (the small late fold
 * passes on the high-priority stream, overlapping other groups' passes), "mle_sub_sums", "mle_sub_reduce", "mle_pass_small", "mle_fold_plan",
 * "mle_fold_sum", "mle_sum_first", "mle_round_hash", "layer_round", "layer_fold", "layer_round_hash". */
int  gkr_ctx_profile(gkr_ctx *ctx, int enable);
int  gkr_ctx_profile_get(gkr_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms,
                         double *algorithmic_bytes);
/* the single launches behind a row (duration and algorithmic bytes of each, in launch order; the first 4096 since
 * the last reset): *count = how many there are, at most `capacity` are written */
int  gkr_ctx_profile_samples(gkr_ctx *ctx, const char *kernel, double *ms, double *bytes, size_t capacity, size_t *count);
int  gkr_ctx_profile_reset(gkr_ctx *ctx);

/* ---- MiMC7 (host; no device needed) ------------------------------------ */
/* multi_hash(arr, key): r = key; for a in arr: r += a + hash(a, r) */
int  gkr_mimc7_multi_hash(const gkr_fr *arr, size_t n, const gkr_fr *key, gkr_fr *out);
int  gkr_mimc7_hash(const gkr_fr *x, const gkr_fr *k, gkr_fr *out);
int  gkr_mimc7_constant(int i, gkr_fr *out);     /* i in 0..90 */
/* the device arithmetic (8x32-bit Montgomery) run on the host, for CPU-side unit tests */
int  gkr_selftest_mul(const gkr_fr *a, const gkr_fr *b, gkr_fr *out);
int  gkr_selftest_wide_sum(const gkr_fr *vals, size_t n, gkr_fr *out);
/* the host transcript's batched hash: eight right-aligned 3-slot round vectors at once */
int  gkr_selftest_hash8(const gkr_fr *vecs, const uint32_t *len, gkr_fr *out, int *used_ifma);
/* the host's share of one multi-round pass of the plain sumcheck: `count` <= 16 sumchecks with 2^J <= 32 sub-block
 * sums each (rows of 32: sums[k*32 + b]) -> per round t < J and sumcheck k (index t*count + k) the round polynomial
 * c1 x + c0, the vector's length (final_len non-null: the lengths of the last round are given), the challenge
 * multi_hash(vector); w[k*32 + b], b < 2^J: the weights eq(r, b) of the fold pass that follows (canonical, first
 * challenge = most significant index bit).  Scalar code; its IFMA-lane form runs beside it when the CPU has it and
 * must agree. */
int  gkr_selftest_host_pass(const gkr_fr *sums, int count, int J, const uint32_t *final_len, gkr_fr *c0, gkr_fr *c1,
                            uint32_t *len, gkr_fr *r, gkr_fr *w, int *used_ifma);
/* the host tail of a phase's product passes (once the tables are down to 2^host_tail_log2 entries the host forms the passes'
 * records itself, from tables the last device pass leaves in pinned memory): tables = W, X, Y of 2^m <= 2^12 canonical entries each,
 * weights = the previous pass's 2^jp canonical weights (jp = 0: none pending), J = the rounds of the pass -> rec: the 72 values
 * the device pass would have left (m[a*8 + b] of the folded tables' 2^J sub-blocks, the sub-block sums of Y at 64 + a).  Scalar code;
 * its eight-lane IFMA form runs beside it when the CPU has it (*used_ifma = 1) and must agree. */
int  gkr_selftest_host_tail(const gkr_fr *tables, int m, int jp, const gkr_fr *weights, int J, gkr_fr *rec, int *used_ifma);
/* the host's share of one product pass of the layer sumcheck (h = W X + Y over three tables): `count` <= 16 sumchecks,
 * recs: 72 values each -- the cross sums m[a*8 + b] = sum_i W[aS+i] X[bS+i] of the 2^J <= 8 sub-blocks, then the Y sums
 * at 64 + a; vec_len[t*count + k] = 2 or 3 -> per round t < J (index t*count + k) the coefficients of
 * c2 X^2 + lin X + c0 and the challenge; w[k*8 + b]: the weights eq(r, b) of the fold that follows (canonical).
 * Scalar code; its IFMA-lane form runs beside it when the CPU has it and must agree. */
int  gkr_selftest_host_prod_pass(const gkr_fr *recs, int count, int J, const uint32_t *vec_len, gkr_fr *c2, gkr_fr *lin,
                                 gkr_fr *c0, gkr_fr *r, gkr_fr *w, int *used_ifma);
/* sum_i a_i b_i through the unreduced 544-bit dot-product accumulator of the fused layer kernel */
int  gkr_selftest_dot(const gkr_fr *a, const gkr_fr *b, size_t n, gkr_fr *out);
/* the pass schedule of a 2^n-point plain sumcheck (host logic): rounds covered by each pass; mfma = 1 default
 * (up to 5 rounds per pass), 0 the v_mad_u64_u32 fold's (up to 3).  *passes = number of passes. */
int  gkr_selftest_pass_schedule(int n, int mfma, uint32_t *rounds, size_t capacity, size_t *passes);
/* q(t) = W(b + t (c - b)) (reduce_multiple_polynomial, poly.rs:469-500) as gkr_prove computes it on the host: W = 2^k
 * evaluations; out = k + 1 slots right-aligned, highest degree first; *out_len = 1 + largest monomial degree of W */
int  gkr_selftest_line_restriction(int k, const gkr_fr *W, const gkr_fr *b, const gkr_fr *c, gkr_fr *out, uint32_t *out_len);
/* one item (<= 32 gates) of a gate-list segment and its combine step, on the host twins of the device code
 * (csrc/gate_seg.h): rows = 0: out0 = e_hi * sum_i e_lo[i] (is_mult[i] ? t[i] : 1), out1 = e_hi * sum_{!is_mult} e_lo[i] t[i]
 * (the U, V sums of sumcheck.rs:50-63); rows = 1: out0 over the add gates, out1 over the mult gates (:97-124) */
int  gkr_selftest_seg_item(const gkr_fr *e_lo, const gkr_fr *t, const uint8_t *is_mult, size_t n, const gkr_fr *e_hi, int rows,
                           gkr_fr *out0, gkr_fr *out1);
/* lo + r (hi - lo) through the fixed-multiplier table the fold kernels use */
int  gkr_selftest_fold(const gkr_fr *lo, const gkr_fr *hi, const gkr_fr *r, gkr_fr *out);

/* ---- plain multilinear sumcheck: prove_sumcheck(g, v), sumcheck.rs:158-161 --- */
/* table: 2^n canonical evaluations on the host.  out_coeffs: n rows x 2 slots;
 * out_len[j] in {1,2}; out_r: n challenges.  n >= 2. */
int  gkr_sumcheck_mle(gkr_ctx *ctx, const gkr_fr *table, int n, gkr_fr *out_coeffs, uint32_t *out_len,
                      gkr_fr *out_r);

/* `batch` independent sumchecks whose tables already sit in device memory,
 * contiguous (table b starts at d_tables + b * 2^n elements).  Inputs are not
 * modified.  Outputs are host arrays of batch x n rows. */
int  gkr_sumcheck_mle_batch_device(gkr_ctx *ctx, const void *d_tables, int n, int batch,
                                   gkr_fr *out_coeffs, uint32_t *out_len, gkr_fr *out_r);

/* ---- GKR layer sumcheck: prove_sumcheck_opt, sumcheck.rs:36-44 ----------- */
/* Layer i has 2^k_i gates; gate g is add (0) or mult (1) of entries left[g],
 * right[g] of layer i+1, which has 2^k_next entries W.  z: k_i challenges.
 * out_coeffs: 2*k_next rows x 3 slots; out_len[j] in {2,3}; out_r: 2*k_next. */
int  gkr_sumcheck_layer(gkr_ctx *ctx, int k_i, int k_next, const uint8_t *gate_type,
                        const uint32_t *left, const uint32_t *right, const gkr_fr *z,
                        const gkr_fr *W, gkr_fr *out_coeffs, uint32_t *out_len, gkr_fr *out_r);

/* ---- one layer sumcheck split across GPUs by GATES ------------------------------------------------------
 * The reference sums the per-gate terms of a round with a rayon map-reduce over the gate list
 * (sumcheck.rs:50-63 b-rounds, :97-124 c-rounds).  In the linear-time form of the layer sumcheck every table the
 * rounds work on is a SUM OVER GATES: U(b), V(b) (2^k_next entries each) for the k_next rounds that bind b, then
 * the row a_u(c), m_u(c) for the k_next rounds that bind c.  So ANY partition of the gates over ranks works: each
 * rank sums its own gates, two sum-over-ranks exchanges per layer (2 * 2^k_next field elements each) complete the
 * tables, and the 2 k_next rounds run on the completed (tiny) tables identically on every rank -- every rank gets
 * the whole transcript, no broadcast.  Bit-exact: modular sums commute.
 *
 * This rank holds gates gate_first .. gate_first + gate_count - 1 of the layer's 2^k_i (the three arrays have
 * gate_count entries; gate_count may be 0).  z, W and the outputs are as for gkr_sumcheck_layer and identical on
 * all ranks.  Needs the host transcript; k_next <= GKR_MAX_K_NEXT as everywhere.
 *
 * allreduce: called twice per layer (three times never), each time with `count` canonical field elements in host
 * memory that it must replace by their sums over all ranks mod r (the same on every rank); 0 = success.  RCCL has
 * no modular sum: gkr_fr_widen / gkr_fr_narrow turn field elements into eight 32-bit limbs held in int64 (an
 * ordinary integer SUM all-reduce of those is exact for < 2^31 ranks) and back (gkr_amd/parallel.py does exactly
 * that over torch.distributed).  One extra element travels with the first exchange: "some rank saw a bad gate", so
 * that all ranks fail together instead of one leaving the others inside a collective. */
typedef int (*gkr_allreduce_fn)(void *user, gkr_fr *values, size_t count);
int  gkr_sumcheck_layer_sharded(gkr_ctx *ctx, int k_i, int k_next, uint64_t gate_first, uint64_t gate_count,
                                const uint8_t *gate_type, const uint32_t *left, const uint32_t *right, const gkr_fr *z,
                                const gkr_fr *W, gkr_allreduce_fn allreduce, void *user, gkr_fr *out_coeffs,
                                uint32_t *out_len, gkr_fr *out_r);
/* The same sumcheck with the gate arrays already in device memory (gkr_device_alloc + gkr_device_upload; gate_count
 * entries each, gates gate_first .. of the layer) -- nothing but z, W and the transcript crosses PCIe.  allreduce ==
 * NULL: the arrays hold the whole layer (gate_first = 0, gate_count = 2^k_i) and no exchange takes place, i.e.
 * gkr_sumcheck_layer on resident gates; otherwise as gkr_sumcheck_layer_sharded.  Gates are validated on the device. */
int  gkr_sumcheck_layer_device(gkr_ctx *ctx, int k_i, int k_next, uint64_t gate_first, uint64_t gate_count,
                               const void *d_gate_type, const void *d_left, const void *d_right, const gkr_fr *z,
                               const gkr_fr *W, gkr_allreduce_fn allreduce, void *user, gkr_fr *out_coeffs,
                               uint32_t *out_len, gkr_fr *out_r);
/* A layer's gates -- the whole layer, or one rank's contiguous share gate_first .. gate_first + gate_count -- kept in
 * device memory across sumchecks, together with the gate lists sorted from them on first use (what gkr_prove keeps per
 * circuit in its cache, for callers that drive prove_sumcheck_opt themselves: one circuit, many z / W).
 * gkr_resident_layer_sumcheck is gkr_sumcheck_layer_device on that layer: nothing but z, W and the transcript crosses
 * PCIe and no sort runs after the first call.  allreduce == NULL needs the whole layer. */
typedef struct gkr_resident_layer gkr_resident_layer;
int  gkr_resident_layer_create(gkr_ctx *ctx, int k_i, int k_next, uint64_t gate_first, uint64_t gate_count,
                               const uint8_t *gate_type, const uint32_t *left, const uint32_t *right,
                               gkr_resident_layer **out);
/* ... and with W (2^k_next values, canonical) already in DEVICE memory: what prover::prove has (the next layer's values come
 * from the forward evaluation, prover.rs:38-43) and what a host that drives prove_sumcheck_opt itself should do -- the
 * upload of a 2^20-value W is 0.5 ms of PCIe, a quarter of that layer's sumcheck.  Whole layers only (no exchange). */
int  gkr_resident_layer_sumcheck_wdev(gkr_ctx *ctx, gkr_resident_layer *layer, const gkr_fr *z, const void *d_W,
                                      gkr_fr *out_coeffs, uint32_t *out_len, gkr_fr *out_r);
int  gkr_resident_layer_sumcheck(gkr_ctx *ctx, gkr_resident_layer *layer, const gkr_fr *z, const gkr_fr *W,
                                 gkr_allreduce_fn allreduce, void *user, gkr_fr *out_coeffs, uint32_t *out_len,
                                 gkr_fr *out_r);
void gkr_resident_layer_free(gkr_ctx *ctx, gkr_resident_layer *layer);
/* The same exchange without leaving the device -- the multi-GPU twin of the rayon reduce at sumcheck.rs:50-63,
 * 97-124 as "one RCCL reduce over xGMI": the caller owns a device buffer of `capacity` int64 (capacity >=
 * gkr_exchange_limbs(k_next)); per exchange the library widens its partial tables into it on the device (eight
 * 32-bit limbs per field element, each in an int64, plus one flag element), calls fn(user, count, hip_stream), which
 * must enqueue an in-place integer SUM all-reduce of d_limbs[0 .. count) ON THAT STREAM (ncclAllReduce(buf, buf, count,
 * ncclInt64, ncclSum, comm, stream); torch.distributed.all_reduce under torch.cuda.ExternalStream(stream)), and
 * reduces the sums mod r on the device.  No host copy, no stream synchronisation: the host next touches the stream
 * when the first round's record lands.  Once a call has passed its argument checks and workspace allocations (which come
 * first and fail alike on every rank given the same arguments), fn is called on every rank the same number of times even
 * when a rank fails locally (the flag element carries "some rank failed", and all ranks then return an error together); fn
 * itself must not fail on one rank only.  (A rank whose device runs out of memory while its peers' do not returns before
 * the first exchange: give every rank the same budget.) */
typedef int (*gkr_allreduce_dev_fn)(void *user, size_t count, void *hip_stream);
typedef struct {
    gkr_allreduce_dev_fn fn;
    void *user;
    int64_t *d_limbs;   /* device memory owned by the caller */
    size_t capacity;    /* in int64 elements */
} gkr_exchange_dev;
size_t gkr_exchange_limbs(int k_next);
int  gkr_resident_layer_sumcheck_dev(gkr_ctx *ctx, gkr_resident_layer *layer, const gkr_fr *z, const gkr_fr *W,
                                     const gkr_exchange_dev *exchange, gkr_fr *out_coeffs, uint32_t *out_len,
                                     gkr_fr *out_r);
/* ---- a sum over ranks the library owns: gkr_exchange_dev backed by RCCL ----------------------------------------------
 * For hosts that are not Python (the reference's Rust binary): no callback of the host's own, no torch.  librccl is
 * loaded on first use (dlopen), not linked.  One rank makes the 128-byte id (gkr_exchange_rccl_unique_id) and hands it
 * to the others by whatever it has; every rank -- a process per GPU, or a thread per GPU of one process, the calls made
 * concurrently -- then calls gkr_exchange_rccl_create(device, id, rank, nranks, capacity), which blocks until all ranks
 * have arrived (ncclCommInitRank) and owns a device buffer of `capacity_limbs` int64.  gkr_exchange_rccl_dev() is the
 * gkr_exchange_dev to pass to gkr_resident_layer_sumcheck_dev / gkr_sumcheck_mle_sharded_dev: its hook queues
 * ncclAllReduce(buf, buf, count, ncclInt64, ncclSum, comm, stream) on the stream the library hands it.  Errors: a status,
 * the text in gkr_exchange_rccl_error() (per thread).  What it stands for in the reference: the rayon reduce at
 * sumcheck.rs:50-63, 97-124 (and :62 for prove_sumcheck), over xGMI instead of over cores.
 *
 * gkr_exchange_rccl_create BLOCKS until all `nranks` ranks have called it with the same id: RCCL's ncclCommInitRank has no
 * timeout, and the library adds none -- a rank that never arrives leaves the others waiting, exactly as in a bare RCCL
 * program; the host that starts the ranks owns that failure mode (tests/test_gpu_sharded.py shows the wait: rank 0 of a
 * world of two, alone, is still inside the call when its parent ends it).  The unit builds without RCCL's development
 * header (the few ABI types are declared locally) and returns GKR_ERR_UNSUPPORTED where librccl cannot be loaded.
 *
 * STATUS: EXPERIMENTAL beyond one device.  The three multi-device mechanisms -- this exchange, gkr_ctx_create_multi and
 * gkr_sumcheck_mle_sharded_dev / gkr_resident_layer_sumcheck_dev across ranks -- are proven bit-exact with logical ranks
 * on one GPU, over gloo with two processes, and through RCCL with ONE rank (torch's communicator and this one); no build
 * of this library has yet had a second physical device (tests/test_gpu_multi_device.py runs the two-device cases and
 * skips where fewer than two GPUs are visible). */
#define GKR_RCCL_ID_BYTES 128
typedef struct gkr_rccl_exchange gkr_rccl_exchange;
int  gkr_exchange_rccl_unique_id(void *id_out /* GKR_RCCL_ID_BYTES */);
int  gkr_exchange_rccl_create(int device_id, const void *unique_id, int rank, int nranks, size_t capacity_limbs,
                              gkr_rccl_exchange **out);
const gkr_exchange_dev *gkr_exchange_rccl_dev(const gkr_rccl_exchange *x);
uint64_t gkr_exchange_rccl_calls(const gkr_rccl_exchange *x);   /* all-reduces queued so far */
void gkr_exchange_rccl_destroy(gkr_rccl_exchange *x);
const char *gkr_exchange_rccl_error(void);

/* ---- one plain sumcheck split over GPUs: prove_sumcheck (sumcheck.rs:158-214), its reduce over the hypercube (the rayon
 * reduce of sumcheck.rs:62) as one RCCL all-reduce per PASS ------------------------------------------------------------
 * A table T of 2^n values is split over P = 2^log2_shards ranks; rank p holds the shard
 *     T_p[h * 2 + x_n] = T[h * 2P + 2p + x_n],   h < 2^(n - log2_shards - 1)
 * (index bits log2_shards .. 1 of an entry are its rank; the last variable stays inside every shard), `batch` shards
 * of 2^(n - log2_shards) values each, contiguous in device memory.  Rounds bind the leading variable, so every pair is
 * rank-local until 2^6 entries per shard are left: the library runs its multi-round passes on the shard (the kernels of
 * gkr_sumcheck_mle_batch_device), and per pass of J <= 5 rounds ONE in-place SUM all-reduce of batch * (2^J + 2) * 8
 * int64 through `exchange` completes the 2^J sub-block sums (they are linear in the table), on the library's stream;
 * every rank then hashes the same round vectors and binds the same challenges.  One more all-reduce gathers the 2^6
 * entries every shard has left, and all ranks finish the last 6 + log2_shards rounds on that tail.  n = 20 on 8 ranks:
 * 3 exchanges + 1 gather instead of 20 per-round reduces.  Every rank returns the whole transcript (outputs as
 * gkr_sumcheck_mle_batch_device: batch x n rows), bit-exact with the unsharded sumcheck.  "Does T depend on x_n" (the
 * last round's length) is a neighbour compare inside the shards, OR-ed over the ranks with the sums.
 * exchange: as for gkr_resident_layer_sumcheck_dev; capacity >= gkr_exchange_limbs_mle(n, log2_shards, batch).  A rank
 * that fails still enters every exchange (the flag travels with the sums) and all ranks return an error together.
 * *out_exchanges (may be NULL): how many times fn was called.  log2_shards = 0 is allowed (one rank; fn may then be a
 * no-op).  Needs the host transcript; 1 <= n - log2_shards <= GKR_MAX_MLE_N. */
size_t gkr_exchange_limbs_mle(int n, int log2_shards, int batch);
int  gkr_sumcheck_mle_sharded_dev(gkr_ctx *ctx, const void *d_shards, int n, int log2_shards, int shard, int batch,
                                  const gkr_exchange_dev *exchange, gkr_fr *out_coeffs, uint32_t *out_len, gkr_fr *out_r,
                                  uint32_t *out_exchanges);
/* host only: count field elements <-> count x 8 int64 (32-bit limbs, least significant first); narrow reduces
 * limb sums of up to 2^31 addends mod r */
int  gkr_fr_widen(const gkr_fr *values, size_t count, int64_t *limbs);
int  gkr_fr_narrow(const int64_t *limbs, size_t count, gkr_fr *values);

/* dense predicate tables A, M (2^{2 k_next} each) = add_i / mult_i restricted to z */
int  gkr_predicate_tables(gkr_ctx *ctx, int k_i, int k_next, const uint8_t *gate_type,
                          const uint32_t *left, const uint32_t *right, const gkr_fr *z,
                          gkr_fr *out_A, gkr_fr *out_M);

/* out[g] = prev[left[g]] (+ | *) prev[right[g]] */
int  gkr_layer_eval(gkr_ctx *ctx, size_t gates, const uint8_t *gate_type, const uint32_t *left,
                    const uint32_t *right, const gkr_fr *prev, size_t n_prev, gkr_fr *out);

/* ---- full GKR proof: prover::prove, prover.rs:6-9 ----------------------- */
/* A layered circuit as the reference's GKRCircuit holds it after
 * convert_r1cs_wtns_gkr (gkr.rs:53-114): layer 0 is the output layer; layer i
 * has 2^k[i] gates wired into layer i+1; the input layer has 2^k[depth] values. */
typedef struct {
    uint32_t depth;                 /* number of gate layers L (Proof.depth = L + 1) */
    const uint32_t *k;              /* L + 1 entries: k[0..L-1] gate layers, k[L] input layer */
    const uint8_t *const *gate_type;/* L arrays of 2^k[i] */
    const uint32_t *const *left;    /* L arrays of 2^k[i], values < 2^k[i+1] */
    const uint32_t *const *right;
} gkr_circuit_desc;

/* Caller-allocated proof buffers (sizes from the k list; gkr_proof_sizes fills
 * the counts).  Mirrors gkr.rs:7-19:
 *   sumcheck_coeffs  sum_i 2 k[i+1] rows x 3 slots (layer i's rows contiguous)
 *   sumcheck_len     one per row
 *   sumcheck_r       one per row
 *   q                per layer k[i+1]+1 slots right-aligned; q_len per layer
 *   z                k[0] + k[1] + ... + k[L] values (z[0] = 0s, prover.rs:16-21)
 *   r                L values (r*)
 *   d_coeffs         2^k[0] monomial coefficients of the output layer's MLE
 *                    (index bit pattern = which variables the monomial carries;
 *                    Proof.d is its non-zero entries, get_multi_ext poly.rs:502-536)
 *   input_coeffs     2^k[L] monomial coefficients of the input layer (Proof.input_func)
 */
typedef struct {
    gkr_fr *sumcheck_coeffs;
    uint32_t *sumcheck_len;
    gkr_fr *sumcheck_r;
    gkr_fr *q;
    uint32_t *q_len;
    gkr_fr *z;
    gkr_fr *r;
    gkr_fr *d_coeffs;
    gkr_fr *input_coeffs;
} gkr_proof_buf;

typedef struct {
    size_t rounds;        /* sum_i 2 k[i+1] */
    size_t q_slots;       /* sum_i (k[i+1] + 1) */
    size_t z_values;      /* sum_i k[i], i = 0..L */
    size_t d_coeffs;      /* 2^k[0] */
    size_t input_coeffs;  /* 2^k[L] */
} gkr_proof_sizes_t;

int  gkr_proof_sizes(const gkr_circuit_desc *circuit, gkr_proof_sizes_t *out);

/* input_values: the 2^k[L] values of the input layer (constants and witness
 * entries already gathered, convert.rs:796-810).  require_zero_output != 0
 * reproduces the reference's assert that output 0 is zero (convert.rs:838). */
int  gkr_prove(gkr_ctx *ctx, const gkr_circuit_desc *circuit, const gkr_fr *input_values,
               int require_zero_output, gkr_proof_buf *out);

/* `batch` proofs of ONE circuit for `batch` witnesses, advanced together (every layer's sumcheck runs
 * batched: one set of launches and one host round trip per round for all proofs) -- the reference proves
 * independent (circuit, input) pairs from a rayon par_iter (aggregator.rs:350-355).  input_values:
 * batch x 2^k[L] values; outs: `batch` caller-allocated proof buffers.  Needs the host transcript. */
int  gkr_prove_batch(gkr_ctx *ctx, const gkr_circuit_desc *circuit, const gkr_fr *input_values, int batch,
                     int require_zero_output, gkr_proof_buf *outs);

/* The proving step of one aggregation round in ONE call (aggregator.rs:341-355: prover::prove mapped over the
 * (circuit, input) pairs with a rayon par_iter): every item is a gkr_prove_batch (one circuit, `batch` witnesses); the
 * items are proven side by side, each by its own thread and child context (stream, workspaces, circuit cache) of `ctx`,
 * kept between calls.  A layer round of a small circuit is latency-bound -- launch, a tiny kernel, the hand-off, the
 * hash -- so independent items in flight fill each other's gaps; threads whose items are done, and threads waiting
 * for a round, take pieces of the others' host work (see gkr_host_help_while).  Items are dealt out by estimated cost,
 * the same way for the same list (so a child context finds its circuits in its cache on the next call).
 * LOCKSTEP GROUPS (round 5; option prove_many_lockstep, default on): items whose circuits share their k list -- the <= 20
 * sub-circuits of one compiled R1CS come in a few shapes -- are proven as ONE chain: one launch per pass and one hand-off
 * for the group, the passes over the gates reading each proof's own gate lists through a per-proof table.  A group holds at
 * most lockstep_max_proofs proofs (default 32: beyond that independent chains overlap better); an item that fails in a
 * group is proven again on its own, so statuses stay per item.  Same bytes as item by item.
 * The circuit cache of a context decides a hit by two 64-bit hashes over the k list and the gate arrays AND a comparison
 * with the gate arrays retained when the entry was made -- byte for byte up to 1 MiB of gate data, sampled 4 KiB blocks of
 * every array beyond: a context shared with callers that may hand in CRAFTED circuits larger than that should be given
 * GKR_NO_CIRCUIT_CACHE / no_circuit_cache (the proof would be wrong, not the memory unsafe: indices were range-checked).
 * max_concurrent: threads / child contexts to use at most; 0 = the CPUs this process may use, less two.
 * Every item's `status` is set; the return value is the first status that is not GKR_OK (gkr_last_error has its text). */
typedef struct {
    const gkr_circuit_desc *circuit;
    const gkr_fr *input_values;      /* batch x 2^k[L] */
    int batch;
    int require_zero_output;
    gkr_proof_buf *outs;             /* `batch` proof buffers */
    int status;                      /* out */
} gkr_prove_item;
int  gkr_prove_many(gkr_ctx *ctx, gkr_prove_item *items, size_t n_items, int max_concurrent);

/* ---- the reference's own argument types at `prove` (host only) ----------------------------------------------------
 * prover::prove(&GKRCircuit<S>, &Input<S>) (prover.rs:6-9) does not receive gate arrays and evaluation tables: a Layer
 * holds its wiring as `wire: (Vec<Vec<S>>, Vec<Vec<S>>)` (gkr.rs:35-51) -- per add gate and per mult gate one vector of
 * k_i + 2 k_next field elements 0 / 1, the bits of `gate || left || right`, most significant first (convert.rs:715-767,
 * convert_binary_to_vec) -- and Input.w[i] (gkr.rs:21-33) is layer i's multilinear extension as a TERM LIST: rows
 * [coeff, e_1 .. e_k] with exponents 0 / 1 (get_multi_ext, poly.rs:502-536; non-zero coefficients only, any order).
 * These three turn the one form into the other, so a binding can sit at `prove` itself without touching convert.rs
 * (INTEGRATION.md writes it out; tests/capi_dropin.c drives it from C):
 *   gkr_layer_from_wires   add_wire: n_add rows, mult_wire: n_mult rows of (k_i + 2 k_next) elements -> gate_type / left /
 *                          right of 2^k_i entries.  n_add + n_mult must be 2^k_i and every gate index appear once (every
 *                          slot of a compiled layer is a gate, convert.rs:209-214); an entry that is neither 0 nor 1, a gate
 *                          named twice or a count that does not fill the layer -> GKR_ERR_INVALID.
 *   gkr_values_from_terms  n_terms rows of (1 + k) elements -> the 2^k evaluations over the hypercube (index = the bit
 *                          string, variable 1 most significant): the inverse of get_multi_ext.  Equal monomials add up
 *                          (add_poly); an exponent other than 0 / 1 -> GKR_ERR_INVALID; n_terms = 0 gives the zero table.
 *   gkr_terms_from_coeffs  a table of 2^k monomial coefficients (gkr_proof_buf.d_coeffs / input_coeffs) -> the term list
 *                          Proof.d / Proof.input_func holds: the non-zero coefficients, ascending monomial index, rows of
 *                          (1 + k) elements.  *n_terms = how many there are; out_terms == NULL only counts;
 *                          GKR_ERR_NOMEM if capacity_terms is too small. */
int  gkr_layer_from_wires(int k_i, int k_next, const gkr_fr *add_wire, size_t n_add, const gkr_fr *mult_wire, size_t n_mult,
                          uint8_t *gate_type, uint32_t *left, uint32_t *right);
int  gkr_values_from_terms(int k, const gkr_fr *terms, size_t n_terms, gkr_fr *out_values);
int  gkr_terms_from_coeffs(int k, const gkr_fr *coeffs, gkr_fr *out_terms, size_t capacity_terms, size_t *n_terms);

/* prover::prove on exactly the reference's argument types, in ONE call (the three adapters above + gkr_prove): per layer i
 * the wire vectors of Layer.wire (add_wire[i]: n_add[i] rows, mult_wire[i]: n_mult[i] rows of k[i] + 2 k[i+1] elements; a
 * pointer may be NULL where its count is 0) and the input layer's term list Input.w[depth] (n_input_terms rows of 1 + k[depth]
 * elements).  Everything else as gkr_prove.  The statuses of the adapters pass through (GKR_ERR_INVALID: wire vectors that do
 * not describe a layer, a term that is not multilinear). */
typedef struct {
    uint32_t depth;                      /* number of gate layers L */
    const uint32_t *k;                   /* L + 1 entries (GKRCircuit::get_k_list) */
    const gkr_fr *const *add_wire;       /* L pointers */
    const size_t *n_add;                 /* L counts */
    const gkr_fr *const *mult_wire;
    const size_t *n_mult;
} gkr_wire_circuit;
int  gkr_prove_wires(gkr_ctx *ctx, const gkr_wire_circuit *circuit, const gkr_fr *input_terms, size_t n_input_terms,
                     int require_zero_output, gkr_proof_buf *out);

/* ---- verifier (host only, no device) ------------------------------------------------------------------------------
 * The relations the reference's verifier checks (python/gkr.py:202-231 with python/sumcheck.py:55-70; verifier.circom:39-71
 * holds the same inside a circuit) on the Rust prover's Proof, which carries no `f` / `add` / `mult` fields: per layer
 * every round's g_j(0) + g_j(1) = the running claim and r_j = multi_hash(g_j); the last claim =
 * add_i(z, b*, c*) (q(0) + q(1)) + mult_i(z, b*, c*) q(0) q(1) with the wiring predicates summed over the circuit's gates
 * (eq tables, O(gates) products on `threads` host threads; 0 = the CPUs this process may use); r* = multi_hash of the last
 * round vector; z[i+1] = l(r*); finally q(r*) of the last layer = input_func(z[L]).  z[0] must be zero (prover.rs:16-21).
 * *accept = 1 / 0; on a rejection *failed_layer / *failed_check (either may be NULL) name the first relation that failed.
 * The return value is a status of the CALL (GKR_ERR_INVALID: null pointers, a gate operand out of range), not the verdict. */
enum {
    GKR_VERIFY_OK = 0,
    GKR_VERIFY_SHAPE = 1,          /* a round vector / q length outside its range */
    GKR_VERIFY_NON_CANONICAL = 2,  /* a proof element >= r */
    GKR_VERIFY_Z0 = 3,             /* z[0] is not the zero vector */
    GKR_VERIFY_ROUND_SUM = 4,      /* g_j(0) + g_j(1) != claim */
    GKR_VERIFY_CHALLENGE = 5,      /* r_j != multi_hash(g_j) */
    GKR_VERIFY_FINAL_CLAIM = 6,    /* g_v(r_v) != add (q0 + q1) + mult q0 q1 */
    GKR_VERIFY_R_STAR = 7,         /* r* != multi_hash(last round vector) */
    GKR_VERIFY_NEXT_Z = 8,         /* z[i+1] != b* + r* (c* - b*) */
    GKR_VERIFY_INPUT = 9           /* q(r*) of the last layer != input_func(z[L]) */
};
int  gkr_verify(const gkr_circuit_desc *circuit, const gkr_proof_buf *proof, int threads, int *accept, uint32_t *failed_layer,
                uint32_t *failed_check);

/* ---- the proof as input signals of verifier.circom (host only) -----------
 * What the reference does with a Proof right after the path: pad its ragged vectors to the dimensions of
 * the generated verifier component (get_meta aggregator.rs:92-146, modify_proof_for_circom :148-213), print
 * field elements as decimal strings (file_utils.rs:20-28) and merge them into the circuit's input JSON
 * under keys suffixed with the proof's index (CircomInputProof aggregator.rs:20-82, file_utils.rs:49-67).
 * circuit: only depth and k are read.
 * gkr_circom_meta: [depth, largest k, k[0], #terms of D, longest round vector, longest q, #terms of the
 *   input function, k[L], k[0..L]] -- the VerifyGKR template arguments; *count = 8 + L + 1.
 * gkr_circom_input_json: one JSON object with the keys sumcheckProof<i>, sumcheckr<i>, q<i>, D<i>, z<i>,
 *   r<i>, inputFunc<i> (NUL-terminated).  *needed = bytes including the terminator; with out == NULL only
 *   the size is returned; GKR_ERR_NOMEM if capacity is too small.  Term order of D / inputFunc: ascending
 *   monomial index (the reference's order is HashMap order, i.e. unspecified). */
int  gkr_circom_meta(const gkr_circuit_desc *circuit, const gkr_proof_buf *proof, uint32_t *meta, size_t capacity,
                     size_t *count);
int  gkr_circom_input_json(const gkr_circuit_desc *circuit, const gkr_proof_buf *proof, int proof_index, char *out,
                           size_t capacity, size_t *needed);

/* The circom source the reference adds to the user's circuit for the next aggregation round (modify_circom_file,
 * aggregator.rs:215-314), as pure text functions (no circom is run): gkr_circom_verifier_source = the
 * `component verifier[n]` declaration and, per proof, the VerifyGKR(meta) instance, its input signals and the
 * wiring loops; metas = the proofs' meta vectors back to back, meta_len[i] entries each.  gkr_circom_inject = the
 * circuit text with the verifier include after the line `pragma circom 2.0.0;` and the source in front of the
 * first line that is exactly `}`.  Size protocol as gkr_circom_input_json. */
int  gkr_circom_verifier_source(const uint32_t *metas, const size_t *meta_len, size_t proofs, char *out, size_t capacity,
                                size_t *needed);
int  gkr_circom_inject(const char *circuit_text, const char *verifier_source, char *out, size_t capacity, size_t *needed);

/* ---- in front of the path: R1CS + witness -> layered circuits (host only) ------------------------------
 * What the reference does between circom's output files and prover::prove: read the iden3 `.r1cs` and `.wtns`
 * containers (third-party r1cs-file / wtns-file crates; aggregator.rs:341,345,399,404), turn every constraint
 * <A,w> * <B,w> - <C,w> = 0 into an expression tree (convert_constraints_to_nodes, convert.rs:360-632), and
 * compile the trees into at most 20 layered circuits whose every layer has 2^k add / mult gates over the next
 * one (compile, convert.rs:154-358; get_k :140-152).  gkr_layered_circuit hands out the gkr_circuit_desc that
 * gkr_prove / gkr_prove_batch take; gkr_layered_input_values gathers a witness into the input layer's values
 * (calculate_input, convert.rs:796-810) -- the forward evaluation and the "output 0 is zero" assertion
 * (:812-838) happen inside gkr_prove (require_zero_output).
 * Formats: `.r1cs` = magic "r1cs", version 1, sections (type u32, size u64): 1 header {field size 32, prime,
 * nWires, nPubOut, nPubIn, nPrvIn, nLabels u64, nConstraints}, 2 constraints {per constraint three linear
 * combinations: nTerms, then (wire u32, coefficient 32 B LE) per term}, 3 wire2label; `.wtns` = magic "wtns",
 * version 2, sections 1 {field size, prime, nWitness} and 2 {values, 32 B LE}.  Only BN254 Fr is accepted.
 * The writers exist because there is no circom in the build image: test fixtures have to be written. */
typedef struct gkr_r1cs gkr_r1cs;
typedef struct gkr_layered gkr_layered;
typedef struct {
    uint32_t n_wires, n_pub_out, n_pub_in, n_prv_in;
    uint64_t n_labels;
    size_t n_constraints, n_terms;   /* n_terms: over all A, B, C of all constraints */
} gkr_r1cs_info_t;

int  gkr_r1cs_parse(const void *bytes, size_t len, gkr_r1cs **out);
/* flat form: term_counts has 3 entries per constraint (A, B, C), wires / coeffs list all terms in that order */
int  gkr_r1cs_build(uint32_t n_wires, uint32_t n_pub_out, uint32_t n_pub_in, uint32_t n_prv_in, size_t n_constraints,
                    const uint32_t *term_counts, const uint32_t *wires, const gkr_fr *coeffs, gkr_r1cs **out);
int  gkr_r1cs_info(const gkr_r1cs *r1cs, gkr_r1cs_info_t *out);
int  gkr_r1cs_export(const gkr_r1cs *r1cs, uint32_t *term_counts, uint32_t *wires, gkr_fr *coeffs);
/* *needed = size of the file image; with out == NULL only the size is returned; GKR_ERR_NOMEM if too small */
int  gkr_r1cs_serialize(const gkr_r1cs *r1cs, void *out, size_t capacity, size_t *needed);
void gkr_r1cs_free(gkr_r1cs *r1cs);
int  gkr_wtns_parse(const void *bytes, size_t len, gkr_fr *out, size_t capacity, size_t *count);
int  gkr_wtns_serialize(const gkr_fr *values, size_t count, void *out, size_t capacity, size_t *needed);

/* bad_constraint (may be NULL): index of the constraint behind GKR_ERR_UNSUPPORTED */
int  gkr_r1cs_compile(const gkr_r1cs *r1cs, gkr_layered **out, size_t *bad_constraint);
int  gkr_layered_count(const gkr_layered *layered, uint32_t *circuits);
/* the arrays *out points to stay owned by `layered` */
int  gkr_layered_circuit(const gkr_layered *layered, uint32_t index, gkr_circuit_desc *out);
/* input layer of circuit `index`: slot s holds witness[wire[s]], or constant[s] where wire[s] == UINT32_MAX */
int  gkr_layered_input_layer(const gkr_layered *layered, uint32_t index, const uint32_t **wire, const gkr_fr **constant,
                             size_t *slots);
int  gkr_layered_input_values(const gkr_layered *layered, uint32_t index, const gkr_fr *witness, size_t n_witness,
                              gkr_fr *out_values);
void gkr_layered_free(gkr_layered *layered);

/* ---- step-wise sessions: one sumcheck split across GPUs -------------------
 * The reference reduces each round's per-assignment polynomials with a rayon
 * map-reduce (sumcheck.rs:50-63,65-78,97-124); across GPUs that reduce is one
 * tiny all-reduce per round.  The hypercube is partitioned by its TRAILING
 * log2(P) variables (rank p owns index low bits == p), so every pair of a
 * leading-variable round is rank-local.  A session does one rank's table work;
 * the caller owns the collective and the transcript (gkr_amd/parallel.py).
 * P = 1 gives the whole sumcheck with an external transcript. */
typedef struct gkr_layer_session gkr_layer_session;
typedef struct gkr_mle_session gkr_mle_session;

/* shard `shard` of `nshards` (power of two, <= 2^k_next) of the layer sumcheck:
 * keeps the gates whose right operand % nshards == shard; runs 2 k_next - log2(nshards) rounds */
int  gkr_layer_session_open(gkr_ctx *ctx, int k_i, int k_next, const uint8_t *gate_type,
                            const uint32_t *left, const uint32_t *right, const gkr_fr *z,
                            const gkr_fr *W, uint32_t nshards, uint32_t shard,
                            gkr_layer_session **out);
/* the redundant tail after the all-gather of the shards' last entries: tables of 2^kc entries */
int  gkr_layer_session_open_tables(gkr_ctx *ctx, int kc, const gkr_fr *A, const gkr_fr *M,
                                   const gkr_fr *wb, const gkr_fr *Wc, gkr_layer_session **out);
int  gkr_layer_session_dep(gkr_ctx *ctx, const gkr_layer_session *s, uint32_t *out_dep, uint32_t count);
int  gkr_layer_session_rounds(const gkr_layer_session *s, uint32_t *done, uint32_t *total);
/* this shard's partial sums of the current round: out[0] = c0, out[1] = g(1), out[2] = c2 */
int  gkr_layer_session_sums(gkr_ctx *ctx, gkr_layer_session *s, gkr_fr *out);
int  gkr_layer_session_bind(gkr_ctx *ctx, gkr_layer_session *s, const gkr_fr *r);
/* after the last local round: out = { A, M, Wc, W(b*) } of this shard */
int  gkr_layer_session_tail(gkr_ctx *ctx, gkr_layer_session *s, gkr_fr *out);
void gkr_layer_session_close(gkr_ctx *ctx, gkr_layer_session *s);

/* plain multilinear sumcheck on a device-resident table of 2^n entries (a shard or the whole table) */
int  gkr_mle_session_open(gkr_ctx *ctx, const void *d_table, int n, gkr_mle_session **out);
/* out[0] = sum of the low half, out[1] = sum of the high half of the current table */
int  gkr_mle_session_sums(gkr_ctx *ctx, gkr_mle_session *s, gkr_fr *out, uint32_t *out_dep);
int  gkr_mle_session_bind(gkr_ctx *ctx, gkr_mle_session *s, const gkr_fr *r);
int  gkr_mle_session_value(gkr_ctx *ctx, gkr_mle_session *s, gkr_fr *out);
void gkr_mle_session_close(gkr_ctx *ctx, gkr_mle_session *s);
int  gkr_device_tables_differ(gkr_ctx *ctx, const void *d_a, const void *d_b, size_t count,
                              uint32_t *out_differ);

/* ---- device memory helpers (so callers need no HIP of their own) -------- */
int  gkr_device_alloc(gkr_ctx *ctx, size_t bytes, void **d_ptr);
int  gkr_device_free(gkr_ctx *ctx, void *d_ptr);
int  gkr_device_upload(gkr_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int  gkr_device_download(gkr_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* synthetic table generator used by bench.py and the full-size parity tests
 * (definition: element i, limb j = mix64(seed + (4 i + j + 1) * 0x9E3779B97F4A7C15),
 * top limb masked to 61 bits; mix64 = splitmix64's finaliser) */
int  gkr_device_fill_table(gkr_ctx *ctx, void *d_table, size_t count, uint64_t seed);
/* the same stream of values, the 2^(n - log2_shards) entries rank `shard` of gkr_sumcheck_mle_sharded_dev holds of the
 * 2^n-entry table gkr_device_fill_table would write (bench.py --mode mle-split; no table ever exists in one piece) */
int  gkr_device_fill_shard(gkr_ctx *ctx, void *d_shard, int n, int log2_shards, int shard, uint64_t seed);
int  gkr_device_synchronize(gkr_ctx *ctx);
/* What this box itself gives, measured now (SURVEY section 8d asks bench.py to quote them beside the vendor
 * peaks): a plain device-to-device copy of `bytes` (>= 64 MiB; read + write counted) and a read-only stream, in
 * GB/s, and the chip-wide rate of 254-bit Montgomery products (dependent chains, 16 waves per SIMD) -- the
 * arithmetic ceiling of the gate passes of prove_sumcheck_opt (sumcheck.rs:50-63, 97-124).  No reference
 * counterpart: measurement only. */
/* host only: microseconds per Mimc7::multi_hash (sumcheck.rs:84,129,152) of a len-element round vector on ONE thread
 * of this host: sixteen transcripts side by side in AVX-512 IFMA lanes (time per hash; 0 without IFMA), and one
 * transcript on the scalar code.  The legs of the bench that are bound by the transcript quote their floor from it. */
int  gkr_ubench_host_hash(int len, double *us_per_hash_lanes16, double *us_per_hash_scalar);
int  gkr_ubench_ceilings(gkr_ctx *ctx, size_t bytes, double *copy_GBps, double *read_GBps, double *modmul_per_sec);

#ifdef __cplusplus
}
#endif
#endif /* GKR_AMD_H */
