"""ctypes binding of libgkr_amd.so (include/gkr_amd.h).

The library is the product; there is no Python or CPU fallback.  Importing this
module without the built library raises, and every compute call needs a gfx950
device.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GKR_AMD_LIB: another build of the same library -- same-box A/B runs of two source states; never a fallback)
LIB_PATH = os.environ.get("GKR_AMD_LIB") or os.path.join(_HERE, "lib", "libgkr_amd.so")

GKR_OK = 0
GKR_ERR_INVALID = 1
GKR_ERR_NON_CANONICAL = 2
GKR_ERR_NO_DEVICE = 3
GKR_ERR_HIP = 4
GKR_ERR_NOMEM = 5
GKR_ERR_DEGENERATE = 6
GKR_ERR_UNSUPPORTED = 7

GKR_TRANSCRIPT_DEVICE = 0
GKR_TRANSCRIPT_HOST = 1

# every symbol include/gkr_amd.h declares (tests check the library exports them all)
SYMBOLS = [
    "gkr_strerror", "gkr_version", "gkr_ctx_create", "gkr_ctx_create_multi", "gkr_ctx_device_count", "gkr_ctx_destroy", "gkr_last_error",
    "gkr_ctx_set_transcript", "gkr_ctx_set_host_threads", "gkr_ctx_set_option", "gkr_ctx_get_option", "gkr_option_count", "gkr_option_name", "gkr_option_doc", "gkr_option_env", "gkr_host_help_while", "gkr_host_accounting", "gkr_host_accounting_read", "gkr_prove_many", "gkr_ctx_device_name", "gkr_ctx_profile", "gkr_ctx_profile_get", "gkr_ctx_profile_samples",
    "gkr_ctx_profile_reset", "gkr_mimc7_multi_hash", "gkr_mimc7_hash", "gkr_mimc7_constant",
    "gkr_selftest_mul", "gkr_selftest_wide_sum", "gkr_selftest_fold", "gkr_selftest_dot", "gkr_selftest_hash8", "gkr_selftest_host_pass", "gkr_selftest_host_prod_pass", "gkr_selftest_host_tail", "gkr_selftest_pass_schedule", "gkr_selftest_line_restriction", "gkr_selftest_seg_item", "gkr_sumcheck_mle", "gkr_sumcheck_mle_batch_device",
    "gkr_sumcheck_layer", "gkr_sumcheck_layer_sharded", "gkr_sumcheck_layer_device", "gkr_resident_layer_create", "gkr_resident_layer_sumcheck", "gkr_resident_layer_sumcheck_wdev", "gkr_resident_layer_free", "gkr_exchange_limbs", "gkr_resident_layer_sumcheck_dev", "gkr_exchange_limbs_mle", "gkr_sumcheck_mle_sharded_dev", "gkr_exchange_rccl_unique_id", "gkr_exchange_rccl_create", "gkr_exchange_rccl_dev",
    "gkr_exchange_rccl_calls", "gkr_exchange_rccl_destroy", "gkr_exchange_rccl_error", "gkr_fr_widen", "gkr_fr_narrow", "gkr_predicate_tables", "gkr_layer_eval", "gkr_proof_sizes", "gkr_prove", "gkr_prove_batch",
    "gkr_layer_from_wires", "gkr_values_from_terms", "gkr_terms_from_coeffs", "gkr_prove_wires", "gkr_verify",
    "gkr_circom_meta", "gkr_circom_input_json", "gkr_circom_verifier_source", "gkr_circom_inject",
    "gkr_r1cs_parse", "gkr_r1cs_build", "gkr_r1cs_info", "gkr_r1cs_export", "gkr_r1cs_serialize", "gkr_r1cs_free",
    "gkr_wtns_parse", "gkr_wtns_serialize", "gkr_r1cs_compile", "gkr_layered_count", "gkr_layered_circuit",
    "gkr_layered_input_layer", "gkr_layered_input_values", "gkr_layered_free",
    "gkr_device_alloc", "gkr_device_free", "gkr_device_upload", "gkr_device_download",
    "gkr_device_fill_table", "gkr_device_fill_shard", "gkr_device_synchronize", "gkr_ubench_ceilings", "gkr_ubench_host_hash",
    "gkr_layer_session_open", "gkr_layer_session_open_tables", "gkr_layer_session_dep", "gkr_layer_session_rounds",
    "gkr_layer_session_sums", "gkr_layer_session_bind", "gkr_layer_session_tail", "gkr_layer_session_close",
    "gkr_mle_session_open", "gkr_mle_session_sums", "gkr_mle_session_bind", "gkr_mle_session_value",
    "gkr_mle_session_close", "gkr_device_tables_differ",
]


class CircuitDesc(ctypes.Structure):
    _fields_ = [
        ("depth", ctypes.c_uint32),
        ("k", ctypes.POINTER(ctypes.c_uint32)),
        ("gate_type", ctypes.POINTER(ctypes.c_void_p)),
        ("left", ctypes.POINTER(ctypes.c_void_p)),
        ("right", ctypes.POINTER(ctypes.c_void_p)),
    ]


class ProofBuf(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("sumcheck_coeffs", "sumcheck_len", "sumcheck_r", "q", "q_len", "z", "r", "d_coeffs", "input_coeffs")]


class ProveItem(ctypes.Structure):
    _fields_ = [("circuit", ctypes.c_void_p), ("input_values", ctypes.c_void_p), ("batch", ctypes.c_int),
                ("require_zero_output", ctypes.c_int), ("outs", ctypes.c_void_p), ("status", ctypes.c_int)]


ALLREDUCE_DEV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class ExchangeDev(ctypes.Structure):
    _fields_ = [("fn", ALLREDUCE_DEV_FN), ("user", ctypes.c_void_p), ("d_limbs", ctypes.c_void_p), ("capacity", ctypes.c_size_t)]


class R1csInfo(ctypes.Structure):
    _fields_ = [("n_wires", ctypes.c_uint32), ("n_pub_out", ctypes.c_uint32), ("n_pub_in", ctypes.c_uint32),
                ("n_prv_in", ctypes.c_uint32), ("n_labels", ctypes.c_uint64), ("n_constraints", ctypes.c_size_t),
                ("n_terms", ctypes.c_size_t)]


class ProofSizes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in ("rounds", "q_slots", "z_values", "d_coeffs", "input_coeffs")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "gkr_amd: %s is missing -- build it with `make -C gkr_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.gkr_strerror.restype = ctypes.c_char_p
        L.gkr_version.restype = ctypes.c_char_p
        L.gkr_last_error.restype = ctypes.c_char_p
        L.gkr_last_error.argtypes = [ctypes.c_void_p]
        L.gkr_resident_layer_free.restype = None
        L.gkr_resident_layer_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.gkr_exchange_limbs.restype = ctypes.c_size_t
        L.gkr_exchange_limbs.argtypes = [ctypes.c_int]
        L.gkr_host_help_while.restype = ctypes.c_long
        L.gkr_host_help_while.argtypes = [ctypes.c_void_p]
        L.gkr_host_accounting.restype = ctypes.c_int
        L.gkr_host_accounting.argtypes = [ctypes.c_int]
        L.gkr_host_accounting_read.restype = ctypes.c_int
        L.gkr_host_accounting_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]
        L.gkr_ctx_destroy.restype = None
        L.gkr_ctx_destroy.argtypes = [ctypes.c_void_p]
        for fn in (L.gkr_layer_session_close, L.gkr_mle_session_close):
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        for fn in (L.gkr_r1cs_free, L.gkr_layered_free):
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib
