"""gkr_amd -- MI355X-native GKR sumcheck prover (hot path of jeong0982/gkr).

The numeric work lives in lib/libgkr_amd.so (hand-written HIP for gfx950 behind
the C ABI of include/gkr_amd.h).  This package is the host-side mirror of the
reference's prover interface; it has no CPU fallback.
"""

from .field import MODULUS, from_limbs, to_limbs
from .prover import (Context, GKRCircuit, GkrError, Layer, Proof, default_context, multi_hash, prove,
                     prove_sumcheck, prove_sumcheck_opt)

from .verifier import verify
from .aggregate import aggregated_input, circom_input, circom_meta

__all__ = ["verify", "aggregated_input", "circom_input", "circom_meta", "MODULUS", "from_limbs", "to_limbs", "Context", "GKRCircuit", "GkrError", "Layer", "Proof",
           "default_context", "multi_hash", "prove", "prove_sumcheck", "prove_sumcheck_opt"]
