"""CPU oracle for the GKR sumcheck hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, link or execute it, and there only as the checker / the timed CPU
baseline -- never as the thing being measured or shipped.  The product path
(``gkr_amd``) fails loudly when its HIP library is missing; it never falls
back to this code.

Contents
  field.py     BN254 Fr constants (halo2curves bn256::Fr, rust/Cargo.toml:21)
  mimc7.py     keccak-256 + MiMC7-91 (third-party mimc-rs; call sites
               rust/src/gkr/sumcheck.rs:45,84,129,152 and prover.rs:10,78)
  termlist.py  term-list restatement of rust/src/gkr/{poly,sumcheck,prover}.rs
               and the wiring / input builders of rust/src/convert.rs:703-849
  dense.py     dense-table algorithm (pure Python ints), shown equal to
               termlist.py by tests/test_oracle_equivalence.py
  c/           plain-C dense oracle (gcc), used for 2^16..2^20 sizes and as the
               timed CPU baseline ("port")

Pinning status (see DESIGN.md "Oracle"):
  * termlist.py / dense.py are pinned against golden vectors produced by
    importing the reference's own Python prover (python/gkr.py,
    python/sumcheck.py, python/poly.py) in the build container -- see
    tests/golden/make_golden.py.  The reference's one missing third-party
    import (``ethsnarks``: FQ field class and mimc_hash) is supplied by a
    stand-in inside that script; no reference code is replaced.
  * MiMC7 itself lives in the un-vendored, un-pinned git dependency
    ``jeong0982/mimc-rs`` (rust/Cargo.toml:28).  It is restated from the
    published circomlib MiMC7 algorithm and checked against the public
    circomlib/mimc-rs known answers; against the fork the reference links it
    is "parity unpinned".
  * The Rust-only behaviours the Python reference does not share (z[0] = 0,
    short round vectors when W does not depend on a variable) are restated
    from the Rust sources line by line and are NOT pinned by an executed
    reference (no Rust toolchain in the image): "parity unpinned" for those
    two rules.
"""
