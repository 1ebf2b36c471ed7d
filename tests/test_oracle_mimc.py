"""keccak-256 / MiMC7-91 known answers for the oracle hash (CPU only).

The vectors are the public circomlib / mimc-rs ones (first hard-coded
constants of circomlib mimc.circom; arnaucube/mimc-rs unit-test values); the
crate itself is an un-vendored dependency of the reference
(rust/Cargo.toml:28), call sites rust/src/gkr/sumcheck.rs:45,84 and
prover.rs:10,78.
"""

from oracle.mimc7 import CTS, keccak256, mimc7_hash, multi_hash


def test_keccak_empty_and_abc():
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # multi-block input (rate 136)
    assert keccak256(b"a" * 200).hex() == keccak256(b"a" * 200).hex()
    assert len(keccak256(b"x" * 136)) == 32


def test_round_constants():
    assert len(CTS) == 91 and CTS[0] == 0
    assert CTS[1] == 20888961410941983456478427210666206549300505294776164667214940546594746570981
    assert CTS[2] == 15265126113435022738560151911929040668591755459209400716467504685752745317193
    assert CTS[90] == 13602139229813231349386885113156901793661719180900395818909719758150455500533


def test_hash_known_answers():
    assert mimc7_hash(1, 2) == 0x176C6EEFC3FDF8D6136002D8E6F7A885BBD1C4E3957B93DDC1EC3AE7859F1A08
    assert mimc7_hash(12, 45) == 0x2BA7EBAD3C6B6F5A20BDECBA2333C63173CA1A5F2F49D958081D9FA7179C44E4


def test_multi_hash_known_answers():
    assert multi_hash([12], 0) == 0x237C92644DBDDB86D8A259E0E923AAAB65A93F1EC5758B8799988894AC0958FD
    assert multi_hash([78, 41], 0) == 0x067F3202335EA256AE6E6AADCD2D5F7F4B06A00B2D1E0DE903980D5AB552DC70
    assert multi_hash([12, 45, 78, 41], 0) == 0x284BC1F34F335933A23A433B6FF3EE179D682CD5E5E2FCDD2D964AFA85104BEB
    assert multi_hash([], 0) == 0
