"""BN254 scalar field constants (oracle side; test infrastructure only).

The reference's field is ``halo2curves::bn256::Fr`` (rust/src/aggregator.rs:9,
rust/Cargo.toml:21).  Elements cross every boundary as 32-byte little-endian
canonical representations (rust/src/gkr/sumcheck.rs:10-22).
"""

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
assert P == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

R = (1 << 256) % P          # Montgomery radix used by both the C oracle and the product
R2 = (R * R) % P
INV64 = (-pow(P, -1, 1 << 64)) % (1 << 64)
INV32 = (-pow(P, -1, 1 << 32)) % (1 << 32)


def to_le_bytes(x: int) -> bytes:
    return (x % P).to_bytes(32, "little")


def from_le_bytes(b: bytes) -> int:
    v = int.from_bytes(b, "little")
    if v >= P:
        raise ValueError("non-canonical field element")
    return v


def to_limbs64(x: int):
    x %= P
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs64(l) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))
