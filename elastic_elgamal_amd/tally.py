"""Tally stage on top of the GPU primitives (examples/voting.rs:122-177): combine verified
decryption shares by Lagrange interpolation in the exponent and read the vote counts off a discrete-log table.

Host code only schedules; every group operation runs on the GPU through :class:`elastic_elgamal_amd.Ristretto`.
Mirrors ``lagrange_coefficients`` / ``Params::combine_shares`` (src/sharing/mod.rs:139-170,302-325) and
``DiscreteLogTable`` (src/encryption.rs:260-298).
"""
from __future__ import annotations

L = 2**252 + 27742317777372353535851937790883648493


def _sc(x: int) -> bytes:
    return (x % L).to_bytes(32, "little")


def lagrange_coefficients(indexes):
    """(denominators^-1, scale) exactly as src/sharing/mod.rs:139-170 (zero-based indexes, points index + 1)."""
    inv = []
    for i in indexes:
        d = 1
        for j in indexes:
            d = d * ((i + 1) if i == j else (j - i)) % L      # sign folded in: (j - i) mod l
        inv.append(pow(d, L - 2, L))
    scale = 1
    for i in indexes:
        scale = scale * (i + 1) % L
    return inv, scale


def combine_shares(group, threshold: int, shares):
    """``Params::combine_shares``: shares = [(participant index, dh_element bytes)], at least `threshold` of them.
    Returns the combined dh element [x]R, or None if there are too few shares."""
    shares = list(shares)[:threshold]
    if len(shares) < threshold:
        return None
    idx = [i for i, _ in shares]
    inv, scale = lagrange_coefficients(idx)
    # restored = sum_i [denominator_i^-1] share_i ; result = [scale] restored
    scalars = b"".join(_sc(c * scale) for c in inv)
    out, ok = group.vartime_multi_mul(len(shares), scalars, b"".join(s for _, s in shares))
    if ok != b"\x01":
        raise ValueError("invalid decryption share element")
    return out


class DiscreteLogTable:
    """``DiscreteLogTable::new(values)``: maps [m]G (canonical encoding) back to m."""

    def __init__(self, group, values):
        values = list(values)
        enc = group.mul_generator(b"".join(_sc(v) for v in values))
        self.table = {enc[32 * k : 32 * k + 32]: v for k, v in enumerate(values)}

    def get(self, element: bytes):
        return self.table.get(element)


def decrypt_total(group, table: DiscreteLogTable, ciphertext: bytes, combined_dh: bytes):
    """``VerifiableDecryption::decrypt``: blinded_element - dh looked up in the table (None if absent)."""
    m, ok = group.element_add(ciphertext[32:64], combined_dh, subtract=True)
    return table.get(m)
