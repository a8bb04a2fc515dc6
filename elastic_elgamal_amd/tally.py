"""Tally stage (examples/voting.rs:122-177): combine verified decryption shares by Lagrange interpolation
in the exponent and read the vote counts off a discrete-log table.

Thin ctypes mirror of the C entry points (`eg_combine_shares`, `eg_dlog_table_*`, include/eg_hip.h), which mirror
``lagrange_coefficients`` / ``Params::combine_shares`` (src/sharing/mod.rs:139-170,302-325) and ``DiscreteLogTable``
(src/encryption.rs:260-298).  All scalar and group arithmetic runs on the GPU primitives inside the library.
"""
from __future__ import annotations

import ctypes as C

from . import _check, _load


def combine_shares(group, threshold: int, shares, n_shares: int | None = None):
    """``Params::combine_shares``: shares = [(participant index, dh_element bytes)]; the first `threshold` are used.
    Returns the combined dh element [x]R, or None if there are too few shares.  `n_shares` = participants of the key set
    (default: large enough for the indexes given)."""
    shares = list(shares)
    n = len(shares)
    total = n_shares if n_shares is not None else max([i for i, _ in shares] + [threshold - 1]) + 1
    idx = (C.c_uint64 * max(n, 1))(*[i for i, _ in shares])
    out = C.create_string_buffer(32)
    combined = C.c_int(0)
    _check(_load().eg_combine_shares(group.ctx._h, total, threshold, n, idx, b"".join(s for _, s in shares), out, C.byref(combined)))
    return out.raw if combined.value else None


class DiscreteLogTable:
    """``DiscreteLogTable::new(values)``: maps [m]G (canonical encoding) back to m."""

    def __init__(self, group, values):
        values = list(values)
        arr = (C.c_uint64 * max(len(values), 1))(*values)
        self._h = C.c_void_p()
        _check(_load().eg_dlog_table_create(group.ctx._h, len(values), arr, C.byref(self._h)))

    def get(self, element: bytes):
        v, found = C.c_uint64(0), C.create_string_buffer(1)
        _check(_load().eg_dlog_table_get(self._h, 1, element, C.byref(v), found))
        return int(v.value) if found.raw[0] else None

    def close(self):
        if getattr(self, "_h", None):
            _load().eg_dlog_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decrypt_total(group, table: DiscreteLogTable, ciphertext: bytes, combined_dh: bytes):
    """``VerifiableDecryption::decrypt``: blinded_element - dh looked up in the table (None if absent)."""
    m, ok = group.element_add(ciphertext[32:64], combined_dh, subtract=True)
    return table.get(m)
