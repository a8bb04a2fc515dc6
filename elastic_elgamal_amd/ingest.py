"""Verification of ballots that arrive as deserialised objects (the reference's serde layout, `serde.py`) rather than
as packed bytes: one verdict per ballot, including the variants the packed layout cannot express.

A packed batch has one shape, so `OptionsLenMismatch` (src/app/choice.rs:149-158,365) and `LenMismatch`
(src/proofs/mod.rs:73-99; checked at ring.rs:311-316, range.rs:556-560, mul.rs:199-204) never arise inside the C ABI.
They arise here, where a ballot object may carry any number of choices / responses / partial ciphertexts.  The order of
events of the reference is kept:

1. deserialisation: an object that is structurally malformed (a string that is not unpadded base64url of 32 bytes, fewer
   than 2 `ring_responses` / `ciphertext_responses`, a `sum_proof` that does not match the election kind, a missing field)
   fails on its own with `Malformed` - in the reference a serde failure is per object, so one voter's junk never blocks the
   verification of the others; then the first element (document order) that is not a canonical scalar / valid point fails
   the ballot (`BadScalar` / `BadPoint` with the index of the item, serde.rs:191-206,254-269);
2. `check_options_count`;
3. the proofs in the order of `verify`, each beginning with its own length check.  A length mismatch in a later proof is
   only reported if the earlier proofs verify, so well-shaped earlier parts ARE verified on the GPU: the ill-shaped part is
   replaced by an all-zero proof of the right shape (which cannot verify) and the verdict is read off the status word.

Nothing here computes group or field arithmetic: validity of elements and every proof is decided by the GPU library.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field

from . import range_decomposition
from .serde import SerdeError, b64url_decode

ST_OK, ST_BAD_SCALAR, ST_BAD_POINT, ST_OPTIONS_LEN, ST_SUM_CHALLENGE, ST_RANGE_LEN, ST_RANGE_CHALLENGE = 0, 1, 2, 3, 4, 5, 6
ST_QV_VARIANT_LEN, ST_QV_VARIANT_CHALLENGE, ST_QV_CREDIT_RANGE_LEN, ST_QV_CREDIT_RANGE_CHALLENGE = 7, 8, 9, 10
ST_QV_CREDIT_EQUIV_LEN, ST_QV_CREDIT_EQUIV_CHALLENGE = 11, 12
ST_MALFORMED = 13      # the object does not deserialise (EG_ST_MALFORMED): that voter's ballot only, the batch goes on
ZERO = bytes(32)


def status(kind: int, detail: int = 0) -> int:
    return kind | (detail << 8)


@dataclass
class _Items:
    """32-byte items of one ballot in document order, tagged 'P' (group element) or 'S' (scalar)."""
    kinds: list = field(default_factory=list)
    data: list = field(default_factory=list)

    def point(self, s):
        self.kinds.append("P"); self.data.append(b64url_decode(s))

    def scalar(self, s):
        self.kinds.append("S"); self.data.append(b64url_decode(s))

    def ct(self, c):
        self.point(c["random_element"]); self.point(c["blinded_element"])

    def ring(self, p):
        if len(p["ring_responses"]) < 2:
            raise SerdeError("invalid length of ring_responses, expected at least 2")      # VecHelper<_, 2>, serde.rs:303-355
        self.scalar(p["common_challenge"])
        for r in p["ring_responses"]:
            self.scalar(r)


def _first_invalid(items_per_ballot, grp):
    """Status of the first invalid item of each ballot (document order), or None: one batched call per item type."""
    pts = b"".join(d for it in items_per_ballot for k, d in zip(it.kinds, it.data) if k == "P")
    scs = b"".join(d for it in items_per_ballot for k, d in zip(it.kinds, it.data) if k == "S")
    pok = grp.element_roundtrip(pts)[1] if pts else b""
    sok = grp.deserialize_scalar_ok(scs) if scs else b""
    out, pi, si = [], 0, 0
    for it in items_per_ballot:
        st = None
        for idx, k in enumerate(it.kinds):
            if k == "P":
                good = pok[pi]; pi += 1
            else:
                good = sok[si]; si += 1
            if not good and st is None:
                st = status(ST_BAD_POINT if k == "P" else ST_BAD_SCALAR, idx)
        out.append(st)
    return out


def parse_range(text: str):
    """'3 * 0..7 + 0..3' -> [(step, size), ...] as RangeDecomposition's Display prints it (range.rs:110-124)."""
    rings = []
    for term in text.split(" + "):
        m = re.fullmatch(r"(?:(\d+) \* )?0\.\.(\d+)", term.strip())
        if not m:
            raise ValueError(f"cannot parse range decomposition {text!r}")
        rings.append((int(m.group(1) or 1), int(m.group(2))))
    return rings


# ------------------------------------------------------------------------------------------------ EncryptedChoice
def verify_choice_objects(params, grp, objs):
    """`obj.verify(&params)` for EncryptedChoice objects (dicts in the serde layout).  Returns (status words, tally of
    the accepted ballots as n_options x 64 bytes).  `grp` is the `Ristretto` primitive backend of the same context."""
    n, single = params.n_options, params.single
    statuses = [None] * len(objs)
    good_idx, good_packed = [], []
    odd_idx, odd_items, odd_objs = [], [], []
    for i, o in enumerate(objs):
        it = _Items()
        try:
            for c in o["choices"]:
                it.ct(c)
            it.ring(o["range_proof"])
            sp = o.get("sum_proof")
            if single != (sp is not None):
                raise SerdeError("sum_proof does not match the kind of election")      # S::Proof is a different type
            if sp is not None:
                it.scalar(sp["challenge"]); it.scalar(sp["response"])
        except (SerdeError, KeyError, TypeError, AttributeError):
            statuses[i] = status(ST_MALFORMED)
            continue
        if len(o["choices"]) == n and len(o["range_proof"]["ring_responses"]) == 2 * n:
            good_idx.append(i); good_packed.append(b"".join(it.data))
        else:
            odd_idx.append(i); odd_items.append(it); odd_objs.append(o)
    tally = None
    if good_idx or not objs:
        st, tally = params.verify_batch(b"".join(good_packed))
        for i, s in zip(good_idx, st):
            statuses[i] = s
    if tally is None:
        tally = params.verify_batch(b"")[1]
    if odd_idx:
        substitutes, sub_for = [], []
        for i, it, o, bad in zip(odd_idx, odd_items, odd_objs, _first_invalid(odd_items, grp)):
            if bad is not None:
                statuses[i] = bad
            elif len(o["choices"]) != n:
                statuses[i] = status(ST_OPTIONS_LEN)
            elif not single:
                statuses[i] = status(ST_RANGE_LEN)          # nothing is verified before the ring proof
            else:                                            # the sum proof is verified before the ring proof's length check
                cts = b"".join(it.data[: 2 * n])
                substitutes.append(cts + ZERO * (1 + 2 * n) + b"".join(it.data[-2:]))
                sub_for.append(i)
        if substitutes:
            st, _ = params.verify_batch(b"".join(substitutes), with_tally=False)
            for i, s in zip(sub_for, st):
                statuses[i] = s if (s & 0xFF) == ST_SUM_CHALLENGE else status(ST_RANGE_LEN)
    return statuses, tally


# ------------------------------------------------------------------------------------------------ QuadraticVotingBallot
def isqrt(x: int) -> int:
    import math
    return math.isqrt(x)


def verify_qv_objects(params, grp, objs):
    """`obj.verify(&params)` for QuadraticVotingBallot objects.  Returns (status words, tally of the accepted ballots)."""
    n = params.n_options
    vote_rings = parse_range(range_decomposition(isqrt(params.credits) + 1))      # quadratic_voting.rs:63-76
    credit_rings = parse_range(range_decomposition(params.credits + 1))
    statuses = [None] * len(objs)
    good_idx, good_packed, odd = [], [], []

    def range_shape_ok(p, rings):
        return len(p["partial_ciphertexts"]) == len(rings) - 1 and len(p["ring_responses"]) == sum(s for _, s in rings)

    def dummy_range(rings):
        return ZERO * (2 * (len(rings) - 1)) + ZERO * (1 + sum(s for _, s in rings))

    for i, o in enumerate(objs):
        it = _Items()
        spans = []                      # (first item, end item) of every vote / credit block and of the final proof
        try:
            for v in list(o["votes"]) + [o["credit"]]:
                a = len(it.data)
                it.ct(v["ciphertext"])
                for c in v["range_proof"]["partial_ciphertexts"]:
                    it.ct(c)
                it.ring(v["range_proof"])
                spans.append((a, len(it.data)))
            p = o["credit_equivalence_proof"]
            a = len(it.data)
            if len(p["ciphertext_responses"]) < 2:
                raise SerdeError("invalid length of ciphertext_responses, expected at least 2")   # VecHelper<_, 2>, mul.rs:89-90
            it.scalar(p["challenge"])
            for r in p["ciphertext_responses"]:
                it.scalar(r)
            it.scalar(p["sum_response"])
            spans.append((a, len(it.data)))
        except (SerdeError, KeyError, TypeError, AttributeError):
            statuses[i] = status(ST_MALFORMED)
            continue
        shapes = [range_shape_ok(v["range_proof"], vote_rings) for v in o["votes"]]
        shapes.append(range_shape_ok(o["credit"]["range_proof"], credit_rings))
        shapes.append(len(p["ciphertext_responses"]) == 2 * len(o["votes"]))
        if len(o["votes"]) == n and all(shapes):
            good_idx.append(i); good_packed.append(b"".join(it.data))
        else:
            odd.append((i, it, o, spans, shapes))
    tally = None
    if good_idx or not objs:
        st, tally = params.verify_batch(b"".join(good_packed))
        for i, s in zip(good_idx, st):
            statuses[i] = s
    if tally is None:
        tally = params.verify_batch(b"")[1]
    if odd:
        substitutes, sub_for = [], []
        for (i, it, o, spans, shapes), bad in zip(odd, _first_invalid([x[1] for x in odd], grp)):
            if bad is not None:
                statuses[i] = bad
                continue
            if len(o["votes"]) != n:
                statuses[i] = status(ST_OPTIONS_LEN)
                continue
            first_bad = shapes.index(False)             # 0..n-1 votes, n credit range, n+1 credit equivalence
            if first_bad == 0:
                statuses[i] = status(ST_QV_VARIANT_LEN, 0)   # nothing is verified before the first vote's length check
                continue
            blocks = []
            for k, (a, b) in enumerate(spans):
                if shapes[k]:
                    blocks.append(b"".join(it.data[a:b]))
                elif k < n:
                    blocks.append(b"".join(it.data[a : a + 2]) + dummy_range(vote_rings))
                elif k == n:
                    blocks.append(b"".join(it.data[a : a + 2]) + dummy_range(credit_rings))
                else:
                    blocks.append(ZERO * (2 + 2 * n))
            substitutes.append(b"".join(blocks))
            sub_for.append((i, first_bad))
        if substitutes:
            st, _ = params.verify_batch(b"".join(substitutes), with_tally=False)
            for (i, first_bad), s in zip(sub_for, st):
                kind, detail = s & 0xFF, s >> 8
                # position of the reported failure in verify's order: votes 0..n-1, credit range, credit equivalence
                pos = detail if kind in (ST_QV_VARIANT_LEN, ST_QV_VARIANT_CHALLENGE) else n if kind == ST_QV_CREDIT_RANGE_CHALLENGE else n + 1
                if kind != ST_OK and pos < first_bad:
                    statuses[i] = s                       # an earlier, well-shaped proof fails first
                elif first_bad < n:
                    statuses[i] = status(ST_QV_VARIANT_LEN, first_bad)
                elif first_bad == n:
                    statuses[i] = status(ST_QV_CREDIT_RANGE_LEN)
                else:
                    statuses[i] = status(ST_QV_CREDIT_EQUIV_LEN)
    return statuses, tally


def unpack_qv_ballot(packed: bytes, n_options: int, credits: int) -> dict:
    """Packed QuadraticVotingBallot -> serde-layout object (inverse of serde.pack_qv_ballot)."""
    from .serde import b64url_encode

    def ct(b):
        return {"random_element": b64url_encode(b[:32]), "blinded_element": b64url_encode(b[32:64])}

    def block(buf, off, rings):
        c = ct(buf[off : off + 64]); off += 64
        partial = [ct(buf[off + 64 * k : off + 64 * k + 64]) for k in range(len(rings) - 1)]
        off += 64 * (len(rings) - 1)
        e0 = b64url_encode(buf[off : off + 32]); off += 32
        m = sum(s for _, s in rings)
        resp = [b64url_encode(buf[off + 32 * k : off + 32 * k + 32]) for k in range(m)]
        off += 32 * m
        return {"ciphertext": c, "range_proof": {"partial_ciphertexts": partial, "common_challenge": e0, "ring_responses": resp}}, off

    vote_rings = parse_range(range_decomposition(isqrt(credits) + 1))
    credit_rings = parse_range(range_decomposition(credits + 1))
    off, votes = 0, []
    for _ in range(n_options):
        v, off = block(packed, off, vote_rings)
        votes.append(v)
    credit, off = block(packed, off, credit_rings)
    items = [packed[k : k + 32] for k in range(off, len(packed), 32)]
    if len(items) != 2 + 2 * n_options:
        raise SerdeError("invalid packed length")
    return {"votes": votes, "credit": credit,
            "credit_equivalence_proof": {"challenge": b64url_encode(items[0]),
                                         "ciphertext_responses": [b64url_encode(x) for x in items[1:-1]],
                                         "sum_response": b64url_encode(items[-1])}}


# ------------------------------------------------------------------------------------------------ ballots as JSON text
def verify_choice_json(params, grp, text):
    """`EncryptedChoice::verify` for ballots given as JSON text in the reference's serde layout (what examples/voting.rs:195-198
    prints): (status words, tally of the accepted ballots).  The whole path runs below the C ABI (`eg_verify_choice_json`): the native
    packer, the GPU batch, and for objects whose shape is not the election's the object path of csrc/wire_json.hpp, which applies the
    reference's order of checks exactly like `verify_choice_objects` above (the tests compare the two).  One voter's junk - an object
    that is not JSON, lacks a field, repeats one - is `Malformed` for that voter only."""
    return params.verify_json(text)


def verify_qv_json(params, grp, text):
    """`QuadraticVotingBallot::verify` for ballots given as JSON text (`eg_verify_qv_json`)."""
    return params.verify_json(text)
