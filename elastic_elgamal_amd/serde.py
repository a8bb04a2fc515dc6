"""Wire ingest: the reference's human-readable serde layout <-> the packed binary layout of include/eg_hip.h
(SURVEY.md 8f row 2).  Pure host work.

In human-readable formats every scalar / group element is a base64url string WITHOUT padding
(src/serde.rs:19-80, ScalarHelper :179-207, ElementHelper :242-270); structures follow the serde derives of
`EncryptedChoice` (src/app/choice.rs:276-280), `RingProof` (src/proofs/ring.rs:282-287), `LogEqualityProof`
(src/proofs/log_equality.rs:96-101), `QuadraticVotingBallot` (src/app/quadratic_voting.rs:205-217), `RangeProof`
(src/proofs/range.rs:446-450, `inner` flattened) and `SumOfSquaresProof` (src/proofs/mul.rs:86-93).  This is what
`examples/voting.rs:195-198` prints with serde_json.  Canonicity of scalars and validity of elements are NOT checked
here: the GPU verifier does that and reports BadScalar / BadPoint with the item index, like serde would fail.
"""
from __future__ import annotations

import base64
import binascii


class SerdeError(ValueError):
    pass


def b64url_decode(s: str, size: int = 32) -> bytes:
    """base64url without padding (serde.rs:29-47); wrong alphabet, padding or length is an error."""
    if not isinstance(s, str) or "=" in s:
        raise SerdeError("expected an unpadded base64url string")
    try:
        raw = base64.urlsafe_b64decode(s + "=" * (-len(s) % 4))
    except (binascii.Error, ValueError) as e:
        raise SerdeError(f"invalid base64url: {e}") from None
    if base64.urlsafe_b64encode(raw).rstrip(b"=").decode() != s:
        raise SerdeError("non-canonical base64url string")
    if len(raw) != size:
        raise SerdeError(f"invalid byte length {len(raw)}, expected {size}")   # serde.rs:197,260
    return raw


def b64url_encode(b: bytes) -> str:
    return base64.urlsafe_b64encode(b).rstrip(b"=").decode()


def _ct(c) -> bytes:
    return b64url_decode(c["random_element"]) + b64url_decode(c["blinded_element"])


def _ct_json(b: bytes) -> dict:
    return {"random_element": b64url_encode(b[:32]), "blinded_element": b64url_encode(b[32:64])}


def _scalars(xs, minimum=0) -> bytes:
    if len(xs) < minimum:
        raise SerdeError(f"invalid length {len(xs)}, expected at least {minimum}")    # VecHelper<_, MIN>, serde.rs:303-355
    return b"".join(b64url_decode(x) for x in xs)


def _ring_proof(p) -> bytes:
    return b64url_decode(p["common_challenge"]) + _scalars(p["ring_responses"], 2)


def _range_proof(p) -> bytes:
    return b"".join(_ct(c) for c in p["partial_ciphertexts"]) + _ring_proof(p)


def pack_encrypted_choice(obj: dict) -> bytes:
    """EncryptedChoice JSON/YAML object -> choices || RingProof::to_bytes || LogEqualityProof::to_bytes."""
    out = b"".join(_ct(c) for c in obj["choices"]) + _ring_proof(obj["range_proof"])
    sp = obj.get("sum_proof")
    if sp:                                  # SingleChoice; MultiChoice serialises `()` (null)
        out += b64url_decode(sp["challenge"]) + b64url_decode(sp["response"])
    return out


def unpack_encrypted_choice(packed: bytes, n_options: int, single: bool) -> dict:
    want = n_options * 64 + 32 * (1 + 2 * n_options) + (64 if single else 0)
    if len(packed) != want:
        raise SerdeError(f"invalid packed length {len(packed)}, expected {want}")
    it = [packed[i : i + 32] for i in range(0, len(packed), 32)]
    obj = {
        "choices": [_ct_json(packed[64 * k : 64 * k + 64]) for k in range(n_options)],
        "range_proof": {
            "common_challenge": b64url_encode(it[2 * n_options]),
            "ring_responses": [b64url_encode(x) for x in it[2 * n_options + 1 : 4 * n_options + 1]],
        },
        "sum_proof": None,
    }
    if single:
        obj["sum_proof"] = {"challenge": b64url_encode(it[-2]), "response": b64url_encode(it[-1])}
    return obj


def pack_qv_ballot(obj: dict) -> bytes:
    """QuadraticVotingBallot object -> per vote (ct || partials || e0 || responses), credit likewise, sum-of-squares."""
    out = b""
    for v in obj["votes"]:
        out += _ct(v["ciphertext"]) + _range_proof(v["range_proof"])
    out += _ct(obj["credit"]["ciphertext"]) + _range_proof(obj["credit"]["range_proof"])
    p = obj["credit_equivalence_proof"]
    out += b64url_decode(p["challenge"]) + _scalars(p["ciphertext_responses"], 2) + b64url_decode(p["sum_response"])
    return out


def pack_range_encryption(obj: dict) -> bytes:
    """{"ciphertext": .., "proof": RangeProof} as produced by tests/snapshots.rs:93-105."""
    return _ct(obj["ciphertext"]) + _range_proof(obj["proof"])


def pack_ballots(objs, packer=pack_encrypted_choice) -> bytes:
    """Concatenate many ballots; all must have the same packed size (same election parameters)."""
    parts = [packer(o) for o in objs]
    if parts and any(len(p) != len(parts[0]) for p in parts):
        raise SerdeError("ballots of different shapes in one batch")     # OptionsLenMismatch territory (choice.rs:149-158)
    return b"".join(parts)
