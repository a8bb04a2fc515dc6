#!/usr/bin/env python3
"""Convert the reference's Ristretto golden snapshots into packed binary fixtures.

Input : /root/reference/tests/snapshots/snapshots__*-ristretto.snap  (insta YAML / text files produced by
        the reference's tests/snapshots.rs with ChaChaRng::seed_from_u64(12345)) -- test DATA held by the
        reference's own test-suite.
Output: tests/golden/snapshots_ristretto.json  (hex of the packed wire layout used by this repo, i.e. the
        concatenation of the reference's own to_bytes formats in serde field order, plus the generating
        parameters from tests/snapshots.rs).

        /root/reference/src/serde.rs #[cfg(test)] -- the two REJECTING inputs the reference's tests hold for Ristretto (a string that is
        not a group element, a non-canonical scalar) with the error text each test expects -> tests/golden/rejections_ristretto.json.

Run in the build container only (the reference tree does not exist on the GPU box); the JSON is committed.
"""
import base64
import json
import sys
from pathlib import Path

import yaml

SNAP = Path("/root/reference/tests/snapshots")
OUT = Path(__file__).resolve().parent / "snapshots_ristretto.json"


def unb64(s: str) -> bytes:
    return base64.urlsafe_b64decode(s + "=" * (-len(s) % 4))


def load(name: str):
    text = (SNAP / f"snapshots__{name}-ristretto.snap").read_text()
    # insta header: '---\n<meta>\n---\n<body>'
    parts = text.split("---\n")
    body = parts[-1]
    return body


def ct(c) -> bytes:
    return unb64(c["random_element"]) + unb64(c["blinded_element"])


def ring_proof(p) -> bytes:
    return unb64(p["common_challenge"]) + b"".join(unb64(s) for s in p["ring_responses"])


def range_proof(p) -> bytes:
    return b"".join(ct(c) for c in p["partial_ciphertexts"]) + ring_proof(p)


def rejections() -> dict:
    """The reference's own negative vectors: string literals of the unit tests in src/serde.rs (test DATA: an input and the error text
    the test asserts), looked up by test name so that a moved line does not silently change what is pinned."""
    import re

    src = Path("/root/reference/src/serde.rs").read_text()
    lines = src.splitlines()

    def case(test_name: str, which: int = 0):
        start = next(i for i, l in enumerate(lines) if f"fn {test_name}()" in l)
        end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("    }"))
        body = "\n".join(lines[start:end])
        # 43-character unpadded base64url strings = 32 bytes; JSON-quoted ("\"...\"") or bare ("...".into())
        found = [(m.group(1), start + 1 + body[:m.start()].count("\n")) for m in re.finditer(r'"\\?"?([A-Za-z0-9_-]{43})\\?"?"', body)]
        text, line = found[which]
        after = body[body.index(text):]
        err = re.search(r'contains\("([^"]+)"\)', after).group(1)
        raw = unb64(text)
        assert len(raw) == 32
        return {"b64": text, "hex": raw.hex(), "error_contains": err, "source": f"src/serde.rs:{line} ({test_name})"}

    return {
        "_source": "slowli/elastic-elgamal src/serde.rs unit tests: the inputs the reference itself rejects for Ristretto",
        "non_element": case("public_key_deserialization_of_non_element"),
        "non_canonical_scalar": case("secret_key_deserialization_of_invalid_scalar"),
        "scalar_helper_invalid_scalar": case("scalar_helper_invalid_scalar"),
        "element_helper_invalid_element": case("element_helper_invalid_element"),
    }


def main() -> None:
    (OUT.parent / "rejections_ristretto.json").write_text(json.dumps(rejections(), indent=1) + "\n")
    out = {
        "_source": "slowli/elastic-elgamal tests/snapshots/*-ristretto.snap (tests/snapshots.rs, seed 12345)",
        "seed": 12345,
        "public_key_b64": "pq226cCujVTCbm5WtczXoWuw4ZUavk1-5wKOPU7KhTE",
    }

    y = yaml.safe_load(load("encrypted-choice"))
    out["encrypted-choice"] = {
        "params": {"options": 5, "single": True, "choice": 3},
        "packed": (b"".join(ct(c) for c in y["choices"]) + ring_proof(y["range_proof"])
                   + unb64(y["sum_proof"]["challenge"]) + unb64(y["sum_proof"]["response"])).hex(),
    }
    y = yaml.safe_load(load("encrypted-multi-choice"))
    out["encrypted-multi-choice"] = {
        "params": {"options": 5, "single": False, "choices": [0, 1, 1, 0, 1]},
        "packed": (b"".join(ct(c) for c in y["choices"]) + ring_proof(y["range_proof"])).hex(),
    }
    y = yaml.safe_load(load("qv-ballot"))
    packed = b""
    for v in y["votes"]:
        packed += ct(v["ciphertext"]) + range_proof(v["range_proof"])
    packed += ct(y["credit"]["ciphertext"]) + range_proof(y["credit"]["range_proof"])
    p = y["credit_equivalence_proof"]
    packed += unb64(p["challenge"]) + b"".join(unb64(s) for s in p["ciphertext_responses"]) + unb64(p["sum_response"])
    out["qv-ballot"] = {"params": {"options": 5, "credits": 15, "votes": [3, 0, 1, 0, 2]}, "packed": packed.hex()}

    y = yaml.safe_load(load("range-encryption"))
    out["range-encryption"] = {
        "params": {"upper_bound": 100, "value": 42, "label": "ciphertext_range"},
        "packed": (ct(y["ciphertext"]) + range_proof(y["proof"])).hex(),
    }
    y = yaml.safe_load(load("zero-encryption"))
    out["zero-encryption"] = {
        "packed": (ct(y["ciphertext"]) + unb64(y["proof"]["challenge"]) + unb64(y["proof"]["response"])).hex()
    }
    out["zero-encryption-bin"] = {"packed": unb64(load("zero-encryption-bin").strip()).hex()}
    y = yaml.safe_load(load("bool-encryption"))
    out["bool-encryption"] = {"params": {"value": True}, "packed": (ct(y["ciphertext"]) + ring_proof(y["proof"])).hex()}
    out["bool-encryption-bin"] = {"packed": unb64(load("bool-encryption-bin").strip()).hex()}
    y = yaml.safe_load(load("ciphertext"))
    out["ciphertext"] = {"params": {"value": 42}, "packed": ct(y).hex()}
    out["ciphertext-bin"] = {"packed": unb64(load("ciphertext-bin").strip()).hex()}
    y = yaml.safe_load(load("sum-sq-proof"))
    out["sum-sq-proof"] = {
        "params": {"values": [1, 3, 3, 7, 5], "label": "test"},
        "packed": (unb64(y["challenge"]) + b"".join(unb64(s) for s in y["ciphertext_responses"])
                   + unb64(y["sum_response"])).hex(),
    }
    serde = {name: yaml.safe_load(load(name)) for name in
             ("encrypted-choice", "encrypted-multi-choice", "qv-ballot", "range-encryption")}
    (OUT.parent / "snapshots_serde.json").write_text(json.dumps(serde, indent=1) + "\n")
    OUT.write_text(json.dumps(out, indent=1) + "\n")
    print(f"wrote {OUT}", file=sys.stderr)


if __name__ == "__main__":
    main()
