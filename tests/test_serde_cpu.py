"""Host-side wire ingest (elastic_elgamal_amd/serde.py) against the reference's snapshot objects."""
import json
from pathlib import Path

import pytest

from elastic_elgamal_amd import serde

GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def objs():
    return json.loads((GOLD / "snapshots_serde.json").read_text())


def test_pack_matches_golden_packed(objs, golden):
    assert serde.pack_encrypted_choice(objs["encrypted-choice"]).hex() == golden["encrypted-choice"]["packed"]
    assert serde.pack_encrypted_choice(objs["encrypted-multi-choice"]).hex() == golden["encrypted-multi-choice"]["packed"]
    assert serde.pack_qv_ballot(objs["qv-ballot"]).hex() == golden["qv-ballot"]["packed"]
    assert serde.pack_range_encryption(objs["range-encryption"]).hex() == golden["range-encryption"]["packed"]


def test_roundtrip_and_json(objs, golden):
    packed = bytes.fromhex(golden["encrypted-choice"]["packed"])
    obj = serde.unpack_encrypted_choice(packed, 5, True)
    assert obj == objs["encrypted-choice"]
    assert serde.pack_encrypted_choice(json.loads(json.dumps(obj))) == packed
    multi = bytes.fromhex(golden["encrypted-multi-choice"]["packed"])
    assert serde.pack_encrypted_choice(serde.unpack_encrypted_choice(multi, 5, False)) == multi
    assert serde.pack_ballots([obj, obj]) == packed * 2


def test_rejections(objs):
    # serde.rs:402-404,427-429 style failures: bad length, bad alphabet, padding, too few responses
    import copy

    o = copy.deepcopy(objs["encrypted-choice"])
    o["choices"][0]["random_element"] = o["choices"][0]["random_element"][:-2]
    with pytest.raises(serde.SerdeError):
        serde.pack_encrypted_choice(o)
    o = copy.deepcopy(objs["encrypted-choice"])
    o["sum_proof"]["challenge"] += "="
    with pytest.raises(serde.SerdeError):
        serde.pack_encrypted_choice(o)
    o = copy.deepcopy(objs["encrypted-choice"])
    o["range_proof"]["ring_responses"] = o["range_proof"]["ring_responses"][:1]
    with pytest.raises(serde.SerdeError):
        serde.pack_encrypted_choice(o)
    o = copy.deepcopy(objs["encrypted-choice"])
    o["choices"][1]["blinded_element"] = "!!" + o["choices"][1]["blinded_element"][2:]
    with pytest.raises(serde.SerdeError):
        serde.pack_encrypted_choice(o)
    with pytest.raises(serde.SerdeError):
        serde.unpack_encrypted_choice(b"\0" * 100, 5, True)
    with pytest.raises(serde.SerdeError):
        serde.pack_ballots([objs["encrypted-choice"], objs["encrypted-multi-choice"]])
