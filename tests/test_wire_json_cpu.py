"""Native wire ingest (csrc/wire_json.hpp behind eg_choice_pack_json / eg_qv_pack_json; host-only C++): JSON in the reference's
serde layout -> packed ballots.  Checked against the reference's snapshot objects, against the Python mirror (serde.py) and on
the rejections of src/serde.rs:29-47,191-206,254-269,303-355.  No GPU needed."""
import copy
import json
import time
from pathlib import Path

import pytest

import elastic_elgamal_amd as eg
from elastic_elgamal_amd import ingest, serde

GOLD = Path(__file__).resolve().parent / "golden"
MAL, RESHAPE = 13, eg.PACK_RESHAPE


@pytest.fixture(scope="module")
def objs():
    return json.loads((GOLD / "snapshots_serde.json").read_text())


def test_snapshot_objects_pack_to_the_golden_bytes(objs, golden):
    c, m, q = objs["encrypted-choice"], objs["encrypted-multi-choice"], objs["qv-ballot"]
    for text in (json.dumps([c]), json.dumps(c, indent=2), json.dumps(c) + "\n" + json.dumps(c)):
        packed, st = eg.pack_json(text, 5, single=True)
        assert set(st) == {0} and packed[:736].hex() == golden["encrypted-choice"]["packed"] and len(packed) == 736 * len(st)
    packed, st = eg.pack_json(json.dumps([m, m, m]), 5, single=False, threads=2)
    assert st == [0, 0, 0] and packed == bytes.fromhex(golden["encrypted-multi-choice"]["packed"]) * 3
    packed, st = eg.pack_json(json.dumps(q, indent=1), 5, credits=15)
    assert st == [0] and packed.hex() == golden["qv-ballot"]["packed"]
    assert eg.pack_json("", 5, single=True) == (b"", []) and eg.pack_json("[]", 5, single=True) == (b"", [])
    # field order does not matter, unknown fields are skipped (serde's default)
    shuffled = {"sum_proof": c["sum_proof"], "extra": {"a": [1, "x", None]}, "range_proof": dict(reversed(list(c["range_proof"].items()))),
                "choices": [dict(reversed(list(x.items()))) for x in c["choices"]]}
    packed, st = eg.pack_json(json.dumps(shuffled), 5, single=True)
    assert st == [0] and packed.hex() == golden["encrypted-choice"]["packed"]


def _choice_rejections(c):
    cases = []

    def mut(fn, want):
        o = copy.deepcopy(c)
        fn(o)
        cases.append((o, want))

    mut(lambda o: o["choices"][0].__setitem__("random_element", o["choices"][0]["random_element"][:-2]), MAL)      # 31 bytes
    mut(lambda o: o["choices"][0].__setitem__("random_element", o["choices"][0]["random_element"] + "AA"), MAL)    # 33+ bytes
    mut(lambda o: o["sum_proof"].__setitem__("challenge", o["sum_proof"]["challenge"] + "="), MAL)                 # padding
    mut(lambda o: o["choices"][1].__setitem__("blinded_element", "!!" + o["choices"][1]["blinded_element"][2:]), MAL)   # alphabet
    mut(lambda o: o["choices"][1].__setitem__("blinded_element", o["choices"][1]["blinded_element"].replace("-", "+").replace("_", "/")
                                              if ("-" in o["choices"][1]["blinded_element"] or "_" in o["choices"][1]["blinded_element"])
                                              else "+" + o["choices"][1]["blinded_element"][1:]), MAL)           # standard alphabet
    mut(lambda o: o["range_proof"].__setitem__("common_challenge", o["range_proof"]["common_challenge"][:-1] + "B"), MAL)   # trailing bits
    mut(lambda o: o["range_proof"].__setitem__("ring_responses", o["range_proof"]["ring_responses"][:1]), MAL)     # VecHelper<_, 2>
    mut(lambda o: o.__setitem__("sum_proof", None), MAL)                                                           # wrong proof kind
    mut(lambda o: o.pop("range_proof"), MAL)                                                                       # missing field
    mut(lambda o: o["choices"][2].pop("blinded_element"), MAL)
    mut(lambda o: o["choices"].__setitem__(0, "AAAA"), MAL)                                                        # wrong type
    mut(lambda o: o["range_proof"].__setitem__("common_challenge", 17), MAL)
    mut(lambda o: o["choices"].pop(), RESHAPE)                                                                     # OptionsLenMismatch
    mut(lambda o: o["choices"].append(o["choices"][0]), RESHAPE)
    mut(lambda o: o["range_proof"]["ring_responses"].pop(), RESHAPE)                                               # LenMismatch
    mut(lambda o: o["range_proof"]["ring_responses"].extend(o["range_proof"]["ring_responses"][:2]), RESHAPE)
    return cases


def test_rejections_follow_serde_and_python_mirror(objs):
    c = objs["encrypted-choice"]
    cases = _choice_rejections(c)
    batch = [c] + [o for o, _ in cases] + [c]
    packed, st = eg.pack_json(json.dumps(batch), 5, single=True, threads=3)
    assert st == [0] + [w for _, w in cases] + [0]
    good = serde.pack_encrypted_choice(c)
    assert packed[:736] == good == packed[-736:]
    assert all(packed[736 * k : 736 * (k + 1)] == bytes(736) for k in range(1, len(batch) - 1))      # rejected slots are zeroed
    # the Python mirror rejects exactly the MALFORMED ones
    for o, want in cases:
        if want == MAL:
            with pytest.raises((serde.SerdeError, KeyError, TypeError, AttributeError)):
                serde.pack_encrypted_choice(o)
                if o.get("sum_proof") is None:
                    raise serde.SerdeError("sum_proof does not match the kind of election")
        else:
            assert len(serde.pack_encrypted_choice(o)) != 736
    # duplicate fields cannot be built from a dict: write the text
    text = json.dumps(c)
    dup = text[:-1] + ', "sum_proof": ' + json.dumps(c["sum_proof"]) + "}"
    assert eg.pack_json(dup, 5, single=True)[1] == [MAL]
    for junk in (text + " trailing", "not json", "[" + text + ",]x"):
        with pytest.raises(eg.EgError):
            eg.pack_json(junk, 5, single=True)
    with pytest.raises(eg.EgError):
        eg.pack_json("[" + text, 5, single=True)


def test_qv_shapes_and_rejections(objs, oracle, golden):
    q = objs["qv-ballot"]
    o1 = copy.deepcopy(q); o1["votes"].pop()                                             # OptionsLenMismatch
    o2 = copy.deepcopy(q); o2["credit"]["range_proof"]["partial_ciphertexts"] = []       # LenMismatch("admissible values")
    o3 = copy.deepcopy(q); o3["votes"][1]["range_proof"]["ring_responses"].pop()
    o4 = copy.deepcopy(q); o4["credit_equivalence_proof"]["ciphertext_responses"] = o4["credit_equivalence_proof"]["ciphertext_responses"][:4]
    o5 = copy.deepcopy(q); o5["credit_equivalence_proof"]["ciphertext_responses"] = []   # VecHelper<_, 2>
    o6 = copy.deepcopy(q); o6["votes"][0]["ciphertext"]["random_element"] = "AAAA"
    o7 = copy.deepcopy(q); del o7["credit"]["range_proof"]["common_challenge"]
    packed, st = eg.pack_json(json.dumps([q, o1, o2, o3, o4, o5, o6, o7, q]), 5, credits=15, threads=4)
    assert st == [0, RESHAPE, RESHAPE, RESHAPE, RESHAPE, MAL, MAL, MAL, 0]
    size = len(packed) // 9
    assert packed[:size].hex() == golden["qv-ballot"]["packed"] == packed[-size:].hex()
    # other parameters: ballots from the oracle prover through the Python mirror's unpack
    import base64
    pk = base64.urlsafe_b64decode(golden["public_key_b64"] + "=")
    for n, credits in ((3, 9), (2, 100), (4, 30)):
        oq = oracle.QvParams(pk, n, credits)
        raw = oq.generate_batch(5, 0, 6)
        sz = len(raw) // 6
        dicts = [ingest.unpack_qv_ballot(raw[i * sz : (i + 1) * sz], n, credits) for i in range(6)]
        packed, st = eg.pack_json("\n".join(json.dumps(d) for d in dicts), n, credits=credits)
        assert st == [0] * 6 and packed == raw


def test_native_ingest_throughput(objs):
    """Objects per second of the native packer against the Python mirror on the same text (the GPU consumes ~5 M ballots/s)."""
    c = objs["encrypted-choice"]
    n = 20000
    text = "[" + ",".join([json.dumps(c)] * n) + "]"
    t0 = time.perf_counter()
    packed, st = eg.pack_json(text, 5, single=True, max_objects=n)
    native = time.perf_counter() - t0
    assert st == [0] * n and packed[-736:] == serde.pack_encrypted_choice(c)
    t0 = time.perf_counter()
    ref = serde.pack_ballots(json.loads(text)[:2000])
    python = (time.perf_counter() - t0) * (n / 2000)
    assert ref == packed[: 2000 * 736]
    print(f"native {n / native:,.0f} objects/s ({len(text) / native / 1e6:.0f} MB/s of JSON), python mirror {n / python:,.0f} objects/s")
    assert native < python / 5


def test_cpp_mirror_packs_json(tmp_path, objs):
    """include/elastic_elgamal_hip.hpp: pack_choice_json / pack_qv_json on the host (no GPU), compiled with g++."""
    import subprocess

    root = Path(__file__).resolve().parent.parent
    one = json.dumps(objs["encrypted-choice"])
    q = json.dumps(objs["qv-ballot"])
    (tmp_path / "c.json").write_text("[" + one + "," + one + ',{"choices":1}]')
    (tmp_path / "q.json").write_text(q + "\n" + q)
    (tmp_path / "t.cpp").write_text(r"""
        #include "elastic_elgamal_hip.hpp"
        #include <fstream>
        #include <sstream>
        #include <cstdio>
        using namespace elastic_elgamal_hip;
        static std::string slurp(const char* p) { std::ifstream f(p); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
        int main(int argc, char** argv) {
          PackedJson c = pack_choice_json(5, true, slurp(argv[1]), 2);
          PackedJson q = pack_qv_json(5, 15, slurp(argv[2]), 2);
          bool threw = false;
          try { pack_choice_json(5, true, "garbage"); } catch (const Error&) { threw = true; }
          printf("%zu %u %u %u %zu | %zu %u %u %zu | %d\n", c.status.size(), c.status[0], c.status[1], c.status[2], c.accepted().size(),
                 q.status.size(), q.status[0], q.status[1], q.accepted().size(), (int)threw);
          return 0;
        }""")
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", f"-I{root / 'include'}", str(tmp_path / "t.cpp"), f"-L{root / 'elastic_elgamal_amd'}", "-leg_hip",
                           f"-Wl,-rpath,{root / 'elastic_elgamal_amd'}", "-o", str(exe)])
    out = subprocess.run([str(exe), str(tmp_path / "c.json"), str(tmp_path / "q.json")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    qsize = len(serde.pack_qv_ballot(objs["qv-ballot"]))
    assert out.stdout.strip() == f"3 0 0 13 {2 * 736} | 2 0 0 {2 * qsize} | 1"


def test_escaped_strings_and_keys_are_unescaped_like_serde_json(objs, golden):
    """serde_json unescapes strings and keys before base64 decoding / field matching: a ballot whose emitter wrote a
    character as \\uXXXX or '/' as '\\/' is the same ballot; a bad escape, an escape to another character or to a non-ASCII code
    point is Malformed for that object only."""
    c = objs["encrypted-choice"]
    plain = json.dumps(c)
    want = golden["encrypted-choice"]["packed"]
    e0 = c["range_proof"]["common_challenge"]
    esc_first = "\\u%04x" % ord(e0[0]) + e0[1:]
    esc_mid = e0[:20] + "\\u%04X" % ord(e0[20]) + e0[21:]
    for variant in (plain.replace('"' + e0 + '"', '"' + esc_first + '"'),
                    plain.replace('"' + e0 + '"', '"' + esc_mid + '"'),
                    plain.replace('"choices"', '"ch\\u006fices"'),
                    plain.replace('"random_element"', '"random\\u005felement"', 1),
                    plain.replace('"sum_proof"', '"sum\\u005Fproof"')):
        assert variant != plain
        packed, st = eg.pack_json("[" + variant + "," + plain + "]", 5, single=True)
        assert st == [0, 0] and packed.hex() == want * 2, variant[:80]
    for variant in (plain.replace('"' + e0 + '"', '"' + "\\u0021" + e0[1:] + '"'),            # '!' is not in the alphabet
                    plain.replace('"' + e0 + '"', '"' + "\\u00e9" + e0[1:] + '"'),            # non-ASCII code point
                    plain.replace('"' + e0 + '"', '"' + "\\x41" + e0[1:] + '"'),              # not a JSON escape
                    plain.replace('"' + e0 + '"', '"' + "\\u00" + e0[1:] + '"'),              # truncated \u
                    plain.replace('"' + e0 + '"', '"' + e0 + "\\n" + '"')):                   # 44 characters after unescaping
        _, st = eg.pack_json("[" + variant + "," + plain + "]", 5, single=True)
        assert st == [MAL, 0], variant[:80]
    # an escaped spelling of an unknown field is still an unknown field; of a duplicate field, a duplicate
    assert eg.pack_json(plain[:-1] + ', "\\u0065xtra": 1}', 5, single=True)[1] == [0]
    assert eg.pack_json(plain[:-1] + ', "ch\\u006fices": []}', 5, single=True)[1] == [MAL]
