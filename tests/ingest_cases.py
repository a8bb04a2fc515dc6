"""Shared scenario builder for the object-ingest tests (CPU: oracle-backed shims; GPU: the real library)."""
import copy

from elastic_elgamal_amd import ingest, serde

L = 2**252 + 27742317777372353535851937790883648493
BAD_POINT = serde.b64url_encode(b"\xff" * 32)
BAD_SCALAR = serde.b64url_encode(L.to_bytes(32, "little"))


def flip(s: str) -> str:
    """another canonical scalar: flip the lowest bit of the encoding"""
    b = bytearray(serde.b64url_decode(s)); b[0] ^= 1
    return serde.b64url_encode(bytes(b))


def choice_cases(valid_objs, single: bool):
    """(name, object, expected status) built from valid EncryptedChoice objects with n options."""
    n = len(valid_objs[0]["choices"])
    S = ingest.status
    cases = [("valid", valid_objs[0], S(ingest.ST_OK))]
    o = copy.deepcopy(valid_objs[1]); o["choices"].append(o["choices"][0]); o["range_proof"]["ring_responses"] += o["range_proof"]["ring_responses"][:2]
    cases.append(("extra choice", o, S(ingest.ST_OPTIONS_LEN)))
    o = copy.deepcopy(valid_objs[1]); o["choices"].pop()
    cases.append(("missing choice", o, S(ingest.ST_OPTIONS_LEN)))
    o = copy.deepcopy(valid_objs[2]); o["range_proof"]["ring_responses"].pop()
    cases.append(("missing response", o, S(ingest.ST_RANGE_LEN)))
    o = copy.deepcopy(valid_objs[2]); o["range_proof"]["ring_responses"].append(o["range_proof"]["ring_responses"][0])
    cases.append(("extra response", o, S(ingest.ST_RANGE_LEN)))
    if single:
        o = copy.deepcopy(valid_objs[3]); o["range_proof"]["ring_responses"].pop(); o["sum_proof"]["response"] = flip(o["sum_proof"]["response"])
        cases.append(("missing response, bad sum proof", o, S(ingest.ST_SUM_CHALLENGE)))
    o = copy.deepcopy(valid_objs[3]); o["choices"].append(o["choices"][0]); o["choices"][1]["blinded_element"] = BAD_POINT
    cases.append(("extra choice, invalid point", o, S(ingest.ST_BAD_POINT, 3)))
    o = copy.deepcopy(valid_objs[4]); o["range_proof"]["ring_responses"].pop(); o["range_proof"]["ring_responses"][2] = BAD_SCALAR
    cases.append(("missing response, non-canonical scalar", o, S(ingest.ST_BAD_SCALAR, 2 * n + 1 + 2)))
    o = copy.deepcopy(valid_objs[4]); o["range_proof"]["ring_responses"][0] = flip(o["range_proof"]["ring_responses"][0])
    cases.append(("well-shaped, tampered ring", o, S(ingest.ST_RANGE_CHALLENGE)))
    o = copy.deepcopy(valid_objs[5]); o["choices"][0]["random_element"] = BAD_POINT
    cases.append(("well-shaped, invalid point", o, S(ingest.ST_BAD_POINT, 0)))
    cases.append(("valid again", valid_objs[6], S(ingest.ST_OK)))
    return cases


def qv_cases(valid_objs):
    n = len(valid_objs[0]["votes"])
    S = ingest.status
    cases = [("valid", valid_objs[0], S(ingest.ST_OK))]
    o = copy.deepcopy(valid_objs[1]); o["votes"].pop(); o["credit_equivalence_proof"]["ciphertext_responses"] = o["credit_equivalence_proof"]["ciphertext_responses"][:-2]
    cases.append(("missing vote", o, S(ingest.ST_OPTIONS_LEN)))
    o = copy.deepcopy(valid_objs[1]); o["votes"][1]["range_proof"]["ring_responses"].pop()
    cases.append(("vote 1 short", o, S(ingest.ST_QV_VARIANT_LEN, 1)))
    o = copy.deepcopy(valid_objs[2]); o["votes"][1]["range_proof"]["ring_responses"].pop()
    o["votes"][0]["range_proof"]["ring_responses"][0] = flip(o["votes"][0]["range_proof"]["ring_responses"][0])
    cases.append(("vote 1 short, vote 0 tampered", o, S(ingest.ST_QV_VARIANT_CHALLENGE, 0)))
    o = copy.deepcopy(valid_objs[2]); o["votes"][0]["range_proof"]["ring_responses"].append(o["votes"][0]["range_proof"]["ring_responses"][0])
    cases.append(("vote 0 long", o, S(ingest.ST_QV_VARIANT_LEN, 0)))
    o = copy.deepcopy(valid_objs[3]); o["credit"]["range_proof"]["partial_ciphertexts"] = o["credit"]["range_proof"]["partial_ciphertexts"][:-1]
    cases.append(("credit range lacks a partial ciphertext", o, S(ingest.ST_QV_CREDIT_RANGE_LEN)))
    o = copy.deepcopy(valid_objs[3]); o["credit_equivalence_proof"]["ciphertext_responses"].pop()
    cases.append(("credit equivalence short", o, S(ingest.ST_QV_CREDIT_EQUIV_LEN)))
    o = copy.deepcopy(valid_objs[4]); o["credit_equivalence_proof"]["ciphertext_responses"].pop()
    o["credit"]["range_proof"]["ring_responses"][1] = flip(o["credit"]["range_proof"]["ring_responses"][1])
    cases.append(("credit equivalence short, credit range tampered", o, S(ingest.ST_QV_CREDIT_RANGE_CHALLENGE)))
    o = copy.deepcopy(valid_objs[4]); o["credit_equivalence_proof"]["ciphertext_responses"].pop()
    o["votes"][n - 1]["ciphertext"]["blinded_element"] = BAD_POINT
    rp = valid_objs[0]["votes"][0]["range_proof"]
    vote_items = 2 + 2 * len(rp["partial_ciphertexts"]) + 1 + len(rp["ring_responses"])     # items of one vote block
    cases.append(("credit equivalence short, invalid point in the last vote", o, S(ingest.ST_BAD_POINT, (n - 1) * vote_items + 1)))
    o = copy.deepcopy(valid_objs[5]); o["credit_equivalence_proof"]["sum_response"] = flip(o["credit_equivalence_proof"]["sum_response"])
    cases.append(("well-shaped, tampered sum of squares", o, S(ingest.ST_QV_CREDIT_EQUIV_CHALLENGE)))
    cases.append(("valid again", valid_objs[6], S(ingest.ST_OK)))
    return cases
