"""ctypes wrapper around oracle/liboracle.so -- CPU ORACLE (test infrastructure, NOT the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (elastic_elgamal_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_DIR = Path(__file__).resolve().parent
_LIB_PATH = _DIR / "liboracle.so"


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (building the checker is not using it)."""
    srcs = [p for p in _DIR.glob("*.c")] + [_DIR / "eg_oracle.h"]
    if force or not _LIB_PATH.exists() or any(p.stat().st_mtime > _LIB_PATH.stat().st_mtime for p in srcs):
        subprocess.check_call(["make", "-C", str(_DIR), "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _declare(_lib)
        _lib.or_init()
    return _lib


class ChaChaRng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("counter", C.c_uint64)]


def _declare(l: C.CDLL) -> None:
    vp, u8p, u32p, u64p = C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    sig = {
        "or_init": (None, []),
        "or_const_bytes": (None, [C.c_int, u8p]),
        "or_sc_from_wide": (None, [u8p, u8p]),
        "or_sc_is_canonical": (C.c_int, [u8p]),
        "or_sc_add": (None, [u8p, u8p, u8p]),
        "or_sc_sub": (None, [u8p, u8p, u8p]),
        "or_sc_neg": (None, [u8p, u8p]),
        "or_sc_mul": (None, [u8p, u8p, u8p]),
        "or_sc_muladd": (None, [u8p, u8p, u8p, u8p]),
        "or_sc_invert": (None, [u8p, u8p]),
        "or_merlin_init": (None, [vp, u8p]),
        "or_merlin_append": (None, [vp, u8p, u8p, C.c_size_t]),
        "or_merlin_append_u64": (None, [vp, u8p, C.c_uint64]),
        "or_merlin_challenge": (None, [vp, u8p, u8p, C.c_size_t]),
        "or_keccak_calls": (C.c_uint64, []),
        "or_rng_seed_from_u64": (None, [C.POINTER(ChaChaRng), C.c_uint64]),
        "or_rng_fill64": (None, [C.POINTER(ChaChaRng), u8p]),
        "or_keypair_from_seed": (None, [C.c_uint64, u8p, u8p, C.POINTER(ChaChaRng)]),
        "or_isqrt": (C.c_uint64, [C.c_uint64]),
        "or_choice_params_new": (vp, [u8p, C.c_int, C.c_int]),
        "or_qv_params_new": (vp, [u8p, C.c_int, C.c_uint64]),
        "or_prepared_range_new": (vp, [C.c_uint64]),
        "or_pubkey_new": (vp, [u8p]),
        "or_choice_params_pk": (vp, [vp]),
        "or_qv_params_pk": (vp, [vp]),
        "or_qv_vote_range": (vp, [vp]),
        "or_qv_credit_range": (vp, [vp]),
        "or_prepared_range_name": (C.c_int, [vp, u8p, C.c_size_t]),
        "or_prepared_range_rings": (C.c_int, [vp, u64p, u64p]),
        "or_prepared_range_table": (None, [vp, C.c_int, C.c_int, u8p]),
        "or_free": (None, [vp]),
        "or_choice_ballot_size": (C.c_size_t, [C.c_int, C.c_int]),
        "or_range_proof_size": (C.c_size_t, [vp]),
        "or_qv_ballot_size": (C.c_size_t, [vp]),
        "or_choice_verify": (C.c_uint32, [vp, u8p]),
        "or_range_verify": (C.c_uint32, [vp, vp, u8p, u8p]),
        "or_sumsq_verify": (C.c_uint32, [vp, C.c_int, u8p, u8p, u8p, u8p]),
        "or_qv_verify": (C.c_uint32, [vp, u8p]),
        "or_choice_verify_object": (C.c_uint32, [vp, C.c_int, C.c_int, u8p]),
        "or_qv_verify_object": (C.c_uint32, [vp, C.POINTER(C.c_int), u8p]),
        "or_verify_zero": (C.c_uint32, [vp, u8p]),
        "or_verify_bool": (C.c_uint32, [vp, u8p]),
        "or_encrypt_u64": (None, [vp, C.c_uint64, C.POINTER(ChaChaRng), u8p]),
        "or_encrypt_zero": (None, [vp, C.POINTER(ChaChaRng), u8p]),
        "or_encrypt_bool": (None, [vp, C.c_int, C.POINTER(ChaChaRng), u8p]),
        "or_encrypt_range": (None, [vp, vp, C.c_uint64, C.POINTER(ChaChaRng), u8p]),
        "or_choice_new": (None, [vp, u8p, C.POINTER(ChaChaRng), u8p]),
        "or_qv_new": (None, [vp, u64p, C.POINTER(ChaChaRng), u8p]),
        "or_sumsq_snapshot": (None, [vp, C.c_int, u64p, C.POINTER(ChaChaRng), u8p, u8p]),
        "or_select_single": (None, [C.c_uint64, C.c_int, u8p]),
        "or_select_multi": (None, [C.c_uint64, C.c_int, C.c_int, u8p]),
        "or_select_qv": (None, [C.c_uint64, C.c_int, C.c_uint64, u64p]),
        "or_choice_generate_batch": (None, [vp, C.c_uint64, C.c_size_t, C.c_size_t, C.c_int, vp, C.c_int]),
        "or_qv_generate_batch": (None, [vp, C.c_uint64, C.c_size_t, C.c_size_t, vp, C.c_int]),
        "or_choice_verify_batch": (None, [vp, C.c_size_t, vp, vp, C.c_int]),
        "or_qv_verify_batch": (None, [vp, C.c_size_t, vp, vp, C.c_int]),
        "or_tally": (None, [C.c_int, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, C.c_size_t, vp]),
        "or_decryption_share_new": (C.c_int, [u8p, u8p, C.c_uint64, C.c_uint64, u8p, C.c_uint64, C.POINTER(ChaChaRng), u8p]),
        "or_decryption_share_verify": (C.c_uint32, [u8p, C.c_uint64, C.c_uint64, u8p, C.c_uint64, u8p]),
        "or_point_double_mul_generator": (C.c_int, [u8p, u8p, u8p, u8p]),
        "or_point_multi_mul": (C.c_int, [C.c_size_t, u8p, u8p, u8p]),
        "or_point_mul_generator": (None, [u8p, u8p]),
        "or_point_add": (C.c_int, [u8p, u8p, C.c_int, u8p]),
        "or_point_roundtrip": (C.c_int, [u8p, u8p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args


# --------------------------------------------------------------------------- status words
OK, BAD_SCALAR, BAD_POINT, OPTIONS_LEN, SUM_CHALLENGE, RANGE_LEN, RANGE_CHALLENGE = range(7)
QV_VARIANT_LEN, QV_VARIANT_CHALLENGE, QV_CREDIT_RANGE_LEN, QV_CREDIT_RANGE_CHALLENGE = 7, 8, 9, 10
QV_CREDIT_EQUIV_LEN, QV_CREDIT_EQUIV_CHALLENGE = 11, 12


def status(kind: int, detail: int = 0) -> int:
    return kind | (detail << 8)


def _buf(n: int) -> C.Array:
    return C.create_string_buffer(n)


# --------------------------------------------------------------------------- scalars / constants
def const_bytes(which: int) -> bytes:
    b = _buf(32)
    lib().or_const_bytes(which, b)
    return b.raw


def sc_from_wide(wide: bytes) -> bytes:
    b = _buf(32)
    lib().or_sc_from_wide(b, wide)
    return b.raw


def sc_is_canonical(s: bytes) -> bool:
    return bool(lib().or_sc_is_canonical(s))


def _sc2(fn, a: bytes, b: bytes) -> bytes:
    r = _buf(32)
    fn(r, a, b)
    return r.raw


def sc_add(a, b): return _sc2(lib().or_sc_add, a, b)
def sc_sub(a, b): return _sc2(lib().or_sc_sub, a, b)
def sc_mul(a, b): return _sc2(lib().or_sc_mul, a, b)


def sc_neg(a: bytes) -> bytes:
    r = _buf(32)
    lib().or_sc_neg(r, a)
    return r.raw


def sc_invert(a: bytes) -> bytes:
    r = _buf(32)
    lib().or_sc_invert(r, a)
    return r.raw


# --------------------------------------------------------------------------- points
def point_mul_generator(k: bytes) -> bytes:
    r = _buf(32)
    lib().or_point_mul_generator(k, r)
    return r.raw


def point_double_mul_generator(k: bytes, p: bytes, r_: bytes):
    r = _buf(32)
    rc = lib().or_point_double_mul_generator(k, p, r_, r)
    return None if rc else r.raw


def point_multi_mul(ks: bytes, ps: bytes):
    n = len(ks) // 32
    r = _buf(32)
    rc = lib().or_point_multi_mul(n, ks, ps, r)
    return None if rc else r.raw


def point_add(a: bytes, b: bytes, sub: bool = False):
    r = _buf(32)
    rc = lib().or_point_add(a, b, int(sub), r)
    return None if rc else r.raw


def point_roundtrip(a: bytes):
    """decode then encode; None if `a` is not a valid ristretto255 encoding."""
    r = _buf(32)
    rc = lib().or_point_roundtrip(a, r)
    return None if rc else r.raw


# --------------------------------------------------------------------------- transcript
class Merlin:
    def __init__(self, label: bytes):
        self._st = _buf(256)
        lib().or_merlin_init(self._st, label)

    def append(self, label: bytes, msg: bytes) -> None:
        lib().or_merlin_append(self._st, label, msg, len(msg))

    def append_u64(self, label: bytes, x: int) -> None:
        lib().or_merlin_append_u64(self._st, label, x)

    def challenge(self, label: bytes, n: int) -> bytes:
        out = _buf(n)
        lib().or_merlin_challenge(self._st, label, out, n)
        return out.raw

    @property
    def pos(self) -> int:
        return self._st.raw[200]

    @property
    def state(self) -> bytes:
        return self._st.raw[:200]


# --------------------------------------------------------------------------- rng / keys
def rng_from_u64(seed: int) -> ChaChaRng:
    r = ChaChaRng()
    lib().or_rng_seed_from_u64(C.byref(r), seed)
    return r


def rng_fill64(r: ChaChaRng) -> bytes:
    b = _buf(64)
    lib().or_rng_fill64(C.byref(r), b)
    return b.raw


def keypair_from_seed(seed: int):
    """(sk, pk, rng positioned after the keypair draw) as in tests/snapshots.rs:32-33."""
    sk, pk, rng = _buf(32), _buf(32), ChaChaRng()
    lib().or_keypair_from_seed(seed, sk, pk, C.byref(rng))
    return sk.raw, pk.raw, rng


# --------------------------------------------------------------------------- params objects
class ChoiceParams:
    """ChoiceParams::single / ::multi (choice.rs:132-196)."""

    def __init__(self, pk: bytes, n_options: int, single: bool = True):
        self.ptr = lib().or_choice_params_new(pk, n_options, int(single))
        if not self.ptr:
            raise ValueError("invalid public key or parameters beyond the oracle's capacity")
        self.pk, self.n_options, self.single = pk, n_options, single
        self.ballot_size = lib().or_choice_ballot_size(n_options, int(single))
        self.pk_ptr = lib().or_choice_params_pk(self.ptr)

    def new_ballot(self, flags, rng: ChaChaRng) -> bytes:
        out = _buf(self.ballot_size)
        lib().or_choice_new(self.ptr, bytes(int(bool(f)) for f in flags), C.byref(rng), out)
        return out.raw

    def verify(self, ballot: bytes) -> int:
        assert len(ballot) == self.ballot_size
        return lib().or_choice_verify(self.ptr, ballot)

    def verify_object(self, choices, common_challenge: bytes, responses, sum_proof=None) -> int:
        """EncryptedChoice::verify on an object of ANY shape (oracle/objects.c): choices = [(R, B) bytes pairs],
        responses = list of 32-byte scalars, sum_proof = (challenge, response) for single-choice elections."""
        items = b"".join(r + b for r, b in choices) + common_challenge + b"".join(responses)
        if self.single:
            items += sum_proof[0] + sum_proof[1]
        return lib().or_choice_verify_object(self.ptr, len(choices), len(responses), items)

    def generate_batch(self, base_seed: int, first: int, n: int, n_selected: int = 0, threads: int = 0) -> bytes:
        out = _buf(n * self.ballot_size)
        lib().or_choice_generate_batch(self.ptr, base_seed, first, n, n_selected, out, threads or os.cpu_count())
        return out.raw

    def verify_batch(self, ballots: bytes, threads: int = 0):
        n = len(ballots) // self.ballot_size
        st = (C.c_uint32 * n)()
        lib().or_choice_verify_batch(self.ptr, n, ballots, st, threads or os.cpu_count())
        return list(st)

    def tally(self, ballots: bytes, statuses) -> bytes:
        n = len(ballots) // self.ballot_size
        st = (C.c_uint32 * n)(*statuses)
        out = _buf(64 * self.n_options)
        lib().or_tally(self.n_options, self.ballot_size, n, ballots, st, 0, 64, out)
        return out.raw


class PreparedRange:
    def __init__(self, upper_bound: int = 0, ptr=None):
        self.ptr = ptr if ptr is not None else lib().or_prepared_range_new(upper_bound)
        nm = _buf(256)
        lib().or_prepared_range_name(self.ptr, nm, 256)
        self.name = nm.value.decode()
        sizes, steps = (C.c_uint64 * 16)(), (C.c_uint64 * 16)()
        n = lib().or_prepared_range_rings(self.ptr, sizes, steps)
        self.rings = [(int(sizes[i]), int(steps[i])) for i in range(n)]
        self.proof_size = lib().or_range_proof_size(self.ptr)

    def table(self, ring: int, j: int) -> bytes:
        b = _buf(32)
        lib().or_prepared_range_table(self.ptr, ring, j, b)
        return b.raw


class PublicKey:
    def __init__(self, pk: bytes):
        self.ptr = lib().or_pubkey_new(pk)
        if not self.ptr:
            raise ValueError("invalid public key or parameters beyond the oracle's capacity")
        self.bytes = pk

    def encrypt_u64(self, value: int, rng) -> bytes:
        out = _buf(64)
        lib().or_encrypt_u64(self.ptr, value, C.byref(rng), out)
        return out.raw

    def encrypt_zero(self, rng) -> bytes:
        out = _buf(128)
        lib().or_encrypt_zero(self.ptr, C.byref(rng), out)
        return out.raw

    def encrypt_bool(self, value: bool, rng) -> bytes:
        out = _buf(160)
        lib().or_encrypt_bool(self.ptr, int(value), C.byref(rng), out)
        return out.raw

    def encrypt_range(self, rng_obj: PreparedRange, value: int, rng) -> bytes:
        out = _buf(64 + rng_obj.proof_size)
        lib().or_encrypt_range(self.ptr, rng_obj.ptr, value, C.byref(rng), out)
        return out.raw

    def verify_zero(self, blob: bytes) -> int:
        return lib().or_verify_zero(self.ptr, blob)

    def verify_bool(self, blob: bytes) -> int:
        return lib().or_verify_bool(self.ptr, blob)

    def verify_range(self, rng_obj: PreparedRange, blob: bytes, label: bytes = b"ciphertext_range") -> int:
        return lib().or_range_verify(self.ptr, rng_obj.ptr, blob, label)

    def sumsq_snapshot(self, values, rng):
        n = len(values)
        vals = (C.c_uint64 * n)(*values)
        cts, proof = _buf(64 * (n + 1)), _buf(32 * (2 * n + 2))
        lib().or_sumsq_snapshot(self.ptr, n, vals, C.byref(rng), cts, proof)
        return cts.raw, proof.raw

    def verify_sumsq(self, cts: bytes, sum_ct: bytes, proof: bytes, label: bytes) -> int:
        return lib().or_sumsq_verify(self.ptr, len(cts) // 64, cts, sum_ct, proof, label)


class QvParams:
    """QuadraticVotingParams::new (quadratic_voting.rs:63-76)."""

    def __init__(self, pk: bytes, n_options: int, credits: int):
        self.ptr = lib().or_qv_params_new(pk, n_options, credits)
        if not self.ptr:
            raise ValueError("invalid public key or parameters beyond the oracle's capacity")
        self.pk, self.n_options, self.credits = pk, n_options, credits
        self.ballot_size = lib().or_qv_ballot_size(self.ptr)
        self.vote_range = PreparedRange(ptr=lib().or_qv_vote_range(self.ptr))
        self.credit_range = PreparedRange(ptr=lib().or_qv_credit_range(self.ptr))
        self.vote_size = 64 + self.vote_range.proof_size
        self.credit_size = 64 + self.credit_range.proof_size

    def new_ballot(self, votes, rng: ChaChaRng) -> bytes:
        out = _buf(self.ballot_size)
        v = (C.c_uint64 * len(votes))(*votes)
        lib().or_qv_new(self.ptr, v, C.byref(rng), out)
        return out.raw

    def verify(self, ballot: bytes) -> int:
        assert len(ballot) == self.ballot_size
        return lib().or_qv_verify(self.ptr, ballot)

    def verify_object(self, blocks, sumsq) -> int:
        """QuadraticVotingBallot::verify on an object of ANY shape (oracle/objects.c).  blocks = votes then credit, each
        (ciphertext (R, B), [partial (R, B), ..], common_challenge, [responses]); sumsq = (challenge, [responses], sum_response)."""
        shape, items = [len(blocks) - 1], b""
        for ct, partials, e0, resp in blocks:
            shape += [len(partials), len(resp)]
            items += ct[0] + ct[1] + b"".join(r + b for r, b in partials) + e0 + b"".join(resp)
        shape.append(len(sumsq[1]))
        items += sumsq[0] + b"".join(sumsq[1]) + sumsq[2]
        arr = (C.c_int * len(shape))(*shape)
        return lib().or_qv_verify_object(self.ptr, arr, items)

    def generate_batch(self, base_seed: int, first: int, n: int, threads: int = 0) -> bytes:
        out = _buf(n * self.ballot_size)
        lib().or_qv_generate_batch(self.ptr, base_seed, first, n, out, threads or os.cpu_count())
        return out.raw

    def verify_batch(self, ballots: bytes, threads: int = 0):
        n = len(ballots) // self.ballot_size
        st = (C.c_uint32 * n)()
        lib().or_qv_verify_batch(self.ptr, n, ballots, st, threads or os.cpu_count())
        return list(st)

    def tally(self, ballots: bytes, statuses) -> bytes:
        n = len(ballots) // self.ballot_size
        st = (C.c_uint32 * n)(*statuses)
        out = _buf(64 * self.n_options)
        lib().or_tally(self.n_options, self.ballot_size, n, ballots, st, 0, self.vote_size, out)
        return out.raw


def select_single(seed: int, n: int):
    b = _buf(n)
    lib().or_select_single(seed, n, b)
    return list(b.raw)


def select_multi(seed: int, n: int, k: int):
    b = _buf(n)
    lib().or_select_multi(seed, n, k, b)
    return list(b.raw)


def select_qv(seed: int, n: int, credits: int):
    v = (C.c_uint64 * n)()
    lib().or_select_qv(seed, n, credits, v)
    return list(v)


def decryption_share_new(sk_share: bytes, ct_random: bytes, shares: int, threshold: int, shared_key: bytes, index: int, rng) -> bytes:
    """dh || challenge || response for one participant and one ciphertext (participant.rs:163-186)."""
    out = _buf(96)
    rc = lib().or_decryption_share_new(sk_share, ct_random, shares, threshold, shared_key, index, C.byref(rng), out)
    if rc:
        raise ValueError("invalid ciphertext element")
    return out.raw


def decryption_share_verify(key_share: bytes, shares: int, threshold: int, shared_key: bytes, index: int, item: bytes) -> int:
    return lib().or_decryption_share_verify(key_share, shares, threshold, shared_key, index, item)
