"""elastic_elgamal_amd -- MI355X (gfx950) batch verifier for slowli/elastic-elgamal ballots.

Python host-side mirror of the reference interface for the ballot-verification hot path, bound with
ctypes to the C ABI of ``libeg_hip.so`` (``include/eg_hip.h``).  Names follow the reference:

* :class:`Ristretto`              -- the ``Group`` backend (src/group/ristretto.rs), batched
* :class:`ChoiceParams`           -- ``ChoiceParams::single / ::multi`` (src/app/choice.rs:132-196) with
  ``verify_batch`` = ``EncryptedChoice::verify`` for every ballot (choice.rs:358-380) + the homomorphic tally
  of examples/voting.rs:199-203
* :class:`QuadraticVotingParams`  -- ``QuadraticVotingParams::new`` (src/app/quadratic_voting.rs:63-76) with
  ``verify_batch`` = ``QuadraticVotingBallot::verify`` (quadratic_voting.rs:291-329)

There is NO CPU path in this package: if the HIP library is missing or no gfx950 device is usable, importing
works but creating a :class:`Context` raises.  (The CPU oracle lives under ``oracle/`` and is test-only.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

__all__ = [
    "Context", "Ristretto", "ChoiceParams", "QuadraticVotingParams", "PublicKeyVerifier", "DecryptionShareVerifier", "SumOfSquaresVerifier", "EgError", "library_path", "build",
    "STATUS_NAMES", "status_kind", "status_detail", "pack_json", "JsonPacker", "PACK_RESHAPE", "verify_json_multi", "json_stream_multi",
]

_PKG = Path(__file__).resolve().parent
_LIB = Path(os.environ.get("EG_LIB", str(_PKG / "libeg_hip.so")))   # EG_LIB: alternate build for A/B measurements

OK, BAD_SCALAR, BAD_POINT, OPTIONS_LEN, SUM_CHALLENGE, RANGE_LEN, RANGE_CHALLENGE = range(7)
QV_VARIANT_LEN, QV_VARIANT_CHALLENGE, QV_CREDIT_RANGE_LEN, QV_CREDIT_RANGE_CHALLENGE = 7, 8, 9, 10
QV_CREDIT_EQUIV_LEN, QV_CREDIT_EQUIV_CHALLENGE = 11, 12
MALFORMED = 13
STATUS_NAMES = {
    0: "Ok", 1: "BadScalar", 2: "BadPoint", 3: "OptionsLenMismatch", 4: "Sum(ChallengeMismatch)",
    5: "Range(LenMismatch)", 6: "Range(ChallengeMismatch)", 7: "Variant(LenMismatch)", 8: "Variant(ChallengeMismatch)",
    9: "CreditRange(LenMismatch)", 10: "CreditRange(ChallengeMismatch)", 11: "CreditEquivalence(LenMismatch)",
    12: "CreditEquivalence(ChallengeMismatch)", 13: "Malformed",
}


def status_kind(s: int) -> int:
    return s & 0xFF


def status_detail(s: int) -> int:
    return s >> 8


class EgError(RuntimeError):
    pass


def effective_cores() -> int:
    """Hardware threads this process may actually use (affinity mask and cgroup CPU quota): the default size of the parser's thread pool.
    os.cpu_count() is the machine's - 256 on a box that grants 16 - and oversubscribing the pool 16-fold costs an order of magnitude."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return n


def library_path() -> Path:
    return _LIB


def build(verbose: bool = False) -> Path:
    """Compile csrc/eg_hip.hip for gfx950 with hipcc into the in-tree libeg_hip.so (no GPU needed)."""
    import subprocess

    deps = list((_PKG / "csrc").glob("*.hip")) + list((_PKG / "csrc").glob("*.cuh")) + list((_PKG / "csrc").glob("*.h*"))
    deps.append(_PKG.parent / "include" / "eg_hip.h")
    if _LIB.exists() and all(d.stat().st_mtime <= _LIB.stat().st_mtime for d in deps):
        return _LIB
    cmd = ["make", "-C", str(_PKG / "csrc"), "-j4"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return _LIB


_lib = None
_RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)      # eg_json_release_fn
# blocks given back by the library to feed_owned_ptr callers: counted by the library's own eg_json_release_count (a C function: a Python
# callback would take the interpreter lock on the stream's worker thread once per block, in competition with the feeding thread)
_released_blocks = C.c_size_t(0)


def released_blocks() -> int:
    return int(_released_blocks.value)


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB.exists():
        raise EgError(
            f"{_LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # PyTorch-ROCm wheels carry their OWN libamdhip64 / HSA runtime.  Two HIP runtimes in one process do not share the device: whichever
    # loads second reports "No HIP GPUs are available".  A process that uses torch for device memory (tests, bench.py) must therefore load
    # torch's runtime first; libeg_hip.so then binds to the copy that is already there.  (A host without torch is unaffected.)
    import sys
    if "torch" not in sys.modules and not os.environ.get("EG_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(str(_LIB))
    vp, cp, sz = C.c_void_p, C.c_char_p, C.c_size_t
    sig = {
        "eg_init": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "eg_destroy": (None, [vp]),
        "eg_last_error": (cp, []),
        "eg_device_name": (C.c_int, [vp, cp, sz]),
        "eg_synchronize": (C.c_int, [vp]),
        "eg_scalar_from_wide_batch": (C.c_int, [vp, sz, cp, cp]),
        "eg_scalar_is_canonical_batch": (C.c_int, [vp, sz, cp, cp]),
        "eg_scalar_muladd_batch": (C.c_int, [vp, sz, cp, cp, cp, cp]),
        "eg_scalar_neg_batch": (C.c_int, [vp, sz, cp, cp]),
        "eg_scalar_invert_batch": (C.c_int, [vp, sz, cp, cp]),
        "eg_point_is_identity_batch": (C.c_int, [vp, sz, cp, cp, cp]),
        "eg_point_roundtrip_batch": (C.c_int, [vp, sz, cp, cp, cp]),
        "eg_point_add_batch": (C.c_int, [vp, sz, cp, cp, C.c_int, cp, cp]),
        "eg_mul_generator_batch": (C.c_int, [vp, sz, cp, cp]),
        "eg_vartime_double_mul_generator_batch": (C.c_int, [vp, sz, cp, cp, cp, cp, cp]),
        "eg_vartime_multi_mul_batch": (C.c_int, [vp, sz, sz, cp, cp, cp, cp]),
        "eg_abi_version": (C.c_int, []),
        "eg_msm_scratch_bytes_ctx": (sz, [vp, sz, sz]),
        "eg_selfbench_fmul": (C.c_int, [vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "eg_choice_prepare_wide_tables": (C.c_int, [vp]),
        "eg_qv_prepare_wide_tables": (C.c_int, [vp]),
        "eg_combine_shares": (C.c_int, [vp, C.c_uint64, C.c_uint64, sz, C.POINTER(C.c_uint64), cp, cp, C.POINTER(C.c_int)]),
        "eg_dlog_table_create": (C.c_int, [vp, sz, C.POINTER(C.c_uint64), C.POINTER(vp)]),
        "eg_dlog_table_destroy": (None, [vp]),
        "eg_dlog_table_get": (C.c_int, [vp, sz, cp, C.POINTER(C.c_uint64), cp]),
        "eg_vartime_multi_mul_batch_device": (C.c_int, [vp, sz, sz, vp, vp, vp, vp, vp, vp, vp]),
        "eg_prepared_point_size": (sz, []),
        "eg_points_prepare_device": (C.c_int, [vp, sz, vp, vp, vp, vp]),
        "eg_vartime_multi_mul_prepared_batch_device": (C.c_int, [vp, sz, sz, vp, vp, vp, vp, vp, vp]),
        "eg_choice_params_create": (C.c_int, [vp, cp, C.c_int, C.c_int, C.POINTER(vp)]),
        "eg_choice_params_destroy": (None, [vp]),
        "eg_choice_ballot_size": (sz, [C.c_int, C.c_int]),
        "eg_verify_choice_batch": (C.c_int, [vp, sz, vp, vp, vp]),
        "eg_verify_choice_batch_device": (C.c_int, [vp, sz, vp, vp, vp]),
        "eg_choice_tally_reset": (C.c_int, [vp]),
        "eg_choice_tally_add": (C.c_int, [vp, cp]),
        "eg_qv_tally_add": (C.c_int, [vp, cp]),
        "eg_choice_tally_encode": (C.c_int, [vp, cp]),
        "eg_qv_params_create": (C.c_int, [vp, cp, C.c_int, C.c_uint64, C.POINTER(vp)]),
        "eg_qv_params_destroy": (None, [vp]),
        "eg_qv_ballot_size": (sz, [vp]),
        "eg_verify_qv_batch": (C.c_int, [vp, sz, vp, vp, vp]),
        "eg_verify_qv_batch_device": (C.c_int, [vp, sz, vp, vp, vp]),
        "eg_qv_tally_reset": (C.c_int, [vp]),
        "eg_qv_tally_encode": (C.c_int, [vp, cp]),
        "eg_verify_choice_batch_multi": (C.c_int, [C.POINTER(vp), C.c_int, sz, vp, vp, vp]),
        "eg_verify_qv_batch_multi": (C.c_int, [C.POINTER(vp), C.c_int, sz, vp, vp, vp]),
        "eg_verify_choice_batch_multi_device": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(sz), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]),
        "eg_verify_qv_batch_multi_device": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(sz), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]),
        "eg_choice_tally_encode_multi": (C.c_int, [C.POINTER(vp), C.c_int, cp]),
        "eg_qv_tally_encode_multi": (C.c_int, [C.POINTER(vp), C.c_int, cp]),
        "eg_choice_tally_reset_async": (C.c_int, [vp, vp]),
        "eg_choice_tally_encode_device": (C.c_int, [vp, vp, vp]),
        "eg_qv_tally_reset_async": (C.c_int, [vp, vp]),
        "eg_qv_tally_encode_device": (C.c_int, [vp, vp, vp]),
        "eg_points_sum_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp]),
        "eg_proof_params_create": (C.c_int, [vp, cp, C.c_int, C.c_uint64, C.POINTER(vp)]),
        "eg_share_params_create": (C.c_int, [vp, cp, C.c_uint64, C.c_uint64, C.c_uint64, cp, C.POINTER(vp)]),
        "eg_proof_params_destroy": (None, [vp]),
        "eg_proof_item_size": (sz, [vp]),
        "eg_verify_proof_batch": (C.c_int, [vp, sz, vp, vp]),
        "eg_verify_proof_batch_device": (C.c_int, [vp, sz, vp, vp, vp]),
        "eg_choice_encrypt_batch": (C.c_int, [vp, C.c_uint64, sz, sz, C.c_int, cp]),
        "eg_choice_encrypt_batch_device": (C.c_int, [vp, C.c_uint64, sz, sz, C.c_int, vp, vp]),
        "eg_qv_encrypt_batch_device": (C.c_int, [vp, C.c_uint64, sz, sz, vp, vp]),
        "eg_choice_encrypt_selected_batch": (C.c_int, [vp, C.c_uint64, sz, sz, C.c_uint64, vp, cp]),
        "eg_choice_encrypt_selected_batch_device": (C.c_int, [vp, C.c_uint64, sz, sz, C.c_uint64, vp, vp, vp]),
        "eg_qv_encrypt_votes_batch": (C.c_int, [vp, C.c_uint64, sz, sz, C.c_uint64, vp, cp]),
        "eg_qv_encrypt_votes_batch_device": (C.c_int, [vp, C.c_uint64, sz, sz, C.c_uint64, vp, vp, vp]),
        "eg_sumsq_params_create": (C.c_int, [vp, cp, C.c_int, cp, sz, C.POINTER(vp)]),
        "eg_merlin_challenge_batch": (C.c_int, [vp, sz, cp, sz, cp, sz, cp, sz, cp, sz, cp, sz]),
        "eg_choice_pack_json": (C.c_int, [C.c_int, C.c_int, cp, sz, C.c_int, sz, vp, vp, C.POINTER(sz)]),
        "eg_qv_pack_json": (C.c_int, [C.c_int, C.c_uint64, cp, sz, C.c_int, sz, vp, vp, C.POINTER(sz)]),
        "eg_qv_ballot_size_for": (sz, [C.c_int, C.c_uint64]),
        "eg_verify_choice_json": (C.c_int, [vp, vp, sz, C.c_int, sz, vp, C.POINTER(sz), vp]),
        "eg_verify_qv_json": (C.c_int, [vp, vp, sz, C.c_int, sz, vp, C.POINTER(sz), vp]),
        "eg_verify_choice_json_multi": (C.c_int, [C.POINTER(vp), C.c_int, vp, sz, C.c_int, sz, vp, C.POINTER(sz), vp]),
        "eg_verify_qv_json_multi": (C.c_int, [C.POINTER(vp), C.c_int, vp, sz, C.c_int, sz, vp, C.POINTER(sz), vp]),
        "eg_verify_choice_json_begin_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]),
        "eg_verify_qv_json_begin_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]),
        "eg_verify_choice_json_begin": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
        "eg_verify_qv_json_begin": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
        "eg_verify_json_feed": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
        "eg_verify_json_feed_owned": (C.c_int, [vp, vp, sz, vp, vp, C.POINTER(sz)]),
        "eg_json_release_count": (None, [vp, vp, sz]),
        "eg_verify_json_take": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
        "eg_verify_json_end": (C.c_int, [vp, vp, sz, C.POINTER(sz), C.POINTER(sz), vp]),
        "eg_verify_json_abort": (None, [vp]),
        "eg_range_decomposition": (C.c_int, [C.c_uint64, cp, sz]),
        "eg_plan_describe": (C.c_int, [C.c_int, C.c_int, C.c_uint64, cp, sz]),
        "eg_profile_enable": (C.c_int, [vp, C.c_int]),
        "eg_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
        "eg_profile_read_tables": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "eg_selfcheck_generator_table": (C.c_int, [vp, C.c_int, sz, C.c_uint64, C.POINTER(C.c_uint64)]),
        "eg_comb_table_bits": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        if "EG_LIB" in os.environ and not hasattr(lib, name):
            continue             # an alternate (older) build selected for an A/B measurement may lack the newest entry points
        fn = getattr(lib, name)  # raises AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if "EG_LIB" not in os.environ and lib.eg_abi_version() != ABI_VERSION:
        raise EgError(f"{_LIB} speaks ABI version {lib.eg_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


PACK_RESHAPE = 0xFFFFFFFE
ABI_VERSION = 6          # include/eg_hip.h: EG_ABI_VERSION


def pack_json(text, n_options: int, single: bool | None = None, credits: int | None = None, threads: int = 0, max_objects: int = 0):
    """Native wire ingest (csrc/wire_json.hpp; no GPU needed): ballots in serde's JSON layout (one array, or objects back to
    back) -> (packed bytes of ALL objects, status words).  Object k occupies packed[k*size:(k+1)*size] and is valid iff
    status[k] == 0; MALFORMED = does not deserialise, PACK_RESHAPE = wrong number of choices / responses (object path).
    `credits` selects QuadraticVotingBallot, otherwise EncryptedChoice with `single`."""
    lib = _load()
    data = text.encode() if isinstance(text, str) else bytes(text)
    if not threads:
        threads = effective_cores()
    size = lib.eg_qv_ballot_size_for(n_options, credits) if credits is not None else lib.eg_choice_ballot_size(n_options, int(bool(single)))
    if not size:
        raise EgError("bad election parameters")
    if not max_objects:
        # the library counts the objects itself (max_objects = 0: nothing is written, *n_objects says how many there are); counting a
        # key word instead would undercount texts that hold junk objects, and one voter's junk must never block the others' ballots
        n0 = sz_t(0)
        if credits is not None:
            rc = lib.eg_qv_pack_json(n_options, credits, data, len(data), threads, 0, None, None, C.byref(n0))
        else:
            rc = lib.eg_choice_pack_json(n_options, int(bool(single)), data, len(data), threads, 0, None, None, C.byref(n0))
        if rc != 0 and n0.value == 0:
            _check(rc)
        max_objects = n0.value
    packed = C.create_string_buffer(max(max_objects * size, 1))
    status = (C.c_uint32 * max(max_objects, 1))()
    n = sz_t(0)
    if credits is not None:
        _check(lib.eg_qv_pack_json(n_options, credits, data, len(data), threads, max_objects, packed, status, C.byref(n)))
    else:
        _check(lib.eg_choice_pack_json(n_options, int(bool(single)), data, len(data), threads, max_objects, packed, status, C.byref(n)))
    return packed.raw[: n.value * size], list(status[: n.value])


sz_t = C.c_size_t


class JsonPacker:
    """The native packer with caller-owned, reusable output buffers (what a native host does): no allocation or copy per call.
    `pack(text)` returns the number of objects; `packed` (a ctypes buffer) and `status` hold the results."""

    def __init__(self, n_options: int, max_objects: int, single: bool | None = None, credits: int | None = None, threads: int = 0):
        lib = _load()
        self.n_options, self.single, self.credits, self.max_objects = n_options, single, credits, max_objects
        self.threads = threads or effective_cores()
        self.ballot_size = (lib.eg_qv_ballot_size_for(n_options, credits) if credits is not None
                            else lib.eg_choice_ballot_size(n_options, int(bool(single))))
        if not self.ballot_size:
            raise EgError("bad election parameters")
        self.packed = C.create_string_buffer(max(max_objects * self.ballot_size, 1))
        self.status = (C.c_uint32 * max(max_objects, 1))()

    def pack(self, data: bytes) -> int:
        n = sz_t(0)
        lib = _load()
        if self.credits is not None:
            _check(lib.eg_qv_pack_json(self.n_options, self.credits, data, len(data), self.threads, self.max_objects, self.packed,
                                       self.status, C.byref(n)))
        else:
            _check(lib.eg_choice_pack_json(self.n_options, int(bool(self.single)), data, len(data), self.threads, self.max_objects,
                                           self.packed, self.status, C.byref(n)))
        return n.value


def range_decomposition(upper_bound: int) -> str:
    """``RangeDecomposition::optimal(upper_bound).to_string()`` by the product's host code (no GPU needed)."""
    b = C.create_string_buffer(1024)
    _check(_load().eg_range_decomposition(upper_bound, b, 1024))
    return b.value.decode()


def plan_describe(kind: str, n_options: int = 0, credits_or_bound: int = 0) -> dict:
    """Summary of the flattened verification plan (host logic only, no GPU needed)."""
    import json

    kinds = {"single": 0, "multi": 1, "qv": 2, "zero": 3, "bool": 4, "range": 5, "sumsq": 6}
    b = C.create_string_buffer(4096)
    _check(_load().eg_plan_describe(kinds[kind], n_options, credits_or_bound, b, 4096))
    return json.loads(b.value.decode())


def exported_symbols():
    """Names declared in include/eg_hip.h that the library must export (used by the CPU-side ABI test)."""
    import re

    hdr = (_PKG.parent / "include" / "eg_hip.h").read_text()
    return sorted(set(re.findall(r"\b(eg_[a-z0-9_]+)\s*\(", hdr)))


def _check(rc: int) -> None:
    if rc != 0:
        raise EgError(f"libeg_hip error {rc}: {_load().eg_last_error().decode()}")


class Context:
    """One GPU context per process (eg_init).  No reference analogue: the backend there is a ZST."""

    def __init__(self, device: int = 0):
        lib = _load()
        self._h = C.c_void_p()
        _check(lib.eg_init(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            _load().eg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def name(self) -> str:
        b = C.create_string_buffer(256)
        _check(_load().eg_device_name(self._h, b, 256))
        return b.value.decode()

    def synchronize(self):
        _check(_load().eg_synchronize(self._h))

    def points_sum_device(self, n_ranks: int, n_points: int, d_in: int, d_out: int, stream: int = 0, d_bad: int = 0):
        """d_out[k] = sum over ranks of d_in[r][k] (32-byte encodings): merge of all-gathered per-GPU tallies.
        d_bad: device uint32 counter (zeroed by the caller) of slots that received an undecodable encoding."""
        _check(_load().eg_points_sum_device(self._h, n_ranks, n_points, d_in, d_out, d_bad or None, stream))

    def merlin_challenges(self, proto: bytes, msg_label: bytes, msgs, chal_label: bytes, out_len: int = 64):
        """``Transcript::new(proto); append_message(msg_label, m); challenge_bytes(chal_label, out_len)`` for every message m
        (all of one length) on the GPU: the transcript layer of proofs/mod.rs:39-57 as a primitive."""
        msgs = list(msgs)
        n = len(msgs)
        ml = len(msgs[0]) if n else 0
        if any(len(m) != ml for m in msgs):
            raise ValueError("messages must have one length")
        out = C.create_string_buffer(max(n * out_len, 1))
        _check(_load().eg_merlin_challenge_batch(self._h, n, proto, len(proto), msg_label, len(msg_label), b"".join(msgs), ml,
                                                 chal_label, len(chal_label), out, out_len))
        return [out.raw[i * out_len : (i + 1) * out_len] for i in range(n)]

    def profile_enable(self, on=True):
        """on = 2: profile and run the chunks of a call serially on one work set (a launch's duration is then its own)."""
        _check(_load().eg_profile_enable(self._h, int(on)))

    def profile_read(self):
        """(dominant-kernel ms total, launches, whole-verify ms total) since the last read; HIP events."""
        a, n, b = C.c_double(), C.c_uint64(), C.c_double()
        _check(_load().eg_profile_read(self._h, C.byref(a), C.byref(n), C.byref(b)))
        return a.value, n.value, b.value

    def profile_read_tables(self):
        """(k_base_tables ms total, launches) for the launches folded in by the last profile_read()."""
        a, n = C.c_double(), C.c_uint64()
        _check(_load().eg_profile_read_tables(self._h, C.byref(a), C.byref(n)))
        return a.value, n.value


    def selfbench_fmul(self, seconds: float = 2.0):
        """(G field multiplications/s, shader clock MHz) of the shipped fe_mul in a bare chain sustained for `seconds` on this box
        (eg_selfbench_fmul): the VALU roof bench.py quotes its fractions against."""
        g, f = C.c_double(), C.c_double()
        _check(_load().eg_selfbench_fmul(self._h, float(seconds), C.byref(g), C.byref(f)))
        return g.value, f.value

    def comb_table_bits(self):
        """(window bits of the comb tables built at start-up, window bits of the wide tables or 0 while they do not exist)."""
        a, b = C.c_int(), C.c_int()
        _check(_load().eg_comb_table_bits(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def selfcheck_generator_table(self, wide: bool = False, samples: int = 4096, seed: int = 1) -> int:
        """Entries of the generator's comb table that differ from an entry-by-entry recomputation (must be 0); wide=True
        checks (and first builds) the wide table that large batches use."""
        bad = C.c_uint64()
        _check(_load().eg_selfcheck_generator_table(self._h, int(wide), samples, seed, C.byref(bad)))
        return bad.value


class Ristretto:
    """Batched ``Group`` backend for Ristretto255 (src/group/ristretto.rs).  All arguments are concatenated
    32-byte encodings (64-byte for wide scalars); every method runs one GPU lane per problem."""

    SCALAR_SIZE = 32
    ELEMENT_SIZE = 32

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def scalar_from_random_bytes(self, wide: bytes) -> bytes:  # ristretto.rs:34-38
        n = len(wide) // 64
        out = C.create_string_buffer(32 * n)
        _check(_load().eg_scalar_from_wide_batch(self.ctx._h, n, wide, out))
        return out.raw

    def deserialize_scalar_ok(self, scalars: bytes) -> bytes:  # ristretto.rs:59-62
        n = len(scalars) // 32
        ok = C.create_string_buffer(n)
        _check(_load().eg_scalar_is_canonical_batch(self.ctx._h, n, scalars, ok))
        return ok.raw

    def scalar_muladd(self, a: bytes, b: bytes, c: bytes) -> bytes:
        n = len(a) // 32
        out = C.create_string_buffer(32 * n)
        _check(_load().eg_scalar_muladd_batch(self.ctx._h, n, a, b, c, out))
        return out.raw

    def scalar_neg(self, a: bytes) -> bytes:
        n = len(a) // 32
        out = C.create_string_buffer(32 * n)
        _check(_load().eg_scalar_neg_batch(self.ctx._h, n, a, out))
        return out.raw

    def invert_scalars(self, a: bytes) -> bytes:  # group/mod.rs:104-118, ristretto.rs:40-52
        n = len(a) // 32
        out = C.create_string_buffer(32 * n)
        _check(_load().eg_scalar_invert_batch(self.ctx._h, n, a, out))
        return out.raw

    def is_identity(self, elements: bytes):
        """ElementOps::is_identity (ristretto.rs:80-82): (is_identity flags, ok flags)."""
        n = len(elements) // 32
        f, ok = C.create_string_buffer(n), C.create_string_buffer(n)
        _check(_load().eg_point_is_identity_batch(self.ctx._h, n, elements, f, ok))
        return f.raw, ok.raw

    def element_neg(self, a: bytes):
        return self.element_add(bytes(len(a)), a, subtract=True)

    def element_roundtrip(self, elements: bytes):
        """deserialize_element + serialize_element (ristretto.rs:88-95): (re-encodings, ok flags)."""
        n = len(elements) // 32
        out, ok = C.create_string_buffer(32 * n), C.create_string_buffer(n)
        _check(_load().eg_point_roundtrip_batch(self.ctx._h, n, elements, out, ok))
        return out.raw, ok.raw

    def element_add(self, a: bytes, b: bytes, subtract: bool = False):
        n = len(a) // 32
        out, ok = C.create_string_buffer(32 * n), C.create_string_buffer(n)
        _check(_load().eg_point_add_batch(self.ctx._h, n, a, b, int(subtract), out, ok))
        return out.raw, ok.raw

    def mul_generator(self, k: bytes) -> bytes:  # ristretto.rs:105-121
        n = len(k) // 32
        out = C.create_string_buffer(32 * n)
        _check(_load().eg_mul_generator_batch(self.ctx._h, n, k, out))
        return out.raw

    def vartime_double_mul_generator(self, k: bytes, p: bytes, r: bytes):  # ristretto.rs:131-137
        n = len(k) // 32
        out, ok = C.create_string_buffer(32 * n), C.create_string_buffer(n)
        _check(_load().eg_vartime_double_mul_generator_batch(self.ctx._h, n, k, p, r, out, ok))
        return out.raw, ok.raw

    def vartime_multi_mul(self, terms: int, scalars: bytes, points: bytes):  # ristretto.rs:139-145
        n = len(scalars) // (32 * terms) if terms else 0
        out, ok = C.create_string_buffer(32 * max(n, 1)), C.create_string_buffer(max(n, 1))
        _check(_load().eg_vartime_multi_mul_batch(self.ctx._h, n, terms, scalars, points, out, ok))
        return out.raw[: 32 * n], ok.raw[:n]

    def msm_scratch_bytes(self, n: int, terms: int) -> int:
        return int(_load().eg_msm_scratch_bytes_ctx(self.ctx._h, n, terms))

    def prepare_points_device(self, n: int, d_encodings: int, d_prepared: int, d_ok: int, stream: int = 0):
        """Decodes n elements once into prepared points (prepared_point_size() bytes each, 16-byte aligned) for repeated products over
        them.  d_ok (n bytes, device) is mandatory: an encoding that does not decode is prepared as the identity, and this is the only
        place that says so."""
        _check(_load().eg_points_prepare_device(self.ctx._h, n, d_encodings, d_prepared, d_ok, stream))

    def vartime_multi_mul_prepared_device(self, n: int, terms: int, d_scalars: int, d_prepared: int, d_out: int, d_r: int = 0, d_scratch: int = 0,
                                          stream: int = 0):
        """vartime_multi_mul over prepared points (no decoding per term)."""
        _check(_load().eg_vartime_multi_mul_prepared_batch_device(self.ctx._h, n, terms, d_scalars, d_prepared, d_r, d_scratch, d_out, stream))

    def vartime_multi_mul_device(self, n: int, terms: int, d_scalars: int, d_points: int, d_out: int, d_r: int = 0, d_scratch: int = 0,
                                 d_ok: int = 0, stream: int = 0):
        """The multi-scalar multiplication on device buffers, asynchronous on `stream` (eg_vartime_multi_mul_batch_device)."""
        _check(_load().eg_vartime_multi_mul_batch_device(self.ctx._h, n, terms, d_scalars, d_points, d_r, d_scratch, d_out, d_ok, stream))


def prepared_point_size() -> int:
    return int(_load().eg_prepared_point_size())


def verify_batch_multi(per_device, ballots: bytes, with_tally: bool = True):
    """``eg_verify_*_batch_multi``: one batch over several params objects of the same election, each on its own context (normally one
    per GPU of the process): contiguous slabs, one host thread per object inside the library, tallies merged in the library.
    Returns (status words in ballot order, tally of this batch or None)."""
    per_device = list(per_device)
    p0 = per_device[0]
    if any(type(p) is not type(p0) for p in per_device):
        raise ValueError("params objects of different kinds")
    n = len(ballots) // p0.ballot_size
    if n * p0.ballot_size != len(ballots):
        raise ValueError("ballots is not a whole number of packed ballots")
    arr = (C.c_void_p * len(per_device))(*[p._h for p in per_device])
    st = (C.c_uint32 * max(n, 1))()
    tally = C.create_string_buffer(64 * p0.n_options) if with_tally else None
    buf = (C.c_char * max(len(ballots), 1)).from_buffer_copy(ballots or b"\0")
    fn = getattr(_load(), f"eg_verify_{p0._prefix}_batch_multi")
    _check(fn(arr, len(per_device), n, buf, st, tally))
    return list(st[:n]), (tally.raw if with_tally else None)


def verify_batch_multi_host_ptr(per_device, n: int, ballots_ptr: int, status_ptr: int, with_tally: bool = False):
    """``eg_verify_*_batch_multi`` on raw host addresses (e.g. ONE pinned torch tensor holding the whole batch): what a single-process
    host with its ballots in host memory calls; no Python-side copies.  Returns the tally of this batch or None."""
    per_device = list(per_device)
    p0 = per_device[0]
    arr = (C.c_void_p * len(per_device))(*[p._h for p in per_device])
    tally = C.create_string_buffer(64 * p0.n_options) if with_tally else None
    fn = getattr(_load(), f"eg_verify_{p0._prefix}_batch_multi")
    _check(fn(arr, len(per_device), n, ballots_ptr, status_ptr, tally))
    return tally.raw if with_tally else None


def verify_batch_multi_device(per_device, counts, d_ballots, d_status, streams=None, with_tally: bool = False):
    """``eg_verify_*_batch_multi_device``: slab d (counts[d] ballots at device pointer d_ballots[d], verdicts to d_status[d]) is already
    resident on the GPU of per_device[d]; streams[d] a stream of that device (None: the null streams).  Returns when every verdict is
    written; with_tally: the tally of this call's batch alone (the running tallies advance either way)."""
    per_device = list(per_device)
    p0, k = per_device[0], len(per_device)
    if any(type(p) is not type(p0) for p in per_device):
        raise ValueError("params objects of different kinds")
    if not (len(counts) == len(d_ballots) == len(d_status) == k) or (streams is not None and len(streams) != k):
        raise ValueError("one count, ballot pointer, status pointer (and stream) per params object")
    arr = (C.c_void_p * k)(*[p._h for p in per_device])
    cnt = (C.c_size_t * k)(*counts)
    db = (C.c_void_p * k)(*d_ballots)
    ds = (C.c_void_p * k)(*d_status)
    ss = (C.c_void_p * k)(*streams) if streams is not None else None
    tally = C.create_string_buffer(64 * p0.n_options) if with_tally else None
    fn = getattr(_load(), f"eg_verify_{p0._prefix}_batch_multi_device")
    _check(fn(arr, k, cnt, db, ds, ss, tally))
    return tally.raw if with_tally else None


def verify_json_multi_into(per_device, data: bytes, status, threads: int = 0, tally=None) -> int:
    """``eg_verify_*_json_multi`` with caller-owned buffers (status: ctypes uint32 array): ONE parser, the packed windows dealt to the
    params objects (one per GPU), verdicts in text order.  Returns the object count."""
    per_device = list(per_device)
    p0 = per_device[0]
    arr = (C.c_void_p * len(per_device))(*[p._h for p in per_device])
    n = sz_t(0)
    fn = getattr(_load(), f"eg_verify_{p0._prefix}_json_multi")
    _check(fn(arr, len(per_device), data, len(data), threads or effective_cores(), len(status), status, C.byref(n), tally))
    return n.value


def verify_json_multi(per_device, text, max_objects: int = 0, threads: int = 0, with_tally: bool = True):
    """JSON text -> (status words in text order, tally of this call) over several params objects of one election."""
    per_device = list(per_device)
    data = text.encode() if isinstance(text, str) else bytes(text)
    if not max_objects:
        max_objects = data.count(b"{") + 1
    st = (C.c_uint32 * max(max_objects, 1))()
    tally = C.create_string_buffer(64 * per_device[0].n_options) if with_tally else None
    n = verify_json_multi_into(per_device, data, st, threads, tally)
    return list(st[:n]), (tally.raw if with_tally else None)


def json_stream_multi(per_device, threads: int = 0) -> "JsonStream":
    """``eg_verify_*_json_begin_multi``: a JSON stream over several params objects (feed / take / end / abort as for one)."""
    return JsonStream(list(per_device), threads)


def tally_encode_multi(per_device) -> bytes:
    """Sum of the running tallies of several params objects of one election (``eg_*_tally_encode_multi``)."""
    per_device = list(per_device)
    p0 = per_device[0]
    arr = (C.c_void_p * len(per_device))(*[p._h for p in per_device])
    out = C.create_string_buffer(64 * p0.n_options)
    _check(getattr(_load(), f"eg_{p0._prefix}_tally_encode_multi")(arr, len(per_device), out))
    return out.raw


class JsonStream:
    """``eg_verify_*_json_begin / eg_verify_json_feed / _take / _end / _abort``: the JSON text of a batch of ballots in pieces of any size
    (a ballot may straddle pieces); verdicts and tally are those of ``verify_json`` on the concatenated text.  Between begin and
    end / abort the params object belongs to the stream."""

    def __init__(self, params, threads: int = 0):
        import weakref

        self._h = C.c_void_p()
        self.objects = 0
        if isinstance(params, (list, tuple)):           # several params objects of one election (one per GPU): eg_verify_*_json_begin_multi
            self.params, self.all_params = params[0], list(params)
            arr = (C.c_void_p * len(params))(*[p._h for p in params])
            fn = getattr(_load(), f"eg_verify_{params[0]._prefix}_json_begin_multi")
            _check(fn(arr, len(params), threads or effective_cores(), C.byref(self._h)))
            for p in params:                            # a params object that is destroyed takes the whole stream with it
                if not hasattr(p, "_streams"):
                    p._streams = weakref.WeakSet()
                p._streams.add(self)
        else:
            self.params, self.all_params = params, [params]
            fn = getattr(_load(), f"eg_verify_{params._prefix}_json_begin")
            _check(fn(params._h, threads or effective_cores(), C.byref(self._h)))

    def feed(self, piece) -> int:
        """The next piece of the text (bytes / bytearray / memoryview / str); returns the number of complete objects seen so far."""
        if isinstance(piece, str):
            piece = piece.encode()
        piece = bytes(piece)
        n = C.c_size_t(0)
        _check(_load().eg_verify_json_feed(self._h, C.cast(C.c_char_p(piece), C.c_void_p), len(piece), C.byref(n)))
        self.objects = n.value
        return n.value

    def feed_ptr(self, ptr: int, length: int) -> int:
        n = C.c_size_t(0)
        _check(_load().eg_verify_json_feed(self._h, C.c_void_p(ptr), length, C.byref(n)))
        self.objects = n.value
        return n.value

    def feed_owned_ptr(self, ptr: int, length: int) -> int:
        """``eg_verify_json_feed_owned``: the library reads the block at `ptr` in place - no copy on this thread, no wait for the worker -
        until the stream has been ended or aborted; the CALLER keeps the memory alive until then (the release function is the library's own
        eg_json_release_count: released_blocks() says how many blocks have come back)."""
        n = C.c_size_t(0)
        lib = _load()
        _check(lib.eg_verify_json_feed_owned(self._h, C.c_void_p(ptr), length, C.cast(lib.eg_json_release_count, C.c_void_p), C.byref(_released_blocks),
                                             C.byref(n)))
        self.objects = n.value
        return n.value

    def take(self, cap: int = 1 << 20):
        """Verdicts that are final so far (in order, from where the last take stopped).  Never waits for the GPU to drain, but it does take
        the context's lock: it may wait for the worker to finish the window of the piece it is cutting, which - while the staging ring is
        full - includes the worker's own wait for the oldest submission (include/eg_hip.h, `take`)."""
        st = (C.c_uint32 * max(cap, 1))()
        n = C.c_size_t(0)
        _check(_load().eg_verify_json_take(self._h, st, cap, C.byref(n)))
        return list(st[: n.value])

    def end(self, with_tally: bool = True, cap: int | None = None):
        """Flushes the stream: (verdicts not yet taken, tally of the stream's ballots or None).  The stream is gone afterwards."""
        cap = self.objects + 1024 if cap is None else cap
        tally = C.create_string_buffer(64 * self.params.n_options) if with_tally else None
        h, self._h = self._h, None
        while True:
            st = (C.c_uint32 * max(cap, 1))()
            n, total = C.c_size_t(0), C.c_size_t(0)
            rc = _load().eg_verify_json_end(h, st, cap, C.byref(n), C.byref(total), tally)
            if not (rc and n.value > cap):
                break
            cap = n.value                      # the stream is still open (and flushed): it had cut more objects than feed() had reported yet
        _check(rc)
        self.objects = total.value
        return list(st[: n.value]), (tally.raw if with_tally else None)

    def end_into(self, status, with_tally: bool = False):
        """As end(), verdicts into a caller-owned ctypes uint32 array (bench.py: no Python list of a million words)."""
        n, total = C.c_size_t(0), C.c_size_t(0)
        tally = C.create_string_buffer(64 * self.params.n_options) if with_tally else None
        h, self._h = self._h, None
        rc = _load().eg_verify_json_end(h, status, len(status), C.byref(n), C.byref(total), tally)
        if rc and n.value > len(status):
            self._h = h                        # no room for the n.value verdicts that are left: the stream is still open
        _check(rc)
        self.objects = total.value
        return n.value, (tally.raw if with_tally else None)

    def abort(self):
        h, self._h = getattr(self, "_h", None), None
        # a params object that has been destroyed took its open stream with it (eg_*_params_destroy aborts it): the handle is stale then
        if h and all(getattr(p, "_h", None) for p in self.all_params):
            _load().eg_verify_json_abort(h)

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


class _BatchParams:
    _prefix = ""

    def json_stream(self, threads: int = 0) -> JsonStream:
        import weakref

        st = JsonStream(self, threads)
        if not hasattr(self, "_streams"):
            self._streams = weakref.WeakSet()
        self._streams.add(st)
        return st

    def _forget_streams(self):
        """The library aborts a stream that is still open when its params object is destroyed: the Python handles must not outlive it."""
        for st in list(getattr(self, "_streams", ())):
            st._h = None

    def _fn(self, name):
        return getattr(_load(), f"eg_{self._prefix}_{name}")

    def verify_batch(self, ballots: bytes, with_tally: bool = True):
        """verify() for every packed ballot; returns (status words, tally bytes or None).
        The returned tally is that of THIS batch: the component-wise sum of the ciphertexts of its accepted ballots,
        n_options x (R || B).  The running tally inside the params object accumulates them as well, exactly as
        verify_batch_device does (tally_reset / tally_add / tally_encode operate on the running tally)."""
        n = len(ballots) // self.ballot_size
        if n * self.ballot_size != len(ballots):
            raise ValueError("ballots is not a whole number of packed ballots")
        st = (C.c_uint32 * max(n, 1))()
        tally = C.create_string_buffer(64 * self.n_options) if with_tally else None
        buf = (C.c_char * max(len(ballots), 1)).from_buffer_copy(ballots or b"\0")
        fn = getattr(_load(), f"eg_verify_{self._prefix}_batch")
        _check(fn(self._h, n, buf, st, tally))
        return list(st[:n]), (tally.raw if with_tally else None)

    def verify_json(self, text, max_objects: int = 0, threads: int = 0, with_tally: bool = True):
        """JSON text (serde's layout; one array or objects back to back) -> (status words, tally of this call) through the
        native entry point: host threads pack piece k+1 while the GPU verifies piece k.  Objects that do not deserialise get
        MALFORMED; objects of another shape than the election's get the reference's verdict (OptionsLenMismatch / LenMismatch in
        verify()'s order) from the library's object path."""
        data = text.encode() if isinstance(text, str) else bytes(text)
        if not threads:
            threads = effective_cores()
        if not max_objects:
            max_objects = data.count(b"{") + 1
        st = (C.c_uint32 * max(max_objects, 1))()
        n = sz_t(0)
        tally = C.create_string_buffer(64 * self.n_options) if with_tally else None
        fn = getattr(_load(), f"eg_verify_{self._prefix}_json")
        _check(fn(self._h, data, len(data), threads, max_objects, st, C.byref(n), tally))
        return list(st[: n.value]), (tally.raw if with_tally else None)

    def verify_json_into(self, data: bytes, status, threads: int, tally=None) -> int:
        """The same with caller-owned buffers (status: ctypes uint32 array): the bare C call, for timing.  Returns the object count."""
        n = sz_t(0)
        fn = getattr(_load(), f"eg_verify_{self._prefix}_json")
        _check(fn(self._h, data, len(data), threads, len(status), status, C.byref(n), tally))
        return n.value

    def verify_batch_host_ptr(self, n: int, ballots_ptr: int, status_ptr: int, tally_ptr: int = 0):
        """The host-buffer entry point on raw host addresses (e.g. pinned torch tensors): no Python-side copies."""
        fn = getattr(_load(), f"eg_verify_{self._prefix}_batch")
        _check(fn(self._h, n, ballots_ptr, status_ptr, tally_ptr or None))

    def verify_batch_device(self, n: int, d_ballots: int, d_status: int, stream: int = 0):
        """Asynchronous device-pointer variant (torch tensors' data_ptr()); tally accumulates on the device."""
        fn = getattr(_load(), f"eg_verify_{self._prefix}_batch_device")
        _check(fn(self._h, n, d_ballots, d_status, stream))

    def prepare_wide_tables(self):
        """Build the wide fixed-base comb tables now instead of inside the first large verify call (eg_*_prepare_wide_tables)."""
        _check(self._fn("prepare_wide_tables")(self._h))

    def tally_reset(self, stream: int = 0):
        if stream:
            _check(self._fn("tally_reset_async")(self._h, stream))
        else:
            _check(self._fn("tally_reset")(self._h))

    def tally_encode_device(self, d_out: int, stream: int = 0):
        """Canonical encodings of the running tally (n_options x 64 bytes) into device memory, asynchronously."""
        _check(self._fn("tally_encode_device")(self._h, d_out, stream))

    def tally_add(self, encoded: bytes):
        """running tally += an encoded tally (checkpoint / resume, merging batches verified elsewhere)."""
        if len(encoded) != 64 * self.n_options:
            raise ValueError("encoded tally must be n_options x 64 bytes")
        _check(self._fn("tally_add")(self._h, encoded))

    def tally_encode(self) -> bytes:
        out = C.create_string_buffer(64 * self.n_options)
        _check(self._fn("tally_encode")(self._h, out))
        return out.raw


class ChoiceParams(_BatchParams):
    """``ChoiceParams::single(pk, n)`` / ``::multi(pk, n)`` (choice.rs:160-196)."""

    _prefix = "choice"

    def __init__(self, ctx: Context, public_key: bytes, options_count: int, single: bool = True):
        self.ctx, self.public_key, self.n_options, self.single = ctx, public_key, options_count, single
        self._h = C.c_void_p()
        _check(_load().eg_choice_params_create(ctx._h, public_key, options_count, int(single), C.byref(self._h)))
        self.ballot_size = _load().eg_choice_ballot_size(options_count, int(single))

    @property
    def kind_name(self) -> str:
        """The name plan_describe() knows this election's plan by."""
        return "single" if self.single else "multi"

    @classmethod
    def single_choice(cls, ctx, public_key, options_count):
        return cls(ctx, public_key, options_count, True)

    @classmethod
    def multi_choice(cls, ctx, public_key, options_count):
        return cls(ctx, public_key, options_count, False)

    def encrypt_batch(self, base_seed: int, first: int, n: int, n_selected: int = 0) -> bytes:
        """EncryptedChoice::new for n synthetic voters (GPU), returned packed in host memory."""
        out = C.create_string_buffer(max(n * self.ballot_size, 1))
        _check(_load().eg_choice_encrypt_batch(self._h, base_seed, first, n, n_selected, out))
        return out.raw[: n * self.ballot_size]

    def encrypt_batch_device(self, base_seed: int, first: int, n: int, d_out: int, n_selected: int = 0, stream: int = 0):
        """EncryptedChoice::new for n synthetic voters, written packed to device memory."""
        _check(_load().eg_choice_encrypt_batch_device(self._h, base_seed, first, n, n_selected, d_out, stream))

    def encrypt_selected(self, base_seed: int, first: int, selections, rng_skip: int = 0) -> bytes:
        """``EncryptedChoice::single(params, choice, rng)`` / ``::new(params, &[bool], rng)`` (choice.rs:296-349) on the GPU for the
        caller's choices: `selections` = one bitmask per ballot (bit k <=> option k chosen); ballot i uses
        ChaChaRng::seed_from_u64(base_seed + first + i) after `rng_skip` 64-byte draws.  Returns the packed ballots."""
        sel = list(selections)
        n = len(sel)
        sw = (self.n_options + 31) // 32                       # bitmask words per ballot
        words = [(int(m) >> (32 * w)) & 0xFFFFFFFF for m in sel for w in range(sw)]
        arr = (C.c_uint32 * max(len(words), 1))(*words)
        out = C.create_string_buffer(max(n * self.ballot_size, 1))
        _check(_load().eg_choice_encrypt_selected_batch(self._h, base_seed, first, n, rng_skip, arr, out))
        return out.raw[: n * self.ballot_size]

    def close(self):
        if getattr(self, "_h", None):
            _load().eg_choice_params_destroy(self._h)
            self._forget_streams()
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PublicKeyVerifier:
    """Batched ``PublicKey::verify_zero / verify_bool / verify_range`` (src/keys/impls.rs:59-69,100-112,142-151)."""

    ZERO, BOOL, RANGE = 0, 1, 2

    def __init__(self, ctx: Context, public_key: bytes, kind: int, upper_bound: int = 0):
        self.ctx, self.kind = ctx, kind
        self._h = C.c_void_p()
        _check(_load().eg_proof_params_create(ctx._h, public_key, kind, upper_bound, C.byref(self._h)))
        self.item_size = _load().eg_proof_item_size(self._h)

    def verify_batch(self, items: bytes):
        n = len(items) // self.item_size
        if n * self.item_size != len(items):
            raise ValueError("items is not a whole number of packed proofs")
        st = (C.c_uint32 * max(n, 1))()
        buf = (C.c_char * max(len(items), 1)).from_buffer_copy(items or b"\0")
        _check(_load().eg_verify_proof_batch(self._h, n, buf, st))
        return list(st[:n])

    def close(self):
        if getattr(self, "_h", None):
            _load().eg_proof_params_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SumOfSquaresVerifier(PublicKeyVerifier):
    """Batched ``SumOfSquaresProof::verify`` (src/proofs/mul.rs:190-260) with ``Transcript::new(label)``.
    item = n value ciphertexts || sum-of-squares ciphertext || challenge || 2n ciphertext responses || sum response."""

    def __init__(self, ctx: Context, public_key: bytes, n_values: int, label: bytes):
        self.ctx, self.kind = ctx, 4
        self._h = C.c_void_p()
        _check(_load().eg_sumsq_params_create(ctx._h, public_key, n_values, label, len(label), C.byref(self._h)))
        self.item_size = _load().eg_proof_item_size(self._h)


class DecryptionShareVerifier(PublicKeyVerifier):
    """Batched ``PublicKeySet::verify_share`` for one participant (src/sharing/key_set.rs:209-228).
    item = ciphertext.random_element || dh_element || challenge || response."""

    def __init__(self, ctx: Context, shared_key: bytes, shares: int, threshold: int, index: int, participant_key: bytes):
        self.ctx, self.kind = ctx, 3
        self._h = C.c_void_p()
        _check(_load().eg_share_params_create(ctx._h, shared_key, shares, threshold, index, participant_key, C.byref(self._h)))
        self.item_size = _load().eg_proof_item_size(self._h)


class QuadraticVotingParams(_BatchParams):
    """``QuadraticVotingParams::new(pk, options, credits)`` (quadratic_voting.rs:63-76)."""

    _prefix = "qv"
    kind_name = "qv"

    def __init__(self, ctx: Context, public_key: bytes, options_count: int, credits: int):
        self.ctx, self.public_key, self.n_options, self.credits = ctx, public_key, options_count, credits
        self._h = C.c_void_p()
        _check(_load().eg_qv_params_create(ctx._h, public_key, options_count, credits, C.byref(self._h)))
        self.ballot_size = _load().eg_qv_ballot_size(self._h)

    def encrypt_batch_device(self, base_seed: int, first: int, n: int, d_out: int, stream: int = 0):
        """QuadraticVotingBallot::new for n synthetic voters, written packed to device memory."""
        _check(_load().eg_qv_encrypt_batch_device(self._h, base_seed, first, n, d_out, stream))

    def encrypt_votes(self, base_seed: int, first: int, votes, rng_skip: int = 0) -> bytes:
        """``QuadraticVotingBallot::new(params, votes, rng)`` (quadratic_voting.rs:234-284) on the GPU for the caller's votes
        (one list of n_options integers per ballot); RNG as in ChoiceParams.encrypt_selected."""
        votes = [list(v) for v in votes]
        n = len(votes)
        if any(len(v) != self.n_options for v in votes):
            raise ValueError("every ballot needs n_options votes")
        flat = [x for v in votes for x in v]
        arr = (C.c_uint32 * max(len(flat), 1))(*flat)
        out = C.create_string_buffer(max(n * self.ballot_size, 1))
        _check(_load().eg_qv_encrypt_votes_batch(self._h, base_seed, first, n, rng_skip, arr, out))
        return out.raw[: n * self.ballot_size]

    def close(self):
        if getattr(self, "_h", None):
            _load().eg_qv_params_destroy(self._h)
            self._forget_streams()
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
