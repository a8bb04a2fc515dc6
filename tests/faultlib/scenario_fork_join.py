"""Child process of tests/test_gpu_parity.py::test_failure_between_fork_and_join_leaves_a_usable_engine.  Runs with
EG_LIB = tests/faultlib/libeg_hip_faults.so (the library built WITH fault points): an error return between the fork onto the two work
sets' streams and the join must still tie the streams back into the caller's, and the same params object must then verify a clean batch
with the right verdicts and tally (no stale share of set 1, no kernel of the failed call still writing into the workspace or the
status buffer).  Exit code 0 = every assertion held."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import elastic_elgamal_amd as eg  # noqa: E402
from oracle import oracle  # noqa: E402

assert eg.library_path().name == "libeg_hip_faults.so", eg.library_path()
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
op = oracle.ChoiceParams(pk, 5, True)
p = eg.ChoiceParams.single_choice(ctx, pk, 5)
n = 140000                                           # more than half the resident lanes: the call forks onto both work sets
d = torch.empty(n * p.ballot_size, dtype=torch.uint8, device="cuda")
p.encrypt_batch_device(99, 0, n, d.data_ptr())
ctx.synchronize()
st = torch.full((n,), 77, dtype=torch.int32, device="cuda")
p.tally_reset()
p.verify_batch_device(n, d.data_ptr(), st.data_ptr())
ctx.synchronize()
assert int((st == 0).sum()) == n
good_tally = p.tally_encode()
sample = bytes(d[: 64 * p.ballot_size].cpu().numpy().tobytes())
assert op.verify_batch(sample) == [0] * 64
os.environ["EG_TEST_FAIL_after_fork"] = "1"         # the fault point reads it per call (test build only)
p.tally_reset()
try:
    p.verify_batch_device(n, d.data_ptr(), st.data_ptr())
except eg.EgError as e:
    assert "injected failure" in str(e), e
else:
    raise AssertionError("the fault point did not fire")
del os.environ["EG_TEST_FAIL_after_fork"]
# the caller's (null) stream was joined: work enqueued on it now runs after the failed call's kernels, so this fill wins
st.fill_(55)
torch.cuda.synchronize()
ctx.synchronize()
assert int((st == 55).sum()) == n, "kernels of the failed call wrote after the caller's stream went on"
p.tally_reset()
p.verify_batch_device(n, d.data_ptr(), st.data_ptr())
ctx.synchronize()
assert int((st == 0).sum()) == n
assert p.tally_encode() == good_tally
ballots = bytearray(sample)
ballots[5 * p.ballot_size + 100] ^= 1
st_h, t_h = p.verify_batch(bytes(ballots))           # the host form on the same object
assert st_h == op.verify_batch(bytes(ballots)) and t_h == op.tally(bytes(ballots), st_h)
# ---- a multi call (several contexts in one process) that fails in ONE slab puts EVERY running tally back (eg_hip.h "AFTER A FAILURE") ----
ctx2 = eg.Context(0)
p2 = eg.ChoiceParams.single_choice(ctx2, pk, 5)
small = 1000                                          # slab 1 is too small to fork: it succeeds and advances its tally, slab 0 fails
d2 = d[: small * p.ballot_size].clone()
st2 = torch.full((small,), 77, dtype=torch.int32, device="cuda")
for o in (p, p2):
    o.tally_reset()
t_batch = eg.verify_batch_multi_device([p, p2], [n, small], [d.data_ptr(), d2.data_ptr()], [st.data_ptr(), st2.data_ptr()], with_tally=True)
assert int((st == 0).sum()) == n and int((st2 == 0).sum()) == small
before = (p.tally_encode(), p2.tally_encode(), eg.tally_encode_multi([p, p2]))
assert t_batch == before[2] and before[0] == good_tally and before[1] == op.tally(bytes(d2.cpu().numpy().tobytes()), [0] * small)
os.environ["EG_TEST_FAIL_after_fork"] = "1"
try:
    eg.verify_batch_multi_device([p, p2], [n, small], [d.data_ptr(), d2.data_ptr()], [st.data_ptr(), st2.data_ptr()])
except eg.EgError as e:
    assert "slab 0 of 2" in str(e) and "injected failure" in str(e) and "as it was before the call" in str(e), e
else:
    raise AssertionError("the fault point did not fire in the multi call")
assert int((st2 == 0).sum()) == small                # slab 1 did run (and had advanced its tally before the roll-back)
del os.environ["EG_TEST_FAIL_after_fork"]
assert (p.tally_encode(), p2.tally_encode(), eg.tally_encode_multi([p, p2])) == before, "a failed multi call moved a running tally"
t_again = eg.verify_batch_multi_device([p, p2], [n, small], [d.data_ptr(), d2.data_ptr()], [st.data_ptr(), st2.data_ptr()], with_tally=True)
assert t_again == t_batch                              # the retry counts the batch once
grp = eg.Ristretto(ctx)
assert eg.tally_encode_multi([p, p2]) == grp.element_add(t_batch, t_batch)[0]
print("fork/join fault scenario ok")
