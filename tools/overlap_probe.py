#!/usr/bin/env python3
"""Developer probe: one batch on one stream against its two halves on two streams (two contexts, so that nothing is shared), to
see whether a second stream fills the launch tails of the first.   usage: overlap_probe.py [n_total] [single|qv]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import elastic_elgamal_amd as eg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mode = sys.argv[2] if len(sys.argv) > 2 else "single"
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
mk = (lambda c: eg.ChoiceParams(c, pk, 5, True)) if mode == "single" else (lambda c: eg.QuadraticVotingParams(c, pk, 5, 20))
c0, c1, c2 = eg.Context(0), eg.Context(0), eg.Context(0)
p0, p1, p2 = mk(c0), mk(c1), mk(c2)
d = torch.empty(n * p0.ballot_size, dtype=torch.uint8, device="cuda")
p0.encrypt_batch_device(1, 0, n, d.data_ptr()); c0.synchronize()
st = torch.empty(n, dtype=torch.int32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = n // 2 // 256 * 256
def whole():
    p0.verify_batch_device(n, d.data_ptr(), st.data_ptr(), s1.cuda_stream)
def halves():
    p1.verify_batch_device(h, d.data_ptr(), st.data_ptr(), s1.cuda_stream)
    p2.verify_batch_device(n - h, d.data_ptr() + h * p0.ballot_size, st.data_ptr() + 4 * h, s2.cuda_stream)
for name, fn in (("one stream", whole), ("two streams", halves), ("one stream", whole), ("two streams", halves)):
    best = 0
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t0
        best = max(best, n / dt)
    print(f"{name}: best {best:,.0f} ballots/s  accepted {int((st == 0).sum())}")
