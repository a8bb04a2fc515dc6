#!/usr/bin/env python3
"""Developer tool (one-off soak, beyond the committed tests): differential fuzz of the HIP path against the oracle with FRESH seeds and many
election shapes - ballots generated on the GPU, 1-3 random bit flips anywhere in nine ballots out of ten, every status word and the tally
compared.   usage: tools/fuzz_soak.py [seed] [ballots per shape]"""
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import elastic_elgamal_amd as eg
from oracle import oracle

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else int(time.time())
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
shapes = [("single", 5, 0), ("single", 2, 0), ("single", 9, 0), ("multi", 16, 3), ("multi", 6, 2), ("multi", 3, 1), ("qv", 5, 20), ("qv", 4, 12), ("qv", 3, 50), ("qv", 2, 7)]
bad = 0
for k, (kind, opts, extra) in enumerate(shapes):
    m = n if kind != "qv" else n // 3
    if kind == "qv":
        p, op, kw = eg.QuadraticVotingParams(ctx, pk, opts, extra), oracle.QvParams(pk, opts, extra), {}
    else:
        p, op = eg.ChoiceParams(ctx, pk, opts, kind == "single"), oracle.ChoiceParams(pk, opts, kind == "single")
        kw = {"n_selected": extra} if kind == "multi" else {}
    sz = p.ballot_size
    d = torch.empty(m * sz, dtype=torch.uint8, device="cuda")
    p.encrypt_batch_device(seed0 + k, 0, m, d.data_ptr(), **kw)
    ctx.synchronize()
    ballots = bytearray(d.cpu().numpy().tobytes())
    rnd = random.Random(seed0 * 31 + k)
    for b in range(m):
        if b % 10 == 0:
            continue
        for _ in range(rnd.randrange(1, 4)):
            ballots[b * sz + rnd.randrange(sz)] ^= 1 << rnd.randrange(8)
    ballots = bytes(ballots)
    t0 = time.time()
    want = op.verify_batch(ballots, threads=16)
    t1 = time.time()
    got, tally = p.verify_batch(ballots)
    same = got == want and tally == op.tally(ballots, want)
    kinds = sorted({w & 0xFF for w in want})
    print(f"{kind:6s} options {opts:2d} extra {extra:2d}: {m} ballots, {want.count(0)} accepted, kinds {kinds}, oracle {t1 - t0:.1f} s: {'IDENTICAL' if same else 'MISMATCH'}", flush=True)
    if not same:
        bad += 1
        diff = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:5]
        print("   first differences (index, gpu, oracle):", diff, flush=True)
    p.close()
print(f"seed {seed0}: {'all shapes identical' if not bad else str(bad) + ' shapes differ'}")
raise SystemExit(1 if bad else 0)
