#!/usr/bin/env python3
"""Developer probe: the whole wire path (eg_verify_choice_json) on 1 M single-choice ballots as one JSON array, for several
host-thread counts.   usage: json_probe.py [threads ...]"""
import ctypes, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import elastic_elgamal_amd as eg
from elastic_elgamal_amd import serde

pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
p = eg.ChoiceParams(ctx, pk, 5, True)
n, distinct = 1_000_000, 1000
d = torch.empty(distinct * p.ballot_size, dtype=torch.uint8, device="cuda")
p.encrypt_batch_device(1, 0, distinct, d.data_ptr()); ctx.synchronize()
raw = bytes(d.cpu().numpy())
one = [json.dumps(serde.unpack_encrypted_choice(raw[i * p.ballot_size:(i + 1) * p.ballot_size], 5, True)) for i in range(distinct)]
text = ("[" + ",".join(one * (n // distinct)) + "]").encode()
st = (ctypes.c_uint32 * n)()
for t in [int(x) for x in sys.argv[1:]] or [8, 12, 14, 15, 16]:
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        got = p.verify_json_into(text, st, t)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"threads {t}: {n / best / 1e6:.3f} M ballots/s  ({best * 1e3:.1f} ms, {got} objects, {sum(1 for i in range(0, n, 997) if st[i] == 0)} sampled ok)")
