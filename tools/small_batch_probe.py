#!/usr/bin/env python3
"""Developer probe: latency of small device-resident batches (eg_verify_choice_batch_device + synchronize), single-choice 5 options."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import elastic_elgamal_amd as eg

pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
p = eg.ChoiceParams(ctx, pk, 5, True)
N = 1 << 16
d = torch.empty(N * p.ballot_size, dtype=torch.uint8, device="cuda")
p.encrypt_batch_device(1, 0, N, d.data_ptr()); ctx.synchronize()
st = torch.empty(N, dtype=torch.int32, device="cuda")
for n in (1, 64, 256, 1024, 4096, 16384, 65536):
    for _ in range(3):
        p.verify_batch_device(n, d.data_ptr(), st.data_ptr()); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        p.verify_batch_device(n, d.data_ptr(), st.data_ptr()); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        p.verify_batch_device(n, d.data_ptr(), st.data_ptr())
    enq = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    print(f"n = {n:6d}: {dt * 1e3:7.3f} ms per call ({n / dt / 1e6:6.3f} M ballots/s), host enqueue {enq * 1e3:6.3f} ms, ok {int((st[:n] == 0).sum())}", flush=True)
