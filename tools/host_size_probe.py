#!/usr/bin/env python3
"""Developer probe: throughput of the host-pointer entry (eg_verify_choice_batch: pinned buffer in, verdicts out) by call size."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import elastic_elgamal_amd as eg

pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
p = eg.ChoiceParams(ctx, pk, 5, True)
N = 1 << 20
d = torch.empty(N * p.ballot_size, dtype=torch.uint8, device="cuda")
p.encrypt_batch_device(1, 0, N, d.data_ptr()); ctx.synchronize()
host = torch.empty(N * p.ballot_size, dtype=torch.uint8, pin_memory=True); host.copy_(d)
st = torch.empty(N, dtype=torch.int32, pin_memory=True)
p.verify_batch_host_ptr(N, host.data_ptr(), st.data_ptr()) if hasattr(p, "verify_batch_host_ptr") else None
for n in (16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    best = 0
    for it in range(4):
        t0 = time.perf_counter()
        for k in range(0, N, n):
            p.verify_batch_host_ptr(n, host.data_ptr() + k * p.ballot_size, st.data_ptr() + 4 * k)
        dt = time.perf_counter() - t0
        best = max(best, N / dt)
    print(f"call size {n:8d}: {best/1e6:.3f} M ballots/s  ({n / best * 1e3:.2f} ms per call)", flush=True)
