#!/usr/bin/env python3
"""Developer probe: chunk workspace per ballot and work set, measured (free device memory before / after a batch of two full chunks)."""
import os, sys
from pathlib import Path
os.environ["EG_COMB_BIG_BITS"] = "0"          # keep the wide comb tables out of the measurement
os.environ["EG_CHUNK"] = "262144"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import elastic_elgamal_amd as eg

pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
n = 2 * 262144
for name, make, kw in (("single-5", lambda: eg.ChoiceParams(ctx, pk, 5, True), {}), ("multi-16", lambda: eg.ChoiceParams(ctx, pk, 16, False), {"n_selected": 3}),
                       ("qv-5-20", lambda: eg.QuadraticVotingParams(ctx, pk, 5, 20), {})):
    p = make()
    d = torch.empty(n * p.ballot_size, dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    p.encrypt_batch_device(1, 0, n, d.data_ptr(), **kw); ctx.synchronize()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    p.verify_batch_device(n, d.data_ptr(), st.data_ptr()); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    print(f"{name}: {(free0 - free1) / n:.0f} bytes of chunk workspace per ballot and work set ({(free0 - free1) / 2**30:.1f} GiB for two sets of 262144), accepted {int((st == 0).sum())}")
    del p, d, st
