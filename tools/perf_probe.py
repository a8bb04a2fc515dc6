#!/usr/bin/env python3
"""Developer probe (not the benchmark): time verify on GPU-generated ballots.
usage: perf_probe.py <n_total> <single|multi|qv> [iters]      (EG_LIB selects an alternate library build)"""
import sys, time, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import elastic_elgamal_amd as eg

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mode = sys.argv[2] if len(sys.argv) > 2 else "single"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
n_opt = int(os.environ.get("EG_PROBE_OPTIONS", "0"))      # another election size than the BASELINE configs' (single / multi)
if mode == "single":
    p = eg.ChoiceParams(ctx, pk, n_opt or 5, True); d = torch.empty(n_total * p.ballot_size, dtype=torch.uint8, device="cuda")
    p.encrypt_batch_device(1, 0, n_total, d.data_ptr()); ctx.synchronize()
elif mode == "multi":
    p = eg.ChoiceParams(ctx, pk, n_opt or 16, False); d = torch.empty(n_total * p.ballot_size, dtype=torch.uint8, device="cuda")
    p.encrypt_batch_device(1, 0, n_total, d.data_ptr(), n_selected=3); ctx.synchronize()
else:
    p = eg.QuadraticVotingParams(ctx, pk, 5, int(os.environ.get("EG_PROBE_CREDITS", "20"))); d = torch.empty(n_total * p.ballot_size, dtype=torch.uint8, device="cuda")
    p.encrypt_batch_device(1, 0, n_total, d.data_ptr()); ctx.synchronize()
st = torch.empty(n_total, dtype=torch.int32, device="cuda")
ctx.profile_enable(True)
best = 0
for it in range(iters):
    p.tally_reset()
    torch.cuda.synchronize(); t0 = time.time()
    p.verify_batch_device(n_total, d.data_ptr(), st.data_ptr())
    ctx.synchronize(); dt = time.time() - t0
    msm_ms, launches, all_ms = ctx.profile_read()
    best = max(best, n_total / dt)
    print(f"  iter {it}: {n_total/dt:,.0f} ballots/s  wall {dt*1e3:.1f} ms  msm {msm_ms:.1f} ms / {launches} launches  ok={int((st==0).sum())}")
print(f"{os.environ.get('EG_LIB','default')} {mode} n={n_total}: best {best:,.0f} ballots/s   [{ctx.name}]")
if os.environ.get("EG_PROBE_HOST"):
    # host-buffer entry point (PCIe-inclusive): pageable numpy buffer and pinned torch buffer
    import ctypes as C
    lib = eg._load()
    fn = getattr(lib, "eg_verify_qv_batch" if mode == "qv" else "eg_verify_choice_batch")
    for kind in ("pageable", "pinned"):
        h = d.cpu()
        if kind == "pinned":
            h = h.pin_memory()
        hs = torch.empty(n_total, dtype=torch.int32)
        tally = C.create_string_buffer(64 * p.n_options)
        for it in range(3):
            t0 = time.time()
            rc = fn(p._h, n_total, C.c_void_p(h.data_ptr()), C.c_void_p(hs.data_ptr()), tally)
            dt = time.time() - t0
            print(f"  host[{kind}] iter {it}: rc={rc} {n_total/dt:,.0f} ballots/s  wall {dt*1e3:.1f} ms  ok={int((hs==0).sum())}")
