#!/usr/bin/env python3
"""Developer probe: one vartime_multi_mul of n terms, device-resident operands (eg_vartime_multi_mul_batch_device), by n."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import elastic_elgamal_amd as eg

ctx = eg.Context(0)
grp = eg.Ristretto(ctx)
L = 2**252 + 27742317777372353535851937790883648493
big = 1 << 22
g = torch.Generator(device="cpu"); g.manual_seed(1)
sc = torch.randint(0, 256, (big, 32), dtype=torch.uint8, generator=g); sc[:, 31] &= 0x0f
sc = sc.cuda()
# points: [x]G made on the GPU from the same scalars
pts = torch.empty(big * 32, dtype=torch.uint8, device="cuda")
grp_pts = grp.mul_generator(bytes(sc[:4096].cpu().numpy().tobytes()))
base = torch.frombuffer(bytearray(grp_pts), dtype=torch.uint8).cuda()
pts = base.repeat(big // 4096)
out = torch.empty(32, dtype=torch.uint8, device="cuda")
prep = torch.empty(big * 96, dtype=torch.uint8, device="cuda")
import os
sizes = [(n, "default") for n in (1 << 12, 1 << 14, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22)]
sizes = [(n, mode) for n, _ in sizes for mode in (("straus", "buckets") if n >= 1 << 14 else ("straus",))]
ctxs = {}
for mode in ("straus", "buckets"):        # the switch is read once, by eg_init: one context per forced path
    os.environ["EG_MSM_BUCKET_MIN"] = str(1 << 30) if mode == "straus" else "4096"
    ctxs[mode] = eg.Ristretto(eg.Context(0))
for n, mode in sizes:
    grp = ctxs[mode]
    scratch = torch.empty(max(grp.msm_scratch_bytes(1, n), 16), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        grp.vartime_multi_mul_device(1, n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), 0, scratch.data_ptr()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        grp.vartime_multi_mul_device(1, n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), 0, scratch.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    first = bytes(out.cpu().numpy())
    # the same product over PREPARED points (decoded once: eg_points_prepare_device)
    t0 = time.perf_counter()
    pok = torch.empty(n, dtype=torch.uint8, device="cuda")
    grp.prepare_points_device(n, pts.data_ptr(), prep.data_ptr(), pok.data_ptr()); torch.cuda.synchronize()
    t_prep = time.perf_counter() - t0
    for _ in range(2):
        grp.vartime_multi_mul_prepared_device(1, n, sc.data_ptr(), prep.data_ptr(), out.data_ptr(), 0, scratch.data_ptr()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        grp.vartime_multi_mul_prepared_device(1, n, sc.data_ptr(), prep.data_ptr(), out.data_ptr(), 0, scratch.data_ptr())
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t0) / 5
    same = bytes(out.cpu().numpy()) == first
    print(f"{n:8d} terms  {mode:8s}: {dt * 1e3:8.3f} ms  ({n / dt / 1e6:7.1f} M terms/s)   prepared points: {dp * 1e3:8.3f} ms ({n / dp / 1e6:7.1f} M terms/s; "
          f"prepare once {t_prep * 1e3:.3f} ms; same result: {same})   result {first.hex()[:16]}", flush=True)
