#!/usr/bin/env python3
"""Developer probe: the bucket path of one n-term vartime_multi_mul (run under rocprofv3 --kernel-trace --stats for the per-kernel split).
usage: msm_bucket_probe.py <log2 terms> [reps]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["EG_MSM_BUCKET_MIN"] = "4096"
import torch
import elastic_elgamal_amd as eg

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = eg.Context(0)
grp = eg.Ristretto(ctx)
g = torch.Generator(device="cpu"); g.manual_seed(1)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sc[:, 31] &= 0x0f
sc = sc.cuda()
base = torch.frombuffer(bytearray(grp.mul_generator(bytes(sc[:4096].cpu().numpy().tobytes()))), dtype=torch.uint8).cuda()
pts = base.repeat(n // 4096)
out = torch.empty(32, dtype=torch.uint8, device="cuda")
scratch = torch.empty(max(grp.msm_scratch_bytes(1, n), 16), dtype=torch.uint8, device="cuda")
for _ in range(2):
    grp.vartime_multi_mul_device(1, n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), 0, scratch.data_ptr()); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    grp.vartime_multi_mul_device(1, n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), 0, scratch.data_ptr())
torch.cuda.synchronize()
print(f"{n} terms, buckets: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms, scratch {scratch.numel() / 2**20:.0f} MiB")
