#!/usr/bin/env python3
"""Developer probe: what do memory-side bytes cost in watts on this chip?  Samples the package power (hwmon, as bench.py's ClockSampler)
while the GPU idles, while it streams device-to-device copies at full HBM speed, and at two throttled rates (the same copies with gaps),
and prints watts per TB/s of memory-side traffic (reads + writes).  The verifier moves ~1.6 TB/s of table gathers and stores.
usage: hbm_power_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

dev = torch.device("cuda", 0)
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
a.fill_(1); torch.cuda.synchronize()


def run(label, seconds, duty):
    s = bench.ClockSampler(torch, 0); s.start()
    t0 = time.perf_counter(); moved = 0
    while time.perf_counter() - t0 < seconds:
        if duty > 0:
            t1 = time.perf_counter()
            b.copy_(a); torch.cuda.synchronize(); moved += 2 * n          # n bytes read + n bytes written
            busy = time.perf_counter() - t1
            if duty < 1: time.sleep(busy * (1 - duty) / duty)
        else:
            time.sleep(0.05)
    dt = time.perf_counter() - t0
    c = s.stop()
    print(f"{label:28s}: {moved / dt / 1e12:6.2f} TB/s memory-side (read + write)   {c['power_w']:7.1f} W   sclk {c['sclk_mhz']:.0f} MHz", flush=True)
    return moved / dt / 1e12, c["power_w"]


r0 = run("idle", 2.0, 0)
r1 = run("copies back to back", 3.0, 1.0)
r2 = run("copies, 50 % duty", 3.0, 0.5)
r3 = run("copies, 25 % duty", 3.0, 0.25)
for name, r in (("full", r1), ("50 %", r2), ("25 %", r3)):
    print(f"  {name}: ({r[1]:.0f} - {r0[1]:.0f}) W / {r[0]:.2f} TB/s = {(r[1] - r0[1]) / max(r[0], 1e-9):.0f} W per TB/s above idle")
