#!/usr/bin/env python3
"""Developer probe (not the benchmark): time verify on oracle-generated ballots tiled in HBM."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import base64, torch
import elastic_elgamal_amd as eg
from oracle import oracle as o

n_unique = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_total = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
mode = sys.argv[3] if len(sys.argv) > 3 else "single"
sk, pk, _ = o.keypair_from_seed(12345)
ctx = eg.Context(0)
print(ctx.name)
if mode == "single":
    op = o.ChoiceParams(pk, 5, True); p = eg.ChoiceParams(ctx, pk, 5, True); uniq = op.generate_batch(1, 0, n_unique)
elif mode == "multi":
    op = o.ChoiceParams(pk, 16, False); p = eg.ChoiceParams(ctx, pk, 16, False); uniq = op.generate_batch(1, 0, n_unique, n_selected=3)
else:
    op = o.QvParams(pk, 5, 20); p = eg.QuadraticVotingParams(ctx, pk, 5, 20); uniq = op.generate_batch(1, 0, n_unique)
reps = (n_total + n_unique - 1) // n_unique
host = torch.frombuffer(bytearray(uniq), dtype=torch.uint8).repeat(reps)[: n_total * p.ballot_size]
d = host.cuda()
st = torch.empty(n_total, dtype=torch.int32, device="cuda")
ctx.profile_enable(True)
for it in range(3):
    p.tally_reset()
    torch.cuda.synchronize(); t0 = time.time()
    p.verify_batch_device(n_total, d.data_ptr(), st.data_ptr())
    ctx.synchronize(); dt = time.time() - t0
    msm_ms, launches, all_ms = ctx.profile_read()
    print(f"iter {it}: {n_total/dt:,.0f} ballots/s  wall {dt*1e3:.1f} ms  msm {msm_ms:.1f} ms over {launches} launches  all {all_ms:.1f} ms  ok={int((st==0).sum())}")
