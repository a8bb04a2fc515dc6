#!/usr/bin/env python3
"""Developer probe: the timeline of one eg_verify_choice_json call on 1 M ballots (EG_JSON_TRACE=1: submissions and landings on stderr)."""
import ctypes as C, json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["EG_JSON_TRACE"] = "1"
import torch
import elastic_elgamal_amd as eg
from elastic_elgamal_amd import serde
m = 1_000_000
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
p = eg.ChoiceParams(ctx, pk, 5, True)
d = torch.empty(1000 * p.ballot_size, dtype=torch.uint8, device="cuda")
p.encrypt_batch_device(1, 0, 1000, d.data_ptr()); ctx.synchronize()
raw = bytes(d.cpu().numpy().tobytes())
one = [json.dumps(serde.unpack_encrypted_choice(raw[i * p.ballot_size:(i + 1) * p.ballot_size], 5, True)) for i in range(1000)]
text = ("[" + ",".join(one * (m // 1000)) + "]").encode()
st = (C.c_uint32 * m)()
for k in range(3):
    print(f"--- call {k}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); p.verify_json_into(text, st, eg.effective_cores()); dt = time.perf_counter() - t0
    print(f"--- call {k}: {dt*1e3:.1f} ms = {m/dt/1e6:.3f} M/s", file=sys.stderr, flush=True)
