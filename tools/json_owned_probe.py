#!/usr/bin/env python3
"""Developer probe: 1 M ballots as JSON text through eg_verify_json_feed_owned in pieces of several sizes; time of the feeding loop and of
the whole stream, against the one-shot entry.  EG_JSON_TRACE=1 in the environment adds the timeline of submissions on stderr."""
import ctypes as C, json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
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
base = C.cast(C.c_char_p(text), C.c_void_p).value
cores = eg.effective_cores()
for _ in range(2):
    t0 = time.perf_counter(); p.verify_json_into(text, st, cores); dt = time.perf_counter() - t0
print(f"one-shot: {dt*1e3:.1f} ms = {m/dt/1e6:.3f} M/s")
for piece in (1 << 20, 1 << 22, 1 << 24, 1 << 26):
    for mode in ("owned", "copy"):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            js = p.json_stream(threads=cores)
            f = js.feed_owned_ptr if mode == "owned" else js.feed_ptr
            for at in range(0, len(text), piece):
                f(base + at, min(piece, len(text) - at))
            t1 = time.perf_counter()
            js.end_into(st)
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0)
        print(f"pieces of {piece >> 20:3d} MB {mode:5s}: feeding loop {best[1]*1e3:7.1f} ms, whole stream {best[0]*1e3:7.1f} ms = {m/best[0]/1e6:.3f} M/s")
