#!/usr/bin/env python3
"""Developer probe: the streaming JSON entry (eg_verify_json_begin / _feed / _end) against the one-shot entry on the same 1 M-ballot text,
by piece size and by EG_JSON_FIRST_MIN (packed ballots the stream waits for before its first GPU submission; read at params creation).
usage: json_stream_probe.py [ballots] [threads]"""
import ctypes as C, json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import elastic_elgamal_amd as eg
from elastic_elgamal_amd import serde

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
def effective_cores() -> int:
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return n
threads = int(sys.argv[2]) if len(sys.argv) > 2 else effective_cores()
pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
ctx = eg.Context(0)
p = eg.ChoiceParams(ctx, pk, 5, True)
distinct = 1000
d = torch.empty(distinct * p.ballot_size, dtype=torch.uint8, device="cuda")
p.encrypt_batch_device(1, 0, distinct, d.data_ptr()); ctx.synchronize()
raw = bytes(d.cpu().numpy().tobytes())
one = [json.dumps(serde.unpack_encrypted_choice(raw[i * p.ballot_size:(i + 1) * p.ballot_size], 5, True)) for i in range(distinct)]
text = ("[" + ",".join(one * (m // distinct)) + "]").encode()
base = C.cast(C.c_char_p(text), C.c_void_p).value
want = (C.c_uint32 * m)()
# device-resident rate of the same ballots, for the ratio
db = d.repeat(m // distinct); st = torch.empty(m, dtype=torch.int32, device="cuda")
best = 1e9
for _ in range(4):
    p.tally_reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); p.verify_batch_device(m, db.data_ptr(), st.data_ptr()); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
resident = m / best
print(f"resident: {resident/1e6:.3f} M/s ({best*1e3:.1f} ms); text {len(text)/1e9:.2f} GB, {threads} threads", flush=True)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); assert p.verify_json_into(text, want, threads) == m; best = min(best, time.perf_counter() - t0)
print(f"one-shot eg_verify_choice_json: {m/best/1e6:.3f} M/s ({best*1e3:.1f} ms) = {m/best/resident:.3f} of resident", flush=True)
p.close()
w = np.frombuffer(want, dtype=np.uint32)
for first_min in (8192, 16384, 32768, 131072):
    os.environ["EG_JSON_FIRST_MIN"] = str(first_min)
    q = eg.ChoiceParams(ctx, pk, 5, True)
    for piece in (256 << 20, 64 << 20, 8 << 20, 1 << 20):
        best = 1e9
        for _ in range(3):
            got = (C.c_uint32 * m)()
            t0 = time.perf_counter()
            s = q.json_stream(threads=threads)
            for at in range(0, len(text), piece):
                s.feed_ptr(base + at, min(piece, len(text) - at))
            n, _ = s.end_into(got)
            best = min(best, time.perf_counter() - t0)
            assert n == m and np.array_equal(np.frombuffer(got, dtype=np.uint32), w)
        print(f"stream first_min={first_min:6d} piece={piece >> 20:4d} MB: {m/best/1e6:.3f} M/s ({best*1e3:.1f} ms) = {m/best/resident:.3f} of resident", flush=True)
    q.close()
