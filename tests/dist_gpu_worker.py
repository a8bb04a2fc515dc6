"""One rank of tests/test_gpu_parity.py::test_two_ranks_real_gpu_tallies: a FRESH process (it initialises the GPU itself) that joins a
gloo group, verifies its shard_range slab of a GPU-made batch with the HIP engine, exchanges tally encodings with the other rank
through elastic_elgamal_amd.distributed.gather_tallies and merges them on the device with eg_points_sum_device.
(examples/voting.rs:199-203 of the reference is the loop this stands in for.)
usage: dist_gpu_worker.py RANK WORLD PORT TOTAL SEED OUT.json"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


def tamper(view, first, n):
    """flip one response bit of every ballot whose GLOBAL index is 37 mod 100 (1 % of the batch, the same ballots however it is cut)"""
    idx = torch.arange(first, first + n, device=view.device)
    rows = torch.nonzero(idx % 100 == 37).flatten()
    view[rows, view.shape[1] - 32] ^= 1
    return int(rows.numel())


def main():
    rank, world, port, total, seed = (int(x) for x in sys.argv[1:6])
    out = Path(sys.argv[6])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import elastic_elgamal_amd as eg
    from elastic_elgamal_amd import distributed as egd

    pk = bytes.fromhex("a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531")
    ctx = eg.Context(0)
    p = eg.ChoiceParams.single_choice(ctx, pk, 5)
    lo, hi = egd.shard_range(total, rank, world)
    n = hi - lo
    d = torch.empty(n * p.ballot_size, dtype=torch.uint8, device="cuda")
    p.encrypt_batch_device(seed, lo, n, d.data_ptr())             # ballot i of the batch is the same whichever rank makes it
    ctx.synchronize()
    n_bad = tamper(d.view(n, p.ballot_size), lo, n)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    p.tally_reset()
    p.verify_batch_device(n, d.data_ptr(), st.data_ptr())          # EncryptedChoice::verify x n + tally, on the GPU
    ctx.synchronize()
    accepted_local = int((st == 0).sum())
    local = torch.frombuffer(bytearray(p.tally_encode()), dtype=torch.uint8)
    gathered = egd.gather_tallies(local)                            # the ONE exchange (gloo here, RCCL in bench.py)
    assert gathered.shape == (world, 64 * 5)
    g = gathered.cuda()
    merged = torch.empty(64 * 5, dtype=torch.uint8, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.points_sum_device(world, 10, g.data_ptr(), merged.data_ptr(), d_bad=bad.data_ptr())
    ctx.synchronize()
    accepted = egd.sum_over_ranks(accepted_local, "cpu")
    out.write_text(json.dumps({"rank": rank, "range": [lo, hi], "merged": bytes(merged.cpu().numpy()).hex(), "accepted": accepted,
                               "accepted_local": accepted_local, "tampered_local": n_bad, "d_bad": int(bad.item()),
                               "local": bytes(local.numpy()).hex()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
