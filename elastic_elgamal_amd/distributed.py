"""Multi-GPU plumbing: one process per GPU, ballots sharded contiguously, ONE exchange per batch.

The path shards embarrassingly (ballots are independent): rank r verifies ballots
[shard_range(total, r, world)) and accumulates its own homomorphic tally (examples/voting.rs:199-203).  The only
collective is an all-gather of the per-rank tallies (n_options x 64 bytes of canonical encodings) over
RCCL/xGMI; point addition is not an RCCL reduction op, so every rank then sums the gathered encodings itself
(`Context.points_sum_device`).  With gloo the same helpers run on CPU tensors (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

# bench.py --force-dist sets this: the helpers then run their collectives even with one rank (RCCL at world size 1 on a one-GPU box)
ALWAYS_COLLECTIVE = False


def _single() -> bool:
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not ALWAYS_COLLECTIVE)


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slab [begin, end) of ballots for `rank`; slabs differ by at most one ballot."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return total * rank // world, total * (rank + 1) // world


def gather_tallies(local: torch.Tensor) -> torch.Tensor:
    """All-gather of the per-rank tally encodings.  local: uint8 [n_options*64] -> uint8 [world, n_options*64],
    rank-major (row r is rank r's tally).  A single collective per batch; no data-path collective elsewhere."""
    if _single():
        return local.reshape(1, -1).clone()
    world = dist.get_world_size()
    flat = local.contiguous().view(-1)
    out = torch.empty(world * flat.numel(), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo":      # CPU tests, and bench.py's one-GPU rehearsal of N > 1: gloo gathers host tensors
        host = flat.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        out = torch.cat(parts).to(local.device)
    else:
        dist.all_gather_into_tensor(out, flat)
    return out.view(world, flat.numel())


def max_over_ranks(seconds: float, device) -> float:
    if _single():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: int, device) -> int:
    if _single():
        return value
    t = torch.tensor([value], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_rows(row, device) -> list[list[float]]:
    """One row of numbers per rank -> every rank's rows, rank-major (bench.py's `per_rank` block: step time, clock, power, accepted ...).
    A measurement helper, called once after the timed loop; not on the data path."""
    if _single():
        return [list(map(float, row))]
    t = torch.tensor(list(map(float, row)), dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [p.cpu().tolist() for p in parts]
