"""Rank script of tests/test_gpu_multirank.py::test_bench_multi_rank_failures_are_loud_and_early: bench.py's main() with the tally
all-gather of rank 1 handing back one wrong byte - what a broken transport would do.  The fault is installed HERE, by the test's own
rank script (a wrapper around elastic_elgamal_amd.distributed.gather_tallies); the shipped module has no such switch."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from elastic_elgamal_amd import distributed as egd  # noqa: E402

_real = egd.gather_tallies


def _corrupting(local):
    out = _real(local)
    if dist.is_initialized() and dist.get_rank() == 1:
        out = out.clone()
        out.view(-1)[0] ^= 0x5A
    return out


egd.gather_tallies = _corrupting

if __name__ == "__main__":
    bench.main()
