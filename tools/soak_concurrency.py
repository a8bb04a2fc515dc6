#!/usr/bin/env python3
"""Developer tool: the concurrency tests of round 6 (shared per-lane workspace, held multi-GPU calls, the multi-lane JSON stream, handed-over
blocks) several times over in ONE process, to shake out races that one pass does not show.   usage: tools/soak_concurrency.py [rounds]"""
import sys
from pathlib import Path

import pytest

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
root = Path(__file__).resolve().parent.parent
expr = "two_params_objects or multi_calls_hold or json_entries_over or feed_owned or concurrent_host_calls"
bad = 0
for k in range(rounds):
    rc = pytest.main([str(root / "tests" / "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", expr, "-p", "no:cacheprovider"])
    print(f"soak round {k + 1} of {rounds}: rc {int(rc)}", flush=True)
    bad += int(rc) != 0
    if bad:
        break
raise SystemExit(1 if bad else 0)
