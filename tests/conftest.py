import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    return json.loads((ROOT / "tests" / "golden" / "snapshots_ristretto.json").read_text())


@pytest.fixture(scope="session")
def rejections():
    """The reference's own REJECTING inputs for Ristretto (src/serde.rs unit tests), copied by tests/golden/make_golden.py."""
    import json

    return json.loads((ROOT / "tests" / "golden" / "rejections_ristretto.json").read_text())


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.lib()
    return o


@pytest.fixture(scope="session", autouse=True)
def _product_library_built():
    """The in-tree libeg_hip.so normally travels with the snapshot; build it (hipcc, no GPU needed) if it is missing."""
    import elastic_elgamal_amd as eg

    if not eg.library_path().exists():
        eg.build()
