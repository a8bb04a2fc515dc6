"""CPU-side checks of the product: the C-ABI library loads and exports every symbol declared in
include/eg_hip.h; the host plan builders (pure host logic) produce the reference's shapes.  No compute calls."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib_path():
    import elastic_elgamal_amd as eg

    p = eg.library_path()
    if not p.exists():
        eg.build()
    return p


def test_library_exports_every_declared_symbol(lib_path):
    import elastic_elgamal_amd as eg

    lib = C.CDLL(str(lib_path))
    names = eg.exported_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/eg_hip.h but not exported"


def test_no_oracle_in_product():
    # the product must not include / link / import anything under oracle/
    for f in (ROOT / "elastic_elgamal_amd").rglob("*"):
        if f.is_file() and f.suffix in {".py", ".hip", ".cuh", ".hpp", ".h", ".cpp"}:
            txt = f.read_text()
            assert not re.search(r'#include\s+"[^"]*oracle', txt), f
            assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
    out = subprocess.run(["ldd", str(ROOT / "elastic_elgamal_amd" / "libeg_hip.so")], capture_output=True, text=True).stdout
    assert "liboracle" not in out


def test_ballot_sizes_and_missing_gpu_is_loud(lib_path):
    import elastic_elgamal_amd as eg

    lib = eg._load()
    assert lib.eg_choice_ballot_size(5, 1) == 736      # SURVEY 8a / BASELINE.md section 4
    assert lib.eg_choice_ballot_size(16, 0) == 2080
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(eg.EgError):
            eg.Context(0)   # no silent CPU fallback


def test_cpp_host_header_compiles(tmp_path, lib_path):
    # the C++ mirror of the reference interface and the voting example build against the C ABI without a GPU
    root = ROOT
    exe = tmp_path / "voting"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", f"-I{root / 'include'}", str(root / "examples" / "voting.cpp"),
                           f"-L{root / 'elastic_elgamal_amd'}", "-leg_hip", f"-Wl,-rpath,{root / 'elastic_elgamal_amd'}",
                           "-o", str(exe)])
    assert exe.exists()
