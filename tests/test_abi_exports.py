"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every symbol include/air_hip.h
declares; the ctypes table matches the header (no compute calls -- no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "air_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(air_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def libpath():
    from attend_infer_repeat_amd import build
    return build.build()


def test_header_and_ctypes_table_agree():
    from attend_infer_repeat_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(libpath):
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = set(re.findall(r"\bT (air_[a-z0-9_]+)\b", out))
    missing = [s for s in _declared() if s not in exported]
    assert not missing, missing


def test_library_loads_and_reports_abi(libpath):
    from attend_infer_repeat_amd import _lib
    lib = _lib.load()
    assert lib.air_abi_version() == _lib.ABI_VERSION == 9
    from attend_infer_repeat_amd import build
    assert lib.air_build_digest().decode() == build.source_digest()
    assert lib.air_status_string(-2).decode().startswith("AIR_E_SHAPE")
    assert lib.air_gemm_workspace_bytes(64, 256, 2500) == 16 * 64 * 256 * 4


def test_argument_errors_are_reported_not_crashed(libpath):
    """NULL / bad-shape arguments return AIR_E_* before any launch (safe without a GPU)."""
    from attend_infer_repeat_amd import _lib
    lib = _lib.load()
    assert lib.air_st_read_fwd(None, None, None, 1, 1, 4, 4, 2, 2, None) == -1
    assert lib.air_lstm_pointwise_fwd(None, None, None, None, None, 4, 4, 1.0, None) == -1
    assert lib.air_gemm(0, 0, 0, 4, 4, 1, 4, 1, 4, 1, 4, None, 0, None, 0, 0.0, None, None, 0, None) == -2


def test_code_object_is_gfx950_only(libpath):
    data = open(libpath, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data))
    assert targets == {b"gfx950"}, targets


def test_product_does_not_import_oracle():
    """The HIP path must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "attend_infer_repeat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
