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


def test_stable_contract_did_not_change():
    """include/air_hip.h marks every declaration AIR_API (the stable contract: SURVEY 8(b)'s operator boundary + plumbing) or AIR_ENGINE_API
    (engine plan entries that change with every fold).  The stable prototypes are pinned: a change here needs a new AIR_ABI_VERSION,
    a new tests/golden/abi_stable.txt and a line in INTEGRATION.md's change log."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = re.findall(r"AIR_API\s+((?:int|const char \*)\s*air_[a-z0-9_]+\s*\([^;]*\));", src, flags=re.S)
    have = sorted(re.sub(r"\s+", " ", x).strip() for x in protos)
    want = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "abi_stable.txt")) if l.strip() and not l.startswith("#")]
    assert have == want
    # every declaration carries exactly one of the two marks, and SURVEY 8(b)'s nine exports are on the stable side
    unmarked = re.findall(r"^(?:int|const char \*) ?air_[a-z0-9_]+\(", src, flags=re.M)
    assert not unmarked, unmarked
    names = {re.search(r"(air_[a-z0-9_]+)\s*\(", x).group(1) for x in have}
    for n in ("air_st_read_fwd", "air_st_read_bwd", "air_st_write_fwd", "air_st_write_bwd", "air_linear_fwd", "air_linear_bwd",
              "air_lstm_pointwise_fwd", "air_lstm_pointwise_bwd", "air_gauss_sample_fwd", "air_gauss_sample_bwd", "air_rec_loglik_fwd",
              "air_rec_loglik_bwd", "air_numsteps_fwd", "air_numsteps_bwd", "air_nvil", "air_rmsprop_centered", "air_canvas_unroll_fwd",
              "air_canvas_unroll_bwd", "air_graph_launch", "air_allreduce_sum"):
        assert n in names, n


def test_library_exports_every_declared_symbol(libpath):
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = set(re.findall(r"\bT (air_[a-z0-9_]+)\b", out))
    missing = [s for s in _declared() if s not in exported]
    assert not missing, missing


def test_library_loads_and_reports_abi(libpath):
    from attend_infer_repeat_amd import _lib
    lib = _lib.load()
    assert lib.air_abi_version() == _lib.ABI_VERSION == 10 and lib.air_engine_abi_version() == _lib.ENGINE_ABI_VERSION == 5
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
