"""HIP runtime settings applied at package import (attend_infer_repeat_amd/runtime_env.py): host logic, no GPU needed."""

def test_runtime_env_is_applied_on_import_and_yields_to_the_user(monkeypatch):
    """attend_infer_repeat_amd.runtime_env: the HIP runtime settings are in the environment once the package is imported, a value
    the user exported is left alone, and AIR_RUNTIME_ENV=0 switches the mechanism off."""
    import os
    from attend_infer_repeat_amd import runtime_env as R
    for k, v in R.SETTINGS.items():
        assert os.environ.get(k) is not None                       # package import ran apply()
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")      # the user's export wins
    assert R.apply()["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "1"
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    assert R.apply()["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == R.SETTINGS["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
    monkeypatch.setenv("AIR_RUNTIME_ENV", "0")
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    assert R.apply() == {} and "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ
    monkeypatch.delenv("AIR_RUNTIME_ENV")
    R.apply()


def test_runtime_env_warns_when_the_hip_runtime_is_already_up(monkeypatch):
    """the setting is read once by the HIP runtime; an import after torch initialised HIP cannot apply it and says so."""
    import pytest
    import torch
    from attend_infer_repeat_amd import runtime_env as R
    import os
    import warnings
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    # the user exported the key before the process started: it IS in effect whatever the import order -- no warning, not late
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        R.apply()
    assert R.late is False
    # the package is the one trying to set it, too late
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    with pytest.warns(RuntimeWarning, match="cannot take effect"):
        R.apply()
    assert R.late is True
    monkeypatch.undo()
    R.apply()
    assert R.late is False
