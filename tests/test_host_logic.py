"""Host-side logic of the engine that needs no GPU."""
import pytest


def test_flat_layout_switch_carves_disjoint_aligned_arrays(monkeypatch):
    """AIR_FLAT_LAYOUT (placement probe of round 5): the five flat arrays as one arena, packed or on staggered 2 MiB slots -- disjoint,
    zero-filled, 16-byte aligned (vectorised operand loads), the stagger as asked; unset = one allocation each."""
    import torch
    from attend_infer_repeat_amd.engine import AIREngine
    n = 2629124                                                   # configs[1]'s parameter count
    dev = torch.device("cpu")
    monkeypatch.delenv("AIR_FLAT_LAYOUT", raising=False)
    plain = AIREngine._alloc_flat(5, n, dev)
    assert len(plain) == 5 and all(t.shape == (n,) and t.dtype == torch.float32 for t in plain)
    for mode, stagger in (("packed", None), ("stagger:0", 0), ("stagger:4096", 4096), ("stagger:8448", 8448)):
        monkeypatch.setenv("AIR_FLAT_LAYOUT", mode)
        arrs = AIREngine._alloc_flat(5, n, dev)
        ptr = [t.data_ptr() for t in arrs]
        assert all(t.shape == (n,) and t.dtype == torch.float32 and not t.any() for t in arrs)
        assert all(p % 16 == 0 for p in ptr) and all(b - a >= 4 * n for a, b in zip(ptr, ptr[1:]))
        if stagger is not None:
            assert [p % (2 << 20) for p in ptr] == [k * stagger for k in range(5)]
        arrs[1].fill_(1.0)
        assert not arrs[0].any() and not arrs[2].any()
    monkeypatch.setenv("AIR_FLAT_LAYOUT", "stagger:10")
    with pytest.raises(ValueError):
        AIREngine._alloc_flat(5, n, dev)
    monkeypatch.setenv("AIR_FLAT_LAYOUT", "diagonal")
    with pytest.raises(ValueError):
        AIREngine._alloc_flat(5, n, dev)


def test_gradient_summaries_norm_ratio_histogram():
    """evaluation.py:221-248 on plain tensors: global norm, per-variable mean |g| / (|v| + 1e-8), and the histogram's content."""
    import torch
    from attend_infer_repeat_amd.evaluation import gradient_summaries
    g = {"a/w": torch.tensor([[3.0, -4.0], [0.0, 0.0]]), "b": torch.tensor([1.0, float("nan"), 2.0])}
    v = {"a/w": torch.ones(2, 2), "b": torch.full((3,), 2.0)}
    out = gradient_summaries({"a/w": g["a/w"]}, v, histogram=True, bins=4)
    assert abs(out["grad_norm"] - 5.0) < 1e-12 and abs(out["grad_ratio/a/w"] - 7.0 / 4) < 1e-6
    h = out["grad_hist/a/w"]
    assert sum(h["counts"]) == 4 and len(h["edges"]) == 5 and h["edges"][0] == -4.0 and h["edges"][-1] == 3.0 and h["non_finite"] == 0
    assert h["counts"][0] == 1 and h["counts"][-1] == 1 and h["counts"][2] == 2          # -4 | (0, 0) | 3
    hb = gradient_summaries({"b": g["b"]}, v, norm=False, ratio=False, histogram=True, bins=2)["grad_hist/b"]
    assert hb["non_finite"] == 1 and sum(hb["counts"]) == 2
    assert "grad_hist/a/w" not in gradient_summaries({"a/w": g["a/w"]}, v)               # off by default (JSON lines, not event files)


def test_roctx_ranges_are_opt_in(monkeypatch):
    """AIR_ROCTX=1 wraps every eagerly issued plan entry in a roctx range (engine._run); off by default, and a missing library is not an error."""
    import attend_infer_repeat_amd.engine as E
    monkeypatch.setattr(E, "_ROCTX", [False, None])
    monkeypatch.delenv("AIR_ROCTX", raising=False)
    assert E._roctx() is None
    monkeypatch.setattr(E, "_ROCTX", [False, None])
    monkeypatch.setenv("AIR_ROCTX", "1")
    lib = E._roctx()
    if lib is not None:                          # (the ROCm image ships libroctx64; nested depth is what push returns)
        d0 = lib.roctxRangePushA(b"00 air_test"); d1 = lib.roctxRangePushA(b"01 air_test")
        assert d1 == d0 + 1
        lib.roctxRangePop(); lib.roctxRangePop()
    calls = []
    eng = E.AIREngine.__new__(E.AIREngine)
    E.AIREngine._run(eng, [(lambda a, sp: calls.append((a, sp)) or 0, (7,), "air_fake")], "S")
    assert calls == [(7, "S")]

