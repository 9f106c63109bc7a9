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
