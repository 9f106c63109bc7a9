"""Host-side data pipeline (reference: data/data.py:19-158): multi-digit synthesis from templates, nums encoding, idx reader."""
import gzip
import struct

import numpy as np
import pytest

from attend_infer_repeat_amd import data as D


def _templates(n=40, seed=0):
    """fake 'digits': 28x28 with a random filled box strictly inside (so the tight box is known)"""
    rng = np.random.default_rng(seed)
    t = np.zeros((n, 28, 28), np.uint8)
    boxes = []
    for i in range(n):
        h, w = rng.integers(6, 20, 2)
        y, x = rng.integers(1, 28 - h), rng.integers(1, 28 - w)
        t[i, y:y + h, x:x + w] = rng.integers(1, 256, (h, w))
        boxes.append(((y, x), (h, w)))
    return t, boxes


def test_tight_box_matches_construction():
    t, boxes = _templates()
    for img, box in zip(t, boxes):
        assert D._tight_box(img) == ((int(box[0][0]), int(box[0][1])), (int(box[1][0]), int(box[1][1])))
    assert D._tight_box(np.zeros((28, 28))) == ((0, 0), (0, 0))


def test_create_multi_mnist_contract_and_no_overlap():
    t, boxes = _templates()
    labels = np.arange(len(t)) % 10
    d = D.create_multi_mnist(t, labels, canvas_size=(50, 50), n_objects=(0, 2), n_samples=200, seed=3)
    imgs, nums, lab = d["imgs"], d["nums"], d["labels"]
    assert imgs.shape == (200, 50, 50) and imgs.dtype == np.uint8
    assert nums.shape == (3, 200, 1) and lab.shape == (200, 2)
    counts = nums.sum(0)[:, 0]
    assert set(np.unique(counts)) <= {0, 1, 2} and len(np.unique(counts)) == 3
    assert np.all(nums[:-1] >= nums[1:])                                   # cumulative one-hot: ones first
    assert np.all(nums[-1] == 0)
    # without overlap the pasted pixel mass equals the sum of the chosen templates' masses; templates are identified by label
    # only modulo 10, so check the weaker invariant: every image's mass is the sum of some n template masses
    masses = sorted(int(x.sum()) for x in t)
    for i in range(200):
        m = int(imgs[i].astype(np.int64).sum())
        n = int(counts[i])
        if n == 0:
            assert m == 0
        else:
            assert m >= sum(masses[:n]) and m <= sum(masses[-n:])
    # empty canvases stay exactly zero (the model relies on a zero background)
    assert (imgs[counts == 0] == 0).all()


def test_create_multi_mnist_is_reproducible_and_float_templates_work():
    t, _ = _templates()
    a = D.create_multi_mnist(t, None, n_samples=50, seed=5)
    b = D.create_multi_mnist(t, None, n_samples=50, seed=5)
    assert np.array_equal(a["imgs"], b["imgs"]) and np.array_equal(a["nums"], b["nums"])
    c = D.create_multi_mnist(t.astype(np.float32) / 255.0, None, n_samples=50, seed=5)
    assert np.array_equal(a["imgs"], c["imgs"])
    d = D.create_multi_mnist(t, None, n_samples=20, seed=5, expand_nums=False)
    assert d["nums"].shape == (20,)


def test_crowded_canvas_retries_until_it_fits():
    t = np.zeros((4, 28, 28), np.uint8)
    t[:, 2:26, 2:26] = 200                                                  # 24x24 boxes: two fit a 50x50 canvas only side by side
    d = D.create_multi_mnist(t, None, canvas_size=(50, 50), n_objects=(2,), n_samples=30, seed=1)
    counts = d["nums"].sum(0)[:, 0]
    two = d["imgs"][counts == 2]
    assert (two > 0).reshape(len(two), -1).sum(1).tolist() == [2 * 24 * 24] * len(two)   # never overlapping


def test_load_mnist_idx_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    images = rng.integers(0, 256, (6000, 28, 28), dtype=np.uint8)
    labels = rng.integers(0, 10, 6000, dtype=np.uint8)
    with gzip.open(tmp_path / "train-images-idx3-ubyte.gz", "wb") as f:
        f.write(struct.pack(">IIII", 2051, 6000, 28, 28) + images.tobytes())
    with open(tmp_path / "train-labels-idx1-ubyte", "wb") as f:
        f.write(struct.pack(">II", 2049, 6000) + labels.tobytes())
    x, y = D.load_mnist_idx(str(tmp_path), "train")
    assert x.shape == (1000, 28, 28) and np.array_equal(x, images[5000:]) and np.array_equal(y, labels[5000:])
    xv, yv = D.load_mnist_idx(str(tmp_path), "validation")
    assert xv.shape == (5000, 28, 28) and np.array_equal(yv, labels[:5000])
    with pytest.raises(FileNotFoundError):
        D.load_mnist_idx(str(tmp_path), "test")


def test_create_dataset_script_writes_loadable_pickles(tmp_path):
    """scripts/create_dataset.py end to end on a fake idx directory; data.load_data reads what it wrote."""
    from attend_infer_repeat_amd.scripts import create_dataset
    t, _ = _templates(6000, seed=2)
    labels = (np.arange(6000) % 10).astype(np.uint8)
    mn = tmp_path / "MNIST_data"
    mn.mkdir()
    with open(mn / "train-images-idx3-ubyte", "wb") as f:
        f.write(struct.pack(">IIII", 2051, 6000, 28, 28) + t.tobytes())
    with open(mn / "train-labels-idx1-ubyte", "wb") as f:
        f.write(struct.pack(">II", 2049, 6000) + labels.tobytes())
    # (the real script writes 60000 / 10000 samples; patch the sizes down for the test)
    orig = create_dataset.create_multi_mnist
    create_dataset.create_multi_mnist = lambda tm, lb, n_samples, seed: orig(tm, lb, n_samples=min(n_samples, 64), seed=seed)
    try:
        create_dataset.main(["--mnist-dir", str(mn), "--out-dir", str(tmp_path)])
    finally:
        create_dataset.create_multi_mnist = orig
    d = D.load_data(str(tmp_path / "mnist_train.pickle"))
    assert d["imgs"].dtype == np.float32 and d["imgs"].shape == (64, 50, 50) and 0.0 <= d["imgs"].min() and d["imgs"].max() <= 1.0
    assert d["nums"].shape == (3, 64, 1) and d["nums"].dtype == np.float32
    assert (tmp_path / "mnist_validation.pickle").exists()


def test_procedural_digit_templates_have_mnist_like_boxes():
    """The stand-in for the MNIST download (data.py:38): ten glyph classes whose tight boxes -- what create_multi_mnist crops and
    pastes (data.py:81-92) -- look like MNIST's: about 20 pixels tall inside the 28 x 28 field, ones narrow, 8-25 % ink."""
    from attend_infer_repeat_amd import data as D
    t, labels = D.procedural_digit_templates(300, seed=4)
    assert t.shape == (300, 28, 28) and t.dtype == np.uint8 and set(np.unique(labels)) == set(range(10))
    boxes = [D._tight_box(x)[1] for x in t]
    hs, ws = np.array([b[0] for b in boxes]), np.array([b[1] for b in boxes])
    assert 15 <= hs.mean() <= 22 and hs.max() <= 26 and ws.max() <= 26
    assert ws[labels == 1].mean() < 0.6 * ws[labels == 8].mean()
    ink = (t > 0).mean()
    assert 0.08 < ink < 0.25
    assert np.array_equal(t, D.procedural_digit_templates(300, seed=4)[0])          # deterministic in the seed
    d = D.procedural_multi_mnist(64, seed=1, n_templates=100)
    assert d["imgs"].shape == (64, 50, 50) and d["nums"].shape == (3, 64, 1)
    counts = d["nums"].sum(0).reshape(-1)
    assert set(np.unique(counts)) <= {0, 1, 2} and (d["imgs"][counts == 0] == 0).all()


def test_gradient_summaries_and_attention_box():
    """evaluation.py:169-180,221-248 (global gradient norm, mean |g| / (|v| + 1e-8) per variable) and the box convention of
    evaluation.py:23-28 on hand-computable inputs."""
    import torch
    from attend_infer_repeat_amd.evaluation import attention_box, gradient_summaries
    g = {"a/w": torch.tensor([[3.0, 0.0], [0.0, 4.0]]), "b": torch.tensor([12.0])}
    v = {"a/w": torch.tensor([[1.0, 2.0], [4.0, 8.0]]), "b": torch.tensor([-3.0])}
    out = gradient_summaries(g, v)
    assert abs(out["grad_norm"] - 13.0) < 1e-9                                      # sqrt(9 + 16 + 144)
    assert abs(out["grad_ratio/a/w"] - (3.0 / 1.0 + 0 + 0 + 4.0 / 8.0) / 4) < 1e-6 and abs(out["grad_ratio/b"] - 4.0) < 1e-6
    # identity transform covers the canvas; half-size glimpse shifted right by half an extent sits in the right half
    assert attention_box([1.0, 0.0, 1.0, 0.0], 50, 40) == (0.0, 0.0, 50.0, 40.0)
    left, top, w, h = attention_box([0.5, 0.5, 0.5, 0.0], 50, 40)
    assert (left, top, w, h) == (25.0, 10.0, 25.0, 20.0)
