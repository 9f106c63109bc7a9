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
    assert D._tight_box(np.zeros((28, 28))) == ((1, 1), (0, 0))             # data.py:19-23 on an empty projection: 0 - 0 + 1


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
    # float templates go through scipy.misc.bytescale's arithmetic (data.py:55): [min, max] -> [0, 255]; MNIST's floats are
    # uint8 / 255 with min 0 and (almost always) max 1, which maps back to the original bytes
    t2 = t.copy(); t2[:, 14, 14] = 255
    a2 = D.create_multi_mnist(t2, None, n_samples=50, seed=5)
    c = D.create_multi_mnist(t2.astype(np.float32) / 255.0, None, n_samples=50, seed=5)
    assert np.array_equal(a2["imgs"], c["imgs"])
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


# ---- f2: the generator against the restatement of data.py:35-107 (oracle/data_oracle.py), element for element ----------------
def _gappy_templates(n=60, seed=9):
    """procedural digits plus the cases the reference's box arithmetic treats specially: supports with an empty row / column
    inside (dim_coords counts non-empty rows), a single pixel, a full-field template, an empty one"""
    t, labels = D.procedural_digit_templates(n, seed=seed)
    t = t.copy()
    t[0] = 0; t[0, 5:9, 6:20] = 200; t[0, 15:22, 8:12] = 90          # two blobs, rows 9..14 empty
    t[1] = 0; t[1, 4:24, 3:8] = 255; t[1, 6:20, 16:21] = 17          # columns 8..15 empty
    t[2] = 0; t[2, 13, 13] = 1
    t[3] = 131
    return t, labels


@pytest.mark.parametrize("n_objects,canvas,seed,overlap", [((0, 2), (50, 50), 0, False), ((0, 2), (50, 50), 1, False),
                                                           ((1, 2), (40, 64), 2, False), ((2,), (34, 34), 3, False),
                                                           ((0, 1), (28, 28), 4, True), ((0, 2), (50, 50), 5, True)])
def test_create_multi_mnist_equals_the_reference_generator_draw_for_draw(n_objects, canvas, seed, overlap):
    """Same templates, same generator state => the same canvases, labels and counts as data.py:35-107, including its quirks (one
    object: both coordinates from one draw; five tries per SAMPLE; count-based boxes) -- and the SAME generator state afterwards,
    i.e. not one draw more or less."""
    from oracle import data_oracle as DO
    t, labels = _gappy_templates()
    if canvas == (28, 28):
        t = t[4:]; labels = labels[4:]                     # (keep the full-field template out of a canvas of its own size)
    ra, rb = np.random.RandomState(100 + seed), np.random.RandomState(100 + seed)
    want = DO.create_mnist(t, labels, ra, canvas_size=canvas, n_objects=n_objects, n_samples=300, with_overlap=overlap)
    got = D.create_multi_mnist(t, labels, canvas_size=canvas, n_objects=n_objects, n_samples=300, with_overlap=overlap, rng=rb)
    for k in ("imgs", "labels", "nums"):
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
    assert ra.randint(1 << 30) == rb.randint(1 << 30)
    # seed= is RandomState(seed): the reference's `np.random.seed(seed)` before running its script
    again = D.create_multi_mnist(t, labels, canvas_size=canvas, n_objects=n_objects, n_samples=300, with_overlap=overlap, seed=100 + seed)
    assert np.array_equal(again["imgs"], want["imgs"])


def test_single_objects_lie_on_the_diagonal_like_the_reference():
    """data.py:58-60: make_p draws rand(n) with n = the sample's object count, so one object gets ONE number for both coordinates."""
    t, labels = D.procedural_digit_templates(50, seed=1)
    d = D.create_multi_mnist(t, labels, n_objects=(1,), n_samples=100, seed=7)
    assert d["nums"].sum() > 30
    for img in d["imgs"]:
        ys, xs = np.nonzero(img.sum(1))[0], np.nonzero(img.sum(0))[0]
        if ys.size == 0:                                    # randint(max + 1): counts 0 or 1
            continue
        (y0, h), (x0, w) = (ys[0], ys[-1] - ys[0] + 1), (xs[0], xs[-1] - xs[0] + 1)
        u_y, u_x = y0 / max(50 - h, 1), x0 / max(50 - w, 1)
        assert abs(u_y - u_x) <= 0.5 / max(50 - h, 1) + 0.5 / max(50 - w, 1) + 1e-9


def test_multi_mnist_fixture():
    """tests/golden/multi_mnist_case.npz (written by tests/golden/make_data_golden.py from oracle/data_oracle.py): templates in,
    dataset out.  Both the oracle and the product reproduce it."""
    import os
    from oracle import data_oracle as DO
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "multi_mnist_case.npz"))
    for make in (lambda: DO.create_mnist(z["templates"], z["labels"], np.random.RandomState(int(z["seed"])), n_samples=int(z["n_samples"])),
                 lambda: D.create_multi_mnist(z["templates"], z["labels"], n_samples=int(z["n_samples"]), seed=int(z["seed"]))):
        d = make()
        assert np.array_equal(d["imgs"], z["imgs"]) and np.array_equal(d["labels"], z["out_labels"]) and np.array_equal(d["nums"], z["nums"])


def test_more_than_two_objects_is_an_extension():
    """The reference cannot broadcast rand(3) against two free ranges (ValueError in make_p); the product places 3-4 digits
    (BASELINE configs[3]) with two draws per try."""
    from oracle import data_oracle as DO
    t, labels = D.procedural_digit_templates(40, seed=2)
    with pytest.raises(ValueError):
        DO.create_mnist(t, labels, np.random.RandomState(0), canvas_size=(100, 100), n_objects=(3,), n_samples=4)
    d = D.create_multi_mnist(t, labels, canvas_size=(100, 100), n_objects=(0, 4), n_samples=60, seed=0)
    assert d["nums"].shape == (5, 60, 1) and set(np.unique(d["nums"].sum(0))) == {0, 1, 2, 3, 4}


# ---- f3: content of the progress figure (evaluation.py:31-65) ------------------------------------------------------------------
def check_progress_figure(fig, obs, canvas, glimpse, presence, where, step_probs, n_cols):
    """Every panel of the reference's figure, read back from the matplotlib objects: row 0 the inputs, rows 1..T the canvas after
    each step with ONE red rectangle -- the box of evaluation.py:23-28 -- exactly where the step is present, rows T+1..2T the
    glimpses titled '<presence> with p(<t+1>) = <prob>'."""
    from matplotlib.patches import Rectangle
    from attend_infer_repeat_amd.evaluation import attention_box
    T = canvas.shape[0]
    H, W = obs.shape[1:]
    axes = np.array(fig.axes).reshape(2 * T + 1, n_cols)
    n_boxes = 0
    for col in range(n_cols):
        assert np.array_equal(axes[0, col].images[0].get_array(), obs[col])
        for t in range(T):
            ax = axes[1 + t, col]
            assert np.array_equal(ax.images[0].get_array(), canvas[t, col]) and ax.images[0].get_clim() == (0, 1)
            boxes = [p for p in ax.patches if isinstance(p, Rectangle)]
            if presence[t, col] > .5:
                assert len(boxes) == 1
                left, top, bw, bh = attention_box(where[t, col], W, H)
                r = boxes[0]
                assert np.allclose([r.get_x(), r.get_y(), r.get_width(), r.get_height()], [left - .5, top - .5, bw, bh], atol=1e-4)
                assert r.get_edgecolor()[:3] == (1.0, 0.0, 0.0) and r.get_facecolor()[3] == 0.0
                # the same box in the reference's own variables (evaluation.py:23-28: bbox = [y - .5, x - .5, height * sy, width * sx])
                sx, tx, sy, ty = (float(v) for v in where[t, col])
                assert np.allclose([r.get_x(), r.get_y()], [W * (1. - sx + tx) / 2 - .5, H * (1. - sy + ty) / 2 - .5], atol=1e-4)
                n_boxes += 1
            else:
                assert len(boxes) == 0
            gax = axes[1 + T + t, col]
            assert np.array_equal(gax.images[0].get_array(), glimpse[t, col])
            assert gax.get_title() == '{:d} with p({:d}) = {:.02f}'.format(int(presence[t, col]), t + 1, float(step_probs[col, t]))
    return n_boxes


def test_make_fig_content_with_a_stand_in_model():
    import torch
    from attend_infer_repeat_amd.evaluation import make_fig
    from attend_infer_repeat_amd.utils import AttrDict
    rng = np.random.default_rng(0)
    T, B, H, W, h, w = 3, 12, 50, 40, 20, 16
    f = lambda *s: torch.from_numpy(rng.random(s).astype(np.float32))
    pres = torch.from_numpy((rng.random((T, B, 1)) > 0.4).astype(np.float32))
    probs = f(B, T + 1)
    air = AttrDict(obs=f(B, H, W), canvas=f(T, B, H, W), glimpse=f(T, B, h, w), presence=pres, where=f(T, B, 4) * 2 - 0.5,
                   max_steps=T, batch_size=B, num_steps_distrib=AttrDict(prob=lambda: probs))
    fig = make_fig(air, n_samples=10)
    n = check_progress_figure(fig, air.obs.numpy(), air.canvas.numpy(), air.glimpse.numpy(), pres.numpy()[..., 0], air.where.numpy(),
                              probs.numpy()[:, 1:], 10)
    assert n == int(pres[:, :10].sum())
    import matplotlib.pyplot as plt
    plt.close(fig)
