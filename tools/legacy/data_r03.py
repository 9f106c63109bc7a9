"""Synthetic multi-MNIST-shaped batches with the tensor contract of the reference's feeder.

The reference builds its dataset offline from an MNIST download (data/data.py:35-107) and feeds it through
tf.py_func (data.py:121-158); neither is available here (no network), so benchmarks and smoke tests use stroke
blobs with the same contract: `imgs` float32 [B,H,W] in [0,1] with an exactly-zero background and 0..max_objects
non-overlapping-ish objects, `nums` float32 [max_objects+1,B,1] one-hot-cumulative (data.py:101-105).
"""
import numpy as np


def synthetic_multi_mnist(batch, img_size=(50, 50), max_objects=2, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W = img_size
    imgs = np.zeros((batch, H, W), np.float32)
    nums = np.zeros((max_objects + 1, batch, 1), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(batch):
        n = int(rng.integers(0, max_objects + 1))
        nums[:n, b, 0] = 1.0
        for _ in range(n):
            s = max(2, int(min(H, W) * rng.uniform(0.2, 0.45)))
            cy = rng.uniform(s / 2, max(s / 2 + 1e-3, H - s / 2))
            cx = rng.uniform(s / 2, max(s / 2 + 1e-3, W - s / 2))
            ang, r = rng.uniform(0, np.pi), s * 0.4
            x0, y0 = cx - r * np.cos(ang), cy - r * np.sin(ang)
            x1, y1 = cx + r * np.cos(ang), cy + r * np.sin(ang)
            px, py, dxs, dys = xx - x0, yy - y0, x1 - x0, y1 - y0
            tt = np.clip((px * dxs + py * dys) / (dxs * dxs + dys * dys + 1e-9), 0, 1)
            d = np.sqrt((px - tt * dxs) ** 2 + (py - tt * dys) ** 2)
            imgs[b] = np.maximum(imgs[b], np.clip(1.6 - d, 0, 1))          # anti-aliased stroke
    return imgs, nums


# ---- feeders (reference: data/data.py:110-158) -------------------------------------------------------------------------
def load_data(path):
    """data.py:110-118 for Python-3 pickles: dict(imgs uint8 [N,H,W], nums [max_objects+1,N,1], ...) -> float32 arrays."""
    import pickle
    with open(path, "rb") as f:
        data = pickle.load(f)
    data["imgs"] = data["imgs"].astype(np.float32) / 255.0
    data["nums"] = data["nums"].astype(np.float32)
    return data


def synthetic_dataset(n_samples, img_size=(50, 50), max_objects=2, seed=0):
    imgs, nums = synthetic_multi_mnist(n_samples, img_size, max_objects, seed)
    return dict(imgs=imgs, nums=nums)


class DeviceFeeder(object):
    """HBM-resident counterpart of `tensors_from_data` (data.py:121-158).  The reference feeds every step through
    tf.py_func (a host round trip per step); here the whole dataset lives on the GPU and a batch is an index gather.
    shuffle=True samples with replacement like `np.random.choice(n, batch_size)` (data.py:131-132); shuffle=False walks
    the data in order (the reference's non-shuffled feeder always returns the FIRST batch -- SURVEY B-8 -- which is a bug
    we do not reproduce)."""

    def __init__(self, data, batch_size, device, shuffle=False, seed=0):
        import torch
        self.torch = torch
        self.imgs = torch.as_tensor(data["imgs"], dtype=torch.float32, device=device)
        self.nums = torch.as_tensor(data["nums"], dtype=torch.float32, device=device)      # [max_objects+1, N, 1]
        self.n = self.imgs.shape[0]
        self.batch_size, self.shuffle, self._pos = int(batch_size), shuffle, 0
        self.gen = torch.Generator(device=device).manual_seed(seed)

    @property
    def num_batches(self):
        return self.n // self.batch_size

    def __call__(self):
        torch = self.torch
        if self.shuffle:
            idx = torch.randint(0, self.n, (self.batch_size,), device=self.imgs.device, generator=self.gen)
        else:
            idx = (torch.arange(self.batch_size, device=self.imgs.device) + self._pos) % self.n
            self._pos = (self._pos + self.batch_size) % self.n
        return self.imgs.index_select(0, idx), self.nums.index_select(1, idx)

    def state_dict(self):
        """Position of the feeder (sampler state / cursor) so that a resumed run draws the batches an uninterrupted one would."""
        return {"pos": int(self._pos), "gen": self.gen.get_state().cpu()}

    def load_state_dict(self, sd):
        self._pos = int(sd["pos"])
        self.gen.set_state(sd["gen"])


# ---- multi-MNIST synthesis from digit templates (reference: data/data.py:19-107) ----------------------------------------
def _tight_box(template):
    """(y0, x0), (height, width) of the non-zero support of a template (data.py:19-32: first..last non-zero row / column)."""
    rows = np.flatnonzero(template.sum(1) > 0)
    cols = np.flatnonzero(template.sum(0) > 0)
    if rows.size == 0 or cols.size == 0:
        return (0, 0), (0, 0)
    return (int(rows[0]), int(cols[0])), (int(rows[-1] - rows[0] + 1), int(cols[-1] - cols[0] + 1))


def _resize_templates(templates, obj_size):
    if tuple(templates.shape[1:]) == tuple(obj_size):
        return templates
    import torch
    t = torch.as_tensor(templates, dtype=torch.float32)[:, None]
    t = torch.nn.functional.interpolate(t, size=tuple(obj_size), mode="bilinear", align_corners=False)
    return t[:, 0].clamp_(0, 255).round_().numpy().astype(templates.dtype)


def create_multi_mnist(templates, labels=None, canvas_size=(50, 50), obj_size=(28, 28), n_objects=(0, 2), n_samples=None,
                       dtype=np.uint8, expand_nums=True, with_overlap=False, seed=0, max_tries=5):
    """Multi-digit canvases from single-digit templates, the generator of the reference's dataset script (data.py:35-107).

    templates [N, h, w] (uint8 0..255 or float 0..1 -- MNIST digits when available; the container has no network, so the
    caller supplies them, e.g. `load_mnist_idx`), labels [N] optional.  Per sample: n ~ U{0..max(n_objects)} distinct templates,
    each cropped to the tight bounding box of its non-zero pixels and pasted at a uniformly random position where the box fits;
    without overlap a position is redrawn while the box hits an occupied box, at most `max_tries` redraws per SAMPLE, after
    which the whole sample is started again (data.py:84-97).  Returns dict(imgs [n,H,W] dtype, labels [n,max] uint8,
    nums [max+1,n,1] cumulative one-hot (data.py:101-105) or [n] counts)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    templates = np.asarray(templates)
    if templates.dtype != np.uint8 and dtype == np.uint8:
        templates = np.clip(np.round(templates * 255.0), 0, 255).astype(np.uint8)
    templates = _resize_templates(templates, obj_size)
    n_templates = templates.shape[0]
    n_samples = n_templates if n_samples is None else int(n_samples)
    max_objects = int(max(np.atleast_1d(n_objects)))
    H, W = canvas_size
    imgs = np.zeros((n_samples, H, W), dtype=dtype)
    lab = np.zeros((n_samples, max_objects), dtype=np.uint8)
    nums = rng.integers(0, max_objects + 1, size=n_samples).astype(np.uint8)
    boxes = [_tight_box(t) for t in templates]
    occupancy = np.zeros((H, W), dtype=bool)

    def position(size):
        return np.round(rng.random(2) * (np.asarray([H, W]) - np.asarray(size))).astype(np.int64)

    i = 0
    while i < n_samples:
        n, tries, retry = int(nums[i]), 0, False
        occupancy[...] = False
        idx = rng.choice(n_templates, size=n, replace=False) if n > 0 else []
        for j, k in enumerate(idx):
            (y0, x0), (sh, sw) = boxes[k]
            if sh > H or sw > W:
                raise ValueError("template box %s does not fit the canvas %s" % ((sh, sw), (H, W)))
            p = position((sh, sw))
            if not with_overlap:
                while occupancy[p[0]:p[0] + sh, p[1]:p[1] + sw].any() and tries < max_tries:
                    p = position((sh, sw))
                    tries += 1
                if tries == max_tries and occupancy[p[0]:p[0] + sh, p[1]:p[1] + sw].any():
                    retry = True
                    break
            imgs[i, p[0]:p[0] + sh, p[1]:p[1] + sw] = templates[k, y0:y0 + sh, x0:x0 + sw]
            occupancy[p[0]:p[0] + sh, p[1]:p[1] + sw] = True
            if labels is not None:
                lab[i, j] = labels[k]
        if retry:
            imgs[i] = 0
            lab[i] = 0
        else:
            i += 1
    if expand_nums:
        expanded = np.zeros((max_objects + 1, n_samples, 1), dtype=np.uint8)
        for s, n in enumerate(nums):
            expanded[:n, s] = 1
        nums = expanded
    return dict(imgs=imgs, labels=lab, nums=nums)


# ---- procedural digit templates (no network, no MNIST files in the container) ----------------------------------------------
# Seven-segment skeletons of the ten digits, drawn as anti-aliased poly-lines with per-sample slant, aspect, stroke width, corner
# rounding and endpoint jitter into the 20x20 box MNIST size-normalises its digits to, centred in a 28x28 field (so the tight
# boxes create_multi_mnist crops -- data.py:81-92 -- have MNIST-like statistics: ~20 pixels tall, 5-18 wide, ~10-20 % ink).
_SEGMENTS = {  # unit-square coordinates (x right, y down): a top, b upper right, c lower right, d bottom, e lower left, f upper left, g middle
    "a": ((0, 0), (1, 0)), "b": ((1, 0), (1, .5)), "c": ((1, .5), (1, 1)), "d": ((0, 1), (1, 1)),
    "e": ((0, .5), (0, 1)), "f": ((0, 0), (0, .5)), "g": ((0, .5), (1, .5))}
_DIGIT_SEGMENTS = ["abcdef", "bc", "abged", "abgcd", "fgbc", "afgcd", "afgedc", "abc", "abcdefg", "abfgcd"]


def procedural_digit_templates(n, seed=0, size=28):
    """n digit-like glyphs: uint8 [n, size, size] (0 background, up to 255 ink) and their labels uint8 [n]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, size, size), np.uint8)
    labels = rng.integers(0, 10, size=n).astype(np.uint8)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for i in range(n):
        d = int(labels[i])
        h = rng.uniform(14.0, 17.5)
        w = h * (rng.uniform(0.18, 0.3) if d == 1 else rng.uniform(0.45, 0.8))
        slant = rng.uniform(-0.25, 0.25)
        thick = rng.uniform(1.1, 2.0)
        cx, cy = size / 2 + rng.uniform(-1, 1), size / 2 + rng.uniform(-1, 1)
        img = np.zeros((size, size), np.float32)
        for sname in _DIGIT_SEGMENTS[d]:
            (ax, ay), (bx, by) = _SEGMENTS[sname]
            pts = []
            for (ux, uy) in ((ax, ay), (bx, by)):
                ux, uy = ux + rng.normal(0, 0.04), uy + rng.normal(0, 0.03)
                y = cy + (uy - 0.5) * h
                x = cx + (ux - 0.5) * w + slant * (0.5 - uy) * h
                pts.append((x, y))
            (x0, y0), (x1, y1) = pts
            # a bowed stroke: three-point poly-line through a displaced midpoint
            mx, my = (x0 + x1) / 2 + rng.normal(0, 0.6), (y0 + y1) / 2 + rng.normal(0, 0.6)
            for (px0, py0), (px1, py1) in (((x0, y0), (mx, my)), ((mx, my), (x1, y1))):
                dxs, dys = px1 - px0, py1 - py0
                tt = np.clip(((xx - px0) * dxs + (yy - py0) * dys) / (dxs * dxs + dys * dys + 1e-9), 0, 1)
                dist = np.sqrt((xx - px0 - tt * dxs) ** 2 + (yy - py0 - tt * dys) ** 2)
                img = np.maximum(img, np.clip(thick + 0.5 - dist, 0, 1))
        out[i] = np.clip(np.round(img * 255.0), 0, 255).astype(np.uint8)
    return out, labels


def procedural_multi_mnist(n_samples, canvas_size=(50, 50), n_objects=(0, 2), seed=0, n_templates=4000):
    """A multi-MNIST-shaped dataset in the reference's format (dict(imgs uint8, labels, nums) of create_multi_mnist, i.e. of
    data.py:35-107) from procedural digit templates: the reference's generator end to end, with only the MNIST download replaced."""
    templates, labels = procedural_digit_templates(n_templates, seed=seed)
    return create_multi_mnist(templates, labels, canvas_size=canvas_size, n_objects=n_objects, n_samples=n_samples,
                              seed=seed + 1)


def load_mnist_idx(directory, partition="train"):
    """MNIST digits from the standard idx-ubyte files (train-images-idx3-ubyte[.gz], ...), for create_multi_mnist.
    (The reference downloads them through tensorflow.examples.tutorials.mnist, data.py:38; there is no network here.)"""
    import gzip
    import os
    import struct
    stem = {"train": "train", "validation": "train", "test": "t10k"}[partition]

    def read(name):
        for cand in (name, name + ".gz"):
            path = os.path.join(directory, cand)
            if os.path.exists(path):
                opener = gzip.open if cand.endswith(".gz") else open
                with opener(path, "rb") as f:
                    return f.read()
        raise FileNotFoundError(os.path.join(directory, name))

    raw = read("%s-images-idx3-ubyte" % stem)
    magic, n, h, w = struct.unpack(">IIII", raw[:16])
    if magic != 2051:
        raise ValueError("not an idx3 image file")
    images = np.frombuffer(raw, np.uint8, offset=16).reshape(n, h, w)
    raw = read("%s-labels-idx1-ubyte" % stem)
    magic, n2 = struct.unpack(">II", raw[:8])
    if magic != 2049 or n2 != n:
        raise ValueError("label file does not match the image file")
    labels = np.frombuffer(raw, np.uint8, offset=8)
    if partition == "train":           # tensorflow's read_data_sets holds out the first 5000 training digits for validation
        return images[5000:], labels[5000:]
    if partition == "validation":
        return images[:5000], labels[:5000]
    return images, labels
