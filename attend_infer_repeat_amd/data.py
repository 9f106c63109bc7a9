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
