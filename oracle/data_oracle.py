"""CPU restatement of the reference's multi-MNIST synthesis (attend_infer_repeat/data/data.py:19-107) -- TEST INFRASTRUCTURE ONLY.

The reference draws from the GLOBAL legacy numpy generator (np.random.randint / choice / rand) and reads MNIST through a
TensorFlow download; here the digit templates and the generator are INJECTED (`templates`, `rng` = a numpy RandomState), and
the call sequence on `rng` is the reference's, draw for draw:

    nums    = rng.randint(max_objects + 1, size=n_samples, dtype=uint8)              data.py:52
    per sample attempt:  indices = rng.choice(n_templates, n, replace=False)        data.py:73
        per object and per placement try:  u = rng.rand(n)                          data.py:58-60  (n = objects of the SAMPLE)

Quirks restated, not fixed:
  * make_p multiplies rand(n) -- n numbers, n = the sample's object count -- with the 2-vector of free positions: with ONE object
    both coordinates come from the SAME draw (the digit sits on the canvas diagonal), with two objects y gets the first and x the
    second draw, and three or more objects cannot broadcast (ValueError) -- the generator only works for n_objects <= 2.
  * a sample is abandoned when the try counter reaches 5 even if the fifth draw found a free spot (data.py:89-91), and the
    counter runs over all objects of the sample.
  * dim_coords measures a template's extent as the NUMBER of non-empty rows / columns and places the start at
    last - count + 1 (data.py:19-23): a glyph with an empty row inside its support is cropped short at the top.
Third party, not in the tree: scipy.misc.imresize(x, obj_size) (data.py:55) = bytescale + PIL resize; `bytescale` below restates
its published arithmetic for float templates; a resize to a different obj_size is not restated (templates are injected at
obj_size).  PARITY: pinned only to the reference's source text (no fixture exists in /root/reference for this path).
"""
import numpy as np


def bytescale(data):
    """scipy.misc.bytescale(data) with its defaults (cmin = data.min(), cmax = data.max(), low = 0, high = 255): uint8 passes through."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = 255.0 / cscale
    return (((data - cmin) * scale).clip(0, 255) + 0.5).astype(np.uint8)


def dim_coords(proj):                                   # data.py:19-23
    nz = np.asarray(proj) > 0
    count = int(nz.sum())
    last = int(np.argmax(np.arange(len(nz)) * nz))
    return last - count + 1, count


def template_dimensions(template):                      # data.py:26-32
    y0, hy = dim_coords(template.sum(1))
    x0, wx = dim_coords(template.sum(0))
    return (y0, x0), (hy, wx)


def create_mnist(templates, labels, rng, canvas_size=(50, 50), n_objects=(0, 2), n_samples=None, dtype=np.uint8,
                 expand_nums=True, with_overlap=False, n_tries=5):
    """data.py:35-107 with the MNIST partition replaced by (`templates` [N, h, w] already at obj_size, `labels` [N]) and the
    global np.random replaced by `rng`."""
    n_templates = len(templates)
    if n_samples is None:
        n_samples = n_templates
    max_objects = sorted(int(v) for v in np.atleast_1d(n_objects).ravel())[-1]
    canvas = np.asarray(canvas_size)
    imgs = np.zeros((n_samples,) + tuple(canvas_size), dtype=dtype)
    out_labels = np.zeros((n_samples, max_objects), dtype=np.uint8)
    nums = rng.randint(max_objects + 1, size=n_samples, dtype=np.uint8)
    occupancy = np.zeros(tuple(canvas_size), dtype=bool)
    sample = 0
    while sample < n_samples:
        n = nums[sample]
        tries, give_up = 0, False
        if n > 0:
            picked = rng.choice(n_templates, n, replace=False)
            occupancy[...] = False
            for j in range(int(n)):
                t = bytescale(templates[picked[j]])
                out_labels[sample, j] = labels[picked[j]]
                (ty, tx), (sy, sx) = template_dimensions(t)
                free = canvas - np.asarray([sy, sx])
                pos = np.round(rng.rand(n) * free).astype(np.int32)
                if not with_overlap:
                    while occupancy[pos[0]:pos[0] + sy, pos[1]:pos[1] + sx].any() and tries < n_tries:
                        pos = np.round(rng.rand(n) * free).astype(np.int32)
                        tries += 1
                    if tries == n_tries:
                        give_up = True
                        break
                imgs[sample, pos[0]:pos[0] + sy, pos[1]:pos[1] + sx] = t[ty:ty + sy, tx:tx + sx]
                occupancy[pos[0]:pos[0] + sy, pos[1]:pos[1] + sx] = True
        if give_up:
            imgs[sample, ...] = 0
        else:
            sample += 1
    if expand_nums:
        wide = np.zeros((max_objects + 1, n_samples, 1), dtype=np.uint8)
        for s, n in enumerate(nums):
            wide[:n, s] = 1
        nums = wide
    return dict(imgs=imgs, labels=out_labels, nums=nums)
