#!/usr/bin/env python
"""Writes tests/golden/multi_mnist_case.npz: a small multi-MNIST synthesis case (templates in, dataset out) produced by the CPU
restatement of the reference's generator (oracle/data_oracle.py <- attend_infer_repeat/data/data.py:35-107).  The reference itself
cannot run here (Python 2, TensorFlow MNIST download), so -- like the other fixtures in this directory -- the vectors come from
the restatement and are PARITY UNPINNED with respect to the original; what they pin is the product against the restatement and
both against accidental change."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import data_oracle as DO  # noqa: E402
from attend_infer_repeat_amd import data as D  # noqa: E402  (procedural templates only: the inputs of the case)


def main():
    templates, labels = D.procedural_digit_templates(24, seed=12)
    templates = templates.copy()
    templates[0] = 0; templates[0, 5:9, 6:20] = 200; templates[0, 15:22, 8:12] = 90      # a support with empty rows inside
    seed, n_samples = 1234, 48
    d = DO.create_mnist(templates, labels, np.random.RandomState(seed), n_samples=n_samples)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "multi_mnist_case.npz"), templates=templates, labels=labels,
                        seed=seed, n_samples=n_samples, imgs=d["imgs"], out_labels=d["labels"], nums=d["nums"])
    print("wrote multi_mnist_case.npz", d["imgs"].shape, int(d["nums"].sum()))


if __name__ == "__main__":
    main()
