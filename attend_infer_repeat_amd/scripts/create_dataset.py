#!/usr/bin/env python
"""Counterpart of the reference's dataset script (scripts/create_dataset.sh -> data/data.py:160-174): writes
mnist_train.pickle (60000 canvases) and mnist_validation.pickle (10000) of multi-digit MNIST for scripts/multi_mnist.py.

The reference downloads MNIST through TensorFlow; this container has no network, so `--mnist-dir` must hold the standard
idx-ubyte files (train-images-idx3-ubyte[.gz], train-labels-idx1-ubyte[.gz]).
"""
import argparse
import os
import os.path as osp
import pickle
import sys

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from attend_infer_repeat_amd.data import create_multi_mnist, load_mnist_idx  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mnist-dir", default="data/MNIST_data")
    ap.add_argument("--out-dir", default="data")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    os.makedirs(args.out_dir, exist_ok=True)
    for i, (partition, n) in enumerate((("train", 60000), ("validation", 10000))):       # data.py:161-162
        print('Processing partition "{}"'.format(partition))
        templates, labels = load_mnist_idx(args.mnist_dir, partition)
        data = create_multi_mnist(templates, labels, n_samples=n, seed=args.seed + i)
        path = osp.join(args.out_dir, "mnist_{}.pickle".format(partition))
        print('saving to "{}"'.format(path))
        with open(path, "wb") as f:
            pickle.dump(data, f, pickle.HIGHEST_PROTOCOL)


if __name__ == "__main__":
    main()
