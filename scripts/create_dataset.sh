#!/usr/bin/env bash
# counterpart of the reference's scripts/create_dataset.sh; needs the MNIST idx files under data/MNIST_data (no network here)
python attend_infer_repeat_amd/scripts/create_dataset.py "$@"
