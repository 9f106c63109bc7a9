#!/bin/bash
# counterpart of the reference's scripts/train_multi_mnist.sh
cd "$(dirname "$0")/.."
python -m attend_infer_repeat_amd.scripts.multi_mnist "$@"
