#!/bin/bash
# round 6: the f1 statistics of round 5 (12 seeds x 300 k updates, --guard-degenerate 1e-6) on the final round-6 binary: the glimpse-space
# canvas backward, the lean attend read and the gather fold change roundings, not the algorithm -- the distribution must look the same
export GUARD=${GUARD:-1e-6} TAG=${TAG:-r06_t}
bash tools/runs/r05_train.sh
