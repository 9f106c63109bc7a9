# final session of round 6: evidence of the final binary (tag r06_zz), then the 12-seed training statistics on it
bash tools/runs/r06_final.sh r06_zz
GUARD=1e-6 TAG=r06_tr bash tools/runs/r06_train.sh > gpurun_out/r06_tr_train.log 2>&1
tail -22 gpurun_out/r06_tr_train.log
