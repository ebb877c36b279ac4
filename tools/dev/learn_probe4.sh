#!/bin/bash
# Fourth learning probe: the headline workload's task (configs/basic12.yaml --task walk: Solo12, walk, torque) for 1.2e10 env-steps on one GPU
# (probe 3 stopped at 4e9 with success 0.46 and still rising).  python writes straight into the log under gpurun_out/ (no pipe: a buffered
# pipe looks like a hung run to gpurun); a line every 400 updates.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/learn4; mkdir -p $O; cd $R
COMMON="--num-agents 4096 --use-gae --use-linear-lr-decay --lr 2.5e-4 --entropy-coef 0.01 --clip-param 0.1 --ppo-epoch 5 --mini-batch-size 32768 --seed 1 --log-interval 400"
echo "== configs/basic12.yaml --task walk (Solo12, walk, torque), 1.2e10 env-steps" > $O/basic12_walk_12e9.log
timeout -k 10 960 python -u train_ppo.py $COMMON --config-file configs/basic12.yaml --task walk --num-env-steps 1.2e10 >> $O/basic12_walk_12e9.log 2>&1
grep -c Updates $O/basic12_walk_12e9.log
