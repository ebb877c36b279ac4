#!/bin/bash
# Third learning probe (longer horizons): Solo12 walk for 4e9 env-steps, Solo8 walk (configs/basic.yaml unmodified) for 8e9, Solo12 pointgoal 2e9.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/learn3; mkdir -p $O; cd $R
COMMON="--num-agents 4096 --use-gae --use-linear-lr-decay --lr 2.5e-4 --entropy-coef 0.01 --clip-param 0.1 --ppo-epoch 5 --mini-batch-size 32768 --seed 1 --log-interval 80"
run() { echo "== $1"; shift; python -u train_ppo.py $COMMON "$@" 2>&1 | grep -v amdgpu.ids | paste - - - ; }
run "configs/basic12.yaml --task walk (Solo12, walk, torque), 4e9 env-steps" --config-file configs/basic12.yaml --task walk --num-env-steps 4.0e9 > $O/basic12_walk_4e9.log
run "configs/basic12.yaml unmodified (Solo12, pointgoal), 2e9 env-steps" --config-file configs/basic12.yaml --num-env-steps 2.0e9 > $O/basic12_pointgoal_2e9.log
run "configs/basic.yaml unmodified (Solo8, walk, torque, treadmill), 8e9 env-steps" --config-file configs/basic.yaml --num-env-steps 8.0e9 > $O/basic_walk_8e9.log
tail -n 2 $O/*.log
