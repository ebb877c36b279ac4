#!/bin/bash
# Learning probe on the README's own recipe (Next #3): reference README.md:34 hyper-parameters (lr 2.5e-4, entropy 0.01, clip 0.1,
# 5 PPO epochs, GAE, linear LR decay, seed 1), 4096 envs, mini-batch scaled to 50 per epoch, >= 1e9 env-steps per run.
# usage (GPU box): bash tools/dev/learn_probe.sh [env-steps]     logs -> gpurun_out/learn/*.log
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/learn; mkdir -p $O; cd $R
S=${1:-1.0e9}
COMMON="--num-agents 4096 --use-gae --use-linear-lr-decay --lr 2.5e-4 --entropy-coef 0.01 --clip-param 0.1 --ppo-epoch 5 --mini-batch-size 32768 --seed 1 --log-interval 20 --num-env-steps $S"
run() { echo "== $1"; shift; python -u train_ppo.py $COMMON "$@" 2>&1 | grep -v amdgpu.ids | paste - - - ; }
run "configs/basic.yaml unmodified (Solo8, walk, torque, treadmill)" --config-file configs/basic.yaml > $O/basic_walk.log
run "configs/basic12.yaml unmodified (Solo12, pointgoal, torque)" --config-file configs/basic12.yaml > $O/basic12_pointgoal.log
run "configs/basic12.yaml --task walk (Solo12, walk, torque)" --config-file configs/basic12.yaml --task walk > $O/basic12_walk.log
run "configs/basic.yaml --task stand (Solo8, stand, torque, treadmill)" --config-file configs/basic.yaml --task stand > $O/basic_stand.log
# ablation: torque held over the frame_skip sub-steps (K8 off)
sed 's/^task: walk/task: walk\nhold_torque: 1/' configs/basic.yaml > $O/basic_hold.yaml
run "configs/basic.yaml + hold_torque: 1" --config-file $O/basic_hold.yaml > $O/basic_walk_hold.log
tail -n 4 $O/*.log
