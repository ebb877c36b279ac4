#!/bin/bash
# Second learning probe: (a) configs/basic.yaml unmodified for 4e9 env-steps (does walk learn to stay up given time?), (b) the same recipe with
# use_urdf_inertia: 1 for 1e9 (K2 ablation).   logs -> gpurun_out/learn2/*.log
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/learn2; mkdir -p $O; cd $R
COMMON="--num-agents 4096 --use-gae --use-linear-lr-decay --lr 2.5e-4 --entropy-coef 0.01 --clip-param 0.1 --ppo-epoch 5 --mini-batch-size 32768 --seed 1 --log-interval 40"
run() { echo "== $1"; shift; python -u train_ppo.py $COMMON "$@" 2>&1 | grep -v amdgpu.ids | paste - - - ; }
run "configs/basic.yaml unmodified (Solo8, walk, torque, treadmill), 4e9 env-steps" --config-file configs/basic.yaml --num-env-steps 4.0e9 > $O/basic_walk_4e9.log
sed 's/^task: walk/task: walk\nuse_urdf_inertia: 1/' configs/basic.yaml > $O/basic_urdf.yaml
run "configs/basic.yaml + use_urdf_inertia: 1, 1e9 env-steps" --config-file $O/basic_urdf.yaml --num-env-steps 1.0e9 > $O/basic_walk_urdf.log
tail -n 3 $O/*.log
