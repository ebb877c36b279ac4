#!/bin/bash
# Trains the small policies committed under tests/golden/policies/ (VERDICT r03 "Next" #2: trained-policy parity test): README recipe
# (lr 2.5e-4, entropy 0.01, clip 0.1, 5 epochs, GAE, linear lr decay), 4096 envs, one GPU.  python writes straight into a log under
# gpurun_out/ (no pipe).  usage: train_fixture_policies.sh [which...]   which = stand8 pointgoal12 walk8 walk12 walk12long walk8long walk12seed2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/policies; mkdir -p $O; cd $R
COMMON="--num-agents 4096 --use-gae --use-linear-lr-decay --lr 2.5e-4 --entropy-coef 0.01 --clip-param 0.1 --ppo-epoch 5 --mini-batch-size 32768 --seed 1 --log-interval 100 --save-interval 100000"
run() {   # name config task steps limit [extra flags, e.g. "--seed 2": the last occurrence of a flag wins]
  echo "== $1: $2 --task $3, $4 env-steps" > $O/$1.log
  timeout -k 10 $5 python -u train_ppo.py $COMMON $6 --config-file configs/$2 --task $3 --num-env-steps $4 --logdir $O/$1 --timestamp run >> $O/$1.log 2>&1 || return 1
  cp $O/$1/*/solo.pt $O/$1.pt && tail -3 $O/$1.log
}
for w in ${@:-stand8 pointgoal12 walk8}; do
  case $w in
    stand8) run stand8 basic.yaml stand 1e9 200 || exit 1;;
    pointgoal12) run pointgoal12 basic12.yaml pointgoal 2e9 300 || exit 1;;
    walk8) run walk8 basic.yaml walk 8e9 900 || exit 1;;
    walk12) run walk12 basic12.yaml walk 8e9 900 || exit 1;;
    walk12long) run walk12long basic12.yaml walk 1.6e10 1150 || exit 1;;
    walk8long) run walk8long basic.yaml walk 1.6e10 1150 || exit 1;;
    walk12seed2) run walk12seed2 basic12.yaml walk 1.6e10 1150 "--seed 2" || exit 1;;
  esac
done
