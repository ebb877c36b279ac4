#!/bin/bash
# true cost of the idempotent phases: launch time with the phase run twice minus the normal launch time
cd $GRAFT_REPO_ROOT
for d in ${DUPS:-"" SOLO_DUP_FRONT SOLO_DUP_LEGS SOLO_DUP_BASE SOLO_DUP_FINISH}; do
  SOLORL_BUILD_DEFINES="$d" python -m solorl_amd.build -f > /dev/null 2>&1
  echo "build [$d]"; python -u tools/dev/bench_iters.py 4096 2>&1 | grep "iters 50 frame_skip 4"
done
python -m solorl_amd.build -f > /dev/null 2>&1
