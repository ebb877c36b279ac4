#!/bin/bash
# Second evidence call of a round: PMC passes (tools/dev/pmc_round.sh) and the per-wavefront timing histogram on a PREBUILT -DSOLO_WAVE_TIMING
# library (tools/dev/build_variant.py timing -DSOLO_WAVE_TIMING -> ab_libs/timing.so; take ab_libs/ out of .gpurunignore for this call).
# usage: gpu_round2.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04}
cd $R
bash tools/dev/pmc_round.sh $T > $O/${T}_pmc_round.log 2>&1; echo "pmc rc $?"; tail -12 $O/${T}_pmc_round.log
cd $R
SOLORL_BUILD_DEFINES="SOLO_WAVE_TIMING" SOLORL_LIB=$R/ab_libs/timing.so python -u tools/dev/wave_hist.py 4096 2>&1 | grep -v amdgpu > $O/${T}_wave_hist.txt; echo "wave_hist rc $?"; cat $O/${T}_wave_hist.txt
SOLORL_BUILD_DEFINES="SOLO_WAVE_TIMING" SOLORL_LIB=$R/ab_libs/timing.so python -u tools/dev/wave_hist.py 8192 2>&1 | grep -v amdgpu > $O/${T}_wave_hist_8192.txt; echo "wave_hist 8192 rc $?"
