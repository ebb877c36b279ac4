#!/bin/bash
# kernel trace of the PPO loop for A/B variants of the PPO kernels (ab_libs/<name>.so, tools/dev/build_ppo_variant.py): gpu_ppo_ab.sh name ...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  export SOLORL_LIB=$R/ab_libs/$v.so
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo_$v -o ppo -- python3 $R/tools/dev/prof_ppo.py > $O/prof_ppo_$v.log 2>&1; echo "rocprof $v rc $?"
  python3 $R/tools/dev/rocpd_export.py stats /tmp/prof_ppo_$v/ppo_results.db $O/ppo_kernel_stats_$v.csv > /dev/null 2>&1; echo "== $v"; grep "ppo_" $O/ppo_kernel_stats_$v.csv
done
