#!/bin/bash
# PPO-side kernels: their tests, then a kernel trace of the PPO loop (rollout graph + update graphs)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -u -m pytest tests/test_train_gpu.py tests/test_golden_ppo.py tests/test_reference_pinned_gpu.py -m gpu -q --durations=5 -x > $O/ppo_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|FAILED|Error|assert" $O/ppo_pytest_gpu.log | tail -20
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o ppo -- python3 $R/tools/dev/prof_ppo.py > $O/prof_ppo.log 2>&1; echo "rocprof rc $?"; tail -2 $O/prof_ppo.log
python3 $R/tools/dev/rocpd_export.py stats /tmp/prof_ppo/ppo_results.db $O/${1:-r04}_ppo_kernel_stats.csv 2>&1 | tail -3 || find /tmp/prof_ppo | head
head -12 $O/${1:-r04}_ppo_kernel_stats.csv
# optional A/B variants of the PPO kernels (ab_libs/<name>.so, tools/dev/build_ppo_variant.py): gpu_ppo.sh TAG name ...
shift
for v in "$@"; do
  export SOLORL_LIB=$R/ab_libs/$v.so
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo_$v -o ppo -- python3 $R/tools/dev/prof_ppo.py > $O/prof_ppo_$v.log 2>&1; echo "rocprof $v rc $?"
  python3 $R/tools/dev/rocpd_export.py stats /tmp/prof_ppo_$v/ppo_results.db $O/ppo_kernel_stats_$v.csv > /dev/null 2>&1; echo "== $v"; grep "ppo_\|step_kernel" $O/ppo_kernel_stats_$v.csv
done
