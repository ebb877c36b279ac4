#!/bin/bash
# PPO side after a kernel change: its tests, the step_act A/B, the PPO kernel trace, the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -u -m pytest tests/test_train_gpu.py tests/test_golden_ppo.py tests/test_reference_pinned_gpu.py -m gpu -q --durations=5 -x > $O/ppo_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; grep -E "passed|failed|FAILED|Error|assert" $O/ppo_pytest_gpu.log | tail -20
[ $rc -eq 0 ] || exit $rc
python -u tools/dev/ab_step_act.py 2>&1 | grep -v amdgpu > $O/${1:-r04}_step_act_ab.txt; cat $O/${1:-r04}_step_act_ab.txt
python -u bench.py --strict --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc $?"; python3 -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d['f64']['value'], json.dumps(d['ppo_loop']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o ppo -- python3 $R/tools/dev/prof_ppo.py > $O/prof_ppo.log 2>&1; echo "rocprof rc $?"
python3 $R/tools/dev/rocpd_export.py stats /tmp/prof_ppo/ppo_results.db $O/${1:-r04}_ppo_kernel_stats.csv 2>&1 | tail -3
head -7 $O/${1:-r04}_ppo_kernel_stats.csv
