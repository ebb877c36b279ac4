#!/bin/bash
# Round 4, GPU call: step_act tests, trained-policy tests on the retrained fixtures, bench line (A/B of the rollout with and without solorl_step_act).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -u -m pytest tests -m gpu -q --durations=5 -s -k "step_act or trained_policy or graphed_rollout" > $O/r04d_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|FAILED|Error|trained\[" $O/r04d_pytest_gpu.log | tail -40
python -u bench.py --strict --no-f64 --no-cpu-baseline > $O/r04d_bench.json 2> $O/r04d_bench.err; echo "bench rc $?"; python -c "
import json; d=json.load(open('$O/r04d_bench.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['ppo_loop']['env_steps_per_s'], d['ppo_loop']['rollout_env_steps_per_s'], d['ppo_loop']['rollout_ms'], d['ppo_loop']['update_ms'])"
SOLORL_STEP_ACT=0 python -u bench.py --strict --no-f64 --no-cpu-baseline > $O/r04d_bench_two_launch.json 2> $O/r04d_bench2.err; echo "bench rc $?"; python -c "
import json; d=json.load(open('$O/r04d_bench_two_launch.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['ppo_loop']['env_steps_per_s'], d['ppo_loop']['rollout_env_steps_per_s'], d['ppo_loop']['rollout_ms'], d['ppo_loop']['update_ms'])"
