#!/bin/bash
# One GPU-box call: GPU test suite, smoke, bench (1 rank + 2- and 4-rank rehearsals on the one GPU: the box admits six GPU processes, so
# the 8-rank node shape is rehearsed with gloo on the CPU instead, tests/test_dist_cpu.py), kernel trace, batch-size sweep, friction A/B.
# usage: gpu_round.sh [tag]      outputs under gpurun_out/<tag>_*
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04}
cd $R
rm -f $O/parity_stats.json $O/trained_policy_stats.json
if [ -z "$SKIP_TESTS" ]; then
python -u -m pytest tests -m gpu -x -q --durations=8 -s > $O/${T}_pytest_gpu.log 2>&1; rc=$?; grep -E "passed|failed|FAILED|Error" $O/${T}_pytest_gpu.log | tail -15
[ $rc -eq 0 ] || exit $rc
fi
python -u -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1 && echo "smoke ok" && \
python -u bench.py --strict > $O/${T}_bench.json 2> $O/${T}_bench.err && echo "bench ok" && cat $O/${T}_bench.json && \
python -u bench.py --gpus 2 --rehearse-on-one-gpu --steps 100 --ppo-steps 32 --ppo-epoch 1 --no-f64 > $O/${T}_bench_g2.json 2> $O/${T}_bench_g2.err && echo "bench g2 ok" && \
python -u bench.py --gpus 4 --rehearse-on-one-gpu --envs-per-gpu 1024 --steps 100 --ppo-steps 32 --ppo-epoch 1 --no-f64 > $O/${T}_bench_g4.json 2> $O/${T}_bench_g4.err && echo "bench g4 ok" && \
python -u tools/dev/bench_sizes.py 1024 4096 8192 16384 65536 262144 2>&1 | grep -v amdgpu > $O/${T}_sizes.txt && BENCH_CFG=pd50 python -u tools/dev/bench_sizes.py 8192 2>&1 | grep -v amdgpu >> $O/${T}_sizes.txt && cat $O/${T}_sizes.txt && \
python -u tools/dev/bench_friction.py 2>&1 | grep -v amdgpu > $O/${T}_friction_ms.txt && cat $O/${T}_friction_ms.txt && \
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats -d $O/${T}_prof_stats -o st -- python3 $R/bench.py --ppo-steps 0 --no-cpu-baseline > $O/${T}_prof_stats.log 2>&1 && echo "stats ok" && \
python3 $R/tools/dev/rocpd_export.py stats $O/${T}_prof_stats/st_results.db $O/${T}_kernel_stats.csv && cat $O/${T}_kernel_stats.csv | head -8
