#!/bin/bash
# One GPU-box call: GPU test suite, smoke, bench (1 rank + a 2-rank rehearsal on the one GPU), kernel trace, PMC passes.
# usage: gpu_round.sh [tag]      outputs under gpurun_out/<tag>_*
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r02}
cd $R
python -u -m pytest tests -m gpu -x -q --durations=8 > $O/${T}_pytest_gpu.log 2>&1; rc=$?; tail -15 $O/${T}_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -u -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1 && echo "smoke ok" && \
python -u bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err && echo "bench ok" && cat $O/${T}_bench.json && \
python -u bench.py --gpus 2 --rehearse-on-one-gpu --steps 100 --ppo-steps 32 --ppo-epoch 1 > $O/${T}_bench_g2.json 2> $O/${T}_bench_g2.err && echo "bench g2 ok" && cat $O/${T}_bench_g2.json && \
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats -d $O/${T}_prof_stats -o st -- python3 $R/bench.py --ppo-steps 0 --no-cpu-baseline > $O/${T}_prof_stats.log 2>&1 && echo "stats ok" && \
python3 $R/tools/dev/rocpd_export.py stats $O/${T}_prof_stats/st_results.db $O/${T}_kernel_stats.csv && cat $O/${T}_kernel_stats.csv | head -8
