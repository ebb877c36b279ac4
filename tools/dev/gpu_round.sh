#!/bin/bash
# One GPU-box call: GPU test suite, smoke, bench, kernel trace and the two HBM-traffic PMC passes.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -u -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 && echo "pytest ok" && \
python -u -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && echo "smoke ok" && \
python -u bench.py > $O/bench.json 2> $O/bench.err && echo "bench ok" && \
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st -- python3 $R/bench.py --steps 200 --warmup 20 --ppo-steps 0 --no-cpu-baseline > $O/prof_stats.log 2>&1 && echo "stats ok" && \
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 $R/tools/dev/prof_step.py > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 $R/tools/dev/prof_step.py > $O/pmc_write.log 2>&1 && \
python3 $R/tools/dev/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch.txt && python3 $R/tools/dev/pmc_summary.py $O/pmc_write > $O/pmc_write.txt && echo "pmc ok"
tail -3 $O/pytest_gpu.log; cat $O/bench.json; cat $O/pmc_fetch.txt $O/pmc_write.txt
