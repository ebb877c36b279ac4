#!/bin/bash
# Round 4, first GPU call: the whole -m gpu suite on the new defaults (no -x: the quantile references in tests/golden/parity_measured.json
# were measured with round 3's pyramid, every mismatch is wanted), friction-model A/B of the step kernel, bench line, fixture policies.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -u -m pytest tests -m gpu -q --durations=8 > $O/r04a_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -25 $O/r04a_pytest_gpu.log
python -u tools/dev/bench_friction.py 2>&1 | grep -v amdgpu > $O/r04a_friction_ms.txt; cat $O/r04a_friction_ms.txt
python -u bench.py --strict > $O/r04a_bench.json 2> $O/r04a_bench.err; echo "bench rc $?"; cat $O/r04a_bench.json
bash tools/dev/train_fixture_policies.sh stand8 pointgoal12
