#!/bin/bash
# Round 4, GPU call 3: the whole -m gpu suite incl. tests/test_parity_gpu3.py (no -x), fp64 512-register A/B.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/parity_stats.json $O/trained_policy_stats.json
python -u -m pytest tests -m gpu -q --durations=10 -s > $O/r04b_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|FAILED|Error" $O/r04b_pytest_gpu.log | tail -30
python -u tools/dev/ab_f64.py default ab_libs/f64_w1g1.so ab_libs/f64_w1g2.so ab_libs/f64_w1g3.so ab_libs/f64_w1g5.so 2>&1 | grep -v amdgpu > $O/r04b_f64_ab.txt; cat $O/r04b_f64_ab.txt
