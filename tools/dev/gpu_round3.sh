#!/bin/bash
# quick GPU check of a subset of tests: gpu_round3.sh "<pytest -k expression>"
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -u -m pytest tests -m gpu -q --durations=5 -s -x -k "$1" > $O/quick_pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|FAILED|Error|assert" $O/quick_pytest_gpu.log | tail -20
