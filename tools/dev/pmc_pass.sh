#!/bin/bash
# usage: pmc_pass.sh <tag> <counter> [<counter> ...]   -- one rocprofv3 --pmc pass over tools/dev/prof_step.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" -d $O/pmc_$tag -o p -- python3 $R/tools/dev/prof_step.py > $O/pmc_$tag.log 2>&1
rc=$?
python3 $R/tools/dev/rocpd_export.py pmc $O/pmc_$tag/p_results.db $O/pmc_$tag.txt
exit $rc
