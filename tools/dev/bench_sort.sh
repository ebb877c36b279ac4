#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in 8192 16384 65536; do for s in 0 1; do SOLORL_SORT=$s python -u tools/dev/bench_n.py $n 2>&1 | grep "env-steps" | sed "s/$/ sort=$s/"; done; done
