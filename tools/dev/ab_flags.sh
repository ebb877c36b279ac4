#!/bin/bash
# A/B of compiler flag sets on the headline bench (value = env-steps/s, graph replay); usage: ab_flags.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  SOLORL_BUILD_FLAGS="$f" python -m solorl_amd.build -f > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --ppo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[%s] %.2f M env-steps/s  %.4f ms' % (sys.argv[1], d['value']/1e6, d['ms_per_step']))" "$f"
  done
done
python -m solorl_amd.build -f > /dev/null 2>&1
