#!/bin/bash
# A/B of environment switches inside ONE gpurun call: tools/ab_env.sh "CCSP_X=1" "CCSP_X=2 CCSP_Y=3" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg python $R/bench.py $BENCH_ARGS --no-cpu-baseline --no-roofline --no-evaluate 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read())['value'])")
    echo "$cfg: $v"
  done
done
