#!/bin/bash
# row GEMM staging mode (CCSP_ROW_MODE) across batch sizes, inside ONE gpurun call
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, env, args
  v=$(env $2 python $R/bench.py $3 --no-cpu-baseline --no-roofline --no-evaluate 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], d.get('mala',{}).get('value_recomputing_every_evaluation',''))")
  echo "$1 [$2]: $v"
}
for rep in 1 2; do
  for g in 32 48 64 96; do for m in 4 0 2; do run "c2 g$g" "CCSP_ROW_MODE=$m" "--graphs-per-gpu $g"; done; done
  for g in 16 32; do for m in 4 0; do run "c5 g$g" "CCSP_ROW_MODE=$m" "--config c5 --graphs-per-gpu $g"; done; done
done
