#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { v=$(env $2 python $R/bench.py $3 --no-cpu-baseline --no-roofline --no-evaluate 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'])"); echo "$1 [$2]: $v"; }
for rep in 1 2; do
  for g in 16 32 64 96 128; do for m in direct stream; do run "c2 g$g" "CCSP_NODE=$m" "--graphs-per-gpu $g"; done; done
  for g in 16 32; do for m in direct stream; do run "c5 g$g" "CCSP_NODE=$m" "--config c5 --graphs-per-gpu $g"; done; done
done
