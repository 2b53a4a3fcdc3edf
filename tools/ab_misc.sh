#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { v=$(env $2 python $R/bench.py $3 --no-cpu-baseline --no-roofline --no-evaluate 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'])"); echo "$1 [$2]: $v"; }
for rep in 1 2; do
  run c2 "CCSP_LANES=2" ""; run c2 "CCSP_LANES=3" ""; run c2 "CCSP_LANES=1" ""
  run c5 "CCSP_LANES=1" "--config c5"; run c5 "CCSP_LANES=2 CCSP_LANE_MIN_EDGES=1000" "--config c5"
  run "c2 g128" "CCSP_LANES=1" "--graphs-per-gpu 128"; run "c2 g128" "CCSP_LANES=2 CCSP_LANE_MIN_EDGES=1000" "--graphs-per-gpu 128"
  run "c2 g512" "CCSP_LANES=2" "--graphs-per-gpu 512"; run "c2 g512" "CCSP_LANES=3" "--graphs-per-gpu 512"
done
