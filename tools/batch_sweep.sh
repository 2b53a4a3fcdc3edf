#!/bin/bash
# samples/s and us per evaluation against batch size, one lane (tools only)
R=${GRAFT_REPO_ROOT:-/root/repo}
for B in "$@"; do
  CCSP_LANES=1 python $R/bench.py --no-cpu-baseline --no-roofline --graphs-per-gpu $B --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('graphs %4d  %7.1f samples/s  %6.1f us/eval' % ($B, r['value'], r['ms_per_step']*1e3/11000))"
done
