#!/bin/bash
# round 3: phase traces of the C2-sized kernels in the regimes the benchmark runs them (one lane, two lanes, throughput)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "256 1" "256 2" "1024 1"; do
  set -- $cfg
  echo "=== $1 graphs, CCSP_LANES=$2 ==="
  CCSP_LANES=$2 python tools/trace_run.py $1
done
