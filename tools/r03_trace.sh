#!/bin/bash
# round 3: phase traces (tools/trace_run.py) of several trace builds in one call.  usage: tools/r03_trace.sh "<graphs> <lanes>" [kernel-name-prefix] so1 so2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cfg=$1; shift
K=${1}; shift
set -- $cfg "$K" "$@"
G=$1; L=$2; K=$3; shift; shift; shift
for so in "$@"; do
  echo "=== $(basename $so .so): $G graphs, CCSP_LANES=$L ==="
  CCSP_SO=$R/tools/$so CCSP_LANES=$L python tools/trace_run.py $G 2>/dev/null | grep -A 18 "^$K"
done
