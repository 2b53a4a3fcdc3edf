#!/bin/bash
# inter-kernel gaps on the chain's streams: rocprofv3 --kernel-trace of a short chain segment, then per queue the time between
# one kernel's end and the next one's start, keyed by the pair of kernels.   usage: tools/gaps.sh <tag> <graphs> ["ENV=..." ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; G=$2; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); rm -rf /tmp/gp_$i
  env $cfg timeout 600 rocprofv3 --kernel-trace -d /tmp/gp_$i --output-format csv -- python $R/tools/segment_run.py $G > $OUT/gaps_$i.log 2>&1
  f=$(find /tmp/gp_$i -name "*kernel_trace.csv" | head -1)
  echo "== $cfg ($G graphs)" | tee -a $OUT/gaps.txt
  python $R/tools/gaps.py "$f" | tee -a $OUT/gaps.txt
done
