#!/bin/bash
# rocprofv3 --kernel-trace --stats of 30 single evaluations of the StructDiffusion baseline (tools/profile_sd.py) -> gpurun_out/<tag>/kernel_stats_sd.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sd
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sd -- python $R/tools/profile_sd.py 30 > $OUT/prof_sd.log 2>&1
f=$(find /tmp/prof_sd -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats_sd.csv
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-72s calls %5s avg %8.2f us' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:72], r['Calls'], float(r['AverageNs']) / 1e3))
PY
