#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench chain per configuration (tools only).
# usage: tools/prof_stats.sh <tag> ["ENV=.. ENV=.." ...]   -> gpurun_out/<tag>/stats_<i>.csv + stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
: > $OUT/stats.txt
for cfg in "$@"; do
  i=$((i+1))
  rm -rf /tmp/st_$i
  env $cfg timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st_$i --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS > $OUT/stats_$i.log 2>&1
  f=$(find /tmp/st_$i -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/stats_$i.csv
  echo "== $cfg" >> $OUT/stats.txt
  tail -1 $OUT/stats_$i.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('   samples/s under the profiler: %.1f' % r['value'])" >> $OUT/stats.txt 2>&1
  python - "$f" >> $OUT/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print('   %-60s calls %7s  avg %8.2f us  min %8.2f  max %8.2f  %5.1f %%' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60], r['Calls'],
          float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
cat $OUT/stats.txt
