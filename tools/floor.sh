#!/bin/bash
# per-kernel durations of single evaluations at small batch sizes (latency floor), default build + tools/abl_*.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for so in "" $(ls $R/tools/abl_*.so 2>/dev/null); do
  for B in ${FLOOR_BATCHES:-1 128}; do
    rm -rf /tmp/kk
    CCSP_SO=$so rocprofv3 --kernel-trace --stats -d /tmp/kk --output-format csv -- python $R/tools/profile_eval.py 50 $B > /dev/null 2>&1
    echo "== $(basename "${so:-default}" .so) graphs=$B"
    python - "$(find /tmp/kk -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_rowgemm_bf', 'k_edge_bf', 'k_node<')):
        print('   %-26s avg %7.1f us  min %7.1f' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:26], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
  done
done
