#!/bin/bash
# rocprofv3 kernel stats of tools/profile_eval.py for every ablation build tools/abl_*.so (tools only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/abl_*; for so in "" $(ls $R/tools/abl_*.so 2>/dev/null); do
  tag=$(basename "${so:-default}" .so)
  CCSP_SO=$so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abl_$tag --output-format csv -- python $R/tools/profile_eval.py ${EVALS:-100} ${GRAPHS:-256} > /dev/null 2>&1
  f=$(find /tmp/abl_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_rowgemm', 'k_edge', 'k_node', 'k_fused')):
        print('   %-44s avg %8.1f us  min %8.1f' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
