#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by several %): default + every tools/abl_*.so, twice
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for so in "" $(ls $R/tools/abl_*.so 2>/dev/null); do
    v=$(CCSP_SO=$so python $R/tools/bench_so.py --no-cpu-baseline --no-roofline --no-evaluate "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read())['value'])")
    echo "$(basename "${so:-default}" .so) $v"
  done
done
