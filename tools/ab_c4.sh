#!/bin/bash
# C4 (256 x 12 triangles, MALA, energy mode) with EVERY evaluation recomputed (CCSP_MALA_REUSE=0: the kernels at full work) and with the
# reuse of unmoved states, for a list of environment settings, inside ONE gpurun call.   usage: tools/ab_c4.sh "CCSP_EDGE_FB=0" "CCSP_EDGE_FB=2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "$@"; do
    a=$(env $cfg CCSP_MALA_REUSE=0 python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read())['value'])")
    b=$(env $cfg python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'])")
    echo "$cfg: recomputing $a  with reuse $b"
  done
done
