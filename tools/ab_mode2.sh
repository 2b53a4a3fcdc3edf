R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "CCSP_ROW_MODE=0" "CCSP_ROW_MODE=2" "CCSP_ROW_MODE=9"; do
    v=$(env $cfg CCSP_SO=$R/tools/abl_try_mode2.so python $R/tools/bench_so.py $BENCH_ARGS --no-cpu-baseline --no-roofline --no-evaluate --no-strict-fp32 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read())['value'])")
    echo "$cfg: $v"
  done
done
