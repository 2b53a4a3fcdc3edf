#!/bin/bash
# What one rank costs its HOST: the C2 step of bench.py at N = 1 under shrinking core budgets, two lanes and one lane.
# Per run: samples/s, CPU-seconds (user + sys of the whole process: the lane threads' enqueue work included) per step and busy cores = CPU-seconds
# per second of chain.  The table is what bench.py's launcher rule stands on (select_lanes: one lane per rank below 2.5 usable cores per rank).
# usage: tools/host_budget.sh <tag>   -> gpurun_out/<tag>/host_budget.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
F=$OUT/host_budget.txt
echo "# bench.py --config c2 --steps 3 --warmup 1 at N = 1 (256 graphs x 8 objects, T = 1000 ULA S = 10); host: $(nproc) cpus visible, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" > $F
echo "# pin            lanes  samples/s  host_cpu_s_per_step  busy_cores  ms_per_step" >> $F
run() { # label, taskset prefix, CCSP_LANES
  line=$($2 env CCSP_LANES=$3 python bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-evaluate --no-strict-fp32 2>/dev/null | tail -1)
  echo "$line" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-16s %5s  %9.1f  %19.3f  %10.2f  %11.1f' % ('$1', d['lanes'], d['value'], d['host_cpu_s_per_step'][0], d['host_cores_busy_per_rank'][0], d['ms_per_step']))" >> $F
}
for rep in 1 2; do
  run "unpinned" "" 2
  run "unpinned" "" 1
  run "taskset 0-3" "taskset -c 0-3" 2
  run "taskset 0-1" "taskset -c 0-1" 2
  run "taskset 0-1" "taskset -c 0-1" 1
  run "taskset 0" "taskset -c 0" 2
  run "taskset 0" "taskset -c 0" 1
done
# C4 (energy mode: one lane, one enqueueing thread = the caller's) and the transformer baseline (two lanes) for the record
for c in c4 sd; do
  line=$(python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  echo "$line" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-16s %5s  %9.1f  %19.3f  %10.2f  %11.1f' % ('$c unpinned', d['lanes'], d['value'], d['host_cpu_s_per_step'][0], d['host_cores_busy_per_rank'][0], d['ms_per_step']))" >> $F
done
cat $F
