#!/bin/bash
# What one rank costs its HOST, and what happens when the host is short of cores.  bench.py --config c2 at N = 1:
#   (1) unconstrained and under `taskset`, two lanes and one lane: samples/s, CPU-seconds per step (user + sys of every thread of the process: the
#       lane threads' enqueue work and the waiting Python thread's spin included), busy cores = CPU-seconds per second of chain;
#   (2) with K busy-spinning competitor processes inside the same cgroup (the GPU boxes grant 16 cores): K = 7 x the busy cores of one rank is what
#       seven OTHER ranks of an 8-rank run put next to this one -- the only way to see 8-rank host contention with one GPU.
# The table is what bench.py's launcher rule stands on (select_lanes).   usage: tools/host_budget.sh <tag>   -> gpurun_out/<tag>/host_budget.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
F=$OUT/host_budget.txt
echo "# bench.py --config c2 --steps 3 --warmup 1 at N = 1 (256 graphs x 8 objects, T = 1000 ULA S = 10); host: $(nproc) cpus visible, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" > $F
echo "# setting                     lanes  samples/s  host_cpu_s_per_step  busy_cores  ms_per_step  affinity_cpus" >> $F
run() { # label, prefix command, CCSP_LANES
  line=$($2 env CCSP_LANES=$3 python bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-evaluate --no-strict-fp32 2>/dev/null | tail -1)
  echo "$line" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-28s %5s  %9.1f  %19.3f  %10.2f  %11.1f  %13s' % ('$1', d['lanes'], d['value'], d['host_cpu_s_per_step'][0], d['host_cores_busy_per_rank'][0], d['ms_per_step'], d['host_budget'].get('affinity_cpus_at_end')))" >> $F
}
run "unpinned" "" 2
run "unpinned" "" 1
run "taskset -c 0-1" "taskset -c 0-1" 2
run "taskset -c 0" "taskset -c 0" 2
run "taskset -c 0" "taskset -c 0" 1
# competitors: K processes that spin (what the other ranks' enqueue threads and waiting threads are to this rank)
for K in 8 13 16 19 24; do
  pids=""
  for i in $(seq $K); do ( while :; do :; done ) & pids="$pids $!"; done
  run "with $K spinning processes" "" 2
  run "with $K spinning processes" "" 1
  kill $pids 2>/dev/null; wait $pids 2>/dev/null
done
# C4 (energy mode: one lane, the caller's thread enqueues) and the transformer baseline (two lanes) for the record
for c in c4 sd; do
  line=$(python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  echo "$line" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-28s %5s  %9.1f  %19.3f  %10.2f  %11.1f' % ('$c unpinned', d['lanes'], d['value'], d['host_cpu_s_per_step'][0], d['host_cores_busy_per_rank'][0], d['ms_per_step']))" >> $F
done
cat $F
