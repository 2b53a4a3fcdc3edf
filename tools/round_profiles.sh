#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call (tools only):
#   bench lines of c2 / c4 / c5, the RCCL code path on one rank (weights broadcast, gather, MALA global-batch all_reduce),
#   rocprofv3 --kernel-trace --stats of one chain per configuration, PMC passes per configuration.
# usage: tools/round_profiles.sh <tag>      -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd $R
for c in c2 c4 c5; do
  python bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  tail -c 400 $OUT/bench_$c.json; echo
done
python bench.py --gpus 1 --force-dist --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-evaluate > $OUT/rccl_c2_force_dist.log 2>&1
python bench.py --config c4 --gpus 1 --force-dist --mala-global-batch --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/rccl_c4_mala_global_batch.log 2>&1
python bench.py --config c4 --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/c4_replica_for_comparison.log 2>&1
tail -c 300 $OUT/rccl_c2_force_dist.log; echo; tail -c 300 $OUT/rccl_c4_mala_global_batch.log; echo
for c in c2 c4 c5; do
  # (c4: the kernels at full work -- every evaluation recomputed)
  BENCH_ARGS="--config $c --no-evaluate --no-strict-fp32" bash tools/prof_stats.sh ${T}_stats_$c "CCSP_MALA_REUSE=0" > /dev/null 2>&1
  cp $R/gpurun_out/${T}_stats_$c/stats_1.csv $OUT/kernel_stats_$c.csv
  bash tools/pmc_run.sh ${T}_pmc_$c $c > /dev/null 2>&1
  cp $R/gpurun_out/${T}_pmc_$c/summary.txt $OUT/pmc_$c.txt
done
BENCH_ARGS="--config c2 --no-evaluate --no-strict-fp32" bash tools/prof_stats.sh ${T}_stats_c2_1lane "CCSP_LANES=1" > /dev/null 2>&1
cp $R/gpurun_out/${T}_stats_c2_1lane/stats_1.csv $OUT/kernel_stats_c2_1lane.csv
ls -la $OUT
