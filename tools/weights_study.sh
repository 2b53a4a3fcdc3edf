#!/bin/bash
# (tools only) what the int8 storage of the bench weights costs, and a longer-trained checkpoint for the overflow fixture:
# trains with tools/train_gpu.py (checkpoint at 12k steps + final), each saved int8-row-quantised AND fp32, then the solved
# sweep of diffusion-ccsp_amd/checker.py on both storages of the same weights.  usage: tools/weights_study.sh <tag> <minutes>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT /tmp/w
cd $R
SAVE_FP32=1 CKPT_KSTEPS=12 python tools/train_gpu.py $2 /tmp/w/new.npz > $OUT/train.log 2>&1
tail -3 $OUT/train.log
for f in new_12k new_12k_fp32 new new_fp32; do
  echo "== $f" | tee -a $OUT/solved.txt
  SWEEP_OBJECTS=3,6,8 python tools/solved_sweep.py 256 /tmp/w/$f.npz 2>/dev/null | tee -a $OUT/solved.txt
done
cp /tmp/w/new.npz $OUT/qualitative_h256_long.npz
cp /tmp/w/new_12k.npz $OUT/qualitative_h256_12k_new.npz
ls -la $OUT
