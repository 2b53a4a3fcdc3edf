#!/bin/bash
# Weights that do NOT travel with the repository (round 5: what is pushed to a GPU box went from 106 MB to ~60 MB) and how to get them back.
# There is no network on the build / GPU boxes, so "fetch" means: copy from a directory you give (CCSP_WEIGHTS_DIR), else train on this GPU box.
#
#   weights/qualitative_h256_ref30k.npz    TRAVELS (bench.py --config c2; int8-row storage, 10 MB)
#   weights/qualitative_h256_ref300k.npz   TRAVELS (tests: the reference recipe's final checkpoint, reference-generated goldens chain_q256_ref300k_*)
#   weights/qualitative_h256_ref30k_fp32.npz   dropped: the same checkpoint in fp32 (34 MB).  SAVE_FP32=1 below writes it again.
#   weights/qualitative_h256_trained.npz       dropped: round 2's 12 000-step bench weights (its golden was regenerated with ref30k)
#   weights/qualitative_h256_50k.npz           dropped: round 1's 50 000-step checkpoint (its mixed-overflow case is now chain_q256_ref300k_S3_B32)
#
# usage (GPU box): tools/fetch_or_train_weights.sh ref30k|ref300k|trained|50k
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
want=${1:-ref30k}
case $want in
  ref30k)  out=weights/qualitative_h256_ref30k.npz;  env="TRAIN_RECIPE=reference TRAIN_STEPS=30000"; minutes=30 ;;
  ref300k) out=weights/qualitative_h256_ref300k.npz; env="TRAIN_RECIPE=reference TRAIN_STEPS=300000"; minutes=120 ;;
  trained) out=weights/qualitative_h256_trained.npz; env="TRAIN_OBJECTS=2,8"; minutes=2.2 ;;      # (wall-clock budgeted: ~12 000 steps)
  50k)     out=weights/qualitative_h256_50k.npz;     env="TRAIN_OBJECTS=2,8"; minutes=9 ;;        # (~50 000 steps)
  *) echo "unknown checkpoint $want"; exit 2 ;;
esac
if [ -n "$CCSP_WEIGHTS_DIR" ] && [ -f "$CCSP_WEIGHTS_DIR/$(basename $out)" ]; then
  cp "$CCSP_WEIGHTS_DIR/$(basename $out)" "$R/$out"; echo "copied $out from $CCSP_WEIGHTS_DIR"; exit 0
fi
echo "training $out on this box ($env; ~10 s per 1000 steps on one MI355X; GPU training is not bit-reproducible: goldens made with the original"
echo "checkpoint do not carry over to a re-trained one -- regenerate them with oracle/gen_golden.py)"
cd "$R" && env $env python tools/train_gpu.py $minutes $out          # (SAVE_FP32=1 also writes <out>_fp32.npz)
