#!/bin/bash
# determinism soak of every configuration with the shipped kernel selection, plus lanes that run the mid-size row GEMM (MODE 6)
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in c2 c4 c5 small; do python $R/tools/soak.py 12 $c 2>&1 | tail -1; done
CCSP_ROW_MODE=6 python $R/tools/soak.py 8 c2 2>&1 | tail -1
for g in 96 192; do python - $g <<'PY'
import os, sys, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from bench import load_weights
from diffusion_ccsp_amd import ConstraintDiffuser, GaussianDiffusion, worlds
g = int(sys.argv[1])
dev = torch.device("cuda:0")
den = ConstraintDiffuser(dims=worlds.MODE_DIMS["qualitative"], hidden_dim=256, input_mode="qualitative", EBM="ULA", device=dev, verbose=False)
den.load_state_dict(load_weights(os.path.join(R, "tests/golden/weights_qualitative_h256.npz")))
gd = GaussianDiffusion(den, timesteps=1000, EBM="ULA", samples_per_step=10)
b = worlds.qualitative_batch(g, 8, seed=5).to_torch(dev)
ref = gd.sample(b, seed=42).clone()
ok = all(torch.equal(gd.sample(b, seed=42), ref) for _ in range(6))
print("soak %d graphs (lanes on the mid-size row GEMM):" % g, ok, bool(torch.isfinite(ref).all()))
PY
done
