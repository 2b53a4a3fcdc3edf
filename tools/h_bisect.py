"""which stage breaks at a hidden_dim outside 64 / 128 / 256?  (tools only)  usage: python tools/h_bisect.py H"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
from diffusion_ccsp_amd import ConstraintDiffuser, worlds
H = int(sys.argv[1])
dev = torch.device('cuda:0')
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (1 + np.abs(b).max()))
den = ConstraintDiffuser(dims=worlds.MODE_DIMS['qualitative'], hidden_dim=H, input_mode='qualitative', EBM='ULA', device=dev, verbose=False)
den.reset_parameters(3)
W = {k: v.cpu().numpy() for k, v in den.state_dict().items()}
b = worlds.qualitative_batch(5, 6, seed=H).to_torch()
om = oracle.OracleModel(W, worlds.MODE_DIMS['qualitative'], H, 13, timesteps=20, samples_per_step=2)
og = om.graph(b)
rng = np.random.default_rng(1)
poses = (rng.standard_normal((b.x.shape[0], 4)) * 0.7).astype(np.float32)
pe = den.pose_encoder(torch.from_numpy(poses)).cpu().numpy()
# numpy encoder
def silu(v): return v / (1 + np.exp(-v))
def enc(x, name):
    h = silu(x @ W[name + '.0.weight'].T + W[name + '.0.bias'])
    return silu(h @ W[name + '.2.weight'].T + W[name + '.2.bias'])
print('H', H, 'mode', os.environ.get('CCSP_MMA', 'default'))
print(' pose_encoder op   ', rel(pe, enc(poses, 'pose_encoder')))
ge = den.geom_encoder(b.x[:, :2]).cpu().numpy()
print(' geom_encoder op   ', rel(ge, enc(b.x[:, :2].numpy(), 'geom_encoder')))
te = den.time_mlp(torch.tensor([7.0])).cpu().numpy()
eo = den.edge_outputs(torch.from_numpy(poses), b, 7).cpu().numpy()
eo_o = og.edge_outputs(poses, 7)
m = np.isfinite(eo_o).all(axis=(1, 2))
print(' edge outputs      ', rel(eo[m], eo_o[m]), 'edges', int(m.sum()))
ea = b.edge_attr.numpy()
for t in sorted(set(ea[m].astype(int))):
    s = m & (ea == t)
    print('   type %2d  %.3g' % (t, rel(eo[s], eo_o[s])))
out = den(torch.from_numpy(poses), b, torch.tensor([7]), eval=True).cpu().numpy()
print(' denoise           ', rel(out, og.denoise(poses, 7)))
# energy mode
from diffusion_ccsp_amd import ConstraintDiffuser as CD
dene = CD(dims=worlds.MODE_DIMS['diffuse_pairwise'], hidden_dim=H, input_mode='diffuse_pairwise', EBM='MALA', energy_wrapper=True, device=dev, verbose=False)
dene.reset_parameters(5)
be = worlds.triangular_batch(3, 7, seed=H).to_torch()
ome = oracle.OracleModel({k: v.cpu().numpy() for k, v in dene.state_dict().items()}, worlds.MODE_DIMS['diffuse_pairwise'], H, 2, timesteps=20, samples_per_step=2, energy_wrapper=True)
oge = ome.graph(be)
poses = (rng.standard_normal((be.x.shape[0], 4)) * 0.5).astype(np.float32)
grad, E = dene(torch.from_numpy(poses), be, torch.tensor([1]), tag='EBM')
want, Ew = oge.energy_grad(poses, 1)
g = grad.cpu().numpy()
print(' energy            ', abs(float(E) - Ew) / (1 + abs(Ew)))
print(' gradient          ', rel(g, want), 'per column', [float('%.2g' % (np.abs(g[:, c] - want[:, c]).max() / (1 + np.abs(want).max()))) for c in range(4)])
print(' worst nodes       ', np.argsort(-np.abs(g - want).max(axis=1))[:6].tolist(), 'of', g.shape[0])
