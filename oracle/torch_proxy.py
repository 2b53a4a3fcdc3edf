"""TEST / BASELINE INFRASTRUCTURE ONLY -- a cost-faithful PyTorch-CPU *proxy* of the reference
sampler, written from SURVEY.md Appendix A (not from the reference's files).

Purpose: the reference's Python cannot travel to the GPU box, so bench.py's ``cpu_baseline`` leg
times this proxy on the box's host cores instead ("kind": "port").  To make that number stand for
"the reference's PyTorch-CPU path" the proxy keeps the reference's op sequence *and its
redundancies*: the geometry encoder runs on every evaluation, the edge subset of every constraint
type is found with ``where`` + a host round trip, the ``[E_i, 5H]`` input is materialised with
``cat``, the time MLP runs once per edge row, outputs are accumulated with ``scatter_add_`` /
``bincount``.  tests/test_oracle_golden.py checks its outputs against the reference-generated
golden vectors; oracle/gen_golden.py's companion check (oracle/certify_proxy.py) times it next to
the imported reference in the build container.

The product (diffusion-ccsp_amd/) never imports this file.
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F


def _lin(x, w, b):
    return F.linear(x, w, b)


class ProxyDiffuser(object):
    def __init__(self, weights, dims, hidden_dim, n_types, normalize=True):
        # parameters like the reference module's (requires_grad): in energy mode autograd then records the same graph
        self.W = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32).clone().requires_grad_(True) for k, v in weights.items()}
        self.dims, self.H, self.C, self.normalize = dims, hidden_dim, n_types, normalize
        self.grasp = len(dims) == 3
        self.P = dims[-1][0]

    def _mlp2(self, name, x, last_act=True):
        h = F.silu(_lin(x, self.W[name + '.0.weight'], self.W[name + '.0.bias']))
        h = _lin(h, self.W[name + '.2.weight'], self.W[name + '.2.bias'])
        return F.silu(h) if last_act else h

    def time_mlp(self, t_rows):
        half = self.H // 2
        emb = math.log(10000) / (half - 1)
        emb = torch.exp(torch.arange(half) * -emb)
        emb = t_rows[:, None] * emb[None, :]
        emb = torch.cat((emb.sin(), emb.cos()), dim=-1)
        h = F.mish(_lin(emb, self.W['time_mlp.1.weight'], self.W['time_mlp.1.bias']))
        return _lin(h, self.W['time_mlp.3.weight'], self.W['time_mlp.3.bias'])

    def energy_grad(self, poses_in, batch, t):
        """energy mode (ComposedEBMDenoiseFn, denoise_fn.py:57-83,373-375,539-548): E = sum over (edge, slot) |o - pose|^2 and
        dE/dposes through autograd, like the reference (every call is a forward AND a backward pass)"""
        with torch.enable_grad():
            poses = poses_in.detach().clone().requires_grad_(True)
            energy = self.__call__(poses, batch, t, energy=True)
            grad = torch.autograd.grad(energy, poses)[0]
        return grad.detach(), energy.detach()

    def __call__(self, poses_in, batch, t, energy=False):
        H, P = self.H, self.P
        x = batch.x.clone()
        geoms_emb = self._mlp2('geom_encoder', x[:, :self.dims[0][2]])
        poses_emb = self._mlp2('pose_encoder', poses_in)
        grasp_emb = self._mlp2('grasp_encoder', x[:, self.dims[1][1]:self.dims[1][2]]) if self.grasp else None
        edge_index = batch.edge_index.T
        out = torch.zeros_like(poses_in)
        cnt = torch.zeros_like(poses_in[:, 0])
        total_energy = 0
        for i in range(self.C):
            edges = torch.where(batch.edge_attr == i)[0]
            edges = edges.detach().cpu().numpy()
            if edges.shape[0] == 0:
                continue
            args = torch.stack([edge_index[edges][:, 0], edge_index[edges][:, 1]], dim=1)
            # [E_i, 1] rows through the time MLP as a 3-D batch, then [:, 0] -- the shape the reference feeds (denoise_fn.py:328)
            temb = self.time_mlp(t.unsqueeze(0).expand(edges.shape[0], *t.shape))[:, 0]
            parts = [geoms_emb[args].reshape(len(edges), -1), poses_emb[args].reshape(len(edges), -1), temb]
            if self.grasp:
                parts = [grasp_emb[args[:, 0]]] + parts
            inputs = torch.cat(parts, dim=-1)
            h = F.silu(_lin(inputs, self.W['mlps.%d.0.weight' % i], self.W['mlps.%d.0.bias' % i]))
            h = torch.stack([h[:, :H], h[:, H:]], dim=1)
            o = self._mlp2('pose_decoder', h, last_act=False)
            if energy:
                total_energy = total_energy + ((o - poses_in[args]) ** 2).sum()
                continue
            flat = args.reshape(-1)
            o = o.reshape(-1, P)
            out.scatter_add_(0, flat.unsqueeze(-1).expand(o.shape), o)
            cnt += torch.bincount(flat, minlength=out.shape[0])
        if energy:
            return total_energy
        if self.normalize:
            out /= torch.sqrt(cnt.unsqueeze(-1))
        m = batch.mask.bool()
        out[m] = x[:, -P:][m]
        return out


class ProxyStructDiffuser(ProxyDiffuser):
    """cost-faithful port of the StructDiffusion baseline (reference denoise_fn.py:391-451, transformer.py:43-82), op for op:
    the geometry and pose encoders on every call, the time MLP on [N, 1, H] rows, a Python loop over the graphs that slices each sequence, adds the
    positional encoding, applies ln_pre, pads to 8 tokens and builds the FLOAT pad mask (+1.0 on padded rows / columns; `[-0:]` marks everything of
    an unpadded graph), the masks repeated graph-major while nn.MultiheadAttention reads them head-major, four pre-LN blocks
    (x + MHA(ln_1 x); x + ln_2(c_proj(QuickGELU(c_fc x)))), ln_post, the last H channels, a second Python loop that gathers the real tokens, the
    pose decoder, the mask fill.  nn.MultiheadAttention.forward in eval mode is F.multi_head_attention_forward with need_weights=True (the
    reference keeps the attention weights of every block): called here with the same arguments."""
    L, HEADS, LAYERS = 8, 2, 4

    def __init__(self, weights, dims, hidden_dim, n_types=0, normalize=True):
        ProxyDiffuser.__init__(self, weights, dims, hidden_dim, n_types, normalize)
        self.width = hidden_dim * (3 if self.grasp else 2)
        pe = torch.zeros(5000, self.width)
        position = torch.arange(0, 5000).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, self.width, 2) * -(math.log(10000.0) / self.width))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.unsqueeze(0)
        # the transformer is made of stock torch modules in the reference (nn.LayerNorm, nn.MultiheadAttention, nn.Linear): the same stock modules here,
        # so that what a module call costs in Python -- 256 ln_pre calls per evaluation, MultiheadAttention's fast-path checks -- is in the baseline too
        nn = torch.nn

        def ln(name):
            m = nn.LayerNorm(self.width)
            m.weight, m.bias = nn.Parameter(self.W[name + '.weight'].detach().clone()), nn.Parameter(self.W[name + '.bias'].detach().clone())
            return m.eval()

        def lin(name, n_in, n_out):
            m = nn.Linear(n_in, n_out)
            m.weight, m.bias = nn.Parameter(self.W[name + '.weight'].detach().clone()), nn.Parameter(self.W[name + '.bias'].detach().clone())
            return m.eval()
        self.ln_pre, self.ln_post = ln('ln_pre'), ln('ln_post')
        self.blocks = []
        for l in range(self.LAYERS):
            pre = 'transformer.resblocks.%d.' % l
            attn = nn.MultiheadAttention(self.width, self.HEADS)
            attn.in_proj_weight = nn.Parameter(self.W[pre + 'attn.in_proj_weight'].detach().clone())
            attn.in_proj_bias = nn.Parameter(self.W[pre + 'attn.in_proj_bias'].detach().clone())
            attn.out_proj.weight = nn.Parameter(self.W[pre + 'attn.out_proj.weight'].detach().clone())
            attn.out_proj.bias = nn.Parameter(self.W[pre + 'attn.out_proj.bias'].detach().clone())
            self.blocks.append(dict(ln_1=ln(pre + 'ln_1'), ln_2=ln(pre + 'ln_2'), attn=attn.eval(), c_fc=lin(pre + 'mlp.c_fc', self.width, 4 * self.width),
                                    c_proj=lin(pre + 'mlp.c_proj', 4 * self.width, self.width), dropout=nn.Dropout(p=0.1).eval()))

    def __call__(self, poses_in, batch, t, energy=False):
        from einops import rearrange, repeat          # (the reference's own helpers: ~30 us of Python per call, 256 calls per evaluation -- part of its cost)
        H, P, W = self.H, self.P, self.W
        x = batch.x.clone()
        geoms_emb = self._mlp2('geom_encoder', x[:, :self.dims[0][2]])
        poses_emb = self._mlp2('pose_encoder', poses_in)
        time_emb = self.time_mlp(t.unsqueeze(0).expand(geoms_emb.shape[0], *t.shape))[:, 0]
        poses_emb = poses_emb + time_emb
        obj_emb = torch.cat([geoms_emb, poses_emb], dim=-1)
        if self.grasp:
            obj_emb = torch.cat([self._mlp2('grasp_encoder', x[:, self.dims[1][1]:self.dims[1][2]]), obj_emb], dim=-1)
        sequences, attn_masks, indices = [], [], []
        for j in range(batch.batch.max().item() + 1):
            seq = obj_emb[batch.batch == j]
            pe = self.pe[:, :seq.shape[0], :]
            if hasattr(batch, 'shuffled'):
                pe = pe[:, batch.shuffled[batch.batch == j], :]
            seq += rearrange(pe, 'b n c -> (b n) c')
            xs = self.ln_pre(seq)
            padding_len = self.L - xs.shape[0]
            indices.append(xs.shape[0])
            sequences.append(F.pad(xs, (0, 0, 0, padding_len), 'constant', 0))
            attn_mask = torch.zeros(self.L, self.L)
            attn_mask[:, -padding_len:] = True
            attn_mask[-padding_len:, :] = True
            attn_masks.append(attn_mask)
        xs = torch.stack(sequences, dim=1)                                  # [8, B, width]
        masks = torch.stack(attn_masks)
        masks = repeat(masks, 'b l1 l2 -> (repeat b) l1 l2', repeat=self.HEADS)
        weights = None
        for blk in self.blocks:
            y = blk['ln_1'](xs)
            attn, attn_w = blk['attn'](y, y, y, attn_mask=masks)
            weights = attn_w.unsqueeze(1) if weights is None else torch.cat([weights, attn_w.unsqueeze(1)], dim=1)
            xs = xs + attn
            h = blk['c_fc'](xs)
            h = h * torch.sigmoid(1.702 * h)
            h = blk['c_proj'](h)
            xs = xs + blk['ln_2'](blk['dropout'](h))
        xs = self.ln_post(xs)
        xs = xs[:, :, -H:]
        poses_out = torch.cat([xs[:indices[j], j] for j in range(len(indices))], dim=0)
        poses_out = self._mlp2('pose_decoder', poses_out, last_act=False)
        m = batch.mask.bool()
        poses_out[m] = x[:, -P:][m]
        return poses_out


def cosine_schedule(T):
    steps = T + 1
    xs = np.linspace(0, steps, steps)
    ac = np.cos(((xs / steps) + 0.008) / 1.008 * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - ac[1:] / ac[:-1], 0, 0.999)
    alphas = 1 - betas
    acp = np.cumprod(alphas)
    prev = np.append(1.0, acp[:-1])
    f = lambda a: torch.tensor(a, dtype=torch.float32)  # noqa: E731
    pv = betas * (1 - prev) / (1 - acp)
    return dict(betas=f(betas), a=f(np.sqrt(1 / acp)), b=f(np.sqrt(1 / acp - 1)), kappa=f(np.sqrt(1 / (1 - acp))),
                lv=f(np.log(np.maximum(pv, 1e-20))), c1=f(betas * np.sqrt(prev) / (1 - acp)),
                c2=f((1 - prev) * np.sqrt(alphas) / (1 - acp)))


@torch.no_grad()
def timestep(model, sch, batch, x, t, S, noise_fn):
    """one timestep of the ULA chain: ancestral step + S Langevin steps + mask reset"""
    tt = torch.full((1,), t, dtype=torch.long)
    eps = model(x, batch, tt)
    x0 = sch['a'][t] * x - sch['b'][t] * eps
    mean = sch['c1'][t] * x0 + sch['c2'][t] * x
    x = mean + (0 if t == 0 else 1) * (0.5 * sch['lv'][t]).exp() * noise_fn()
    ss = 2 * sch['betas'][t]
    std = (2 * ss) ** .5
    for _ in range(S):
        grad = -model(x, batch, tt) * sch['kappa'][t]
        x = x + grad * ss + noise_fn() * std
    m = batch.mask.bool()
    x[m] = batch.x[:, model.dims[-1][1]:model.dims[-1][2]][m]
    return x


def mala_timestep(model, sch, batch, x, t, S, noise_fn, uniform_fn):
    """one timestep of the MALA chain on an energy-mode model: ancestral step with epsilon = dE/dposes, then S inner steps of
    AnnealedMALASampler.sample_step (ddpm.py:1013-1047) -- the gradient at x plus the energy function at x and at the proposal,
    each a forward + backward pass as in the reference (energy_function -> neg_logp_unnorm -> model.forward(tag='EBM'))"""
    tt = torch.full((1,), t, dtype=torch.long)
    eps, _ = model.energy_grad(x, batch, tt)
    x0 = sch['a'][t] * x - sch['b'][t] * eps
    mean = sch['c1'][t] * x0 + sch['c2'][t] * x
    x = mean + (0 if t == 0 else 1) * (0.5 * sch['lv'][t]).exp() * noise_fn()
    ss = 2 * sch['betas'][t]
    std = (2 * ss) ** .5
    for _ in range(S):
        g, _ = model.energy_grad(x, batch, tt)
        grad = -g * sch['kappa'][t]
        mu = x + grad * ss
        x_hat = mu + noise_fn() * std
        logp_x = -model.energy_grad(x, batch, tt)[1] * sch['kappa'][t]
        logp_x_hat = -model.energy_grad(x_hat, batch, tt)[1] * sch['kappa'][t]
        dist = torch.distributions.Normal(mu, torch.ones_like(x) * std)
        logp_accept = logp_x_hat - logp_x + dist.log_prob(x).sum(1) - dist.log_prob(x_hat).sum(1)
        accept = (uniform_fn() < torch.exp(logp_accept)).float()
        x = accept[:, None] * x_hat + (1 - accept[:, None]) * x
    m = batch.mask.bool()
    x[m] = batch.x[:, model.dims[-1][1]:model.dims[-1][2]][m]
    return x


@torch.no_grad()
def sample(model, batch, T, S, noise_fn):
    sch = cosine_schedule(T)
    gt = batch.x[:, model.dims[-1][1]:model.dims[-1][2]]
    x = 0.5 * noise_fn()
    m = batch.mask.bool()
    x[m] = gt[m]
    for t in reversed(range(T)):
        x = timestep(model, sch, batch, x, t, S, noise_fn)
    return x


def time_baseline(weights, dims, hidden_dim, n_types, batch, T=1000, S=10, n_timesteps=3, budget_s=25.0,
                  thread_candidates=(8, 16, 32, 64), sampler='ULA', model_kind='Diffusion-CCSP'):
    """times full timesteps (1+S evaluations each) of the proxy on the host cores and extrapolates x T.
    PyTorch-CPU does not scale to every core of a large host on these small matrices (128 threads were
    4x slower than 8 on the GPU box), so one timestep is timed per candidate thread count first and the
    fastest setting is used -- the baseline is the best the CPU path does on this box.
    Returns dict(samples_per_s, sec_per_timestep, cores, sample)."""
    model = (ProxyStructDiffuser if model_kind == 'StructDiffusion' else ProxyDiffuser)(weights, dims, hidden_dim, n_types)
    sch = cosine_schedule(T)
    g = torch.Generator().manual_seed(0)
    N, P = batch.x.shape[0], dims[-1][0]
    noise_fn = lambda: torch.randn((N, P), generator=g)  # noqa: E731
    uniform_fn = lambda: torch.rand((N,), generator=g)  # noqa: E731
    mala = sampler == 'MALA'
    if mala:
        step = lambda xx, t, s: mala_timestep(model, sch, batch, xx, t, s, noise_fn, uniform_fn)  # noqa: E731
        evals = lambda s: 1 + 3 * s  # noqa: E731  (forward + backward each)
    else:
        step = lambda xx, t, s: timestep(model, sch, batch, xx, t, s, noise_fn)  # noqa: E731
        evals = lambda s: 1 + s  # noqa: E731
    x = 0.5 * noise_fn()
    max_threads = torch.get_num_threads()
    cands = sorted(set(min(c, max_threads) for c in thread_candidates))
    step(x, T // 2, 0 if mala else 1)                             # warm-up (allocator, threads)
    t_all = time.time()
    trial = {}
    for c in cands:
        torch.set_num_threads(c)
        step(x, T // 2, 0)
        t0 = time.time()
        step(x, T // 2, 1 if mala else 2)                         # 4 (MALA) / 3 evaluations
        trial[c] = (time.time() - t0) / (4.0 if mala else 3.0)
        if time.time() - t_all > budget_s * 0.5:
            break
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    done, t0 = 0, time.time()
    for k in range(n_timesteps):
        x = step(x, T - 1 - k, S)
        done += 1
        if time.time() - t_all > budget_s:
            break
    dt = (time.time() - t0) / done
    torch.set_num_threads(max_threads)
    n_graphs = int(batch.num_graphs) if hasattr(batch, 'num_graphs') else 1
    return dict(samples_per_s=n_graphs / (dt * T), sec_per_timestep=dt, cores=best,
                sec_per_eval_by_threads={int(k): float(v) for k, v in trial.items()},
                sample='%d full %s timesteps (%d network evaluations%s) of the %d-graph batch at %d threads (best of %s), '
                       'extrapolated x%d/%d' % (done, sampler, done * evals(S), ', forward + backward each' if mala else '', n_graphs, best,
                                                list(trial.keys()), T, done))
