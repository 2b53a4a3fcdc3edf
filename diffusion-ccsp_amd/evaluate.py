"""Evaluation harness + checkpoint I/O around the sampler: the call pattern and success
accounting of the reference's ``Trainer.evaluate`` (networks/ddpm.py:558-843) and ``Trainer.save/load``
(:496-517), so that a reference checkpoint ``logs/<run>/model-<k>.pt`` drops in and ``solve_csp.py``-style
evaluation produces the same JSON log schema.

Reproduced behaviour
  * per test set: a first pass with batches of up to 100 graphs repeated ``tries[0]`` times (stopping once
    every graph has been solved), then a second pass with batch size 1 repeated ``tries[1]`` times
    (ddpm.py:446-449,591-603);
  * per try: ``result = model.sample(batch, ...)``, poses clamped to [-1, 1] (:620), NaN graphs skipped
    (:644), a graph counts as solved iff the checker returns no violation (:704-713);
  * bookkeeping: ``success_list`` [(graph, try)], ``success_rounds`` {graph: first successful try},
    ``sampling_time`` [(seconds, graphs, seconds per graph)], ``all_failure_modes`` per try,
    ``success_rate`` (solved at try 0) and ``success_rate_top3`` (solved in any try), and
    ``model_ave_sample_time`` from ``model.sample_loop_time`` (:823-836);
  * the log is written to ``<render_dir>/denoised_t=<milestone>.json`` after every try (:779-795).
Not reproduced: rendering, wandb, robot/stability branches (pybullet), rejection sampling.
"""
import json
import os
import time

import numpy as np
import torch

from . import checker, worlds


def save_checkpoint(path, model, step=0):
    """Trainer.save (ddpm.py:496-501): {'step', 'model': GaussianDiffusion.state_dict()}"""
    sd = {k: (v.detach().cpu() if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in model.state_dict().items()}
    torch.save({'step': int(step), 'model': sd}, path)


def load_checkpoint(path, model):
    """Trainer.load (ddpm.py:503-514); returns the stored step"""
    data = torch.load(path, map_location='cpu')
    model.load_state_dict(data['model'])
    return int(data.get('step', 0))


def create_sampler(input_mode='qualitative', hidden_dim=256, timesteps=1000, EBM='ULA', samples_per_step=10,
                   step_sizes='2*self.betas', normalize=True, energy_wrapper=False, ebm_per_steps=1, device='cuda',
                   checkpoint=None):
    """the model-construction part of train_utils.create_trainer (train_utils.py:266-286) with the reference's
    flag names and defaults (-hidden_dim 256 -timesteps 1000 -EBM ... -samples_per_step 10 -step_sizes ...)"""
    from . import ComposedEBMDenoiseFn, ConstraintDiffuser, GaussianDiffusion
    if EBM in ('HMC', 'MALA'):                       # train_utils.py:115-116
        energy_wrapper = True
    mode_key = 'robot_box' if 'robot' in input_mode else ('diffuse_pairwise' if input_mode not in worlds.MODE_DIMS else input_mode)
    dims = worlds.MODE_DIMS[mode_key]
    den = ConstraintDiffuser(dims=dims, hidden_dim=hidden_dim, EBM=EBM, input_mode=input_mode, normalize=normalize,
                             energy_wrapper=energy_wrapper, device=device, verbose=False)
    den.reset_parameters(0)
    fn = ComposedEBMDenoiseFn(den, ebm_per_steps) if (EBM and energy_wrapper) else den
    gd = GaussianDiffusion(fn, timesteps=timesteps, EBM=EBM, samples_per_step=samples_per_step, step_sizes=step_sizes)
    if checkpoint is not None:
        load_checkpoint(checkpoint, gd)
    return gd


class Evaluator(object):
    """test_sets: {name (e.g. number of objects): [graph dicts with x, edge_index, edge_attr, mask, world_dims]}"""

    def __init__(self, model, test_sets, render_dir, batch_sizes=(100, 1), check_fn=None, device=None):
        self.model = model
        self.test_sets = test_sets
        self.render_dir = render_dir
        self.batch_sizes = batch_sizes
        self.check_fn = check_fn or checker.evaluate_graph
        self.device = device if device is not None else getattr(model, 'device', 'cpu')
        self.num_test_samples = 0 if not test_sets else len(list(test_sets.values())[0])
        os.makedirs(render_dir, exist_ok=True)

    def _loaders(self, graphs):
        out = []
        for bs in self.batch_sizes:
            out.append([(list(range(i, min(i + bs, len(graphs)))), worlds.collate(graphs[i:i + bs])) for i in range(0, len(graphs), bs)])
        return out

    def evaluate(self, milestone, tries=(10, 0), return_history=False, save_log=False, run_all=False, resume_eval=False,
                 seed=None, **kwargs):
        json_name = os.path.join(self.render_dir, 'denoised_t=%s.json' % milestone)
        log = {}
        if resume_eval and os.path.isfile(json_name):
            log = json.load(open(json_name, 'r'))
        call = 0
        for i, graphs in self.test_sets.items():
            key = str(i)
            if resume_eval and key in log:
                continue
            success_list, succeeded, success_rounds, sampling_time, failure_modes = [], [], {}, [], {}
            percentage = 0.0
            n_total = len(graphs)
            for m, loader in enumerate(self._loaders(graphs)):
                count = 0
                for idxs, batch_np in loader:
                    for k in range(tries[m]):
                        if m == 0 and len(succeeded) == n_total:
                            break
                        if m == 1:
                            k += tries[0]
                        batch = batch_np.to_torch(self.device)
                        failure_modes.setdefault(str(k), {})
                        start = time.time()
                        result = self.model.sample(batch, return_history=return_history,
                                                   **({} if seed is None else {'seed': seed + call}), **kwargs)
                        passed = time.time() - start
                        call += 1
                        poses = result[0] if return_history else result
                        poses = poses.detach().cpu().numpy().clip(-1.0, 1.0)                  # ddpm.py:620
                        sampling_time.append((passed, len(idxs), passed / len(idxs)))
                        gid = np.asarray(batch_np.batch)
                        ei, ea = np.asarray(batch_np.edge_index), np.asarray(batch_np.edge_attr)
                        for local, j in enumerate(idxs):
                            count += 1
                            nodes = np.nonzero(gid == local)[0]
                            feats = np.concatenate([np.asarray(batch_np.x)[nodes, :2], poses[nodes]], axis=1)
                            if np.isnan(feats).any():                                          # ddpm.py:644
                                continue
                            if m == 0 and j in succeeded:
                                continue
                            n0 = int(nodes[0])
                            sel = np.nonzero(gid[ei[0]] == local)[0]
                            given = [(worlds.QUALITATIVE_CONSTRAINTS[int(ea[e])], int(ei[0, e]) - n0, int(ei[1, e]) - n0)
                                     for e in sel if 0 <= int(ea[e]) < len(worlds.QUALITATIVE_CONSTRAINTS)]
                            evaluations = self.check_fn(feats, batch_np.world_dims[local], given)
                            if len(evaluations) == 0:
                                success_list.append((j, k))
                                if j not in success_rounds:
                                    succeeded.append(j)
                                    success_rounds[j] = k
                            else:
                                failure_modes[str(k)][str(j)] = [list(e) for e in evaluations]
                        percentage = len(succeeded) / max(1, n_total)
                        if m == 0 and count != 0:
                            self._summarize(key, success_list, n_total, succeeded, success_rounds, log)
                        log.setdefault(key, {})
                        log[key].update({'success_rounds': {str(a): b for a, b in success_rounds.items()},
                                         'sampling_time': sampling_time, 'all_failure_modes': failure_modes,
                                         'eval_tries': k, 'visualize': False})
                        with open(json_name, 'w') as f:
                            json.dump(log, f)
                if m == 0 and count != 0:
                    self._summarize(key, success_list, n_total, succeeded, success_rounds, log, final=True)
            with open(json_name, 'w') as f:
                json.dump(log, f)
            if not run_all and percentage == 0:          # ddpm.py:797-800: stop at the first test set nothing solves
                break
        return log

    def _summarize(self, key, success_list, count, succeeded, success_rounds, log, final=False):
        """summarize_success_rate (ddpm.py:823-843): top-1 = share of graphs solved at try 0, top-k ('success_rate_top3') = share
        solved at any try, average sampling time per graph from the model's sample_loop_time window; the closing call of a
        test set (send_wandb=True in the reference) also empties that window (:837)"""
        top1 = round(len([s for s in success_rounds.values() if s == 0]) / count, 3)
        topk = round(len(succeeded) / count, 3)
        times = getattr(self.model, 'sample_loop_time', None) or [0.0]
        log.setdefault(key, {})
        log[key].update({'success': [list(s) for s in success_list], 'success_rate': top1, 'success_rate_top3': topk,
                         'model_ave_sample_time': sum(times) / len(times) / count})
        if final and hasattr(self.model, 'sample_loop_time'):
            self.model.sample_loop_time = []
