"""evaluation harness (next row 8f-2): try/loader pattern, success accounting and JSON log schema of
Trainer.evaluate, with a scripted sampler on CPU and the real sampler on the GPU; checkpoint round trip"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import weights, worlds
from diffusion_ccsp_amd import evaluate


class ScriptedSampler(object):
    """returns the ground-truth poses from try `good_from` on, noise before; graph 1 is never solved"""

    def __init__(self, good_from=1):
        self.calls, self.good_from, self.sample_loop_time = 0, good_from, []

    def sample(self, batch, return_history=False, **kw):
        gt = batch.x[:, 2:6].clone()
        k = self.calls
        self.calls += 1
        self.sample_loop_time.append(0.01)
        if k < self.good_from:
            gt = torch.zeros_like(gt)                     # everything piled at the centre: collisions
        else:
            bad_nodes = (torch.as_tensor(batch.batch) == 1)
            gt[bad_nodes] = 0.0
        return (gt, [gt]) if return_history else gt


def make_sets(n=4, objs=3, seed=2):
    rng = np.random.default_rng(seed)
    gs = []
    for _ in range(n):
        wd = worlds.sample_qualitative_world(rng, objs)
        gs.append(worlds.encode_qualitative(wd['nodes'], wd['constraints']))
    return {objs: gs}


def test_accounting_and_log_schema(tmp_path):
    ev = evaluate.Evaluator(ScriptedSampler(good_from=1), make_sets(), str(tmp_path), device='cpu')
    log = ev.evaluate(7, tries=(3, 0))
    rec = log['3']
    # try 0 solves nothing; try 1 solves graphs 0, 2, 3; graph 1 never
    assert rec['success_rate'] == 0.0 and rec['success_rate_top3'] == 0.75
    assert rec['success_rounds'] == {'0': 1, '2': 1, '3': 1}
    assert sorted(tuple(s) for s in rec['success']) == [(0, 1), (2, 1), (3, 1)]      # solved graphs are not re-checked
    assert rec['eval_tries'] == 2 and len(rec['sampling_time']) == 3
    assert set(rec) >= {'success', 'success_rate', 'success_rate_top3', 'model_ave_sample_time', 'success_rounds',
                        'sampling_time', 'all_failure_modes', 'eval_tries', 'visualize'}
    assert '1' in rec['all_failure_modes']['2']
    on_disk = json.load(open(os.path.join(str(tmp_path), 'denoised_t=7.json')))
    assert on_disk['3']['success_rate_top3'] == 0.75
    # all graphs solved at try 0 -> later tries are skipped
    ev2 = evaluate.Evaluator(ScriptedSampler(good_from=0), {3: [make_sets()[3][0]]}, str(tmp_path), device='cpu')
    log2 = ev2.evaluate(8, tries=(5, 0))
    assert log2['3']['success_rate'] == 1.0 and len(log2['3']['sampling_time']) == 1
    # second loader (batch size 1) continues the try numbering
    ev3 = evaluate.Evaluator(ScriptedSampler(good_from=99), make_sets(2), str(tmp_path), device='cpu')
    log3 = ev3.evaluate(9, tries=(1, 2), run_all=True)
    assert log3['3']['eval_tries'] == 2 and len(log3['3']['sampling_time']) == 1 + 2 * 2


def test_summary_matches_the_reference_method(tmp_path):
    """tests/golden/evaluate_summary.json: the reference's own Trainer.summarize_success_rate (ddpm.py:823-843) run on scripted
    bookkeeping states by oracle/gen_golden.py -- top-1 / top-k rounding, the per-graph sampling time, the record's keys, and
    the emptied sample_loop_time window after the closing call of a test set"""
    cases = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'evaluate_summary.json')))
    assert len(cases) >= 10

    class M(object):
        pass
    for c in cases:
        m = M()
        m.sample_loop_time = list(c['sample_loop_time'])
        ev = evaluate.Evaluator(m, {}, str(tmp_path), device='cpu')
        log = {}
        ev._summarize(str(c['i']), [tuple(s) for s in c['success_list']], c['count'], list(c['succeeded']),
                      {int(k): v for k, v in c['success_rounds'].items()}, log, final=c['final'])
        got = json.loads(json.dumps(log[str(c['i'])]))
        assert got == c['expected'], (got, c['expected'])
        assert m.sample_loop_time == c['sample_loop_time_after']


@pytest.mark.gpu
def test_real_sampler_and_checkpoint_roundtrip(device, tmp_path):
    gd = evaluate.create_sampler('qualitative', hidden_dim=64, timesteps=100, EBM='ULA', samples_per_step=2, device=device)
    gd.load_state_dict({'denoise_fn.' + k: v for k, v in weights('weights_qualitative_h64.npz').items()})
    sets = make_sets(6, 3)
    ev = evaluate.Evaluator(gd, sets, str(tmp_path))
    log = ev.evaluate(0, tries=(2, 1), run_all=True, seed=3)
    rec = log['3']
    assert 0.0 <= rec['success_rate'] <= rec['success_rate_top3'] <= 1.0
    assert rec['model_ave_sample_time'] > 0 and rec['sampling_time'][0][1] == 6
    # checkpoint in the reference's format: {'step', 'model': {schedule buffers + denoise_fn.*}}
    path = os.path.join(str(tmp_path), 'model-3.pt')
    evaluate.save_checkpoint(path, gd, step=123)
    data = torch.load(path)
    assert data['step'] == 123 and 'betas' in data['model'] and 'denoise_fn.mlps.12.0.weight' in data['model']
    gd2 = evaluate.create_sampler('qualitative', hidden_dim=64, timesteps=100, EBM='ULA', samples_per_step=2, device=device,
                                  checkpoint=path)
    b = worlds.collate(sets[3]).to_torch(device)
    assert torch.equal(gd.sample(b, seed=9), gd2.sample(b, seed=9))


@pytest.mark.gpu
def test_schedule_follows_reloaded_weights(device):
    """regression: reloading weights destroys and re-creates the native model, which may land on the freed
    address; the GaussianDiffusion schedule (samples_per_step, step sizes) must be re-applied to it regardless"""
    gd = evaluate.create_sampler('qualitative', hidden_dim=64, timesteps=20, EBM='ULA', samples_per_step=3, device=device)
    sd = {'denoise_fn.' + k: v for k, v in weights('weights_qualitative_h64.npz').items()}
    b = worlds.collate(make_sets(2, 3)[3]).to_torch(device)
    ref = None
    for _ in range(6):
        gd.load_state_dict(sd)
        x = gd.sample(b, seed=4)
        assert gd.chain_stats()['evals'] == 20 * (1 + 3)
        ref = x if ref is None else ref
        assert torch.equal(x, ref)
