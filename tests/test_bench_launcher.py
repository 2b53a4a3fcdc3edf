"""`python bench.py --gpus N` as the driver invokes it (no torchrun around it): the launcher re-executes itself as N ranks.  CPU coverage of
that path every round: world size 2 over gloo with --dry-run (launcher -> init_process_group -> one flat weight broadcast -> gather ->
max-over-ranks timing -> ONE JSON line).  Nothing is sampled in a dry run (the HIP path has no CPU fallback), value is null."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, **kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'CCSP_LANES')}
    env.update(kw.get('env', {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, env=env, timeout=kw.get('timeout', 600), cwd=ROOT)


@pytest.mark.parametrize('config', ['c2', 'c5'])
def test_gpus_2_self_launches_two_ranks_gloo_dry_run(config):
    r = _run('--gpus', '2', '--backend', 'gloo', '--dry-run', '--steps', '2', '--warmup', '1', '--config', config)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                     # rank 0 prints ONE JSON line, rank 1 none
    rec = json.loads(lines[0])
    assert rec['dry_run'] is True and rec['value'] is None
    assert rec['n_gpus'] == 2 and rec['world'] == 2 and rec['steps'] == 2 and rec['warmup'] == 1
    c = rec['communicator']
    assert c['backend'] == 'gloo' and c['world'] == 2
    assert c['allreduce_of_[1,rank]'] == [2.0, 1.0] == c['allreduce_expected']       # only two DISTINCT ranks sum to [2, 0 + 1]
    assert len(rec['per_rank_ms_per_step']) == 2
    assert rec['config']['weights_broadcast_equal_to_file_on_every_rank'] is True    # rank 1 never read the file's values: they came over the wire
    assert rec['config']['gathered_rows_in_rank_order'] is True
    assert len(rec['config']['nodes_per_rank']) == 2
    assert 'launching 2 ranks' in r.stderr


def test_gloo_without_dry_run_is_refused():
    r = _run('--gpus', '1', '--backend', 'gloo')
    assert r.returncode != 0 and 'dry-run' in (r.stderr + r.stdout)


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--dry-run'], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_lanes_rule_budgets_the_host_for_n_ranks():
    """bench.select_lanes: a two-lane rank keeps 2.67 host cores busy, a one-lane rank 1.9 (profiles/r06_host_budget.txt), so the launcher falls back
    to one lane per rank when the container grants fewer than 2.2 cores per rank -- the GPU boxes' 16-core quota under 8 ranks -- and says so"""
    sys.path.insert(0, ROOT)
    import bench
    lanes, why = bench.select_lanes(8, 16)
    assert lanes == 1 and '16 usable cores' in why and '8 ranks' in why
    assert bench.select_lanes(8, 20)[0] is None                 # 20 >= 2.2 x 8: the library default (two lanes)
    assert bench.select_lanes(7, 16)[0] is None                 # measured: seven two-lane ranks on 16 cores 499 samples/s each, one-lane 466
    assert bench.select_lanes(4, 16)[0] is None and bench.select_lanes(1, 16)[0] is None and bench.select_lanes(2, 8)[0] is None
    assert bench.select_lanes(1, 2)[0] == 1 and bench.select_lanes(2, 4)[0] == 1
    lanes, why = bench.select_lanes(8, 16, '2')                  # an explicit CCSP_LANES always wins
    assert lanes is None and 'set by the caller' in why


def test_dry_run_line_carries_the_host_budget():
    """the N > 1 line shows what each rank cost its host and which lane count the rule chose (here: 4 usable cores for 2 ranks -> one lane)"""
    r = _run('--gpus', '2', '--backend', 'gloo', '--dry-run', '--steps', '2', '--warmup', '1', env={'CCSP_BENCH_HOST_CORES': '4'})
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    hb = rec['host_budget']
    assert hb['usable_cores'] == 4 and hb['ccsp_lanes_env'] == '1' and 'one lane per rank' in hb['rule']
    assert len(rec['host_cpu_s_per_step']) == 2 and all(v >= 0 for v in rec['host_cpu_s_per_step'])
    r = _run('--gpus', '2', '--backend', 'gloo', '--dry-run', '--steps', '1', '--warmup', '0', env={'CCSP_BENCH_HOST_CORES': '16'})
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert rec['host_budget']['ccsp_lanes_env'] is None and 'library default' in rec['host_budget']['rule']
