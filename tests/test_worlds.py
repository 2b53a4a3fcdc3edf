"""synthetic constraint-graph generator vs the reference's labeller (golden) and layout contract"""
import numpy as np

from conftest import golden, worlds


def test_labeller_matches_reference_golden():
    z = golden('labeller')
    n_cases = len([k for k in z.files if k.endswith('/boxes')])
    assert n_cases >= 30
    for i in range(n_cases):
        boxes = z['case%d/boxes' % i]
        objects = worlds.tray_objects([tuple(b) for b in boxes], 3.0, 2.0)
        got = worlds.qualitative_constraints(objects, None, 1.0)
        want = [(worlds.QUALITATIVE_CONSTRAINTS[t], a, b) for t, a, b in z['case%d/cons' % i].tolist()]
        assert got == want, 'case %d' % i


def test_qualitative_batch_contract():
    b = worlds.qualitative_batch(16, 8, seed=5)
    assert b.x.shape == (16 * 9, 6) and b.x.dtype == np.float32
    assert b.edge_index.dtype == np.int64 and b.edge_index.shape[0] == 2
    assert b.edge_attr.dtype == np.float32 and b.mask.dtype == np.int8
    assert b.mask.sum() == 16 and (b.mask[::9] == 1).all()
    # container rows: [1, 1 | 0, 0, 0, 0]
    assert np.allclose(b.x[::9], [1, 1, 0, 0, 0, 0])
    # edges never cross graphs (block-diagonal collation)
    ga = b.edge_index[0] // 9
    gb = b.edge_index[1] // 9
    assert (ga == gb).all()
    per_graph = np.bincount(ga, minlength=16)
    assert per_graph.min() >= 8 + 28 and per_graph.max() <= 110
    # 'in' edges point object -> container, 28 cfree pairs per graph
    types = b.edge_attr.astype(int)
    assert (np.bincount(ga[types == 0], minlength=16) == 8).all()
    assert (np.bincount(ga[types == 6], minlength=16) == 28).all()
    assert (b.edge_index[1][types == 0] % 9 == 0).all()
    # w >= l after the swap, yaw encodes it
    assert (b.x[b.mask == 0][:, 0] * 3 >= b.x[b.mask == 0][:, 1] * 2 - 1e-6).all()


def test_other_world_shapes():
    t = worlds.triangular_batch(3, 12, seed=1)
    assert t.x.shape == (39, 7) and t.edge_index.shape == (2, 3 * 78)
    r = worlds.robot_box_batch(3, 10, seed=1)
    assert r.x.shape == (33, 28) and r.edge_index.shape == (2, 3 * 55)
    assert set(np.unique(r.edge_attr)) == {0.0, 1.0}


def test_deterministic():
    a = worlds.qualitative_batch(4, 8, seed=5)
    b = worlds.qualitative_batch(4, 8, seed=5)
    assert np.array_equal(a.x, b.x) and np.array_equal(a.edge_index, b.edge_index)
