"""pre_transform restatement (next row 8f-3) against the reference's own data_transform_cn_diffuse_batch"""
import numpy as np

from conftest import golden, worlds
from diffusion_ccsp_amd import transforms

NAMES = {'qualitative': worlds.QUALITATIVE_CONSTRAINTS, 'diffuse_pairwise': worlds.PUZZLE_CONSTRAINTS,
         'stability_flat': worlds.STABILITY_CONSTRAINTS, 'robot_box': worlds.ROBOT_CONSTRAINTS}


def test_pre_transform_matches_reference():
    z = golden('pre_transform')
    tags = sorted(set(k.split('/')[0] for k in z.files if k.endswith('/raw_x')))
    assert len(tags) >= 8
    for tag in tags:
        mode = str(z[tag + '/mode'])
        raw_edges = [(NAMES[mode][int(t)], int(a), int(b)) for t, a, b in z[tag + '/raw_edges']]
        out = transforms.pre_transform(z[tag + '/raw_x'], raw_edges, mode)
        assert np.array_equal(out['x'], z[tag + '/x']), tag            # bit-exact: float64 arithmetic, one cast
        assert np.array_equal(out['edge_index'], z[tag + '/edge_index']), tag
        assert np.array_equal(out['edge_attr'], z[tag + '/edge_attr']), tag
        assert np.array_equal(out['mask'], z[tag + '/mask']), tag
        assert np.allclose(out['world_dims'], z[tag + '/world_dims']), tag


def test_stability_json_encoder_matches_reference():
    z = golden('pre_transform')
    container = dict(shelf_extent=z['stab/container_extent'].tolist(), shelf_pose=z['stab/container_pose'].tolist())
    placements = [dict(extents=e.tolist(), centroid=c.tolist(), theta=float(t))
                  for e, c, t in zip(z['stab/extents'], z['stab/centroids'], z['stab/thetas'])]
    raw_x, edges = transforms.stability_raw_graph(container, placements, z['stab/supports'].tolist())
    assert np.array_equal(raw_x.astype(np.float32), z['stab/ref_raw_x'])
    want = [(worlds.STABILITY_CONSTRAINTS[int(t)], int(a), int(b)) for t, a, b in z['stab/ref_raw_edges']]
    assert edges == want


def test_robot_json_encoder_matches_reference():
    """robot_data_json_to_pt (data_transforms.py:203-269): tray + grasped objects -> the 29-column raw rows and the
    'gin' / 'gfree' edges, then through pre_transform to the 28-column sampler input"""
    z = golden('pre_transform')
    keys = transforms.GRASP_SIDES
    placements = []
    for k in range(len(z['robotjson/scale'])):
        p = dict(name='Bottle_%d' % (int(z['robotjson/mobility'][k]) if k < 5 else 0), extent=z['robotjson/extent'][k].tolist(),
                 scale=float(z['robotjson/scale'][k]), grasp_id=int(z['robotjson/grasp_id'][k]),
                 grasp_side=[[keys[int(z['robotjson/grasp_side'][k][0])], int(z['robotjson/grasp_side'][k][1])]],
                 pick_pose=[z['robotjson/pick_pose'][k][:3].tolist(), z['robotjson/pick_pose'][k][3:].tolist()])
        if k < 5:
            p['place_pose'] = [z['robotjson/place_pos'][k].tolist(), z['robotjson/place_quat'][k].tolist()]
        placements.append(p)
    raw_x, edges = transforms.robot_raw_graph(dict(tray_dim=z['robotjson/tray_dim'].tolist(), tray_pose=z['robotjson/tray_pose'].tolist()),
                                              placements, scene_id=77)
    assert raw_x.shape == (7, 29)
    assert np.array_equal(raw_x.astype(np.float32), z['robotjson/ref_raw_x'])
    want = [(worlds.ROBOT_CONSTRAINTS[int(t)], int(a), int(b)) for t, a, b in z['robotjson/ref_raw_edges']]
    assert edges == want and len(edges) == 6 + 15
    out = transforms.pre_transform(raw_x.astype(np.float32), edges, 'robot_box')
    assert np.array_equal(out['x'], z['robotjson/x']) and np.array_equal(out['edge_index'], z['robotjson/edge_index'])


def test_encode_qualitative_is_the_same_function():
    rng = np.random.default_rng(0)
    wd = worlds.sample_qualitative_world(rng, 5)
    a = worlds.encode_qualitative(wd['nodes'], wd['constraints'])
    b = transforms.pre_transform(wd['nodes'], wd['constraints'], 'qualitative')
    for k in ('x', 'edge_index', 'edge_attr', 'mask'):
        assert np.array_equal(a[k], b[k])
