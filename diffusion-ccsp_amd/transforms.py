"""Raw graph -> the sampler's input contract, without torch_geometric.

Restates ``data_transform_cn_diffuse_batch`` (reference networks/data_transforms.py:26-200), the
stability json->graph encoder (:272-303) and the robot (TableToBoxWorld) json->graph encoder (:203-269).  A raw graph is what the reference's world generators save
(envs/worlds.py:247-358): ``x`` rows ``[type, features...]`` and ``edge_index`` entries
``(constraint name, arg1, arg2)``.  The output is the dict consumed by ``worlds.collate``:
x [n, F] fp32 normalised (geometry columns then pose columns), edge_index [2, E] int64, edge_attr [E]
fp32 (type id), mask [n] int8 (node 0 = container), world_dims.  Pinned by golden vectors produced by
the reference function itself (tests/golden/pre_transform.npz, oracle/gen_golden.py).
"""
import numpy as np

from .worlds import (PUZZLE_CONSTRAINTS, QUALITATIVE_CONSTRAINTS, ROBOT_CONSTRAINTS, STABILITY_CONSTRAINTS)


def pre_transform(raw_x, raw_edges, input_mode):
    """raw_x: [n, 1 + k] rows [type, ...]; raw_edges: [(name, a, b)]"""
    # raw graphs are stored as float32 tensors (envs/data_utils.save_graph_data); the reference reads them
    # back with .tolist(), i.e. computes in float64 on fp32-rounded inputs, then casts once to fp32
    raw_x = np.asarray(raw_x, dtype=np.float32).astype(np.float64)
    w_tray, l_tray = float(raw_x[0, 1]), float(raw_x[0, 2])
    world_dims = (w_tray, l_tray)
    all_constraints = PUZZLE_CONSTRAINTS
    feats = []
    for dd in raw_x.tolist():
        n = len(dd)
        if n == 5:                                               # box: [type, w, l, x, y]   (:57-64)
            typ, w, l, x, y = dd
            feats.append([w / w_tray, l / l_tray, x / (w_tray / 2), y / (l_tray / 2)])
        elif n == 7:
            if 'diffuse_pairwise' in input_mode:                 # triangle P1 with theta (:69-85)
                if dd[0] == 0:
                    typ, w, l, _, x, y, _ = dd
                    feats.append([w / w_tray, l / l_tray, 0, x, y, 0])
                else:
                    typ, l, x3, y3, x1, y1, r1 = dd
                    feats.append([l / w_tray, x3 / w_tray, y3 / l_tray, x1 / (w_tray / 2), y1 / (l_tray / 2), r1 / np.pi])
            elif dd[0] == 0:                                     # container of the sin/cos box encodings (:90-95)
                typ, w, l, x, y, _, _ = dd
                feats.append([w / w_tray, l / l_tray, x, y, 0, 0])
            elif 'stability' in input_mode:                      # (:97-100)
                all_constraints = STABILITY_CONSTRAINTS
                feats.append(dd[1:3] + dd[3:])
            elif 'qualitative' in input_mode:                    # (:101-109): stored sn, cs -> pose [x, y, cs, sn]
                all_constraints = QUALITATIVE_CONSTRAINTS
                _, w, l, x, y, sn, cs = dd
                feats.append([w / w_tray, l / l_tray, x / (w_tray / 2), y / (l_tray / 2), cs, sn])
            else:
                raise ValueError('7-column rows need a diffuse_pairwise / stability / qualitative input_mode')
        elif n == 8:                                             # triangle P1 with sin/cos (:112-127)
            if dd[0] == 0:
                typ, w, l, _, x, y, _, _ = dd
                feats.append([w / w_tray, l / l_tray, 0, x, y, 0, 0])
            else:
                typ, l, x3, y3, x1, y1, cs, sn = dd
                feats.append([l / w_tray, x3 / w_tray, y3 / l_tray, x1 / (w_tray / 2), y1 / (l_tray / 2), cs, sn])
        elif n in (22, 29, 36):                                  # robot: geometry 8, the rest as is (:159-166)
            feats.append(dd[1:9] + dd[9:])
            all_constraints = ROBOT_CONSTRAINTS
            world_dims = tuple(dd[4:6])
        else:
            raise ValueError('unsupported raw row width %d' % n)
    if 'stability' in input_mode:
        all_constraints = STABILITY_CONSTRAINTS
    elif 'qualitative' in input_mode and 'robot' not in input_mode:
        all_constraints = QUALITATIVE_CONSTRAINTS
    x = np.asarray(feats, dtype=np.float32)
    edge_attr = np.asarray([all_constraints.index(e[0]) for e in raw_edges], dtype=np.float32)
    edge_index = np.asarray([[e[1], e[2]] for e in raw_edges], dtype=np.int64).T.reshape(2, -1)
    mask = np.zeros(x.shape[0], dtype=np.int8)
    mask[0] = 1                                                  # conditioned_variables = [0] (:45,182-186)
    return dict(x=x, edge_index=edge_index, edge_attr=edge_attr, mask=mask, world_dims=world_dims)


def stability_raw_graph(container, placements, supports, input_mode='stability_flat'):
    """``stability_data_json_to_pt`` (:272-303): shelf extent/pose + placed boxes + support pairs ->
    (raw_x [n, 7] rows [type, w, l, x, y, sn, cs], raw_edges)"""
    w0, l0 = container['shelf_extent'][:2]
    x0, y0 = container['shelf_pose'][:2]
    rows = [[0, w0, l0, 0, 0, 0, 0]]
    for obj in placements:
        w, l = obj['extents'][:2]
        x, y = obj['centroid'][:2]
        x = (x - x0) / w0 * 2
        y = (y - y0) / l0 * 2
        yaw = obj['theta']
        if 'flat' in input_mode and w > l:
            l, w = obj['extents'][:2]
            yaw = yaw + np.pi / 2
        rows.append([1, w / w0, l / l0, x, y, np.sin(yaw), np.cos(yaw)])
    n = len(rows)
    edges = [('within', i, 0) for i in range(1, n)]
    edges += [('supportedby', int(i), int(j)) for i, j in supports]
    for i in range(1, n):
        for j in range(i + 1, n):
            if ('supportedby', i, j) not in edges and ('supportedby', j, i) not in edges:
                edges.append(('cfree', i, j))
    return np.asarray(rows, dtype=np.float64), edges


GRASP_SIDES = ("x+", "x-", "y+", "y-", "z+")


def yaw_from_quat(q):
    """yaw (rotation about z) of a quaternion (x, y, z, w) -- the third component of pybullet's getEulerFromQuaternion,
    which the reference reaches through pybullet_planning.euler_from_quat (data_transforms.py:236)"""
    x, y, z, w = [float(v) for v in q]
    return float(np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z)))


def robot_raw_graph(container, placements, scene_id=0, qualitative_constraints=()):
    """``robot_data_json_to_pt`` (:203-269): tray dimensions / pose + grasped objects -> (raw_x [n, 29] rows
    [type | w/w0, l/l0, h/h0, w0, l0, h0, x0, y0 | mobility_id, scale | one-hot grasp side (5) | grasp_id | x, y, z, sn, cs |
    pick pose (7)], raw_edges 'gin' i -> 0 and 'gfree' j -> i for j > i).  Objects carry their ``grasp_side``
    ([(side, sign)], envs/data_utils.py:675-678); looking it up from a grasp quaternion needs the absent
    packing_models submodule (data_utils.py:686-693) and is not restated."""
    w0, l0 = [float(v) for v in container['tray_dim'][:2]]
    x0, y0, z0 = [float(v) for v in container['tray_pose']]
    h0 = 0.25
    rows = [[0] + [1, 1, 0, w0, l0, h0, x0, y0, 0, 0] + [0] * 6 + [0] * 5 + [0] * 7]
    for obj in placements:
        w, l, h = obj['extent']
        if 'place_pose' in obj:
            x, y, z = obj['place_pose'][0]
            x = (x - x0) / w0 * 2
            y = (y - y0) / l0 * 2
            z = z / h0
            yaw = yaw_from_quat(obj['place_pose'][1])
            mobility_id = int(obj['name'].split('_')[1])
        else:
            x, y, z, yaw = 0, 0, 0, 0
            mobility_id = scene_id
        sides = {k: 0 for k in GRASP_SIDES}
        sides.update({k[0]: abs(k[1]) for k in obj['grasp_side']})
        pick_pose = list(obj['pick_pose'][0]) + list(obj['pick_pose'][1])
        rows.append([1] + [w / w0, l / l0, h / h0, w0, l0, h0, x0, y0, mobility_id, obj['scale']] + list(sides.values())
                    + [obj['grasp_id']] + [x, y, z, np.sin(yaw), np.cos(yaw)] + pick_pose)
    n = len(rows)
    edges = [('gin', i, 0) for i in range(1, n)]
    for i in range(1, n):
        for j in range(i + 1, n):
            edges.append(('gfree', j, i))
    edges += [tuple(c) for c in qualitative_constraints]
    return np.asarray(rows, dtype=np.float64), edges
