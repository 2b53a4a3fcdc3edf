"""The "solved?" check of RandomSplitQualitativeWorld samples -- what turns
samples/s into *solved* samples/s.

In the reference, ``Trainer.evaluate`` (networks/ddpm.py:620-713) clamps the sampled poses to [-1, 1],
rebuilds a scene per graph (envs/data_utils.py:221-258,299-313 -> envs/worlds.py:662-712,766-770) and
calls ``check_constraints_satisfied`` (envs/worlds.py:734-764): a sample is solved iff

  1. no pair of bodies collides (FCL box-box between tiles and the four tray walls; wall-wall pairs and
     anything touching 'bottom' are ignored -- envs/worlds.py:380-388, envs/collisions.py:58-130), and
  2. every given constraint is found among the constraints of the reconstructed scene ('in' and all
     'cfree' pairs by construction, plus the qualitative labeller's output with the boxes' rotations),
     after ``expand_unordered_constraints`` on both sides (envs/data_utils.py:418-424).

Restated here without trimesh / python-fcl:
  * the labeller is worlds.qualitative_constraints (pinned against the reference by golden vectors);
  * FCL is not importable in the build container, so box-box collision is a separating-axis test on the
    oriented footprints (all bodies span the same z range).  Strictly overlapping / strictly apart boxes agree
    with any exact box-box query by geometry.  Contact convention = FCL's: its box-box narrow phase (boxBox2, the
    ODE dBoxBox test python-fcl's fcl.collide runs for two fcl.Box shapes, envs/collisions.py:62-66,118-127)
    declares an axis separating only when the gap along it is STRICTLY positive (`s2 = |t.axis| - (ra + rb);
    if (s2 > 0) return 0`), so boxes at exactly zero distance are in contact = colliding.  Same rule here (round 2
    had the opposite: touching = free).  It only matters on a measure-zero set of poses; the generator's own scenes
    keep > 1e-4 clearance.  tests/test_checker.py checks the rule against an independent polygon-distance computation
    on several hundred configurations at gaps -1e-6, 0 (exactly representable ones), +1e-6 and +1e-4.

Reference quirk kept: the feature columns are stored [w, l, x, y, cs, sn] but unpacked as
``w, l, x, y, sn, cs`` (data_utils.py:246), so the yaw used downstream is atan2(col4, col5).
"""
import math
from collections import OrderedDict

import numpy as np

from .worlds import IGNORED_CONSTRAINTS, QUALITATIVE_CONSTRAINTS, qualitative_constraints, tray_objects

UNORDERED = ('close-to', 'away-from', 'h-aligned', 'v-aligned', 'cfree')


def yaw_from_sn_cs(sn, cs):
    """envs/data_utils.py:360-364"""
    total = math.sqrt(sn ** 2 + cs ** 2)
    if total == 0.0 or math.isnan(total):
        return float('nan')                      # numpy gives nan here (0/0); such a sample is never solved
    return math.atan2(sn / total, cs / total)


def reconstruct(features, world_dims):
    """normalised rows [w, l, x, y, c4, c5] (row 0 = container) -> [(cx, cy, bw, bl, yaw)] of the tiles
    (envs/data_utils.py:241-249)"""
    w_tray, l_tray = world_dims
    out = []
    for f in np.asarray(features, dtype=np.float64)[1:]:
        bw, bl, x, y, c4, c5 = [float(v) for v in f[:6]]
        out.append((x * w_tray / 2, y * l_tray / 2, bw * w_tray, bl * l_tray, yaw_from_sn_cs(c4, c5)))
    return out


def _corners(cx, cy, bw, bl, yaw):
    c, s = math.cos(yaw), math.sin(yaw)
    pts = []
    for dx, dy in ((bw / 2, bl / 2), (-bw / 2, bl / 2), (-bw / 2, -bl / 2), (bw / 2, -bl / 2)):
        pts.append((cx + c * dx - s * dy, cy + s * dx + c * dy))
    return pts


def rects_overlap(a, b):
    """separating-axis test of two oriented rectangles (cx, cy, w, l, yaw); FCL's rule: an axis separates only if the gap
    along it is strictly positive, so boxes at zero distance collide"""
    if any(math.isnan(v) for v in a) or any(math.isnan(v) for v in b):
        return True                              # an undefined pose counts as a violation
    pa, pb = _corners(*a), _corners(*b)
    for rect in (a, b):
        c, s = math.cos(rect[4]), math.sin(rect[4])
        for ax in ((c, s), (-s, c)):
            ja = [p[0] * ax[0] + p[1] * ax[1] for p in pa]
            jb = [p[0] * ax[0] + p[1] * ax[1] for p in pb]
            if max(ja) < min(jb) or max(jb) < min(ja):
                return False
    return True


def collisions(tiles, world_dims, t=0.1):
    """colliding (label, label) pairs among tiles and the tray walls (envs/mesh_utils.py:174-191 for the
    wall boxes; wall-wall and 'bottom' pairs are not reported, envs/worlds.py:380-388)"""
    w, l = world_dims
    bodies = OrderedDict()
    bodies['north'] = (0.0, (l + t) / 2, w, t, 0.0)
    bodies['south'] = (0.0, -(l + t) / 2, w, t, 0.0)
    bodies['west'] = (-(w + t) / 2, 0.0, t, l + 2 * t, 0.0)
    bodies['east'] = ((w + t) / 2, 0.0, t, l + 2 * t, 0.0)
    walls = set(bodies)
    for i, r in enumerate(tiles):
        bodies['tile_%d' % i] = r
    names = list(bodies)
    out = []
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            if names[i] in walls and names[j] in walls:
                continue
            if rects_overlap(bodies[names[i]], bodies[names[j]]):
                out.append((names[i], names[j]))
    return out


def expand_unordered(cons):
    """envs/data_utils.py:418-424"""
    out = []
    for c in cons:
        if c[0] in UNORDERED:
            out.append((c[0], c[2], c[1]))
        out.append(tuple(c))
    return out


def current_constraints(tiles, world_dims):
    """'in' + all-pairs 'cfree' (envs/worlds.py:136-144) + the labeller on the rebuilt scene with rotations
    (envs/worlds.py:729-732)"""
    w, l = world_dims
    n = len(tiles) + 1
    objects = tray_objects([(r[0], r[1], r[2], r[3]) for r in tiles], w, l)
    rotations = {'tile_%d' % i: r[4] for i, r in enumerate(tiles)}
    cons = [('in', i, 0) for i in range(1, n)]
    cons += [('cfree', i, j) for i in range(1, n - 1) for j in range(i + 1, n)]
    cons += qualitative_constraints(objects, rotations=rotations, scale=min(w / 3, l / 2))
    return [c for c in cons if c[0] not in IGNORED_CONSTRAINTS]


def evaluate_graph(features, world_dims, given):
    """[] if solved, else the list of collisions or missing constraints (envs/worlds.py:734-764)"""
    feats = np.asarray(features, dtype=np.float64)
    if np.isnan(feats).any():
        return [('nan',)]
    tiles = reconstruct(feats, world_dims)
    col = collisions(tiles, world_dims)
    if col:
        return col
    cur = set(expand_unordered(current_constraints(tiles, world_dims)))
    giv = expand_unordered([c for c in given if c[0] not in IGNORED_CONSTRAINTS])
    return [c for c in giv if c not in cur]


def solved_mask(poses, batch):
    """per graph of a collated qualitative batch: True iff the sampled poses [N, 4] (clamped to [-1, 1] like
    Trainer.evaluate, ddpm.py:620) solve the graph's constraints"""
    poses = np.clip(np.asarray(poses, dtype=np.float64), -1.0, 1.0)
    x = np.asarray(batch.x, dtype=np.float64)
    gid = np.asarray(batch.batch)
    ei = np.asarray(batch.edge_index)
    ea = np.asarray(batch.edge_attr)
    n_graphs = int(gid.max()) + 1
    out = np.zeros(n_graphs, dtype=bool)
    egid = gid[ei[0]]
    for j in range(n_graphs):
        nodes = np.nonzero(gid == j)[0]
        n0 = int(nodes[0])
        feats = np.concatenate([x[nodes, :2], poses[nodes]], axis=1)          # Trainer.get_all_features
        sel = np.nonzero(egid == j)[0]
        given = []
        for e in sel:                                                         # constraint_from_edge_attr
            typ = int(ea[e])
            if 0 <= typ < len(QUALITATIVE_CONSTRAINTS):
                given.append((QUALITATIVE_CONSTRAINTS[typ], int(ei[0, e]) - n0, int(ei[1, e]) - n0))
        wd = batch.world_dims[j] if hasattr(batch, 'world_dims') else (3.0, 2.0)
        out[j] = len(evaluate_graph(feats, wd, given)) == 0
    return out
