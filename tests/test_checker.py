"""the "solved?" check (next row 8f-1): ground-truth scenes are solved, perturbed ones are not"""
import numpy as np

from conftest import worlds
from diffusion_ccsp_amd import checker


def test_ground_truth_poses_are_solved():
    b = worlds.qualitative_batch(40, 8, seed=3)
    gt = b.x[:, 2:6]
    ok = checker.solved_mask(gt, b)
    assert ok.all()
    b3 = worlds.qualitative_batch(20, 3, seed=4)
    assert checker.solved_mask(b3.x[:, 2:6], b3).all()


def test_collisions_and_violations_are_detected():
    b = worlds.qualitative_batch(6, 5, seed=9)
    gt = b.x[:, 2:6].copy()
    # put object 1 of graph 0 on top of object 2 -> collision
    bad = gt.copy()
    bad[1, :2] = bad[2, :2]
    ok = checker.solved_mask(bad, b)
    assert not ok[0] and ok[1:].all()
    # push an object through the east wall
    bad = gt.copy()
    bad[7, 0] = 0.999
    assert not checker.solved_mask(bad, b)[1]
    # NaN poses are never solved (Trainer.evaluate skips them, ddpm.py:644)
    bad = gt.copy()
    bad[13, 1] = np.nan
    assert not checker.solved_mask(bad, b)[2]
    # poses are clamped to [-1, 1] first (ddpm.py:620)
    far = gt.copy()
    far[1, 0] = 50.0
    assert np.array_equal(checker.solved_mask(far, b), checker.solved_mask(np.clip(far, -1, 1), b))


def test_oriented_overlap():
    a = (0.0, 0.0, 2.0, 1.0, 0.0)
    assert checker.rects_overlap(a, (1.5, 0.0, 2.0, 1.0, 0.0))
    assert checker.rects_overlap(a, (2.0, 0.0, 2.0, 1.0, 0.0))                # touching edges: in contact = colliding (FCL's rule)
    assert not checker.rects_overlap(a, (2.0 + 2.0 ** -30, 0.0, 2.0, 1.0, 0.0))
    assert not checker.rects_overlap(a, (0.0, 1.6, 2.0, 1.0, 0.0))
    # a diamond that a bounding-box test would flag but SAT separates
    assert not checker.rects_overlap((0.0, 0.0, 1.0, 1.0, 0.0), (1.3, 1.3, 1.0, 1.0, np.pi / 4))
    assert checker.rects_overlap((0.0, 0.0, 1.0, 1.0, 0.0), (0.8, 0.8, 1.0, 1.0, np.pi / 4))
    # swapped box: stored (w, l) with yaw -pi/2 has the footprint (l, w)
    assert checker.rects_overlap((0.0, 0.0, 2.0, 0.2, -np.pi / 2), (0.0, 0.9, 0.5, 0.5, 0.0))
    assert not checker.rects_overlap((0.0, 0.0, 2.0, 0.2, 0.0), (0.0, 0.9, 0.5, 0.5, 0.0))


def test_missing_constraint_is_reported():
    rng = np.random.default_rng(1)
    wd = worlds.sample_qualitative_world(rng, 4)
    g = worlds.encode_qualitative(wd['nodes'], wd['constraints'])
    given = [(worlds.QUALITATIVE_CONSTRAINTS[int(t)], int(a), int(b)) for t, a, b in zip(g['edge_attr'], g['edge_index'][0], g['edge_index'][1])]
    feats = g['x']
    assert checker.evaluate_graph(feats, (3.0, 2.0), given) == []
    # a constraint that does not hold in the scene is reported as missing
    fake = ('center-in', 1, 0) if ('center-in', 1, 0) not in given else ('left-in', 1, 0)
    missing = checker.evaluate_graph(feats, (3.0, 2.0), given + [fake])
    assert missing == [fake] or missing == []  # ('left-in' may hold by chance; 'center-in' cannot be both)


def _poly_distance(pa, pb):
    """distance of two convex polygons, from scratch and without any separating-axis reasoning: 0 if they share a point
    (an edge crossing, a vertex on an edge, or one inside the other), else the smallest vertex-to-segment distance"""
    def seg_pt(p, a, b):
        ax, ay, bx, by = a[0], a[1], b[0], b[1]
        dx, dy = bx - ax, by - ay
        t = ((p[0] - ax) * dx + (p[1] - ay) * dy) / (dx * dx + dy * dy)
        t = min(1.0, max(0.0, t))
        return np.hypot(p[0] - (ax + t * dx), p[1] - (ay + t * dy))

    def orient(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])

    def inside(p, poly):                      # closed polygon (counter-clockwise)
        return all(orient(poly[i], poly[(i + 1) % len(poly)], p) >= 0 for i in range(len(poly)))

    def cross(a, b, c, d):                    # closed segments ab and cd share a point
        o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
        if (o1 > 0) != (o2 > 0) and (o3 > 0) != (o4 > 0) and o1 != 0 and o2 != 0 and o3 != 0 and o4 != 0:
            return True
        on = lambda p, q, r: orient(p, q, r) == 0 and min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])  # noqa: E731
        return on(a, b, c) or on(a, b, d) or on(c, d, a) or on(c, d, b)
    ea = [(pa[i], pa[(i + 1) % 4]) for i in range(4)]
    eb = [(pb[i], pb[(i + 1) % 4]) for i in range(4)]
    if any(cross(a, b, c, d) for a, b in ea for c, d in eb) or inside(pa[0], pb) or inside(pb[0], pa):
        return 0.0
    return min(min(seg_pt(p, a, b) for p in pa for a, b in eb), min(seg_pt(p, a, b) for p in pb for a, b in ea))


def test_contact_convention_vs_polygon_distance():
    """FCL's box-box test (boxBox2 / ODE dBoxBox: an axis separates only if the gap along it is > 0) reports boxes at zero
    distance as colliding; python-fcl is not importable here, so the rule is checked against an independent computation:
    collide <=> the polygons' distance is 0.  Several hundred configurations: random orientations with the second box placed
    at a gap of -1e-6 / +1e-6 / +1e-4 from a face of the first (face-face, corner-face and corner-corner approaches), and exactly
    representable axis-aligned and 90-degree configurations at a gap of exactly 0, edge and corner contact."""
    rng = np.random.default_rng(7)
    n = 0
    for _ in range(150):
        w1, l1, w2, l2 = rng.uniform(0.1, 1.0, 4)
        th = rng.uniform(-np.pi, np.pi)
        hx = (abs(np.cos(th)) * w2 + abs(np.sin(th)) * l2) / 2        # half extent of the turned box along x
        y = rng.uniform(-(l1 + l2) / 2, (l1 + l2) / 2) * rng.choice([0.2, 1.0])
        for gap in (-1e-6, 1e-6, 1e-4):
            a = (0.0, 0.0, w1, l1, 0.0)
            b = (w1 / 2 + hx + gap, y, w2, l2, th)
            d = _poly_distance(checker._corners(*a), checker._corners(*b))
            want = bool(d == 0.0)
            assert checker.rects_overlap(a, b) is want and checker.rects_overlap(b, a) is want, (a, b, gap, d)
            # configurations whose nearest feature (the turned box's leftmost corner) faces the first box's right face are
            # decided by the gap alone
            left = min(checker._corners(*b))
            if abs(left[1]) <= l1 / 2 - 1e-3:
                assert want is (gap < 0), (a, b, gap)
            n += 1
    # exact contact: dyadic numbers, rotations by multiples of 90 degrees (cos / sin exact up to the sign of zero)
    for w1, l1, w2, l2 in ((0.5, 0.25, 0.25, 0.5), (1.0, 0.5, 0.5, 0.5), (0.75, 0.5, 0.25, 0.125)):
        a = (0.0, 0.0, w1, l1, 0.0)
        for b in ((w1 / 2 + w2 / 2, 0.0, w2, l2, 0.0),                          # face to face
                  (w1 / 2 + w2 / 2, l1 / 2 + l2 / 2, w2, l2, 0.0),              # corner to corner
                  (0.0, l1 / 2 + l2 / 2, w2, l2, 0.0),                          # face to face along y
                  (w1 / 2 + w2 / 2, 0.125, w2, l2, 0.0)):                       # shifted along the shared face
            assert _poly_distance(checker._corners(*a), checker._corners(*b)) == 0.0
            assert checker.rects_overlap(a, b) and checker.rects_overlap(b, a), (a, b)
            for dx in (2.0 ** -20, 2.0 ** -10):                                 # ... and free as soon as there is a gap
                bb = (b[0] + (dx if b[0] > 0 else 0.0), b[1] + (dx if b[0] == 0 else 0.0), b[2], b[3], b[4])
                assert not checker.rects_overlap(a, bb), (a, bb)
            n += 1
    assert n >= 450
    # tiles against the tray walls: a tile whose edge lies exactly on the inner wall face is in contact (collides), 2^-20 inside it is free
    w, l = 3.0, 2.0
    tile = lambda x: [(x, 0.0, 0.5, 0.5, 0.0)]  # noqa: E731
    assert checker.collisions(tile(w / 2 - 0.25), (w, l)) == [('east', 'tile_0')]
    assert checker.collisions(tile(w / 2 - 0.25 - 2.0 ** -20), (w, l)) == []
    assert checker.collisions(tile(w / 2 - 0.25 + 1e-6), (w, l)) == [('east', 'tile_0')]
    assert checker.collisions(tile(-w / 2 + 0.25 - 1e-6), (w, l)) == [('west', 'tile_0')]
    # the generator's own scenes keep a positive clearance between tiles (paddings), so no ground truth sits on the boundary
    b = worlds.qualitative_batch(10, 8, seed=21)
    for j in range(10):
        nodes = np.nonzero(np.asarray(b.batch) == j)[0]
        feats = np.concatenate([b.x[nodes, :2], b.x[nodes, 2:6]], axis=1)
        tiles = checker.reconstruct(feats, (3.0, 2.0))
        shrunk = [(t[0], t[1], t[2] - 1e-4, t[3] - 1e-4, t[4]) for t in tiles]
        grown = [(t[0], t[1], t[2] + 1e-4, t[3] + 1e-4, t[4]) for t in tiles]
        assert checker.collisions(shrunk, (3.0, 2.0)) == [] and checker.collisions(grown, (3.0, 2.0)) == []
