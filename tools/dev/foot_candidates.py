"""Prototype for round 5's first K6 step: a direction-binned candidate table for the support vertex of the FOOT hull.

The support vertex of direction d (the vertex maximising d . v) over all directions of one cell of a cube map of the sphere is one of a
handful of vertices.  This script reads the foot hull (tools/compile_hulls.py's walk over the reference's meshes; data only), builds the
table -- per cell: the vertices that are optimal at any of S x S samples of the cell, plus their neighbours on the hull -- and verifies it
against brute force on random directions.  Output: table statistics (what the engine would carry) and the worst support-value error.
usage: foot_candidates.py [K cells per cube-face edge = 16] [S samples per cell edge = 12] [checks = 2000000]"""
import os, sys
import numpy as np
from scipy.spatial import ConvexHull
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from compile_hulls import robot_hulls                                   # noqa: E402
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
NCHK = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
desc = "/root/reference/solo_description"


def cell_of(d):
    """cube-map cell of unit directions d [n, 3]: face = 2 * major axis + (negative?), (u, v) = the other two components / |major|"""
    a = np.abs(d); ax = a.argmax(1); n = len(d); i = np.arange(n)
    major = d[i, ax]
    u = d[i, (ax + 1) % 3] / np.abs(major); v = d[i, (ax + 2) % 3] / np.abs(major)
    cu = np.minimum((0.5 * (u + 1) * K).astype(int), K - 1); cv = np.minimum((0.5 * (v + 1) * K).astype(int), K - 1)
    return ((2 * ax + (major < 0)) * K + cu) * K + cv


for tag, urdf in (("solo12", "solo12.urdf"), ("solo8", "solo.urdf")):
    order, tables, link_table = robot_hulls(desc, urdf)
    foot = [i for i, n in enumerate(order) if n.endswith("FOOT")]
    assert len({link_table[i] for i in foot}) == 1, "the four feet share one hull"
    V = tables[link_table[foot[0]]].astype(np.float64)
    hull = ConvexHull(V)
    nb = [set() for _ in V]
    for tri in hull.simplices:
        for a in tri:
            nb[a].update(int(b) for b in tri if b != a)
    # samples of every cell
    g = (np.arange(S) + 0.5) / S
    cand = [set() for _ in range(6 * K * K)]
    for face in range(6):
        ax, neg = face // 2, face % 2
        for cu in range(K):
            for cv in range(K):
                u = (2 * (cu + g) / K - 1)[:, None] * np.ones((1, S)); v = np.ones((S, 1)) * (2 * (cv + g) / K - 1)[None, :]
                d = np.zeros((S * S, 3)); d[:, ax] = -1.0 if neg else 1.0; d[:, (ax + 1) % 3] = u.ravel(); d[:, (ax + 2) % 3] = v.ravel()
                # plus the cell's corners and edges (the optimal vertex changes along them)
                best = np.unique((d @ V.T).argmax(1))
                c = cand[(face * K + cu) * K + cv]
                for b in best:
                    c.add(int(b)); c.update(nb[int(b)])
    sizes = np.array([len(c) for c in cand])
    # verification on random directions
    rng = np.random.default_rng(1)
    worst, miss = 0.0, 0
    for chunk in range(NCHK // 100000):
        d = rng.standard_normal((100000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        sup = (d @ V.T)
        bf = sup.max(1); cells = cell_of(d)
        for ci in np.unique(cells):
            sel = np.nonzero(cells == ci)[0]
            idx = np.fromiter(cand[ci], int)
            got = sup[np.ix_(sel, idx)].max(1)
            err = bf[sel] - got
            worst = max(worst, float(err.max())); miss += int((err > 0).sum())
    print("%s foot hull: %d vertices; cube map %d x %d x 6 = %d cells; candidates per cell: mean %.1f, median %d, max %d; table %d entries (u16: %.1f KB)" % (
        tag, len(V), K, K, 6 * K * K, sizes.mean(), np.median(sizes), sizes.max(), sizes.sum(), 2 * sizes.sum() / 1024))
    print("  %d random directions: the true support vertex is not among the cell's candidates in %d cases; worst support-value shortfall %.3g m" % (NCHK // 100000 * 100000, miss, worst))
