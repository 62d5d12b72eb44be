"""Seeded stand-ins for BASELINE.json configs[3] -- the SuiteSparse matrices nlpkkt120, ldoor and thermal2 -- for when the
files themselves are not on the box (there is no network here; the reference only ships a downloader,
testing/UF/PyUFTest.py, and reads such files through performance/spmv/spmv.cu:70-79).

Each generator follows what the collection publishes about its matrix (dimensions, entries, row-length range and mean --
the figures below are the collection's own summary statistics, quoted from memory of its pages, hence "like"; when
CMI_SUITESPARSE_DIR holds the real file, `load()` reads it and the generators are not used):

  name        rows       entries      entries/row (min .. mean .. max)   structure
  thermal2    1 228 045   8 580 313   1 ..  6.99 .. 11                   unstructured P1 FEM thermal problem (planar-like mesh)
  ldoor         952 203  42 493 817   28 .. 44.6 .. 77                   structural FEM, 3 dof per node: dense 3x3 blocks, banded
  nlpkkt120   3 542 400  95 117 792   5 .. 26.85 .. 28                   KKT system of a 3-D PDE-constrained optimisation:
                                                                         [[H, A^T], [A, 0]] with 27-point couplings between the halves

`scale` shrinks a matrix for the parity tests (same construction, fewer rows).  Returns numpy CSR arrays
(Ap int32, Aj int32 ascending inside a row, Ax float64) -- set-up code, numpy only.
"""
import os

import numpy as np

PUBLISHED = {
    "thermal2": {"rows": 1228045, "entries": 8580313, "min": 1, "mean": 6.99, "max": 11},
    "ldoor": {"rows": 952203, "entries": 42493817, "min": 28, "mean": 44.63, "max": 77},
    "nlpkkt120": {"rows": 3542400, "entries": 95117792, "min": 5, "mean": 26.85, "max": 28},
}


def _csr_from_coo(rows, cols, ri, ci, rng):
    """sorted, de-duplicated CSR with random values (diagonally dominant: the SpMV does not care, CG would)."""
    key = ri.astype(np.int64) * cols + ci.astype(np.int64)
    key = np.unique(key)
    ri = (key // cols).astype(np.int64)
    ci = (key % cols).astype(np.int32)
    Ap = np.zeros(rows + 1, np.int64)
    np.add.at(Ap, ri + 1, 1)
    Ap = np.cumsum(Ap)
    Ax = rng.standard_normal(len(ci))
    return Ap.astype(np.int32), ci, Ax


def thermal2_like(scale=1.0, seed=7):
    """Unstructured planar mesh: jittered triangulated grid (every interior node has 6 neighbours; random diagonal flips
    and deleted edges spread that to 1..11 as in the published histogram), nodes numbered along a Morton curve of their
    cell (a mesh generator's locality, not a random permutation)."""
    rng = np.random.default_rng(seed)
    n_target = int(PUBLISHED["thermal2"]["rows"] * scale)
    g = max(4, int(round(np.sqrt(n_target))))
    n = g * g
    iy, ix = np.divmod(np.arange(n), g)

    def morton(a, b):
        def spread(v):
            v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
            for s, m in ((16, 0x0000FFFF0000FFFF), (8, 0x00FF00FF00FF00FF), (4, 0x0F0F0F0F0F0F0F0F), (2, 0x3333333333333333), (1, 0x5555555555555555)):
                v = (v | (v << np.uint64(s))) & np.uint64(m)
            return v
        return spread(a) | (spread(b) << np.uint64(1))

    order = np.argsort(morton(ix, iy), kind="stable")
    label = np.empty(n, np.int64)
    label[order] = np.arange(n)
    edges = []
    for dx, dy in ((1, 0), (0, 1)):  # grid edges
        ok = (ix + dx < g) & (iy + dy < g)
        edges.append((np.arange(n)[ok], (np.arange(n) + dx + dy * g)[ok]))
    ok = (ix + 1 < g) & (iy + 1 < g)  # one diagonal per cell, orientation random
    flip = rng.random(n) < 0.5
    a = np.where(flip, np.arange(n) + 1, np.arange(n))[ok]
    b = np.where(flip, np.arange(n) + g, np.arange(n) + g + 1)[ok]
    edges.append((a, b))
    # a few longer-range couplings (refinement transitions) and deleted edges -> the tails of the histogram
    extra = rng.integers(0, n, size=n // 6)
    ex_ok = (ix[extra] + 2 < g) & (iy[extra] + 1 < g)
    edges.append((extra[ex_ok], extra[ex_ok] + 2 + g))
    u = np.concatenate([e[0] for e in edges])
    v = np.concatenate([e[1] for e in edges])
    keep = rng.random(len(u)) > 0.04
    u, v = label[u[keep]], label[v[keep]]
    ri = np.concatenate([u, v, np.arange(n)])
    ci = np.concatenate([v, u, np.arange(n)])
    return _csr_from_coo(n, n, ri, ci, rng)


def ldoor_like(scale=1.0, seed=11):
    """3 dof per node, dense 3x3 blocks; a node couples to itself and to some of the 24 nodes of the 5 x 5 window around it in
    a banded numbering (a thin shell: mesh lines of ~sqrt(nodes)/2 nodes).  Every node draws ~11 of them (with repeats) and the pattern is
    symmetrised, which leaves ~14 neighbours on average: 15 x 3 = 45 entries per row, at most 25 x 3 = 75."""
    rng = np.random.default_rng(seed)
    nodes = max(16, int(PUBLISHED["ldoor"]["rows"] * scale) // 3)
    w = max(4, int(np.sqrt(nodes) / 2))  # nodes per mesh line
    deg = np.clip(np.round(rng.normal(11.2, 2.6, size=nodes)), 4, 22).astype(np.int64)  # draws per node (before symmetrising)
    src = np.repeat(np.arange(nodes), deg)
    # neighbours: offsets (dx, dline) with |dx| <= 2, |dline| <= 2 lines -> index distance up to ~2w; draws that leave the
    # mesh are dropped (boundary nodes have fewer neighbours)
    dx = rng.integers(-2, 3, size=len(src))
    dl = rng.integers(-2, 3, size=len(src))
    dst = src + dx + dl * w
    inside = (dst >= 0) & (dst < nodes) & ((src % w) + dx >= 0) & ((src % w) + dx < w)
    src, dst = src[inside], dst[inside]
    bu = np.concatenate([src, dst, np.arange(nodes)])  # symmetric pattern + diagonal block
    bv = np.concatenate([dst, src, np.arange(nodes)])
    key = np.unique(bu * nodes + bv)
    bu, bv = key // nodes, key % nodes
    # expand every node pair to a dense 3x3 block
    ri = (bu[:, None, None] * 3 + np.arange(3)[None, :, None]).repeat(3, axis=2).reshape(-1)
    ci = (bv[:, None, None] * 3 + np.arange(3)[None, None, :]).repeat(3, axis=1).reshape(-1)
    n = nodes * 3
    return _csr_from_coo(n, n, ri, ci, rng)


def nlpkkt_like(scale=1.0, seed=13):
    """[[D, A^T], [A, 0]] with A a 27-point coupling on a g^3 grid (g = 120 at scale 1): rows of the first half hold their
    diagonal and 27 columns in the SECOND half, rows of the second half 27 columns in the first -- the two gather windows
    of a row block lie half the matrix apart."""
    rng = np.random.default_rng(seed)
    g = max(3, int(round(120 * scale ** (1.0 / 3.0))))
    n1 = g ** 3
    r = np.arange(n1)
    ix, iy, iz = r % g, (r // g) % g, r // (g * g)
    ri_l, ci_l = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (ix + dx >= 0) & (ix + dx < g) & (iy + dy >= 0) & (iy + dy < g) & (iz + dz >= 0) & (iz + dz < g)
                ri_l.append(r[ok])
                ci_l.append((r + dx + dy * g + dz * g * g)[ok])
    ar, ac = np.concatenate(ri_l), np.concatenate(ci_l)
    n = 2 * n1
    ri = np.concatenate([r, ac, ar + n1])        # D ; A^T (row = column of A, col = n1 + row of A) ; A
    ci = np.concatenate([r, ar + n1, ac])
    return _csr_from_coo(n, n, ri, ci, rng)


def _match_extremes(Ap, Aj, Ax, want_min, want_max, seed, picks=48):
    """Give the stand-in the collection's published row-length EXTREMES (the generators missed them -- ldoor-like min 6
    for 28, thermal2-like min 2 for 1, nlpkkt-like min 8 for 5): `picks` of the shortest rows are cut down to `want_min` entries
    (the diagonal stays), `picks` of the longest rows are filled up to `want_max` with columns next to their own (new columns,
    ascending order kept).  A few dozen rows out of millions: the mean does not move, the kernels see the published range."""
    rng = np.random.default_rng(seed)
    rows = len(Ap) - 1
    lens = np.diff(Ap).astype(np.int64)
    ri = np.repeat(np.arange(rows, dtype=np.int64), lens)
    keep = np.ones(len(Aj), bool)
    add_r, add_c = [], []
    if lens.min() > want_min:
        order = np.argsort(lens, kind="stable")[:picks]
        for r in order:
            a, b = int(Ap[r]), int(Ap[r + 1])
            cols = Aj[a:b]
            drop = [k for k in range(a, b) if cols[k - a] != r]          # never the diagonal
            n_drop = (b - a) - want_min
            for k in drop[len(drop) - n_drop:]:
                keep[k] = False
    fill = [(int(r), want_min) for r in np.nonzero(lens < want_min)[0]]           # rows below the published minimum: filled up to it
    if lens.max() < want_max:
        fill += [(int(r), want_max) for r in np.argsort(-lens, kind="stable")[:picks]]
    if fill:
        for r, target in fill:
            have = set(int(c) for c in Aj[Ap[r]:Ap[r + 1]])
            need = target - len(have)
            c = int(r)
            step = 1
            while need > 0 and step < 10 * want_max:
                for cand in (c + step, c - step):
                    if need > 0 and 0 <= cand < rows and cand not in have:
                        have.add(cand); add_r.append(int(r)); add_c.append(cand); need -= 1
                step += 1
    if keep.all() and not add_r:
        return Ap, Aj, Ax
    r2 = np.concatenate([ri[keep], np.array(add_r, np.int64)])
    c2 = np.concatenate([Aj[keep].astype(np.int64), np.array(add_c, np.int64)])
    v2 = np.concatenate([Ax[keep], rng.standard_normal(len(add_r))])
    order = np.lexsort((c2, r2))
    r2, c2, v2 = r2[order], c2[order], v2[order]
    Ap2 = np.zeros(rows + 1, np.int64)
    np.add.at(Ap2, r2 + 1, 1)
    return np.cumsum(Ap2).astype(np.int32), c2.astype(np.int32), v2


def _with_extremes(gen, name):
    def run(scale=1.0, seed=None):
        Ap, Aj, Ax = gen(scale) if seed is None else gen(scale, seed)
        p = PUBLISHED[name]
        return _match_extremes(Ap, Aj, Ax, p["min"], p["max"], 1000 + len(name))
    run.__name__ = gen.__name__
    return run


GENERATORS = {"thermal2": _with_extremes(thermal2_like, "thermal2"), "ldoor": _with_extremes(ldoor_like, "ldoor"),
              "nlpkkt120": _with_extremes(nlpkkt_like, "nlpkkt120")}


def load(name, scale=1.0):
    """(Ap, Aj, Ax, source): the real matrix when CMI_SUITESPARSE_DIR/<name>.mtx (or <name>/<name>.mtx) exists -- dimensions
    then come from the file -- else the seeded stand-in; `source` says which."""
    d = os.environ.get("CMI_SUITESPARSE_DIR", "")
    for cand in (os.path.join(d, name + ".mtx"), os.path.join(d, name, name + ".mtx")):
        if d and os.path.exists(cand):
            import scipy.io
            import scipy.sparse
            M = scipy.sparse.csr_matrix(scipy.io.mmread(cand))
            M.sort_indices()
            return M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64), f"file {cand}"
    Ap, Aj, Ax = GENERATORS[name](scale)
    return Ap, Aj, Ax, f"seeded stand-in {GENERATORS[name].__name__}(scale={scale}) -- CMI_SUITESPARSE_DIR has no {name}.mtx"


def stats(Ap, Aj):
    lens = np.diff(Ap)
    rows = len(lens)
    ri = np.repeat(np.arange(rows), lens)
    return {"rows": rows, "entries": int(Ap[-1]), "min": int(lens.min()), "mean": float(lens.mean()), "max": int(lens.max()),
            "bandwidth": int(np.abs(Aj.astype(np.int64) - ri).max()) if len(Aj) else 0}


if __name__ == "__main__":
    import sys
    sc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
    for nm in GENERATORS:
        Ap, Aj, Ax, src = load(nm, sc)
        print(nm, stats(Ap, Aj), "published", PUBLISHED[nm], "|", src)
