"""Synthetic workloads of BASELINE.json / SURVEY.md 8(d) (numpy only; CCS with int64 indices)."""
import numpy as np


def _tril_ccs_from_coo(n, rows, cols, vals):
    """Lower-triangular CCS (sorted rows, duplicates summed) from COO triplets with rows >= cols."""
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    key = cols.astype(np.int64) * n + rows
    uniq, start = np.unique(key, return_index=True)
    v = np.add.reduceat(vals, start) if vals.size else vals
    r = (uniq % n).astype(np.int64)
    c = (uniq // n).astype(np.int64)
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, c + 1, 1)
    np.cumsum(colptr, out=colptr)
    return colptr, r, v.astype(np.float64)


def laplacian_2d(g, h=None):
    """5-point Laplacian A = I(x)T + T(x)I on a g x h grid, lower triangle as CCS (config 2:
    g = h = 1000 -> n = 1e6, 2 998 000 stored entries)."""
    h = g if h is None else h
    n = g * h
    idx = np.arange(n, dtype=np.int64)
    x = idx % g
    rows = [idx, idx[x + 1 < g] + 1, idx[idx + g < n] + g]
    cols = [idx, idx[x + 1 < g], idx[idx + g < n]]
    vals = [np.full(n, 4.0), np.full(rows[1].size, -1.0), np.full(rows[2].size, -1.0)]
    return (n,) + _tril_ccs_from_coo(n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


def laplacian_3d(g, h=None, d=None):
    """7-point Laplacian on a g x h x d grid (a g^3 cube by default), lower triangle (config 5 uses g = 200)."""
    h = g if h is None else h
    d = g if d is None else d
    n = g * h * d
    idx = np.arange(n, dtype=np.int64)
    x = idx % g
    y = (idx // g) % h
    m1, m2, m3 = x + 1 < g, y + 1 < h, idx + g * h < n
    rows = np.concatenate([idx, idx[m1] + 1, idx[m2] + g, idx[m3] + g * h])
    cols = np.concatenate([idx, idx[m1], idx[m2], idx[m3]])
    vals = np.concatenate([np.full(n, 6.0), np.full(m1.sum() + m2.sum() + m3.sum(), -1.0)])
    return (n,) + _tril_ccs_from_coo(n, rows, cols, vals)


def stencil21_2d(g, seed=20):
    """'~20 nnz/row' north-star variant: radius-2 box+cross stencil (21 points) on a g x g grid,
    random negative couplings, diagonally dominant (a_ii = 1 + sum |a_ij|), seed 20."""
    n = g * g
    rng = np.random.default_rng(seed)
    idx = np.arange(n, dtype=np.int64)
    x, y = idx % g, idx // g
    # 5x5 box minus its 4 corners = 21 points; keep the half with the larger node index
    offs = [(dx, dy) for dy in range(0, 3) for dx in range(-2, 3)
            if (dy > 0 or dx > 0) and not (abs(dx) == 2 and abs(dy) == 2)]
    rows, cols, vals = [], [], []
    diag = np.ones(n)
    for dx, dy in offs:
        m = (x + dx >= 0) & (x + dx < g) & (y + dy < g)
        a = idx[m]
        b = a + dx + dy * g
        v = -rng.uniform(0.1, 1.0, a.size)
        rows.append(b)
        cols.append(a)
        vals.append(v)
        np.add.at(diag, a, -v)
        np.add.at(diag, b, -v)
    rows.append(idx)
    cols.append(idx)
    vals.append(diag)
    return (n,) + _tril_ccs_from_coo(n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


def sym_matvec(n, colptr, rowind, values, X):
    """Y = A X for a symmetric matrix stored as its lower triangle (numpy; test/bench helper)."""
    X2 = X.reshape(n, -1)
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(colptr))
    Y = np.zeros_like(X2)
    np.add.at(Y, rowind, values[:, None] * X2[cols])
    off = rowind != cols
    np.add.at(Y, cols[off], values[off, None] * X2[rowind[off]])
    return Y.reshape(X.shape)


def lp_grid(gx, gy, seed=4):
    """Config-4b structured LP: variables on a gx x gy grid (n = gx*gy), G = M with
    4 stacked n-row blocks; row j of block t has one entry at node j and one at its neighbour in
    direction t (down/up/right/left; at the boundary the neighbour is j itself and the duplicate
    is summed).  Strictly feasible by construction.  Returns dict with G as CCS (ml x n), c, h."""
    n = gx * gy
    rng = np.random.default_rng(seed)
    idx = np.arange(n, dtype=np.int64)
    x, y = idx % gx, idx // gx
    nb = [np.where(y + 1 < gy, idx + gx, idx), np.where(y > 0, idx - gx, idx),
          np.where(x + 1 < gx, idx + 1, idx), np.where(x > 0, idx - 1, idx)]
    rows, cols, vals = [], [], []
    for t in range(4):
        own = rng.uniform(0.5, 1.5, n) * rng.choice([-1.0, 1.0], n)
        oth = rng.uniform(0.5, 1.5, n) * rng.choice([-1.0, 1.0], n)
        rows += [t * n + idx, t * n + idx]
        cols += [idx, nb[t]]
        vals += [own, oth]
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    ml = 4 * n
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    key = cols * ml + rows
    uniq, start = np.unique(key, return_index=True)
    v = np.add.reduceat(vals, start)
    r, c = uniq % ml, uniq // ml
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, c + 1, 1)
    np.cumsum(colptr, out=colptr)
    x0 = rng.standard_normal(n)
    s0 = rng.uniform(0.5, 1.5, ml)
    z0 = rng.uniform(0.5, 1.5, ml)
    Gx0 = np.zeros(ml)
    np.add.at(Gx0, r, v * x0[c])
    h = Gx0 + s0
    cvec = np.zeros(n)
    np.add.at(cvec, c, -v * z0[r])
    return {"ml": ml, "n": n, "Gp": colptr, "Gi": r.astype(np.int64), "Gx": v, "c": cvec, "h": h, "x0": x0, "s0": s0, "z0": z0}


def lp_grid_eq(gx, gy, p, seed=9):
    """lp_grid plus p equality constraints A x = b (goldens G8): general G AND equality rows, the branch of
    misc.kkt_chol2 with K = A S^-1 A' and a non-diagonal S (misc.py:1476-1487).  Row i of A has three entries N(0, 1) in
    random columns; b = A x0 with lp_grid's interior point x0, so the problem stays strictly feasible."""
    L = lp_grid(gx, gy)
    n = L["n"]
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(p, dtype=np.int64), 3)
    cols = np.concatenate([rng.choice(n, 3, replace=False) for _ in range(p)]).astype(np.int64)
    vals = rng.standard_normal(3 * p)
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.add.at(Ap, cols + 1, 1)
    np.cumsum(Ap, out=Ap)
    b = np.zeros(p)
    np.add.at(b, rows, vals * L["x0"][cols])
    L.update({"p": p, "Ap": Ap, "Ai": rows, "Ax": vals, "b": b})
    return L


def qp_grid(gx, gy):
    """Convex QP on the config-4b grid (goldens G5): G, h of lp_grid (strictly feasible), P = I + (5-point grid
    Laplacian) / 4 as lower CCS, q = default_rng(8) standard normal."""
    L = lp_grid(gx, gy)
    n, cp, ri, vx = laplacian_2d(gx, gy)
    Px = vx * 0.25
    Px[cp[:-1]] += 1.0
    L.update({"Pp": cp, "Pi": ri, "Px": Px, "q": np.random.default_rng(8).standard_normal(n)})
    return L


def lp_grid_std(gx, gy):
    """Config 4a: the literal "m = 50k, n = 200k" standard-form LP at grid size gx x gy --
    minimise c'x  s.t.  A x = b, x >= 0  with A = M' (n_grid x 4 n_grid, M of lp_grid), G = -I, h = 0.
    x0, z0 ~ U(0.5, 1.5) (seeds 5, 6), y0 ~ N(0, 1) (seed 7), b = A x0, c = z0 - M y0: strictly feasible.
    Returns dict with A as CCS (p x nv), G as CCS (nv x nv, -I), c, h, b."""
    L = lp_grid(gx, gy)
    p, nv = L["n"], L["ml"]                            # equality rows = grid nodes, variables = rows of M
    # A = M' : CCS of A (p x nv) = CSR of M; build from M's CCS (ml x n)
    Mi, Mx = L["Gi"], L["Gx"]
    Mcols = np.repeat(np.arange(L["n"], dtype=np.int64), np.diff(L["Gp"]))
    order = np.lexsort((Mcols, Mi))                    # A column j = M row j, entries sorted by A-row (= M column)
    Ai, Ax, Aj = Mcols[order], Mx[order], Mi[order]
    Ap = np.zeros(nv + 1, dtype=np.int64)
    np.add.at(Ap, Aj + 1, 1)
    np.cumsum(Ap, out=Ap)
    x0 = np.random.default_rng(5).uniform(0.5, 1.5, nv)
    z0 = np.random.default_rng(6).uniform(0.5, 1.5, nv)
    y0 = np.random.default_rng(7).standard_normal(p)
    b = np.zeros(p); np.add.at(b, Ai, Ax * x0[Aj])
    My0 = np.zeros(nv); np.add.at(My0, Aj, Ax * y0[Ai])      # M y0 = A' y0
    c = z0 - My0
    Gp = np.arange(nv + 1, dtype=np.int64)
    return {"p": p, "n": nv, "ml": nv, "Ap": Ap, "Ai": Ai.astype(np.int64), "Ax": Ax, "b": b, "c": c,
            "Gp": Gp, "Gi": np.arange(nv, dtype=np.int64), "Gx": -np.ones(nv), "h": np.zeros(nv)}


def convdiff_2d(g, seed=0):
    """Unsymmetric 5-point convection-diffusion operator on a g x g grid with a random local wind (full CCS, int64):
    the LU counterpart of laplacian_2d -- structurally symmetric, numerically unsymmetric, not diagonally dominant."""
    rng = np.random.default_rng(seed)
    n = g * g
    idx = np.arange(n, dtype=np.int64).reshape(g, g)
    I, J, V = [idx.ravel()], [idx.ravel()], [4.0 + 0.1 * rng.random(n)]
    for a, b, s in ((idx[1:, :], idx[:-1, :], 1.0), (idx[:-1, :], idx[1:, :], -1.0), (idx[:, 1:], idx[:, :-1], 1.0), (idx[:, :-1], idx[:, 1:], -1.0)):
        w = rng.standard_normal(a.size) * 0.8
        I.append(a.ravel()); J.append(b.ravel()); V.append(-1.0 + s * w)
    I, J, V = np.concatenate(I), np.concatenate(J), np.concatenate(V)
    order = np.lexsort((I, J))
    I, J, V = I[order], J[order], V[order]
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, J + 1, 1)
    np.cumsum(colptr, out=colptr)
    return n, colptr, I.astype(np.int64), V
