"""Pins the CPU oracle (oracle/kvx_oracle.c) before anything trusts it:
known answers from the reference's documentation (doc/source/spsolvers.rst:555-563, 580-585,
700-708, 759-772), dense numpy cross-checks, and the reference's own test matrices."""
import os

import numpy as np
import pytest

from oracle.kvx_oracle import OracleChol, atda, spmv

# spsolvers.rst:556: A = spmatrix([10,3,5,-2,5,2], [0,2,1,3,2,3], [0,0,1,1,2,3])
DOC_P = np.array([0, 2, 4, 5, 6])
DOC_I = np.array([0, 2, 1, 3, 2, 3])
DOC_V = np.array([10.0, 3, 5, -2, 5, 2])
DOC_DENSE = np.array([[10, 0, 3, 0], [0, 5, 0, -2], [3, 0, 5, 0], [0, -2, 0, 2.0]])


def test_doc_linsolve_known_answer():
    F = OracleChol(4, DOC_P, DOC_I)
    F.factorize(DOC_V)
    X = np.asfortranarray(np.arange(8, dtype=float).reshape(4, 2, order="F"))
    F.solve(X)
    # printed to 3 significant digits at spsolvers.rst:560-563 / 705-708
    doc = np.array([[-1.46e-01, 4.88e-02], [1.33e+00, 4.00e+00], [4.88e-01, 1.17e+00], [2.83e+00, 7.50e+00]])
    assert np.allclose(X, doc, rtol=5e-3)
    assert np.allclose(DOC_DENSE @ X, np.arange(8, dtype=float).reshape(4, 2, order="F"), atol=1e-13)


def test_doc_inverse_known_answer():
    F = OracleChol(4, DOC_P, DOC_I)
    F.factorize(DOC_V)
    X = np.asfortranarray(np.eye(4))
    F.solve(X)
    doc = np.array([[1.22e-01, 0, -7.32e-02, 0], [0, 3.33e-01, 0, 3.33e-01],
                    [-7.32e-02, 0, 2.44e-01, 0], [0, 3.33e-01, 0, 8.33e-01]])   # spsolvers.rst:582-585
    assert np.allclose(X, doc, atol=6e-4)


def test_doc_logdet_known_answer():
    F = OracleChol(4, DOC_P, DOC_I)
    F.factorize(DOC_V)
    assert abs(2.0 * np.sum(np.log(F.diag())) - 5.50533153593) < 1e-10     # spsolvers.rst:765


def _rand_spd(n, dens, seed):
    import scipy.sparse as sp
    M = sp.random(n, n, dens, random_state=seed, format="csc")
    S = (M @ M.T + sp.eye(n) * (1.0 + seed)).tocsc()
    L = sp.tril(S).tocsc()
    L.sort_indices()
    return S, L


@pytest.mark.parametrize("uplo", ["L", "U"])
def test_against_dense_cholesky(uplo):
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    S, L = _rand_spd(150, 0.04, 2)
    p = rng.permutation(150)
    T = L if uplo == "L" else sp.triu(S).tocsc()
    T.sort_indices()
    F = OracleChol(150, T.indptr, T.indices, uplo, p)
    F.factorize(T.data)
    Ld = np.linalg.cholesky(S.toarray()[np.ix_(p, p)])
    Lp, Li, Lx = F.L()
    Lo = np.zeros((150, 150))
    for j in range(150):
        Lo[Li[Lp[j]:Lp[j + 1]], j] = Lx[Lp[j]:Lp[j + 1]]
    assert np.abs(Lo - Ld).max() < 1e-12 * np.abs(Ld).max()
    assert F.lnz == np.count_nonzero(np.abs(Ld) > 1e-300)
    b = rng.standard_normal(150)
    for sys_, ref in ((4, np.linalg.solve(Ld, b)), (5, np.linalg.solve(Ld.T, b)), (7, b[p]), (1, np.linalg.solve(Ld @ Ld.T, b))):
        x = b.copy()
        F.solve(x, sys=sys_)
        assert np.allclose(x, ref, rtol=1e-10, atol=1e-12)
    x = b.copy(); F.solve(x, sys=7); F.solve(x, sys=8)
    assert np.array_equal(x, b)


def test_other_triangle_is_ignored_and_errors():
    # cholmod.c:137-157: entries in the other triangle are silently ignored
    import scipy.sparse as sp
    S, L = _rand_spd(40, 0.1, 5)
    full = S.tocsc(); full.sort_indices()
    junk = full.copy(); junk.data = junk.data.copy()
    up = junk.indices < np.repeat(np.arange(40), np.diff(junk.indptr))
    junk.data[up] = 1e9
    F = OracleChol(40, junk.indptr, junk.indices, "L")
    F.factorize(junk.data)
    b = np.ones(40); x = b.copy(); F.solve(x)
    assert np.linalg.norm(S @ x - b) < 1e-10
    with pytest.raises(ValueError):
        OracleChol(3, [0, 1, 2, 3], [0, 1, 2], "L", [0, 0, 1])
    # not positive definite -> ArithmeticError(minor)
    F = OracleChol(2, [0, 2, 3], [0, 1, 1], "L")
    with pytest.raises(ArithmeticError) as e:
        F.factorize([1.0, 2.0, 1.0])
    assert e.value.args[0] == 1
    with pytest.raises(ArithmeticError):
        F.solve(np.ones(2))
    with pytest.raises(ValueError):
        OracleChol(2, [0, 2, 3], [0, 1, 1]).solve(np.ones(2))     # symbolic factor


@pytest.mark.parametrize("name", ["bcsstk13", "bcsstk24"])
def test_reference_test_matrices(golden_dir, name):
    """Residual test in the style of tests/test_sparse_solvers.py:239-259 on the reference's own
    matrices (lower triangle as stored, uplo='L'); bcsstk13 has cond ~1.1e10."""
    from kvxopt_amd.workloads import sym_matvec
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n, cp, ri, v = int(z["n"]), z["colptr"], z["rowind"], z["values"]
    B = np.random.default_rng(13).standard_normal((n, 3))
    X = np.asfortranarray(B.copy())
    F = OracleChol(n, cp, ri, "L")
    F.factorize(v)
    F.solve(X)
    R = sym_matvec(n, cp, ri, v, X) - B
    nrmA = np.sqrt(2 * np.sum(v ** 2) - np.sum(v[ri == np.repeat(np.arange(n), np.diff(cp))] ** 2))
    backward = np.linalg.norm(R) / (nrmA * np.linalg.norm(X) + np.linalg.norm(B))
    assert backward < 1e-14
    assert np.linalg.norm(R) / np.linalg.norm(B) < 5e-8


def test_atda_and_spmv_against_dense():
    import scipy.sparse as sp
    rng = np.random.default_rng(7)
    G = sp.random(40, 12, 0.2, random_state=3, format="csc"); G.sort_indices()
    di = rng.uniform(0.5, 2, 40)
    Sd = (G.T @ sp.diags(di ** 2) @ G).toarray()
    pat = sp.tril((abs(G).T @ abs(G))).tocsc(); pat.sort_indices()
    Gs, Sx = atda(40, 12, G.indptr, G.indices, G.data, di, pat.indptr, pat.indices)
    ref = Sd[pat.indices, np.repeat(np.arange(12), np.diff(pat.indptr))]
    assert np.allclose(Sx, ref, rtol=1e-13, atol=1e-14)
    x = rng.standard_normal(12); y = rng.standard_normal(40)
    y2 = y.copy(); spmv("N", 40, 12, G.indptr, G.indices, G.data, x, y2, 2.0, -0.5)
    assert np.allclose(y2, 2 * (G @ x) - 0.5 * y)
    x2 = x.copy(); spmv("T", 40, 12, G.indptr, G.indices, G.data, y, x2, -1.0, 1.0)
    assert np.allclose(x2, x - G.T @ y)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_host_supernodal_baseline_matches_the_simplicial_oracle(threads):
    """oracle/kvx_supernodal.c (the all-cores CPU baseline of bench.py): same solutions as the pinned simplicial oracle on
    the GPU library's own supernodes and permutation, for any thread count; a non-positive pivot is reported with the
    oracle's column."""
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    from oracle.kvx_oracle import OracleChol, OracleSupernodal
    for (n, cp, ri, v) in (workloads.laplacian_2d(70, 41), workloads.laplacian_3d(12)):
        F = Factor(n, cp, ri)                                    # host-side analysis only
        O = OracleChol(n, cp, ri, "L", F.perm()); O.factorize(v)
        S = OracleSupernodal.from_factor(n, cp, ri, F, threads=threads); S.factorize(v)
        B = np.random.default_rng(3).standard_normal((n, 3))
        x1 = np.asfortranarray(B.copy()); O.solve(x1)
        x2 = np.asfortranarray(B.copy()); S.solve(x2)
        assert np.abs(x1 - x2).max() / np.abs(x1).max() < 1e-12
        bad = v.copy(); bad[cp[int(F.perm()[n // 2])]] = -3.0
        with pytest.raises(ArithmeticError) as eo:
            O.factorize(bad)
        with pytest.raises(ArithmeticError) as es:
            S.factorize(bad)
        assert es.value.args[0] == eo.value.args[0]
        S.factorize(v)
        x3 = np.asfortranarray(B.copy()); S.solve(x3)
        assert np.abs(x3 - x1).max() / np.abs(x1).max() < 1e-12


def test_cpu_conelp_restatement_reproduces_the_reference_traces(golden_dir):
    """oracle/lp_oracle.py (the CPU baseline of the IPM metric) against golden G4 from the reference's conelp: same iteration
    counts, same solutions, same certificates -- over the simplicial oracle and over the host supernodal Cholesky."""
    import json
    import scipy.sparse as sp
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    from oracle import lp_oracle
    g = np.load(os.path.join(golden_dir, "g4_conelp.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_conelp.json")))["cases"]
    for name, gx, gy in (("grid6x5", 6, 5), ("grid25x20", 25, 20)):
        P = workloads.lp_grid(gx, gy)
        G = sp.csc_matrix((P["Gx"], P["Gi"], P["Gp"]), shape=(P["ml"], P["n"]))
        Spat = sp.tril((abs(G.T) @ abs(G)).tocsc()).tocsc(); Spat.sort_indices()
        Sp, Si = Spat.indptr.astype(np.int64), Spat.indices.astype(np.int64)
        Fs = Factor(P["n"], Sp, Si)
        sup, nrows, parent, level = Fs.supernodes()
        rp, ri = Fs.front_rows()
        for structure in (None, (Fs.perm(), sup, rp, ri, parent, Sp, Si)):
            sol = lp_oracle.conelp_l(P["c"], P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"], P["h"], structure=structure, threads=2)
            assert sol["status"] == meta[name]["status"] == "optimal" and sol["iterations"] == meta[name]["iterations"]
            assert np.abs(sol["x"] - g[name + "_x"]).max() < 1e-9 and np.abs(sol["z"] - g[name + "_z"]).max() < 1e-9
            assert abs(sol["primal objective"] - meta[name]["primal objective"]) < 1e-9 * max(1.0, abs(meta[name]["primal objective"]))

    def ccs(V, I, J, shape):
        A = sp.csc_matrix((V, (I, J)), shape=shape); A.sort_indices()
        return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data
    Gp, Gi, Gx = ccs([-1.0, 1.0], [0, 1], [0, 0], (2, 1))
    sol = lp_oracle.conelp_l(np.array([1.0]), 2, 1, Gp, Gi, Gx, np.array([-1.0, 0.0]))
    assert sol["status"] == "primal infeasible" and sol["iterations"] == meta["primal_infeasible"]["iterations"]
    assert np.allclose(sol["z"], meta["primal_infeasible"]["z"], atol=1e-9)
    Gp, Gi, Gx = ccs([-1.0, -1.0], [0, 1], [0, 1], (2, 2))
    sol = lp_oracle.conelp_l(np.array([-1.0, 0.5]), 2, 2, Gp, Gi, Gx, np.array([0.0, 0.0]))
    assert sol["status"] == "dual infeasible" and sol["iterations"] == meta["dual_infeasible"]["iterations"]
    assert np.allclose(sol["x"], meta["dual_infeasible"]["x"], rtol=1e-9)


def _blocks(vec, off, dims):
    out, o = [], off
    for m in dims:
        out.append(vec[o:o + m * m].reshape((m, m), order="F"))
        o += m * m
    return out


def test_cone_block_restatements_against_the_reference_goldens():
    """oracle/kvx_oracle.py restates the 'q' and 's' parts of the Nesterov-Todd scaling (misc.py:290-419, 582-634;
    misc_solvers.c:144-240, 343-397, 671-770, 845-882, 1029-1160) in numpy; pinned here on the goldens G15 / G16 the reference
    itself produced (values to 1e-12; the singular vectors of an 's' block up to the sign of each column, so r and rti are
    compared through r r', rti rti' and through scale() with the reference's own r)."""
    import os
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    from oracle import kvx_oracle as orc
    rel = lambda a, b: np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300)
    # --- 'q' (G15): dims = {'l': 4, 'q': [5, 1, 9]}
    g = np.load(os.path.join(gd, "g15_q_cone_scaling.npz"))
    for tag, k in (("a", 0), ("b", 3)):
        ind, iv = k + 4, 0
        for m in (5, 1, 9):
            v, beta, lm = orc.compute_scaling_q(g[tag + "_s"][ind:ind + m].copy(), g[tag + "_z"][ind:ind + m].copy())
            assert rel(v, g[tag + "_v"][iv:iv + m]) < 1e-12 and rel(lm, g[tag + "_lmbda"][ind:ind + m]) < 1e-12
            bidx = [5, 1, 9].index(m)
            assert abs(beta - g[tag + "_beta"][bidx]) < 1e-12 * beta
            for inv in "NI":
                for c in range(2):
                    got = orc.scale_q(g[tag + "_X"][ind:ind + m, c], v, beta, inv)
                    assert rel(got, g["%s_scale_N%s" % (tag, inv)][ind:ind + m, c]) < 1e-12
            assert rel(orc.sprod_q(g[tag + "_x1"][ind:ind + m], g[tag + "_y1"][ind:ind + m]), g[tag + "_sprod"][ind:ind + m]) < 1e-12
            ind += m
            iv += m
    # --- 's' (G16): dims = {'l': 3, 'q': [4], 's': [3, 1, 6]}
    g = np.load(os.path.join(gd, "g16_s_cone_scaling.npz"))
    sd = [3, 1, 6]
    for tag, k in (("a", 0), ("b", 2)):
        nlq = k + 3 + 4
        Sb, Zb = _blocks(g[tag + "_s"], nlq, sd), _blocks(g[tag + "_z"], nlq, sd)
        Rb, Tb = _blocks(g[tag + "_r"], 0, sd), _blocks(g[tag + "_rti"], 0, sd)
        X1, Y1 = _blocks(g[tag + "_x1"], nlq, sd), _blocks(g[tag + "_y1"], nlq, sd)
        il, sdot, tmax = nlq, float(np.dot(g[tag + "_x1"][:nlq], g[tag + "_y1"][:nlq])), -np.inf
        for i, m in enumerate(sd):
            r, rti, lm = orc.compute_scaling_s(Sb[i], Zb[i])
            assert rel(lm, g[tag + "_lmbda"][il:il + m]) < 1e-12
            assert rel(r @ r.T, Rb[i] @ Rb[i].T) < 1e-11 and rel(rti @ rti.T, Tb[i] @ Tb[i].T) < 1e-11
            for tr in "NT":
                for inv in "NI":
                    for c in range(2):
                        Xc = _blocks(g[tag + "_X"][:, c], nlq, sd)[i]
                        want = _blocks(g["%s_scale_%s%s" % (tag, tr, inv)][:, c], nlq, sd)[i]
                        assert rel(orc.scale_s(Xc, Rb[i], Tb[i], tr, inv), want) < 1e-12, (tr, inv)
            lmk = g[tag + "_lmbda"][il:il + m]
            assert rel(orc.scale2_s(lmk, X1[i], "N"), _blocks(g[tag + "_scale2_N"], nlq, sd)[i]) < 1e-12
            assert rel(orc.scale2_s(lmk, X1[i], "I"), _blocks(g[tag + "_scale2_I"], nlq, sd)[i]) < 1e-12
            assert rel(orc.sprod_s(X1[i], Y1[i]), _blocks(g[tag + "_sprod"], nlq, sd)[i]) < 1e-12
            yd = g[tag + "_yd"][il:il + m]
            assert rel(orc.sprod_s(X1[i], yd, "D"), _blocks(g[tag + "_sprod_D"], nlq, sd)[i]) < 1e-12
            assert rel(orc.sinv_s(X1[i], yd), _blocks(g[tag + "_sinv"], nlq, sd)[i]) < 1e-12
            sdot += orc.sdot_s(X1[i], Y1[i])
            t, ev = orc.max_step_s(X1[i])
            tmax = max(tmax, t)
            assert rel(ev, g[tag + "_max_step_sigma"][il - nlq:il - nlq + m]) < 1e-12
            il += m
        assert abs(sdot - float(g[tag + "_sdot"])) < 1e-12 * abs(sdot)
        x1 = g[tag + "_x1"]
        tq = orc.max_step_q(x1[k + 3:k + 7])
        assert abs(max(tmax, tq, float(np.max(-x1[:k + 3]))) - float(g[tag + "_max_step"])) < 1e-12
        packed = np.concatenate([orc.pack_s(B) for B in X1])
        assert np.array_equal(packed, g[tag + "_pack"][2 + nlq:2 + nlq + packed.size])
        # update_scaling: lambda and the new scaling point through r r', rti rti'
        Ls, Lz = _blocks(g[tag + "_us_s_in"], nlq, sd), _blocks(g[tag + "_us_z_in"], nlq, sd)
        Rn, Tn = _blocks(g[tag + "_us_r"], 0, sd), _blocks(g[tag + "_us_rti"], 0, sd)
        il = nlq
        for i, m in enumerate(sd):
            r2, t2, lm2 = orc.update_scaling_s(Rb[i], Tb[i], Ls[i], Lz[i])
            assert rel(lm2, g[tag + "_us_lmbda"][il:il + m]) < 1e-12
            assert rel(r2 @ r2.T, Rn[i] @ Rn[i].T) < 1e-11 and rel(t2 @ t2.T, Tn[i] @ Tn[i].T) < 1e-11
            il += m
