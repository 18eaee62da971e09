"""CPU-side checks: the oracle's numpy restatements against the reference goldens, the host
containers, and argument/option validation of the cholmod mirror that needs no GPU."""
import os

import numpy as np
import pytest

from kvxopt_amd import base, cholmod, workloads
from kvxopt_amd.base import matrix, spmatrix
from oracle import kvx_oracle as orc


@pytest.mark.parametrize("ml", [1, 7, 1000])
def test_oracle_nt_restatement_vs_reference(golden_dir, ml):
    g = np.load(os.path.join(golden_dir, "g1_nt_scaling.npz"))
    k = lambda name: g["ml%d_%s" % (ml, name)]
    d, di, lm = orc.compute_scaling_l(k("s"), k("z"))
    assert np.allclose(d, k("d"), rtol=1e-15) and np.allclose(di, k("di"), rtol=1e-15) and np.allclose(lm, k("lmbda"), rtol=1e-15)
    for tr in "NT":
        for inv in "NI":
            x = k("X").copy()
            orc.scale_l(x, d, di, inv)
            assert np.allclose(x, k("scale_%s%s" % (tr, inv)), rtol=1e-15)
    assert np.allclose(orc.scale2_l(lm, k("x1").copy()), k("scale2_N"), rtol=1e-15)
    assert np.allclose(orc.scale2_l(lm, k("x1").copy(), "I"), k("scale2_I"), rtol=1e-15)
    assert np.allclose(orc.sprod_l(k("x1").copy(), k("y1")), k("sprod"), rtol=1e-15)
    assert np.allclose(orc.sinv_l(k("x1").copy(), k("y1")), k("sinv"), rtol=1e-15)
    assert np.allclose(orc.ssqr_l(k("x1")), k("ssqr"), rtol=1e-15)
    assert abs(orc.sdot_l(k("x1"), k("y1")) - float(k("sdot"))) < 1e-12 * max(1, abs(float(k("sdot"))))
    assert orc.max_step_l(k("x1")) == float(k("max_step"))
    dd, lm2, s, z = k("d").copy(), k("lmbda").copy(), k("us_ds").copy(), k("us_dz").copy()
    di2 = orc.update_scaling_l(dd, lm2, s, z)
    for got, name in ((s, "us_s"), (z, "us_z"), (dd, "us_d"), (di2, "us_di"), (lm2, "us_lmbda")):
        assert np.allclose(got, k(name), rtol=1e-15), name


def test_oracle_assembly_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_assembly.npz"))
    Gs, Sx = orc.atda(40, 12, g["G_cp"], g["G_ri"], g["G_v"], g["di"], g["S_cp"], g["S_ri"])
    assert np.allclose(Gs, g["Gs_v"], rtol=1e-15) and np.allclose(Sx, g["S_v"], rtol=1e-13, atol=1e-15)
    Gs, Sx = orc.atda(40, 12, g["G_cp"], g["G_ri"], g["G_v"], g["di2"], g["S_cp"], g["S_ri"])
    assert np.allclose(Sx, g["S2_v"], rtol=1e-13, atol=1e-15)


def test_oracle_kkt_system_vs_reference(golden_dir):
    """The reduced-system algebra of misc.kkt_chol2 restated with the oracle's Cholesky reproduces the
    reference's kkt_chol2 output (p = 0)."""
    g = np.load(os.path.join(golden_dir, "g3_kkt_chol2.npz"))
    ml, n = 30, 8
    Gp, Gi, Gx = g["p0_G_cp"], g["p0_G_ri"], g["p0_G_v"]
    di = 1.0 / g["p0_d"]
    Gd = np.zeros((ml, n)); Gd[Gi, np.repeat(np.arange(n), np.diff(Gp))] = Gx
    S = Gd.T @ np.diag(di * di) @ Gd
    Sl = np.tril(S)
    cp = np.arange(0, n * n + 1, n); ri = np.tile(np.arange(n), n)
    O = orc.OracleChol(n, cp, ri, "L")
    O.factorize(Sl.reshape(-1, order="F"))
    z = di * g["p0_bz"]
    x = g["p0_bx"] + Gd.T @ (di * z)
    O.solve(x)
    z = di * (Gd @ x) - z
    assert np.allclose(x, g["p0_x"], rtol=1e-10) and np.allclose(z, g["p0_z"], rtol=1e-10)


def test_containers():
    A = spmatrix([1.0, 2.0, 3.0, 4.0], [2, 0, 2, 1], [0, 0, 0, 1], (3, 2))     # duplicate (2,0) summed, rows sorted
    assert np.array_equal(A.colptr, [0, 2, 3]) and np.array_equal(A.rowind, [0, 2, 1]) and np.array_equal(A.values, [2.0, 4.0, 4.0])
    assert A.T.size == (2, 3) and np.array_equal(A.T.todense(), A.todense().T)
    B = spmatrix([5.0], [1], [0], (3, 2))
    A += B
    assert A.todense()[1, 0] == 5.0 and len(A) == 4
    E = spmatrix([], [], [], (0, 4))
    assert E.T.size == (4, 0) and len(E.T) == 0
    m = matrix([[1.0, 2.0], [3.0, 4.0]])                      # list of columns, as kvxopt
    assert m.size == (2, 2) and m[1, 0] == 2.0 and m[2] == 3.0
    assert np.array_equal((m * 2.0).a, [[2.0, 6.0], [4.0, 8.0]])
    d = base.spdiag(matrix([1.0, 2.0, 3.0]))
    assert np.array_equal(d.todense(), np.diag([1.0, 2.0, 3.0]))
    cp, ri, v = A.CCS
    assert cp.typecode == "i" and v.typecode == "d"


def test_cholmod_argument_checks_without_gpu():
    A = spmatrix([10, 3, 5, -2, 5, 2], [0, 2, 1, 3, 2, 3], [0, 0, 1, 1, 2, 3])
    F = cholmod.symbolic(A)                                     # symbolic analysis is host-only
    assert F.name == "CHOLMOD SYM D FACTOR L" and cholmod.symbolic(A, uplo="U").name.endswith("U")
    with pytest.raises(ValueError):
        cholmod.solve(F, matrix(np.ones(4)))                    # "called with symbolic factor"
    with pytest.raises(TypeError):
        cholmod.symbolic(matrix(np.ones(4)))
    cholmod.options["postorder"] = 1                            # must be a bool (cholmod.c:113-115)
    with pytest.raises(ValueError):
        cholmod.symbolic(A)
    cholmod.options.clear()
    cholmod.options.update({"postorder": True, "print": 0, "dbound": 0.0, "nmethods": 1})
    cholmod.symbolic(A, p=matrix([3, 2, 1, 0], tc="i"))
    cholmod.options.clear()
    with pytest.raises(TypeError):
        cholmod.symbolic(A, p=matrix([0.0, 1.0, 2.0, 3.0]))
    # K = 0 x 0 (p = 0 in kkt_chol2, misc.py:1486): analysis of an empty matrix must not raise
    K = spmatrix([], [], [], (0, 0))
    cholmod.symbolic(K)


def test_lp_generator_is_feasible():
    P = workloads.lp_grid(6, 5)
    assert P["ml"] == 120 and P["n"] == 30 and P["Gp"][-1] == len(P["Gx"])
    per_row = np.bincount(P["Gi"], minlength=P["ml"])
    assert per_row.max() <= 2 and per_row.min() >= 1


def _dense_of(m, n, cp, ri, v):
    D = np.zeros((m, n)); D[ri, np.repeat(np.arange(n), np.diff(cp))] = v
    return D


@pytest.mark.parametrize("tag,p", [("p0", 0), ("p2", 2)])
def test_golden_nonlinear_block_is_the_kkt_solution(golden_dir, tag, p):
    """G12 pins the mnl > 0 branch of misc.kkt_chol2 (misc.py:1488-1500): the reference's output (ux, uy, W uz) solves
        [H A' GG'; A 0 0; GG 0 -W'W] [ux; uy; uz] = [bx; by; bz],   GG = [Df; G],  W = diag([dnl; d])
    -- checked here with a dense numpy solve, so the fixture and the restated system agree before the GPU test uses it."""
    g = np.load(os.path.join(golden_dir, "g12_nonlinear_block.npz"))
    mnl, ml, n = 5, 30, 9
    G = _dense_of(ml, n, g[tag + "_G_cp"], g[tag + "_G_ri"], g[tag + "_G_v"])
    Df = _dense_of(mnl, n, g[tag + "_Df_cp"], g[tag + "_Df_ri"], g[tag + "_Df2_v"])
    Hl = _dense_of(n, n, g[tag + "_H_cp"], g[tag + "_H_ri"], g[tag + "_H2_v"])
    H = Hl + np.tril(Hl, -1).T
    A = _dense_of(p, n, g[tag + "_A_cp"], g[tag + "_A_ri"], g[tag + "_A_v"]) if p else np.zeros((0, n))
    w = np.concatenate([g[tag + "_W1_dnl"], g[tag + "_W1_d"]])
    GG = np.vstack([Df, G])
    m = mnl + ml
    K = np.block([[H, A.T, GG.T], [A, np.zeros((p, p)), np.zeros((p, m))], [GG, np.zeros((m, p)), -np.diag(w * w)]])
    u = np.linalg.solve(K, np.concatenate([g[tag + "_bx"], g[tag + "_by"], g[tag + "_bz"]]))
    assert np.allclose(u[:n], g[tag + "_x"], rtol=1e-9, atol=1e-11)
    assert np.allclose(u[n:n + p], g[tag + "_y"], rtol=1e-9, atol=1e-11)
    assert np.allclose(w * u[n + p:], g[tag + "_z"], rtol=1e-9, atol=1e-11)


def test_object_caches_key_on_the_device_and_take_numpy_options(monkeypatch):
    """the KKT / symbolic / linsolve caches hold device objects, so the current HIP device is part of every key (a
    hit after a device switch would hand out buffers, streams and graphs of the other device), and option values of any scalar
    type (numpy integers) make a key."""
    import numpy as np
    from kvxopt_amd import _lib, lp
    lp.clear_cache()
    built = []

    def build():
        built.append(object())
        return built[-1]
    pat = [np.array([0, 1, 2]), np.array([0, 1])]
    opts = {"nd_leaf": np.int64(96), "postorder": True, "dbound": np.float64(1e-9)}
    monkeypatch.setattr(_lib, "current_device", lambda: 0)
    a = lp._kkt_for("t", (2, 2), pat, opts, build, lambda k: None)
    assert lp._kkt_for("t", (2, 2), pat, dict(opts), build, lambda k: None) is a and len(built) == 1     # same device: a hit
    monkeypatch.setattr(_lib, "current_device", lambda: 1)
    b = lp._kkt_for("t", (2, 2), pat, opts, build, lambda k: None)
    assert b is not a and len(built) == 2                                                                  # other device: a new object
    # the release hook empties all three caches (the out-of-memory retry and release_cached() go through it)
    from kvxopt_amd import cholmod, klu
    cholmod._SYMBOLIC_CACHE["x"] = 1
    klu._LINSOLVE_CACHE["x"] = 1
    for f in _lib._CACHE_CLEARERS:
        f()
    assert not lp._KKT_CACHE and not cholmod._SYMBOLIC_CACHE and not klu._LINSOLVE_CACHE
    # a MemoryError from a build releases the caches and retries once
    calls = []

    def flaky():
        calls.append(1)
        if len(calls) == 1:
            raise MemoryError("pool exhausted")
        return "ok"
    monkeypatch.setattr(_lib, "release_cached", lambda: calls.append("released"))
    assert _lib.retry_after_release(flaky) == "ok" and calls == [1, "released", 1]


def test_pattern_digest_separates_dtype_length_and_content(monkeypatch):
    """The digest behind the KKT-cache key (lp._pattern_key): arrays that differ in content, in length, in dtype, or in how a total is
    split between two arrays hash differently; the blake2b fallback (no xxhash module) behaves the same."""
    import builtins
    from kvxopt_amd import _lib, lp
    a = np.arange(12, dtype=np.int64)
    def all_distinct():
        ds = [_lib.pattern_digest(a), _lib.pattern_digest(a[:11]), _lib.pattern_digest(a.astype(np.int32)),
              _lib.pattern_digest(a[:6], a[6:]), _lib.pattern_digest(a[:5], a[5:]), _lib.pattern_digest(a + (np.arange(12) == 7)),
              _lib.pattern_digest(np.zeros(0, dtype=np.int64)), _lib.pattern_digest()]
        assert len(set(ds)) == len(ds) and all(len(d) == 16 for d in ds)
        assert _lib.pattern_digest(a) == _lib.pattern_digest(a.copy())
    all_distinct()
    assert lp._pattern_key(a, a[:3]) == lp._pattern_key(a.astype(np.int32), a[:3].tolist())     # (index arrays are taken as int64)
    real_import = builtins.__import__
    def no_xxhash(name, *args, **kw):
        if name == "xxhash":
            raise ImportError(name)
        return real_import(name, *args, **kw)
    monkeypatch.setattr(builtins, "__import__", no_xxhash)
    all_distinct()
