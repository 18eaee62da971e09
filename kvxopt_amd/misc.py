"""Orthant ('l' cone) subset of `kvxopt.misc` / `kvxopt.misc_solvers` on MI355X, plus the KKT solver
factory `kkt_chol2` -- same names, argument meaning and in-place semantics as the reference
(src/python/misc.py, src/C/misc_solvers.c).  Second-order-cone and semidefinite blocks are out of
scope (kkt_chol2 rejects them in the reference too, misc.py:1381-1384).  The nonlinear block of
cvxprog (`mnl` leading entries scaled by W['dnl'], misc.py:262-270, 432-442, 48-60) IS covered: it is
one more diagonal block in front of the 'l' block, so every kernel simply runs over mnl + ml entries.

Host-array compatibility layer: arguments are host `matrix` objects (ours or kvxopt's); every
operation runs through the HIP kernels of libkvxhip.so (upload, kernel, download).  The device-
resident fast path for whole interior-point iterations is `kvxopt_amd.lp`.
"""
import ctypes

import numpy as np

from . import _lib, base, cholmod
from ._lib import DeviceBuffer, lib, raise_for
from .base import matrix, spmatrix


def _only_l(dims, what):
    if dims.get("q") or dims.get("s"):
        raise NotImplementedError("%s: only the orthant ('l') cone is implemented on the GPU path" % what)


def _buf(x):
    return base._dense_buffer(x)


def _up(a):
    return DeviceBuffer.from_array(np.ascontiguousarray(a, dtype=np.float64))


def _sync():
    raise_for(lib().kvx_dev_sync())


def compute_scaling(s, z, lmbda, dims, mnl=None):
    """misc.py:250-287 (nonlinear and 'l' blocks): W['d'] = sqrt(s./z), W['di'] = 1./d, lmbda = sqrt(s.*z); with
    mnl given (cvxprog), the first mnl entries make W['dnl'], W['dnli'] by the same formulas."""
    _only_l(dims, "compute_scaling")
    k = 0 if mnl is None else int(mnl)
    m = k + dims["l"]
    sb, _ = _buf(s)
    zb, _ = _buf(z)
    lb, _ = _buf(lmbda)
    _lib.require_device()
    ds, dz = _up(sb[:m]), _up(zb[:m])
    dd, ddi, dl = DeviceBuffer(8 * max(m, 1)), DeviceBuffer(8 * max(m, 1)), DeviceBuffer(8 * max(m, 1))
    raise_for(lib().kvx_nt_compute_scaling_dev(m, ds.ptr, dz.ptr, dd.ptr, ddi.ptr, dl.ptr))
    _sync()
    d, di = dd.download(np.float64, m), ddi.download(np.float64, m)
    W = {}
    if mnl is not None:
        W["dnl"], W["dnli"] = matrix(d[:k].copy(), (k, 1)), matrix(di[:k].copy(), (k, 1))
    W.update({"d": matrix(d[k:].copy(), (m - k, 1)), "di": matrix(di[k:].copy(), (m - k, 1)),
              "v": [], "beta": [], "r": [], "rti": []})
    lb[:m] = dl.download(np.float64, m)
    return W


def _diag_of(W, inverse=False):
    """[dnl; d] (or the inverses) as one contiguous vector, and the length of the nonlinear part."""
    d, _ = _buf(W["di"] if inverse else W["d"])
    if "dnl" in W:
        dn, _ = _buf(W["dnli"] if inverse else W["dnl"])
        return np.concatenate([dn, d]), dn.size
    return np.ascontiguousarray(d), 0


def update_scaling(W, lmbda, s, z):
    """misc.py:422-464 (nonlinear and 'l' blocks), in place: s:=sqrt(s), z:=sqrt(z), d:=d.*s./z, di:=1./d,
    lmbda:=s.*z (W['dnl'], W['dnli'] likewise on the leading mnl entries)."""
    if W.get("v") or W.get("r"):
        raise NotImplementedError("update_scaling: only the orthant ('l') cone")
    dcat, k = _diag_of(W)
    m = dcat.size
    sb, _ = _buf(s)
    zb, _ = _buf(z)
    lb, _ = _buf(lmbda)
    _lib.require_device()
    ds, dz, dd = _up(sb[:m]), _up(zb[:m]), _up(dcat)
    ddi, dl = DeviceBuffer(8 * max(m, 1)), DeviceBuffer(8 * max(m, 1))
    raise_for(lib().kvx_nt_update_scaling_dev(m, ds.ptr, dz.ptr, dd.ptr, ddi.ptr, dl.ptr))
    _sync()
    sb[:m] = ds.download(np.float64, m)
    zb[:m] = dz.download(np.float64, m)
    d, di = dd.download(np.float64, m), ddi.download(np.float64, m)
    if k:
        _buf(W["dnl"])[0][:] = d[:k]
        _buf(W["dnli"])[0][:] = di[:k]
    _buf(W["d"])[0][:] = d[k:]
    _buf(W["di"])[0][:] = di[k:]
    lb[:m] = dl.download(np.float64, m)


def scale(x, W, trans="N", inverse="N"):
    """misc_solvers.c:85-141 / misc.py:36-82 (nonlinear and 'l' blocks): x := [dnl; d].*x ('N') or [dnli; di].*x ('I')
    for every column of x; trans is irrelevant for a diagonal scaling."""
    if W.get("v") or W.get("r"):
        raise NotImplementedError("scale: only the orthant ('l') cone")
    w, _ = _diag_of(W, inverse != "N")
    xb, size = _buf(x)
    m = w.size
    if m == 0:
        return
    _lib.require_device()
    dx, dw = _up(xb), _up(w)
    raise_for(lib().kvx_nt_scale_dev(m, size[1], size[0], dx.ptr, dw.ptr))
    _sync()
    xb[:] = dx.download(np.float64, xb.size)


def scale2(lmbda, x, dims, mnl=0, inverse="N"):
    """misc_solvers.c:256-298 ('l' block): x := x./lmbda ('N') or x.*lmbda ('I')."""
    _only_l(dims, "scale2")
    m = mnl + dims["l"]
    lb, _ = _buf(lmbda)
    xb, _ = _buf(x)
    if m == 0:
        return
    _lib.require_device()
    dx, dl = _up(xb[:m]), _up(lb[:m])
    raise_for(lib().kvx_nt_scale2_dev(m, dl.ptr, dx.ptr, 1 if inverse == "I" else 0))
    _sync()
    xb[:m] = dx.download(np.float64, m)


def _binary(kernel, x, y, m):
    xb, _ = _buf(x)
    yb, _ = _buf(y)
    if m == 0:
        return
    _lib.require_device()
    dx, dy = _up(xb[:m]), _up(yb[:m])
    raise_for(kernel(m, dx.ptr, dy.ptr))
    _sync()
    xb[:m] = dx.download(np.float64, m)


def sprod(x, y, dims, mnl=0, diag="N"):
    """misc_solvers.c:634-669 ('l' block): x := x.*y."""
    _only_l(dims, "sprod")
    _binary(lib().kvx_nt_sprod_dev, x, y, mnl + dims["l"])


def sinv(x, y, dims, mnl=0):
    """misc_solvers.c:775-800 ('l' block): x := x./y."""
    _only_l(dims, "sinv")
    _binary(lib().kvx_nt_sinv_dev, x, y, mnl + dims["l"])


def ssqr(x, y, dims, mnl=0):
    """misc.py:945-952 ('l' block): x := y.*y."""
    _only_l(dims, "ssqr")
    _binary(lib().kvx_nt_ssqr_dev, x, y, mnl + dims["l"])


def sdot(x, y, dims, mnl=0):
    """misc_solvers.c:991-1018 ('l' block): sum_i x_i*y_i."""
    _only_l(dims, "sdot")
    m = mnl + dims["l"]
    xb, _ = _buf(x)
    yb, _ = _buf(y)
    if m == 0:
        return 0.0
    _lib.require_device()
    dx, dy = _up(xb[:m]), _up(yb[:m])
    r = ctypes.c_double()
    raise_for(lib().kvx_nt_sdot_dev(m, dx.ptr, dy.ptr, ctypes.byref(r)))
    return r.value


def max_step(x, dims, mnl=0, sigma=None):
    """misc_solvers.c:1052-1071 ('l' block): max_i(-x_i)."""
    _only_l(dims, "max_step")
    m = mnl + dims["l"]
    xb, _ = _buf(x)
    if m == 0:
        return -np.finfo(np.float64).max
    _lib.require_device()
    dx = _up(xb[:m])
    r = ctypes.c_double()
    raise_for(lib().kvx_nt_max_step_dev(m, dx.ptr, ctypes.byref(r)))
    return r.value


def kkt_chol2(G, dims, A, mnl=0):
    """KKT solver factory, sparse-G branch of misc.py:1352-1567 (same state machine: first call
    fixes the pattern of S = G' W^{-1} W^{-T} G (+H); later calls refactor numerically on the same
    symbolic analysis; K = A S^{-1} A' is refactored with a fresh analysis every call; the singular-S
    fallback adds A'A).  Returns factor(W, H=None, Df=None) -> solve(x, y, z).

    mnl > 0 (cvxprog, misc.py:1396-1400, 1413-1415, 1423-1424, 1452-1453, 1523, 1560-1561): Df is the mnl x n sparse
    Jacobian of the nonlinear constraints, stacked on top of G:  S = Df' Wnl^-2 Df + G' Wl^-2 G + H; its pattern is
    fixed by the first call like G's.  A dense Df (or dense G, H) sends the reference to its LAPACK branch, which is out
    of scope here; a dense A (the reference's "mixed" branch, misc.py:1476-1481) is accepted."""
    if dims.get("q") or dims.get("s"):
        raise ValueError("kktsolver option 'kkt_chol2' is implemented only for problems with no "
                         "second-order or semidefinite cone constraints")
    if isinstance(G, matrix):
        raise NotImplementedError("kkt_chol2: the dense-G LAPACK branch (misc.py:1429,1467-1472) is out of scope")
    p, n = A.size
    if isinstance(A, matrix):
        # mixed branch of the reference (sparse S, dense A: misc.py:1476-1481 forms a dense K with LAPACK).  Here a dense A
        # is A with every entry stored: Asct = L^-1 P A' and K = Asct' Asct come out of the same device kernels as for a
        # sparse A, K is simply a full p x p pattern (one front).
        Ad = np.asarray(A.a, dtype=np.float64).reshape(p, n)
        A = spmatrix.from_ccs(p, n, np.arange(n + 1, dtype=np.int64) * p, np.tile(np.arange(p, dtype=np.int64), n),
                              np.ascontiguousarray(Ad.T).reshape(-1))
    ml = dims["l"]
    F = {"firstcall": True, "singular": False}

    def factor(W, H=None, Df=None):
        if F["firstcall"]:
            gm, gn, gcp, gri, gv = base._as_ccs(G)
            F["Gs"] = spmatrix.from_ccs(gm, gn, gcp.copy(), gri.copy(), np.zeros(gv.size))
            if mnl:
                if isinstance(Df, matrix) or not hasattr(Df, "CCS"):
                    raise NotImplementedError("kkt_chol2: a dense Df takes the reference's LAPACK branch (out of scope)")
                dm, dn, dcp, dri, dv = base._as_ccs(Df)
                F["Dfs"] = spmatrix.from_ccs(dm, dn, dcp.copy(), dri.copy(), np.zeros(dv.size))
            F["S"] = spmatrix([], [], [], (n, n))
            F["K"] = spmatrix([], [], [], (p, p))
        if isinstance(H, matrix):
            raise NotImplementedError("kkt_chol2: a dense H takes the reference's LAPACK branch (out of scope)")
        # Dfs = Wnl^{-1} * Df
        if mnl:
            base.gemm(base.spdiag(W["dnli"]), Df, F["Dfs"], partial=True)
        # Gs = Wl^{-1} * G
        base.gemm(base.spdiag(W["di"]), G, F["Gs"], partial=True)
        if F["firstcall"]:
            base.syrk(F["Gs"], F["S"], trans="T")
            if mnl:
                base.syrk(F["Dfs"], F["S"], trans="T", beta=1.0)
            if H is not None:
                F["S"] += H
            try:
                F["Sf"] = cholmod.symbolic(F["S"])
                cholmod.numeric(F["S"], F["Sf"])
            except ArithmeticError:
                F["singular"] = True
                base.syrk(F["Gs"], F["S"], trans="T")
                if mnl:
                    base.syrk(F["Dfs"], F["S"], trans="T", beta=1.0)
                base.syrk(A, F["S"], trans="T", beta=1.0)
                if H is not None:
                    F["S"] += H
                F["Sf"] = cholmod.symbolic(F["S"])
                cholmod.numeric(F["S"], F["Sf"])
            F["firstcall"] = False
        else:
            base.syrk(F["Gs"], F["S"], trans="T", partial=True)
            if mnl:
                base.syrk(F["Dfs"], F["S"], trans="T", beta=1.0, partial=True)
            if H is not None:
                F["S"] += H
            if F["singular"]:
                base.syrk(A, F["S"], trans="T", beta=1.0, partial=True)
            cholmod.numeric(F["S"], F["Sf"])
        # Asct := L^{-1}*P*A'.  Factor K = Asct'*Asct.
        Asct = cholmod.spsolve(F["Sf"], A.T, sys=7)
        Asct = cholmod.spsolve(F["Sf"], Asct, sys=4)
        F["K"] = spmatrix([], [], [], (p, p))
        base.syrk(Asct, F["K"], trans="T")
        Kf = cholmod.symbolic(F["K"])
        cholmod.numeric(F["K"], Kf)

        def solve(x, y, z):
            # z := W^{-1} * z = W^{-1} * bz
            scale(z, W, trans="T", inverse="I")
            # x := L^{-1} * P * (x + Gs'*z (+ A'*y if singular))
            if mnl:
                base.gemv(F["Dfs"], z, x, trans="T", beta=1.0)
            base.gemv(F["Gs"], z, x, offsetx=mnl, trans="T", beta=1.0)
            if F["singular"]:
                base.gemv(A, y, x, trans="T", beta=1.0)
            cholmod.solve(F["Sf"], x, sys=7)
            cholmod.solve(F["Sf"], x, sys=4)
            # y := K^{-1} * (Asct'*x - y)
            base.gemv(Asct, x, y, trans="T", beta=-1.0)
            cholmod.solve(Kf, y)
            # x := P' * L^{-T} * (x - Asct*y)
            base.gemv(Asct, y, x, alpha=-1.0, beta=1.0)
            cholmod.solve(F["Sf"], x, sys=5)
            cholmod.solve(F["Sf"], x, sys=8)
            # W*z := Gs*x - z
            if mnl:
                base.gemv(F["Dfs"], x, z, beta=-1.0)
            base.gemv(F["Gs"], x, z, beta=-1.0, offsety=mnl)

        return solve

    return factor
