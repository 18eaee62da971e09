"""Orthant ('l' cone) subset of `kvxopt.misc` / `kvxopt.misc_solvers` on MI355X, plus the KKT solver
factory `kkt_chol2` -- same names, argument meaning and in-place semantics as the reference
(src/python/misc.py, src/C/misc_solvers.c).  Second-order-cone and semidefinite blocks are out of
scope (kkt_chol2 rejects them in the reference too, misc.py:1381-1384).

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


def _run1(fn, x, *dev_args_builder):
    raise NotImplementedError


def _up(a):
    return DeviceBuffer.from_array(np.ascontiguousarray(a, dtype=np.float64))


def _sync():
    raise_for(lib().kvx_dev_sync())


def compute_scaling(s, z, lmbda, dims, mnl=None):
    """misc.py:250-287 ('l' block): W['d'] = sqrt(s./z), W['di'] = 1./d, lmbda = sqrt(s.*z)."""
    _only_l(dims, "compute_scaling")
    if mnl:
        raise NotImplementedError("nonlinear blocks are out of scope")
    m = dims["l"]
    sb, _ = _buf(s)
    zb, _ = _buf(z)
    lb, _ = _buf(lmbda)
    _lib.require_device()
    ds, dz = _up(sb[:m]), _up(zb[:m])
    dd, ddi, dl = DeviceBuffer(8 * max(m, 1)), DeviceBuffer(8 * max(m, 1)), DeviceBuffer(8 * max(m, 1))
    raise_for(lib().kvx_nt_compute_scaling_dev(m, ds.ptr, dz.ptr, dd.ptr, ddi.ptr, dl.ptr))
    _sync()
    W = {"d": matrix(dd.download(np.float64, m), (m, 1)), "di": matrix(ddi.download(np.float64, m), (m, 1)),
         "v": [], "beta": [], "r": [], "rti": []}
    lb[:m] = dl.download(np.float64, m)
    return W


def update_scaling(W, lmbda, s, z):
    """misc.py:422-464 ('l' block), in place: s:=sqrt(s), z:=sqrt(z), d:=d.*s./z, di:=1./d, lmbda:=s.*z."""
    if W.get("v") or W.get("r"):
        raise NotImplementedError("update_scaling: only the orthant ('l') cone")
    db, _ = _buf(W["d"])
    dib, _ = _buf(W["di"])
    m = db.size
    sb, _ = _buf(s)
    zb, _ = _buf(z)
    lb, _ = _buf(lmbda)
    _lib.require_device()
    ds, dz, dd = _up(sb[:m]), _up(zb[:m]), _up(db)
    ddi, dl = DeviceBuffer(8 * max(m, 1)), DeviceBuffer(8 * max(m, 1))
    raise_for(lib().kvx_nt_update_scaling_dev(m, ds.ptr, dz.ptr, dd.ptr, ddi.ptr, dl.ptr))
    _sync()
    sb[:m] = ds.download(np.float64, m)
    zb[:m] = dz.download(np.float64, m)
    db[:] = dd.download(np.float64, m)
    dib[:] = ddi.download(np.float64, m)
    lb[:m] = dl.download(np.float64, m)


def scale(x, W, trans="N", inverse="N"):
    """misc_solvers.c:85-141 / misc.py:36-82 ('l' block): x := d.*x ('N') or di.*x ('I') for every
    column of x; trans is irrelevant for a diagonal scaling."""
    if W.get("v") or W.get("r"):
        raise NotImplementedError("scale: only the orthant ('l') cone")
    w, _ = _buf(W["d"] if inverse == "N" else W["di"])
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
    fallback adds A'A).  Returns factor(W, H=None) -> solve(x, y, z)."""
    if dims.get("q") or dims.get("s"):
        raise ValueError("kktsolver option 'kkt_chol2' is implemented only for problems with no "
                         "second-order or semidefinite cone constraints")
    if mnl:
        raise NotImplementedError("nonlinear blocks (cvxprog) are out of scope")
    p, n = A.size
    ml = dims["l"]
    F = {"firstcall": True, "singular": False}

    def factor(W, H=None, Df=None):
        if F["firstcall"]:
            gm, gn, gcp, gri, gv = base._as_ccs(G)
            F["Gs"] = spmatrix.from_ccs(gm, gn, gcp.copy(), gri.copy(), np.zeros(gv.size))
            F["S"] = spmatrix([], [], [], (n, n))
            F["K"] = spmatrix([], [], [], (p, p))
        # Gs = Wl^{-1} * G
        base.gemm(base.spdiag(W["di"]), G, F["Gs"], partial=True)
        if F["firstcall"]:
            base.syrk(F["Gs"], F["S"], trans="T")
            if H is not None:
                F["S"] += H
            try:
                F["Sf"] = cholmod.symbolic(F["S"])
                cholmod.numeric(F["S"], F["Sf"])
            except ArithmeticError:
                F["singular"] = True
                base.syrk(F["Gs"], F["S"], trans="T")
                base.syrk(A, F["S"], trans="T", beta=1.0)
                if H is not None:
                    F["S"] += H
                F["Sf"] = cholmod.symbolic(F["S"])
                cholmod.numeric(F["S"], F["Sf"])
            F["firstcall"] = False
        else:
            base.syrk(F["Gs"], F["S"], trans="T", partial=True)
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
            base.gemv(F["Gs"], z, x, trans="T", beta=1.0)
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
            base.gemv(F["Gs"], x, z, beta=-1.0)

        return solve

    return factor
