"""Device-resident cone-LP interior-point driver for the orthant cone (inequality form, p = 0):

    minimize c'x  subject to  G x + s = h,  s >= 0          (G sparse, ml x n)

A restatement of the reference's `coneprog.conelp` (src/python/coneprog.py:31-1436) specialised to
dims = {'l': ml, 'q': [], 's': []} and no equality constraints, with the default KKT solver
`misc.kkt_chol2` (src/python/misc.py:1352-1567).  Every vector lives in HBM for the whole solve; per
iteration the host sees only scalars (gap, residual norms, step lengths).  Per iteration, as in the
reference: 1 numeric refactorisation of S = G' diag(di^2) G on a fixed symbolic analysis,
3 KKT solves, 2 products with G and 2 with G', the NT-scaling update -- all HIP kernels of
libkvxhip.so.  No CPU fallback.

Equality constraints (p > 0, K = A S^{-1} A') are device-resident too: KKTDiagEqDev (diagonal S, sparse K on a fixed
pattern) and KKTGenEqDev (general S, dense K in HBM, any p).  `kvxopt_amd.misc.kkt_chol2` is built on the same classes.
"""
import collections
import ctypes
import math
import os
import sys
import time

import numpy as np

from . import _lib, base
from ._lib import DeviceBuffer, lib, raise_for
from .chol import Factor

EXPON = 3          # coneprog.py:423
STEP = 0.99        # coneprog.py:424
# KVX_LP_UNFUSED=1: one launch per BLAS-1-sized operation, as in rounds 1-2 (the library reads the same variable); the fused launches
# of round 3 do the same arithmetic with the same roundings -- tests/test_kkt_gpu.py compares the two bit for bit
_UNFUSED = os.environ.get("KVX_LP_UNFUSED", "0") not in ("", "0")
_TRACE = os.environ.get("KVX_LP_TRACE", "0") not in ("", "0")      # per-iteration wall times of conelp on stderr
# KVX_LP_PYCALLS=1: the fused launches of an iteration issued one by one from Python (round 3) instead of through the four calls
# kvx_lp_iter_* (round 4: same kernels, same order, issued from C)
_PYCALLS = os.environ.get("KVX_LP_PYCALLS", "0") not in ("", "0")


def _sides(items):
    """ctypes array of kvx_kkt_side from (xin, xscale, zin, xout, xoscale, zout, zoscale) tuples of DVecs and floats."""
    arr = (_lib.KktSide * 2)()
    for k, (xin, xs, zin, xout, xos, zout, zos) in enumerate(items):
        a = arr[k]
        a.xin, a.xscale, a.zin = xin.ptr, float(xs), zin.ptr
        a.xout, a.xoscale, a.zout, a.zoscale = xout.ptr, float(xos), zout.ptr, float(zos)
    return arr


class DVec:
    """A float64 vector in HBM with the BLAS-1 / NT-scaling operations of the C ABI."""

    def __init__(self, n, init=None):
        self.n = int(n)
        self.buf = DeviceBuffer(8 * max(self.n, 1))
        if init is not None:
            self.set(init)

    @property
    def ptr(self):
        return self.buf.ptr

    def set(self, a):
        self.buf.upload(np.ascontiguousarray(a, dtype=np.float64).reshape(-1))
        return self

    def get(self):
        return self.buf.download(np.float64, self.n)

    def fill(self, v):
        raise_for(lib().kvx_vec_fill_dev(self.n, float(v), self.ptr)); return self

    def copy_from(self, x):
        raise_for(lib().kvx_vec_copy_dev(self.n, x.ptr, self.ptr)); return self

    def axpy(self, x, alpha=1.0):                       # self += alpha * x
        raise_for(lib().kvx_vec_axpy_dev(self.n, float(alpha), x.ptr, self.ptr)); return self

    def lincomb(self, a, x, b=0.0, y=None):             # self := a * x + b * y (one pass; copy + axpy / copy + scal)
        raise_for(lib().kvx_vec_lincomb_dev(self.n, float(a), x.ptr, float(b) if y is not None else 0.0,
                                            (y if y is not None else x).ptr, self.ptr)); return self

    def scal(self, alpha):
        raise_for(lib().kvx_vec_scal_dev(self.n, float(alpha), self.ptr)); return self

    def addc(self, c):
        raise_for(lib().kvx_vec_addc_dev(self.n, float(c), self.ptr)); return self

    def mul(self, y):                                   # self .*= y   (misc.scale / sprod / scale2 'I')
        raise_for(lib().kvx_nt_sprod_dev(self.n, self.ptr, y.ptr)); return self

    def div(self, y):                                   # self ./= y   (sinv / scale2 'N')
        raise_for(lib().kvx_nt_sinv_dev(self.n, self.ptr, y.ptr)); return self

    def sqr_of(self, y):                                # self := y.*y  (misc.ssqr)
        raise_for(lib().kvx_nt_ssqr_dev(self.n, self.ptr, y.ptr)); return self

    def xmy(self, a, x, y, b=0.0):                      # self := a * x.*y + b * self
        raise_for(lib().kvx_vec_xmy_dev(self.n, float(a), x.ptr, y.ptr, float(b), self.ptr)); return self

    def dot(self, y):
        if self.n == 0:
            return 0.0
        r = ctypes.c_double()
        raise_for(lib().kvx_nt_sdot_dev(self.n, self.ptr, y.ptr, ctypes.byref(r)))
        return r.value

    def nrm2(self):
        return math.sqrt(self.dot(self))

    def max_step(self):                                 # misc.max_step 'l' block: max_i(-x_i)
        r = ctypes.c_double()
        raise_for(lib().kvx_nt_max_step_dev(self.n, self.ptr, ctypes.byref(r)))
        return r.value


def reduce_multi(items):
    """Several reductions with ONE host synchronisation (kvx_nt_reduce_multi_dev).  items: ("dot", x, y) or ("max", x)
    with DVec operands; entries whose operand is not a DVec (the empty y-blocks of p = 0) yield 0.0.  Bitwise the values
    of DVec.dot / DVec.max_step."""
    live = [(k, it) for k, it in enumerate(items) if isinstance(it[1], DVec) and it[1].n > 0]
    out = [0.0] * len(items)
    if not live:
        return out
    m = len(live)
    kind = (ctypes.c_int32 * m)(*[0 if it[0] == "dot" else 1 for _, it in live])
    n = (ctypes.c_int64 * m)(*[it[1].n for _, it in live])
    xs = (ctypes.c_void_p * m)(*[it[1].ptr for _, it in live])
    ys = (ctypes.c_void_p * m)(*[(it[2].ptr if it[0] == "dot" else None) for _, it in live])
    res = (ctypes.c_double * m)()
    raise_for(lib().kvx_nt_reduce_multi_dev(m, kind, n, xs, ys, res))
    for j, (k, _) in enumerate(live):
        out[k] = float(res[j])
    return out


class SpMatDev:
    """CCS matrix resident in HBM (int64 indices as in the reference, kvxopt.h:46), together with the CCS of its
    transpose: both directions of the mat-vec then run as row gathers (kvx_spmv_dev 'T') -- the column-scatter form of
    'N' needs FP64 atomics, whose rounding depends on the order of arrival: with it two runs of the interior-point loop
    differ in the last bits, without it they are bitwise identical (like the factorisation and the solves)."""

    def __init__(self, m, n, colptr, rowind, values):
        self.m, self.n = int(m), int(n)
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowind = np.ascontiguousarray(rowind, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=np.float64)
        self.cp = DeviceBuffer.from_array(colptr)
        self.ri = DeviceBuffer.from_array(rowind) if len(rowind) else DeviceBuffer(8)
        self.vx = DeviceBuffer.from_array(values) if len(values) else DeviceBuffer(8)
        cols = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(colptr))
        # (row, column) order: the entries come column by column, so a STABLE sort by row is the two-key sort; counts by bincount
        # (lexsort + np.add.at were 6 of the 8 ms this constructor took for 400 000 entries)
        order = np.argsort(rowind, kind="stable")
        self._order = order                               # CCS position of every entry of the transposed copy
        tp = np.zeros(self.m + 1, dtype=np.int64)
        if len(rowind):
            np.cumsum(np.bincount(rowind, minlength=self.m), out=tp[1:])
        self.max_row = int(np.diff(tp).max()) if self.m else 0         # most entries in a row / column (the fused kernels size
        self.max_col = int(np.diff(colptr).max()) if self.n else 0     # their lane groups by them)
        self.tcp = DeviceBuffer.from_array(tp)
        self.tri = DeviceBuffer.from_array(cols[order]) if len(rowind) else DeviceBuffer(8)
        self.tvx = DeviceBuffer.from_array(values[order]) if len(values) else DeviceBuffer(8)

    def set_values(self, values):
        """New values on the same pattern (both copies)."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        if values.size:
            self.vx.upload(values)
            self.tvx.upload(np.ascontiguousarray(values[self._order]))

    def gemv(self, x, y, trans="N", alpha=1.0, beta=0.0):
        """y := alpha*op(A)*x + beta*y  (base.gemv -> sparse.c:1073-1104)."""
        if trans == "N":                                  # A x = (A')' x: gather over the rows of A
            raise_for(lib().kvx_spmv_dev(ord("T"), self.n, self.m, self.tcp.ptr, self.tri.ptr, self.tvx.ptr,
                                         float(alpha), x.ptr, float(beta), y.ptr))
        else:
            raise_for(lib().kvx_spmv_dev(ord("T"), self.m, self.n, self.cp.ptr, self.ri.ptr, self.vx.ptr,
                                         float(alpha), x.ptr, float(beta), y.ptr))


class KKTChol2Dev:
    """Device-resident `misc.kkt_chol2` for sparse G and p = 0 (misc.py:1389-1563)."""

    def __init__(self, ml, n, Gp, Gi, Gx, chol_opts=None, Pp=None, Pi=None, Px=None):
        """Pp, Pi, Px: optional lower-triangular CCS of the QP Hessian H (coneqp): S = H + G' W^-1 W^-T G
        (misc.py:1425-1426, 1454-1455)."""
        self.ml, self.n = ml, n
        Gp = np.ascontiguousarray(Gp, dtype=np.int64)
        Gi = np.ascontiguousarray(Gi, dtype=np.int64)
        h = ctypes.c_void_p()
        self.Px = None
        if Pp is not None:
            Pp = np.ascontiguousarray(Pp, dtype=np.int64)
            Pi = np.ascontiguousarray(Pi, dtype=np.int64)
            self.Px = DVec(max(len(Px), 1), Px if len(Px) else None)
            raise_for(lib().kvx_atda_plan(ml, n, _lib.pi(Gp), _lib.pi(Gi), _lib.pi(Pp), _lib.pi(Pi), ctypes.byref(h)))
        else:
            raise_for(lib().kvx_atda_plan(ml, n, _lib.pi(Gp), _lib.pi(Gi), None, None, ctypes.byref(h)))
        self._plan = h
        snz = ctypes.c_int64()
        raise_for(lib().kvx_atda_pattern(h, ctypes.byref(snz), None, None))
        self.Sp = np.empty(n + 1, dtype=np.int64)
        Si = np.empty(max(snz.value, 1), dtype=np.int64)
        raise_for(lib().kvx_atda_pattern(h, ctypes.byref(snz), _lib.pi(self.Sp), _lib.pi(Si)))
        self.Si = Si[:snz.value].copy()
        # first call of the reference fixes the pattern of S and analyses it once (misc.py:1422-1432)
        self.fac = Factor(n, self.Sp, self.Si, "L", None, chol_opts)
        self.G = SpMatDev(ml, n, Gp, Gi, Gx)
        self.w = DVec(ml)
        self.t = DVec(ml)
        self.Sx = DVec(self.Si.size)
        self.di = None
        self.nfactor = 0
        self.async_solves = False      # True: solves are only enqueued; check() after the next host synchronisation

    def __del__(self):
        if getattr(self, "_plan", None):
            lib().kvx_atda_free(self._plan)
            self._plan = None

    def reset(self, Gx, Px=None):
        """New values of G (and H) on the patterns given at construction: the object -- product map, analysis of S, captured
        launch graphs -- serves another problem of the same structure (see `_kkt_for`)."""
        self.G.set_values(Gx)
        if Px is not None:
            self.set_hessian(Px)
        self.di = None
        self.nfactor = 0
        self.async_solves = False

    def set_hessian(self, Px):
        """New values of H on the pattern given at construction (cvxprog: H changes at every iteration)."""
        if self.Px is not None and len(Px):
            self.Px.set(Px)

    def factor(self, di, sync=True):
        """S = G' diag(di)^2 G on the fixed pattern, numeric refactorisation (misc.py:1418-1462).
        Raises ArithmeticError when S is not positive definite -- with sync=False only the NEXT solve does (the solve
        is queued behind the factorisation without a host round trip)."""
        self._assemble(di)
        self.fac.factorize_dev(self.Sx.ptr, sync=sync)
        self.di = di
        self.nfactor += 1

    def _assemble(self, di):
        if _UNFUSED:
            self.w.sqr_of(di)
            raise_for(lib().kvx_atda_assemble_dev(self._plan, self.G.vx.ptr, self.w.ptr,
                                                  None if self.Px is None else self.Px.ptr, self.Sx.ptr))
        else:                                            # the square is taken while G is scaled: one launch less
            raise_for(lib().kvx_atda_assemble_sq_dev(self._plan, self.G.vx.ptr, di.ptr,
                                                     None if self.Px is None else self.Px.ptr, self.Sx.ptr))

    def _x2buf(self):
        if getattr(self, "_x2", None) is None:
            self._x2 = DVec(2 * max(self.n, 1))
        return self._x2

    def _pre(self, sides, nrhs):
        G = self.G
        raise_for(lib().kvx_kkt_solve_pre_dev(self.ml, self.n, G.cp.ptr, G.ri.ptr, G.vx.ptr, G.max_col, self.di.ptr, nrhs, sides,
                                              self._x2buf().ptr, max(1, self.n)))

    def _post(self, sides, nrhs):
        G = self.G
        raise_for(lib().kvx_kkt_solve_post_dev(self.ml, self.n, G.tcp.ptr, G.tri.ptr, G.tvx.ptr, G.max_row, self.di.ptr, nrhs, sides,
                                               self._x2buf().ptr, max(1, self.n)))

    def solve_sides(self, items):
        """One or two KKT solves with the current factor, each side given as (xin, xscale, zin, xout, xoscale, zout, zoscale):
        x2 := xscale*xin + Gs' W^-1 zin ; x2 := S^-1 x2 ; xout := xoscale*x2 ; zout := zoscale*(Gs x2 - W^-1 zin)
        (misc.py:1489-1563 with p = 0) in three enqueues: kvx_kkt_solve_pre_dev, the triangular solves, kvx_kkt_solve_post_dev."""
        nrhs = len(items)
        sides = _sides(items)
        self._pre(sides, nrhs)
        self.fac.solve_dev(self._x2buf().ptr, 0, nrhs, max(1, self.n), sync=not self.async_solves)
        self._post(sides, nrhs)

    def factor_solve_sides(self, di, items):
        """factor(di) and solve_sides(items) with the factorisation and the triangular solves as ONE enqueue
        (kvx_chol_factorize_solve_async_dev).  Enqueue only -- check() after the next host synchronisation."""
        nrhs = len(items)
        sides = _sides(items)
        self._assemble(di)
        self.di = di
        self.nfactor += 1
        self._pre(sides, nrhs)
        self.fac.factorize_solve_async_dev(self.Sx.ptr, self._x2buf().ptr, nrhs, max(1, self.n))
        self._post(sides, nrhs)

    def factor_solve2(self, di, xa, za, xb, zb):
        """factor(di) and solve2(xa, za, xb, zb) as ONE enqueue (kvx_chol_factorize_solve_async_dev): the two right-hand sides
        do not depend on the factor, so they are formed first and the forward sweep runs beside the factorisation of the top of
        the tree.  Same kernels, same order: bitwise what factor() followed by solve2() gives.  Enqueue only -- check() after
        the next host synchronisation."""
        n = self.n
        if not _UNFUSED:
            return self.factor_solve_sides(di, [(xa, 1.0, za, xa, 1.0, za, 1.0), (xb, 1.0, zb, xb, 1.0, zb, 1.0)])
        self._x2buf()
        self._assemble(di)
        self.di = di
        self.nfactor += 1
        for k, (x, z) in enumerate(((xa, za), (xb, zb))):
            z.mul(di)
            self.t.xmy(1.0, di, z)
            self.G.gemv(self.t, x, trans="T", alpha=1.0, beta=1.0)
            raise_for(lib().kvx_vec_copy_dev(n, x.ptr, self._x2.ptr + 8 * n * k))
        self.fac.factorize_solve_async_dev(self.Sx.ptr, self._x2.ptr, 2, max(1, n))
        for k, (x, z) in enumerate(((xa, za), (xb, zb))):
            raise_for(lib().kvx_vec_copy_dev(n, self._x2.ptr + 8 * n * k, x.ptr))
            self.G.gemv(x, self.t, trans="N")
            z.xmy(1.0, di, self.t, -1.0)

    def check(self):
        """Raise ArithmeticError if the last (asynchronous) factorisation failed; synchronises the factor's stream."""
        self.fac.status()

    def solve(self, x, z):
        """Overwrites (x, z) with (ux, W*uz) (misc.py:1489-1563 with p = 0)."""
        di = self.di
        if not _UNFUSED:
            return self.solve_sides([(x, 1.0, z, x, 1.0, z, 1.0)])
        z.mul(di)                                        # z := W^{-1} z                (misc.py:1513)
        self.t.xmy(1.0, di, z)                           # t := di .* z
        self.G.gemv(self.t, x, trans="T", alpha=1.0, beta=1.0)   # x += Gs' z       (misc.py:1524)
        self.fac.solve_dev(x.ptr, 0, 1, max(1, self.n), sync=not self.async_solves)  # x := S^{-1} x (sys 7,4 then 5,8: misc.py:1531-1558)
        self.G.gemv(x, self.t, trans="N")                # t := G x
        z.xmy(1.0, di, self.t, -1.0)                     # z := Gs x - z                (misc.py:1563)

    def solve2(self, xa, za, xb, zb):
        """Two KKT systems with the same factor in ONE two-column triangular solve (the interior-point iteration has
        two right-hand sides that do not depend on each other: the (-c, h) system and the predictor)."""
        di, n = self.di, self.n
        if not _UNFUSED:
            return self.solve_sides([(xa, 1.0, za, xa, 1.0, za, 1.0), (xb, 1.0, zb, xb, 1.0, zb, 1.0)])
        self._x2buf()
        for k, (x, z) in enumerate(((xa, za), (xb, zb))):
            z.mul(di)
            self.t.xmy(1.0, di, z)
            self.G.gemv(self.t, x, trans="T", alpha=1.0, beta=1.0)
            raise_for(lib().kvx_vec_copy_dev(n, x.ptr, self._x2.ptr + 8 * n * k))
        self.fac.solve_dev(self._x2.ptr, 0, 2, max(1, n), sync=not self.async_solves)
        for k, (x, z) in enumerate(((xa, za), (xb, zb))):
            raise_for(lib().kvx_vec_copy_dev(n, self._x2.ptr + 8 * n * k, x.ptr))
            self.G.gemv(x, self.t, trans="N")
            z.xmy(1.0, di, self.t, -1.0)


class KKTUserHost:
    """The reference's plug-in point `kktsolver(W) -> f(x, y, z)` (coneprog.py:323-344, 571-585) under the device-resident
    conelp: the caller's factory is handed the scaling on the host (W['d'], W['di'] as base.matrix; the other fields of the
    'l'-cone scaling empty, as misc.compute_scaling leaves them) at every factorisation, and its f the right-hand sides as
    base.matrix columns at every solve; (ux, uy, W uz) go back to HBM.  Functional, not fast -- every solve crosses PCIe --
    and the place where `kvxopt_amd.misc.kkt_chol2` (host-array mirror of the reference's default) or any user solver plugs in."""

    def __init__(self, ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, factory):
        self.ml, self.n, self.p = ml, n, p
        self.G = SpMatDev(ml, n, Gp, Gi, Gx)
        self.A = SpMatDev(p, n, Ap, Ai, Ax) if p else None
        self.factory = factory
        self.f = None
        self.di = None
        self.nfactor = 0
        self.async_solves = False

    def factor(self, di, sync=True):
        dih = di.get()
        empty = base.matrix(0.0, (0, 1))
        W = {"d": base.matrix(1.0 / dih), "di": base.matrix(dih), "dnl": empty, "dnli": base.matrix(0.0, (0, 1)),
             "v": [], "beta": [], "r": [], "rti": []}
        self.f = self.factory(W)                          # ArithmeticError / ValueError of the caller's factorisation pass through
        self.di = di
        self.nfactor += 1

    def solve(self, x, y, z):
        xm = base.matrix(x.get())
        ym = base.matrix(y.get()) if self.p else base.matrix(0.0, (0, 1))
        zm = base.matrix(z.get())
        self.f(xm, ym, zm)
        x.set(xm._a)
        if self.p:
            y.set(ym._a)
        z.set(zm._a)

    def check(self):
        pass


class KKTDiagEqDev:
    """Device-resident `misc.kkt_chol2` with equality constraints (p > 0) for a G whose columns have disjoint row
    supports (every row of G holds at most one entry -- G = -I of a standard-form LP).  Then
    S = G' W^-1 W^-T G is DIAGONAL and the reference's K = A S^-1 A' (misc.py:1483-1487, 1545: sparse triangular
    solves with the factor of S, then a syrk) is one more fixed-pattern assembly, K = sum_k (1/S_kk) A[:,k] A[:,k]',
    factored numerically on a symbolic analysis done once (the reference re-analyses K at every call, misc.py:1486)."""

    def __init__(self, ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, chol_opts=None):
        self.ml, self.n, self.p = ml, n, p
        Gi = np.ascontiguousarray(Gi, dtype=np.int64)
        if Gi.size and np.bincount(Gi, minlength=ml).max() > 1:
            raise NotImplementedError("device-resident conelp with equality constraints needs a G with at most one "
                                      "entry per row (diagonal S); use kvxopt_amd.misc.kkt_chol2 (host arrays) otherwise")
        self.G = SpMatDev(ml, n, Gp, Gi, Gx)
        self.G2 = SpMatDev(ml, n, Gp, Gi, np.asarray(Gx, dtype=np.float64) ** 2)
        self.A = SpMatDev(p, n, Ap, Ai, Ax)
        # CCS of A' (n x p) = CSR of A
        Ap = np.asarray(Ap, dtype=np.int64); Ai = np.asarray(Ai, dtype=np.int64); Ax = np.asarray(Ax, dtype=np.float64)
        cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
        order = np.lexsort((cols, Ai))
        self._at_order = order
        ATi, ATx = cols[order], Ax[order]
        ATp = np.zeros(p + 1, dtype=np.int64)
        np.add.at(ATp, Ai + 1, 1)
        np.cumsum(ATp, out=ATp)
        h = ctypes.c_void_p()
        raise_for(lib().kvx_atda_plan(n, p, _lib.pi(ATp), _lib.pi(np.ascontiguousarray(ATi)), None, None, ctypes.byref(h)))
        self._plan = h
        self.A2 = SpMatDev(p, n, Ap, Ai, Ax ** 2)                     # diag(K) = (A o A) S^-1
        self.kd = DVec(max(p, 1))
        self.kscale = 1.0
        knz = ctypes.c_int64()
        raise_for(lib().kvx_atda_pattern(h, ctypes.byref(knz), None, None))
        self.Kp = np.empty(p + 1, dtype=np.int64)
        Ki = np.empty(max(knz.value, 1), dtype=np.int64)
        raise_for(lib().kvx_atda_pattern(h, ctypes.byref(knz), _lib.pi(self.Kp), _lib.pi(Ki)))
        self.Ki = Ki[:knz.value].copy()
        # K is factored after scaling to max(diag K) = 1 with the pivot rule "d <= 1e-30 -> 1e128" (cholmod's dbound in its
        # drop-the-row form): near the solution S^-1 = diag(x ./ z) spans twenty orders of magnitude and K = A S^-1 A' loses
        # rank in floating point; the reference stops there with status 'unknown' (coneprog.py:1078-1109), interior-point
        # codes built on normal equations zero the affected multipliers instead.
        ko = {"dbound": 1e-15, "dbound_drop": 1}
        ko.update(chol_opts or {})
        self.fac = Factor(p, self.Kp, self.Ki, "L", None, ko)
        self.ATx = DVec(max(ATx.size, 1), ATx if ATx.size else None)
        self.Kx = DVec(self.Ki.size)
        self.w, self.t = DVec(ml), DVec(ml)
        self.sdiag, self.sinv, self.u = DVec(n), DVec(n), DVec(n)
        self.di = None
        self.nfactor = 0
        self.async_solves = False

    def __del__(self):
        if getattr(self, "_plan", None):
            lib().kvx_atda_free(self._plan)
            self._plan = None

    def reset(self, Gx, Ax):
        """New values of G and A on the patterns given at construction (see `_kkt_for`)."""
        Gx = np.asarray(Gx, dtype=np.float64); Ax = np.asarray(Ax, dtype=np.float64)
        self.G.set_values(Gx); self.G2.set_values(Gx ** 2)
        self.A.set_values(Ax); self.A2.set_values(Ax ** 2)
        if Ax.size:
            self.ATx.set(Ax[self._at_order])
        self.kscale = 1.0
        self.di = None
        self.nfactor = 0
        self.async_solves = False

    def factor(self, di, sync=True):
        self.w.sqr_of(di)
        self.G2.gemv(self.w, self.sdiag, trans="T")                   # S_kk = sum_i di_i^2 G_ik^2
        self.sinv.fill(1.0).div(self.sdiag)
        self.A2.gemv(self.sinv, self.kd, trans="N", alpha=-1.0)       # -diag(K)
        smin, kmax = reduce_multi([("max", self.sdiag), ("max", self.kd)])   # one host round trip: -min S_kk, max diag K
        if -smin <= 0.0 or not (kmax > 0.0) or not np.isfinite(kmax):  # S singular / K = 0
            raise ArithmeticError(0)
        self.kscale = 1.0 / kmax
        self.sinv.scal(self.kscale)                                   # K' = A (S^-1 / max diag K) A'
        raise_for(lib().kvx_atda_assemble_dev(self._plan, self.ATx.ptr, self.sinv.ptr, None, self.Kx.ptr))
        self.sinv.scal(kmax)
        self.fac.factorize_dev(self.Kx.ptr, sync=sync)
        self.di = di
        self.nfactor += 1

    def check(self):
        self.fac.status()

    def solve(self, x, y, z):
        """Overwrites (x, y, z) = (bx, by, bz) with (ux, uy, uz) of misc.py:1489-1563."""
        di = self.di
        z.mul(di)
        self.t.xmy(1.0, di, z)
        self.G.gemv(self.t, x, trans="T", alpha=1.0, beta=1.0)        # x := bx + G' W^-1 W^-T bz
        self.u.xmy(1.0, self.sinv, x)                                 # u := S^-1 x
        self.A.gemv(self.u, y, trans="N", alpha=1.0, beta=-1.0)       # y := A S^-1 x - by
        self.fac.solve_dev(y.ptr, 0, 1, max(1, self.p), sync=not self.async_solves)   # y := K'^-1 y
        y.scal(self.kscale)                                           # K^-1 = K'^-1 / max diag K: uy
        self.A.gemv(y, x, trans="T", alpha=-1.0, beta=1.0)
        x.mul(self.sinv)                                              # x := S^-1 (x - A' uy) = ux
        self.G.gemv(x, self.t, trans="N")
        z.xmy(1.0, di, self.t, -1.0)                                  # z := W^-T (G ux - bz) = uz

    def solve2(self, xa, ya, za, xb, yb, zb):
        """Two KKT systems, one two-column solve with the factor of K (see KKTChol2Dev.solve2)."""
        di, p = self.di, self.p
        if getattr(self, "_y2", None) is None:
            self._y2 = DVec(2 * max(p, 1))
        for k, (x, y, z) in enumerate(((xa, ya, za), (xb, yb, zb))):
            z.mul(di)
            self.t.xmy(1.0, di, z)
            self.G.gemv(self.t, x, trans="T", alpha=1.0, beta=1.0)
            self.u.xmy(1.0, self.sinv, x)
            self.A.gemv(self.u, y, trans="N", alpha=1.0, beta=-1.0)
            raise_for(lib().kvx_vec_copy_dev(p, y.ptr, self._y2.ptr + 8 * p * k))
        self.fac.solve_dev(self._y2.ptr, 0, 2, max(1, p), sync=not self.async_solves)
        self._y2.scal(self.kscale)
        for k, (x, y, z) in enumerate(((xa, ya, za), (xb, yb, zb))):
            raise_for(lib().kvx_vec_copy_dev(p, self._y2.ptr + 8 * p * k, y.ptr))
            self.A.gemv(y, x, trans="T", alpha=-1.0, beta=1.0)
            x.mul(self.sinv)
            self.G.gemv(x, self.t, trans="N")
            z.xmy(1.0, di, self.t, -1.0)


class KKTGenEqDev:
    """Device-resident `misc.kkt_chol2` with equality constraints and a GENERAL sparse G (S not diagonal), for a moderate
    number p of equality rows.  The reference forms Asct = L^-1 P A' by sparse triangular solves, K = Asct' Asct by a sparse
    syrk and re-analyses K at every call (misc.py:1476-1487).  Here X = S^-1 A' is one multi-right-hand-side solve with the
    factor of S (n x p dense), K = A X a dense p x p matrix (one gather mat-mat), factored as a single dense front by the
    same Cholesky (dense lower pattern, analysed once).  KKT solve (misc.py:1489-1563 written with S^-1):
    u = S^-1 (bx + G' W^-1 W^-T bz), uy = K^-1 (A u - by), ux = S^-1 (bx + G'.. - A' uy), uz = W^-T (G ux - bz)."""

    BLOCK_BYTES = 1 << 30        # X = S^-1 A' is formed in column blocks of about this size (n x cols doubles)

    def __init__(self, ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, chol_opts=None, Pp=None, Pi=None, Px=None):
        """Pp, Pi, Px: optional lower-triangular CCS of the QP Hessian (coneqp): S = P + G' W^-1 W^-T G.
        Any p: K = A S^-1 A' is structurally dense whenever S is irreducible (every column of L^-1 P A' reaches the root of
        the elimination tree), so it is held as a dense p x p matrix in HBM (20 GB at p = 50 000) and X = S^-1 A' is formed
        and consumed in column blocks -- never as a whole."""
        self.ml, self.n, self.p = ml, n, p
        self.cols = max(1, min(max(p, 1), self.BLOCK_BYTES // (8 * max(n, 1))))
        self.S = KKTChol2Dev(ml, n, Gp, Gi, Gx, chol_opts, Pp, Pi, Px)
        self.G = self.S.G
        self.A = SpMatDev(p, n, Ap, Ai, Ax)
        # CCS of A' (n x p) = CSR of A
        Ap = np.asarray(Ap, dtype=np.int64); Ai = np.asarray(Ai, dtype=np.int64); Ax = np.asarray(Ax, dtype=np.float64)
        cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
        order = np.lexsort((cols, Ai))
        self._at_order = order
        ATp = np.zeros(p + 1, dtype=np.int64)
        np.add.at(ATp, Ai + 1, 1)
        np.cumsum(ATp, out=ATp)
        self.AT = SpMatDev(n, p, ATp, cols[order], Ax[order])
        self.X = DVec(max(n * self.cols, 1))
        self.x_whole = p <= self.cols and os.environ.get("KVX_KKT_NO_X") != "1"     # X = S^-1 A' of the last factorisation is held as a whole
        self.Kd = DVec(max(p * p, 1))
        self.Kx = DVec(max(p * (p + 1) // 2, 1))
        Kp = np.zeros(p + 1, dtype=np.int64)
        Kp[1:] = np.cumsum(np.arange(p, 0, -1))
        Ki = np.concatenate([np.arange(j, p, dtype=np.int64) for j in range(p)]) if p else np.zeros(0, np.int64)
        # K is dense: natural order, one front; scaled to max diag = 1 with the pivot floor of KKTDiagEqDev
        self.fac = Factor(p, Kp, Ki, "L", None, {"ordering": 1, "dbound": 1e-15, "dbound_drop": 1})
        self.kdiag = DVec(max(p, 1))
        self.u, self.r = DVec(n), DVec(n)
        self.t = self.S.t
        self.kscale = 1.0
        self.di = None
        self.nfactor = 0
        self._async = False

    def reset(self, Gx, Ax, Px=None):
        """New values of G, A (and H) on the patterns given at construction (see `_kkt_for`)."""
        Ax = np.asarray(Ax, dtype=np.float64)
        self.S.reset(Gx, Px)
        self.A.set_values(Ax)
        self.AT.set_values(Ax[self._at_order])
        self.kscale = 1.0
        self.di = None
        self.nfactor = 0
        self.async_solves = False

    @property
    def async_solves(self):
        return self._async

    @async_solves.setter
    def async_solves(self, v):
        self._async = v
        self.S.async_solves = v

    def factor(self, di, sync=True):
        n, p = self.n, self.p
        self.S.factor(di, sync=False)                                  # S = G' W^-1 W^-T G: assembly + numeric refactorisation
        for c0 in range(0, p, self.cols):                              # column blocks of X = S^-1 A' and of K = A X
            nb = min(self.cols, p - c0)
            raise_for(lib().kvx_dense_from_ccs_dev(n, nb, self.AT.cp.ptr + 8 * c0, self.AT.ri.ptr, self.AT.vx.ptr, self.X.ptr, n))
            self.S.fac.solve_dev(self.X.ptr, 0, nb, n, sync=False)     # (a failed S surfaces in check())
            raise_for(lib().kvx_spmm_t_dev(p, nb, self.AT.cp.ptr, self.AT.ri.ptr, self.AT.vx.ptr, self.X.ptr, n,
                                           self.Kd.ptr + 8 * c0 * p, p))
        raise_for(lib().kvx_pack_lower_dev(p, self.Kd.ptr, p, self.Kx.ptr))
        # scale to max diag K = 1 (one host round trip; it also surfaces a failed factorisation of S)
        raise_for(lib().kvx_vec_copy_strided_dev(p, self.Kd.ptr, p + 1, self.kdiag.ptr))
        self.kdiag.scal(-1.0)
        kmax = self.kdiag.max_step()
        self.S.check()
        if not (kmax > 0.0) or not np.isfinite(kmax):
            raise ArithmeticError(0)
        self.kscale = 1.0 / kmax
        self.Kx.scal(self.kscale)
        self.fac.factorize_dev(self.Kx.ptr, sync=sync)
        self.di = di
        self.nfactor += 1

    def check(self):
        self.S.check()
        self.fac.status()

    def solve(self, x, y, z):
        """Overwrites (x, y, z) = (bx, by, bz) with (ux, uy, uz)."""
        self._solve_cols(((x, y, z),))

    def solve2(self, xa, ya, za, xb, yb, zb):
        """Two KKT systems: their solves with S and with K are two-column solves (see KKTChol2Dev.solve2)."""
        self._solve_cols(((xa, ya, za), (xb, yb, zb)))

    def _solve_cols(self, systems):
        # u = S^-1 (bx + G' W^-1 W^-T bz), uy = K^-1 (A u - by), ux = u - X uy with X = S^-1 A' of the last factorisation when the
        # whole of it is at hand (p columns fit one block): S^-1 (b - A' uy) = S^-1 b - X uy, a dense product (80 MB at n = 50 000,
        # p = 200: 20 us) instead of a second sweep through the factor of S (0.5 ms).  Otherwise the second solve as in misc.py:1545-1553.
        di, n, p = self.di, self.n, self.p
        nc = len(systems)
        if getattr(self, "_u2", None) is None:
            self._u2 = DVec(2 * max(n, 1))
            self._y2 = DVec(2 * max(p, 1))
        U, Y = self._u2, self._y2
        for c, (x, y, z) in enumerate(systems):
            z.mul(di)
            self.t.xmy(1.0, di, z)
            self.G.gemv(self.t, x, trans="T", alpha=1.0, beta=1.0)    # x := bx + G' W^-1 W^-T bz
            raise_for(lib().kvx_vec_copy_dev(n, x.ptr, U.ptr + 8 * n * c))
        self.S.fac.solve_dev(U.ptr, 0, nc, max(1, n), sync=not self._async)           # u := S^-1 x
        for c, (x, y, z) in enumerate(systems):
            raise_for(lib().kvx_vec_copy_dev(n, U.ptr + 8 * n * c, self.u.ptr))
            self.A.gemv(self.u, y, trans="N", alpha=1.0, beta=-1.0)   # y := A u - by
            raise_for(lib().kvx_vec_copy_dev(p, y.ptr, Y.ptr + 8 * p * c))
        self.fac.solve_dev(Y.ptr, 0, nc, max(1, p), sync=not self._async)
        for c, (x, y, z) in enumerate(systems):
            raise_for(lib().kvx_vec_copy_dev(p, Y.ptr + 8 * p * c, y.ptr))
            y.scal(self.kscale)                                       # uy = K^-1 (A u - by)
        if self.x_whole:
            for c, (x, y, z) in enumerate(systems):
                raise_for(lib().kvx_vec_copy_dev(n, U.ptr + 8 * n * c, x.ptr))
                raise_for(lib().kvx_dense_gemv_dev(n, p, 1, -1.0, self.X.ptr, n, y.ptr, p, 1.0, x.ptr, n))   # ux = u - X uy
        else:
            for c, (x, y, z) in enumerate(systems):
                self.A.gemv(y, x, trans="T", alpha=-1.0, beta=1.0)    # x := x - A' uy
                raise_for(lib().kvx_vec_copy_dev(n, x.ptr, U.ptr + 8 * n * c))
            self.S.fac.solve_dev(U.ptr, 0, nc, max(1, n), sync=not self._async)       # ux = S^-1 (...)
            for c, (x, y, z) in enumerate(systems):
                raise_for(lib().kvx_vec_copy_dev(n, U.ptr + 8 * n * c, x.ptr))
        for c, (x, y, z) in enumerate(systems):
            self.G.gemv(x, self.t, trans="N")
            z.xmy(1.0, di, self.t, -1.0)                              # uz = W^-T (G ux - bz)


# The KKT objects of the last few constraint structures are kept: a sequence of cone programs on the same patterns of G, A (and
# P) -- the usual way an interior-point solver is deployed: receding-horizon control, parameter sweeps, branch and bound -- pays
# for the product map, the analysis of S (and K), the device set-up and the capture of the launch graphs once; only the values
# are refreshed.  cholmod.linsolve / klu.linsolve keep their analyses the same way.  `clear_cache()` releases them.
_KKT_CACHE = collections.OrderedDict()
_KKT_CACHE_MAX = 4


def clear_cache():
    """Release the cached KKT objects (device memory, analyses, launch graphs)."""
    _KKT_CACHE.clear()


_lib.register_cache(clear_cache)


def _opts_key(chol_opts):
    """Hashable key of an options dict whatever the value types (numpy integers, bools, floats): sorted (name, repr) pairs."""
    return tuple(sorted((str(k), repr(v.item() if hasattr(v, "item") else v)) for k, v in (chol_opts or {}).items()))


def _pattern_key(*arrays):
    return _lib.pattern_digest(*[np.ascontiguousarray(a, dtype=np.int64) for a in arrays])


def _kkt_for(kind, dims_key, patterns, chol_opts, build, refresh):
    """The cached KKT object of this structure with its values refreshed, or a new one (build())."""
    if os.environ.get("KVX_LP_NO_CACHE") == "1":
        return build()
    # the device is part of the key: the buffers, streams and launch graphs of a KKT object live on the device it was built on
    key = (kind, dims_key, _pattern_key(*patterns), _opts_key(chol_opts), _lib.current_device())
    kkt = _KKT_CACHE.pop(key, None)
    if kkt is None:
        kkt = _lib.retry_after_release(build)
    else:
        refresh(kkt)
    _KKT_CACHE[key] = kkt
    while len(_KKT_CACHE) > _KKT_CACHE_MAX:
        _KKT_CACHE.popitem(last=False)
    return kkt


def conelp(c, G, h, dims=None, A=None, b=None, options=None, chol_opts=None, primalstart=None, dualstart=None, kktsolver=None):
    """Solve the LP  minimize c'x  s.t.  Gx <= h, Ax = b  on the GPU.  c: (n,), h: (ml,), G: spmatrix-like
    (ml x n, sparse); A (p x n, sparse), b (p,) optional -- with equality constraints either G has at most one entry per
    row (standard form: KKTDiagEqDev, sparse K on a fixed pattern) or any other sparse G (KKTGenEqDev, dense K in HBM).  primalstart = {'x', 's'},
    dualstart = {'y', 'z'} (y optional) as in the reference (coneprog.py:683-737): s and z must be strictly positive.  Returns the reference's result dictionary (coneprog.py:962-974) with numpy arrays."""
    _lib.require_device()
    opts = {"maxiters": 100, "abstol": 1e-7, "reltol": 1e-6, "feastol": 1e-7, "show_progress": False, "refinement": 0}
    opts.update(options or {})
    MAXITERS, ABSTOL, RELTOL, FEASTOL = opts["maxiters"], opts["abstol"], opts["reltol"], opts["feastol"]
    show = opts["show_progress"]
    REFINEMENT = opts["refinement"]                      # coneprog.py:502-507: default 0 when there are no 'q' / 's' cones
    if not isinstance(REFINEMENT, (int, np.integer)) or REFINEMENT < 0:
        raise ValueError("options['refinement'] must be a nonnegative integer")
    ml, n, Gp, Gi, Gx = base._as_ccs(G)
    if dims is not None and (dims.get("q") or dims.get("s") or dims.get("l", ml) != ml):
        raise NotImplementedError("only the orthant cone dims = {'l': G.size[0], 'q': [], 's': []} runs on the GPU")
    c_h = np.asarray(c._a if isinstance(c, base.matrix) else c, dtype=np.float64).reshape(-1)
    h_h = np.asarray(h._a if isinstance(h, base.matrix) else h, dtype=np.float64).reshape(-1)
    if c_h.size != n or h_h.size != ml:
        raise TypeError("dimensions of c, G, h do not match")
    p = 0
    if A is not None:
        p, na, Ap, Ai, Ax = base._as_ccs(A)
        if na != n:
            raise TypeError("'A' must have %d columns" % n)
        b_h = np.asarray(b._a if isinstance(b, base.matrix) else b, dtype=np.float64).reshape(-1)
        if b_h.size != p:
            raise TypeError("'b' must have length %d" % p)
    if p > n or p + ml < n:
        raise ValueError("Rank(A) < p or Rank([G; A]) < n")          # coneprog.py:572-573

    class _NoY:                                                      # p = 0: the y-blocks of the algorithm are empty
        def __getattr__(self, name):
            return lambda *a, **k: self
        def dot(self, other):
            return 0.0
        def nrm2(self):
            return 0.0
        def get(self):
            return np.zeros(0)

    if kktsolver is not None:
        # the reference's plug-in point (coneprog.py:323-344): kktsolver(W) returns f(x, y, z); host round trips per call
        if not callable(kktsolver):
            raise ValueError("kktsolver must be a function W -> f(x, y, z) (the reference's named solvers 'ldl', 'ldl2', 'qr', "
                             "'chol', 'chol2' are not part of this path: 'chol2' is what runs on the GPU by default)")
        if p == 0:
            Ap, Ai, Ax = np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0)
        kkt = KKTUserHost(ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, kktsolver)
        fused = False
        kfactor_solve2 = None
        if p > 0:
            Ad = kkt.A
            bv = DVec(p, b_h)
            y, dy, y1, ry, hry = (DVec(p) for _ in range(5))
            def Af(u, v, trans="N", alpha=1.0, beta=0.0):
                Ad.gemv(u, v, trans=trans, alpha=alpha, beta=beta)
        else:
            bv = y = dy = y1 = ry = hry = _NoY()
            def Af(u, v, trans="N", alpha=1.0, beta=0.0):
                if trans == "T" and beta == 0.0:
                    v.fill(0.0)
        ksolve = kkt.solve
        def ksolve2(xa, ya, za, xb, yb, zb):
            kkt.solve(xa, ya, za)
            kkt.solve(xb, yb, zb)
    elif p > 0:
        Gi64 = np.asarray(Gi, dtype=np.int64)
        diag_s = not (Gi64.size and np.bincount(Gi64, minlength=ml).max() > 1)
        cls = KKTDiagEqDev if diag_s else KKTGenEqDev
        kkt = _kkt_for(cls.__name__, (ml, n, p), (Gp, Gi, Ap, Ai), chol_opts,
                       lambda: cls(ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, chol_opts), lambda k: k.reset(Gx, Ax))
        Ad = kkt.A
        bv = DVec(p, b_h)
        y, dy, y1, ry, hry = (DVec(p) for _ in range(5))
        ksolve = kkt.solve
        ksolve2 = kkt.solve2
        kfactor_solve2 = None
        fused = False
        def Af(u, v, trans="N", alpha=1.0, beta=0.0):
            Ad.gemv(u, v, trans=trans, alpha=alpha, beta=beta)
    else:
        kkt = _kkt_for("KKTChol2Dev", (ml, n, 0), (Gp, Gi), chol_opts,
                       lambda: KKTChol2Dev(ml, n, Gp, Gi, Gx, chol_opts), lambda k: k.reset(Gx))
        bv = y = dy = y1 = ry = hry = _NoY()
        def ksolve(xx, yy, zz):
            kkt.solve(xx, zz)
        def ksolve2(xa, ya, za, xb, yb, zb):
            kkt.solve2(xa, za, xb, zb)
        def kfactor_solve2(dd, xa, ya, za, xb, yb, zb):
            kkt.factor_solve2(dd, xa, za, xb, zb)
        fused = not _UNFUSED and REFINEMENT == 0         # p = 0: the short launches of an iteration fused (kkt.hip, "round 3")
        def Af(u, v, trans="N", alpha=1.0, beta=0.0):
            if trans == "T" and beta == 0.0:
                v.fill(0.0)                                          # A' y with p = 0: the zero vector
    Gd = kkt.G
    cv, hv = DVec(n, c_h), DVec(ml, h_h)
    x, dx, x1, rx, hrx = (DVec(n) for _ in range(5))
    s, z, ds, dz, z1, rz, hrz, th, ws3, tmp, lmbda, lmbdasq, d, di = (DVec(ml) for _ in range(14))

    resx0 = max(1.0, cv.nrm2())
    resy0 = max(1.0, bv.nrm2())
    resz0 = max(1.0, hv.nrm2())

    t_loop = [None]
    t_phase = [0.0, 0.0, 0.0]
    t_mark = [0.0]
    fast = fused and not _PYCALLS and isinstance(kkt, KKTChol2Dev) and kkt.Px is None
    if fast:
        ctx = _lib.LpCtx(ml, n, Gd.cp.ptr, Gd.ri.ptr, Gd.vx.ptr, Gd.max_col, Gd.tcp.ptr, Gd.tri.ptr, Gd.tvx.ptr, Gd.max_row,
                         kkt._plan, kkt.fac._h, kkt.Sx.ptr, kkt._x2buf().ptr, x.ptr, s.ptr, z.ptr, cv.ptr, hv.ptr, hrx.ptr, rx.ptr,
                         hrz.ptr, rz.ptr, lmbda.ptr, d.ptr, di.ptr, ds.ptr, dz.ptr, dx.ptr, x1.ptr, z1.ptr, th.ptr, ws3.ptr)
        out10 = (ctypes.c_double * 10)()
    next_stats = None                                    # the residual statistics of the coming iteration, when the update has made them

    if REFINEMENT:
        # iterative refinement of the Newton systems (coneprog.py:599-631 res(), :1110-1195 f6_no_ir, :1211-1235 f6), operation by
        # operation on the device vectors; scalars (tau, kappa blocks) travel in one-element lists.  For the orthant cone
        # W = diag(d): scale(., W, inverse='I') multiplies by di, scale(., W, trans='T') by d.
        wx, wx2 = DVec(n), DVec(n)
        wy, wy2 = (DVec(p), DVec(p)) if p else (_NoY(), _NoY())
        wz, ws, wz2, ws2, rs3, rz3 = (DVec(ml) for _ in range(6))

        def f6_no_ir_g(sc, bx, by, bz, btau, bs, bkappa):
            by.scal(-1.0)
            bs.div(lmbda).scal(-1.0)                     # s := -lmbda o\ bs
            rs3.copy_from(bs).mul(d)
            bz.axpy(rs3).scal(-1.0)                      # z := -(bz + W' s)
            ksolve(bx, by, bz)
            bkappa[0] = -bkappa[0] / sc["lmbda_g"]
            btau[0] += bkappa[0] / sc["dgi"]
            btau[0] = sc["dgi"] * (btau[0] + cv.dot(bx) + bv.dot(by) + th.dot(bz)) / (1.0 + z1.dot(z1))
            bx.axpy(x1, btau[0]); by.axpy(y1, btau[0]); bz.axpy(z1, btau[0])
            bs.axpy(bz, -1.0)
            bkappa[0] -= btau[0]

        def res_g(sc, ux, uy, uz, utau, us, ukappa, vx, vy, vz, vtau, vs, vkappa):
            dg_ = 1.0 / sc["dgi"]
            Af(uy, vx, trans="T", alpha=-1.0, beta=1.0)
            rz3.copy_from(uz).mul(di)                    # W^{-1} uz
            Gd.gemv(rz3, vx, trans="T", alpha=-1.0, beta=1.0)
            vx.axpy(cv, -utau[0] / dg_)
            Af(ux, vy, trans="N", alpha=1.0, beta=1.0)
            vy.axpy(bv, -utau[0] / dg_)
            Gd.gemv(ux, vz, trans="N", alpha=1.0, beta=1.0)
            vz.axpy(hv, -utau[0] / dg_)
            rs3.copy_from(us).mul(d)
            vz.axpy(rs3)
            vtau[0] += dg_ * ukappa[0] + cv.dot(ux) + bv.dot(uy) + hv.dot(rz3)
            rs3.copy_from(us).axpy(uz).mul(lmbda)
            vs.axpy(rs3)
            vkappa[0] += sc["lmbda_g"] * (utau[0] + ukappa[0])

        def f6_g(sc, bx, by, bz, btau, bs, bkappa):
            wx.copy_from(bx); wy.copy_from(by); wz.copy_from(bz); ws.copy_from(bs)
            wtau, wkappa = [btau[0]], [bkappa[0]]
            f6_no_ir_g(sc, bx, by, bz, btau, bs, bkappa)
            for _ in range(REFINEMENT):
                wx2.copy_from(wx); wy2.copy_from(wy); wz2.copy_from(wz); ws2.copy_from(ws)
                wtau2, wkappa2 = [wtau[0]], [wkappa[0]]
                res_g(sc, bx, by, bz, btau, bs, bkappa, wx2, wy2, wz2, wtau2, ws2, wkappa2)
                f6_no_ir_g(sc, wx2, wy2, wz2, wtau2, ws2, wkappa2)
                bx.axpy(wx2); by.axpy(wy2); bz.axpy(wz2)
                btau[0] += wtau2[0]
                bs.axpy(ws2)
                bkappa[0] += wkappa2[0]

    def result(status, iters, gap, relgap, pcost, dcost, pres, dres, pinfres, dinfres, xs=True, zs=True, msg=None):
        if show:                                         # the reference's closing line (coneprog.py:791,941,961,985,1010,1094)
            print(msg or {"optimal": "Optimal solution found.", "primal infeasible": "Certificate of primal infeasibility found.",
                          "dual infeasible": "Certificate of dual infeasibility found."}[status])
        return {"x": x.get() if xs else None, "y": y.get() if zs else None,
                "s": s.get() if xs else None, "z": z.get() if zs else None,
                "status": status, "gap": gap, "relative gap": relgap,
                "primal objective": pcost, "dual objective": dcost,
                "primal infeasibility": pres, "dual infeasibility": dres,
                "primal slack": -s.max_step() if xs else None, "dual slack": -z.max_step() if zs else None,
                "residual as primal infeasibility certificate": pinfres,
                "residual as dual infeasibility certificate": dinfres,
                "iterations": iters, "factorizations": kkt.nfactor,
                # wall time of the interior-point loop proper (coneprog.py:859-1436), i.e. without the symbolic
                # analysis and the starting point; not a key of the reference's dictionary
                "loop seconds": (time.perf_counter() - t_loop[0]) if t_loop[0] is not None else 0.0,
                # the same, split at the three host synchronisations of an iteration: [residual norms -> first direction
                # (assembly, factorisation, two solves), -> second direction (one solve), -> update + residuals of the next]
                "phase seconds": list(t_phase)}

    # ---- starting point (coneprog.py:662-822): factor with W = I ------------------------------------
    d.fill(1.0); di.fill(1.0)
    try:
        kkt.factor(di)
    except ArithmeticError:
        raise ValueError("Rank(A) < p or Rank([G; A]) < n")
    def _vec(v):
        return np.ascontiguousarray(np.asarray(v._a if isinstance(v, base.matrix) else v, dtype=np.float64).reshape(-1))

    if primalstart is None:
        x.fill(0.0); dy.copy_from(bv); s.copy_from(hv)
        ksolve(x, dy, s)
        s.scal(-1.0)
    else:                                                        # coneprog.py:703-705
        xs0, ss0 = _vec(primalstart["x"]), _vec(primalstart["s"])
        if xs0.size != n or ss0.size != ml:
            raise TypeError("primalstart has the wrong dimensions")
        x.set(xs0); s.set(ss0)
    ts = s.max_step()
    if ts >= 0 and primalstart is not None:
        raise ValueError("initial s is not positive")
    if dualstart is None:
        dx.copy_from(cv).scal(-1.0); y.fill(0.0); z.fill(0.0)
        ksolve(dx, y, z)
    else:                                                        # coneprog.py:731-733
        zs0 = _vec(dualstart["z"])
        if zs0.size != ml:
            raise TypeError("dualstart has the wrong dimensions")
        if p and "y" in dualstart:
            y.set(_vec(dualstart["y"]))
        elif p:
            y.fill(0.0)
        z.set(zs0)
    tz = z.max_step()
    if tz >= 0 and dualstart is not None:
        raise ValueError("initial z is not positive")
    nrms, nrmz = s.nrm2(), z.nrm2()
    if primalstart is None and dualstart is None:
        gap = s.dot(z)
        pcost = cv.dot(x)
        dcost = -bv.dot(y) - hv.dot(z)
        relgap = gap / -pcost if pcost < 0.0 else (gap / dcost if dcost > 0.0 else None)
        if ts <= 0 and tz <= 0 and (gap <= ABSTOL or (relgap is not None and relgap <= RELTOL)):
            rx.copy_from(cv); Af(y, rx, trans="T", alpha=1.0, beta=1.0); Gd.gemv(z, rx, trans="T", alpha=1.0, beta=1.0)
            resx = rx.nrm2()
            ry.copy_from(bv); Af(x, ry, trans="N", alpha=1.0, beta=-1.0)
            resy = ry.nrm2()
            Gd.gemv(x, rz, trans="N"); rz.axpy(s); rz.axpy(hv, -1.0)
            resz = rz.nrm2()
            return result("optimal", 0, gap, relgap, pcost, dcost, max(resy / resy0, resz / resz0), resx / resx0, None, None)
    # (coneprog.py:806-842: a computed start is pushed into the cone, a given one is taken as it is)
    if primalstart is None and ts >= -1e-8 * max(nrms, 1.0):
        s.addc(1.0 + ts)
    if dualstart is None and tz >= -1e-8 * max(nrmz, 1.0):
        z.addc(1.0 + tz)

    tau, kappa = 1.0, 1.0
    gap = s.dot(z)
    lmbda.fill(0.0)
    dg = dgi = lmbda_g = 1.0
    t_loop[0] = time.perf_counter()
    for iters in range(MAXITERS + 1):
        # residuals (coneprog.py:861-896); their norms and the objectives come back in one reduction call
        if fast:
            if next_stats is None:
                raise_for(lib().kvx_lp_iter_residuals(ctypes.byref(ctx), tau, out10))
                next_stats = tuple(out10)
        elif fused:                                      # the six launches below in one
            raise_for(lib().kvx_lp_residuals_dev(ml, n, Gd.cp.ptr, Gd.ri.ptr, Gd.vx.ptr, Gd.max_col, Gd.tcp.ptr, Gd.tri.ptr, Gd.tvx.ptr, Gd.max_row,
                                                 x.ptr, z.ptr, s.ptr, cv.ptr, hv.ptr, tau, hrx.ptr, rx.ptr, hrz.ptr, rz.ptr))
        else:
            Af(y, hrx, trans="T", alpha=-1.0, beta=0.0)
            Gd.gemv(z, hrx, trans="T", alpha=-1.0, beta=1.0)
            rx.lincomb(1.0, hrx, -tau, cv)
            Af(x, hry, trans="N")
            ry.lincomb(1.0, hry, -tau, bv)
            Gd.gemv(x, hrz, trans="N"); hrz.axpy(s)
            rz.lincomb(1.0, hrz, -tau, hv)
        if fast:
            (v_hrx, v_rx, v_hry, v_ry, v_hrz, v_rz, cx, by, hz, lam2) = next_stats
            next_stats = None
        else:
            (v_hrx, v_rx, v_hry, v_ry, v_hrz, v_rz, cx, by, hz, lam2) = reduce_multi(
                [("dot", hrx, hrx), ("dot", rx, rx), ("dot", hry, hry), ("dot", ry, ry), ("dot", hrz, hrz), ("dot", rz, rz),
                 ("dot", cv, x), ("dot", bv, y), ("dot", hv, z), ("dot", lmbda, lmbda)])
        t_now = time.perf_counter()
        if iters > 0:
            t_phase[2] += t_now - t_mark[0]
        t_mark[0] = t_now
        hresx, resx = math.sqrt(v_hrx), math.sqrt(v_rx) / tau
        hresy, resy = math.sqrt(v_hry), math.sqrt(v_ry) / tau
        hresz, resz = math.sqrt(v_hrz), math.sqrt(v_rz) / tau
        if iters > 0:
            gap = (math.sqrt(lam2) / tau) ** 2           # (coneprog.py:1436; lmbda of the previous update)
        rt = kappa + cx + by + hz
        pcost, dcost = cx / tau, -(by + hz) / tau
        relgap = gap / -pcost if pcost < 0.0 else (gap / dcost if dcost > 0.0 else None)
        pres = max(resy / resy0, resz / resz0)
        dres = resx / resx0
        pinfres = hresx / resx0 / (-hz - by) if hz + by < 0.0 else None
        dinfres = max(hresy / resy0, hresz / resz0) / (-cx) if cx < 0.0 else None
        if show:
            if iters == 0:
                print("% 10s% 12s% 10s% 8s% 7s % 5s" % ("pcost", "dcost", "gap", "pres", "dres", "k/t"))
            print("%2d: % 8.4e % 8.4e % 4.0e% 7.0e% 7.0e% 7.0e" % (iters, pcost, dcost, gap, pres, dres, kappa / tau))

        if (pres <= FEASTOL and dres <= FEASTOL and (gap <= ABSTOL or (relgap is not None and relgap <= RELTOL))) \
                or iters == MAXITERS:
            x.scal(1.0 / tau); y.scal(1.0 / tau); s.scal(1.0 / tau); z.scal(1.0 / tau)
            if iters == MAXITERS:
                return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, pinfres, dinfres,
                              msg="Terminated (maximum number of iterations reached).")
            return result("optimal", iters, gap, relgap, pcost, dcost, pres, dres, None, None)
        elif pinfres is not None and pinfres <= FEASTOL:
            y.scal(1.0 / (-hz - by)); z.scal(1.0 / (-hz - by))
            return result("primal infeasible", iters, None, None, None, 1.0, None, None, pinfres, None, xs=False)
        elif dinfres is not None and dinfres <= FEASTOL:
            x.scal(1.0 / (-cx)); s.scal(1.0 / (-cx))
            return result("dual infeasible", iters, None, None, -1.0, None, None, None, None, dinfres, zs=False)

        # NT scaling at the first iteration (coneprog.py:1031-1043 -> misc.py:284-287)
        if iters == 0:
            raise_for(lib().kvx_nt_compute_scaling_dev(ml, s.ptr, z.ptr, d.ptr, di.ptr, lmbda.ptr))
            dg = math.sqrt(kappa / tau)
            dgi = math.sqrt(tau / kappa)
            lmbda_g = math.sqrt(tau * kappa)
            lam2 = lmbda.dot(lmbda)
        if not fused:
            lmbdasq.sqr_of(lmbda)                        # (fused: formed inside kvx_lp_newton_rhs_dev)
        lmbdasq_g = lmbda_g ** 2

        mu = (lam2 + lmbda_g ** 2) / (1 + ml)
        sigma = 0.0
        wkappa3 = 0.0
        st8 = {}                                          # dkappa, dtau of the Newton step under construction

        def newton_rhs(i):
            # right-hand side of the Newton system (coneprog.py:1250-1298) and the first half of f6_no_ir (:1130-1160)
            dkappa = lmbdasq_g
            if i == 1:
                dkappa += wkappa3 - sigma * mu
            # ds := -(lmbdasq (+ ws3 - sigma mu)) o\ lmbda,  dz := -((1 - sigma) rz + W' ds): one fused kernel
            raise_for(lib().kvx_lp_newton_rhs_dev(ml, None if fused else lmbdasq.ptr, ws3.ptr if i == 1 else None,
                                                  sigma * mu if i == 1 else 0.0, 1.0 - sigma, rz.ptr, lmbda.ptr, d.ptr, ds.ptr, dz.ptr))
            if not fused:                                # fused: dx := (1 - sigma) rx is the first operation of the KKT solve
                dx.lincomb(1.0 - sigma, rx)
                dy.lincomb(-(1.0 - sigma), ry)
            st8["dtau"] = (1.0 - sigma) * rt
            st8["dkappa"] = dkappa

        # factor + the two solves that do not depend on each other (coneprog.py:1066-1077 and the predictor's f3):
        # one two-column triangular solve with the new factor
        try:
            kkt.async_solves = True                      # enqueue only: the host runs ahead of the GPU up to the next scalar
            if fast:
                pass                                     # (the predictor's launches go out with its second half, below: kvx_lp_iter_predictor)
            elif fused:
                # as below, with the vector operations around the solves inside their two launches: x1 := -c, z1 := h, the
                # scaling of (x1, z1) by dgi, dx := (1 - sigma) rx
                newton_rhs(0)
                kkt.factor_solve_sides(di, [(cv, -1.0, hv, x1, dgi, z1, dgi), (rx, 1.0 - sigma, dz, dx, 1.0, dz, 1.0)])
            elif REFINEMENT:
                kkt.factor(di, sync=False)               # (the directions are solved one by one inside f6_g)
                x1.lincomb(-1.0, cv)
                y1.copy_from(bv)
                z1.copy_from(hv)
                ksolve(x1, y1, z1)
            elif kfactor_solve2 is not None:
                # the factorisation and the two solves in one enqueue: their right-hand sides do not depend on the factor
                x1.lincomb(-1.0, cv)
                y1.copy_from(bv)
                z1.copy_from(hv)
                newton_rhs(0)
                kfactor_solve2(di, x1, y1, z1, dx, dy, dz)
            else:
                kkt.factor(di, sync=False)               # a failed factorisation surfaces in the solve right below
                x1.lincomb(-1.0, cv)
                y1.copy_from(bv)
                z1.copy_from(hv)
                newton_rhs(0)
                ksolve2(x1, y1, z1, dx, dy, dz)
            if fast:
                pass
            elif fused:
                th.xmy(1.0, hv, di)
            else:
                x1.scal(dgi); y1.scal(dgi); z1.scal(dgi)
                th.copy_from(hv).mul(di)                 # th = W^{-T} h      (coneprog.py:1126-1128)
        except ArithmeticError:
            kkt.async_solves = False
            x.scal(1.0 / tau); y.scal(1.0 / tau); s.scal(1.0 / tau); z.scal(1.0 / tau)
            return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, pinfres, dinfres, msg="Terminated (singular KKT matrix).")

        z1z1 = -1.0                                      # computed on the device with the first direction
        out4 = (ctypes.c_double * 4)()
        for i in (0, 1):
            if REFINEMENT:
                # the direction with iterative refinement (coneprog.py:1250-1333 around f6): right-hand side as the reference
                # sets it up, then f6, the Mehrotra products, the scaling by lmbda and the step bounds, one operation at a time
                sc = {"dgi": dgi, "lmbda_g": lmbda_g}
                if i == 0:
                    try:
                        kkt.check()                      # (a failed factorisation must not be refined)
                    except ArithmeticError:
                        kkt.async_solves = False
                        x.scal(1.0 / tau); y.scal(1.0 / tau); s.scal(1.0 / tau); z.scal(1.0 / tau)
                        return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, pinfres, dinfres,
                                      msg="Terminated (singular KKT matrix).")
                ds.copy_from(lmbdasq)
                dkap = [lmbdasq_g]
                if i == 1:
                    ds.axpy(ws3).addc(-sigma * mu)
                    dkap[0] += wkappa3 - sigma * mu
                dx.lincomb(1.0 - sigma, rx)
                dy.lincomb(1.0 - sigma, ry)
                dz.lincomb(1.0 - sigma, rz)
                dta = [(1.0 - sigma) * rt]
                f6_g(sc, dx, dy, dz, dta, ds, dkap)
                dtau, dkappa = dta[0], dkap[0]
                if i == 0:
                    ws3.copy_from(ds).mul(dz)
                    wkappa3 = dtau * dkappa
                ds.div(lmbda); dz.div(lmbda)             # misc.scale2(lmbda, .) on the orthant
                ts, tz = ds.max_step(), dz.max_step()
                tt = -dtau / lmbda_g
                tk = -dkappa / lmbda_g
                t = max(0.0, ts, tz, tt, tk)
                step = 1.0 if t == 0.0 else (min(1.0, 1.0 / t) if i == 0 else min(1.0, STEP / t))
                if i == 0:
                    sigma = (1.0 - step) ** EXPON
                continue
            if fast:
                # the launches of the direction up to its scalars in ONE call (kvx_lp_iter_predictor / _corrector: what newton_rhs,
                # factor_solve_sides / solve_sides, th.xmy and kvx_lp_second_half_dev enqueue, in that order, issued from C)
                dkappa = -(lmbdasq_g if i == 0 else lmbdasq_g + (wkappa3 - sigma * mu)) / lmbda_g      # (newton_rhs's association)
                dtau0 = (1.0 - sigma) * rt + dkappa / dgi
                try:
                    if i == 0:
                        kkt.di = di
                        kkt.nfactor += 1
                        raise_for(lib().kvx_lp_iter_predictor(ctypes.byref(ctx), dgi, dtau0, out4))
                    else:
                        raise_for(lib().kvx_lp_iter_corrector(ctypes.byref(ctx), sigma * mu, 1.0 - sigma, dgi, dtau0, z1z1, out4))
                except ArithmeticError:
                    kkt.async_solves = False
                    x.scal(1.0 / tau); y.scal(1.0 / tau); s.scal(1.0 / tau); z.scal(1.0 / tau)
                    return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, pinfres, dinfres, msg="Terminated (singular KKT matrix).")
            elif i == 1:
                newton_rhs(1)
                if fused:
                    kkt.solve_sides([(rx, 1.0 - sigma, dz, dx, 1.0, dz, 1.0)])
                else:
                    ksolve(dx, dy, dz)
            # second half of f6_no_ir (coneprog.py:1162-1195), dz += dtau z1, ds -= dz, [ws3 := ds o dz for the corrector
            # (coneprog.py:1303-1306)], the scaling by lmbda and the step bounds (coneprog.py:1314-1321): dtau is formed on the
            # device from the inner products, ONE host round trip per direction (the first one after the factorisation)
            if not fast:
                dkappa = -st8["dkappa"] / lmbda_g
                dtau0 = st8["dtau"] + dkappa / dgi
                raise_for(lib().kvx_lp_second_half_dev(ml, n, p, cv.ptr, bv.ptr if p else None, th.ptr, x1.ptr, y1.ptr if p else None,
                                                       z1.ptr, lmbda.ptr, dx.ptr, dy.ptr if p else None, dz.ptr, ds.ptr,
                                                       ws3.ptr if i == 0 else None, dgi, dtau0, z1z1, out4))
            dtau, z1z1, ts, tz = out4[0], out4[1], out4[2], out4[3]
            t_now = time.perf_counter()
            t_phase[i] += t_now - t_mark[0]
            if _TRACE:
                print("conelp iteration %d direction %d: %.0f us" % (iters, i, 1e6 * (t_now - t_mark[0])), file=sys.stderr)
            t_mark[0] = t_now
            if i == 0:
                try:
                    kkt.check()                          # the stream is idle by now: no extra wait
                except ArithmeticError:
                    kkt.async_solves = False
                    x.scal(1.0 / tau); y.scal(1.0 / tau); s.scal(1.0 / tau); z.scal(1.0 / tau)
                    return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, pinfres, dinfres,
                                  msg="Terminated (singular KKT matrix).")
            dkappa -= dtau
            if i == 0:
                wkappa3 = dtau * dkappa
            tt = -dtau / lmbda_g
            tk = -dkappa / lmbda_g
            t = max(0.0, ts, tz, tt, tk)
            if t == 0.0:
                step = 1.0
            else:
                step = min(1.0, 1.0 / t) if i == 0 else min(1.0, STEP / t)
            if i == 0:
                sigma = (1.0 - step) ** EXPON

        # update (coneprog.py:1336-1436)
        # scaled iterates, NT scaling update and unscaled s, z (coneprog.py:1343-1432, misc.py:444-464): one fused kernel
        if fast:
            pass                                         # (with the residuals of the next iteration, below)
        elif fused:
            raise_for(lib().kvx_lp_update_x_dev(ml, n, step, ds.ptr, dz.ptr, d.ptr, di.ptr, lmbda.ptr, s.ptr, z.ptr, dx.ptr, x.ptr))
        else:
            x.axpy(dx, step)
            y.axpy(dy, step)
            raise_for(lib().kvx_lp_update_dev(ml, step, ds.ptr, dz.ptr, d.ptr, di.ptr, lmbda.ptr, s.ptr, z.ptr))
        dg *= math.sqrt(1.0 - step * tk) / math.sqrt(1.0 - step * tt)
        dgi = 1.0 / dg
        lmbda_g *= math.sqrt(1.0 - step * tt) * math.sqrt(1.0 - step * tk)
        kappa, tau = lmbda_g / dgi, lmbda_g * dgi
        if fast:
            # the update, the residuals of the next iteration (with the new tau) and their reductions in ONE call
            raise_for(lib().kvx_lp_iter_update(ctypes.byref(ctx), step, tau, out10))
            next_stats = tuple(out10)
    raise AssertionError("unreachable")


class SymSpMatDev:
    """Symmetric sparse matrix resident in HBM, given by its lower triangle (the 'L' storage base.symv reads for
    the quadratic term, coneprog.py:1889-1893): y := alpha*P*x + beta*y as one pass over the lower triangle and one
    transposed pass over its strictly lower part."""

    def __init__(self, n, colptr, rowind, values):
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowind = np.ascontiguousarray(rowind, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=np.float64)
        cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(colptr))
        if np.any(rowind < cols):
            raise ValueError("P must be given by its lower triangle")
        strict = rowind > cols
        sp = np.zeros(n + 1, dtype=np.int64)
        np.add.at(sp, cols[strict] + 1, 1)
        np.cumsum(sp, out=sp)
        self.low = SpMatDev(n, n, colptr, rowind, values)
        self.strict = SpMatDev(n, n, sp, rowind[strict], values[strict])

    def symv(self, x, y, alpha=1.0, beta=0.0):
        self.low.gemv(x, y, trans="N", alpha=alpha, beta=beta)
        self.strict.gemv(x, y, trans="T", alpha=alpha, beta=1.0)


def _lower_ccs(P, n):
    """Lower triangle (i >= j) of a spmatrix-like P as sorted CCS; entries above the diagonal are ignored, as the
    reference's symmetric kernels do."""
    m, n2, Pp, Pi, Px = base._as_ccs(P)
    if m != n or n2 != n:
        raise TypeError("'P' must be a 'd' matrix of size (%d, %d)" % (n, n))
    Pp = np.asarray(Pp, dtype=np.int64); Pi = np.asarray(Pi, dtype=np.int64); Px = np.asarray(Px, dtype=np.float64)
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(Pp))
    keep = Pi >= cols
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, cols[keep] + 1, 1)
    np.cumsum(cp, out=cp)
    return cp, Pi[keep].copy(), Px[keep].copy()


def coneqp(P, q, G, h, options=None, chol_opts=None, A=None, b=None, initvals=None, kktsolver=None):
    """Solve the convex QP  minimize (1/2) x'Px + q'x  s.t.  Gx <= h, Ax = b  on the GPU (orthant cone): the reference's coneqp (coneprog.py:1440-2547) with its default KKT solver for sparse G,
    misc.kkt_chol2 with H = P.  P: spmatrix-like, its lower triangle is used.  Returns the reference's result
    dictionary (coneprog.py:2216-2221) with numpy arrays, plus "factorizations"."""
    _lib.require_device()
    opts = {"maxiters": 100, "abstol": 1e-7, "reltol": 1e-6, "feastol": 1e-7, "show_progress": False,
            "refinement": 0, "use_correction": True}                        # coneprog.py:1768-1781, 1862-1865
    opts.update(options or {})
    MAXITERS, ABSTOL, RELTOL, FEASTOL = opts["maxiters"], opts["abstol"], opts["reltol"], opts["feastol"]
    show, refinement, correction = opts["show_progress"], int(opts["refinement"]), bool(opts["use_correction"])
    ml, n, Gp, Gi, Gx = base._as_ccs(G)
    q_h = np.asarray(q._a if isinstance(q, base.matrix) else q, dtype=np.float64).reshape(-1)
    h_h = np.asarray(h._a if isinstance(h, base.matrix) else h, dtype=np.float64).reshape(-1)
    if q_h.size != n or h_h.size != ml:
        raise TypeError("dimensions of q, G, h do not match")
    if ml == 0:
        raise ValueError("coneqp on the GPU needs at least one inequality (dims['l'] > 0)")
    Pp, Pi, Px = _lower_ccs(P, n)
    p = 0
    if A is not None:
        p, na, Ap, Ai, Ax = base._as_ccs(A)
        if na != n:
            raise TypeError("'A' must have %d columns" % n)
        b_h = np.asarray(b._a if isinstance(b, base.matrix) else b, dtype=np.float64).reshape(-1)
        if b_h.size != p:
            raise TypeError("'b' must have length %d" % p)
    if kktsolver is not None:
        # the reference's plug-in point (coneprog.py:1969-1981): kktsolver(W) returns f(x, y, z) for the system with H = P; host
        # round trips per factorisation and solve (KKTUserHost, as under conelp)
        if not callable(kktsolver):
            raise ValueError("kktsolver must be a function W -> f(x, y, z) (the reference's named solvers are not part of this path: "
                             "'chol2' is what runs on the GPU by default)")
        if p == 0:
            Ap, Ai, Ax = np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0)
        kkt = KKTUserHost(ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, kktsolver)
    elif p > 0:
        kkt = _kkt_for("KKTGenEqDev+P", (ml, n, p), (Gp, Gi, Ap, Ai, Pp, Pi), chol_opts,
                       lambda: KKTGenEqDev(ml, n, Gp, Gi, Gx, p, Ap, Ai, Ax, chol_opts, Pp, Pi, Px), lambda k: k.reset(Gx, Ax, Px))
    else:
        kkt = _kkt_for("KKTChol2Dev+P", (ml, n, 0), (Gp, Gi, Pp, Pi), chol_opts,
                       lambda: KKTChol2Dev(ml, n, Gp, Gi, Gx, chol_opts, Pp, Pi, Px), lambda k: k.reset(Gx, Px))
    if p > 0:
        Ad = kkt.A
        bv = DVec(p, b_h)
        y, dy, ry = DVec(p), DVec(p), DVec(p)
        if refinement:
            wy, wy2 = DVec(p), DVec(p)
        resy0 = max(1.0, bv.nrm2())
        ksolve = kkt.solve
    else:
        y = dy = ry = wy = wy2 = None
        resy0 = 1.0
        if kktsolver is not None:
            ksolve = kkt.solve
        else:
            def ksolve(xx, yy, zz):
                kkt.solve(xx, zz)
    Gd, Pd = kkt.G, SymSpMatDev(n, Pp, Pi, Px)
    qv, hv = DVec(n, q_h), DVec(ml, h_h)
    x, dx, rx, tmpx = (DVec(n) for _ in range(4))
    s, z, ds, dz, rz, ws3, tmp, lmbda, lmbdasq, d, di = (DVec(ml) for _ in range(11))
    if refinement:
        wx, wx2 = DVec(n), DVec(n)
        wz, ws, wz2, ws2 = (DVec(ml) for _ in range(4))
    resx0 = max(1.0, qv.nrm2())
    resz0 = max(1.0, hv.nrm2())

    def result(status, iters, gap, relgap, pcost, dcost, pres, dres, msg=None):
        if show:                                         # coneprog.py:2222-2227, 2269
            print(msg or "Optimal solution found.")
        return {"x": x.get(), "y": y.get() if p else np.zeros(0), "s": s.get(), "z": z.get(), "status": status, "gap": gap,
                "relative gap": relgap, "primal objective": pcost, "dual objective": dcost,
                "primal infeasibility": pres, "dual infeasibility": dres,
                "primal slack": -s.max_step(), "dual slack": -z.max_step(), "iterations": iters,
                "factorizations": kkt.nfactor}

    # ---- starting point (coneprog.py:2044-2150)
    if initvals is None:
        # factor with W = I, solve [P A' G'; A 0 0; G 0 -I][x; y; z] = [-q; b; h], s = -z, push s and z into the cone
        d.fill(1.0); di.fill(1.0)
        try:
            kkt.factor(di)
        except ArithmeticError:
            raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
        x.copy_from(qv).scal(-1.0)
        if p:
            y.copy_from(bv)
        z.copy_from(hv)
        try:
            ksolve(x, y, z)
            if p:
                kkt.check()
        except ArithmeticError:
            raise ValueError("Rank(A) < p or Rank([P; G; A]) < n")
        s.copy_from(z).scal(-1.0)
        ts = s.max_step()
        if ts >= -1e-8 * max(s.nrm2(), 1.0):
            s.addc(1.0 + ts)
        tz = z.max_step()
        if tz >= -1e-8 * max(z.nrm2(), 1.0):
            z.addc(1.0 + tz)
    else:
        # user-supplied values (coneprog.py:2108-2150): missing x, y default to 0, missing s, z to the cone's identity
        def _vec(v, length):
            a = np.ascontiguousarray(np.asarray(v._a if isinstance(v, base.matrix) else v, dtype=np.float64).reshape(-1))
            if a.size != length:
                raise TypeError("initvals has the wrong dimensions")
            return a
        x.set(_vec(initvals["x"], n)) if "x" in initvals else x.fill(0.0)
        if "s" in initvals:
            s.set(_vec(initvals["s"], ml))
            if s.max_step() >= 0:
                raise ValueError("initial s is not positive")
        else:
            s.fill(1.0)
        if p:
            y.set(_vec(initvals["y"], p)) if "y" in initvals else y.fill(0.0)
        if "z" in initvals:
            z.set(_vec(initvals["z"], ml))
            if z.max_step() >= 0:
                raise ValueError("initial z is not positive")
        else:
            z.fill(1.0)
    gap = s.dot(z)

    def f4_no_ir(bx, by, bz, bs):
        # [P A' G'; A 0 0; G 0 -W'W][ux; uy; W^-1 uz] = [bx; by; bz - W'(lmbda o\ bs)],  us = lmbda o\ bs - uz   (coneprog.py:2283-2313)
        bs.div(lmbda)
        tmp.xmy(1.0, bs, d)
        bz.axpy(tmp, -1.0)
        ksolve(bx, by, bz)
        bs.axpy(bz, -1.0)

    def res(ux, uy, uz, us, vx, vy, vz, vs):
        # residual of the Newton equations (coneprog.py:1929-1960)
        Pd.symv(ux, vx, alpha=-1.0, beta=1.0)
        if p:
            Ad.gemv(uy, vx, trans="T", alpha=-1.0, beta=1.0)
            Ad.gemv(ux, vy, trans="N", alpha=-1.0, beta=1.0)
        tmp.xmy(1.0, uz, di)
        Gd.gemv(tmp, vx, trans="T", alpha=-1.0, beta=1.0)
        Gd.gemv(ux, vz, trans="N", alpha=-1.0, beta=1.0)
        tmp.xmy(1.0, us, d)
        vz.axpy(tmp, -1.0)
        tmp.lincomb(1.0, us, 1.0, uz)
        tmp.mul(lmbda)
        vs.axpy(tmp, -1.0)

    def f4(bx, by, bz, bs):
        if refinement:
            wx.copy_from(bx); wz.copy_from(bz); ws.copy_from(bs)
            if p:
                wy.copy_from(by)
        f4_no_ir(bx, by, bz, bs)
        for _ in range(refinement):
            wx2.copy_from(wx); wz2.copy_from(wz); ws2.copy_from(ws)
            if p:
                wy2.copy_from(wy)
            res(bx, by, bz, bs, wx2, wy2, wz2, ws2)
            f4_no_ir(wx2, wy2, wz2, ws2)
            bx.axpy(wx2); bz.axpy(wz2); bs.axpy(ws2)
            if p:
                by.axpy(wy2)

    for iters in range(MAXITERS + 1):
        # residuals and objectives (coneprog.py:2167-2203): one reduction call for the five inner products
        rx.copy_from(qv)
        Pd.symv(x, rx, alpha=1.0, beta=1.0)
        tmpx.copy_from(rx)                               # P x + q, for f0
        if p:
            Ad.gemv(y, rx, trans="T", alpha=1.0, beta=1.0)
            ry.copy_from(bv)
            Ad.gemv(x, ry, trans="N", alpha=1.0, beta=-1.0)      # ry = A x - b
        Gd.gemv(z, rx, trans="T", alpha=1.0, beta=1.0)
        rz.lincomb(1.0, s, -1.0, hv)
        Gd.gemv(x, rz, trans="N", alpha=1.0, beta=1.0)
        xPq, xq, v_rx, v_rz, zrz, v_ry, yry = reduce_multi([("dot", x, tmpx), ("dot", x, qv), ("dot", rx, rx), ("dot", rz, rz),
                                                             ("dot", z, rz), ("dot", ry, ry), ("dot", y, ry)])
        f0 = 0.5 * (xPq + xq)
        resx, resz, resy = math.sqrt(v_rx), math.sqrt(v_rz), math.sqrt(v_ry)
        pcost = f0
        dcost = f0 + yry + zrz - gap
        relgap = gap / -pcost if pcost < 0.0 else (gap / dcost if dcost > 0.0 else None)
        pres, dres = max(resy / resy0, resz / resz0), resx / resx0
        if show:
            if iters == 0:
                print("% 10s% 12s% 10s% 8s% 7s" % ("pcost", "dcost", "gap", "pres", "dres"))
            print("%2d: % 8.4e % 8.4e % 4.0e% 7.0e% 7.0e" % (iters, pcost, dcost, gap, pres, dres))
        if (pres <= FEASTOL and dres <= FEASTOL and (gap <= ABSTOL or (relgap is not None and relgap <= RELTOL))) \
                or iters == MAXITERS:
            if iters == MAXITERS:
                return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, msg="Terminated (maximum number of iterations reached).")
            return result("optimal", iters, gap, relgap, pcost, dcost, pres, dres)

        # scaling (coneprog.py:2230-2231) and KKT factorisation
        if iters == 0:
            raise_for(lib().kvx_nt_compute_scaling_dev(ml, s.ptr, z.ptr, d.ptr, di.ptr, lmbda.ptr))
        lmbdasq.sqr_of(lmbda)
        try:
            kkt.factor(di)
            if p:
                kkt.check()
        except ArithmeticError:
            if iters == 0:
                raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
            return result("unknown", iters, gap, relgap, pcost, dcost, pres, dres, msg="Terminated (singular KKT matrix).")

        mu = gap / ml
        sigma, eta = 0.0, 0.0
        for i in (0, 1):
            # right-hand sides (coneprog.py:2367-2390)
            ds.fill(0.0)
            if correction and i == 1:
                ds.axpy(ws3, -1.0)
            ds.axpy(lmbdasq, -1.0).addc(sigma * mu)
            dx.lincomb(-1.0 + eta, rx)
            if p:
                dy.lincomb(-1.0 + eta, ry)
            dz.lincomb(-1.0 + eta, rz)
            f4(dx, dy, dz, ds)
            dsdz = ds.dot(dz)
            if correction and i == 0:
                ws3.xmy(1.0, ds, dz)
            # step to the boundary (coneprog.py:2431-2451)
            ds.div(lmbda); dz.div(lmbda)
            t = max(0.0, *reduce_multi([("max", ds), ("max", dz)]))
            if t == 0.0:
                step = 1.0
            else:
                step = min(1.0, 1.0 / t) if i == 0 else min(1.0, STEP / t)
            if i == 0:
                sigma = min(1.0, max(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** EXPON
                eta = 0.0

        # update iterates and scaling (coneprog.py:2454-2545)
        x.axpy(dx, step)
        if p:
            y.axpy(dy, step)
        ds.scal(step).addc(1.0); dz.scal(step).addc(1.0)
        ds.mul(lmbda); dz.mul(lmbda)
        raise_for(lib().kvx_nt_update_scaling_dev(ml, ds.ptr, dz.ptr, d.ptr, di.ptr, lmbda.ptr))
        s.xmy(1.0, lmbda, d)
        z.xmy(1.0, lmbda, di)
        gap = lmbda.dot(lmbda)
    raise AssertionError("unreachable")
