"""`kvxopt.misc` / `kvxopt.misc_solvers` on MI355X -- the Nesterov-Todd scaling operations for the nonlinear, 'l', 'q' and
's' blocks and the storage helpers of the 's' blocks -- plus the KKT solver factory `kkt_chol2`: same names, argument
meaning and in-place semantics as the reference (src/python/misc.py, src/C/misc_solvers.c).  kkt_chol2 itself takes
'l' (and nonlinear) blocks only, as in the reference (misc.py:1381-1384).  The nonlinear block of cvxprog (`mnl` leading
entries scaled by W['dnl'], misc.py:262-270, 432-442, 48-60) is one more diagonal block in front of the 'l' block, so
every kernel simply runs over mnl + ml entries.

Host-array compatibility layer: arguments are host `matrix` objects (ours or kvxopt's); every
operation runs through the HIP kernels of libkvxhip.so (upload, kernel, download).  The device-
resident fast path for whole interior-point iterations is `kvxopt_amd.lp`.
"""
import ctypes

import numpy as np

from . import _lib, base
from ._lib import DeviceBuffer, lib, raise_for
from .base import matrix, spmatrix


class _SBlocks:
    """Device tables of the 's' section of a vector: offsets of the m_k x m_k blocks, of their diagonals / eigenvalues and
    of their packed lower triangles (csrc/kkt_s.hip: one workgroup per block, all blocks in one launch)."""

    def __init__(self, sdims):
        self.dims = [int(m) for m in sdims]
        if any(m < 0 or m > 4096 for m in self.dims):
            raise ValueError("semidefinite blocks must have an order between 0 and 4096 (one workgroup per block)")
        self.ns = len(self.dims)
        d = np.asarray(self.dims, dtype=np.int64)
        self.off2, self.off1, self.offp = (np.zeros(self.ns + 1, dtype=np.int64) for _ in range(3))
        np.cumsum(d * d, out=self.off2[1:])
        np.cumsum(d, out=self.off1[1:])
        np.cumsum(d * (d + 1) // 2, out=self.offp[1:])
        self.tot2, self.tot1, self.totp = int(self.off2[-1]), int(self.off1[-1]), int(self.offp[-1])
        self.d2, self.d1, self.dp = (DeviceBuffer.from_array(a) for a in (self.off2, self.off1, self.offp))

    def split(self, flat):
        """list of m_k x m_k matrices from the concatenated blocks"""
        return [matrix(flat[self.off2[k]:self.off2[k + 1]].copy(), (m, m)) for k, m in enumerate(self.dims)]


def _s_blocks(dims):
    sd = [int(m) for m in (dims.get("s") or [])]
    return _SBlocks(sd) if sd and sum(sd) else None


def _cat(mats):
    return np.concatenate([np.asarray(_buf(a)[0], dtype=np.float64) for a in mats]) if mats else np.zeros(0)


def _scratch(n):
    return DeviceBuffer(8 * max(int(n), 1))


def _q_offsets(q):
    """Device table of the cone boundaries inside the 'q' section: [0, q0, q0 + q1, ...]."""
    off = np.zeros(len(q) + 1, dtype=np.int64)
    np.cumsum(np.asarray(q, dtype=np.int64), out=off[1:])
    return off, DeviceBuffer.from_array(off)


def _q_apply(dims, m, call, *host_vecs):
    """Run one 'q'-block kernel on the slices [m, m + sum(q)) of the host vectors: `call(nq, off_dev, *device slices)`;
    the first vector is written back."""
    q = list(dims.get("q") or [])
    if not q:
        return
    off, doff = _q_offsets(q)
    tot = int(off[-1])
    bufs = [_buf(v)[0] for v in host_vecs]
    devs = [_up(b[m:m + tot]) for b in bufs]
    raise_for(call(len(q), doff.ptr, *[d.ptr for d in devs]))
    _sync()
    bufs[0][m:m + tot] = devs[0].download(np.float64, tot)


def _buf(x):
    return base._dense_buffer(x)


def _up(a):
    return DeviceBuffer.from_array(np.ascontiguousarray(a, dtype=np.float64))


def _sync():
    raise_for(lib().kvx_dev_sync())


def compute_scaling(s, z, lmbda, dims, mnl=None):
    """misc.py:250-352 (nonlinear, 'l' and 'q' blocks): W['d'] = sqrt(s./z), W['di'] = 1./d, lmbda = sqrt(s.*z); with
    mnl given (cvxprog), the first mnl entries make W['dnl'], W['dnli'] by the same formulas; for every second-order cone
    the unit-hyperbolic-norm vector W['v'][k] and W['beta'][k] with (beta_k (2 v_k v_k' - J)) z_k = lambda_k."""
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
    q = list(dims.get("q") or [])
    if q:
        # 'q' blocks (misc.py:290-352): hyperbolic Householder scaling beta_k (2 v_k v_k' - J), one workgroup per cone
        off, doff = _q_offsets(q)
        tot = int(off[-1])
        dsq, dzq = _up(sb[m:m + tot]), _up(zb[m:m + tot])
        dv, dlq, db = DeviceBuffer(8 * tot), DeviceBuffer(8 * tot), DeviceBuffer(8 * len(q))
        raise_for(lib().kvx_ntq_compute_scaling_dev(len(q), doff.ptr, dsq.ptr, dzq.ptr, dv.ptr, db.ptr, dlq.ptr))
        _sync()
        vall = dv.download(np.float64, tot)
        W["v"] = [matrix(vall[off[i]:off[i + 1]].copy(), (q[i], 1)) for i in range(len(q))]
        W["beta"] = [float(b) for b in db.download(np.float64, len(q))]
        lb[m:m + tot] = dlq.download(np.float64, tot)
    S = _s_blocks(dims)
    if S is not None:
        # 's' blocks (misc.py:354-419): r_k' z_k r_k = r_k^-1 s_k r_k^-T = diag(lambda_k), rti_k = r_k^-T; lambda_k is the
        # vector of singular values of Lz' Ls (descending), stored behind the 'q' part of lmbda
        ind = m + sum(q)
        dss, dzs = _up(sb[ind:ind + S.tot2]), _up(zb[ind:ind + S.tot2])
        dr, drt, dls, wk = _scratch(S.tot2), _scratch(S.tot2), _scratch(S.tot1), _scratch(4 * S.tot2)
        st = DeviceBuffer.from_array(np.array([_NOFAIL], dtype=np.int32))
        raise_for(lib().kvx_nts_compute_scaling_dev(S.ns, S.d2.ptr, S.d1.ptr, dss.ptr, dzs.ptr, dr.ptr, drt.ptr, dls.ptr, wk.ptr, st.ptr))
        _sync()
        bad = int(st.download(np.int32, 1)[0])
        if bad != _NOFAIL:
            raise ArithmeticError(bad + 1)                  # lapack.potrf on a block that is not positive definite
        W["r"], W["rti"] = S.split(dr.download(np.float64, S.tot2)), S.split(drt.download(np.float64, S.tot2))
        lb[ind:ind + S.tot1] = dls.download(np.float64, S.tot1)
    return W


_NOFAIL = 2 ** 31 - 1


def _diag_of(W, inverse=False):
    """[dnl; d] (or the inverses) as one contiguous vector, and the length of the nonlinear part."""
    d, _ = _buf(W["di"] if inverse else W["d"])
    if "dnl" in W:
        dn, _ = _buf(W["dnli"] if inverse else W["dnl"])
        return np.concatenate([dn, d]), dn.size
    return np.ascontiguousarray(d), 0


def update_scaling(W, lmbda, s, z):
    """misc.py:422-634, in place.  Nonlinear and 'l' blocks: s:=sqrt(s), z:=sqrt(z), d:=d.*s./z, di:=1./d, lmbda:=s.*z
    (W['dnl'], W['dnli'] likewise on the leading mnl entries); 'q' and 's' blocks below."""
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
    if W.get("v"):
        # 'q' blocks (misc.py:467-580): s, z leave as st / a, zt / b; v, beta, lambda are updated
        q = [int(_buf(v)[1][0]) for v in W["v"]]
        off, doff = _q_offsets(q)
        tot = int(off[-1])
        dsq, dzq, dlq = _up(sb[m:m + tot]), _up(zb[m:m + tot]), DeviceBuffer(8 * tot)
        dv = _up(np.concatenate([np.asarray(_buf(v)[0], dtype=np.float64) for v in W["v"]]))
        db = _up(np.asarray(W["beta"], dtype=np.float64))
        raise_for(lib().kvx_ntq_update_scaling_dev(len(q), doff.ptr, dsq.ptr, dzq.ptr, dv.ptr, db.ptr, dlq.ptr))
        _sync()
        sb[m:m + tot] = dsq.download(np.float64, tot)
        zb[m:m + tot] = dzq.download(np.float64, tot)
        lb[m:m + tot] = dlq.download(np.float64, tot)
        vall = dv.download(np.float64, tot)
        for i, v in enumerate(W["v"]):
            _buf(v)[0][:] = vall[off[i]:off[i + 1]]
        W["beta"][:] = [float(b) for b in db.download(np.float64, len(q))]
    if W.get("r"):
        # 's' blocks (misc.py:582-634): s_k, z_k hold the Cholesky factors Ls, Lz of the new iterates in the current scaling;
        # with Lz' Ls = U diag(lambda_k) V':  r_k := r_k Ls V diag(lambda_k)^-1/2, rti_k := rti_k Lz U diag(lambda_k)^-1/2;
        # s_k, z_k leave as U and V' (what lapack.gesvd stores there)
        S = _SBlocks([int(_buf(r)[1][0]) for r in W["r"]])
        ind = m + sum(int(_buf(v)[1][0]) for v in (W.get("v") or []))
        dss, dzs = _up(sb[ind:ind + S.tot2]), _up(zb[ind:ind + S.tot2])
        dr, drt = _up(_cat(W["r"])), _up(_cat(W["rti"]))
        dls, wk = _scratch(S.tot1), _scratch(4 * S.tot2)
        raise_for(lib().kvx_nts_update_scaling_dev(S.ns, S.d2.ptr, S.d1.ptr, dss.ptr, dzs.ptr, dr.ptr, drt.ptr, dls.ptr, wk.ptr))
        _sync()
        sb[ind:ind + S.tot2] = dss.download(np.float64, S.tot2)
        zb[ind:ind + S.tot2] = dzs.download(np.float64, S.tot2)
        lb[ind:ind + S.tot1] = dls.download(np.float64, S.tot1)
        rall, tall = dr.download(np.float64, S.tot2), drt.download(np.float64, S.tot2)
        for i in range(S.ns):
            _buf(W["r"][i])[0][:] = rall[S.off2[i]:S.off2[i + 1]]
            _buf(W["rti"][i])[0][:] = tall[S.off2[i]:S.off2[i + 1]]


def scale(x, W, trans="N", inverse="N"):
    """misc_solvers.c:85-240 / misc.py:36-164.  Nonlinear and 'l' blocks: x := [dnl; d].*x ('N') or [dnli; di].*x ('I') for
    every column of x (trans is irrelevant for a diagonal scaling); 'q' and 's' blocks below."""
    w, _ = _diag_of(W, inverse != "N")
    xb, size = _buf(x)
    m = w.size
    nq = len(W.get("v") or [])
    nsb = len(W.get("r") or [])
    if m == 0 and nq == 0 and nsb == 0:
        return
    _lib.require_device()
    dx = _up(xb)
    if m:
        dw = _up(w)
        raise_for(lib().kvx_nt_scale_dev(m, size[1], size[0], dx.ptr, dw.ptr))
    if nq:
        # 'q' blocks (misc_solvers.c:144-186): x_k := beta_k (2 v_k v_k' - J) x_k, or the inverse; symmetric, so `trans` is moot
        q = [int(_buf(v)[1][0]) for v in W["v"]]
        off, doff = _q_offsets(q)
        dv = _up(np.concatenate([np.asarray(_buf(v)[0], dtype=np.float64) for v in W["v"]]))
        db = _up(np.asarray(W["beta"], dtype=np.float64))
        raise_for(lib().kvx_ntq_scale_dev(nq, doff.ptr, dv.ptr, db.ptr, dx.ptr + 8 * m, size[0], size[1], 1 if inverse != "N" else 0))
    if nsb:
        # 's' blocks (misc_solvers.c:188-240): x_k := r' X r ('N','N'), r X r' ('T','N'), rti X rti' ('N','I'), rti' X rti
        # ('T','I'), X the symmetric matrix in the lower triangle of x_k; only the lower triangle is written
        R = W["r"] if inverse == "N" else W["rti"]
        S = _SBlocks([int(_buf(r)[1][0]) for r in R])
        ind = m + sum(int(_buf(v)[1][0]) for v in (W.get("v") or []))
        dR, wk = _up(_cat(R)), _scratch(S.tot2 * size[1])
        form = 1 if (inverse == "N") == (trans == "T") else 0
        raise_for(lib().kvx_nts_scale_dev(S.ns, S.d2.ptr, S.d1.ptr, dR.ptr, dx.ptr + 8 * ind, size[0], size[1], form, wk.ptr, S.tot2))
    _sync()
    xb[:] = dx.download(np.float64, xb.size)


def scale2(lmbda, x, dims, mnl=0, inverse="N"):
    """misc_solvers.c:256-397.  Nonlinear and 'l' blocks: x := x./lmbda ('N') or x.*lmbda ('I'); 'q' blocks: the hyperbolic
    form (:301-341); 's' blocks: x_k(i, j) divided ('N') or multiplied ('I') by sqrt(l_i) sqrt(l_j) (:343-397)."""
    m = mnl + dims["l"]
    lb, _ = _buf(lmbda)
    xb, _ = _buf(x)
    _lib.require_device()
    if m:
        dx, dl = _up(xb[:m]), _up(lb[:m])
        raise_for(lib().kvx_nt_scale2_dev(m, dl.ptr, dx.ptr, 1 if inverse == "I" else 0))
        _sync()
        xb[:m] = dx.download(np.float64, m)
    inv = 1 if inverse == "I" else 0                      # 'q' blocks: misc_solvers.c:301-341
    _q_apply(dims, m, lambda nq, off, dxq, dlq: lib().kvx_ntq_scale2_dev(nq, off, dlq, dxq, inv), x, lmbda)
    S = _s_blocks(dims)
    if S is not None:
        ind = m + sum(dims.get("q") or [])
        dxs, dls = _up(xb[ind:ind + S.tot2]), _up(lb[ind:ind + S.tot1])
        raise_for(lib().kvx_nts_scale2_dev(S.ns, S.d2.ptr, S.d1.ptr, dls.ptr, dxs.ptr, inv))
        _sync()
        xb[ind:ind + S.tot2] = dxs.download(np.float64, S.tot2)


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
    """misc_solvers.c:634-770: x := y o x.  'l' block: x.*y; 'q' blocks :671-700; 's' blocks :700-770: the lower triangle of
    (Y X + X Y) / 2, with y holding full blocks (diag 'N'; their upper triangles are filled in as the reference does) or
    only their diagonals (diag 'D')."""
    _binary(lib().kvx_nt_sprod_dev, x, y, mnl + dims["l"])
    _q_apply(dims, mnl + dims["l"], lambda nq, off, dx, dy: lib().kvx_ntq_prod_dev(nq, off, dx, dy, 0), x, y)   # misc_solvers.c:671-700
    S = _s_blocks(dims)
    if S is not None:
        ind = mnl + dims["l"] + sum(dims.get("q") or [])
        xb, yb = _buf(x)[0], _buf(y)[0]
        dxs = _up(xb[ind:ind + S.tot2])
        if diag == "N":
            dys, wk = _up(yb[ind:ind + S.tot2]), _scratch(S.tot2)
            raise_for(lib().kvx_nts_prod_dev(S.ns, S.d2.ptr, S.d1.ptr, dxs.ptr, dys.ptr, 0, wk.ptr))
            _sync()
            yb[ind:ind + S.tot2] = dys.download(np.float64, S.tot2)
        else:
            dys = _up(yb[ind:ind + S.tot1])
            raise_for(lib().kvx_nts_prod_dev(S.ns, S.d2.ptr, S.d1.ptr, dxs.ptr, dys.ptr, 1, None))
            _sync()
        xb[ind:ind + S.tot2] = dxs.download(np.float64, S.tot2)


def sinv(x, y, dims, mnl=0):
    """misc_solvers.c:775-882: the inverse of x := y o x.  'l' block: x./y; 'q' blocks :803-835; 's' blocks :845-882 (y holds
    only the diagonals): the lower triangle of x_k divided entrywise by (y_i + y_j) / 2."""
    _binary(lib().kvx_nt_sinv_dev, x, y, mnl + dims["l"])
    _q_apply(dims, mnl + dims["l"], lambda nq, off, dx, dy: lib().kvx_ntq_prod_dev(nq, off, dx, dy, 1), x, y)   # misc_solvers.c:803-835
    S = _s_blocks(dims)
    if S is not None:
        ind = mnl + dims["l"] + sum(dims.get("q") or [])
        xb, yb = _buf(x)[0], _buf(y)[0]
        dxs, dys = _up(xb[ind:ind + S.tot2]), _up(yb[ind:ind + S.tot1])
        raise_for(lib().kvx_nts_prod_dev(S.ns, S.d2.ptr, S.d1.ptr, dxs.ptr, dys.ptr, 2, None))
        _sync()
        xb[ind:ind + S.tot2] = dxs.download(np.float64, S.tot2)


def ssqr(x, y, dims, mnl=0):
    """misc.py:945-959: x := y o y; the 's' components of x and y are diagonal and only the diagonals are stored, so they are
    squared entrywise like the 'l' block."""
    _binary(lib().kvx_nt_ssqr_dev, x, y, mnl + dims["l"])
    _q_apply(dims, mnl + dims["l"], lambda nq, off, dx, dy: lib().kvx_ntq_prod_dev(nq, off, dx, dy, 2), x, y)   # misc.py:951-959
    ns1 = sum(int(m) for m in (dims.get("s") or []))
    if ns1:
        ind = mnl + dims["l"] + sum(dims.get("q") or [])
        xb, yb = _buf(x)[0], _buf(y)[0]
        dxs, dys = _up(xb[ind:ind + ns1]), _up(yb[ind:ind + ns1])
        raise_for(lib().kvx_nt_ssqr_dev(ns1, dxs.ptr, dys.ptr))
        _sync()
        xb[ind:ind + ns1] = dxs.download(np.float64, ns1)


def sdot(x, y, dims, mnl=0):
    """misc_solvers.c:991-1046: sum_i x_i*y_i over the nonlinear, 'l' and 'q' entries plus, per 's' block, the trace inner
    product of the symmetric matrices stored in the lower triangles (diagonal + twice the strict lower part)."""
    m = mnl + dims["l"] + sum(dims.get("q") or [])          # misc_solvers.c:1009-1012: one dot over the 'l' and 'q' entries
    xb, _ = _buf(x)
    yb, _ = _buf(y)
    S = _s_blocks(dims)
    if m == 0 and S is None:
        return 0.0
    _lib.require_device()
    a = 0.0
    if m:
        dx, dy = _up(xb[:m]), _up(yb[:m])
        r = ctypes.c_double()
        raise_for(lib().kvx_nt_sdot_dev(m, dx.ptr, dy.ptr, ctypes.byref(r)))
        a = r.value
    if S is not None:
        dxs, dys, dout = _up(xb[m:m + S.tot2]), _up(yb[m:m + S.tot2]), _scratch(S.ns)
        raise_for(lib().kvx_nts_dot_dev(S.ns, S.d2.ptr, S.d1.ptr, dxs.ptr, dys.ptr, dout.ptr))
        _sync()
        for v in dout.download(np.float64, S.ns):          # block after block, as the reference accumulates
            a += float(v)
    return a


def max_step(x, dims, mnl=0, sigma=None):
    """misc_solvers.c:1052-1160: min {t | x + t e >= 0} = max(max_i(-x_i), max_k(|x_k1| - x_k0), max_k(-lambda_min(x_k)));
    0.0 for an empty x (misc_solvers.c:1099).  With `sigma` the eigenvalues of the 's' blocks (ascending) are stored there and
    their eigenvectors replace the blocks of x (:1128-1133)."""
    m = mnl + dims["l"]
    q = list(dims.get("q") or [])
    xb, _ = _buf(x)
    S = _s_blocks(dims)
    if m + sum(q) == 0 and S is None:
        return 0.0
    _lib.require_device()
    t = -np.finfo(np.float32).max                        # the reference starts from -FLT_MAX (misc_solvers.c:1063)
    if m:
        dx = _up(xb[:m])
        r = ctypes.c_double()
        raise_for(lib().kvx_nt_max_step_dev(m, dx.ptr, ctypes.byref(r)))
        t = max(t, r.value)
    if q:
        off, doff = _q_offsets(q)
        tot = int(off[-1])
        dxq, dout = _up(xb[m:m + tot]), DeviceBuffer(8 * len(q))
        raise_for(lib().kvx_ntq_max_step_dev(len(q), doff.ptr, dxq.ptr, dout.ptr))
        _sync()
        t = max(t, float(dout.download(np.float64, len(q)).max()))
    if S is not None:
        ind = m + sum(q)
        dxs, dout, wk = _up(xb[ind:ind + S.tot2]), _scratch(S.ns), _scratch(3 * S.tot2 + 2 * S.tot1)
        dsg = _scratch(S.tot1) if sigma is not None else None
        raise_for(lib().kvx_nts_max_step_dev(S.ns, S.d2.ptr, S.d1.ptr, dxs.ptr, dsg.ptr if dsg else None, dout.ptr, wk.ptr))
        _sync()
        t = max(t, float(dout.download(np.float64, S.ns).max()))
        if sigma is not None:
            _buf(sigma)[0][:S.tot1] = dsg.download(np.float64, S.tot1)
            xb[ind:ind + S.tot2] = dxs.download(np.float64, S.tot2)
    return t


# ---- storage helpers of the 's' blocks (misc_solvers.c:412-632, 887-988) -------------------------------------------------
def _s_section(dims, mnl):
    return mnl + dims["l"] + sum(dims.get("q") or [])


def pack(x, y, dims, mnl=0, offsetx=0, offsety=0):
    """misc_solvers.c:412-468: y := x with the 's' blocks in packed storage (lower triangles by columns, off-diagonal
    entries scaled by sqrt(2))."""
    xb, yb = _buf(x)[0], _buf(y)[0]
    nlq = _s_section(dims, mnl)
    yb[offsety:offsety + nlq] = xb[offsetx:offsetx + nlq]
    S = _s_blocks(dims)
    if S is None:
        return
    _lib.require_device()
    dfull, dpk = _up(xb[offsetx + nlq:offsetx + nlq + S.tot2]), _scratch(S.totp)
    raise_for(lib().kvx_nts_pack_dev(S.ns, S.d2.ptr, S.d1.ptr, S.dp.ptr, dfull.ptr, dpk.ptr, 0))
    _sync()
    yb[offsety + nlq:offsety + nlq + S.totp] = dpk.download(np.float64, S.totp)


def pack2(x, dims, mnl=0):
    """misc_solvers.c:476-544: in-place pack of every column of the matrix x (the diagonal entries are copied as they are)."""
    xb, size = _buf(x)
    S = _s_blocks(dims)
    if S is None:
        return
    _lib.require_device()
    nlq = _s_section(dims, mnl)
    X = xb.reshape(size, order="F")
    for c in range(size[1]):
        col = X[:, c]
        dfull, dpk = _up(col[nlq:nlq + S.tot2]), _scratch(S.totp)
        raise_for(lib().kvx_nts_pack_dev(S.ns, S.d2.ptr, S.d1.ptr, S.dp.ptr, dfull.ptr, dpk.ptr, 2))
        _sync()
        col[nlq:nlq + S.totp] = dpk.download(np.float64, S.totp)


def unpack(x, y, dims, mnl=0, offsetx=0, offsety=0):
    """misc_solvers.c:552-608: y := x with the 's' blocks unpacked into the lower triangles (off-diagonal entries scaled by
    1/sqrt(2)); the strict upper triangles of y are not touched."""
    xb, yb = _buf(x)[0], _buf(y)[0]
    nlq = _s_section(dims, mnl)
    yb[offsety:offsety + nlq] = xb[offsetx:offsetx + nlq]
    S = _s_blocks(dims)
    if S is None:
        return
    _lib.require_device()
    dfull, dpk = _up(yb[offsety + nlq:offsety + nlq + S.tot2]), _up(xb[offsetx + nlq:offsetx + nlq + S.totp])
    raise_for(lib().kvx_nts_pack_dev(S.ns, S.d2.ptr, S.d1.ptr, S.dp.ptr, dfull.ptr, dpk.ptr, 1))
    _sync()
    yb[offsety + nlq:offsety + nlq + S.tot2] = dfull.download(np.float64, S.tot2)


def _tri(x, sdims, offset, mode):
    S = _SBlocks(sdims) if sdims and sum(sdims) else None
    if S is None:
        return
    _lib.require_device()
    xb = _buf(x)[0]
    d = _up(xb[offset:offset + S.tot2])
    raise_for(lib().kvx_nts_tri_dev(S.ns, S.d2.ptr, S.d1.ptr, d.ptr, mode))
    _sync()
    xb[offset:offset + S.tot2] = d.download(np.float64, S.tot2)


def symm(x, n, offset=0):
    """misc_solvers.c:610-632: fills in the upper triangle of the n x n symmetric matrix stored at x[offset:] ('L' storage)."""
    _tri(x, [int(n)], offset, 0)


def trisc(x, dims, offset=0):
    """misc_solvers.c:887-938: upper triangles of the 's' blocks := 0, strict lower triangles scaled by 2."""
    _tri(x, [int(m) for m in (dims.get("s") or [])], offset + dims["l"] + sum(dims.get("q") or []), 1)


def triusc(x, dims, offset=0):
    """misc_solvers.c:940-988: strict lower triangles of the 's' blocks scaled by 1/2."""
    _tri(x, [int(m) for m in (dims.get("s") or [])], offset + dims["l"] + sum(dims.get("q") or []), 2)


class _Chol2Device:
    """Device side of one `kkt_chol2` factory: the stacked constraint matrix J = [Df; G] (plus the rows of A when S turned out
    singular), its row weights, the assembly plan of S = J' diag(w^2) J + H and the KKT object of kvxopt_amd.lp that owns
    the Cholesky factors.  Built at the first factor() call -- the call that fixes the sparsity patterns in the reference
    too (misc.py:1405-1432) -- and refilled with new values afterwards."""

    def __init__(self, G, A, mnl, Df, H, with_A_rows):
        from . import lp
        gm, n, gcp, gri, gv = base._as_ccs(G)
        self.n, self.ml, self.mnl = n, gm, mnl
        self.p = A.size[0]
        am, an, acp, ari, av = base._as_ccs(A)
        blocks = []                                   # (row offset, colptr, rowind, value count) of every block of J
        if mnl:
            dm, dn, dcp, dri, dv = base._as_ccs(Df)
            if (dm, dn) != (mnl, n):
                raise TypeError("Df must be an mnl x n sparse matrix")
            blocks.append((0, dcp, dri, dv.size))
        blocks.append((mnl, gcp, gri, gv.size))
        if with_A_rows:
            blocks.append((mnl + gm, acp, ari, av.size))
        rows = np.concatenate([ri + off for off, cp, ri, cnt in blocks]) if blocks else np.zeros(0, np.int64)
        cols = np.concatenate([np.repeat(np.arange(n, dtype=np.int64), np.diff(cp)) for off, cp, ri, cnt in blocks])
        self._perm = np.lexsort((rows, cols))         # J's CCS order; also the gather of its value array from the blocks'
        self.Jp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(cols, minlength=n), out=self.Jp[1:])
        self.Ji = np.ascontiguousarray(rows[self._perm])
        self.mj = mnl + gm + (am if with_A_rows else 0)
        self.with_A_rows = with_A_rows
        self._gv, self._av = gv, av
        Hp = Hi = Hx = None
        self._hkeep = None
        if H is not None:
            hm, hn, hcp, hri, hv = base._as_ccs(H)
            hcol = np.repeat(np.arange(hn, dtype=np.int64), np.diff(hcp))
            self._hkeep = hri >= hcol                 # S is stored by its lower triangle: that part of H is added
            Hp = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(np.bincount(hcol[self._hkeep], minlength=n), out=Hp[1:])
            Hi, Hx = np.ascontiguousarray(hri[self._hkeep]), np.ascontiguousarray(hv[self._hkeep])
            self._hpat = (hcp.copy(), hri.copy())
        Jx = self._values(Df)
        diag_s = (self.p > 0 and H is None and mnl == 0 and not with_A_rows and
                  (self.Ji.size == 0 or np.bincount(self.Ji, minlength=max(self.mj, 1)).max() <= 1))
        if self.p == 0:
            self.kkt = lp.KKTChol2Dev(self.mj, n, self.Jp, self.Ji, Jx, None, Hp, Hi, Hx)
            self.J, self.chol = self.kkt.G, self.kkt
        elif diag_s:
            self.kkt = lp.KKTDiagEqDev(self.mj, n, self.Jp, self.Ji, Jx, self.p, acp, ari, av)
            self.J, self.chol = None, None
        else:
            self.kkt = lp.KKTGenEqDev(self.mj, n, self.Jp, self.Ji, Jx, self.p, acp, ari, av, None, Hp, Hi, Hx)
            self.J, self.chol = self.kkt.G, self.kkt.S
        self.Adev = lp.SpMatDev(am, an, acp, ari, av) if (with_A_rows and self.p) else None
        self.w = lp.DVec(max(self.mj, 1))
        self.x, self.z = lp.DVec(max(n, 1)), lp.DVec(max(self.mj, 1))
        self.y = lp.DVec(max(self.p, 1))
        self._first = True

    def _values(self, Df):
        parts = []
        if self.mnl:
            parts.append(base._as_ccs(Df)[4])
        parts.append(self._gv)
        if self.with_A_rows:
            parts.append(self._av)
        return np.ascontiguousarray(np.concatenate(parts)[self._perm]) if parts else np.zeros(0)

    def refactor(self, W, H, Df):
        """New weights (and Df / H values) on the fixed patterns; numeric refactorisation.  ArithmeticError if S (or K) is
        not positive definite."""
        wl = [np.asarray(_buf(W["dnli"])[0], dtype=np.float64)[:self.mnl]] if self.mnl else []
        wl.append(np.asarray(_buf(W["di"])[0], dtype=np.float64)[:self.ml])
        if self.with_A_rows:
            wl.append(np.ones(self.p))
        w = np.concatenate(wl) if self.mj else np.zeros(1)
        if w.size != max(self.mj, 1) and self.mj:
            raise TypeError("W['di'] does not match the number of inequality rows")
        self.w.set(w)
        if self.mnl and not self._first and self.J is not None:
            self.J.set_values(self._values(Df))
        if H is not None:
            if self._hkeep is None:
                raise ValueError("kkt_chol2: H appeared after the first call fixed the pattern of S without it")
            hm, hn, hcp, hri, hv = base._as_ccs(H)
            if hri.size != self._hpat[1].size or not np.array_equal(hcp, self._hpat[0]) or not np.array_equal(hri, self._hpat[1]):
                raise ValueError("kkt_chol2: H must keep the sparsity pattern of the first call")
            if not self._first and self.chol is not None:
                self.chol.set_hessian(np.ascontiguousarray(hv[self._hkeep]))
        self._first = False
        self.kkt.factor(self.w, sync=True)

    def solve(self, x, y, z):
        """(x, y, z) := (ux, uy, W uz) of the KKT system, host vectors in and out; the arithmetic stays on the device."""
        xb, yb, zb = _buf(x)[0], _buf(y)[0], _buf(z)[0]
        n, p, m = self.n, self.p, self.mnl + self.ml
        self.x.set(xb[:n])
        zz = np.zeros(max(self.mj, 1))
        zz[:m] = zb[:m]
        self.z.set(zz)                                   # (the rows of A appended to J carry no right-hand side)
        if p:
            self.y.set(yb[:p])
            if self.with_A_rows:
                self.Adev.gemv(self.y, self.x, trans="T", alpha=1.0, beta=1.0)   # singular S: bx + A' by (misc.py:1525-1526)
            self.kkt.solve(self.x, self.y, self.z)
            yb[:p] = self.y.get()[:p]
        else:
            self.kkt.solve(self.x, self.z)
        xb[:n] = self.x.get()[:n]
        zb[:m] = self.z.get()[:m]


def _full_pattern(M, lower=False):
    """A dense `matrix` as an spmatrix with every entry (or every entry of the lower triangle) stored; anything else unchanged."""
    if not isinstance(M, matrix):
        return M
    r, c = M.size
    D = np.asarray(M.a, dtype=np.float64).reshape(r, c)
    if lower:
        I, J = np.nonzero(np.tril(np.ones((r, c), dtype=bool)).T)       # column by column
        I, J = J, I
        return spmatrix(D[I, J], I, J, (r, c))
    return spmatrix.from_ccs(r, c, np.arange(c + 1, dtype=np.int64) * r, np.tile(np.arange(r, dtype=np.int64), c),
                             np.ascontiguousarray(D.T).reshape(-1))


def kkt_chol2(G, dims, A, mnl=0):
    """KKT solver factory of the reference for sparse G (misc.py:1352-1567): returns factor(W, H=None, Df=None), which
    returns solve(x, y, z) overwriting the right-hand side (bx, by, bz) with (ux, uy, W uz) of

        [ H   A'  J' ] [ux]   [bx]
        [ A   0   0  ] [uy] = [by],      J = [Df; G]  (Df: the mnl x n Jacobian block of cvxprog, absent for cone LPs / QPs).
        [ J   0 -W'W ] [uz]   [bz]

    Same contract as the reference: the first factor() call fixes the sparsity patterns (of G, Df, H -- hence of
    S = J' W^-1 W^-T J + H) and analyses S once, later calls refactor numerically (misc.py:1405-1462); if the first S is not
    positive definite, A'A is added to it from then on and the solves compensate (misc.py:1433-1447, 1525-1526); a failing
    factorisation raises ArithmeticError.  Not the reference's program: everything between the host vectors of the caller
    and the result runs in HBM on kvxopt_amd.lp's device classes -- S assembled by one gather kernel on a product map,
    K = A S^-1 A' as a fixed-pattern assembly when S is diagonal and as a dense matrix otherwise (the reference rebuilds
    Asct = L^-1 P A' by sparse triangular solves and re-analyses K at every call, misc.py:1483-1487), both Cholesky factors
    resident between factor() and solve().  Dense G, A, H or Df (the reference's LAPACK branches) are taken as matrices with
    every entry stored and run through the same kernels."""
    if dims.get("q") or dims.get("s"):
        raise ValueError("kktsolver option 'kkt_chol2' is implemented only for problems with no "
                         "second-order or semidefinite cone constraints")
    p, n = A.size
    # dense operands (the reference's LAPACK branches, misc.py:1401-1404, 1428-1429, 1464-1481): every entry stored, the same
    # kernels on a full pattern -- S is then one dense front
    G = _full_pattern(G)
    A = _full_pattern(A)
    state = {"dev": None}

    def factor(W, H=None, Df=None):
        H = _full_pattern(H, lower=True)          # (the lower triangle is what the factorisation reads)
        if mnl:
            Df = _full_pattern(Df)
            if not hasattr(Df, "CCS"):
                raise TypeError("Df must be a 'd' matrix or spmatrix")
        _lib.require_device()
        if state["dev"] is None:
            dev = _Chol2Device(G, A, mnl, Df, H, with_A_rows=False)
            try:
                dev.refactor(W, H, Df)
            except ArithmeticError:
                if p == 0:
                    raise
                dev = _Chol2Device(G, A, mnl, Df, H, with_A_rows=True)      # S singular: S + A'A from now on
                dev.refactor(W, H, Df)
            state["dev"] = dev
        else:
            state["dev"].refactor(W, H, Df)
        return state["dev"].solve

    return factor
