"""TEST / BENCH INFRASTRUCTURE ONLY -- CPU restatement of the reference's conelp main loop for the orthant cone in
inequality form (dims = {'l': ml}, no equality rows): `coneprog.conelp` lines 420-1436 specialised as SURVEY.md
Appendix A spells out, with the KKT solver of `misc.kkt_chol2` for p = 0 (misc.py:1352-1567): S = G' diag(di^2) G assembled
on a fixed pattern and factored by a CPU Cholesky on a fixed symbolic analysis at every iteration.

Used as the CPU baseline beside the device-resident `kvxopt_amd.lp.conelp` (bench.py, `ipm.cpu_baseline`) and pinned on
the reference's own traces (golden G4: same iteration counts, same solution) in tests/test_oracle.py.  The Cholesky is
`oracle.kvx_oracle.OracleSupernodal` (host supernodal, all cores) or `OracleChol` (simplicial, 1 core).  Nothing in
kvxopt_amd/ imports this module.
"""
import math
import time

import numpy as np

from . import kvx_oracle as orc

EXPON, STEP = 3, 0.99                        # coneprog.py:423-424


def conelp_l(c, ml, n, Gp, Gi, Gx, h, structure=None, threads=None, maxiters=100, abstol=1e-7, reltol=1e-6, feastol=1e-7):
    """minimise c'x  s.t.  G x + s = h, s >= 0 (G: ml x n CCS).  `structure`: (perm, super, front rowptr, front rowidx,
    parent, Sp, Si) of an analysis of tril(G'G) to factor on (then the supernodal host Cholesky runs on it); None: the
    simplicial oracle with its own ordering.  Returns the reference's result dict entries used by the tests plus timings."""
    import scipy.sparse as sp
    c = np.asarray(c, float).reshape(-1); h = np.asarray(h, float).reshape(-1)
    G = sp.csc_matrix((Gx, Gi, Gp), shape=(ml, n))
    GT = G.T.tocsc()
    if structure is None:
        Spat = sp.tril((abs(GT) @ abs(G)).tocsc()).tocsc(); Spat.sort_indices()
        Sp, Si = Spat.indptr.astype(np.int64), Spat.indices.astype(np.int64)
        chol = orc.OracleChol(n, Sp, Si, "L")
    else:
        perm, sup, rp, ri, parent, Sp, Si = structure
        chol = orc.OracleSupernodal(n, Sp, Si, (perm, sup, rp, ri, parent), threads)
    Gp64, Gi64, Gx64 = np.asarray(Gp, np.int64), np.asarray(Gi, np.int64), np.asarray(Gx, float)
    stats = {"factor_s": 0.0, "solve_s": 0.0, "assemble_s": 0.0, "factorizations": 0}

    def kkt(d, di):
        """misc.kkt_chol2, p = 0: factor S = G' di^2 G; return f(x, z) that overwrites them with ux, W uz."""
        t0 = time.perf_counter()
        _, Sx = orc.atda(ml, n, Gp64, Gi64, Gx64, di, Sp, Si)              # misc.py:1418-1462 on the fixed pattern
        t1 = time.perf_counter()
        chol.factorize(Sx)                                                  # ArithmeticError -> caller
        stats["assemble_s"] += t1 - t0; stats["factor_s"] += time.perf_counter() - t1; stats["factorizations"] += 1

        def f(x, z):
            t = time.perf_counter()
            z *= di                                                         # z := W^-T z               (misc.py:1513)
            x += GT @ (di * z)                                              # x := x + Gs' z           (misc.py:1524)
            xs = np.asfortranarray(x.reshape(n, 1)); chol.solve(xs); x[:] = xs[:, 0]   # (misc.py:1531-1532)
            z[:] = di * (G @ x) - z                                         # W uz := Gs x - z          (misc.py:1563)
            stats["solve_s"] += time.perf_counter() - t
        return f

    resx0, resz0 = max(1.0, math.sqrt(c @ c)), max(1.0, math.sqrt(h @ h))  # coneprog.py:649-651
    # starting point (coneprog.py:670-822)
    f = kkt(np.ones(ml), np.ones(ml))
    x = np.zeros(n); z0 = h.copy(); f(x, z0); s = -z0
    xd = -c.copy(); z = np.zeros(ml); f(xd, z)
    ts, tz = float(np.max(-s)), float(np.max(-z))
    nrms, nrmz = math.sqrt(s @ s), math.sqrt(z @ z)
    if ts >= -1e-8 * max(nrms, 1.0):
        s += 1.0 + ts
    if tz >= -1e-8 * max(nrmz, 1.0):
        z += 1.0 + tz
    tau = kappa = 1.0
    gap = float(s @ z)
    d = di = lmbda = None
    dg = dgi = lmbda_g = 1.0
    t_loop = time.perf_counter()
    status, iters = "unknown", 0
    for iters in range(maxiters + 1):
        hrx = -(GT @ z); rx = hrx - c * tau; resx = math.sqrt(rx @ rx) / tau            # coneprog.py:861-896
        hrz = s + G @ x; rz = hrz - h * tau; resz = math.sqrt(rz @ rz) / tau
        cx, hz = float(c @ x), float(h @ z)
        rt = kappa + cx + hz
        pcost, dcost = cx / tau, -hz / tau                                               # :898-923
        if pcost < 0.0:
            relgap = gap / -pcost
        elif dcost > 0.0:
            relgap = gap / dcost
        else:
            relgap = None
        pres, dres = resz / resz0, resx / resx0
        pinfres = math.sqrt(hrx @ hrx) / resx0 / (-hz) if hz < 0.0 else None
        dinfres = math.sqrt(hrz @ hrz) / resz0 / (-cx) if cx < 0.0 else None
        if (pres <= feastol and dres <= feastol and (gap <= abstol or (relgap is not None and relgap <= reltol))) or iters == maxiters:
            status = "unknown" if iters == maxiters else "optimal"                      # :937-960
            x, s, z = x / tau, s / tau, z / tau
            break
        if pinfres is not None and pinfres <= feastol:
            status = "primal infeasible"; z = z / (-hz); x = s = None
            break
        if dinfres is not None and dinfres <= feastol:
            status = "dual infeasible"; x = x / (-cx); s = s / (-cx); z = None
            break
        if iters == 0:                                                                   # :1031-1043
            d = np.sqrt(s / z); di = 1.0 / d; lmbda = np.sqrt(s * z)
            dg = math.sqrt(kappa / tau); dgi = math.sqrt(tau / kappa); lmbda_g = math.sqrt(tau * kappa)
        lmbdasq = lmbda * lmbda; lmbdasq_g = lmbda_g * lmbda_g                          # :1046-1047
        try:
            f3 = kkt(d, di)                                                              # :1067
        except ArithmeticError:
            status = "unknown"; x, s, z = x / tau, s / tau, z / tau
            break
        x1 = -c.copy(); z1 = h.copy(); f3(x1, z1); x1 *= dgi; z1 *= dgi                # :1071-1077
        th = di * h                                                                      # :1126-1128
        mu = (float(lmbda @ lmbda) + lmbda_g * lmbda_g) / (1 + ml); sigma = 0.0         # :1248-1249
        dsdz_save = dkdt_save = None
        for i in (0, 1):
            ds = lmbdasq.copy(); dkappa = lmbdasq_g                                     # :1273-1292
            if i == 1:
                ds += dsdz_save - sigma * mu; dkappa += dkdt_save - sigma * mu
            dx = (1.0 - sigma) * rx; dz = (1.0 - sigma) * rz; dtau = (1.0 - sigma) * rt
            # f6_no_ir (:1130-1195)
            ds = -ds / lmbda
            dz = -(dz + d * ds)
            f3(dx, dz)
            dkappa = -dkappa / lmbda_g
            dtau = dtau + dkappa / dgi
            dtau = dgi * (dtau + float(c @ dx) + float(th @ dz)) / (1.0 + float(z1 @ z1))
            dx += dtau * x1; dz += dtau * z1
            ds = ds - dz; dkappa = dkappa - dtau
            if i == 0:
                dsdz_save = ds * dz; dkdt_save = dtau * dkappa                           # :1303-1306
            ds = ds / lmbda; dz = dz / lmbda                                             # :1314-1315
            tsx, tzx = float(np.max(-ds)), float(np.max(-dz))
            tt, tk = -dtau / lmbda_g, -dkappa / lmbda_g
            t = max(0.0, tsx, tzx, tt, tk)
            if t == 0.0:
                step = 1.0
            else:
                step = min(1.0, 1.0 / t) if i == 0 else min(1.0, STEP / t)
            if i == 0:
                sigma = (1.0 - step) ** EXPON
        x += step * dx                                                                   # :1336-1436
        ds = (1.0 + step * ds) * lmbda; dz = (1.0 + step * dz) * lmbda
        ds = np.sqrt(ds); dz = np.sqrt(dz)                                               # update_scaling (misc.py:444-464)
        d = d * ds / dz; di = 1.0 / d; lmbda = ds * dz
        dg *= math.sqrt(1.0 - step * tk) / math.sqrt(1.0 - step * tt); dgi = 1.0 / dg
        lmbda_g *= math.sqrt(1.0 - step * tt) * math.sqrt(1.0 - step * tk)
        s = d * lmbda; z = di * lmbda
        kappa = lmbda_g / dgi; tau = lmbda_g * dgi
        gap = float(lmbda @ lmbda) / (tau * tau)
    loop_s = time.perf_counter() - t_loop
    return {"status": status, "x": x, "s": s, "z": z, "iterations": iters, "loop seconds": loop_s,
            "primal objective": (float(c @ x) if x is not None and status != "dual infeasible" else None), **stats}
