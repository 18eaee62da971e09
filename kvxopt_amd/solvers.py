"""`kvxopt.solvers` names for the device-resident interior-point drivers of kvxopt_amd.lp (orthant cone only):

    conelp(c, G, h, dims=None, A=None, b=None, primalstart=None, dualstart=None)     coneprog.py:420
    coneqp(P, q, G, h, dims=None, A=None, b=None, initvals=None)                     coneprog.py:1440
    lp(c, G, h, A=None, b=None, primalstart=None, dualstart=None)                    coneprog.py:2551 (-> conelp)
    qp(P, q, G, h, A=None, b=None, initvals=None)                                    coneprog.py:4120 (-> coneqp)
    options                                                                          the module-level dict of the reference

Like the reference, algorithm parameters come from `solvers.options` ('maxiters', 'abstol', 'reltol', 'feastol',
'refinement', 'show_progress'); a keyword `options=` overrides it per call.  `conelp / lp / coneqp / qp (..., kktsolver=f)`
take the reference's plug-in, a function `W -> g(x, y, z)` (coneprog.py:323-344, 1969-1981; host round trips per factorisation
and solve, lp.KKTUserHost); the named solvers ('ldl', 'ldl2', 'qr', 'chol', 'chol2') and `solver=` (external codes) are not part of
this path and raise.
"""
from . import lp as _lp

options = {}


def _opts(kw):
    o = {"show_progress": True}                          # the reference's default (coneprog.py:456, 1803)
    o.update(options)
    o.update(kw.pop("options", None) or {})
    if kw.pop("solver", None) is not None:
        raise NotImplementedError("kvxopt_amd.solvers runs misc.kkt_chol2 on the GPU; 'solver' is not selectable")
    if kw:
        raise TypeError("unexpected arguments: %s" % ", ".join(sorted(kw)))
    return o


def _kkt(kw):
    k = kw.pop("kktsolver", None)
    if k is not None and not callable(k):
        raise NotImplementedError("kvxopt_amd.solvers runs misc.kkt_chol2 on the GPU; the named KKT solver '%s' is not selectable "
                                  "(a function W -> f(x, y, z) is)" % k)
    return k


def conelp(c, G, h, dims=None, A=None, b=None, primalstart=None, dualstart=None, **kw):
    k = _kkt(kw)
    return _lp.conelp(c, G, h, dims=dims, A=A, b=b, options=_opts(kw), primalstart=primalstart, dualstart=dualstart, kktsolver=k)


def coneqp(P, q, G, h, dims=None, A=None, b=None, initvals=None, **kw):
    k = _kkt(kw)
    if dims is not None and (dims.get("q") or dims.get("s")):
        raise NotImplementedError("only the orthant cone runs on the GPU")
    return _lp.coneqp(P, q, G, h, _opts(kw), None, A=A, b=b, initvals=initvals, kktsolver=k)


def lp(c, G, h, A=None, b=None, primalstart=None, dualstart=None, **kw):
    """solvers.lp (coneprog.py:2551-2790): conelp on the orthant, result keys as the reference returns them."""
    sol = conelp(c, G, h, None, A, b, primalstart, dualstart, **kw)
    return sol


def qp(P, q, G, h, A=None, b=None, initvals=None, **kw):
    """solvers.qp (coneprog.py:4120-4330): coneqp on the orthant."""
    return coneqp(P, q, G, h, None, A, b, initvals, **kw)
