"""MPS front-end (kvxopt_amd/mps.py) against the reference's own fixture and reader (modeling.op.fromfile,
src/python/modeling.py:2760-3060; reference test tests/test_modeling.py:59-63).  Golden: tests/golden/g7_boeing2.json,
written by tests/golden/make_goldens.py from the reference itself (pure reference: dense LAPACK kkt solver).
CPU tests check the reader and the standard-form conversion (independent LP solver: SciPy HiGHS); the GPU test runs the
device-resident interior-point loop on the converted problem."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from kvxopt_amd import mps

def _ln(t="", n1="", n2="", v1=None, n3="", v2=None):
    """One fixed-format data line: type 1:3, names 4:12 / 14:22 / 39:47, numbers 24:36 / 49:61."""
    f1 = ("%12g" % v1) if v1 is not None else " " * 12
    f2 = ("%12g" % v2) if v2 is not None else ""
    return (" " + t.ljust(2) + " " + n1.ljust(8) + "  " + n2.ljust(8) + "  " + f1 + "   " + n3.ljust(8) + "  " + f2).rstrip()


SMALL = "\n".join(
    ["NAME          SMALL", "ROWS"] +
    [_ln(t, n) for t, n in (("N", "COST"), ("N", "OTHER"), ("L", "LIM1"), ("G", "LIM2"), ("E", "MYEQN"), ("E", "RNGEQ"), ("G", "EMPTY"))] +
    ["COLUMNS",
     _ln("", "X1", "COST", 1.0, "LIM1", 1.0), _ln("", "X1", "LIM2", 1.0, "OTHER", 5.0),
     _ln("", "X2", "COST", 2.0, "LIM1", 1.0), _ln("", "X2", "MYEQN", -1.0),
     _ln("", "X3", "COST", -1.0, "MYEQN", 1.0),
     _ln("", "X4", "COST", 1.5, "RNGEQ", 1.0), _ln("", "X4", "LIM2", 1.0),
     _ln("", "X5", "COST", -0.5, "RNGEQ", 1.0),
     _ln("", "X6", "COST", 1.0, "LIM1", 1.0),
     "RHS",
     _ln("", "RHS", "COST", -3.0, "LIM1", 4.0), _ln("", "RHS", "LIM2", 1.0, "MYEQN", 7.0), _ln("", "RHS", "RNGEQ", 2.0),
     _ln("", "RHS2", "LIM1", 99.0),
     "RANGES",
     _ln("", "RNG", "LIM1", 2.5, "RNGEQ", -1.5),
     "BOUNDS",
     _ln("UP", "BND", "X1", 4.0), _ln("LO", "BND", "X2", -1.0), _ln("UP", "BND", "X2", 1.0), _ln("MI", "BND", "X3"),
     _ln("UP", "BND", "X3", 9.0), _ln("FR", "BND", "X4"), _ln("FX", "BND", "X6", 0.5), _ln("PL", "BND", "X5"),
     "ENDATA", ""])


def _sp(A):
    return sp.csc_matrix((A.values, A.rowind, A.colptr), shape=A.size)


def test_reader_follows_the_reference_semantics(tmp_path):
    fn = tmp_path / "small.mps"
    fn.write_text(SMALL)
    P = mps.read_mps(str(fn))
    assert P.name == "SMALL" and P.objective_row == "COST" and P.c0 == 3.0          # RHS of the objective row: minus the constant
    assert P.rows == ["LIM1", "LIM2", "MYEQN", "RNGEQ", "EMPTY"] and P.cols == ["X1", "X2", "X3", "X4", "X5", "X6"]
    assert P.rowtype == {"LIM1": "L", "LIM2": "G", "MYEQN": "E", "RNGEQ": "E", "EMPTY": "G"}
    assert P.rhs["LIM1"] == 4.0                                                      # second RHS vector ignored
    assert P.ranges == {"LIM1": 2.5, "RNGEQ": -1.5}
    assert P.bounds == {"X1": [0.0, 4.0], "X2": [-1.0, 1.0], "X3": [None, 9.0], "X4": [None, None], "X5": [0.0, None], "X6": [0.5, 0.5]}
    assert "X1" not in P.coeff.get("OTHER", {})                                      # later N rows are ignored


def test_standard_form_is_equivalent(tmp_path):
    """Solve the converted problem and the natural-form one with HiGHS: same optimum, recovered x feasible."""
    fn = tmp_path / "small.mps"
    fn.write_text(SMALL)
    P = mps.read_mps(str(fn))
    c, A, b, c0, recover = mps.standard_form(P)
    r = linprog(c, A_eq=_sp(A), b_eq=b, bounds=(0, None), method="highs")
    assert r.status == 0
    x = recover(r.x)
    # natural form by hand: ranges as modeling.py:2962-2995
    cols = P.cols
    cn = np.array([P.coeff["COST"].get(k, 0.0) for k in cols])
    def row(label):
        return np.array([P.coeff[label].get(k, 0.0) for k in cols])
    A_ub = np.array([row("LIM1"), -row("LIM1"), -row("LIM2"), row("RNGEQ"), -row("RNGEQ")])
    b_ub = np.array([4.0, -(4.0 - 2.5), -1.0, 2.0, -(2.0 - 1.5)])
    rn = linprog(cn, A_ub=A_ub, b_ub=b_ub, A_eq=row("MYEQN")[None, :], b_eq=[7.0],
                 bounds=[(0, 4), (-1, 1), (None, 9), (None, None), (0, None), (0.5, 0.5)], method="highs")
    assert rn.status == 0
    assert abs((r.fun + c0) - (rn.fun + 3.0)) < 1e-9
    xv = np.array([x[k] for k in cols])
    assert np.all(A_ub @ xv <= b_ub + 1e-9) and abs(row("MYEQN") @ xv - 7.0) < 1e-9 and x["X6"] == 0.5


def test_boeing2_reader_and_standard_form_against_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g7_boeing2.json")))
    P = mps.read_mps(os.path.join(golden_dir, "boeing2.mps"))
    assert P.name == "BOEING2" and len(P.cols) == g["n_variables"] == 143 and len(P.rows) == 166
    x = g["x"]
    obj = sum(a * x[k] for k, a in P.coeff[P.objective_row].items()) + P.c0
    assert abs(obj - g["objective"]) < 1e-9 * abs(g["objective"])                    # same objective function as the reference parsed
    for label in P.rows:                                                             # the reference's solution satisfies the rows as parsed here
        f = sum(a * x[k] for k, a in P.coeff[label].items()) - P.rhs.get(label, 0.0)
        t, rng = P.rowtype[label], P.ranges.get(label)
        lo, hi = {"L": (-abs(rng) if rng is not None else -np.inf, 0.0), "G": (0.0, abs(rng) if rng is not None else np.inf),
                  "E": (0.0, 0.0) if not rng else ((0.0, rng) if rng > 0 else (rng, 0.0))}[t]
        assert lo - 1e-5 <= f <= hi + 1e-5, label
    for k, (lo, up) in P.bounds.items():
        assert (lo is None or x[k] >= lo - 1e-6) and (up is None or x[k] <= up + 1e-6)
    c, A, b, c0, recover = mps.standard_form(P)
    r = linprog(c, A_eq=_sp(A), b_eq=b, bounds=(0, None), method="highs")
    assert r.status == 0 and abs(r.fun + c0 - g["objective"]) < 1e-5 * abs(g["objective"])
    assert np.linalg.matrix_rank(_sp(A).toarray()) == A.size[0]                      # K = A S^-1 A' is nonsingular


def test_boeing2_natural_form_has_the_reference_shape(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g7_boeing2.json")))
    P = mps.read_mps(os.path.join(golden_dir, "boeing2.mps"))
    c, G, h, A, b, c0 = mps.natural_form(P)
    assert c.size == g["n_variables"] and A.size[0] == g["n_equalities"]
    assert G.size[0] == g["n_inequalities"]            # after the reference dropped its 26 constraints without variables
    r = linprog(c, A_ub=_sp(G), b_ub=h, A_eq=_sp(A), b_eq=b, bounds=(None, None), method="highs")
    assert r.status == 0 and abs(r.fun + c0 - g["objective"]) < 1e-5 * abs(g["objective"])


@pytest.mark.gpu
def test_boeing2_natural_form_on_the_gpu(golden_dir):
    """The reference's own formulation (general G + 4 equality rows) on the device: lp.KKTGenEqDev."""
    from kvxopt_amd import _lib
    _lib.require_device()
    g = json.load(open(os.path.join(golden_dir, "g7_boeing2.json")))
    out = mps.solve(os.path.join(golden_dir, "boeing2.mps"), form="natural")
    assert out["status"] == "optimal", out["iterations"]
    assert abs(out["objective"] - g["objective"]) <= 1e-5 * abs(g["objective"])
    xr = g["x"]
    assert max(abs(out["x"][k] - xr[k]) for k in xr) <= 1e-3 * max(1.0, max(abs(v) for v in xr.values()))   # the optimum is not unique to more


@pytest.mark.gpu
def test_boeing2_on_the_gpu_matches_the_reference(golden_dir):
    """tests/test_modeling.py:59-63 asserts status == 'optimal'; the golden adds the reference's objective value."""
    from kvxopt_amd import _lib
    _lib.require_device()
    g = json.load(open(os.path.join(golden_dir, "g7_boeing2.json")))
    out = mps.solve(os.path.join(golden_dir, "boeing2.mps"))
    assert out["status"] == "optimal", (out["iterations"], {k: out["sol"][k] for k in ("gap", "relative gap", "primal infeasibility", "dual infeasibility")})
    assert abs(out["objective"] - g["objective"]) <= 1e-5 * abs(g["objective"])      # both stop at reltol 1e-6
    P = out["problem"]
    x = out["x"]
    for label in P.rows:
        f = sum(a * x[k] for k, a in P.coeff[label].items()) - P.rhs.get(label, 0.0)
        t, rng = P.rowtype[label], P.ranges.get(label)
        lo, hi = {"L": (-abs(rng) if rng is not None else -np.inf, 0.0), "G": (0.0, abs(rng) if rng is not None else np.inf),
                  "E": (0.0, 0.0) if not rng else ((0.0, rng) if rng > 0 else (rng, 0.0))}[t]
        scale = 1.0 + abs(P.rhs.get(label, 0.0))
        assert lo - 1e-5 * scale <= f <= hi + 1e-5 * scale, label
