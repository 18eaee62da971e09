"""Fixed-format MPS front-end for the LP path ("front-ends and formats either side of the path").

`read_mps` follows the reference's reader `modeling.op.fromfile` (src/python/modeling.py:2760-3060) field by field:
fixed columns (type 1:3, names 4:12 / 14:22 / 39:47, numbers 24:36 / 49:61), the first N row is the objective (its RHS
entry is minus the constant term, :2862-2863), only the first RHS / RANGES / BOUNDS vector is read, bound types
LO UP FX FR MI PL (:2917-2960), range semantics :2962-2995 (L: -|R| <= f <= 0, G: 0 <= f <= |R|, E: 0 <= f <= R or
R <= f <= 0 with f = a'x - rhs), rows without variables are dropped after a consistency check (:3010-3025).

The reference hands the problem to conelp in inequality form with equality rows (general G and A together), a branch
the device-resident driver does not have.  `standard_form` therefore rewrites it as  min c'y  s.t.  A y = b, y >= 0
(shifted / reflected / split variables, one slack per inequality row, one extra row per finite two-sided bound or
range), which is exactly BASELINE configs[3]'s shape: G = -I, so S is diagonal and K = A S^-1 A' is refactored on the
GPU at every iteration (`lp.KKTDiagEqDev`, misc.py:1483-1487, 1545).  `solve` runs `lp.conelp` on it and maps the solution back.
"""
import numpy as np

from . import lp as _lp
from .base import spmatrix


class MPSProblem:
    """minimise  c'x + c0  subject to the rows (type, coefficients, rhs, range) and the variable bounds."""

    def __init__(self):
        self.name = ""
        self.objective_row = None
        self.rows = []            # constraint row labels in file order
        self.rowtype = {}         # label -> 'E' | 'L' | 'G'
        self.cols = []            # column labels in order of first appearance
        self.coeff = {}           # row label -> {column label: value}   (objective row included)
        self.rhs = {}             # row label -> value
        self.ranges = {}          # row label -> value
        self.bounds = {}          # column label -> [lower or None, upper or None]

    @property
    def c0(self):
        return -self.rhs.get(self.objective_row, 0.0)


def _num(s):
    return float(s.strip())


def read_mps(filename):
    P = MPSProblem()
    with open(filename, "r") as f:
        lines = f.read().split("\n")
    it = iter(lines)

    def advance_to(tag, msg):
        for s in it:
            if s[:len(tag)] == tag:
                return s
        raise SyntaxError(msg)

    s = advance_to("NAME", "EOF reached before 'NAME' section was found")
    P.name = s[14:22].strip()
    advance_to("ROWS", "EOF reached before 'ROWS' section was found")
    section = "ROWS"
    rhslabel = rangeslabel = boundslabel = None
    collabel = None
    seen_endata = False
    extra_n = set()           # further 'N' rows: declared, ignored
    for s in it:
        if len(s.strip()) == 0 or s[0] == "*":
            continue
        head = s[:7]
        if head.startswith("COLUMNS"):
            section = "COLUMNS"; continue
        if s[:3] == "RHS" and section in ("COLUMNS", "ROWS"):
            section = "RHS"; continue
        if s[:6] == "RANGES":
            section = "RANGES"; continue
        if s[:6] == "BOUNDS":
            section = "BOUNDS"; continue
        if s[:6] == "ENDATA":
            seen_endata = True
            break
        if section == "ROWS":
            t = s[1:3].strip()
            label = s[4:12].strip()
            if t in ("E", "L", "G"):
                P.rows.append(label); P.rowtype[label] = t; P.coeff[label] = {}
            elif t == "N":
                if P.objective_row is None:                 # first occurrence of 'N' counts (modeling.py:2809-2813)
                    P.objective_row = label; P.coeff[label] = {}
                else:
                    extra_n.add(label)
            else:
                raise ValueError("unknown row type '%s'" % t)
        elif section == "COLUMNS":
            if s[4:12].strip():
                collabel = s[4:12].strip()
            if collabel not in P.bounds:
                P.cols.append(collabel); P.bounds[collabel] = [0.0, None]
            for a, b in ((s[14:22], s[24:36]), (s[39:47], s[49:61])):
                label = a.strip()
                if not label:
                    continue
                if label in P.coeff:
                    P.coeff[label][collabel] = _num(b)
                elif label not in extra_n:
                    raise KeyError("no row label '%s'" % label)
        elif section == "RHS":
            if rhslabel is not None and rhslabel != s[4:12].strip():
                continue
            rhslabel = s[4:12].strip()
            for a, b in ((s[14:22], s[24:36]), (s[39:47], s[49:61])):
                label = a.strip()
                if label:
                    if label not in P.coeff:
                        raise KeyError("no row label '%s'" % label)
                    P.rhs[label] = _num(b)
        elif section == "RANGES":
            if rangeslabel is not None and rangeslabel != s[4:12].strip():
                continue
            rangeslabel = s[4:12].strip()
            for a, b in ((s[14:22], s[24:36]), (s[39:47], s[49:61])):
                label = a.strip()
                if label:
                    if label not in P.rowtype:
                        raise KeyError("no row label '%s'" % label)
                    P.ranges[label] = _num(b)
        elif section == "BOUNDS":
            if boundslabel is not None and boundslabel != s[4:12].strip():
                continue
            boundslabel = s[4:12].strip()
            col = s[14:22].strip()
            if col not in P.bounds:
                raise ValueError("unknown column label '%s'" % col)
            t = s[1:3].strip()
            bd = P.bounds[col]
            if t == "LO":
                bd[0] = _num(s[24:36])
            elif t == "UP":
                bd[1] = _num(s[24:36])
            elif t == "FX":
                bd[0] = bd[1] = _num(s[24:36])
            elif t == "FR":
                bd[0] = bd[1] = None
            elif t == "MI":
                bd[0] = None
            elif t == "PL":
                pass
            else:
                raise ValueError("unknown bound type '%s'" % t)
    if not seen_endata:
        raise SyntaxError("EOF reached before 'ENDATA' was found")
    if P.objective_row is None:
        raise SyntaxError("no objective ('N') row")
    return P


def standard_form(P):
    """min c'y + c0  s.t.  A y = b, y >= 0.  Returns (c, A as spmatrix, b, c0, recover) with recover(y) -> {column: x}."""
    ycols = 0
    xmap = {}                      # column -> (kind, index, shift): x = shift + y | shift - y | y[i] - y[i+1] | fixed
    extra_rows = []                # (index of y, rhs): y + t = rhs with a new slack t
    c0 = P.c0
    for col in P.cols:
        lo, up = P.bounds[col]
        if lo is not None and up is not None and lo == up:
            xmap[col] = ("fixed", -1, lo)
        elif lo is not None:
            xmap[col] = ("plus", ycols, lo)
            if up is not None:
                if up < lo:
                    raise ValueError("empty bounds for variable '%s'" % col)
                extra_rows.append((ycols, up - lo))
            ycols += 1
        elif up is not None:
            xmap[col] = ("minus", ycols, up)
            ycols += 1
        else:
            xmap[col] = ("free", ycols, 0.0)
            ycols += 2
    I, J, V, b = [], [], [], []

    def add_terms(r, coeffs):
        """a'x over the y variables in row r; returns the constant that moves to the right-hand side."""
        const = 0.0
        for col, a in coeffs.items():
            kind, j, sh = xmap[col]
            if kind == "fixed":
                const += a * sh
            elif kind == "plus":
                I.append(r); J.append(j); V.append(a); const += a * sh
            elif kind == "minus":
                I.append(r); J.append(j); V.append(-a); const += a * sh
            else:
                I.extend((r, r)); J.extend((j, j + 1)); V.extend((a, -a))
        return const

    r = 0
    nslack = 0
    slack_rows = []                # (row, sign, range or None)
    for label in P.rows:
        coeffs = P.coeff[label]
        t = P.rowtype[label]
        rhs = P.rhs.get(label, 0.0)
        rng = P.ranges.get(label)
        if not any(xmap[cn][0] != "fixed" for cn in coeffs):
            const = sum(a * xmap[cn][2] for cn, a in coeffs.items()) - rhs        # f = a'x - rhs with every variable fixed / absent
            lo_ok = {"E": const == 0.0 if not rng else (0 <= const <= rng if rng > 0 else rng <= const <= 0),
                     "L": const <= 0.0 and (rng is None or const >= -abs(rng)),
                     "G": const >= 0.0 and (rng is None or const <= abs(rng))}[t]
            if not lo_ok:
                raise ValueError("constraint '%s' has no variables and an inconsistent righthand side" % label)
            continue
        const = add_terms(r, coeffs)
        b.append(rhs - const)
        if t == "E" and not rng:
            pass
        elif t == "L" or (t == "E" and rng < 0):
            slack_rows.append((r, 1.0, abs(rng) if rng is not None else None))
        else:                                                  # 'G', or 'E' with a positive range
            slack_rows.append((r, -1.0, abs(rng) if rng is not None else None))
        r += 1
    nstruct = ycols
    for (row, sign, rng) in slack_rows:                        # a'x + sign * s = rhs, s >= 0 (and s <= |R|: s + t = |R|)
        sj = nstruct + nslack
        nslack += 1
        I.append(row); J.append(sj); V.append(sign)
        if rng is not None:
            extra_rows.append((sj, rng))
    ny = nstruct + nslack
    for (j, ub) in extra_rows:                                 # y_j + t = ub
        I.extend((r, r)); J.extend((j, ny)); V.extend((1.0, 1.0))
        b.append(ub)
        ny += 1
        r += 1
    c = np.zeros(ny)
    for col, a in P.coeff[P.objective_row].items():
        kind, j, sh = xmap[col]
        if kind == "fixed":
            c0 += a * sh
        elif kind == "plus":
            c[j] += a; c0 += a * sh
        elif kind == "minus":
            c[j] -= a; c0 += a * sh
        else:
            c[j] += a; c[j + 1] -= a
    A = spmatrix(V, I, J, (r, ny))

    def recover(y):
        x = {}
        for col in P.cols:
            kind, j, sh = xmap[col]
            x[col] = sh if kind == "fixed" else sh + y[j] if kind == "plus" else sh - y[j] if kind == "minus" else y[j] - y[j + 1]
        return x

    return c, A, np.array(b, dtype=np.float64), c0, recover


def natural_form(P):
    """The reference's own formulation (modeling.py:2962-3007, then op._inmatrixform): min c'x + c0 s.t. G x <= h, A x = b
    over the file's variables -- inequality rows, range rows and finite bounds in G, 'E' rows and FX bounds in A.
    Returns (c, G, h, A or None, b, c0)."""
    cols = {k: j for j, k in enumerate(P.cols)}
    n = len(P.cols)
    GI, GJ, GV, h, AI, AJ, AV, b = [], [], [], [], [], [], [], []

    def add_ineq(coeffs, sign, rhs):                      # sign * a'x <= rhs
        r = len(h)
        for k, a in coeffs.items():
            GI.append(r); GJ.append(cols[k]); GV.append(sign * a)
        h.append(rhs)

    def add_eq(coeffs, rhs):
        r = len(b)
        for k, a in coeffs.items():
            AI.append(r); AJ.append(cols[k]); AV.append(a)
        b.append(rhs)

    for label in P.rows:
        co, t, rhs, rng = P.coeff[label], P.rowtype[label], P.rhs.get(label, 0.0), P.ranges.get(label)
        if not co:
            continue                                       # rows without variables: checked by standard_form's rule, dropped
        if t == "L":
            add_ineq(co, 1.0, rhs)
            if rng is not None:
                add_ineq(co, -1.0, -(rhs - abs(rng)))
        elif t == "G":
            add_ineq(co, -1.0, -rhs)
            if rng is not None:
                add_ineq(co, 1.0, rhs + abs(rng))
        elif not rng:
            add_eq(co, rhs)
        elif rng > 0:
            add_ineq(co, -1.0, -rhs); add_ineq(co, 1.0, rhs + rng)
        else:
            add_ineq(co, 1.0, rhs); add_ineq(co, -1.0, -(rhs + rng))
    for k in P.cols:
        lo, up = P.bounds[k]
        if lo is not None and up is not None and lo == up:
            add_eq({k: 1.0}, lo)
            continue
        if lo is not None:
            add_ineq({k: 1.0}, -1.0, -lo)
        if up is not None:
            add_ineq({k: 1.0}, 1.0, up)
    c = np.zeros(n)
    for k, a in P.coeff[P.objective_row].items():
        c[cols[k]] = a
    G = spmatrix(GV, GI, GJ, (len(h), n))
    A = spmatrix(AV, AI, AJ, (len(b), n)) if b else None
    return c, G, np.array(h), A, np.array(b), P.c0


def solve(filename, options=None, form="standard"):
    """Read, convert and solve on the GPU.  Returns status / objective / x (by column label) / iterations as
    `modeling.op.solve` exposes them (op.status, op.objective.value(), variable.value).
    form = "standard": min c'y, A y = b, y >= 0 (any size; K = A S^-1 A' sparse, S diagonal);
    form = "natural": the reference's own G x <= h, A x = b (general G; needs at most 2048 equality rows)."""
    P = read_mps(filename)
    if form == "natural":
        c, G, h, A, b, c0 = natural_form(P)
        sol = _lp.conelp(c, G, h, A=A, b=b if A is not None else None, options=options)
        out = {"status": sol["status"], "iterations": sol["iterations"], "sol": sol, "problem": P,
               "natural_form": {"inequalities": int(G.size[0]), "equalities": 0 if A is None else int(A.size[0]), "cols": int(c.size)}}
        if sol["x"] is not None:
            x = np.asarray(sol["x"]).reshape(-1)
            out["x"] = dict(zip(P.cols, x.tolist()))
            out["objective"] = float(c @ x + c0)
        return out
    c, A, b, c0, recover = standard_form(P)
    ny = c.size
    ar = np.arange(ny, dtype=np.int64)
    G = spmatrix(-np.ones(ny), ar, ar, (ny, ny))
    sol = _lp.conelp(c, G, np.zeros(ny), A=A, b=b, options=options)
    out = {"status": sol["status"], "iterations": sol["iterations"], "sol": sol, "problem": P,
           "standard_form": {"rows": int(A.size[0]), "cols": int(ny), "nnz": int(A.values.size)}}
    if sol["x"] is not None:
        y = np.asarray(sol["x"]).reshape(-1)
        out["x"] = recover(y)
        out["objective"] = float(c @ y + c0)
    return out
