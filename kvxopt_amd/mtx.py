"""MatrixMarket coordinate reader: the front-end the reference's sparse-solver tests use to load their fixtures
(tests/test_sparse_solvers.py:36-68, `read_mtx`), as a function of the package.

    read_mtx(path, symmetric="as_stored") -> spmatrix

The reference's reader takes the `rows cols nnz` line and then one `i j value` triplet per line, exactly as stored: for a
file whose banner says `symmetric` that is the stored (lower) triangle -- what cholmod wants -- and `klu` factors that
triangle as a general matrix.  `symmetric="as_stored"` (default) reproduces this; `symmetric="expand"` mirrors the
off-diagonal entries of a `symmetric` / `skew-symmetric` file into the full matrix.  `pattern` files get the value 1.0;
`integer` values are read as floats; `complex` and `array` (dense) files are rejected (real sparse path only).
Duplicate entries are summed, as `spmatrix(V, I, J, size)` does in the reference.
"""
import numpy as np

from .base import spmatrix


def read_mtx(path, symmetric="as_stored"):
    if symmetric not in ("as_stored", "expand"):
        raise ValueError("symmetric must be 'as_stored' or 'expand'")
    field, symm = "real", "general"
    size = None
    rows = []
    with open(path, "r") as fd:
        for lineno, line in enumerate(fd):
            if line.startswith("%"):
                if lineno == 0 and line.lower().startswith("%%matrixmarket"):
                    tok = line.lower().split()
                    if len(tok) < 5 or tok[1] != "matrix":
                        raise ValueError("%s: malformed MatrixMarket banner" % path)
                    if tok[2] != "coordinate":
                        raise ValueError("%s: only the 'coordinate' (sparse) format is read" % path)
                    field, symm = tok[3], tok[4]
                    if field not in ("real", "integer", "pattern"):
                        raise TypeError("%s: field '%s' is not supported (real matrices only)" % (path, field))
                    if symm not in ("general", "symmetric", "skew-symmetric"):
                        raise TypeError("%s: symmetry '%s' is not supported" % (path, symm))
                continue
            if not line.strip():
                continue
            if size is None:
                t = line.split()
                if len(t) != 3:
                    raise ValueError("%s: expected 'rows cols nnz', got %r" % (path, line.strip()))
                size = (int(t[0]), int(t[1]), int(t[2]))
                continue
            rows.append(line)
    if size is None:
        raise ValueError("%s: no size line" % path)
    m, n, nnz = size
    if len(rows) != nnz:
        raise ValueError("%s: %d entries announced, %d found" % (path, nnz, len(rows)))
    ncol = 2 if field == "pattern" else 3
    if nnz:
        data = np.array(" ".join(rows).split(), dtype=np.float64)
        if data.size != ncol * nnz:
            raise ValueError("%s: every entry line must hold %d numbers" % (path, ncol))
        data = data.reshape(nnz, ncol)
        I = data[:, 0].astype(np.int64) - 1
        J = data[:, 1].astype(np.int64) - 1
        V = np.ones(nnz) if field == "pattern" else data[:, 2].copy()
    else:
        I = J = np.zeros(0, dtype=np.int64)
        V = np.zeros(0)
    if nnz and (I.min() < 0 or I.max() >= m or J.min() < 0 or J.max() >= n):
        raise ValueError("%s: index out of range" % path)
    if symmetric == "expand" and symm != "general":
        off = I != J
        sign = -1.0 if symm == "skew-symmetric" else 1.0
        I, J, V = np.concatenate([I, J[off]]), np.concatenate([J, I[off]]), np.concatenate([V, sign * V[off]])
    return spmatrix(V, I, J, (m, n))
