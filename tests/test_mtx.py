"""MatrixMarket front-end (kvxopt_amd/mtx.py) -- the reader of the reference's sparse-solver tests
(tests/test_sparse_solvers.py:36-68) as a package function.  CPU only."""
import os

import numpy as np
import pytest

from kvxopt_amd import mtx


def _write(path, banner, size, lines):
    with open(path, "w") as fd:
        fd.write(banner + "\n% a comment\n%\n")
        fd.write("%d %d %d\n" % size)
        fd.write("\n".join(lines) + ("\n" if lines else ""))


@pytest.mark.parametrize("name,banner", [("bcsstk13", "symmetric"), ("bp_800", "general")])
def test_fixture_round_trip(golden_dir, tmp_path, name, banner):
    """The .npz fixtures are the reference's .mtx data files read the reference's way (tests/golden/convert_mtx.py);
    written back as MatrixMarket text (shuffled, 17 significant digits) the reader returns the same CCS arrays bit for bit
    -- with a `symmetric` banner too, since the reference reads the triplets as stored."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, cp, ri, v = int(g["n"]), g["colptr"], g["rowind"], g["values"]
    cols = np.repeat(np.arange(n), np.diff(cp))
    order = np.random.default_rng(3).permutation(ri.size)
    lines = ["%d %d %.17g" % (ri[k] + 1, cols[k] + 1, v[k]) for k in order]
    p = str(tmp_path / (name + ".mtx"))
    _write(p, "%%MatrixMarket matrix coordinate real " + banner, (n, n, ri.size), lines)
    A = mtx.read_mtx(p)
    assert A.size == (n, n)
    assert np.array_equal(A.colptr, cp) and np.array_equal(A.rowind, ri) and np.array_equal(A.values, v)


def test_symmetry_fields_and_errors(tmp_path):
    p = str(tmp_path / "a.mtx")
    _write(p, "%%MatrixMarket matrix coordinate real symmetric", (3, 3, 4), ["1 1 2.0", "3 1 -1.5", "2 2 4", "3 3 1e1"])
    L = mtx.read_mtx(p).todense()
    assert np.array_equal(L, [[2.0, 0, 0], [0, 4.0, 0], [-1.5, 0, 10.0]])
    F = mtx.read_mtx(p, symmetric="expand").todense()
    assert np.array_equal(F, L + np.tril(L, -1).T)
    _write(p, "%%MatrixMarket matrix coordinate real skew-symmetric", (2, 2, 1), ["2 1 3.0"])
    assert np.array_equal(mtx.read_mtx(p, symmetric="expand").todense(), [[0, -3.0], [3.0, 0]])
    _write(p, "%%MatrixMarket matrix coordinate pattern general", (2, 3, 2), ["1 3", "2 1"])
    assert np.array_equal(mtx.read_mtx(p).todense(), [[0, 0, 1.0], [1.0, 0, 0]])
    _write(p, "%%MatrixMarket matrix coordinate integer general", (2, 2, 3), ["1 1 2", "1 1 3", "2 2 -7"])      # duplicates are summed
    assert np.array_equal(mtx.read_mtx(p).todense(), [[5.0, 0], [0, -7.0]])
    _write(p, "%%MatrixMarket matrix coordinate real general", (4, 2, 0), [])
    E = mtx.read_mtx(p)
    assert E.size == (4, 2) and len(E) == 0
    _write(p, "%%MatrixMarket matrix coordinate complex general", (1, 1, 1), ["1 1 1.0 2.0"])
    with pytest.raises(TypeError):
        mtx.read_mtx(p)
    _write(p, "%%MatrixMarket matrix array real general", (1, 1, 1), ["1.0"])
    with pytest.raises(ValueError):
        mtx.read_mtx(p)
    _write(p, "%%MatrixMarket matrix coordinate real general", (2, 2, 2), ["1 1 1.0"])
    with pytest.raises(ValueError):
        mtx.read_mtx(p)
    _write(p, "%%MatrixMarket matrix coordinate real general", (2, 2, 1), ["3 1 1.0"])
    with pytest.raises(ValueError):
        mtx.read_mtx(p)
    with pytest.raises(ValueError):
        mtx.read_mtx(p, symmetric="maybe")
