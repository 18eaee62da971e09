# scratch GPU check: factor+solve vs oracle on small problems
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from kvxopt_amd.chol import Factor
from oracle.kvx_oracle import OracleChol

def lap2d(g):
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
    A = (sp.kron(sp.eye(g), T) + sp.kron(T, sp.eye(g))).tocsc()
    L = sp.tril(A).tocsc(); L.sort_indices(); return A, L

def check(name, A, L, nrhs=3, oracle=True):
    n = A.shape[0]
    t0 = time.time(); F = Factor(n, L.indptr, L.indices); t1 = time.time()
    inf = F.info(); print(name, {k: inf[k] for k in ('n','lnz','flops','nsuper','lsize','nlevels','max_front')}, 'analyze %.2fs' % (t1-t0))
    F.factorize(L.data)
    rng = np.random.default_rng(1); B = rng.standard_normal((n, nrhs)); X = np.asfortranarray(B.copy())
    F.solve(X)
    res = np.linalg.norm(A @ X - B) / np.linalg.norm(B)
    print('  rel residual %.3e' % res, 'timing ms', F.timing())
    if oracle:
        O = OracleChol(n, L.indptr, L.indices, 'L', F.perm()); O.factorize(L.data)
        assert O.lnz == inf['lnz'], (O.lnz, inf['lnz'])
        Xo = np.asfortranarray(B.copy()); O.solve(Xo)
        print('  vs oracle max rel diff %.3e' % (np.abs(X - Xo).max() / np.abs(Xo).max()))
        d = F.diag(); do = O.diag(); print('  diag diff %.3e' % (np.abs(d - do).max() / np.abs(do).max()))
        # sys 4,5,7,8
        for s in (4, 5, 7, 8, 1):
            Y = np.asfortranarray(B.copy()); Yo = np.asfortranarray(B.copy()); F.solve(Y, sys=s); O.solve(Yo, sys=s)
            print('   sys', s, 'diff %.3e' % (np.abs(Y - Yo).max() / max(np.abs(Yo).max(), 1e-300)))
    assert res < 1e-10
    return F

V=[10,3,5,-2,5,2]; I=[0,2,1,3,2,3]; J=[0,0,1,1,2,3]
A4 = sp.csc_matrix((V,(I,J)),shape=(4,4)); A4f = (A4 + sp.tril(A4,-1).T).tocsc(); A4.sort_indices()
check('doc4', A4f, A4, 2)
for g in (8, 30, 100, 300):
    A, L = lap2d(g); check('lap%d' % g, A, L)
# random SPD
rng = np.random.default_rng(0); n = 3000
M = sp.random(n, n, 0.002, random_state=1, format='csc'); S = (M @ M.T + sp.eye(n) * 2).tocsc(); L = sp.tril(S).tocsc(); L.sort_indices()
check('rand3000', S, L)
A, L = lap2d(1000); t=time.time(); F = check('lap1000', A, L, 1, oracle=False); print('total', time.time()-t)
for i in range(3):
    F.factorize(L.data); X = np.ones(A.shape[0]); F.solve(X); print('timing ms', F.timing())
# not PD
A, L = lap2d(20); d = L.data.copy(); d[L.indptr[150]] = -5.0
F = Factor(400, L.indptr, L.indices)
try:
    F.factorize(d); print('ERROR: no exception')
except ArithmeticError as e:
    print('ArithmeticError minor', e.args)
print('DONE')
