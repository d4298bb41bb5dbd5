"""numpy experiment behind DESIGN.md 4.5 / 9: how many sweeps do Jacobi-type eigensolvers need on the matrices of this path
(RBF + noise: a few hundred graded eigenvalues above a cluster of thousands at the noise floor)?
  python tools/jac_sweeps_experiment.py <n> "two one"
two : two-sided cyclic Jacobi on K (round-robin parallel ordering, threshold eps ||K||_F) -- what csrc/jacobi.hpp runs by blocks
one : one-sided Jacobi on the rows of L^T after a Cholesky factorisation K = L L^T (the usual preconditioning), relative and
      absolute stopping rule
Result at n = 512: 16 sweeps two-sided, 17 (absolute rule) / 24 (relative rule) one-sided: the Cholesky preconditioning does
NOT shorten the iteration on these matrices, so it was not built."""
import numpy as np, sys, time
def rr_pairs(n):
    # circle method rounds
    idx = list(range(n))
    for r in range(n - 1):
        a = [idx[i] for i in range(n // 2)]
        b = [idx[n - 1 - i] for i in range(n // 2)]
        yield np.array(a), np.array(b)
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
def two_sided(K, tolf=1.0, maxs=60):
    A = K.copy(); n = len(A)
    tol = tolf * 2.2e-16 * np.linalg.norm(A)
    for s in range(maxs):
        nrot = 0
        for p, q in rr_pairs(n):
            apq = A[p, q]; m = np.abs(apq) > tol
            if not m.any(): continue
            nrot += m.sum()
            tau = np.where(m, (A[q, q] - A[p, p]) / (2 * np.where(m, apq, 1.)), 0.)
            t = np.where(m, np.sign(tau + (tau == 0)) / (np.abs(tau) + np.sqrt(1 + tau * tau)), 0.)
            c = 1 / np.sqrt(1 + t * t); sn = t * c
            Ap, Aq = A[p, :].copy(), A[q, :].copy()
            A[p, :] = c[:, None] * Ap - sn[:, None] * Aq
            A[q, :] = sn[:, None] * Ap + c[:, None] * Aq
            Ap, Aq = A[:, p].copy(), A[:, q].copy()
            A[:, p] = c[None, :] * Ap - sn[None, :] * Aq
            A[:, q] = sn[None, :] * Ap + c[None, :] * Aq
        off = np.sqrt(max(0., np.linalg.norm(A) ** 2 - np.sum(np.diag(A) ** 2)))
        print('  two-sided sweep %d: %d rotations, off %.2e' % (s, nrot, off), flush=True)
        if nrot == 0: return s + 1
    return maxs
def one_sided(B, rel=True, tolf=1.0, maxs=60, absn=None):
    # rows of B are orthogonalised: B <- J^T B
    B = B.copy(); n = len(B)
    eps = 2.2e-16
    for s in range(maxs):
        nrot = 0
        for p, q in rr_pairs(n):
            Bp, Bq = B[p], B[q]
            gpp = np.einsum('ij,ij->i', Bp, Bp); gqq = np.einsum('ij,ij->i', Bq, Bq); gpq = np.einsum('ij,ij->i', Bp, Bq)
            if rel: m = np.abs(gpq) > tolf * eps * np.sqrt(gpp * gqq) * 4
            else: m = np.abs(gpq) > tolf * eps * absn
            if not m.any(): continue
            nrot += m.sum()
            tau = np.where(m, (gqq - gpp) / (2 * np.where(m, gpq, 1.)), 0.)
            t = np.where(m, np.sign(tau + (tau == 0)) / (np.abs(tau) + np.sqrt(1 + tau * tau)), 0.)
            c = 1 / np.sqrt(1 + t * t); sn = t * c
            B[p] = c[:, None] * Bp - sn[:, None] * Bq
            B[q] = sn[:, None] * Bp + c[:, None] * Bq
        print('  one-sided sweep %d: %d rotations' % (s, nrot), flush=True)
        if nrot == 0: return s + 1, B
    return maxs, B
n = int(sys.argv[1]); mode = sys.argv[2]
rng = np.random.default_rng(n)
x = np.sort(rng.uniform(0., 0.1 * n, n))
d = x[:, None] - x[None, :]
K = np.exp(-0.5 * d * d / 4.) + 1e-2 * np.eye(n)
w = np.linalg.eigvalsh(K)
print('n', n, 'lambda range', w[0], w[-1], 'frob', np.linalg.norm(K))
if 'two' in mode:
    print('two-sided sweeps:', two_sided(K))
if 'one' in mode:
    L = np.linalg.cholesky(K)
    for rel in (True, False):
        s, B = one_sided(L.T.copy(), rel=rel, absn=np.linalg.norm(K))
        lam = np.sort(np.einsum('ij,ij->i', B, B))
        print('one-sided on L^T (rel=%s) sweeps: %d ; max rel eigenvalue error %.2e' % (rel, s, np.max(np.abs(lam - w) / w)))
if 'onek' in mode:
    s, B = one_sided(K.copy(), rel=True)
    print('one-sided on K sweeps', s)
