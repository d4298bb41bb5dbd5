"""Developer script (not collected by pytest): first-light checks on the GPU box."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from oracle import cosmogp_oracle as orc
from tests.conftest import ulp_diff

d = _lib.Device(0)
print('selftest', d.selftest_mfma())
print('microbench mfma TF/s, HBM write GB/s:', d.microbench())

def kp1(h): return np.array([h[0]**2, h[0]**2, h[1]**2, 0, 0, 0.])
def kp2(h, compat=True):
    vi = orc.inv_l_2d(h)
    return np.array([1.0 if compat else h[0]**2, h[0]**2, vi[0,0], vi[0,1], vi[1,0], vi[1,1]])

rng = np.random.default_rng(0)
for n, m in ((100, 7), (256, 64), (300, 130), (1000, 200)):
    x = np.sort(rng.uniform(0, 50, n)); xn = np.linspace(0, 50, m)
    e = rng.uniform(0.05, 0.1, n); hyp = np.array([0.7, 3.0])
    y = np.sin(x) + 0.1 * rng.standard_normal(n)
    K = orc.rbf_kernel_1d(x, hyp, y_err=e, nugget=0.02)
    H = orc.rbf_kernel_1d(x, hyp, new_x=xn)
    dg = orc.diag_noise(n, e, 0., 0.02)
    Kd = d.rbf(1, x, kp1(hyp), diag=dg)
    Hd = d.rbf(1, x, kp1(hyp), xnew=xn)
    print(n, 'K ulp', ulp_diff(K, Kd), 'H ulp', ulp_diff(H, Hd))
    mu, var, ll = orc.fit_predict_cho(K, y, H, np.full(m, hyp[0]**2 + 0.02**2))
    d.set_data(1, x, y)
    t = time.time()
    lld, ldd, mud, vard = d.fit(kp1(hyp), dg, xn, None, 0.02**2)
    print('  fit time', time.time() - t)
    print('  ll', ll, lld, 'rel', abs(ll - lld) / abs(ll))
    print('  mu err', np.abs(mu - mud).max(), 'var err', np.abs(var - vard).max())
    mu2, var2, cov2 = d.predict(xn[: max(1, m // 2)], None, 0.02**2, want_cov=True)
    print('  predict err', np.abs(mu2 - mu[: len(mu2)]).max(), np.abs(var2 - var[: len(mu2)]).max(),
          'cov diag err', np.abs(np.diag(cov2) - var2).max())
    L = d.fetch(_lib.MAT_L)
    print('  L err', np.abs(L - np.linalg.cholesky(K)).max())
    Ki = d.fetch(_lib.MAT_KINV)
    print('  Kinv err', np.abs(Ki - np.linalg.inv(K)).max() / np.abs(Ki).max())
    inv, ld = d.cholesky_inverse(K)
    print('  cholesky_inverse err', np.abs(inv - np.linalg.inv(K)).max() / np.abs(inv).max(), 'logdet', ld, np.linalg.slogdet(K)[1])

# 2D
n, m = 500, 100
X = rng.uniform(0, 30, (n, 2)); Xn = rng.uniform(0, 30, (m, 2)); e = rng.uniform(0.05, 0.1, n)
hyp = np.array([0.8, 3.0, 2.5, 1.2]); y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(n)
K = orc.rbf_kernel_2d(X, hyp, y_err=e); H = orc.rbf_kernel_2d(X, hyp, new_x=Xn)
Kd = d.rbf(2, X, kp2(hyp), diag=orc.diag_noise(n, e, 0, 0)); Hd = d.rbf(2, X, kp2(hyp), xnew=Xn)
print('2D K ulp', ulp_diff(K, Kd), 'H ulp', ulp_diff(H, Hd))
mu, var, ll = orc.fit_predict_cho(K, y, H, np.full(m, 1.0))
d.set_data(2, X, y)
lld, ldd, mud, vard = d.fit(kp2(hyp), orc.diag_noise(n, e, 0, 0), Xn, None, 0.)
print('  ll', ll, lld, 'mu err', np.abs(mu - mud).max(), 'var err', np.abs(var - vard).max())

# batch
B = 5
xs = [np.sort(rng.uniform(-12, 42, k)) for k in (200, 512, 300, 129, 512)]
es = [rng.uniform(0.02, 0.1, len(a)) for a in xs]
ys = [np.sin(0.2 * a) + 0.05 * rng.standard_normal(len(a)) for a in xs]
hyp = np.array([0.5, 8.0]); xn = np.linspace(-12, 42, 19)
ll, info, mu, var = d.batch_fit(1, xs, ys, [orc.diag_noise(len(a), e_, 0, 0) for a, e_ in zip(xs, es)], kp1(hyp), xn, None, 0.)
for b in range(B):
    K = orc.rbf_kernel_1d(xs[b], hyp, y_err=es[b]); H = orc.rbf_kernel_1d(xs[b], hyp, new_x=xn)
    m0, v0, l0 = orc.fit_predict_cho(K, ys[b], H, np.full(19, hyp[0]**2))
    print('batch', b, info[b], abs(l0 - ll[b]) / abs(l0), np.abs(m0 - mu[b]).max(), np.abs(v0 - var[b]).max())

# timing big
for n, m, dim in ((4096, 1024, 1), (8192, 2500, 2)):
    if dim == 1:
        x = np.sort(rng.uniform(0, 400, n)); xn = rng.uniform(0, 400, m); k = kp1([1., 2.])
    else:
        x = rng.uniform(0, 2048, (n, 2)); xn = rng.uniform(0, 2048, (m, 2)); k = kp2([1., 110., 110., 0.])
    y = rng.standard_normal(n) * 0.1
    d.set_data(dim, x, y)
    dg = np.full(n, 0.01)
    t = time.time(); out = d.fit(k, dg, xn, None, 0.); print(n, 'first fit', time.time() - t, 'll', out[0])
    ms, st = d.bench_fit(5)
    print('  ms/fit', ms, 'stages', st)
    print('  potrf-only flops', n**3 / 3 / 1e9, 'GF; total gemm GF', st[7] / 1e9, ' gemm TF/s', st[7] / st[3] / 1e9)
