"""Host threads with a handle each drive DIFFERENT entry points at the same time -- a pipelined cgp_batch_fit (ragged objects, host
arrays in), packed objects of tens of points, one large single-launch fit, a 512-object batch of 512-point light curves (four
diagonal blocks per workgroup), leave-one-out pulls, the eigen route and a gradient -- and every result is compared bit for bit
with the same call done alone.  python tools/mixed_concurrent_stress.py <seconds>"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K


def make_jobs():
    rng = np.random.default_rng(11)
    kp = K.kernel_params_1d([1., 2.], 0.02)
    jobs = []

    def objs(B, lo, hi):
        ns = rng.integers(lo, hi + 1, B)
        xs = [np.sort(rng.uniform(0., 0.15 * n + 5., n)) for n in ns]
        ys = [np.sin(0.05 * x) + 0.1 * rng.standard_normal(len(x)) for x in xs]
        dg = [K.diag_term(len(x), np.full(len(x), 0.1), 0., 0.) for x in xs]
        return xs, ys, dg
    xn19 = rng.uniform(0., 40., 19)
    xs, ys, dg = objs(600, 130, 300)

    def pipelined(dev, a=(xs, ys, dg)):
        r = dev.batch_fit(1, a[0], a[1], a[2], kp, xn19, None, 0.)
        return np.concatenate([r[0], r[2].ravel(), r[3].ravel()])
    jobs.append(('pipelined batch 600 x n<=300', pipelined))
    xs2, ys2, dg2 = objs(500, 10, 64)

    def packed(dev, a=(xs2, ys2, dg2)):
        r = dev.batch_fit(1, a[0], a[1], a[2], kp, xn19[:7], None, 0.)
        return np.concatenate([r[0], r[2].ravel(), r[3].ravel()])
    jobs.append(('packed 500 x n<=64', packed))
    n3 = 4100
    x3 = np.sort(rng.uniform(0., 600., n3))
    y3 = np.sin(0.05 * x3) + 0.1 * rng.standard_normal(n3)
    d3 = K.diag_term(n3, np.full(n3, 0.1), 0., 0.)
    xn3 = rng.uniform(0., 600., 200)

    def single(dev):
        dev.set_data(1, x3, y3)
        r = dev.fit(kp, d3, xn3, None, 0.)
        return np.concatenate([[r[0]], r[2], r[3]])
    jobs.append(('single-launch fit n=4100', single))
    xs4 = np.stack([np.sort(rng.uniform(0., 80., 512)) for _ in range(512)])
    ys4 = np.sin(0.05 * xs4) + 0.1 * rng.standard_normal(xs4.shape)
    dg4 = np.full(xs4.shape, 0.01)

    def lightcurves(dev):
        r = dev.batch_fit(1, xs4, ys4, dg4, kp, xn19, None, 0.)
        return np.concatenate([r[0], r[2].ravel(), r[3].ravel()])
    jobs.append(('512 x 512-point light curves', lightcurves))
    xs5, ys5, dg5 = objs(40, 60, 200)

    def pulls(dev, a=(xs5, ys5, dg5)):
        dev.batch_set_data(1, a[0], a[1], a[2])
        al, be, g, info = dev.batch_loo(kp)
        return np.concatenate([np.concatenate([v[b, :len(a[0][b])] for b in range(len(a[0]))]) for v in (al, be, g)] + [info.astype(float)])
    jobs.append(('leave-one-out of 40 objects', pulls))
    x6 = np.sort(rng.uniform(0., 60., 300))
    K6 = np.exp(-0.5 * (x6[:, None] - x6[None, :]) ** 2 / 4.)

    def eig(dev):
        r = dev.svd_inverse(K6, 1e-9)
        return np.ravel(np.asarray(r[0] if isinstance(r, tuple) else r, dtype=float))
    jobs.append(('eigen route n=300 (noise-free)', eig))
    return jobs


def run_mixed(budget=30.):
    jobs = make_jobs()
    devs = [_lib.Device(0) for _ in jobs]
    ref = [fn(d) for d, (_, fn) in zip(devs, jobs)]
    count, bad, errors = [0] * len(jobs), [0] * len(jobs), []
    t_end = time.time() + budget

    def worker(i):
        try:
            while time.time() < t_end:
                got = jobs[i][1](devs[i])
                count[i] += 1
                if not np.array_equal(got, ref[i], equal_nan=True):
                    bad[i] += 1
        except Exception as exc:
            errors.append('%s: %r' % (jobs[i][0], exc))
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for d in devs:
        d.close()
    return [j[0] for j in jobs], count, bad, errors


if __name__ == '__main__':
    names, count, bad, errors = run_mixed(float(sys.argv[1]) if len(sys.argv) > 1 else 30.)
    for n_, c, b in zip(names, count, bad):
        print('%-34s %6d calls, %d differing from the solo run' % (n_, c, b))
    print('errors', errors)
    sys.exit(1 if (any(bad) or errors) else 0)
