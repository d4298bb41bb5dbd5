"""Several host threads, each with its own handle, run single-launch factorisations at the same time (persistent kernels of
different streams compete for the CUs: a launch may run with only part of its workgroups resident); every result against the
same call done alone.  python tools/dag_concurrent_stress.py <seconds> <threads>"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K


def run_concurrent(budget=30., nth=3, sizes=(4100, 2600, 1300, 3300)):
    """(calls per thread, results differing from the solo run per thread) after `budget` seconds"""
    saved = {k: os.environ.get(k) for k in ('CGP_DAG_MIN_NB', 'CGP_DAG_BATCH_MIN_NB')}
    os.environ['CGP_DAG_MIN_NB'] = '2'
    os.environ['CGP_DAG_BATCH_MIN_NB'] = '2'
    jobs = []
    rng = np.random.default_rng(5)
    for t in range(nth):
        n = sizes[t % len(sizes)]
        B = [1, 3, 6, 2][t % 4]
        xs = [np.sort(rng.uniform(0., 0.15 * n, n)) for _ in range(B)]
        ys = [np.sin(0.05 * x) + 0.1 * rng.standard_normal(n) for x in xs]
        xn = rng.uniform(0., 0.15 * n, 200)
        jobs.append((n, B, xs, ys, xn))
    kp = K.kernel_params_1d([1., 2.], 0.02)

    def run(dev, job):
        n, B, xs, ys, xn = job
        diags = [K.diag_term(n, np.full(n, 0.1), 0., 0.)] * B
        if B == 1:
            dev.set_data(1, xs[0], ys[0])
            r = dev.fit(kp, diags[0], xn, None, 0.)
            return np.concatenate([[r[0]], r[2], r[3]])
        r = dev.batch_fit(1, xs, ys, diags, kp, xn, None, 0.)
        return np.concatenate([r[0], r[2].ravel(), r[3].ravel()])

    devs = [_lib.Device(0) for _ in range(nth)]
    ref = [run(d, j) for d, j in zip(devs, jobs)]
    bad = [0] * nth
    count = [0] * nth
    errors = []
    t_end = time.time() + budget

    def worker(i):
        try:
            while time.time() < t_end:
                got = run(devs[i], jobs[i])
                count[i] += 1
                if not np.array_equal(got, ref[i]):
                    bad[i] += 1
        except Exception as exc:                # a give-up of the bounded spins arrives here as HipError
            errors.append(repr(exc))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(nth)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for d in devs:
        d.close()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return count, bad, errors


if __name__ == '__main__':
    count, bad, errors = run_concurrent(float(sys.argv[1]) if len(sys.argv) > 1 else 30., int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    print('calls per thread', count, 'results differing from the solo run', bad, 'errors', errors)
    sys.exit(1 if (any(bad) or errors) else 0)
