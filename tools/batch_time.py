"""Device-resident time of one batched fit (cgp_batch_fit's launch sequence, cgp_bench_fit) for the two routes of the
batched entry points: single-launch tile DAG with several objects side by side (default from CGP_DAG_BATCH_MIN_NB tile
columns on) and one launch per step (CGP_NO_DAG_BATCH=1).
usage: python tools/batch_time.py "<n>:<objects>:<m> ..." [ENV=VALUE ...]   (environment applied before the handle is created)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[2:]:
    k, v = kv.split('=', 1)
    os.environ[k] = v
from cosmogp_amd import _lib

rng = np.random.default_rng(0)
for spec in sys.argv[1].split():
    n, B, m = (int(v) for v in spec.split(':'))
    out = []
    for route in ('dag', 'steps'):
        if route == 'steps':
            os.environ['CGP_NO_DAG_BATCH'] = '1'
        else:
            os.environ.pop('CGP_NO_DAG_BATCH', None)
        dev = _lib.Device(0)
        xs = [np.sort(rng.uniform(0, 50, n)) for _ in range(B)]
        rs = [np.sin(x) + 0.1 * rng.normal(size=n) for x in xs]
        dev.batch_set_data(1, xs, rs, [np.full(n, 0.04)] * B)
        kp = np.array([1.0, 1.0, 1.0 / 4.0, 0., 0., 0., 0.])
        from cosmogp_amd.gaussian_process import _kparams
        kp = _kparams(1, np.array([1.0, 2.0]), 0.0)
        xnew = np.linspace(0, 50, m) if m else None
        dev.batch_fit_resident(kp, xnew=xnew)
        ms, _ = dev.bench_fit(5, stages=False)
        out.append(ms)
        dev.close()
    fl = B * (n ** 3 / 3. + n * n * (m + 1.)) * 1e-9
    print('n=%5d objects=%3d m=%5d : dag %8.3f ms (%5.1f TF/s)   steps %8.3f ms (%5.1f TF/s)   dag/steps %.3f'
          % (n, B, m, out[0], fl / out[0], out[1], fl / out[1], out[0] / out[1]), flush=True)
