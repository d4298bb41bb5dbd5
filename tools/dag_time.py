"""Single-fit timings: single-launch factorisation (default) vs the launch-per-step route, log-likelihood-only and full fits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cosmogp_amd import _lib  # noqa: E402
from cosmogp_amd import kernel as K  # noqa: E402


def run(wl_name, env):
    for k in ('CGP_NO_DAG', 'CGP_DAG_DEPTH', 'CGP_DAG_GRID'):
        os.environ.pop(k, None)
    os.environ.update(env)
    wl = bench.make_workload(wl_name)
    dev = _lib.Device(0)
    dev.set_data(wl['dim'], wl['x'], wl['y'])
    diag = K.diag_term(wl['n'], wl['e'], 0., 0.)
    out = {}
    for label, xn in (('chol', None), ('full', wl['xnew'])):
        dev.fit(wl['kp'], diag, xn, None, 0.)
        dev.bench_fit(3, stages=False)
        ms = min(dev.bench_fit(5, stages=False)[0] for _ in range(3))
        out[label] = ms
    dev.close()
    n = wl['n']
    return out, n ** 3 / 3. / (out['chol'] * 1e-3) / 1e12


if __name__ == '__main__':
    for wl in sys.argv[1:] or ['c3', 'c2']:
        for env in ({'CGP_NO_DAG': '1'}, {}, {'CGP_DAG_DEPTH': '2'}, {'CGP_DAG_DEPTH': '4'}, {'CGP_DAG_DEPTH': '8'}):
            out, tf = run(wl, env)
            print('%s %-22s chol-only %.3f ms (%.1f TFLOP/s)  full fit %.3f ms' % (wl, env or 'dag depth 1', out['chol'], tf, out['full']),
                  flush=True)
