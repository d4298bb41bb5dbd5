"""Where the time of an API-level call goes (host arrays in, results out): upload vs fit, per workload."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cosmogp_amd import _lib  # noqa: E402
from cosmogp_amd import kernel as K  # noqa: E402

for name in sys.argv[1:] or ['c3', 'c2', 'c4']:
    wl = bench.make_workload(name)
    bd = _lib.Device(0)
    if 'B' in wl:
        xs, ys = wl['xs'], wl['ys']
        diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]
        dim = 1
    else:
        objs = [wl] + [bench.make_workload(name, 0, obj=i) for i in range(1, 8)]
        xs, ys = [o['x'] for o in objs], [o['y'] for o in objs]
        diags = [K.diag_term(o['n'], o['e'], 0., 0.) for o in objs]
        dim = wl['dim']
    for rep in range(4):
        t0 = time.perf_counter()
        bd.batch_set_data(dim, xs, ys, diags)
        t1 = time.perf_counter()
        out = bd.batch_fit_resident(wl['kp'], wl['xnew'], None, 0.)
        t2 = time.perf_counter()
        print('%s: set_data %.2f ms, fit_resident %.2f ms' % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    ms = bd.bench_fit_objects(3, len(xs))
    print('%s: resident pass of %d objects: %.2f ms' % (name, len(xs), ms), flush=True)
    bd.close()

# the same after other host threads have driven other handles (the runtime's synchronisation mode)
if os.environ.get('CGP_API_TIME_THREADS'):
    import threading
    wl = bench.make_workload('c4')
    diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]
    devs = [_lib.Device(0) for _ in range(4)]
    for d in devs:
        d.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
    ths = [threading.Thread(target=lambda d=d: d.bench_fit_objects(50, 128)) for d in devs]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for rep in range(4):
        t0 = time.perf_counter()
        devs[0].batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
        print('after threads: batch_fit %.2f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
