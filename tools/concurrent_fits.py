"""Developer probe: do two independent fits on two handles (two streams, separate HBM buffers) overlap
usefully on one GPU?  (leaf chain of one fit beside the bulk updates of the other)"""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
import bench

for wname in ('c2', 'c3'):
    wl = bench.make_workload(wname, 0)
    devs = [_lib.Device(0) for _ in range(3)]
    diag = K.diag_term(wl['n'], wl['e'], 0., 0.)
    for d in devs:
        d.set_data(wl['dim'], wl['x'], wl['y'])
        d.fit(wl['kp'], diag, wl['xnew'], None, 0.)
        d.bench_fit(2, stages=False)
    iters = 12
    t0 = time.perf_counter(); devs[0].bench_fit(iters, stages=False); t1 = time.perf_counter()
    print(wname, 'sequential: %.3f ms per fit' % ((t1 - t0) * 1e3 / iters))
    for nthr in (2, 3):
        ths = [threading.Thread(target=lambda d=d: d.bench_fit(iters // nthr, stages=False)) for d in devs[:nthr]]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        t1 = time.perf_counter()
        print(wname, '%d concurrent handles: %.3f ms per fit (aggregate)' % (nthr, (t1 - t0) * 1e3 / (iters // nthr * nthr)))
    for d in devs: d.close()
