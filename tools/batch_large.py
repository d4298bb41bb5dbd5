"""Developer probe: B independent large fits through the BATCHED launch sequence (lock-step by construction)
versus B handles in flight."""
import sys, time, threading
import numpy as np
sys.path.insert(0, '.')
from cosmogp_amd import _lib, kernel as K
import bench
for wname in ('c2', 'c3'):
    wl = bench.make_workload(wname)
    diag = K.diag_term(wl['n'], wl['e'], 0., 0.)
    for B, nh in ((6, 2), (8, 2), (8, 3), (12, 1), (12, 2), (16, 1), (16, 2)):
        if True:
            devs = [_lib.Device(0) for _ in range(nh)]
            for d in devs:
                d.batch_fit(wl['dim'], [wl['x']] * B, [wl['y']] * B, [diag] * B, wl['kp'], wl['xnew'], None, 0.)
                d.bench_fit(1, stages=False)
            reps = 6 if wname == 'c3' else 20
            def run(d): d.bench_fit(reps, stages=False)
            t0 = time.perf_counter()
            ths = [threading.Thread(target=run, args=(d,)) for d in devs]
            [t.start() for t in ths]; [t.join() for t in ths]
            dt = time.perf_counter() - t0
            print('%s batch B=%d x %d handles: %.1f fits/s (%.3f ms per fit)' % (wname, B, nh, B * nh * reps / dt, dt / (B * nh * reps) * 1e3), flush=True)
            del devs
