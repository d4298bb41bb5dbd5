import os, sys, time, threading
import numpy as np
sys.path.insert(0, '/root/repo')
import bench
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
wl = bench.make_workload('c4', 0)
diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]
px, py, pd = np.stack(wl['xs']), np.stack(wl['ys']), np.stack(diags)
for trial in range(6):
    callers = [_lib.Device(0) for _ in range(2)]
    for c_ in callers:
        c_.batch_fit(1, px, py, pd, wl['kp'], wl['xnew'], None, 0.)
    reps = 64
    def caller(c_):
        for _ in range(reps):
            c_.batch_fit(1, px, py, pd, wl['kp'], wl['xnew'], None, 0.)
    ths = [threading.Thread(target=caller, args=(c_,)) for c_ in callers]
    t0 = time.perf_counter()
    for t_ in ths: t_.start()
    for t_ in ths: t_.join()
    print('2 callers: %.0f fits/s' % (2 * reps * 128 / (time.perf_counter() - t0)), flush=True)
    for c_ in callers: c_.close()
