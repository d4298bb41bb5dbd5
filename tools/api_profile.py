import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import bench
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
wl = bench.make_workload('c4', 0)
dev = _lib.Device(0)
diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]
for _ in range(3):
    dev.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
def t(f, n=20):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print('batch_fit total        %.3f ms' % t(lambda: dev.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)))
print('batch_set_data         %.3f ms' % t(lambda: dev.batch_set_data(1, wl['xs'], wl['ys'], diags)))
print('batch_fit_resident     %.3f ms' % t(lambda: dev.batch_fit_resident(wl['kp'], wl['xnew'], None, 0.)))
print('device-resident pass   %.3f ms' % dev.bench_fit(20, stages=False)[0])
x = np.concatenate([np.asarray(a, dtype=np.float64).reshape(len(a), -1) for a in wl['xs']]).ravel()
print('python concat x        %.3f ms' % t(lambda: np.concatenate([np.asarray(a, dtype=np.float64).reshape(len(a), -1) for a in wl['xs']]).ravel()))
print('python concat y        %.3f ms' % t(lambda: np.concatenate(wl['ys'])))
