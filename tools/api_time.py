import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
wl = bench.make_workload('c3')
objs = [wl] + [bench.make_workload('c3', 0, obj=i) for i in range(1, 8)]
bd = _lib.Device(0)
diags = [K.diag_term(o['n'], o['e'], 0., 0.) for o in objs]
xs=[o['x'] for o in objs]; ys=[o['y'] for o in objs]
for rep in range(4):
    t0=time.perf_counter(); bd.batch_set_data(2, xs, ys, diags); t1=time.perf_counter()
    out = bd.batch_fit_resident(wl['kp'], wl['xnew'], None, 0.); t2=time.perf_counter()
    print('set_data %.2f ms, fit_resident %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3), flush=True)
ms = bd.bench_fit_objects(3, 8)
print('resident pass of 8: %.2f ms' % ms)
