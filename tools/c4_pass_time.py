"""Config 4 (128 light curves of 512 points, 19 prediction points) on ONE handle: device time of a pass, first-touch generation
(default) against the assembly launches (CGP_NO_FUSED_ASM=1), with the stage split of the profiling entry point.
python tools/c4_pass_time.py [objects]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wl = bench.make_workload('c4', nobj=nobj)
diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]
for flag in (None, '1'):
    if flag:
        os.environ['CGP_NO_FUSED_ASM'] = flag
    else:
        os.environ.pop('CGP_NO_FUSED_ASM', None)
    dev = _lib.Device(0)
    dev.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
    best = min(dev.bench_fit(200, stages=False)[0] for _ in range(3))
    ms, st = dev.bench_fit(20, stages=True)
    print('%-28s %.4f ms per pass = %.0f fits/s on one handle; stages (ms): %s'
          % ('assembly launches' if flag else 'first-touch generation', best, nobj / best * 1e3, np.round(np.asarray(st), 4)), flush=True)
    dev.close()
