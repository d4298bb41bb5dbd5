"""Config 4 (128 light curves of 512 points, 19 prediction points) on ONE handle under several environments (each one a fresh
handle: the CGP_* switches are read at cgp_create): device time of a pass, stage split.
python tools/c4_ab.py [objects] [ENV=VAL,ENV=VAL ...]   (first variant: the defaults)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = [''] + sys.argv[2:]
wl = bench.make_workload('c4', nobj=nobj)
diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]
base = dict(os.environ)
ref = None
for v in variants:
    os.environ.clear(); os.environ.update(base)
    for kv in [t for t in v.split(',') if t]:
        k, val = kv.split('=')
        os.environ[k] = val
    dev = _lib.Device(0)
    out = dev.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
    same = '' if ref is None else ('  same bits' if all(np.array_equal(a, b) for a, b in zip(ref, out)) else '  DIFFERENT BITS (max |d mu| %.3g)' % np.max(np.abs(ref[2] - out[2])))
    if ref is None:
        ref = out
    best = min(dev.bench_fit(200, stages=False)[0] for _ in range(3))
    ms, st = dev.bench_fit(20, stages=True)
    print('%-40s %.4f ms per pass = %7.0f fits/s on one handle; potrf %.4f trsm %.4f gemm %.4f finish %.4f%s'
          % (v or 'defaults', best, nobj / best * 1e3, st[1], st[2], st[3], st[5], same), flush=True)
    dev.close()
