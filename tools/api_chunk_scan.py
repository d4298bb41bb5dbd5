"""API rate of config 4 (1024 / 2048 light curves per cgp_batch_fit call, host arrays in, results out, ONE handle) against the chunk
of the upload / compute pipeline and the object count from which the four-block factorisation kernel is used.
python tools/api_chunk_scan.py            (spawns one process per setting)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    B = int(sys.argv[2])
    wl = bench.make_workload('c4', nobj=B)
    dg = np.stack([K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']])
    xs, ys = np.stack(wl['xs']), np.stack(wl['ys'])
    bd = _lib.Device(0)
    for _ in range(3):
        bd.batch_fit(1, xs, ys, dg, wl['kp'], wl['xnew'], None, 0.)
    best = 0.
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(6):
            bd.batch_fit(1, xs, ys, dg, wl['kp'], wl['xnew'], None, 0.)
        best = max(best, 6 * B / (time.perf_counter() - t0))
    print('%d per call: %.0f fits/s' % (B, best), flush=True)
    sys.exit(0)
for p4 in ('512', '256', '128'):
    for ch in ('256', '512', '128'):
        for B in ('1024', '2048'):
            env = dict(os.environ, CGP_POTRF4_MIN=p4, CGP_PIPE_CHUNK=ch)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', B], env=env, capture_output=True, text=True, timeout=300)
            print('potrf4_min %s chunk %s: %s' % (p4, ch, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
