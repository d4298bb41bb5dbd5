"""Leave-one-out pulls (build_pull.compute_pull): one factorisation per object (cgp_batch_loo, default) against the literal
refits of every (object, point) pair (LOO_BY_REFITTING): python tools/pull_time.py "<objects>x<n> ..." [max n for the literal form]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosmogp_amd as cg
pm = sys.modules['cosmogp_amd.pull']
lit_max = int(sys.argv[2]) if len(sys.argv) > 2 else 600
for spec in sys.argv[1].split():
    B, n = (int(v) for v in spec.split('x'))
    rng = np.random.default_rng(B * 1000 + n)
    xs = [np.sort(rng.uniform(0., 0.4 * n + 4., n)) for _ in range(B)]
    es = [rng.uniform(0.05, 0.2, n) for _ in range(B)]
    ys = [np.sin(x / 3.) + e * rng.standard_normal(n) for x, e in zip(xs, es)]
    times, res = [], []
    for literal in (False, True):
        if literal and n > lit_max:
            times.append(float('nan')); res.append(None); continue
        pm.LOO_BY_REFITTING = literal
        best = 1e9
        for _ in range(2):
            bp = cg.build_pull(ys, xs, [0.9, 2.5], nugget=0.03, y_err=es)
            t0 = time.perf_counter(); bp.compute_pull(); best = min(best, time.perf_counter() - t0)
        times.append(best); res.append(np.array(bp.pull))
    pm.LOO_BY_REFITTING = False
    diff = np.abs(res[0] - res[1]).max() if res[1] is not None else float('nan')
    print('%5d objects x %5d points: one factorisation per object %9.2f ms   refits %10.2f ms   (x%.0f)   max |pull diff| %.1e   std %.3f'
          % (B, n, times[0] * 1e3, times[1] * 1e3, times[1] / times[0], diff, bp.pull_std), flush=True)
