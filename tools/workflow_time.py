"""A reference user's workflow on light-curve-like data through the Python classes (cosmogp_amd as cosmogp):
gaussian_process_nobject -> find_hyperparameters -> get_prediction -> build_pull.  python tools/workflow_time.py <objects> <points>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cosmogp_amd as cosmogp

B, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
xs = [np.sort(rng.uniform(-12., 42., n)) for _ in range(B)]
es = [rng.uniform(0.03, 0.1, n) for _ in range(B)]
ys = [0.5 * np.sin(x / 8.) + e * rng.standard_normal(n) for x, e in zip(xs, es)]
grid = np.linspace(-12., 42., 60)

def lap(label, t0):
    t1 = time.perf_counter(); print('%-34s %9.2f ms' % (label, (t1 - t0) * 1e3), flush=True); return t1

for rep in range(2):
    print('--- pass %d (%d objects x %d points)' % (rep, B, n))
    t = time.perf_counter()
    gp = cosmogp.gaussian_process_nobject(ys, xs, y_err=es, kernel='RBF1D')
    t = lap('constructor', t)
    gp.find_hyperparameters(hyperparameter_guess=[0.5, 5.], nugget=False)
    t = lap('find_hyperparameters (fmin)', t)
    gp.get_prediction(new_binning=grid, COV=False)
    t = lap('get_prediction (60-point grid)', t)
    bp = cosmogp.build_pull(ys, xs, gp.hyperparameters, nugget=gp.nugget, y_err=es)
    bp.compute_pull()
    t = lap('build_pull.compute_pull', t)
    print('    hyperparameters', np.round(gp.hyperparameters, 4), 'pull std %.3f' % bp.pull_std)
