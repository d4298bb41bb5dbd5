import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd as cg
rng = np.random.default_rng(0)
for n in (20, 50, 64, 65, 100):
    x = np.sort(rng.uniform(0., 20., n)); y = np.sin(x)
    gp = cg.gaussian_process(y, x, kernel='RBF1D')              # noise-free: the default route is the eigensolver
    gp.hyperparameters = [1., 2.]
    gp.get_prediction(new_binning=np.linspace(0., 20., 30))
    t0 = time.perf_counter()
    for _ in range(5): gp.get_prediction(new_binning=np.linspace(0., 20., 30))
    print('n=%3d noise-free default get_prediction: %.2f ms' % (n, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
