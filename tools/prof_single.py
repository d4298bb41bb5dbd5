import sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, '/root/repo')
import cosmogp_amd as cosmogp
for n in (60, 1000):
    rng = np.random.default_rng(0)
    x = np.sort(rng.uniform(0., 0.5 * n, n)); e = rng.uniform(0.03, 0.1, n); y = np.sin(x / 3.) + e * rng.standard_normal(n)
    gp = cosmogp.gaussian_process(y, x, y_err=e, kernel='RBF1D')
    gp.compute_log_likelihood([0.5, 5.])
    t0 = time.perf_counter()
    for _ in range(50): gp.compute_log_likelihood([0.5, 5.1])
    print('n=%d compute_log_likelihood per call %.3f ms' % (n, (time.perf_counter() - t0) / 50 * 1e3))
    t0 = time.perf_counter(); gp.find_hyperparameters(hyperparameter_guess=[0.5, 5.], nugget=False); print('   find_hyperparameters %.1f ms' % ((time.perf_counter() - t0) * 1e3))
    grid = np.linspace(0., 0.5 * n, 200)
    gp.get_prediction(new_binning=grid)
    t0 = time.perf_counter()
    for _ in range(20): gp.get_prediction(new_binning=grid)
    print('   get_prediction per call %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): gp.compute_log_likelihood([0.5, 5.2])
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(10)
