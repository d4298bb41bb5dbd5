import sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, '/root/repo')
import cosmogp_amd as cosmogp
B, n = 1000, 40
rng = np.random.default_rng(0)
xs = [np.sort(rng.uniform(-12., 42., n)) for _ in range(B)]
es = [rng.uniform(0.03, 0.1, n) for _ in range(B)]
ys = [0.5 * np.sin(x / 8.) + e * rng.standard_normal(n) for x, e in zip(xs, es)]
gp = cosmogp.gaussian_process_nobject(ys, xs, y_err=es, kernel='RBF1D')
gp.compute_log_likelihood([0.5, 5.])
t0 = time.perf_counter()
for _ in range(20): gp.compute_log_likelihood([0.5, 5.1])
print('compute_log_likelihood per call %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): gp.compute_log_likelihood([0.5, 5.2])
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
