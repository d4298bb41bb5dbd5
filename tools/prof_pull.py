import sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, '/root/repo')
import cosmogp_amd as cosmogp
B, n = 1000, 40
rng = np.random.default_rng(0)
xs = [np.sort(rng.uniform(-12., 42., n)) for _ in range(B)]
es = [rng.uniform(0.03, 0.1, n) for _ in range(B)]
ys = [0.5 * np.sin(x / 8.) + e * rng.standard_normal(n) for x, e in zip(xs, es)]
bp = cosmogp.build_pull(ys, xs, [0.48, 14.6], y_err=es); bp.compute_pull()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    bp = cosmogp.build_pull(ys, xs, [0.48, 14.6], y_err=es); bp.compute_pull()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
