"""Developer timing: joint likelihood / gradient of many small objects (the reference's typical nobject use)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import cosmogp_amd as cg
rng = np.random.default_rng(1)
for B, n in ((1000, 40), (256, 300)):
    xs = [np.sort(rng.uniform(-12., 42., n)) for _ in range(B)]
    es = [rng.uniform(0.02, 0.1, n) for _ in range(B)]
    ys = [0.5 * np.sin(x / 8.) + e * rng.standard_normal(n) for x, e in zip(xs, es)]
    gp = cg.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es)
    h = [0.5, 8.]
    gp.compute_log_likelihood(h, svd_method=False); gp.compute_log_likelihood_gradient(h)
    t0 = time.perf_counter(); gp.compute_log_likelihood(h, svd_method=False); t1 = time.perf_counter()
    gp.compute_log_likelihood_gradient(h); t2 = time.perf_counter()
    print('%d objects x N=%d: joint log-likelihood %.2f ms, with gradient %.2f ms' % (B, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    t3 = time.perf_counter(); gp.find_hyperparameters(hyperparameter_guess=[0.8, 5.], svd_method=False, method='L-BFGS-B'); t4 = time.perf_counter()
    hb = gp.hyperparameters.copy(); nb = gp.fit_result.nfev
    gp.find_hyperparameters(hyperparameter_guess=[0.8, 5.], svd_method=False); t5 = time.perf_counter()
    print('   L-BFGS-B fit %.0f ms (%d evaluations) -> %s ; Nelder-Mead fit %.0f ms -> %s' % ((t4 - t3) * 1e3, nb, hb, (t5 - t4) * 1e3, gp.hyperparameters))
