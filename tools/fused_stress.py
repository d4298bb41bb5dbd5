"""Randomised check of the first-touch generation (csrc/kernels.hpp, GenSrc) and, since round 4, of everything else that must not
show in a number: random dimension, sizes, batch, prediction grid (shared or one per object), mean, hyper-parameters; every output
of cgp_batch_fit / cgp_fit on the DEFAULT route -- first touch, XCD-affine launches, compact tall rows, objects of <= 64 points packed
two / four per tile, the call pipelined in chunks of 8 / 16 objects or not at all -- against the same call with the assembly
launches on the plain grid, one tile per object, one pass (CGP_NO_FUSED_ASM=1 CGP_NO_XCD=1 CGP_PIPE_CHUNK=0), bit for bit.
python tools/fused_stress.py <seconds> [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K


def run_fused_stress(budget, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_bad = 0
    while time.time() < t_end:
        dim = int(rng.choice([1, 2]))
        nmax = int(rng.choice([3, 17, 64, 127, 128, 129, 200, 256, 300, 512, 513, 700, 900, 1100, 1500, 1900]))
        B = int(rng.choice([1, 1, 2, 3, 7, 20, 37, 64]))
        if nmax * nmax * B > 1500 * 1500 * 8:
            B = 2
        ns = [nmax] + [int(rng.integers(max(2, nmax // 3), nmax + 1)) for _ in range(B - 1)]
        m = int(rng.choice([0, 1, 19, 63, 64, 65, 130, 400]))
        if dim == 1:
            xs = [np.sort(rng.uniform(0., 30., n)) for n in ns]
            ys = [np.sin(x / 3.) + 0.1 * rng.standard_normal(len(x)) for x in xs]
            xn = rng.uniform(0., 30., m) if m else None
            kp = K.kernel_params_1d([float(rng.uniform(0.5, 1.5)), float(rng.uniform(1., 4.))], float(rng.choice([0., 0.03])))
        else:
            xs = [rng.uniform(0., 30., (n, 2)) for n in ns]
            ys = [np.sin(x[:, 0] / 3.) * np.cos(x[:, 1] / 4.) + 0.1 * rng.standard_normal(len(x)) for x in xs]
            xn = rng.uniform(0., 30., (m, 2)) if m else None
            lxy = float(rng.choice([0., 2.5]))
            kp = K.kernel_params_2d([float(rng.uniform(0.5, 1.5)), float(rng.uniform(3., 6.)), float(rng.uniform(3., 6.)), lxy],
                                    sigma2_in_train=bool(rng.integers(0, 2)), nugget=float(rng.choice([0., 0.03])))
        diags = [K.diag_term(len(y), rng.uniform(0.05, 0.2, len(y)), 0., 0.) for y in ys]
        mean = rng.uniform(-1., 1., m) if (m and rng.random() < 0.5) else None
        single = B == 1 and rng.random() < 0.5
        per_obj = (not single) and m > 0 and rng.random() < 0.3
        if per_obj:
            xn = rng.uniform(0., 30., (B, m) if dim == 1 else (B, m, 2))
            mean = None if mean is None else mean
        chunk = str(rng.choice(['', '8', '16']))
        outs = []
        for flag in (None, '1'):
            for k_ in ('CGP_NO_FUSED_ASM', 'CGP_NO_XCD', 'CGP_PIPE_CHUNK'):
                os.environ.pop(k_, None)
            if flag:
                os.environ.update(CGP_NO_FUSED_ASM='1', CGP_NO_XCD='1', CGP_PIPE_CHUNK='0')
            elif chunk:
                os.environ['CGP_PIPE_CHUNK'] = chunk
            dev = _lib.Device(0)
            if single:
                dev.set_data(dim, xs[0], ys[0])
                r = dev.fit(kp, diags[0], xn, mean, 0.01)
                res = [np.array([r[0], r[1]])] + ([r[2], r[3]] if m else [])
            else:
                r = dev.batch_fit(dim, xs, ys, diags, kp, xn, mean, 0.01, xnew_per_object=per_obj)
                res = [r[0], r[1]] + ([r[2], r[3]] if m else [])
            dev.close()
            outs.append(res)
        n_cases += 1
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*outs))
        if not same:
            n_bad += 1
            if verbose:
                print('MISMATCH dim=%d B=%d ns=%s m=%d per_obj=%s chunk=%r single=%s' % (dim, B, ns[:4], m, per_obj, chunk, single), flush=True)
    for k_ in ('CGP_NO_FUSED_ASM', 'CGP_NO_XCD', 'CGP_PIPE_CHUNK'):
        os.environ.pop(k_, None)
    return n_cases, n_bad


if __name__ == '__main__':
    n_cases, n_bad = run_fused_stress(float(sys.argv[1]) if len(sys.argv) > 1 else 30., int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print('%d cases, %d mismatches' % (n_cases, n_bad))
    sys.exit(1 if n_bad else 0)
