"""Left-looking panel route of small objects (csrc/smallchol.hpp) against the default recursive launch-per-step route (the left-looking one is opt-in: CGP_LL=1) and
the oracle: ragged batches, 1D / 2D, first touch and assembled input.  python tools/ll_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
from oracle import cosmogp_oracle as orc

base = dict(os.environ)
worst = 0.
for dim in (1, 2):
    rng = np.random.default_rng(9 + dim)
    ns = [512, 300, 257, 640, 1000, 1500, 130, 1800, 512, 256]
    if dim == 1:
        xs = [np.sort(rng.uniform(0., 30., n)) for n in ns]
        ys = [np.sin(x / 3.) + 0.1 * rng.standard_normal(len(x)) for x in xs]
        xn = rng.uniform(0., 30., 19)
        hyp = [0.9, 2.5]
        kp = K.kernel_params_1d(hyp, 0.03)
        kern = orc.rbf_kernel_1d
    else:
        xs = [rng.uniform(0., 30., (n, 2)) for n in ns]
        ys = [np.sin(x[:, 0] / 3.) * np.cos(x[:, 1] / 4.) + 0.1 * rng.standard_normal(len(x)) for x in xs]
        xn = rng.uniform(0., 30., (19, 2))
        hyp = [1.0, 4.0, 5.0, 3.0]
        kp = K.kernel_params_2d(hyp, nugget=0.03)
        kern = orc.rbf_kernel_2d
    es = [rng.uniform(0.05, 0.2, n) for n in ns]
    diags = [K.diag_term(len(y), e, 0., 0.) for y, e in zip(ys, es)]
    outs = {}
    for name, env in (('ll', {'CGP_LL': '1'}), ('ll_asm', {'CGP_LL': '1', 'CGP_NO_FUSED_ASM': '1'}), ('rec', {})):
        os.environ.clear(); os.environ.update(base); os.environ.update(env)
        dev = _lib.Device(0)
        r = dev.batch_fit(dim, xs, ys, diags, kp, xn, None, 0.03 ** 2)
        r2 = dev.batch_fit(dim, xs, ys, diags, kp, None, None, 0.)
        dev.set_data(dim, xs[0], ys[0])
        one = dev.fit(kp, diags[0], xn, None, 0.03 ** 2)
        outs[name] = (r, r2, one)
        dev.close()
    a, b, c = outs['ll'], outs['ll_asm'], outs['rec']
    print('dim', dim, 'info', a[0][1].tolist())
    print('  ll vs assembled input: bit-identical', all(np.array_equal(u, v) for u, v in zip(a[0], b[0])) and np.array_equal(a[1][0], b[1][0]))
    print('  batch object 0 vs its single fit: bit-identical', a[0][0][0] == a[2][0] and np.array_equal(a[0][2][0], a[2][2]) and np.array_equal(a[0][3][0], a[2][3]))
    dl = np.max(np.abs(a[0][0] - c[0][0]) / np.abs(c[0][0]))
    dm = np.max(np.abs(a[0][2] - c[0][2])) / np.max(np.abs(c[0][2]))
    dv = np.max(np.abs(a[0][3] - c[0][3]))
    print('  ll vs recursive route: rel loglik %.2e, mu %.2e of max, var %.2e' % (dl, dm, dv))
    worst = max(worst, dl, dm, dv)
    for bq in (0, 4, 7):
        Kc = kern(xs[bq], hyp, y_err=es[bq], nugget=0.03)
        H = kern(xs[bq], hyp, new_x=xn)
        amp = hyp[0] ** 2 if dim == 1 else 1.0
        m0, v0, l0 = orc.fit_predict_cho(Kc, ys[bq], H, np.full(19, amp + 0.03 ** 2))
        print('  object %d (n = %d) vs oracle: rel loglik %.2e, mu %.2e, var %.2e' % (bq, ns[bq], abs(l0 - a[0][0][bq]) / abs(l0), np.max(np.abs(m0 - a[0][2][bq])), np.max(np.abs(v0 - a[0][3][bq]))))
print('worst ll vs recursive', worst)
