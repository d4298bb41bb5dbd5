"""Single-launch factorisation (csrc/dag.hpp): one persistent kernel walks the tile DAG of the blocked Cholesky with
dependency counters in HBM.  It is the default route of a single-object fit from 26 tile columns on (N > 3200) and of
batched fits from 16; the launch-per-step route
(CGP_NO_DAG=1) computes the same tiles with a different association of the sums, so the two agree to rounding
(tolerances below), each of them is run-to-run bit-reproducible, and both agree with the oracle."""
import numpy as np
import pytest

from oracle import cosmogp_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _problem(n, m, seed, dim=1):
    rng = np.random.default_rng(seed)
    if dim == 1:
        x = np.sort(rng.uniform(0., 0.15 * n, n))
        xn = rng.uniform(0., 0.15 * n, m) if m else None
        y = np.sin(0.05 * x) + 0.3 * np.cos(0.31 * x) + 0.1 * rng.standard_normal(n)
    else:
        s_ = 24. * np.sqrt(n)
        x = rng.uniform(0., s_, (n, 2))
        xn = rng.uniform(0., s_, (m, 2)) if m else None
        y = np.sin(x[:, 0] / 150.) * np.cos(x[:, 1] / 170.) + 0.1 * rng.standard_normal(n)
    return x, y, np.full(n, 0.1), xn


def _fit(env, monkeypatch, dim, x, y, e, xn, reps=1):
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    for k in ('CGP_NO_DAG', 'CGP_NO_GRAPH', 'CGP_DAG_DEPTH', 'CGP_DAG_GRID', 'CGP_LOOKAHEAD', 'CGP_DAG_MIN_NB'):
        monkeypatch.delenv(k, raising=False)
    # cgp_fit takes the single launch from 26 tile columns on by default (below, the launch-per-step graph is faster); these
    # tests run it on every size
    monkeypatch.setenv('CGP_DAG_MIN_NB', '2')
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = _lib.Device(0)
    kp = K.kernel_params_1d([1., 2.]) if dim == 1 else K.kernel_params_2d([1., 110., 90., 300.])
    dev.set_data(dim, x, y)
    outs = [dev.fit(kp, K.diag_term(len(y), e, 0., 0.), xn, None, 0.) for _ in range(reps)]
    dev.close()
    return outs


def _close(a, b, tol):
    assert abs(a[0] - b[0]) <= tol * abs(b[0]) and abs(a[1] - b[1]) <= tol * abs(b[1])
    if a[2] is not None:
        np.testing.assert_allclose(a[2], b[2], rtol=0, atol=tol * 100)
        np.testing.assert_allclose(a[3], b[3], rtol=0, atol=tol * 100)


@pytest.mark.parametrize('n,m,dim', [(129, 0, 1), (256, 5, 1), (300, 40, 1), (640, 0, 2), (1000, 333, 1), (2048, 100, 2),
                                     (3000, 1, 1), (4096, 1024, 1)])
def test_dag_equals_launch_route_and_oracle(monkeypatch, n, m, dim):
    x, y, e, xn = _problem(n, m, 100 + n, dim)
    a = _fit({}, monkeypatch, dim, x, y, e, xn, reps=3)
    b = _fit({'CGP_NO_DAG': '1'}, monkeypatch, dim, x, y, e, xn)[0]
    # replaying the captured launch gives the same bits: the sums of every tile have a fixed order whatever the
    # interleaving of the workgroups
    for o in a[1:]:
        assert o[0] == a[0][0] and o[1] == a[0][1]
        if m:
            assert np.array_equal(o[2], a[0][2]) and np.array_equal(o[3], a[0][3])
    _close(a[0], b, 1e-12)
    if n <= 1000:
        hyp = [1., 2.] if dim == 1 else [1., 110., 90., 300.]
        kern = orc.rbf_kernel_1d if dim == 1 else orc.rbf_kernel_2d
        Kc = kern(x, hyp, y_err=e)
        if m:
            mu, var, ll = orc.fit_predict_cho(Kc, y, kern(x, hyp, new_x=xn), np.full(m, 1.))
            np.testing.assert_allclose(a[0][2], mu, rtol=0, atol=1e-9)
            np.testing.assert_allclose(a[0][3], var, rtol=0, atol=1e-9)
        else:
            ll = orc.log_likelihood_gp(y, x, 'RBF1D' if dim == 1 else 'RBF2D', hyp, 0., y_err=e, svd_method=False)[0]
        assert abs(a[0][0] - ll) <= 1e-11 * abs(ll)


def test_dag_eager_few_workgroups_and_depths(monkeypatch):
    """Eager launch = captured launch bit for bit; a grid of 3 workgroups (one leaf lane, two workers) and of 17 walk
    the same list (no assumption on residency or dispatch order); deeper far updates (CGP_DAG_DEPTH) re-associate sums."""
    x, y, e, xn = _problem(1500, 200, 7)
    ref = _fit({}, monkeypatch, 1, x, y, e, xn)[0]
    for env in ({'CGP_NO_GRAPH': '1'}, {'CGP_DAG_GRID': '3'}, {'CGP_DAG_GRID': '17'}):
        o = _fit(env, monkeypatch, 1, x, y, e, xn)[0]
        assert o[0] == ref[0] and o[1] == ref[1] and np.array_equal(o[2], ref[2]) and np.array_equal(o[3], ref[3])
    for depth in ('2', '4'):
        o = _fit({'CGP_DAG_DEPTH': depth}, monkeypatch, 1, x, y, e, xn)[0]
        _close(o, ref, 1e-12)


def test_dag_not_positive_definite(monkeypatch):
    """A non-positive pivot is reported like LAPACK's info and nothing hangs (NaNs flow through the counters)."""
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    monkeypatch.delenv('CGP_NO_DAG', raising=False)
    x = np.linspace(0., 5., 700)
    dev = _lib.Device(0)
    dev.set_data(1, x, np.sin(x))
    with pytest.raises(np.linalg.LinAlgError):
        dev.fit(K.kernel_params_1d([1., 2.]), np.zeros(700))
    # the handle stays usable
    ll = dev.fit(K.kernel_params_1d([1., 2.]), np.full(700, 0.01))[0]
    ref = orc.log_likelihood_gp(np.sin(x), x, 'RBF1D', [1., 2.], 0., y_err=np.full(700, 0.1), svd_method=False)[0]
    assert abs(ll - ref) <= 1e-9 * abs(ref)
    dev.close()


def test_dag_c3_size(monkeypatch):
    """BASELINE config[2] size through the single launch: equals the launch-per-step route to rounding."""
    x, y, e, xn = _problem(8192, 2500, 3, dim=2)
    a = _fit({}, monkeypatch, 2, x, y, e, xn)[0]
    b = _fit({'CGP_NO_DAG': '1'}, monkeypatch, 2, x, y, e, xn)[0]
    _close(a, b, 1e-11)


# ---- several objects per launch (the batched entry points from 16 tile columns on; forced here for small objects) ------------

def _batch(env, monkeypatch, xs, ys, es, xn, chunks=None):
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    for k in ('CGP_NO_DAG', 'CGP_NO_DAG_BATCH', 'CGP_DAG_BATCH_MIN_NB', 'CGP_DAG_BMAX', 'CGP_DAG_GRID', 'CGP_NO_GRAPH', 'CGP_DAG_BLAZY'):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = _lib.Device(0)
    kp = K.kernel_params_1d([1., 2.])
    chunks = chunks or [list(range(len(xs)))]
    ll, info, mu, var = [], [], [], []
    for c in chunks:
        o = dev.batch_fit(1, [xs[i] for i in c], [ys[i] for i in c], [K.diag_term(len(ys[i]), es[i], 0., 0.) for i in c], kp, xn, None, 0.)
        ll.append(o[0]); info.append(o[1]); mu.append(o[2]); var.append(o[3])
    dev.close()
    return np.concatenate(ll), np.concatenate(info), np.concatenate(mu), np.concatenate(var)


def test_dag_batch_equals_launch_route_oracle_and_any_chunking(monkeypatch):
    """Seven ragged objects (300..640 points, padded to 640 = 5 tile columns) through cgp_batch_fit on the single-launch
    route: equal to the launch-per-step route to rounding and to the oracle; bit-identical however the batch is cut into
    calls and launches (CGP_DAG_BMAX = objects per launch), with 3 workgroups (fewer workers than objects need) and eagerly."""
    rng = np.random.default_rng(5)
    ns = [640, 300, 513, 640, 384, 601, 450]
    probs = [_problem(n, 0, 900 + i) for i, n in enumerate(ns)]
    xs, ys, es = [p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs]
    xn = rng.uniform(0., 40., 70)
    force = {'CGP_DAG_BATCH_MIN_NB': '2'}
    ref = _batch(force, monkeypatch, xs, ys, es, xn)
    steps = _batch({'CGP_NO_DAG_BATCH': '1'}, monkeypatch, xs, ys, es, xn)
    assert not ref[1].any() and not steps[1].any()
    np.testing.assert_allclose(ref[0], steps[0], rtol=1e-12)
    np.testing.assert_allclose(ref[2], steps[2], rtol=0, atol=1e-10)
    np.testing.assert_allclose(ref[3], steps[3], rtol=0, atol=1e-10)
    for i in (1, 5):
        Kc = orc.rbf_kernel_1d(xs[i], [1., 2.], y_err=es[i])
        mu, var, ll = orc.fit_predict_cho(Kc, ys[i], orc.rbf_kernel_1d(xs[i], [1., 2.], new_x=xn), np.full(len(xn), 1.))
        assert abs(ref[0][i] - ll) <= 1e-11 * abs(ll)
        np.testing.assert_allclose(ref[2][i], mu, rtol=0, atol=1e-9)
        np.testing.assert_allclose(ref[3][i], var, rtol=0, atol=1e-9)
    variants = [dict(force, CGP_DAG_BMAX='2'), dict(force, CGP_DAG_BMAX='3', CGP_DAG_GRID='5'), dict(force, CGP_NO_GRAPH='1')]
    for env in variants:
        got = _batch(env, monkeypatch, xs, ys, es, xn)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), env
    # the objects one by one: every object alone has 3..5 tile columns of its own (its padding differs from the batch's), so
    # only objects of the batch's full width are bit-identical to their solo run
    solo = _batch(force, monkeypatch, xs, ys, es, xn, chunks=[[i] for i in range(len(ns))])
    for i, n in enumerate(ns):
        if n > 512:
            assert solo[0][i] == ref[0][i] and np.array_equal(solo[2][i], ref[2][i]) and np.array_equal(solo[3][i], ref[3][i])
        else:
            assert abs(solo[0][i] - ref[0][i]) <= 1e-12 * abs(ref[0][i])
    cut = _batch(force, monkeypatch, xs, ys, es, xn, chunks=[[0, 3, 5], [1], [2, 4, 6]])
    # (a chunk of narrower objects is padded less: same numbers to rounding there, bit-identical where the width is the same)
    assert np.array_equal(cut[0][:3], ref[0][[0, 3, 5]]) and np.array_equal(cut[2][:3], ref[2][[0, 3, 5]])


def test_dag_batch_lazy_and_eager_schedules_bit_identical(monkeypatch):
    """The batched route picks the task list by the launch's work (api.hip, dag_list: work-bound launches get the laziest list --
    window 1, lead 1, 16 panels per far update --, chain-bound ones window 2, lead 2, 8 panels).  The grouping of a tile's panels
    into tasks must not change a bit (a task boundary is an exact store and reload): six objects of 1400..1536 points (12 tile
    columns) with 300 prediction points under both lists (CGP_DAG_BLAZY), every output compared bitwise; and the gradient /
    leave-one-out layout likewise."""
    rng = np.random.default_rng(15)
    ns = [1536, 1400, 1500, 1536, 1450, 1530]
    probs = [_problem(n, 0, 300 + i) for i, n in enumerate(ns)]
    xs, ys, es = [p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs]
    xn = rng.uniform(0., 40., 300)
    force = {'CGP_DAG_BATCH_MIN_NB': '2'}
    eager = _batch(dict(force, CGP_DAG_BLAZY='0'), monkeypatch, xs, ys, es, xn)
    lazy = _batch(dict(force, CGP_DAG_BLAZY='1'), monkeypatch, xs, ys, es, xn)
    auto = _batch(force, monkeypatch, xs, ys, es, xn)
    assert not eager[1].any()
    assert all(np.array_equal(a, b) for a, b in zip(eager, lazy))
    assert all(np.array_equal(a, b) for a, b in zip(eager, auto))
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    kp = K.kernel_params_1d([1., 2.])
    outs = []
    for flag in ('0', '1'):
        monkeypatch.setenv('CGP_DAG_BATCH_MIN_NB', '2')
        monkeypatch.setenv('CGP_DAG_BLAZY', flag)
        dev = _lib.Device(0)
        dev.batch_fit(1, xs, ys, [K.diag_term(len(y), e, 0., 0.) for y, e in zip(ys, es)], kp, None, None, 0.)
        g = dev.batch_fit_grad_resident(kp)
        loo = dev.batch_loo(kp)
        # (entries of the leave-one-out arrays beyond an object's own length are undefined)
        outs.append([np.asarray(v) for v in g] + [np.concatenate([loo[q][i, :ns[i]] for i in range(len(ns))]) for q in range(3)] + [loo[3]])
        dev.close()
    assert all(np.array_equal(a, b) for a, b in zip(*outs))


def test_dag_batch_reports_the_failing_object(monkeypatch):
    """A non-positive-definite object inside a batch on the single-launch route: its info word is set (LAPACK's leading
    minor), the other objects of the launch are untouched."""
    probs = [_problem(384, 0, 70 + i) for i in range(4)]
    xs, ys, es = [p[0] for p in probs], [p[1] for p in probs], [p[2].copy() for p in probs]
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    monkeypatch.setenv('CGP_DAG_BATCH_MIN_NB', '2')
    dev = _lib.Device(0)
    kp = K.kernel_params_1d([1., 2.])
    diags = [K.diag_term(384, e, 0., 0.) for e in es]
    good = dev.batch_fit(1, xs, ys, diags, kp, None, None, 0.)
    diags[2] = diags[2].copy()
    diags[2][200] = -5.0                       # K[200, 200] = 1 - 5 < 0
    bad = dev.batch_fit(1, xs, ys, diags, kp, None, None, 0.)
    dev.close()
    assert not good[1].any()
    assert bad[1][2] == 201 and not bad[1][[0, 1, 3]].any()
    assert np.array_equal(bad[0][[0, 1, 3]], good[0][[0, 1, 3]])


def test_dag_batch_default_route_at_17_tile_columns(monkeypatch):
    """The default configuration: three objects of 2100 / 2150 / 2176 points (17 tile columns: the batched entry point takes
    the single-launch route by itself) with a prediction grid, against the launch-per-step route and the oracle; the sweep of
    hyper-parameter candidates (cgp_sweep_loglik) over one of them against its candidates evaluated alone."""
    ns = [2100, 2150, 2176]
    probs = [_problem(n, 0, 4100 + i) for i, n in enumerate(ns)]
    xs, ys, es = [p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs]
    xn = np.random.default_rng(2).uniform(0., 300., 150)
    dag = _batch({}, monkeypatch, xs, ys, es, xn)
    steps = _batch({'CGP_NO_DAG_BATCH': '1'}, monkeypatch, xs, ys, es, xn)
    assert not dag[1].any()
    np.testing.assert_allclose(dag[0], steps[0], rtol=1e-12)
    np.testing.assert_allclose(dag[2], steps[2], rtol=0, atol=1e-10)
    np.testing.assert_allclose(dag[3], steps[3], rtol=0, atol=1e-10)
    Kc = orc.rbf_kernel_1d(xs[1], [1., 2.], y_err=es[1])
    mu, var, ll = orc.fit_predict_cho(Kc, ys[1], orc.rbf_kernel_1d(xs[1], [1., 2.], new_x=xn), np.full(len(xn), 1.))
    assert abs(dag[0][1] - ll) <= 1e-11 * abs(ll)
    np.testing.assert_allclose(dag[2][1], mu, rtol=0, atol=1e-9)
    np.testing.assert_allclose(dag[3][1], var, rtol=0, atol=1e-9)
    # sweep: 5 candidates in one call = each candidate in a call of its own, bit for bit
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    dev = _lib.Device(0)
    dev.set_data(1, xs[2], ys[2])
    diag = K.diag_term(ns[2], es[2], 0., 0.)
    kps = np.array([K.kernel_params_1d([a, l]) for a, l in ((1., 2.), (0.7, 3.), (1.3, 1.5), (0.9, 2.5), (1.1, 4.))])
    ll5, info5 = dev.sweep_loglik(kps, diag)
    alone = np.array([dev.sweep_loglik(kps[i:i + 1], diag)[0][0] for i in range(5)])
    dev.close()
    assert not info5.any() and np.array_equal(ll5, alone)
    want = orc.log_likelihood_gp(ys[2], xs[2], 'RBF1D', [0.7, 3.], 0., y_err=es[2], svd_method=False)[0]
    assert abs(ll5[1] - want) <= 1e-11 * abs(want)


# ---- gradient / leave-one-out layout ([K; residual block; identity rows -> W = L^-T]) on the single-launch route ----------------

def _grad_and_loo(env, monkeypatch, xs, ys, es):
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    for k in ('CGP_NO_DAG', 'CGP_NO_DAG_BATCH', 'CGP_DAG_BATCH_MIN_NB', 'CGP_DAG_MIN_NB', 'CGP_DAG_GRID', 'CGP_DAG_BMAX'):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = _lib.Device(0)
    kp = K.kernel_params_1d([1., 2.], 0.03)
    diags = [K.diag_term(len(y), e, 0., 0.) for y, e in zip(ys, es)]
    dev.set_data(1, xs[0], ys[0])
    single = dev.fit_grad(kp, diags[0])
    dev.batch_set_data(1, xs, ys, diags)
    bgrad = dev.batch_fit_grad_resident(kp)
    loo = dev.batch_loo(kp)
    dev.close()
    return single, bgrad, loo


def test_dag_gradient_and_loo_layout(monkeypatch):
    """cgp_fit_grad, cgp_batch_fit_grad and cgp_batch_loo factor [K; residual block; identity rows] in one sweep; the identity
    row tiles are triangular (row tile t is zero left of column tile t), which the task list honours (no tasks for the zero
    tiles, counters start at t).  Single-launch route (forced for these 3-5 tile-column objects, with 5 workgroups too)
    against the launch-per-step route and the oracle's gradient."""
    ns = [600, 385, 640]
    probs = [_problem(n, 0, 5200 + i) for i, n in enumerate(ns)]
    xs, ys, es = [p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs]
    force = {'CGP_DAG_MIN_NB': '2', 'CGP_DAG_BATCH_MIN_NB': '2'}
    ref = _grad_and_loo({'CGP_NO_DAG': '1'}, monkeypatch, xs, ys, es)
    for env in (force, dict(force, CGP_DAG_GRID='5', CGP_DAG_BMAX='2')):
        got = _grad_and_loo(env, monkeypatch, xs, ys, es)
        assert abs(got[0][0] - ref[0][0]) <= 1e-12 * abs(ref[0][0])
        np.testing.assert_allclose(got[0][2], ref[0][2], rtol=1e-9, atol=1e-9)
        assert not got[1][1].any()
        np.testing.assert_allclose(got[1][0], ref[1][0], rtol=1e-12)
        np.testing.assert_allclose(got[1][2], ref[1][2], rtol=1e-9, atol=1e-9)
        assert not got[2][3].any()
        for i, n in enumerate(ns):
            for q in range(3):
                np.testing.assert_allclose(got[2][q][i, :n], ref[2][q][i, :n], rtol=1e-9, atol=1e-10)
    # the oracle's gradient for the single object (moment sums -> chain rule on the host)
    from cosmogp_amd import kernel as K
    grad = K.gradient_from_sums_1d([1., 2.], 0.03, ref[0][2])
    want = orc.log_likelihood_gradient(ys[0], xs[0], 'RBF1D', [1., 2.], 0.03, y_err=es[0])
    np.testing.assert_allclose(grad, want, rtol=1e-7, atol=1e-7)


def test_dag_gradient_default_route_at_4096(monkeypatch):
    """Default thresholds: a 4096-point object's gradient and leave-one-out sweep take the single launch by themselves."""
    x, y, e, _ = _problem(4096, 0, 77)
    got = _grad_and_loo({}, monkeypatch, [x], [y], [e])
    ref = _grad_and_loo({'CGP_NO_DAG': '1'}, monkeypatch, [x], [y], [e])
    assert abs(got[0][0] - ref[0][0]) <= 1e-12 * abs(ref[0][0])
    np.testing.assert_allclose(got[0][2], ref[0][2], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(got[1][2], ref[1][2], rtol=1e-8, atol=1e-8)
    for q in range(3):
        np.testing.assert_allclose(got[2][q][0], ref[2][q][0], rtol=1e-8, atol=1e-9)


def test_dag_give_up_is_reported_on_first_and_replayed_calls(monkeypatch):
    """A bounded spin that gives up (forced: CGP_DAG_TIMEOUT_MS=0 makes the first unsatisfied poll of any lane give up) must
    come back as an error of the call -- on the eager / capturing first call of a shape, on the graph REPLAYS after it (the
    launch's own control words are cleared by every launch; the handle's sticky words keep the first give-up), and on the
    sweep entry point, which shares the route -- never as CGP_OK with unfinished numbers.  (round-2 advisor finding)"""
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    x, y, e, _ = _problem(2048, 0, 77, 1)
    for k in ('CGP_NO_DAG', 'CGP_NO_GRAPH', 'CGP_DAG_GRID', 'CGP_NO_DAG_BATCH'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('CGP_DAG_MIN_NB', '2')
    monkeypatch.setenv('CGP_DAG_BATCH_MIN_NB', '2')
    monkeypatch.setenv('CGP_DAG_TIMEOUT_MS', '0')
    dev = _lib.Device(0)
    kp = K.kernel_params_1d([1., 2.])
    dev.set_data(1, x, y)
    dg = K.diag_term(len(y), e, 0., 0.)
    for call in range(3):               # capture, replay, replay
        with pytest.raises(_lib.HipError, match='gave up'):
            dev.fit(kp, dg, None, None, 0.)
    kps = np.stack([K.kernel_params_1d([1., 2.]), K.kernel_params_1d([0.9, 2.5]), K.kernel_params_1d([1.1, 1.5])])
    for call in range(2):
        with pytest.raises(_lib.HipError, match='gave up'):
            dev.sweep_loglik(kps, dg)
    dev.close()
    # the same handle type with the default bound: same inputs, a result
    monkeypatch.delenv('CGP_DAG_TIMEOUT_MS')
    dev = _lib.Device(0)
    dev.set_data(1, x, y)
    ll = dev.fit(kp, dg, None, None, 0.)[0]
    ll2, info = dev.sweep_loglik(kps, dg)
    dev.close()
    assert np.isfinite(ll) and info[0] == 0 and abs(ll2[0] - ll) <= 1e-11 * abs(ll)


def _tool(name):
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', name + '.py')
    spec = importlib.util.spec_from_file_location('cgp_tool_' + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dag_randomised_stress_subset(monkeypatch):
    """A seeded ~12-s pass of tools/dag_stress.py: random sizes / batch sizes / prediction grids / grids of 3, 9, 40 and all
    workgroups / objects per launch, every entry point that takes the single-launch route (fit, gradient, batched fit,
    leave-one-out), each result against the launch-per-step route.  (The full 150-s run of round 2 -- 3287 cases -- stays a
    tool; this subset is what a compiler bump must keep green.)"""
    for k in ('CGP_NO_DAG', 'CGP_DAG_GRID', 'CGP_DAG_MIN_NB', 'CGP_DAG_BATCH_MIN_NB', 'CGP_DAG_BMAX', 'CGP_NO_GRAPH'):
        monkeypatch.delenv(k, raising=False)
    cases, bad, worst = _tool('dag_stress').run_stress(12., seed=20263, sizes=(130, 257, 400, 640, 900, 1300, 2050, 2600), verbose=True)
    assert cases >= 20 and bad == 0, (cases, bad, worst)


def test_dag_concurrent_launches_subset(monkeypatch):
    """Four host threads, four handles, ~6 s: persistent launches of different streams compete for the CUs (a launch may run with
    only part of its workgroups resident), every result bit-identical to the same call done alone, and no bounded spin gives
    up (tools/dag_concurrent_stress.py)."""
    for k in ('CGP_NO_DAG', 'CGP_DAG_GRID', 'CGP_DAG_BMAX', 'CGP_NO_GRAPH', 'CGP_DAG_TIMEOUT_MS'):
        monkeypatch.delenv(k, raising=False)
    count, bad, errors = _tool('dag_concurrent_stress').run_concurrent(6., 4, sizes=(2600, 1300, 900, 2050))
    assert not errors and not any(bad) and min(count) >= 3, (count, bad, errors)


def test_mixed_entry_points_concurrently_subset(monkeypatch):
    """Six host threads, a handle each, ~5 s, DIFFERENT entry points at once: a pipelined cgp_batch_fit, packed objects of tens
    of points, a single-launch fit, 512 light curves per call (four diagonal blocks per workgroup), leave-one-out pulls and the
    eigen route; every result bit-identical to the same call done alone (tools/mixed_concurrent_stress.py)."""
    for k in ('CGP_NO_DAG', 'CGP_DAG_GRID', 'CGP_NO_GRAPH', 'CGP_DAG_TIMEOUT_MS', 'CGP_PIPE_CHUNK'):
        monkeypatch.delenv(k, raising=False)
    names, count, bad, errors = _tool('mixed_concurrent_stress').run_mixed(5.)
    assert not errors and not any(bad) and min(count) >= 2, (names, count, bad, errors)
