"""CPU tests of the host-side mirror (no device calls): scalar preparation, mean function,
initial guess, lazy lists, sharding helpers -- against the golden vectors / oracle."""
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle import cosmogp_oracle as orc
import cosmogp_amd
from cosmogp_amd import kernel as K
from cosmogp_amd import multigpu
from cosmogp_amd.gaussian_process import _LazyList


def test_kernel_params_match_reference_expressions():
    h = np.array([0.5, 3.1, 2.2, 1.7])
    kp = K.kernel_params_2d(h)
    vi = orc.inv_l_2d(h)
    assert kp[0] == 1.0 and kp[1] == h[0] ** 2
    assert np.array_equal(kp[2:6], vi.ravel()) and kp[6] == 0.
    assert K.kernel_params_2d(h, nugget=0.3)[6] == 0.3 ** 2
    assert K.kernel_params_2d(h, sigma2_in_train=True)[0] == h[0] ** 2
    kp1 = K.kernel_params_1d([0.7, 6.5])
    assert kp1[0] == 0.7 ** 2 and kp1[1] == 0.7 ** 2 and kp1[2] == 6.5 ** 2 and kp1.shape == (7,)
    # (y_err^2 + floor^2) + nugget^2: resident vector + kp[6] reproduce the reference's order
    e = np.array([0.1, 0.2, 0.3])
    assert np.array_equal(K.diag_term(3, e, 0.01, 0.) + K.kernel_params_1d([1., 1.], 0.03)[6],
                          K.diag_term(3, e, 0.01, 0.03))


def test_diag_term_order():
    e = np.array([0.1, 0.2, 0.3])
    assert np.array_equal(K.diag_term(3, e, 0.01, 0.03), e ** 2 + 0.01 ** 2 + 0.03 ** 2)
    assert np.array_equal(K.diag_term(3, None, 0., 0.5), np.full(3, 0.25))
    assert np.array_equal(K.diag_term(3, e, 0.01, 0.03), orc.diag_noise(3, e, 0.01, 0.03))


def test_init_rbf_golden():
    g = load_golden('init_rbf')
    xs = [g['x1_%d' % i] for i in range(int(g['n1']))]
    ys = [g['y1_%d' % i] for i in range(int(g['n1']))]
    np.testing.assert_allclose(cosmogp_amd.init_rbf(xs, ys), g['guess_1d'], rtol=1e-15)
    Xs = [g['x2_%d' % i] for i in range(int(g['n2']))]
    Ys = [g['y2_%d' % i] for i in range(int(g['n2']))]
    np.testing.assert_allclose(cosmogp_amd.init_rbf(Xs, Ys), g['guess_2d'], rtol=1e-15)


def test_constructor_matches_reference_state():
    g = load_golden('predict_mean_1d_n80')
    gp = cosmogp_amd.gaussian_process(g['y'], g['x'], y_err=g['e'], Mean_Y=g['mean_y'], Time_mean=g['mean_x'])
    np.testing.assert_allclose(gp.y0[0], g['y0'], rtol=1e-14)
    assert gp.N_sn == 1 and gp.nugget == 0. and gp.fit_nugget is False
    g2 = load_golden('loglik_2d')
    gp2 = cosmogp_amd.gaussian_process_2d(g2['y'], g2['x'], y_err=g2['e'])
    np.testing.assert_allclose(gp2.hyperparameters, g2['init_hyp'], rtol=1e-15)
    with pytest.raises(AssertionError):
        cosmogp_amd.gaussian_process(g['y'], g['x'], kernel='MATERN')
    gp3 = cosmogp_amd.gaussian_process(g['y'], g['x'])
    assert np.array_equal(gp3.y_err[0], np.zeros(80)) and np.array_equal(gp3.y0, np.zeros(1))


def test_return_mean_vs_oracle():
    rng = np.random.default_rng(3)
    x = np.sort(rng.uniform(0, 10, 30))
    y = np.sin(x)
    tm = np.linspace(-1, 11, 25)
    my = np.cos(tm)
    xn = np.linspace(0, 10, 7)
    for kw in (dict(), dict(mean_y=my, mean_xaxis=tm), dict(mean_y=my, mean_xaxis=tm, diff=0.3)):
        np.testing.assert_array_equal(cosmogp_amd.return_mean(y, x, **kw), orc.return_mean(y, x, **kw))
        np.testing.assert_array_equal(cosmogp_amd.return_mean(y, x, new_x=xn, **kw),
                                      orc.return_mean(y, x, new_x=xn, **kw))
    # 2D mean: the reference calls bisplrep(task=1) with no prior task=0 call (mean.py:58-59), which SciPy 1.15 rejects:
    # the oracle (a faithful restatement) raises, so this branch has NO reference output (parity unpinned).  The package
    # tries the same call, then fits the same smoothing spline from scratch (task=0) and evaluates it in one device call.
    import scipy.interpolate as inter
    from cosmogp_amd import mean as cmean
    from fake_device import FakeDevice
    X = rng.uniform(0, 5, (40, 2))
    Y = np.sin(X[:, 0]) + X[:, 1]
    g = np.linspace(0, 5, 7)
    XM = np.array([[a, b] for a in g for b in g])
    YM = np.sin(XM[:, 0]) + XM[:, 1]
    with pytest.raises(Exception):
        orc.return_mean(Y, X, mean_y=YM, mean_xaxis=XM)
    dev = FakeDevice()
    got = cmean.interpolate_mean_2d(XM, YM, X, device=dev)
    assert cmean.LAST_BISPLREP_TASK in (0, 1) and dev.calls == ['bispev:40']          # ONE evaluation call for all points
    tck = inter.bisplrep(XM[:, 0], XM[:, 1], YM, task=0)
    np.testing.assert_allclose(got, [inter.bisplev(p[0], p[1], tck) for p in X], rtol=1e-13)


def test_lazy_list():
    calls = []

    def make(i):
        calls.append(i)
        return np.full((2, 2), float(i))

    ll = _LazyList(3, make)
    assert len(ll) == 3 and calls == []
    assert ll[1][0, 0] == 1. and ll[1][1, 1] == 1. and calls == [1]
    assert ll[-1][0, 0] == 2.
    assert [a[0, 0] for a in ll] == [0., 1., 2.] and sorted(calls) == [0, 1, 2]
    with pytest.raises(IndexError):
        ll[3]


def test_shard_helpers():
    for n, w in ((1024, 8), (10, 4), (3, 8), (0, 2)):
        blocks = [multigpu.shard_bounds(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1
    costs = np.array([512, 100, 300, 512, 64, 700, 50, 90]) ** 3.
    parts = multigpu.shard_by_cost(costs, 3)
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [costs[p].sum() for p in parts]
    assert max(loads) <= 700 ** 3. + 1e-9 + min(loads)


def test_empty_inputs_need_no_device():
    """Empty point sets give empty matrices (what NumPy broadcasting gives in the reference), without
    touching the GPU."""
    x = np.zeros(0)
    assert cosmogp_amd.rbf_kernel_1d(x, [1., 1.]).shape == (0, 0)
    assert cosmogp_amd.rbf_kernel_1d(np.linspace(0, 1, 5), [1., 1.], new_x=x).shape == (0, 5)
    assert cosmogp_amd.rbf_kernel_2d(np.zeros((0, 2)), [1., 1., 1., 0.]).shape == (0, 0)
    assert orc.rbf_kernel_1d(x, [1., 1.]).shape == (0, 0)
    assert orc.rbf_kernel_1d(np.linspace(0, 1, 5), [1., 1.], new_x=x).shape == (0, 5)


def test_batch_chunks_respect_grid_and_memory_limits(monkeypatch):
    """Batched launch sequences carry the object index in gridDim.z (<= 65535) and allocate
    (Np + round_up(m + 1, 128)) x Np doubles per object: both bound the chunk size."""
    import cosmogp_amd
    import importlib
    gpm = importlib.import_module('cosmogp_amd.gaussian_process')
    from fake_device import FakeDevice
    ys = [np.zeros(5)] * 70000
    xs = [np.arange(5.)] * 70000
    gp = cosmogp_amd.gaussian_process_nobject(ys, xs, kernel='RBF1D', device=FakeDevice())
    chunks = gp._batch_chunks(list(range(70000)))
    assert max(len(c) for c in chunks) <= gpm.MAX_BATCH_OBJECTS and sum(len(c) for c in chunks) == 70000
    monkeypatch.setattr(gpm, 'BATCH_BYTES_LIMIT', 10 * (128 + 1024) * 128 * 8)
    small = gp._batch_chunks(list(range(100)), m=1000)          # 1001 -> 1024 tall rows per object
    assert max(len(c) for c in small) == 10
    assert max(len(c) for c in gp._batch_chunks(list(range(100)), m=0)) == 45


def test_batch_chunks_give_short_objects_buckets_of_their_own():
    """Objects of <= 32 / <= 64 points are fitted four / two per tile when EVERY object of a call is that short
    (csrc/api.hip, batch_fit_packed): the host side therefore gives each class its own launch sequence -- from 8
    objects of the class on; fewer stay with the other 128-point objects (one call less)."""
    import cosmogp_amd
    from fake_device import FakeDevice
    lens = [20] * 9 + [50] * 8 + [100] * 3 + [300] * 2
    gp = cosmogp_amd.gaussian_process_nobject([np.zeros(n) for n in lens], [np.arange(float(n)) for n in lens],
                                              kernel='RBF1D', device=FakeDevice())
    chunks = gp._batch_chunks(list(range(len(lens))))
    assert sorted(map(sorted, chunks)) == sorted([list(range(0, 9)), list(range(9, 17)), [17, 18, 19], [20, 21]])
    lens = [20] * 7 + [50] * 8 + [100] * 3            # seven of the shortest class: they join the plain 128-point bucket
    gp = cosmogp_amd.gaussian_process_nobject([np.zeros(n) for n in lens], [np.arange(float(n)) for n in lens],
                                              kernel='RBF1D', device=FakeDevice())
    chunks = gp._batch_chunks(list(range(len(lens))))
    assert sorted(map(sorted, chunks)) == sorted([list(range(0, 7)) + [15, 16, 17], list(range(7, 15))])
    # the gradient needs the identity rows of every object: no packing, one bucket per padded size
    assert sorted(map(sorted, gp._batch_chunks(list(range(len(lens))), grad=True))) == [list(range(18))]


def test_bench_file_fallback_communicator(tmp_path):
    """bench.py's stand-in for the case that the RCCL communicator cannot be created: barrier / all-gather / max over
    files, three processes."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'fc.py'
    script.write_text(textwrap.dedent('''
        import os, sys, numpy as np
        sys.path.insert(0, %r)
        import bench
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        c = bench.FileComm(r, w)
        c.barrier()
        g = c._allgather(np.full((2, 3), float(r)))
        assert g.shape == (w, 2, 3) and all((g[i] == i).all() for i in range(w)), g
        assert c.allreduce_max(10. + r) == 10. + w - 1
        c.close()
    ''' % root))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE='3',
                                                                       MASTER_PORT=str(40000 + os.getpid() % 20000)))
             for r in range(3)]
    try:
        assert [p.wait(timeout=60) for p in procs] == [0, 0, 0]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_bench_launch_ranks_end_to_end(tmp_path, capfd):
    """bench.py --gpus N run plainly starts its ranks itself (bench.launch_ranks): rank 0's single JSON line is passed through
    and the exit code is 0 when every rank succeeds; a rank that dies takes the others down (they would wait for it inside a
    collective for ever) and the launcher exits non-zero.  Driven here with stand-in ranks that only use the file
    communicator (CGP_BENCH_COMM=file is the only way bench.py itself ever uses it), three processes, no GPU."""
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'rank.py'
    script.write_text(textwrap.dedent('''
        import json, os, sys, time, numpy as np
        sys.path.insert(0, %r)
        import bench
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["CGP_BENCH_CHILD"] == "1" and os.environ["MASTER_ADDR"] == "127.0.0.1"
        c = bench.FileComm(r, w)
        c.barrier()
        if sys.argv[1] == "die" and r == 1:
            sys.exit(7)
        if sys.argv[1] == "die":
            time.sleep(60)                      # the survivors "hang in the collective": the launcher must end them
        worst = c.allreduce_max(float(r))
        if r == 0:
            print(json.dumps(dict(metric="gp_fits_per_sec", n_gpus=w, worst=worst)))
        c.close()
    ''' % root))
    sys.path.insert(0, root)
    import bench
    import json
    import time
    rc = bench.launch_ranks(3, script=str(script), argv=['ok'])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1 and json.loads(out[0]) == dict(metric='gp_fits_per_sec', n_gpus=3, worst=2.0)
    t0 = time.time()
    rc = bench.launch_ranks(3, script=str(script), argv=['die'])
    out = capfd.readouterr().out.strip()
    assert rc != 0 and out == '' and time.time() - t0 < 40


def test_replayed_traffic_constant_belongs_to_the_device_code_at_head():
    """`roofline.traffic` of the bench line is a replayed constant from a committed PMC pass (rocprofv3 cannot wrap the bench).  Every
    traffic file records the sha256 of the device sources it was collected on (tools/pmc_traffic.py, tools/pmc_pass_traffic.py ->
    bench.kernel_source_hash: kernels.hpp, rbfmath.hpp, dag.hpp, dag_tasks.hpp, smallchol.hpp); for every BASELINE configuration the
    newest file must belong to the code at HEAD -- a kernel change without re-collecting the passes (tools/final_pass_r04.sh 3)
    fails here, and the bench line's `roofline.traffic_source_current` says the same at run time."""
    import bench
    cur = bench.kernel_source_hash()
    for wl, dag, nobj in (('c3', True, 8), ('c2', True, 16), ('c5', True, 4), ('c4', False, 128)):
        path, rec = bench.traffic_file(wl, dag, nobj)
        assert path is not None and os.path.basename(path).startswith('r04_'), (wl, path)
        assert rec.get('kernel_sources_sha256') == cur, '%s was collected on other device sources: re-run tools/final_pass_r04.sh 3' % path
        assert rec['traffic_bytes_per_launch'] > 0
