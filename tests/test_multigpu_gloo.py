"""N > 1 path on CPU: world_size-2 gloo process group; objects are sharded over ranks with no
data-path collective and gathered once at the end (cosmogp_amd/multigpu.py).  The per-rank
compute is the oracle-backed FakeDevice (no GPU here); the GPU form of the same test is
tests/test_gpu_parity.py::test_sharded_equals_serial (single rank) and bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import cosmogp_amd
    from cosmogp_amd import multigpu
    from fake_device import FakeDevice
    from gloo_comm import GlooComm
    from conftest import load_golden

    comm = GlooComm()
    assert comm.world == world and comm.rank == rank
    assert comm.agree_any(rank == 1) and not comm.agree_any(False)
    assert comm.allreduce_max(float(rank)) == world - 1
    # raw gathers
    n_units = 5
    mine = comm.my_units(n_units)
    vals = comm.gather_units(np.array([10. * u for u in mine]), n_units)
    assert np.array_equal(vals, 10. * np.arange(n_units))
    items = [None] * n_units
    for u in mine:
        items[u] = np.arange(u + 1, dtype=float)
    full = comm.gather_unit_arrays(items, n_units)
    assert all(np.array_equal(full[u], np.arange(u + 1)) for u in range(n_units))

    # sharded n-object GP == the golden (reference) answer on every rank
    g = load_golden('predict_nobject_1d')
    n = int(g['nobj'])
    xs, ys, es = ([g['%s_%d' % (k, i)] for i in range(n)] for k in 'xye')
    dev = FakeDevice()
    gp = cosmogp_amd.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es, device=dev)
    gp.comm = comm
    gp.hyperparameters = g['hyp']
    gp.get_prediction(new_binning=g['xnew'], svd_method=False)
    for i in range(n):
        np.testing.assert_allclose(gp.Prediction[i], g['pred_%d' % i], rtol=0, atol=1e-10)
        np.testing.assert_allclose(gp.variance[i], g['covdiag_%d' % i], rtol=0, atol=1e-10)
    # each rank only computed its own block
    n_mine = len(comm.my_units(n))
    assert any(c in ('batch_fit:%d' % n_mine, 'fit') for c in dev.calls)
    assert 'batch_fit:%d' % n not in dev.calls or world == 1

    g = load_golden('loglik_1d')
    n = int(g['nobj'])
    xs, ys, es = ([g['%s_%d' % (k, i)] for i in range(n)] for k in 'xye')
    gp = cosmogp_amd.gaussian_process_nobject(ys, xs, kernel='RBF1D', y_err=es, device=FakeDevice())
    gp.comm = comm
    gp.nugget = float(g['nugget'])
    gp.compute_log_likelihood(g['hyp'], svd_method=False)
    assert gp.log_likelihood.shape == (1,)
    np.testing.assert_allclose(gp.log_likelihood, g['ll3_chol'], rtol=1e-11)
    # analytic gradient of the joint likelihood: every rank reduces its own objects, one gather of (loglik, grad) rows
    import importlib
    orc_ = importlib.import_module('oracle.cosmogp_oracle')
    grad = gp.compute_log_likelihood_gradient(g['hyp'])
    want = sum(orc_.log_likelihood_gradient(ys[i], xs[i], 'RBF1D', g['hyp'], float(g['nugget']), y_err=es[i], y_mean=0.)
               for i in range(n))
    np.testing.assert_allclose(grad, want[:2], rtol=1e-9)
    np.testing.assert_allclose(gp.log_likelihood, g['ll3_chol'], rtol=1e-11)
    # hyper-parameter sweep: candidates sharded over ranks, data resident, one gather
    table = np.array([[0.5, 6.], [0.6, 7.], [0.7, 8.], [0.8, 9.], [0.9, 10.]])
    gp1 = cosmogp_amd.gaussian_process(ys[0], xs[0], kernel='RBF1D', y_err=es[0], device=FakeDevice())
    got = multigpu.sweep_log_likelihood(gp1, table, comm)
    import importlib
    orc = importlib.import_module('oracle.cosmogp_oracle')
    ref = [orc.total_log_likelihood([ys[0]], [xs[0]], 'RBF1D', h, 0., [es[0]], [0.], svd_method=False)[0]
           for h in table]
    np.testing.assert_allclose(got, ref, rtol=1e-11)
    # one lock-step launch sequence per rank for its block of candidates, data uploaded once
    assert gp1.device.calls.count('set_data') == 1
    assert gp1.device.calls.count('sweep_loglik:%d' % len(comm.my_units(5))) == 1
    # a non-positive-definite object on ONE rank: every rank must leave the Cholesky route together (no rank may
    # sit in a gather its peer never enters) -- raise everywhere with svd_method=False ...
    xs2 = [np.linspace(0., 1., 6), np.linspace(0., 1., 7), np.linspace(0., 1., 5), np.linspace(0., 1., 8)]
    ys2 = [np.sin(3. * x) for x in xs2]
    es2 = [np.full(len(x), 0.1) for x in xs2]
    es2[3] = np.zeros(8)
    xs2[3] = np.array([0., 0., 0.2, 0.3, 0.5, 0.6, 0.8, 1.])        # duplicate point, no noise: singular K on rank 1
    gpb = cosmogp_amd.gaussian_process_nobject(ys2, xs2, kernel='RBF1D', y_err=es2, device=FakeDevice())
    gpb.comm = comm
    gpb.hyperparameters = np.array([1., 0.5])
    raised = False
    try:
        gpb.get_prediction(new_binning=np.linspace(0., 1., 4), svd_method=False)
    except np.linalg.LinAlgError:
        raised = True
    assert raised
    raised = False
    try:
        gpb.compute_log_likelihood(np.array([1., 0.5]), svd_method=False)
    except np.linalg.LinAlgError:
        raised = True
    assert raised
    # ... and fall back to the eigensolver route together with svd_method=True (forced through Cholesky first)
    import importlib
    gpm = importlib.import_module('cosmogp_amd.gaussian_process')
    gpm.SVD_VIA_CHOLESKY_WHEN_PD = True
    gpb.get_prediction(new_binning=np.linspace(0., 1., 4), svd_method=True)
    assert all(p is not None and len(p) == 4 for p in gpb.Prediction)
    ser = cosmogp_amd.gaussian_process_nobject(ys2, xs2, kernel='RBF1D', y_err=es2, device=FakeDevice())
    ser.hyperparameters = np.array([1., 0.5])
    ser.get_prediction(new_binning=np.linspace(0., 1., 4), svd_method=True)
    for i in range(4):
        np.testing.assert_allclose(gpb.Prediction[i], ser.Prediction[i], rtol=0, atol=1e-6)
    gpm.SVD_VIA_CHOLESKY_WHEN_PD = 'auto'
    comm.barrier()
    open(os.path.join(out_dir, 'ok_%d' % rank), 'w').write('ok')
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(str(tmp_path))) == ['ok_0', 'ok_1']


def test_single_process_fake_device_matches_golden():
    """Same host logic without a process group (batched path, lazy attributes)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import cosmogp_amd
    from fake_device import FakeDevice
    from conftest import load_golden
    g = load_golden('c1_predict_1d_n256')
    dev = FakeDevice()
    gp = cosmogp_amd.gaussian_process(g['y'], g['x'], y_err=g['e'], device=dev)
    gp.hyperparameters = g['hyp']
    gp.get_prediction(new_binning=g['xnew'], svd_method=False)
    np.testing.assert_allclose(gp.Prediction[0], g['pred'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(gp.variance[0], np.diag(g['cov']), rtol=0, atol=1e-10)
    n_fit = dev.calls.count('fit')
    np.testing.assert_allclose(gp.covariance_matrix[0], g['cov'], rtol=0, atol=1e-9)   # lazy, no re-fit
    assert dev.calls.count('fit') == n_fit
    gp.compute_log_likelihood(g['hyp'], svd_method=False)
    np.testing.assert_allclose(gp.log_likelihood, g['loglik'], rtol=1e-11)
    assert dev.calls.count('set_data') == 1            # data stayed resident
