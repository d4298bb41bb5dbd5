#!/usr/bin/env python
"""bench.py -- GP fits/s and single-fit fp64 Cholesky GF/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM -- the objects of ONE launch sequence: 8 distinct objects for c3, 16 for c2, 128 for c4, 4 candidates for c5
(`config.fits_per_step_per_gpu`) --: assemble K (RBF), blocked Cholesky with the cross matrix and the residual riding along
as tall rows, predictive mean + variance at M points, log-likelihood.  No host copies inside the
timed region (`value` is device-resident fits/s; the PCIe-inclusive rate is `roofline.api_fits_per_sec`, it is never `value`).

Workloads (seeds fixed):
  c3 (default at N=1)  2D RBF GP, N=8192, M=2500, single fit     -- the size BASELINE.json's metric names
  c2                   1D RBF GP, N=4096, M=1024, single fit
  c4                   1D light curves, N=512, M=19, 128 fits per GPU per step (1024 on 8 GPUs)
N > 1: one process per GPU, every rank fits its own objects (weak scaling, no data-path collective);
RCCL -- called directly inside libcosmogp_hip (cgp_comm_*), no PyTorch -- does the barrier, the
max-over-ranks of the timing and the ONE all-gather of the predictions.  `python bench.py --gpus N` run
plainly starts its N ranks itself (child processes, spawned before the parent touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the launcher's RANK / LOCAL_RANK /
WORLD_SIZE are used as they are.

Rank 0 prints ONE JSON line.  roofline = the dominant kernel (k_chol_dag, the persistent tile-DAG launch; k_gemm_nt on the
launch-per-step route of small objects; fp64 MFMA): algorithmic flops per launch / average launch duration measured with HIP
events on the kernel's own stream; roofline.single_fit / single_fit_n4096 = one fit at a time at N = 8192 / 4096.
Launched as a rank, config.rccl_ranks = ncclCommCount and extra.multi_gpu = BASELINE configs 4 and 5 over the ranks.
cpu_baseline = the CPU oracle (NumPy/SciPy restatement of the reference) on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPEC_FP64_MFMA_TFLOPS = 78.6      # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (AMD datasheet class figure)
SPEC_HBM_GBS = 8000.0


def kernel_source_hash():
    """sha256 over the device sources of the library: the PMC traffic files under profiles/ record it at collection time
    (tools/pmc_traffic.py, tools/pmc_pass_traffic.py), and the bench line says whether the replayed constant still belongs to
    the code that ran (`roofline.traffic_source_current`; tests/test_host_logic.py checks the default workload's file)."""
    import hashlib
    h = hashlib.sha256()
    # (the device code of every kernel a fit launches; api.hip -- host orchestration -- is left out: it changes with every
    # entry point added, its schedule constants are recorded in the bench line itself)
    for f in ('kernels.hpp', 'rbfmath.hpp', 'dag.hpp', 'dag_tasks.hpp', 'smallchol.hpp', 'potrf4.hpp'):
        with open(os.path.join(ROOT, 'cosmogp_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def traffic_file(workload, dag, nobj_pass):
    """(path, record) of the newest committed PMC traffic file of this workload, or (None, None)"""
    for rnd in ('r04', 'r03', 'r02', 'r01'):
        tpath = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s%s.json'
                             % (rnd, workload, ('_single' if nobj_pass == 1 else '_dagbatch') if dag else ''))
        if os.path.exists(tpath):
            return tpath, json.load(open(tpath))
    return None, None


def make_workload(name, rank=0, obj=0, nobj=128):
    from cosmogp_amd import kernel as K
    if name == 'c3':
        rng = np.random.default_rng(3 + 1000 * rank + 37 * obj)
        n, m = 8192, 2500
        X = rng.uniform(0., 2048., (n, 2))
        g = np.linspace(20., 2028., 50)
        Xn = np.array([[a, b] for a in g for b in g])
        y = np.sin(X[:, 0] / 150.) * np.cos(X[:, 1] / 170.) + 0.1 * rng.standard_normal(n)
        hyp = [1., 110., 110., 0.]
        return dict(dim=2, n=n, m=m, x=X, y=y, e=np.full(n, 0.1), xnew=Xn, hyp=hyp,
                    kp=K.kernel_params_2d(hyp), label='2D RBF GP N=8192 M=2500 (BASELINE config[2])')
    if name == 'c2':
        rng = np.random.default_rng(2 + 1000 * rank + 37 * obj)
        n, m = 4096, 1024
        x = np.sort(rng.uniform(0., 400., n))
        y = np.sin(0.05 * x) + 0.3 * np.cos(0.31 * x) + 0.1 * rng.standard_normal(n)
        hyp = [1., 2.]
        return dict(dim=1, n=n, m=m, x=x, y=y, e=np.full(n, 0.1),
                    xnew=np.random.default_rng(20 + 1000 * rank).uniform(0., 400., m), hyp=hyp,
                    kp=K.kernel_params_1d(hyp), label='1D RBF GP N=4096 M=1024 (BASELINE config[1])')
    if name == 'c4':
        B, n, m = int(nobj), 512, 19
        xs, ys, es = [], [], []
        for i in range(B):
            rng = np.random.default_rng(4000 + i + 1000 * rank)
            x = np.sort(rng.uniform(-12., 42., n))
            e = rng.uniform(0.02, 0.1, n)
            xs.append(x)
            es.append(e)
            ys.append(0.5 * np.sin(x / 8.) + e * rng.standard_normal(n))
        hyp = [0.5, 8.]
        return dict(dim=1, n=n, m=m, B=B, xs=xs, ys=ys, es=es, xnew=np.linspace(-12., 42., m), hyp=hyp,
                    kp=K.kernel_params_1d(hyp),
                    label='%d x 1D light-curve GPs N=512 M=19 per GPU per step (BASELINE config[3]: 1024 light curves; with the default '
                          '512 per pass x 2 handles all 1024 are in flight on the GPU)' % B)
    if name == 'c5':
        rng = np.random.default_rng(5)                      # same field on every rank, pairs differ
        n = 16384
        X = rng.uniform(0., 4096., (n, 2))
        y = np.sin(X[:, 0] / 300.) * np.cos(X[:, 1] / 340.) + 0.1 * rng.standard_normal(n)
        sig = np.logspace(-0.5, 0.5, 4)
        ell = np.logspace(1.5, 2.5, 8)
        pairs = [(s_, l_) for s_ in sig for l_ in ell]       # 32 (sigma, l) candidates
        mine = pairs[(rank * 4) % 32:(rank * 4) % 32 + 4]    # 4 per GPU (all 32 on 8 GPUs)
        return dict(dim=2, n=n, m=0, x=X, y=y, e=np.full(n, 0.1), xnew=None, sweep=mine,
                    hyp=[mine[0][0], mine[0][1], mine[0][1], 0.],
                    kp=K.kernel_params_2d([mine[0][0], mine[0][1], mine[0][1], 0.]),
                    label='2D RBF GP N=16384, log-likelihood sweep, 4 (sigma,l) pairs per GPU per step '
                          '(BASELINE config[4] / 8)')
    raise SystemExit('unknown workload %r' % name)


def _cpu_info():
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    blas, threads = 'unknown', os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        infos = [p for p in threadpool_info() if p.get('user_api') == 'blas']
        if infos:
            blas = ', '.join(sorted(set('%s %s (%s)' % (p.get('internal_api'), p.get('version'), p.get('threading_layer', p.get('architecture', '')))
                                        for p in infos)))
            threads = max(p.get('num_threads', 1) for p in infos)
    except Exception:
        pass
    return dict(cpu_model=model, physical_cores=int(phys), logical_cpus=int(os.cpu_count()), blas=blas), int(threads)


def _best_of(fn, max_reps=5, budget_s=8.0):
    """best wall time of up to max_reps runs of fn(), stopping once budget_s is spent (at least one run)."""
    best, spent, reps = np.inf, 0., 0
    while reps < max_reps and (reps == 0 or spent < budget_s):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best, spent, reps = min(best, dt), spent + dt, reps + 1
    return best, reps


def _c4_pool_fit(args):
    """one N=512 light-curve fit on ONE BLAS thread (worker of the process pool)"""
    from oracle import cosmogp_oracle as orc
    x, y, e, hyp, xnew = args
    Kc = orc.rbf_kernel_1d(x, hyp, y_err=e)
    H = orc.rbf_kernel_1d(x, hyp, new_x=xnew)
    mu, var, ll = orc.fit_predict_cho(Kc, y, H, np.full(len(xnew), hyp[0] ** 2))
    return float(ll)


def cpu_baseline(wl):
    """The oracle (NumPy/SciPy restatement of the reference) timed on the host cores, on a bounded sample of the
    same workload: warm-up, best of up to 5, all BLAS threads and ONE thread, BLAS vendor and
    CPU model reported; the batch workload also through a process pool (one fit per worker, one thread each)."""
    from oracle import cosmogp_oracle as orc
    from scipy import linalg as la
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    info, threads = _cpu_info()
    t_start = time.perf_counter()

    def one_thread(fn):
        if threadpool_limits is None:
            return None
        with threadpool_limits(limits=1, user_api='blas'):
            return _best_of(fn, 3, 4.0)

    if 'B' in wl:
        nfit = 4
        args = [(wl['xs'][b], wl['ys'][b], wl['es'][b], wl['hyp'], wl['xnew']) for b in range(wl['B'])]

        def serial():
            for b in range(nfit):
                _c4_pool_fit(args[b])
        serial()                                                   # warm-up
        best, reps = _best_of(serial, 5, 6.0)
        one = one_thread(serial)
        out = dict(value=nfit / best, unit='fits/s', cores=threads, kind='port',
                   sample='%d of the %d fits of one step (N=512, M=19), cho_factor/cho_solve form, one after the other; '
                          'best of %d after a warm-up' % (nfit, wl['B'], reps), seconds=round(best, 4))
        if one:
            out['one_thread'] = dict(value=nfit / one[0], unit='fits/s', cores=1, seconds=round(one[0], 4))
        # the fair CPU form of embarrassingly parallel fits: a process pool, one fit per worker, one BLAS thread each
        try:
            import multiprocessing as mp
            nw = min(info['physical_cores'], wl['B'])
            env_keep = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
            os.environ.update(OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
            with mp.get_context('spawn').Pool(nw) as pool:
                # (time-outs: a pool whose workers cannot start would respawn them forever)
                pool.map_async(_c4_pool_fit, args[:nw]).get(timeout=120)          # warm-up (imports, first touch)
                t0 = time.perf_counter()
                pool.map_async(_c4_pool_fit, args, chunksize=max(1, len(args) // (4 * nw))).get(timeout=120)
                dtp = time.perf_counter() - t0
            for k, v in env_keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            out['process_pool'] = dict(value=len(args) / dtp, unit='fits/s', workers=nw, threads_per_worker=1,
                                       sample='all %d fits of one step' % len(args), seconds=round(dtp, 3))
        except Exception as exc:                                   # a box without fork/spawn rights: report, do not die
            out['process_pool'] = dict(error=repr(exc))
        out.update(info)
        out['baseline_seconds_total'] = round(time.perf_counter() - t_start, 2)
        return out

    kern = orc.rbf_kernel_1d if wl['dim'] == 1 else orc.rbf_kernel_2d
    sweep = 'sweep' in wl
    n = wl['n']
    # bounded: the full-size CPU factorisation of C5 alone takes minutes -> a quarter of the points, N^3 law;
    # C2 / C3 run at full size with all threads
    ns_ = n // 4 if sweep else n
    scale = (n / float(ns_)) ** 3
    xs_, ys_, es_ = wl['x'][:ns_], wl['y'][:ns_], wl['e'][:ns_]
    kw = dict(sigma2_in_train=True) if sweep else {}
    stage, last = {}, {}

    def full_fit(x_=xs_, y_=ys_, e_=es_, record=True):
        t0 = time.perf_counter()
        Kc = kern(x_, wl['hyp'], y_err=e_, **kw)
        H = None if sweep else kern(x_, wl['hyp'], new_x=wl['xnew'])
        t1 = time.perf_counter()
        if sweep:
            c = la.cho_factor(Kc, lower=True)
            la.cho_solve(c, y_)
        else:
            orc.fit_predict_cho(Kc, y_, H, np.full(wl['m'], 1.0))
        t2 = time.perf_counter()
        if record and (not stage or t2 - t0 < stage['total']):
            stage.update(assembly=t1 - t0, factor_solve=t2 - t1, total=t2 - t0)
        last.update(assembly=t1 - t0, factor_solve=t2 - t1)

    nw_ = max(256, ns_ // 8)
    full_fit(xs_[:nw_], ys_[:nw_], es_[:nw_], record=False)        # warm-up (BLAS threads, page faults) at N/8
    best, reps = _best_of(full_fit, 5, 26.0)
    # a reduced sample is scaled stage by stage: assembly ~ N^2, factor / solve ~ N^3
    est = stage['assembly'] * (n / float(ns_)) ** 2 + stage['factor_solve'] * scale
    cap = ('' if threads >= info['physical_cores'] else
           '; BLAS threads = %d of %d physical cores: the thread cap of this OpenBLAS build, every thread it offers is used'
           % (threads, info['physical_cores']))
    out = dict(value=1.0 / est, unit='fits/s', cores=threads, kind='port', seconds=round(best, 3),
               sample=('%s at N=%d%s: assembly %.2f s + cho_factor/cho_solve%s %.2f s; best of %d after a warm-up%s'
                       % ('1 log-likelihood' if sweep else '1 full fit of the workload', ns_,
                          (' scaled to N=%d (assembly x %.0f, factorisation x %.0f)' % (n, (n / float(ns_)) ** 2, scale))
                          if scale > 1 else '', stage['assembly'], '' if sweep else '/predict', stage['factor_solve'], reps, cap)))
    # ONE BLAS thread: at a quarter of the size (the full size would take minutes), scaled by the N^3 law
    n1 = max(512, ns_ // 4)
    one = one_thread(lambda: full_fit(xs_[:n1], ys_[:n1], es_[:n1], record=False))
    if one:
        est1 = last['assembly'] * (n / float(n1)) ** 2 + last['factor_solve'] * (n / float(n1)) ** 3
        out['one_thread'] = dict(value=1.0 / est1, unit='fits/s', cores=1,
                                 sample='the same at N=%d (assembly %.2f s + factorisation %.2f s) scaled to N=%d '
                                        '(assembly x %.0f, factorisation x %.0f)'
                                        % (n1, last['assembly'], last['factor_solve'], n, (n / float(n1)) ** 2,
                                           (n / float(n1)) ** 3))
    if not sweep:
        # the reference's own algorithm (inv_matrix.py:23-25 explicit inverse, then dots) at half the size
        # (it takes ~3x the cho form), scaled by the N^3 law
        nr = n // 2
        t0 = time.perf_counter()
        Kc = kern(wl['x'][:nr], wl['hyp'], y_err=wl['e'][:nr])
        H = kern(wl['x'][:nr], wl['hyp'], new_x=wl['xnew'])
        t1 = time.perf_counter()
        inv = orc.cholesky_inverse(Kc)
        mu_ref = H @ (inv @ wl['y'][:nr])
        var_ref = 1.0 - np.einsum('ij,ij->i', H @ inv, H)
        dtr = time.perf_counter() - t0
        out['reference_form'] = dict(value=1.0 / ((t1 - t0) * 4.0 + (dtr - (t1 - t0)) * 8.0), unit='fits/s',
                                     seconds=round(dtr, 3),
                                     note='assembly + explicit inverse (cholesky, inv(L), inv_L.T @ inv_L) and dot '
                                          'products like cosmogp/inv_matrix.py:23-25, Gaussian_process.py:334-361 '
                                          '(variance diagonal only) at N=%d, scaled to N=%d (assembly x 4, the rest x 8)' % (nr, n),
                                     checksum=float(mu_ref[0] + var_ref[0]))
    out.update(info)
    out['baseline_seconds_total'] = round(time.perf_counter() - t_start, 2)
    return out


def multi_gpu_extras(dev, comm, rank, world, local_rank, args):
    """BASELINE configs 4 and 5 are defined over the GPUs of a node: under --gpus N every rank also times its share of
    them -- 1024 / N light curves (N = 512, M = 19) per pass through cgp_batch_fit's resident path, and the 32-candidate
    log-likelihood sweep at N = 16384 through cgp_sweep_loglik_sharded (each rank factors 32 / N candidates, ONE
    ncclAllGather inside the library) -- between barriers, max over ranks; whole-job rates go to rank 0's line."""
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    out = {}
    if comm is None:
        return out

    def timed(fn, reps):
        fn()                                  # warm-up (graph capture, workspaces)
        comm.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        comm.barrier()
        return comm.allreduce_max(time.perf_counter() - t0) / reps

    try:
        # ---- config 4: 1024 light curves over the ranks, contiguous blocks
        per = 1024 // world
        w4 = make_workload('c4', rank, nobj=per)
        d4 = _lib.Device(local_rank)
        diags = [K.diag_term(w4['n'], e, 0., 0.) for e in w4['es']]
        ll, info, mu, var = d4.batch_fit(1, w4['xs'], w4['ys'], diags, w4['kp'], w4['xnew'], None, 0.)
        assert not info.any()
        dt = timed(lambda: d4.bench_fit(4, stages=False), 3) / 4
        g = comm._allgather(np.stack([mu.ravel(), var.ravel()]))           # the one gather of the path
        out['c4'] = dict(fits_per_sec=round(world * per / dt, 1), fits_per_rank_per_pass=per, ms_per_pass=round(dt * 1e3, 4),
                         gathered_shape=list(g.shape), note='1024 x (N=512, M=19) over %d ranks, device-resident passes' % world)
        d4.close()
    except Exception as exc:                                               # reported, never fatal for the headline
        out['c4'] = dict(error=repr(exc)[:300])
    try:
        # ---- config 5: 32 (sigma, l) candidates at N = 16384 over the ranks
        w5 = make_workload('c5', 0)
        sig = np.logspace(-0.5, 0.5, 4)
        ell = np.logspace(1.5, 2.5, 8)
        kps = np.array([K.kernel_params_2d([s_, l_, l_, 0.], sigma2_in_train=True) for s_ in sig for l_ in ell])
        if comm.backend != 'rccl':
            raise RuntimeError('needs the RCCL communicator of the library (cgp_sweep_loglik_sharded)')
        dev.set_data(w5['dim'], w5['x'], w5['y'])
        dg = K.diag_term(w5['n'], w5['e'], 0., 0.)
        res = {}

        def sweep():
            res['ll'], res['info'] = dev.sweep_loglik_sharded(kps, dg, chunk=4)
        dt = timed(sweep, 1)
        assert not res['info'].any() and np.all(np.isfinite(res['ll']))
        out['c5'] = dict(candidates_per_sec=round(len(kps) / dt, 3), seconds_per_sweep=round(dt, 4), candidates=len(kps),
                         candidates_per_rank=-(-len(kps) // world),
                         note='32 log-likelihoods of the 2D N=16384 object over %d ranks (cgp_sweep_loglik_sharded)' % world)
    except Exception as exc:
        out['c5'] = dict(error=repr(exc)[:300])
    return {'multi_gpu': out}


def other_configs(local_rank):
    """The default line is BASELINE config[2]; the other configurations that fit one GPU -- config[1] (c2), config[3] (c4),
    config[4] (c5) -- are timed by child processes of THIS run (each its own `bench.py --workload cX`, a timed region of <= 2 s
    after its warm-up), so that the driver's one command covers every BASELINE configuration: device-resident rate, API rate
    (host arrays in, results out), the dominant kernel's roofline fraction, the end-to-end fraction and the PMC traffic
    constant with its currency flag.  A child that fails is reported, never fatal."""
    import subprocess
    out = {}
    for name, steps, warm in (('c2', 10, 2), ('c4', 200, 20), ('c5', 6, 1)):
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', name, '--gpus', '1', '--no-cpu-baseline',
               '--steps', str(steps), '--warmup', str(warm)]
        env = dict(os.environ, COSMOGP_DEVICE=str(local_rank), CGP_BENCH_NO_CONFIGS='1')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
            env.pop(k, None)
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
            d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1])
            rf = d.get('roofline') or {}
            ex = d.get('extra') or {}
            o = dict(workload=d['config']['workload'], fits_per_sec=d['value'], ms_per_step=d['ms_per_step'], steps=d['steps'],
                     fits_per_step=d['config']['fits_per_step_per_gpu'], handles=d['config']['handles_per_gpu'],
                     api_fits_per_sec=rf.get('api_fits_per_sec'),
                     dominant_kernel=(rf.get('kernel') or '').split(' ')[0], dominant_kernel_frac=rf.get('frac'),
                     frac_end_to_end=rf.get('frac_end_to_end'), traffic=rf.get('traffic'),
                     traffic_unit=rf.get('traffic_unit', 'bytes per launch'),
                     traffic_source_current=rf.get('traffic_source_current'),
                     single_fit=rf.get('single_fit'), seconds=round(time.perf_counter() - t0, 1))
            for k in ('api_fits_per_sec_1024_per_call', 'api_fits_per_sec_2callers', 'small_objects'):
                if k in ex:
                    o[k] = ex[k]
            out[name] = o
        except Exception as exc:
            out[name] = dict(error=repr(exc)[:300], seconds=round(time.perf_counter() - t0, 1))
    return out


def launch_ranks(n_gpus, script=None, argv=None):
    """`python bench.py --gpus N` run plainly: start N ranks (one per GPU) as child processes BEFORE anything in
    this process touches the GPU, pass rank 0's JSON line through, return the worst exit code.  A rank that dies takes
    the others with it (they would wait for it inside the collective for ever)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), CGP_BENCH_CHILD='1')
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else argv),
                                      env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()))
    reader.start()
    rc = 0
    while any(p_.poll() is None for p_ in procs):
        bad = [p_ for p_ in procs if p_.poll() not in (None, 0)]
        if bad:
            rc = abs(bad[0].returncode) or 1
            for p_ in procs:
                if p_.poll() is None:
                    p_.terminate()
            break
        time.sleep(0.05)
    for p_ in procs:
        try:
            rc = max(rc, abs(p_.wait(timeout=30)))
        except subprocess.TimeoutExpired:
            p_.kill()
            rc = max(rc, 1)
    reader.join(timeout=30)
    sys.stdout.write((out[0] if out else b'').decode())
    sys.stdout.flush()
    return rc


class FileComm(object):
    """Stand-in used ONLY if the RCCL communicator of the library cannot be created (bench.py must still report the N-rank
    measurement, labelled as such in `config.collective`): barrier / all-gather / max over files in the node's temporary
    directory, one process per GPU on one node.  Not part of the product."""

    def __init__(self, rank, world):
        import tempfile
        self.rank, self.world, self.seq = rank, world, 0
        self.dir = os.path.join(tempfile.gettempdir(), 'cgp_bench_sync_%d_%s_%d'
                                % (os.getuid(), os.environ.get('MASTER_PORT', '0'), os.getppid()))
        os.makedirs(self.dir, exist_ok=True)
        self.backend = 'file'

    def _exchange(self, arr, timeout=600.):
        self.seq += 1
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        mine = os.path.join(self.dir, '%d_%d.npy' % (self.seq, self.rank))
        with open(mine + '.tmp', 'wb') as f:
            np.save(f, arr)
        os.replace(mine + '.tmp', mine)
        out, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, '%d_%d.npy' % (self.seq, r))
            while True:
                try:
                    with open(path, 'rb') as f:
                        out.append(np.load(f))
                    break
                except (OSError, ValueError, EOFError):
                    if time.time() - t0 > timeout:
                        raise RuntimeError('rank %d: no %s after %.0f s' % (self.rank, path, timeout))
                    time.sleep(0.0002)
        return out

    def barrier(self):
        self._exchange(np.zeros(1))

    def _allgather(self, block):
        return np.stack(self._exchange(block))

    def allreduce_max(self, value):
        return float(max(v[0] for v in self._exchange(np.array([float(value)]))))

    def close(self):
        self.barrier()
        # every rank has READ the last exchange before rank 0 removes the directory
        open(os.path.join(self.dir, 'done_%d' % self.rank), 'w').close()
        if self.rank == 0:
            import shutil
            t0 = time.time()
            while time.time() - t0 < 30. and not all(os.path.exists(os.path.join(self.dir, 'done_%d' % r))
                                                     for r in range(self.world)):
                time.sleep(0.001)
            shutil.rmtree(self.dir, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default 10; c4: 200 -- its step is 1.1 ms)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps before them (default 2; c4: 20)')
    ap.add_argument('--workload', default='c3', choices=['c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the one-fit-at-a-time measurements (latency, Cholesky-only rate, stage split) and the '
                         'ceiling microbenchmarks: every kernel of the process is then one of the timed configuration '
                         '(for the rocprofv3 --stats / --pmc passes committed under profiles/)')
    ap.add_argument('--batch', type=int, default=-1,
                    help='c2 / c3: independent fits (distinct objects of the configured size) per launch sequence. '
                         'They go through the batched entry point (cgp_batch_fit, the path gaussian_process_nobject '
                         'uses): one launch sequence factors all of them in lock-step, which amortises the '
                         'latency-bound leaf chain. -1 = auto (8); 1 = one object per launch sequence (then '
                         '--streams handles overlap whole fits instead)')
    ap.add_argument('--streams', type=int, default=0,
                    help='independent fits in flight per GPU (separate handles = separate HIP streams and HBM '
                         'buffers); 0 = auto: 2 when c2 / c3 run in lock-step batches (--batch), 4 with --batch 1, '
                         '4 for c4 (232k -> 290k fits/s), 1 for the lock-step sweep')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == 'c4' else 10
    if args.warmup is None:
        args.warmup = 20 if args.workload == 'c4' else 2

    if args.gpus > 1 and 'RANK' not in os.environ:
        # not under a launcher: this process only starts the ranks (it never initialises HIP)
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print('note: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)

    import threading
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    dev = _lib.Device(local_rank)
    dev.selftest_mfma()
    comm = None
    collective = None
    rccl_ranks = None
    if world > 1 or 'RANK' in os.environ:          # launched as a rank (any world size): RCCL inside the library
        from cosmogp_amd import multigpu
        if os.environ.get('CGP_BENCH_COMM') == 'file':
            # asked for explicitly (a debugging aid, e.g. on a box whose RCCL cannot start): never chosen silently
            comm = FileComm(rank, world)
            collective = 'FILE STAND-IN (CGP_BENCH_COMM=file): not RCCL'
            rccl_ranks = None
        else:
            try:
                comm = multigpu.Comm(dev, rank=rank, world=world)
            except Exception as exc:
                # a multi-rank number that did not go through RCCL is not the measurement the contract asks for: fail
                print('rank %d: RCCL communicator failed (%s); set CGP_BENCH_COMM=file to run with the file stand-in'
                      % (rank, exc), file=sys.stderr)
                sys.exit(3)
            rccl_ranks = comm.ranks_reported           # ncclCommCount
            if rccl_ranks != world:
                print('rank %d: RCCL reports %d ranks, launched with %d' % (rank, rccl_ranks, world), file=sys.stderr)
                sys.exit(3)
            collective = 'RCCL inside the library (cgp_comm_*: ncclAllGather / ncclAllReduce on the handle\'s stream)'
    # config 4 on one GPU is the WHOLE configuration (1024 light curves) in flight: 512 per pass x 2 handles; every further
    # rank brings its own 1024 (weak scaling: the per-GPU work does not change with N)
    wl = make_workload(args.workload, rank, nobj=int(os.environ.get('CGP_BENCH_C4_NOBJ', '512')))
    batched = 'B' in wl
    # Batches of objects with >= 16 tile columns (N >= 2048) are factored by ONE persistent launch per group of objects
    # (csrc/dag.hpp, several leaf lanes + one interleaved task list); smaller objects and CGP_NO_DAG_BATCH=1 take one launch
    # per step.  The persistent launch owns every CU, so one handle is enough to fill the GPU.
    dag_batch = (not batched and -(-wl['n'] // 128) >= int(os.environ.get('CGP_DAG_BATCH_MIN_NB', '16'))
                 and 'CGP_NO_DAG_BATCH' not in os.environ and 'CGP_NO_DAG' not in os.environ)
    if batched:
        lock = 1
    elif 'sweep' in wl:
        lock = len(wl['sweep']) if args.batch < 0 else max(1, args.batch)      # candidates per launch sequence
    else:
        # objects per launch sequence
        lock = (16 if (dag_batch and wl['n'] <= 4096) else 8) if args.batch < 0 else max(1, args.batch)
    nstreams = args.streams if args.streams > 0 else ((2 if wl['B'] >= 512 else 4) if batched else ((1 if dag_batch else 2) if lock > 1 else 4))
    if 'sweep' in wl and args.streams <= 0:
        nstreams = 1 if lock > 1 else 4
    # Independent fits are the unit of this path (objects, candidates): with several in flight the
    # latency-bound leaf chain of one fit runs in the gaps of the others' bulk updates.  Every handle has
    # its own stream and its own copy of the inputs in HBM; each fit is still one full pass.
    extra_devs = [_lib.Device(local_rank) for _ in range(nstreams - 1)]

    # ---- inputs to HBM + one checked fit (also sizes every workspace) -------------------------
    if batched:
        diags = [K.diag_term(wl['n'], e, 0., 0.) for e in wl['es']]

        def step():
            return dev.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
        ll, info, mu, var = step()
        assert not info.any() and np.all(np.isfinite(mu))
        for d in extra_devs:
            d.batch_fit(1, wl['xs'], wl['ys'], diags, wl['kp'], wl['xnew'], None, 0.)
        units_per_step = wl['B']
    elif 'sweep' in wl:
        dev.set_data(wl['dim'], wl['x'], wl['y'])
        diag = K.diag_term(wl['n'], wl['e'], 0., 0.)
        kps = [K.kernel_params_2d([s_, l_, l_, 0.], sigma2_in_train=True) for s_, l_ in wl['sweep']]
        lls = [dev.fit(kp, diag)[0] for kp in kps]
        assert np.all(np.isfinite(lls))
        for d in extra_devs:
            d.set_data(wl['dim'], wl['x'], wl['y'])
            d.fit(kps[0], diag)
        mu, var = np.array(lls), np.zeros(len(lls))
        units_per_step = len(kps)
    else:
        dev.set_data(wl['dim'], wl['x'], wl['y'])
        diag = K.diag_term(wl['n'], wl['e'], 0., 0.)
        ll, ld, mu, var = dev.fit(wl['kp'], diag, wl['xnew'], None, 0.)
        assert np.isfinite(ll) and np.all(np.isfinite(mu)) and var.min() > -1e-9
        if lock > 1:
            # `lock` distinct objects per launch sequence, resident on every handle of the timed region (dev keeps
            # the single object for the roofline pass); object 0 must come out bit-identical to its single fit
            objs = [wl] + [make_workload(args.workload, rank, obj=i) for i in range(1, lock)]
            bdiags = [K.diag_term(o['n'], o['e'], 0., 0.) for o in objs]
            fit_devs = extra_devs + [_lib.Device(local_rank)]
            for d in fit_devs:
                bll, binfo, bmu, bvar = d.batch_fit(wl['dim'], [o['x'] for o in objs], [o['y'] for o in objs], bdiags,
                                                    wl['kp'], wl['xnew'], None, 0.)
                # object 0 of the lock-step batch against its single fit (cgp_fit = single-launch factorisation, the
                # batch = launch-per-step route: the same tiles, sums associated differently)
                assert not binfo.any() and abs(bll[0] - ll) <= 1e-11 * abs(ll)
                assert np.allclose(bmu[0], mu, rtol=0, atol=1e-9) and np.allclose(bvar[0], var, rtol=0, atol=1e-9)
        else:
            for d in extra_devs:
                d.set_data(wl['dim'], wl['x'], wl['y'])
                out = d.fit(wl['kp'], diag, wl['xnew'], None, 0.)
                assert out[0] == ll and np.array_equal(out[2], mu)          # every handle computes the same fit
        # a step = one pass over the objects of a launch sequence (`lock` distinct objects, one persistent launch): every
        # step of the timed region is then the same launch shape (round 2 counted single fits and ended on a partial pass)
        units_per_step = lock

    def sync_all():
        for d in set([dev] + extra_devs + (fit_devs if (lock > 1 and 'sweep' not in wl) else [])):
            d._lib.cgp_synchronize(d._h)
        if comm is not None:
            comm.barrier()                 # ncclAllReduce on dev's stream + stream synchronise

    all_devs = fit_devs if (lock > 1 and 'sweep' not in wl) else [dev] + extra_devs

    stagger_s = float(os.environ.get('CGP_BENCH_STAGGER_MS', '0')) * 1e-3

    def run_on(d, k, offset, slot=0):
        if stagger_s > 0. and slot > 0:
            time.sleep(stagger_s * slot)
        # k device-resident fits (or batches of fits) back to back on this handle's stream, then sync:
        # the last cgp_fit / cgp_batch_fit left every input in HBM
        if 'sweep' in wl and lock > 1:
            # lock-step sweep: up to `lock` candidates per launch sequence (cgp_sweep_loglik: 7 doubles per candidate
            # go down, 1 comes back, data resident and shared)
            i = 0
            while i < k:
                c = min(lock, k - i)
                ll_, info_ = d.sweep_loglik(np.array([kps[(offset + i + j) % len(kps)] for j in range(c)]), None)
                assert not info_.any()
                i += c
        elif 'sweep' in wl:
            # one hyper-parameter set per evaluation: 7 doubles go down, 1 comes back (data resident)
            for i in range(k):
                d.fit(kps[(offset + i) % len(kps)], None)
        elif k > 0 and lock > 1:
            d.bench_fit_objects(k, lock)          # k passes over the whole resident batch
        elif k > 0:
            d.bench_fit(k, stages=False)

    def run_steps(k):
        # EXACTLY k steps (k * units for the sweep), dealt over the handles; ctypes drops the GIL in the call
        total = k * (len(kps) if 'sweep' in wl else 1)
        share = [total // len(all_devs) + (1 if i < total % len(all_devs) else 0) for i in range(len(all_devs))]
        if len(all_devs) == 1:
            return run_on(all_devs[0], share[0], 0)
        ths, off = [], 0
        for slot, (d, n_) in enumerate(zip(all_devs, share)):
            ths.append(threading.Thread(target=run_on, args=(d, n_, off, slot)))
            off += n_
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    run_steps(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    if comm is not None:                           # the one gather of the path: predictions of every rank
        gathered = comm._allgather(np.stack([np.ravel(mu), np.ravel(var)]))
        assert gathered.shape[0] == world
    sync_all()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)

    # ---- roofline of the dominant kernel (event-instrumented pass, rank 0's GPU) --------------
    roof, extra = None, {}
    if True:
        # the update kernel exactly as it runs in the timed region: in lock-step mode one launch covers `lock` objects
        # (the stage pass runs on a handle of the timed region), otherwise one object
        rdev = all_devs[0] if (lock > 1 and 'sweep' not in wl) else dev
        ms_fit, st = rdev.bench_fit(min(args.steps, 5) if 'sweep' not in wl else 1, stages=True)
        n = wl['n']
        gemm_ms, launches, flops = st[3], st[6], st[7]
        dag = st[11] > 0. and gemm_ms == 0.          # the single-launch factorisation (csrc/dag.hpp), one or several objects
        nobj_pass = (wl['B'] if batched else lock) if (lock > 1 or batched) else 1
        if dag:
            # the dominant kernel is the persistent launch: algorithmic flops per object = N^3/3 (factor) + N^2 (M + 1) (the M
            # cross rows and the residual row ride through the same sweep); one launch per group of objects
            gemm_ms, launches = st[11], max(1., launches)
            flops = nobj_pass * (float(n) ** 3 / 3.0 + float(n) ** 2 * (wl.get('m', 0) + 1))
        ll_route = (not dag) and gemm_ms == 0.
        if ll_route:
            # small objects on the left-looking panel route (csrc/smallchol.hpp): no update launches at all -- the dominant kernels
            # are k_ll_diag + k_ll_slab (+ the diagonal-block / panel-solve kernels of panel 0): the whole factorisation
            gemm_ms, launches = st[1] + st[2], max(1., 2. + 3. * (-(-n // 128) - 1))
            flops = nobj_pass * (float(n) ** 3 / 3.0 + float(n) ** 2 * (wl.get('m', 0) + 1))
        ach = flops / (gemm_ms * 1e-3) / 1e12
        # HBM bytes per launch (per PASS for the launch-per-step route of small objects) from the separate rocprofv3 --pmc
        # passes (NOT this run); the file records the hash of the device sources it was collected on
        traffic, tsrc, tcur = None, None, None
        tpath, trec = traffic_file(args.workload, dag, nobj_pass)
        if tpath:
            traffic = trec['traffic_bytes_per_launch']
            tsrc = os.path.relpath(tpath, ROOT)
            tcur = trec.get('kernel_sources_sha256') == kernel_source_hash()
        roof = dict(bound='mfma', achieved=round(ach, 3), peak=SPEC_FP64_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=round(ach / SPEC_FP64_MFMA_TFLOPS, 4), traffic=traffic,
                    kernel=('k_chol_dag (persistent launch: one leaf lane per object + ticketed TRSM / update tasks, '
                            'v_mfma_f64_4x4x4_4b_f64)' if dag else
                            ('k_ll_diag + k_ll_slab (left-looking panels of small objects, csrc/smallchol.hpp) with k_potrf_diag / k_trsm_panel '
                             'for panel 0: every launch of the factorisation' if ll_route else
                             'k_gemm_nt (v_mfma_f64_4x4x4_4b_f64), both tile forms')),
                    launches_per_fit=int(launches),
                    flops_per_launch=flops / launches, avg_launch_ms=round(gemm_ms / launches, 5),
                    objects_per_launch=lock,
                    measured='HIP events around every launch of one eager pass over the %d objects of a launch '
                             'sequence, on its stream (compare with the rocprofv3 summary of `bench.py --no-extras`)'
                             % (wl['B'] if batched else lock))
        if st[9] > 0:       # per tile form, to be matched against the per-kernel rows of rocprofv3 --stats
            roof['by_form'] = {
                'k_gemm_nt<128,64,4,2,16>': dict(launches=int(st[9]), avg_launch_ms=round(st[8] / st[9], 5),
                                                 achieved=round(st[10] / (st[8] * 1e-3) / 1e12, 3)),
                'k_gemm_nt<64,64,2,4,16>': dict(launches=int(launches - st[9]),
                                                avg_launch_ms=round((gemm_ms - st[8]) / max(1., launches - st[9]), 5),
                                                achieved=round((flops - st[10]) / ((gemm_ms - st[8]) * 1e-3) / 1e12, 3))}
        if lock > 1 or batched:
            roof['launches_per_pass'] = roof.pop('launches_per_fit')
        roof['objects_per_launch'] = wl['B'] if batched else lock
        if traffic is not None:
            roof['traffic_source'] = ('replayed constant, not measured by this run: %s (PMC FETCH_SIZE x2, gfx950 '
                                      'wide-load correction, + WRITE_SIZE per launch, separate rocprofv3 --pmc passes of '
                                      '`bench.py --no-extras`)' % tsrc)
            # does the constant belong to the device code that just ran?  (sha256 of csrc/kernels.hpp, rbfmath.hpp, dag.hpp,
            # dag_tasks.hpp recorded by tools/pmc_traffic.py at collection time; tests/test_host_logic.py checks the default's)
            roof['traffic_source_current'] = bool(tcur)
            if batched:
                roof['traffic_unit'] = 'bytes per pass: every launch of the %d objects (tools/pmc_pass_traffic.py)' % wl['B']
                roof['traffic_bytes_per_fit'] = round(traffic / wl['B'], 1)
    if not batched and not args.no_extras:
        if 'sweep' in wl:
            dev.fit(kps[0], None)                    # back to one candidate at a time on this handle
        t_one = time.perf_counter()
        dev.bench_fit(3, stages=False)
        single_ms = (time.perf_counter() - t_one) * 1e3 / 3
        # single-fit Cholesky rate (the second half of BASELINE.json's metric): log-likelihood-only fits, i.e.
        # assemble + blocked Cholesky of the N x N matrix with one residual row, N^3/3 flops per fit
        dev.fit(wl['kp'], None)
        dev.bench_fit(2, stages=False)
        t_c = time.perf_counter()
        dev.bench_fit(5, stages=False)
        chol_ms = (time.perf_counter() - t_c) * 1e3 / 5
        dev.fit(wl['kp'], None, wl['xnew'], None, 0.) if wl['xnew'] is not None else None
        ms_fit, st = dev.bench_fit(2, stages=True)     # stage split of ONE fit at a time (single-object handle)
        potrf_ms = st[1] + st[2] + st[3] + st[11]          # [11]: the single-launch factorisation (csrc/dag.hpp)
        chol_flops = n ** 3 / 3.0
        # the sweep also carries the M+1 tall rows (n^2 (m+1) flops): time share by flops
        sweep_flops = chol_flops + float(n) ** 2 * (wl['m'] + 1)
        extra = dict(single_fit_latency_ms=round(single_ms, 4),
                     single_fit_cholesky_ms=round(chol_ms, 4),
                     single_fit_cholesky_gflops=round(n ** 3 / 3.0 / (chol_ms * 1e-3) / 1e9, 1),
                     ms_per_fit_events=round(ms_fit, 4),
                     single_fit_gemm_tflops=(round(st[7] / (st[3] * 1e-3) / 1e12, 2) if st[3] > 0 else None),
                     stage_ms=dict(assemble_K=round(st[0], 4), potrf_diag=round(st[1], 4), trsm_panel=round(st[2], 4),
                                   gemm_update=round(st[3], 4), assemble_cross=round(st[4], 4), finish=round(st[5], 4),
                                   single_launch_factorisation=round(st[11], 4)),
                     sweep_gflops=round(sweep_flops / (potrf_ms * 1e-3) / 1e9, 1),
                     cholesky_gflops_equiv=round(sweep_flops / (potrf_ms * 1e-3) / 1e9, 1),
                     assemble_K_GBs=round(8. * n * (n + 1) / 2 / (st[0] * 1e-3) / 1e9, 1),
                     assemble_K_frac_of_hbm=round(8. * n * (n + 1) / 2 / (st[0] * 1e-3) / 1e9 / SPEC_HBM_GBS, 4))
    if not batched and not args.no_extras and roof is not None and 'single_fit_cholesky_gflops' in extra:
        # the two numbers BASELINE.json's metric names for ONE fit at a time, next to the dominant-kernel roofline
        roof['single_fit'] = dict(latency_ms=extra['single_fit_latency_ms'],
                                  fits_per_sec=round(1e3 / extra['single_fit_latency_ms'], 2),
                                  cholesky_ms=extra['single_fit_cholesky_ms'],
                                  cholesky_gflops=extra['single_fit_cholesky_gflops'],
                                  frac=round(extra['single_fit_cholesky_gflops'] / 1e3 / SPEC_FP64_MFMA_TFLOPS, 4),
                                  note='one fit at a time on one handle (no lock-step batch, no second stream): full fit '
                                       'latency, and N^3/3 flops over the wall time of a log-likelihood-only fit '
                                       '(assembly + blocked Cholesky with one residual row)')
    if not batched and not args.no_extras and args.workload == 'c3' and roof is not None:
        # the N = 4096 half of BASELINE.json's metric ("single-fit Cholesky GF/s at N=4096/8192"): the first 4096 points of
        # the same object, one log-likelihood-only fit at a time on a handle of its own, and its full fit (M = 2500)
        n4 = 4096
        d4 = _lib.Device(local_rank)
        d4.set_data(wl['dim'], wl['x'][:n4], wl['y'][:n4])
        dg4 = K.diag_term(n4, wl['e'][:n4], 0., 0.)
        d4.fit(wl['kp'], dg4)
        d4.bench_fit(2, stages=False)
        t_c = time.perf_counter()
        d4.bench_fit(8, stages=False)
        c4_ms = (time.perf_counter() - t_c) * 1e3 / 8
        d4.fit(wl['kp'], dg4, wl['xnew'], None, 0.)
        d4.bench_fit(2, stages=False)
        t_c = time.perf_counter()
        d4.bench_fit(5, stages=False)
        f4_ms = (time.perf_counter() - t_c) * 1e3 / 5
        d4.close()
        g4 = n4 ** 3 / 3.0 / (c4_ms * 1e-3) / 1e9
        extra['single_fit_n4096'] = dict(latency_ms=round(f4_ms, 4), cholesky_ms=round(c4_ms, 4), cholesky_gflops=round(g4, 1),
                                         frac=round(g4 / 1e3 / SPEC_FP64_MFMA_TFLOPS, 4),
                                         note='2D RBF GP, the first 4096 points of the same object: N^3/3 flops over the wall time '
                                              'of a log-likelihood-only fit; latency_ms = its full fit with M = 2500')
        roof['single_fit_n4096'] = extra['single_fit_n4096']
    if not args.no_extras and 'sweep' not in wl:
        # API-level rate, the SURVEY.md 8(d) definition of a fit: host arrays in -> device -> mu / var / loglik on the
        # host (PCIe-inclusive; it is never `value`)
        adev = dev
        if batched:
            # the objects have equal lengths: ONE (B, n) array per quantity is the natural host form (no per-object work in
            # the binding); the list-of-arrays form the reference's classes hold is timed beside it
            px, py, pd = np.stack(wl['xs']), np.stack(wl['ys']), np.stack(diags)

            def api_rate(x_, y_, d_, nobj_=None, blocks=3, reps_a=8):
                # median of `blocks` blocks of `reps_a` calls (single blocks of a few ms each scatter by 10 % from run to run)
                adev.batch_fit(1, x_, y_, d_, wl['kp'], wl['xnew'], None, 0.)
                dts = []
                for _ in range(blocks):
                    t_a = time.perf_counter()
                    for _ in range(reps_a):
                        out_ = adev.batch_fit(1, x_, y_, d_, wl['kp'], wl['xnew'], None, 0.)
                    dts.append((time.perf_counter() - t_a) / reps_a)
                return (nobj_ or wl['B']) / sorted(dts)[len(dts) // 2], out_
            rate_p, out_p = api_rate(px, py, pd)
            rate_l, out_l = api_rate(wl['xs'], wl['ys'], diags)
            assert np.array_equal(out_p[0], out_l[0]) and np.array_equal(out_p[2], out_l[2])       # same bits either way
            extra['api_fits_per_sec'] = round(rate_p, 2)
            extra['api_fits_per_sec_lists'] = round(rate_l, 2)
            # two callers, one handle each (the binding releases the GIL inside the call): one caller's upload and download
            # run under the other's device pass
            import threading
            ncall = max(2, int(os.environ.get('CGP_BENCH_API_CALLERS', '2')))
            callers = [adev] + [_lib.Device(local_rank) for _ in range(ncall - 1)]
            for c_ in callers:
                c_.batch_fit(1, px, py, pd, wl['kp'], wl['xnew'], None, 0.)
            reps_t = 48          # (16 calls per thread were as long as the threads' start-up: 180k-330k from run to run)

            def caller(c_):
                for _ in range(reps_t):
                    c_.batch_fit(1, px, py, pd, wl['kp'], wl['xnew'], None, 0.)
            ths = [threading.Thread(target=caller, args=(c_,)) for c_ in callers]
            t_a = time.perf_counter()
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
            extra['api_fits_per_sec_%dcallers' % ncall] = round(len(callers) * reps_t * wl['B'] / (time.perf_counter() - t_a), 2)
            for c_ in callers[1:]:
                c_.close()
            extra['api_note'] = ('cgp_batch_fit with host arrays in (upload of all %d objects: coordinates, residuals, diagonals as '
                                 'one (B, n) array each) and loglik / mu / var back on the host, ONE handle, one call after the '
                                 'other; _lists: the same from lists of per-object arrays; _<k>callers: k host threads with a handle each' % wl['B'])
            # BASELINE config[3] as ONE call on one GPU: all 1024 light curves in one cgp_batch_fit, ONE handle, one caller; the
            # library deals them into chunks of 256 and uploads chunk k + 1 under the device pass of chunk k (round 4)
            try:
                w1k = make_workload('c4', rank, nobj=1024)
                d1k = [K.diag_term(w1k['n'], e, 0., 0.) for e in w1k['es']]
                PX, PY, PD = np.stack(w1k['xs']), np.stack(w1k['ys']), np.stack(d1k)
                o1 = adev.batch_fit(1, PX, PY, PD, wl['kp'], wl['xnew'], None, 0.)
                assert not o1[1].any() and np.array_equal(o1[2][:wl['B']], out_p[2])       # the same objects, the same bits
                extra['api_fits_per_sec_1024_per_call'] = round(api_rate(PX, PY, PD, nobj_=1024, reps_a=6)[0], 2)
                tb = min(adev.bench_fit(20, stages=False)[0] for _ in range(2))
                extra['resident_fits_per_sec_1024_one_handle'] = round(1024 / tb * 1e3, 1)
                del PX, PY, PD, w1k, d1k
            except Exception as exc:
                extra['api_fits_per_sec_1024_per_call'] = dict(error=repr(exc)[:200])
            # the reference's actual use (docs/notebook/1D_kernel_example_with_noise.ipynb, pull.py:51-66): many objects of tens
            # of points -- 1000 light curves of 40 points, 19 prediction points, one handle
            try:
                rs = np.random.default_rng(40)
                sx = np.sort(rs.uniform(-12., 42., (1000, 40)), axis=1)
                se = rs.uniform(0.02, 0.1, (1000, 40))
                sy = 0.5 * np.sin(sx / 8.) + se * rs.standard_normal((1000, 40))
                sd = se ** 2
                adev.batch_fit(1, sx, sy, sd, wl['kp'], wl['xnew'], None, 0.)
                t_a = time.perf_counter()
                for _ in range(8):
                    so = adev.batch_fit(1, sx, sy, sd, wl['kp'], wl['xnew'], None, 0.)
                api_s = 8 * 1000 / (time.perf_counter() - t_a)
                ts = min(adev.bench_fit(50, stages=False)[0] for _ in range(2))
                extra['small_objects'] = dict(objects=1000, n=40, m=19, resident_fits_per_sec=round(1000 / ts * 1e3, 1),
                                              ms_per_pass=round(ts, 4), api_fits_per_sec=round(api_s, 1),
                                              packed=os.environ.get('CGP_NO_PACK') is None,
                                              note='1000 x (n = 40, M = 19) per call, one handle; objects of <= 64 points are packed '
                                                   'two per 128 x 128 tile as a block diagonal (csrc/api.hip, round 4)')
                assert not so[1].any()
            except Exception as exc:
                extra['small_objects'] = dict(error=repr(exc)[:200])
        else:
            def api_one():
                adev.set_data(wl['dim'], wl['x'], wl['y'])
                return adev.fit(wl['kp'], diag, wl['xnew'], None, 0.)
            api_one()
            t_a = time.perf_counter()
            reps_a = 3
            for _ in range(reps_a):
                api_one()
            dt_1 = (time.perf_counter() - t_a) / reps_a
            extra['api_single_fit_ms'] = round(dt_1 * 1e3, 4)
            extra['api_single_fits_per_sec'] = round(1. / dt_1, 2)
            if lock > 1:
                bd = all_devs[0]
                bd.batch_fit(wl['dim'], [o['x'] for o in objs], [o['y'] for o in objs], bdiags, wl['kp'], wl['xnew'], None, 0.)
                reps_a = 5
                t_a = time.perf_counter()
                for _ in range(reps_a):
                    bd.batch_fit(wl['dim'], [o['x'] for o in objs], [o['y'] for o in objs], bdiags, wl['kp'], wl['xnew'],
                                 None, 0.)
                dt_a = (time.perf_counter() - t_a) / reps_a
                extra['api_fits_per_sec'] = round(lock / dt_a, 2)
                extra['api_note'] = ('cgp_batch_fit, %d objects per call: host arrays in (coordinates, residuals, '
                                     'diagonals, grid), mu / var / loglik back on the host, ONE handle; '
                                     'api_single_*: cgp_fit_set_data + cgp_fit of one object' % lock)
            else:
                extra['api_fits_per_sec'] = extra['api_single_fits_per_sec']
    if roof is not None and 'api_fits_per_sec' in extra:
        # SURVEY.md 8(d) counts a fit from host arrays in to results on the host: that rate, next to the device-resident `value`
        roof['api_fits_per_sec'] = extra['api_fits_per_sec']
    if world > 1 or 'RANK' in os.environ:
        extra.update(multi_gpu_extras(dev, comm, rank, world, local_rank, args))
    if rank == 0:
        if extra.get('assemble_K_GBs'):
            # second roofline object: the covariance build (K1), HBM-write bound by its algorithmic bytes
            # (lower triangle of K, 8 N (N + 1) / 2 B), measured with HIP events on one fit at a time
            extra['roofline_assembly'] = dict(bound='hbm', achieved=extra['assemble_K_GBs'], peak=SPEC_HBM_GBS, unit='GB/s',
                                              frac=extra['assemble_K_frac_of_hbm'], kernel='k_assemble_train',
                                              note='algorithmic bytes = the lower triangle, 8 N (N + 1) / 2; the kernel is fp64-VALU-bound '
                                                   '(~40 operations per element in 2D, ~30 in 1D, every one rounded like the '
                                                   'reference: tools/asm_ablate.hip, DESIGN.md 4.2), stores alone reach 6 TB/s')
        if not args.no_extras:
            mfma_tf, hbm_gbs = dev.microbench()
            extra.update(measured_mfma_f64_ceiling_tflops=round(mfma_tf, 2), measured_hbm_write_gbs=round(hbm_gbs, 1))
            if roof is not None:
                roof['frac_of_measured_ceiling'] = round(roof['achieved'] / mfma_tf, 4)
        value = n_gpus * args.steps * units_per_step / elapsed
        if roof is not None:
            # end to end: the algorithmic flops of every fit of the timed region, N^3/3 + N^2 (M + 1), over its wall time -- next
            # to `frac`, which is the dominant kernel alone (for the launch-per-step route of small objects that kernel is 3
            # of ~20 launches: this is the number to read there)
            fl_fit = float(wl['n']) ** 3 / 3.0 + float(wl['n']) ** 2 * (wl.get('m', 0) + 1)
            roof['frac_end_to_end'] = round(value / n_gpus * fl_fit / 1e12 / SPEC_FP64_MFMA_TFLOPS, 4)
            roof['flops_per_fit'] = fl_fit
        if (n_gpus == 1 and args.workload == 'c3' and not args.no_extras and 'RANK' not in os.environ
                and 'CGP_BENCH_NO_CONFIGS' not in os.environ):
            extra['configs'] = other_configs(local_rank)
        line = dict(metric='gp_fits_per_sec', value=round(value, 4), unit='fits/s', n_gpus=n_gpus,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 4),
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                    config=dict(workload=wl['label'] + ((', %d distinct objects per lock-step launch sequence x %d handles'
                                                         % (lock, nstreams)) if (lock > 1 and not batched and 'sweep' not in wl)
                                                        else ''),
                                n=wl['n'], m=wl['m'], dim=wl['dim'],
                                fits_per_step_per_gpu=units_per_step,
                                fits_in_flight_per_gpu=nstreams * (wl['B'] if batched else lock),
                                objects_per_launch_sequence=(wl['B'] if batched else lock), handles_per_gpu=nstreams,
                                collective=collective, rccl_ranks=rccl_ranks,
                                value_definition='device-resident fits/s: every input already in HBM when the timed region starts, '
                                                 'results left in HBM (the contract of bench.py); roofline.api_fits_per_sec is the '
                                                 'SURVEY.md 8(d) rate of the same workload: host arrays in, mu / var / loglik back on '
                                                 'the host, PCIe included',
                                factorisation_route=('single launch: persistent tile-DAG kernel k_chol_dag, one leaf lane per object'
                                                     if (roof is not None and 'k_chol_dag' in roof['kernel'])
                                                     else ('left-looking panels: k_potrf_diag / k_trsm_panel (panel 0), k_ll_diag / k_ll_slab'
                                                           if (roof is not None and 'k_ll_diag' in roof['kernel'])
                                                           else 'one launch per step: k_potrf_diag / k_trsm_panel / k_gemm_nt')),
                                parallelism='independent fits, 1 rank per GPU, %d handles (HIP streams) per GPU, '
                                            '%d distinct objects per launch sequence' % (nstreams, wl['B'] if batched else lock)),
                    roofline=roof)
        if n_gpus == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(wl)
        line['extra'] = extra
        print(json.dumps(line))
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == '__main__':
    main()
