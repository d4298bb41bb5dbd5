#!/usr/bin/env python
"""bench.py -- GP fits/s and single-fit fp64 Cholesky GF/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM: assemble K (RBF), blocked Cholesky with the cross matrix and the residual riding along
as tall rows, predictive mean + variance at M points, log-likelihood.  No host copies inside the
timed region (the PCIe-inclusive rate is noted in DESIGN.md, it is never `value`).

Workloads (seeds fixed):
  c3 (default at N=1)  2D RBF GP, N=8192, M=2500, single fit     -- the size BASELINE.json's metric names
  c2                   1D RBF GP, N=4096, M=1024, single fit
  c4                   1D light curves, N=512, M=19, 128 fits per GPU per step (1024 on 8 GPUs)
N > 1: one process per GPU (torch.distributed.run), every rank fits its own objects (weak
scaling, no data-path collective); RCCL is used for the barrier, the max-over-ranks of the timing
and one all_gather of the predictions per step batch.

Rank 0 prints ONE JSON line.  roofline = the dominant kernel (k_gemm_nt, fp64 MFMA): algorithmic
flops per launch / average launch duration measured with HIP events on the kernel's own stream.
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


def make_workload(name, rank=0, obj=0):
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
                    kp=K.kernel_params_2d(hyp), label='2D RBF GP N=8192 M=2500 single fit (BASELINE config[2])')
    if name == 'c2':
        rng = np.random.default_rng(2 + 1000 * rank + 37 * obj)
        n, m = 4096, 1024
        x = np.sort(rng.uniform(0., 400., n))
        y = np.sin(0.05 * x) + 0.3 * np.cos(0.31 * x) + 0.1 * rng.standard_normal(n)
        hyp = [1., 2.]
        return dict(dim=1, n=n, m=m, x=x, y=y, e=np.full(n, 0.1),
                    xnew=np.random.default_rng(20 + 1000 * rank).uniform(0., 400., m), hyp=hyp,
                    kp=K.kernel_params_1d(hyp), label='1D RBF GP N=4096 M=1024 single fit (BASELINE config[1])')
    if name == 'c4':
        B, n, m = 128, 512, 19
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
                    label='128 x 1D light-curve GPs N=512 M=19 per GPU per step (BASELINE config[3] / 8)')
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


def cpu_baseline(wl):
    """The oracle timed on the host cores on a bounded sample of the same workload."""
    from oracle import cosmogp_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    t0 = time.perf_counter()
    if 'B' in wl:
        nfit = 4
        for b in range(nfit):
            Kc = orc.rbf_kernel_1d(wl['xs'][b], wl['hyp'], y_err=wl['es'][b])
            H = orc.rbf_kernel_1d(wl['xs'][b], wl['hyp'], new_x=wl['xnew'])
            orc.fit_predict_cho(Kc, wl['ys'][b], H, np.full(wl['m'], wl['hyp'][0] ** 2))
        sample = '%d of the %d fits of one step (N=512, M=19), cho_factor/cho_solve form' % (nfit, wl['B'])
    elif 'sweep' in wl:
        # bounded sample: ONE log-likelihood of the sweep at a quarter of the points (N = 4096 of 16384),
        # scaled by the N^3 law to the full size; the full-size CPU factorisation alone takes minutes
        from scipy import linalg as la
        nfit = 1
        ns_ = wl['n'] // 4
        Kc = orc.rbf_kernel_2d(wl['x'][:ns_], wl['hyp'], y_err=wl['e'][:ns_], sigma2_in_train=True)
        c = la.cho_factor(Kc, lower=True)
        la.cho_solve(c, wl['y'][:ns_])
        dt_small = time.perf_counter() - t0
        scale = (wl['n'] / float(ns_)) ** 3
        return dict(value=1.0 / (dt_small * scale), unit='fits/s', cores=int(threads), kind='port',
                    sample='1 log-likelihood at N=%d (%.2f s) scaled by (N/n)^3 = %.0f to N=%d' %
                           (ns_, dt_small, scale, wl['n']), seconds=round(dt_small, 3))
    else:
        nfit = 1
        kern = orc.rbf_kernel_1d if wl['dim'] == 1 else orc.rbf_kernel_2d
        Kc = kern(wl['x'], wl['hyp'], y_err=wl['e'])
        H = kern(wl['x'], wl['hyp'], new_x=wl['xnew'])
        t1 = time.perf_counter()
        orc.fit_predict_cho(Kc, wl['y'], H, np.full(wl['m'], 1.0))
        t2 = time.perf_counter()
        sample = ('1 full fit of the workload (assembly %.2f s + cho_factor/cho_solve/predict %.2f s)'
                  % (t1 - t0, t2 - t1))
        # the reference's own algorithm (inv_matrix.py:23-25 explicit inverse, then dots) on the same matrices
        inv = orc.cholesky_inverse(Kc)
        mu_ref = H @ (inv @ wl['y'])
        var_ref = 1.0 - np.einsum('ij,ij->i', H @ inv, H)
        t3 = time.perf_counter()
        dt = t2 - t0
        return dict(value=nfit / dt, unit='fits/s', cores=int(threads), kind='port', sample=sample,
                    seconds=round(dt, 3),
                    reference_form=dict(value=1.0 / ((t1 - t0) + (t3 - t2)), unit='fits/s',
                                        note='same assembly + explicit inverse (cholesky, inv(L), inv_L.T @ inv_L) '
                                             'and dot products like cosmogp/inv_matrix.py:23-25, '
                                             'Gaussian_process.py:334-361 (variance diagonal only)',
                                        seconds=round((t1 - t0) + (t3 - t2), 3),
                                        checksum=float(mu_ref[0] + var_ref[0])))
    dt = time.perf_counter() - t0
    return dict(value=nfit / dt, unit='fits/s', cores=int(threads), kind='port', sample=sample,
                seconds=round(dt, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
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

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1 or 'RANK' in os.environ:          # launched by torch.distributed.run (any world size)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl')
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print('note: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world),
              file=sys.stderr)

    import threading
    from cosmogp_amd import _lib
    from cosmogp_amd import kernel as K
    dev = _lib.Device(local_rank)
    dev.selftest_mfma()
    wl = make_workload(args.workload, rank)
    batched = 'B' in wl
    if batched:
        lock = 1
    elif 'sweep' in wl:
        lock = len(wl['sweep']) if args.batch < 0 else max(1, args.batch)      # candidates per launch sequence
    else:
        lock = 8 if args.batch < 0 else max(1, args.batch)                     # objects per launch sequence
    nstreams = args.streams if args.streams > 0 else (4 if batched else (2 if lock > 1 else 4))
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
                assert not binfo.any() and bll[0] == ll and np.array_equal(bmu[0], mu) and np.array_equal(bvar[0], var)
        else:
            for d in extra_devs:
                d.set_data(wl['dim'], wl['x'], wl['y'])
                out = d.fit(wl['kp'], diag, wl['xnew'], None, 0.)
                assert out[0] == ll and np.array_equal(out[2], mu)          # every handle computes the same fit
        units_per_step = 1

    def sync_all():
        for d in set([dev] + extra_devs + (fit_devs if (lock > 1 and 'sweep' not in wl) else [])):
            d._lib.cgp_synchronize(d._h)
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

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
            # k fits = k // lock passes over the whole resident batch + one pass over the first k % lock objects
            if k // lock:
                d.bench_fit_objects(k // lock, lock)
            if k % lock:
                d.bench_fit_objects(1, k % lock)
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

    if lock > 1 and 'sweep' not in wl:
        # capture the launch graph of the partial pass each handle will need (k % lock objects) outside the timed region
        for i, d in enumerate(all_devs):
            k_i = args.steps // len(all_devs) + (1 if i < args.steps % len(all_devs) else 0)
            if k_i % lock:
                d.bench_fit_objects(1, k_i % lock)
    run_steps(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    if dist is not None:                           # the one gather of the path: predictions of every rank
        import torch
        send = torch.from_numpy(np.ascontiguousarray(np.stack([np.ravel(mu), np.ravel(var)]))).cuda()
        recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv.view(-1, *send.shape[1:]), send)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (event-instrumented pass, rank 0's GPU) --------------
    roof, extra = None, {}
    if True:
        # the update kernel exactly as it runs in the timed region: in lock-step mode one launch covers `lock` objects
        # (the stage pass runs on a handle of the timed region), otherwise one object
        rdev = all_devs[0] if (lock > 1 and 'sweep' not in wl) else dev
        ms_fit, st = rdev.bench_fit(min(args.steps, 5) if 'sweep' not in wl else 1, stages=True)
        n = wl['n']
        gemm_ms, launches, flops = st[3], st[6], st[7]
        ach = flops / (gemm_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_%s.json' % args.workload)
        if os.path.exists(tpath):          # HBM bytes per launch from the separate rocprofv3 --pmc passes
            traffic = json.load(open(tpath))['traffic_bytes_per_launch']
        roof = dict(bound='mfma', achieved=round(ach, 3), peak=SPEC_FP64_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=round(ach / SPEC_FP64_MFMA_TFLOPS, 4), traffic=traffic,
                    kernel='k_gemm_nt (v_mfma_f64_4x4x4_4b_f64), both tile forms', launches_per_fit=int(launches),
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
            roof['traffic_note'] = ('PMC FETCH_SIZE (x2, gfx950 wide-load correction) + WRITE_SIZE per launch, '
                                    'separate rocprofv3 passes, see profiles/')
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
        potrf_ms = st[1] + st[2] + st[3]
        chol_flops = n ** 3 / 3.0
        # the sweep also carries the M+1 tall rows (n^2 (m+1) flops): time share by flops
        sweep_flops = chol_flops + float(n) ** 2 * (wl['m'] + 1)
        extra = dict(single_fit_latency_ms=round(single_ms, 4),
                     single_fit_cholesky_ms=round(chol_ms, 4),
                     single_fit_cholesky_gflops=round(n ** 3 / 3.0 / (chol_ms * 1e-3) / 1e9, 1),
                     ms_per_fit_events=round(ms_fit, 4),
                     single_fit_gemm_tflops=round(st[7] / (st[3] * 1e-3) / 1e12, 2),
                     stage_ms=dict(assemble_K=round(st[0], 4), potrf_diag=round(st[1], 4), trsm_panel=round(st[2], 4),
                                   gemm_update=round(st[3], 4), assemble_cross=round(st[4], 4), finish=round(st[5], 4)),
                     sweep_gflops=round(sweep_flops / (potrf_ms * 1e-3) / 1e9, 1),
                     cholesky_gflops_equiv=round(sweep_flops / (potrf_ms * 1e-3) / 1e9, 1),
                     assemble_K_GBs=round(8. * n * (n + 1) / 2 / (st[0] * 1e-3) / 1e9, 1),
                     assemble_K_frac_of_hbm=round(8. * n * (n + 1) / 2 / (st[0] * 1e-3) / 1e9 / SPEC_HBM_GBS, 4))
    if rank == 0:
        if extra.get('assemble_K_GBs'):
            # second roofline object: the covariance build (K1), HBM-write bound by its algorithmic bytes
            # (lower triangle of K, 8 N (N + 1) / 2 B), measured with HIP events on one fit at a time
            extra['roofline_assembly'] = dict(bound='hbm', achieved=extra['assemble_K_GBs'], peak=SPEC_HBM_GBS, unit='GB/s',
                                              frac=extra['assemble_K_frac_of_hbm'], kernel='k_assemble_train',
                                              note='also bounded by fp64 VALU issue (~60 instructions per element), DESIGN.md 4.2')
        if not args.no_extras:
            mfma_tf, hbm_gbs = dev.microbench()
            extra.update(measured_mfma_f64_ceiling_tflops=round(mfma_tf, 2), measured_hbm_write_gbs=round(hbm_gbs, 1))
            if roof is not None:
                roof['frac_of_measured_ceiling'] = round(roof['achieved'] / mfma_tf, 4)
        value = n_gpus * args.steps * units_per_step / elapsed
        line = dict(metric='gp_fits_per_sec', value=round(value, 4), unit='fits/s', n_gpus=n_gpus,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 4),
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                    config=dict(workload=wl['label'], n=wl['n'], m=wl['m'], dim=wl['dim'],
                                fits_per_step_per_gpu=units_per_step,
                                fits_in_flight_per_gpu=nstreams * (wl['B'] if batched else lock),
                                objects_per_launch_sequence=(wl['B'] if batched else lock), handles_per_gpu=nstreams,
                                parallelism='independent fits, 1 rank per GPU, %d handles (HIP streams) per GPU, '
                                            '%d distinct objects per launch sequence' % (nstreams, wl['B'] if batched else lock)),
                    roofline=roof)
        if n_gpus == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(wl)
        line['extra'] = extra
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
