"""Sharding independent GP fits over the GPUs of one node.

Objects (light curves, focal-plane patches) and hyperparameter candidates are independent
units: rank r of W owns a contiguous block of them, runs them on its own GPU through its own
libcosmogp_hip handle, and ONE gather at the end assembles the per-unit results
(predictions / variances / log-likelihood scalars).  No data-path collective exists: nothing
is exchanged while fitting.  The gather goes through torch.distributed -- backend "nccl" is
RCCL over xGMI on ROCm; "gloo" is used by the CPU tests -- which is the only thing torch is
used for in this package (imported lazily, never by the single-GPU path).
"""
import os

import numpy as np


def shard_bounds(n_units, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_cost(costs, world):
    """Longest-processing-time-first assignment for ragged units (cost ~ N^3).
    Returns a list of index lists, one per rank."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind='stable')
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += costs[i]
    return [sorted(v) for v in out]


class Comm(object):
    """Thin view of the default torch.distributed process group (one process per GPU)."""

    def __init__(self, init_if_needed=False, backend=None):
        import torch.distributed as dist
        self._dist = dist
        if not dist.is_initialized():
            if not init_if_needed:
                raise RuntimeError('torch.distributed is not initialised (launch with '
                                   'python -m torch.distributed.run, or pass init_if_needed=True)')
            import torch
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            if backend == 'nccl':
                torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            dist.init_process_group(backend=backend)
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.backend = dist.get_backend()

    def _tensor(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        if self.backend == 'nccl':
            t = t.cuda()
        return t

    def my_units(self, n_units):
        lo, hi = shard_bounds(n_units, self.rank, self.world)
        return list(range(lo, hi))

    def barrier(self):
        self._dist.barrier()

    def gather_units(self, local, n_units):
        """local: float64 array with one row (or scalar) per unit of this rank's block.
        Returns the (n_units, ...) array on every rank (all_gather of equal-size padded blocks)."""
        import torch
        local = np.asarray(local, dtype=np.float64)
        tail = local.shape[1:]
        width = int(np.prod(tail)) if tail else 1
        per = -(-n_units // self.world)
        buf = np.zeros((per, width))
        if len(local):
            buf[:len(local)] = local.reshape(len(local), width)
        send = self._tensor(buf)
        recv = torch.empty((self.world * per, width), dtype=send.dtype, device=send.device)
        self._dist.all_gather_into_tensor(recv, send)
        recv = recv.cpu().numpy().reshape(self.world, per, width)
        out = np.empty((n_units, width))
        for r in range(self.world):
            lo, hi = shard_bounds(n_units, r, self.world)
            out[lo:hi] = recv[r, :hi - lo]
        return out.reshape((n_units,) + tail)

    def gather_unit_arrays(self, items, n_units):
        """items: list of length n_units holding a 1-D array for this rank's units (None elsewhere),
        possibly of different lengths.  Returns the complete list on every rank."""
        mine = self.my_units(n_units)
        lens = np.zeros(len(mine))
        for k, u in enumerate(mine):
            lens[k] = len(items[u])
        all_lens = self.gather_units(lens, n_units).astype(int)
        width = int(all_lens.max()) if n_units else 0
        block = np.zeros((len(mine), width))
        for k, u in enumerate(mine):
            block[k, :len(items[u])] = items[u]
        full = self.gather_units(block, n_units)
        return [full[u, :all_lens[u]].copy() for u in range(n_units)]


#: HBM the candidates of one sweep launch sequence may occupy (P x (Np + 128) x Np x 8 bytes)
SWEEP_BYTES_LIMIT = 48 << 30


def sweep_log_likelihood(gp, table, comm=None, svd_method=False, streams=None):
    """Log-likelihood of `gp` for every hyper-parameter row of `table` (P x nhyp, plus a trailing
    nugget column when gp.fit_nugget) -- BASELINE config[4]: candidates are independent units.
    Rank r evaluates its contiguous block on its own GPU and ONE all_gather returns the P values on every
    rank.  A single-object GP on the Cholesky route evaluates its candidates in LOCK-STEP: cgp_sweep_loglik
    factors up to 256 candidate matrices in one launch sequence (data resident and shared, one 7-double
    parameter block per candidate down, one double back), in chunks of SWEEP_BYTES_LIMIT; everything else
    (several objects, svd_method=True) goes candidate by candidate.  A candidate whose matrix is not
    positive definite raises numpy.linalg.LinAlgError like the serial loop.  comm=None: single process.
    (`streams` is accepted for compatibility and ignored.)"""
    table = np.asarray(table, dtype=np.float64)
    n_units = len(table)
    mine = list(range(n_units)) if comm is None else comm.my_units(n_units)
    vals = np.zeros(len(mine))
    if not svd_method and gp.N_sn == 1 and len(mine) > 0:
        from .gaussian_process import _kparams
        dev = gp.device
        if gp._resident != (dev.generation, 0):
            dev.set_data(gp._dim, gp.Time[0], gp._resid(0))
            gp._resident = (dev.generation, 0)
        gp._factor_token = None
        kps = np.empty((len(mine), 7))
        for pos, k in enumerate(mine):
            row = table[k]
            hyp, nug = (row[:-1], row[-1]) if gp.fit_nugget else (row, gp.nugget)
            kps[pos] = _kparams(gp._dim, hyp, nug)
        n = len(gp.y[0])
        npad = -(-n // 128) * 128
        per = (npad + 128) * npad * 8
        chunk = int(max(1, min(256, SWEEP_BYTES_LIMIT // per)))
        diag = gp._diag(0)
        for lo in range(0, len(mine), chunk):
            ll, info = dev.sweep_loglik(kps[lo:lo + chunk], diag)
            bad = np.nonzero(info)[0]
            if len(bad):
                raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite '
                                            '(candidate %d)' % (info[bad[0]], mine[lo + bad[0]]))
            vals[lo:lo + chunk] = ll
    else:
        saved_comm = gp.comm
        gp.comm = None                 # objects are NOT sharded here: candidates are
        try:
            for pos, k in enumerate(mine):
                gp.compute_log_likelihood(table[k], svd_method=svd_method)
                vals[pos] = gp.log_likelihood[0]
        finally:
            gp.comm = saved_comm
    if comm is None:
        return vals
    return comm.gather_units(vals, n_units)
