"""Sharding independent GP fits over the GPUs of one node.

Objects (light curves, focal-plane patches; the loops of Gaussian_process.py:207-211 and :304-335) and
hyperparameter candidates are independent units: rank r of W owns a contiguous block of them, runs
them on its own GPU through its own libcosmogp_hip handle, and ONE gather at the end assembles the
per-unit results (predictions / variances / log-likelihood scalars).  No data-path collective exists:
nothing is exchanged while fitting.

The gather is RCCL (ncclAllGather over xGMI) called INSIDE the library on the handle's stream
(cgp_comm_* of include/cosmogp_hip.h); this module only bootstraps it: rank 0 asks the library for the
128-byte ncclUniqueId and hands it to the other ranks through a file in the node's temporary directory
(one process per GPU on ONE node), then every rank calls cgp_comm_init.  No PyTorch anywhere in the package.
The collective-free logic lives in CommBase so that a CPU test can run it over another transport
(tests/gloo_comm.py: world-size-2 gloo).
"""
import os
import tempfile
import time

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


class CommBase(object):
    """What the sharded paths need from a communicator: rank, world and ONE primitive,
    _allgather(block) with block a (per, width) float64 array of the same shape on every rank,
    returning the (world, per, width) array on every rank."""

    rank = 0
    world = 1

    def _allgather(self, block):
        raise NotImplementedError

    def barrier(self):
        self._allgather(np.zeros((1, 1)))

    def my_units(self, n_units):
        lo, hi = shard_bounds(n_units, self.rank, self.world)
        return list(range(lo, hi))

    def allreduce_max(self, value):
        return float(self._allgather(np.full((1, 1), float(value))).max())

    def agree_any(self, flag):
        """True on every rank when `flag` is set on at least one: ranks must take the same branch before the
        next collective (a rank whose block hit a non-positive pivot must not leave its peers in a gather)."""
        return self.allreduce_max(1. if flag else 0.) > 0.

    def gather_units(self, local, n_units):
        """local: float64 array with one row (or scalar) per unit of this rank's block.
        Returns the (n_units, ...) array on every rank (all-gather of equal-size padded blocks)."""
        local = np.asarray(local, dtype=np.float64)
        tail = local.shape[1:]
        width = int(np.prod(tail)) if tail else 1
        per = -(-n_units // self.world)
        buf = np.zeros((per, max(width, 1)))
        if len(local) and width:
            buf[:len(local), :width] = local.reshape(len(local), width)
        recv = np.asarray(self._allgather(buf)).reshape(self.world, per, max(width, 1))
        out = np.empty((n_units, width))
        for r in range(self.world):
            lo, hi = shard_bounds(n_units, r, self.world)
            out[lo:hi] = recv[r, :hi - lo, :width]
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


def _id_file(port):
    base = os.environ.get('CGP_COMM_ID_FILE')
    if base:
        return base
    # every rank of one job has the same parent (torch.distributed.run's agent, or bench.py's launcher): its pid AND its
    # start time are in the name, so a file left behind by a crashed job (another launcher process, even one that got the
    # same pid and port) can never be read as this job's id
    ppid = os.getppid()
    try:
        with open('/proc/%d/stat' % ppid) as f:
            born = f.read().rsplit(')', 1)[1].split()[19]          # field 22: start time of the process, in clock ticks
    except (OSError, IndexError):
        born = '0'
    return os.path.join(tempfile.gettempdir(), 'cgp_rccl_id_%d_%s_%d_%s' % (os.getuid(), port, ppid, born))


def exchange_unique_id(rank, world, make_id, timeout=300.):
    """Rank 0 creates the id and publishes it (atomic rename); the others wait for the file."""
    path = _id_file(os.environ.get('MASTER_PORT', '0'))
    if world == 1:
        return make_id(), path
    if rank == 0:
        uid = make_id()
        try:
            os.unlink(path)                    # (a second communicator of the same job: never leave the first id readable)
        except OSError:
            pass
        tmp = '%s.tmp.%d' % (path, os.getpid())
        with open(tmp, 'wb') as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid, path
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                uid = f.read()
            if len(uid) == 128:
                return uid, path
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError('rank %d: no RCCL unique id at %s after %.0f s' % (rank, path, timeout))
        time.sleep(0.01)


class Comm(CommBase):
    """RCCL communicator of this rank, owned by a libcosmogp_hip handle (one process per GPU).
    rank / world / local device come from the arguments or from the launcher's environment
    (RANK, WORLD_SIZE, LOCAL_RANK: torch.distributed.run and bench.py --gpus N both export them)."""

    def __init__(self, device=None, rank=None, world=None):
        from . import _lib
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.device = device if device is not None else _lib.default_device()
        # one node by construction: RCCL's bootstrap sockets stay on the loopback interface unless the caller chose one
        # (a container's other interfaces / host name need not resolve)
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        uid, path = exchange_unique_id(self.rank, self.world, _lib.comm_unique_id)
        self.device.comm_init(self.rank, self.world, uid)
        self.backend = 'rccl'
        self.ranks_reported = self.device.comm_count()      # what RCCL itself says (bench.py records it: config.rccl_ranks)
        self.barrier()                       # every rank has read the id: rank 0 may remove the file
        if self.rank == 0 and self.world > 1:
            try:
                os.unlink(path)
            except OSError:
                pass

    def _allgather(self, block):
        block = np.ascontiguousarray(block, dtype=np.float64)
        return self.device.comm_allgather(block).reshape((self.world,) + block.shape)

    def barrier(self):
        self.device.comm_barrier()

    def allreduce_max(self, value):
        from . import _lib
        return float(self.device.comm_allreduce(np.array([float(value)]), _lib.OP_MAX)[0])

    def close(self):
        self.device.comm_destroy()


#: HBM the candidates of one sweep launch sequence may occupy (P x (Np + 128) x Np x 8 bytes)
SWEEP_BYTES_LIMIT = 48 << 30


def sweep_log_likelihood(gp, table, comm=None, svd_method=False, streams=None):
    """Log-likelihood of `gp` for every hyper-parameter row of `table` (P x nhyp, plus a trailing
    nugget column when gp.fit_nugget) -- BASELINE config[4]: candidates are independent units.
    Rank r evaluates its contiguous block on its own GPU and ONE all-gather returns the P values on every
    rank.  A single-object GP on the Cholesky route evaluates its candidates in LOCK-STEP: cgp_sweep_loglik
    factors up to 256 candidate matrices in one launch sequence (data resident and shared, one 7-double
    parameter block per candidate down, one double back), in chunks of SWEEP_BYTES_LIMIT; with an RCCL
    communicator on the GP's own handle the whole sharded sweep is ONE library call (cgp_sweep_loglik_sharded:
    the gather runs device to device).  Everything else (several objects, svd_method=True) goes candidate by
    candidate.  A candidate whose matrix is not positive definite raises numpy.linalg.LinAlgError on every
    rank, like the serial loop.  comm=None: single process.  (`streams` is accepted for compatibility and
    ignored.)"""
    table = np.asarray(table, dtype=np.float64)
    n_units = len(table)
    mine = list(range(n_units)) if comm is None else comm.my_units(n_units)
    vals = np.zeros(len(mine))
    if not svd_method and gp.N_sn == 1:
        from .gaussian_process import _kparams
        dev = gp.device
        gp._make_resident(0)
        gp._factor_token = None

        def kp_of(k):
            row = table[k]
            hyp, nug = (row[:-1], row[-1]) if gp.fit_nugget else (row, gp.nugget)
            return _kparams(gp._dim, hyp, nug)

        n = len(gp.y[0])
        npad = -(-n // 128) * 128
        per = (npad + 128) * npad * 8
        chunk = int(max(1, min(256, SWEEP_BYTES_LIMIT // per)))
        diag = gp._diag(0)
        if comm is not None and getattr(comm, 'backend', None) == 'rccl' and comm.device is dev:
            kps = np.array([kp_of(k) for k in range(n_units)])
            ll, info = dev.sweep_loglik_sharded(kps, diag, chunk)
            bad = np.nonzero(info)[0]
            if len(bad):
                raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite '
                                            '(candidate %d)' % (info[bad[0]], bad[0]))
            return ll
        err = None
        kps = np.empty((len(mine), 7))
        for pos, k in enumerate(mine):
            kps[pos] = kp_of(k)
        for lo in range(0, len(mine), chunk):
            ll, info = dev.sweep_loglik(kps[lo:lo + chunk], diag)
            bad = np.nonzero(info)[0]
            if len(bad):
                err = np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite '
                                            '(candidate %d)' % (info[bad[0]], mine[lo + bad[0]]))
                break
            vals[lo:lo + chunk] = ll
    else:
        saved_comm = gp.comm
        gp.comm = None                 # objects are NOT sharded here: candidates are
        err = None
        try:
            for pos, k in enumerate(mine):
                gp.compute_log_likelihood(table[k], svd_method=svd_method)
                vals[pos] = gp.log_likelihood[0]
        except np.linalg.LinAlgError as e:
            err = e
        finally:
            gp.comm = saved_comm
    if comm is None:
        if err is not None:
            raise err
        return vals
    if comm.agree_any(err is not None):
        raise err if err is not None else np.linalg.LinAlgError('a candidate on another rank is not positive definite')
    return comm.gather_units(vals, n_units)
