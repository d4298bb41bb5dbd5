"""Randomised stress of the single-launch route: random sizes, batch sizes, prediction grids, grids of workgroups; every result
against the launch-per-step route.  python tools/dag_stress.py <seconds> [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K


def run_stress(budget, seed=1, sizes=(130, 257, 400, 640, 900, 1300, 2050, 2600, 3300, 4100), verbose=True):
  """(cases, mismatches, worst error) after `budget` seconds of random cases (tests/test_dag.py runs a short seeded pass)"""
  rng = np.random.default_rng(seed)
  t_end = time.time() + budget
  n_cases = n_bad = 0
  worst = 0.
  while time.time() < t_end:
      n = int(rng.choice(list(sizes)))
      B = int(rng.choice([1, 1, 2, 3, 5, 9, 17]))
      if n * n * B > 4100 * 4100 * 3:
          B = 1
      m = int(rng.choice([0, 1, 19, 130, 700]))
      grid = rng.choice(['', '3', '9', '40'])
      ns = [n] + [int(rng.integers(max(2, n // 2), n + 1)) for _ in range(B - 1)]
      xs = [np.sort(rng.uniform(0., 0.15 * k + 3., k)) for k in ns]
      ys = [np.sin(0.05 * x) + 0.3 * np.cos(0.31 * x) + 0.1 * rng.standard_normal(len(x)) for x in xs]
      es = [rng.uniform(0.05, 0.2, k) for k in ns]
      xn = rng.uniform(0., 0.15 * n, m) if m else None
      kp = K.kernel_params_1d([float(rng.uniform(0.5, 1.5)), float(rng.uniform(1., 4.))], 0.02)
      outs = []
      single = B == 1 and rng.random() < 0.5
      bmax = str(int(rng.integers(1, 5))) if rng.random() < 0.3 else ''
      blazy = rng.choice(['', '0', '1'])           # task list of the batched route: by the launch's work / eager / lazy
      for route in ('dag', 'steps'):
          for k in ('CGP_NO_DAG', 'CGP_DAG_GRID', 'CGP_DAG_MIN_NB', 'CGP_DAG_BATCH_MIN_NB', 'CGP_DAG_BMAX', 'CGP_DAG_BLAZY'):
              os.environ.pop(k, None)
          if route == 'steps':
              os.environ['CGP_NO_DAG'] = '1'
          else:
              os.environ['CGP_DAG_MIN_NB'] = '2'; os.environ['CGP_DAG_BATCH_MIN_NB'] = '2'
              if grid:
                  os.environ['CGP_DAG_GRID'] = grid
              if bmax:
                  os.environ['CGP_DAG_BMAX'] = bmax
              if blazy:
                  os.environ['CGP_DAG_BLAZY'] = blazy
          dev = _lib.Device(0)
          diags = [K.diag_term(len(y), e, 0., 0.) for y, e in zip(ys, es)]
          if single:
              dev.set_data(1, xs[0], ys[0])
              r = dev.fit(kp, diags[0], xn, None, 0.)
              res = (np.array([r[0]]), None if xn is None else r[2][None, :], None if xn is None else r[3][None, :])
              g = dev.fit_grad(kp, diags[0])
              res = res + (np.asarray(g[2]),)
          else:
              r = dev.batch_fit(1, xs, ys, diags, kp, xn, None, 0.)
              assert not r[1].any()
              loo = dev.batch_loo(kp)
              res = (r[0], r[2], r[3], np.concatenate([loo[q][i, :ns[i]] for q in range(3) for i in range(B)]))
          dev.close()
          outs.append(res)
      a, b = outs
      err = float(np.max(np.abs(a[0] - b[0]) / np.abs(b[0])))
      if a[1] is not None:
          err = max(err, float(np.max(np.abs(a[1] - b[1]))), float(np.max(np.abs(a[2] - b[2]))))
      err2 = float(np.max(np.abs(a[3] - b[3]) / (np.abs(b[3]) + 1e-6)))
      n_cases += 1
      worst = max(worst, err, err2 * 1e-2)
      # (err2: gradient sums / leave-one-out vectors, entries near zero compared relatively with a 1e-6 floor: 1e-5 is rounding)
      if err > 1e-9 or err2 > 1e-5:
          n_bad += 1
          if verbose: print('MISMATCH n=%d B=%d m=%d grid=%r: %.2e %.2e' % (n, B, m, grid, err, err2), flush=True)
  for k in ('CGP_NO_DAG', 'CGP_DAG_GRID', 'CGP_DAG_MIN_NB', 'CGP_DAG_BATCH_MIN_NB', 'CGP_DAG_BMAX', 'CGP_DAG_BLAZY'):
      os.environ.pop(k, None)
  return n_cases, n_bad, worst


if __name__ == '__main__':
    n_cases, n_bad, worst = run_stress(float(sys.argv[1]) if len(sys.argv) > 1 else 60., int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print('%d cases, %d mismatches, worst %.2e' % (n_cases, n_bad, worst))
    sys.exit(1 if n_bad else 0)
