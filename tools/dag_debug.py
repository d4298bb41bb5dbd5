"""Guarded smoke run of the single-launch factorisation: each case in its own child process under a hard timeout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from cosmogp_amd import _lib
from cosmogp_amd import kernel as K
n, m = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
x = np.sort(rng.uniform(0., 0.15 * n, n)); y = np.sin(0.05 * x) + 0.1 * rng.standard_normal(n)
xn = rng.uniform(0., 0.15 * n, m) if m else None
print('create', flush=True)
dev = _lib.Device(0)
dev.set_data(1, x, y)
print('fit', flush=True)
t = time.time()
try:
    out = dev.fit(K.kernel_params_1d([1., 2.]), K.diag_term(n, np.full(n, 0.1), 0., 0.), xn, None, 0.)
    print('dag   ', out[0], out[1], time.time() - t, flush=True)
except Exception as e:
    print('ERROR', e, flush=True)
import os
os.environ['CGP_NO_DAG'] = '1'
d2 = _lib.Device(0)
d2.set_data(1, x, y)
o2 = d2.fit(K.kernel_params_1d([1., 2.]), K.diag_term(n, np.full(n, 0.1), 0., 0.), xn, None, 0.)
print('launch', o2[0], o2[1], flush=True)
if m:
    print('max |dmu|', np.abs(out[2] - o2[2]).max(), 'max |dvar|', np.abs(out[3] - o2[3]).max(), flush=True)
'''
if __name__ == '__main__':
    cases = sys.argv[1:] or ['256,0', '256,5', '1024,100']
    for c in cases:
        n, m = c.split(',')
        print('=== case', c, flush=True)
        try:
            r = subprocess.run([sys.executable, '-c', CHILD % ROOT, n, m], timeout=60, capture_output=True, text=True)
            print(r.stdout[-2000:], r.stderr[-2000:], 'rc', r.returncode, flush=True)
        except subprocess.TimeoutExpired as e:
            print('TIMEOUT (hard hang)', (e.stdout or b'')[-1000:], (e.stderr or b'')[-1000:], flush=True)
            break
