import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, cosmogp_amd
from conftest import load_golden
g = load_golden('inverse_n96'); K = g['K_singular']; n=len(K)
dev = cosmogp_amd._lib.default_device()
w, v = np.linalg.eigh(K)
for th in (1e-15, 1e-14, 1e-13, 1e-11, 1e-8):
    inv, ld, nd = dev.svd_inverse(K, thresh=th)
    keep = np.abs(w) > th
    want = (v[:, keep] / w[keep]) @ v[:, keep].T
    print('thresh %g dropped %d (eigh would drop %d) |KK+K-K| %.2e  |inv-want|/|want| %.2e  |inv| %.2e asym %.2e' % (
        th, nd, (~keep).sum(), np.abs(K@inv@K-K).max(), np.abs(inv-want).max()/np.abs(want).max(), np.abs(inv).max(), np.abs(inv-inv.T).max()))
