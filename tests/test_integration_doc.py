"""INTEGRATION.md section 1 is executable documentation: the reference-side ctypes stub it shows (`cosmogp/_hip.py`,
replacing the seam at cosmogp/Gaussian_process.py:6-9 and the kernel binding at :140-154) is extracted from the
markdown AS WRITTEN and run against the reference's golden vectors.  If the document rots, this test notices."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, ulp_diff


def _python_blocks_of_section_1():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text.split('## 1. Operator level', 1)[1].split('## 2. Fused level', 1)[0]
    return re.findall(r'```python\n(.*?)```', sec, flags=re.S)


def test_integration_doc_has_the_stub():
    """(no GPU) the section holds the two code blocks and they at least compile; every entry point they bind is declared
    in include/cosmogp_hip.h"""
    blocks = _python_blocks_of_section_1()
    assert len(blocks) == 2
    header = open(os.path.join(ROOT, 'include', 'cosmogp_hip.h')).read()
    for b in blocks:
        compile(b, 'INTEGRATION.md', 'exec')
        for sym in set(re.findall(r'_l\.(cgp_[a-z_]+)', b)):
            assert re.search(r'\b%s\s*\(' % sym, header), sym


@pytest.mark.gpu
def test_integration_doc_stub_runs_against_goldens():
    """The stub exactly as a maintainer of the reference would paste it (only the library path is made absolute):
    rbf_kernel_1d against k1d_n64 (<= 4 ulp, the tolerance of tests/test_gpu_parity.py), cholesky_inverse and svd_inverse
    against inverse_n96 (inv_matrix.py:21-31 / :4-18), LinAlgError on the singular matrix like the reference."""
    from cosmogp_amd import _lib
    blocks = _python_blocks_of_section_1()
    ns = {}
    for b in blocks:
        assert "'libcosmogp_hip.so'" in b or '_l.' in b
        exec(compile(b.replace("'libcosmogp_hip.so'", repr(_lib.LIB_PATH)), 'INTEGRATION.md', 'exec'), ns)
    g = load_golden('k1d_n64')
    ng, fl = float(g['nugget']), float(g['floor'])
    assert ulp_diff(ns['rbf_kernel_1d'](g['x'], g['hyp']), g['K_plain']) <= 4
    assert ulp_diff(ns['rbf_kernel_1d'](g['x'], g['hyp'], nugget=ng, floor=fl, y_err=g['yerr']), g['K_noise']) <= 4
    H = ns['rbf_kernel_1d'](g['x'], g['hyp'], new_x=g['xnew'])
    assert H.shape == g['H'].shape and ulp_diff(H, g['H']) <= 4
    assert ulp_diff(ns['rbf_kernel_1d'](g['xnew'], g['hyp'], nugget=ng), g['Kss']) <= 4
    gi = load_golden('inverse_n96')
    inv, ld = ns['cholesky_inverse'](gi['K'], return_logdet=True)
    np.testing.assert_allclose(inv, gi['chol_inv'], rtol=0, atol=1e-8 * np.abs(gi['chol_inv']).max())
    assert abs(ld - float(gi['chol_logdet'])) <= 1e-11 * abs(ld)
    assert np.array_equal(ns['cholesky_inverse'](gi['K']), inv)
    with pytest.raises(np.linalg.LinAlgError):
        ns['cholesky_inverse'](gi['K_singular'])
    sinv, sld = ns['svd_inverse'](gi['K'], return_logdet=True)
    np.testing.assert_allclose(sinv, gi['svd_inv'], rtol=0, atol=1e-10 * np.abs(gi['svd_inv']).max())
    assert abs(sld - float(gi['svd_logdet'])) <= 1e-11 * abs(sld)
    assert ns['_l'].cgp_destroy(ns['_h']) == 0
