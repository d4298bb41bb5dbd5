"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads, exports every
symbol include/cosmogp_hip.h declares, and the product path has no CPU fallback."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cosmogp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cgp_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    from cosmogp_amd import _lib
    assert sorted(_lib.ABI_SYMBOLS) == declared_symbols()


def test_library_loads_and_exports_every_symbol():
    from cosmogp_amd import _lib
    lib = _lib.load_library()
    assert lib.cgp_version() == 1
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), name


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, 'include', 'cosmogp_hip.h')).read()
    assert 'torch' not in text.replace('no torch', '') and 'at::' not in text


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'cosmogp_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
                assert 'cosmogp_oracle' not in src, f


def test_core_package_does_not_import_torch():
    code = ("import sys; import cosmogp_amd; "
            "assert 'torch' not in sys.modules, 'core package must not import torch'")
    subprocess.check_call([sys.executable, '-c', code], cwd=ROOT)


@pytest.mark.skipif(os.path.exists('/dev/kfd') and os.access('/dev/kfd', os.R_OK), reason='GPU present')
def test_fails_loudly_without_gpu():
    import cosmogp_amd
    from cosmogp_amd import _lib
    assert _lib.device_count() == 0
    with pytest.raises(cosmogp_amd.HipUnavailable):
        _lib.Device(0)
    x = np.linspace(0., 1., 8)
    with pytest.raises(cosmogp_amd.HipUnavailable):
        cosmogp_amd.rbf_kernel_1d(x, [1., 1.])
    with pytest.raises(cosmogp_amd.HipUnavailable):
        cosmogp_amd.cholesky_inverse(np.eye(4))
    gp = cosmogp_amd.gaussian_process(np.sin(x), x, y_err=np.full(8, 0.1))      # construction is host only
    with pytest.raises(cosmogp_amd.HipUnavailable):
        gp.get_prediction(svd_method=False)
