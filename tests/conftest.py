import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available() -> bool:
    try:
        from mxx_amd import _ffi

        return _ffi.detected_gpu_device_count() > 0
    except Exception:
        return False


class _HipEnv:
    """MXX_HIP_* switches are cached per context at creation: set / unset + gpupoly_reload_env."""

    def __init__(self):
        self._saved = {}

    def _reload(self):
        from mxx_amd import _ffi

        _ffi.reload_env()

    def set(self, name, value):
        self._saved.setdefault(name, os.environ.get(name))
        os.environ[name] = value
        self._reload()

    def unset(self, name):
        self._saved.setdefault(name, os.environ.get(name))
        os.environ.pop(name, None)
        self._reload()

    def restore(self):
        for name, old in self._saved.items():
            if old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = old
        self._saved.clear()
        self._reload()


@pytest.fixture
def hip_env():
    env = _HipEnv()
    yield env
    env.restore()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    import mxx_amd

    if mxx_amd.detected_gpu_device_count() == 0:
        pytest.fail("no GPU visible: gpu-marked tests must run on an MI355X (no CPU fallback exists)")
    return mxx_amd


_PARAM_CACHE = {}


def make_params(gpu_mod, oracle_mod, n, depth, bits, base_bits):
    key = (n, depth, bits, base_bits)
    if key not in _PARAM_CACHE:
        moduli = oracle_mod.gen_crt_basis(n, depth, bits)
        _PARAM_CACHE[key] = gpu_mod.GpuDCRTPolyParams(n, moduli, base_bits)
    return _PARAM_CACHE[key]


def rand_matrix(oracle_mod, seed, rows, cols, moduli, n):
    return oracle_mod.random_matrix(seed, rows, cols, moduli, n)
