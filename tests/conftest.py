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


def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 2^64."""
    if n < 2:
        return False
    for b in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % b == 0:
            return n == b
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for b in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(b, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def high_rejection_moduli(n: int, count: int) -> list:
    """Primes = 1 (mod 2n) just below 2^64 / 4.5: 2^64 mod q is about q / 2, so one 64-bit draw in nine is
    rejected by the uniform sampler (24-bit limbs reject one in 2^40: their overflow path would never run in a test)."""
    out, q = [], (1 << 64) * 2 // 9
    q -= (q - 1) % (2 * n)
    while len(out) < count:
        if is_prime(q):
            out.append(q)
        q -= 2 * n
    return out
