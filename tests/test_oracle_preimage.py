"""CPU: the restatement of the whole trapdoor / preimage chain satisfies the reference's predicates
(src/sampler/trapdoor/gpu.rs:547-811): A*[R;E;I] == G, A*x == u exactly, x small."""
import math

import numpy as np
import pytest


@pytest.mark.parametrize("n,depth,bits,base,d,cols", [(64, 2, 16, 4, 1, 2), (32, 2, 24, 12, 2, 3), (16, 2, 51, 17, 1, 1)])
def test_cpu_preimage_chain(oracle, n, depth, bits, base, d, cols):
    moduli = oracle.gen_crt_basis(n, depth, bits)
    sigma = 4.578
    seed = bytes(range(32))
    r, e, a = oracle.trapdoor_gen(moduli, n, base, sigma, d, seed)
    k, c, s = oracle.preimage_params(moduli, n, base, sigma, d)
    L = len(moduli)
    ident = np.zeros((d * k, d * k, L, n), dtype=np.uint64)
    for i in range(d * k):
        ident[i, i] = 1
    rei = np.concatenate([r, e, ident], axis=0)
    assert np.array_equal(oracle.matmul(a, rei, moduli), oracle.gadget_matrix(d, moduli, n, base))
    target = oracle.matrix_ntt(oracle.random_matrix(7, d, cols, moduli, n), moduli)
    x = oracle.preimage(moduli, n, base, sigma, r, e, a, target, seed)
    assert x.shape[:2] == (d * (k + 2), cols)
    assert np.array_equal(oracle.matmul(a, x, moduli), target)
    xc = oracle.matrix_ntt(x, moduli, inverse=True)
    Q = math.prod(int(q) for q in moduli)
    worst = 0
    for row in range(xc.shape[0]):
        for col in range(cols):
            for i in range(0, n, max(1, n // 8)):
                v = oracle.crt_reconstruct([int(xc[row, col, l, i]) for l in range(L)], moduli)
                worst = max(worst, min(v, Q - v))
    assert 0 < worst < 6.5 * s + 6.5 * math.sqrt(d * k * n) * 6.0 * sigma * c
    x2 = oracle.preimage(moduli, n, base, sigma, r, e, a, target, bytes(range(1, 33)))
    assert not np.array_equal(x, x2) and np.array_equal(oracle.matmul(a, x2, moduli), target)
