"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle.c).

The reference has no golden vectors for this path and cannot be built or imported
here, so the fixtures are produced by the oracle and every one is
additionally pinned by a schoolbook negacyclic product in tests/test_oracle.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, n, depth, bits, base_bits  (reference test parameter sets)
    ("n4_d2_b17_base1", 4, 2, 17, 1),
    ("n16_d3_b18_base6", 16, 3, 18, 6),
    ("n128_d2_b16_base4", 128, 2, 16, 4),
    ("n128_d2_b17_base1", 128, 2, 17, 1),
    ("n16_d2_b51_base17", 16, 2, 51, 17),
]

for name, n, depth, bits, base in CASES:
    moduli = O.gen_crt_basis(n, depth, bits)
    seed = 0x6D7878 ^ (n * 1000 + bits)
    a = O.random_matrix(seed, 2, 3, moduli, n)
    b = O.random_matrix(seed + 1, 3, 2, moduli, n)
    m = O.random_matrix(seed + 2, 2, 2, moduli, n)
    a_eval = O.matrix_ntt(a, moduli)
    b_eval = O.matrix_ntt(b, moduli)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        moduli=np.asarray(moduli, dtype=np.uint64),
        n=n,
        base_bits=base,
        a_coeff=a,
        a_eval=a_eval,
        b_eval=b_eval,
        ab_eval=O.matmul(a_eval, b_eval, moduli),
        m_coeff=m,
        m_decomposed=O.decompose(m, moduli, base),
        gadget_eval=O.gadget_matrix(2, moduli, n, base),
    )
    print("wrote", name)

# seeded samplers (round 3 keying: eight draws per ChaCha20 block for uniform / bit / ternary, a stream per coefficient
# for the Gaussian): a 2 x 3 matrix and a column window of it, n = 16, three 18-bit limbs, plus a 51-bit pair
SEED = bytes((7 * i + 3) & 0xFF for i in range(32))
for name, n, depth, bits in (("samplers_n16_d3_b18", 16, 3, 18), ("samplers_n8_d2_b51", 8, 2, 51)):
    moduli = O.gen_crt_basis(n, depth, bits)
    out = {"moduli": np.asarray(moduli, dtype=np.uint64), "n": n, "seed": np.frombuffer(SEED, dtype=np.uint8)}
    for dist, sigma in (("uniform", 0.0), ("bit", 0.0), ("ternary", 0.0), ("gauss", 4.578)):
        out[dist] = O.sample_distribution(2, 3, moduli, n, dist, sigma, SEED)
        out[dist + "_window"] = O.sample_distribution(2, 2, moduli, n, dist, sigma, SEED, full_ncol=3, col_offset=1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)

# the same samplers under the REFERENCE device RNG's own keying (round 4: MXX_HIP_RNG_COMPAT=reference; restated from
# cuda/src/ChaCha.cu:104-167 + cuda/src/matrix/MatrixSampling.cu:239-289 in oracle/oracle_sampling.c): a fixture a CUDA build
# of the reference could be checked against, should one ever be run next to this library
for name, n, depth, bits in (("samplers_refkey_n16_d3_b18", 16, 3, 18), ("samplers_refkey_n8_d2_b51", 8, 2, 51)):
    moduli = O.gen_crt_basis(n, depth, bits)
    out = {"moduli": np.asarray(moduli, dtype=np.uint64), "n": n, "seed": np.frombuffer(SEED, dtype=np.uint8)}
    for dist, sigma in (("uniform", 0.0), ("bit", 0.0), ("ternary", 0.0), ("gauss", 4.578)):
        out[dist] = O.sample_distribution_refkey(2, 3, moduli, n, dist, sigma, SEED)
        out[dist + "_window"] = O.sample_distribution_refkey(2, 2, moduli, n, dist, sigma, SEED, full_ncol=3, col_offset=1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)
