"""CPU-only: pins the sampler restatement (oracle/oracle_sampling.c)."""
import numpy as np
import pytest

from oracle import oracle as O

SEED = bytes(range(32))


def test_chacha20_block_rfc8439_kat():
    """RFC 8439 section 2.3.2 test vector."""
    key = bytes(range(32))
    nonce = bytes([0, 0, 0, 9, 0, 0, 0, 0x4A, 0, 0, 0, 0])
    state = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
    state += [int.from_bytes(key[4 * i : 4 * i + 4], "little") for i in range(8)]
    state += [1] + [int.from_bytes(nonce[4 * i : 4 * i + 4], "little") for i in range(3)]
    out = O.chacha20_block(state)
    assert [hex(int(x)) for x in out[:4]] == ["0xe4e7f110", "0x15593bd1", "0x1fdd0f50", "0xc47120a3"]
    stream = b"".join(int(x).to_bytes(4, "little") for x in out)
    assert stream[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"
    assert stream[-8:].hex() == "cbd083e8a2503c4e"


def test_rng_stream_is_keyed():
    a = O.rng_stream(SEED, 1, 1, 1, 0x6F70656E66686531, 24)
    b = O.rng_stream(SEED, 1, 1, 1, 0x6F70656E66686531, 24)
    c = O.rng_stream(SEED, 2, 1, 1, 0x6F70656E66686531, 24)
    d = O.rng_stream(SEED, 1, 1, 1, 0x6F70656E66686532, 24)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)
    assert len(set(int(x) for x in a)) == 24  # block counter advances


def test_rng_streams_do_not_overlap():
    """Block b of stream (s0, s1) must not be block 0 of stream (s0 + b, s1): the reference's layout
    (counter = stream0, cuda/src/ChaCha.cu:138-149) has that overlap, and stream ids are consecutive
    (poly + 1, column + 1, tower + 1), so neighbouring polynomials would share keystream."""
    tag = 0x6F70656E66686532
    seen = set()
    for s0 in range(1, 6):
        for s1 in range(1, 4):
            words = O.rng_stream(SEED, s0, s1, 0, tag, 40)  # five 8-word blocks
            for b in range(5):
                blk = tuple(int(x) for x in words[8 * b : 8 * b + 8])
                assert blk not in seen, (s0, s1, b)
                seen.add(blk)
    # stream words above 2^32 still select distinct streams (48 bits each are kept)
    a = O.rng_stream(SEED, 1 << 33, 1, 0, tag, 8)
    b = O.rng_stream(SEED, 1, 1 << 33, 0, tag, 8)
    c = O.rng_stream(SEED, 0, 0, 0, tag, 8)
    assert not np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(b, c)


def test_detmath_against_libm():
    """Box-Muller's log / cos(2 pi u) (mxx_amd/csrc/detmath.h, compiled into both sides) stay within
    an ulp or two of libm: the samplers' normals are the same distribution, now reproducible."""
    import ctypes as C
    import math

    lib = O.lib()
    lib.orc_det_log.restype = C.c_double
    lib.orc_det_log.argtypes = [C.c_double]
    lib.orc_det_cos2pi.restype = C.c_double
    lib.orc_det_cos2pi.argtypes = [C.c_double]
    rng = np.random.default_rng(3)
    us = np.concatenate([rng.random(20000), rng.random(2000) * 2.0 ** -rng.integers(1, 52, 2000) + 2.0**-53,
                         [2.0**-53, 1 - 2.0**-53, 0.5, 0.25, 0.75, 0.125, 1 / 3]])
    for u in us:
        want = math.log(u)
        assert abs(lib.orc_det_log(u) - want) <= 2 * abs(np.spacing(want)) + 1e-300
        assert abs(lib.orc_det_cos2pi(u) - math.cos(2 * math.pi * u)) < 1e-15
    assert lib.orc_det_cos2pi(0.25) == 0.0 or abs(lib.orc_det_cos2pi(0.25)) < 1e-16
    assert lib.orc_det_cos2pi(0.5) == -1.0 and lib.orc_det_cos2pi(0.0) == 1.0


@pytest.mark.parametrize("sigma,mean", [(4.578, 0.0), (1.0, 0.3), (137.5, -20.25), (0.8, 0.0)])
def test_karney_moments(sigma, mean):
    x = O.karney_samples(SEED, 7, mean, sigma, 40000).astype(np.float64)
    assert abs(x.mean() - mean) < 5 * sigma / np.sqrt(len(x)) + 0.02
    assert abs(x.std() / max(sigma, (sigma**2 + 1 / 12) ** 0.5) - 1) < 0.05 or sigma < 1.0
    assert np.abs(x - mean).max() < 6.5 * sigma + 1


def test_distribution_sampler_properties_and_windows():
    n, moduli = 64, O.gen_crt_basis(64, 3, 20)
    full = O.sample_distribution(2, 5, moduli, n, "uniform", 0, SEED)
    assert all(int(full[:, :, l].max()) < q for l, q in enumerate(moduli))
    win = O.sample_distribution(2, 2, moduli, n, "uniform", 0, SEED, full_ncol=5, col_offset=2)
    assert np.array_equal(win, full[:, 2:4])  # src/sampler/gpu.rs:323-361
    for dist, lo, hi in [("bit", 0, 1), ("ternary", -1, 1)]:
        s = O.sample_distribution(2, 3, moduli, n, dist, 0, SEED)
        c = O.centered(s[:, :, 0], moduli[0])
        assert c.min() >= lo and c.max() <= hi and len(np.unique(c)) == hi - lo + 1
        for l, q in enumerate(moduli):  # the same integer in every limb
            assert np.array_equal(O.centered(s[:, :, l], q), c)
    g = O.sample_distribution(3, 3, moduli, n, "gauss", 3.2, SEED)
    c = O.centered(g[:, :, 0], moduli[0])
    assert np.abs(c).max() < 6 * 3.2 and abs(c.std() - 3.2) < 0.3  # |x| < 6 sigma (sampler/gpu.rs:363-400)


def test_uniform_sampler_keying_and_overflow_streams():
    """One keystream block serves eight uniform draws: the residue of coefficient i of limb l is word i of stream
    (gpoly + 1, 0) under sub-key (tag, l + 1); a rejected word is replaced by the first accepted word of the overflow
    stream (gpoly + 1, i + 1).  Recomputed here from raw keystream words, with moduli that reject one draw in nine."""
    from conftest import high_rejection_moduli

    n, tag = 16, 0x6f70656e66686531
    moduli = high_rejection_moduli(n, 2)
    rows, cols, full, off = 2, 2, 5, 3
    got = O.sample_distribution(rows, cols, moduli, n, "uniform", 0, SEED, full_ncol=full, col_offset=off)
    rejected = 0
    for r in range(rows):
        for c in range(cols):
            gpoly = r * full + off + c
            for l, q in enumerate(moduli):
                thr = (2**64 - 1) - (2**64 - 1) % q
                words = O.rng_stream(SEED, gpoly + 1, 0, l + 1, tag, n)
                for i in range(n):
                    x = int(words[i])
                    if x >= thr:
                        rejected += 1
                        x = next(int(w) for w in O.rng_stream(SEED, gpoly + 1, i + 1, l + 1, tag, 64) if int(w) < thr)
                    assert int(got[r, c, l, i]) == x % q
    assert rejected >= 3, "the overflow path was not exercised"
    # bit / ternary: the same positional word under sub-key (tag, 0), shared by all limbs
    for dist, t in (("bit", 0x6f70656e66686533), ("ternary", 0x6f70656e66686534)):
        s = O.sample_distribution(1, 1, moduli, n, dist, 0, SEED)
        words = O.rng_stream(SEED, 1, 0, 0, t, n)
        for i in range(n):
            z = int(words[i]) & 1 if dist == "bit" else (0, 1, -1)[int(words[i]) % 3]
            assert [int(v) for v in s[0, 0, :, i]] == [z % q for q in moduli]


@pytest.mark.parametrize("n,depth,bits,base", [(16, 2, 17, 1), (16, 2, 16, 4), (16, 3, 17, 5), (8, 2, 51, 17), (16, 2, 24, 12)])
def test_g_sampler_relation_on_cpu(n, depth, bits, base):
    """G * gauss_samp_gq(M) == M (gpu_dcrt_poly.rs:2381-2542)."""
    moduli = O.gen_crt_basis(n, depth, bits)
    M = O.random_matrix(9, 2, 2, moduli, n)
    c = (2**base + 1) * 4.578
    z = O.gauss_samp_gq(M, moduli, base, c, SEED)
    G = O.gadget_matrix(2, moduli, n, base)
    prod = O.matmul(G, O.matrix_ntt(z, moduli), moduli)
    assert np.array_equal(O.matrix_ntt(prod, moduli, inverse=True), M)
    zc = O.centered(z[:, :, 0], moduli[0])
    assert np.abs(zc).max() < 8 * c  # short digits


def test_p1_sampler_statistics():
    """conditional sampling reproduces mean c_scale*tp2 and the requested covariance diagonal."""
    n, moduli = 32, O.gen_crt_basis(32, 2, 24)
    d = 1
    rng = np.random.default_rng(1)

    def small(shape, bound):
        v = rng.integers(-bound, bound + 1, size=shape)
        return np.stack([np.mod(v, q).astype(np.uint64) for q in moduli], axis=-2)

    a = small((d, d, n), 50)
    dm = small((d, d, n), 50)
    b = small((d, d, n), 20)
    sigma, s = 10.0, 400.0
    sv, up = O.p1_covariance(a, b, dm, moduli, sigma, s, 4.0)
    assert sv.shape == (n, 2) and np.all(sv > 0)
    ac = O.centered(a[0, 0, 0], moduli[0])
    assert np.allclose(sv[:, 1] ** 2, s * s - sigma * sigma * O.centered(dm[0, 0, 0], moduli[0]))
    cols = 400
    tp2 = small((2 * d, cols, n), 1000)
    c_scale = -(sigma * sigma) / (s * s - sigma * sigma)
    out = O.sample_p1(tp2, moduli, sv, up, c_scale, SEED)
    z = O.centered(out[:, :, 0], moduli[0]).astype(np.float64)
    mu = c_scale * O.centered(tp2[:, :, 0], moduli[0])
    resid = z - mu
    assert abs(resid.mean()) < 5 * s / np.sqrt(resid.size) * 2
    assert abs(resid[1].std() / s - 1) < 0.1


def test_karney_lazy_deviates_tie_path_is_exercised_and_unbiased():
    """Round 3 keying: a uniform deviate is one 16-bit draw, its 37 low bits are drawn only when a comparison ties
    (one in 65536).  Over 400 k integers the tie path runs hundreds of times; the moments stay those of the discrete
    Gaussian (a wrong tie rule would bias every 4000th sample, far below what moments see - the point of this test is
    that the path RUNS here on the CPU; the GPU tests compare it bit for bit against this restatement)."""
    before = O.karney_ties()
    x = O.karney_samples(SEED, 11, 0.37, 4.578, 400000).astype(np.float64)
    ties = O.karney_ties() - before
    assert 30 <= ties <= 400, ties   # ~17 comparisons per integer / 65536
    assert abs(x.mean() - 0.37) < 0.03
    assert abs(x.std() - 4.578) < 0.03


def test_reference_keyed_sampler_restates_the_reference_layout():
    """oracle.sample_distribution_refkey (what MXX_HIP_RNG_COMPAT=reference must reproduce): the keying of
    cuda/src/ChaCha.cu:104-167 + cuda/src/matrix/MatrixSampling.cu:239-289 recomputed HERE from raw ChaCha20 blocks - the
    HChaCha20 sub-key of (seed, tag, stream2), stream0 = global polynomial index + 1 in the 64-bit block counter, stream1
    = coefficient + 1 in the nonce, first 64-bit word of the first block - for the uniform (with its rejection loop), bit
    and ternary distributions; column windows commute; the default keying gives a different matrix."""
    import ctypes as C

    n, moduli = 16, O.gen_crt_basis(16, 2, 17)
    seed = O._seed_words(SEED)

    def first_words(gpoly, coeff, stream2, tag, count):
        x = np.zeros(16, dtype=np.uint32)
        x[:4] = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
        for i in range(4):
            x[4 + 2 * i], x[5 + 2 * i] = int(seed[i]) & 0xFFFFFFFF, int(seed[i]) >> 32
        x[12], x[13], x[14], x[15] = tag & 0xFFFFFFFF, tag >> 32, stream2 & 0xFFFFFFFF, stream2 >> 32
        # HChaCha20 = the block function without the feed-forward: out - in
        blk = O.chacha20_block(x)
        h = (blk.astype(np.uint64) - x.astype(np.uint64)) & 0xFFFFFFFF
        st = np.zeros(16, dtype=np.uint32)
        st[:4] = x[:4]
        st[4:8], st[8:12] = h[0:4], h[12:16]
        out, ctr = [], gpoly + 1
        while len(out) < count:
            st[12], st[13], st[14], st[15] = ctr & 0xFFFFFFFF, ctr >> 32, (coeff + 1) & 0xFFFFFFFF, (coeff + 1) >> 32
            b = O.chacha20_block(st)
            out += [int(b[2 * j]) | (int(b[2 * j + 1]) << 32) for j in range(8)]
            ctr += 1
        return out[:count]

    full = {d: O.sample_distribution_refkey(2, 3, moduli, n, d, 0.0, SEED) for d in ("uniform", "bit", "ternary")}
    for row, col, i in ((0, 0, 0), (1, 2, 15), (0, 1, 7)):
        gpoly = row * 3 + col
        for l, q in enumerate(moduli):
            ws = first_words(gpoly, i, l + 1, 0x6F70656E66686531, 16)
            thr = (2**64 - 1) - ((2**64 - 1) % q)
            want = next(w for w in ws if w < thr) % q
            assert int(full["uniform"][row, col, l, i]) == want
        w = first_words(gpoly, i, 0, 0x6F70656E66686533, 1)[0]
        assert int(full["bit"][row, col, 0, i]) == (w & 1)
        w = first_words(gpoly, i, 0, 0x6F70656E66686534, 1)[0] % 3
        assert int(O.centered(full["ternary"][row, col, 0], moduli[0])[i]) == (0 if w == 0 else (1 if w == 1 else -1))
    for d, sigma in (("uniform", 0.0), ("gauss", 4.578), ("bit", 0.0), ("ternary", 0.0)):
        a = O.sample_distribution_refkey(2, 5, moduli, n, d, sigma, SEED)
        win = O.sample_distribution_refkey(2, 2, moduli, n, d, sigma, SEED, full_ncol=5, col_offset=2)
        assert np.array_equal(win, a[:, 2:4])  # src/sampler/gpu.rs:323-361
        assert not np.array_equal(a, O.sample_distribution(2, 5, moduli, n, d, sigma, SEED))
    g = O.centered(O.sample_distribution_refkey(4, 4, moduli, 64, "gauss", 4.578, SEED)[:, :, 0], moduli[0]).astype(np.float64)
    assert abs(g.mean()) < 0.5 and 4.0 < g.std() < 5.2
