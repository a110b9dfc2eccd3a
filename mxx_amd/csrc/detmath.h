/* detmath.h — log and cos(2*pi*u) as fixed sequences of IEEE-754 double operations.
 *
 * The Box-Muller step of the samplers (rng.h: rng_standard_normal; reference
 * cuda/src/matrix/MatrixSampling.cu:20-29, MatrixTrapdoor.cu:701-833) is the only place on the
 * path that calls transcendental functions.  libm and the device math library round them
 * differently in the last place, which made the G-lattice sampler the one sampler whose output
 * could not be replayed bit for bit on the CPU.  The functions below use only +, -, *, / and
 * integer bit manipulation in a fixed order, so any IEEE-754 implementation that is built without
 * floating-point contraction (-ffp-contract=off, both libgpupoly.so and the CPU restatement) returns
 * the same bits.  Accuracy is ~1 ulp (tests/test_oracle_sampling.py checks against libm); the
 * samplers need determinism, not correct rounding.  sqrt is IEEE-exact on both sides already.
 *
 * Plain C so that the CPU restatement (oracle/oracle_sampling.c, test infrastructure) compiles the
 * very same text.
 */
#ifndef MXX_DETMATH_H
#define MXX_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DETMATH_FN __host__ __device__ static inline
#else
#define DETMATH_FN static inline
#endif

DETMATH_FN uint64_t detmath_bits(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, sizeof(u));
    return u;
}

DETMATH_FN double detmath_from_bits(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, sizeof(x));
    return x;
}

/* natural logarithm of a positive, finite, normal double (the samplers pass u in [2^-53, 1)).
 * Argument reduction x = 2^k * m with m in [sqrt(1/2), sqrt(2)), then with f = m - 1,
 * s = f / (2 + f): log(m) = f - f^2/2 + s * (f^2/2 + R(s^2)), R an even minimax polynomial on
 * [0, 0.1716] (the classic reduction; coefficients are the well-known degree-14 set). */
DETMATH_FN double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double c1 = 6.666666666666735130e-01, c2 = 3.999999999940941908e-01, c3 = 2.857142874366239149e-01,
                 c4 = 2.222219843214978396e-01, c5 = 1.818357216161805012e-01, c6 = 1.531383769920937332e-01,
                 c7 = 1.479819860511658591e-01;
    uint64_t bits = detmath_bits(x);
    uint32_t hx = (uint32_t)(bits >> 32);
    int32_t k = (int32_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    /* mantissa >= sqrt(2): halve it and bump the exponent, so m lies in [sqrt(1/2), sqrt(2)) */
    const uint32_t up = (hx + 0x95f64u) & 0x100000u;
    k += (int32_t)(up >> 20);
    hx |= up ^ 0x3ff00000u;
    const double m = detmath_from_bits(((uint64_t)hx << 32) | (bits & 0xffffffffull));
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (c2 + w * (c4 + w * c6));
    const double t2 = z * (c1 + w * (c3 + w * (c5 + w * c7)));
    const double r = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + r) + dk * ln2_lo)) - f);
}

/* sin and cos on [-pi/4, pi/4] (odd / even minimax polynomials) */
DETMATH_FN double detmath_sin_kernel(double t) {
    const double s1 = -1.66666666666666324348e-01, s2 = 8.33333333332248946124e-03, s3 = -1.98412698298579493134e-04,
                 s4 = 2.75573137070700676789e-06, s5 = -2.50507602534068634195e-08, s6 = 1.58969099521155010221e-10;
    const double z = t * t;
    const double v = z * t;
    const double r = s2 + z * (s3 + z * (s4 + z * (s5 + z * s6)));
    return t + v * (s1 + z * r);
}

DETMATH_FN double detmath_cos_kernel(double t) {
    const double k1 = 4.16666666666666019037e-02, k2 = -1.38888888888741095749e-03, k3 = 2.48015872894767294178e-05,
                 k4 = -2.75573143513906633035e-07, k5 = 2.08757232129817482790e-09, k6 = -1.13596475577881948265e-11;
    const double z = t * t;
    const double r = z * (k1 + z * (k2 + z * (k3 + z * (k4 + z * (k5 + z * k6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * r);
}

/* cos(2*pi*u) for u in [0, 1]: the quadrant is taken from 4u exactly (4u and 4u - round(4u) are
 * exact in binary floating point), so only the final angle d*(pi/2), |d| <= 1/2, is rounded. */
DETMATH_FN double det_cos2pi(double u) {
    const double half_pi = 1.57079632679489661923;
    const double w = 4.0 * u;
    const int q = (int)(w + 0.5);      /* round half up; w >= 0 */
    const double d = w - (double)q;    /* in [-1/2, 1/2], exact */
    const double t = d * half_pi;
    switch (q & 3) {
        case 0: return detmath_cos_kernel(t);
        case 1: return -detmath_sin_kernel(t);
        case 2: return -detmath_cos_kernel(t);
        default: return detmath_sin_kernel(t);
    }
}

#endif /* MXX_DETMATH_H */
