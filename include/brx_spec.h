/*
 * include/brx_spec.h -- the NUMERICAL SPEC shared by the HIP kernels and the CPU oracle.
 *
 * Badread draws from CPython's MT19937 (`random`) and numpy's legacy RandomState
 * (badread/simulate.py:34-36).  A GPU cannot consume one sequential stream from thousands of
 * wavefronts, so this build replaces both with a counter-based generator: every draw is a pure
 * function of (seed, read index, stream id, draw index).  That makes results independent of batch
 * size, launch geometry and GPU count, and lets the CPU oracle reproduce the GPU bit-for-bit.
 *
 * Everything here uses only IEEE-754 +,-,*,/,sqrt and integer ops, so that gcc (oracle) and
 * hipcc (device) agree to the last bit.  BOTH must be compiled with -ffp-contract=off and without
 * fast-math.  No libm transcendental is called: log/exp are restated below (fdlibm-style kernels).
 *
 * Distributions restated (same laws as the numpy / random calls in the reference, different
 * algorithms and therefore different streams -- parity is distributional):
 *   gamma      np.random.gamma      badread/fragment_lengths.py:51   (Marsaglia-Tsang)
 *   beta       np.random.beta       badread/identities.py:89, simulate.py:387 (two gammas)
 *   normal     np.random.normal     badread/identities.py:92          (Marsaglia polar)
 *   geometric  np.random.geometric  badread/simulate.py:466,475,478   (inversion)
 *   randint / random / choices      badread/misc.py:156-182, simulate.py:174,189,215,294
 */
#ifndef BRX_SPEC_H
#define BRX_SPEC_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BRX_HD __host__ __device__ inline
#else
#define BRX_HD static inline
#endif

/* ------------------------------------------------------------------ stream ids */
enum {
    BRX_ST_PLAN  = 1,   /* sequential draws of the per-read planner (lengths, coordinates, glitches) */
    BRX_ST_BASES = 2,   /* position-addressed random bases: ctr1 = segment serial, ctr0 = pos/64      */
    BRX_ST_MUT   = 3,   /* mutate loop: ctr0/ctr1 = iteration index                                 */
    BRX_ST_WIN   = 4,   /* in-loop alignment window position: ctr0 = alignment serial               */
    BRX_ST_QS    = 5,   /* qscore sampling: ctr0 = read position / 4                                */
    BRX_ST_NAME  = 6    /* 128-bit read name                                                        */
};

/* ------------------------------------------------------------------ Philox4x32-10 */
BRX_HD void brx_mulhilo32(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

BRX_HD void brx_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        brx_mulhilo32(0xD2511F53u, c0, &hi0, &lo0);
        brx_mulhilo32(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* one 128-bit block for (seed, read, stream, index) */
BRX_HD void brx_draw4(uint64_t seed, uint64_t read, uint32_t stream, uint64_t index, uint32_t out[4]) {
    uint32_t ctr[4], key[2];
    ctr[0] = (uint32_t)index;
    ctr[1] = (uint32_t)(index >> 32);
    ctr[2] = (uint32_t)read;
    ctr[3] = ((uint32_t)(read >> 32) & 0x00FFFFFFu) | (stream << 24);
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    brx_philox4x32_10(ctr, key, out);
}

/* floor(x * n / 2^64): uniform integer in [0, n) from 64 random bits (bias < n / 2^64) */
BRX_HD uint64_t brx_mulhi64(uint64_t x, uint64_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(x, n);
#else
    return (uint64_t)(((unsigned __int128)x * (unsigned __int128)n) >> 64);
#endif
}

/* sequential generator used by the planner (one per read) */
typedef struct {
    uint64_t seed, read;
    uint32_t stream;
    uint64_t index;
    uint32_t buf[4];
    int have;
} brx_rng;

BRX_HD void brx_rng_init(brx_rng *g, uint64_t seed, uint64_t read, uint32_t stream) {
    g->seed = seed; g->read = read; g->stream = stream; g->index = 0; g->have = 0;
    g->buf[0] = g->buf[1] = g->buf[2] = g->buf[3] = 0;
}

BRX_HD uint32_t brx_next_u32(brx_rng *g) {
    if (g->have == 0) {
        brx_draw4(g->seed, g->read, g->stream, g->index, g->buf);
        g->index += 1;
        g->have = 4;
    }
    uint32_t v = g->buf[4 - g->have];
    g->have -= 1;
    return v;
}

BRX_HD uint64_t brx_next_u64(brx_rng *g) {
    uint64_t lo = brx_next_u32(g);
    uint64_t hi = brx_next_u32(g);
    return (hi << 32) | lo;
}

/* uniform double in [0,1) with 53 random bits (same construction as MT genrand_res53) */
BRX_HD double brx_next_double(brx_rng *g) {
    uint32_t a = brx_next_u32(g) >> 5, b = brx_next_u32(g) >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

/* uniform integer in [0, n), n >= 1 */
BRX_HD uint64_t brx_next_below(brx_rng *g, uint64_t n) {
    return brx_mulhi64(brx_next_u64(g), n);
}

/* ------------------------------------------------------------------ bit casts */
BRX_HD uint64_t brx_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
BRX_HD double brx_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

/* ------------------------------------------------------------------ log / exp (fdlibm kernels) */
/* natural log of a positive, finite, normal double; relative error < 1 ulp */
BRX_HD double brx_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = brx_d2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u) {            /* subnormal or zero: scale up (never hit by the samplers) */
        if ((u << 1) == 0) return -1.0e300 * 1.0e300;
        x = x * 18014398509481984.0;   /* 2^54 */
        u = brx_d2u(x); hx = (uint32_t)(u >> 32); k -= 54;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    x = brx_u2d(u);
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* e^x for finite x; returns 0 below -745, saturates above 709 */
BRX_HD double brx_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x > 709.0) return 1.0e300 * 1.0e300;
    if (x < -745.0) return 0.0;
    double hi, lo;
    int k;
    double ax = x < 0.0 ? -x : x;
    if (ax > 0.34657359027997264) {            /* 0.5 ln2 */
        if (ax >= 1.0397207708399179) k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
        else k = (x < 0.0) ? -1 : 1;
        hi = x - (double)k * ln2HI;
        lo = (double)k * ln2LO;
        x = hi - lo;
    } else { k = 0; hi = x; lo = 0.0; }
    double xx = x * x;
    double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    if (k < -1021) {                           /* two-step scale into the subnormal range */
        y = y * brx_u2d((uint64_t)(0x3ff - 1000) << 52);
        k += 1000;
        if (k < -1021) return 0.0;
    }
    return y * brx_u2d((uint64_t)(0x3ff + k) << 52);
}

BRX_HD double brx_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dsqrt_rn(x);
#else
    return __builtin_sqrt(x);
#endif
}

/* ------------------------------------------------------------------ samplers */
BRX_HD double brx_normal(brx_rng *g) {
    for (;;) {
        double u1 = 2.0 * brx_next_double(g) - 1.0;
        double u2 = 2.0 * brx_next_double(g) - 1.0;
        double s = u1 * u1 + u2 * u2;
        if (s >= 1.0 || s == 0.0) continue;
        return u1 * brx_sqrt(-2.0 * brx_log(s) / s);
    }
}

/* standard gamma(shape), shape > 0 */
BRX_HD double brx_std_gamma(brx_rng *g, double shape) {
    double boost = 1.0;
    if (shape < 1.0) {
        double u = 1.0 - brx_next_double(g);               /* (0,1] */
        boost = brx_exp(brx_log(u) / shape);
        shape += 1.0;
    }
    double d = shape - 1.0 / 3.0;
    double c = 1.0 / brx_sqrt(9.0 * d);
    for (;;) {
        double x = brx_normal(g);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = 1.0 - brx_next_double(g);               /* (0,1] */
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return boost * d * v;
        if (brx_log(u) < 0.5 * x2 + d * (1.0 - v + brx_log(v))) return boost * d * v;
    }
}

BRX_HD double brx_beta(brx_rng *g, double a, double b) {
    double x = brx_std_gamma(g, a);
    double y = brx_std_gamma(g, b);
    return x / (x + y);
}

/* geometric on {1,2,...} with success probability p (np.random.geometric) */
BRX_HD int64_t brx_geometric(brx_rng *g, double p) {
    if (p >= 1.0) return 1;
    double u = 1.0 - brx_next_double(g);                   /* (0,1] */
    double r = brx_log(u) / brx_log(1.0 - p);
    int64_t n = (int64_t)r;
    if ((double)n < r) n += 1;                             /* ceil */
    if (n < 1) n = 1;
    return n;
}

/* Python round() on a non-negative double: round-half-to-even */
BRX_HD int64_t brx_round_half_even(double x) {
    double fl = (double)(int64_t)x;                        /* x >= 0 -> floor */
    double diff = x - fl;
    int64_t n = (int64_t)fl;
    if (diff > 0.5) return n + 1;
    if (diff < 0.5) return n;
    return (n & 1) ? n + 1 : n;
}

/* random base code 0..3 at position `pos` of random segment `serial` of a read */
BRX_HD uint32_t brx_random_base(uint64_t seed, uint64_t read, uint32_t serial, uint64_t pos) {
    uint32_t o[4];
    brx_draw4(seed, read, BRX_ST_BASES, ((uint64_t)serial << 32) | (pos >> 6), o);
    return (o[(pos >> 4) & 3] >> (2 * (pos & 15))) & 3u;
}

#endif /* BRX_SPEC_H */
