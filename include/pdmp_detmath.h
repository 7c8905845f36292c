/*
 * pdmp_detmath.h -- the numerical contract shared by the gfx950 kernels and the CPU oracle.
 *
 * The reference (ZigZagBoomerang.jl) draws its uniforms from a *sequential* Xoroshiro128Plus stream
 * (src/ZigZagBoomerang.jl:6-10, src/sfact.jl:166) and takes logarithms with Julia's libm.  Neither
 * exists on a GPU, and neither is bit-reproducible across a host compiler and gfx950.  This header
 * replaces exactly those two things -- and nothing else -- by functions that give the SAME BITS on
 * x86-64 (gcc/clang, -ffp-contract=off) and on gfx950 (hipcc, -ffp-contract=off):
 *
 *   pdmp_philox4x32_10   counter-based RNG (Salmon et al., SC'11), keyed (seed_lo, seed_hi)
 *   pdmp_u01             draw #n of a chain's stream as a double in the OPEN interval (0,1)
 *   pdmp_log             natural log for positive normal doubles, < 1 ulp, only + - * / on doubles
 *   pdmp_exp             exp, < 1 ulp, only + - * / on doubles (logistic targets)
 *   pdmp_randexp         -log(u)                       (replaces Random.randexp, src/poissontime.jl:77)
 *   pdmp_randn           Box-Muller normal             (replaces Random.randn,   src/dynamics.jl:115)
 *   pdmp_randn2          both Box-Muller branches of one block (the d-vector refresh of the non-factorised samplers)
 *
 * Everything else (poisson_time, ab, the event loop) is restated INDEPENDENTLY in oracle/ and in the
 * HIP kernels, so that a transcription error on one side shows up as a parity failure.
 *
 * IEEE-754 binary64 +,-,*,/ and sqrt are correctly rounded on both targets; no fused multiply-add is
 * used anywhere in this file (a*b+c is always two roundings), so the only requirement on the compiler
 * is that it neither contracts nor re-associates: build with -ffp-contract=off and without -ffast-math.
 */
#ifndef PDMP_DETMATH_H
#define PDMP_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PDMP_HD __host__ __device__ static inline
#else
#define PDMP_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

/* ---------------------------------------------------------------- bit casts */
PDMP_HD uint64_t pdmp_f2u(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}
PDMP_HD double pdmp_u2f(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof x);
    return x;
}

/* ---------------------------------------------------------------- Philox4x32-10 */
#define PDMP_PHILOX_M0 0xD2511F53u
#define PDMP_PHILOX_M1 0xCD9E8D57u
#define PDMP_PHILOX_W0 0x9E3779B9u
#define PDMP_PHILOX_W1 0xBB67AE85u

typedef struct {
    uint32_t v[4];
} pdmp_u32x4;

PDMP_HD pdmp_u32x4 pdmp_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                       uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PDMP_PHILOX_M0 * (uint64_t)c0;
        uint64_t p1 = (uint64_t)PDMP_PHILOX_M1 * (uint64_t)c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += PDMP_PHILOX_W0;
        k1 += PDMP_PHILOX_W1;
    }
    pdmp_u32x4 out;
    out.v[0] = c0;
    out.v[1] = c1;
    out.v[2] = c2;
    out.v[3] = c3;
    return out;
}

/* Stream identifiers: the third counter word separates independent uses of one chain's key. */
#define PDMP_STREAM_MAIN 0u   /* the seeded `rng` of the reference drivers (src/sfact.jl:166)     */
#define PDMP_STREAM_GLOBAL 1u /* draws the reference takes from Julia's GLOBAL rng (src/sfact.jl:80) */
#define PDMP_STREAM_INIT 2u   /* x0/theta0 synthesis for benchmarks                              */

/* 64 random bits for draw #n of stream `stream` of the chain keyed by `seed`. */
PDMP_HD uint64_t pdmp_bits64(uint64_t seed, uint32_t stream, uint64_t n) {
    pdmp_u32x4 r = pdmp_philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), stream, 0u, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
    return ((uint64_t)r.v[0] << 32) | (uint64_t)r.v[1];
}

/* 52 random bits -> (m + 1/2) * 2^-52, exact in binary64, in [2^-53, 1 - 2^-53]: log(u) is finite. */
PDMP_HD double pdmp_bits_to_u01(uint64_t bits) {
    return ((double)(bits >> 12) + 0.5) * 0x1.0p-52;
}

PDMP_HD double pdmp_u01(uint64_t seed, uint32_t stream, uint64_t n) {
    return pdmp_bits_to_u01(pdmp_bits64(seed, stream, n));
}

/* Uniform integer in [0, n) for n < 2^32 (replaces rand(1:n), src/sfact.jl:80): multiply-shift. */
PDMP_HD uint32_t pdmp_randint(uint64_t seed, uint32_t stream, uint64_t n_draw, uint32_t n) {
    uint64_t b = pdmp_bits64(seed, stream, n_draw);
    return (uint32_t)(((b >> 32) * (uint64_t)n) >> 32);
}

/* ---------------------------------------------------------------- log */
/*
 * Classic argument reduction x = 2^k (1+f), sqrt(1/2) < 1+f <= sqrt(2), s = f/(2+f),
 * log(1+f) = 2s + s*R(s^2) with a degree-14 minimax polynomial (the Sun fdlibm coefficients).
 * Valid for positive, finite, NORMAL x (all callers pass u in [2^-53, 1) or ratios thereof).
 */
PDMP_HD double pdmp_log(double x) {
    const double ln2_hi = 0x1.62e42fee00000p-1;  /* 6.93147180369123816490e-01 */
    const double ln2_lo = 0x1.a39ef35793c76p-33; /* 1.90821492927058770002e-10 */
    const double Lg1 = 0x1.5555555555593p-1;     /* 6.666666666666735130e-01 */
    const double Lg2 = 0x1.999999997fa04p-2;     /* 3.999999999940941908e-01 */
    const double Lg3 = 0x1.2492494229359p-2;     /* 2.857142874366239149e-01 */
    const double Lg4 = 0x1.c71c51d8e78afp-3;     /* 2.222219843214978396e-01 */
    const double Lg5 = 0x1.7466496cb03dep-3;     /* 1.818357216161805012e-01 */
    const double Lg6 = 0x1.39a09d078c69fp-3;     /* 1.531383769920937332e-01 */
    const double Lg7 = 0x1.2f112df3e5244p-3;     /* 1.479819860511658591e-01 */

    uint64_t ix = pdmp_f2u(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    uint32_t lx = (uint32_t)ix;
    int32_t k = (int32_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u; /* mantissa above sqrt(2): halve it */
    hx |= (i ^ 0x3ff00000u);
    k += (int32_t)(i >> 20);
    double m = pdmp_u2f(((uint64_t)hx << 32) | (uint64_t)lx);

    double f = m - 1.0;
    double hfsq = (0.5 * f) * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return ((((s * (hfsq + R)) + (dk * ln2_lo)) - hfsq) + f) + (dk * ln2_hi);
}

/* ---------------------------------------------------------------- exp */
/*
 * exp(x) = 2^k exp(r), k = round(x/ln2), r = x - k ln2 (two-part constant), exp(r) = 1 + r + r c/(2 - c) with the
 * classic degree-5 polynomial c = r - r^2 (P1 + r^2 (P2 + ...)) (Sun fdlibm).  < 1 ulp; only + - * / on doubles, so the
 * result is bit-identical on x86-64 and gfx950.  Saturates: x > 709.78 -> +Inf, x < -745.13 -> 0.
 * Replaces Base.exp inside the logistic targets' sigmoid (scripts/logistic.jl:33).
 */
PDMP_HD double pdmp_exp(double x) {
    const double ln2_hi = 0x1.62e42fee00000p-1;   /* 6.93147180369123816490e-01 */
    const double ln2_lo = 0x1.a39ef35793c76p-33;  /* 1.90821492927058770002e-10 */
    const double invln2 = 0x1.71547652b82fep+0;   /* 1.44269504088896338700e+00 */
    const double P1 = 0x1.555555555553ep-3;       /*  1.66666666666666019037e-01 */
    const double P2 = -0x1.6c16c16bebd93p-9;      /* -2.77777777770155933842e-03 */
    const double P3 = 0x1.1566aaf25de2cp-14;      /*  6.61375632143793436117e-05 */
    const double P4 = -0x1.bbd41c5d26bf1p-20;     /* -1.65339022054652515390e-06 */
    const double P5 = 0x1.6376972bea4d0p-25;      /*  4.13813679705723846039e-08 */
    if (x != x) return x;
    if (x > 709.782712893384) return __builtin_inf();
    if (x < -745.1332191019411) return 0.0;
    const double kf = x * invln2 + ((x < 0) ? -0.5 : 0.5);
    const int32_t k = (int32_t)kf; /* truncation toward zero of (x/ln2 +- 1/2) = round to nearest */
    const double dk = (double)k;
    const double hi = x - dk * ln2_hi;
    const double lo = dk * ln2_lo;
    const double r = hi - lo;
    const double z = r * r;
    const double c = r - z * (P1 + z * (P2 + z * (P3 + z * (P4 + z * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    /* scale by 2^k in two steps so that subnormal results round once */
    if (k >= -1021 && k <= 1023) {
        return y * pdmp_u2f((uint64_t)(1023 + k) << 52);
    } else if (k > 1023) {
        return (y * 0x1.0p+1023) * pdmp_u2f((uint64_t)(1023 + (k - 1023)) << 52);
    } else {
        return (y * pdmp_u2f((uint64_t)(1023 + (k + 1000)) << 52)) * 0x1.0p-1000;
    }
}

PDMP_HD double pdmp_randexp_from_u(double u) {
    return -pdmp_log(u);
}

/* ---------------------------------------------------------------- sin/cos of 2*pi*v, v in [0,1) */
/*
 * Box-Muller needs cos(2 pi v), sin(2 pi v).  Reduce v to an octant exactly (v*8 is exact), then
 * evaluate Taylor/minimax polynomials on |r| <= pi/4.  Accuracy ~1e-16 absolute; determinism is what
 * matters here, not the last ulp.
 */
PDMP_HD double pdmp_sin_poly(double r) {
    /* sin r = r + r^3 * (S1 + r^2 (S2 + ...)), fdlibm __kernel_sin coefficients */
    const double S1 = -0x1.5555555555549p-3;  /* -1.66666666666666324348e-01 */
    const double S2 = 0x1.111111110f8a6p-7;   /*  8.33333333332248946124e-03 */
    const double S3 = -0x1.a01a019c161d5p-13; /* -1.98412698298579493134e-04 */
    const double S4 = 0x1.71de357b1fe7dp-19;  /*  2.75573137070700676789e-06 */
    const double S5 = -0x1.ae5e68a2b9cebp-26; /* -2.50507602534068634195e-08 */
    const double S6 = 0x1.5d93a5acfd57cp-33;  /*  1.58969099521155010221e-10 */
    double z = r * r;
    double v = z * r;
    double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return r + v * (S1 + z * p);
}
PDMP_HD double pdmp_cos_poly(double r) {
    /* cos r = 1 - r^2/2 + r^4 * (C1 + r^2 (C2 + ...)), fdlibm __kernel_cos coefficients */
    const double C1 = 0x1.555555555554cp-5;   /*  4.16666666666666019037e-02 */
    const double C2 = -0x1.6c16c16c15177p-10; /* -1.38888888888741095749e-03 */
    const double C3 = 0x1.a01a019cb1590p-16;  /*  2.48015872894767294178e-05 */
    const double C4 = -0x1.27e4f809c52adp-22; /* -2.75573143513906633035e-07 */
    const double C5 = 0x1.1ee9ebdb4b1c4p-29;  /*  2.08757232129817482790e-09 */
    const double C6 = -0x1.8fae9be8838d4p-37; /* -1.13596475577881948265e-11 */
    double z = r * r;
    double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * p;
}
PDMP_HD void pdmp_sincos2pi(double v, double* s_out, double* c_out) {
    const double pi_4 = 0x1.921fb54442d18p-1; /* pi/4 rounded */
    double v8 = v * 8.0;                      /* exact */
    int32_t oct = (int32_t)v8;                /* 0..7 */
    double fr = v8 - (double)oct;             /* exact, in [0,1) */
    /* angle = (oct + fr) * pi/4 ; fold to r in [-pi/4, pi/4] around the nearest multiple of pi/2 */
    int32_t q = (oct + 1) >> 1;               /* 0..4 quadrant index */
    double r = ((oct & 1) ? (fr - 1.0) : fr) * pi_4;
    double sr = pdmp_sin_poly(r);
    double cr = pdmp_cos_poly(r);
    switch (q & 3) {
    case 0: *s_out = sr;  *c_out = cr;  break;
    case 1: *s_out = cr;  *c_out = -sr; break;
    case 2: *s_out = -sr; *c_out = -cr; break;
    default: *s_out = -cr; *c_out = sr; break;
    }
}

/* sin and cos of an arbitrary angle |x| < 2^20: three-term Cody-Waite reduction by pi/2 (every product n*pio2_k is exact),
 * then the kernels above.  Replaces Base.sincos in the Boomerang rotation (src/sfact.jl:29-36, src/dynamics.jl:29-36). */
PDMP_HD void pdmp_sincos(double x, double* s_out, double* c_out) {
    const double invpio2 = 0x1.45f306dc9c883p-1; /* 2/pi */
    const double pio2_1 = 0x1.921fb54400000p+0;  /* first 33 bits of pi/2 */
    const double pio2_2 = 0x1.0b4611a600000p-34; /* next 33 bits */
    const double pio2_3 = 0x1.3198a2e000000p-69; /* next 33 bits */
    const double pio2_3t = 0x1.b839a252049c1p-104;
    const double fnr = x * invpio2 + ((x < 0) ? -0.5 : 0.5);
    const int32_t n = (int32_t)fnr;
    const double fn = (double)n;
    double r = x - fn * pio2_1;
    r = r - fn * pio2_2;
    r = r - fn * pio2_3;
    r = r - fn * pio2_3t;
    const double sr = pdmp_sin_poly(r);
    const double cr = pdmp_cos_poly(r);
    switch (n & 3) {
    case 0: *s_out = sr;  *c_out = cr;  break;
    case 1: *s_out = cr;  *c_out = -sr; break;
    case 2: *s_out = -sr; *c_out = -cr; break;
    default: *s_out = -cr; *c_out = sr; break;
    }
}

#if defined(__HIPCC__)
#define PDMP_SQRT(x) __builtin_sqrt(x)
#else
#define PDMP_SQRT(x) __builtin_sqrt(x)
#endif

/* Standard normal from two uniforms (Box-Muller, cosine branch only: one normal per draw pair). */
PDMP_HD double pdmp_randn_from_u(double u1, double u2) {
    double rad = PDMP_SQRT(-2.0 * pdmp_log(u1));
    double s, c;
    pdmp_sincos2pi(u2, &s, &c);
    (void)s;
    return rad * c;
}

/* Normal draw #n of a stream: consumes the two 52-bit halves... of two Philox words of ONE block. */
PDMP_HD double pdmp_randn(uint64_t seed, uint32_t stream, uint64_t n) {
    pdmp_u32x4 r = pdmp_philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), stream, 0u, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
    double u1 = pdmp_bits_to_u01(((uint64_t)r.v[0] << 32) | (uint64_t)r.v[1]);
    double u2 = pdmp_bits_to_u01(((uint64_t)r.v[2] << 32) | (uint64_t)r.v[3]);
    return pdmp_randn_from_u(u1, u2);
}

/* Both Box-Muller branches of block #n: z0 = r cos 2πu2, z1 = r sin 2πu2 (two independent standard normals per Philox block). */
PDMP_HD void pdmp_randn2(uint64_t seed, uint32_t stream, uint64_t n, double* z0, double* z1) {
    pdmp_u32x4 r = pdmp_philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), stream, 0u, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
    double u1 = pdmp_bits_to_u01(((uint64_t)r.v[0] << 32) | (uint64_t)r.v[1]);
    double u2 = pdmp_bits_to_u01(((uint64_t)r.v[2] << 32) | (uint64_t)r.v[3]);
    double rad = PDMP_SQRT(-2.0 * pdmp_log(u1));
    double s, c;
    pdmp_sincos2pi(u2, &s, &c);
    *z0 = rad * c;
    *z1 = rad * s;
}

#endif /* PDMP_DETMATH_H */
