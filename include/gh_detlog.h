/*
 * gh_detlog.h -- a bit-reproducible double-precision log10 for host and device.
 *
 * Why: Gretel's path extension picks the arg-max of sums of log10(conditional)
 * (reference gretel/gretel.py:155-174) and accumulates log10(marginal)
 * (gretel/gretel.py:185-186).  The reference calls libm's log10 through
 * Python's math.log10.  libm (glibc) and the GPU math library (ocml) are both
 * "within an ulp or so" but not bit-identical to each other, so a near tie
 * could resolve differently on the two sides and the recovered SNP sequence
 * would stop being bit-exact.  This header gives ONE sequence of IEEE-754
 * binary64 operations (+, -, *, / and integer bit moves; no fma contraction --
 * compile with -ffp-contract=off) that gcc on x86-64 and hipcc on gfx950
 * evaluate identically.  Error < 1 ulp (tests/test_detlog.py checks it against
 * a 50-digit reference and against libm).
 *
 * Algorithm: the classic argument reduction x = 2^k * (1+f),
 * sqrt(2)/2 < 1+f < sqrt(2), log(1+f) = f - f^2/2 + s*(f^2/2 + R(s^2)),
 * s = f/(2+f), R a degree-7 minimax polynomial (Sun fdlibm's published
 * coefficients), followed by a hi/lo split multiplication by 1/ln(10) and
 * addition of k*log10(2) in two pieces.
 *
 * Special cases: +0/-0 -> -inf, x<0 -> nan, +inf -> +inf, nan -> nan.
 */
#ifndef GH_DETLOG_H
#define GH_DETLOG_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GH_HD __host__ __device__ __forceinline__
#else
#define GH_HD static inline
#endif

GH_HD uint64_t gh_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
GH_HD double gh_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* the arithmetic of gh_log10 for a normal, positive, finite x (k0 = exponent carried in by the caller's
 * subnormal scaling); straight-line code, so that several calls interleave on the GPU */
GH_HD double gh_log10_normal(double x, int32_t k0)
{
    const double ivln10hi = 4.34294481878168880939e-01;   /* 0x3fdbcb7b15200000 */
    const double ivln10lo = 2.50829467116452752298e-11;   /* 0x3dbb9438ca9aadd5 */
    const double log10_2hi = 3.01029995663611771306e-01;  /* 0x3FD34413509F6000 */
    const double log10_2lo = 3.69423907715893078616e-13;  /* 0x3D59FEF311F12B36 */
    const double Lg1 = 6.666666666666735130e-01;
    const double Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01;
    const double Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01;
    const double Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;


    uint64_t u = gh_d2u(x);
    int32_t hx = (int32_t)(u >> 32);
    uint32_t lx = (uint32_t)u;
    int32_t k = k0;

    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;       /* 1 iff mantissa > sqrt(2) */
    u = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | lx;   /* normalise x or x/2 */
    x = gh_u2d(u);
    k += (i >> 20);
    double y = (double)k;
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;

    /* r = log(1+f) - f + f*f/2 */
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double r = s * (hfsq + R);

    /* hi + lo = f - hfsq + r, hi with its low 32 bits cleared */
    double hi = f - hfsq;
    hi = gh_u2d(gh_d2u(hi) & 0xffffffff00000000ULL);
    double lo = (f - hi) - hfsq + r;

    double val_hi = hi * ivln10hi;
    double y2 = y * log10_2hi;
    double val_lo = y * log10_2lo + (lo + hi) * ivln10lo + lo * ivln10hi;

    /* extra-precision sum y2 + val_hi */
    double ww = y2 + val_hi;
    val_lo += (y2 - ww) + val_hi;
    val_hi = ww;

    return val_lo + val_hi;
}

/* 1 iff gh_log10(x) takes the straight-line path: 2^-1022 <= x < inf */
GH_HD int gh_log10_is_normal(double x)
{
    const int32_t hx = (int32_t)(gh_d2u(x) >> 32);
    return hx >= 0x00100000 && hx < 0x7ff00000;
}

GH_HD double gh_log10(double x)
{
    const double two54 = 1.80143985094819840000e+16;      /* 2^54 */
    uint64_t u = gh_d2u(x);
    int32_t hx = (int32_t)(u >> 32);
    uint32_t lx = (uint32_t)u;
    int32_t k = 0;

    if (hx < 0x00100000) {                       /* x < 2^-1022, zero, or negative */
        if (((hx & 0x7fffffff) | (int32_t)(lx != 0)) == 0)
            return -gh_u2d(0x7ff0000000000000ULL);           /* log(+-0) = -inf */
        if (hx < 0)
            return gh_u2d(0x7ff8000000000000ULL);            /* log(-#) = nan */
        k -= 54;
        x *= two54;                              /* subnormal: scale up */
        hx = (int32_t)(gh_d2u(x) >> 32);
    }
    if (hx >= 0x7ff00000)
        return x + x;                            /* inf or nan */
    return gh_log10_normal(x, k);                /* log10(1) comes out as +0 by itself */
}


#endif /* GH_DETLOG_H */
