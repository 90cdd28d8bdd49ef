/*
 * gh_detlog.h -- the host's log10, bit for bit, on host and device.
 *
 * Why: Gretel's path extension picks the arg-max of sums of log10(conditional)
 * (reference gretel/gretel.py:155-174) and accumulates log10(marginal)
 * (gretel/gretel.py:185-186).  The reference calls libm's log10 through Python's
 * math.log10 (gretel/gretel.py:2).  Candidates whose evidence is mathematically
 * equal but arrives through different operands (the same counts in another order,
 * a.b = c.d) end up an ulp or two apart -- or exactly tied -- depending on how
 * every single log10 was rounded: tests/test_log10_audit.py counts such steps
 * (86 within 4 ulp among 1.2e8, next to 11 000 exact ties), and a log10 that is
 * merely "accurate to an ulp" decides about one step in 5e6 differently from
 * libm -- one path in a hundred at 50 000 SNPs (rounds 1-3 shipped such a log10;
 * profiles/r4_log10_audit_fdlibm.json is its census).  So this header is not
 * another log10: it is glibc's, restated operation by operation --
 *
 *   log10(x):  glibc sysdeps/ieee754/dbl-64/e_log10.c (__ieee754_log10):
 *              x = 2^k m, m in [1,2) for k >= 0, [0.5,1) for k < 0 (y = k or k+1);
 *              (y*log10_2lo + ivln10*log(m)) + y*log10_2hi, plain multiplies and adds;
 *   log(m):    glibc sysdeps/ieee754/dbl-64/e_log.c (__log, glibc >= 2.28; Szabolcs
 *              Nagy's table-driven log), in the form x86-64 hosts with FMA run it
 *              (the ifunc picks __log_fma wherever the CPU has FMA and AVX2: every
 *              x86-64 server since 2013): r = fma(z, 1/c, -1) and the fused
 *              multiply-adds exactly where that build has them (read off the
 *              instruction sequence of Ubuntu GLIBC 2.35-0ubuntu3.11's libm.so.6);
 *   table:     include/gh_logtab.inc (tools/gen_logtab.py).
 *
 * Provenance and licence.  The table-driven log() restated here -- its decomposition x = 2^k z, the 128 subintervals of
 * [0x1.6p-1, 0x1.6p0), the two polynomials and their coefficients, and the table { 1/c, log c } -- is Szabolcs Nagy's
 * double-precision log of the Arm Optimized Routines (math/log.c, math/log_data.c, LOG_TABLE_BITS = 7, LOG_POLY_ORDER = 6,
 * LOG_POLY1_ORDER = 12), which glibc 2.28 imported unchanged as sysdeps/ieee754/dbl-64/e_log.c / e_log_data.c:
 *
 *     Copyright (c) 2018, Arm Limited.
 *     SPDX-License-Identifier: MIT OR Apache-2.0 WITH LLVM-exception
 *
 * This header is an independent restatement of that published algorithm (no source text of either project is reproduced);
 * the constants are the algorithm's numerical data.  The table's construction is checkable without either source and
 * tools/gen_logtab.py --verify (tests/test_detlog.py) checks it with exact arithmetic: every c lies within 2^29 ulp of its
 * subinterval's centre (Arm's search range for 1/c) and every log c IS round(2^43 ln(1/invc)) / 2^43 -- the second column
 * follows from the first.  The first build of this header read the table out of a distribution's libm binary; the committed
 * values are the same 256 doubles, now carried with their origin, and `gen_logtab.py --from-libm` remains only as a
 * cross-check that the running libm still holds them.  The wrapper log10 (split of the exponent, ivln10, log10_2hi/lo) is
 * fdlibm's e_log10.c as glibc ships it (Copyright (C) 1993 by Sun Microsystems, Inc.; "Permission to use, copy, modify, and
 * distribute this software is freely granted, provided that this notice is preserved").
 *
 * Every operation below is an IEEE-754 binary64 +, -, *, or an EXPLICIT fma;
 * compile with -ffp-contract=off so the compiler adds none of its own.  gcc on
 * x86-64 and hipcc on gfx950 then evaluate it identically, and identically to
 * the libm it restates: tests/test_detlog.py compares it with the running libm
 * on 4e7 arguments (every exponent, the near-one interval, every table
 * interval's edges, the operands the hot path produces) on the CPU, and the GPU
 * suite compares the device function with the host's libm the same way
 * (tests/test_gpu_detlog.py).  On a host whose libm takes another road (no FMA;
 * a glibc before 2.28; another libc) that test says so -- the reference's own
 * answers on near-ties differ between such hosts too.
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

/* { 1/c, log c } per subinterval: glibc's __log_data.tab */
static const double gh_logtab_host[256] = {
#include "gh_logtab.inc"
};
#if defined(__HIPCC__)
__device__ static const double gh_logtab_dev[256] = {
#include "gh_logtab.inc"
};
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define GH_LOGTAB gh_logtab_dev
#else
#define GH_LOGTAB gh_logtab_host
#endif

/* how a caller wants the logarithm scheduled (a compile-time constant at every call; the VALUES never depend on it):
 * GH_LOG_BOTH   evaluate both branches of log() and select -- straight-line code, several logarithms interleave (k_rwseg, k_rw:
 *               the logarithms sit in a chain of dependent steps of one workgroup and there are registers to spare);
 * GH_LOG_SERIAL one logarithm after the other, the branch taken as a branch (k_marg<T,true>, k_lt: a wave per SIMD more is
 *               worth more than instruction-level parallelism -- five interleaved logarithms cost 13 VGPRs). */
#define GH_LOG_BOTH 1
#define GH_LOG_SERIAL 2
/* glibc's log() for 0.5 <= x < 2 (what its log10 hands it), the FMA build.  glibc branches on the argument; here both
 * (mode & GH_LOG_BOTH) both branches are evaluated and one is selected (the same operations on the same operands either
 * way): straight-line code, so that the five or six logarithms a lane of k_rw / k_rwseg takes interleave.
 * tab = { 1/c, log c } x 128 (gh_logtab_*: the kernels hand in their copy in LDS). */
GH_HD double gh_log_reduced_tab(double x, const double *tab, int mode)
{
    const int both = (mode & GH_LOG_BOTH) != 0;
    const uint64_t ix = gh_d2u(x);
    /* (a) 1 - 0x1p-4 <= x < 1 + 0x1.09p-4: a polynomial in r = x - 1, its leading terms in two pieces (x = 1 gives +0 by itself) */
    const int near_one = ix - 0x3fee000000000000ULL < 0x0003090000000000ULL;
    double y_near = 0.0;
    if (both || near_one) {
        const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3,
                     B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3,
                     B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4,
                     B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        double a = __builtin_fma(r, B2, B1);
        double b = __builtin_fma(r, B5, B4);
        double c = __builtin_fma(r, B8, B7);
        a = __builtin_fma(r2, B3, a);
        b = __builtin_fma(r2, B6, b);
        c = __builtin_fma(r2, B9, c);
        c = __builtin_fma(r3, B10, c);
        double y = __builtin_fma(c, r3, b);
        y = __builtin_fma(y, r3, a);
        /* rhi = the top 26 bits of r; hi + lo = r - rhi*rhi/2 */
        const double t = __builtin_fma(r, 0x1p27, r);
        const double rhi = __builtin_fma(-0x1p27, r, t);
        const double rlo = r - rhi;
        const double rr = rhi * rhi;
        const double hi = __builtin_fma(rr, B0, r);
        double lo = __builtin_fma(rr, B0, r - hi);
        lo = __builtin_fma(B0 * rlo, r + rhi, lo);
        y = __builtin_fma(y, r3, lo);
        y_near = hi + y;
    }
    /* (b) x = 2^k z, z in [0x1.6p-1, 0x1.6p0), split into 128 intervals; log x = k ln2 + log c + log1p(z/c - 1) */
    double y_main = 0.0;
    if (both || !near_one) {
        const double ln2hi = 0x1.62e42fefa3800p-1, ln2lo = 0x1.ef35793c76730p-45;
        const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3,
                     A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
        const uint64_t tmp = ix - 0x3fe6000000000000ULL;
        const int i = (int)((tmp >> 45) & 127);
        const int32_t k = (int32_t)((int64_t)tmp >> 52);
        const double z = gh_u2d(ix - (tmp & 0xfff0000000000000ULL));
        const double kd = (double)k;
        const double invc = tab[2 * i], logc = tab[2 * i + 1];
        const double r = __builtin_fma(z, invc, -1.0);
        const double w = __builtin_fma(kd, ln2hi, logc);
        const double hi = w + r;
        const double lo = __builtin_fma(kd, ln2lo, (w - hi) + r);
        const double r2 = r * r;
        const double p = __builtin_fma(r, A2, A1);
        const double q = __builtin_fma(r, A4, A3);
        const double lo2 = __builtin_fma(r2, A0, lo);
        const double pq = __builtin_fma(q, r2, p);
        const double y = __builtin_fma(r * r2, pq, lo2);
        y_main = y + hi;
    }
    return near_one ? y_near : y_main;
}

/* log10 of a normal, positive, finite x (k0 = exponent carried in by the caller's subnormal scaling):
 * glibc's __ieee754_log10 behind its special cases */
GH_HD double gh_log10_normal_tab(double x, int32_t k0, const double *tab, int mode)
{
    const double ivln10 = 0x1.bcb7b1526e50ep-2;       /* 1/ln 10 */
    const double log10_2hi = 0x1.34413509f6000p-2;
    const double log10_2lo = 0x1.9fef311f12b36p-42;
    const uint64_t u = gh_d2u(x);
    const int32_t k = k0 + (int32_t)(u >> 52) - 1023;
    const int32_t i = (int32_t)((uint32_t)k >> 31);               /* 1 iff k < 0: then m in [0.5, 1) and y = k + 1 */
    const double m = gh_u2d((u & 0x000fffffffffffffULL) | ((uint64_t)(0x3ff - i) << 52));
    const double y = (double)(k + i);
    const double z = y * log10_2lo + ivln10 * gh_log_reduced_tab(m, tab, mode);
    const double out = z + y * log10_2hi;
#if defined(__HIP_DEVICE_COMPILE__)
    if (mode & GH_LOG_SERIAL) __builtin_amdgcn_sched_barrier(0);
#endif
    return out;
}
GH_HD double gh_log10_normal(double x, int32_t k0) { return gh_log10_normal_tab(x, k0, GH_LOGTAB, GH_LOG_SERIAL); }

/* 1 iff gh_log10(x) takes the straight-line path: 2^-1022 <= x < inf */
GH_HD int gh_log10_is_normal(double x)
{
    const int32_t hx = (int32_t)(gh_d2u(x) >> 32);
    return hx >= 0x00100000 && hx < 0x7ff00000;
}

GH_HD double gh_log10_tab(double x, const double *tab, int mode)
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
    return gh_log10_normal_tab(x, k, tab, mode);
}
GH_HD double gh_log10(double x) { return gh_log10_tab(x, GH_LOGTAB, GH_LOG_SERIAL); }


#endif /* GH_DETLOG_H */
