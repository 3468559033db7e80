// tl_math.h -- deterministic fp64 log10 / 10^x shared by the HIP kernels and their host-side
// emulation build.
//
// The reference calls glibc's log10() (psycho_1.c:245,254, psycho_3.c:158) and pow(10.0, x)
// (psycho_1.c:373) per frame.  A GPU has no glibc, and two different libms differ in the last
// ulp, so the device path carries its own implementation built ONLY from IEEE-754 + - * / and
// integer operations (no FMA contraction: the translation unit is compiled -ffp-contract=off).
// The same source compiled for the host gives bit-identical results, which is what lets the CPU
// test-suite check the device algorithm exactly.  Accuracy: < 1 ulp (log10), < 1 ulp (pow10) --
// measured against glibc in tests/test_tl_math.py; the residual last-ulp differences against
// glibc only matter when a dB value lands within 1 ulp of a decision threshold.
//
// Algorithm: the classic fdlibm/FreeBSD msun kernels (k_log.h + e_log10.c hi/lo recombination,
// e_exp.c rational form), restated.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define TL_HD __host__ __device__ __forceinline__
#else
#define TL_HD static inline
#endif

TL_HD uint64_t tl_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
TL_HD double tl_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

// log10(x) for finite x > 0 (normal or subnormal).  x <= 0 / inf / nan are not needed by the
// encoder (energies are clamped at 1e-20 before the call) and return -inf / x.
TL_HD double tl_log10(double x)
{
    const double two54 = 18014398509481984.0;
    const double ivln10hi = 4.34294481878168880939e-01;   // 0x3fdbcb7b15200000
    const double ivln10lo = 2.50829467116452752298e-11;   // 0x3dbb9438ca9aadd5
    const double log10_2hi = 3.01029995663611771306e-01;  // 0x3FD34413509F6000
    const double log10_2lo = 3.69423907715893078616e-13;  // 0x3D59FEF311F12B36
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = tl_d2u(x);
    int32_t hx = (int32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000) {
        if ((u << 1) == 0) return -1.0 / 0.0 * 1.0;        // log10(+-0) = -inf
        if (hx < 0) return (x - x) / (x - x);
        k -= 54;
        x *= two54;
        u = tl_d2u(x);
        hx = (int32_t)(u >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    if (u == 0x3ff0000000000000ull) return 0.0;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    u = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (u & 0xffffffffull);   // x or x/2 in [sqrt(.5), sqrt(2))
    x = tl_u2d(u);
    k += (i >> 20);
    double y = (double)k;
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    // k_log1p(f) = log(1+f) - f + f*f/2
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double r = s * (hfsq + (t2 + t1));
    double hi = f - hfsq;
    hi = tl_u2d(tl_d2u(hi) & 0xffffffff00000000ull);
    double lo = (f - hi) - hfsq + r;
    double val_hi = hi * ivln10hi;
    double y2 = y * log10_2hi;
    double val_lo = y * log10_2lo + (lo + hi) * ivln10lo + lo * ivln10hi;
    double ww = y2 + val_hi;
    val_lo += (y2 - ww) + val_hi;
    val_hi = ww;
    return val_lo + val_hi;
}

// 10^x for |x| < 300.
TL_HD double tl_pow10(double x)
{
    const double log2_10 = 3.32192809488736218171e+00;
    const double log10_2hi = 3.01029995663611771306e-01;  // 13 trailing zero bits: n*hi exact for |n| < 2^13
    const double log10_2lo = 3.69423907715893078616e-13;
    const double ln10hi = 2.3025850653648376;               // 0x40026bb1b8000000 (26 significant bits)
    const double ln10lo = 2.7629208037533617e-08;           // ln(10) - ln10hi
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    double t = x * log2_10;
    double n = (double)(long long)(t + (t >= 0 ? 0.5 : -0.5));
    // r = x - n*log10(2) in hi/lo, |r| <= 0.5*log10(2) ~ 0.1505
    double rhi = x - n * log10_2hi;
    double rlo = n * log10_2lo;
    double r = rhi - rlo;
    // e = r * ln(10), hi/lo
    double rh = tl_u2d(tl_d2u(r) & 0xfffffffff8000000ull);       // 26-bit head: rh*ln10hi exact
    double rt = (rhi - rh) - rlo;
    double hi = rh * ln10hi;
    double lo = rt * ln10hi + r * ln10lo;
    double e = hi + lo;
    double elo = (hi - e) + lo;
    // exp(e), fdlibm rational form; e small (|e| < 0.35)
    double tt = e * e;
    double c = e - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    double yv = 1.0 - ((-elo - (e * c) / (2.0 - c)) - e);     // = 1 + e + elo + e*c/(2-c)
    // scale by 2^n
    int ni = (int)n;
    uint64_t u = tl_d2u(yv);
    int ex = (int)((u >> 52) & 0x7ff) + ni;
    if (ex <= 0) return yv * tl_u2d((uint64_t)(ni + 1023 + 200) << 52) * tl_u2d((uint64_t)(1023 - 200) << 52);
    if (ex >= 0x7ff) return yv * tl_u2d(0x7fe0000000000000ull) * 2.0;
    return tl_u2d((u & 0x800fffffffffffffull) | ((uint64_t)ex << 52));
}
