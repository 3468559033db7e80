// tl_math.h -- deterministic fp64 log10 / 10^x shared by the HIP kernels and their host-side
// emulation build.
//
// The reference calls glibc's log10() (psycho_1.c:245,254, psycho_3.c:158) and pow(10.0, x)
// (psycho_1.c:373) per frame.  A GPU has no glibc, and two different libms differ in the last
// ulp, so the device path carries its own implementation built ONLY from IEEE-754 + - * / and
// integer operations (no FMA contraction: the translation unit is compiled -ffp-contract=off).
// The same source compiled for the host gives bit-identical results, which is what lets the CPU
// test-suite check the device algorithm exactly.  Accuracy: within 2 ulp (log10) / 1 ulp (pow10, log, exp,
// sin, cos, atan2) of glibc -- tests/test_emu_parity.py; the residual last-ulp differences against
// glibc only matter when a dB value lands within 1 ulp of a decision threshold.
//
// Algorithm: the classic fdlibm/FreeBSD msun kernels (k_log.h + e_log10.c hi/lo recombination,
// e_exp.c rational form), restated.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
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

// log10(x) for positive NORMAL finite x only, straight-line (no branches): the same operations as tl_log10 on
// that domain, so the same bits.  The encoder's spectra are clamped at 1e-20 before the call; for any other input
// the result is unspecified (and discarded by the caller's select), never a trap.
TL_HD double tl_log10_pn(double x)
{
    const double ivln10hi = 4.34294481878168880939e-01, ivln10lo = 2.50829467116452752298e-11;
    const double log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = tl_d2u(x);
    const bool one = u == 0x3ff0000000000000ull;
    int32_t hx = (int32_t)(u >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i = (hx + 0x95f64) & 0x100000;
    u = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (u & 0xffffffffull);
    x = tl_u2d(u);
    k += (i >> 20);
    const double y = (double)k;
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double r = s * (hfsq + (t2 + t1));
    double hi = f - hfsq;
    hi = tl_u2d(tl_d2u(hi) & 0xffffffff00000000ull);
    const double lo = (f - hi) - hfsq + r;
    double val_hi = hi * ivln10hi;
    const double y2 = y * log10_2hi;
    double val_lo = y * log10_2lo + (lo + hi) * ivln10lo + lo * ivln10hi;
    const double ww = y2 + val_hi;
    val_lo += (y2 - ww) + val_hi;
    val_hi = ww;
    const double res = val_lo + val_hi;
    return one ? 0.0 : res;
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

// ------------------------------------------------------------------------------------------
// natural log, exp, sin, cos, atan2 for psy model 2 (psycho_2.c:127-133,184,192,235,245, fft.c:1258-1263).
// Same construction as above: fdlibm kernels restated with IEEE + - * / only.

TL_HD double tl_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = tl_d2u(x);
    int32_t hx = (int32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000) {
        if ((u << 1) == 0) return -1.0 / 0.0 * 1.0;
        if (hx < 0) return (x - x) / (x - x);
        k -= 54; x *= 18014398509481984.0; u = tl_d2u(x); hx = (int32_t)(u >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = tl_u2d(((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (u & 0xffffffffull));
    k += (i >> 20);
    const double f = x - 1.0, dk = (double)k;
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

TL_HD double tl_exp(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x > 709.0) return 1.0 / 0.0 * 1.0;
    if (x < -745.0) return 0.0;
    const double t = x * invln2;
    const int k = (int)(t + (t >= 0 ? 0.5 : -0.5));
    const double hi = x - k * ln2HI, lo = k * ln2LO;
    const double r = hi - lo;
    const double tt = r * r;
    const double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k == 0) return y;
    uint64_t u = tl_d2u(y);
    const int ex = (int)((u >> 52) & 0x7ff) + k;
    if (ex <= 0) return y * tl_u2d((uint64_t)(k + 1023 + 200) << 52) * tl_u2d((uint64_t)(1023 - 200) << 52);
    return tl_u2d((u & 0x800fffffffffffffull) | ((uint64_t)ex << 52));
}

// reduce x to r = y0 + y1 in [-pi/4, pi/4], return the quadrant (|x| < ~1e5: two/three-term Cody-Waite)
TL_HD int tl_rem_pio2(double x, double *y0, double *y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21,
                 pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    const double t0 = x * invpio2;
    const int n = (int)(t0 + (t0 >= 0 ? 0.5 : -0.5));
    const double fn = (double)n;
    double r = x - fn * pio2_1, w = fn * pio2_1t;
    double y = r - w;
    const int ex = (int)((tl_d2u(x) >> 52) & 0x7ff);
    int ey = (int)((tl_d2u(y) >> 52) & 0x7ff);
    if (ex - ey > 16) {                       // cancellation: second iteration
        double t = r;
        w = fn * pio2_2; r = t - w; w = fn * pio2_2t - ((t - r) - w);
        y = r - w;
        ey = (int)((tl_d2u(y) >> 52) & 0x7ff);
        if (ex - ey > 49) {                   // third iteration
            t = r;
            w = fn * pio2_3; r = t - w; w = fn * pio2_3t - ((t - r) - w);
            y = r - w;
        }
    }
    *y0 = y; *y1 = (r - y) - w;
    return n;
}
TL_HD double tl_ksin(double x, double y, int iy)
{   // both forms are evaluated and one is selected: a wave's lanes are on both sides of the reduction threshold anyway
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x, r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double plain = x + v * (S1 + z * r);
    const double tail = x - ((z * (0.5 * y - v * r) - y) - v * S1);
    return iy == 0 ? plain : tail;
}
TL_HD double tl_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const int32_t ix = (int32_t)(tl_d2u(x) >> 32) & 0x7fffffff;
    const double zr = z * r - x * y;
    const double small = 1.0 - (0.5 * z - zr);                      // |x| < 0.3
    const double qx = ix > 0x3fe90000 ? 0.28125 : tl_u2d((uint64_t)(uint32_t)(ix - 0x00200000) << 32);
    const double hz = 0.5 * z - qx, a = 1.0 - qx;
    return ix < 0x3FD33333 ? small : a - (hz - zr);
}
TL_HD void tl_sincos(double x, double *sn, double *cs)
{
    const int32_t ix = (int32_t)(tl_d2u(x) >> 32) & 0x7fffffff;
    const bool reduce = ix > 0x3fe921fb;                             // |x| > pi/4
    double r0, r1;
    const int nr = tl_rem_pio2(x, &r0, &r1);                         // harmless below the threshold (its result is dropped)
    const double y0 = reduce ? r0 : x, y1 = reduce ? r1 : 0.0;
    const int n = reduce ? nr : 0;
    const double s = tl_ksin(y0, y1, reduce ? 1 : 0), c = tl_kcos(y0, y1);
    const bool swap = n & 1;
    const double a = swap ? c : s, b = swap ? s : c;                 // quadrants: (s, c), (c, -s), (-s, -c), (-c, s)
    *sn = (n & 2) ? -a : a;
    *cs = (((n + 1) & 2) != 0) ? -b : b;
}

TL_HD double tl_atan(double x)
{   // fdlibm s_atan.c, written without branches: the five argument ranges differ in the quotient that is formed (num/den), in
    // the constant (hi, lo) added back and in nothing else, so both are selected and ONE division runs whatever mix of
    // ranges the lanes of a wave are in.  The small-argument range (no reduction) is num/den = |x|/1 with hi = lo = 0:
    // hi - ((t*p - lo) - t) is then t - t*p, the range's own formula, and the sign is applied at the end as in the others.
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    const int32_t hx = (int32_t)(tl_d2u(x) >> 32), ix = hx & 0x7fffffff;
    const double ax = tl_u2d(tl_d2u(x) & 0x7fffffffffffffffull);
    const bool r_small = ix < 0x3fdc0000, r0 = ix < 0x3fe60000, r1 = ix < 0x3ff30000, r2 = ix < 0x40038000;
    const double num = r_small ? ax : r0 ? 2.0 * ax - 1.0 : r1 ? ax - 1.0 : r2 ? ax - 1.5 : -1.0;
    const double den = r_small ? 1.0 : r0 ? 2.0 + ax : r1 ? ax + 1.0 : r2 ? 1.0 + 1.5 * ax : ax;
    const double hi = r_small ? 0.0 : r0 ? 4.63647609000806093515e-01 : r1 ? 7.85398163397448278999e-01
                    : r2 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00;
    const double lo = r_small ? 0.0 : r0 ? 2.26987774529616870924e-17 : r1 ? 3.06161699786838301793e-17
                    : r2 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17;
    const double t = num / den;
    const double z = t * t, w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const double zz = hi - ((t * (s1 + s2) - lo) - t);
    double res = hx < 0 ? -zz : zz;
    res = ix < 0x3e400000 ? x : res;                                  // |x| < 2^-27
    const double big = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return ix >= 0x44100000 ? (hx > 0 ? big : -big) : res;           // |x| >= 2^66
}
TL_HD double tl_atan2(double y, double x)
{   // fdlibm e_atan2.c for finite arguments, without branches (the quotient and the arctangent are always formed; the
    // special cases select over them), so that several calls can be in flight in one lane
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16, pi_o_2 = 1.5707963267948965580E+00;
    const uint64_t ux = tl_d2u(x), uy = tl_d2u(y);
    const int32_t hx = (int32_t)(ux >> 32), hy = (int32_t)(uy >> 32), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    const int k = (iy - ix) >> 20;
    const double q = y / x;
    double z = tl_atan(q < 0 ? -q : q);
    z = (hx < 0 && k < -60) ? 0.0 : z;
    z = k > 60 ? pi_o_2 + 0.5 * pi_lo : z;
    double res = m == 0 ? z : m == 1 ? -z : m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
    res = (ux << 1) == 0 ? (hy < 0 ? -pi_o_2 : pi_o_2) : res;       // x = +-0
    res = (uy << 1) == 0 ? (m <= 1 ? y : m == 2 ? pi : -pi) : res;  // y = +-0 (first in the original)
    return res;
}
