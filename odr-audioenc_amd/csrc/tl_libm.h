// tl_libm.h -- the reference's transcendental arithmetic, operation for operation.
//
// The reference calls the HOST's libm per frame: log10 / pow (psycho_1.c:245,254,373, psycho_3.c:158), sqrt,
// cos+sin (gcc fuses the pairs of psycho_2.c:127-132 into one sincos call), log, exp (psycho_2.c:114-137,185,190,
// 235,245, psycho_4.c:184-222,267,277,307,317) and atan2 (fft.c:1269-1274).  The oracle is the reference built here,
// so "the reference's result" is what Ubuntu glibc 2.35 (libm.so.6, x86-64, an FMA-capable CPU: this container's
// Xeon and the GPU box's EPYC alike) returns.  One ulp of difference decides tone tests and allocation ties on
// degenerate signals (round-2 soak), so a GPU path that wants the reference's bytes has to produce glibc's bits.
// This header restates the routines glibc 2.35 runs for those calls:
//
//   log    sysdeps/ieee754/dbl-64/e_log.c  (Szabolcs Nagy's table-driven log, N = 128), the build the ifunc resolver
//          picks on FMA machines (__log_fma, libm.so.6 0x76660)
//   log10  sysdeps/ieee754/dbl-64/e_log10.c (one baseline build, 0x29510): y*log10_2lo + ivln10*log(m) + y*log10_2hi
//   exp    e_exp.c (__exp_fma 0x76470)           pow   e_pow.c (__pow_fma 0x768b0): log_inline + exp_inline
//   sincos s_sincos.c + s_sin.c (IBM Accurate Mathematical Library as trimmed in 2.28+; ONE baseline build 0x2fa80,
//          no FMA anywhere in it)                atan2 e_atan2.c + uatan.tbl (__atan2_fma 0x78060)
//
// The *_fma builds are compiled with -mfma -mavx2 and GCC's default -ffp-contract=fast, so WHICH products are fused
// into which sums is the compiler's choice, not the source's: every fma() below is one vfmadd/vfmsub/vfnmadd of the
// disassembly (`objdump -d libm.so.6`, addresses above), every separate * and + one vmulsd / vaddsd; the
// expression trees were read off the instruction stream, not guessed.  This translation unit is compiled
// -ffp-contract=off, so nothing else gets fused.  Tables: tl_libm_tables.inc (tools/extract_libm_tables.py reads
// them out of the same libm.so.6, bit patterns, nothing retyped).
//
// Proof: tools/libm_agree.c runs every function here against the host's libm on >= 1e8 arguments each
// (profiles/libm_agree_r03.txt: 0 differing results), tests/test_libm_agree.py on a smaller sample in the CPU suite.
// Domain notes are at each function; outside them the functions still return what glibc returns unless stated.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TLM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TLM_HD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define TLM_TABLE_QUAL static __device__ const
#else
#define TLM_TABLE_QUAL static const
#endif
#include "tl_libm_tables.inc"

TLM_HD uint64_t tlm_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
TLM_HD double tlm_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#define TL_HD TLM_HD
TLM_HD uint64_t tl_d2u(double d) { return tlm_d2u(d); }       // the names the kernels use
TLM_HD double tl_u2d(uint64_t u) { return tlm_u2d(u); }
#define TLM_FMA(a, b, c) __builtin_fma((a), (b), (c))
// A select that stays a select (v_cndmask) on the device.  As a FUNCTION: the arms are its parameters -- evaluated by the caller, plain
// locals here -- so the front end emits a select instruction, not a branch with a phi.  (From `c ? f(x) : g(x)` it emits control flow;
// the optimiser then sinks each arm's computation, loads included, into its side of the branch and nothing turns that back: a
// divergent region in the middle of a routine, behind every scheduling fence the caller has set.)
TLM_HD double tlm_sel(bool c, double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_unpredictable(c) ? a : b;
#else
    return c ? a : b;
#endif
}
TLM_HD uint32_t tlm_sel_u32(bool c, uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_unpredictable(c) ? a : b;
#else
    return c ? a : b;
#endif
}
#define TLM_SEL(c, a, b) tlm_sel((c), (a), (b))
// a scheduling fence for callers that put a routine's phases apart (tlm_sincos_reduce / _poly / _finish, tlm_atan2_head / _mid / _tail):
// nothing is moved across it, so a table row requested before it is not waited for until the work between the fences has been issued
#if defined(__HIP_DEVICE_COMPILE__)
#define TLM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TLM_SCHED_FENCE() ((void)0)
#endif
#define TLM_D(bits) tlm_u2d(bits##ull)
// TLM_FMA_K(a, b, bits): fma(a, b, K) with a CONSTANT addend.  gfx950's three-operand fp64 instructions take no 64-bit literal, and the
// compiler builds such a Horner step as two v_mov_b32 into the destination of a v_fmac_f64 -- three vector issue slots per coefficient
// (it does so from a scalar register pair too: it prefers the two-address form).  Spelled out, the step is ONE v_fma_f64 whose addend
// is a scalar pair (two s_mov_b32, which issue beside the vector stream).  v_fma_f64 is the IEEE fused multiply-add: the same bits.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline __attribute__((always_inline)) double tlm_fma_k_(double a, double b, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#define TLM_FMA_K(a, b, bits) tlm_fma_k_((a), (b), TLM_D(bits))
#else
#define TLM_FMA_K(a, b, bits) TLM_FMA((a), (b), TLM_D(bits))
#endif
// TLM_MIN_NN(a, b): the smaller of two numbers, NEITHER a NaN (the caller's precondition): `b < a ? b : a` in one v_min_f64.  Through
// fmin() the compiler first canonicalises every operand that comes from memory (a v_max_f64 x, x each: the instruction quiets signalling
// NaNs and the language's fmin must not) -- twice the instructions for a case that cannot occur here.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline __attribute__((always_inline)) double tlm_min_nn_(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline __attribute__((always_inline)) double tlm_max_nn_(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define TLM_MIN_NN(a, b) tlm_min_nn_((a), (b))
#define TLM_MAX_NN(a, b) tlm_max_nn_((a), (b))
#else
#define TLM_MIN_NN(a, b) ((b) < (a) ? (b) : (a))
#define TLM_MAX_NN(a, b) ((a) < (b) ? (b) : (a))
#endif

// ------------------------------------------------------------------------------------------------------------
// log (e_log.c).  TAB = {invc, logc}[128] as bit patterns (global table or an LDS copy of it).
// The two evaluation paths as straight-line pieces so that device code can run both and select.

// |x - 1| small: x in [1 - 2^-4, 1 + 0x1.09p-4)  (e_log.c "close to 1.0" branch; __log_fma 0x76760-0x76832)
TLM_HD double tlm_log_near1(double x)
{
    const double B0 = TLM_D(0xbfe0000000000000), B1 = tlm_u2d(tlm_log_poly1[1]), B2 = tlm_u2d(tlm_log_poly1[2]),
                 B3 = tlm_u2d(tlm_log_poly1[3]), B4 = tlm_u2d(tlm_log_poly1[4]), B5 = tlm_u2d(tlm_log_poly1[5]),
                 B6 = tlm_u2d(tlm_log_poly1[6]), B7 = tlm_u2d(tlm_log_poly1[7]), B8 = tlm_u2d(tlm_log_poly1[8]),
                 B9 = tlm_u2d(tlm_log_poly1[9]), B10 = tlm_u2d(tlm_log_poly1[10]);
    const double r = x - 1.0;
    const double r2 = r * r;
    const double r3 = r * r2;
    const double q1 = TLM_FMA(r2, B3, TLM_FMA(B2, r, B1));
    const double q2 = TLM_FMA(r2, B6, TLM_FMA(B5, r, B4));
    const double q3 = TLM_FMA(r3, B10, TLM_FMA(r2, B9, TLM_FMA(B8, r, B7)));
    const double y0 = TLM_FMA(TLM_FMA(q3, r3, q2), r3, q1);
    const double t = TLM_FMA(r, 0x1p27, r);                 // r + w, w = r * 2^27
    const double rhi = TLM_FMA(-0x1p27, r, t);              // (r + w) - w
    const double rr = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = TLM_FMA(rr, B0, r);                   // r + rhi*rhi*B0
    double lo = TLM_FMA(rr, B0, r - hi);
    lo = TLM_FMA(B0 * rlo, rhi + r, lo);
    return hi + TLM_FMA(y0, r3, lo);
}

// the table path for positive NORMAL ix = bits of x (__log_fma 0x766a1-0x7675d)
template <typename TAB>
TLM_HD double tlm_log_main(uint64_t ix, TAB tab)
{
    const double ln2hi = tlm_u2d(tlm_log_ln2[0]), ln2lo = tlm_u2d(tlm_log_ln2[1]);
    const double A0 = tlm_u2d(tlm_log_poly[0]), A1 = tlm_u2d(tlm_log_poly[1]), A2 = tlm_u2d(tlm_log_poly[2]),
                 A3 = tlm_u2d(tlm_log_poly[3]), A4 = tlm_u2d(tlm_log_poly[4]);
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)(tmp >> 45) & 127;
    const int k = (int32_t)(uint32_t)(tmp >> 32) >> 20;            // (int64_t)tmp >> 52, from the high word (one shift, one 32-bit convert)
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double invc = tlm_u2d(tab[2 * i]), logc = tlm_u2d(tab[2 * i + 1]);
    const double z = tlm_u2d(iz);
    const double kd = (double)k;
    const double r = TLM_FMA(z, invc, -1.0);
    const double w = TLM_FMA(kd, ln2hi, logc);
    const double hi = w + r;
    const double lo = TLM_FMA(kd, ln2lo, (w - hi) + r);
    const double r2 = r * r;
    const double p = TLM_FMA(TLM_FMA(r, A4, A3), r2, TLM_FMA(r, A2, A1));
    return TLM_FMA(r * r2, p, TLM_FMA(r2, A0, lo)) + hi;
}

TLM_HD bool tlm_log_is_near1(uint64_t ix)
{
    return ix - 0x3fee000000000000ull < 0x3ff1090000000000ull - 0x3fee000000000000ull;
}

// log(x), any x (the special cases return what __log_fma returns; errno is not modelled)
template <typename TAB>
TLM_HD double tlm_log_t(double x, TAB tab)
{
    uint64_t ix = tlm_d2u(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (tlm_log_is_near1(ix)) return ix == 0x3ff0000000000000ull ? 0.0 : tlm_log_near1(x);
    if (top - 0x0010 >= 0x7ff0 - 0x0010) {
        if (ix * 2 == 0) return -1.0 / 0.0 * 1.0;
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000) || (top & 0x7ff0) == 0x7ff0) return (x - x) / (x - x);
        ix = tlm_d2u(x * 0x1p52) - (52ull << 52);
    }
    return tlm_log_main(ix, tab);
}
TLM_HD double tlm_log(double x) { return tlm_log_t(x, tlm_log_tab); }

// log for positive normal x without branches (both paths, one select): what a wave runs anyway when its lanes disagree
template <typename TAB>
TLM_HD double tlm_log_pn(double x, TAB tab)
{
    const uint64_t ix = tlm_d2u(x);
    const double a = tlm_log_near1(x), b = tlm_log_main(ix, tab);
    // (e_log.c returns 0 for x == 1.0 before anything else -- for the directed rounding modes; to nearest tlm_log_near1(1.0) is +0.0
    // itself: r = +0 makes every product a zero and every sum +0.  tests/test_libm_agree.py pins it.)
    return tlm_log_is_near1(ix) ? a : b;
}

// ------------------------------------------------------------------------------------------------------------
// log10 (e_log10.c, __ieee754_log10 0x29510: baseline build, separate multiplies and adds)
template <typename TAB>
TLM_HD double tlm_log10_t(double x, TAB tab)
{
    const double ivln10 = TLM_D(0x3fdbcb7b1526e50e), log10_2hi = TLM_D(0x3fd34413509f6000),
                 log10_2lo = TLM_D(0x3d59fef311f12b36);
    int64_t hx = (int64_t)tlm_d2u(x);
    int32_t k = 0;
    if (hx < 0x0010000000000000ll) {
        if ((hx & 0x7fffffffffffffffll) == 0) return -0x1p54 / (x < 0 ? -x : x);
        if (hx < 0) return (x - x) / (x - x);
        k -= 54;
        x *= 0x1p54;
        hx = (int64_t)tlm_d2u(x);
    }
    if (hx >= 0x7ff0000000000000ll) return x + x;
    k += (int32_t)(hx >> 52) - 1023;
    const int32_t i = (int32_t)((uint32_t)k >> 31);
    hx = (hx & 0x000fffffffffffffll) | ((int64_t)(0x3ff - i) << 52);
    const double y = (double)(k + i);
    const double z = y * log10_2lo + ivln10 * tlm_log_t(tlm_u2d((uint64_t)hx), tab);
    return z + y * log10_2hi;
}
TLM_HD double tlm_log10(double x) { return tlm_log10_t(x, tlm_log_tab); }

// log10 for positive NORMAL x, straight-line.  The mantissa handed to log is in [0.5, 2): always normal.
template <typename TAB>
TLM_HD double tlm_log10_pn(double x, TAB tab)
{
    const double ivln10 = TLM_D(0x3fdbcb7b1526e50e), log10_2hi = TLM_D(0x3fd34413509f6000),
                 log10_2lo = TLM_D(0x3d59fef311f12b36);
    const uint64_t hx = tlm_d2u(x);
    const int32_t k = (int32_t)(hx >> 52) - 1023;
    const int32_t i = (int32_t)((uint32_t)k >> 31);
    const uint64_t m = (hx & 0x000fffffffffffffull) | ((uint64_t)(0x3ff - i) << 52);
    const double y = (double)(k + i);
    const double z = y * log10_2lo + ivln10 * tlm_log_pn(tlm_u2d(m), tab);
    return z + y * log10_2hi;
}

// The same function in two halves for callers that compute MANY logarithms per wave.  e_log.c's "close to 1.0" branch is taken by the
// mantissas in [1 - 2^-4, 1 + 0x1.09p-4): about 9 % of all arguments -- so a wave of 64 lanes always has some, and tlm_log10_pn makes every
// lane pay for both branches.  tlm_log10_main_pn runs the table branch alone and reports which lanes it was NOT the right one for;
// the caller collects those arguments (a few dozen of five hundred) and puts them through tlm_log10_near1_pn together.  Per argument the
// operations are those of tlm_log10_pn on the branch e_log.c takes for it.
template <typename TAB>
TLM_HD double tlm_log10_main_pn(double x, TAB tab, bool *near1)
{
    const double ivln10 = TLM_D(0x3fdbcb7b1526e50e), log10_2hi = TLM_D(0x3fd34413509f6000),
                 log10_2lo = TLM_D(0x3d59fef311f12b36);
    const uint64_t hx = tlm_d2u(x);
    const int32_t k = (int32_t)(hx >> 52) - 1023;
    const int32_t i = (int32_t)((uint32_t)k >> 31);
    const uint64_t m = (hx & 0x000fffffffffffffull) | ((uint64_t)(0x3ff - i) << 52);
    const double y = (double)(k + i);
    *near1 = tlm_log_is_near1(m);
    const double z = y * log10_2lo + ivln10 * tlm_log_main(m, tab);
    return z + y * log10_2hi;
}
TLM_HD double tlm_log10_near1_pn(double x)
{
    const double ivln10 = TLM_D(0x3fdbcb7b1526e50e), log10_2hi = TLM_D(0x3fd34413509f6000),
                 log10_2lo = TLM_D(0x3d59fef311f12b36);
    const uint64_t hx = tlm_d2u(x);
    const int32_t k = (int32_t)(hx >> 52) - 1023;
    const int32_t i = (int32_t)((uint32_t)k >> 31);
    const uint64_t m = (hx & 0x000fffffffffffffull) | ((uint64_t)(0x3ff - i) << 52);
    const double y = (double)(k + i);
    const double z = y * log10_2lo + ivln10 * tlm_log_near1(tlm_u2d(m));
    return z + y * log10_2hi;
}

// ------------------------------------------------------------------------------------------------------------
// exp (e_exp.c).  exp_inline of e_pow.c is the same code with an extra low word and a sign bias, so one body
// serves both: tlm_exp_core(x, xtail, with_tail).  __exp_fma 0x76470, exp part of __pow_fma 0x76a05-0x76ad3.
TLM_HD double tlm_exp_special(double tmp, uint64_t sbits, uint64_t ki)
{   // e_exp.c specialcase(): |x| >= 512, the scale 2^(k/N) may be outside the normal range
    if ((ki & 0x80000000u) == 0) {
        sbits -= 1009ull << 52;
        const double scale = tlm_u2d(sbits);
        return 0x1p1009 * TLM_FMA(scale, tmp, scale);
    }
    sbits += 1022ull << 52;
    const double scale = tlm_u2d(sbits);
    const double st = tmp * scale;
    double y = scale + st;
    if ((y < 0 ? -y : y) < 1.0) {
        const double one = y < 0 ? -1.0 : 1.0;               // exp: always +1; pow's exp_inline: sign of y
        double lo = (scale - y) + st;
        const double hi = one + y;
        lo = ((one - hi) + y) + lo;
        y = (hi + lo) - one;
        if (y == 0.0) y = tlm_u2d(sbits & 0x8000000000000000ull);
    }
    return 0x1p-1022 * y;
}

template <bool WITH_TAIL>
TLM_HD double tlm_exp_core(double x, double xtail)
{
    const double invln2N = tlm_u2d(tlm_exp_head[0]), shift = tlm_u2d(tlm_exp_head[1]),
                 negln2hiN = tlm_u2d(tlm_exp_head[2]), negln2loN = tlm_u2d(tlm_exp_head[3]),
                 C2 = tlm_u2d(tlm_exp_head[4]), C3 = tlm_u2d(tlm_exp_head[5]), C4 = tlm_u2d(tlm_exp_head[6]),
                 C5 = tlm_u2d(tlm_exp_head[7]);
    uint32_t abstop = (uint32_t)(tlm_d2u(x) >> 52) & 0x7ff;
    if (abstop - 0x3c9 >= 0x3f) {
        if ((int32_t)(abstop - 0x3c9) < 0) return 1.0 + x;                  // |x| < 2^-54
        if (abstop >= 0x409) {                                              // |x| >= 1024
            if (!WITH_TAIL) {
                if (tlm_d2u(x) == 0xfff0000000000000ull) return 0.0;
                if (abstop >= 0x7ff) return 1.0 + x;
            }
            return (tlm_d2u(x) >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;
        }
        abstop = 0;                                                         // 512 <= |x| < 1024
    }
    const double kd0 = TLM_FMA(x, invln2N, shift);
    const uint64_t ki = tlm_d2u(kd0);
    const double kd = kd0 - shift;
    double r = TLM_FMA(kd, negln2loN, TLM_FMA(kd, negln2hiN, x));
    if (WITH_TAIL) r = xtail + r;
    const int idx = 2 * (int)(ki & 127);
    const double tail = tlm_u2d(tlm_exp_tab[idx]);
    const uint64_t sbits = tlm_exp_tab[idx + 1] + (ki << 45);
    const double r2 = r * r;
    const double tmp = TLM_FMA(TLM_FMA(r, C5, C4), r2 * r2, TLM_FMA(TLM_FMA(C3, r, C2), r2, r + tail));
    if (abstop == 0) return tlm_exp_special(tmp, sbits, ki);
    const double scale = tlm_u2d(sbits);
    return TLM_FMA(scale, tmp, scale);
}
TLM_HD double tlm_exp(double x) { return tlm_exp_core<false>(x, 0.0); }

// ------------------------------------------------------------------------------------------------------------
// pow (e_pow.c): log_inline (__pow_fma 0x768fb-0x769f8) for positive normal x.
TLM_HD double tlm_pow_log(uint64_t ix, double *tail)
{
    const double ln2hi = tlm_u2d(tlm_powlog_head[0]), ln2lo = tlm_u2d(tlm_powlog_head[1]);
    const double A0 = tlm_u2d(tlm_powlog_head[2]), A1 = tlm_u2d(tlm_powlog_head[3]), A2 = tlm_u2d(tlm_powlog_head[4]),
                 A3 = tlm_u2d(tlm_powlog_head[5]), A4 = tlm_u2d(tlm_powlog_head[6]), A5 = tlm_u2d(tlm_powlog_head[7]),
                 A6 = tlm_u2d(tlm_powlog_head[8]);
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)(tmp >> 45) & 127;
    const int k = (int32_t)(uint32_t)(tmp >> 32) >> 20;            // (int64_t)tmp >> 52, from the high word (one shift, one 32-bit convert)
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double z = tlm_u2d(iz), kd = (double)k;
    const double invc = tlm_u2d(tlm_powlog_tab[4 * i]), logc = tlm_u2d(tlm_powlog_tab[4 * i + 2]),
                 logctail = tlm_u2d(tlm_powlog_tab[4 * i + 3]);
    const double r = TLM_FMA(z, invc, -1.0);
    const double t1 = TLM_FMA(kd, ln2hi, logc);
    const double t2 = t1 + r;
    const double lo1 = TLM_FMA(kd, ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = A0 * r;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = TLM_FMA(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double p = TLM_FMA(ar2, TLM_FMA(TLM_FMA(r, A6, A5), ar2, TLM_FMA(r, A4, A3)), TLM_FMA(r, A2, A1));
    const double lo = TLM_FMA(ar3, p, ((lo1 + lo2) + lo3) + lo4);
    const double y = hi + lo;
    *tail = (hi - y) + lo;
    return y;
}

// pow(x, y) for finite x > 0 and finite y (the reference only ever raises 10.0 and 8/3)
TLM_HD double tlm_pow_pos(double x, double y)
{
    uint64_t ix = tlm_d2u(x);
    const uint64_t iy = tlm_d2u(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52) & 0x7ff;
    if (topy - 0x3be >= 0x43e - 0x3be) {
        if (2 * iy == 0) return 1.0;
        if (ix == 0x3ff0000000000000ull) return 1.0;
        if (topy < 0x3be) return ix > 0x3ff0000000000000ull ? 1.0 + y : 1.0 - y;       // |y| < 2^-65
        return (ix > 0x3ff0000000000000ull) == (topy < 0x800 && !(iy >> 63)) ? 0x1p769 * 0x1p769 : 0x1p-767 * 0x1p-767;
    }
    if (topx == 0) ix = (tlm_d2u(x * 0x1p52) & 0x7fffffffffffffffull) - (52ull << 52);
    double lo;
    const double hi = tlm_pow_log(ix, &lo);
    const double ehi = y * hi;
    const double elo = TLM_FMA(y, lo, TLM_FMA(hi, y, -ehi));
    return tlm_exp_core<true>(ehi, elo);
}

// pow(10.0, y): log_inline(10.0) does not depend on y -- its two words are constants
// (tests/test_libm_agree.py recomputes them with tlm_pow_log and compares).
#define TLM_LOG10_HI 0x40026bb1bbb55516ull
#define TLM_LOG10_LO 0xbcaf48ad48200000ull
TLM_HD double tlm_pow10(double y)
{
    const uint32_t topy = (uint32_t)(tlm_d2u(y) >> 52) & 0x7ff;
    if (topy < 0x3be) return 1.0;                             // |y| < 2^-65 (and +-0): 1.0 + y rounds to 1.0
    const double hi = tlm_u2d(TLM_LOG10_HI), lo = tlm_u2d(TLM_LOG10_LO);
    const double ehi = y * hi;
    const double elo = TLM_FMA(y, lo, TLM_FMA(hi, y, -ehi));
    return tlm_exp_core<true>(ehi, elo);
}

// ------------------------------------------------------------------------------------------------------------
// sincos (s_sincos.c / s_sin.c; sincos 0x2fa80: the one baseline build -- no fused operation anywhere).
// |x| < 105414350 (reduce_sincos range); the encoder's phases stay below 11.
#define TLM_SN3 TLM_D(0xbfc5555555555515)
#define TLM_SN5 TLM_D(0x3f811110e829872f)
#define TLM_CS2 0.5
#define TLM_CS4 TLM_D(0xbfa5555555555535)
#define TLM_CS6 TLM_D(0x3f56c16bedd9e239)
#define TLM_BIG TLM_D(0x42c8000000000000)

template <typename TAB>
TLM_HD double tlm_do_cos(double x, double dx, TAB tab)
{   // s_sin.c do_cos
    if (x < 0) dx = -dx;
    const double ax = x < 0 ? -x : x;
    const double u = TLM_BIG + ax;
    x = (ax - (u - TLM_BIG)) + dx;
    const double xx = x * x;
    const double s = x + (x * xx) * (TLM_SN3 + xx * TLM_SN5);
    const double c = xx * (TLM_CS2 + xx * (TLM_CS4 + xx * TLM_CS6));
    const int k = (int)(uint32_t)tlm_d2u(u) * 4;
    const double sn = tlm_u2d(tab[k]), ssn = tlm_u2d(tab[k + 1]), cs = tlm_u2d(tab[k + 2]), ccs = tlm_u2d(tab[k + 3]);
    const double cor = ((ccs - s * ssn) - cs * c) - sn * s;
    return cs + cor;
}

TLM_HD double tlm_taylor_sin(double xx, double x, double dx)
{   // s_sin.c TAYLOR_SIN / POLYNOMIAL
    const double s1 = TLM_D(0xbfc5555555555555), s2 = TLM_D(0x3f81111111110ece), s3 = TLM_D(0xbf2a01a019db08b8),
                 s4 = TLM_D(0x3ec71de27b9a7ed9), s5 = TLM_D(0xbe5addffc2fcdf59);
    const double p = ((((s5 * xx + s4) * xx + s3) * xx + s2) * xx) + s1;
    const double t = (p * x - 0.5 * dx) * xx + dx;
    return x + t;
}

template <typename TAB>
TLM_HD double tlm_do_sin(double x, double dx, TAB tab)
{   // s_sin.c do_sin
    const double xold = x;
    const double ax = x < 0 ? -x : x;
    if (ax < 0.126) return tlm_taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    const double u = TLM_BIG + ax;
    x = ax - (u - TLM_BIG);
    const double xx = x * x;
    const double s = x + (dx + (x * xx) * (TLM_SN3 + xx * TLM_SN5));
    const double c = x * dx + xx * (TLM_CS2 + xx * (TLM_CS4 + xx * TLM_CS6));
    const int k = (int)(uint32_t)tlm_d2u(u) * 4;
    const double sn = tlm_u2d(tab[k]), ssn = tlm_u2d(tab[k + 1]), cs = tlm_u2d(tab[k + 2]), ccs = tlm_u2d(tab[k + 3]);
    const double cor = ((ssn + s * ccs) - sn * c) + cs * s;
    const double r = sn + cor;
    return tlm_u2d((tlm_d2u(r) & 0x7fffffffffffffffull) | (tlm_d2u(xold) & 0x8000000000000000ull));
}

template <typename TAB>
TLM_HD void tlm_sincos_t(double x, double *sinx, double *cosx, TAB tab)
{
    const double hp0 = TLM_D(0x3ff921fb54442d18), hp1 = TLM_D(0x3c91a62633145c07);
    const uint64_t sx = tlm_d2u(x) & 0x8000000000000000ull;
    const int32_t k = (int32_t)(tlm_d2u(x) >> 32) & 0x7fffffff;
    if (k < 0x400368fd) {
        if (k < 0x3e400000) { *sinx = x; *cosx = 1.0; return; }                    // |x| < 2^-27
        if (k < 0x3feb6000) {                                                      // |x| < 0.855469
            *sinx = tlm_do_sin(x, 0.0, tab);
            *cosx = tlm_do_cos(x, 0.0, tab);
            return;
        }
        const double ax = x < 0 ? -x : x;                                          // |x| < 2.426265
        const double y = hp0 - ax;
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        const double s = tlm_do_cos(a, da, tab);
        *sinx = tlm_u2d((tlm_d2u(s) & 0x7fffffffffffffffull) | sx);
        *cosx = tlm_do_sin(a, da, tab);
        return;
    }
    // reduce_sincos (|x| < 105414350)
    const double hpinv = TLM_D(0x3fe45f306dc9c883), toint = TLM_D(0x4338000000000000), mp1 = TLM_D(0x3ff921fb58000000),
                 mp2 = TLM_D(0xbe4dde973c000000), pp3 = TLM_D(0xbc8cb3b398000000), pp4 = TLM_D(0xbacd747f23e32ed7);
    const double t = x * hpinv + toint;
    const double xn = t - toint;
    const int n = (int)(uint32_t)tlm_d2u(t) & 3;
    const double y = (x - xn * mp1) - xn * mp2;
    double t1 = xn * pp3;
    const double t2 = y - t1;
    double db = (y - t2) - t1;
    t1 = xn * pp4;
    const double b = t2 - t1;
    db += (t2 - b) - t1;
    // s_sincos.c: for n = 1, 2 the reduced argument is negated; odd n swaps the outputs; n & 2 negates the cosine
    double a = b, da = db;
    if (n == 1 || n == 2) { a = -a; da = -da; }
    const double s = tlm_do_sin(a, da, tab), c0 = tlm_do_cos(a, da, tab);
    const double c = (n & 2) ? -c0 : c0;
    *sinx = (n & 1) ? c : s;
    *cosx = (n & 1) ? s : c;
}
TLM_HD void tlm_sincos(double x, double *s, double *c) { tlm_sincos_t(x, s, c, tlm_sincostab); }

// ------------------------------------------------------------------------------------------------------------
// atan2 (e_atan2.c, __atan2_fma 0x78060) for finite arguments.
template <typename TAB>
TLM_HD double tlm_atan2_t(double y, double x, TAB cij)
{
    const double hpi = TLM_D(0x3ff921fb54442d18), hpi1 = TLM_D(0x3c91a62633145c07), opi = TLM_D(0x400921fb54442d18),
                 opi1 = TLM_D(0x3ca1a62633145c07);
    const double d3 = TLM_D(0xbfd5555555555555), d5 = TLM_D(0x3fc99999999997fd), d7 = TLM_D(0xbfc24924923f7603),
                 d9 = TLM_D(0x3fbc71c6e5129a3b), d11 = TLM_D(0xbfb7458022b13c25), d13 = TLM_D(0x3fb375f08b31cbce);
    const uint64_t bx = tlm_d2u(x), by = tlm_d2u(y), sy = by & 0x8000000000000000ull;
    const int32_t ux = (int32_t)(bx >> 32), uy = (int32_t)(by >> 32);
    if ((by << 1) == 0) return (bx >> 63) ? tlm_u2d(tlm_d2u(opi) | sy) : y;    // y = +-0 (x = +-0 included)
    if ((bx << 1) == 0) return tlm_u2d(tlm_d2u(hpi) | sy);     // x = +-0
    double ax = tlm_u2d(bx & 0x7fffffffffffffffull), ay = tlm_u2d(by & 0x7fffffffffffffffull);
    const int32_t de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
    if (de >= 59768832) return tlm_u2d(tlm_d2u(hpi) | sy);
    if (de <= -59768832) {
        if (x > 0) return tlm_u2d(tlm_d2u(ay / ax) | sy);       // (the underflow flag is not modelled)
        return tlm_u2d(tlm_d2u(opi) | sy);
    }
    if (ax < 0x1p-500 || ay < 0x1p-500) { ax *= 0x1p500; ay *= 0x1p500; }
    if (ax > 0x1p500 || ay > 0x1p500) { ax *= 0x1p-500; ay *= 0x1p-500; }
    double u, du, z;
    const bool ylx = ay < ax;
    {
        const double num = ylx ? ay : ax, den = ylx ? ax : ay;
        u = num / den;
        const double v = den * u;
        const double vv = TLM_FMA(den, u, -v);                  // EMULV
        du = ((num - v) - vv) / den;
    }
    const bool small = u < 0.0625;
    double zz = 0, v = 0;
    if (small) {
        v = u * u;
        zz = TLM_FMA(TLM_FMA(TLM_FMA(TLM_FMA(TLM_FMA(d13, v, d11), v, d9), v, d7), v, d5), v, d3);
    } else {
        const int i = (int)(uint32_t)tlm_d2u(TLM_FMA(u, 256.0, 0x1p52)) - 16;     // (TWO52 + TWO8*u) - TWO52 as an integer: the low word of the sum
        // row i of cij: {x_i, atan(x_i), c2..c6}
        const double c0 = tlm_u2d(cij[7 * i]);
        if (x > 0 && ylx) {                                     // (i): EADD(u - c0, du) keeps the low word
            const double t3 = u - c0;
            const double w = t3 + du;
            const double at3 = t3 < 0 ? -t3 : t3, adu = du < 0 ? -du : du;
            const double dv = at3 > adu ? (t3 - w) + du : (du - w) + t3;
            const double t1 = tlm_u2d(cij[7 * i + 1]), t2 = tlm_u2d(cij[7 * i + 2]);
            const double p = TLM_FMA(TLM_FMA(TLM_FMA(tlm_u2d(cij[7 * i + 6]), w, tlm_u2d(cij[7 * i + 5])), w,
                                             tlm_u2d(cij[7 * i + 4])), w, tlm_u2d(cij[7 * i + 3]));
            zz = TLM_FMA(w, t2, TLM_FMA(dv, t2, (w * w) * p));
            z = t1 + zz;
            return tlm_u2d((tlm_d2u(z) & 0x7fffffffffffffffull) | sy);
        }
        v = (u - c0) + du;
        const double p = TLM_FMA(TLM_FMA(TLM_FMA(TLM_FMA(tlm_u2d(cij[7 * i + 6]), v, tlm_u2d(cij[7 * i + 5])), v,
                                                 tlm_u2d(cij[7 * i + 4])), v, tlm_u2d(cij[7 * i + 3])), v,
                                 tlm_u2d(cij[7 * i + 2]));
        const double a1 = tlm_u2d(cij[7 * i + 1]);
        if (x > 0) { zz = TLM_FMA(-v, p, hpi1); z = (hpi - a1) + zz; }          // (ii)
        else if (!ylx && ax < ay) { zz = TLM_FMA(v, p, hpi1); z = (hpi + a1) + zz; }   // (iii)
        else { zz = TLM_FMA(-v, p, opi1); z = (opi - a1) + zz; }                // (iv)
        return tlm_u2d((tlm_d2u(z) & 0x7fffffffffffffffull) | sy);
    }
    const double uv = u * v;
    if (x > 0) {
        if (ylx) z = u + TLM_FMA(uv, zz, du);                                   // (i)
        else {                                                                  // (ii): ESUB(hpi, u)
            const double t2 = hpi - u;
            const double cor = (hpi - t2) - u;
            z = ((((cor + hpi1) - du) - uv * zz)) + t2;
        }
    } else if (!ylx && ax < ay) {                                               // (iii): EADD(hpi, u)
        const double t2 = hpi + u;
        const double cor = (hpi - t2) + u;
        z = (((cor + hpi1) + du) + uv * zz) + t2;
    } else {                                                                    // (iv): ESUB(opi, u)
        const double t2 = opi - u;
        const double cor = (opi - t2) - u;
        z = (((cor + opi1) - du) - uv * zz) + t2;
    }
    return tlm_u2d((tlm_d2u(z) & 0x7fffffffffffffffull) | sy);
}
TLM_HD double tlm_atan2(double y, double x) { return tlm_atan2_t(y, x, tlm_atan_cij); }

// ------------------------------------------------------------------------------------------------------------
// IEEE division and square root WITHOUT the exponent scaling (round 6).  The compiler expands an fp64 `/` into v_div_scale x 2, v_rcp,
// two Newton steps on the reciprocal, a product, a residual, v_div_fmas, v_div_fixup (11 instructions) and sqrt() into a scale test,
// v_ldexp, v_rsq, nine refinement steps, v_ldexp, a class test and two selects (18): the scale / fixup parts only matter for operands
// within ~2^70 of the exponent limits, for zero and for infinities.  The forms below are the SAME refinement steps on unscaled
// operands -- for every operand the scaled forms would leave unscaled they execute the same arithmetic, hence return the same bits,
// i.e. the correctly rounded result (the final fused step sees the exact residual) -- and a reciprocal refined once serves every
// quotient by the same divisor (e_atan2.c divides twice by max(|x|, |y|)).
// Domain: divisor and square-root argument normal and in [2^-500, 2^500], dividend zero or in [2^-900, 2^500]; sqrt(+0) = +0 is a select.
// The seed is v_rcp_f64 / v_rsq_f64 on the device; on the host (emulation, tools) the exact 1/d, 1/sqrt(x) rounded -- the refinement
// converges to the same result from any seed good to ~2^-20 (tools/libm_agree.cpp perturbs the seed to show it, 1e8 operands each).
TLM_HD double tlm_rcp_seed(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(d);
#else
    return 1.0 / d;
#endif
}
TLM_HD double tlm_rsq_seed(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#else
    return 1.0 / __builtin_sqrt(x);
#endif
}
TLM_HD double tlm_recip_from(double d, double r)
{   // two Newton steps: r (1 + e) with e = 1 - d r
    double e = TLM_FMA(-d, r, 1.0);
    r = TLM_FMA(r, e, r);
    e = TLM_FMA(-d, r, 1.0);
    return TLM_FMA(r, e, r);
}
TLM_HD double tlm_div_by_recip(double n, double d, double r)
{   // n / d with r = the refined reciprocal of d: quotient estimate, its exact residual, one fused correction
    const double q = n * r;
    const double e = TLM_FMA(-d, q, n);
    return TLM_FMA(e, r, q);
}
TLM_HD double tlm_sqrt_from(double x, double y)
{   // sqrt(x) from y ~ 1 / sqrt(x): g ~ sqrt(x), h ~ 1 / (2 sqrt(x)) refined together once, then g twice with its exact residual
    double g = x * y;
    double h = y * 0.5;
    const double r = TLM_FMA(-h, g, 0.5);
    g = TLM_FMA(g, r, g);
    h = TLM_FMA(h, r, h);
    double d = TLM_FMA(-g, g, x);
    g = TLM_FMA(d, h, g);
    d = TLM_FMA(-g, g, x);
    return TLM_FMA(d, h, g);
}
TLM_HD double tlm_div_ns(double n, double d) { return tlm_div_by_recip(n, d, tlm_recip_from(d, tlm_rcp_seed(d))); }
TLM_HD double tlm_sqrt_ns(double x) { const double g = tlm_sqrt_from(x, tlm_rsq_seed(x)); return TLM_SEL(x == 0.0, x, g); }      // (x = 0: the seed is infinite)
TLM_HD double tlm_sqrt_nz(double x) { return tlm_sqrt_from(x, tlm_rsq_seed(x)); }                                                // x > 0 by the caller's contract

// ------------------------------------------------------------------------------------------------------------
// Straight-line forms for the device: the lanes of a wave are spread over every branch of the routines above, so
// the wave executes all of them anyway; here the branches' COMMON work is done once and the results are selected.
// Same operations on the selected path, hence the same bits (tools/libm_agree.cpp checks these forms as well).

// sincos for |x| < 105414350.  All four argument ranges end in do_sin(a, da) and do_cos(a, da) of some reduced
// argument; the ranges differ in (a, da), in which output takes which value and in the signs.  Round 6: the ranges are folded into
// ONE quadrant number so that the outputs take two 64-bit selects instead of six, and a 64-bit select is two instructions:
//   * |x| < 0.855469 is reduce_sincos with xn FORCED to +0 and n = 0: y = (x - 0) - 0 = x, every correction term is (x - x) - 0 = +0,
//     so (a, da) = (x, +0) -- do_sin(x, 0.0) / do_cos(x, 0.0) as s_sincos.c calls them (x = -0.0 reduces to +0: a takes x's high word);
//   * 0.855469 <= |x| < 2.426265 ("pi/2 - |x|") hands out sin = |cos(a)| with x's sign and cos = sin(a): that is quadrant 1 for
//     x > 0 and quadrant 3 for x < 0 of the general rule (sin = +-cos(a), cos = sin(a)) because cos(a) >= cos(0.855469) > 0 there;
//   * |x| < 2^-27 (sin = x, cos = 1.0 in s_sincos.c) needs no case: TAYLOR_SIN gives x + (p x) x^2 with |(p x) x^2| < 2^-56 |x|, which
//     rounds to x (the sign of -0.0 comes back with the copysign below), and do_cos gives 1.0 - c with c <= x^2 / 2 < 2^-55, which rounds to 1.0 (row 0 of the table is sin 0, cos 1 exactly);
//   * do_cos flips dx for a < 0, do_sin's table branch for a <= 0: they differ at a = +-0 only, where do_sin takes TAYLOR_SIN anyway,
//     and for a = -0.0 the cosine does not depend on dx's sign (row 0: sn = ssn = ccs = 0, so cor = (0 - s*0 - c) - 0*s = -c either
//     way) -- both use the sign BIT of a.
// tools/libm_agree.cpp runs this form against the host's sincos (every range, every boundary, 1e8 arguments per round).
// The routine in three PHASES so that a caller can put other work between them (the psy-2 kernel's line loop, mp2_psy24.h): the table
// row is requested at the end of phase one and first needed in phase three, and everything the row does not enter -- both polynomial
// sets, about forty operations -- is phase two.  Left to itself the compiler requests the row eight instructions before it waits for it.
struct TlmSinCosA { double a, da, aa, xr, dx; uint64_t sa; uint32_t q; int row; };
struct TlmSinCosB { double cs_s, cs_c, ty, sn_s, sn_c; };
TLM_HD TlmSinCosA tlm_sincos_reduce(double x)
{   // phase one: the reduced argument (a, da), the quadrant, the table row
    const double hp0 = TLM_D(0x3ff921fb54442d18), hp1 = TLM_D(0x3c91a62633145c07);
    const double hpinv = TLM_D(0x3fe45f306dc9c883), toint = TLM_D(0x4338000000000000), mp1 = TLM_D(0x3ff921fb58000000),
                 mp2 = TLM_D(0xbe4dde973c000000), pp3 = TLM_D(0xbc8cb3b398000000), pp4 = TLM_D(0xbacd747f23e32ed7);
    TlmSinCosA r;
    const uint64_t bx = tlm_d2u(x);
    const uint32_t hx = (uint32_t)(bx >> 32);
    const int32_t k = (int32_t)(hx & 0x7fffffffu);
    const double ax = tlm_u2d(bx & 0x7fffffffffffffffull);
    const bool p1 = k < 0x3feb6000, p12 = k < 0x400368fd;
    // reduce_sincos (xn = +0 in range 1)
    const double t = x * hpinv + toint;
    const double xn = TLM_SEL(p1, 0.0, t - toint);
    const uint32_t n = (uint32_t)tlm_d2u(t) & 3u;
    const double y = (x - xn * mp1) - xn * mp2;
    const double t1 = xn * pp3;
    const double t2 = y - t1;
    const double t1b = xn * pp4;
    const double b = t2 - t1b;
    const double db = ((y - t2) - t1) + ((t2 - b) - t1b);
    // the quadrant: 0 in range 1; 1 / 3 in range 2 (x > 0 / x < 0); n in range 3, whose reduced argument is negated for n = 1, 2
    r.q = p1 ? 0u : p12 ? (1u | (hx >> 30 & 2u)) : n;
    const uint64_t flip = p12 ? 0ull : (uint64_t)((n + 1u) & 2u) << 62;
    // range 2: pi/2 - |x|
    const double yy = hp0 - ax;
    const double a2 = yy + hp1;
    const double da2 = (yy - a2) + hp1;
    const bool p2 = p12 && !p1;
    const uint64_t a23 = tlm_d2u(TLM_SEL(p2, a2, tlm_u2d(tlm_d2u(b) ^ flip)));
    // (range 1: b = x for every x but -0.0, whose reduction comes out as +0 -- the high word is x's own there: one 32-bit select)
    r.a = tlm_u2d((uint64_t)tlm_sel_u32(p1, hx, (uint32_t)(a23 >> 32)) << 32 | (uint32_t)a23);
    r.da = TLM_SEL(p2, da2, tlm_u2d(tlm_d2u(db) ^ flip));
    // do_sin / do_cos on (a, da): shared table row, first reduction and dx = da with a's sign
    r.sa = tlm_d2u(r.a) & 0x8000000000000000ull;
    r.aa = tlm_u2d(tlm_d2u(r.a) ^ r.sa);
    const double u = TLM_BIG + r.aa;
    r.xr = r.aa - (u - TLM_BIG);
    r.row = (int)(uint32_t)tlm_d2u(u) * 4;
    r.dx = tlm_u2d(tlm_d2u(r.da) ^ r.sa);
    return r;
}
TLM_HD TlmSinCosB tlm_sincos_poly(const TlmSinCosA &r)
{   // phase two: what the table row does not enter
    TlmSinCosB p;
    {
        const double xc = r.xr + r.dx;
        const double xx = xc * xc;
        p.cs_s = xc + (xc * xx) * (TLM_SN3 + xx * TLM_SN5);
        p.cs_c = xx * (TLM_CS2 + xx * (TLM_CS4 + xx * TLM_CS6));
    }
    {
        p.ty = tlm_taylor_sin(r.a * r.a, r.a, r.da);
        const double xx = r.xr * r.xr;
        p.sn_s = r.xr + (r.dx + (r.xr * xx) * (TLM_SN3 + xx * TLM_SN5));
        p.sn_c = r.xr * r.dx + xx * (TLM_CS2 + xx * (TLM_CS4 + xx * TLM_CS6));
    }
    return p;
}
// phase three: the row's four values against the polynomials -> do_sin's value (signed) and do_cos's (negated in quadrants 2, 3).  In an
// EVEN quadrant (r.q & 1 == 0) they are the sine and the cosine, in an odd one the cosine and the sine: tlm_sincos_finish hands them out;
// a caller that only forms symmetric expressions of several pairs can do with fewer exchanges (tl_psy2_pass).
TLM_HD void tlm_sincos_finish_raw(const TlmSinCosA &r, const TlmSinCosB &p, double sn, double ssn, double cs, double ccs, double *dosin, double *docos)
{
    const double cosv = cs + (((ccs - p.cs_s * ssn) - cs * p.cs_c) - sn * p.cs_s);
    const double rt = sn + (((ssn + p.sn_s * ccs) - sn * p.sn_c) + cs * p.sn_s);
    // do_sin returns the table value with a's sign; TAYLOR_SIN's value has a's sign by itself -- except for a = -0.0, where it is
    // +0 and s_sincos.c's own |x| < 2^-27 case returns x: one copysign after the select serves both
    const double sel = TLM_SEL(r.aa < 0.126, p.ty, rt);
    const double sinv = tlm_u2d((tlm_d2u(sel) & 0x7fffffffffffffffull) | r.sa);
    // quadrants 2 and 3 negate the cosine
    *dosin = sinv;
    *docos = tlm_u2d(tlm_d2u(cosv) ^ ((uint64_t)(r.q & 2u) << 62));
}
TLM_HD void tlm_sincos_finish(const TlmSinCosA &r, const TlmSinCosB &p, double sn, double ssn, double cs, double ccs, double *sinx, double *cosx)
{
    double sinv, c3;
    tlm_sincos_finish_raw(r, p, sn, ssn, cs, ccs, &sinv, &c3);
    const bool odd = (r.q & 1u) != 0;                                   // odd quadrants swap
    *sinx = TLM_SEL(odd, c3, sinv);
    *cosx = TLM_SEL(odd, sinv, c3);
}
template <typename TAB>
TLM_HD void tlm_sincos_sl(double x, double *sinx, double *cosx, TAB tab)
{
    const TlmSinCosA r = tlm_sincos_reduce(x);
    const double sn = tlm_u2d(tab[r.row]), ssn = tlm_u2d(tab[r.row + 1]), cs = tlm_u2d(tab[r.row + 2]), ccs = tlm_u2d(tab[r.row + 3]);
    const TlmSinCosB p = tlm_sincos_poly(r);
    tlm_sincos_finish(r, p, sn, ssn, cs, ccs, sinx, cosx);
}

// atan2 for finite arguments: one quotient / remainder pair, both evaluation forms (polynomial below 1/16, table row
// above), the four quadrant combinations selected at the end.
// SCALE = false leaves out e_atan2.c's rescaling of operands below 2^-500 / above 2^500.  That is the same function wherever
// max(|y|, |x|) lies in [2^-443, 2^500]: an operand below 2^-500 is then more than 57 binades from the other, the result comes
// from the extreme-ratio branches (which use the unscaled operands) and the quotient path's value is discarded.  The encoder's
// spectra qualify (|.| < 2^25, and the call is only made when a*a + b*b >= 0.001).
template <bool SCALE = true, typename TAB>
TLM_HD double tlm_atan2_sl(double y, double x, TAB cij)
{
    const double hpi = TLM_D(0x3ff921fb54442d18), hpi1 = TLM_D(0x3c91a62633145c07), opi = TLM_D(0x400921fb54442d18),
                 opi1 = TLM_D(0x3ca1a62633145c07);
    const uint64_t bx = tlm_d2u(x), by = tlm_d2u(y), sy = by & 0x8000000000000000ull;
    const int32_t ux = (int32_t)(bx >> 32), uy = (int32_t)(by >> 32);
    const bool xpos = !(bx >> 63);
    double ax = tlm_u2d(bx & 0x7fffffffffffffffull), ay = tlm_u2d(by & 0x7fffffffffffffffull);
    const double ax0 = ax, ay0 = ay;
    const int32_t de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
    if (SCALE) {
        const bool dn = ax < 0x1p-500 || ay < 0x1p-500;
        ax = dn ? ax * 0x1p500 : ax; ay = dn ? ay * 0x1p500 : ay;
        const bool up = ax > 0x1p500 || ay > 0x1p500;
        ax = up ? ax * 0x1p-500 : ax; ay = up ? ay * 0x1p-500 : ay;
    }
    const bool ylx = ay < ax;
    const double num = ylx ? ay : ax, den = ylx ? ax : ay;
    double u, du, rd = 0;
    if (SCALE) u = num / den;
    else {
        // both quotients by ONE refined reciprocal of den, without the divisions' exponent scaling (tlm_div_by_recip): den is in
        // [2^-443, 2^500] by this form's contract; where num is more than 2^57 below it (or zero) the extreme-ratio branches below
        // discard the quotients, whatever they are
        rd = tlm_recip_from(den, tlm_rcp_seed(den));
        u = tlm_div_by_recip(num, den, rd);
    }
    // form B's table row is requested as soon as the first quotient names it; the second quotient and all of form A run while it is
    // on its way (the phased form below, tlm_atan2_head / _mid / _tail, lets a caller pin that order with scheduling fences)
    int i = (int)(uint32_t)tlm_d2u(TLM_FMA(u, 256.0, 0x1p52)) - 16;       // the low word of TWO52 + TWO8*u (defined for the NaN of 0/0 too: the lane's result is discarded)
    i = i < 0 ? 0 : i > 240 ? 240 : i;                                    // (only form A's lanes can leave the table)
    const double c0 = tlm_u2d(cij[7 * i]), c1 = tlm_u2d(cij[7 * i + 1]), c2 = tlm_u2d(cij[7 * i + 2]), c3 = tlm_u2d(cij[7 * i + 3]),
                 c4 = tlm_u2d(cij[7 * i + 4]), c5 = tlm_u2d(cij[7 * i + 5]), c6 = tlm_u2d(cij[7 * i + 6]);
    {
        const double pv = den * u;
        const double res = (num - pv) - TLM_FMA(den, u, -pv);
        du = SCALE ? res / den : tlm_div_by_recip(res, den, rd);
    }
    // form A: u < 1/16
    const double v = u * u;
    // (d13 .. d3 of e_atan2.c; the constant addends from scalar registers: TLM_FMA_K)
    const double pa = TLM_FMA_K(TLM_FMA_K(TLM_FMA_K(TLM_FMA_K(TLM_FMA_K(TLM_D(0x3fb375f08b31cbce), v, 0xbfb7458022b13c25), v, 0x3fbc71c6e5129a3b), v,
                                                    0xbfc24924923f7603), v, 0x3fc99999999997fd), v, 0xbfd5555555555555);
    const double uv = u * v;
    const double zA1 = u + TLM_FMA(uv, pa, du);
    const double zz = uv * pa;
    // form B: table row
    const double t3 = u - c0;
    const double w = t3 + du;
    const double p3 = TLM_FMA(TLM_FMA(TLM_FMA(c6, w, c5), w, c4), w, c3);
    const double p2 = TLM_FMA(p3, w, c2);
    const bool q1 = xpos && ylx, q2 = xpos && !ylx, q3 = !xpos && !ylx && ax < ay;      // else (iv)
    double zA, zB;
    {   // (i)
        const double at3 = t3 < 0 ? -t3 : t3, adu = du < 0 ? -du : du;
        const double dv = at3 > adu ? (t3 - w) + du : (du - w) + t3;
        const double zB1 = c1 + TLM_FMA(w, c2, TLM_FMA(dv, c2, (w * w) * p3));
        // (ii), (iii), (iv): pi/2 -, pi/2 +, pi - ; the three share their shape: base (-/+) u, base1 (-/+) stuff
        const double base = q3 || q2 ? hpi : opi, base1 = q3 || q2 ? hpi1 : opi1;
        const uint64_t sg = q3 ? 0 : 0x8000000000000000ull;              // (iii) adds, (ii) and (iv) subtract
        const double su = tlm_u2d(tlm_d2u(u) ^ sg), sdu = tlm_u2d(tlm_d2u(du) ^ sg);
        const double t2 = base + su;
        const double cor = (base - t2) + su;
        const double zA2 = (((cor + base1) + sdu) + tlm_u2d(tlm_d2u(zz) ^ sg)) + t2;
        const double zB2 = (base + tlm_u2d(tlm_d2u(c1) ^ sg)) + TLM_FMA(tlm_u2d(tlm_d2u(w) ^ sg), p2, base1);
        zA = q1 ? zA1 : zA2;
        zB = q1 ? zB1 : zB2;
    }
    double z = u < 0.0625 ? zA : zB;
    // extreme ratios and zeros
    z = de <= -59768832 ? (xpos ? ay0 / ax0 : opi) : z;
    z = de >= 59768832 ? hpi : z;
    z = (bx << 1) == 0 ? hpi : z;
    z = (by << 1) == 0 ? (xpos ? 0.0 : opi) : z;
    return tlm_u2d((tlm_d2u(z) & 0x7fffffffffffffffull) | sy);
}

// The no-rescale form in three PHASES (as tlm_sincos_reduce / _poly / _finish): head = the first quotient and the table row it names,
// mid = the second quotient and all of form A (nothing of the row enters), tail = form B on the row, the selection, the extreme ratios.
// max(|y|, |x|) in [2^-443, 2^500] as for tlm_atan2_sl<false>; tools/libm_agree.cpp runs head + mid + tail against the host's atan2.
struct TlmAtanA { uint64_t bx, by; double ax, ay, num, den, rd, u; int32_t de; int i; bool xpos, ylx; };
struct TlmAtanB { double du, zA; };
TLM_HD TlmAtanA tlm_atan2_head(double y, double x)
{
    TlmAtanA r;
    r.bx = tlm_d2u(x); r.by = tlm_d2u(y);
    const int32_t ux = (int32_t)(r.bx >> 32), uy = (int32_t)(r.by >> 32);
    r.xpos = !(r.bx >> 63);
    r.ax = tlm_u2d(r.bx & 0x7fffffffffffffffull); r.ay = tlm_u2d(r.by & 0x7fffffffffffffffull);
    r.de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
    r.ylx = r.ay < r.ax;
    r.num = TLM_MIN_NN(r.ax, r.ay); r.den = TLM_MAX_NN(r.ay, r.ax);    // (= ylx ? ay : ax and ylx ? ax : ay: finite magnitudes, equal where they tie)
    r.rd = tlm_recip_from(r.den, tlm_rcp_seed(r.den));
    r.u = tlm_div_by_recip(r.num, r.den, r.rd);
    int i = (int)(uint32_t)tlm_d2u(TLM_FMA(r.u, 256.0, 0x1p52)) - 16;
    r.i = i < 0 ? 0 : i > 240 ? 240 : i;
    return r;
}
TLM_HD TlmAtanB tlm_atan2_mid(const TlmAtanA &r)
{
    const double hpi = TLM_D(0x3ff921fb54442d18), hpi1 = TLM_D(0x3c91a62633145c07), opi = TLM_D(0x400921fb54442d18),
                 opi1 = TLM_D(0x3ca1a62633145c07);
    TlmAtanB m;
    const double u = r.u;
    const double pv = r.den * u;
    m.du = tlm_div_by_recip((r.num - pv) - TLM_FMA(r.den, u, -pv), r.den, r.rd);
    const double v = u * u;
    const double pa = TLM_FMA_K(TLM_FMA_K(TLM_FMA_K(TLM_FMA_K(TLM_FMA_K(TLM_D(0x3fb375f08b31cbce), v, 0xbfb7458022b13c25), v, 0x3fbc71c6e5129a3b), v,
                                                    0xbfc24924923f7603), v, 0x3fc99999999997fd), v, 0xbfd5555555555555);
    const double uv = u * v;
    const double zA1 = u + TLM_FMA(uv, pa, m.du);
    const double zz = uv * pa;
    const bool q1 = r.xpos && r.ylx, q2 = r.xpos && !r.ylx, q3 = !r.xpos && !r.ylx && r.ax < r.ay;
    const double base = q3 || q2 ? hpi : opi, base1 = q3 || q2 ? hpi1 : opi1;
    const uint64_t sg = q3 ? 0 : 0x8000000000000000ull;
    const double su = tlm_u2d(tlm_d2u(u) ^ sg), sdu = tlm_u2d(tlm_d2u(m.du) ^ sg);
    const double t2 = base + su;
    const double cor = (base - t2) + su;
    const double zA2 = (((cor + base1) + sdu) + tlm_u2d(tlm_d2u(zz) ^ sg)) + t2;
    m.zA = TLM_SEL(q1, zA1, zA2);
    return m;
}
TLM_HD double tlm_atan2_tail(const TlmAtanA &r, const TlmAtanB &m, double c0, double c1, double c2, double c3, double c4, double c5, double c6)
{
    const double hpi = TLM_D(0x3ff921fb54442d18), hpi1 = TLM_D(0x3c91a62633145c07), opi = TLM_D(0x400921fb54442d18),
                 opi1 = TLM_D(0x3ca1a62633145c07);
    const double u = r.u, du = m.du;
    const double t3 = u - c0;
    const double w = t3 + du;
    const double p3 = TLM_FMA(TLM_FMA(TLM_FMA(c6, w, c5), w, c4), w, c3);
    const double p2 = TLM_FMA(p3, w, c2);
    const bool q1 = r.xpos && r.ylx, q2 = r.xpos && !r.ylx, q3 = !r.xpos && !r.ylx && r.ax < r.ay;
    const double at3 = t3 < 0 ? -t3 : t3, adu = du < 0 ? -du : du;
    const double dv = TLM_SEL(at3 > adu, (t3 - w) + du, (du - w) + t3);
    const double zB1 = c1 + TLM_FMA(w, c2, TLM_FMA(dv, c2, (w * w) * p3));
    const double base = q3 || q2 ? hpi : opi, base1 = q3 || q2 ? hpi1 : opi1;
    const uint64_t sg = q3 ? 0 : 0x8000000000000000ull;
    const double zB2 = (base + tlm_u2d(tlm_d2u(c1) ^ sg)) + TLM_FMA(tlm_u2d(tlm_d2u(w) ^ sg), p2, base1);
    const double zB = TLM_SEL(q1, zB1, zB2);
    double z = TLM_SEL(u < 0.0625, m.zA, zB);
    // extreme ratios and zeros: an explicit, rarely taken branch BEHIND form B.  (As selects the compiler turns them into a branch of its
    // own around the division and then sinks form B -- with the loads of the table row -- into that branch's other side, past every fence
    // the caller has set.)
    if (__builtin_expect(r.de <= -59768832 || r.de >= 59768832 || (r.bx << 1) == 0 || (r.by << 1) == 0, 0)) {
        z = r.de <= -59768832 ? (r.xpos ? r.ay / r.ax : opi) : z;
        z = r.de >= 59768832 ? hpi : z;
        z = (r.bx << 1) == 0 ? hpi : z;
        z = (r.by << 1) == 0 ? (r.xpos ? 0.0 : opi) : z;
    }
    return tlm_u2d((tlm_d2u(z) & 0x7fffffffffffffffull) | (r.by & 0x8000000000000000ull));
}
template <typename TAB>
TLM_HD double tlm_atan2_phased(double y, double x, TAB cij)
{
    const TlmAtanA r = tlm_atan2_head(y, x);
    const double c0 = tlm_u2d(cij[7 * r.i]), c1 = tlm_u2d(cij[7 * r.i + 1]), c2 = tlm_u2d(cij[7 * r.i + 2]), c3 = tlm_u2d(cij[7 * r.i + 3]),
                 c4 = tlm_u2d(cij[7 * r.i + 4]), c5 = tlm_u2d(cij[7 * r.i + 5]), c6 = tlm_u2d(cij[7 * r.i + 6]);
    const TlmAtanB m = tlm_atan2_mid(r);
    return tlm_atan2_tail(r, m, c0, c1, c2, c3, c4, c5, c6);
}

// exp / pow(10, y) for results inside the normal range (|x| < 512 resp. |y * ln 10| < 512), no branches.
template <bool WITH_TAIL>
TLM_HD double tlm_exp_sl(double x, double xtail)
{
    const double invln2N = tlm_u2d(tlm_exp_head[0]), shift = tlm_u2d(tlm_exp_head[1]),
                 negln2hiN = tlm_u2d(tlm_exp_head[2]), negln2loN = tlm_u2d(tlm_exp_head[3]),
                 C2 = tlm_u2d(tlm_exp_head[4]), C3 = tlm_u2d(tlm_exp_head[5]), C4 = tlm_u2d(tlm_exp_head[6]),
                 C5 = tlm_u2d(tlm_exp_head[7]);
    const uint32_t abstop = (uint32_t)(tlm_d2u(x) >> 52) & 0x7ff;
    const double kd0 = TLM_FMA(x, invln2N, shift);
    const uint64_t ki = tlm_d2u(kd0);
    const double kd = kd0 - shift;
    double r = TLM_FMA(kd, negln2loN, TLM_FMA(kd, negln2hiN, x));
    if (WITH_TAIL) r = xtail + r;
    const int idx = 2 * (int)(ki & 127);
    const double tail = tlm_u2d(tlm_exp_tab[idx]);
    const double scale = tlm_u2d(tlm_exp_tab[idx + 1] + (ki << 45));
    const double r2 = r * r;
    const double tmp = TLM_FMA(TLM_FMA(r, C5, C4), r2 * r2, TLM_FMA(TLM_FMA(C3, r, C2), r2, r + tail));
    const double res = TLM_FMA(scale, tmp, scale);
    return abstop < 0x3c9 ? 1.0 + x : res;                          // |x| < 2^-54
}
TLM_HD double tlm_pow10_sl(double y)
{
    const double hi = tlm_u2d(TLM_LOG10_HI), lo = tlm_u2d(TLM_LOG10_LO);
    const double ehi = y * hi;
    const double elo = TLM_FMA(y, lo, TLM_FMA(hi, y, -ehi));
    const double res = tlm_exp_sl<true>(ehi, elo);
    return ((uint32_t)(tlm_d2u(y) >> 52) & 0x7ff) < 0x3be ? 1.0 : res;      // |y| < 2^-65, +-0
}
