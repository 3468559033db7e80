// libm_agree -- csrc/tl_libm.h against the host's libm, bit for bit.
//
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -pthread -I odr-audioenc_amd/csrc tools/libm_agree.cpp -o build/libm_agree -lm
//   build/libm_agree [millions of arguments per function, default 100] [threads, default all] [seed]
//
// For every function the encoder's device path takes from tl_libm.h, draws arguments from several distributions
// (the encoder's own ranges, all binades, neighbourhoods of the branch points of each routine) and compares the
// result with what libm.so.6 of THIS machine returns (raw bits; NaN equals NaN).  Prints one line per function:
// arguments tried, results differing, first few offenders.  Exit status 1 if anything differs.
// Test infrastructure (third-party arithmetic = glibc 2.35 libm); output kept in profiles/.
#include <math.h>
#include <gnu/libc-version.h>
#include <string>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>
#include <mutex>
#include "tl_libm.h"

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9e3779b97f4a7c15ull + 0x1234567) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545f4914f6cdd1dull; }
    double unit() { return (double)(next() >> 11) * 0x1p-53; }                       // [0, 1)
    double uniform(double a, double b) { return a + (b - a) * unit(); }
    double mant() { return tlm_u2d(0x3ff0000000000000ull | (next() >> 12)); }        // [1, 2)
    double binade(int emin, int emax) { return ldexp(mant(), emin + (int)(next() % (uint64_t)(emax - emin + 1))); }
    double sign() { return (next() & 1) ? -1.0 : 1.0; }
};

static bool same(double a, double b) { return (a != a && b != b) || tlm_d2u(a) == tlm_d2u(b); }

struct Tally {
    const char *name;
    std::atomic<uint64_t> n{0}, bad{0};
    std::mutex m;
    std::vector<std::string> examples;
    void miss(const char *fmt, double a, double b, double got, double want)
    {
        bad++;
        std::lock_guard<std::mutex> g(m);
        if (examples.size() < 5) {
            char buf[256];
            snprintf(buf, sizeof buf, fmt, a, b, got, want);
            examples.push_back(buf);
        }
    }
};

static Tally T_log{"log"}, T_log10{"log10"}, T_log10pn{"log10 (straight-line form, positive normal)"}, T_log10split{"log10 (table branch / near-1 branch apart, positive normal)"}, T_exp{"exp"},
    T_pow10{"pow(10, y)"}, T_pow{"pow(x > 0, y)"}, T_sin{"sincos: sin"}, T_cos{"sincos: cos"}, T_atan2{"atan2"},
    T_sinsl{"sincos, straight-line form: sin"}, T_cossl{"sincos, straight-line form: cos"}, T_atan2sl{"atan2, straight-line form"}, T_atan2ns{"atan2, straight-line, no rescaling (max|.| in [2^-443, 2^500])"}, T_expsl{"exp, straight-line form (|x| < 512)"},
    T_pow10sl{"pow(10, y), straight-line form (|y| < 222)"},
    T_divns{"n / d by a refined reciprocal, no exponent scaling (seed perturbed by up to 2^-20)"}, T_sqrtns{"sqrt by refinement, no exponent scaling (seed perturbed by up to 2^-20)"};

static void chk1(Tally &t, double x, double got, double want)
{
    t.n++;
    if (!same(got, want)) t.miss("x=%a%.0s got %a want %a", x, 0, got, want);
}

static void worker(int id, uint64_t per_fn, uint64_t seed)
{
    Rng r(seed * 1000 + id);
    for (uint64_t i = 0; i < per_fn; i++) {
        // ---- log / log10
        double x;
        switch (i % 8) {
        case 0: x = r.binade(-1074, 1023); break;                                    // everything incl. subnormal-ish
        case 1: x = r.uniform(0.9, 1.1); break;                                      // around the near-1 branch
        case 2: x = r.mant() * (r.next() & 1 ? 1.0 : 0.5); break;                    // what log10 hands to log
        case 3: x = r.uniform(1e-20, 1e-10); break;
        case 4: x = r.binade(-70, 40); break;                                        // energies of the encoder
        case 5: x = tlm_u2d((r.next() % 0x000fffffffffffffull) + 1); break;          // subnormals
        case 6: x = tlm_u2d(0x3fee000000000000ull + (r.next() % 0x0003090000000000ull) - 0x800 + (r.next() & 0xfff)); break;
        default: x = r.uniform(0, 70000.0 * 70000.0); break;
        }
        chk1(T_log, x, tlm_log(x), log(x));
        chk1(T_log10, x, tlm_log10(x), log10(x));
        if (tlm_d2u(x) >= 0x0010000000000000ull && tlm_d2u(x) < 0x7ff0000000000000ull)
            chk1(T_log10pn, x, tlm_log10_pn(x, tlm_log_tab), log10(x));
        if (tlm_d2u(x) >= 0x0010000000000000ull && tlm_d2u(x) < 0x7ff0000000000000ull) {
            bool near1;
            const double a = tlm_log10_main_pn(x, tlm_log_tab, &near1);
            chk1(T_log10split, x, near1 ? tlm_log10_near1_pn(x) : a, log10(x));
        }
        // ---- exp
        switch (i % 6) {
        case 0: x = r.uniform(-750, 715); break;
        case 1: x = r.uniform(-60, 5); break;                                        // encoder: exp(-bc * LN_TO_LOG10)
        case 2: x = r.sign() * r.binade(-60, 10); break;
        case 3: x = r.uniform(-1, 1); break;
        case 4: x = r.sign() * r.binade(-1074, -50); break;
        default: x = r.uniform(-745.2, -707.0); break;                               // subnormal results
        }
        chk1(T_exp, x, tlm_exp(x), exp(x));
        if (fabs(x) < 512) chk1(T_expsl, x, tlm_exp_sl<false>(x, 0.0), exp(x));
        // ---- pow(10, y), pow(x, y)
        switch (i % 5) {
        case 0: x = r.uniform(-330, 310); break;
        case 1: x = r.uniform(-20, 20); break;                                       // encoder: -0.1 * (dB sum)
        case 2: x = r.sign() * r.binade(-80, 8); break;
        case 3: x = r.uniform(-324, -300); break;
        default: x = -0.1 * (double)(int64_t)(r.next() % 4001 - 2000) * 0.1; break;
        }
        chk1(T_pow10, x, tlm_pow10(x), pow(10.0, x));
        if (fabs(x) < 222) chk1(T_pow10sl, x, tlm_pow10_sl(x), pow(10.0, x));
        {
            const double bx = (i & 1) ? r.binade(-300, 300) : r.uniform(0.01, 100.0);
            const double lim = 700.0 / fabs(log(bx) + 1e-300);
            const double by = (i & 2) ? r.uniform(-1, 1) * (lim < 1e6 ? lim : 1e6) : r.sign() * r.binade(-70, 3);
            T_pow.n++;
            const double got = tlm_pow_pos(bx, by), want = pow(bx, by);
            if (!same(got, want)) T_pow.miss("x=%a y=%a got %a want %a", bx, by, got, want);
        }
        // ---- sincos
        switch (i % 6) {
        case 0: x = r.uniform(-12, 12); break;                                       // encoder: phi, 2*phi - phi'
        case 1: x = r.uniform(-0.9, 0.9); break;
        case 2: x = r.sign() * r.binade(-40, 25); break;                                // < 105414350: reduce_sincos range
        case 3: x = r.uniform(-105414350.0, 105414350.0); break;
        case 4: x = r.sign() * (r.uniform(0.8, 2.5)); break;
        default: x = (double)(int64_t)(r.next() % 64) * 0.78539816339744830962 * r.sign() + r.uniform(-1e-6, 1e-6) * (r.next() & 1); break;
        }
        {
            double s, c, s0, c0;
            tlm_sincos(x, &s, &c);
            sincos(x, &s0, &c0);
            chk1(T_sin, x, s, s0);
            chk1(T_cos, x, c, c0);
            tlm_sincos_sl(x, &s, &c, tlm_sincostab);
            chk1(T_sinsl, x, s, s0);
            chk1(T_cossl, x, c, c0);
        }
        // ---- atan2
        {
            double y, xx;
            switch (i % 8) {
            case 0: y = r.sign() * r.binade(-40, 40); xx = r.sign() * r.binade(-40, 40); break;
            case 1: y = r.uniform(-1, 1); xx = r.uniform(-1, 1); break;              // spectra
            case 2: y = r.sign() * r.binade(-1022, 1023); xx = r.sign() * r.binade(-1022, 1023); break;
            case 3: xx = r.sign() * r.binade(-20, 20); y = xx * r.uniform(-1.001, 1.001); break;   // |y| ~ |x|
            case 4: xx = r.sign() * r.binade(-20, 20); y = xx * r.uniform(-0.07, 0.07); break;     // around u = 1/16
            case 5: y = r.sign() * r.binade(-20, 20); xx = y * r.uniform(-0.07, 0.07); break;
            case 6: y = (double)(int64_t)(r.next() % 65536 - 32768); xx = (double)(int64_t)(r.next() % 65536 - 32768); break;
            default: y = (r.next() % 16 == 0) ? 0.0 * r.sign() : r.uniform(-3e4, 3e4) * r.unit();
                     xx = (r.next() % 16 == 0) ? 0.0 * r.sign() : r.uniform(-3e4, 3e4) * r.unit(); break;
            }
            T_atan2.n++;
            const double got = tlm_atan2(y, xx), want = atan2(y, xx);
            if (!same(got, want)) T_atan2.miss("y=%a x=%a got %a want %a", y, xx, got, want);
            T_atan2sl.n++;
            const double got2 = tlm_atan2_sl(y, xx, tlm_atan_cij);
            if (!same(got2, want)) T_atan2sl.miss("y=%a x=%a got %a want %a", y, xx, got2, want);
            {   // the unscaled division / square root cores of round 6: any seed of hardware quality must give the correctly rounded result
                double d, n;
                switch (i % 5) {
                case 0: d = r.binade(-500, 500); n = r.binade(-900, 500); break;                 // the stated domain
                case 1: d = r.uniform(0.02, 4e7); n = d * r.uniform(0x1p-57, 1.0); break;            // atan2 of the encoder: num / den
                case 2: d = r.mant(); n = r.mant() * (r.next() & 1 ? 1.0 : 0.5); break;
                case 3: d = r.binade(-5, 30); n = d * r.mant() * 0x1p-106; break;                     // the second quotient: a residual over den
                default: d = r.uniform(0.02, 4e9); n = r.uniform(0.0, 4e9); break;                   // c = sqrt(..) / (r + |r'|)
                }
                const double pert = 1.0 + r.uniform(-0x1p-20, 0x1p-20);
                T_divns.n++;
                const double q = tlm_div_by_recip(n, d, tlm_recip_from(d, (1.0 / d) * pert));
                if (!same(q, n / d)) T_divns.miss("n=%a d=%a got %a want %a", n, d, q, n / d);
                double x2;
                switch (i % 3) { case 0: x2 = r.binade(-500, 500); break; case 1: x2 = r.uniform(0.0005, 1e18); break; default: x2 = r.mant() * (r.next() & 1 ? 1.0 : 2.0); break; }
                T_sqrtns.n++;
                const double sq = tlm_sqrt_from(x2, (1.0 / sqrt(x2)) * pert);
                if (!same(sq, sqrt(x2))) T_sqrtns.miss("x=%a%.0s got %a want %a", x2, 0.0, sq, sqrt(x2));
            }
            const double big = fabs(y) > fabs(xx) ? fabs(y) : fabs(xx);
            if (big >= 0x1p-443 && big <= 0x1p500) {          // the domain of the form without operand rescaling
                T_atan2ns.n++;
                const double got3 = tlm_atan2_sl<false>(y, xx, tlm_atan_cij);
                if (!same(got3, want)) T_atan2ns.miss("y=%a x=%a got %a want %a", y, xx, got3, want);
                const double got4 = tlm_atan2_phased(y, xx, tlm_atan_cij);        // head + mid + tail, what the psy-2 kernel's line loop calls
                if (!same(got4, want)) T_atan2ns.miss("(phased) y=%a x=%a got %a want %a", y, xx, got4, want);
            }
        }
    }
}

int main(int argc, char **argv)
{
    const uint64_t millions = argc > 1 ? strtoull(argv[1], 0, 10) : 100;
    unsigned nthreads = argc > 2 ? (unsigned)atoi(argv[2]) : std::thread::hardware_concurrency();
    const uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 10) : 1;
    if (nthreads < 1) nthreads = 1;
    const uint64_t per = millions * 1000000ull / nthreads;
    printf("libm_agree: tl_libm.h against this machine's libm (%s), %llu M arguments per function, %u threads, seed %llu\n",
           gnu_get_libc_version(), (unsigned long long)millions, nthreads, (unsigned long long)seed);
    {   // the two words of log_inline(10.0) that tlm_pow10 carries as constants
        double lo;
        const double hi = tlm_pow_log(tlm_d2u(10.0), &lo);
        const bool ok = tlm_d2u(hi) == TLM_LOG10_HI && tlm_d2u(lo) == TLM_LOG10_LO;
        printf("log_inline(10.0) = %a + %a (%#llx, %#llx): constants in tl_libm.h %s\n", hi, lo,
               (unsigned long long)tlm_d2u(hi), (unsigned long long)tlm_d2u(lo), ok ? "match" : "DO NOT MATCH");
        if (!ok) return 1;
    }
    {   // the constants line 512 of psy model 2 uses (mp2_wave.h tl_psy2): sincos of the double nearest pi, atan2(+0, x < 0)
        double s0, c0;
        sincos(3.141592653589793116, &s0, &c0);
        const bool ok = tlm_d2u(s0) == 0x3ca1a62633145c07ull && tlm_d2u(c0) == 0xbff0000000000000ull &&
                        tlm_d2u(atan2(0.0, -1.0)) == 0x400921fb54442d18ull && tlm_d2u(atan2(0.0, -0.0)) == 0x400921fb54442d18ull && atan2(0.0, 2.0) == 0.0;
        printf("sincos(pi) = (%a, %a), atan2(+0, -1) = %a: constants in mp2_wave.h %s\n", s0, c0, atan2(0.0, -1.0), ok ? "match" : "DO NOT MATCH");
        if (!ok) return 1;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(worker, (int)t, per, seed);
    for (auto &t : th) t.join();
    uint64_t bad = 0;
    for (Tally *t : {&T_log, &T_log10, &T_log10pn, &T_log10split, &T_exp, &T_pow10, &T_pow, &T_sin, &T_cos, &T_atan2, &T_sinsl, &T_cossl, &T_atan2sl, &T_atan2ns, &T_expsl, &T_pow10sl, &T_divns, &T_sqrtns}) {
        printf("%-48s %12llu arguments  %llu differ (%.4f %% bit-equal)\n", t->name, (unsigned long long)t->n.load(),
               (unsigned long long)t->bad.load(), 100.0 * (double)(t->n - t->bad) / (double)t->n);
        for (auto &e : t->examples) printf("    %s\n", e.c_str());
        bad += t->bad;
    }
    return bad ? 1 : 0;
}
