// mp2_fht.h -- Hann window + 1024-point fast Hartley transform + energies of one channel (psycho_1.c:57-76,215-239, fft.c:1092-1293): tl_psy_spectrum.
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ------------------------------------------------------------------------------------------
// K3: 1024-point FHT (fft.c:78-1185), parallel over 64 lanes.  The swap list of fft.c:85-1090 is the 10-bit reversal.
//
// Head in registers: lane L loads the windowed samples i = L + 64*it (it = 0..15); their bit-reversed slots are
// 16*rev6(L) + rev4(it) -- exactly one block of 16 consecutive points, the unit the first pass (groups of four,
// fft.c:1092-1102) and the k=2 pass (fft.c:1104-1184 with k1=4) work on.  So the lane runs both passes on its own sixteen
// values without touching LDS and stores the block once (tl_fht_head / tl_fht_store).
// Layout: logical index i lives at i ^ (i >> 5) (a permutation inside each group of 32 doubles).  With it the stored
// blocks, the k=4/6/8 butterflies and the energy reads spread over the LDS banks (at most ~3 lanes per bank instead of
// up to 32); index fields that do not share bits pass through the map separately: FX(a|b) = FX(a) ^ FX(b).
#define TL_FX(i) ((i) ^ ((i) >> 5))
TL_FN void tl_fht_head(double (&e)[16], const double (*TL_RESTRICT tw)[4])
{
    const double SQRT2 = 1.4142135623730951454746218587388284504414;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int g = 0; g < 16; g += 4) {                               // fft.c:1092-1102
        const double f1 = e[g] - e[g + 1], f0 = e[g] + e[g + 1], f3 = e[g + 2] - e[g + 3], f2 = e[g + 2] + e[g + 3];
        e[g + 2] = f0 - f2; e[g] = f0 + f2; e[g + 3] = f1 - f3; e[g + 1] = f1 + f3;
    }
    {   // k=2 pass, i = 0: fi = block, gi = block + 2 (k1 = 4, k2 = 8, k3 = 12)
        const double f1 = e[0] - e[4], f0 = e[0] + e[4], f3 = e[8] - e[12], f2 = e[8] + e[12];
        e[8] = f0 - f2; e[0] = f0 + f2; e[12] = f1 - f3; e[4] = f1 + f3;
        const double g1 = e[2] - e[6], g0 = e[2] + e[6], g3 = SQRT2 * e[14], g2 = SQRT2 * e[10];
        e[10] = g0 - g2; e[2] = g0 + g2; e[14] = g1 - g3; e[6] = g1 + g3;
    }
    {   // k=2 pass, i = 1: fi = block + 1, gi = block + 3; one twiddle set for every block
        const double c1 = tw[0][0], s1 = tw[0][1], c2 = tw[0][2], s2 = tw[0][3];
        double a, b2, g0, f0, f1, g1, f2, g2, f3, g3;
        b2 = s2 * e[5] - c2 * e[7]; a = c2 * e[5] + s2 * e[7];
        f1 = e[1] - a; f0 = e[1] + a; g1 = e[3] - b2; g0 = e[3] + b2;
        b2 = s2 * e[13] - c2 * e[15]; a = c2 * e[13] + s2 * e[15];
        f3 = e[9] - a; f2 = e[9] + a; g3 = e[11] - b2; g2 = e[11] + b2;
        b2 = s1 * f2 - c1 * g3; a = c1 * f2 + s1 * g3;
        e[9] = f0 - a; e[1] = f0 + a; e[15] = g1 - b2; e[7] = g1 + b2;
        b2 = c1 * g2 - s1 * f3; a = s1 * g2 + c1 * f3;
        e[11] = g0 - a; e[3] = g0 + a; e[13] = f1 - b2; e[5] = f1 + b2;
    }
}
TL_FN double *tl_fht_at(double *x, int byte_offset) { return (double *)((char *)x + byte_offset); }
TL_FN int tl_rev6(int lane) { int r = 0; for (int b = 0; b < 6; b++) r |= ((lane >> b) & 1) << (5 - b); return r; }
TL_FN void tl_fht_store(double *x, int lane, const double (&e)[16])
{
    const int l = tl_rev6(lane), base = ((16 * l) ^ (l >> 1)) << 3;  // FX(16*l + t) = base ^ t for t < 16 (as byte offsets: tl_fht_at)
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int t = 0; t < 16; t++) *tl_fht_at(x, base ^ (t << 3)) = e[t];
}
// Twiddles (c1,s1,c2,s2) of the (up to) two general butterflies a lane runs in pass K, and where the butterflies are (TlTables::fht_fg_lane:
// the byte offsets of their points f0 and g0 in the transform buffer, 16 bits each); fetched one pass ahead.
template <int K>
TL_FN void tl_fht_twiddles(double (&t)[8], uint32_t (&fg)[2], const TlTables *TL_RESTRICT T, int lane)
{   // rows in lane order (TlTables::fht_tw_lane): one address per lane, no index arithmetic
    const double (*tw)[4] = T->fht_tw_lane[(K - 4) / 2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) t[4 * it + q] = tw[lane + 64 * it][q];
        fg[it] = T->fht_fg_lane[(K - 4) / 2][lane + 64 * it];
    }
}
template <int K>
TL_FN void tl_fht_pass(double *x, const double (&t)[8], const uint32_t (&fg)[2], int lane)
{   // fft.c:1104-1184: one pass = 128 independent 8-point butterflies: per block of 4*k1 points one with trivial /
    // sqrt(2) twiddles (i = 0) and kx-1 general ones.  The general ones are dealt to 128 slots (two per lane) by the host's table -- which slot
    // runs which (block, i) is chosen for the LDS banks, csrc/mp2_host.cpp -- and the trivial ones follow in their own step, so a wave never
    // runs both code paths for one batch of butterflies.
    // Addresses: block, i (or k1-i) and q*k1 occupy disjoint bit fields, so each goes through TL_FX on its own.
    const double SQRT2 = 1.4142135623730951454746218587388284504414;
    constexpr int k1 = 1 << K, k2 = k1 << 1, k4 = k2 << 1, k3 = k2 + k1, kx = k1 >> 1;
    constexpr int NBLK = 128 / kx;
    constexpr int q1 = TL_FX(k1), q2 = TL_FX(k2), q3 = TL_FX(k3);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int g = lane + 64 * it;
        if (fg[it] == 0xffffffffu || g >= 127) continue;            // an idle slot of the host's dealing; k=8's slot 127 is that pass's TRIVIAL butterfly (below)
        const double c1 = t[4 * it], s1 = t[4 * it + 1], c2 = t[4 * it + 2], s2 = t[4 * it + 3];
        // (byte offsets: an exchange partner's address is (offset ^ constant) + base, ONE v_xad_u32; as indices it is an exclusive-or and then a shift-add)
        // slot g = the general butterfly (block, i) the host dealt to it -- by LDS banks, csrc/mp2_host.cpp tl_build_tables --: f0 at FX(block * k4) ^ FX(i),
        // g0 at FX(block * k4) ^ FX(k1 - i), as byte offsets from the table (the dealing and the layout map are the host's)
        const int F = (int)(fg[it] & 0xffffu), G = (int)(fg[it] >> 16);
        double *f0p = tl_fht_at(x, F), *f1p = tl_fht_at(x, F ^ (q1 << 3)), *f2p = tl_fht_at(x, F ^ (q2 << 3)), *f3p = tl_fht_at(x, F ^ (q3 << 3));
        double *g0p = tl_fht_at(x, G), *g1p = tl_fht_at(x, G ^ (q1 << 3)), *g2p = tl_fht_at(x, G ^ (q2 << 3)), *g3p = tl_fht_at(x, G ^ (q3 << 3));
        double a, b2, g0, f0, f1, g1, f2, g2, f3, g3;
        b2 = s2 * *f1p - c2 * *g1p; a = c2 * *f1p + s2 * *g1p;
        f1 = *f0p - a; f0 = *f0p + a; g1 = *g0p - b2; g0 = *g0p + b2;
        b2 = s2 * *f3p - c2 * *g3p; a = c2 * *f3p + s2 * *g3p;
        f3 = *f2p - a; f2 = *f2p + a; g3 = *g2p - b2; g2 = *g2p + b2;
        b2 = s1 * f2 - c1 * g3; a = c1 * f2 + s1 * g3;
        *f2p = f0 - a; *f0p = f0 + a; *g3p = g1 - b2; *g1p = g1 + b2;
        b2 = c1 * g2 - s1 * f3; a = s1 * g2 + c1 * f3;
        *g2p = g0 - a; *g0p = g0 + a; *f3p = f1 - b2; *f1p = f1 + b2;
    }
    if (lane < NBLK) {
        const int F = TL_FX(lane * k4) << 3, G = F ^ (TL_FX(kx) << 3);
        double *f0p = tl_fht_at(x, F), *f1p = tl_fht_at(x, F ^ (q1 << 3)), *f2p = tl_fht_at(x, F ^ (q2 << 3)), *f3p = tl_fht_at(x, F ^ (q3 << 3));
        double *g0p = tl_fht_at(x, G), *g1p = tl_fht_at(x, G ^ (q1 << 3)), *g2p = tl_fht_at(x, G ^ (q2 << 3)), *g3p = tl_fht_at(x, G ^ (q3 << 3));
        double f1 = *f0p - *f1p, f0 = *f0p + *f1p, f3 = *f2p - *f3p, f2 = *f2p + *f3p;
        *f2p = f0 - f2; *f0p = f0 + f2; *f3p = f1 - f3; *f1p = f1 + f3;
        double g1 = *g0p - *g1p, g0 = *g0p + *g1p, g3 = SQRT2 * *g3p, g2 = SQRT2 * *g2p;
        *g2p = g0 - g2; *g0p = g0 + g2; *g3p = g1 - g3; *g1p = g1 + g3;
    }
}

// Hann window of samples [t-192, t+832) + FHT + energy (psycho_1.c:57-76,215-239, fft.c:1278-1293).
// Leaves energy[i] in w.u.fft[TL_EX(i)], i = 0..512.
// A stream's PCM as the kernel sees it in HBM: this frame (planar [2][1152]) and the 480 samples per
// channel that precede it (the stream state on the first frame of a launch, the previous input frame after).
// Per "channel" c of the wave: the two channels of a stereo stream -- or, for a PAIR of mono streams sharing a wave (tl_frame_unit),
// channel 0 of each of the two streams.
struct TlPcmView { const int16_t *cur[2]; const int16_t *hist[2]; };

TL_FN void tl_psy_spectrum(TlPsyLds &w, const TlTables *TL_RESTRICT T, const TlPcmView &pv, int ch, long long *sp)
{
    double *x = w.u.fft;
    long long *sq = (sp && ch == 0) ? sp + 16 : nullptr;      // channel 0's pass-by-pass stamps: slots 24..30 of the frame's record
    TL_STAMP(sq, 0);
    // twiddles travel one pass ahead of their use (twc: k=4 with the window, twb: k=6 during pass 4, twa: k=8 during pass 6): two sets live at most
    PA(double, twa, 8); PA(double, twb, 8); PA(double, twc, 8);
    PA(uint32_t, fga, 2); PA(uint32_t, fgb, 2); PA(uint32_t, fgc, 2);
    TL_LANES_BEGIN
    {
        // sample i = lane + 64*it of the analysis window: the last 192 samples of the history (it < 3), then the
        // first 832 of the frame.  The loads are issued in batches ahead of their use.  Slot of i inside the lane's block
        // of sixteen: rev4(it).
        const int16_t *hs = (ch ? pv.hist[1] : pv.hist[0]) + (TL_HIST - 192) + lane;      // (a select, not an indexed array: that would live in scratch)
        const int16_t *cs = (ch ? pv.cur[1] : pv.cur[0]) - 192 + lane;
        // (sample / 32768) * window (psycho_1.c:57-76) as sample * (window / 32768): the division by a power of two is exact on either
        // factor and commutes with the rounding of the product (nothing near the subnormal range), so the bits are the same
        const double *hann = T->hann_s;
        TL_LAUNDER(hann);
        tl_fht_twiddles<4>(L(twc), L(fgc), T, lane);
        double e[16];
        // the sixteen PCM samples are this unit's first touch of its input (HBM, not L2): all of them are requested before anything is
        // used; the window's coefficients (L2) follow in two batches of eight (sixteen doubles more in flight would spill)
        int vs[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 16; it++) vs[it] = it < 3 ? hs[64 * it] : cs[64 * it];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 16; it++) TL_KEEP(vs[it]);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int half = 0; half < 16; half += 8) {
            double h[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) h[q] = hann[lane + 64 * (half + q)];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) {
                const int it = half + q;
                const int r4 = ((it & 1) << 3) | ((it & 2) << 1) | ((it & 4) >> 1) | ((it & 8) >> 3);
                e[r4] = (double)vs[it] * h[q];
            }
        }
        tl_fht_head(e, T->fht_tw);
        tl_fht_store(x, lane, e);
    }
    TL_LANES_END
    TL_STAMP(sq, 1);
    TL_STAMP(sq, 2);
    TL_STAMP(sq, 3);
    TL_LANES_BEGIN tl_fht_twiddles<6>(L(twb), L(fgb), T, lane); tl_fht_pass<4>(x, L(twc), L(fgc), lane); TL_LANES_END
    TL_STAMP(sq, 4);
    TL_LANES_BEGIN tl_fht_twiddles<8>(L(twa), L(fga), T, lane); tl_fht_pass<6>(x, L(twb), L(fgb), lane); TL_LANES_END
    TL_STAMP(sq, 5);
    // Last pass (k=8) and energies (fft.c:1278-1293) in one go: the eight outputs of a k=8 butterfly are x[i+256q] and
    // x[256-i+256q], and line j pairs with 1024-j -- so butterfly i holds both members of the pairs of lines i, 256-i, 256+i
    // and 512-i (the trivial butterfly: lines 0, 128, 256, 384, 512).  The transform is never written back: every input is
    // read first (the energies go to natural positions, which are other lanes' inputs), then each lane squares its own pairs.
    {
        constexpr int k1 = 256, kx = 128;
        constexpr int q1 = TL_FX(256), q2 = TL_FX(512), q3 = TL_FX(768);
        PA(double, fv, 8); PA(double, gv, 8);
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 2; it++) {
            const int g = lane + 64 * it;                            // general butterflies i = 1 + g (g < 127); g = 127: the trivial one
            // (byte offsets; the table's entry 127 of this pass is the trivial butterfly: f0 at 0, g0 at FX(kx))
            const int F = (int)(L(fga)[it] & 0xffffu), G = (int)(L(fga)[it] >> 16);
            L(fv)[4 * it] = *tl_fht_at(x, F); L(fv)[4 * it + 1] = *tl_fht_at(x, F ^ (q1 << 3)); L(fv)[4 * it + 2] = *tl_fht_at(x, F ^ (q2 << 3)); L(fv)[4 * it + 3] = *tl_fht_at(x, F ^ (q3 << 3));
            L(gv)[4 * it] = *tl_fht_at(x, G); L(gv)[4 * it + 1] = *tl_fht_at(x, G ^ (q1 << 3)); L(gv)[4 * it + 2] = *tl_fht_at(x, G ^ (q2 << 3)); L(gv)[4 * it + 3] = *tl_fht_at(x, G ^ (q3 << 3));
        }
        TL_LANES_END
        TL_LANES_BEGIN
        const double SQRT2 = 1.4142135623730951454746218587388284504414;
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 2; it++) {
            const int g = lane + 64 * it;
            const double fi0 = L(fv)[4 * it], fi1 = L(fv)[4 * it + 1], fi2 = L(fv)[4 * it + 2], fi3 = L(fv)[4 * it + 3];
            const double gi0 = L(gv)[4 * it], gi1 = L(gv)[4 * it + 1], gi2 = L(gv)[4 * it + 2], gi3 = L(gv)[4 * it + 3];
            if (g < 127) {
                const int i = 1 + g;
                const double c1 = L(twa)[4 * it], s1 = L(twa)[4 * it + 1], c2 = L(twa)[4 * it + 2], s2 = L(twa)[4 * it + 3];
                double a, b2, g0, f0, f1, g1, f2, g2, f3, g3;
                b2 = s2 * fi1 - c2 * gi1; a = c2 * fi1 + s2 * gi1;
                f1 = fi0 - a; f0 = fi0 + a; g1 = gi0 - b2; g0 = gi0 + b2;
                b2 = s2 * fi3 - c2 * gi3; a = c2 * fi3 + s2 * gi3;
                f3 = fi2 - a; f2 = fi2 + a; g3 = gi2 - b2; g2 = gi2 + b2;
                b2 = s1 * f2 - c1 * g3; a = c1 * f2 + s1 * g3;
                const double o_f2 = f0 - a, o_f0 = f0 + a, o_g3 = g1 - b2, o_g1 = g1 + b2;     // x[i+512], x[i], x[1024-i], x[512-i]
                b2 = c1 * g2 - s1 * f3; a = s1 * g2 + c1 * f3;
                const double o_g2 = g0 - a, o_g0 = g0 + a, o_f3 = f1 - b2, o_f1 = f1 + b2;     // x[768-i], x[256-i], x[i+768], x[i+256]
                // E[j] = (x[j]^2 + x[1024-j]^2) / 2 with a = x[j] first, as in the reference
                x[TL_EX(i)] = (o_f0 * o_f0 + o_g3 * o_g3) / 2.0;
                x[TL_EX(256 - i)] = (o_g0 * o_g0 + o_f3 * o_f3) / 2.0;
                x[TL_EX(256 + i)] = (o_f1 * o_f1 + o_g2 * o_g2) / 2.0;
                x[TL_EX(512 - i)] = (o_g1 * o_g1 + o_f2 * o_f2) / 2.0;
            } else if (g == 127) {
                double f1 = fi0 - fi1, f0 = fi0 + fi1, f3 = fi2 - fi3, f2 = fi2 + fi3;
                const double o_f2 = f0 - f2, o_f0 = f0 + f2, o_f3 = f1 - f3, o_f1 = f1 + f3;     // x[512], x[0], x[768], x[256]
                double g1 = gi0 - gi1, g0 = gi0 + gi1, g3 = SQRT2 * gi3, g2 = SQRT2 * gi2;
                const double o_g2 = g0 - g2, o_g0 = g0 + g2, o_g3 = g1 - g3, o_g1 = g1 + g3;     // x[640], x[128], x[896], x[384]
                x[0] = o_f0 * o_f0;                                   // TL_EX leaves multiples of 256 where they are
                x[512] = o_f2 * o_f2;
                x[256] = (o_f1 * o_f1 + o_f3 * o_f3) / 2.0;
                x[TL_EX(128)] = (o_g0 * o_g0 + o_g3 * o_g3) / 2.0;
                x[TL_EX(384)] = (o_g1 * o_g1 + o_g2 * o_g2) / 2.0;
            }
        }
        TL_LANES_END
    }
    TL_STAMP(sq, 6);
}
