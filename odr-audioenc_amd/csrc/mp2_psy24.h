// mp2_psy24.h -- psy models 2 and 4 (psycho_2.c, psycho_4.c): one 576-sample pass of one channel (tl_psy2_pass).
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ------------------------------------------------------------------------------------------
// psy model 2 (psycho_2.c:52-254, psycho_2_fft fft.c:1230-1275), one 576-sample pass of channel `ch`.
// Two passes per frame; a pass needs the 480 samples before its 544 new ones -- the stream's PCM history on pass 0,
// samples 96..575 of the frame on pass 1 (the reference's savebuf shift by 576).
// Line-parallel: FFT, unpredictability (sincos/atan2/sqrt per line), thresholds; partition-parallel:
// grouping, spreading, SNR.  Every sum is one lane's sequential chain in the reference's order.
//
// The prediction state.  The reference keeps r = sqrt(energy) and phi of the two previous passes per line
// (psycho_2.c:111-116, 300-306) -- but there is no recurrence in it: both are functions of that pass's transform alone
// (`lthr`, the one true feedback of the model, is dead for Layer II, psycho_2.c:214-224).  So a run of passes can start
// anywhere: two SEED passes (transform, square root, arctangent -- no unpredictability, nothing after it) over the 1152
// samples before it rebuild exactly the state the chain would have carried there.  During a run the state lives in the wave's
// REGISTERS: line lane + 64 it in slot `it` of r1/p1 (previous pass) and r2/p2 (the pass before), line 512 in four LDS words.
// glibc's sincos table as the kernel holds it in LDS: the rows' halves apart -- (sn, ssn) of row k at [2 k], (cs, ccs) at [220 + 2 k] -- instead of
// 32-byte rows: a 16-byte gather of sixteen lanes then spreads over sixteen bank quads, not eight (row k's first half alone sat on banks 8 k .. 8 k + 3,
// the other four idle during that read).  `row` = 4 k as tlm_sincos_reduce returns it.  The emulation reads glibc's own layout.
#ifdef TL_EMULATE
#define TL_SCT(t, row, j) ((t)[(row) + (j)])
#else
#define TL_SCT(t, row, j) ((t)[((row) >> 1) + ((j) & 1) + 220 * ((j) >> 1)])
#endif
#define TL_P2_L512(w) ((w).px + 532)     /* r1, r2, p1, p2 of line 512 (c[] ends at px[512], the padded fthr[] at px[528]) */
template <bool SEED>
TL_FN void tl_psy2_pass(TlPsy2Lds &w, const TlTables *TL_RESTRICT T, const TlPsy2Tables *TL_RESTRICT P, const TlPcmView &pv, int ch, int pass,
                        PARGA(double, r1, 8), PARGA(double, r2, 8), PARGA(double, p1, 8), PARGA(double, p2, 8),
                        PARG(double, snr0), double *smr_out, const uint64_t *sct, long long *sq, const TlPsy2State *load_state = nullptr)
{   // sct: glibc's __sincostab (tl_libm.h), the workgroup's LDS copy on the device
    // load_state: the run's FIRST pass fetches the prediction state itself (32 KB per channel from HBM) -- behind the transform's own
    // loads and with three transform passes to arrive in, instead of at the head of the unit where the first wait for a PCM sample
    // is a wait for all of it (one frame per launch, the tick shape: a unit is two passes and this was a seventh of its time)
    double *x = w.u.fft;
    double *cw = w.px, *ge = w.u.fft + 520;  // c[] (unpredictability), then fthr[]; partition sums in the dead upper half of the FHT buffer:
    double *ecb = ge + 128, *nb = ecb + 64;  // ge[2 j], ge[2 j + 1] = grouped energy and weighted unpredictability of partition j (one 16-byte read per term of the spreading sums)
    double *l5 = TL_P2_L512(w);
    if (TL_P2_LEVEL >= 7) return;                                    // diagnostic builds only (mp2_wave.h)
    {
        TL_STAMP(sq, 0);
        PA(double, twa, 8); PA(double, twb, 8); PA(double, twc, 8);
        PA(uint32_t, fga, 2); PA(uint32_t, fgb, 2); PA(uint32_t, fgc, 2);
        TL_LANES_BEGIN
        {
            // sample i = lane + 64*it of the pass's 1024-sample window (psycho_2.c:84-92); loads in batches of eight ahead
            // of their use; slot of i inside the lane's block of sixteen: rev4(it) (see tl_fht_head)
            const double *win = P->window;
            TL_LAUNDER(win);
            const int16_t *pvh = ch ? pv.hist[1] : pv.hist[0], *pvc = ch ? pv.cur[1] : pv.cur[0];
            tl_fht_twiddles<4>(L(twc), L(fgc), T, lane);
            double e[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int half = 0; half < 16; half += 8) {
                int16_t v[8]; double h[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int q = 0; q < 8; q++) {
                    const int i = lane + 64 * (half + q);
                    if (pass == 0) v[q] = i < TL_HIST ? pvh[i] : pvc[i - TL_HIST];
                    else v[q] = pvc[96 + i];
                    h[q] = win[i];
                }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int q = 0; q < 8; q++) {
                    const int it = half + q;
                    const int r4 = ((it & 1) << 3) | ((it & 2) << 1) | ((it & 4) >> 1) | ((it & 8) >> 3);
                    e[r4] = h[q] * (double)v[q];
                }
            }
            tl_fht_twiddles<6>(L(twb), L(fgb), T, lane);
            tl_fht_head(e, T->fht_tw);
            tl_fht_store(x, lane, e);
        }
        TL_LANES_END
        PA(double, s5, 4);
        TL_LANES_BEGIN
        tl_fht_twiddles<8>(L(twa), L(fga), T, lane);
        if (load_state) {
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int it = 0; it < 8; it++) {
                const int j = lane + 64 * it;
                L(r1)[it] = load_state->r[ch][0][j]; L(r2)[it] = load_state->r[ch][1][j]; L(p1)[it] = load_state->phi[ch][0][j]; L(p2)[it] = load_state->phi[ch][1][j];
            }
            if (lane == 0) { L(s5)[0] = load_state->r[ch][0][512]; L(s5)[1] = load_state->r[ch][1][512]; L(s5)[2] = load_state->phi[ch][0][512]; L(s5)[3] = load_state->phi[ch][1][512]; }
        }
        tl_fht_pass<4>(x, L(twc), L(fgc), lane);
        TL_LANES_END
        TL_LANES_BEGIN
        tl_fht_pass<6>(x, L(twb), L(fgb), lane);
        if (load_state && lane == 0) { l5[0] = L(s5)[0]; l5[1] = L(s5)[1]; l5[2] = L(s5)[2]; l5[3] = L(s5)[3]; }
        TL_LANES_END
        TL_LANES_BEGIN tl_fht_pass<8>(x, L(twa), L(fga), lane); TL_LANES_END
        TL_STAMP(sq, 1);
        // energy + phase (fft.c:1246-1275), unpredictability (psycho_2.c:119-140).
        // 64 lines per step: the transform is read through the layout map first, then the step's energies are written in
        // natural order (their slots hold nothing a later step reads).
        // Lines 0..511 are eight full steps of the wave; line 512 would be a ninth with ONE lane at work, at the price of a full
        // step (two sincos, an atan2, two square roots for every lane).  It needs no arctangent of its own (its phase is 0 or pi,
        // fft.c:1274) and line 0 needs none either and no sincos of its phase (phi = 0, fft.c:1257-1259), so in step 0 lane 0 puts
        // line 512's PREDICTED phase through its first sincos slot and finishes that line with a few extra operations.
        PV(double, e512);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) {
            PV(double, xa); PV(double, xb); PV(double, xc);
            TL_LANES_BEGIN
            const int j = lane + 64 * it;
            L(xa) = x[TL_FX(j)];
            L(xb) = j >= 1 ? x[TL_FX(1024 - j)] : 0.0;
            L(xc) = it == 0 ? x[TL_FX(512)] : 0.0;
            TL_LANES_END
            TL_LANES_BEGIN
            const int j = lane + 64 * it;
            {
                const bool first = j == 0;                           // lane 0 of step 0: lines 0 and 512
                double r_o5 = 0, r_n5 = 0, p_o5 = 0, p_n5 = 0;      // state of line 512
                if (it == 0) { r_o5 = l5[0]; r_n5 = l5[1]; p_o5 = l5[2]; p_n5 = l5[3]; }
                const double a = L(xa), b = L(xb);
                if (TL_P2_LEVEL >= 6) { L(r1)[it] = a; L(p1)[it] = b; continue; }   // diagnostic: the transform alone, kept alive through the state
                // The step in FIVE stretches with scheduling fences between them (tl_libm.h: tlm_atan2_head / _mid / _tail, tlm_sincos_reduce /
                // _poly / _finish).  Every routine here ends in a table row -- the arctangent's in L1 / L2, the two sincos rows in LDS -- and the
                // compiler, left alone, requests a row a handful of instructions before it waits for it: under the round-5 kernel a wave
                // spent 53 % of this stage's cycles in s_waitcnt (profiles/class_budget_r06_psy2.txt), and three such waves leave a SIMD idle
                // one cycle in six.  So: (1) the arctangent's first quotient and the PREDICTED phase's reduction, both rows requested;
                // (2) the second quotient, form A, both polynomial sets of the predicted phase -- ~75 operations no row enters;
                // (3) form B -> this pass's phase, the predicted phase's sine / cosine, the phase's own reduction, its row requested;
                // (4) its polynomial sets and the square root of the energy; (5) its sine / cosine and the unpredictability measure.
                // (TL_P2_SUB, diagnostic builds only: 1 = no sincos of the predicted phase, 2 = no sincos at all, 3 = no arctangent, 4 = no square roots / division)
                constexpr bool full = !SEED && TL_P2_LEVEL < 5, sc2 = full && TL_P2_SUB != 1 && TL_P2_SUB != 2, sc1 = full && TL_P2_SUB != 2;
                TlmAtanA at = {}; TlmAtanB am = {};
                double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
                if (TL_P2_SUB != 3) {
                    at = tlm_atan2_head(-a, b);
                    const uint64_t *arow = tlm_atan_cij + 7 * at.i;
                    c0 = tl_u2d(arow[0]); c1 = tl_u2d(arow[1]); c2 = tl_u2d(arow[2]); c3 = tl_u2d(arow[3]); c4 = tl_u2d(arow[4]); c5 = tl_u2d(arow[5]); c6 = tl_u2d(arow[6]);
                }
                double e = (a * a + b * b) / 2.0;
                TlmSinCosA s2 = {}; TlmSinCosB v2 = {};
                double q_sn = 0, q_ssn = 0, q_cs = 0, q_ccs = 0, r_prime = 0;
                if (sc2) {
                    s2 = tlm_sincos_reduce(2.0 * L(p1)[it] - L(p2)[it]);
                    q_sn = tl_u2d(TL_SCT(sct, s2.row, 0)); q_ssn = tl_u2d(TL_SCT(sct, s2.row, 1)); q_cs = tl_u2d(TL_SCT(sct, s2.row, 2)); q_ccs = tl_u2d(TL_SCT(sct, s2.row, 3));
                }
                TLM_SCHED_FENCE();
                if (TL_P2_SUB != 3) am = tlm_atan2_mid(at);
                if (sc2) v2 = tlm_sincos_poly(s2);
                if (full) r_prime = 2.0 * L(r1)[it] - L(r2)[it];
                TLM_SCHED_FENCE();
                double phi = (TL_P2_SUB == 3 ? b - a : tlm_atan2_tail(at, am, c0, c1, c2, c3, c4, c5, c6)) + 3.14159265358979 / 4;
                const bool low = e < 0.0005;
                e = TLM_MAX_NN(e, 0.0005); phi = tlm_sel(low, 0.0, phi);
                e = tlm_sel(first, a * a, e); phi = tlm_sel(first, 0.0, phi);      // line 0: energy x_real[0]^2, phase 0
                double spp = 0, cpp = 0, w_sn = 0, w_ssn = 0, w_cs = 0, w_ccs = 0;
                TlmSinCosA s1 = {}; TlmSinCosB v1 = {};
                // Steps 1..7 need no sine and cosine by NAME: t1^2 + t2^2 below is symmetric in (cosines, sines), so where both phases' quadrants
                // are of one parity the two raw pairs are used as they come, and where they differ ONE pair is exchanged (the phase's own)
                if (sc2) { if (it == 0 || !sc1) tlm_sincos_finish(s2, v2, q_sn, q_ssn, q_cs, q_ccs, &spp, &cpp); else tlm_sincos_finish_raw(s2, v2, q_sn, q_ssn, q_cs, q_ccs, &spp, &cpp); }
                else if (full) { spp = L(p1)[it] + L(p2)[it]; cpp = r_prime; }
                if (sc1) {
                    s1 = tlm_sincos_reduce(tlm_sel(first, 2.0 * p_o5 - p_n5, phi));
                    w_sn = tl_u2d(TL_SCT(sct, s1.row, 0)); w_ssn = tl_u2d(TL_SCT(sct, s1.row, 1)); w_cs = tl_u2d(TL_SCT(sct, s1.row, 2)); w_ccs = tl_u2d(TL_SCT(sct, s1.row, 3));
                }
                TLM_SCHED_FENCE();
                const double rn = TL_P2_SUB == 4 ? e + 1.0 : it == 0 ? tlm_sqrt_ns(e) : tlm_sqrt_nz(e);   // e >= 0.0005, or line 0's x^2 (step 0; zero included): far from the exponent limits
                if (sc1) v1 = tlm_sincos_poly(s1);
                TLM_SCHED_FENCE();
                double spp5 = 0, cpp5 = 0;
                if (full) {
                    double sp = phi, cp = rn;
                    if (sc1 && (it == 0 || !sc2)) tlm_sincos_finish(s1, v1, w_sn, w_ssn, w_cs, w_ccs, &sp, &cp);
                    else if (sc1) {
                        double ds, dc;
                        tlm_sincos_finish_raw(s1, v1, w_sn, w_ssn, w_cs, w_ccs, &ds, &dc);
                        const bool sw = ((s1.q ^ s2.q) & 1u) != 0;
                        sp = tlm_sel(sw, dc, ds); cp = tlm_sel(sw, ds, dc);
                    }
                    spp5 = sp; cpp5 = cp;                                // sincos of line 512's predicted phase (lane 0 of step 0)
                    sp = tlm_sel(first, 0.0, sp); cp = tlm_sel(first, 1.0, cp);         // sincos(0.0)
                    const double t1 = rn * cp - r_prime * cpp;
                    const double t2 = rn * sp - r_prime * spp;
                    const double t3 = rn + fabs(r_prime);
                    // (t3 >= sqrt(0.0005) but for line 0; t1^2 + t2^2 is zero or above 1e-70: the unscaled square root and division, tl_libm.h)
                    // (stored as the partition sums' TERM energy x c, psycho_2.c:153: one multiplication here with 64 lanes at work instead of
                    // one per line in the sums, where a handful of lanes walk the wide partitions)
                    // (steps 1..7: t3 >= sqrt(0.0005), no zero divisor to select around)
                    cw[j] = e * (TL_P2_SUB == 4 ? t1 * t1 + t2 * t2 + t3 : it == 0 ? tlm_sel(t3 != 0, tlm_div_ns(tlm_sqrt_ns(t1 * t1 + t2 * t2), t3), 0.0)
                                                                               : tlm_div_ns(tlm_sqrt_ns(t1 * t1 + t2 * t2), t3));
                    x[j] = e;
                }
                L(r2)[it] = L(r1)[it]; L(r1)[it] = rn; L(p2)[it] = L(p1)[it]; L(p1)[it] = phi;
                if (it == 0) {                                       // line 512 (psycho_2.c:110-140 with fft.c:1274's phase), finished by lane 0
                    const double c5 = L(xc);
                    const double e5 = c5 * c5;
                    const bool neg5 = (tl_d2u(c5) >> 63) != 0;       // atan2(+0.0, x) = pi for x < 0 and x = -0, else +0
                    const double phi5 = neg5 ? tl_u2d(0x400921fb54442d18ull) : 0.0;
                    const double rn5 = tlm_sqrt_ns(e5);
                    double c512 = 0;
                    if (!SEED) {
                        const double sp5 = neg5 ? tl_u2d(0x3ca1a62633145c07ull) : 0.0, cp5 = neg5 ? -1.0 : 1.0;   // glibc's sincos of that pi / of 0
                        const double r_prime5 = 2.0 * r_o5 - r_n5;
                        const double t15 = rn5 * cp5 - r_prime5 * cpp5;
                        const double t25 = rn5 * sp5 - r_prime5 * spp5;
                        const double t35 = rn5 + fabs(r_prime5);
                        c512 = tlm_sel(t35 != 0, tlm_div_ns(tlm_sqrt_ns(t15 * t15 + t25 * t25), t35), 0.0);
                    }
                    if (first) { l5[0] = rn5; l5[1] = r_o5; l5[2] = phi5; l5[3] = p_o5; if (!SEED) cw[512] = e5 * c512; }
                    L(e512) = e5;                                    // slot 512 of the transform buffer still holds a point step 7 reads
                }
            }
            TL_LANES_END
        }
        if (SEED || TL_P2_LEVEL >= 5) return;
        TL_LANES_BEGIN
        if (lane == 0) x[512] = L(e512);
        TL_LANES_END
        TL_STAMP(sq, 2);
        if (TL_P2_LEVEL >= 4) {                                          // diagnostic: c[] and the energies kept alive through the record
            TL_LANES_BEGIN
            if (lane < 32) { if (pass == 0) L(snr0) = cw[lane] + x[lane + 64]; else smr_out[lane] = L(snr0) + cw[lane + 128] + x[lane + 192]; }
            TL_LANES_END
            return;
        }
        const double *energy = x;
        // the lane's first sixteen spreading coefficients (two batches of TL_P2_B), requested here, used after the partition sums
        PA(double, sva, TL_P2_B); PA(double, svb, TL_P2_B);
        TL_LANES_BEGIN
        {
            const double *sb = &P->s_band[0][0];
            TL_LAUNDER(sb);                                              // (loads through it stay behind this point)
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < TL_P2_B; q++) { L(sva)[q] = sb[64 * q + lane]; L(svb)[q] = sb[64 * (TL_P2_B + q) + lane]; }
        }
        TL_LANES_END
        // grouped energy / weighted unpredictability per partition (psycho_2.c:146-155): two chains of additions per partition, each in
        // line order.  The widest partitions set the stage's length (67 and 77 lines, most others under 10), so the TWO chains of a
        // partition run on two lanes -- lanes 0..31 add the energies, lanes 32..63 the terms energy x c the line loop has left in c[] --
        // in two rounds of 32 partitions: the upper 32 first (all the wide ones), then the rest.  Four lines per LDS round trip, the next
        // four requested before these are added, the batch loop unrolled by two (no register rotation).
        TL_LANES_BEGIN
        {
            const int l = lane & 31, half = lane >> 5, np = P->npart;
            const double *arr = half ? cw : energy;
            const int p0 = np - 32 + l, p1 = l;
            const bool ok0 = p0 >= 0, ok1 = p1 < np - 32;
            const int lo0 = P->part_lo[ok0 ? p0 : 0], hi0 = P->part_hi[ok0 ? p0 : 0], lo1 = P->part_lo[ok1 ? p1 : 0], hi1 = P->part_hi[ok1 ? p1 : 0];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int r = 0; r < 2; r++) {
                const bool ok = r ? ok1 : ok0;
                const int hi = r ? hi1 : hi0;
                int j = r ? lo1 : lo0;
                double acc = 0;
                if (ok) {
                    double ev[4];
                    if (j + 4 <= hi) {
#ifndef TL_EMULATE
#pragma unroll
#endif
                        for (int q = 0; q < 4; q++) ev[q] = arr[j + q];
#ifndef TL_EMULATE
#pragma unroll 2
#endif
                        for (; j + 8 <= hi; j += 4) {
                            double en[4];
#ifndef TL_EMULATE
#pragma unroll
#endif
                            for (int q = 0; q < 4; q++) en[q] = arr[j + 4 + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
                            for (int q = 0; q < 4; q++) acc += ev[q];
#ifndef TL_EMULATE
#pragma unroll
#endif
                            for (int q = 0; q < 4; q++) ev[q] = en[q];
                        }
#ifndef TL_EMULATE
#pragma unroll
#endif
                        for (int q = 0; q < 4; q++) acc += ev[q];
                        j += 4;
                    }
                    for (; j < hi; j++) acc += arr[j];
                    ge[2 * (r ? p1 : p0) + half] = acc;
                }
            }
            if (np < 64 && lane >= np) { ge[2 * lane] = 0; ge[2 * lane + 1] = 0; }    // (the spreading window may reach past the last partition: zeros, as before)
        }
        TL_LANES_END
        TL_STAMP(sq, 3);
        if (TL_P2_LEVEL >= 3) {
            TL_LANES_BEGIN
            if (lane < 32) { if (pass == 0) L(snr0) = ge[lane] + ge[lane + 64]; else smr_out[lane] = L(snr0) + ge[lane + 32] + ge[lane + 96]; }
            TL_LANES_END
            return;
        }
        // spreading (psycho_2.c:161-175), required SNR (:181-193), permissible noise (:200-204)
        // Row `lane` of the spreading function is zero outside a band of at most TL_P2_BAND partitions (TlPsy2Tables::s_band): the
        // sums run over the band's window in ascending order -- the reference's order with its zero coefficients left out, which is
        // what the reference does itself (:165).  Batches of TL_P2_B coefficients, two batches ahead of their use: the first two have been
        // under way since before the partition sums, batch i + 2 is requested before batch i is summed (TL_TIE pins that order: the
        // compiler otherwise requests all of them at once and spills the run's r / phi state).  A table whose band is narrower
        // (model 4: 29) stops early.
        TL_LANES_BEGIN
        {
            double e = 0, c = 0;
            const double *gel = ge + 2 * P->band_lo[lane];
            const int nbat = (P->band_w + TL_P2_B - 1) / TL_P2_B;         // (uniform)
            double svc[TL_P2_B];
#define TL_P2_LOAD(dst, i) do { if ((i) < nbat) { const double *sb_ = &P->s_band[(i) * TL_P2_B][0]; TL_TIE(sb_, e); \
                                 _Pragma("unroll") for (int q = 0; q < TL_P2_B; q++) dst[q] = sb_[64 * q + lane]; } } while (0)
#define TL_P2_SUM(src, i) do { if ((i) < nbat) { _Pragma("unroll") for (int q = 0; q < TL_P2_B; q++) { double ge_, gc_; TL_LD2(gel + 2 * ((i) * TL_P2_B + q), ge_, gc_); e += src[q] * ge_; c += src[q] * gc_; } } } while (0)
            TL_P2_LOAD(svc, 2); TL_P2_SUM(L(sva), 0);
            TL_P2_LOAD(L(sva), 3); TL_P2_SUM(L(svb), 1);
            TL_P2_LOAD(L(svb), 4); TL_P2_SUM(svc, 2);
            TL_P2_LOAD(svc, 5); TL_P2_SUM(L(sva), 3);
            TL_P2_SUM(L(svb), 4);
            TL_P2_SUM(svc, 5);
#undef TL_P2_LOAD
#undef TL_P2_SUM
            static_assert(TL_P2_BAND == 6 * TL_P2_B, "six batches");
            double cb = e != 0 ? c / e : 0;
            if (cb < .05) cb = 0.05; else if (cb > .5) cb = 0.5;
            const double tb = -0.434294482 * tlm_log_pn(cb, tlm_log_tab) - 0.301029996;
            double bc = P->tmn[lane] * tb + 5.5 * (1.0 - tb);
            bc = bc > P->bmaxk[lane] ? bc : P->bmaxk[lane];
            bc = tlm_exp_sl<false>(-bc * 0.2302585093, 0.0);
            ecb[lane] = e;
            nb[lane] = P->den[lane] != 0 ? e * bc / P->den[lane] : 0;
        }
        TL_LANES_END
        TL_STAMP(sq, 4);
        if (TL_P2_LEVEL >= 2) {
            TL_LANES_BEGIN
            if (lane < 32) { if (pass == 0) L(snr0) = nb[lane] + ecb[lane + 32]; else smr_out[lane] = L(snr0) + nb[lane + 32] + ecb[lane]; }
            TL_LANES_END
            return;
        }
        // threshold per line (psycho_2.c:205-224): c[] is dead, reuse it for fthr[]
        // (all nine lines of a lane at once: as a rolled loop every trip was a load of the partition number, then a dependent LDS read of its
        // permissible noise, then the store -- nine memory round trips in a row for nine maxima; round 6: the table reads of all trips are
        // requested first, then the LDS reads, then the stores -- one round trip of each kind)
        // Both arrays the subband stage walks are laid out PADDED here -- line j at index j + (j >> 5) = lane + (lane >> 5) + 66 q: a subband's
        // lane starts 16 lines after its neighbour's, 128 bytes, one LDS bank pair for all 32 of them; with one slot of padding per 32 lines
        // the 32 starts fall on 32 different bank pairs.  The thresholds are written that way; the energies are moved in place (all nine
        // reads of every lane before the first write; the partition sums above them are dead).
        PA(double, ej, 9);
        TL_LANES_BEGIN
        {
            int pj[9]; double aj[9], tj[9];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 9; q++) { const int j = lane + 64 * q < 512 ? lane + 64 * q : 512; pj[q] = P->partition[j]; aj[q] = P->absthr[j]; }
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 9; q++) { tj[q] = nb[pj[q]]; L(ej)[q] = energy[lane + 64 * q < 512 ? lane + 64 * q : 512]; }
            const int jp = lane + (lane >> 5);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 9; q++) { if (lane + 64 * q <= 512) cw[jp + 66 * q] = TLM_MAX_NN(aj[q], tj[q]); }      // (= tj > aj ? tj : aj: no NaN among permissible noise and threshold in quiet)
        }
        TL_LANES_END
        TL_LANES_BEGIN
        {
            const int jp = lane + (lane >> 5);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 9; q++) { if (lane + 64 * q <= 512) x[jp + 66 * q] = L(ej)[q]; }
        }
        TL_LANES_END
        TL_STAMP(sq, 5);
        if (TL_P2_LEVEL >= 1) {
            TL_LANES_BEGIN
            if (lane < 32) { if (pass == 0) L(snr0) = cw[lane] + cw[lane + 256]; else smr_out[lane] = L(snr0) + cw[lane + 64] + cw[lane + 448]; }
            TL_LANES_END
            return;
        }
        // 32 subbands (psycho_2.c:227-246): lanes 0..31 walk a subband's 17 thresholds (their minimum below subband 13, their sum above),
        // lanes 32..63 the same subband's 17 energies, both in line order from the padded arrays; the energy sums cross over through LDS
        PV(double, sb_min); PV(double, sb_sum);
        TL_LANES_BEGIN
        {
            const int sb = lane & 31;
            const double *a = (lane < 32 ? cw : x) + 16 * sb + (sb >> 1);
            double m = 60802371420160.0, t = 0.0;
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 16; k++) { const double v = a[k]; m = TLM_MIN_NN(m, v); t += v; }
            { const double v = a[16 + (sb & 1)]; m = TLM_MIN_NN(m, v); t += v; }
            L(sb_min) = m; L(sb_sum) = t;
            if (lane >= 32) nb[sb] = t;
        }
        TL_LANES_END
        TL_LANES_BEGIN
        if (lane < 32) {
            const double sum_energy = nb[lane];
            double snr = lane < 13 ? sum_energy / (L(sb_min) * 17.0) : sum_energy / L(sb_sum);
            snr = 4.342944819 * tlm_log_pn(snr, tlm_log_tab);
            if (pass == 0) L(snr0) = snr;
            else smr_out[lane] = L(snr0) > snr ? L(snr0) : snr;
        }
        TL_LANES_END
        TL_STAMP(sq, 6);
    }
}
