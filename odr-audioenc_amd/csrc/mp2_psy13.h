// mp2_psy13.h -- psy models 1 and 3 (psycho_1.c, psycho_3.c): tone labelling, noise bands, dB-sum chains, decimation, thresholds, SMR record.
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ------------------------------------------------------------------------------------------
// Candidate record used by the tone labelling of psy 1 and psy 3: bits 0-8 line index, bits 21.. the line's run.  Only candidates
// whose RIGHT neighbours pass are recorded: the right side of a candidate is never touched by an earlier tone (the erasure reach
// of every earlier tone ends below the candidate), so that half of the test is decided in parallel from the original spectrum,
// for all 500 lines.  The left half depends on which earlier candidates were confirmed: the walk that follows reads the left
// neighbours of the CANDIDATES (tl_cand_left: "neighbour j fails" as bit j - 2) and resolves them against its state.

// power density in dB of one line (psycho_1.c:241-248, psycho_3.c:152-160), straight-line so that several lines' logarithms
// (long dependent chains) can be in flight together
// The logarithm is glibc 2.35's own (tl_libm.h: table-driven log, no division, then e_log10.c's recombination) -- bit-equal to
// the reference's libm by construction.  That matters: on degenerate spectra (a lone impulse: hundreds of lines of nearly
// equal level) the tone tests and the allocation compare values that differ in the last bits (GPU soak, round 2).
#define TL_LOGTAB(db) ((const uint64_t *)((db) + 1002))     /* the log table rides behind the dB-sum table in the workgroup's LDS block (TlTables::dblog) */
// The floor (`energy < 1E-20 ? -200 + POWERNORM : ...`, psycho_1.c:243-246) is one maximum: log10 of the double 1E-20 is exactly -20.0
// in glibc and in its restatement here (tests/test_libm_agree.py pins it), so 10 * log10(max(e, 1E-20)) + POWERNORM is the reference's
// value on either side of the test -- (-200.0 + POWERNORM) is the same sum -- and the argument of the logarithm is always normal.
TL_FN double tl_power_db(double e, const uint64_t *lt)
{
    return 10 * tlm_log10_pn(__builtin_fmax(e, 1E-20), lt) + TL_POWERNORM;
}
// The 512 logarithms of a channel in two halves (tl_libm.h: tlm_log10_main_pn / tlm_log10_near1_pn).  About one argument in eleven takes
// e_log.c's "close to 1.0" branch (26 operations the other ten have no use for -- and every wave of 64 has some): tl_power_db_main runs
// the table branch and says which lanes it does not serve, their LINE NUMBERS are filed in a list (tl_power_defer: the candidate array,
// not in use yet, as 512 16-bit slots), and tl_power_near1 puts up to 64 filed lines through the other branch at once -- energy re-read,
// level overwritten.  One or two such passes per channel instead of eight.
TL_FN double tl_power_db_main(double e, const uint64_t *lt, bool *near1)
{
    return 10 * tlm_log10_main_pn(__builtin_fmax(e, 1E-20), lt, near1) + TL_POWERNORM;
}
#define TL_DEFER(w) ((uint16_t *)(w).cinfo)
TL_FN void tl_power_near1(TlPsyLds &w, int pos0, int cnt)
{
    TL_LANES_BEGIN
    if (lane < cnt) {
        const int i = TL_DEFER(w)[pos0 + lane];
        const double e = w.u.fft[TL_EX(i)];
        TL_PX(w)[i] = 10 * tlm_log10_near1_pn(__builtin_fmax(e, 1E-20)) + TL_POWERNORM;
    }
    TL_LANES_END
}
// lanes with `flag` append `line` to the list (lane order), ndef grows
#define TL_POWER_DEFER(w, flag, line_expr, ndef) do { \
    const uint64_t m_ = TL_BALLOT(flag); \
    TL_LANES_BEGIN if (L(flag)) TL_DEFER(w)[(ndef) + __builtin_popcountll(m_ & ((1ull << lane) - 1ull))] = (uint16_t)(line_expr); TL_LANES_END \
    (ndef) += __builtin_popcountll(m_); } while (0)
TL_FN int tl_run_psy1(int c) { return (c < 3 || c > 500) ? 0 : c < 63 ? 2 : c < 127 ? 3 : c < 255 ? 6 : 12; }   // psycho_1.c:289-298
TL_FN int tl_run_psy3(int c) { return c < 63 ? 2 : c < 127 ? 3 : c < 255 ? 6 : 12; }                            // psycho_3.c:212-215

// Tone candidates of one 64-line chunk: local maxima 2..499 whose right-hand neighbours (distance 2..run) pass the
// 7 dB test; the left-hand failures are recorded as a bit mask for the walk that follows (psycho_1.c:267-300,
// psycho_3.c:186-236).  RMAX is the largest run inside the chunk, so the neighbour reads are straight-line code
// and overlap; PSY3 selects psycho_3's strict maximum and its (peak - neighbour) < 7 form of the test.
template <int RMAX, bool PSY3>
TL_FN void tl_cand_chunk(TlPsyLds &w, int c8, int &ncand)
{
    const double *px = TL_PX(w);
    PV(bool, isc); PV(uint32_t, rec);
    TL_LANES_BEGIN
    const int i = 64 * c8 + lane - 1;                               // chunks start one line early: the run lengths change at 63, 127, 255
    const bool inr = i >= 2 && i < 500;
    const int ii = inr ? i : 16;
    // every neighbour is read before the first test (TL_KEEP: otherwise the compiler reads each one only if the
    // tests so far passed -- a chain of dependent LDS round trips)
    double a[RMAX + 1];
    const double pk = px[ii], b1 = px[ii - 1];
#pragma unroll
    for (int j = 1; j <= RMAX; j++) a[j] = px[ii + j];
#pragma unroll
    for (int j = 1; j <= RMAX; j++) TL_KEEP(a[j]);
    bool cnd = inr && pk > b1 && (PSY3 ? pk > a[1] : pk >= a[1]);
    // the run of a line inside the chunk is the chunk's RMAX (the chunks are cut where the run changes), or 0 (psycho_1's lines
    // below 3): one per-lane flag instead of a `j <= run` per neighbour
    const int run = PSY3 ? tl_run_psy3(ii) : tl_run_psy1(ii);
    const bool has = run != 0;
    const double max = pk - 7;
    bool fail = false;
#pragma unroll
    for (int j = 2; j <= RMAX; j++) fail = fail || (PSY3 ? (pk - a[j]) < 7.0 : max < a[j]);
    cnd = cnd && !(has && fail);
    // The LEFT-hand neighbours are not looked at here: what they decide depends on the walk, and the walk looks at them for the
    // candidates alone (tl_cand_left) -- a few dozen lines instead of five hundred.
    L(isc) = cnd; L(rec) = (uint32_t)i | ((uint32_t)run << 21);     // line | run << 21
    TL_LANES_END
    const uint64_t m = TL_BALLOT(isc);
    TL_LANES_BEGIN
    if ((m >> lane) & 1ull) w.cinfo[ncand + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = L(rec);
    TL_LANES_END
    ncand += __builtin_popcountll(m);
}
// "left neighbour j fails the 7 dB test" for j = 2..run of candidate line c with level pk, bit j - 2 (psycho_1.c:289-300,
// psycho_3.c:217-226), from the still-original spectrum.  All eleven neighbours are read whatever the run is (px[] sits behind
// the transform buffer: c - 12 is inside the wave's block for every c >= 2, and what lies there are finite energies); the bits
// beyond the run are masked off.
template <bool PSY3>
TL_FN uint32_t tl_cand_left(const double *px, int c, int run, double pk)
{
    double b[11];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int j = 2; j <= 12; j++) b[j - 2] = px[c - j];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int j = 2; j <= 12; j++) TL_KEEP(b[j - 2]);
    const double max = pk - 7;
    uint32_t lf = 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int j = 2; j <= 12; j++) lf |= (PSY3 ? (pk - b[j - 2]) < 7.0 : max < b[j - 2]) ? 1u << (j - 2) : 0u;
    return lf & (run >= 2 ? (1u << (run - 1)) - 1u : 0u);
}

// psy model 1 (psycho_1.c:22-87, :215-581); result in w.smr[ch][0..sblimit).
//
// Per channel the model is a FRONT (spectrum, power, tone labelling, compaction of the lines each critical band sums),
// the per-band dB-sum CHAINS (sequential by definition: up to 164 dependent table look-ups in the widest band, on 27
// lanes) and a BACK (band centres, decimation, thresholds, SMR).  For two channels the chains of both run side by side
// on the two halves of the wave (tl_psy1_stereo): channel 0's front results wait in registers while channel 1's front
// uses the LDS arrays.
struct TlPsy1Ch { int nconf, nlist; bool dead_head; };

TL_FN TlPsy1Ch tl_psy1_front(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                             const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const double *energy = w.u.fft;                                   // line i at TL_EX(i)
    double *px = TL_PX(w);
    TL_PRIO2(TL_PS_FHT);
    TL_STAMP(sp, 0);
    if (TL_EXP_LEVEL < 8) tl_psy_spectrum(w, T, pv, ch, sp);
    TL_STAMP(sp, 1);
    TL_PRIO2(TL_PS_POW);

    // power density spectrum (psycho_1.c:241-248); spike (psycho_1.c:252-257)
    // The spike sums read 16 consecutive energies per lane; the energies' XOR layout keeps those reads off each other's
    // LDS banks.  Only subbands below sblimit (<= 30) are ever used.
    int ndef = 0;                                                     // lines filed for the logarithm's other branch (tl_power_near1)
    for (int h = 0; h < (TL_EXP_LEVEL >= 7 ? 0 : 2); h++) {           // four lines per lane at a time
        PV(bool, n0); PV(bool, n1); PV(bool, n2); PV(bool, n3);
        TL_LANES_BEGIN
        {
            const int i0 = lane + 256 * h;
            double e[4], v[4];
            bool nr[4];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 4; q++) e[q] = energy[TL_EX(i0 + 64 * q)];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 4; q++) v[q] = tl_power_db_main(e[q], TL_LOGTAB(db), &nr[q]);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 4; q++) {
                const int i = i0 + 64 * q;
                px[i] = v[q];                                           // (a filed line's value is overwritten by tl_power_near1)
                w.ptype[i] = 0;
            }
            L(n0) = nr[0]; L(n1) = nr[1]; L(n2) = nr[2]; L(n3) = nr[3];
        }
        TL_LANES_END
        TL_POWER_DEFER(w, n0, lane + 256 * h, ndef);
        TL_POWER_DEFER(w, n1, lane + 256 * h + 64, ndef);
        TL_POWER_DEFER(w, n2, lane + 256 * h + 128, ndef);
        TL_POWER_DEFER(w, n3, lane + 256 * h + 192, ndef);
        while (ndef >= 64) { ndef -= 64; TL_DBG_NEAR1_FULL(); tl_power_near1(w, ndef, 64); }
    }
    if (ndef) tl_power_near1(w, 0, ndef);
    TL_LANES_BEGIN
    if (lane < (TL_EXP_LEVEL >= 7 ? 0 : 30)) {
        double e[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int j = 0; j < 16; j++) e[j] = energy[16 * lane + (j ^ (lane & 15))];      // == energy[TL_EX(16 * lane + j)]
        // 1E-20 + sum of 2^30 e[j] in line order (psycho_1.c:252-257) as 2^30 times (1E-20 / 2^30 + sum of e[j]): scaling every operand of a
        // chain of additions by a power of two scales every partial sum exactly (no operand is anywhere near the subnormal range)
        double sum = 1E-20 * (1.0 / 1073741824);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int j = 0; j < 16; j++) sum += e[j];
        const double spk = 10.0 * tlm_log10_pn(1073741824 * sum, TL_LOGTAB(db));
        L(rec)[ch] = spk;                                           // final as it is: straight to the record (nothing to park)
    } else if (lane < 32) L(rec)[ch] = 0.0;
    TL_LANES_END
    TL_PRIO2(1);                                                      // from here to the thresholds: serial stages

    // ---- tonal components (psycho_1.c:267-340) ----
    // (1) local maxima 2..499 whose right-hand neighbours pass the 7 dB test, compacted ascending
    int ncand = 0;
    if (TL_EXP_LEVEL < 6) {
    tl_cand_chunk<2, false>(w, 0, ncand);                           // lines -1..62: run 2
    tl_cand_chunk<3, false>(w, 1, ncand);                           // 63..126: run 3
    for (int c8 = 2; c8 < 4; c8++) tl_cand_chunk<6, false>(w, c8, ncand);     // 127..254: run 6
    for (int c8 = 4; c8 < 8; c8++) tl_cand_chunk<12, false>(w, c8, ncand);    // 255..510: run 12
    }
    TL_STAMP(sp, 2);
    // (2) which candidates become tones.  The reference walks its list once, in line order; what a candidate's fate depends on is the
    //     walk's state -- the last confirmed tone (`last`), its run and whether its left neighbour was erased -- and that state only
    //     changes at a CONFIRMATION.  So instead of one scalar iteration per candidate (sixty dependent scalar instructions each: the
    //     walk was 9 % of a frame's cycles with every lane idle) the candidates sit in lanes and every round evaluates all of them
    //     against the current state at once: the first one that passes is the next confirmed tone -- every candidate before it was
    //     rejected under the same state, exactly as the sequential walk rejects them -- the state moves, the lanes after it go on.
    //     One round per confirmed tone (+ 1) instead of one iteration per candidate.  The bookkeeping of the reference's list is
    //     kept: the erasure reach R = last + run(last), the last_but_one relinking (psycho_1.c:313-316).
    int nconf = 0;
    bool any_erased;                                                  // a confirmed tone erased its predecessor: the chain order is not 0..nconf-1
    {
        int last = -1, run_last = 0, last_var = 0;
        any_erased = false;
        for (int kb = 0; kb < (TL_EXP_LEVEL >= 5 ? 0 : ncand); kb += 64) {          // 64 candidates per pass (there are rarely more)
        PV(int, cc); PV(int, crun); PV(int, clf); PV(double, cpx); PV(bool, act); PV(bool, dep);
        TL_LANES_BEGIN
        const bool in = kb + lane < ncand;
        const uint32_t info = w.cinfo[in ? kb + lane : 0];
        const uint32_t pinfo = w.cinfo[in && kb + lane > 0 ? kb + lane - 1 : 0];    // the candidate before this one
        L(cc) = (int)(info & 511u); L(crun) = (int)(info >> 21);
        L(cpx) = px[L(cc)];
#if defined(TL_WALK_SKIP) && (TL_WALK_SKIP & 1)
        L(clf) = 0;                                                   // diagnostic build: what the left-neighbour tests cost
#else
        L(clf) = (int)tl_cand_left<false>(px, L(cc), L(crun), L(cpx));
#endif
        L(act) = in;
        // A candidate at least run(previous candidate) + run(its own) + 1 lines above the candidate before it is out of every earlier
        // tone's reach whatever the walk's state is when it gets there: `last` is at or below that previous candidate and runs grow
        // with the line, so d - run_last - 1 >= run, every one of its left neighbours is original, and its fate is lfail == 0 -- the
        // value the test below gives it under ANY earlier state (d > run, so neither the summed-level test nor an erasure applies,
        // and its left neighbour is not the end of a reach: var = 0).  Such candidates need no round of their own.
        L(dep) = in && kb + lane > 0 && L(cc) - (int)(pinfo & 511u) < (int)(pinfo >> 21) + L(crun) + 1;
        TL_LANES_END
        for (;;) {
            PV(bool, okv); PV(bool, needx);
            TL_LANES_BEGIN
            const int c = L(cc), run = L(crun);
            const uint32_t lfail = (uint32_t)L(clf);
            bool ok, nx = false;
            if (last < 0) ok = lfail == 0;
            else {
                // neighbours c-j <= R were erased to DBMIN by `last` (they pass), except `last` itself,
                // which carries its summed level; neighbours above R (or below last-run_last) are original
                // bit j-2 set for j in [2, run] with c-j > R (j <= c-R-1) or c-j < last-run_last (j >= c-last+run_last+1)
                const int d = c - last;
                const int hi_j = run < d - run_last - 1 ? run : d - run_last - 1;
                const int lo_j = d + run_last + 1 > 2 ? d + run_last + 1 : 2;
                uint32_t orig = hi_j >= 2 ? (1u << (hi_j - 1)) - 1u : 0u;
                if (lo_j <= run) orig |= ((1u << (run - 1)) - 1u) & ~((1u << (lo_j - 2)) - 1u);
                ok = d > run_last && !(lfail & orig);                // d <= run_last: unlinked by the help loop, psycho_1.c:309-312
                nx = ok && d >= 2 && d <= run;                       // then `last` itself is among its neighbours, with its summed level
            }
            L(okv) = ok && L(act); L(needx) = nx && L(act);
            TL_LANES_END
            if (TL_BALLOT(needx) != 0ull) {                           // rare: a candidate within its run of the last tone
                TL_LANES_BEGIN
                const double xl = tl_add_db(db, px[last], tl_add_db(db, last_var ? TL_DBMIN : px[last - 1], px[last + 1]));
                if (L(needx) && L(cpx) - 7 < xl) L(okv) = false;
                TL_LANES_END
            }
            const uint64_t m = TL_BALLOT(okv);
            if (m == 0ull) break;                                     // everything left of this pass is rejected (only unlinked, psycho_1.c:330-338)
            const int wl = __builtin_ctzll(m);
            const int c = TL_READLANE_I32(cc, wl), run = TL_READLANE_I32(crun, wl);
            // confirmed.  Its left neighbour c-1 was erased iff it is exactly the end of `last`'s reach.
            const int var = (last >= 0 && run_last >= 1 && c - 1 == last + run_last) ? 1 : 0;
            // With it, in the same round: every passing candidate between it and the next state-DEPENDENT candidate still to be
            // decided (see `dep` above) -- their verdicts under the state of this round are their verdicts under any state.
            PV(bool, depact);
            TL_LANES_BEGIN L(depact) = L(dep) && L(act) && lane > wl; TL_LANES_END
            const uint64_t dm = TL_BALLOT(depact);
            const uint64_t upto = dm ? (1ull << __builtin_ctzll(dm)) - 1ull : ~0ull;     // lanes below the next dependent one
            const uint64_t batch = m & upto;                           // wl and the independent passing candidates after it
            const int nb = __builtin_popcountll(batch), wlast = 63 - __builtin_clzll(batch);
            const int i0 = nconf;
            TL_LANES_BEGIN
            if ((batch >> lane) & 1ull) {
                const int i = i0 + __builtin_popcountll(batch & ((1ull << lane) - 1ull));
                if (i < TL_TONE_MAX) {
                    w.conf_c[i] = (int16_t)(L(cc) | ((lane == wl ? var : 0) << 12));
                    w.conf_nxt[i] = (int16_t)((lane == wlast || i + 1 >= TL_TONE_MAX) ? TL_LAST : i + 1);
                }
            }
            TL_LANES_END
            if (i0 < TL_TONE_MAX && i0 > 0) {                         // the round's first tone against the tone before it
                if (c - last <= run) {                                // erases the previous tone, psycho_1.c:313-316,322-326
                    any_erased = true;
                    w.conf_nxt[i0 - 1] = TL_STOP;
                    w.conf_c[i0 - 1] = (int16_t)(w.conf_c[i0 - 1] | (1 << 13));
                    if (i0 >= 2) w.conf_nxt[i0 - 2] = (int16_t)i0;
                } else w.conf_nxt[i0 - 1] = (int16_t)i0;
            }
            nconf = i0 + nb < TL_TONE_MAX ? i0 + nb : (i0 < TL_TONE_MAX ? TL_TONE_MAX : i0);
            last = TL_READLANE_I32(cc, wlast); run_last = TL_READLANE_I32(crun, wlast); last_var = wlast == wl ? var : 0;
            TL_LANES_BEGIN L(act) = L(act) && lane > wlast; TL_LANES_END
            TL_DBG_ROUND();
        }
        }
        TL_SYNC();
    }
#if defined(TL_WALK_SKIP) && (TL_WALK_SKIP & 2)
    nconf = 0;                                                        // diagnostic build: what levels, erasures and the list cost
#endif
#if defined(TL_WALK_SKIP) && (TL_WALK_SKIP & 4)
    if (nconf > 1) nconf = 1;
#endif
    // (3) levels of the confirmed tones from the still-original spectrum (psycho_1.c:317-321)
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {
        const int cc = w.conf_c[i], c = cc & 511, var = (cc >> 12) & 1;
        w.tone_x[i] = tl_add_db(db, px[c], tl_add_db(db, var ? TL_DBMIN : px[c - 1], px[c + 1]));
    }
    TL_LANES_END
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) { const int c = w.conf_c[i] & 511; px[c] = w.tone_x[i]; w.ptype[c] = TL_T_TONE; }
    TL_LANES_END
    // (4) erasures (psycho_1.c:322-326); a tone erased by its successor ends up DBMIN / not TONE
    //     Straight-line: a tone's run is 2, 3, 6 or 12 (it is a line 3..499), so the stores are four nested groups behind three tests,
    //     each store at a constant offset from ONE address per array (px + c - 12 is inside the wave's block: px[] lies behind the
    //     transform) -- instead of a loop of `run` trips with four address computations each.
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {
        const int c = w.conf_c[i] & 511, run = tl_run_psy1(c);
        double *pb = px + (c - 12);
        uint8_t *tb = w.ptype;
#define TL_ERASE1(j) do { pb[12 - (j)] = TL_DBMIN; pb[12 + (j)] = TL_DBMIN; tb[c - (j)] = 0; tb[c + (j)] = 0; } while (0)
        if (run >= 2) {
            TL_ERASE1(1); TL_ERASE1(2);
            if (run >= 3) {
                TL_ERASE1(3);
                if (run >= 6) {
                    TL_ERASE1(4); TL_ERASE1(5); TL_ERASE1(6);
                    if (run >= 12) { TL_ERASE1(7); TL_ERASE1(8); TL_ERASE1(9); TL_ERASE1(10); TL_ERASE1(11); TL_ERASE1(12); }
                }
            }
        }
#undef TL_ERASE1
    }
    TL_LANES_END
    // (5) the tone list in chain order (psycho_1.c list head `*tone`): walk the links, then decimate
    //     in parallel (psycho_1.c:416-428): drop erased tones and tones below the threshold in quiet
    int nlist = 0;
    if (!any_erased) {                                                // every link points to the next tone: the chain is 0..nconf-1
        nlist = nconf;
        TL_LANES_BEGIN
        for (int i = lane; i < nconf; i += 64) w.tlist[i] = (int16_t)i;
        TL_LANES_END
    } else {
        PV(int, nx0); PV(int, nx1);                                   // the links in registers: the walk reads lanes, not LDS
        TL_LANES_BEGIN L(nx0) = w.conf_nxt[lane]; L(nx1) = w.conf_nxt[64 + lane < TL_TONE_MAX ? 64 + lane : 0]; TL_LANES_END
        int i = nconf ? 0 : TL_LAST, guard = 0;
        while (i != TL_LAST && i != TL_STOP && guard++ < TL_TONE_MAX) {
            w.tlist[nlist++] = (int16_t)i;
            i = i < 64 ? TL_READLANE_I32(nx0, i) : TL_READLANE_I32(nx1, i - 64);
        }
        TL_SYNC();
    }
    TL_STAMP(sp, 3);

    // ---- noise components (psycho_1.c:356-376) ----
    // Line-parallel preparation: the lines a band will actually sum (not tonal, not erased) are compacted in
    // ascending order together with their weight terms, so the sequential part is a bare dB-sum chain.
    // vt[] overwrites the energies and vp[] the power spectrum in place (a compacted position is never above its line, and a
    // chunk of 64 lines is read completely before its entries are written).  power[] is gone after this: the dead-head
    // replay rebuilds what it reads (tl_psy1_deadhead).
    const int nbands = C->p1_ncb - 1;
    {
        double *vt = w.u.fft, *vp = px;
        int nvalid = 0;
        PA(uint32_t, linfo, 8); PA(double, lrw, 8);                 // the table reads of all eight chunks in one batch
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int c8 = 0; c8 < 8; c8++) { L(linfo)[c8] = C->p1_lineinfo[64 * c8 + lane]; L(lrw)[c8] = C->p1_linerw[64 * c8 + lane]; }
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int base = 0; base < (TL_EXP_LEVEL >= 4 ? 0 : 512); base += 64) {
            PV(bool, ok); PV(double, tv); PV(double, pvv); PV(int, bnd);
            TL_LANES_BEGIN
            const int j = base + lane;
            const uint32_t info = L(linfo)[base >> 6];
            const int lo = (int)((info >> 8) & 0xfffu), hi = (int)(info >> 20);
            bool v = false; double t = 0, p = 0;
            if (info) {                                             // line inside the bands
                p = px[j];
                v = w.ptype[j] != TL_T_TONE && p != TL_DBMIN;
                t = tl_div_by(1073741824 * energy[TL_EX(j)] * (double)(j - lo), (double)(hi - lo), L(lrw)[base >> 6]);   // == num / (hi - lo)
            }
            L(ok) = v; L(tv) = t; L(pvv) = p; L(bnd) = (info && j == lo) ? (int)(info & 0xffu) : -1;
            TL_LANES_END
            const uint64_t m = TL_BALLOT(ok);
            TL_LANES_BEGIN
            const int pos = nvalid + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (L(bnd) >= 0) w.bandoff[L(bnd)] = (int16_t)pos;     // first line of its band
            if (L(ok)) { vt[pos] = L(tv); vp[pos] = L(pvv); }
            TL_LANES_END
            nvalid += __builtin_popcountll(m);
        }
        w.bandoff[nbands] = (int16_t)nvalid;
        TL_SYNC();
    }
    TlPsy1Ch r;
    r.nconf = nconf; r.nlist = nlist;
    // The reference keeps tones and noise components in ONE linked list field (power[].next).  The two chains only interact
    // when the head of the tone chain is a tone that was erased by its successor (psycho_1.c:313-316 with last_but_one ==
    // LAST): its line is no longer TONE, so a noise centre may land on it and splice the noise chain into the tone chain.
    // That (rare) case is replayed pointer by pointer (tl_psy1_deadhead); otherwise the chains are independent.
    r.dead_head = nconf > 0 && ((w.conf_c[0] >> 13) & 1);
    TL_DBG_TONES(nconf, r.dead_head); TL_DBG_CAND(ncand);
    return r;
}

// weight sums of the bands (psycho_1.c:364-366), ascending line order; lane b < nbands owns band b.  Only used where the
// weights cannot ride along with the dB-sum chain (channel 0 of a stereo frame, whose terms leave LDS before its chain runs).
TL_FN void tl_psy1_weights(TlPsyLds &w, int nbands, PARG(double, wt))
{
    TL_LANES_BEGIN
    double weight = 0.0;
    if (lane < nbands) {
        const double *vt = w.u.fft;
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 16 <= i1; i += 16) {                             // sixteen operands per LDS round trip, summed in order
            double t[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 16; q++) t[q] = vt[i + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 16; q++) weight += t[q];
        }
        for (; i < i1; i++) weight += vt[i];
    }
    L(wt) = weight;
    TL_LANES_END
}

// dB sums and weight sums of the bands of ONE channel (levels at TL_PX, weight terms at fft): lane b < nbands
TL_FN void tl_psy1_chain(TlPsyLds &w, const double *TL_RESTRICT db, int nbands, PARG(double, bsum), PARG(double, wt))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN, weight = 0.0;
    if (lane < nbands) {
        const double *vt = w.u.fft, *vp = TL_PX(w);
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 4 <= i1; i += 4) {                               // operands of four steps in flight per LDS round trip
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            const double t0 = vt[i], t1 = vt[i + 1], t2 = vt[i + 2], t3 = vt[i + 3];
            sum = tl_add_db(db, p0, sum); weight += t0;
            sum = tl_add_db(db, p1, sum); weight += t1;
            sum = tl_add_db(db, p2, sum); weight += t2;
            sum = tl_add_db(db, p3, sum); weight += t3;
        }
        for (; i < i1; i++) { sum = tl_add_db(db, vp[i], sum); weight += vt[i]; }
    }
    L(bsum) = sum; L(wt) = weight;
    TL_LANES_END
}

// dB sums of BOTH channels at once: lanes 0..31 walk channel 0's bands (levels parked at fft[], ranges in r0/r1), lanes 32..63
// channel 1's (levels at TL_PX, ranges from bandoff[]).  Result: lane b holds channel 0's sum, lane 32+b channel 1's.
TL_FN void tl_psy1_chain2(TlPsyLds &w, const double *TL_RESTRICT db, int nbands, PARG(int, r0), PARG(int, r1), PARG(double, bsum))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN;
    const int band = lane & 31;
    if (band < nbands) {
        const bool second = lane >= 32;
        const double *vp = second ? TL_PX(w) : w.u.fft;
        const int i0 = second ? (int)w.bandoff[band] : L(r0), i1 = second ? (int)w.bandoff[band + 1] : L(r1);
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                               // eight steps' operands per LDS round trip
            double p[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) p[q] = vp[i + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) sum = tl_add_db(db, p[q], sum);
        }
        for (; i + 4 <= i1; i += 4) {
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            sum = tl_add_db(db, p0, sum); sum = tl_add_db(db, p1, sum); sum = tl_add_db(db, p2, sum); sum = tl_add_db(db, p3, sum);
        }
        for (; i < i1; i++) sum = tl_add_db(db, vp[i], sum);
    }
    L(bsum) = sum;
    TL_LANES_END
}

// band centres (psycho_1.c:367-388) from the sums and weights of lanes b < nbands; needs ptype[] of the channel
// the band limits of lane b (the configuration record is in HBM / L2: requested BEFORE the chains run, used by tl_psy1_centres after them)
TL_FN void tl_psy1_limits(const TlConfig *TL_RESTRICT C, int nbands, PARG(int, blo), PARG(int, bhi))
{
    TL_LANES_BEGIN
    const int b = lane < nbands ? lane : 0;
    L(blo) = C->p1_cbound[b]; L(bhi) = C->p1_cbound[b + 1];
    TL_LANES_END
}
TL_FN void tl_psy1_centres(TlPsyLds &w, int nbands, PARG(double, bsum), PARG(double, wt), PARG(int, blo), PARG(int, bhi))
{
    TL_LANES_BEGIN
    if (lane < nbands) {
        const int lo = L(blo), hi = L(bhi);
        const double sum = L(bsum), weight = L(wt);
        int centre;
        if (sum <= TL_DBMIN) centre = (hi + lo) / 2;
        else {
            double index = weight * tlm_pow10_sl(-0.1 * sum);
            centre = lo + (int)(index * (double)(hi - lo));
        }
        centre = centre < 1 ? 1 : centre > 510 ? 510 : centre;     // out-of-range only on non-finite input (UB in the reference)
        if (w.ptype[centre] == TL_T_TONE) { if (w.ptype[centre + 1] == TL_T_TONE) centre++; else centre--; }
        w.nsum[lane] = sum; w.ncentre[lane] = (int16_t)centre;
    }
    TL_LANES_END
}

// individual + global masking thresholds, minimum per subband, SMR (psycho_1.c:480-581) from the masker lists
TL_FN void tl_psy1_thresholds(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, int ntone, int nnoise, PARGA(double, rec, 4), long long *sp)
{
    TL_STAMP(sp, 5);
    TL_PRIO2(TL_PS_THR);

    TL_DBG_DUMP("psy1", ch, ntone, nnoise, TL_MK_X(w), TL_MK_BARK(w));
    // ---- individual + global masking thresholds on the table lines (psycho_1.c:480-532) ----
    const int sub = C->p1_sub;
    // the lane's table rows (bark and threshold in quiet of its two lines: the configuration record in HBM / L2) are requested before
    // the masker records are built and the order test runs, so that their latency is not the first thing the walk waits for
    PV(double, tb0); PV(double, tb1); PV(double, th0); PV(double, th1); PV(int, mmn); PV(int, mmj);
    TL_LANES_BEGIN
    {
        const int k0 = 1 + 2 * lane, k1 = k0 + 1;
        const int q0 = k0 < sub ? k0 : 0, q1 = k1 < sub ? k1 : q0;
        L(tb0) = C->p1_bark[q0]; L(tb1) = C->p1_bark[q1]; L(th0) = C->p1_hear[q0]; L(th1) = C->p1_hear[q1];
        L(mmn) = C->p1_mm_n[lane & 31]; L(mmj) = C->p1_mm_j0[lane & 31];          // rows of the subband's minimum (used at the very end)
    }
    TL_LANES_END
    TL_LANES_BEGIN
    for (int t = lane; t < ntone + nnoise; t += 64) tl_masker_consts(TL_MK4(w), TL_MK_X(w), TL_MK_BARK(w), t, t < ntone);
    TL_LANES_END
    // Each lane folds the maskers into two ADJACENT table lines at once (two independent dB-sum chains).  A masker only
    // reaches lines with -3 <= dz < 8 bark, so a lane first finds the first and last masker (tones, then noise, in list
    // order) that reaches either of its lines and walks only that span; the per-line range test stays in the walk, so
    // nothing depends on the lists being sorted.
    const bool srt = ntone < 128 && nnoise < 64 && tl_maskers_sorted(TL_MK_BARK(w), ntone, nnoise);
    for (int base = 1; base < (TL_EXP_LEVEL >= 1 ? 0 : sub); base += 128) {       // 126..132 lines: one full pass + a 4-line tail at most
        TL_LANES_BEGIN
        const int k0 = base + 2 * lane, k1 = k0 + 1;
        const bool h0 = k0 < sub, h1 = k1 < sub;
        if (h0) {
            double bk0 = L(tb0), bk1 = L(tb1), hr0 = L(th0), hr1 = L(th1);      // the first pass uses the rows fetched above
            if (base != 1) {                                         // (a second pass is a 4-line tail at most; the branch is wave-uniform)
                bk0 = C->p1_bark[k0]; bk1 = C->p1_bark[h1 ? k1 : k0]; hr0 = C->p1_hear[k0]; hr1 = C->p1_hear[h1 ? k1 : k0];
            }
            const double blo = (bk0 < bk1 ? bk0 : bk1) - 8.0, bhi = (bk0 < bk1 ? bk1 : bk0) + 3.0;
            const TlMasker *mk = TL_MK4(w);
            const int nm = ntone + nnoise;
            int a0, a1, b0, b1;                                     // spans inside the tone part and inside the noise part
            if (srt) tl_mask_spans_sorted<64, 32>(TL_MK_BARK(w), ntone, nnoise, blo, bhi, a0, a1, b0, b1);
            else tl_mask_spans(mk, nm, ntone, blo, bhi, a0, a1, b0, b1);
            // one walk over the tone span, then the noise span, two maskers per trip: their four masking terms do not depend on
            // the running sums and are computed while the masker reads and the previous table look-ups are under way
            double x0 = TL_DBMIN, x1 = TL_DBMIN;
            const int nt = a1 >= a0 ? a1 - a0 + 1 : 0, cnt = nt + (b1 >= b0 ? b1 - b0 + 1 : 0);
            int t = nt ? a0 : b0;
            TL_DBG_WALK(ch, lane, cnt);
            const TlMaskK kk = tl_mask_consts();
            for (int i = 0; i < cnt; i += 2) {
                const int tA = t, tB = tA == a1 ? b0 : tA + 1;
                t = tB == a1 ? b0 : tB + 1;
                const TlMasker *pA = &mk[tA & (TL_MASKER_MAX - 1)], *pB = &mk[tB & (TL_MASKER_MAX - 1)];
                const double bA = pA->bark, avA = pA->av, bB = pB->bark, avB = pB->av;
                const bool two = i + 1 < cnt;                           // an odd walk ends with a masker that reaches nothing
                const double mA0 = tl_mask_term_w(pA, bA - bk0, avA, kk.far_hi), mA1 = tl_mask_term_w(pA, bA - bk1, avA, kk.far_hi);
                const double mB0 = tl_mask_term_w(pB, bB - bk0, avB, kk.far_hi, two), mB1 = tl_mask_term_w(pB, bB - bk1, avB, kk.far_hi, two);
                tl_add_db2_k(db, kk.k1000, x0, mA0, x1, mA1);
                tl_add_db2_k(db, kk.k1000, x0, mB0, x1, mB1);
            }
            TL_LTG(w)[k0] = tl_add_db(db, C->br_per_ch < 96 ? hr0 : hr0 - 12.0, x0);
            if (h1) TL_LTG(w)[k1] = tl_add_db(db, C->br_per_ch < 96 ? hr1 : hr1 - 12.0, x1);
        }
        TL_LANES_END
    }
    TL_STAMP(sp, 6);

    // ---- minimum per subband (psycho_1.c:541-559) and SMR (psycho_1.c:568-581) ----
    TL_LANES_BEGIN
    if (lane < C->sblimit) {
        double m;
        int n = L(mmn), j0 = L(mmj);
        if (n == 0) m = C->p1_hear[sub - 1];
        else {
            m = tl_min_rows(TL_LTG(w), j0, n, 0.0, true);
        }
        L(rec)[2 + ch] = m;                                         // the encoder finishes the line (tl_encode_frame, TL_PSY_EXT): SMR = max(spike, scale level) - m, psycho_1.c:575-580
    } else if (lane < 32) L(rec)[2 + ch] = 0.0;                     // subbands the model leaves alone
    TL_LANES_END
}

// band levels, decimation (psycho_1.c:390-470) and everything after; the regular (not dead-head) case
// (stamp4: the two-channel path stamps the chain stage itself -- channel 1's slot 4 = both chains begin, channel 0's slot 4 = both chains
// end, tl_psy1_stereo -- so that a parked channel's waiting time is not booked as its "noise bands"; tools/stage_profile.py)
TL_FN void tl_psy1_back(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, const TlPsy1Ch &st, PARGA(double, rec, 4), long long *sp, bool stamp4 = true)
{
    const int nbands = C->p1_ncb - 1, nlist = st.nlist;
    int ntone = 0, nnoise = 0;
    if (stamp4) TL_STAMP(sp, 4);
    if (TL_EXP_LEVEL >= 2) { tl_psy1_thresholds(w, db, C, ch, 0, 0, rec, sp); return; }
    // The reference now writes every band's sum to power[centre] in band order -- a later band overwrites an earlier one
    // that chose the same line, and (through the centre+1 rule above) a centre may even land on a tone's line
    // (psycho_1.c:390-398) -- and the decimation reads the levels back from power[].  The same values without the array:
    // a band's level is the sum of the LAST band with its centre, a tone's level is its own unless a band centre sits on
    // its line.  (Lane reads, no LDS round trips.)
    PV(int, ncen); PV(double, nlev); PV(int, nsh); PV(int, nsl); PV(bool, ontone);
    TL_LANES_BEGIN
    L(ncen) = lane < nbands ? (int)w.ncentre[lane] : -1 - lane;
    const double v = lane < nbands ? w.nsum[lane] : 0.0;
    L(nlev) = v; L(nsh) = (int)(uint32_t)(tl_d2u(v) >> 32); L(nsl) = (int)(uint32_t)tl_d2u(v);
    L(ontone) = lane < nbands && w.ptype[L(ncen)] == TL_T_TONE;
    TL_LANES_END
    const bool centre_on_tone = TL_BALLOT(ontone) != 0ull;            // only then can a tone's level be replaced (rare)
    // Two bands with the same centre are rare too.  Every band writes its index at its centre in a scratch map (the candidate
    // records are dead by now) and reads it back: with all centres distinct every band finds itself; otherwise some band
    // finds another one (whichever write lands last) and the overwrite order is resolved band by band.
    PV(bool, shared_c);
    uint8_t *cmark = (uint8_t *)w.cinfo;                              // centres are 1..510
    TL_LANES_BEGIN if (lane < nbands) cmark[L(ncen)] = (uint8_t)lane; TL_LANES_END
    TL_LANES_BEGIN L(shared_c) = lane < nbands && cmark[L(ncen)] != (uint8_t)lane; TL_LANES_END
    if (TL_BALLOT(shared_c) != 0ull)
        for (int b = 1; b < nbands; b++) {
            const int cb = TL_READLANE_I32(ncen, b);
            const double vb = tl_u2d(((uint64_t)(uint32_t)TL_READLANE_I32(nsh, b) << 32) | (uint32_t)TL_READLANE_I32(nsl, b));
            TL_LANES_BEGIN if (lane < b && L(ncen) == cb) L(nlev) = vb; TL_LANES_END
        }

    // ---- decimation (psycho_1.c:409-470) ----
    {
        // the noise components' table values (bark and threshold in quiet of their centres) are requested NOW and used after the tones:
        // the configuration lives in HBM / L2, and the tones' own table reads are then under way at the same time
        PV(double, nbk0); PV(double, nhear0);
        TL_LANES_BEGIN
        const int c0 = lane < nbands ? L(ncen) : 1;
        L(nbk0) = C->p1_lbark[c0]; L(nhear0) = C->p1_lhear[c0];
        TL_LANES_END
        // tones: keep if not erased and not below the threshold in quiet (order preserved)
        for (int base = 0; base < nlist; base += 64) {
            PV(bool, keep); PV(double, kx); PV(double, kb); PV(int, tline); PV(int, tcc); PV(double, tbk); PV(double, thr);
            TL_LANES_BEGIN
            double x = 0; int c = -1000 - lane, cc = 0;
            if (base + lane < nlist) { const int ti = w.tlist[base + lane]; cc = w.conf_c[ti]; c = cc & 511; x = w.tone_x[ti]; }
            L(kx) = x; L(tline) = c; L(tcc) = cc;
            // the tone's table values are requested as soon as its line is known (the configuration record is in HBM / L2)
            L(tbk) = C->p1_lbark[c < 0 ? 0 : c]; L(thr) = C->p1_lhear[c < 0 ? 0 : c];
            TL_LANES_END
            if (centre_on_tone)
                for (int b = 0; b < nbands; b++) {                  // a band centre on the tone's line replaces its level
                    const int cb = TL_READLANE_I32(ncen, b);
                    const double vb = tl_u2d(((uint64_t)(uint32_t)TL_READLANE_I32(nsh, b) << 32) | (uint32_t)TL_READLANE_I32(nsl, b));
                    TL_LANES_BEGIN if (L(tline) == cb) L(kx) = vb; TL_LANES_END
                }
            TL_LANES_BEGIN
            bool kp = false; double bk = 0;
            if (base + lane < nlist) {
                bk = L(tbk);
                kp = !((L(tcc) >> 13) & 1) && !(L(kx) < L(thr));
            }
            L(keep) = kp; L(kb) = bk;
            TL_LANES_END
            const uint64_t m = TL_BALLOT(keep);
            TL_LANES_BEGIN
            if ((m >> lane) & 1ull) {
                const int pos = ntone + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                TL_MK_X(w)[pos] = L(kx); TL_MK_BARK(w)[pos] = L(kb);
            }
            TL_LANES_END
            ntone += __builtin_popcountll(m);
        }
        // tones closer than 0.5 bark: keep the stronger (psycho_1.c:443-469).  The walk compares each tone with the current
        // survivor; as long as no two NEIGHBOURS of the list are that close the survivor is always the previous tone and
        // nothing is merged, which one line-parallel comparison settles.  Only otherwise the sequential walk runs.
        {
            PV(bool, closep);
            TL_LANES_BEGIN
            bool cl = false;
            for (int q = 1 + lane; q < ntone; q += 64) cl = cl || (TL_MK_BARK(w)[q] - TL_MK_BARK(w)[q - 1] < 0.5);
            L(closep) = cl;
            TL_LANES_END
            if (TL_BALLOT(closep) != 0ull) {
                int n = 0;                // compacted in place: entries [0,n) are final, (xi,bi) is the current survivor
                double xi = TL_MK_X(w)[0], bi = TL_MK_BARK(w)[0];
                for (int q = 1; q < ntone; q++) {
                    const double xn = TL_MK_X(w)[q], bn = TL_MK_BARK(w)[q];
                    if (bn - bi < 0.5) {
                        if (xn > xi) { xi = xn; bi = bn; }           // drop i, continue from next
                    } else { TL_MK_X(w)[n] = xi; TL_MK_BARK(w)[n] = bi; n++; xi = xn; bi = bn; }
                }
                TL_MK_X(w)[n] = xi; TL_MK_BARK(w)[n] = bi; n++;
                ntone = n;
                TL_SYNC();
            }
        }
        // noise: band order, keep if not below the threshold in quiet (psycho_1.c:429-442)
        PV(bool, keepn); PV(double, nx); PV(double, nb);
        TL_LANES_BEGIN
        bool kp = false; double x = 0, bk = 0;
        if (lane < nbands) {
            x = L(nlev); bk = L(nbk0);
            kp = !(x < L(nhear0));
        }
        L(keepn) = kp; L(nx) = x; L(nb) = bk;
        TL_LANES_END
        const uint64_t mn = TL_BALLOT(keepn);
        TL_LANES_BEGIN
        if ((mn >> lane) & 1ull) {
            const int pos = ntone + __builtin_popcountll(mn & ((1ull << lane) - 1ull));
            TL_MK_X(w)[pos] = L(nx); TL_MK_BARK(w)[pos] = L(nb);
        }
        TL_LANES_END
        nnoise = __builtin_popcountll(mn);
    }
    tl_psy1_thresholds(w, db, C, ch, ntone, nnoise, rec, sp);
}

// the dead-head replay (see tl_psy1_front): works on power[] (px) and the shared links like the reference
TL_FN void tl_psy1_deadhead(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, const TlPsy1Ch &st, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1, nconf = st.nconf;
    const uint8_t *map = C->p1_map;
    double *px = TL_PX(w);
    int ntone = 0, nnoise = 0;
    TL_STAMP(sp, 4);
    {
        // power[] as the replay needs it.  The array itself was compacted in place (tl_psy1_front), but the replay only ever
        // reads the lines of its chains: a confirmed tone's line holds the tone's summed level (psycho_1.c:317-321) unless
        // its successor erased it (DBMIN, :322-326); every other line the replay can reach is non-tonal and inside the
        // bands, i.e. consumed by its band (DBMIN, psycho_1.c:363).
        TL_LANES_BEGIN
        for (int j = lane; j < 520; j += 64) px[j] = TL_DBMIN;
        TL_LANES_END
        TL_LANES_BEGIN
        for (int i = lane; i < nconf; i += 64) { const int cc = w.conf_c[i]; if (!((cc >> 13) & 1)) px[cc & 511] = w.tone_x[i]; }
        TL_LANES_END
        TL_DBG_DUMP("deadhead", ch, 0, 0, px, px);
        int16_t *pnext = (int16_t *)w.cinfo;                // candidate records are dead by now
        TL_LANES_BEGIN
        for (int i = lane; i < 512; i += 64) pnext[i] = TL_STOP;
        TL_LANES_END
        TL_LANES_BEGIN
        for (int i = lane; i < nconf; i += 64) {
            const int nx = w.conf_nxt[i];
            pnext[w.conf_c[i] & 511] = (int16_t)(nx >= 0 ? (w.conf_c[nx] & 511) : nx);
        }
        TL_LANES_END
        int tone = w.conf_c[0] & 511, noise = 0;
        {   // noise chain in band order (psycho_1.c:390-398)
            int last = TL_LAST;
            for (int i = 0; i < nbands; i++) {
                const int centre = w.ncentre[i];
                if (last == TL_LAST) noise = centre;
                else { pnext[centre] = TL_LAST; pnext[last] = (int16_t)centre; }
                px[centre] = w.nsum[i]; w.ptype[centre] = TL_T_NOISE; last = centre;
            }
        }
        {   // psycho_1.c:409-470 verbatim on the shared links
            int i = tone, old = TL_STOP, guard = 0;
            while (i != TL_LAST && i != TL_STOP && guard++ < 600) {
                if (px[i] < C->p1_hear[map[i]]) {
                    w.ptype[i] = 0; px[i] = TL_DBMIN;
                    if (old == TL_STOP) tone = pnext[i]; else pnext[old] = pnext[i];
                } else old = i;
                i = pnext[i];
            }
            i = noise; old = TL_STOP; guard = 0;
            while (i != TL_LAST && i != TL_STOP && guard++ < 600) {
                if (px[i] < C->p1_hear[map[i]]) {
                    w.ptype[i] = 0; px[i] = TL_DBMIN;
                    if (old == TL_STOP) noise = pnext[i]; else pnext[old] = pnext[i];
                } else old = i;
                i = pnext[i];
            }
            i = tone; old = TL_STOP; guard = 0;
            while (i != TL_LAST && i != TL_STOP && guard++ < 600) {
                const int nx = pnext[i];
                if (nx == TL_LAST) break;
                if (nx == TL_STOP) break;                     // (the reference would index power[-100]; never reached in practice)
                if (C->p1_bark[map[nx]] - C->p1_bark[map[i]] < 0.5) {
                    if (px[nx] > px[i]) {
                        if (old == TL_STOP) tone = nx; else pnext[old] = (int16_t)nx;
                        w.ptype[i] = 0; px[i] = TL_DBMIN; i = nx;
                    } else {
                        w.ptype[nx] = 0; px[nx] = TL_DBMIN;
                        pnext[i] = pnext[nx]; old = i;
                    }
                } else { old = i; i = nx; }
            }
            guard = 0;
            for (int t = tone; t != TL_LAST && t != TL_STOP && ntone < TL_MASKER_MAX - 32 && guard++ < 600; t = pnext[t]) {
                TL_MK_X(w)[ntone] = px[t]; TL_MK_BARK(w)[ntone] = C->p1_bark[map[t]]; ntone++;
            }
            guard = 0;
            for (int t = noise; t != TL_LAST && t != TL_STOP && ntone + nnoise < TL_MASKER_MAX && guard++ < 600; t = pnext[t]) {
                TL_MK_X(w)[ntone + nnoise] = px[t]; TL_MK_BARK(w)[ntone + nnoise] = C->p1_bark[map[t]]; nnoise++;
            }
        }
        TL_SYNC();
    }
    tl_psy1_thresholds(w, db, C, ch, ntone, nnoise, rec, sp);
}

// one channel start to end (mono streams; stereo streams when a dead-head case forces the plain order)
TL_FN void tl_psy1_finish(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, const TlPsy1Ch &st, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1;
    PV(double, wt); PV(double, bsum); PV(int, blo); PV(int, bhi);
    tl_psy1_limits(C, nbands, blo, bhi);
    TL_PRIO(1); tl_psy1_chain(w, db, nbands, bsum, wt); TL_PRIO(0);
    tl_psy1_centres(w, nbands, bsum, wt, blo, bhi);
    if (st.dead_head) tl_psy1_deadhead(w, db, C, ch, st, rec, sp); else tl_psy1_back(w, db, C, ch, st, rec, sp);
}
TL_FN void tl_psy1(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                   const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const TlPsy1Ch st = tl_psy1_front(w, T, db, C, pv, ch, rec, sp);
    tl_psy1_finish(w, db, C, ch, st, rec, sp);
}

// Both channels of a stereo frame.  Order: front(0) -> park channel 0's front results in registers -> front(1) -> the dB-sum
// chains of both channels side by side -> back(1) -> channel 0's results return to the LDS arrays -> back(0).
// Parked: the compacted levels (<= 466 doubles: 8 per lane), the tone records (conf_c, tlist, tone_x), the spike levels, the
// band ranges and the weight sums.  ptype[] is not parked: after the tone labelling a line is TONE exactly if it is the line
// of a confirmed tone that was not erased by its successor, so it is rebuilt from conf_c.  A dead-head channel (see
// tl_psy1_front) falls back to the plain per-channel order.
TL_FN void tl_psy1_stereo(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                          const TlConfig *TL_RESTRICT C, const TlPcmView &pv, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1;
    long long *sp0 = sp ? sp + 8 : nullptr, *sp1 = sp ? sp + 16 : nullptr;
    const TlPsy1Ch s0 = tl_psy1_front(w, T, db, C, pv, 0, rec, sp0);
    if (s0.dead_head) {                                               // plain order for both channels
        tl_psy1_finish(w, db, C, 0, s0, rec, sp0);
        tl_psy1(w, T, db, C, pv, 1, rec, sp1);
        return;
    }
    // ---- park channel 0 ----
    PV(double, wt0); PV(int, r0); PV(int, r1);
    PA(double, pvp, 8); PV(int, pcc); PV(int, ptl); PV(double, ptx0); PV(double, ptx1);
    if (TL_EXP_LEVEL < 3) tl_psy1_weights(w, nbands, wt0);
    TL_LANES_BEGIN
    L(r0) = lane < nbands ? (int)w.bandoff[lane] : 0; L(r1) = lane < nbands ? (int)w.bandoff[lane + 1] : 0;
    const double *vp = TL_PX(w);
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) L(pvp)[k] = lane + 64 * k < 504 ? vp[lane + 64 * k] : 0.0;
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    L(pcc) = (int)((uint32_t)(uint16_t)w.conf_c[lane] | ((uint32_t)(uint16_t)w.conf_c[hi] << 16));
    L(ptl) = (int)((uint32_t)(uint16_t)w.tlist[lane] | ((uint32_t)(uint16_t)w.tlist[hi] << 16));
    L(ptx0) = w.tone_x[lane]; L(ptx1) = w.tone_x[hi];
    TL_LANES_END
    // ---- channel 1's front; a dead-head channel 1 is finished in the plain order first ----
    const TlPsy1Ch s1 = tl_psy1_front(w, T, db, C, pv, 1, rec, sp1);
    PV(double, bsum); PV(double, wt1); PV(int, blo); PV(int, bhi);
    tl_psy1_limits(C, nbands, blo, bhi);                              // (used after the chains, by both channels' centres)
    if (s1.dead_head) {
        tl_psy1_finish(w, db, C, 1, s1, rec, sp1);
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int k = 0; k < 8; k++) if (lane + 64 * k < 504) TL_PX(w)[lane + 64 * k] = L(pvp)[k];
        if (lane <= nbands) w.bandoff[lane] = (int16_t)(lane < nbands ? L(r0) : 0);
        TL_LANES_END
        // bandoff[nbands] = end of the last band
        {
            const int last_end = TL_READLANE_I32(r1, nbands - 1);
            TL_LANES_BEGIN if (lane == 0) w.bandoff[nbands] = (int16_t)last_end; TL_LANES_END
        }
        PV(double, wdummy);
        // the weight terms are gone; tl_psy1_chain's weight output is ignored (the parked sums are used)
        TL_LANES_BEGIN
        for (int i = lane; i < 504; i += 64) { uint64_t z = 0; TL_KEEP(z); w.u.fft[i] = tl_u2d(z); }   // (a zero made here, not a register kept through the frame)
        TL_LANES_END
        TL_PRIO(1); tl_psy1_chain(w, db, nbands, bsum, wdummy); TL_PRIO(0);
    } else {
        // ---- both chains: channel 1's weight sums first (its terms sit where channel 0's levels go) ----
        if (TL_EXP_LEVEL < 3) tl_psy1_weights(w, nbands, wt1);
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int k = 0; k < 8; k++) if (lane + 64 * k < 504) w.u.fft[lane + 64 * k] = L(pvp)[k];
        TL_LANES_END
        TL_STAMP(sp1, 4);
        if (TL_EXP_LEVEL < 3) { TL_PRIO(1); tl_psy1_chain2(w, db, nbands, r0, r1, bsum); TL_PRIO(0); }
        TL_STAMP(sp0, 4);                                               // both channels' chains: sp1[4] -> sp0[4]
        // ---- back(1): its sums move from lanes 32+b to lanes b ----
        PV(double, bsum1);
#ifdef TL_EMULATE
        for (int lane = 0; lane < 64; ++lane) bsum1[lane] = bsum[(lane + 32) & 63];
#else
        bsum1 = __shfl(bsum, (int)((threadIdx.x + 32u) & 63u), 64);
#endif
        if (TL_EXP_LEVEL < 3) tl_psy1_centres(w, nbands, bsum1, wt1, blo, bhi);
        tl_psy1_back(w, db, C, 1, s1, rec, sp1, false);
    }
    // ---- channel 0 returns to the LDS arrays ----
    TL_LANES_BEGIN
    for (int i = lane; i < 520; i += 64) w.ptype[i] = 0;
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    w.conf_c[lane] = (int16_t)(L(pcc) & 0xffff); w.tlist[lane] = (int16_t)(L(ptl) & 0xffff); w.tone_x[lane] = L(ptx0);
    if (64 + lane < TL_TONE_MAX) { w.conf_c[hi] = (int16_t)((uint32_t)L(pcc) >> 16); w.tlist[hi] = (int16_t)((uint32_t)L(ptl) >> 16); w.tone_x[hi] = L(ptx1); }
    TL_LANES_END
    TL_LANES_BEGIN
    for (int i = lane; i < s0.nconf; i += 64) { const int cc = w.conf_c[i]; if (!((cc >> 13) & 1)) w.ptype[cc & 511] = TL_T_TONE; }
    TL_LANES_END
    if (TL_EXP_LEVEL < 3) tl_psy1_centres(w, nbands, bsum, wt0, blo, bhi);
    tl_psy1_back(w, db, C, 0, s0, rec, sp0, s1.dead_head);
}

// ------------------------------------------------------------------------------------------
// psy model 3 (psycho_3.c:71-432) for channel `ch`; result in w.smr[ch][0..32).
TL_FN int tl_psy3_front(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                   const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const double *energy = w.u.fft;                                   // line i at TL_EX(i)
    double *px = TL_PX(w);
    const double *bark = C->p3_bark, *ath = C->p3_ath;
    TL_PRIO2(TL_PS_FHT);
    TL_STAMP(sp, 0);
    tl_psy_spectrum(w, T, pv, ch, sp);
    TL_STAMP(sp, 1);
    TL_PRIO2(TL_PS_POW);

    // power[1..512] (psycho_3.c:152-160); power[0] is an uninitialised slot in the reference, pinned
    // to 0.0 (oracle/mp2_oracle.c:psy3_run)
    // and the strongest line Xmax of each subband (psycho_3.c:163-183; line 512 is skipped, see oracle) -> the output record.  A subband's 16 lines sit in one row of
    // 16 lanes, so its maximum is a row reduction of the values just computed (no strided re-read of px).
    // The logarithms in two halves as in tl_psy1_front (tl_power_db_main / tl_power_near1).  Line 0 has no logarithm (pinned), line 512 has one
    // and no lane of its own: lane 0's first slot computes line 512 instead of line 0 (a ninth pass for one lane would cost what a pass of 64 does).
    PA(double, pxa, 8);
    int ndef = 0;
    for (int h = 0; h < 2; h++) {                                   // four lines per lane at a time
        PV(bool, n0); PV(bool, n1); PV(bool, n2); PV(bool, n3);
        TL_LANES_BEGIN
        {
            double e[4], v[4];
            bool nr[4];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 4; q++) { const int i = lane + 64 * (4 * h + q); e[q] = energy[TL_EX(i == 0 ? 512 : i)]; }
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 4; q++) v[q] = tl_power_db_main(e[q], TL_LOGTAB(db), &nr[q]);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 4; q++) { const int i = lane + 64 * (4 * h + q); px[i == 0 ? 512 : i] = v[q]; }
            if (h == 0 && lane == 0) px[0] = 0.0;
            L(n0) = nr[0]; L(n1) = nr[1]; L(n2) = nr[2]; L(n3) = nr[3];
        }
        TL_LANES_END
        TL_POWER_DEFER(w, n0, (h == 0 && lane == 0) ? 512 : lane + 256 * h, ndef);
        TL_POWER_DEFER(w, n1, lane + 256 * h + 64, ndef);
        TL_POWER_DEFER(w, n2, lane + 256 * h + 128, ndef);
        TL_POWER_DEFER(w, n3, lane + 256 * h + 192, ndef);
        while (ndef >= 64) { ndef -= 64; TL_DBG_NEAR1_FULL(); tl_power_near1(w, ndef, 64); }
    }
    if (ndef) tl_power_near1(w, 0, ndef);
    TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) { const int i = lane + 64 * k; L(pxa)[k] = i == 0 ? TL_DBMIN : px[i]; }
    TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int it = 0; it < 8; it++) {
        PV(double, pxv); PV(double, pxm);
        TL_LANES_BEGIN L(pxv) = L(pxa)[it]; TL_LANES_END
        TL_ROW16_MAX_F64(pxm, pxv);
        // subband 4 it + r: the maximum sits in lane 16 r + 15; the record keeps it in lane 4 it + r (the encoder takes the
        // maximum with the scalefactor level, psycho_3.c:180-182)
        PV(double, xm);
        TL_LANES_BEGIN L(xm) = TL_DBMIN < L(pxm) ? L(pxm) : TL_DBMIN; TL_LANES_END
        TL_LANES_BEGIN
        {
            const double v = TL_OTHER(xm, , 16 * (lane & 3) + 15);
            if (lane < 32 && (lane >> 2) == it) L(rec)[ch] = v;
        }
        TL_LANES_END
    }
    TL_PRIO2(1);                                                      // from here to the thresholds: serial stages
    // ---- tone labelling (psycho_3.c:186-247) ----
    // (1) local maxima 2..499 whose right-hand neighbours are >= 7 dB down, compacted ascending
    int ncand = 0;
    tl_cand_chunk<2, true>(w, 0, ncand);                           // lines -1..62: run 2
    tl_cand_chunk<3, true>(w, 1, ncand);                           // 63..126: run 3
    for (int c8 = 2; c8 < 4; c8++) tl_cand_chunk<6, true>(w, c8, ncand);     // 127..254: run 6
    for (int c8 = 4; c8 < 8; c8++) tl_cand_chunk<12, true>(w, c8, ncand);    // 255..510: run 12
    TL_STAMP(sp, 2);
    // (2) which candidates become tones: the candidates in lanes, one round per confirmed tone (see tl_psy1_front).  A confirmed tone k
    //     erases lines k-sr..k+sr (itself included) to DBMIN (psycho_3.c:243-244); a later maximum inside that reach R has power
    //     DBMIN and always fails, one above R sees erased left neighbours (always >= 7 dB down) and original ones beyond.
    int nconf = 0;
    {
        int R = -1;
        for (int kb = 0; kb < ncand; kb += 64) {
        PV(int, ck); PV(int, csr); PV(int, clf); PV(bool, act); PV(bool, dep);
        TL_LANES_BEGIN
        const bool in = kb + lane < ncand;
        const uint32_t info = w.cinfo[in ? kb + lane : 0];
        const uint32_t pinfo = w.cinfo[in && kb + lane > 0 ? kb + lane - 1 : 0];    // the candidate before this one
        L(ck) = (int)(info & 511u); L(csr) = (int)(info >> 21);
        L(clf) = (int)tl_cand_left<true>(px, L(ck), L(csr), px[L(ck)]);
        L(act) = in;
        // state-independent candidates as in tl_psy1_front: at least sr(previous candidate) + sr + 1 lines above the candidate before
        // it, a candidate is above every earlier reach R with all its left neighbours original -- its verdict is clf == 0 under any R
        L(dep) = in && kb + lane > 0 && L(ck) - (int)(pinfo & 511u) < (int)(pinfo >> 21) + L(csr) + 1;
        TL_LANES_END
        for (;;) {
            PV(bool, okv);
            TL_LANES_BEGIN
            const int k = L(ck), sr = L(csr);
            const int hi_j = sr < k - R - 1 ? sr : k - R - 1;         // bit j-2 set for j in [2, sr] with k-j > R
            const uint32_t orig = hi_j >= 2 ? (1u << (hi_j - 1)) - 1u : 0u;
            L(okv) = L(act) && k > R && !((uint32_t)L(clf) & orig);
            TL_LANES_END
            const uint64_t m = TL_BALLOT(okv);
            if (m == 0ull) break;
            const int wl = __builtin_ctzll(m);
            // the first passing candidate, and with it every passing state-independent one up to the next dependent candidate
            PV(bool, depact);
            TL_LANES_BEGIN L(depact) = L(dep) && L(act) && lane > wl; TL_LANES_END
            const uint64_t dm = TL_BALLOT(depact);
            const uint64_t batch = m & (dm ? (1ull << __builtin_ctzll(dm)) - 1ull : ~0ull);
            const int nb = __builtin_popcountll(batch), wlast = 63 - __builtin_clzll(batch);
            const int i0 = nconf, Rold = R;
            TL_LANES_BEGIN
            if ((batch >> lane) & 1ull) {
                const int i = i0 + __builtin_popcountll(batch & ((1ull << lane) - 1ull));
                if (i < TL_TONE_MAX) w.conf_c[i] = (int16_t)(L(ck) | ((lane == wl && L(ck) - 1 <= Rold) ? (1 << 12) : 0));
            }
            TL_LANES_END
            nconf = i0 + nb < TL_TONE_MAX ? i0 + nb : (i0 < TL_TONE_MAX ? TL_TONE_MAX : i0);
            R = TL_READLANE_I32(ck, wlast) + TL_READLANE_I32(csr, wlast);
            TL_LANES_BEGIN L(act) = L(act) && lane > wlast; TL_LANES_END
            TL_DBG_ROUND();
        }
        }
        TL_SYNC();
    }
    // (3) tone levels from the still-original spectrum (psycho_3.c:238-239); kept aside until the energies
    //     are dead (the masker lists share the FHT buffer)
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {
        const int cc = w.conf_c[i], k = cc & 511;
        const double temp = tl_add_db(db, (cc >> 12) & 1 ? TL_DBMIN : px[k - 1], px[k]);
        w.tone_x[i] = tl_add_db(db, temp, px[k + 1]);
    }
    TL_LANES_END
    // (4) erasures
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {                         // straight-line, as in tl_psy1_front: sr is 2, 3, 6 or 12
        const int k = w.conf_c[i] & 511, sr = tl_run_psy3(k);
        double *pb = px + (k - 12);
#define TL_ERASE3(j) do { pb[12 - (j)] = TL_DBMIN; pb[12 + (j)] = TL_DBMIN; } while (0)
        pb[12] = TL_DBMIN; TL_ERASE3(1); TL_ERASE3(2);
        if (sr >= 3) {
            TL_ERASE3(3);
            if (sr >= 6) {
                TL_ERASE3(4); TL_ERASE3(5); TL_ERASE3(6);
                if (sr >= 12) { TL_ERASE3(7); TL_ERASE3(8); TL_ERASE3(9); TL_ERASE3(10); TL_ERASE3(11); TL_ERASE3(12); }
            }
        }
#undef TL_ERASE3
    }
    TL_LANES_END
    TL_STAMP(sp, 3);
    // ---- noise per critical band (psycho_3.c:264-304) + decimation (:313-320); one lane per band ----
    const int nb = C->p3_cbands;
    // Line-parallel preparation as in psy 1: the lines that are summed (not erased) are compacted in ascending
    // order -- levels in place in px[], energies in place in the FHT buffer (a compacted position is always below
    // its line), and each entry's distance j - lo from its band's first line (the factor of its centre-of-gravity term
    // (j-lo)*e, psycho_3.c:283-289) as 16 bits in the candidate records' place, which are dead by now -- so the per-band
    // part is three bare chains.
    {
        double *ve = w.u.fft, *vp = px;
        uint16_t *vj = (uint16_t *)w.cinfo;                          // [512]
        static_assert(sizeof(w.cinfo) >= 512 * sizeof(uint16_t), "distance records");
        PV(double, e512);
        TL_LANES_BEGIN L(e512) = energy[512]; TL_LANES_END
        int nvalid = 0;
        PA(uint32_t, linfo, 9);                                     // the table reads of all nine chunks in one batch
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int c8 = 0; c8 < 9; c8++) L(linfo)[c8] = 64 * c8 + lane < 520 ? C->p3_lineinfo[64 * c8 + lane] : 0u;
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int base = 0; base < 576; base += 64) {                // lines 1..512
            PV(bool, ok); PV(double, ev); PV(int, dj); PV(double, pvv); PV(int, bnd);
            TL_LANES_BEGIN
            const int j = base + lane;
            const uint32_t info = L(linfo)[base >> 6];
            const int lo = (int)((info >> 8) & 0xfffu);
            bool v = false; double e = 0, p = 0;
            if (info) {
                p = px[j];
                v = p != TL_DBMIN;
                e = j == 512 ? L(e512) : energy[TL_EX(j)];
            }
            L(ok) = v; L(ev) = e; L(dj) = j - lo; L(pvv) = p; L(bnd) = (info && j == lo) ? (int)(info & 0xffu) : -1;
            TL_LANES_END
            const uint64_t m = TL_BALLOT(ok);
            TL_LANES_BEGIN
            const int pos = nvalid + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (L(bnd) >= 0) w.bandoff[L(bnd)] = (int16_t)pos;     // first line of its band
            if (L(ok)) { ve[pos] = L(ev); vj[pos] = (uint16_t)L(dj); vp[pos] = L(pvv); }
            TL_LANES_END
            nvalid += __builtin_popcountll(m);
        }
        w.bandoff[nb] = (int16_t)nvalid;
        TL_SYNC();
    }
    return nconf;
}

// energy sum and centre-of-gravity sum of the bands (psycho_3.c:283-289), ascending line order; lane b < nb.  Used on their own
// for stereo frames, where only the levels take part in the shared dB-sum chain.
TL_FN void tl_psy3_moments(TlPsyLds &w, int nb, PARG(double, es), PARG(double, cg))
{
    TL_LANES_BEGIN
    double esum = 0, cw = 0;
    if (lane < nb) {
        const double *ve = w.u.fft;
        const uint16_t *vj = (const uint16_t *)w.cinfo;
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                               // operands of eight steps per LDS round trip, summed in order
            double e[8], c[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) { e[q] = ve[i + q]; c[q] = (int)vj[i + q] * e[q]; }     // (j - lo) * e, psycho_3.c:287
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) { esum += e[q]; cw += c[q]; }
        }
        for (; i < i1; i++) { esum += ve[i]; cw += (int)vj[i] * ve[i]; }
    }
    L(es) = esum; L(cg) = cw;
    TL_LANES_END
}

// dB sums, energy sums and centre-of-gravity sums of the bands of ONE channel: lane b < nb
TL_FN void tl_psy3_chain(TlPsyLds &w, const double *TL_RESTRICT db, int nb, PARG(double, bsum), PARG(double, es), PARG(double, cg))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN, esum = 0, cw = 0;
    if (lane < nb) {
        const double *ve = w.u.fft, *vp = TL_PX(w);
        const uint16_t *vj = (const uint16_t *)w.cinfo;
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 4 <= i1; i += 4) {                               // operands of four steps in flight per LDS round trip
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            const double e0 = ve[i], e1 = ve[i + 1], e2 = ve[i + 2], e3 = ve[i + 3];
            const double c0 = (int)vj[i] * e0, c1 = (int)vj[i + 1] * e1, c2 = (int)vj[i + 2] * e2, c3 = (int)vj[i + 3] * e3;
            sum = tl_add_db(db, p0, sum); esum += e0; cw += c0;
            sum = tl_add_db(db, p1, sum); esum += e1; cw += c1;
            sum = tl_add_db(db, p2, sum); esum += e2; cw += c2;
            sum = tl_add_db(db, p3, sum); esum += e3; cw += c3;
        }
        for (; i < i1; i++) { sum = tl_add_db(db, vp[i], sum); esum += ve[i]; cw += (int)vj[i] * ve[i]; }
    }
    L(bsum) = sum; L(es) = esum; L(cg) = cw;
    TL_LANES_END
}

// dB sums of BOTH channels at once: lanes 0..31 walk channel 0's bands (levels parked at fft[], ranges in r0/r1), lanes 32..63
// channel 1's (levels in px[], ranges from bandoff[]).  Lane b holds channel 0's sum, lane 32+b channel 1's.
TL_FN void tl_psy3_chain2(TlPsyLds &w, const double *TL_RESTRICT db, int nb, PARG(int, r0), PARG(int, r1), PARG(double, bsum))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN;
    const int band = lane & 31;
    if (band < nb) {
        const bool second = lane >= 32;
        const double *vp = second ? TL_PX(w) : w.u.fft;
        const int i0 = second ? (int)w.bandoff[band] : L(r0), i1 = second ? (int)w.bandoff[band + 1] : L(r1);
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                               // eight steps' operands per LDS round trip
            double p[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) p[q] = vp[i + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) sum = tl_add_db(db, p[q], sum);
        }
        for (; i + 4 <= i1; i += 4) {
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            sum = tl_add_db(db, p0, sum); sum = tl_add_db(db, p1, sum); sum = tl_add_db(db, p2, sum); sum = tl_add_db(db, p3, sum);
        }
        for (; i < i1; i++) sum = tl_add_db(db, vp[i], sum);
    }
    L(bsum) = sum;
    TL_LANES_END
}

// band centres, decimation, thresholds, SMR (psycho_3.c:290-432) from the sums of lanes b < nb
TL_FN void tl_psy3_back(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, int nconf,
                        PARG(double, bsum), PARG(double, es), PARG(double, cg), PARGA(double, rec, 4), long long *sp, bool stamp4 = true)
{
    const double *bark = C->p3_bark, *ath = C->p3_ath;
    const int nb = C->p3_cbands;
    PV(bool, keepn); PV(double, nx); PV(double, nbk);
    // the first 64 tones' table values (bark, threshold in quiet: the configuration record in HBM / L2) are requested before the noise
    // components' -- whose addresses are themselves table values -- so that the two chains of reads overlap
    PV(double, tbk0); PV(double, tath0);
    TL_LANES_BEGIN
    const int k0 = lane < nconf ? (w.conf_c[lane] & 511) : 0;
    L(tbk0) = bark[k0]; L(tath0) = ath[k0];
    TL_LANES_END
    TL_LANES_BEGIN
    bool kp = false; double xn = 0, bk = 0;
    if (lane < nb) {
        const int lo = C->p3_cbidx[lane], hi = C->p3_cbidx[lane + 1];
        const double sum = L(bsum), esum = L(es), cw = L(cg);
        // esum == 0: the reference indexes with (int)(0/0) and segfaults; defined as the band centre
        int centre = (sum <= TL_DBMIN || esum == 0) ? (lo + hi) / 2 : lo + (int)(cw / esum);
        centre = centre < 1 ? 1 : centre > 512 ? 512 : centre;
        xn = sum; bk = bark[centre];
        kp = !(xn < ath[centre]);
    }
    L(keepn) = kp; L(nx) = xn; L(nbk) = bk;
    TL_LANES_END
    const uint64_t mn = TL_BALLOT(keepn);       // ascending line order == band order (centres stay in their band)
    // tones: decimation against the threshold in quiet (psycho_3.c:321-326), compaction in ascending line order
    int ntone = 0;
    for (int base = 0; base < nconf; base += 64) {
        PV(bool, keep); PV(double, kx); PV(double, kb);
        TL_LANES_BEGIN
        bool kp2 = false; double x = 0, bk2 = 0;
        if (base + lane < nconf) {
            x = w.tone_x[base + lane];
            if (base == 0) { bk2 = L(tbk0); kp2 = !(x < L(tath0)); }
            else { const int k = w.conf_c[base + lane] & 511; bk2 = bark[k]; kp2 = !(x < ath[k]); }
        }
        L(keep) = kp2; L(kx) = x; L(kb) = bk2;
        TL_LANES_END
        const uint64_t m = TL_BALLOT(keep);
        TL_LANES_BEGIN
        if ((m >> lane) & 1ull) {
            const int pos = ntone + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            TL_MK_X(w)[pos] = L(kx); TL_MK_BARK(w)[pos] = L(kb);
        }
        TL_LANES_END
        ntone += __builtin_popcountll(m);
    }
    TL_LANES_BEGIN
    if ((mn >> lane) & 1ull) {
        const int pos = ntone + __builtin_popcountll(mn & ((1ull << lane) - 1ull));
        TL_MK_X(w)[pos] = L(nx); TL_MK_BARK(w)[pos] = L(nbk);
    }
    TL_LANES_END
    const int nnoise = __builtin_popcountll(mn);
    if (stamp4) TL_STAMP(sp, 4);                                      // (the two-channel path stamps its chain stage itself: tl_psy3_stereo)
    TL_STAMP(sp, 5);
    TL_PRIO2(TL_PS_THR);
    // ---- thresholds on the 136 subsampled lines (psycho_3.c:339-406) ----
    // the lane's two subset lines and their bark values (two DEPENDENT reads of the configuration record in HBM / L2) are requested
    // before the masker records are built and the order test runs
    PV(int, sl0); PV(int, sl1); PV(double, sb0); PV(double, sb1); PV(int, sbj); PV(int, sbn);
    TL_LANES_BEGIN
    L(sl0) = C->p3_subset[2 * lane]; L(sl1) = C->p3_subset[2 * lane + 1];
    L(sb0) = bark[L(sl0)]; L(sb1) = bark[L(sl1)];
    L(sbj) = C->p3_sb_j0[lane & 31]; L(sbn) = C->p3_sb_n[lane & 31];               // rows of the subband's minimum (used at the very end)
    TL_LANES_END
    TL_LANES_BEGIN
    for (int t = lane; t < ntone + nnoise; t += 64) tl_masker_consts(TL_MK4(w), TL_MK_X(w), TL_MK_BARK(w), t, t < ntone);
    TL_LANES_END
    const bool srt = ntone < 128 && nnoise < 64 && tl_maskers_sorted(TL_MK_BARK(w), ntone, nnoise);
    // lines 0..127: every lane folds the maskers into two ADJACENT lines (two independent dB-sum chains at a time) and
    // walks only the maskers that can reach one of them (-3 <= dz < 8 bark; the exact test stays in the step)
    TL_LANES_BEGIN
    {
        const int j0 = 2 * lane, j1 = j0 + 1;
        const int line0 = L(sl0), line1 = L(sl1);
        const double b0 = L(sb0), b1 = L(sb1);
        const TlMasker *mk = TL_MK4(w);
        int ta0, ta1, tb0, tb1;
        if (srt) tl_mask_spans_sorted<64, 32>(TL_MK_BARK(w), ntone, nnoise, (b0 < b1 ? b0 : b1) - 8.0, (b0 < b1 ? b1 : b0) + 3.0, ta0, ta1, tb0, tb1);
        else tl_mask_spans(mk, ntone + nnoise, ntone, (b0 < b1 ? b0 : b1) - 8.0, (b0 < b1 ? b1 : b0) + 3.0, ta0, ta1, tb0, tb1);
        double lt0 = TL_DBMIN, ln0 = TL_DBMIN, lt1 = TL_DBMIN, ln1 = TL_DBMIN;
        uint32_t far_hi = 0xC0F00000u;
        TL_PIN(far_hi);
        for (int t = ta0; t <= ta1; t++) {
            const double mb = mk[t].bark, av = mk[t].av;
            lt0 = tl_mask_step(db, lt0, &mk[t], mb - b0, av, far_hi);
            lt1 = tl_mask_step(db, lt1, &mk[t], mb - b1, av, far_hi);
        }
        for (int t = tb0; t <= tb1; t++) {
            const double mb = mk[t].bark, av = mk[t].av;
            ln0 = tl_mask_step(db, ln0, &mk[t], mb - b0, av, far_hi);
            ln1 = tl_mask_step(db, ln1, &mk[t], mb - b1, av, far_hi);
        }
        const double g0 = tl_add_db(db, ln0, lt0), g1 = tl_add_db(db, ln1, lt1);
        TL_LTG(w)[j0] = tl_add_db(db, C->br_per_ch < 96 ? ath[line0] : ath[line0] - 12.0, g0);
        TL_LTG(w)[j1] = tl_add_db(db, C->br_per_ch < 96 ? ath[line1] : ath[line1] - 12.0, g1);
    }
    TL_LANES_END
    // lines 128..135: the tone sum and the noise sum of a line are independent chains (psycho_3.c:350-395), so
    // lanes 0..7 run the tone chains and lanes 8..15 the noise chains of the eight lines side by side
    TL_LANES_BEGIN
#if TL_P3_NOTAIL
    if (lane < 16) w.nsum[lane] = TL_DBMIN;                           // diagnostic builds only (tools/variant_hip.sh notail): what the eight top lines cost
    if (false) {
#else
    if (lane < 16) {
#endif
        const int j = 128 + (lane & 7), line = C->p3_subset[j];
        const double bj = bark[line];
        const TlMasker *mk = TL_MK4(w);
        int ta0, ta1, tb0, tb1;                                       // only the maskers that can reach the line (it is one of the top eight)
        if (srt) tl_mask_spans_sorted<64, 32>(TL_MK_BARK(w), ntone, nnoise, bj - 8.0, bj + 3.0, ta0, ta1, tb0, tb1);
        else tl_mask_spans(mk, ntone + nnoise, ntone, bj - 8.0, bj + 3.0, ta0, ta1, tb0, tb1);
        const int t0 = lane < 8 ? ta0 : tb0, t1 = lane < 8 ? ta1 : tb1;
        double acc = TL_DBMIN;
        uint32_t far_hi = 0xC0F00000u;
        TL_PIN(far_hi);
        for (int t = t0; t <= t1; t++) {
            const double mb = mk[t].bark, av = mk[t].av;
            acc = tl_mask_step(db, acc, &mk[t], mb - bj, av, far_hi);
        }
        w.nsum[lane] = acc;
    }
    TL_LANES_END
    TL_LANES_BEGIN
    if (lane < 8) {
        const int j = 128 + lane, line = C->p3_subset[j];
        const double g = tl_add_db(db, w.nsum[8 + lane], w.nsum[lane]);
        TL_LTG(w)[j] = tl_add_db(db, C->br_per_ch < 96 ? ath[line] : ath[line] - 12.0, g);
    }
    TL_LANES_END
    TL_STAMP(sp, 6);
    // ---- minimum per subband + SMR (psycho_3.c:409-432); subset rows of subband sb are contiguous ----
    TL_LANES_BEGIN
    if (lane < 32) {
        double m = 999999.9;
        const int j0 = L(sbj), n = L(sbn);
        m = tl_min_rows(TL_LTG(w), j0, n, m, false);
        L(rec)[2 + ch] = m;
    }
    TL_LANES_END
}


TL_FN void tl_psy3(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                   const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const int nconf = tl_psy3_front(w, T, db, C, pv, ch, rec, sp);
    PV(double, bsum); PV(double, es); PV(double, cg);
    TL_PRIO(1); tl_psy3_chain(w, db, C->p3_cbands, bsum, es, cg); TL_PRIO(0);
    tl_psy3_back(w, db, C, ch, nconf, bsum, es, cg, rec, sp);
}

// Both channels of a stereo frame, organised like tl_psy1_stereo: front(0) -> channel 0's compacted levels, tone records, Lsb
// and band moments wait in registers -> front(1) -> both dB-sum chains side by side -> back(1) -> back(0).
TL_FN void tl_psy3_stereo(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                          const TlConfig *TL_RESTRICT C, const TlPcmView &pv, PARGA(double, rec, 4), long long *sp)
{
    const int nb = C->p3_cbands;
    long long *sp0 = sp ? sp + 8 : nullptr, *sp1 = sp ? sp + 16 : nullptr;
    const int nconf0 = tl_psy3_front(w, T, db, C, pv, 0, rec, sp0);
    PV(double, es0); PV(double, cg0); PV(int, r0); PV(int, r1);
    PA(double, pvp, 8); PV(int, pcc); PV(double, ptx0); PV(double, ptx1);
    tl_psy3_moments(w, nb, es0, cg0);
    TL_LANES_BEGIN
    L(r0) = lane < nb ? (int)w.bandoff[lane] : 0; L(r1) = lane < nb ? (int)w.bandoff[lane + 1] : 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) L(pvp)[k] = TL_PX(w)[lane + 64 * k];
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    L(pcc) = (int)((uint32_t)(uint16_t)w.conf_c[lane] | ((uint32_t)(uint16_t)w.conf_c[hi] << 16));
    L(ptx0) = w.tone_x[lane]; L(ptx1) = w.tone_x[hi];
    TL_LANES_END
    const int nconf1 = tl_psy3_front(w, T, db, C, pv, 1, rec, sp1);
    PV(double, es1); PV(double, cg1); PV(double, bsum); PV(double, bsum1);
    tl_psy3_moments(w, nb, es1, cg1);
    TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) w.u.fft[lane + 64 * k] = L(pvp)[k];   // channel 1's energies are summed: the buffer's lower half is free
    TL_LANES_END
    TL_STAMP(sp1, 4);                                                   // both channels' chains: sp1[4] -> sp0[4] (tools/stage_profile.py)
    TL_PRIO(1); tl_psy3_chain2(w, db, nb, r0, r1, bsum); TL_PRIO(0);
    TL_STAMP(sp0, 4);
#ifdef TL_EMULATE
    for (int lane = 0; lane < 64; ++lane) bsum1[lane] = bsum[(lane + 32) & 63];
#else
    bsum1 = __shfl(bsum, (int)((threadIdx.x + 32u) & 63u), 64);
#endif
    tl_psy3_back(w, db, C, 1, nconf1, bsum1, es1, cg1, rec, sp1, false);
    TL_LANES_BEGIN
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    w.conf_c[lane] = (int16_t)(L(pcc) & 0xffff); w.tone_x[lane] = L(ptx0);
    if (64 + lane < TL_TONE_MAX) { w.conf_c[hi] = (int16_t)((uint32_t)L(pcc) >> 16); w.tone_x[hi] = L(ptx1); }
    TL_LANES_END
    tl_psy3_back(w, db, C, 0, nconf0, bsum, es0, cg0, rec, sp0, false);
}
