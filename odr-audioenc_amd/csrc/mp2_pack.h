// mp2_pack.h -- the encoder of one frame after the model: scalefactors, transmission pattern, SMR, allocation, quantiser, fields, CRCs, DAB tail (tl_encode_frame, tl_encode_pair).
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ------------------------------------------------------------------------------------------
// One frame of one stream.  lane = 2*sb + ch owns subband sb of channel ch.  PSY (the psy model) is a compile-time
// parameter: one kernel per model keeps each kernel's code and register footprint to what that model needs.
// Where a frame of the frame-parallel encode kernel goes (exactly one of bytes / words is set): the output slot it waits in
// for its successor's ScF-CRC, or -- the last frame of a launch -- the batch's pending buffer (big-endian words like
// TlStreamState::pending); and the slot for its own ScF-CRC bytes, which tl_finish_stream stores into the frame before it.
struct TlFrameOut { uint8_t *bytes; uint32_t *words; uint8_t *scfcrc; };
template <int PSY, int NCH = 0>
TL_FN void tl_encode_frame(TlMainLds &w, const TlTables *TL_RESTRICT T, const TlBlockShared *TL_RESTRICT B,
                           const TlConfig *TL_RESTRICT C, const TlPsyOut *TL_RESTRICT PO,
                           const TlPcmView &pv, int xpad_len, const TlFrameOut &fo,
                           const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, int padding, TlTaps *taps, long long *sp)
{
    constexpr int FB = TlMainLds::kFbBatch;
    const int nch = NCH ? NCH : C->nch, sblimit = C->sblimit;
    PA(double, smp, 36);            // sb_sample[ch][gr][bl][sb] of this lane's (sb,ch), b = gr*12+bl
    PA(int, scf, 3);

    TL_STAMP(sp, 0);
    TL_PRIO2(TL_PS_FB);
    // ---- K1: polyphase filterbank ----
    tl_filterbank(w, B, enw_s, nch, smp);

    TL_STAMP(sp, 1);
    TL_PRIO2(1);
#if !defined(TL_EMULATE) && TL_ENC_LEVEL >= 5
    for (int b = 0; b < 36; b++) TL_KEEP(smp[b]);
    scf[0] = scf[1] = scf[2] = 0;
#endif
    // the lane's allocation line and field width come from the configuration record (HBM / L2): requested here, used by the allocation
    PV(int, a_ln); PV(int, a_nbal);
    TL_LANES_BEGIN
    {
        const int c = lane & 1, sb = lane >> 1;
        const bool live = c < nch && sb < sblimit;
        L(a_ln) = live ? C->line[sb] : 0;
        L(a_nbal) = live ? C->nbal[sb] : 0;
    }
    TL_LANES_END
    // ---- K2: scalefactors (encode_new.c:179-230) + find_sf_max (:260-277) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (TL_ENC_LEVEL >= 5) { }
    else if (c < nch && sb < sblimit) {
        unsigned lo = 63;
        for (int gr = 0; gr < 3; gr++) {
            double m = fabs(L(smp)[gr * 12 + 11]);
            for (int j = 10; j >= 0; j--) { double t = fabs(L(smp)[gr * 12 + j]); if (t > m) m = t; }
            unsigned idx = tl_sf_index(B->scalefactor, m);
            L(scf)[gr] = (int)idx;
            w.scf[c][gr][sb] = (uint8_t)idx;
            if (idx < lo) lo = idx;
        }
        w.minidx[c][sb] = (uint8_t)lo;
    } else {
        L(scf)[0] = L(scf)[1] = L(scf)[2] = 0;
        if (sb >= sblimit || c >= nch) { w.minidx[c][sb] = 63; w.scf[c][0][sb] = w.scf[c][1][sb] = w.scf[c][2][sb] = 0; }
    }
    TL_LANES_END

    // joint stereo: scalefactors of .5*(L+R) (toolame.c:332-337, encode_new.c:237-246)
    if (TL_ENC_LEVEL < 5 && C->mode0 == 1) {
        for (int gr = 0; gr < 3; gr++) {
            PV(double, jm);
            TL_LANES_BEGIN L(jm) = 0.0; TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int j = 11; j >= 0; j--) {
                PV(double, other);
#ifdef TL_EMULATE
                for (int lane = 0; lane < 64; ++lane) other[lane] = smp[lane ^ 1][gr * 12 + j];
#else
                other = tld_swap1_f64(smp[gr * 12 + j]);
#endif
                TL_LANES_BEGIN
                double t = fabs(.5 * (L(smp)[gr * 12 + j] + L(other)));     // ch0 lane: .5*(L+R)
                if (j == 11 || t > L(jm)) L(jm) = t;
                TL_LANES_END
            }
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            if (c == 0 && sb < sblimit) w.jscale[gr][sb] = (uint8_t)tl_sf_index(B->scalefactor, L(jm));
            TL_LANES_END
        }
    }

    if (taps) {
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        for (int b = 0; b < 36; b++) taps->sb_sample[c][b / 12][b % 12][sb] = L(smp)[b];
        for (int gr = 0; gr < 3; gr++) { taps->scalar_pre[c][gr][sb] = w.scf[c][gr][sb]; if (c == 0) taps->j_scale[gr][sb] = C->mode0 == 1 && sb < sblimit ? w.jscale[gr][sb] : 0; }
        taps->max_sc[c][sb] = (c < nch && sb < sblimit) ? B->scalefactor[w.minidx[c][sb]] : 1E-20;
        TL_LANES_END
    }

    TL_STAMP(sp, 2);
    // ---- K3/K4: psychoacoustic model -> SMR (toolame.c:361-452) ----
    if constexpr (TL_ENC_LEVEL >= 5) { }
    else if constexpr (PSY == 0) {                                    // psycho_0.c:52-68
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        if (c < nch) {
            int m = sb < sblimit ? (int)w.minidx[c][sb] : 0;     // scalar[] above sblimit stays 0 (toolame.c:132)
            w.smr[c][sb] = 2.0 * (30.0 - m) - C->p0_athmin[sb];
        }
        TL_LANES_END
    } else if constexpr (PSY == 2) {
        TL_LANES_BEGIN                                           // models 2 and 4: the psy-2 kernel left the SMR itself
        const int c = lane & 1, sb = lane >> 1;
        if (c < nch) w.smr[c][sb] = PO->a[c][sb];
        TL_LANES_END
    } else {
        // models 1 and 3: the model (tl_frame_unit ran it before this frame body) left, per (channel, subband), the level A that
        // competes with the scalefactor level (in smr[]) and the minimum masking threshold m (in psy_m[]); the SMR line itself
        // needs this frame's scalefactors and is finished here:
        // psycho_1.c:575-580 (max = scale level; if (spike > max) max = spike; smr = max - ltmin) and psycho_3.c:180-182,428
        // (Lsb = max(Xmax, scale level); smr = Lsb - ltmin) are the same three operations.
        static_assert(PSY == TL_PSY_EXT, "models 1 and 3");
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        if (c < nch) {
            const double a = w.smr[c][sb], m = w.psy_m[c][sb];
            const double val = B->scale_db[w.minidx[c][sb]];
            const double top = a > val ? a : val;
            w.smr[c][sb] = top - m;
        }
        TL_LANES_END
    }

    TL_STAMP(sp, 3);
    // ---- sf_transmission_pattern (encode_new.c:288-354, ISO Table C.4) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (TL_ENC_LEVEL < 5 && c < nch && sb < sblimit) {
        int s0 = L(scf)[0], s1 = L(scf)[1], s2 = L(scf)[2];
        int d0 = s0 - s1, d1 = s1 - s2;
        int c0 = d0 <= -3 ? 0 : d0 < 0 ? 1 : d0 == 0 ? 2 : d0 < 3 ? 3 : 4;
        int c1 = d1 <= -3 ? 0 : d1 < 0 ? 1 : d1 == 0 ? 2 : d1 < 3 ? 3 : 4;
        // pattern of the class pair: where each transmitted scalefactor comes from, and scfsi (no branches: a lane per cell)
        const unsigned p = B->sfpat[c0 * 5 + c1];
        const int m02 = s0 > s2 ? s2 : s0;                              // pattern 444: the larger scalefactor (smaller index) of the outer two
        const unsigned q0 = p & 3u, q1 = (p >> 2) & 3u, q2 = (p >> 4) & 3u;
        const int n0 = q0 == 0 ? s0 : q0 == 1 ? s1 : q0 == 2 ? s2 : m02;
        const int n1 = q1 == 0 ? s0 : q1 == 1 ? s1 : q1 == 2 ? s2 : m02;
        const int n2 = q2 == 0 ? s0 : q2 == 1 ? s1 : q2 == 2 ? s2 : m02;
        const int sel = (int)(p >> 6);
        s0 = n0; s1 = n1; s2 = n2;
        L(scf)[0] = s0; L(scf)[1] = s1; L(scf)[2] = s2;
        w.scf[c][0][sb] = (uint8_t)s0; w.scf[c][1][sb] = (uint8_t)s1; w.scf[c][2][sb] = (uint8_t)s2;
        w.scfsi[c][sb] = (uint8_t)sel;
    } else w.scfsi[c][sb] = 0;
    w.balloc[c][sb] = 0;
    TL_LANES_END

    // ---- K5: bit allocation (encode_new.c:733-886, :634-705, :1061-1187) ----
    const int lg_frame = C->frame_bytes + padding;                  // availbits.c:64: (whole + extra) slots
    int adb = lg_frame * 8 - (C->dab_ext * 8 + (xpad_len ? xpad_len : 2) * 8);     // toolame.c:292-301
    int mode = C->mode0, mode_ext = C->mode_ext0, jsbound = C->jsbound0;
    // per-lane constants of the allocation loops
    PV(int, a_sfs); PV(int, a_sfs_o); PV(double, a_smr); PV(double, a_smr_o);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = c < nch && sb < sblimit;
    L(a_sfs) = live ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0;
    L(a_sfs_o) = (live && nch == 2) ? 6 * tl_sfs_count(w.scfsi[1 - c][sb]) : 0;
    L(a_smr) = live ? w.smr[c][sb] : 0.0;
    L(a_smr_o) = (live && nch == 2) ? w.smr[1 - c][sb] : 0.0;
    TL_LANES_END
    if (TL_ENC_LEVEL < 4 && C->mode0 == 1) {
        // try plain stereo, then jsbound 16, 12, 8, 4 (encode_new.c:803-819).  What a cell needs for "no audible noise"
        // (bits_for_nonoise_new, encode_new.c:634-705) does not depend on the trial: the SNR column of an allocation line
        // is increasing, so the first allocation that masks the cell's own SMR is the number of allocations that do not,
        // and above jsbound (where the search goes on against the other channel's SMR) it is the larger of the two counts.
        // Both counts and both prices are computed once; a trial only selects and sums.
        PV(int, nz_own); PV(int, nz_jnt);
        TL_LANES_BEGIN
        int bo = 0, bj = 0;
        if (lane < 2 * sblimit) {
            const int ln = L(a_ln), maxAlloc = (1 << L(a_nbal)) - 1;
            double sv[15];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 15; q++) sv[q] = B->snr_line[ln][q];
            int n1 = 0, n2 = 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 15; q++) {
                const bool inr = q < maxAlloc - 1;
                n1 += (inr && !((sv[q] - L(a_smr)) >= 0.0)) ? 1 : 0;
                n2 += (inr && !((sv[q] - L(a_smr_o)) >= 0.0)) ? 1 : 0;
            }
            n2 = n2 > n1 ? n2 : n1;
            bo = n1 > 0 ? B->bits12_line[ln][n1] + 2 + L(a_sfs) : 0;
            bj = n2 > 0 ? B->bits12_line[ln][n2] + 4 + L(a_sfs) + L(a_sfs_o) : 0;
        }
        L(nz_own) = bo; L(nz_jnt) = bj;
        TL_LANES_END
        mode = 0; mode_ext = 0; jsbound = sblimit;
        int tries = 0, try_ext = 4;
        for (;;) {
            PV(int, need);
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            int bitsn = 0;
            if (sb < sblimit && c < (sb < jsbound ? nch : 1))
                bitsn = ((nch == 2 && sb >= jsbound) ? L(nz_jnt) : L(nz_own)) + L(a_nbal);     // + the bbal share of this (sb,ch)
            L(need) = bitsn;
            TL_LANES_END
            int rq = 32 + 16 + TL_WAVE_SUM_I32(need);
            if (tries == 0) {
                if (rq > adb) { mode = 1; } else break;
            } else if (!(rq > adb && try_ext > 0)) { mode_ext = try_ext; break; }
            --try_ext; jsbound = 4 * (try_ext + 1); tries++;          // 16, 12, 8, 4
        }
    }
    int adb_left;
    {   // a_bit_allocation_new (encode_new.c:1078-1187).  Every lane carries its cell's order-preserving mnr key,
        // its ba and the price of its next step; one wave arg-min per iteration.
        // The reference marks a cell used=2 when its next step does not fit.  The bits left only shrink and a
        // cell's price only changes when it wins, so a cell that does not fit now never fits later: such cells
        // are left out of the arg-min right away (same result, no iterations spent on refusals).
        //
        // A cell's mnr only grows with its allocation (the SNR column of an allocation line is increasing), so the
        // greedy order is the merge of the cells' ascending key lists.  Every cell therefore carries the key and the
        // price of its next step AND of the step after it; with M = the smallest second key in the wave, the cells
        // whose next key is below M are exactly the greedy order's next events (no second step can come before
        // them).  If together they still fit, they are all taken in one round; the one-at-a-time loop takes over
        // when a round no longer fits (or is empty), so the refusal rule above is applied event by event.
        PV(int, ba);
        adb_left = tl_allocate(B, adb, nch, sblimit, jsbound, a_ln, a_nbal, a_sfs, a_sfs_o, a_smr, ba);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        w.balloc[c][sb] = (uint8_t)((c < nch && sb < sblimit) ? L(ba) : 0);
        TL_LANES_END
    }

    TL_STAMP(sp, 4);
    // ---- K6: header, CRC, bit_alloc, scfsi, scalefactors, quantised samples -> LDS frame ----
    uint32_t *frame = w.u.frame[0];
    TL_LANES_BEGIN
    for (int i = lane; i < ((lg_frame + 3) >> 2) + 2; i += 64) frame[i] = 0;      // this frame's words (+ 2: tl_put_bits48)
    TL_LANES_END
    PV(int, f_ba); PV(int, f_sel); PV(int, f_scf); PV(int, f_smp);
    PV(int, o_ba); PV(int, o_sel); PV(int, o_scf); PV(int, o_smp);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = c < nch && sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0;
    const bool own = sb < sblimit && c < (sb < jsbound ? nch : 1);     // transmits bit_alloc + samples
    L(f_ba) = own ? L(a_nbal) : 0;
    L(f_sel) = (live && ba) ? 2 : 0;
    L(f_scf) = (live && ba) ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0;
    L(f_smp) = (own && ba) ? B->bits12_line[L(a_ln)][ba] / 12 : 0;     // group * bits of the cell's quantiser class
    TL_LANES_END
    TL_WAVE_EXSCAN_I32(o_ba, f_ba); TL_WAVE_EXSCAN_I32(o_sel, f_sel);
    TL_WAVE_EXSCAN_I32(o_scf, f_scf); TL_WAVE_EXSCAN_I32(o_smp, f_smp);
    const int n_ba = TL_WAVE_SUM_I32(f_ba), n_sel = TL_WAVE_SUM_I32(f_sel);
    const int n_scf = TL_WAVE_SUM_I32(f_scf), n_smp = TL_WAVE_SUM_I32(f_smp);
    const int p_ba = 48, p_sel = p_ba + n_ba, p_scf = p_sel + n_sel, p_smp = p_scf + n_scf;

    if (TL_ENC_LEVEL < 3) {
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (lane == 0) {     // write_header (encode_new.c:356-373)
        uint32_t h = (0xfffu << 20) | ((uint32_t)C->version << 19) | (2u << 17) | (0u << 16)
                   | ((uint32_t)C->br_idx << 12) | ((uint32_t)C->fs_idx << 10) | ((uint32_t)padding << 9) | (0u << 8)
                   | ((uint32_t)mode << 6) | ((uint32_t)mode_ext << 4);
        TL_ATOMIC_OR(&frame[0], h);
    }
    const bool live = c < nch && sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0;
    if (L(f_ba)) tl_put_bits48(frame, p_ba + L(o_ba), (uint64_t)ba, L(f_ba));
    if (L(f_sel)) {
        const unsigned si = w.scfsi[c][sb];
        tl_put_bits48(frame, p_sel + L(o_sel), si, 2);
        // write_scalefactors (encode_new.c:428-443): scfsi 0 -> three, 1/3 -> first and last, 2 -> one; as one field
        const unsigned s0 = (unsigned)L(scf)[0], s1 = (unsigned)L(scf)[1], s2 = (unsigned)L(scf)[2];
        const unsigned f3 = (s0 << 12) | (s1 << 6) | s2, f2 = (s0 << 6) | s2;
        tl_put_bits48(frame, p_scf + L(o_scf), si == 0 ? f3 : si == 2 ? s0 : f2, L(f_scf));
    }
    TL_LANES_END
    }

    TL_PRIO2(TL_PS_Q);
    // quantise (encode_new.c:479-547) + write_samples_new (:560-598): 12 rounds of 3 samples
    if (TL_ENC_LEVEL < 2) {
        const bool any_joint = (nch == 2) && jsbound < sblimit;      // joint-coded subbands exist in this frame
        // per-lane constants of the frame: quantiser class and its coefficients, the three scalefactors
        PV(int, q_ba); PV(int, q_nb); PV(int, q_grp); PV(int, q_s2n); PV(int, q_steps);
        PV(double, q_a); PV(double, q_b); PV(double, q_s2nf); PA(double, q_sf, 3); PA(double, q_rsf, 3);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        const bool own = sb < sblimit && c < (sb < jsbound ? nch : 1);
        const int ba = own ? w.balloc[c][sb] : 0;
        const unsigned qi = ba ? B->qinfo_line[L(a_ln)][ba] : 0u;   // class, bits and grouping from the shared LDS copy
        const int q = (int)(qi & 31u);
        const bool joint = any_joint && sb >= jsbound;
        // The field of a triple (encode_new.c:574-592) is ONE Horner form A + M (v1 + M C): three separate codewords of nb bits are
        // v2 + 2^nb (v1 + 2^nb v0), a grouped codeword is v0 + steps (v1 + steps v2) -- M = 2^nb or steps, (A, C) = (v2, v0) or (v0, v2).
        L(q_ba) = ba; L(q_nb) = (int)((qi >> 5) & 31u); L(q_grp) = ((qi >> 10) & 1u) ? 3 : 1; L(q_s2n) = K->steps2n[q];
        L(q_steps) = ((qi >> 10) & 1u) ? 1 << L(q_nb) : K->steps[q];                                // M
        L(q_a) = K->qa[q]; L(q_b) = K->qb[q]; L(q_s2nf) = K->steps2n_f[q];
        for (int gr = 0; gr < 3; gr++) {
            L(q_sf)[gr] = B->scalefactor[joint ? w.jscale[gr][sb] : L(scf)[gr]];
            L(q_rsf)[gr] = 1.0 / L(q_sf)[gr];                          // one division per granule instead of twelve
        }
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int r = 0; r < 12; r++) {
            const int gr = r >> 2, j0 = (r & 3) * 3;
            PA(double, oth, 3);
#ifdef TL_EMULATE
            for (int lane = 0; lane < 64; ++lane) for (int x = 0; x < 3; x++) oth[lane][x] = smp[lane ^ 1][gr * 12 + j0 + x];
#else
            if (any_joint) {
#pragma unroll
                for (int x = 0; x < 3; x++) oth[x] = tld_swap1_f64(smp[gr * 12 + j0 + x]);
            } else { oth[0] = oth[1] = oth[2] = 0.0; }
#endif
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            unsigned v[3] = {0, 0, 0};
            if (L(q_ba)) {
                const bool joint = any_joint && sb >= jsbound;
                const double sfv = L(q_sf)[gr], rsf = L(q_rsf)[gr];
                for (int x = 0; x < 3; x++) {
                    double s = L(smp)[gr * 12 + j0 + x];
                    if (joint) s = .5 * (s + L(oth)[x]);
                    double d = tl_div_by(s, sfv, rsf);                   // == s / sfv (encode_new.c:507,511)
                    d = d * L(q_a) + L(q_b);
                    const bool neg = !(d >= 0);                          // encode_new.c:528-534; d + 0.0 changes no quantised value
                    d += TL_SELECT(neg, 1.0, 0.0);
                    const unsigned qv = (unsigned)(d * L(q_s2nf));
                    v[x] = qv | (neg ? 0u : (unsigned)L(q_s2n));
                }
                // three codewords of nb bits, or one codeword v0 + v1*steps + v2*steps^2 of nb bits (encode_new.c:574-592): one field
                const int nb = L(q_nb);
                const int pos = p_smp + r * n_smp + L(o_smp);
                const bool three = L(q_grp) == 3;
                const unsigned M = (unsigned)L(q_steps);
                const unsigned fa = TL_SELECT(three, v[2], v[0]), fc = TL_SELECT(three, v[0], v[2]);
                const unsigned inner = v[1] + M * fc;                                    // < 2^32: nb <= 16
                tl_put_bits48(frame, pos, (uint64_t)fa + (uint64_t)M * (uint64_t)inner, L(q_grp) * nb);
            }
            if (taps) for (int x = 0; x < 3; x++) taps->subband[c][gr][j0 + x][sb] = (c < nch) ? v[x] : 0;
            TL_LANES_END
        }
    }

    TL_STAMP(sp, 5);
    TL_PRIO2(1);
    // CRC-16 over header bits 16..31, bit_alloc and scfsi fields (crc.c:12-41).
    // Protected message M = frame bits [16,32) then [48,p_scf), n bits.  The register after M with preset I is
    // (I(x) x^n + M(x) x^16) mod P -- linear over GF(2) -- so every lane takes one byte of M (a byte of the frame: the message is
    // byte aligned in it) and adds up bit_k * x^(16 + bits after the byte + k) mod P, starting from a table value and
    // multiplying by x per step; the two bytes of the preset ride on lanes 62/63; one XOR-reduce.
    unsigned crc16 = 0;
    if (TL_ENC_LEVEL < 1) {
    {
        const int n = 16 + (p_scf - 48);
        PV(uint32_t, part);
        TL_LANES_BEGIN
        uint32_t acc = 0;
        const bool preset = lane >= 62;
        const int first = 8 * lane;                                   // message bits [first, first + cnt)
        if (first < n || preset) {
            const int byte = lane < 2 ? lane + 2 : lane + 4;          // frame byte holding them
            const int cnt = preset ? 8 : (n - first < 8 ? n - first : 8);
            const int e0 = preset ? n + 8 * (63 - lane) : 16 + (n - first - cnt);     // exponent of the byte's last bit
            // bit k of the byte contributes x^(e0 + k) mod P: eight reads of the power table (one address, constant offsets) and a masked
            // XOR per bit -- instead of stepping x^e through the polynomial in registers (nine operations per bit)
            const uint16_t *xt = &K->crc_xpow[e0];
            const unsigned v = preset ? 0xffu : ((frame[byte >> 2] >> (24 - 8 * (byte & 3))) & 0xffu) >> (8 - cnt);
            unsigned x8[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 8; k++) x8[k] = xt[k];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 8; k++) acc ^= (0u - ((v >> k) & 1u)) & x8[k];            // bits past cnt are zero
        }
        L(part) = acc;
        TL_LANES_END
        crc16 = TL_WAVE_XOR_U32(part) & 0xffffu;
    }
    // ScF-CRC (crc.c:58-97, toolame.c:527-542).  Every (sb,ch) lane packs the 3 MSBs of the scalefactors it
    // transmits (crc.c:83-96) and folds it on its own; the band groups are combined below.
    const int tail = lg_frame - 2 - C->dab_ext;                     // byte offset of the first ScF-CRC byte
    PV(int, rlen); PV(uint32_t, rcrc);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    uint32_t rec = 0;
    if (c < nch && sb < sblimit && w.balloc[c][sb]) {
        const uint32_t s0 = (uint32_t)L(scf)[0] >> 3, s1 = (uint32_t)L(scf)[1] >> 3, s2 = (uint32_t)L(scf)[2] >> 3;
        switch (w.scfsi[c][sb]) {
        case 0: rec = (9u << 16) | (s0 << 6) | (s1 << 3) | s2; break;
        case 1: case 3: rec = (6u << 16) | (s0 << 3) | s2; break;
        default: rec = (3u << 16) | s0; break;
        }
    }
    if (lane == 0) tl_put_bits(frame, 32, crc16, 16);
    L(rlen) = (int)(rec >> 16); L(rcrc) = rec & 0x1ffu;               // the record's bits; folded below (crc.c:99-113)
    TL_LANES_END
    // The CRC register update is linear over GF(2): the CRC of a band group (records concatenated in (sb,ch) order, crc.c:58-97)
    // is the XOR of rec_l(x) * x^(8 + bits after record l) mod P.  Bits-after from a prefix sum of the lengths, x^e from a table,
    // one XOR scan, then the four group values are differences of that scan at the group boundaries.
    {
        PV(int, lex);
        TL_WAVE_EXSCAN_I32(lex, rlen);
        const int f[5] = {0, 4, 8, 16, 30};
        int gend[4], gfirst[4], glast[4];
        for (int g = 0; g < 4; g++) {
            gfirst[g] = f[g]; glast[g] = f[g + 1] > sblimit ? sblimit : f[g + 1];
            const int e = 2 * glast[g];
            gend[g] = (g < C->dab_ext && glast[g] > gfirst[g]) ? TL_READLANE_I32(lex, e) : 0;      // e <= 60
        }
        PV(uint32_t, part); PV(uint32_t, pscan);
        TL_LANES_BEGIN
        const int sb = lane >> 1;
        const int g = sb < 4 ? 0 : sb < 8 ? 1 : sb < 16 ? 2 : 3;
        const int after = (g == 0 ? gend[0] : g == 1 ? gend[1] : g == 2 ? gend[2] : gend[3]) - L(lex) - L(rlen);
        const int e0 = after + 8;                                    // <= 252 + 8: inside crc8_xpow[]
        unsigned xp = K->crc8_xpow[e0 < 0 ? 0 : e0 > 319 ? 319 : e0];
        unsigned acc = 0;
        const unsigned rb = L(rcrc);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int b = 0; b < 9; b++) {                                // acc = rec(x) * x^(8 + after) mod P, shift-and-add in GF(2)
            acc ^= ((rb >> b) & 1u) ? xp : 0u;
            xp = ((xp << 1) & 0xffu) ^ ((xp & 0x80u) ? 0x1Du : 0u);
        }
        L(part) = (L(rlen) && sb < sblimit) ? acc : 0u;
        TL_LANES_END
        TL_WAVE_INCL_XSCAN_U32(pscan, part);
        unsigned c8g[4];
        for (int g = 0; g < 4; g++) {
            c8g[g] = 0;
            if (g < C->dab_ext && glast[g] > gfirst[g]) {
                c8g[g] = (unsigned)TL_READLANE_I32(pscan, 2 * glast[g] - 1);
                if (gfirst[g] > 0) c8g[g] ^= (unsigned)TL_READLANE_I32(pscan, 2 * gfirst[g] - 1);
            }
        }
        TL_LANES_BEGIN
        if (lane < C->dab_ext) {
            const int grp = C->dab_ext - 1 - lane;                  // transmission order: i = dab_ext-1 .. 0
            const unsigned c8 = (grp == 0 ? c8g[0] : grp == 1 ? c8g[1] : grp == 2 ? c8g[2] : c8g[3]) & 0xffu;
            tl_put_bits(frame, (tail + lane) * 8, c8, 8);
            w.ncentre[lane] = (int16_t)c8;                           // reused as a 4-entry scratch
        }
        TL_LANES_END
    }
    TL_LANES_BEGIN
    // X-PAD + F-PAD bytes (toolame.c:515-524,544-551): xpad[] holds xpad_len bytes in transmission order
    if (xpad_len) {
        const int xstart = lg_frame - C->dab_ext - xpad_len;        // X-PAD sits right before the ScF-CRC
        for (int i = lane; i < xpad_len; i += 64) {
            int bytepos = i < xpad_len - 2 ? xstart + i : lg_frame - 2 + (i - (xpad_len - 2));
            tl_put_bits(frame, bytepos * 8, w.xpad[i], 8);
        }
    }
    TL_LANES_END

    } else {
        TL_LANES_BEGIN
        if (lane < 4) w.ncentre[lane] = 0;
        TL_LANES_END
    }
    if (taps) {
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        taps->smr[c][sb] = (c < nch && (C->psy != 1 || sb < sblimit)) ? w.smr[c][sb] : 0.0;   // psy 1 leaves sb >= sblimit unset
        taps->scfsi[c][sb] = w.scfsi[c][sb]; taps->bit_alloc[c][sb] = w.balloc[c][sb];
        for (int gr = 0; gr < 3; gr++) taps->scalar[c][gr][sb] = w.scf[c][gr][sb];
        if (lane == 0) { taps->adb_left = adb_left; taps->mode = mode; taps->mode_ext = mode_ext; taps->jsbound = jsbound; taps->crc16 = (int)crc16; }
        if (lane < 4) taps->scfcrc[lane] = lane < C->dab_ext ? (uint8_t)w.ncentre[lane] : 0;
        TL_LANES_END
    }

    TL_STAMP(sp, 6);
    // ---- emit (toolame.c:527-542 keeps "one frame in memory" to patch its ScF-CRC slot with the next frame's CRC) ----
    const int nwords = (lg_frame + 3) >> 2;
    // frames of a stream are encoded by different waves in any order: this one only files its frame and its ScF-CRC;
    // tl_finish_stream puts each frame's CRC into the frame before it once the launch's frames are all there
    TL_LANES_BEGIN
    for (int i = lane; i < nwords; i += 64) {
        if (fo.words) fo.words[i] = frame[i];
        else {
            const uint32_t le = tl_bswap(frame[i]);
            const int rem = lg_frame - 4 * i;
            if (rem >= 4) ((uint32_t *)fo.bytes)[i] = le;
            else for (int b = 0; b < rem; b++) fo.bytes[4 * i + b] = (uint8_t)(le >> (8 * b));
        }
    }
    if (lane < 4) fo.scfcrc[lane] = lane < C->dab_ext ? (uint8_t)w.ncentre[lane] : 0;
    TL_LANES_END
    TL_PRIO2(0);
    TL_STAMP(sp, 7);
}

// ------------------------------------------------------------------------------------------
// TWO mono streams of ONE configuration in one wave.  A lone mono frame leaves every second lane idle from the filterbank to the
// packing (lane = 2*sb + ch, ch = 0 only) and costs as many instructions as a stereo frame; here lane = 2*sb + u owns subband sb of
// UNIT u, the u-th of the two streams (same frame index f of the launch).  Filterbank, scalefactors, transmission pattern, quantiser
// and packing run for both units at once, the bit allocation too (tl_allocate_pair); the CRC-16 folds use one half-wave per unit.  What a unit produces is what
// tl_encode_frame produces for it alone: the operations per cell are the same text, and every wave-level sum, scan and minimum is
// taken over the unit's own lanes (a scan over both units carries unit 0 in the low and unit 1 in the high half of a word:
// sums stay below 2^16, XORs never carry).  toolame.c:267-554 twice, the `nch` loop of toolame.c:308-312 turned into lanes.
template <int PSY>
TL_FN void tl_encode_pair(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const TlConfig *TL_RESTRICT C, const TlPsyOut *const (&PO)[2],
                          const int (&xpad_len)[2], const uint8_t *const (&xpad_src)[2], const TlFrameOut (&fo)[2],
                          const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, const int (&padding)[2])
{
    const int sblimit = C->sblimit;
    const int padpk = padding[0] | (padding[1] << 1);             // both units' padding bits in one scalar (an array indexed by the lane would live in scratch)
    PA(double, smp, 36);
    PA(int, scf, 3);
    TL_PRIO2(TL_PS_FB);
    tl_filterbank(w, B, enw_s, 2, smp);
    TL_PRIO2(1);
    // ---- scalefactors (encode_new.c:179-230) + find_sf_max (:260-277) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (sb < sblimit) {
        unsigned lo = 63;
        for (int gr = 0; gr < 3; gr++) {
            double m = fabs(L(smp)[gr * 12 + 11]);
            for (int j = 10; j >= 0; j--) { double t = fabs(L(smp)[gr * 12 + j]); if (t > m) m = t; }
            unsigned idx = tl_sf_index(B->scalefactor, m);
            L(scf)[gr] = (int)idx;
            if (idx < lo) lo = idx;
        }
        w.minidx[c][sb] = (uint8_t)lo;
    } else {
        L(scf)[0] = L(scf)[1] = L(scf)[2] = 0;
        w.minidx[c][sb] = 63;
    }
    TL_LANES_END
    // ---- SMR (toolame.c:361-452), as in tl_encode_frame with c = the unit ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if constexpr (PSY == 0) {
        const int m = sb < sblimit ? (int)w.minidx[c][sb] : 0;
        w.smr[c][sb] = 2.0 * (30.0 - m) - C->p0_athmin[sb];
    } else if constexpr (PSY == 2) {
        w.smr[c][sb] = (c ? PO[1] : PO[0])->a[0][sb];
    } else {
        const double a = w.smr[c][sb], m = w.psy_m[c][sb];
        const double val = B->scale_db[w.minidx[c][sb]];
        const double top = a > val ? a : val;
        w.smr[c][sb] = top - m;
    }
    TL_LANES_END
    // ---- sf_transmission_pattern (encode_new.c:288-354, ISO Table C.4) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (sb < sblimit) {
        int s0 = L(scf)[0], s1 = L(scf)[1], s2 = L(scf)[2];
        int d0 = s0 - s1, d1 = s1 - s2;
        int c0 = d0 <= -3 ? 0 : d0 < 0 ? 1 : d0 == 0 ? 2 : d0 < 3 ? 3 : 4;
        int c1 = d1 <= -3 ? 0 : d1 < 0 ? 1 : d1 == 0 ? 2 : d1 < 3 ? 3 : 4;
        const unsigned p = B->sfpat[c0 * 5 + c1];
        const int m02 = s0 > s2 ? s2 : s0;
        const unsigned q0 = p & 3u, q1 = (p >> 2) & 3u, q2 = (p >> 4) & 3u;
        const int n0 = q0 == 0 ? s0 : q0 == 1 ? s1 : q0 == 2 ? s2 : m02;
        const int n1 = q1 == 0 ? s0 : q1 == 1 ? s1 : q1 == 2 ? s2 : m02;
        const int n2 = q2 == 0 ? s0 : q2 == 1 ? s1 : q2 == 2 ? s2 : m02;
        L(scf)[0] = n0; L(scf)[1] = n1; L(scf)[2] = n2;
        w.scfsi[c][sb] = (uint8_t)(p >> 6);
    } else w.scfsi[c][sb] = 0;
    w.balloc[c][sb] = 0;
    TL_LANES_END
    // ---- bit allocation (encode_new.c:733-886, :1061-1187): both units at once, each over its own lanes ----
    int lg_frame[2];
    {
        PV(int, a_ln); PV(int, a_nbal); PV(int, a_sfs); PV(double, a_smr); PV(int, ba);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        const bool live = sb < sblimit;
        L(a_ln) = live ? C->line[sb] : 0;
        L(a_nbal) = live ? C->nbal[sb] : 0;
        L(a_sfs) = live ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0;
        L(a_smr) = live ? w.smr[c][sb] : 0.0;
        TL_LANES_END
        int adb[2];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int u = 0; u < 2; u++) {
            lg_frame[u] = C->frame_bytes + padding[u];                                       // availbits.c:64
            adb[u] = lg_frame[u] * 8 - (C->dab_ext * 8 + (xpad_len[u] ? xpad_len[u] : 2) * 8);    // toolame.c:292-301
        }
        tl_allocate_pair(B, adb[0], adb[1], sblimit, a_ln, a_nbal, a_sfs, a_smr, ba);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        w.balloc[c][sb] = (uint8_t)(sb < sblimit ? L(ba) : 0);
        TL_LANES_END
    }
    // ---- header, bit_alloc, scfsi, scalefactors, quantised samples -> the two LDS frames ----
    TL_LANES_BEGIN
    for (int i = lane; i < ((lg_frame[0] + 3) >> 2) + 2; i += 64) w.u.frame[0][i] = 0;
    for (int i = lane; i < ((lg_frame[1] + 3) >> 2) + 2; i += 64) w.u.frame[1][i] = 0;
    TL_LANES_END
    PV(int, f_ba); PV(int, f_sel); PV(int, f_scf); PV(int, f_smp); PV(int, a_ln2);
    PV(int, o_ba); PV(int, o_sel); PV(int, o_scf); PV(int, o_smp);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0, sh = 16 * c;
    L(a_ln2) = live ? C->line[sb] : 0;
    L(f_ba) = (live ? (int)C->nbal[sb] : 0) << sh;
    L(f_sel) = (ba ? 2 : 0) << sh;
    L(f_scf) = (ba ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0) << sh;
    L(f_smp) = (ba ? B->bits12_line[L(a_ln2)][ba] / 12 : 0) << sh;
    TL_LANES_END
    TL_WAVE_EXSCAN_I32(o_ba, f_ba); TL_WAVE_EXSCAN_I32(o_sel, f_sel);
    TL_WAVE_EXSCAN_I32(o_scf, f_scf); TL_WAVE_EXSCAN_I32(o_smp, f_smp);
    const int s_ba = TL_WAVE_SUM_I32(f_ba), s_sel = TL_WAVE_SUM_I32(f_sel), s_scf = TL_WAVE_SUM_I32(f_scf), s_smp = TL_WAVE_SUM_I32(f_smp);
    int p_sel[2], p_scf[2], p_smp[2], n_smp[2];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++) {
        const int sh = 16 * u;
        p_sel[u] = 48 + ((s_ba >> sh) & 0xffff); p_scf[u] = p_sel[u] + ((s_sel >> sh) & 0xffff); p_smp[u] = p_scf[u] + ((s_scf >> sh) & 0xffff);
        n_smp[u] = (s_smp >> sh) & 0xffff;
    }
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1, sh = 16 * c;
    uint32_t *frame = w.u.frame[c];
    if (lane < 2) {      // write_header (encode_new.c:356-373), lane u for unit u
        uint32_t h = (0xfffu << 20) | ((uint32_t)C->version << 19) | (2u << 17) | (0u << 16)
                   | ((uint32_t)C->br_idx << 12) | ((uint32_t)C->fs_idx << 10) | ((uint32_t)((padpk >> c) & 1) << 9) | (0u << 8)
                   | ((uint32_t)C->mode0 << 6) | ((uint32_t)C->mode_ext0 << 4);
        TL_ATOMIC_OR(&frame[0], h);
    }
    const bool live = sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0;
    const int nb_ba = (L(f_ba) >> sh) & 0xffff;
    if (nb_ba) tl_put_bits48(frame, 48 + ((L(o_ba) >> sh) & 0xffff), (uint64_t)ba, nb_ba);
    if (ba) {
        const unsigned si = w.scfsi[c][sb];
        tl_put_bits48(frame, (c ? p_sel[1] : p_sel[0]) + ((L(o_sel) >> sh) & 0xffff), si, 2);
        const unsigned s0 = (unsigned)L(scf)[0], s1 = (unsigned)L(scf)[1], s2 = (unsigned)L(scf)[2];
        const unsigned f3 = (s0 << 12) | (s1 << 6) | s2, f2 = (s0 << 6) | s2;
        tl_put_bits48(frame, (c ? p_scf[1] : p_scf[0]) + ((L(o_scf) >> sh) & 0xffff), si == 0 ? f3 : si == 2 ? s0 : f2, (L(f_scf) >> sh) & 0xffff);
    }
    TL_LANES_END
    TL_PRIO2(TL_PS_Q);
    // quantise (encode_new.c:479-547) + write_samples_new (:560-598): 12 rounds of 3 samples
    {
        PV(int, q_ba); PV(int, q_nb); PV(int, q_grp); PV(int, q_s2n); PV(int, q_steps); PV(int, q_pos); PV(int, q_rstep);
        PV(double, q_a); PV(double, q_b); PV(double, q_s2nf); PA(double, q_sf, 3); PA(double, q_rsf, 3);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1, sh = 16 * c;
        const int ba = sb < sblimit ? w.balloc[c][sb] : 0;
        const unsigned qi = ba ? B->qinfo_line[L(a_ln2)][ba] : 0u;
        const int q = (int)(qi & 31u);
        L(q_ba) = ba; L(q_nb) = (int)((qi >> 5) & 31u); L(q_grp) = ((qi >> 10) & 1u) ? 3 : 1; L(q_s2n) = K->steps2n[q];
        L(q_steps) = ((qi >> 10) & 1u) ? 1 << L(q_nb) : K->steps[q];                                // M of the field's Horner form (tl_encode_frame)
        L(q_a) = K->qa[q]; L(q_b) = K->qb[q]; L(q_s2nf) = K->steps2n_f[q];
        for (int gr = 0; gr < 3; gr++) {
            L(q_sf)[gr] = B->scalefactor[L(scf)[gr]];
            L(q_rsf)[gr] = 1.0 / L(q_sf)[gr];
        }
        L(q_pos) = (c ? p_smp[1] : p_smp[0]) + ((L(o_smp) >> sh) & 0xffff);
        L(q_rstep) = c ? n_smp[1] : n_smp[0];
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int r = 0; r < 12; r++) {
            const int gr = r >> 2, j0 = (r & 3) * 3;
            TL_LANES_BEGIN
            if (L(q_ba)) {
                const double sfv = L(q_sf)[gr], rsf = L(q_rsf)[gr];
                unsigned v[3];
                for (int x = 0; x < 3; x++) {
                    double d = tl_div_by(L(smp)[gr * 12 + j0 + x], sfv, rsf);                // == s / sfv (encode_new.c:507,511)
                    d = d * L(q_a) + L(q_b);
                    const bool neg = !(d >= 0);                                            // encode_new.c:528-534
                    d += TL_SELECT(neg, 1.0, 0.0);
                    const unsigned qv = (unsigned)(d * L(q_s2nf));
                    v[x] = qv | (neg ? 0u : (unsigned)L(q_s2n));
                }
                const int nb = L(q_nb);
                const bool three = L(q_grp) == 3;
                const unsigned M = (unsigned)L(q_steps);
                const unsigned fa = TL_SELECT(three, v[2], v[0]), fc = TL_SELECT(three, v[0], v[2]);
                const unsigned inner = v[1] + M * fc;
                tl_put_bits48(w.u.frame[lane & 1], L(q_pos) + r * L(q_rstep), (uint64_t)fa + (uint64_t)M * (uint64_t)inner, L(q_grp) * nb);
            }
            TL_LANES_END
        }
    }
    TL_PRIO2(1);
    // ---- CRC-16 (crc.c:12-41) of both frames: lanes 0..31 fold unit 0's message bytes, lanes 32..63 unit 1's (a mono frame protects
    //      at most 16 + 94 + 60 bits: 22 bytes; the preset's two bytes ride on lanes 30/31 of each half) ----
    {
        PV(uint32_t, part0); PV(uint32_t, part1);
        TL_LANES_BEGIN
        const int u = lane >> 5, l5 = lane & 31;
        const uint32_t *frame = w.u.frame[u];
        const int n = 16 + ((u ? p_scf[1] : p_scf[0]) - 48);
        uint32_t acc = 0;
        const bool preset = l5 >= 30;
        const int first = 8 * l5;
        if (first < n || preset) {
            const int byte = l5 < 2 ? l5 + 2 : l5 + 4;
            const int cnt = preset ? 8 : (n - first < 8 ? n - first : 8);
            const int e0 = preset ? n + 8 * (31 - l5) : 16 + (n - first - cnt);
            const uint16_t *xt = &K->crc_xpow[e0];                   // as in tl_encode_frame: the power table instead of stepping x^e
            const unsigned v = preset ? 0xffu : ((frame[byte >> 2] >> (24 - 8 * (byte & 3))) & 0xffu) >> (8 - cnt);
            unsigned x8[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 8; k++) x8[k] = xt[k];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 8; k++) acc ^= (0u - ((v >> k) & 1u)) & x8[k];
        }
        L(part0) = u == 0 ? acc : 0u; L(part1) = u == 1 ? acc : 0u;
        TL_LANES_END
        const unsigned crc0 = TL_WAVE_XOR_U32(part0) & 0xffffu, crc1 = TL_WAVE_XOR_U32(part1) & 0xffffu;
        TL_LANES_BEGIN
        if (lane < 2) tl_put_bits(w.u.frame[lane], 32, lane ? crc1 : crc0, 16);
        TL_LANES_END
    }
    // ---- ScF-CRC (crc.c:58-97, toolame.c:527-542), both units through ONE sum scan and ONE XOR scan (unit u in bits 16u..16u+15) ----
    {
        PV(int, rlen); PV(uint32_t, rcrc); PV(int, rl2); PV(int, lex);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        uint32_t rec = 0;
        if (sb < sblimit && w.balloc[c][sb]) {
            const uint32_t s0 = (uint32_t)L(scf)[0] >> 3, s1 = (uint32_t)L(scf)[1] >> 3, s2 = (uint32_t)L(scf)[2] >> 3;
            switch (w.scfsi[c][sb]) {
            case 0: rec = (9u << 16) | (s0 << 6) | (s1 << 3) | s2; break;
            case 1: case 3: rec = (6u << 16) | (s0 << 3) | s2; break;
            default: rec = (3u << 16) | s0; break;
            }
        }
        L(rlen) = (int)(rec >> 16); L(rcrc) = rec & 0x1ffu; L(rl2) = L(rlen) << (16 * c);
        TL_LANES_END
        TL_WAVE_EXSCAN_I32(lex, rl2);
        const int f[5] = {0, 4, 8, 16, 30};
        int gend[4], gfirst[4], glast[4];
        for (int g = 0; g < 4; g++) {
            gfirst[g] = f[g]; glast[g] = f[g + 1] > sblimit ? sblimit : f[g + 1];
            gend[g] = (g < C->dab_ext && glast[g] > gfirst[g]) ? TL_READLANE_I32(lex, 2 * glast[g]) : 0;      // both units' sums, packed; lane <= 60
        }
        PV(uint32_t, part); PV(uint32_t, pscan);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1, sh = 16 * c;
        const int g = sb < 4 ? 0 : sb < 8 ? 1 : sb < 16 ? 2 : 3;
        const int ge = ((g == 0 ? gend[0] : g == 1 ? gend[1] : g == 2 ? gend[2] : gend[3]) >> sh) & 0xffff;
        const int after = ge - ((L(lex) >> sh) & 0xffff) - L(rlen);
        const int e0 = after + 8;
        unsigned xp = K->crc8_xpow[e0 < 0 ? 0 : e0 > 319 ? 319 : e0];
        unsigned acc = 0;
        const unsigned rb = L(rcrc);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int b = 0; b < 9; b++) {
            acc ^= ((rb >> b) & 1u) ? xp : 0u;
            xp = ((xp << 1) & 0xffu) ^ ((xp & 0x80u) ? 0x1Du : 0u);
        }
        L(part) = (L(rlen) && sb < sblimit) ? acc << sh : 0u;
        TL_LANES_END
        TL_WAVE_INCL_XSCAN_U32(pscan, part);
        unsigned c8g[4];                                             // per group: unit 0's CRC in bits 0..7, unit 1's in bits 16..23
        for (int g = 0; g < 4; g++) {
            c8g[g] = 0;
            if (g < C->dab_ext && glast[g] > gfirst[g]) {
                c8g[g] = (unsigned)TL_READLANE_I32(pscan, 2 * glast[g] - 1);
                if (gfirst[g] > 0) c8g[g] ^= (unsigned)TL_READLANE_I32(pscan, 2 * gfirst[g] - 1);
            }
        }
        TL_LANES_BEGIN
        const int u = lane >> 5, l5 = lane & 31;
        if (l5 < C->dab_ext) {
            const int grp = C->dab_ext - 1 - l5;                     // transmission order: i = dab_ext-1 .. 0
            const unsigned c8 = ((grp == 0 ? c8g[0] : grp == 1 ? c8g[1] : grp == 2 ? c8g[2] : c8g[3]) >> (16 * u)) & 0xffu;
            const int tail = (u ? lg_frame[1] : lg_frame[0]) - 2 - C->dab_ext;
            tl_put_bits(w.u.frame[u], (tail + l5) * 8, c8, 8);
            w.ncentre[4 * u + l5] = (int16_t)c8;
        }
        TL_LANES_END
    }
    // ---- X-PAD + F-PAD bytes (toolame.c:515-524,544-551), straight from the launch's X-PAD records ----
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++)
        if (xpad_len[u]) {
            const int xl = xpad_len[u], xstart = lg_frame[u] - C->dab_ext - xl;
            TL_LANES_BEGIN
            for (int i = lane; i < xl; i += 64) {
                const int bytepos = i < xl - 2 ? xstart + i : lg_frame[u] - 2 + (i - (xl - 2));
                tl_put_bits(w.u.frame[u], bytepos * 8, xpad_src[u][i], 8);
            }
            TL_LANES_END
        }
    // ---- emit: each unit files its frame and its ScF-CRC (tl_finish_stream puts the CRC into the frame before it) ----
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++) {
        const int nwords = (lg_frame[u] + 3) >> 2;
        const uint32_t *frame = w.u.frame[u];
        TL_LANES_BEGIN
        for (int i = lane; i < nwords; i += 64) {
            if (fo[u].words) fo[u].words[i] = frame[i];
            else {
                const uint32_t le = tl_bswap(frame[i]);
                const int rem = lg_frame[u] - 4 * i;
                if (rem >= 4) ((uint32_t *)fo[u].bytes)[i] = le;
                else for (int b = 0; b < rem; b++) fo[u].bytes[4 * i + b] = (uint8_t)(le >> (8 * b));
            }
        }
        if (lane < 4) fo[u].scfcrc[lane] = lane < C->dab_ext ? (uint8_t)w.ncentre[4 * u + lane] : 0;
        TL_LANES_END
    }
    TL_PRIO2(0);
}
