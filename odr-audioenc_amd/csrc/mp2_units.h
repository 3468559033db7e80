// mp2_units.h -- units of work of the kernels: slot recurrence, PCM views, psy / psy-2 / encode / frame units, the finish pass.
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ------------------------------------------------------------------------------------------
// The configuration index of stream s.  A batch whose streams all share ONE configuration (a homogeneous fleet: the common case) passes no
// table: the index is 0 and a unit does not begin with a load that everything else about its configuration depends on.
TL_FN int tl_cfg_index(const TlLaunch &A, int s) { return A.stream_cfg ? A.stream_cfg[s] : 0; }
// One step of the padding recurrence (availbits.c:49-62): does the next frame carry a padding slot?  fp64 as in the reference.
TL_FN int tl_slot_step(double &lag, double frac)
{
    if (frac == 0) return 0;
    if (lag > (frac - 1.0)) { lag -= frac; return 0; }
    lag += (1 - frac);
    return 1;
}
// Split path, 44.1 / 22.05 kHz only: the recurrence is sequential, the frames are not -- so one lane per stream runs it over
// the launch's frames first and leaves every frame's padding bit for the units (and the state after the launch for the finish pass).
TL_FN void tl_slots_stream(const TlLaunch &A, int s)
{
    const double frac = A.configs[tl_cfg_index(A, s)].pad_frac;
    double lag = A.state[s].slot_lag;
    for (int f = 0; f < A.nframes; f++) A.padbits[(size_t)f * (size_t)A.nstreams + (size_t)s] = (uint8_t)tl_slot_step(lag, frac);
    A.newlag[s] = lag;
}

// A stream's PCM around frame f of a launch: the frame itself and the 480 samples per channel before it (the stream state
// on the first frame of a launch, the previous input frame after).
TL_FN TlPcmView tl_pcm_view(const TlLaunch &A, const TlStreamState *st, int s, int f)
{
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    TlPcmView pv;
    pv.cur[0] = A.pcm + slot * 2304; pv.cur[1] = pv.cur[0] + 1152;
    pv.hist[0] = f == 0 ? &st->hist[0][0] : A.pcm + (slot - (size_t)A.nstreams) * 2304 + (1152 - TL_HIST);
    pv.hist[1] = f == 0 ? &st->hist[1][0] : pv.hist[0] + 1152;
    return pv;
}

// The same for a PAIR of mono streams sharing a wave: "channel" u is channel 0 of stream s[u]
TL_FN TlPcmView tl_pcm_view_pair(const TlLaunch &A, int s0, int s1, int f)
{
    const TlPcmView a = tl_pcm_view(A, &A.state[s0], s0, f), b = tl_pcm_view(A, &A.state[s1], s1, f);
    TlPcmView pv;
    pv.cur[0] = a.cur[0]; pv.hist[0] = a.hist[0]; pv.cur[1] = b.cur[0]; pv.hist[1] = b.hist[0];
    return pv;
}

// One unit of the psy kernel (models 1 and 3): both channels of frame f of stream s -> A.psy_out[f][s].  The model reads
// nothing but PCM (the window of a frame: the last 192 samples before it and its first 832), so units are independent of each
// other -- of other streams AND of other frames of the same stream -- and the kernel runs them in any order on any wave.
template <int PSY, int NCH = 0>     // NCH = 2: every stream of the launch's list is a two-channel stream (the kernel variant of all-stereo lists: no look at the configuration record before the first transform)
TL_FN void tl_psy_unit(TlPsyLds &w, const double *TL_RESTRICT db, const TlLaunch &A, int s, int f, PARGA(double, rec, 4), int s2 = -1, int *ci_out = nullptr)
{   // s2 >= 0: a PAIR of mono streams of one configuration -- the model runs its two-channel form on channel 0 of s and of s2
    // rec: the model's result per subband, in the registers of lane = subband: [ch] the level that competes with the scalefactor
    // level, [2 + ch] the minimum masking threshold (SMR = max(level, scale_db[min scalefactor index]) - threshold is the encoder's
    // line: psycho_1.c:568-581 with level = spike level; psycho_3.c:163-183,409-432 with level = strongest line of the subband)
    const TlTables *T = A.tables;
    const int ci = TL_UNI_I(tl_cfg_index(A, s));                        // (uniform by construction; said so, the record's address is a scalar and its rows are read base + lane offset)
    if (ci_out) *ci_out = ci;                                                 // the encoder phase of the same unit takes it from here (one round trip less when it starts)
    const TlConfig *C = &A.configs[ci];
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    const TlPcmView pv = s2 >= 0 ? tl_pcm_view_pair(A, s, s2, f) : tl_pcm_view(A, &A.state[s], s, f);
    TL_LANES_BEGIN
    L(rec)[0] = 0.0; L(rec)[1] = 0.0; L(rec)[2] = 0.0; L(rec)[3] = 0.0;     // the model writes every subband of the channels it runs
    TL_LANES_END
#ifdef TL_NO_PSY_STAMPS
    long long *sp = nullptr;
#else
    long long *sp = A.stamps ? A.stamps + slot * 32 : nullptr;
#endif
    TL_STAMP(sp, 15);                                                 // unit begin (slots 8..14 / 16..22: the channels' stages, 24..30: FHT passes)
    // The unit's PCM is its first touch of HBM, and the model gets to its loads only after the configuration record has come in (two dependent
    // round trips: record index, then the record's rows).  One load per 128-byte line of both channels' analysis windows is issued HERE, so
    // that the lines travel while those round trips are under way: lanes 0..12 / 16..28 the frame's first 832 samples, 13..15 / 29..31 the
    // last 192 of the history.  The values are not used (TL_KEEP after the model: the register is simply released there).
    PV(int, touch);
    TL_LANES_BEGIN
    {
        const int c = (lane >> 4) & 1, j = lane & 15;
        const int16_t *p = j < 13 ? (c ? pv.cur[1] : pv.cur[0]) + 64 * j : (c ? pv.hist[1] : pv.hist[0]) + (TL_HIST - 192) + 64 * (j - 13);
        L(touch) = lane < 32 ? (int)*p : 0;                              // (a mono stream's second half of the slot is allocated too: read, not used)
    }
    TL_LANES_END
    if constexpr (TL_EXP_LEVEL >= 9) { }                              // diagnostic build: no model at all (tools/class_budget.sh: what the encoder phase alone issues)
    else if constexpr (PSY == 1) {
        if (NCH == 2 || s2 >= 0 || C->nch == 2) tl_psy1_stereo(w, T, db, C, pv, rec, sp);
        else tl_psy1(w, T, db, C, pv, 0, rec, sp ? sp + 8 : nullptr);
    } else {
        if (NCH == 2 || s2 >= 0 || C->nch == 2) tl_psy3_stereo(w, T, db, C, pv, rec, sp);
        else tl_psy3(w, T, db, C, pv, 0, rec, sp ? sp + 8 : nullptr);
    }
    TL_LANES_BEGIN TL_KEEP(L(touch)); TL_LANES_END
    TL_STAMP(sp, 23);                                                 // unit end
}

// Models 2 and 4 on the split path.  One unit = frames [f0, f1) of ONE channel of one stream, in order (the two channels of a
// stream share nothing).  The r/phi prediction state of the run lives in the wave's registers (tl_psy2_pass).  Where a run starts
// at the launch's first frame the state comes from the stream's record (what the previous launch left; the passes before it
// are PCM this launch cannot see); anywhere else two seed passes over frame f0 - 1 rebuild it.  The run that ends the launch
// leaves the state for the next one -- in the OTHER of the record's two copies, so that it can never be read by a run of the
// same launch that starts at frame 0 and is scheduled later.  It leaves the SMR itself in TlPsyOut::a (the model's last line
// needs no scalefactors).
TL_FN void tl_psy2_chain(TlPsy2Lds &w, const TlLaunch &A, int s, int ch, int f0, int f1, const uint64_t *sct)
{
    const TlConfig *C = &A.configs[TL_UNI_I(tl_cfg_index(A, s))];    // (uniform by construction: said so, the record's and the tables' addresses are scalars)
    if (ch >= TL_UNI_I(C->nch) || f0 >= f1) return;
    const TlPsy2Tables *P = &A.psy2_tables[TL_UNI_I(C->psy2_tab)];
    PA(double, r1, 8); PA(double, r2, 8); PA(double, p1, 8); PA(double, p2, 8); PV(double, snr0);
    double *l5 = TL_P2_L512(w);
    const TlPsy2State *S0 = nullptr;
    if (f0 == 0) {
        // the run starts with the stream's carried state: its first pass fetches it, behind the transform's own loads (tl_psy2_pass)
        S0 = &A.psy2_state[2 * (size_t)s + (size_t)A.psy2_flip];
        TL_LANES_BEGIN
        L(snr0) = 0.0;
        TL_LANES_END
    } else {
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) { L(r1)[it] = 0.0; L(r2)[it] = 0.0; L(p1)[it] = 0.0; L(p2)[it] = 0.0; }
        if (lane == 0) { l5[0] = 0.0; l5[1] = 0.0; l5[2] = 0.0; l5[3] = 0.0; }
        L(snr0) = 0.0;
        TL_LANES_END
        const TlPcmView pv = tl_pcm_view(A, &A.state[s], s, f0 - 1);
        tl_psy2_pass<true>(w, A.tables, P, pv, ch, 0, r1, r2, p1, p2, snr0, nullptr, sct, nullptr);
        tl_psy2_pass<true>(w, A.tables, P, pv, ch, 1, r1, r2, p1, p2, snr0, nullptr, sct, nullptr);
    }
    for (int f = f0; f < f1; f++) {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
        const TlPcmView pv = tl_pcm_view(A, &A.state[s], s, f);
        long long *sp = A.stamps ? A.stamps + slot * 32 + 8 + 8 * ch : nullptr;
        tl_psy2_pass<false>(w, A.tables, P, pv, ch, 0, r1, r2, p1, p2, snr0, &A.psy_out[slot].a[ch][0], sct, sp, f == f0 ? S0 : nullptr);
        tl_psy2_pass<false>(w, A.tables, P, pv, ch, 1, r1, r2, p1, p2, snr0, &A.psy_out[slot].a[ch][0], sct, nullptr);
    }
    if (f1 == A.nframes) {
        TlPsy2State *S = &A.psy2_state[2 * (size_t)s + (size_t)(1 - A.psy2_flip)];
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) {
            const int j = lane + 64 * it;
            S->r[ch][0][j] = L(r1)[it]; S->r[ch][1][j] = L(r2)[it]; S->phi[ch][0][j] = L(p1)[it]; S->phi[ch][1][j] = L(p2)[it];
        }
        if (lane == 0) { S->r[ch][0][512] = l5[0]; S->r[ch][1][512] = l5[1]; S->phi[ch][0][512] = l5[2]; S->phi[ch][1][512] = l5[3]; }
        TL_LANES_END
    }
}
// Unit u of the psy-2 kernel's work list -> (chain, first frame, end frame).  The launch's chains (TlLaunch::chain_list: first
// channels of the list's streams, then the second channels of its stereo streams) are dealt to the waves longest first: chains
// [0, p2_nwhole) as ONE unit each, every chain after them cut into p2_k runs of p2_plen frames -- so that the last round of
// waves is as full as the ones before it (the host picks the cut, tl_psy2_plan in mp2_host.cpp).
TL_FN bool tl_psy2_unit(const TlLaunch &A, int u, int &chain, int &f0, int &f1)
{
    if (u < A.p2_nwhole) { chain = u; f0 = 0; f1 = A.nframes; return true; }
    const int v = u - A.p2_nwhole, k = A.p2_k;
    chain = A.p2_nwhole + v / k;
    f0 = (v - (v / k) * k) * A.p2_plen;
    f1 = f0 + A.p2_plen < A.nframes ? f0 + A.p2_plen : A.nframes;
    return f0 < f1;
}

// [history | frame] -> LDS in 8-byte pieces, 120 + 288 per channel.  All of a lane's loads are issued before the first LDS
// write so the HBM latency is paid once per frame, not once per piece.
// The pieces are dealt run by run -- history and frame of each channel, four runs with ONE base address each -- not as one index space
// over all of them: piece lane + 64 it of a run is a load at base + 8 lane + 512 it (an immediate), stored at the run's LDS offset + the same,
// and only a run's last, partial trip is predicated.  (One index space straddles the run boundaries inside every trip: a pointer select,
// a bounds test and an execution-mask region per piece -- 290 vector and 200 scalar instructions per frame for 26 copies.)
TL_FN void tl_stage_pcm(TlMainLds &w, const TlPcmView &pv, int nch)
{
    TL_LANES_BEGIN
    {
        constexpr int HP = TL_HIST / 4, CP = 1152 / 4;                      // 120 + 288 pieces per channel
        constexpr int HT = (HP + 63) / 64, CT = (CP + 63) / 64;             // trips per run: 2 and 5, the last ones partial (56 and 32 lanes)
        uint64_t vh[2][HT], vc[2][CT];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int c = 0; c < 2; c++) {
            if (c >= nch) break;                                            // (wave-uniform)
            const uint64_t *h = (const uint64_t *)(c ? pv.hist[1] : pv.hist[0]), *q = (const uint64_t *)(c ? pv.cur[1] : pv.cur[0]);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int it = 0; it < HT; it++) { vh[c][it] = 0; if (64 * it + 64 <= HP || lane < HP - 64 * it) vh[c][it] = h[lane + 64 * it]; }
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int it = 0; it < CT; it++) { vc[c][it] = 0; if (64 * it + 64 <= CP || lane < CP - 64 * it) vc[c][it] = q[lane + 64 * it]; }
        }
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int c = 0; c < 2; c++) {
            if (c >= nch) break;
            uint64_t *d = (uint64_t *)&w.u.fbk.pcm[c][0];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int it = 0; it < HT; it++) if (64 * it + 64 <= HP || lane < HP - 64 * it) d[lane + 64 * it] = vh[c][it];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int it = 0; it < CT; it++) if (64 * it + 64 <= CP || lane < CP - 64 * it) d[HP + lane + 64 * it] = vc[c][it];
        }
    }
    TL_LANES_END
}
// X-PAD bytes of a slot -> LDS; returns the length the frame carries.  The contract is 0 or 2..pad_len (toolame.c:515-516,
// odr-audioenc.cpp:803,830-834); anything else -- more than the stream's toolame_set_pad() length, more than the record
// holds -- is treated as "no PAD this frame" (tl_build_config has made sure that pad_len itself fits into the frame).
TL_FN int tl_stage_xpad(TlMainLds &w, const TlLaunch &A, const TlConfig *C, size_t slot)
{
    if (!A.xpad_len) return 0;
    int xl = A.xpad_len[slot];
    if (xl < 2 || xl > TL_MAX_XPAD || xl > C->dab_length) xl = 0;
    TL_LANES_BEGIN
    for (int i = lane; i < xl; i += 64) w.xpad[i] = A.xpad[slot * TL_MAX_XPAD + i];
    TL_LANES_END
    return xl;
}

// ------------------------------------------------------------------------------------------
// Encode kernel: one unit = frame f of stream s.  Like the psy kernel's units these are
// independent of each other: the filterbank's history is PCM (the previous input frame, or the stream state before frame 0),
// the SMR comes from the psy kernel's record, and the one thing a frame owes its predecessor -- its ScF-CRC, which travels in
// the frame before (toolame.c:527-542) -- is filed aside and put in place by tl_finish_stream.
template <int PSY, int NCH = 0>     // TL_PSY_EXT: SMR from the psy kernel's record (models 1 and 3); 2: the psy-2 kernel's SMR (models 2 and 4); 0: model 0, which needs nothing but this frame's scalefactors
TL_FN void tl_main_unit(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, const TlLaunch &A, int s, int f, int ci = -1)
{   // ci >= 0: the stream's configuration index, known to the caller (tl_frame_unit: the model phase has read it)
    const TlConfig *C = &A.configs[ci >= 0 ? ci : TL_UNI_I(tl_cfg_index(A, s))];
    TlStreamState *st = &A.state[s];
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    const TlPcmView pv = tl_pcm_view(A, st, s, f);
#ifdef TL_NO_MAIN_STAMPS
    long long *sp = nullptr;
#else
    long long *sp = A.stamps ? A.stamps + slot * 32 : nullptr;
#endif
    TL_STAMP(sp, 31);
    tl_stage_pcm(w, pv, NCH ? NCH : C->nch);
    const int xl = tl_stage_xpad(w, A, C, slot);
    TlFrameOut fo;
    fo.bytes = f + 1 < A.nframes ? A.out + (slot + (size_t)A.nstreams) * (size_t)A.out_stride : nullptr;     // waits in the next slot
    fo.words = f + 1 < A.nframes ? nullptr : A.newpend + (size_t)s * TL_MAX_FRAME_WORDS;
    fo.scfcrc = A.scfcrc + slot * 4;
    const int padding = A.padbits ? (int)A.padbits[slot] : 0;
    tl_encode_frame<PSY, NCH>(w, A.tables, B, C, PSY == 2 ? &A.psy_out[slot] : nullptr, pv, xl, fo, enw_s, K, padding,
                                A.taps ? &A.taps[slot] : nullptr, sp);
}

// The same unit for a PAIR of mono streams s0, s1 of one configuration (TlLaunch::partner): frame f of both by one wave (tl_encode_pair)
template <int PSY>
TL_FN void tl_main_pair(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, const TlLaunch &A, int s0, int s1, int f)
{
    const TlConfig *C = &A.configs[tl_cfg_index(A, s0)];
    const int ss[2] = {s0, s1};
    tl_stage_pcm(w, tl_pcm_view_pair(A, s0, s1, f), 2);
    TlFrameOut fo[2];
    const TlPsyOut *po[2];
    const uint8_t *xsrc[2];
    int xl[2], padding[2];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++) {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)ss[u];
        fo[u].bytes = f + 1 < A.nframes ? A.out + (slot + (size_t)A.nstreams) * (size_t)A.out_stride : nullptr;
        fo[u].words = f + 1 < A.nframes ? nullptr : A.newpend + (size_t)ss[u] * TL_MAX_FRAME_WORDS;
        fo[u].scfcrc = A.scfcrc + slot * 4;
        po[u] = PSY == 2 ? &A.psy_out[slot] : nullptr;
        padding[u] = A.padbits ? (int)A.padbits[slot] : 0;
        int x = A.xpad_len ? A.xpad_len[slot] : 0;                  // the contract of tl_stage_xpad
        if (x < 2 || x > TL_MAX_XPAD || x > C->dab_length) x = 0;
        xl[u] = x; xsrc[u] = A.xpad ? A.xpad + slot * TL_MAX_XPAD : nullptr;
    }
    tl_encode_pair<PSY>(w, B, C, po, xl, xsrc, fo, enw_s, K, padding);
}

// Models 1 and 3: one unit = frame f of stream s, psy model first, then the encoder, by the same wave.  The two phases
// share the wave's LDS block (a union: the model's arrays are dead when the encoder starts) and nothing else but the
// model's record, 4 values per subband, which waits in registers until the model is done.
union TlFrameLds { TlPsyLds p; TlMainLds m; };
template <int PSY, int NCH = 0>
TL_FN void tl_frame_unit(TlFrameLds &w, const double *TL_RESTRICT db, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s,
                         const TlPackTables *TL_RESTRICT K, const TlLaunch &Apsy, TL_KARG Amain_p, int s, int f, int s2 = -1)
{   // s2 >= 0: frame f of the two mono streams s and s2 (one configuration) as the two "channels" of the wave
    PA(double, rec, 4);
    int ci = -1;
    tl_psy_unit<PSY, NCH>(w.p, db, Apsy, s, f, rec, s2, &ci);
    TL_SYNC();
    // The encoder phase reads the launch record afresh (device: scalar loads from the kernel-argument segment, issued HERE) and
    // re-derives its pointers from laundered copies of s / f / s2: nothing of the model phase's scalar state stays live across the
    // phases, and nothing of the encoder's is loaded before the model has run.
    TL_LAUNDER(Amain_p); TL_LAUNDER(s); TL_LAUNDER(f); TL_LAUNDER(s2); TL_LAUNDER(ci);
    const TlLaunch Amain = *Amain_p;
    // the model's arrays are dead: its record goes where the encoder expects it (its own SMR array and the one beside it)
    TL_LANES_BEGIN
    if (lane < 32) {
        w.m.smr[0][lane] = L(rec)[0]; w.m.smr[1][lane] = L(rec)[1];
        w.m.psy_m[0][lane] = L(rec)[2]; w.m.psy_m[1][lane] = L(rec)[3];
    }
    TL_LANES_END
    if (s2 >= 0) tl_main_pair<TL_PSY_EXT>(w.m, B, enw_s, K, Amain, s, s2, f);
    else tl_main_unit<TL_PSY_EXT, NCH>(w.m, B, enw_s, K, Amain, s, f, ci);
}

// Which stream shares a wave with stream s?  TlLaunch::partner[s]: the other mono stream of s's configuration it is paired with, or -1.
// The lower-numbered stream of a pair runs the unit for both (returns true, s2 = the partner), the higher one has nothing to do
// (returns false).  Launches with stage taps or cycle stamps (diagnostics, per frame of one stream) run every stream alone.
TL_FN bool tl_unit_partner(const TlLaunch &A, int s, int &s2)
{
    s2 = -1;
#ifdef TL_NO_PAIRS
    return true;                                                      // diagnostic build: every stream alone (what pairing is measured against)
#endif
    if (!A.partner || A.taps || A.stamps) return true;
    const int p = A.partner[s];
    if (p < 0) return true;
    if (p < s) return false;
    s2 = p;
    return true;
}

// After the units of a launch: for stream s, hand out the frame that was pending before the launch (slot 0), store every
// frame's ScF-CRC into the frame before it, make the launch's last frame the pending one, roll the PCM history forward.
TL_FN void tl_finish_stream(const TlLaunch &A, int s)
{
    const TlConfig *C = &A.configs[tl_cfg_index(A, s)];
    TlStreamState *st = &A.state[s];
    const int whole = C->frame_bytes, dab_ext = C->dab_ext, nch = C->nch;
    const bool have_prev = st->frames_done > 0;
    const int prev_len = st->pending_len;
    uint8_t *out0 = A.out + (size_t)s * (size_t)A.out_stride;
    // slot 0: the frame that was pending before the launch, with the ScF-CRC of the launch's first frame
    TL_LANES_BEGIN
    if (have_prev)
        for (int i = lane; i < ((prev_len + 3) >> 2); i += 64) {
            const uint32_t le = tl_bswap(st->pending[i]);
            const int rem = prev_len - 4 * i;
            if (rem >= 4) ((uint32_t *)out0)[i] = le;
            else for (int b = 0; b < rem; b++) out0[4 * i + b] = (uint8_t)(le >> (8 * b));
        }
    TL_LANES_END
    // slot f holds frame f-1 (slot 0: the old pending frame); frame f's ScF-CRC goes 2 + dab_ext bytes before the END of the
    // frame in slot f -- whose length (a padding slot more or less at 44.1 / 22.05 kHz) comes from the slot recurrence
    TL_LANES_BEGIN
    for (int f = lane; f < A.nframes; f += 64) {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
        const int len = f > 0 ? whole + (A.padbits ? (int)A.padbits[slot - (size_t)A.nstreams] : 0) : prev_len;
        if (f > 0 || have_prev) {
            uint8_t *o = A.out + slot * (size_t)A.out_stride + (len - 2 - dab_ext);
            for (int k = 0; k < dab_ext; k++) o[k] = A.scfcrc[slot * 4 + k];
        }
        if (A.out_len) A.out_len[slot] = (f > 0 || have_prev) ? len : 0;
    }
    TL_LANES_END
    const int last_len = whole + (A.padbits ? (int)A.padbits[(size_t)(A.nframes - 1) * (size_t)A.nstreams + (size_t)s] : 0);
    TL_LANES_BEGIN
    for (int i = lane; i < ((last_len + 3) >> 2); i += 64) st->pending[i] = A.newpend[(size_t)s * TL_MAX_FRAME_WORDS + i];
    TL_LANES_END
    {
        const int16_t *last = A.pcm + ((size_t)(A.nframes - 1) * (size_t)A.nstreams + (size_t)s) * 2304;
        TL_LANES_BEGIN
        for (int i = lane; i < (TL_HIST / 2) * 2; i += 64) {
            const int ch = i / (TL_HIST / 2), k = (i % (TL_HIST / 2)) * 2;
            *(uint32_t *)&st->hist[ch][k] = ch < nch ? *(const uint32_t *)(last + ch * 1152 + (1152 - TL_HIST) + k) : 0u;
        }
        if (lane == 0) { st->frames_done += A.nframes; st->pending_len = last_len; if (A.padbits) st->slot_lag = A.newlag[s]; }
        TL_LANES_END
    }
}
