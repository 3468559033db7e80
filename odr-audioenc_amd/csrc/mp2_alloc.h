// mp2_alloc.h -- bit allocation (encode_new.c:1061-1187): tl_allocate, tl_allocate_pair.
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// a_bit_allocation_new (encode_new.c:1078-1187) for the cells of the wave: the one or two channels of a stream, joint pairs included.
// adb: the frame's bits after header extension and PAD (toolame.c:292-301).  Returns the bits left over.
TL_FN int tl_allocate(const TlBlockShared *TL_RESTRICT B, int adb, int nch, int sblimit, int jsbound, PARG(int, a_ln), PARG(int, a_nbal),
                      PARG(int, a_sfs), PARG(int, a_sfs_o), PARG(double, a_smr), PARG(int, ba))
{
    PV(uint64_t, ukey); PV(uint64_t, ukey2); PV(int, nbits); PV(int, cost); PV(int, cost2);
    PV(int, jpair);                                             // lane belongs to a joint-coded pair (steps with its partner)
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = c < nch && sb < sblimit;
    const int maxa = (1 << L(a_nbal)) - 1;
    L(ukey) = live ? tl_mnr_key(B->snr_line[L(a_ln)][0] - L(a_smr)) : ~0ull;
    L(ukey2) = (live && 1 < maxa) ? tl_mnr_key(B->snr_line[L(a_ln)][1] - L(a_smr)) : ~0ull;
    L(ba) = 0;
    L(nbits) = (sb < sblimit && c < (sb < jsbound ? nch : 1)) ? L(a_nbal) : 0;
    // first step of a cell: samples + scfsi + scalefactors (both channels above jsbound), :1139-1147
    L(cost) = live ? B->bits12_line[L(a_ln)][1] + 2 + L(a_sfs) + ((nch == 2 && sb >= jsbound) ? 2 + L(a_sfs_o) : 0) : 0;
    L(cost2) = live ? B->bits12_line[L(a_ln)][2] - B->bits12_line[L(a_ln)][1] : 0;
    L(jpair) = (live && nch == 2 && sb >= jsbound) ? 1 : 0;
    TL_LANES_END
    const int bbal = TL_WAVE_SUM_I32(nbits);
    const int ad = adb - (bbal + 16 + 32);
    int spent = 0;                                              // bspl + bscf + bsel
    const bool any_pair = nch == 2 && jsbound < sblimit;
    for (; TL_ENC_LEVEL < 4;) {                                 // rounds
        PV(uint64_t, keff); PV(uint64_t, k2eff);
        TL_LANES_BEGIN L(keff) = L(ukey); L(k2eff) = L(ukey2); TL_LANES_END
        if (any_pair) {                                         // a pair acts at the smaller of its two keys
            PV(uint64_t, ok1); PV(uint64_t, ok2);
            TL_SWAP1_U64(ok1, ukey); TL_SWAP1_U64(ok2, ukey2);
            TL_LANES_BEGIN
            if (L(jpair)) { if (L(ok1) < L(keff)) L(keff) = L(ok1); if (L(ok2) < L(k2eff)) L(k2eff) = L(ok2); }
            TL_LANES_END
        }
        const uint64_t M = TL_WAVE_MIN_U64(k2eff);
        PV(bool, inb); PV(int, bcost);
        TL_LANES_BEGIN
        L(inb) = L(keff) < M;
        L(bcost) = (L(inb) && !(L(jpair) && (lane & 1))) ? L(cost) : 0;     // a pair pays once
        TL_LANES_END
        const uint64_t bm = TL_BALLOT(inb);
        if (bm == 0ull) break;
        const int csum = TL_WAVE_SUM_I32(bcost);
        bool last_round = false;
        if (csum > ad - spent) {
            // The round does not fit as a whole: admit its events up to the first refusal.  Each cell adds up the
            // prices of the round's events that come no later than its own (equal keys count as earlier, which can
            // only shorten the admitted prefix); the prefix sums grow along the greedy order, so the cells whose
            // sum still fits are exactly a prefix of it.  The event-by-event loop below deals with the rest.
            // The test is ONE compare, key of the event < the cell's own key + 1 (the key of a cell with an event is never ~0; a cell
            // without one compares against 0 and its sum is not read).  A cell's own event passes it as well -- that is the cell's own
            // price, so the sum starts at 0; the two cells of a joint pair share key and price (same allocation line, the two
            // scalefactor selections added up either way round), and the pair's one event is the own event of both.
            PV(int, pre); PV(int, kh); PV(int, kl); PV(uint64_t, kb);
            TL_LANES_BEGIN L(pre) = 0; L(kh) = (int)(uint32_t)(L(keff) >> 32); L(kl) = (int)(uint32_t)L(keff); L(kb) = L(keff) + 1; TL_LANES_END
            uint64_t pm = bm;
            while (pm) {
                const int j = __builtin_ctzll(pm);
                pm &= pm - 1;
                const int cj = TL_READLANE_I32(bcost, j);
                if (cj == 0) continue;                          // the non-paying lane of a pair
                const uint64_t kj = ((uint64_t)(uint32_t)TL_READLANE_I32(kh, j) << 32) | (uint32_t)TL_READLANE_I32(kl, j);
                TL_LANES_BEGIN
                if (kj < L(kb)) L(pre) += cj;
                TL_LANES_END
            }
            TL_LANES_BEGIN
            L(inb) = L(inb) && L(pre) <= ad - spent;
            L(bcost) = L(inb) ? L(bcost) : 0;
            TL_LANES_END
            if (TL_BALLOT(inb) == 0ull) break;
            spent += TL_WAVE_SUM_I32(bcost);
            last_round = true;
        } else spent += csum;
        TL_LANES_BEGIN
        if (L(inb)) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
        if (last_round) break;
    }
    for (; TL_ENC_LEVEL < 4;) {                                 // one event at a time
        // maxmnr_new (encode_new.c:1061-1077): smallest mnr, first in (ch, sb) order
        PV(uint64_t, key);
        TL_LANES_BEGIN
        L(key) = L(cost) <= ad - spent ? L(ukey) : ~0ull;
        TL_LANES_END
        const int wl = TL_WAVE_ARGMIN_U64(key);                  // ch 0 first, then ascending sb
        if (wl < 0) break;
        const int min_sb = wl >> 1;
        spent += TL_READLANE_I32(cost, wl);
        const bool joint_pair = (min_sb >= jsbound && nch == 2);
        TL_LANES_BEGIN
        if (lane == wl || (joint_pair && (lane ^ 1) == wl)) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
    }
    return ad - spent;
}

// The same allocation for the two mono streams sharing a wave, BOTH AT ONCE: lane = 2*sb + u owns cell sb of unit u.  Every minimum and
// sum is taken over the 32 lanes of one parity and lands in all of them (TL_PAR_*), so a unit's greedy loop advances on its own state
// (`room` = bits it may still spend, `ph` = still in the rounds) held in its own lanes, and what a unit's lanes compute is what
// tl_allocate(unit = u) computes for it; the wave leaves a loop when neither unit has anything left in it.
TL_FN void tl_allocate_pair(const TlBlockShared *TL_RESTRICT B, int adb0, int adb1, int sblimit, PARG(int, a_ln), PARG(int, a_nbal),
                            PARG(int, a_sfs), PARG(double, a_smr), PARG(int, ba))
{
    PV(uint64_t, ukey); PV(uint64_t, ukey2); PV(int, nbits); PV(int, cost); PV(int, cost2); PV(int, room); PV(int, ph); PV(int, bbal);
    TL_LANES_BEGIN
    const bool live = (lane >> 1) < sblimit;
    const int maxa = (1 << L(a_nbal)) - 1;
    L(ukey) = live ? tl_mnr_key(B->snr_line[L(a_ln)][0] - L(a_smr)) : ~0ull;
    L(ukey2) = (live && 1 < maxa) ? tl_mnr_key(B->snr_line[L(a_ln)][1] - L(a_smr)) : ~0ull;
    L(ba) = 0;
    L(nbits) = live ? L(a_nbal) : 0;
    L(cost) = live ? B->bits12_line[L(a_ln)][1] + 2 + L(a_sfs) : 0;
    L(cost2) = live ? B->bits12_line[L(a_ln)][2] - B->bits12_line[L(a_ln)][1] : 0;
    TL_LANES_END
    TL_PAR_SUM_I32(bbal, nbits);
    TL_LANES_BEGIN
    L(room) = ((lane & 1) ? adb1 : adb0) - (L(bbal) + 16 + 32);
    L(ph) = 0;
    TL_LANES_END
    for (; TL_ENC_LEVEL < 4;) {                                 // rounds, as in tl_allocate
        PV(uint64_t, k2); PV(uint64_t, M); PV(bool, inb); PV(int, bcost); PV(int, csum); PV(bool, part);
        TL_LANES_BEGIN L(k2) = L(ph) == 0 ? L(ukey2) : ~0ull; TL_LANES_END
        TL_PAR_MIN_U64(M, k2);
        TL_LANES_BEGIN
        L(inb) = L(ph) == 0 && L(ukey) < L(M);
        L(bcost) = L(inb) ? L(cost) | 0x10000 : 0;              // price, and one count per event (a round's prices stay far below 2^16)
        TL_LANES_END
        if (TL_BALLOT(inb) == 0ull) break;
        TL_PAR_SUM_I32(csum, bcost);
        TL_LANES_BEGIN L(part) = L(ph) == 0 && (L(csum) & 0xffff) > L(room); TL_LANES_END
        if (TL_BALLOT(part) != 0ull) {                          // a unit's round does not fit as a whole: its events up to the first refusal
            PV(int, pre); PV(int, kh); PV(int, kl); PV(bool, pin);
            TL_LANES_BEGIN
            L(pre) = 0; L(kh) = (int)(uint32_t)(L(ukey) >> 32); L(kl) = (int)(uint32_t)L(ukey); L(pin) = L(inb) && L(part);
            TL_LANES_END
            // Unit 0's events first, then unit 1's.  A cell counts event j when j's key is no later than its own and j belongs to its
            // unit: ONE compare against its own key + 1 (no key of a live cell is ~0) or against 0 for the other unit's events; the
            // cell's own event passes the test too, which is its own price (so the sum starts at 0; only cells with an event are read).
            PV(uint64_t, kb);
            for (int u = 0; u < 2; u++) {
                TL_LANES_BEGIN L(kb) = (lane & 1) == u ? L(ukey) + 1 : 0ull; TL_LANES_END
                uint64_t pm = TL_BALLOT(pin) & (0x5555555555555555ull << u);
                while (pm) {
                    const int j = __builtin_ctzll(pm);
                    pm &= pm - 1;
                    const int cj = TL_READLANE_I32(bcost, j) & 0xffff;
                    const uint64_t kj = ((uint64_t)(uint32_t)TL_READLANE_I32(kh, j) << 32) | (uint32_t)TL_READLANE_I32(kl, j);
                    TL_LANES_BEGIN
                    if (kj < L(kb)) L(pre) += cj;
                    TL_LANES_END
                }
            }
            TL_LANES_BEGIN
            if (L(part)) { L(inb) = L(inb) && L(pre) <= L(room); L(bcost) = L(inb) ? L(bcost) : 0; }
            TL_LANES_END
            TL_PAR_SUM_I32(csum, bcost);
        }
        TL_LANES_BEGIN
        if (L(ph) == 0) {
            L(room) -= L(csum) & 0xffff;
            if ((L(csum) >> 16) == 0 || L(part)) L(ph) = 1;      // nothing admitted, or the partial round was the unit's last
        }
        if (L(inb)) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
    }
    for (; TL_ENC_LEVEL < 4;) {                                 // one event per unit at a time
        PV(uint64_t, key); PV(uint64_t, mk); PV(bool, win);
        TL_LANES_BEGIN L(key) = L(cost) <= L(room) ? L(ukey) : ~0ull; TL_LANES_END
        TL_PAR_MIN_U64(mk, key);
        TL_LANES_BEGIN L(win) = L(key) == L(mk) && L(key) != ~0ull; TL_LANES_END
        const uint64_t m = TL_BALLOT(win);
        if (m == 0ull) break;
        const uint64_t m0 = m & 0x5555555555555555ull, m1 = m & 0xaaaaaaaaaaaaaaaaull;
        const int wl0 = m0 ? __builtin_ctzll(m0) : -1, wl1 = m1 ? __builtin_ctzll(m1) : -1;     // ascending sb (maxmnr_new, encode_new.c:1061-1077)
        const int c0 = wl0 >= 0 ? TL_READLANE_I32(cost, wl0) : 0, c1 = wl1 >= 0 ? TL_READLANE_I32(cost, wl1) : 0;
        TL_LANES_BEGIN
        L(room) -= (lane & 1) ? c1 : c0;
        if (lane == wl0 || lane == wl1) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
    }
}
