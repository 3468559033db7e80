// mp2_host.cpp -- see mp2_host.h.  Compiled with -ffp-contract=off.
#include "mp2_host.h"
#include "tl_libm.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mp2_tables.inc"

namespace {
const double kRefPi = 3.14159265358979;            // common.h:26 -- truncated in the reference
const int kBitrate[2][15] = {                      // common.c:28-31
    {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160},
    {0, 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384}};
const double kSfreq[2][4] = {{22.05, 24, 16, 0}, {44.1, 48, 32, 0}};   // common.c:26

double bits_to_double(unsigned long long u) { double d; memcpy(&d, &u, 8); return d; }

double ath_db(double f, double value)              // ath.c:7-47
{
    if (f < -.3) f = 3410;
    f /= 1000;
    f = f > 0.01 ? f : 0.01;
    f = f < 18.0 ? f : 18.0;
    double ath = 3.640 * pow(f, -0.8) - 6.800 * exp(-0.6 * pow(f - 3.4, 2.0))
               + 6.000 * exp(-0.15 * pow(f - 8.7, 2.0)) + (0.6 + 0.04 * 0.0) * 0.001 * pow(f, 4.0);
    return ath + value;
}
double freq2bark(double freq)                      // ath.c:73-78
{
    if (freq < 0) freq = 0;
    freq = freq * 0.001;
    return 13.0 * atan(.76 * freq) + 3.5 * atan(freq * freq / (7.5 * 7.5));
}
}  // namespace

void tl_build_tables(TlTables *T)
{
    memset(T, 0, sizeof *T);
    for (int i = 0; i < 512; i++) { T->enwindow[i] = (double)TL_ENWINDOW_E9[i] / 1e9; T->enwindow_s[i] = T->enwindow[i] / 32768; }
    for (int i = 0; i < 63; i++) T->scalefactor[i] = (double)TL_SCALEFACTOR_E14[i] / 1e14;
    T->scalefactor[63] = 1e-20;
    for (int q = 0; q < 18; q++) {
        T->snr[q] = (double)TL_SNR_E2[q] / 100.0;
        T->pack.qa[q] = (double)TL_QUANT_A_E9[q] / 1e9;
        T->pack.qb[q] = (double)TL_QUANT_B_E9[q] / 1e9;
        T->pack.steps[q] = TL_STEPS[q];
        T->pack.steps2n[q] = TL_STEPS2N[q];
        T->pack.steps2n_f[q] = (double)TL_STEPS2N[q];
        T->bits[q] = TL_BITS[q];
        T->group[q] = TL_GROUP[q];
    }
    for (int l = 0; l < 9; l++) {
        T->nbal_line[l] = TL_NBAL[l];
        for (int b = 0; b < 16; b++) T->step_index[l][b] = TL_STEP_INDEX[l * 16 + b];
    }
    for (int l = 0; l < 9; l++)
        for (int b = 0; b < 16; b++) {
            const int q = TL_STEP_INDEX[l * 16 + b];
            T->shared.snr_line[l][b] = T->snr[q];
            T->shared.bits12_line[l][b] = (int16_t)(12 * TL_GROUP[q] * TL_BITS[q]);
            T->shared.qinfo_line[l][b] = (uint16_t)(q | (TL_BITS[q] << 5) | ((TL_GROUP[q] == 3 ? 1 : 0) << 10));
        }
    for (int i = 0; i < 64; i++) T->shared.scalefactor[i] = T->scalefactor[i];
    for (int i = 0; i < 63; i++) T->shared.scale_db[i] = 20 * log10(((double)TL_SCALEFACTOR_E14[i] / 1e14) * 32768) - 10;   // == TlConfig::scale_db (tl_build_config)
    T->shared.scale_db[63] = 20 * log10(1e-20 * 32768) - 10;
    {   // sf_transmission_pattern: pattern[5][5] of encode_new.c:296-301 as (sources of the three scalefactors, scfsi)
        static const unsigned short pat[25] = {0x123, 0x122, 0x122, 0x133, 0x123, 0x113, 0x111, 0x111, 0x444, 0x113,
                                               0x111, 0x111, 0x111, 0x333, 0x113, 0x222, 0x222, 0x222, 0x333, 0x123,
                                               0x123, 0x122, 0x122, 0x133, 0x123};
        memset(T->shared.sfpat, 0, sizeof T->shared.sfpat);
        for (int i = 0; i < 25; i++) {
            const int d0 = (pat[i] >> 8) & 15, d1 = (pat[i] >> 4) & 15, d2 = pat[i] & 15;
            int sel;                                                 // encode_new.c:318-351
            switch (pat[i]) {
            case 0x123: sel = 0; break;
            case 0x122: case 0x133: sel = 3; break;
            case 0x113: sel = 1; break;
            default: sel = 2; break;
            }
            T->shared.sfpat[i] = (uint8_t)((d0 - 1) | ((d1 - 1) << 2) | ((d2 - 1) << 4) | (sel << 6));
        }
    }
    {   // powers of x modulo the CRC-16 polynomial 0x8005 (CRC16_POLYNOMIAL, common.h:45)
        unsigned v = 1;
        for (int e = 0; e < 512; e++) { T->pack.crc_xpow[e] = (uint16_t)v; v <<= 1; if (v & 0x10000u) v = (v ^ 0x18005u) & 0xffffu; }
        {   // Reed-Solomon RS(255,207) of the EDI PFT layer: GF(2^8) with x^8+x^4+x^3+x^2+1, generator polynomial with roots
            // alpha^1..alpha^48 (contrib/edioutput/PFT.cpp:100-107: gfPoly 0x11d, firstRoot 1; contrib/fec/init_rs.h).
            uint8_t *lg = T->rs_log, *ex = T->rs_exp;
            unsigned sr = 1;
            lg[0] = 255;
            for (int i = 0; i < 255; i++) { lg[sr] = (uint8_t)i; ex[i] = (uint8_t)sr; sr <<= 1; if (sr & 0x100u) sr ^= 0x11du; }
            for (int i = 255; i < 512; i++) ex[i] = ex[i - 255];
            uint8_t gen[49];                                         // generator polynomial, gen[48] = 1 (coefficient form)
            gen[0] = 1;
            for (int i = 0, root = 1; i < 48; i++, root++) {
                gen[i + 1] = 1;
                for (int j = i; j > 0; j--) gen[j] = (uint8_t)(gen[j - 1] ^ (gen[j] ? ex[lg[gen[j]] + root] : 0));
                gen[0] = ex[lg[gen[0]] + root];
            }
            for (int u = 0; u < 207; u++) {                          // systematic encoder (LFSR division) on the unit chunk e_u
                uint8_t par[48] = {0};
                for (int i = 0; i < 207; i++) {
                    const uint8_t fb = (uint8_t)((i == u ? 1 : 0) ^ par[0]);
                    for (int j = 0; j < 47; j++) par[j] = (uint8_t)(par[j + 1] ^ ((fb && gen[47 - j]) ? ex[lg[fb] + lg[gen[47 - j]]] : 0));
                    par[47] = (uint8_t)((fb && gen[0]) ? ex[lg[fb] + lg[gen[0]]] : 0);
                }
                for (int j = 0; j < 48; j++) T->rs_mlog[u][j] = par[j] ? lg[par[j]] : 255;
            }
        }
        unsigned ve = 1;
        for (int k = 0; k < 2048; k++) {
            T->edi_xpow8[k] = (uint16_t)ve;
            for (int b = 0; b < 8; b++) { ve <<= 1; if (ve & 0x10000u) ve = (ve ^ 0x11021u) & 0xffffu; }
        }
        unsigned v8 = 1;
        for (int e = 0; e < 320; e++) { T->pack.crc8_xpow[e] = (uint8_t)v8; v8 <<= 1; if (v8 & 0x100u) v8 = (v8 ^ 0x11Du) & 0xffu; }
    }
    // matrixing coefficients: cos scaled by 1e9, rounded half away from zero, scaled back (subband.c:125-137)
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 32; k++) {
            double f = 1e9 * cos((double)((2 * i + 1) * k * kRefPi / 64)), ip;
            if (f >= 0) modf(f + 0.5, &ip); else modf(f - 0.5, &ip);
            T->dct[i][k] = ip * 1e-9;
        }
    for (int r = 0; r < 16; r++)
        for (int k = 0; k < 16; k++)
            for (int par = 0; par < 2; par++) T->shared.dct_t[k][par][r] = T->dct[r][2 * k + par];
    // Hann window with the sqrt(8/3)/N normalisation (psycho_1.c:225-233, psycho_3.c:135-141)
    const double sqrt_8_over_3 = pow(8.0 / 3.0, 0.5);
    for (int i = 0; i < 1024; i++) T->hann[i] = sqrt_8_over_3 * 0.5 * (1 - cos(2.0 * kRefPi * i / 1024)) / 1024;
    for (int i = 0; i < 1024; i++) T->hann_s[i] = T->hann[i] / 32768;           // a power of two: exact
    // dB-sum table (psycho_1.c:170-178, psycho_3.c:249-257)
    for (int i = 0; i < 1000; i++) {
        double x = (double)i / 10.0;
        T->dbtable[i] = 10 * log10(1 + pow(10.0, x / 10.0)) - x;
        T->shared.dbtable[i] = T->dbtable[i];
    }
    T->shared.dbtable[1000] = -0.0; T->shared.dbtable[1001] = -0.0;
    for (int i = 0; i < 1002; i++) T->dblog[i] = T->shared.dbtable[i];
    for (int i = 0; i < 256; i++) T->dblog[1002 + i] = tlm_u2d(tlm_log_tab[i]);
    // Buneman recurrence of fft.c:1139-1149 unrolled into a table (passes k = 2,4,6,8)
    int n = 0;
    for (int k = 2; k <= 8; k += 2) {
        const int kx = (1 << k) >> 1;
        const double t_c = bits_to_double(TL_FHT_COS_BITS[k]), t_s = bits_to_double(TL_FHT_SIN_BITS[k]);
        double c1 = 1, s1 = 0;
        for (int i = 1; i < kx; i++, n++) {
            double t = c1;
            c1 = t * t_c - s1 * t_s;
            s1 = t * t_s + s1 * t_c;
            T->fht_tw[n][0] = c1; T->fht_tw[n][1] = s1;
            T->fht_tw[n][2] = c1 * c1 - s1 * s1; T->fht_tw[n][3] = 2 * (c1 * s1);
        }
    }
    // Which general butterfly (block, i) of pass k the lanes' butterfly g is, is the HOST's choice -- the device takes its two base offsets and its
    // twiddle row from these tables -- and it is made for the LDS banks (MI355X: ds_read_b64 is served per 32-lane half, bank = double index mod 32;
    // ds_write_b64 per 16 lanes, mod 16; layout j -> j ^ (j >> 5)):
    //   k=4 (16 blocks x 7): half h of the 128 slots holds i = 1 + 2h and 2 + 2h, eight blocks at a time (slots 0-7: i, blocks 0-7; 8-15: i + 1,
    //        blocks 0-7; 16-23: i, blocks 8-15; 24-31: i + 1, blocks 8-15): banks i ^ 2 block -- 32 different ones per half, 16 per write group;
    //        the last half has i = 7 alone (16 slots).  Dealt block by block (g / 7, 1 + g % 7) 28 lanes of a half shared 8 banks.
    //   k=6 (4 blocks x 31): half h is block h, i = 1 + slot (slot 31 idle): banks i ^ 8 block.  Dealt densely a half straddled two blocks: 2-way.
    //   k=8 (1 block x 127): i = 1 + g, consecutive addresses; entry 127 is the pass's one trivial butterfly (f0 = 0, g0 = kx), which the fused
    //        last pass of models 1 / 3 deals to slot 127 (tl_psy_spectrum).
    // An idle slot is marked 0xffffffff.  Row of (k, i) in fht_tw: first row of the pass + i - 1 (first rows: k=4 -> 1, k=6 -> 8, k=8 -> 39).
    const int first_row[3] = {1, 8, 39};
    for (int p = 0; p < 3; p++) {
        const int k1 = 1 << (4 + 2 * p), kx = k1 >> 1, nblk = 128 / kx;
        int seen[16][128];
        memset(seen, 0, sizeof seen);
        for (int g = 0; g < 128; g++) {
            const int h = g >> 5, r = g & 31;
            int blk = 0, i = 1;
            bool valid = true, trivial = false;
            if (p == 0) {
                if (h < 3) { const int sub = r >> 3; i = 1 + 2 * h + (sub & 1); blk = (r & 7) + 8 * (sub >> 1); }
                else { i = 7; blk = r; valid = r < 16; }
            } else if (p == 1) { blk = h; i = 1 + r; valid = r < 31; }
            else { i = 1 + g; valid = g < 127; trivial = g == 127; }
            if (!valid && !trivial) {
                for (int q = 0; q < 4; q++) T->fht_tw_lane[p][g][q] = 0.0;
                T->fht_fg_lane[p][g] = 0xffffffffu;
                continue;
            }
            int f0 = blk * 4 * k1 + i, g0 = blk * 4 * k1 + k1 - i;
            if (trivial) { f0 = 0; g0 = kx; i = 1; }
            else if (seen[blk][i]++) { fprintf(stderr, "libtoolame-dab-hip: transform pass %d deals butterfly (%d, %d) twice\n", p, blk, i); abort(); }
            for (int q = 0; q < 4; q++) T->fht_tw_lane[p][g][q] = T->fht_tw[first_row[p] + i - 1][q];
            T->fht_fg_lane[p][g] = (uint32_t)((f0 ^ (f0 >> 5)) << 3) | (uint32_t)((g0 ^ (g0 >> 5)) << 3) << 16;
        }
        for (int blk = 0; blk < nblk; blk++)
            for (int i = 1; i < kx; i++)
                if (!seen[blk][i]) { fprintf(stderr, "libtoolame-dab-hip: transform pass %d never deals butterfly (%d, %d)\n", p, blk, i); abort(); }
    }
}

// A run of frames that does not start at the launch's first frame pays two seed passes (transform + square root + arctangent:
// about half a full pass each, csrc/mp2_wave.h tl_psy2_pass<true>), i.e. about half a frame.  Whole chains fill the complete
// rounds of waves; the chains of the last, partly filled round are cut so that its waves finish together.  Cost of a plan in
// frame times: complete rounds * nframes + rounds of runs * (plen + 0.5).
int tl_psy2_plan(int nchain, int nframes, int slots, int *nwhole, int *k, int *plen)
{
    if (slots < 1) slots = 1;
    int whole = (nchain / slots) * slots, rem = nchain - whole;
    int best_k = 1;
    if (rem > 0 && nframes > 1) {
        double best = (double)nframes;                               // k = 1: the last round takes a whole chain's time
        for (int kk = 2; kk <= nframes; kk++) {
            const int pl = (nframes + kk - 1) / kk;
            if ((long)pl * (kk - 1) >= nframes) continue;            // the last run would be empty: same cut as a smaller k
            const long units = (long)rem * kk;
            const double t = (double)((units + slots - 1) / slots) * ((double)pl + 0.5);
            if (t < best - 1e-9) { best = t; best_k = kk; }
        }
    }
    if (best_k == 1) { *nwhole = nchain; *k = 1; *plen = nframes; return nchain; }
    *nwhole = whole; *k = best_k; *plen = (nframes + best_k - 1) / best_k;
    return whole + rem * best_k;
}

int tl_psy2_slot(long samplerate)
{   // one table set per distinct rate the device path supports (TL_PSY2_SLOTS)
    switch (samplerate) { case 48000: return 0; case 32000: return 1; case 24000: return 2; case 16000: return 3; case 44100: return 4; default: return 5; }
}

// The spreading function by bands (TlPsy2Tables::s_band): for partition j the first and last column of row j of s that is not zero
// AMONG THE PARTITIONS THAT EXIST (columns >= npart carry coefficients too -- the reference fills all 64 x 64 from cbval[] = 0 -- but the
// grouped energies they would multiply are exact +0: a product of +0 adds nothing to a sum of non-negative terms either),
// the window [band_lo, band_lo + TL_P2_BAND) that holds them (pushed down where it would pass column 63), the coefficients of that
// window.  A table whose rows do not fit the window would silently lose terms, so that is fatal here, at table build time.
static void tl_psy2_band(TlPsy2Tables *P)
{
    int lo_[64];
    P->band_w = 0;
    for (int j = 0; j < 64; j++) {
        int lo = 64, hi = -1;
        for (int k = 0; k < P->npart; k++) if (P->s_t[k][j] != 0.0) { if (k < lo) lo = k; hi = k; }
        if (hi < 0) { lo = 0; hi = 0; }
        const int w = hi - lo + 1;
        if (w > TL_P2_BAND) { fprintf(stderr, "libtoolame-dab-hip: spreading band of partition %d is %d wide (TL_P2_BAND %d)\n", j, w, TL_P2_BAND); abort(); }
        if (w > P->band_w) P->band_w = w;
        lo_[j] = lo;
    }
    // the device sums whole batches: the window is the table's widest band rounded up to batches, the same for every partition
    const int window = (P->band_w + TL_P2_B - 1) / TL_P2_B * TL_P2_B;
    for (int j = 0; j < 64; j++) {
        const int lo = lo_[j] + window > 64 ? 64 - window : lo_[j];
        P->band_lo[j] = (int16_t)lo;
        for (int q = 0; q < TL_P2_BAND; q++) P->s_band[q][j] = q < window ? P->s_t[lo + q][j] : 0.0;
    }
}

void tl_build_psy2_tables(TlPsy2Tables *P, long samplerate)
{   // psycho_2_init, psycho_2.c:259-420
    static const double crit_band[27] = {0, 100, 200, 300, 400, 510, 630, 770, 920, 1080, 1270, 1480, 1720, 2000, 2320,
                                         2700, 3150, 3700, 4400, 5300, 6400, 7700, 9500, 12000, 15500, 25000, 30000};
    static const double bmax[27] = {20.0, 20.0, 20.0, 20.0, 20.0, 17.0, 15.0, 10.0, 7.0, 4.4, 4.5, 4.5, 4.5, 4.5,
                                    4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 3.5, 3.5, 3.5};
    const double LN_TO_LOG10 = 0.2302585093;                      // common.h:30
    memset(P, 0, sizeof *P);
    const double sfreq = (double)samplerate;
    int sfreq_idx;                                                // psycho_2.c:287-303
    switch (samplerate) { case 32000: case 16000: sfreq_idx = 0; break; case 44100: case 22050: sfreq_idx = 1; break; default: sfreq_idx = 2; break; }
    for (int j = 0; j < 513; j++) P->absthr[j] = (double)TL_PSY2_ABSTHR_E2[sfreq_idx * 513 + j] / 100.0;
    for (int i = 0; i < 1024; i++) P->window[i] = 0.5 * (1 - cos(2.0 * kRefPi * (i - 0.5) / 1024));
    double fthr[513], cbval[64] = {0}, rnorm[64] = {0}, s[64][64];
    int partition[513], numlines[64] = {0};
    const double freq_mult = sfreq / 1024;
    for (int i = 0; i < 513; i++) {
        const double t = i * freq_mult;
        int j = 1;
        while (t > crit_band[j]) j++;
        fthr[i] = j - 1 + (t - crit_band[j - 1]) / (crit_band[j] - crit_band[j - 1]);
    }
    partition[0] = 0;
    double cnt = 1, bval_lo = fthr[0];
    cbval[0] = fthr[0];
    int i;
    for (i = 1; i < 513; i++) {
        if ((fthr[i] - bval_lo) > 0.33) {
            partition[i] = partition[i - 1] + 1;
            cbval[partition[i - 1]] = cbval[partition[i - 1]] / cnt;
            cbval[partition[i]] = fthr[i];
            bval_lo = fthr[i];
            numlines[partition[i - 1]] = (int)cnt;
            cnt = 1;
        } else {
            partition[i] = partition[i - 1];
            cbval[partition[i]] += fthr[i];
            cnt++;
        }
    }
    numlines[partition[i - 1]] = (int)cnt;
    cbval[partition[i - 1]] = cbval[partition[i - 1]] / cnt;
    P->npart = partition[512] + 1;
    for (int j = 0; j < 64; j++)
        for (int k = 0; k < 64; k++) {
            double t1 = (cbval[k] - cbval[j]) * 1.05, t2, t3;
            if (t1 >= 0.5 && t1 <= 2.5) { t2 = t1 - 0.5; t2 = 8.0 * (t2 * t2 - 2.0 * t2); } else t2 = 0;
            t1 += 0.474;
            t3 = 15.811389 + 7.5 * t1 - 17.5 * sqrt(1.0 + t1 * t1);
            s[k][j] = t3 <= -100 ? 0 : exp((t2 + t3) * LN_TO_LOG10);
        }
    for (int j = 0; j < 64; j++) {
        const double t1 = 15.5 + cbval[j];
        P->tmn[j] = t1 > 24.5 ? t1 : 24.5;
        rnorm[j] = 0;
        for (int k = 0; k < 64; k++) rnorm[j] += s[j][k];
        P->bmaxk[j] = bmax[(int)(cbval[j] + 0.5)];
        P->den[j] = (rnorm[j] && numlines[j]) ? rnorm[j] * numlines[j] : 0.0;
        for (int k = 0; k < 64; k++) P->s_t[k][j] = s[j][k];
        P->part_lo[j] = P->part_hi[j] = 0;
    }
    for (int j = 512; j >= 0; j--) { P->partition[j] = (uint8_t)partition[j]; P->part_lo[partition[j]] = (int16_t)j; }
    for (int j = 0; j < 513; j++) P->part_hi[partition[j]] = (int16_t)(j + 1);
    tl_psy2_band(P);
}

void tl_build_psy4_tables(TlPsy2Tables *P, long samplerate)
{   // psycho_4_init, psycho_4.c:330-413 (+ psycho_4_spreading_function :418-458), mapped onto the psy-2 table record: the
    // per-frame code of psycho_4 (:123-327, non-NEWATAN build) performs the same operations as psycho_2's on these tables.
    static const double minval[27] = {0.0, 20.0, 20.0, 20.0, 20.0, 20.0, 17.0, 15.0, 10.0, 7.0, 4.4, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5,
                                      4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 3.5};      // psycho_4.c:62-76
    const double LN_TO_LOG10 = 0.2302585093;                      // common.h:30
    memset(P, 0, sizeof *P);
    const double sfreq = (double)samplerate;
    for (int i = 0; i < 1024; i++) P->window[i] = 0.5 * (1 - cos(2.0 * kRefPi * (i - 0.5) / 1024));
    double bark[513], cbval[64] = {0}, rnorm[64] = {0}, s[64][64];
    int partition[513], numlines[64] = {0};
    for (int i = 0; i < 513; i++) {
        const double freq = i * sfreq / 1024;
        bark[i] = freq2bark(freq);
        P->absthr[i] = pow(10.0, (ath_db(freq, 0) + 0 /* glopts.athlevel, toolame.c:41 */ + 41.837375) * 0.1);   // ATH_energy, ath.c:49-66
    }
    int partition_count = 0, cbase = 0;
    for (int i = 0; i < 513; i++) {
        if ((bark[i] - bark[cbase]) > 0.33) { cbase = i; partition_count++; }
        partition[i] = partition_count;
        numlines[partition_count]++;
    }
    for (int i = 0; i < 513; i++) cbval[partition[i]] += bark[i];
    for (int i = 0; i < 64; i++) cbval[i] = numlines[i] != 0 ? cbval[i] / numlines[i] : 0;
    for (int i = 0; i < 64; i++)
        for (int j = 0; j < 64; j++) {
            double tempx = 1.05 * (cbval[i] - cbval[j]), x, tempy;
            if (tempx >= 0.5 && tempx <= 2.5) { const double temp = tempx - 0.5; x = 8.0 * (temp * temp - 2.0 * temp); } else x = 0.0;
            tempx += 0.474;
            tempy = 15.811389 + 7.5 * tempx - 17.5 * sqrt(1.0 + tempx * tempx);
            s[i][j] = tempy <= -60.0 ? 0.0 : exp((x + tempy) * LN_TO_LOG10);
            rnorm[i] += s[i][j];
        }
    P->npart = partition_count + 1;
    for (int j = 0; j < 64; j++) {
        const double t1 = 15.5 + cbval[j];
        P->tmn[j] = t1 > 24.5 ? t1 : 24.5;
        P->bmaxk[j] = minval[(int)cbval[j]];
        P->den[j] = (rnorm[j] && numlines[j]) ? rnorm[j] * numlines[j] : 0.0;
        for (int k = 0; k < 64; k++) P->s_t[k][j] = s[j][k];
        P->part_lo[j] = P->part_hi[j] = 0;
    }
    for (int j = 512; j >= 0; j--) { P->partition[j] = (uint8_t)partition[j]; P->part_lo[partition[j]] = (int16_t)j; }
    for (int j = 0; j < 513; j++) P->part_hi[partition[j]] = (int16_t)(j + 1);
    tl_psy2_band(P);
}

int tl_build_config(TlConfig *C, long samplerate, char mode, int kbps, int psy, int pad_len)
{
    memset(C, 0, sizeof *C);
    switch (samplerate) {                                         // SmpFrqIndex, common.c:118-144
    case 48000: C->version = 1; C->fs_idx = 1; break;
    case 32000: C->version = 1; C->fs_idx = 2; break;
    case 24000: C->version = 0; C->fs_idx = 1; break;
    case 16000: C->version = 0; C->fs_idx = 2; break;
    // 44100 / 22050: frames of two lengths (padding slots, availbits.c:49-62).  DAB does not use them
    // (src/odr-audioenc.cpp:560-563 accepts 24000/48000 only); the library does, like libtoolame-dab.
    case 44100: C->version = 1; C->fs_idx = 0; break;
    case 22050: C->version = 0; C->fs_idx = 0; break;
    default: return TL_ERR_SAMPLERATE;
    }
    // toolame.c:204-207 accepts 0..3; model 4 (psycho_4.c, unreachable through the reference's setter) is an extension of
    // the batched API only -- the legacy toolame_set_psy_model() shim keeps rejecting it
    if (psy < 0 || psy > 4) return TL_ERR_PSY;
    C->psy = psy;
    C->psy2_tab = tl_psy2_slot(samplerate) + (psy == 4 ? TL_PSY2_SLOTS : 0);   // slots 0..5: psy 2, 6..11: psy 4
    switch (mode) {                                               // toolame.c:174-200
    case 's': C->mode0 = 0; C->mode_ext0 = 0; break;
    case 'd': C->mode0 = 2; C->mode_ext0 = 0; break;
    case 'j': C->mode0 = 1; C->mode_ext0 = 2; break;
    case 'm': C->mode0 = 3; C->mode_ext0 = 0; break;
    default: return TL_ERR_MODE;
    }
    C->nch = C->mode0 == 3 ? 1 : 2;
    if (kbps == 0) kbps = kBitrate[C->version][10];               // toolame.c:217-218
    C->br_idx = -1;
    for (int i = 1; i < 15; i++) if (kBitrate[C->version][i] == kbps) C->br_idx = i;
    if (C->br_idx < 0) return TL_ERR_BITRATE;
    C->kbps = kbps;
    C->br_per_ch = kbps / C->nch;
    C->dab_ext = 4;                                               // toolame.c:147,225-232
    if (C->version == 1 && (kbps / (C->mode0 == 3 ? 1 : 2) < 56)) C->dab_ext = 2;
    if (pad_len < 0 || pad_len > TL_MAX_XPAD) return TL_ERR_PAD;
    C->dab_length = pad_len;
    {   // alloc table choice (encode_new.c:104-125)
        const int sfrq = (int)kSfreq[C->version][C->fs_idx], b = C->br_per_ch;
        if (C->version == 1) {
            if ((sfrq == 48 && b >= 56) || (b >= 56 && b <= 80)) C->tab = 0;
            else if (sfrq != 48 && b >= 96) C->tab = 1;
            else if (sfrq != 32 && b <= 48) C->tab = 2;
            else C->tab = 3;
        } else C->tab = 4;
    }
    C->sblimit = TL_TABLE_SBLIMIT[C->tab];
    {
        const int jsb[4] = {4, 8, 12, 16};                        // common.c:64-74
        C->jsbound0 = C->mode0 == 1 ? jsb[C->mode_ext0] : C->sblimit;
    }
    for (int sb = 0; sb < 32; sb++) {
        C->line[sb] = TL_LINE[C->tab * 32 + sb];
        C->nbal[sb] = sb < C->sblimit ? TL_NBAL[C->line[sb]] : 0;
    }
    {   // slots per frame (availbits.c:36-67): `whole`, and the fraction that makes some frames one slot longer
        const double average = (1152.0 / kSfreq[C->version][C->fs_idx]) * ((double)kbps / 8.0);
        const int whole = (int)average;
        C->frame_bytes = whole;
        C->pad_frac = average - (double)whole;
        if (whole + (C->pad_frac != 0 ? 1 : 0) > TL_MAX_FRAME_BYTES) return TL_ERR_BITRATE;
        // the PAD of a frame must leave room for header (4), CRC-16 (2), the bit_alloc fields and the ScF-CRC: with less the
        // bit budget of toolame.c:292-301 goes negative (the reference then writes a broken frame); such a pad length is refused
        int bbal = 0;
        for (int sb = 0; sb < C->sblimit; sb++) bbal += C->nbal[sb] * (sb < C->jsbound0 ? C->nch : 1);
        if (pad_len && 8 * (pad_len + C->dab_ext) + 32 + 16 + bbal > 8 * whole) return TL_ERR_PAD;
    }
    for (int i = 0; i < 63; i++) {
        const double sf = (double)TL_SCALEFACTOR_E14[i] / 1e14;
        C->scale_db[i] = 20 * log10(sf * 32768) - 10;             // psycho_1.c:575, psycho_3.c:180
    }
    C->scale_db[63] = 20 * log10(1e-20 * 32768) - 10;

    // ---- psy model 1 tables (psycho_1.c:94-168) ----
    {
        const int t = C->version == 1 ? C->fs_idx : C->fs_idx + 4;
        C->p1_ncb = TL_PSY1_CBOUND[t * 28];
        for (int i = 0; i < C->p1_ncb; i++) C->p1_cbound[i] = TL_PSY1_CBOUND[t * 28 + 1 + i];
        C->p1_sub = TL_PSY1_FREQ_ENTRIES[t] + 1;
        C->p1_line[0] = 0; C->p1_bark[0] = 0.0; C->p1_hear[0] = 0.0;
        for (int i = 1; i < C->p1_sub; i++) {
            C->p1_line[i] = TL_PSY1_LINE[t * 132 + i - 1];
            C->p1_bark[i] = (double)TL_PSY1_BARK_E3[t * 132 + i - 1] / 1000.0;
            C->p1_hear[i] = (double)TL_PSY1_HEAR_E2[t * 132 + i - 1] / 100.0;
        }
        for (int i = 1; i < C->p1_sub; i++)
            for (int j = C->p1_line[i - 1]; j <= C->p1_line[i]; j++) C->p1_map[j] = (uint8_t)i;
        memset(C->p1_lineband, 255, sizeof C->p1_lineband);
        for (int b = 0; b + 1 < C->p1_ncb; b++)
            for (int j = C->p1_cbound[b]; j < C->p1_cbound[b + 1]; j++) C->p1_lineband[j] = (uint8_t)b;
        for (int j = 0; j < 512; j++) { C->p1_lbark[j] = C->p1_bark[C->p1_map[j]]; C->p1_lhear[j] = C->p1_hear[C->p1_map[j]]; }
        memset(C->p1_lineinfo, 0, sizeof C->p1_lineinfo);
        for (int b = 0; b + 1 < C->p1_ncb; b++)
            for (int j = C->p1_cbound[b]; j < C->p1_cbound[b + 1] && j < 512; j++)
                C->p1_lineinfo[j] = (uint32_t)b | ((uint32_t)C->p1_cbound[b] << 8) | ((uint32_t)C->p1_cbound[b + 1] << 20);
        for (int j = 0; j < 512; j++) {
            const uint32_t info = C->p1_lineinfo[j];
            C->p1_linerw[j] = info ? 1.0 / (double)((int)(info >> 20) - (int)((info >> 8) & 0xfffu)) : 0.0;
        }
        // resolve the sequential minimum-mask walk (psycho_1.c:541-559) into per-subband row ranges
        int j = 1;
        for (int sb = 0; sb < C->sblimit; sb++) {
            if (j >= C->p1_sub - 1) { C->p1_mm_j0[sb] = 0; C->p1_mm_n[sb] = 0; continue; }
            int j0 = j;
            while (j < C->p1_sub && (C->p1_line[j] >> 4) == sb) j++;
            C->p1_mm_j0[sb] = (int16_t)j0;
            C->p1_mm_n[sb] = (int16_t)(j > j0 ? j - j0 : 1);     // empty subband: the walk reads row j0 once
        }
    }
    // ---- psy model 3 tables (psycho_3.c:434-512) ----
    {
        const double sfreq = kSfreq[C->version][C->fs_idx] * 1000;
        for (int i = 1; i < 513; i++) {
            const double freq = i * sfreq / 1024;
            C->p3_bark[i] = freq2bark(freq);
            C->p3_ath[i] = ath_db(freq, 0);
        }
        int cbase = 0, cb = 0;
        C->p3_cbidx[0] = 1;
        for (int i = 1; i < 513; i++)
            if ((C->p3_bark[i] - C->p3_bark[cbase]) > 1.0) { cbase = i; cb++; C->p3_cbidx[cb] = (int16_t)cbase; }
        cb++;
        C->p3_cbidx[cb] = 513;
        C->p3_cbands = cb;
        memset(C->p3_lineband, 255, sizeof C->p3_lineband);
        for (int b = 0; b < cb; b++)
            for (int j = C->p3_cbidx[b]; j < C->p3_cbidx[b + 1]; j++) C->p3_lineband[j] = (uint8_t)b;
        memset(C->p3_lineinfo, 0, sizeof C->p3_lineinfo);
        for (int b = 0; b < cb; b++)
            for (int j = C->p3_cbidx[b]; j < C->p3_cbidx[b + 1]; j++)
                C->p3_lineinfo[j] = (uint32_t)b | ((uint32_t)C->p3_cbidx[b] << 8) | ((uint32_t)C->p3_cbidx[b + 1] << 20);
        int n = 0, i = 1;
        for (; i < 3 * 16 + 1; i++) C->p3_subset[n++] = (int16_t)i;
        for (; i < 6 * 16 + 1; i += 2) C->p3_subset[n++] = (int16_t)i;
        for (; i < 12 * 16 + 1; i += 4) C->p3_subset[n++] = (int16_t)i;
        for (; i < 32 * 16 + 1; i += 8) C->p3_subset[n++] = (int16_t)i;
        for (int sb = 0; sb < 32; sb++) { C->p3_sb_j0[sb] = 0; C->p3_sb_n[sb] = 0; }
        for (int j = 135; j >= 0; j--) { const int sb = C->p3_subset[j] >> 4; C->p3_sb_j0[sb] = (int16_t)j; C->p3_sb_n[sb]++; }
    }
    // ---- psy model 0 (psycho_0.c:36-50) ----
    {
        const double per_line = kSfreq[C->version][C->fs_idx] * 1000 / 1024.0;
        for (int sb = 0; sb < 32; sb++) C->p0_athmin[sb] = 1000;
        for (int i = 0; i < 512; i++) {
            const double v = ath_db(i * per_line, 0);
            if (v < C->p0_athmin[i >> 4]) C->p0_athmin[i >> 4] = v;
        }
    }
    return TL_OK;
}
