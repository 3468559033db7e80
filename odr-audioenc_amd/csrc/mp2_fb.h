// mp2_fb.h -- K1: the polyphase analysis filterbank (subband.c:201-310).
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ---- K1: polyphase filterbank (subband.c:201-310), 36 blocks of 32 samples, for the `nlan` channels staged in w.u.fbk.pcm ----
// (the two channels of a stereo stream, one channel, or channel 0 of each of two mono streams sharing the wave)
// Window stage: lane (ch,i) owns yprime[i] and computes exactly the two window outputs it is made of
// (yprime[0]=y[16]; yprime[i]=y[i+16]+y[16-i], i<=16; y[i+16]-y[80-i], i>=17 -- every y is used by one
// yprime only, so nothing is computed twice), each as the reference's ascending 8-tap chain.
// Matrixing stage: lane (ch,sb) owns the even-k chain s0 (sb<16) or the odd-k chain s1 (sb>=16) of row
// min(sb,31-sb); the two halves swap values (a move, not a re-association): s[i]=s0+s1, s[31-i]=s0-s1.
TL_FN void tl_filterbank(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s, const int nch, PARGA(double, smp, 36))
{
    constexpr int FB = TlMainLds::kFbBatch;
        // the reference scales the sample, (pcm/32768)*C (subband.c:233,249); scaling the coefficient instead is the
        // same real product rounded once (2^-15 is exact, nothing underflows), so the bits are identical: enw_s = C / 32768
        // (host table; the encode kernel of the split path reads its workgroup's LDS copy).
        // Window taps as a rolling register file: tap j of block b is tap j+1 of block b+2 (the window advances 32
        // samples per block, the taps are 64 apart), so each block reads two new samples per lane from LDS instead of
        // sixteen (kept as integers and converted at every use: a window of doubles, converted once, measured slower each
        // time it was tried).  Slot of (b, j): [b & 1][((b >> 1) - j) & 7].
        // The coefficients are NOT kept in registers across batches: a batch fetches the eight of its ya-sums, runs them for
        // all its blocks, then the eight of its yb-sums -- 16 registers live instead of 32 next to the 72 of the samples.
        PA(int, xa, 16); PA(int, xb, 16);
        TL_LANES_BEGIN
        const int c = lane & 1, i = lane >> 1;
        const int ya = i == 0 ? 16 : i + 16, yb = i == 0 ? 16 : (i <= 16 ? 16 - i : 80 - i);
        for (int b = 0; b < 2; b++)
            for (int j = 1; j < 8; j++) {
                L(xa)[8 * b + ((0 - j) & 7)] = c < nch ? w.u.fbk.pcm[c][TL_HIST + 32 * b + 31 - ya - 64 * j] : 0;
                L(xb)[8 * b + ((0 - j) & 7)] = c < nch ? w.u.fbk.pcm[c][TL_HIST + 32 * b + 31 - yb - 64 * j] : 0;
            }
        TL_LANES_END
        TlMainLds::YpRows yp = w.yp_rows();
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int b0 = 0; b0 < 36; b0 += FB) {
            TL_LANES_BEGIN
            const int c = lane & 1, i = lane >> 1;
            if (c < nch) {
                const int ya = i == 0 ? 16 : i + 16, yb = i == 0 ? 16 : (i <= 16 ? 16 - i : 80 - i);
                // yprime = ya-sum (i == 0), ya-sum + yb-sum (i <= 16), ya-sum - yb-sum (i >= 17) as ONE addition: the yb-sum
                // with its sign flipped (a - b == a + (-b)) or replaced by -0.0 (a + (-0.0) == a, for every a)
                const uint64_t keep = i == 0 ? 0ull : ~0ull, flip = (i == 0 || i > 16) ? 0x8000000000000000ull : 0ull;
                // the batch's new samples (two per block) and its first coefficients are all requested before the first block is computed
                int na[FB], nb[FB];
                double cf[8], ta[FB];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) {
                    // X[k] = pcm[t0 + 31 - k], t0 = index of the block's first new sample
                    na[bb] = w.u.fbk.pcm[c][TL_HIST + 32 * (b0 + bb) + 31 - ya];
                    nb[bb] = w.u.fbk.pcm[c][TL_HIST + 32 * (b0 + bb) + 31 - yb];
                }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int j = 0; j < 8; j++) cf[j] = enw_s[ya + 64 * j];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) { TL_KEEP(na[bb]); TL_KEEP(nb[bb]); }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) {
                    const int b = b0 + bb, q = 8 * (b & 1), h = b >> 1;
                    L(xa)[q + (h & 7)] = na[bb];
                    double t = (double)L(xa)[q + (h & 7)] * cf[0];
                    for (int j = 1; j < 8; j++) t += (double)L(xa)[q + ((h - j) & 7)] * cf[j];
                    ta[bb] = t;
                }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int j = 0; j < 8; j++) cf[j] = enw_s[yb + 64 * j];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) {
                    const int b = b0 + bb, q = 8 * (b & 1), h = b >> 1;
                    L(xb)[q + (h & 7)] = nb[bb];
                    double t = (double)L(xb)[q + (h & 7)] * cf[0];
                    for (int j = 1; j < 8; j++) t += (double)L(xb)[q + ((h - j) & 7)] * cf[j];
                    yp[bb][TL_YP_ROW * (2 * c + (i & 1)) + (i >> 1)] = ta[bb] + tl_u2d((tl_d2u(t) & keep) ^ flip);
                }
            }
            TL_LANES_END
            PA(double, part, FB);
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1, par = sb < 16 ? 0 : 1, r = sb < 16 ? sb : 31 - sb;
            double acc[FB];
            for (int bb = 0; bb < FB; bb++) acc[bb] = 0.0;
            if (c < nch)
                for (int k = 0; k < 16; k += 2) {
                    const double m0 = B->dct_t[k][par][r], m1 = B->dct_t[k + 1][par][r];      // shared LDS copy, conflict-free per k
                    for (int bb = 0; bb < FB; bb++) {
                        double y0, y1;                                  // yprime[2 k + par], yprime[2 (k + 1) + par]: neighbours in the lane's row
                        TL_LD2(&yp[bb][TL_YP_ROW * (2 * c + par) + k], y0, y1);
                        acc[bb] += m0 * y0;
                        acc[bb] += m1 * y1;
                    }
                }
            for (int bb = 0; bb < FB; bb++) L(part)[bb] = acc[bb];
            TL_LANES_END
            PA(double, oth, FB);
#ifdef TL_EMULATE
            for (int lane = 0; lane < 64; ++lane)
                for (int bb = 0; bb < FB; bb++) oth[lane][bb] = part[2 * (31 - (lane >> 1)) + (lane & 1)][bb];
#else
            {
                const int lane_ = (int)(threadIdx.x & 63u), partner = 2 * (31 - (lane_ >> 1)) + (lane_ & 1);
#pragma unroll
                for (int bb = 0; bb < FB; bb++) oth[bb] = __shfl(part[bb], partner, 64);
            }
#endif
            // s[i] = s0 + s1 on the lanes of the even chains, s[31 - i] = s0 - s1 on the others: two BRANCHES, so that each half of the wave
            // runs one addition per block under its own execution mask -- as selects both results are computed for every lane and two
            // v_cndmask per block pick one (4 instructions per block against 1 + 1)
            TL_LANES_BEGIN
            const int sb = lane >> 1;
            // (lanes of a channel the stream does not have hold part = oth = +0.0 -- their chains were skipped -- and come out as +0.0 either way)
            if (sb < 16) { for (int bb = 0; bb < FB; bb++) L(smp)[b0 + bb] = L(part)[bb] + L(oth)[bb]; }
            else { for (int bb = 0; bb < FB; bb++) L(smp)[b0 + bb] = L(oth)[bb] - L(part)[bb]; }
            TL_LANES_END
        }
}
