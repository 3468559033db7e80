// mp2_dbsum.h -- dB sums, masking terms, exact division, masker spans, scalefactor index, bit writer and CRC step: the helpers every stage shares.
// Part of mp2_wave.h (included from there, in order; lane-SPMD source that compiles for gfx950 and, with TL_EMULATE, as a lane loop).
#ifndef MP2_WAVE_PARTS
#error "include mp2_wave.h"
#endif
// ------------------------------------------------------------------------------------------
TL_FN double tl_add_db(const double *TL_RESTRICT dbtable, double a, double b)
{   // psycho_1.c:180-205 == psycho_3.c:44-69, written without branches (every lane of a wave walks its own
    // chain) and with nothing but the final add behind the table read.  Inside |fdiff| <= 990 the index is the
    // reference's (int)fdiff; beyond it the reference returns the larger operand unchanged, which is
    // operand + table[1000] with table[1000] = -0.0.
    const double fdiff = 10.0 * (a - b);
    const double af = __builtin_fabs(fdiff);
    const int mag = (int)af;                                        // == |(int)fdiff|: truncation is symmetric
    const int idx = TL_SELECT(af > 990.0, 1000, mag);
    const double base = TL_SELECT(fdiff > -1.0, a, b);              // (int)fdiff >= 0
    return base + dbtable[idx];
}
// Two independent dB sums at once: both table entries are requested before either is used.
TL_FN void tl_add_db2(const double *TL_RESTRICT dbtable, double &a0, double b0, double &a1, double b1)
{
    const double f0 = 10.0 * (a0 - b0), f1 = 10.0 * (a1 - b1);
    const double g0 = __builtin_fabs(f0), g1 = __builtin_fabs(f1);
    int i0 = TL_SELECT(g0 > 990.0, 1000, (int)g0), i1 = TL_SELECT(g1 > 990.0, 1000, (int)g1);
    const double s0 = TL_SELECT(f0 > -1.0, a0, b0), s1 = TL_SELECT(f1 > -1.0, a1, b1);
    TL_KEEP(i0); TL_KEEP(i1);
    const double t0 = dbtable[i0], t1 = dbtable[i1];
    a0 = s0 + t0; a1 = s1 + t1;
}
TL_FN void tl_add_db2_k(const double *TL_RESTRICT dbtable, int k1000, double &a0, double b0, double &a1, double b1)
{
    const double f0 = 10.0 * (a0 - b0), f1 = 10.0 * (a1 - b1);
    const double g0 = __builtin_fabs(f0), g1 = __builtin_fabs(f1);
    int i0 = TL_SELECT(g0 > 990.0, k1000, (int)g0), i1 = TL_SELECT(g1 > 990.0, k1000, (int)g1);
    const double s0 = TL_SELECT(f0 > -1.0, a0, b0), s1 = TL_SELECT(f1 > -1.0, a1, b1);
    TL_KEEP(i0); TL_KEEP(i1);
    const double t0 = dbtable[i0], t1 = dbtable[i1];
    a0 = s0 + t0; a1 = s1 + t1;
}
TL_FN uint64_t tl_mnr_key(double mnr)
{   // order-preserving map double -> u64 for the allocation arg-min; ~0 = never chosen (encode_new.c:1068: small = 999999.0)
    uint64_t u = tl_d2u(mnr + 0.0);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    return 999999.0 > mnr ? u : ~0ull;
}
// One step of a threshold chain: x (+) the masking of one masker at distance dz bark, if it reaches the line at all
// (psycho_1.c:489-517 == psycho_3.c:352-394).  The reference's masking function
//   dz < -1: 17*(dz+1) - g     dz < 0: g*dz     dz < 1: -17*dz     else: -(dz-1)*n - 17        (g = 0.4x+6, n = 17-0.15x)
// is, with a = |dz|,  -(A*(a - B) + C):  inside |dz| < 1  A = (dz<0 ? g : 17), B = C = 0;  outside  A = (dz<0 ? 17 : n), B = 1,
// C = (dz<0 ? g : 17) -- the same roundings (negating an operand or a result changes no rounding; adding or subtracting a
// zero changes no bit; at dz = -1 and dz = 1 both neighbouring pieces give the same value), and level + vf = level - (...).
// C is the inside A times B (a product with 0.0 or 1.0 is exact): 64-bit selects cost two instructions, a product one.
// A masker out of reach (dz outside [-3, 8)) enters the dB sum as a level below -65536 dB, which leaves the sum as it
// is (|difference| > 99 dB: the reference returns the larger operand, tl_add_db adds its -0.0 entry).
// The literals of the threshold walk, made once per walk (TL_PIN) instead of once per masker and line.
struct TlMaskK { uint32_t c17_hi, one_hi, far_hi; int k1000; };
TL_FN TlMaskK tl_mask_consts()
{
    TlMaskK k;
    k.c17_hi = 0x40310000u; k.one_hi = 0x3ff00000u; k.far_hi = 0xC0F00000u; k.k1000 = 1000;
    TL_PIN(k.c17_hi); TL_PIN(k.one_hi); TL_PIN(k.far_hi); TL_PIN(k.k1000);
    return k;
}
TL_FN double tl_mask_term(double dz, double av, double g, double n, bool live = true)
{
    const double ad = __builtin_fabs(dz);
    const bool s = dz < 0.0, o = ad >= 1.0;
    const double G = TL_SELECT(s, g, 17.0), H = TL_SELECT(s, 17.0, n);
    const double A = TL_SELECT(o, H, G), Bc = TL_SELECT(o, 1.0, 0.0);
    const double term = av - (A * (ad - Bc) + G * Bc);
    const bool in = live && dz >= -3.0 && dz < 8.0;
    const uint64_t tu = tl_d2u(term);
    const uint32_t hi = TL_SELECT(in, (uint32_t)(tu >> 32), 0xC0F00000u);
    return tl_u2d(((uint64_t)hi << 32) | (tu & 0xffffffffull));
}
// The same term with the walk's pinned literals.  Same operations on the same values: 17.0 = {c17_hi, 0}, 1.0 = {one_hi, 0}.
TL_FN double tl_mask_term_k(const TlMaskK &k, double dz, double av, double g, double n, bool live = true)
{
    const double ad = __builtin_fabs(dz);
    const bool s = dz < 0.0, o = ad >= 1.0;
    const uint64_t gu = tl_d2u(g), nu = tl_d2u(n);
    const uint32_t Gh = TL_SELECT(s, (uint32_t)(gu >> 32), k.c17_hi), Gl = TL_SELECT(s, (uint32_t)gu, 0u);
    const uint32_t Hh = TL_SELECT(s, k.c17_hi, (uint32_t)(nu >> 32)), Hl = TL_SELECT(s, 0u, (uint32_t)nu);
    const uint32_t Ah = TL_SELECT(o, Hh, Gh), Al = TL_SELECT(o, Hl, Gl);
    const double G = tl_u2d(((uint64_t)Gh << 32) | Gl), A = tl_u2d(((uint64_t)Ah << 32) | Al);
    const double Bc = tl_u2d((uint64_t)TL_SELECT(o, k.one_hi, 0u) << 32);
    const double term = av - (A * (ad - Bc) + G * Bc);
    const bool in = live && dz >= -3.0 && dz < 8.0;
    const uint64_t tu = tl_d2u(term);
    const uint32_t hi = TL_SELECT(in, (uint32_t)(tu >> 32), k.far_hi);
    return tl_u2d(((uint64_t)hi << 32) | (tu & 0xffffffffull));
}
// 1 for a negative x, else 0.  On the device one shift of the high word, opaque to the compiler (which otherwise folds it into the
// address arithmetic that follows as shift + and + add: three instructions where shift + shift-add do).
TL_FN int tl_sign_bit(double x)
{
#ifdef TL_EMULATE
    return (int)(tl_d2u(x) >> 63);
#else
    int r;
    asm("v_lshrrev_b32 %0, 31, %1" : "=v"(r) : "v"((uint32_t)(tl_d2u(x) >> 32)));
    return r;
#endif
}
// The same term without a select for its shape.  dzp = masker bark - line bark = -dz (exactly: negation commutes with rounding).
//  * which pair of slopes (inner G, outer H): dz < 0 -> (g, 17), else (17, n) -- the 16-byte window of the masker's record at
//    &g + (dzp < 0): one address computed from the sign bit, one LDS read.  At dz = 0 either window serves (both products are 0).
//  * inside / outside |dz| = 1:  A (ad - Bc) + G Bc  with  (A, Bc) = (H, 1) outside, (G, 0) inside  is  H max(ad - 1, 0) + G min(ad, 1):
//    outside the very same operations (ad - 1.0; G * 1.0 == G); inside G * ad plus a zero in either form, and adding a zero of
//    either sign to a sum changes no bit of it unless the sum is itself a zero -- in which case the term is av - (+-0) = av
//    in either form, av never being -0.0 (a sum of finite non-zero values never rounds to -0).
//  * the reach test -3 <= dz < 8 is -8 < dzp <= 3.
TL_FN double tl_mask_term_w(const TlMasker *TL_RESTRICT m, double dzp, double av, uint32_t far_hi, bool live = true)
{
    const double ad = __builtin_fabs(dzp);
    const double *gh = &m->g + tl_sign_bit(dzp);
    const double G = gh[0], H = gh[1];
    const double t1 = __builtin_fmax(ad - 1.0, 0.0), t2 = __builtin_fmin(ad, 1.0);
    const double term = av - (H * t1 + G * t2);
    const bool in = live && dzp <= 3.0 && dzp > -8.0;
    const uint64_t tu = tl_d2u(term);
    const uint32_t hi = TL_SELECT(in, (uint32_t)(tu >> 32), far_hi);
    return tl_u2d(((uint64_t)hi << 32) | (tu & 0xffffffffull));
}
TL_FN double tl_mask_step(const double *TL_RESTRICT db, double x, const TlMasker *TL_RESTRICT m, double dzp, double av, uint32_t far_hi)
{
    return tl_add_db(db, x, tl_mask_term_w(m, dzp, av, far_hi));
}
TL_FN void tl_masker_consts(TlMasker *TL_RESTRICT mk, const double *TL_RESTRICT mx, const double *TL_RESTRICT mbk, int t, bool tonal)
{
    const double x = mx[t], mb = mbk[t];
    mk[t].bark = mb;
    mk[t].av = tonal ? -1.525 - 0.275 * mb - 4.5 + x : -1.525 - 0.175 * mb - 0.5 + x;
    mk[t].g = 0.4 * x + 6;
    mk[t].c17 = 17.0;
    mk[t].n = 17 - 0.15 * x;
}
// s / d given r = RN(1/d): two residual corrections with fused multiply-adds.  After the first, q is a faithful
// quotient (error ~2u^2 before its rounding); for a faithful q and the correctly rounded reciprocal the second yields the
// correctly rounded quotient (Markstein's theorem; its one exception, a divisor whose significand is all ones, does not
// occur among the divisors used: scalefactors and critical-band widths -- tests/test_emu_parity.py checks them and
// 10^8 quotients incl. the hardest near-midpoint ones).
// No scaling: the encoder's operands are far from the exponent limits.  A zero dividend may come out as +0 where the
// division gives -0; the quantiser adds a non-zero constant next, so no bit depends on it.
TL_FN double tl_div_by(double s, double d, double r)
{
    double q = s * r;
    double e = __builtin_fma(-q, d, s);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-q, d, s);
    return __builtin_fma(e, r, q);
}
// First and last masker with blo < bark <= bhi among the tones [0, ntone) -> a0..a1 and among the noise components
// [ntone, nm) -> b0..b1 (empty: first > last).  Lane-private: called inside a lanes block.
TL_FN void tl_mask_spans(const TlMasker *mk, int nm, int ntone, double blo, double bhi, int &a0, int &a1, int &b0, int &b1)
{
    a0 = nm; a1 = -1; b0 = nm; b1 = -1;
    for (int tb = 0; tb < nm; tb += 32) {                           // 32 maskers -> one hit mask, eight barks per LDS round trip
        uint32_t m = 0;
        for (int t8 = 0; t8 < 32 && tb + t8 < nm; t8 += 8) {
            double mb[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) mb[q] = mk[tb + t8 + q].bark;              // entries past nm (< TL_MASKER_MAX) are masked below
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) m |= (mb[q] > blo && mb[q] <= bhi) ? 1u << (t8 + q) : 0u;
        }
        const int left = nm - tb, tleft = ntone - tb;
        m &= left >= 32 ? ~0u : (1u << left) - 1u;
        const uint32_t tmask = tleft >= 32 ? ~0u : tleft <= 0 ? 0u : (1u << tleft) - 1u;
        const uint32_t mt = m & tmask, mn = m & ~tmask;
        const int ft = tb + __builtin_ctz(mt | 0x80000000u), lt = tb + 31 - __builtin_clz(mt | 1u);
        const int fn = tb + __builtin_ctz(mn | 0x80000000u), ln = tb + 31 - __builtin_clz(mn | 1u);
        a0 = (mt && ft < a0) ? ft : a0; a1 = mt ? lt : a1;
        b0 = (mn && fn < b0) ? fn : b0; b1 = mn ? ln : b1;
    }
}
// The same spans when both lists are ascending in bark -- they are, except after the dead-head replay: tones come in chain order
// (ascending lines), noise components in band order, and bark grows with the line -- by bisection instead of a look at every masker:
// count(B <= v) for v = blo and bhi on the tones mb[0, ntone) and on the noise components mb[ntone, ntone + nnoise), the four
// searches side by side (four reads in flight per level).  STEPS_T / STEPS_N: highest power of two of a count (64: up to 127, 32: up to 63).
// tl_maskers_sorted() decides, wave-uniformly, whether this form may be used.
template <int STEPS_T, int STEPS_N>
TL_FN void tl_mask_spans_sorted(const double *TL_RESTRICT mb, int ntone, int nnoise, double blo, double bhi, int &a0, int &a1, int &b0, int &b1)
{
    int tl = 0, th = 0, nl = 0, nh = 0;
    const double *nbk = mb + ntone;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int step = STEPS_T > STEPS_N ? STEPS_T : STEPS_N; step; step >>= 1) {
        const bool dt = step <= STEPS_T, dn = step <= STEPS_N;
        const int qtl = tl + step, qth = th + step, qnl = nl + step, qnh = nh + step;
        // reads past a list's end stay inside the wave's transform buffer (the masker arrays lie at its start) and are gated by the count tests
        const double vtl = dt ? mb[qtl - 1] : 0.0, vth = dt ? mb[qth - 1] : 0.0;                // (index <= 2 * STEPS_T - 2)
        const double vnl = dn ? nbk[qnl - 1] : 0.0, vnh = dn ? nbk[qnh - 1] : 0.0;
        if (dt) { tl = (qtl <= ntone && vtl <= blo) ? qtl : tl; th = (qth <= ntone && vth <= bhi) ? qth : th; }
        if (dn) { nl = (qnl <= nnoise && vnl <= blo) ? qnl : nl; nh = (qnh <= nnoise && vnh <= bhi) ? qnh : nh; }
    }
    const int nm = ntone + nnoise;
    a0 = th > tl ? tl : nm; a1 = th > tl ? th - 1 : -1;
    b0 = nh > nl ? ntone + nl : nm; b1 = nh > nl ? ntone + nh - 1 : -1;
}
// Are both masker lists ascending in bark?  (wave-uniform; not inside a lanes block)
TL_FN bool tl_maskers_sorted(const double *TL_RESTRICT mb, int ntone, int nnoise)
{
    PV(bool, bad);
    TL_LANES_BEGIN
    bool b = false;
    for (int q = 1 + lane; q < ntone + nnoise; q += 64) b = b || (q != ntone && mb[q] < mb[q - 1]);
    L(bad) = b;
    TL_LANES_END
    return TL_BALLOT(bad) == 0ull;
}
// Running minimum over rows [j0, j0 + n) in the reference's order and with its comparison (`if (m > v) m = v`), four rows
// per LDS round trip; a short last group repeats the last row, which changes nothing.  take_first: m starts as row j0.
TL_FN double tl_min_rows(const double *ltg, int j0, int n, double m, bool take_first)
{
    const int last = j0 + n - 1;
    for (int j = j0; j <= last; j += 4) {
        const double a = ltg[j], b = ltg[j + 1 <= last ? j + 1 : last], c = ltg[j + 2 <= last ? j + 2 : last], d = ltg[j + 3 <= last ? j + 3 : last];
        if (take_first && j == j0) m = a; else if (m > a) m = a;
        if (m > b) m = b;
        if (m > c) m = c;
        if (m > d) m = d;
    }
    return m;
}
// scalefactors transmitted for scfsi 0..3: 3, 2, 1, 2 (encode_new.c:1101, sfsPerScfsi) -- from a constant, not from memory
TL_FN int tl_sfs_count(unsigned scfsi) { return (int)((0x2123u >> (4u * (scfsi & 3u))) & 15u); }
TL_FN unsigned tl_sf_index_ref(const double *TL_RESTRICT sf, double cur_max)
{   // encode_new.c:208-218 as written there (the emulation build checks tl_sf_index against it)
    unsigned i = 32;
    for (unsigned l = 16; l; l >>= 1) { if (cur_max <= sf[i]) i += l; else i -= l; }
    if (cur_max > sf[i]) i--;
    return i;
}
// The same result without the chain of seven dependent table reads.  The table is decreasing, so the search returns
// (number of entries >= cur_max) - 1 (0 when there is none).  Entry i is 2^(1 - i/3) cut to 14 decimals (and [63] = 1e-20): with
// cur_max in [2^e, 2^(e+1)) and i0 = 3(1 - e), every entry above i0 is < 2^e and every entry below i0 - 3 is >= 2^(e+1);
// entry i0 - 3 itself stands for 2^(e+1) but may fall just short of it (the cut), so it is looked at together with the
// three entries in between: four reads, issued together.
TL_FN unsigned tl_sf_index(const double *TL_RESTRICT sf, double cur_max)
{
    const int e = (int)((tl_d2u(cur_max) >> 52) & 0x7ffu) - 1023;
    int i0 = 3 * (1 - e);
    i0 = i0 < 3 ? 3 : i0 > 63 ? 63 : i0;
    const double s3 = sf[i0 - 3], s2 = sf[i0 - 2], s1 = sf[i0 - 1], s0 = sf[i0];
    const int cnt = (i0 - 3) + (cur_max <= s3 ? 1 : 0) + (cur_max <= s2 ? 1 : 0) + (cur_max <= s1 ? 1 : 0) + (cur_max <= s0 ? 1 : 0);
    return (unsigned)(cnt > 0 ? cnt - 1 : 0);
}
TL_FN void tl_put_bits(uint32_t *frame, int pos, uint32_t val, int nbits)
{   // MSB-first bit field at bit offset `pos`; words are big-endian bit order (bitstream.c:130-150)
    if (nbits <= 0) return;
    int w = pos >> 5, o = pos & 31, room = 32 - o;
    if (nbits <= room) TL_ATOMIC_OR(&frame[w], val << (room - nbits));
    else {
        TL_ATOMIC_OR(&frame[w], val >> (nbits - room));
        TL_ATOMIC_OR(&frame[w + 1], val << (32 - (nbits - room)));
    }
}
// The same for a field of 1..48 bits (three codewords of a subband at once), without branches on the field's position:
// the left-aligned value, followed by 32 zero bits, shifted right by the offset inside the first word, is three words.
TL_FN void tl_put_bits48(uint32_t *frame, int pos, uint64_t val, int nbits)
{
    const int w = pos >> 5, o = pos & 31;
    const uint64_t top = val << (64 - nbits);
    TL_ATOMIC_OR(&frame[w], (uint32_t)(top >> (32 + o)));
    TL_ATOMIC_OR(&frame[w + 1], (uint32_t)(top >> o));
    if (o + nbits > 64) TL_ATOMIC_OR(&frame[w + 2], (uint32_t)(((top & 0xffffffffull) << 32) >> o));
}
TL_FN uint32_t tl_get_bit(const uint32_t *frame, int pos) { return (frame[pos >> 5] >> (31 - (pos & 31))) & 1u; }
TL_FN uint32_t tl_bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }
TL_FN unsigned tl_crc_upd(unsigned crc, unsigned data, int len, unsigned poly, unsigned top)
{   // crc.c:43-56 / :99-113
    for (int b = len - 1; b >= 0; b--) {
        unsigned carry = crc & top;
        crc <<= 1;
        if ((!carry) ^ (!((data >> b) & 1u))) crc ^= poly;
    }
    return crc;
}
