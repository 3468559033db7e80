// edi_pft.h -- EDI protection, fragmentation and transport layer (ETSI TS 102 821 clause 7):
// what edi::PFT::Assemble does to one AF packet before it goes out on UDP (contrib/edioutput/PFT.cpp:75-139 Protect,
// :141-232 ProtectAndFragment, :234-320 Assemble; Reed-Solomon RS(255,207) over GF(2^8), field polynomial 0x11d, first
// root 1, through contrib/ReedSolomon.cpp:43-60 and contrib/fec/encode_rs.h).
//
// Lane-SPMD like edi_af.h (include mp2_wave.h first).  One wavefront handles one AF packet:
//   * parity: the RS encoder is linear over GF(2^8), so the 48 parity bytes of a chunk are the XOR over its data bytes
//     d_i of d_i * M[i][.], M[i] = parity of the unit chunk e_i (built once on the host with the textbook LFSR).
//     Lane j < 48 owns parity byte j; products through log/antilog tables.
//   * fragments: byte j of fragment i is byte j*f + i of the RS block (chunk, parity, chunk, parity, ...), or with
//     FEC off a plain slice of the AF packet; every lane copies a strided share.
//   * PF header + CRC-16/CCITT: one lane per fragment.
#pragma once
#include <stdint.h>

#include "edi_types.h"

TL_FN uint32_t tl_crc16_ccitt_byte(uint32_t r, uint32_t by)
{
    r ^= by << 8;
    for (int b = 0; b < 8; b++) r = ((r << 1) & 0xffffu) ^ ((r & 0x8000u) ? 0x1021u : 0u);
    return r;
}

// Wave-private scratch (LDS on the device): the AF packet (read many times by the encoder) and the parity bytes.
struct TlPftScratch { uint8_t af[2048]; uint8_t par[TL_PFT_MAX_CHUNKS * TL_PFT_PARITY]; };
// Encoder tables as the kernel sees them (a workgroup-shared LDS copy on the device, TlTables in the emulation).
struct TlPftTables { const uint8_t *log, *exp, *mlog; };

TL_FN void tl_edi_pft_packet(const TlPftArgs &A, const TlPftTables &R, int s, int f, TlPftScratch &W)
{
    uint8_t *par = W.par;
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    const uint8_t *gaf = A.af + slot * (size_t)A.af_stride;
    uint32_t l = (uint32_t)A.af_len[slot];
    // Pseq counts the AF packets that exist.  A slot of length <= 0 is an ABSENT packet (tl_edi_af_packet leaves the surplus
    // unit slots of a stream that way): no fragments, no sequence number.  A packet longer than its slot / the staging
    // buffer is dropped (no fragments) but counted.  Nothing below divides by a zero chunk count or copies past W.af.
    PV(int, cnt);
    TL_LANES_BEGIN
    int c_ = 0;
    for (int v = lane; v < A.nframes; v += 64) c_ += (v < f && A.af_len[(size_t)v * (size_t)A.nstreams + (size_t)s] > 0) ? 1 : 0;
    L(cnt) = c_;
    TL_LANES_END
    const int before = TL_WAVE_SUM_I32(cnt);                         // packets of this stream before this one, in this call
    if (f == A.nframes - 1) {
        PV(int, cnt2);
        TL_LANES_BEGIN
        int c_ = 0;
        for (int v = lane; v < A.nframes; v += 64) c_ += A.af_len[(size_t)v * (size_t)A.nstreams + (size_t)s] > 0 ? 1 : 0;
        L(cnt2) = c_;
        TL_LANES_END
        const int total = TL_WAVE_SUM_I32(cnt2);
        TL_LANES_BEGIN
        if (lane == 0) A.pseq_out[s] = (uint16_t)(A.pseq[s] + total);
        TL_LANES_END
    }
    const uint32_t lmax = (uint32_t)A.af_stride < (uint32_t)sizeof(W.af) ? (uint32_t)A.af_stride : (uint32_t)sizeof(W.af);
    if ((int32_t)l <= 0 || l > lmax) {
        TL_LANES_BEGIN
        if (lane == 0) A.nfrag[slot] = 0;
        TL_LANES_END
        return;
    }
    TL_LANES_BEGIN
    for (uint32_t i = 4 * (uint32_t)lane; i < l; i += 256) *(uint32_t *)&W.af[i] = *(const uint32_t *)(gaf + i);     // af_stride is a multiple of 4
    TL_LANES_END
    const uint8_t *af = W.af;
    const bool rs = A.fec > 0;
    uint32_t c = 0, k = 0, z = 0, total, nfr, fsz;
    if (rs) {                                                        // PFT.cpp:81-98,166-176
        c = (l + (uint32_t)A.chunk_len - 1) / (uint32_t)A.chunk_len;
        k = (l + c - 1) / c;
        z = c * k - l;
        total = c * (k + TL_PFT_PARITY);
        const uint32_t smax = (c * TL_PFT_PARITY) / ((uint32_t)A.fec + 1);
        nfr = (total + smax - 1) / smax;
        fsz = (total + nfr - 1) / nfr;
        // parity of every chunk, zero padded at the END to 207 bytes (PFT.cpp:105-124): padding contributes nothing
        for (uint32_t ci = 0; ci < c; ci++) {
            TL_LANES_BEGIN
            if (lane < TL_PFT_PARITY) {
                uint32_t acc = 0;
                for (uint32_t i0 = 0; i0 < k; i0 += 8) {             // eight independent products per round trip
#ifndef TL_EMULATE
#pragma unroll
#endif
                    for (uint32_t q = 0; q < 8; q++) {
                        const uint32_t i = i0 + q, pos = ci * k + i;
                        const uint32_t d = (i < k && pos < l) ? af[pos] : 0u;
                        const uint32_t lm = R.mlog[(i < k ? i : 0) * TL_PFT_PARITY + lane];
                        const uint32_t e = R.exp[R.log[d] + lm];      // log[0] = 255 and mlog = 255 for a zero coefficient:
                        acc ^= (d != 0 && lm != 255) ? e : 0u;        // the index stays inside exp[512], the product is dropped
                    }
                }
                par[ci * TL_PFT_PARITY + lane] = (uint8_t)acc;
            }
            TL_LANES_END
        }
    } else {                                                         // PFT.cpp:196-229
        total = l;
        nfr = (l + 1399) / 1400;
        fsz = (l + nfr - 1) / nfr;
    }
    uint8_t *out = A.frags + slot * (size_t)A.max_frags * (size_t)A.frag_stride;
    const uint32_t hdr = 12 + (rs ? 2 : 0) + (A.transport ? 4 : 0) + 2;
    const uint32_t pseq = (uint32_t)(uint16_t)(A.pseq[s] + before);
    // ---- payloads ----
    TL_LANES_BEGIN
    for (uint32_t idx = (uint32_t)lane; idx < nfr * fsz; idx += 64) {
        const uint32_t i = idx / fsz, j = idx - i * fsz;
        uint32_t by = 0;
        bool present = true;
        if (rs) {
            const uint32_t ix = j * nfr + i;                          // interleaved, PFT.cpp:183-192
            if (ix < total) {
                const uint32_t ci = ix / (k + TL_PFT_PARITY), off = ix - ci * (k + TL_PFT_PARITY);
                if (off < k) { const uint32_t pos = ci * k + off; by = pos < l ? af[pos] : 0u; }
                else by = par[ci * TL_PFT_PARITY + (off - k)];
            }
        } else {
            const uint32_t ix = i * fsz + j;                          // plain slices, PFT.cpp:214-226
            present = ix < l;
            if (present) by = af[ix];
        }
        if (present) out[(size_t)i * A.frag_stride + hdr + j] = (uint8_t)by;
    }
    TL_LANES_END
    // ---- PF headers, PFT.cpp:254-308 ----
    TL_LANES_BEGIN
    for (uint32_t i = (uint32_t)lane; i < nfr; i += 64) {
        uint32_t plen_bytes = fsz;
        if (!rs) { const uint32_t lo = i * fsz; plen_bytes = lo >= l ? 0u : (l - lo < fsz ? l - lo : fsz); }
        uint32_t plen = plen_bytes | (rs ? 0x8000u : 0u) | (A.transport ? 0x4000u : 0u);
        const uint32_t n = hdr - 2;                                   // header bytes covered by the CRC
        uint8_t *o = out + (size_t)i * A.frag_stride;
        uint32_t r = 0xffffu;
        for (uint32_t q = 0; q < n; q++) {
            uint32_t by;
            const uint32_t t = q - 12 - (rs ? 2u : 0u);               // index inside the optional transport part
            if (q < 2) by = q == 0 ? 'P' : 'F';
            else if (q < 4) by = (pseq >> (8 * (3 - q))) & 0xffu;
            else if (q < 7) by = (i >> (8 * (6 - q))) & 0xffu;
            else if (q < 10) by = (nfr >> (8 * (9 - q))) & 0xffu;
            else if (q < 12) by = (plen >> (8 * (11 - q))) & 0xffu;
            else if (rs && q < 14) by = q == 12 ? (k & 0xffu) : (z & 0xffu);
            else by = t < 2 ? ((uint32_t)A.addr_source >> (8 * (1 - t))) & 0xffu : ((uint32_t)A.dest_port >> (8 * (3 - t))) & 0xffu;
            o[q] = (uint8_t)by;
            r = tl_crc16_ccitt_byte(r, by);
        }
        r ^= 0xffffu;
        o[n] = (uint8_t)(r >> 8); o[n + 1] = (uint8_t)r;
        A.frag_len[slot * (size_t)A.max_frags + i] = (int32_t)(hdr + plen_bytes);
    }
    if (lane == 0) A.nfrag[slot] = (int32_t)nfr;
    TL_LANES_END
}
