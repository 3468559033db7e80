// edi_af.h -- EDI "AF packet" of one encoded frame: the TAG packet that
// src/Outputs.cpp:194-261 (EDI::write_frame) assembles -- *ptr("DSTI"), dsti, ss0001, ODRa and, every
// ten seconds, ODRv (contrib/edioutput/TagItems.cpp:38-66,202-263,304-356,381-441) -- wrapped by
// contrib/edioutput/AFPacket.cpp:46-94 (SYNC "AF", LEN, SEQ, AR = CRC flag | version 1.0, PT 'T', payload,
// CRC-16/CCITT with init 0xffff and final inversion, contrib/crc.c:247-255).
// This is what an EDI/TCP destination receives; the PFT layer for UDP destinations (Reed-Solomon + fragmentation,
// contrib/edioutput/PFT.cpp) is csrc/edi_pft.h, which takes these packets as its input.
//
// Written in the same lane-SPMD style as mp2_wave.h (include it first): compiled by hipcc for gfx950 and, with
// -DTL_EMULATE, as a lane loop for the CPU tests.  One wavefront builds one packet; the sender state (timestamp,
// sequence and frame counters, version cadence) advances per packet, which is a few integer operations, so the wave of
// packet e simply replays e+1 advances from the state at the start of the call.
//
// UNITS.  What ODR-AudioEnc sends is not the MP2 frame but pieces of 3 * bitrate bytes = 24 ms of the byte stream
// (src/odr-audioenc.cpp:1211-1219: "ODR-DabMux expects frames of length 3*bitrate"), each through send_frame() with its own
// +24 ms timestamp, DLFC and SEQ (src/Outputs.cpp:194-261).  At 48 kHz a frame IS one unit; an MPEG-2 LSF frame carries two
// (24 kHz, 48 ms) or three (16 kHz, 72 ms).  Packet slot v = f * max_upf + u holds unit u of frame f; a stream with fewer
// units per frame than the batch's maximum leaves its surplus slots absent (length 0), and only present units advance the
// sender state.
#pragma once
#include <stdint.h>
#include <string.h>

#include "edi_types.h"

// everything that is the same for all bytes of one packet
struct TlEdiFrame {
    const uint8_t *payload; const uint8_t *version;
    uint32_t n, vlen, taglen, seconds, tsta, uptime;
    uint16_t seq, dsti_hdr; int16_t left, right;
    uint8_t atstf, utco, with_version;
    uint32_t pay_lo;                 // packet offset of the first frame byte
};

// byte `pos` of the AF packet before the CRC (pos < 10 + taglen)
TL_FN uint8_t tl_edi_byte(const TlEdiFrame &F, uint32_t pos)
{
    if (pos < 10) {                                                  // AFPacket.cpp:55-71
        switch (pos) {
        case 0: return 'A'; case 1: return 'F';
        case 2: return (uint8_t)(F.taglen >> 24); case 3: return (uint8_t)(F.taglen >> 16);
        case 4: return (uint8_t)(F.taglen >> 8); case 5: return (uint8_t)F.taglen;
        case 6: return (uint8_t)(F.seq >> 8); case 7: return (uint8_t)F.seq;
        case 8: return 0x80 | 0x10; default: return 'T';
        }
    }
    uint32_t p = pos - 10;
    if (p < 16) {                                                    // TagStarPTR("DSTI"), TagItems.cpp:46-66
        // "*ptr" 00 00 00 40 | "DSTI" 00 00 00 00, little-endian packed (a constant in registers, not an array in memory)
        const uint64_t lo = 0x400000007274702aull, hi = 0x0000000049545344ull;
        return (uint8_t)((p < 8 ? lo >> (8 * p) : hi >> (8 * (p - 8))) & 0xffu);
    }
    p -= 16;
    const uint32_t dsti_len = 2u + (F.atstf ? 8u : 0u);
    if (p < 8 + dsti_len) {                                          // TagDSTI, TagItems.cpp:202-263 (stihf = rfadf = 0)
        const uint32_t bits = dsti_len * 8;
        switch (p) {
        case 0: return 'd'; case 1: return 's'; case 2: return 't'; case 3: return 'i';
        case 4: return (uint8_t)(bits >> 24); case 5: return (uint8_t)(bits >> 16); case 6: return (uint8_t)(bits >> 8); case 7: return (uint8_t)bits;
        case 8: return (uint8_t)(F.dsti_hdr >> 8); case 9: return (uint8_t)F.dsti_hdr;
        case 10: return F.utco;
        case 11: return (uint8_t)(F.seconds >> 24); case 12: return (uint8_t)(F.seconds >> 16);
        case 13: return (uint8_t)(F.seconds >> 8); case 14: return (uint8_t)F.seconds;
        case 15: return (uint8_t)(F.tsta >> 16); case 16: return (uint8_t)(F.tsta >> 8); default: return (uint8_t)F.tsta;
        }
    }
    p -= 8 + dsti_len;
    if (p < 11 + F.n) {                                              // TagSSm id 1, TagItems.cpp:304-356 (istc = 0)
        const uint32_t bits = (3 + F.n) * 8;
        if (p >= 11) return F.payload[p - 11];
        switch (p) {
        case 0: return 's'; case 1: return 's'; case 2: return 0; case 3: return 1;
        case 4: return (uint8_t)(bits >> 24); case 5: return (uint8_t)(bits >> 16); case 6: return (uint8_t)(bits >> 8); case 7: return (uint8_t)bits;
        default: return 0;
        }
    }
    p -= 11 + F.n;
    if (p < 12) {                                                    // TagODRAudioLevels, TagItems.cpp:421-441
        switch (p) {
        case 0: return 'O'; case 1: return 'D'; case 2: return 'R'; case 3: return 'a';
        case 4: case 5: case 6: return 0; case 7: return 0x20;
        case 8: return (uint8_t)((uint16_t)F.left >> 8); case 9: return (uint8_t)F.left;
        case 10: return (uint8_t)((uint16_t)F.right >> 8); default: return (uint8_t)F.right;
        }
    }
    p -= 12;
    {                                                                // TagODRVersion, TagItems.cpp:387-413
        const uint32_t bits = (F.vlen + 4) * 8;
        if (p < 4) return (uint8_t)((0x7652444fu >> (8 * p)) & 0xffu);          // "ODRv"
        if (p < 8) return (uint8_t)(bits >> (8 * (7 - p)));
        if (p < 8 + F.vlen) return F.version[p - 8];
        return (uint8_t)(F.uptime >> (8 * (3 - (p - 8 - F.vlen))));
    }
}

// AF packet of slot v (0-based within this call: unit v % max_upf of frame v / max_upf) of stream s.
TL_FN void tl_edi_af_packet(const TlEdiArgs &A, int s, int v)
{
    TlEdiState st = A.state[s];
    const uint32_t n = (uint32_t)A.unit_bytes[s];
    const int upf = A.frame_bytes[s] / (int)n, f = v / A.max_upf, u = v - f * A.max_upf;
    const size_t pslot = (size_t)v * (size_t)A.nstreams + (size_t)s;
    // frames of this stream that exist: before frame f, and frame f itself (a slot without a frame sends nothing and leaves the
    // sender state -- SEQ, DLFC, timestamp -- where it is)
    int before = f; bool present = true;
    if (A.frame_len) {
        before = 0;
        for (int k = 0; k < f; k++) before += A.frame_len[(size_t)k * (size_t)A.nstreams + (size_t)s] != 0 ? 1 : 0;
        present = A.frame_len[(size_t)f * (size_t)A.nstreams + (size_t)s] != 0;
    }
    if (u >= upf || !present) {                                       // this stream has no unit here
        TL_LANES_BEGIN
        if (lane == 0) A.pkt_len[pslot] = 0;
        TL_LANES_END
        if (!present && f == A.nframes - 1 && u == upf - 1) {         // ... but the call's last slot still owes the state after the call
            for (int k = 0; k < before * upf; k++) {
                st.timestamp += 24u << 14;
                if (st.timestamp > 0xf9FFffu) { st.timestamp -= 0xfa0000u; st.edi_time += 1; st.num_seconds_sent++; }
                st.dlfc = (uint16_t)((st.dlfc + 1) % 5000);
                if (st.send_version_at_time < st.edi_time) st.send_version_at_time += 10;
                st.seq = (uint16_t)(st.seq + 1);
            }
            TL_LANES_BEGIN
            if (lane == 0) A.state_out[s] = st;
            TL_LANES_END
        }
        return;
    }
    const int e = before * upf + u;                                   // units of this stream before this one, in this call
    {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
        // ---- sender state, Outputs.cpp:214-257; units 0..e-1 only advance it ----
        uint16_t dlfc = 0, seq = 0; uint8_t with_version = 0;
        for (int k = 0; k <= e; k++) {
            st.timestamp += 24u << 14;                               // 24 ms at timestamp level 2
            if (st.timestamp > 0xf9FFffu) { st.timestamp -= 0xfa0000u; st.edi_time += 1; st.num_seconds_sent++; }
            dlfc = st.dlfc; st.dlfc = (uint16_t)((st.dlfc + 1) % 5000);
            with_version = 0;
            if (st.send_version_at_time < st.edi_time) { st.send_version_at_time += 10; with_version = 1; }
            seq = st.seq; st.seq = (uint16_t)(st.seq + 1);
        }
        TlEdiFrame F;
        F.payload = A.frames + slot * (size_t)A.out_stride + (size_t)u * n;
        F.version = A.version; F.vlen = (uint32_t)A.version_len; F.n = n;
        F.atstf = st.tist ? 1 : 0;
        F.utco = (uint8_t)(st.tai_utc_offset - 32);                  // TagDSTI::set_edi_time, TagItems.cpp:265-274
        F.seconds = (uint32_t)(st.edi_time - 946684800 + F.utco);
        F.tsta = st.timestamp & 0xffffffu;
        F.dsti_hdr = (uint16_t)((dlfc % 250) | ((dlfc / 250) << 8) | (F.atstf << 14));
        F.left = A.levels ? A.levels[slot * 2] : 0; F.right = A.levels ? A.levels[slot * 2 + 1] : 0;
        F.with_version = with_version;
        F.uptime = st.num_seconds_sent;
        F.seq = seq;
        F.taglen = 16 + (10 + (F.atstf ? 8 : 0)) + (11 + n) + 12 + (F.with_version ? 12 + F.vlen : 0);
        F.pay_lo = 10 + 16 + (10 + (F.atstf ? 8 : 0)) + 11;
        const uint32_t body = 10 + F.taglen;                         // bytes covered by the CRC
        uint8_t *pkt = A.pkts + pslot * (size_t)A.pkt_stride;

        // ---- bytes + CRC.  Lane l owns bytes [l*C, l*C+C), C a multiple of 4 (<= 32): it builds them once as up to eight
        //      words and folds them into its own CRC remainder (lane 0 carries the 0xffff preset).  The register update is
        //      linear over GF(2), so the remainders combine as sum r_l * x^(8 * bytes after the chunk) mod P; the two CRC
        //      bytes are then patched into the word(s) that hold them and every lane stores whole words. ----
        const uint32_t C = 4 * ((body + 2 + 255) / 256);
        PV(uint32_t, part); PA(uint32_t, wd, 8);
        TL_LANES_BEGIN
        uint32_t acc = 0;
        const uint32_t p0 = (uint32_t)lane * C;
        uint32_t r = lane == 0 ? 0xffffu : 0u;
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (uint32_t wi = 0; wi < 8; wi++) {
            uint32_t word = 0;
            if (4 * wi < C) {
                const uint32_t q = p0 + 4 * wi;
                const bool inside = q >= F.pay_lo && q + 4 <= F.pay_lo + F.n;      // four frame bytes: one (unaligned) word load
                uint32_t pw = 0;
                if (inside) memcpy(&pw, F.payload + (q - F.pay_lo), 4);
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t pos = q + k;
                    if (pos < body) {
                        const uint32_t by = inside ? (pw >> (8 * k)) & 0xffu : tl_edi_byte(F, pos);
                        r ^= by << 8;
                        for (int b = 0; b < 8; b++) r = ((r << 1) & 0xffffu) ^ ((r & 0x8000u) ? 0x1021u : 0u);
                        word |= by << (8 * k);
                    }
                }
            }
            L(wd)[wi] = word;
        }
        if (p0 < body) {
            const uint32_t p1 = p0 + C < body ? p0 + C : body;
            uint32_t xp = A.xpow8[body - p1];
            for (int b = 0; b < 16; b++) {                           // acc = r * x^(8*(body-p1)) mod P
                acc ^= ((r >> b) & 1u) ? xp : 0u;
                xp = ((xp << 1) & 0xffffu) ^ ((xp & 0x8000u) ? 0x1021u : 0u);
            }
        }
        L(part) = acc;
        TL_LANES_END
        const uint32_t crc = (TL_WAVE_XOR_U32(part) ^ 0xffffu) & 0xffffu;
        TL_LANES_BEGIN
        const uint32_t p0 = (uint32_t)lane * C;
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (uint32_t wi = 0; wi < 8; wi++) {
            const uint32_t q = p0 + 4 * wi;
            if (4 * wi < C && q < body + 2) {
                uint32_t word = L(wd)[wi];
                if (body >= q && body < q + 4) word |= (crc >> 8) << (8 * (body - q));
                if (body + 1 >= q && body + 1 < q + 4) word |= (crc & 0xffu) << (8 * (body + 1 - q));
                *(uint32_t *)(pkt + q) = word;                        // pkt_stride and C are multiples of 4
            }
        }
        if (lane == 0) A.pkt_len[pslot] = (int32_t)(body + 2);
        TL_LANES_END
    }
    if (f == A.nframes - 1 && u == upf - 1) {                         // the stream's last unit of the call
        TL_LANES_BEGIN
        if (lane == 0) A.state_out[s] = st;
        TL_LANES_END
    }
}
