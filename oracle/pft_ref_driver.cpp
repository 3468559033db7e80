// oracle/pft_ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// EDI PFT layer oracle.  contrib/edioutput/PFT.cpp itself cannot be compiled here: PFT.h pulls in Log.h, which needs
// PACKAGE_NAME from the autoconf-generated config.h (absent; no stand-ins are written).  What CAN be compiled where it
// lies is the arithmetic underneath: the reference's Reed-Solomon wrapper contrib/ReedSolomon.cpp with Phil Karn's
// contrib/fec/{init,encode,decode}_rs_char.c, and contrib/crc.c.  This driver therefore
//   * takes every parity byte and every CRC from that reference code, and
//   * restates the chunking / interleaving / PF-header logic of PFT::Protect (PFT.cpp:75-139), PFT::ProtectAndFragment
//     (:141-232) and PFT::Assemble (:234-320) statement by statement with the same containers (std::vector), so that the
//     device code -- which computes the same bytes with index arithmetic and a linear-map encoder -- is checked against an
//     independent formulation.  Parity status: Reed-Solomon + CRC pinned by the reference, layout pinned by restatement.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "ReedSolomon.h"
extern "C" {
#include "crc.h"
}

#define CEIL_DIV(a, b) (a % b == 0 ? a / b : a / b + 1)      /* PFT.cpp:46 */
static const size_t PARITYBYTES = 48;                      /* PFT.h */

struct PftRef {
    unsigned m_k, m_m;
    uint16_t m_pseq;
    size_t m_num_chunks;
    bool m_transport_header;
    uint16_t m_addr_source, m_dest_port;
};

static std::vector<uint8_t> protect(PftRef &p, std::vector<uint8_t> af_packet)
{
    std::vector<uint8_t> rs_block;
    p.m_num_chunks = CEIL_DIV(af_packet.size(), p.m_k);
    const size_t chunk_len = CEIL_DIV(af_packet.size(), p.m_num_chunks);
    const size_t zero_pad = p.m_num_chunks * chunk_len - af_packet.size();
    ReedSolomon rs_encoder(255, 207, false, 0x11d, 1);       // PFT.cpp:100-107
    for (size_t i = 0; i < zero_pad; i++) af_packet.push_back(0);
    for (size_t i = 0; i < af_packet.size(); i += chunk_len) {
        std::vector<uint8_t> chunk(207);
        std::vector<uint8_t> protection(PARITYBYTES);
        memcpy(&chunk.front(), &af_packet[i], chunk_len);
        rs_encoder.encode(&chunk.front(), &protection.front(), 207);
        chunk.resize(chunk_len);
        rs_block.insert(rs_block.end(), chunk.begin(), chunk.end());
        rs_block.insert(rs_block.end(), protection.begin(), protection.end());
    }
    return rs_block;
}

static std::vector<std::vector<uint8_t>> protect_and_fragment(PftRef &p, const std::vector<uint8_t> &af_packet)
{
    const bool enable_RS = (p.m_m > 0);
    if (enable_RS) {
        std::vector<uint8_t> rs_block = protect(p, af_packet);
        const size_t max_payload_size = (p.m_num_chunks * PARITYBYTES) / (p.m_m + 1);
        const size_t num_fragments = CEIL_DIV(rs_block.size(), max_payload_size);
        const size_t fragment_size = CEIL_DIV(rs_block.size(), num_fragments);
        std::vector<std::vector<uint8_t>> fragments(num_fragments);
        for (size_t i = 0; i < num_fragments; i++) {
            fragments[i].resize(fragment_size);
            for (size_t j = 0; j < fragment_size; j++) {
                const size_t ix = j * num_fragments + i;
                fragments[i][j] = ix < rs_block.size() ? rs_block[ix] : 0;
            }
        }
        return fragments;
    }
    const size_t max_payload_size = 1400;
    const size_t num_fragments = CEIL_DIV(af_packet.size(), max_payload_size);
    const size_t fragment_size = CEIL_DIV(af_packet.size(), num_fragments);
    std::vector<std::vector<uint8_t>> fragments(num_fragments);
    for (size_t i = 0; i < num_fragments; i++)
        for (size_t j = 0; j < fragment_size; j++) {
            const size_t ix = i * fragment_size + j;
            if (ix < af_packet.size()) fragments[i].push_back(af_packet.at(ix));
            else break;
        }
    return fragments;
}

// one stream: nframes AF packets -> fragments [nframes][max_frags][frag_stride], frag_len [nframes][max_frags], nfrag [nframes]
extern "C" int pftref_stream(const uint8_t *af, const int32_t *af_len, int nframes, int af_stride, unsigned fec, unsigned chunk_len,
                             int transport, unsigned addr_source, unsigned dest_port, uint16_t *pseq,
                             uint8_t *frags, int32_t *frag_len, int32_t *nfrag, int max_frags, int frag_stride)
{
    PftRef p{chunk_len, fec, *pseq, 0, transport != 0, (uint16_t)addr_source, (uint16_t)dest_port};
    for (int f = 0; f < nframes; f++) {
        const std::vector<uint8_t> af_packet(af + (size_t)f * af_stride, af + (size_t)f * af_stride + af_len[f]);
        std::vector<std::vector<uint8_t>> fragments = protect_and_fragment(p, af_packet);
        const bool enable_RS = (p.m_m > 0);
        unsigned findex = 0, fcount = (unsigned)fragments.size();
        const size_t chunk_len_b = enable_RS ? CEIL_DIV(af_packet.size(), p.m_num_chunks) : 0;
        const size_t zero_pad = enable_RS ? p.m_num_chunks * chunk_len_b - af_packet.size() : 0;
        if ((int)fragments.size() > max_frags) return -1;
        nfrag[f] = (int32_t)fragments.size();
        for (const auto &fragment : fragments) {
            std::string psync("PF");
            std::vector<uint8_t> packet(psync.begin(), psync.end());
            packet.push_back(p.m_pseq >> 8); packet.push_back(p.m_pseq & 0xFF);
            packet.push_back(findex >> 16); packet.push_back(findex >> 8); packet.push_back(findex & 0xFF);
            findex++;
            packet.push_back(fcount >> 16); packet.push_back(fcount >> 8); packet.push_back(fcount & 0xFF);
            unsigned plen = (unsigned)fragment.size();
            if (enable_RS) plen |= 0x8000;
            if (p.m_transport_header) plen |= 0x4000;
            packet.push_back(plen >> 8); packet.push_back(plen & 0xFF);
            if (enable_RS) { packet.push_back((uint8_t)chunk_len_b); packet.push_back((uint8_t)zero_pad); }
            if (p.m_transport_header) {
                packet.push_back(p.m_addr_source >> 8); packet.push_back(p.m_addr_source & 0xFF);
                packet.push_back(p.m_dest_port >> 8); packet.push_back(p.m_dest_port & 0xFF);
            }
            uint16_t crc = 0xffff;
            crc = crc16(crc, &(packet.front()), packet.size());
            crc ^= 0xffff;
            packet.push_back((crc >> 8) & 0xFF); packet.push_back(crc & 0xFF);
            packet.insert(packet.end(), fragment.begin(), fragment.end());
            if ((int)packet.size() > frag_stride) return -2;
            const size_t o = ((size_t)f * max_frags + (findex - 1));
            memcpy(frags + o * frag_stride, packet.data(), packet.size());
            frag_len[o] = (int32_t)packet.size();
        }
        p.m_pseq++;
    }
    *pseq = p.m_pseq;
    return 0;
}

// ---- receiving side -------------------------------------------------------------------------------------------------
// ODR-AudioEnc only sends; the check that what it sends can be RECEIVED is a receiver written from ETSI TS 102 821 clause 7
// (PF header fields, de-interleaving, erasure positions), with the error correction itself done by the reference's own
// decoder contrib/fec/decode_rs_char.c (Phil Karn's, the code ODR-DabMux's EDI input uses) and the AF CRC by contrib/crc.c.
// Fragments whose `present` flag is 0 are treated as lost: every byte they carried becomes an erasure.
// Returns the AF packet length (> 0), or < 0: -1 header inconsistency, -2 PF header CRC, -3 decoder failure, -4 AF CRC/LEN,
// -5 capacity.
extern "C" {
#include "fec/fec.h"
}
extern "C" int pftref_reassemble(const uint8_t *frags, const int32_t *frag_len, int nfrag, int frag_stride, const uint8_t *present,
                                 uint8_t *af_out, int af_cap, int *corrected)
{
    int fcount = -1, plen = -1, rsk = 0, rsz = 0, pseq = -1;
    bool fec = false, addr = false;
    if (corrected) *corrected = 0;
    for (int i = 0; i < nfrag; i++) {                         // every present fragment: header CRC, consistent fields
        if (!present[i]) continue;
        const uint8_t *p = frags + (size_t)i * frag_stride;
        if (p[0] != 'P' || p[1] != 'F') return -1;
        const int ps = (p[2] << 8) | p[3], fi = (p[4] << 16) | (p[5] << 8) | p[6], fc = (p[7] << 16) | (p[8] << 8) | p[9];
        const int pl = (p[10] << 8) | p[11];
        const bool f = pl & 0x8000, a = pl & 0x4000;
        const int hdr = 12 + (f ? 2 : 0) + (a ? 4 : 0);
        uint16_t crc = 0xffff;
        crc = crc16(crc, p, hdr);
        crc ^= 0xffff;
        if (p[hdr] != (crc >> 8) || p[hdr + 1] != (crc & 0xff)) return -2;
        if (fi != i || fc != nfrag || frag_len[i] != hdr + 2 + (pl & 0x3fff)) return -1;
        if (fcount < 0) { fcount = fc; plen = pl & 0x3fff; fec = f; addr = a; pseq = ps; if (f) { rsk = p[12]; rsz = p[13]; } }
        else if (ps != pseq || f != fec || a != addr || (f && (rsk != p[12] || rsz != p[13])) || (f && plen != (pl & 0x3fff))) return -1;
    }
    if (fcount < 0) return -1;
    const int hdr = 12 + (fec ? 2 : 0) + (addr ? 4 : 0) + 2;
    std::vector<uint8_t> af;
    if (!fec) {                                               // plain slices: nothing may be missing
        for (int i = 0; i < nfrag; i++) {
            if (!present[i]) return -3;
            const uint8_t *p = frags + (size_t)i * frag_stride;
            af.insert(af.end(), p + hdr, p + frag_len[i]);
        }
    } else {
        const size_t total = (size_t)fcount * plen;           // interleaved RS block (+ up to Fcount-1 padding bytes)
        const int cw = rsk + (int)PARITYBYTES;
        const size_t c = total / cw;
        std::vector<uint8_t> block(total, 0), erased(total, 0);
        for (int i = 0; i < nfrag; i++)
            for (int j = 0; j < plen; j++) {
                const size_t ix = (size_t)j * fcount + i;
                if (present[i]) block[ix] = frags[(size_t)i * frag_stride + hdr + j]; else erased[ix] = 1;
            }
        void *rs = init_rs_char(8, 0x11d, 1, 1, (int)PARITYBYTES, 0);
        if (!rs) return -3;
        for (size_t ci = 0; ci < c; ci++) {
            uint8_t word[255];
            int eras[255], ne = 0;
            memset(word, 0, sizeof word);
            for (int o = 0; o < cw; o++) {
                const size_t ix = ci * cw + o;
                const int pos = o < rsk ? o : 207 + (o - rsk);
                word[pos] = block[ix];
                if (erased[ix]) eras[ne++] = pos;
            }
            if (ne > (int)PARITYBYTES) { free_rs_char(rs); return -3; }      // beyond the code's erasure capacity (and the decoder's arrays)
            const int r = decode_rs_char(rs, word, eras, ne);
            if (r < 0) { free_rs_char(rs); return -3; }
            if (corrected) *corrected += r;
            af.insert(af.end(), word, word + rsk);
        }
        free_rs_char(rs);
        if ((int)af.size() < rsz) return -1;
        af.resize(af.size() - rsz);
    }
    // AF packet: "AF", LEN, ..., CRC over everything before it (contrib/edioutput/AFPacket.cpp:46-94)
    if (af.size() < 12 || af[0] != 'A' || af[1] != 'F') return -4;
    const size_t len = ((size_t)af[2] << 24) | ((size_t)af[3] << 16) | ((size_t)af[4] << 8) | af[5];
    if (len + 12 != af.size()) return -4;
    uint16_t crc = 0xffff;
    crc = crc16(crc, af.data(), af.size() - 2);
    crc ^= 0xffff;
    if (af[af.size() - 2] != (crc >> 8) || af[af.size() - 1] != (crc & 0xff)) return -4;
    if ((int)af.size() > af_cap) return -5;
    memcpy(af_out, af.data(), af.size());
    return (int)af.size();
}
