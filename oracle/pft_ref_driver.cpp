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
