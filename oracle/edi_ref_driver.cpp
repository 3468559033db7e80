// oracle/edi_ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the reference's own EDI classes (contrib/edioutput/TagItems.cpp, TagPacket.cpp, AFPacket.cpp and
// contrib/crc.c, compiled where they lie by oracle/Makefile into oracle/_ref/libedi_ref.so) the way
// EDI::write_frame does (src/Outputs.cpp:194-261), minus the sockets: the wall clock and the TAI offset are
// arguments and the AF packet is returned instead of being handed to edi::Sender (Transport.cpp:126-132 calls
// AFPacketiser::Assemble on the same TagPacket).  The tag contents, their order, the AF header, sequence
// numbers and the CRC therefore come from the reference code itself; only this call sequence is restated.
#include <cstdint>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "AFPacket.h"
#include "TagItems.h"
#include "TagPacket.h"

struct EdiRefState {                 // same layout as TlEdiState / tlb_edi_state
    int64_t edi_time, send_version_at_time;
    uint32_t timestamp, num_seconds_sent;
    int32_t tai_utc_offset;
    uint16_t seq, dlfc;
    uint8_t tist, pad_[7];
};

extern "C" int ediref_stream(const uint8_t *frames, int nframes, int frame_bytes, int frame_stride, const int16_t *levels /* [nframes][2] or null */,
                             EdiRefState *st, const char *version, int version_len, uint8_t *pkts, int pkt_stride, int32_t *pkt_len)
{
    edi::TagDSTI tagDSTI;            // EDI::m_edi_tagDSTI
    tagDSTI.dlfc = st->dlfc;
    edi::AFPacketiser packetiser;    // edi::Sender::edi_afPacketiser
    packetiser.OverrideSeq(st->seq);
    const std::string odr_version(version, version + version_len);
    for (int f = 0; f < nframes; f++) {
        edi::TagStarPTR tagStarPtr("DSTI");
        tagDSTI.stihf = false;
        tagDSTI.atstf = st->tist != 0;
        st->timestamp += 24 << 14;
        if (st->timestamp > 0xf9FFff) { st->timestamp -= 0xfa0000; st->edi_time += 1; st->num_seconds_sent++; }
        tagDSTI.set_edi_time((std::time_t)st->edi_time, st->tai_utc_offset);
        tagDSTI.tsta = st->timestamp & 0xffffff;
        tagDSTI.rfadf = false;
        edi::TagSSm tagPayload;
        tagPayload.istd_data = frames + (size_t)f * frame_stride;
        tagPayload.istd_length = (size_t)frame_bytes;
        edi::TagODRAudioLevels tagLevels(levels ? levels[2 * f] : 0, levels ? levels[2 * f + 1] : 0);
        edi::TagODRVersion tagVersion(odr_version, st->num_seconds_sent);
        edi::TagPacket tagpacket(0);  // edi::configuration_t::tagpacket_alignment default
        tagpacket.tag_items.push_back(&tagStarPtr);
        tagpacket.tag_items.push_back(&tagDSTI);
        tagpacket.tag_items.push_back(&tagPayload);
        tagpacket.tag_items.push_back(&tagLevels);
        if (st->send_version_at_time < st->edi_time) { st->send_version_at_time += 10; tagpacket.tag_items.push_back(&tagVersion); }
        const edi::AFPacket af = packetiser.Assemble(tagpacket);
        if ((int)af.size() > pkt_stride) return -1;
        memcpy(pkts + (size_t)f * pkt_stride, af.data(), af.size());
        pkt_len[f] = (int32_t)af.size();
        st->seq = (uint16_t)(st->seq + 1);
    }
    st->dlfc = tagDSTI.dlfc;
    return 0;
}
