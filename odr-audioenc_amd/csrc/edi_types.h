// edi_types.h -- the plain records of the egress stage (EDI AF packets, PFT fragments): what the host side fills in and the kernels of
// edi_af.h / edi_pft.h read.  No device code here, so that the host translation units (tlb_egress.cpp, tlb_tick.cpp) need nothing else.
#pragma once
#include <stdint.h>

// Sender state of one stream: EDI members of src/Outputs.h:150-164 + AFPacketiser::m_seq + TagDSTI::dlfc.
struct TlEdiState {
    int64_t edi_time;                // m_edi_time (POSIX seconds)
    int64_t send_version_at_time;    // m_send_version_at_time
    uint32_t timestamp;              // m_timestamp (level-2 units of 1/16384000 s, 24 bits used on the wire)
    uint32_t num_seconds_sent;       // m_num_seconds_sent
    int32_t tai_utc_offset;          // ClockTAI offset handed to TagDSTI::set_edi_time
    uint16_t seq;                    // AFPacketiser::m_seq
    uint16_t dlfc;                   // TagDSTI::dlfc, modulo 5000
    uint8_t tist;                    // m_tist -> atstf
    uint8_t pad_[7];
};

struct TlEdiArgs {
    const uint8_t *frames;           // [nframes][nstreams][out_stride] whole frames (tlb_encode_* output)
    const int16_t *levels;           // [nframes][nstreams][2] audio levels (tlb_ingest_* peaks) or null -> 0
    const TlEdiState *state;         // [nstreams] sender state before the first frame of this call
    TlEdiState *state_out;           // [nstreams] state after the last frame (a different array: every packet reads `state`)
    const uint8_t *version;          // ODRv version string (not terminated)
    const uint16_t *xpow8;           // x^(8k) mod (x^16+x^12+x^5+1), k = 0..TL_EDI_XPOW-1
    const int32_t *frame_bytes;      // [nstreams]
    const int32_t *unit_bytes;       // [nstreams] 3 * kbps: what one send_frame() carries (divides frame_bytes)
    uint8_t *pkts;                   // [nframes * max_upf][nstreams][pkt_stride]
    int32_t *pkt_len;                // [nframes * max_upf][nstreams]; 0 = absent slot
    const int32_t *frame_len;        // [nframes][nstreams] length of the frame in each input slot, 0 = the stream has no frame there (just created,
                                     // reset or reconfigured: tlb_encode_device_len's d_out_len), or null = every slot holds a frame
    int32_t nstreams, nframes, out_stride, pkt_stride, version_len, max_upf;
};
#define TL_EDI_XPOW 2048             // longest AF packet: 10 + 16 + 18 + 11 + 1728 + 12 + 12 + version < 2048 bytes
#define TL_EDI_MAX_VERSION 64

#define TL_PFT_MAX_CHUNKS 10          // AF packets < 2048 bytes, chunks of up to 207
#define TL_PFT_PARITY 48

struct TlPftArgs {
    const uint8_t *af;                // [nframes][nstreams][af_stride] AF packets (tlb_edi_af_* output)
    const int32_t *af_len;            // [nframes][nstreams]
    const uint16_t *pseq;             // [nstreams] PFT::m_pseq before the first packet of this call
    uint16_t *pseq_out;               // [nstreams] after the last one (a different array)
    uint8_t *frags;                   // [nframes][nstreams][max_frags][frag_stride]: PF header + payload
    int32_t *frag_len;                // [nframes][nstreams][max_frags]
    int32_t *nfrag;                   // [nframes][nstreams]
    int32_t nstreams, nframes, af_stride, max_frags, frag_stride;
    int32_t fec;                      // m: fragments that can be lost (0 = no Reed-Solomon, fragmentation only)
    int32_t chunk_len;                // k_max of the configuration (<= 207)
    int32_t transport;                // 1: PF header carries source/destination
    int32_t addr_source, dest_port;
};
