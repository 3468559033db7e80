/* toolame_batch.h -- C-ABI of the MI355X-native batched DAB MP2 (MPEG-1/2 Layer II) encoder.
 *
 * Drop-in boundary.  The library exports two groups of symbols:
 *
 *  (1) The reference's own nine-function ABI, unchanged, so src/odr-audioenc.cpp:89-92,686-735,
 *      1135-1163 links against this library exactly as it links against libtoolame-dab.a.
 *      Declarations: /root/reference/libtoolame-dab/toolame.h:13-48, export list
 *      /root/reference/libtoolame-dab.sym.  They drive stream 0 of a private one-stream batch and
 *      reproduce toolame_encode_frame()'s bursty 4096-byte output cadence (bitstream.c:46-71).
 *
 *  (2) A handle-based batched API (tlb_*) -- what the reference cannot offer because all of its
 *      state is process-global (toolame.c:24-26,89-118): N independent streams, mixed
 *      configurations, whole-frame output, device-resident buffers and an explicit HIP stream.
 *
 * All pointers are plain C pointers; "d_" parameters are HIP device pointers.  No torch types.
 * Every function returns 0 on success and non-zero on error unless stated otherwise
 * (toolame.h:9-10).  The library needs a gfx950 GPU: there is no CPU fallback, tlb_create()
 * fails with TLB_ERR_NO_DEVICE when none is usable.
 */
#ifndef TOOLAME_BATCH_H
#define TOOLAME_BATCH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * (1) legacy ABI -- replaces libtoolame-dab/toolame.h:13-48 one for one
 * ------------------------------------------------------------------------------------------ */
int toolame_init(void);                                     /* toolame.c:120 */
int toolame_finish(unsigned char *output_buffer, size_t output_buffer_size);   /* toolame.c:155 -> bytes written */
int toolame_enable_byteswap(void);                          /* toolame.c:168 */
int toolame_set_channel_mode(const char mode);              /* toolame.c:174  's','d','j','m' */
int toolame_set_psy_model(int new_model);                   /* toolame.c:202  0..3 */
int toolame_set_bitrate(int brate);                         /* toolame.c:212  kbps */
int toolame_set_samplerate(long sample_rate);               /* toolame.c:239  Hz */
int toolame_set_pad(int pad_len);                           /* toolame.c:250 */
int toolame_encode_frame(short buffer[2][1152], unsigned char *xpad_data, size_t xpad_len,
                         unsigned char *output_buffer, size_t output_buffer_size);   /* toolame.c:267 -> bytes written */

/* ---------------------------------------------------------------------------------------------
 * (2) batched API
 * ------------------------------------------------------------------------------------------ */
#define TLB_MAX_XPAD 256           /* X-PAD + F-PAD bytes per frame: every padlen the caller accepts, 0..255 (src/odr-audioenc.cpp:566) */
#define TLB_SAMPLES_PER_FRAME 1152

enum {
    TLB_OK = 0,
    TLB_ERR_SAMPLERATE = 1,        /* SmpFrqIndex, common.c:118-144; the egress calls: a rate whose frames are no whole number of 24-ms units */
    TLB_ERR_MODE = 2,              /* toolame.c:195-197 */
    TLB_ERR_PSY = 3,               /* toolame.c:204-207 */
    TLB_ERR_BITRATE = 4,           /* common.c:95-116 (the reference exit(-1)s; we return an error) */
    TLB_ERR_PAD = 5,               /* toolame.c:252-255 */
    TLB_ERR_NO_DEVICE = 16,
    TLB_ERR_HIP = 17,
    TLB_ERR_ARG = 18,
};

/* The six knobs odr-audioenc sets (src/odr-audioenc.cpp:687-722), per stream. */
typedef struct {
    long samplerate;               /* 48000, 32000, 24000, 16000, 44100, 22050 */
    char mode;                     /* 's' stereo, 'j' joint stereo, 'd' dual channel, 'm' mono */
    int bitrate;                   /* kbps; 0 = the reference default, bitrate[version][10] (toolame.c:217-218): 192 (MPEG-1) / 96 (LSF) */
    int psy_model;                 /* 0, 1, 2, 3 (toolame.c:202-210); 4 = psycho_4.c, an extension of this API only: the
                                      reference implements it (toolame.c:384-391) but its setter refuses it */
    int pad_len;                   /* toolame_set_pad(): upper bound of xpad_len, 0..TLB_MAX_XPAD; must leave room for header, CRC and
                                      bit allocation in the frame (else TLB_ERR_PAD) */
} tlb_stream_config;

typedef struct tlb_batch tlb_batch;

int tlb_device_count(void);
/* Create `nstreams` encoders on HIP device `device`.  On failure returns NULL and stores a TLB_ERR_*
 * code in *err (if err != NULL). */
tlb_batch *tlb_create(int device, int nstreams, const tlb_stream_config *cfgs, int *err);
void tlb_destroy(tlb_batch *b);
int tlb_reset(tlb_batch *b);                       /* every stream back to the state right after tlb_create(); also the way out after a launch or a
                                                      reconfiguration failed half way (TLB_ERR_HIP from an encode call: the batch refuses work until reset).
                                                      It re-derives the device's stream tables, kernel lists and mono pairing from the host's first; if THAT
                                                      fails it returns the code and the batch stays refused (tlb_destroy is then the way out) */

/* Life cycle of ONE stream inside a live batch.  The reference's unit of restart is the stream -- toolame_init() zeroes one encoder
 * (toolame.c:120-153), toolame_finish() ends one (:155-166), the six setters reconfigure one (toolame.h:13-48), and odr-audioenc
 * restarts a failed input without touching anything else (src/odr-audioenc.cpp:875-902).  With thousands of streams in one batch the
 * same three operations exist per stream.  Each waits for the launches already queued on the batch, then acts; none changes a byte
 * any OTHER stream produces.  After any of them the stream is as right after tlb_create() -- the bytes of a reference process started
 * afresh, not those of an in-process toolame_init() --: its next input frame is its frame 0
 * (history of zeros, no pending frame: the output slot of the call that analyses it is empty, length 0 in the _len variants).
 *   tlb_stream_reset        = a freshly started reference PROCESS for that stream: drops the pending frame, zeroes filterbank and
 *                             model history.  (The reference's own toolame_init(), toolame.c:120-153, called a second time inside one
 *                             process, keeps the file-scope statics of psycho_1/2 -- savebuf, r, phi_sav, lthr, init -- and the
 *                             filterbank FIFO: a re-initialised reference encoder remembers its past, this one does not.)
 *   tlb_stream_finish       = toolame_finish(): copies the pending frame (the bytes still inside the encoder) to out, returns their
 *                             number (0: none yet; a too small buffer gets a truncated copy like the reference's); < 0: -TLB_ERR_*.
 *   tlb_stream_reconfigure  = the setters + toolame_init(): a new sample rate / mode / bitrate / psy model / PAD length.  The frames
 *                             of the new configuration must fit tlb_out_stride() and its units per frame tlb_egress_max_units_per_frame()
 *                             as the batch was created with (else TLB_ERR_ARG / TLB_ERR_SAMPLERATE, nothing changed); illegal settings
 *                             return the setter's error code. */
int tlb_stream_reset(tlb_batch *b, int stream);
int tlb_stream_finish(tlb_batch *b, int stream, uint8_t *out, size_t out_size);
int tlb_stream_reconfigure(tlb_batch *b, int stream, const tlb_stream_config *cfg);

int tlb_nstreams(const tlb_batch *b);
int tlb_frame_bytes(const tlb_batch *b, int stream);   /* 144000*kbps/fs: 384 @128k/48k, 576 @192k/48k ...; at 44.1 / 22.05 kHz the frames
                                                          WITHOUT a padding slot (417 @128k/44.1k) -- some frames are one byte longer,
                                                          see the _len entry points */
int tlb_out_stride(const tlb_batch *b);                /* longest frame over the batch, rounded up to a multiple of 4 */
long tlb_frames_encoded(const tlb_batch *b);           /* per stream */

/* Encode `nframes` frames of every stream, buffers resident in HBM.
 *   d_pcm      int16  [nframes][nstreams][2][1152]   planar like `short buffer[2][1152]`; mono reads [0] only
 *   d_xpad     uint8  [nframes][nstreams][TLB_MAX_XPAD] or NULL: the xpad_len bytes in transmission order
 *              (X-PAD bytes, then the two F-PAD bytes; toolame.c:515-551)
 *   d_xpad_len int32  [nframes][nstreams] or NULL; each 0 or 2..pad_len (any other value: the frame carries no PAD)
 *   d_out      uint8  [nframes][nstreams][tlb_out_stride()]
 * Frame n of a stream is final only once frame n+1 has been analysed (its ScF-CRC is stored in frame
 * n, toolame.c:527-542), so output slot f of this call holds the frame that was pending before input
 * frame f: slot 0 = the last frame of the previous call (undefined on the very first call), slot f =
 * input frame f-1.  tlb_flush_*() hands out the final pending frame.
 * The launch is asynchronous on `hip_stream` (a hipStream_t, NULL = default stream).  Launches of ONE batch must be ordered
 * on one stream (they share the streams' state and the batch's scratch buffers); at most 2^30 (stream, frame) pairs per call. */
int tlb_encode_device(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad,
                      const int32_t *d_xpad_len, uint8_t *d_out, void *hip_stream);
/* 44.1 and 22.05 kHz (libtoolame-dab encodes them; DAB itself does not use them, src/odr-audioenc.cpp:560-563): a frame is
 * tlb_frame_bytes() or one byte more (padding slot, availbits.c:49-62, header bit 9).  The _len variants also report the length
 * of the frame in every output slot (0: none): d_out_len int32 [nframes][nstreams] / out_len [nstreams] for the flush. */
int tlb_encode_device_len(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad, const int32_t *d_xpad_len,
                          uint8_t *d_out, int32_t *d_out_len, void *hip_stream);
int tlb_encode_host_len(tlb_batch *b, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len,
                        uint8_t *out, int32_t *out_len, void *taps);
int tlb_flush_host_len(tlb_batch *b, uint8_t *out, int32_t *out_len);
int tlb_flush_device_len(tlb_batch *b, uint8_t *d_out, int32_t *d_out_len, void *hip_stream);
/* Same with host buffers (synchronous; copies over PCIe).  `taps`, if not NULL, receives
 * [nframes][nstreams] TlTaps records (csrc/mp2_types.h) for stage-level parity tests. */
int tlb_encode_host(tlb_batch *b, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len,
                    uint8_t *out, void *taps);
/* Pinned (page-locked) host memory for the host-buffer entry points: copies from/to it run at PCIe link rate and
 * tlb_encode_host() queues copy-in, launch and copy-out without an intermediate synchronisation.  Pageable memory works
 * too, only slower.  The batch keeps its device staging buffers between calls (no allocation per call after the first). */
void *tlb_host_alloc(size_t bytes);
void tlb_host_free(void *p);
/* Copy every stream's pending (last) frame to out[nstreams][tlb_out_stride()] (host memory).
 * Streams that have not encoded any frame yet get zero bytes.  Does not change state. */
int tlb_flush_host(tlb_batch *b, uint8_t *out);
int tlb_flush_device(tlb_batch *b, uint8_t *d_out, void *hip_stream);

/* Caller-side per-frame glue folded into the batch: what AudioEnc::run() does between the input
 * queue and toolame_encode_frame() -- linear gain with the reference's double-multiply-and-truncate and the positive
 * peak level per channel (src/odr-audioenc.cpp:1030-1051), de-interleaving into `short[2][1152]` (:1139-1152).
 *   d_interleaved int16 [nframes][nstreams][2304]: s16le L R L R ... (mono streams: 1152 samples, rest ignored)
 *   d_pcm         int16 [nframes][nstreams][2][1152]  -- exactly what tlb_encode_device() takes
 *   d_peaks       int16 [nframes][nstreams][2]  (max(0, samples) of the "left"/"right" slots; the caller's silence
 *                 detection is `max(peaks) == 0`, odr-audioenc.cpp:1064)
 * tlb_set_gain_db(stream = -1) sets every stream; gain 0 dB leaves samples untouched like the reference. */
int tlb_set_gain_db(tlb_batch *b, int stream, double gain_db);
int tlb_ingest_device(tlb_batch *b, const int16_t *d_interleaved, int nframes, int16_t *d_pcm, int16_t *d_peaks, void *hip_stream);
int tlb_ingest_host(tlb_batch *b, const int16_t *interleaved, int nframes, int16_t *pcm, int16_t *peaks);
/* The caller's silence accounting (src/odr-audioenc.cpp:1053-1079): per stream, a frame with both peaks 0 adds its duration
 * in whole milliseconds (24 at 48 kHz, 36 at 32 kHz, 48 at 24 kHz) to d_silence_ms[stream], any other frame resets it to 0.
 * The decision itself (`measured_silence_ms > 1000 * silence_timeout` -> stop the stream) stays with the caller. */
int tlb_silence_device(tlb_batch *b, const int16_t *d_peaks, int nframes, uint32_t *d_silence_ms, void *hip_stream);
int tlb_silence_host(tlb_batch *b, const int16_t *peaks, int nframes, uint32_t *silence_ms);

/* Egress framing of the step after the path.
 *
 * UNITS.  ODR-AudioEnc does not send MP2 frames: it cuts the encoder's byte stream into pieces of 3 * bitrate bytes (24 ms)
 * and calls send_frame() once per piece (src/odr-audioenc.cpp:1211-1219, "ODR-DabMux expects frames of length 3*bitrate").
 * At 48 kHz a frame is one unit; an MPEG-2 LSF frame is two (24 kHz) or three (16 kHz).  Every egress buffer below is laid
 * out by SLOT v = frame * tlb_egress_max_units_per_frame() + unit: [nframes * max_upf][nstreams][...]; a stream with fewer
 * units per frame than the batch's maximum leaves its surplus slots absent (length / datasize 0).  For an all-48-kHz batch
 * max_upf is 1 and a slot is a frame.  32 kHz (1.5 units per frame) is not a DAB rate (odr-audioenc.cpp:560-563); a batch
 * that contains it has max_upf 0 and the egress calls return TLB_ERR_SAMPLERATE.
 * TIMING AND LEVELS differ from the reference's send loop, by the reference's burst behaviour: toolame_encode_frame() returns bytes only on
 * the calls on which its 4096-byte bit buffer fills, about one call in ten (bitstream.c:46-71), and odr-audioenc.cpp:1208-1225 then sends
 * every unit but one of what it holds (`while (toolame_buffer.size() > frame_len)`, strictly greater) in one loop with the SAME peak_left /
 * peak_right -- the values current on that call.  A live ZMQ / EDI capture of the reference therefore shows about ten units leaving
 * back to back every ten frames, sharing one level pair, and one unit always held back.  The batch calls emit the units of a frame in that
 * frame's slots with the levels the caller passes for that slot (tlb_tick_run: one frame per tick with that tick's peaks).  Payload bytes,
 * their order, DLFC / SEQ / timestamps per unit are identical; send cadence and the level pair per unit are not.  For a capture-compare
 * tlb_reference_send_schedule() gives the reference's schedule -- how many units it sends during each call -- so a caller can pass, for
 * every unit, the levels of the call the reference sends it on (pure host arithmetic on the configuration, no GPU). */
int tlb_reference_send_schedule(const tlb_stream_config *cfg, int ncalls, int32_t *units_sent);      /* -> bytes still held after the last call; < 0: -TLB_ERR_* */
int tlb_egress_unit_bytes(const tlb_batch *b, int stream);          /* 3 * kbps */
int tlb_egress_units_per_frame(const tlb_batch *b, int stream);     /* 1, 2, 3; 0 = not a whole number */
int tlb_egress_max_units_per_frame(const tlb_batch *b);
/* ZeroMQ part: one ODR-DabMux ZMQ message per unit,
 * `struct zmq_frame_header_t` (src/Outputs.h:76-89: u16 version = 1, u16 encoder = ZMQ_ENCODER_MPEG_L2 = 2, u32 datasize,
 * i16 audiolevel_left, i16 audiolevel_right, packed, little-endian) followed by the unit (Outputs.cpp:101-138).
 *   d_frames uint8 [nframes][nstreams][tlb_out_stride()]   (tlb_encode_device output)
 *   d_peaks  int16 [nframes][nstreams][2] or NULL (levels 0)  (tlb_ingest_device output; a frame's levels go with all its units)
 *   d_msgs   uint8 [nframes * max_upf][nstreams][tlb_zmq_msg_stride()]; message length of a stream = 12 + tlb_egress_unit_bytes()
 * Sockets and CURVE stay with the caller (out of scope). */
int tlb_zmq_msg_stride(const tlb_batch *b);
int tlb_zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream);
int tlb_zmq_frame_host(tlb_batch *b, const uint8_t *frames, const int16_t *peaks, int nframes, uint8_t *msgs);

/* Egress framing, EDI part: the AF packet an EDI/TCP destination receives for one frame --
 * what EDI::write_frame builds (src/Outputs.cpp:194-261): TAG items *ptr("DSTI"), dsti, ss0001 (the frame), ODRa (audio
 * levels) and, every ten seconds, ODRv (version string + uptime) (contrib/edioutput/TagItems.cpp), wrapped by
 * AFPacketiser::Assemble (contrib/edioutput/AFPacket.cpp:46-94: "AF", LEN, SEQ, AR, PT 'T', payload, CRC-16/CCITT).
 * `tlb_edi_state` is the per-stream sender state (EDI::m_timestamp/m_edi_time/m_send_version_at_time/m_num_seconds_sent/
 * m_tist, AFPacketiser::m_seq, TagDSTI::dlfc); the call advances it by nframes.  tlb_edi_state_init() is the first-call
 * branch of write_frame (Outputs.cpp:200-212) with the wall clock and the TAI-UTC offset passed in by the caller.
 *   d_frames  uint8 [nframes][nstreams][tlb_out_stride()]       (tlb_encode_device output)
 *   d_levels  int16 [nframes][nstreams][2] or NULL (levels 0)   (tlb_ingest_device peaks)
 *   d_state   tlb_edi_state [nstreams]
 *   d_pkts    uint8 [nframes * max_upf][nstreams][tlb_edi_af_stride()], d_pkt_len int32 [nframes * max_upf][nstreams] = bytes used
 *             (slot order, see UNITS above: one AF packet per 24-ms unit, each advancing timestamp, DLFC and SEQ; 0 = absent slot)
 * `version` is host memory (<= 64 bytes).  The PFT layer for UDP destinations is tlb_edi_pft_* below; sockets stay with the caller. */
typedef struct tlb_edi_state {
    int64_t edi_time, send_version_at_time;
    uint32_t timestamp, num_seconds_sent;
    int32_t tai_utc_offset;
    uint16_t seq, dlfc;
    uint8_t tist, pad_[7];
} tlb_edi_state;
void tlb_edi_state_init(tlb_edi_state *st, long long now_s, unsigned delay_ms, int tist, int tai_utc_offset);
int tlb_edi_af_stride(const tlb_batch *b, int version_len);
int tlb_edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
                      const char *version, int version_len, uint8_t *d_pkts, int32_t *d_pkt_len, void *hip_stream);
int tlb_edi_af_host(tlb_batch *b, const uint8_t *frames, const int16_t *levels, int nframes, tlb_edi_state *state,
                    const char *version, int version_len, uint8_t *pkts, int32_t *pkt_len);

/* EDI protection, fragmentation and transport layer (ETSI TS 102 821 clause 7): what
 * edi::PFT::Assemble (contrib/edioutput/PFT.cpp:234-320, with Protect :75-139 and ProtectAndFragment :141-232) makes of an
 * AF packet for UDP destinations -- Reed-Solomon RS(255,207) per chunk (fec = m > 0: m fragments may be lost), interleaved
 * fragments, PF header ("PF", Pseq, Findex, Fcount, FEC|Addr|Plen, [RSk, RSz], [Source, Dest], CRC-16) + payload.
 * fec = 0: fragmentation only (1400-byte slices), as the reference's default configuration (EDIConfig.h:69-70).
 *   d_af      uint8 [nframes][nstreams][af_stride], d_af_len int32 [nframes][nstreams]     (tlb_edi_af_device output)
 *   d_pseq    uint16 [nstreams]: PFT::m_pseq per stream, advanced by the number of packets present (d_af_len > 0); `nframes` here
 *             counts packet slots (nframes * max_upf of the AF call)
 *   d_frags   uint8 [nframes][nstreams][max_frags][frag_stride], d_frag_len int32 [nframes][nstreams][max_frags],
 *   d_nfrag   int32 [nframes][nstreams];  max_frags / frag_stride at least what tlb_edi_pft_shape() reports for this af_stride
 *   (every AF packet length up to af_stride is covered; at most 10 chunks per packet, i.e. chunk_len >= af_stride / 10)
 * chunk_len is edi::configuration_t::chunk_len (<= 207), transport/addr_source/dest_port the optional address header. */
int tlb_edi_pft_shape(const tlb_batch *b, int af_stride, int fec, int chunk_len, int transport, int *max_frags, int *frag_stride);
int tlb_edi_pft_device(tlb_batch *b, const uint8_t *d_af, const int32_t *d_af_len, int nframes, int af_stride, uint16_t *d_pseq,
                       int fec, int chunk_len, int transport, int addr_source, int dest_port,
                       uint8_t *d_frags, int32_t *d_frag_len, int32_t *d_nfrag, int max_frags, int frag_stride, void *hip_stream);
int tlb_edi_pft_host(tlb_batch *b, const uint8_t *af, const int32_t *af_len, int nframes, int af_stride, uint16_t *pseq,
                     int fec, int chunk_len, int transport, int addr_source, int dest_port,
                     uint8_t *frags, int32_t *frag_len, int32_t *nfrag, int max_frags, int frag_stride);

/* The real-time loop body as ONE call per tick (src/odr-audioenc.cpp:1030-1051 gain / peak, :1139-1152
 * de-interleave, :1158 toolame_encode_frame, :1208-1225 re-framing into units, src/Outputs.cpp:194-261 EDI::write_frame,
 * contrib/edioutput/PFT.cpp for UDP): one frame of EVERY stream goes host -> PCIe -> ingest -> encode -> EDI AF packets
 * (-> PFT fragments) -> PCIe -> host.  The tick object owns pinned host buffers and all device memory; streams are split into
 * `ngroups` contiguous groups whose copy-in, kernels and copy-out overlap on three HIP streams.
 *   1. fill tlb_tick_pcm(): int16 [nstreams][2304] interleaved s16le (and tlb_tick_xpad() [nstreams][TLB_MAX_XPAD] /
 *      tlb_tick_xpad_len() [nstreams] when created with_xpad)
 *   2. tlb_tick_run(): returns when everything is back in host memory
 *   3. read tlb_tick_peaks() int16 [nstreams][2] and, per stream, tlb_tick_frame() / tlb_tick_packet(unit) /
 *      tlb_tick_fragment(unit, k) (length 0 / count 0: nothing for this stream this tick)
 * What comes out of run number n is the frame that became final during it -- input frame n-1 (one frame of latency, see
 * tlb_encode_device; nothing on the very first run) -- in tlb_tick_units() units of 3*bitrate bytes (one per 48 kHz frame, two
 * per 24 kHz frame, three per 16 kHz frame: a run is one FRAME of every stream, so an LSF batch runs every 48 / 72 ms).
 * The audio levels sent with it are this run's peaks, as in the reference, whose send_frame() carries the peaks current at send
 * time (odr-audioenc.cpp:1213-1219).  tlb_tick_finish() pushes the last (pending) frame of every stream through the same
 * egress stage (toolame_finish at stream end); after it the object only answers the read accessors.
 * egress: TLB_TICK_FRAMES (raw MP2 frames, any sample rate), TLB_TICK_EDI_AF (AF packets, tlb_edi_af_device),
 * TLB_TICK_EDI_PFT (PFT fragments of those packets, tlb_edi_pft_device), TLB_TICK_ZMQ (ZeroMQ messages, tlb_zmq_frame_device).
 * Every run also advances the per-stream silence counter of src/odr-audioenc.cpp:1053-1079 (tlb_tick_silence_ms()).  The EDI sender state of every stream starts as
 * tlb_edi_state_init(now_s, delay_ms, tist, tai_utc_offset) and lives on the device. */
#define TLB_TICK_FRAMES 0
#define TLB_TICK_EDI_AF 1
#define TLB_TICK_EDI_PFT 2
#define TLB_TICK_ZMQ 3             /* ODR-DabMux ZeroMQ messages (tlb_zmq_frame_device): zmq_frame_header_t + unit */
typedef struct tlb_tick tlb_tick;
typedef struct {
    int egress;                    /* TLB_TICK_* */
    int ngroups;                   /* 0 = pick (1 below 2048 streams, 2 below 8192, 4 below 65536, else 8) */
    int with_xpad;                 /* X-PAD side input per tick */
    const char *version; int version_len;     /* ODRv string (EDI) */
    long long now_s; unsigned delay_ms; int tist; int tai_utc_offset;      /* tlb_edi_state_init() arguments */
    int fec, chunk_len, transport, addr_source, dest_port;                 /* PFT layer (chunk_len 0 = 207) */
} tlb_tick_config;
tlb_tick *tlb_tick_create(int device, int nstreams, const tlb_stream_config *cfgs, const tlb_tick_config *tc, int *err);
void tlb_tick_destroy(tlb_tick *t);
int tlb_tick_set_gain_db(tlb_tick *t, int stream, double gain_db);
/* tlb_stream_reset / _finish / _reconfigure for one stream of a tick object, between two runs.  The stream's EDI sender state (SEQ,
 * DLFC, timestamps) keeps running -- one continuous sender whose encoder restarted; until the stream's next frame is final its
 * packets / messages / frame have length 0. */
int tlb_tick_stream_reset(tlb_tick *t, int stream);
int tlb_tick_stream_finish(tlb_tick *t, int stream, uint8_t *out, size_t out_size);
int tlb_tick_stream_reconfigure(tlb_tick *t, int stream, const tlb_stream_config *cfg);
int16_t *tlb_tick_pcm(tlb_tick *t);
uint8_t *tlb_tick_xpad(tlb_tick *t);
int32_t *tlb_tick_xpad_len(tlb_tick *t);
int tlb_tick_run(tlb_tick *t);                 /* = tlb_tick_submit() + tlb_tick_wait() */
/* Ticks overlapped: the object owns TWO sets of pinned host input buffers (and three of outputs).  tlb_tick_submit() queues a tick on the input set the caller has
 * just filled and returns at once; from then on tlb_tick_pcm() / _xpad() / _xpad_len() point at the OTHER input set, which the caller
 * fills while the submitted tick is on its way (odr-audioenc decouples capture from encoding with its input queue,
 * src/odr-audioenc.cpp:904-986).  tlb_tick_wait() waits for the oldest submitted tick; the read accessors (tlb_tick_peaks, _frame,
 * _packet, _message, _fragment, _silence_ms) then show ITS results until the next wait -- the object keeps THREE sets of output buffers, so
 * neither of the two submits that may come before that wait touches them (a sender thread may ship them zero-copy).  Inputs: while two
 * ticks are in flight both input sets belong to queued copies and tlb_tick_pcm() / _xpad() / _xpad_len() return NULL; after the wait
 * the set of the retired tick is handed out again.  At most two ticks in flight: submit, submit, wait, submit, wait ...  Inside, tick t + 1's copy-in starts as soon as tick t's ingest kernel has consumed the device input buffer,
 * so the host-to-device link -- the limit at large stream counts -- never idles between ticks.  The RE-FETCH rule: call tlb_tick_pcm()
 * again after every submit / run, the pointer alternates. */
/* ERRORS.  A device failure inside submit / wait / run / finish (TLB_ERR_HIP, ...; TLB_ERR_ARG for calls out of order is NOT one) leaves the
 * object out of step with itself -- the stream groups queued before the failing one have advanced by a frame, the later ones have not --
 * so it is marked broken and STAYS so: every later submit / wait / run / finish / stream life-cycle call returns TLB_ERR_HIP, the input
 * accessors NULL; the read accessors keep showing the last tick that was waited for.  tlb_tick_status() = 0 or TLB_ERR_HIP.  The way
 * out is tlb_tick_destroy() and a new object (one level up: tlb_node_shard_restart does exactly that for one GPU of a node). */
int tlb_tick_submit(tlb_tick *t);
int tlb_tick_wait(tlb_tick *t);
int tlb_tick_status(const tlb_tick *t);
int tlb_tick_finish(tlb_tick *t);
long tlb_tick_count(const tlb_tick *t);
const int16_t *tlb_tick_peaks(const tlb_tick *t);
int tlb_tick_units(const tlb_tick *t, int stream);
const uint8_t *tlb_tick_frame(const tlb_tick *t, int stream, int *len);
const uint8_t *tlb_tick_packet(const tlb_tick *t, int stream, int unit, int *len);
const uint8_t *tlb_tick_message(const tlb_tick *t, int stream, int unit, int *len);      /* TLB_TICK_ZMQ */
const uint32_t *tlb_tick_silence_ms(const tlb_tick *t);      /* uint32 [nstreams]: the caller's silence counter (tlb_silence_device), after the run */
int tlb_tick_fragments(const tlb_tick *t, int stream, int unit);
const uint8_t *tlb_tick_fragment(const tlb_tick *t, int stream, int unit, int k, int *len);
float tlb_tick_last_ms(tlb_tick *t);           /* first copy-in queued -> last copy-out done of the last run, device clock */

/* ---------------------------------------------------------------------------------------------
 * (3) node level: every GPU of one host behind one handle (csrc/tlb_node.cpp)
 *
 * odr-audioenc is one process per service: AudioEnc::run() (src/odr-audioenc.cpp:819-1276) owns one input, one encoder, one set
 * of outputs.  A head-end that carries a fleet of services on one machine needs the step above: N streams cut into contiguous
 * blocks, block g = streams [g*N/G, (g+1)*N/G) on GPU g (section 8e), one host thread and one tlb_tick (or tlb_batch) per block,
 * nothing shared between blocks -- streams are independent, there is no collective on the data path -- and the counters added up.
 * A "shard" is such a block; `devices[g]` names the HIP device of shard g, and the same device may appear more than once (two
 * shards on one GPU behave as on two).  Every call below that acts on all shards runs on the shards' own threads in parallel and
 * returns when all of them have returned; the first non-zero code wins.  A stream keeps its node-wide index in every accessor.
 * Threading: ONE caller thread drives a node handle (the shards' threads are the node's own; the mailbox of a shard holds one job).
 * The function handed to tlb_node_parallel() runs on those threads and may use the accessors (tlb_node_pcm / _packet / ...), but
 * none of the calls that themselves go to the shards' threads (submit / wait / run / finish / sync / encode / life cycle / gain / copies):
 * they would wait for the thread they are running on.
 *
 * FAULT ISOLATION.  The reference restarts ONE failed input and nothing else (src/odr-audioenc.cpp:875-902, src/InputInterface.h:40);
 * one level up the unit of failure is a GPU.  A shard whose device call fails inside submit / wait / run / finish / encode / flush /
 * sync is marked BROKEN on the spot: its in-flight steps are dropped from the counters, every later node-wide call skips it, the
 * accessors of its streams answer NULL / 0 / length 0, the life-cycle calls of its streams TLB_ERR_HIP.  ALL OTHER SHARDS COMPLETE THE
 * CALL AND STAY IN LOCKSTEP -- their streams never notice.  The call in which a shard breaks returns that shard's code (the caller's
 * cue; later calls return 0 again, or TLB_ERR_HIP once no shard is left alive); tlb_node_shard_status() says which shard, with which TLB_ERR_* and
 * HIP's own error string; tlb_node_shard_restart() destroys the shard's object and makes a fresh one on the shard's own thread: its
 * streams start "as a freshly started reference process" (tlb_stream_reset's contract for the whole block), with the configurations
 * the streams have NOW and the caller's gains; it joins the lockstep at the next submit (first tick: nothing out, one frame of
 * latency).  Restart is legal between steps (no tick in flight / after tlb_node_sync), also on a healthy shard.
 *
 * Two planes, chosen at creation:
 *   TLB_NODE_TICK   the real-time loop: a tlb_tick per shard, host buffers in, packets out (everything of tlb_tick_* per stream).
 *   TLB_NODE_BATCH  device-resident buffers: a tlb_batch per shard, tlb_node_encode_device() takes one device pointer per shard.
 * ------------------------------------------------------------------------------------------ */
#define TLB_NODE_TICK 0
#define TLB_NODE_BATCH 1
typedef struct tlb_node tlb_node;
typedef struct {
    int plane;                     /* TLB_NODE_TICK / TLB_NODE_BATCH */
    tlb_tick_config tick;          /* TICK plane: egress, groups per shard, X-PAD, EDI / PFT parameters (as for tlb_tick_create) */
} tlb_node_config;
typedef struct {
    int shard, device;             /* total record: shard = -1, device = -1 */
    int first, nstreams;           /* the block [first, first + nstreams) */
    long steps;                    /* COMPLETED steps: ticks waited for (TICK) / encode calls retired by tlb_node_sync (BATCH); total record: the minimum over the shards */
    long frames;                   /* (stream, frame) pairs of the completed steps: sum of nstreams * frames per step (steps a shard lost by breaking are not counted) */
    double busy_ns;                /* host clock, this shard: submit (TICK) / the oldest queued encode call (BATCH) -> its results waited for (synced), summed; two ticks in flight overlap; total: the maximum */
    double device_ms;              /* device clock, summed over the completed steps.  TICK: tlb_tick_last_ms() of every tick.  BATCH: every queued encode call has its own
                                      pair of events on the shard's stream (first kernel queued -> last kernel done), added up at tlb_node_sync.  total: the maximum */
    double wall_ns;                /* total record only: first submit -> last wait as the node saw them, summed over steps */
} tlb_node_counter;
/* health and identity of one shard (tlb_node_shard_status) */
#define TLB_SHARD_OK 0
#define TLB_SHARD_BROKEN 1
#define TLB_NODE_WHAT_LEN 192
#define TLB_NODE_NAME_LEN 64
typedef struct {
    int shard, device, first, nstreams;
    int state;                     /* TLB_SHARD_OK / TLB_SHARD_BROKEN */
    int last_err;                  /* TLB_ERR_* of the failure that broke it last (kept after a restart), 0 = never failed */
    long failures, restarts;       /* times it broke / was restarted */
    long lost_steps;               /* steps that were in flight when it broke (not in the counters) */
    char what[TLB_NODE_WHAT_LEN];  /* the failing call, its code and HIP's error string of the shard's thread */
    char device_name[TLB_NODE_NAME_LEN];   /* hipDeviceProp_t::name */
    char pci[24];                  /* domain:bus:device.0 */
    char uuid[36];                 /* hipDeviceProp_t::uuid as 32 hex digits: two shards on DISTINCT GPUs differ here */
    int num_cu, num_xcd;           /* compute units, XCDs (hipDeviceAttributeNumberOfXccs) */
    double hbm_gb;
} tlb_node_shard_info;

/* Pure arithmetic, no GPU: the block of shard g, and what tlb_create() will make of it -- the number of distinct configurations,
 * the streams per kernel list (psy model 0, 1, 2 (with 4), 3: each list is one homogeneous launch) and the mono streams that share
 * waves in pairs.  Returns 0, or the TLB_ERR_* code of the first illegal configuration in the block. */
void tlb_node_partition(int nstreams, int nshards, int shard, int *first, int *n);
int tlb_node_plan_shard(int nstreams, const tlb_stream_config *cfgs, int nshards, int shard,
                        int *first, int *n, int *nconfigs, int list_sizes[4], int *mono_pairs);

tlb_node *tlb_node_create(int nshards, const int *devices, int nstreams, const tlb_stream_config *cfgs, const tlb_node_config *nc, int *err);
void tlb_node_destroy(tlb_node *nd);
int tlb_node_nshards(const tlb_node *nd);
int tlb_node_nstreams(const tlb_node *nd);
int tlb_node_shard_of(const tlb_node *nd, int stream);
/* per_shard: [nshards] records or NULL; total: frames summed, steps = the minimum over the shards, busy_ns / device_ms = the maximum, wall_ns; or NULL */
int tlb_node_counters(const tlb_node *nd, tlb_node_counter *per_shard, tlb_node_counter *total);
/* One line per shard: device ordinal and name, CUs, XCDs, memory, PCI address, UUID, stream block -- what a log of a multi-GPU run
 * should open with (also printed to stderr at creation when TLB_VERBOSE is set in the environment).  Valid until tlb_node_destroy. */
const char *tlb_node_describe(const tlb_node *nd);
/* Returns TLB_SHARD_OK / TLB_SHARD_BROKEN (negative: -TLB_ERR_ARG) and fills *info (may be NULL). */
int tlb_node_shard_status(const tlb_node *nd, int shard, tlb_node_shard_info *info);
/* Destroy + re-create one shard's object (see FAULT ISOLATION above).  now_s >= 0: the EDI timestamp origin of the restarted senders
 * (TICK plane; < 0 keeps the creation-time value).  Returns 0, TLB_ERR_ARG while a step is in flight or after finish, or the code
 * the re-creation failed with (the shard then stays broken and may be restarted again). */
int tlb_node_shard_restart(tlb_node *nd, int shard, long long now_s);
/* Run fn(ctx, shard, first, n) once per shard ON THAT SHARD'S THREAD, all at once, and wait: the caller's own per-block work --
 * filling the pinned PCM of a block from its inputs, shipping a block's packets -- parallel over the shards without a second pool. */
int tlb_node_parallel(tlb_node *nd, void (*fn)(void *ctx, int shard, int first, int n), void *ctx);

/* TICK plane.  tlb_node_pcm(stream) is that stream's int16[2304] slot (interleaved s16le) in its shard's current input set, NULL
 * while two ticks are in flight (see tlb_tick_pcm); the shards tick in lockstep, so submit / wait / run / finish follow the rules
 * of tlb_tick_submit / _wait / _run / _finish.  The read accessors are tlb_tick_*'s with node-wide stream indices. */
int16_t *tlb_node_pcm(tlb_node *nd, int stream);
uint8_t *tlb_node_xpad(tlb_node *nd, int stream);            /* uint8[TLB_MAX_XPAD] */
int32_t *tlb_node_xpad_len(tlb_node *nd, int stream);
int tlb_node_submit(tlb_node *nd);
int tlb_node_wait(tlb_node *nd);
int tlb_node_run(tlb_node *nd);
int tlb_node_finish(tlb_node *nd);
int tlb_node_units(const tlb_node *nd, int stream);
const int16_t *tlb_node_peaks(const tlb_node *nd, int stream);          /* int16[2] */
uint32_t tlb_node_silence_ms(const tlb_node *nd, int stream);
const uint8_t *tlb_node_frame(const tlb_node *nd, int stream, int *len);
const uint8_t *tlb_node_packet(const tlb_node *nd, int stream, int unit, int *len);
const uint8_t *tlb_node_message(const tlb_node *nd, int stream, int unit, int *len);
int tlb_node_fragments(const tlb_node *nd, int stream, int unit);
const uint8_t *tlb_node_fragment(const tlb_node *nd, int stream, int unit, int k, int *len);
/* both planes: gain and the life cycle of one stream (tlb_stream_* / tlb_tick_stream_* of its shard) */
int tlb_node_set_gain_db(tlb_node *nd, int stream, double gain_db);     /* stream = -1: all */
int tlb_node_stream_reset(tlb_node *nd, int stream);
int tlb_node_stream_finish(tlb_node *nd, int stream, uint8_t *out, size_t out_size);
int tlb_node_stream_reconfigure(tlb_node *nd, int stream, const tlb_stream_config *cfg);

/* BATCH plane.  tlb_node_batch(shard) is the shard's tlb_batch (sizes: tlb_out_stride, tlb_frame_bytes; do not encode on it
 * directly).  Device memory of a shard's GPU for callers without a HIP toolchain: tlb_node_device_alloc / _free / _copy_in / _copy_out
 * (synchronous).  tlb_node_encode_device(): element g of every pointer array is shard g's buffer, laid out as tlb_encode_device_len
 * wants it for THAT shard's streams ([nframes][n_g]...); d_xpad / d_xpad_len / d_out_len may be NULL (or hold NULL elements).  Each
 * shard's launch is queued on the shard's own HIP stream by the shard's thread; the call returns when all are queued,
 * tlb_node_sync() when all have finished.  tlb_node_flush_device(): the pending frames, tlb_flush_device_len per shard. */
tlb_batch *tlb_node_batch(tlb_node *nd, int shard);
void *tlb_node_device_alloc(tlb_node *nd, int shard, size_t bytes);
void tlb_node_device_free(tlb_node *nd, int shard, void *d_ptr);
int tlb_node_copy_in(tlb_node *nd, int shard, void *d_dst, const void *src, size_t bytes);
int tlb_node_copy_out(tlb_node *nd, int shard, void *dst, const void *d_src, size_t bytes);
int tlb_node_encode_device(tlb_node *nd, const int16_t *const *d_pcm, int nframes, const uint8_t *const *d_xpad,
                           const int32_t *const *d_xpad_len, uint8_t *const *d_out, int32_t *const *d_out_len);
int tlb_node_flush_device(tlb_node *nd, uint8_t *const *d_out, int32_t *const *d_out_len);
int tlb_node_sync(tlb_node *nd);

/* Diagnostic only: per-stage cycle stamps [nframes][nstreams][32] (csrc/mp2_wave.h TL_STAMP), host buffers. */
int tlb_encode_host_stamps(tlb_batch *b, const int16_t *pcm, int nframes, long long *stamps);

/* Duration in milliseconds of the most recent tlb_encode_device()/tlb_encode_host() kernel, measured
 * with hipEvents on the launch stream (synchronises on that stream).  < 0 on error. */
float tlb_last_kernel_ms(tlb_batch *b);
/* Models 2 and 4 run as two kernels (the psychoacoustic model, then the rest of the encoder).  Their durations for the most
 * recent launch of a batch whose streams all use these models; non-zero return for any other batch (models 1 and 3 run
 * model and encoder in one kernel, model 0 has no model kernel). */
int tlb_last_stage_ms(tlb_batch *b, float *psy_ms, float *encode_ms);
/* Deployment self-check (no GPU needed): the device computes the reference's transcendental calls with its own restatement of glibc
 * 2.35's FMA-path routines (csrc/tl_libm.h), so bit-equality with a reference built on THIS host holds if this host's libm is that
 * libm.  Compares the two on `nsamples` arguments per function (<= 0: 100 000); returns the number of differing results -- 0 means
 * the reference compiled here and this library agree on log10 / pow / log / exp / sincos / atan2 bit for bit. */
long tlb_selfcheck_libm(long nsamples);
/* LDS bytes per wavefront: the largest per-wave block among the kernels (each holds 12 waves per CU). */
int tlb_lds_bytes_per_stream(void);
const char *tlb_version(void);

#ifdef __cplusplus
}
#endif
#endif
