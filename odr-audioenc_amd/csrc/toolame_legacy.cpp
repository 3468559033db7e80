// toolame_legacy.cpp -- the reference's nine-function ABI (libtoolame-dab/toolame.h:13-48) on top of a private one-stream batch.
#include "tlb_internal.h"

// ------------------------------------------------------------------------------------------
// legacy nine-function ABI: stream 0 of a private one-stream batch (libtoolame-dab/toolame.h:13-48)

// The reference hands bytes back only when its 4096-byte bit buffer fills (bitstream.c:46-71): about nine calls in ten return 0.
// The shim knows that cadence arithmetically (frame lengths are a function of the configuration), so it DEFERS the GPU work:
// a call that returns nothing only files its PCM and X-PAD away (pinned host memory); the call on which a burst is due
// encodes every frame filed so far as ONE launch -- frames of a stream are independent (stream, frame) units, so ten frames
// cost one frame's latency -- with one copy in and one copy out.  What the caller sees (return values, bytes, their timing in
// calls) is unchanged: tests/test_hip_parity.py::test_legacy_abi_burst_cadence on every golden case.
struct Legacy {
    bool inited = false;
    long samplerate = 44100;       // toolame_init() sets header.version = MPEG-1 (toolame.c:141) and leaves sampling_frequency at its zero-initialised
                                   // index 0, which is 44.1 kHz in MPEG-1 (common.c:118-144): what a caller gets who never calls toolame_set_samplerate()
    char mode = 's';
    int kbps = 0;
    int psy = 1;                   // DFLT_PSY, encoder.h:11
    int pad_len = 0;
    tlb_batch *batch = nullptr;
    int lg_frame = 0, minimum = 4, fill = 0;     // emulated 4096-byte bit buffer (bitstream.c); lg_frame: a frame without padding slot
    double frac = 0, lag = 0;                    // the slot recurrence on the host (availbits.c:49-62): length of the frame being encoded
    long frame_num = 0;
    std::deque<unsigned char> fifo;              // final bytes not yet handed to the caller
    // deferred frames: pinned host staging for up to kDefer frames (a burst is due long before: 4096 bytes are 78 of the
    // shortest legal frames), filled call by call, encoded when a burst is due or the staging is full
    static const int kDefer = 96;
    int ndefer = 0, stride = 0;
    int16_t *h_pcm = nullptr; uint8_t *h_xpad = nullptr; int32_t *h_xl = nullptr; uint8_t *h_out = nullptr; int32_t *h_len = nullptr;
    void release()
    {
        tlb_host_free(h_pcm); tlb_host_free(h_xpad); tlb_host_free(h_xl); tlb_host_free(h_out); tlb_host_free(h_len);
        h_pcm = nullptr; h_xpad = nullptr; h_xl = nullptr; h_out = nullptr; h_len = nullptr;
    }
};
static Legacy g_legacy;

// encode the deferred frames: slot f of the launch carries the frame that became final while frame f was analysed
static void legacy_run_deferred()
{
    Legacy &g = g_legacy;
    if (!g.ndefer) return;
    if (int rc = tlb_encode_host_len(g.batch, g.h_pcm, g.ndefer, g.h_xpad, g.h_xl, g.h_out, g.h_len, nullptr)) {
        // the reference has no error return from this call (it exit()s on its own fatal errors, mem.c:28); losing frames
        // silently would be worse than stopping
        fprintf(stderr, "libtoolame-dab-hip: encoding on the GPU failed (error %d)\n", rc);
        exit(-1);
    }
    for (int f = 0; f < g.ndefer; f++) {
        const unsigned char *p = g.h_out + (size_t)f * (size_t)g.stride;
        g.fifo.insert(g.fifo.end(), p, p + g.h_len[f]);             // (slot 0 of the very first launch: length 0)
    }
    g.ndefer = 0;
}

static const int kLegacyBuf = 4096;       // common.h BUFFER_SIZE

static int legacy_emit(unsigned char *out, size_t out_size, size_t n)
{
    size_t j = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char c = g_legacy.fifo.front();
        g_legacy.fifo.pop_front();
        if (j < out_size) out[j++] = c;
        else if (j == out_size) { fprintf(stderr, "ERROR: libtoolame output buffer too small (%zu vs %zu)!\n", out_size, n); j = out_size + 1; }
    }
    return (int)(j > out_size ? out_size : j);
}

extern "C" {

int toolame_init(void)
{
    if (g_legacy.batch) { tlb_destroy(g_legacy.batch); g_legacy.batch = nullptr; }
    g_legacy.release();
    g_legacy = Legacy();
    g_legacy.inited = true;
    // Byte parity with the reference is parity with the reference AS BUILT AGAINST glibc 2.35's FMA-path libm (csrc/tl_libm.h).  A
    // maintainer who swaps this library in on a host with another libm would see the CPU reference's bytes move on degenerate
    // signals while these stay: say so once per process (host arithmetic only, a few milliseconds; TLB_NO_LIBM_CHECK silences it).
    static bool checked = false;
    if (!checked && !getenv("TLB_NO_LIBM_CHECK")) {
        checked = true;
        const long bad = tlb_selfcheck_libm(20000);
        if (bad) fprintf(stderr, "libtoolame-dab-hip: note: this host's libm differs from glibc 2.35's FMA-path routines in %ld of 140000 sampled "
                                 "results; the GPU encoder reproduces THAT libm's reference bytes, a reference built here may differ on degenerate signals "
                                 "(INTEGRATION.md section 5)\n", bad);
    }
    return 0;
}
int toolame_enable_byteswap(void) { return 0; }           // glopts.byteswap is never read on this path
int toolame_set_channel_mode(const char mode)
{
    if (mode != 's' && mode != 'd' && mode != 'j' && mode != 'm') { fprintf(stderr, "libtoolame-dab: Bad mode %c\n", mode); return 1; }
    g_legacy.mode = mode;
    return 0;
}
int toolame_set_psy_model(int new_model)
{
    if (new_model < 0 || new_model > 3) { fprintf(stderr, "libtoolame-dab: Invalid PSY model %d\n", new_model); return 1; }
    g_legacy.psy = new_model;
    return 0;
}
int toolame_set_bitrate(int brate)
{   // toolame.c:212-237: the rate is checked HERE, against the MPEG version the sample rate (already set, odr-audioenc.cpp:687-722)
    // selected; the reference's BitrateIndex() prints this message and exit(-1)s (common.c:95-116) -- the shim returns non-zero
    // instead, which sends odr-audioenc down its own "libtoolame-dab init failed" path (odr-audioenc.cpp:724-727)
    TlConfig c;
    const int rc = tl_build_config(&c, g_legacy.samplerate, g_legacy.mode, brate, g_legacy.psy, 0);
    if (rc == TLB_ERR_BITRATE) {
        fprintf(stderr, "BitrateIndex: %d is not a legal bitrate for version %i\n", brate, g_legacy.samplerate >= 32000 ? 1 : 0);
        return 1;
    }
    g_legacy.kbps = brate;
    return 0;
}
int toolame_set_samplerate(long sample_rate)
{
    switch (sample_rate) {
    case 44100: case 48000: case 32000: case 24000: case 22050: case 16000: g_legacy.samplerate = sample_rate; return 0;
    default: fprintf(stderr, "SmpFrqIndex: %ld is not a legal sample rate\n", sample_rate); return -1;
    }
}
int toolame_set_pad(int pad_len)
{
    if (pad_len < 0) { fprintf(stderr, "Invalid XPAD length specified\n"); return 1; }
    // The caller accepts padlen 0..255 (src/odr-audioenc.cpp:566) and every one of them is encoded (TLB_MAX_XPAD = 256).  The reference's
    // setter takes any non-negative number (toolame.c:250-262); a length the device record cannot hold is refused HERE, loudly --
    // never a frame that silently goes out without its PAD.
    if (pad_len > TLB_MAX_XPAD) { fprintf(stderr, "libtoolame-dab-hip: XPAD length %d exceeds the %d bytes this library carries per frame\n", pad_len, TLB_MAX_XPAD); return 1; }
    if (pad_len) g_legacy.pad_len = pad_len;
    return 0;
}

int toolame_encode_frame(short buffer[2][1152], unsigned char *xpad_data, size_t xpad_len, unsigned char *output_buffer,
                         size_t output_buffer_size)
{
    Legacy &g = g_legacy;
    if (!g.batch) {
        tlb_stream_config c = {g.samplerate, g.mode, g.kbps, g.psy, g.pad_len};      // (toolame_set_pad has refused what the record cannot hold)
        int err = 0;
        g.batch = tlb_create(0, 1, &c, &err);
        if (!g.batch) {
            // the reference exit()s on an illegal bitrate (common.c:114); a missing GPU is equally fatal here
            fprintf(stderr, "libtoolame-dab-hip: cannot create the GPU encoder (error %d)\n", err);
            exit(-1);
        }
        g.lg_frame = tlb_frame_bytes(g.batch, 0);
        g.frac = g.batch->h_configs[0].pad_frac; g.lag = 0;
        g.stride = tlb_out_stride(g.batch);
        g.h_pcm = (int16_t *)tlb_host_alloc((size_t)Legacy::kDefer * 2304 * sizeof(int16_t));
        g.h_xpad = (uint8_t *)tlb_host_alloc((size_t)Legacy::kDefer * TLB_MAX_XPAD);
        g.h_xl = (int32_t *)tlb_host_alloc((size_t)Legacy::kDefer * sizeof(int32_t));
        g.h_out = (uint8_t *)tlb_host_alloc((size_t)Legacy::kDefer * (size_t)g.stride);
        g.h_len = (int32_t *)tlb_host_alloc((size_t)Legacy::kDefer * sizeof(int32_t));
        if (!g.h_pcm || !g.h_xpad || !g.h_xl || !g.h_out || !g.h_len) { fprintf(stderr, "libtoolame-dab-hip: out of pinned host memory\n"); exit(-1); }
    }
    // length of THIS frame (the reference's bit buffer fills with it now; its bytes come out of the GPU later)
    int cur_len = g.lg_frame;
    if (g.frac != 0) { if (g.lag > (g.frac - 1.0)) g.lag -= g.frac; else { cur_len++; g.lag += (1 - g.frac); } }
    if (g.frame_num == 0) g.minimum = cur_len + 4;           // toolame.c:298-300: frame 1's length
    // file the frame away
    memcpy(g.h_pcm + (size_t)g.ndefer * 2304, &buffer[0][0], 2304 * sizeof(int16_t));
    unsigned char *xrec = g.h_xpad + (size_t)g.ndefer * TLB_MAX_XPAD;
    int32_t xl = 0;
    memset(xrec, 0, TLB_MAX_XPAD);
    if (xpad_len >= 2 && xpad_data && (int)xpad_len <= g.pad_len) {
        xl = (int32_t)xpad_len;                            // bytes [dab_length-xpad_len, dab_length) in transmission order
        memcpy(xrec, xpad_data + g.pad_len - (int)xpad_len, xpad_len);
    } else if (xpad_len) {
        // outside the contract of toolame.c:515-524 (the reference asserts on 1 and reads before xpad_data[] when xpad_len exceeds
        // toolame_set_pad()'s length): the frame goes out without PAD, and says so
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "libtoolame-dab-hip: xpad_len %zu outside 2..%d (toolame_set_pad), frame sent without PAD\n", xpad_len, g.pad_len); }
    }
    g.h_xl[g.ndefer] = xl;
    g.ndefer++;
    g.frame_num++;
    // bitstream.c:46-71: when the 4096-byte buffer fills, everything but the newest `minimum` bytes is handed out -- bytes of
    // frames up to the one before this, which are final once this frame's ScF-CRC is known: the deferred frames run now
    int written = 0;
    if (g.fill + cur_len >= kLegacyBuf) {
        legacy_run_deferred();
        written = legacy_emit(output_buffer, output_buffer_size, (size_t)(kLegacyBuf - g.minimum));
        g.fill = g.minimum + (g.fill + cur_len - kLegacyBuf);
    } else {
        g.fill += cur_len;
        if (g.ndefer == Legacy::kDefer) legacy_run_deferred();
    }
    return written;
}

int toolame_finish(unsigned char *output_buffer, size_t output_buffer_size)
{
    Legacy &g = g_legacy;
    if (!g.batch) return 0;
    legacy_run_deferred();                                   // frames filed since the last burst
    std::vector<unsigned char> last((size_t)tlb_out_stride(g.batch));
    if (g.frame_num > 0) {
        int32_t last_len = 0;
        if (int rc = tlb_flush_host_len(g.batch, last.data(), &last_len)) { fprintf(stderr, "libtoolame-dab-hip: flushing the GPU encoder failed (error %d)\n", rc); exit(-1); }
        g.fifo.insert(g.fifo.end(), last.begin(), last.begin() + last_len);   // the last frame keeps its own ScF-CRC
    }
    int n = legacy_emit(output_buffer, output_buffer_size, g.fifo.size());
    tlb_destroy(g.batch);
    g.batch = nullptr;
    g.release();
    g.fill = 0; g.frame_num = 0;
    return n;
}

}  // extern "C"
