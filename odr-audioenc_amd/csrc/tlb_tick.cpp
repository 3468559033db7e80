// tlb_tick.cpp -- the real-time loop body as one object (include/toolame_batch.h, tlb_tick_*).  Host C++ above the batch and egress entry points.
#include "tlb_internal.h"

// ------------------------------------------------------------------------------------------
// The caller's real-time loop body as ONE call per tick (include/toolame_batch.h, tlb_tick_*): what AudioEnc::run() does for
// one stream every 24 ms -- gain / peak / de-interleave (src/odr-audioenc.cpp:1030-1051,1139-1152), toolame_encode_frame
// (:1158), re-framing into 3*bitrate-byte units (:1208-1225), EDI::write_frame (src/Outputs.cpp:194-261, optionally the PFT
// layer) -- for every stream of a GPU at once: interleaved PCM in pinned host memory -> PCIe -> tl_ingest_kernel ->
// the encode kernels (one frame per stream) -> tl_edi_af_kernel (-> tl_edi_pft_kernel) -> PCIe -> pinned host memory.
// The streams are split into groups (contiguous ranges, a private tlb_batch each): while group g's kernels run, group g+1's
// PCM comes in and group g-1's packets go out, on three HIP streams (the link is full duplex).
// ------------------------------------------------------------------------------------------
struct TickGroup {
    tlb_batch *b = nullptr;
    int first = 0, n = 0, out_stride = 0, max_upf = 1, af_stride = 0, max_frags = 0, frag_stride = 0;
    int16_t *d_inter = nullptr, *d_pcm = nullptr, *d_peaks = nullptr;
    uint8_t *d_xpad = nullptr; int32_t *d_xl = nullptr;
    uint8_t *d_frames = nullptr; int32_t *d_flen = nullptr;
    tlb_edi_state *d_state = nullptr; uint8_t *d_pkts = nullptr; int32_t *d_plen = nullptr;
    uint16_t *d_pseq = nullptr; uint8_t *d_frags = nullptr; int32_t *d_fraglen = nullptr, *d_nfrag = nullptr;
    uint8_t *d_msgs = nullptr; int msg_stride = 0;              // ZeroMQ egress
    uint32_t *d_silence = nullptr;                              // milliseconds of digital silence so far, per stream
    // this group's slices of the pinned host outputs, THREE sets: tick n's results land in set n % 3.  With two ticks in flight the
    // caller is still reading tick n (valid until the next wait) while tick n + 1 is on its way and tick n + 2 is being submitted:
    // three sets make "until the next wait" true without a copy (two sets let submit n + 2 overwrite what tick n showed)
    uint8_t *h_frames[3] = {}; int32_t *h_flen[3] = {}; uint8_t *h_pkts[3] = {}; int32_t *h_plen[3] = {};
    uint8_t *h_frags[3] = {}; int32_t *h_fraglen[3] = {}, *h_nfrag[3] = {};
    uint8_t *h_msgs[3] = {};
    hipEvent_t ev_in = nullptr, ev_run = nullptr;
    hipEvent_t ev_ingested = nullptr, ev_encoded = nullptr, ev_out = nullptr;   // the group's device buffers are single: the next tick's copy-in waits for this tick's
                                                                                // ingest (d_inter) / encode (X-PAD), its kernels for this tick's copy-out
};
struct tlb_tick {
    int device = 0, nstreams = 0, egress = 0, version_len = 0, with_xpad = 0;
    char version[TL_EDI_MAX_VERSION] = {};
    int fec = 0, chunk_len = 207, transport = 0, addr_source = 0, dest_port = 0;
    std::vector<TickGroup> groups;
    std::vector<int> group_of;                   // stream -> group
    // pinned host buffers (tlb_tick_submit / tlb_tick_wait): two INPUT sets -- the caller fills input set `in_set` while the tick
    // submitted before is still on its way (with two ticks in flight neither set is free: the input accessors return NULL) -- and
    // three OUTPUT sets; results are read from `out_set`, the set of the tick waited for last
    int16_t *h_inter[2] = {}, *h_peaks[3] = {}; uint8_t *h_xpad[2] = {}; int32_t *h_xl[2] = {};
    uint32_t *h_silence[3] = {};
    int in_set = 0, out_set = 0;
    long waited = 0;                             // ticks whose results have been waited for (ticks: submitted)
    std::vector<void *> pinned, dev;
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    hipEvent_t ev0[3] = {}, ev1[3] = {};          // per output set: first copy-in queued / last copy-out done
    long ticks = 0;
    bool finished = false;
    bool broken = false;                         // a device call of submit / wait / finish failed: the groups queued before the failing one have advanced by a frame,
                                                 // the later ones have not -- the object is out of step with itself and every further call but destroy is refused (sticky)
};
// a failing submit / wait / finish: drain what was queued, mark the object, hand the code on
static int tick_fail(tlb_tick *t, int rc);

extern "C" {

void tlb_tick_destroy(tlb_tick *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    for (auto &g : t->groups) {
        if (g.b) tlb_destroy(g.b);
        if (g.ev_in) (void)hipEventDestroy(g.ev_in);
        if (g.ev_run) (void)hipEventDestroy(g.ev_run);
        if (g.ev_ingested) (void)hipEventDestroy(g.ev_ingested);
        if (g.ev_encoded) (void)hipEventDestroy(g.ev_encoded);
        if (g.ev_out) (void)hipEventDestroy(g.ev_out);
    }
    for (void *p : t->dev) (void)hipFree(p);
    for (void *p : t->pinned) (void)hipHostFree(p);
    if (t->s_in) (void)hipStreamDestroy(t->s_in);
    if (t->s_run) (void)hipStreamDestroy(t->s_run);
    if (t->s_out) (void)hipStreamDestroy(t->s_out);
    for (int k = 0; k < 3; k++) { if (t->ev0[k]) (void)hipEventDestroy(t->ev0[k]); if (t->ev1[k]) (void)hipEventDestroy(t->ev1[k]); }
    delete t;
}

static int tick_create_impl(tlb_tick *t, int device, int nstreams, const tlb_stream_config *cfgs, const tlb_tick_config *tc)
{
    t->device = device; t->nstreams = nstreams; t->egress = tc->egress; t->with_xpad = tc->with_xpad ? 1 : 0;
    if (tc->egress < TLB_TICK_FRAMES || tc->egress > TLB_TICK_ZMQ || tc->version_len < 0 || tc->version_len > TL_EDI_MAX_VERSION ||
        (tc->version_len && !tc->version)) return TLB_ERR_ARG;
    t->version_len = tc->version_len;
    if (tc->version_len) memcpy(t->version, tc->version, (size_t)tc->version_len);
    t->fec = tc->fec; t->chunk_len = tc->chunk_len ? tc->chunk_len : 207; t->transport = tc->transport; t->addr_source = tc->addr_source; t->dest_port = tc->dest_port;
    int ng = tc->ngroups > 0 ? tc->ngroups : (nstreams >= 65536 ? 8 : nstreams >= 8192 ? 4 : nstreams >= 2048 ? 2 : 1);      // more groups = a shorter tail behind the last copy-in
    if (ng > nstreams) ng = nstreams;
    t->groups.resize((size_t)ng);
    t->group_of.resize((size_t)nstreams);
    size_t n_frames = 0, n_pkts = 0, n_slots = 0, n_frags = 0, n_fragslots = 0, n_msgs = 0;
    for (int g = 0; g < ng; g++) {
        TickGroup &G = t->groups[(size_t)g];
        G.first = (int)((long)nstreams * g / ng); G.n = (int)((long)nstreams * (g + 1) / ng) - G.first;
        for (int s = G.first; s < G.first + G.n; s++) t->group_of[(size_t)s] = g;
        int err = 0;
        G.b = tlb_create(device, G.n, cfgs + G.first, &err);
        if (!G.b) return err ? err : TLB_ERR_HIP;
        G.out_stride = G.b->out_stride; G.max_upf = G.b->max_upf;
        if (tc->egress == TLB_TICK_ZMQ) {
            if (!G.max_upf) return TLB_ERR_SAMPLERATE;
            G.msg_stride = tlb_zmq_msg_stride(G.b);
        } else if (tc->egress != TLB_TICK_FRAMES) {
            if (!G.max_upf) return TLB_ERR_SAMPLERATE;
            G.af_stride = tlb_edi_af_stride(G.b, tc->version_len);
            if (tc->egress == TLB_TICK_EDI_PFT)
                if (int rc = pft_shape(G.af_stride, t->fec, t->chunk_len, t->transport, &G.max_frags, &G.frag_stride)) return rc;
        } else if (!G.max_upf) G.max_upf = 1;
        n_frames += (size_t)G.n * (size_t)G.out_stride;
        n_slots += (size_t)G.n * (size_t)G.max_upf;
        n_pkts += (size_t)G.n * (size_t)G.max_upf * (size_t)G.af_stride;
        n_msgs += (size_t)G.n * (size_t)G.max_upf * (size_t)G.msg_stride;
        n_fragslots += (size_t)G.n * (size_t)G.max_upf * (size_t)G.max_frags;
        n_frags += (size_t)G.n * (size_t)G.max_upf * (size_t)G.max_frags * (size_t)G.frag_stride;
    }
    HIPCHK(hipSetDevice(device));
    auto pin = [&](size_t bytes) -> void * { void *p = nullptr; if (hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault) != hipSuccess) return nullptr; memset(p, 0, bytes ? bytes : 4); t->pinned.push_back(p); return p; };
    auto dev = [&](size_t bytes) -> void * { void *p = nullptr; if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr; (void)hipMemset(p, 0, bytes ? bytes : 4); t->dev.push_back(p); return p; };
    uint8_t *h_msgs[3], *h_frames[3], *h_pkts[3], *h_frags[3]; int32_t *h_flen[3], *h_plen[3], *h_fraglen[3], *h_nfrag[3];
    for (int k = 0; k < 2; k++) {
        t->h_inter[k] = (int16_t *)pin((size_t)nstreams * 2304 * sizeof(int16_t));
        t->h_xpad[k] = (uint8_t *)pin(t->with_xpad ? (size_t)nstreams * TL_MAX_XPAD : 0);
        t->h_xl[k] = (int32_t *)pin(t->with_xpad ? (size_t)nstreams * sizeof(int32_t) : 0);
        if (!t->h_inter[k] || !t->h_xpad[k] || !t->h_xl[k]) return TLB_ERR_HIP;
    }
    for (int k = 0; k < 3; k++) {
        t->h_peaks[k] = (int16_t *)pin((size_t)nstreams * 2 * sizeof(int16_t));
        t->h_silence[k] = (uint32_t *)pin((size_t)nstreams * sizeof(uint32_t));
        h_msgs[k] = (uint8_t *)pin(n_msgs);
        h_frames[k] = (uint8_t *)pin(n_frames); h_flen[k] = (int32_t *)pin((size_t)nstreams * sizeof(int32_t));
        h_pkts[k] = (uint8_t *)pin(n_pkts); h_plen[k] = (int32_t *)pin(n_slots * sizeof(int32_t));
        h_frags[k] = (uint8_t *)pin(n_frags); h_fraglen[k] = (int32_t *)pin(n_fragslots * sizeof(int32_t)); h_nfrag[k] = (int32_t *)pin(n_slots * sizeof(int32_t));
        if (!t->h_peaks[k] || !t->h_silence[k] || !h_msgs[k] || !h_frames[k] || !h_flen[k] || !h_pkts[k] || !h_plen[k] ||
            !h_frags[k] || !h_fraglen[k] || !h_nfrag[k]) return TLB_ERR_HIP;
    }
    std::vector<tlb_edi_state> st0;
    size_t o_frames = 0, o_slots = 0, o_pkts = 0, o_frags = 0, o_fragslots = 0, o_msgs = 0;
    for (auto &G : t->groups) {
        const size_t n = (size_t)G.n, slots = n * (size_t)G.max_upf;
        G.d_inter = (int16_t *)dev(n * 2304 * 2); G.d_pcm = (int16_t *)dev(n * 2304 * 2); G.d_peaks = (int16_t *)dev(n * 4);
        G.d_xpad = (uint8_t *)dev(t->with_xpad ? n * TL_MAX_XPAD : 0); G.d_xl = (int32_t *)dev(t->with_xpad ? n * 4 : 0);
        G.d_frames = (uint8_t *)dev(n * (size_t)G.out_stride); G.d_flen = (int32_t *)dev(n * 4);
        G.d_state = (tlb_edi_state *)dev(n * sizeof(tlb_edi_state));
        G.d_pkts = (uint8_t *)dev(slots * (size_t)G.af_stride); G.d_plen = (int32_t *)dev(slots * 4);
        G.d_pseq = (uint16_t *)dev(n * 2);
        G.d_msgs = (uint8_t *)dev(slots * (size_t)G.msg_stride); G.d_silence = (uint32_t *)dev(n * 4);
        if (!G.d_msgs || !G.d_silence) return TLB_ERR_HIP;
        for (int k = 0; k < 3; k++) G.h_msgs[k] = h_msgs[k] + o_msgs;
        o_msgs += slots * (size_t)G.msg_stride;
        G.d_frags = (uint8_t *)dev(slots * (size_t)G.max_frags * (size_t)G.frag_stride); G.d_fraglen = (int32_t *)dev(slots * (size_t)G.max_frags * 4); G.d_nfrag = (int32_t *)dev(slots * 4);
        if (!G.d_inter || !G.d_pcm || !G.d_peaks || !G.d_xpad || !G.d_xl || !G.d_frames || !G.d_flen || !G.d_state || !G.d_pkts || !G.d_plen || !G.d_pseq ||
            !G.d_frags || !G.d_fraglen || !G.d_nfrag) return TLB_ERR_HIP;
        for (int k = 0; k < 3; k++) {
            G.h_frames[k] = h_frames[k] + o_frames; G.h_flen[k] = h_flen[k] + G.first; G.h_pkts[k] = h_pkts[k] + o_pkts; G.h_plen[k] = h_plen[k] + o_slots;
            G.h_frags[k] = h_frags[k] + o_frags; G.h_fraglen[k] = h_fraglen[k] + o_fragslots; G.h_nfrag[k] = h_nfrag[k] + o_slots;
        }
        o_frames += n * (size_t)G.out_stride; o_slots += slots; o_pkts += slots * (size_t)G.af_stride;
        o_fragslots += slots * (size_t)G.max_frags; o_frags += slots * (size_t)G.max_frags * (size_t)G.frag_stride;
        st0.resize(n);
        for (size_t i = 0; i < n; i++) tlb_edi_state_init(&st0[i], tc->now_s, tc->delay_ms, tc->tist, tc->tai_utc_offset);
        HIPCHK(hipMemcpy(G.d_state, st0.data(), n * sizeof(tlb_edi_state), hipMemcpyHostToDevice));
        HIPCHK(hipEventCreateWithFlags(&G.ev_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&G.ev_run, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&G.ev_ingested, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&G.ev_encoded, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&G.ev_out, hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&t->s_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&t->s_run, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&t->s_out, hipStreamNonBlocking));
    for (int k = 0; k < 3; k++) { HIPCHK(hipEventCreate(&t->ev0[k])); HIPCHK(hipEventCreate(&t->ev1[k])); }
    return TLB_OK;
}

tlb_tick *tlb_tick_create(int device, int nstreams, const tlb_stream_config *cfgs, const tlb_tick_config *tc, int *err)
{
    if (nstreams <= 0 || !cfgs || !tc) { if (err) *err = TLB_ERR_ARG; return nullptr; }
    tlb_tick *t = new tlb_tick;
    const int rc = tick_create_impl(t, device, nstreams, cfgs, tc);
    if (err) *err = rc;
    if (rc) { tlb_tick_destroy(t); return nullptr; }
    return t;
}

// The input accessors hand out the set the NEXT submit will read.  With two ticks in flight both sets belong to queued copy-ins (the
// set these would name is the one the older tick's host-to-device copy may still be reading): NULL until tlb_tick_wait() has
// retired that tick -- no submit is possible in that state anyway.
static bool tick_input_free(const tlb_tick *t) { return t && !t->finished && !t->broken && t->ticks - t->waited < 2; }
int16_t *tlb_tick_pcm(tlb_tick *t) { return tick_input_free(t) ? t->h_inter[t->in_set] : nullptr; }
uint8_t *tlb_tick_xpad(tlb_tick *t) { return tick_input_free(t) && t->with_xpad ? t->h_xpad[t->in_set] : nullptr; }
int32_t *tlb_tick_xpad_len(tlb_tick *t) { return tick_input_free(t) && t->with_xpad ? t->h_xl[t->in_set] : nullptr; }
const int16_t *tlb_tick_peaks(const tlb_tick *t) { return t ? t->h_peaks[t->out_set] : nullptr; }
long tlb_tick_count(const tlb_tick *t) { return t ? t->ticks : 0; }
int tlb_tick_set_gain_db(tlb_tick *t, int stream, double gain_db)
{
    if (!t || stream < -1 || stream >= t->nstreams) return TLB_ERR_ARG;
    for (auto &G : t->groups) {
        if (stream >= 0 && (stream < G.first || stream >= G.first + G.n)) continue;
        if (int rc = tlb_set_gain_db(G.b, stream < 0 ? -1 : stream - G.first, gain_db)) return rc;
    }
    return TLB_OK;
}

// Life cycle of one stream of a tick object (tlb_stream_reset / _finish / _reconfigure of its group's batch).  The EDI sender state
// of the stream (SEQ, DLFC, timestamps) is NOT touched: the receiver sees one continuous sender whose encoder was restarted, as
// with the reference, whose output object outlives an encoder re-initialisation.  Until the stream's next frame is final its
// slots are empty (length 0).
static TickGroup *tick_group_of(tlb_tick *t, int stream, int *local)
{
    if (!t || stream < 0 || stream >= t->nstreams) return nullptr;
    TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    *local = stream - G.first;
    return &G;
}
int tlb_tick_stream_reset(tlb_tick *t, int stream)
{
    int k; TickGroup *G = tick_group_of(t, stream, &k);
    if (!G || t->finished) return TLB_ERR_ARG;
    if (t->broken) return TLB_ERR_HIP;
    return tlb_stream_reset(G->b, k);
}
int tlb_tick_stream_finish(tlb_tick *t, int stream, uint8_t *out, size_t out_size)
{
    int k; TickGroup *G = tick_group_of(t, stream, &k);
    if (!G || t->finished) return -TLB_ERR_ARG;
    if (t->broken) return -TLB_ERR_HIP;
    return tlb_stream_finish(G->b, k, out, out_size);
}
int tlb_tick_stream_reconfigure(tlb_tick *t, int stream, const tlb_stream_config *cfg)
{
    int k; TickGroup *G = tick_group_of(t, stream, &k);
    if (!G || t->finished) return TLB_ERR_ARG;
    if (t->broken) return TLB_ERR_HIP;
    return tlb_stream_reconfigure(G->b, k, cfg);
}

// egress of the frames sitting in G.d_frames + copy-out, queued on s_run / s_out
static int tick_egress(tlb_tick *t, TickGroup &G, bool have_frames, int set, bool new_input = true)
{
    const size_t n = (size_t)G.n, slots = n * (size_t)G.max_upf;
    if (new_input) if (int rc = tlb_silence_device(G.b, G.d_peaks, 1, G.d_silence, t->s_run)) return rc;       // odr-audioenc.cpp:1053-1079 (the decision stays with the caller)
    if (have_frames && t->egress == TLB_TICK_ZMQ) {
        if (int rc = zmq_frame_device(G.b, G.d_frames, G.d_peaks, 1, G.d_msgs, t->s_run, G.d_flen)) return rc;
    } else if (have_frames && t->egress != TLB_TICK_FRAMES) {
        if (int rc = edi_af_device(G.b, G.d_frames, G.d_peaks, 1, G.d_state, t->version, t->version_len, G.d_pkts, G.d_plen, t->s_run, G.d_flen)) return rc;
        if (t->egress == TLB_TICK_EDI_PFT)
            if (int rc = tlb_edi_pft_device(G.b, G.d_pkts, G.d_plen, G.max_upf, G.af_stride, G.d_pseq, t->fec, t->chunk_len, t->transport, t->addr_source, t->dest_port,
                                            G.d_frags, G.d_fraglen, G.d_nfrag, G.max_frags, G.frag_stride, t->s_run)) return rc;
    }
    HIPCHK(hipEventRecord(G.ev_run, t->s_run));
    HIPCHK(hipStreamWaitEvent(t->s_out, G.ev_run, 0));
    HIPCHK(hipMemcpyAsync(t->h_peaks[set] + (size_t)G.first * 2, G.d_peaks, n * 4, hipMemcpyDeviceToHost, t->s_out));
    HIPCHK(hipMemcpyAsync(t->h_silence[set] + G.first, G.d_silence, n * 4, hipMemcpyDeviceToHost, t->s_out));
    if (have_frames) {                                               // (the very first tick: no frame is final yet, lengths stay 0)
        if (t->egress == TLB_TICK_FRAMES) {
            HIPCHK(hipMemcpyAsync(G.h_frames[set], G.d_frames, n * (size_t)G.out_stride, hipMemcpyDeviceToHost, t->s_out));
            HIPCHK(hipMemcpyAsync(G.h_flen[set], G.d_flen, n * 4, hipMemcpyDeviceToHost, t->s_out));
        } else if (t->egress == TLB_TICK_ZMQ) {
            HIPCHK(hipMemcpyAsync(G.h_msgs[set], G.d_msgs, slots * (size_t)G.msg_stride, hipMemcpyDeviceToHost, t->s_out));
        } else if (t->egress == TLB_TICK_EDI_AF) {
            HIPCHK(hipMemcpyAsync(G.h_pkts[set], G.d_pkts, slots * (size_t)G.af_stride, hipMemcpyDeviceToHost, t->s_out));
            HIPCHK(hipMemcpyAsync(G.h_plen[set], G.d_plen, slots * 4, hipMemcpyDeviceToHost, t->s_out));
        } else {
            HIPCHK(hipMemcpyAsync(G.h_frags[set], G.d_frags, slots * (size_t)G.max_frags * (size_t)G.frag_stride, hipMemcpyDeviceToHost, t->s_out));
            HIPCHK(hipMemcpyAsync(G.h_fraglen[set], G.d_fraglen, slots * (size_t)G.max_frags * 4, hipMemcpyDeviceToHost, t->s_out));
            HIPCHK(hipMemcpyAsync(G.h_nfrag[set], G.d_nfrag, slots * 4, hipMemcpyDeviceToHost, t->s_out));
        }
    }
    HIPCHK(hipEventRecord(G.ev_out, t->s_out));                      // the group's device output buffers are free again once this has passed
    return TLB_OK;
}

static void tick_drain(tlb_tick *t) { (void)hipStreamSynchronize(t->s_in); (void)hipStreamSynchronize(t->s_run); (void)hipStreamSynchronize(t->s_out); }
static int tick_fail(tlb_tick *t, int rc)
{
    tick_drain(t);
    if (!t->broken) fprintf(stderr, "libtoolame-dab-hip: a tick failed half way (code %d): its stream groups are out of step, the tick object refuses further work; destroy it\n", rc);
    t->broken = true;
    return rc;
}
int tlb_tick_status(const tlb_tick *t) { return !t ? TLB_ERR_ARG : t->broken ? TLB_ERR_HIP : TLB_OK; }
#ifdef TLB_FAULT_INJECT
// test builds only (csrc/tlb_debug.h): the nth submit from now fails in its LAST group -- after the groups before it have been queued
int tlb_debug_tick_fail_next(tlb_tick *t, int nth) { if (!t || nth < 0) return TLB_ERR_ARG; t->groups.back().b->fail_in = nth; return TLB_OK; }
#endif

// Queue one tick -- copy-in, ingest, encode, egress, copy-out of every group -- on the input set the caller has just filled, and
// return at once.  tlb_tick_pcm() then points at the OTHER input set: the caller fills the next tick while this one is on its way
// (odr-audioenc decouples capture from encoding with its input queue, src/odr-audioenc.cpp:904-986).  The device buffers of a group
// are single, so across ticks: the next copy-in waits for this tick's ingest kernel, the next kernels for this tick's copy-out --
// the host-to-device link, the limit at large stream counts, never idles between ticks.  At most two ticks may be in flight
// (two host sets): submit, submit, wait, submit, wait, ...
int tlb_tick_submit(tlb_tick *t)
{
    if (!t || t->finished || t->ticks - t->waited >= 2) return TLB_ERR_ARG;
    if (t->broken) return TLB_ERR_HIP;
    if (hipSetDevice(t->device) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    const int set = (int)(t->ticks & 1);                             // == in_set: ticks and input sets alternate together
    const int oset = (int)(t->ticks % 3);                            // output set: the caller may still be reading tick - 2's
    if (hipEventRecord(t->ev0[oset], t->s_in) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    for (auto &G : t->groups) {
        const size_t n = (size_t)G.n;
        int rc = TLB_OK;
        hipError_t e = hipSuccess;
        if (t->ticks > 0) e = hipStreamWaitEvent(t->s_in, G.ev_ingested, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(G.d_inter, t->h_inter[set] + (size_t)G.first * 2304, n * 2304 * sizeof(int16_t), hipMemcpyHostToDevice, t->s_in);
        if (e == hipSuccess && t->with_xpad && t->ticks > 0) e = hipStreamWaitEvent(t->s_in, G.ev_encoded, 0);
        if (e == hipSuccess && t->with_xpad) e = hipMemcpyAsync(G.d_xpad, t->h_xpad[set] + (size_t)G.first * TL_MAX_XPAD, n * TL_MAX_XPAD, hipMemcpyHostToDevice, t->s_in);
        if (e == hipSuccess && t->with_xpad) e = hipMemcpyAsync(G.d_xl, t->h_xl[set] + G.first, n * sizeof(int32_t), hipMemcpyHostToDevice, t->s_in);
        if (e == hipSuccess) e = hipEventRecord(G.ev_in, t->s_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(t->s_run, G.ev_in, 0);
        if (e == hipSuccess && t->ticks > 0) e = hipStreamWaitEvent(t->s_run, G.ev_out, 0);
        if (e != hipSuccess) rc = TLB_ERR_HIP;
        if (!rc) rc = tlb_ingest_device(G.b, G.d_inter, 1, G.d_pcm, G.d_peaks, t->s_run);
        if (!rc && hipEventRecord(G.ev_ingested, t->s_run) != hipSuccess) rc = TLB_ERR_HIP;
        if (!rc) rc = tlb_launch(G.b, G.d_pcm, 1, t->with_xpad ? G.d_xpad : nullptr, t->with_xpad ? G.d_xl : nullptr, G.d_frames, nullptr, t->s_run, nullptr, G.d_flen);
        if (!rc && hipEventRecord(G.ev_encoded, t->s_run) != hipSuccess) rc = TLB_ERR_HIP;
        if (!rc) rc = tick_egress(t, G, t->ticks > 0, oset);
        if (rc) return tick_fail(t, rc);
    }
    if (hipEventRecord(t->ev1[oset], t->s_out) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    t->ticks++;
    t->in_set = (int)(t->ticks & 1);
    return TLB_OK;
}

// Wait for the oldest submitted tick; the read accessors then show ITS results until the next wait (three output sets: neither
// of the two ticks that can be submitted before that wait writes the set this one's results are in).
int tlb_tick_wait(tlb_tick *t)
{
    if (!t || t->waited >= t->ticks) return TLB_ERR_ARG;
    if (t->broken) return TLB_ERR_HIP;
    if (hipSetDevice(t->device) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    const int set = (int)(t->waited % 3);
    if (hipEventSynchronize(t->ev1[set]) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    t->out_set = set;
    t->waited++;
    return TLB_OK;
}

int tlb_tick_run(tlb_tick *t)
{   // one tick start to end: the accessors show its results when the call returns
    if (!t || t->ticks != t->waited) return TLB_ERR_ARG;
    if (t->broken) return TLB_ERR_HIP;
    if (int rc = tlb_tick_submit(t)) return rc;
    return tlb_tick_wait(t);
}

// end of the streams (toolame_finish): the pending frame of every stream through the egress stage; no further run
int tlb_tick_finish(tlb_tick *t)
{
    if (!t || t->finished || t->ticks == 0 || t->ticks != t->waited) return TLB_ERR_ARG;
    if (t->broken) return TLB_ERR_HIP;
    if (hipSetDevice(t->device) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    const int set = (int)(t->ticks % 3);
    for (auto &G : t->groups) {
        int rc = hipStreamWaitEvent(t->s_run, G.ev_out, 0) == hipSuccess ? TLB_OK : TLB_ERR_HIP;
        if (!rc) rc = tlb_flush_device_len(G.b, G.d_frames, G.d_flen, t->s_run);
        if (!rc) {      // the egress sends the levels of the last run with the last frame (they are in the other host set; the device copy is current)
            rc = tick_egress(t, G, true, set, false);
        }
        if (rc) return tick_fail(t, rc);
    }
    if (hipStreamSynchronize(t->s_out) != hipSuccess) return tick_fail(t, TLB_ERR_HIP);
    t->out_set = set;
    t->finished = true;
    return TLB_OK;
}

const uint32_t *tlb_tick_silence_ms(const tlb_tick *t) { return t ? t->h_silence[t->out_set] : nullptr; }
const uint8_t *tlb_tick_message(const tlb_tick *t, int stream, int unit, int *len)
{   // ZeroMQ message = zmq_frame_header_t + unit; the header's datasize field says how much follows (0: absent)
    if (!t || stream < 0 || stream >= t->nstreams || t->egress != TLB_TICK_ZMQ) return nullptr;
    const TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    if (unit < 0 || unit >= G.max_upf) return nullptr;
    const uint8_t *m = G.h_msgs[t->out_set] + ((size_t)unit * (size_t)G.n + (size_t)(stream - G.first)) * (size_t)G.msg_stride;
    uint32_t ds; memcpy(&ds, m + 4, 4);
    if (len) *len = ds ? (int)(12 + ds) : 0;                         // (a set no tick has written yet is all zeros)
    return m;
}
int tlb_tick_units(const tlb_tick *t, int stream)
{
    if (!t || stream < 0 || stream >= t->nstreams) return 0;
    const TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    if (t->egress == TLB_TICK_FRAMES) return 1;
    return tlb_egress_units_per_frame(G.b, stream - G.first);
}
const uint8_t *tlb_tick_frame(const tlb_tick *t, int stream, int *len)
{
    if (!t || stream < 0 || stream >= t->nstreams || t->egress != TLB_TICK_FRAMES) return nullptr;
    const TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    if (len) *len = G.h_flen[t->out_set][stream - G.first];
    return G.h_frames[t->out_set] + (size_t)(stream - G.first) * (size_t)G.out_stride;
}
const uint8_t *tlb_tick_packet(const tlb_tick *t, int stream, int unit, int *len)
{
    if (!t || stream < 0 || stream >= t->nstreams || t->egress != TLB_TICK_EDI_AF) return nullptr;
    const TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    if (unit < 0 || unit >= G.max_upf) return nullptr;
    const size_t slot = (size_t)unit * (size_t)G.n + (size_t)(stream - G.first);
    if (len) *len = G.h_plen[t->out_set][slot];
    return G.h_pkts[t->out_set] + slot * (size_t)G.af_stride;
}
int tlb_tick_fragments(const tlb_tick *t, int stream, int unit)
{
    if (!t || stream < 0 || stream >= t->nstreams || t->egress != TLB_TICK_EDI_PFT) return 0;
    const TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    if (unit < 0 || unit >= G.max_upf) return 0;
    return G.h_nfrag[t->out_set][(size_t)unit * (size_t)G.n + (size_t)(stream - G.first)];
}
const uint8_t *tlb_tick_fragment(const tlb_tick *t, int stream, int unit, int k, int *len)
{
    if (k < 0 || k >= tlb_tick_fragments(t, stream, unit)) return nullptr;
    const TickGroup &G = t->groups[(size_t)t->group_of[(size_t)stream]];
    const size_t slot = (size_t)unit * (size_t)G.n + (size_t)(stream - G.first);
    if (len) *len = G.h_fraglen[t->out_set][slot * (size_t)G.max_frags + (size_t)k];
    return G.h_frags[t->out_set] + (slot * (size_t)G.max_frags + (size_t)k) * (size_t)G.frag_stride;
}
float tlb_tick_last_ms(tlb_tick *t)
{   // first copy-in queued -> last copy-out done, on the device's clock
    float ms = -1.0f;
    if (!t || !t->waited || hipSetDevice(t->device) != hipSuccess) return -1.0f;
    const int set = (int)((t->waited - 1) % 3);                      // the tick waited for last
    if (hipEventSynchronize(t->ev1[set]) != hipSuccess || hipEventElapsedTime(&ms, t->ev0[set], t->ev1[set]) != hipSuccess) return -1.0f;
    return ms;
}

}  // extern "C"
