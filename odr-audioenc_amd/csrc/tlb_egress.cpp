// tlb_egress.cpp -- the step after the path, host side: ZeroMQ messages, EDI AF packets and PFT fragments of the frames a
// batch produced (include/toolame_batch.h).  The kernels are csrc/edi_af.h / edi_pft.h behind tl_kernels.h.
#include "tlb_internal.h"

extern "C" {

int tlb_zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream)
{
    return zmq_frame_device(b, d_frames, d_peaks, nframes, d_msgs, hip_stream, nullptr);
}
// d_frame_len: int32 [nframes][nstreams] or null -- 0 marks a slot without a frame (a stream just reset inside a tick object): no message
}  // extern "C"
int zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream, const int32_t *d_frame_len)
{
    if (!b || !d_frames || !d_msgs || nframes <= 0) return TLB_ERR_ARG;
    if (!b->max_upf) return TLB_ERR_SAMPLERATE;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(tlk_zmq_frame((unsigned)((size_t)nframes * (size_t)b->max_upf * (size_t)b->nstreams), (hipStream_t)hip_stream, d_frames, d_peaks, d_msgs, b->d_configs,
                         b->d_stream_cfg, b->nstreams, b->out_stride, 12 + b->out_stride, b->max_upf, d_frame_len));
    return TLB_OK;
}
extern "C" {

int tlb_zmq_frame_host(tlb_batch *b, const uint8_t *frames, const int16_t *peaks, int nframes, uint8_t *msgs)
{
    DevFree guard_;
    if (!b || !frames || !msgs || nframes <= 0) return TLB_ERR_ARG;
    if (!b->max_upf) return TLB_ERR_SAMPLERATE;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams, ms = 12 + (size_t)b->out_stride, pslots = slots * (size_t)b->max_upf;
    uint8_t *d_f = nullptr, *d_m = nullptr; int16_t *d_p = nullptr;
    DEVALLOC(d_f, slots * (size_t)b->out_stride);
    DEVALLOC(d_m, pslots * ms);
    HIPCHK(hipMemset(d_m, 0, pslots * ms));
    HIPCHK(hipMemcpy(d_f, frames, slots * (size_t)b->out_stride, hipMemcpyHostToDevice));
    if (peaks) { DEVALLOC(d_p, slots * 4); HIPCHK(hipMemcpy(d_p, peaks, slots * 4, hipMemcpyHostToDevice)); }
    int rc = tlb_zmq_frame_device(b, d_f, d_p, nframes, d_m, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(msgs, d_m, pslots * ms, hipMemcpyDeviceToHost));

    return rc;
}

// ---- EDI AF packets (include/toolame_batch.h) ----
static_assert(sizeof(tlb_edi_state) == sizeof(TlEdiState), "tlb_edi_state mirrors TlEdiState");

void tlb_edi_state_init(tlb_edi_state *st, long long now_s, unsigned delay_ms, int tist, int tai_utc_offset)
{   // the first-call branch of EDI::write_frame (src/Outputs.cpp:200-212)
    if (!st) return;
    memset(st, 0, sizeof *st);
    st->edi_time = now_s + delay_ms / 1000;
    st->send_version_at_time = st->edi_time;
    for (int sub_ms = (int)(delay_ms % 1000); sub_ms > 0; sub_ms -= 24) st->timestamp += 24u << 14;
    st->tist = tist ? 1 : 0;
    st->tai_utc_offset = tai_utc_offset;
}

int tlb_edi_af_stride(const tlb_batch *b, int version_len)
{
    if (!b || version_len < 0 || version_len > TL_EDI_MAX_VERSION) return 0;
    return (10 + 16 + 18 + 11 + b->out_stride + 12 + 12 + version_len + 2 + 3) & ~3;
}

int tlb_edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
                      const char *version, int version_len, uint8_t *d_pkts, int32_t *d_pkt_len, void *hip_stream)
{
    return edi_af_device(b, d_frames, d_levels, nframes, d_state, version, version_len, d_pkts, d_pkt_len, hip_stream, nullptr);
}
// d_frame_len: int32 [nframes][nstreams] or null -- 0 marks a slot without a frame: no packet, sender state untouched (csrc/edi_af.h)
}  // extern "C"
int edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
                  const char *version, int version_len, uint8_t *d_pkts, int32_t *d_pkt_len, void *hip_stream, const int32_t *d_frame_len)
{
    if (!b || !d_frames || !d_state || !d_pkts || !d_pkt_len || nframes <= 0 || nframes > 65535 || version_len < 0 || version_len > TL_EDI_MAX_VERSION ||
        (version_len && !version) || (long)nframes * (b->max_upf ? b->max_upf : 1) > 65535) return TLB_ERR_ARG;
    if (!b->max_upf) return TLB_ERR_SAMPLERATE;
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (!b->d_edi_version) {
        // all three buffers or none: a failure half way must not leave the batch looking initialised
        DevFree guard_;
        uint8_t *d_v = nullptr; int32_t *d_fb = nullptr, *d_ub = nullptr; TlEdiState *d_st = nullptr;
        DEVALLOC(d_v, TL_EDI_MAX_VERSION);
        DEVALLOC(d_fb, sizeof(int32_t) * (size_t)b->nstreams);
        DEVALLOC(d_ub, sizeof(int32_t) * (size_t)b->nstreams);
        DEVALLOC(d_st, sizeof(TlEdiState) * (size_t)b->nstreams);
        std::vector<int32_t> fb((size_t)b->nstreams), ub((size_t)b->nstreams);
        for (int s = 0; s < b->nstreams; s++) { fb[(size_t)s] = b->h_configs[b->h_stream_cfg[s]].frame_bytes; ub[(size_t)s] = 3 * b->h_configs[b->h_stream_cfg[s]].kbps; }
        HIPCHK(hipMemcpy(d_fb, fb.data(), sizeof(int32_t) * fb.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ub, ub.data(), sizeof(int32_t) * ub.size(), hipMemcpyHostToDevice));
        guard_.v.clear();
        b->d_edi_version = d_v; b->d_frame_bytes = d_fb; b->d_unit_bytes = d_ub; b->d_edi_state_tmp = d_st;
    }
    // the ODRv string goes to the device when it changes, not on every call (an asynchronous copy from pageable memory may be
    // staged or run synchronously: it would serialise the groups of a tick)
    if (version_len && (version_len != b->edi_version_len || memcmp(b->h_edi_version, version, (size_t)version_len) != 0)) {
        memcpy(b->h_edi_version, version, (size_t)version_len); b->edi_version_len = version_len;
        HIPCHK(hipMemcpyAsync(b->d_edi_version, b->h_edi_version, (size_t)version_len, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));                            // once per string: the host copy may change after this call returns
    }
    TlEdiArgs A;
    A.frame_len = d_frame_len;
    A.frames = d_frames; A.levels = d_levels; A.state = (const TlEdiState *)d_state; A.state_out = b->d_edi_state_tmp; A.version = b->d_edi_version;
    A.xpow8 = b->d_tables->edi_xpow8; A.frame_bytes = b->d_frame_bytes; A.unit_bytes = b->d_unit_bytes; A.pkts = d_pkts; A.pkt_len = d_pkt_len;
    A.nstreams = b->nstreams; A.nframes = nframes; A.out_stride = b->out_stride; A.max_upf = b->max_upf;
    A.pkt_stride = tlb_edi_af_stride(b, version_len); A.version_len = version_len;
    HIPCHK(tlk_edi_af((unsigned)((b->nstreams + 3) / 4), (unsigned)(nframes * b->max_upf), st, A));
    HIPCHK(hipMemcpyAsync(d_state, b->d_edi_state_tmp, sizeof(TlEdiState) * (size_t)b->nstreams, hipMemcpyDeviceToDevice, st));
    return TLB_OK;
}

extern "C" {
int tlb_edi_af_host(tlb_batch *b, const uint8_t *frames, const int16_t *levels, int nframes, tlb_edi_state *state,
                    const char *version, int version_len, uint8_t *pkts, int32_t *pkt_len)
{
    DevFree guard_;
    if (!b || !frames || !state || !pkts || !pkt_len || nframes <= 0) return TLB_ERR_ARG;
    const int stride = tlb_edi_af_stride(b, version_len);
    if (!stride) return TLB_ERR_ARG;
    if (!b->max_upf) return TLB_ERR_SAMPLERATE;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams, pslots = slots * (size_t)b->max_upf;
    uint8_t *d_f = nullptr, *d_p = nullptr; int16_t *d_l = nullptr; tlb_edi_state *d_s = nullptr; int32_t *d_n = nullptr;
    DEVALLOC(d_f, slots * (size_t)b->out_stride);
    DEVALLOC(d_p, pslots * (size_t)stride);
    DEVALLOC(d_s, sizeof(tlb_edi_state) * (size_t)b->nstreams);
    DEVALLOC(d_n, sizeof(int32_t) * pslots);
    HIPCHK(hipMemset(d_p, 0, pslots * (size_t)stride));
    HIPCHK(hipMemcpy(d_f, frames, slots * (size_t)b->out_stride, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_s, state, sizeof(tlb_edi_state) * (size_t)b->nstreams, hipMemcpyHostToDevice));
    if (levels) { DEVALLOC(d_l, slots * 4); HIPCHK(hipMemcpy(d_l, levels, slots * 4, hipMemcpyHostToDevice)); }
    int rc = tlb_edi_af_device(b, d_f, d_l, nframes, d_s, version, version_len, d_p, d_n, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(pkts, d_p, pslots * (size_t)stride, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(pkt_len, d_n, sizeof(int32_t) * pslots, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(state, d_s, sizeof(tlb_edi_state) * (size_t)b->nstreams, hipMemcpyDeviceToHost);

    if (e != hipSuccess) return TLB_ERR_HIP;
    return rc;
}

// ---- EDI PFT layer (include/toolame_batch.h) ----
}  // extern "C"
int pft_shape(int max_af_len, int fec, int chunk_len, int transport, int *max_frags, int *frag_stride)
{   // largest fragment count and fragment size over every AF packet length the batch can produce (PFT.cpp:166-176,199-209)
    if (fec < 0 || fec > 5 || chunk_len < 1 || chunk_len > 207 || max_af_len < 1) return TLB_ERR_ARG;
    int mf = 0, ms = 0;
    for (int l = 1; l <= max_af_len; l++) {
        int nfr, fsz;
        if (fec > 0) {
            const int c = (l + chunk_len - 1) / chunk_len, k = (l + c - 1) / c, total = c * (k + 48), smax = (c * 48) / (fec + 1);
            nfr = (total + smax - 1) / smax; fsz = (total + nfr - 1) / nfr;
            if (c > TL_PFT_MAX_CHUNKS) return TLB_ERR_ARG;
        } else { nfr = (l + 1399) / 1400; fsz = (l + nfr - 1) / nfr; }
        if (nfr > mf) mf = nfr;
        if (fsz > ms) ms = fsz;
    }
    *max_frags = mf;
    *frag_stride = (12 + (fec > 0 ? 2 : 0) + (transport ? 4 : 0) + 2 + ms + 3) & ~3;
    return TLB_OK;
}

extern "C" {
int tlb_edi_pft_shape(const tlb_batch *b, int af_stride, int fec, int chunk_len, int transport, int *max_frags, int *frag_stride)
{
    if (!b || !max_frags || !frag_stride) return TLB_ERR_ARG;
    return pft_shape(af_stride, fec, chunk_len, transport, max_frags, frag_stride);
}

int tlb_edi_pft_device(tlb_batch *b, const uint8_t *d_af, const int32_t *d_af_len, int nframes, int af_stride, uint16_t *d_pseq,
                       int fec, int chunk_len, int transport, int addr_source, int dest_port,
                       uint8_t *d_frags, int32_t *d_frag_len, int32_t *d_nfrag, int max_frags, int frag_stride, void *hip_stream)
{
    if (!b || !d_af || !d_af_len || !d_pseq || !d_frags || !d_frag_len || !d_nfrag || nframes <= 0 || nframes > 65535 || af_stride <= 0 || af_stride > 2048 || (af_stride & 3)) return TLB_ERR_ARG;
    int mf = 0, fs = 0;
    if (int rc = pft_shape(af_stride, fec, chunk_len, transport, &mf, &fs)) return rc;
    if (max_frags < mf || frag_stride < fs) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (!b->d_pseq_tmp) HIPCHK(hipMalloc(&b->d_pseq_tmp, sizeof(uint16_t) * (size_t)b->nstreams));
    TlPftArgs A;
    A.af = d_af; A.af_len = d_af_len; A.pseq = d_pseq; A.pseq_out = b->d_pseq_tmp;
    A.frags = d_frags; A.frag_len = d_frag_len; A.nfrag = d_nfrag;
    A.nstreams = b->nstreams; A.nframes = nframes; A.af_stride = af_stride; A.max_frags = max_frags; A.frag_stride = frag_stride;
    A.fec = fec; A.chunk_len = chunk_len; A.transport = transport ? 1 : 0; A.addr_source = addr_source; A.dest_port = dest_port;
    HIPCHK(tlk_edi_pft((unsigned)((b->nstreams + 3) / 4), (unsigned)nframes, st, A, (const TlTables *)b->d_tables));
    HIPCHK(hipMemcpyAsync(d_pseq, b->d_pseq_tmp, sizeof(uint16_t) * (size_t)b->nstreams, hipMemcpyDeviceToDevice, st));
    return TLB_OK;
}

int tlb_edi_pft_host(tlb_batch *b, const uint8_t *af, const int32_t *af_len, int nframes, int af_stride, uint16_t *pseq,
                     int fec, int chunk_len, int transport, int addr_source, int dest_port,
                     uint8_t *frags, int32_t *frag_len, int32_t *nfrag, int max_frags, int frag_stride)
{
    DevFree guard_;
    if (!b || !af || !af_len || !pseq || !frags || !frag_len || !nfrag || nframes <= 0 || af_stride <= 0 || max_frags <= 0 || frag_stride <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams;
    uint8_t *d_a = nullptr, *d_f = nullptr; int32_t *d_l = nullptr, *d_fl = nullptr, *d_n = nullptr; uint16_t *d_p = nullptr;
    DEVALLOC(d_a, slots * (size_t)af_stride);
    DEVALLOC(d_l, slots * 4);
    DEVALLOC(d_f, slots * (size_t)max_frags * (size_t)frag_stride);
    DEVALLOC(d_fl, slots * (size_t)max_frags * 4);
    DEVALLOC(d_n, slots * 4);
    DEVALLOC(d_p, sizeof(uint16_t) * (size_t)b->nstreams);
    HIPCHK(hipMemset(d_f, 0, slots * (size_t)max_frags * (size_t)frag_stride));
    HIPCHK(hipMemset(d_fl, 0, slots * (size_t)max_frags * 4));
    HIPCHK(hipMemcpy(d_a, af, slots * (size_t)af_stride, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_l, af_len, slots * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_p, pseq, sizeof(uint16_t) * (size_t)b->nstreams, hipMemcpyHostToDevice));
    int rc = tlb_edi_pft_device(b, d_a, d_l, nframes, af_stride, d_p, fec, chunk_len, transport, addr_source, dest_port, d_f, d_fl, d_n, max_frags, frag_stride, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(frags, d_f, slots * (size_t)max_frags * (size_t)frag_stride, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(frag_len, d_fl, slots * (size_t)max_frags * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(nfrag, d_n, slots * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(pseq, d_p, sizeof(uint16_t) * (size_t)b->nstreams, hipMemcpyDeviceToHost);

    if (e != hipSuccess) return TLB_ERR_HIP;
    return rc;
}

}  // extern "C"
