// tlb_batch.cpp -- the batch object of include/toolame_batch.h part (2): creation, the per-model kernel lists, ONE launch of the encode path
// (tlb_launch), the host-buffer entry points with their copy pipeline, the caller's ingest glue and silence counter, the life cycle of one
// stream, flush, the reference's send schedule, the libm self-check and the timing accessors.  Host C++: the kernels are reached through
// tl_kernels.h.
#include "tlb_internal.h"
#include "tlb_plan.h"
#include "tl_libm.h"

extern "C" {

int tlb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int tlb_lds_bytes_per_stream(void) { return (int)tlk_lds_bytes_per_wave(); }      // per WAVE (= per unit in flight): the largest of the kernels' per-wave blocks
#define TLB_STR2(x) #x
#define TLB_STR(x) TLB_STR2(x)
// names the toolchain the kernels came out of: their shape (registers, LDS instruction forms) depends on compiler internals that
// csrc/Makefile sets and tools/check_isa.py verifies on the linked code objects at build time
const char *tlb_version(void)
{
    return "odr-audioenc_amd 0.5 (gfx950, a wavefront per (stream, frame), fp64, glibc 2.35 transcendentals; built with HIP "
           TLB_STR(HIP_VERSION_MAJOR) "." TLB_STR(HIP_VERSION_MINOR) "." TLB_STR(HIP_VERSION_PATCH) ", clang " __clang_version__ ", ISA guard passed)";
}

void tlb_destroy(tlb_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->d_tables) (void)hipFree(b->d_tables);
    if (b->d_configs) (void)hipFree(b->d_configs);
    if (b->d_stream_cfg) (void)hipFree(b->d_stream_cfg);
    if (b->d_state) (void)hipFree(b->d_state);
    if (b->d_gain) (void)hipFree(b->d_gain);
    if (b->d_edi_version) (void)hipFree(b->d_edi_version);
    if (b->d_frame_bytes) (void)hipFree(b->d_frame_bytes);
    if (b->d_unit_bytes) (void)hipFree(b->d_unit_bytes);
    if (b->d_edi_state_tmp) (void)hipFree(b->d_edi_state_tmp);
    if (b->d_pseq_tmp) (void)hipFree(b->d_pseq_tmp);
    for (int p = 0; p < 4; p++) if (b->d_list[p]) (void)hipFree(b->d_list[p]);
    for (int k = 0; k < 12; k++) if (b->stage[k]) (void)hipFree(b->stage[k]);
    for (int i = 0; i < TLB_HOST_CHUNKS; i++) { if (b->ev_in[i]) (void)hipEventDestroy(b->ev_in[i]); if (b->ev_run[i]) (void)hipEventDestroy(b->ev_run[i]); }
    if (b->s_in) (void)hipStreamDestroy(b->s_in);
    if (b->s_run) (void)hipStreamDestroy(b->s_run);
    if (b->s_out) (void)hipStreamDestroy(b->s_out);
    if (b->d_newpend) (void)hipFree(b->d_newpend);
    if (b->d_work) (void)hipFree(b->d_work);
    if (b->d_newlag) (void)hipFree(b->d_newlag);
    if (b->d_psy2_tables) (void)hipFree(b->d_psy2_tables);
    if (b->d_psy2_state) (void)hipFree(b->d_psy2_state);
    if (b->d_chain) (void)hipFree(b->d_chain);
    if (b->d_partner) (void)hipFree(b->d_partner);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->ev_mid) (void)hipEventDestroy(b->ev_mid);
    delete b;
}

// Everything that follows from WHICH stream has WHICH configuration: the per-model stream lists of the kernels, the psy-2 kernel's
// chains, the padding flags, and -- allocated the first time a stream needs them -- the psy 2/4 tables and state and the slot
// recurrence's scratch.  Called at creation and again when a stream is reconfigured (tlb_stream_reconfigure).
static int batch_build_lists(tlb_batch *b)
{
    const int nstreams = b->nstreams;
    for (int p = 0; p < 4; p++) {
        std::vector<int32_t> ids;
        b->pads[p] = false;
        // kernel p serves psy model p; model 4 runs the psy-2 kernel on its own tables (mp2_host.cpp: tl_build_psy4_tables)
        for (int s2 = 0; s2 < nstreams; s2++) { const int m = b->h_configs[b->h_stream_cfg[s2]].psy; if ((m == 4 ? 2 : m) == p) { ids.push_back(s2); b->pads[p] |= b->h_configs[b->h_stream_cfg[s2]].pad_frac != 0; } }
        b->n_list[p] = (int)ids.size();
        if (ids.empty()) continue;
        if (!b->d_list[p]) HIPCHK(hipMalloc(&b->d_list[p], sizeof(int32_t) * (size_t)nstreams));       // room for every stream: a list only changes its content later
        HIPCHK(hipMemcpy(b->d_list[p], ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice));
    }
    {   // mono streams of the same configuration (hence the same model and kernel) in pairs: consecutive ones of the stream order
        std::vector<int32_t> partner;
        tlb_plan_pairs(b->h_configs, b->h_stream_cfg, partner);                                  // (csrc/tlb_plan.h: the one statement of the pairing)
        for (int p = 0; p < 4; p++) { b->list_pairs[p] = false; b->list_stereo[p] = b->n_list[p] > 0; }
        for (int s2 = 0; s2 < nstreams; s2++) {
            const TlConfig &c2 = b->h_configs[b->h_stream_cfg[s2]];
            if (c2.nch != 2) b->list_stereo[c2.psy == 4 ? 2 : c2.psy] = false;
        }
        for (int s2 = 0; s2 < nstreams; s2++)
            if (partner[(size_t)s2] >= 0) { const int m = b->h_configs[b->h_stream_cfg[s2]].psy; b->list_pairs[m == 4 ? 2 : m] = true; }
        if (!b->d_partner) HIPCHK(hipMalloc(&b->d_partner, sizeof(int32_t) * (size_t)nstreams));
        HIPCHK(hipMemcpy(b->d_partner, partner.data(), sizeof(int32_t) * (size_t)nstreams, hipMemcpyHostToDevice));
    }
    if (b->n_list[2]) {
        if (!b->d_psy2_tables) {
            const long rates[TL_PSY2_SLOTS] = {48000, 32000, 24000, 16000, 44100, 22050};
            std::vector<TlPsy2Tables> ht2(2 * TL_PSY2_SLOTS);            // psy 2 per rate, then psy 4 per rate
            for (int i = 0; i < TL_PSY2_SLOTS; i++) {
                tl_build_psy2_tables(&ht2[tl_psy2_slot(rates[i])], rates[i]);
                tl_build_psy4_tables(&ht2[TL_PSY2_SLOTS + tl_psy2_slot(rates[i])], rates[i]);
            }
            HIPCHK(hipMalloc(&b->d_psy2_tables, sizeof(TlPsy2Tables) * ht2.size()));
            HIPCHK(hipMemcpy(b->d_psy2_tables, ht2.data(), sizeof(TlPsy2Tables) * ht2.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMalloc(&b->d_psy2_state, sizeof(TlPsy2State) * 2 * (size_t)nstreams));
            HIPCHK(hipMemset(b->d_psy2_state, 0, sizeof(TlPsy2State) * 2 * (size_t)nstreams));
            HIPCHK(hipMalloc(&b->d_chain, sizeof(int32_t) * 2 * (size_t)nstreams));
        }
        std::vector<int32_t> chains;
        for (int ch = 0; ch < 2; ch++)
            for (int s2 = 0; s2 < nstreams; s2++) {
                const TlConfig &c = b->h_configs[b->h_stream_cfg[s2]];
                if ((c.psy == 2 || c.psy == 4) && ch < c.nch) chains.push_back(s2 | (ch << 30));
            }
        b->n_chain = (int)chains.size();
        HIPCHK(hipMemcpy(b->d_chain, chains.data(), sizeof(int32_t) * chains.size(), hipMemcpyHostToDevice));
    } else b->n_chain = 0;
    if ((b->pads[0] || b->pads[1] || b->pads[2] || b->pads[3]) && !b->d_newlag) {
        HIPCHK(hipMalloc(&b->d_newlag, sizeof(double) * (size_t)nstreams));
        HIPCHK(hipMemset(b->d_newlag, 0, sizeof(double) * (size_t)nstreams));
    }
    return TLB_OK;
}

static int tlb_create_impl(tlb_batch *b, int device, int nstreams, const tlb_stream_config *cfgs)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return TLB_ERR_NO_DEVICE;
    b->device = device;
    b->nstreams = nstreams;
    b->h_stream_cfg.resize(nstreams);
    // streams sharing the six knobs share one config record (keeps the tables L2/L1 resident)
    if (int rc = tlb_plan_configs(nstreams, cfgs, b->h_uniq, b->h_configs, b->h_stream_cfg)) return rc;          // (csrc/tlb_plan.h)
    for (int s = 0; s < nstreams; s++) {
        const int found = b->h_stream_cfg[s];
        {
            const int longest = (b->h_configs[found].frame_bytes + (b->h_configs[found].pad_frac != 0 ? 1 : 0) + 3) & ~3;
            if (longest > b->out_stride) b->out_stride = longest;
        }
        {
            const int unit = 3 * b->h_configs[found].kbps, fb = b->h_configs[found].frame_bytes;
            if (fb % unit) b->max_upf = 0;                          // 32 kHz: 1.5 units per frame -- not a DAB rate (odr-audioenc.cpp:560-563)
            else if (b->max_upf && fb / unit > b->max_upf) b->max_upf = fb / unit;
        }
    }
    HIPCHK(hipSetDevice(device));
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) b->num_cu = n; }
    TlTables *ht = new TlTables;
    tl_build_tables(ht);
    hipError_t e = hipMalloc(&b->d_tables, sizeof(TlTables));
    if (e == hipSuccess) e = hipMemcpy(b->d_tables, ht, sizeof(TlTables), hipMemcpyHostToDevice);
    delete ht;
    HIPCHK(e);
    b->cfg_cap = b->h_configs.size() + 8;                             // room for a few reconfigurations before the array has to move
    HIPCHK(hipMalloc(&b->d_configs, sizeof(TlConfig) * b->cfg_cap));
    HIPCHK(hipMemcpy(b->d_configs, b->h_configs.data(), sizeof(TlConfig) * b->h_configs.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&b->d_stream_cfg, sizeof(int32_t) * nstreams));
    HIPCHK(hipMemcpy(b->d_stream_cfg, b->h_stream_cfg.data(), sizeof(int32_t) * nstreams, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&b->d_state, sizeof(TlStreamState) * (size_t)nstreams));
    HIPCHK(hipMemset(b->d_state, 0, sizeof(TlStreamState) * (size_t)nstreams));
    b->h_gain.assign((size_t)nstreams, 1.0);
    HIPCHK(hipMalloc(&b->d_gain, sizeof(double) * (size_t)nstreams));
    HIPCHK(hipMemcpy(b->d_gain, b->h_gain.data(), sizeof(double) * (size_t)nstreams, hipMemcpyHostToDevice));
    if (int rc = batch_build_lists(b)) return rc;
    {
        HIPCHK(hipMalloc(&b->d_newpend, sizeof(uint32_t) * TL_MAX_FRAME_WORDS * (size_t)nstreams));
        HIPCHK(hipMemset(b->d_newpend, 0, sizeof(uint32_t) * TL_MAX_FRAME_WORDS * (size_t)nstreams));
        HIPCHK(hipMalloc(&b->d_work, sizeof(int32_t) * TL_HEAD_STRIDE * 9));
    }
    HIPCHK(hipEventCreate(&b->ev0));
    HIPCHK(hipEventCreate(&b->ev1));
    HIPCHK(hipEventCreate(&b->ev_mid));
    return TLB_OK;
}

tlb_batch *tlb_create(int device, int nstreams, const tlb_stream_config *cfgs, int *err)
{
    if (nstreams <= 0 || !cfgs) { if (err) *err = TLB_ERR_ARG; return nullptr; }
    tlb_batch *b = new tlb_batch;
    int rc = tlb_create_impl(b, device, nstreams, cfgs);
    if (err) *err = rc;
    if (rc) { tlb_destroy(b); return nullptr; }
    return b;
}

// state of streams [s0, s0 + n) back to what tlb_create() left: the PCM history, the pending frame, the frame counter and the slot
// recurrence (TlStreamState), the psy 2/4 prediction state (both copies), the launch scratch that is per stream
static int batch_clear_streams(tlb_batch *b, int s0, int n)
{
    HIPCHK(hipMemset(b->d_state + s0, 0, sizeof(TlStreamState) * (size_t)n));
    if (b->d_psy2_state) HIPCHK(hipMemset(b->d_psy2_state + 2 * (size_t)s0, 0, sizeof(TlPsy2State) * 2 * (size_t)n));
    HIPCHK(hipMemset(b->d_newpend + (size_t)s0 * TL_MAX_FRAME_WORDS, 0, sizeof(uint32_t) * TL_MAX_FRAME_WORDS * (size_t)n));
    if (b->d_newlag) HIPCHK(hipMemset(b->d_newlag + s0, 0, sizeof(double) * (size_t)n));
    if (b->d_edi_state_tmp) HIPCHK(hipMemset(b->d_edi_state_tmp + s0, 0, sizeof(TlEdiState) * (size_t)n));
    if (b->d_pseq_tmp) HIPCHK(hipMemset(b->d_pseq_tmp + s0, 0, sizeof(uint16_t) * (size_t)n));
    return TLB_OK;
}

int tlb_reset(tlb_batch *b)
{
    if (!b) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipDeviceSynchronize());
    // `broken` has two causes: a launch that failed half way (stream state in motion) and a reconfiguration whose roll-back failed too
    // (tlb_stream_reconfigure: the device's stream -> configuration table, the kernel lists and the mono pairing may then disagree with
    // the host's).  So the device side of everything batch_build_lists() derives is rebuilt from the HOST tables first, and the flag
    // only falls when that and the zeroing went through; otherwise the batch stays refused and tlb_destroy() is the way out.
    HIPCHK(hipMemcpy(b->d_configs, b->h_configs.data(), sizeof(TlConfig) * b->h_configs.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_stream_cfg, b->h_stream_cfg.data(), sizeof(int32_t) * (size_t)b->nstreams, hipMemcpyHostToDevice));
    if (int rc = batch_build_lists(b)) return rc;
    if (int rc = batch_clear_streams(b, 0, b->nstreams)) return rc;
    b->frames = 0; b->psy2_flip = 0; b->work_clean = false; b->broken = false;
    return TLB_OK;
}

// ---- life cycle of ONE stream inside a live batch (include/toolame_batch.h) ----
// The reference's unit of restart is the stream: toolame_init() zeroes one encoder (toolame.c:120-153), toolame_finish() ends one
// (:155-166).  Here thousands share a batch, so the same three operations exist per stream; each waits for the batch's queued
// launches first (they are rare events next to 41.7 frames per second and stream) and touches nothing of any other stream.
int tlb_stream_reset(tlb_batch *b, int stream)
{
    if (!b || stream < 0 || stream >= b->nstreams) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipDeviceSynchronize());
    return batch_clear_streams(b, stream, 1);
}

int tlb_stream_finish(tlb_batch *b, int stream, uint8_t *out, size_t out_size)
{   // toolame_finish(): the bytes still inside the encoder -- here the one pending frame -- then the encoder is as after toolame_init()
    if (!b || stream < 0 || stream >= b->nstreams || (!out && out_size)) return -TLB_ERR_ARG;
    if (hipSetDevice(b->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -TLB_ERR_HIP;
    TlStreamState *st = new TlStreamState;
    hipError_t e = hipMemcpy(st, b->d_state + stream, sizeof(TlStreamState), hipMemcpyDeviceToHost);
    int n = 0;
    if (e == hipSuccess && st->frames_done > 0) {
        n = st->pending_len;
        if ((size_t)n > out_size) n = (int)out_size;                 // a too small buffer gets a truncated copy, like the reference's (bitstream.c:54-58)
        for (int i = 0; i < n; i++) out[i] = (uint8_t)(st->pending[i >> 2] >> (24 - 8 * (i & 3)));
    }
    delete st;
    if (e != hipSuccess) return -TLB_ERR_HIP;
    if (int rc = batch_clear_streams(b, stream, 1)) return -rc;
    return n;
}

int tlb_stream_reconfigure(tlb_batch *b, int stream, const tlb_stream_config *cfg)
{   // the setters of toolame.h:13-48 followed by toolame_init() for ONE stream: new sample rate / mode / bitrate / model / PAD length
    if (!b || !cfg || stream < 0 || stream >= b->nstreams) return TLB_ERR_ARG;
    int found = tlb_find_config(b->h_uniq, *cfg);
    TlConfig c;
    if (found < 0) { if (int rc = tl_build_config(&c, cfg->samplerate, cfg->mode, cfg->bitrate, cfg->psy_model, cfg->pad_len)) return rc; }
    else c = b->h_configs[(size_t)found];
    // the caller's buffers were sized from tlb_out_stride() and tlb_egress_max_units_per_frame(): the new configuration must fit them
    if (((c.frame_bytes + (c.pad_frac != 0 ? 1 : 0) + 3) & ~3) > b->out_stride) return TLB_ERR_ARG;
    {
        const int unit = 3 * c.kbps, upf = c.frame_bytes % unit ? 0 : c.frame_bytes / unit;
        if (b->max_upf && (upf == 0 || upf > b->max_upf)) return TLB_ERR_SAMPLERATE;
    }
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipDeviceSynchronize());
    // "nothing changed" on failure: the device side of a NEW record is prepared first -- a bigger array filled completely before the
    // old one is let go, or the record written into a free slot no stream refers to yet -- and only then do the host lists learn of it.
    if (found < 0) {
        const size_t n_old = b->h_configs.size();
        if (n_old + 1 > b->cfg_cap) {
            TlConfig *nd = nullptr;
            const size_t cap = 2 * (n_old + 1);
            HIPCHK(hipMalloc(&nd, sizeof(TlConfig) * cap));
            hipError_t e = hipMemcpy(nd, b->h_configs.data(), sizeof(TlConfig) * n_old, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(nd + n_old, &c, sizeof(TlConfig), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(nd); HIPCHK(e); }
            (void)hipFree(b->d_configs);                             // (the device is idle: hipDeviceSynchronize above)
            b->d_configs = nd; b->cfg_cap = cap;
        } else HIPCHK(hipMemcpy(b->d_configs + n_old, &c, sizeof(TlConfig), hipMemcpyHostToDevice));
        b->h_uniq.push_back(*cfg); b->h_configs.push_back(c);
        found = (int)n_old;
    }
    {
        const int32_t f32 = found;
        HIPCHK(hipMemcpy(b->d_stream_cfg + stream, &f32, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const int32_t before = b->h_stream_cfg[(size_t)stream];
    b->h_stream_cfg[(size_t)stream] = found;
    if (int rc = batch_build_lists(b)) {
        // the lists are rebuilt from the host table: put the stream back and rebuild; if even that fails the batch is marked broken
        b->h_stream_cfg[(size_t)stream] = before;
        if (hipMemcpy(b->d_stream_cfg + stream, &before, sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess || batch_build_lists(b)) b->broken = true;
        return rc;
    }
    if (b->d_frame_bytes) {                                          // EDI egress: per-stream frame and unit sizes
        const int32_t fb = c.frame_bytes, ub = 3 * c.kbps;
        HIPCHK(hipMemcpy(b->d_frame_bytes + stream, &fb, sizeof fb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->d_unit_bytes + stream, &ub, sizeof ub, hipMemcpyHostToDevice));
    }
    return batch_clear_streams(b, stream, 1);
}

int tlb_nstreams(const tlb_batch *b) { return b ? b->nstreams : 0; }
int tlb_frame_bytes(const tlb_batch *b, int s) { return (b && s >= 0 && s < b->nstreams) ? b->h_configs[b->h_stream_cfg[s]].frame_bytes : 0; }
int tlb_out_stride(const tlb_batch *b) { return b ? b->out_stride : 0; }
long tlb_frames_encoded(const tlb_batch *b) { return b ? b->frames : 0; }
#ifdef TLB_FAULT_INJECT
int tlb_debug_fail_next(tlb_batch *b, int nth) { if (!b || nth < 0) return TLB_ERR_ARG; b->fail_in = nth; return TLB_OK; }
#endif

}  // extern "C"
int tlb_launch(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad, const int32_t *d_xpad_len,
               uint8_t *d_out, TlTaps *d_taps, hipStream_t st, long long *d_stamps, int32_t *d_out_len)
{
    if (!b || !d_pcm || !d_out || nframes <= 0) return TLB_ERR_ARG;
    for (int p = 0; p < 4; p++) if ((long)b->n_list[p] * nframes > (1L << 30)) return TLB_ERR_ARG;   // unit indices are 32-bit; checked for every model before anything is queued
    if (b->broken) { fprintf(stderr, "libtoolame-dab-hip: a launch or a reconfiguration of this batch failed half way; tlb_reset() it before encoding on\n"); return TLB_ERR_HIP; }
    HIPCHK(hipSetDevice(b->device));
    // From the first kernel on the streams' state is in motion.  If anything below fails, the psy-2 state copy the next launch would
    // read may never have been written and the unit counters may be non-zero: the flip is taken back, the counters are re-zeroed by
    // the next launch, and the batch refuses further work until tlb_reset().
    struct Guard { tlb_batch *b; int flip; bool ok; ~Guard() { if (!ok) { b->psy2_flip = flip; b->work_clean = false; b->broken = true; } } } guard_{b, b->psy2_flip, false};
#ifdef TLB_FAULT_INJECT
    // test builds only (csrc/tlb_debug.h): the launch armed by tlb_debug_fail_next() fails here as a device call would -- the guard marks the batch broken
    if (b->fail_in > 0 && --b->fail_in == 0) { fprintf(stderr, "libtoolame-dab-hip: injected fault (tlb_debug_fail_next)\n"); return TLB_ERR_HIP; }
#endif
    TlLaunch A;
    memset(&A, 0, sizeof A);
    A.tables = b->d_tables; A.configs = b->d_configs; A.stream_cfg = b->h_configs.size() == 1 ? nullptr : b->d_stream_cfg; A.state = b->d_state;      // one configuration: index 0 for all, no table (tl_cfg_index)
    A.pcm = d_pcm; A.xpad = d_xpad_len ? d_xpad : nullptr; A.xpad_len = d_xpad ? d_xpad_len : nullptr;
    A.out = d_out; A.out_len = d_out_len; A.taps = d_taps; A.stamps = d_stamps;
    A.psy2_tables = b->d_psy2_tables; A.psy2_state = b->d_psy2_state; A.partner = b->d_partner;
    A.nstreams = b->nstreams; A.nframes = nframes; A.out_stride = b->out_stride;
    {   // TlPsyOut records of this launch (psy-2 kernel -> encode kernel, models 2 and 4 only) and ScF-CRC bytes, grow-only.  NOTE: one buffer per batch -- launches of
        // one batch are ordered on one stream (they share the stream state anyway)
        if (b->n_list[2]) HIPCHK(stage_reserve(b, 5, (size_t)nframes * (size_t)b->nstreams * sizeof(TlPsyOut)));
        HIPCHK(stage_reserve(b, 6, (size_t)nframes * (size_t)b->nstreams * 4));
        A.psy_out = (TlPsyOut *)b->stage[5]; A.scfcrc = (uint8_t *)b->stage[6]; A.newpend = b->d_newpend; A.work = b->d_work;
        if (b->pads[0] || b->pads[1] || b->pads[2] || b->pads[3]) HIPCHK(stage_reserve(b, 7, (size_t)nframes * (size_t)b->nstreams));
    }
    HIPCHK(hipEventRecord(b->ev0, st));
    b->have_mid = false;
    for (int p = 0; p < 4; p++) {
        if (!b->n_list[p]) continue;
        A.stream_list = b->n_list[p] == b->nstreams ? nullptr : b->d_list[p]; A.nlist = b->n_list[p];      // one model in the batch: the list is 0, 1, 2, ... and the kernels need no look at it
        // persistent waves, twelve per CU (three per SIMD) in every kernel; they take their units off a counter
        const long units = (long)b->n_list[p] * nframes;
        A.padbits = b->pads[p] ? (uint8_t *)b->stage[7] : nullptr; A.newlag = b->d_newlag;
        if (b->pads[p]) HIPCHK(tlk_slots((unsigned)((b->n_list[p] + 255) / 256), st, A));
        if (!b->work_clean) HIPCHK(hipMemsetAsync(b->d_work, 0, sizeof(int32_t) * TL_HEAD_STRIDE * 9, st));     // only after a launch that failed half way
        b->work_clean = false;
        long qb = 0;
        if (p == 2) {
            A.chain_list = b->d_chain; A.nchain = b->n_chain; A.psy2_flip = b->psy2_flip;
            const int nunits = tl_psy2_plan(b->n_chain, nframes, b->num_cu * TL_PSY2_WAVES, &A.p2_nwhole, &A.p2_k, &A.p2_plen);
            qb = ((long)nunits + TL_PSY2_WAVES - 1) / TL_PSY2_WAVES;
            if (qb > b->num_cu) qb = b->num_cu;
        }
        if (p == 1 || p == 3) {                                      // psy model and encoder in one kernel
            long mb1 = (units + TL_MAIN_WAVES - 1) / TL_MAIN_WAVES;
            if (mb1 > b->num_cu) mb1 = b->num_cu;
            const bool pr = b->list_pairs[p] && !d_taps && !d_stamps;
            HIPCHK(tlk_frame(p, pr, b->list_stereo[p], (unsigned)mb1, st, A));
            HIPCHK(tlk_finish((unsigned)((b->n_list[p] + 3) / 4), st, A));
            b->work_clean = true;                                    // tl_finish_kernel leaves the counters at zero
            continue;
        }
        if (p == 2) HIPCHK(tlk_psy2((unsigned)qb, st, A));
        if (p == 2 && b->n_list[p] == b->nstreams) { HIPCHK(hipEventRecord(b->ev_mid, st)); b->have_mid = true; }      // models 2/4 only in the batch: psy | encode split of the time
        long mb = (units + TL_MAIN_WAVES - 1) / TL_MAIN_WAVES;
        if (mb > b->num_cu) mb = b->num_cu;
        const bool pr = b->list_pairs[p] && !d_taps && !d_stamps;
        HIPCHK(tlk_main(p, pr, b->list_stereo[p], (unsigned)mb, st, A));                // model 0: no psy kernel before it
        HIPCHK(tlk_finish((unsigned)((b->n_list[p] + 3) / 4), st, A));
        b->work_clean = true;
    }
    HIPCHK(hipEventRecord(b->ev1, st));
    if (b->n_list[2]) b->psy2_flip ^= 1;         // only now: every kernel that writes the other copy has been queued
    guard_.ok = true;
    b->last_stream = st; b->timed = true;
    b->frames += nframes;
    return TLB_OK;
}

extern "C" {

int tlb_encode_device(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad, const int32_t *d_xpad_len,
                      uint8_t *d_out, void *hip_stream)
{
    return tlb_launch(b, d_pcm, nframes, d_xpad, d_xpad_len, d_out, nullptr, (hipStream_t)hip_stream);
}
int tlb_encode_device_len(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad, const int32_t *d_xpad_len,
                          uint8_t *d_out, int32_t *d_out_len, void *hip_stream)
{
    return tlb_launch(b, d_pcm, nframes, d_xpad, d_xpad_len, d_out, nullptr, (hipStream_t)hip_stream, nullptr, d_out_len);
}

int tlb_encode_host(tlb_batch *b, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len,
                    uint8_t *out, void *taps)
{
    return tlb_encode_host_len(b, pcm, nframes, xpad, xpad_len, out, nullptr, taps);
}

int tlb_encode_host_len(tlb_batch *b, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len,
                        uint8_t *out, int32_t *out_len, void *taps)
{
    if (!b || !pcm || !out || nframes <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams;
    const size_t n_pcm = slots * 2304 * sizeof(int16_t), n_out = slots * (size_t)b->out_stride;
    const bool with_xpad = xpad && xpad_len;
    HIPCHK(stage_reserve(b, 0, n_pcm));
    HIPCHK(stage_reserve(b, 1, n_out));
    if (with_xpad) { HIPCHK(stage_reserve(b, 2, slots * TL_MAX_XPAD)); HIPCHK(stage_reserve(b, 3, slots * sizeof(int32_t))); }
    if (taps) HIPCHK(stage_reserve(b, 4, slots * sizeof(TlTaps)));
    if (out_len) HIPCHK(stage_reserve(b, 8, slots * sizeof(int32_t)));
    int32_t *d_len = out_len ? (int32_t *)b->stage[8] : nullptr;
    int16_t *d_pcm = (int16_t *)b->stage[0]; uint8_t *d_out = (uint8_t *)b->stage[1];
    uint8_t *d_xpad = with_xpad ? (uint8_t *)b->stage[2] : nullptr; int32_t *d_xl = with_xpad ? (int32_t *)b->stage[3] : nullptr;
    TlTaps *d_taps = taps ? (TlTaps *)b->stage[4] : nullptr;
    // Big calls go through in up to four chunks of whole frames on three streams: while the kernels of chunk c run, chunk c+1
    // comes in over PCIe and chunk c-1 goes out (the link is full duplex; with pinned host buffers, tlb_host_alloc, the
    // copies run at link rate).  The kernels themselves stay in frame order on one stream -- the streams' state passes from
    // chunk to chunk.  Small calls (the legacy shim: one frame) and tap runs are one chunk.
    const int want = (taps || n_pcm < (8u << 20) || nframes < 2) ? 1 : (nframes < TLB_HOST_CHUNKS ? nframes : TLB_HOST_CHUNKS);
    const int per = (nframes + want - 1) / want;                   // frames per chunk
    const int nchunks = (nframes + per - 1) / per;                 // (5 frames: 2 + 2 + 1, three chunks, not four)
    if (!b->s_in) {
        HIPCHK(hipStreamCreateWithFlags(&b->s_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b->s_run, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b->s_out, hipStreamNonBlocking));
        for (int i = 0; i < TLB_HOST_CHUNKS; i++) { HIPCHK(hipEventCreateWithFlags(&b->ev_in[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b->ev_run[i], hipEventDisableTiming)); }
    }
    // TlPsyOut / ScF-CRC scratch sized for the largest chunk up front (tlb_launch would otherwise re-allocate between chunks)
    {
        if (b->n_list[2]) HIPCHK(stage_reserve(b, 5, (size_t)per * (size_t)b->nstreams * sizeof(TlPsyOut)));
        HIPCHK(stage_reserve(b, 6, (size_t)per * (size_t)b->nstreams * 4));
    }
    HIPCHK(hipMemsetAsync(d_out, 0, n_out, b->s_in));              // bytes the kernels do not write (slot 0 of the first call, tails of short frames) read as 0
    if (taps) HIPCHK(hipMemsetAsync(d_taps, 0, slots * sizeof(TlTaps), b->s_in));
    // From here on copies and kernels are in flight on three streams and touch the caller's buffers: every error path drains
    // them before it returns (a caller that frees or reuses pcm / out on error must not race with a DMA transfer).
#define HIPCHK_DRAIN(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "libtoolame-dab-hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    (void)hipStreamSynchronize(b->s_in); (void)hipStreamSynchronize(b->s_run); (void)hipStreamSynchronize(b->s_out); \
    return TLB_ERR_HIP; } } while (0)
    for (int c = 0, f0 = 0; c < nchunks; c++, f0 += per) {
        const int nf = f0 + per <= nframes ? per : nframes - f0;
        const size_t o = (size_t)f0 * (size_t)b->nstreams, n = (size_t)nf * (size_t)b->nstreams;
        HIPCHK_DRAIN(hipMemcpyAsync(d_pcm + o * 2304, pcm + o * 2304, n * 2304 * sizeof(int16_t), hipMemcpyHostToDevice, b->s_in));
        if (with_xpad) {
            HIPCHK_DRAIN(hipMemcpyAsync(d_xpad + o * TL_MAX_XPAD, xpad + o * TL_MAX_XPAD, n * TL_MAX_XPAD, hipMemcpyHostToDevice, b->s_in));
            HIPCHK_DRAIN(hipMemcpyAsync(d_xl + o, xpad_len + o, n * sizeof(int32_t), hipMemcpyHostToDevice, b->s_in));
        }
        HIPCHK_DRAIN(hipEventRecord(b->ev_in[c], b->s_in));
        HIPCHK_DRAIN(hipStreamWaitEvent(b->s_run, b->ev_in[c], 0));
        int rc = tlb_launch(b, d_pcm + o * 2304, nf, with_xpad ? d_xpad + o * TL_MAX_XPAD : nullptr, with_xpad ? d_xl + o : nullptr,
                            d_out + o * (size_t)b->out_stride, d_taps ? d_taps + o : nullptr, b->s_run, nullptr, d_len ? d_len + o : nullptr);
        if (rc != TLB_OK) { (void)hipStreamSynchronize(b->s_in); (void)hipStreamSynchronize(b->s_run); (void)hipStreamSynchronize(b->s_out); return rc; }
        HIPCHK_DRAIN(hipEventRecord(b->ev_run[c], b->s_run));
        HIPCHK_DRAIN(hipStreamWaitEvent(b->s_out, b->ev_run[c], 0));
        HIPCHK_DRAIN(hipMemcpyAsync(out + o * (size_t)b->out_stride, d_out + o * (size_t)b->out_stride, n * (size_t)b->out_stride, hipMemcpyDeviceToHost, b->s_out));
        if (taps) HIPCHK_DRAIN(hipMemcpyAsync((TlTaps *)taps + o, d_taps + o, n * sizeof(TlTaps), hipMemcpyDeviceToHost, b->s_out));
        if (out_len) HIPCHK_DRAIN(hipMemcpyAsync(out_len + o, d_len + o, n * sizeof(int32_t), hipMemcpyDeviceToHost, b->s_out));
    }
    HIPCHK_DRAIN(hipStreamSynchronize(b->s_out));
    HIPCHK(hipStreamSynchronize(b->s_run));
#undef HIPCHK_DRAIN
    return TLB_OK;
}

// Pinned host memory for callers of the host-buffer entry points (hipHostMalloc): PCIe copies from it run at link rate.
void *tlb_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void tlb_host_free(void *p) { if (p) (void)hipHostFree(p); }

// Diagnostic: per-stage s_memtime stamps [nframes][nstreams][32] (see TL_STAMP in mp2_wave.h).
int tlb_encode_host_stamps(tlb_batch *b, const int16_t *pcm, int nframes, long long *stamps)
{
    DevFree guard_;
    if (!b || !pcm || !stamps || nframes <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams;
    int16_t *d_pcm = nullptr; uint8_t *d_out = nullptr; long long *d_st = nullptr;
    DEVALLOC(d_pcm, slots * 2304 * sizeof(int16_t));
    DEVALLOC(d_out, slots * (size_t)b->out_stride);
    DEVALLOC(d_st, slots * 32 * sizeof(long long));
    HIPCHK(hipMemset(d_st, 0, slots * 32 * sizeof(long long)));
    HIPCHK(hipMemcpy(d_pcm, pcm, slots * 2304 * sizeof(int16_t), hipMemcpyHostToDevice));
    int rc = tlb_launch(b, d_pcm, nframes, nullptr, nullptr, d_out, nullptr, nullptr, d_st);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(stamps, d_st, slots * 32 * sizeof(long long), hipMemcpyDeviceToHost));

    return rc;
}

int tlb_set_gain_db(tlb_batch *b, int stream, double gain_db)
{   // const double linear_gain_correction = pow(10.0, gain_dB / 20.0);  (src/odr-audioenc.cpp:1032)
    if (!b || stream < -1 || stream >= b->nstreams) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    const double g = pow(10.0, gain_db / 20.0);
    for (int s2 = 0; s2 < b->nstreams; s2++) if (stream < 0 || s2 == stream) b->h_gain[(size_t)s2] = g;
    HIPCHK(hipMemcpy(b->d_gain, b->h_gain.data(), sizeof(double) * (size_t)b->nstreams, hipMemcpyHostToDevice));
    return TLB_OK;
}

int tlb_ingest_device(tlb_batch *b, const int16_t *d_interleaved, int nframes, int16_t *d_pcm, int16_t *d_peaks, void *hip_stream)
{
    if (!b || !d_interleaved || !d_pcm || !d_peaks || nframes <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(tlk_ingest((unsigned)((size_t)nframes * (size_t)b->nstreams), (hipStream_t)hip_stream, d_interleaved, d_pcm, d_peaks, b->d_gain, b->d_configs, b->d_stream_cfg, b->nstreams));
    return TLB_OK;
}

int tlb_ingest_host(tlb_batch *b, const int16_t *interleaved, int nframes, int16_t *pcm, int16_t *peaks)
{   // device staging kept between calls, like tlb_encode_host (an application calls this once per chunk of frames)
    if (!b || !interleaved || !pcm || !peaks || nframes <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams;
    HIPCHK(stage_reserve(b, 9, slots * 2304 * 2));
    HIPCHK(stage_reserve(b, 10, slots * 2304 * 2));
    HIPCHK(stage_reserve(b, 11, slots * 2 * 2));
    int16_t *d_in = (int16_t *)b->stage[9], *d_out = (int16_t *)b->stage[10], *d_pk = (int16_t *)b->stage[11];
    HIPCHK(hipMemcpy(d_in, interleaved, slots * 2304 * 2, hipMemcpyHostToDevice));
    int rc = tlb_ingest_device(b, d_in, nframes, d_out, d_pk, nullptr);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(pcm, d_out, slots * 2304 * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(peaks, d_pk, slots * 2 * 2, hipMemcpyDeviceToHost));
    return rc;
}

int tlb_silence_device(tlb_batch *b, const int16_t *d_peaks, int nframes, uint32_t *d_silence_ms, void *hip_stream)
{
    if (!b || !d_peaks || !d_silence_ms || nframes <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(tlk_silence((unsigned)((b->nstreams + 255) / 256), (hipStream_t)hip_stream, d_peaks, d_silence_ms, b->d_configs, b->d_stream_cfg, b->nstreams, nframes));
    return TLB_OK;
}

int tlb_silence_host(tlb_batch *b, const int16_t *peaks, int nframes, uint32_t *silence_ms)
{
    DevFree guard_;
    if (!b || !peaks || !silence_ms || nframes <= 0) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    const size_t slots = (size_t)nframes * (size_t)b->nstreams;
    int16_t *d_p = nullptr; uint32_t *d_m = nullptr;
    DEVALLOC(d_p, slots * 4);
    DEVALLOC(d_m, sizeof(uint32_t) * (size_t)b->nstreams);
    HIPCHK(hipMemcpy(d_p, peaks, slots * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_m, silence_ms, sizeof(uint32_t) * (size_t)b->nstreams, hipMemcpyHostToDevice));
    int rc = tlb_silence_device(b, d_p, nframes, d_m, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(silence_ms, d_m, sizeof(uint32_t) * (size_t)b->nstreams, hipMemcpyDeviceToHost);

    if (e != hipSuccess) return TLB_ERR_HIP;
    return rc;
}

int tlb_zmq_msg_stride(const tlb_batch *b) { return b ? 12 + b->out_stride : 0; }
int tlb_egress_unit_bytes(const tlb_batch *b, int s) { return (b && s >= 0 && s < b->nstreams) ? 3 * b->h_configs[b->h_stream_cfg[s]].kbps : 0; }
int tlb_egress_units_per_frame(const tlb_batch *b, int s)
{
    if (!b || s < 0 || s >= b->nstreams) return 0;
    const TlConfig &c = b->h_configs[b->h_stream_cfg[s]];
    return c.frame_bytes % (3 * c.kbps) ? 0 : c.frame_bytes / (3 * c.kbps);
}
int tlb_egress_max_units_per_frame(const tlb_batch *b) { return b ? b->max_upf : 0; }


int tlb_flush_device_len(tlb_batch *b, uint8_t *d_out, int32_t *d_out_len, void *hip_stream)
{
    if (!b || !d_out) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(tlk_flush((unsigned)b->nstreams, (hipStream_t)hip_stream, b->d_state, b->d_configs, b->d_stream_cfg, d_out, d_out_len, b->nstreams, b->out_stride));
    return TLB_OK;
}
int tlb_flush_device(tlb_batch *b, uint8_t *d_out, void *hip_stream) { return tlb_flush_device_len(b, d_out, nullptr, hip_stream); }

int tlb_flush_host_len(tlb_batch *b, uint8_t *out, int32_t *out_len)
{
    DevFree guard_;
    if (!b || !out) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    uint8_t *d = nullptr; int32_t *dl = nullptr;
    const size_t n = (size_t)b->nstreams * (size_t)b->out_stride;
    DEVALLOC(d, n);
    DEVALLOC(dl, sizeof(int32_t) * (size_t)b->nstreams);
    int rc = tlb_flush_device_len(b, d, dl, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_len) e = hipMemcpy(out_len, dl, sizeof(int32_t) * (size_t)b->nstreams, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return TLB_ERR_HIP;
    return rc;
}
int tlb_flush_host(tlb_batch *b, uint8_t *out) { return tlb_flush_host_len(b, out, nullptr); }

// The reference's send schedule for one stream (host arithmetic, no GPU): toolame_encode_frame() hands bytes back only when its
// 4096-byte bit buffer fills (bitstream.c:46-71), and odr-audioenc sends `while (toolame_buffer.size() > 3 * bitrate)`
// (src/odr-audioenc.cpp:1208-1225) -- so units leave in bursts of about ten during ONE call, all with that call's peak levels, and one
// unit always stays behind.  units_sent[i] = units the reference sends during call i (input frame i).
int tlb_reference_send_schedule(const tlb_stream_config *cfg, int ncalls, int32_t *units_sent)
{
    if (!cfg || ncalls < 0 || (ncalls && !units_sent)) return -TLB_ERR_ARG;
    TlConfig c;
    if (int rc = tl_build_config(&c, cfg->samplerate, cfg->mode, cfg->bitrate, cfg->psy_model, cfg->pad_len)) return -rc;
    const int unit = 3 * c.kbps, buf = 4096;
    double lag = 0;
    int fill = 0, minimum = 4, held = 0;
    for (int i = 0; i < ncalls; i++) {
        int cur = c.frame_bytes;                                     // availbits.c:49-62
        if (c.pad_frac != 0) { if (lag > (c.pad_frac - 1.0)) lag -= c.pad_frac; else { cur++; lag += (1 - c.pad_frac); } }
        if (i == 0) minimum = cur + 4;                               // toolame.c:298-300
        int written = 0;
        if (fill + cur >= buf) { written = buf - minimum; fill = minimum + (fill + cur - buf); }
        else fill += cur;
        held += written;
        int n = 0;
        while (held > unit) { held -= unit; n++; }                   // strictly greater: one unit is held back
        units_sent[i] = n;
    }
    return held;
}

// Is this host's libm the one csrc/tl_libm.h restates?  The reference's bytes depend on what the HOST libm returns for log10 / pow /
// log / exp / sincos / atan2 (glibc 2.35 on an FMA-capable x86-64: the ifunc variants __log_fma, __exp_fma, __pow_fma, __atan2_fma);
// the device computes those routines itself, so on a host with another libm the reference build and this library may part on
// degenerate signals.  Compares the restated routines (their host forms, the very text the kernels compile) with libm on
// `nsamples` arguments per function drawn from the encoder's ranges; returns how many results differ (0: this is that libm).
long tlb_selfcheck_libm(long nsamples)
{
    if (nsamples <= 0) nsamples = 100000;
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st * 0x2545f4914f6cdd1dull; };
    auto unit = [&]() { return (double)(next() >> 11) * 0x1p-53; };
    auto same = [](double a, double b) { uint64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return (a != a && b != b) || x == y; };
    long bad = 0;
    for (long i = 0; i < nsamples; i++) {
        const double e = ldexp(1.0 + unit(), (int)(next() % 90) - 70);            // energies 1e-21 .. 1e6
        const double y = -30.0 + 60.0 * unit(), ph = -8.0 + 16.0 * unit(), ax = ldexp(unit() - 0.5, (int)(next() % 40) - 20), ay = ldexp(unit() - 0.5, (int)(next() % 40) - 20);
        bad += !same(tlm_log10(e), log10(e)) + !same(tlm_log(e), log(e)) + !same(tlm_exp(y), exp(y)) + !same(tlm_pow10(y), pow(10.0, y)) + !same(tlm_atan2(ay, ax), atan2(ay, ax));
        double s1, c1, s2, c2;
        tlm_sincos(ph, &s1, &c1); sincos(ph, &s2, &c2);
        bad += !same(s1, s2) + !same(c1, c2);
    }
    return bad;
}

float tlb_last_kernel_ms(tlb_batch *b)
{
    if (!b || !b->timed) return -1.0f;
    if (hipSetDevice(b->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(b->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.0f;
    return ms;
}

// Durations of the two kernels of the most recent launch of a batch whose streams ALL use psy model 2 or 4 (tl_psy2_kernel,
// then tl_main_kernel<2> + tl_finish_kernel), hipEvents on the launch stream.  Models 1 / 3 run one kernel per launch and
// model 0 has no psy kernel: for those, and for mixed batches, the call returns non-zero.
int tlb_last_stage_ms(tlb_batch *b, float *psy_ms, float *encode_ms)
{
    if (!b || !b->timed || !b->have_mid || !psy_ms || !encode_ms) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipEventSynchronize(b->ev1));
    HIPCHK(hipEventElapsedTime(psy_ms, b->ev0, b->ev_mid));
    HIPCHK(hipEventElapsedTime(encode_ms, b->ev_mid, b->ev1));
    return TLB_OK;
}


}  // extern "C"
