// tlb_internal.h -- what the host translation units of the library share: the batch object, the error macros, and the few internal
// entry points that cross file boundaries (hidden from the dynamic symbol table by exports.map).  Host C++ only: no kernel code.
#pragma once
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <vector>

#include "../../include/toolame_batch.h"
#include "mp2_host.h"
#include "tl_kernels.h"

static_assert(TL_MAX_XPAD == TLB_MAX_XPAD, "xpad record size");
#define TLB_HOST_CHUNKS 4            // tlb_encode_host pipelines a big call in this many chunks of frames

struct tlb_batch {
    int device = 0, nstreams = 0, out_stride = 0;
    long frames = 0;
    std::vector<TlConfig> h_configs;
    std::vector<tlb_stream_config> h_uniq;       // the six knobs of h_configs[i]
    size_t cfg_cap = 0;                          // records d_configs has room for
    std::vector<int32_t> h_stream_cfg;
    TlTables *d_tables = nullptr;
    TlConfig *d_configs = nullptr;
    int32_t *d_stream_cfg = nullptr;
    TlStreamState *d_state = nullptr;
    double *d_gain = nullptr;                    // linear gain per stream (ingest kernel)
    std::vector<double> h_gain;
    int32_t *d_list[4] = {nullptr, nullptr, nullptr, nullptr};   // stream ids per psy model
    int n_list[4] = {0, 0, 0, 0};
    TlPsy2Tables *d_psy2_tables = nullptr;     // 2 * TL_PSY2_SLOTS tables (psy 2 per sample rate, then psy 4 per sample rate; tl_psy2_slot), only when a stream uses psy 2 / 4
    TlPsy2State *d_psy2_state = nullptr;       // two copies per stream; a launch reads copy psy2_flip and writes the other (tl_psy2_chain)
    int psy2_flip = 0;
    int32_t *d_partner = nullptr;              // [nstreams] mono streams of one configuration and model share waves in pairs (tl_encode_pair); -1: alone
    int32_t *d_chain = nullptr;                // psy-2 kernel: (stream, channel) chains of the launch, first channels first
    int n_chain = 0;
    uint8_t *d_edi_version = nullptr;            // EDI: ODRv string and per-stream frame sizes (allocated on first use)
    char h_edi_version[TL_EDI_MAX_VERSION] = {}; // the string d_edi_version holds
    int edi_version_len = -1;
    int32_t *d_frame_bytes = nullptr, *d_unit_bytes = nullptr;
    int max_upf = 1;                             // egress units (3 * kbps bytes) per frame: 1 at 48 kHz, 2 at 24 kHz, 3 at 16 kHz; 0 = a stream's frames are no whole number of units
    TlEdiState *d_edi_state_tmp = nullptr;
    uint16_t *d_pseq_tmp = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;   // ev_mid: between the psy-2 kernel and the encode kernel (models 2/4)
    bool have_mid = false;
    hipStream_t last_stream = nullptr;
    bool timed = false;
    // device staging of the host-buffer entry point (tlb_encode_host): grow-only, created on first use, so a caller that
    // feeds one frame per call (the legacy shim) pays for no allocation after its first frame
    void *stage[12] = {};                        // pcm, out, xpad, xpad_len, taps; [5] = TlPsyOut records (models 2/4), [6] = ScF-CRC bytes, [7] = padding bits, [8] = frame lengths (host entry); [9..11] = tlb_ingest_host: interleaved in, planar out, peaks
    size_t stage_cap[12] = {};
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;   // host-buffer entry point: copy-in / kernels / copy-out
    hipEvent_t ev_in[TLB_HOST_CHUNKS] = {}, ev_run[TLB_HOST_CHUNKS] = {};
    uint32_t *d_newpend = nullptr;               // split path: the launch's last frame of every stream
    double *d_newlag = nullptr;                  // split path, 44.1 / 22.05 kHz: slot recurrence state after the launch
    bool pads[4] = {false, false, false, false}; // some stream of the psy model's list has frames of two lengths
    bool list_pairs[4] = {false, false, false, false};   // the model's list contains mono streams paired in one wave (kernel variant <.., true>)
    bool list_stereo[4] = {false, false, false, false};  // every stream of the model's list has two channels (kernel variant <.., false, 2>: models 1 and 3)
    int32_t *d_work = nullptr;                   // unit counters of the persistent kernels
    bool work_clean = false;                     // ... are zero (tl_finish_kernel zeroes them after use)
    bool broken = false;                         // a launch or a reconfiguration failed half way: stream state, psy-2 state copies and lists may disagree;
                                                 // every further launch is refused (TLB_ERR_HIP) until tlb_reset() has put all streams back to zero
    int num_cu = 256;
    int fail_in = 0;                             // test builds only (-DTLB_FAULT_INJECT, csrc/tlb_debug.h): the fail_in-th launch from now fails
};

static inline hipError_t stage_reserve(tlb_batch *b, int k, size_t bytes)
{
    if (b->stage_cap[k] >= bytes) return hipSuccess;
    if (b->stage[k]) { (void)hipFree(b->stage[k]); b->stage[k] = nullptr; b->stage_cap[k] = 0; }
    hipError_t e = hipMalloc(&b->stage[k], bytes);
    if (e == hipSuccess) b->stage_cap[k] = bytes;
    return e;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "libtoolame-dab-hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    return TLB_ERR_HIP; } } while (0)

// device scratch of the *_host convenience entry points: released on every exit path
struct DevFree { std::vector<void *> v; ~DevFree() { for (void *p : v) (void)hipFree(p); } };
#define DEVALLOC(ptr, bytes) do { HIPCHK(hipMalloc(&(ptr), (bytes))); guard_.v.push_back((void *)(ptr)); } while (0)

// tlb_batch.cpp: one launch of the encode path on `st` (every entry point ends here)
int tlb_launch(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad, const int32_t *d_xpad_len,
               uint8_t *d_out, TlTaps *d_taps, hipStream_t st, long long *d_stamps = nullptr, int32_t *d_out_len = nullptr);
// tlb_egress.cpp: the egress stages with the per-slot frame lengths a tick object has (0 = the slot holds no frame)
int zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream, const int32_t *d_frame_len);
int edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
                  const char *version, int version_len, uint8_t *d_pkts, int32_t *d_pkt_len, void *hip_stream, const int32_t *d_frame_len);
int pft_shape(int max_af_len, int fec, int chunk_len, int transport, int *max_frags, int *frag_stride);
