// tl_kernels.h -- the one door between the host translation units (tlb_batch.cpp, tlb_egress.cpp, tlb_tick.cpp: plain C++, seconds to
// compile) and the kernels (toolame_hip.hip, toolame_psy2.hip: the only files that see mp2_wave.h).  Each launcher queues ONE kernel on
// `st` and returns hipGetLastError(); grid shapes that depend on the kernels' wave counts are computed from the constants below.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "mp2_types.h"
#include "edi_types.h"

#define TL_HEAD_STRIDE 32             // int32 per list head of the persistent kernels' work lists: one 128-byte line each (9 heads)
#ifndef TL_MAIN_WPE
#define TL_MAIN_WPE 3                 // waves per SIMD of the encode kernels
#endif
#define TL_MAIN_WAVES (4 * TL_MAIN_WPE)       // one workgroup per CU: one copy of the tables
#ifndef TL_PSY2_WAVES
#define TL_PSY2_WAVES 12
#endif

hipError_t tlk_slots(unsigned blocks, hipStream_t st, const TlLaunch &A);                         // tl_slots_kernel, 256 threads
hipError_t tlk_frame(int psy, bool pairs, bool stereo, unsigned blocks, hipStream_t st, const TlLaunch &A);    // tl_frame_kernel<1|3, pairs, stereo ? 2 : 0>
hipError_t tlk_main(int psy, bool pairs, bool stereo, unsigned blocks, hipStream_t st, const TlLaunch &A);     // tl_main_kernel<0|2, pairs, stereo ? 2 : 0>
hipError_t tlk_psy2(unsigned blocks, hipStream_t st, const TlLaunch &A);                          // tl_psy2_kernel
hipError_t tlk_finish(unsigned blocks, hipStream_t st, const TlLaunch &A);                        // tl_finish_kernel, 256 threads = 4 streams
hipError_t tlk_ingest(unsigned blocks, hipStream_t st, const int16_t *in, int16_t *out, int16_t *peaks, const double *gain,
                      const TlConfig *configs, const int32_t *stream_cfg, int nstreams);
hipError_t tlk_silence(unsigned blocks, hipStream_t st, const int16_t *peaks, uint32_t *silence_ms, const TlConfig *configs,
                       const int32_t *stream_cfg, int nstreams, int nframes);
hipError_t tlk_zmq_frame(unsigned blocks, hipStream_t st, const uint8_t *frames, const int16_t *peaks, uint8_t *msgs, const TlConfig *configs,
                         const int32_t *stream_cfg, int nstreams, int out_stride, int msg_stride, int max_upf, const int32_t *frame_len);
hipError_t tlk_edi_af(unsigned bx, unsigned by, hipStream_t st, const TlEdiArgs &A);
hipError_t tlk_edi_pft(unsigned bx, unsigned by, hipStream_t st, const TlPftArgs &A, const TlTables *T);
hipError_t tlk_flush(unsigned blocks, hipStream_t st, const TlStreamState *state, const TlConfig *configs, const int32_t *stream_cfg,
                     uint8_t *out, int32_t *out_len, int nstreams, int out_stride);
size_t tlk_lds_bytes_per_wave(void);          // the largest per-wave LDS block among the kernels
