// toolame_hip.hip -- gfx950 kernels + the C-ABI of include/toolame_batch.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
// -ffp-contract=off is load-bearing: the reference is built -std=c99 (no FMA contraction,
// Makefile.am:68) and every MAC chain / the quantiser d*a+b must round twice.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include <deque>
#include <vector>

#include "../../include/toolame_batch.h"
#include "mp2_host.h"
#include "mp2_wave.h"
#include "edi_af.h"
#include "edi_pft.h"
#include "tl_kernel_util.h"

static_assert(TL_MAX_XPAD == TLB_MAX_XPAD, "xpad record size");

#define TLB_HOST_CHUNKS 4            // tlb_encode_host pipelines a big call in this many chunks of frames

// ---- kernels of the encode path ------------------------------------------------------------------------------------------
// One wavefront per unit of work; every kernel is persistent: 12 waves per CU (3 per SIMD) take units off a device-wide counter.
//
//  * psy models 1 and 3 (the ones DAB services use): unit = (stream, frame).  tl_frame_kernel<PSY>: the wave runs the
//    psychoacoustic model (it reads nothing but PCM), then the encoder (filterbank, scalefactors, SMR, bit allocation,
//    quantiser, packing, CRCs) of the same frame.  The filterbank's history is PCM, so frames are independent; the one
//    thing a frame owes its predecessor (its ScF-CRC travels in the frame before) is filed aside.
//  * psy model 0: tl_main_kernel<0>, the encoder alone (the model is three lines on the scalefactors).
//  * psy models 2 and 4 carry prediction state from pass to pass (psycho_2.c:300-306), per channel: tl_psy2_kernel takes
//    one CHANNEL of one stream through the frames of the launch and leaves the SMR in TlPsyOut (HBM); then tl_main_kernel<2>.
//  * tl_finish_kernel, a wave per stream: the pending frame of the last launch to slot 0, ScF-CRCs into place, state.
//
// History of the shape (DESIGN.md section 4 has the counters): round 1 ran the model in the MIDDLE of the encoder (256 VGPRs,
// 2 waves per SIMD); round 2 first split model and encoder into two kernels (3 waves per SIMD each, but the PCM read twice and
// a 1 KB record per frame through HBM), then put them back into one kernel one AFTER the other: the registers needed are the
// maximum of the two phases, not the sum, the LDS blocks a union, the record four values per lane.
// (stream, frame) units come off EIGHT lists, one per XCD (each XCD has its own L2).  The units of a launch, numbered stream by
// stream with frames ascending (u = k * nframes + f), are dealt to the lists in blocks of 32 consecutive units: the waves of
// an XCD work on consecutive frames of a few streams at a time, so the 480 samples of history a frame needs -- the tail of
// the frame before it, which a neighbouring wave is reading as its own PCM -- come out of that XCD's L2 and not over the
// fabric a second time (one stream with many frames spreads over all eight lists just the same).  A wave whose own list
// is empty goes on with the next XCD's (hop): the lists are for locality only, the balance stays that of one queue.
// Returns false when all eight lists are empty.  Which waves share an XCD: workgroups are dealt round-robin over the XCDs
// (observed, MI355X_MICROARCH.md "Workgroup dispatch"), so blockIdx % 8 is the group; a wrong guess costs locality, nothing
// else.  A wave's FIRST unit is its rank within its group (no atomic: three thousand waves asking at once would queue);
// the list heads therefore count from the number of waves of the group.
#define TL_HEAD_STRIDE 32             // int32 per list head: one 128-byte line each
#define TL_LIST_BLOCK_LOG2 5          // 32 consecutive units per block
static __device__ __forceinline__ bool tl_take_unit(int32_t *heads, int nlist, int nframes, int grp, int &hop, int &k, int &f, int first)
{
    const int nunits = nlist * nframes, nblocks = (nunits + (1 << TL_LIST_BLOCK_LOG2) - 1) >> TL_LIST_BLOCK_LOG2;
    while (hop < 8) {
        const int q = (grp + hop) & 7;
        const int lim = ((nblocks - q + 7) >> 3) << TL_LIST_BLOCK_LOG2;              // positions on list q (the launch's last block may be short)
        const int ng = (((int)gridDim.x - q + 7) >> 3) * (int)(blockDim.x >> 6);      // waves whose own list q is: they took positions 0 .. ng-1 by rank
        int v;
        if (first >= 0) { v = first; first = -1; }
        else {
            // Another group's list is looked at before it is drawn from: at the end of a launch every wave walks the other
            // seven lists, and three thousand returning atomics on a word that has nothing left to give queue for 35 us per
            // list; a load does not queue.  A stale value costs one atomic, nothing else.
            if (hop > 0 && ng + __hip_atomic_load(&heads[q * TL_HEAD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= lim) { hop++; continue; }
            v = ng + tl_next_unit(&heads[q * TL_HEAD_STRIDE]);
        }
        if (v >= lim) { hop++; continue; }
        const int u = ((((v >> TL_LIST_BLOCK_LOG2) << 3) + q) << TL_LIST_BLOCK_LOG2) + (v & ((1 << TL_LIST_BLOCK_LOG2) - 1));
        if (u >= nunits) continue;                                   // past the end of the launch's last block: take again
        k = u / nframes; f = u - k * nframes;
        return true;
    }
    return false;
}

// LDS is handed out in granules of 1280 bytes on gfx950 (160 KB / 128): the twelve waves a CU holds at 3 per SIMD are ONE
// workgroup sharing one copy of the tables (three 4-wave workgroups with a copy each do not fit).
#ifndef TL_MAIN_WPE
#define TL_MAIN_WPE 3
#endif
// psy kernel of models 2 and 4: csrc/toolame_psy2.hip, a translation unit of its own (it wants the IR load / store vectorizer the
// other kernels are built without, see csrc/Makefile).  TL_PSY2_WAVES waves per workgroup.
__global__ void tl_psy2_kernel(TlLaunch A);
// encode kernel of the split path: the tables it needs on dependent-load chains (TlBlockShared without the dB-sum table)
struct TlMainShared { double enw_s[512]; char bytes[sizeof(TlBlockShared) - offsetof(TlBlockShared, scalefactor)]; TlPackTables pack; };
static_assert(offsetof(TlBlockShared, dbtable) == 0 && offsetof(TlBlockShared, scalefactor) == sizeof(double) * 1002, "dbtable leads TlBlockShared");
#define TL_MAIN_WAVES (4 * TL_MAIN_WPE)       // one workgroup per CU: one copy of the tables
static_assert((sizeof(TlMainShared) + TL_MAIN_WAVES * sizeof(TlMainLds) + TL_LDS_GRANULE - 1) / TL_LDS_GRANULE <= 128, "the encode workgroup fits a CU");
// PAIRS: the list contains mono streams that share waves in pairs (TlLaunch::partner, tl_encode_pair).  A second instantiation, so
// that the kernel of lists without pairs -- every all-stereo batch -- carries none of the pair code (with it inline the register
// allocation of the stereo path moved from 152 to 168 VGPRs and psy-1 stereo lost 1.1 %).
template <int PSY, bool PAIRS>     // 0: model 0 (no psy kernel); 2: models 2 and 4 (after tl_psy2_kernel)
__global__ void __launch_bounds__(64 * TL_MAIN_WAVES) __attribute__((amdgpu_waves_per_eu(TL_MAIN_WPE, TL_MAIN_WPE))) tl_main_kernel(TlLaunch A)
{
    __shared__ TlMainShared sh;
    __shared__ TlMainLds lds[TL_MAIN_WAVES];
    {
        const double *src = (const double *)&A.tables->shared.scalefactor[0];
        double *dst = (double *)&sh.bytes[0];
        for (int i = (int)threadIdx.x; i < (int)(sizeof(sh.bytes) / 8); i += 64 * TL_MAIN_WAVES) dst[i] = src[i];
        for (int i = (int)threadIdx.x; i < 512; i += 64 * TL_MAIN_WAVES) sh.enw_s[i] = A.tables->enwindow_s[i];
        static_assert(sizeof(TlPackTables) % 8 == 0, "copied as doubles");
        for (int i = (int)threadIdx.x; i < (int)(sizeof(TlPackTables) / 8); i += 64 * TL_MAIN_WAVES) ((double *)&sh.pack)[i] = ((const double *)&A.tables->pack)[i];
    }
    __syncthreads();
    // the encode path never touches B->dbtable: a TlBlockShared pointer whose dbtable part lies before the copied block
    const TlBlockShared *B = (const TlBlockShared *)((const char *)&sh.bytes[0] - offsetof(TlBlockShared, scalefactor));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = (int)blockIdx.x & 7;
    int wave_v = (int)(threadIdx.x >> 6);
    asm volatile("" : "+v"(wave_v));
    TlMainLds &wl = lds[wave_v];
    for (int hop = 0, k, f, first = ((int)blockIdx.x >> 3) * TL_MAIN_WAVES + wave; tl_take_unit(A.work + TL_HEAD_STRIDE, A.nlist, A.nframes, grp, hop, k, f, first); first = -1) {
        const int s = __builtin_amdgcn_readfirstlane(A.stream_list[k]);
        if constexpr (PAIRS) {
            int s2;
            if (!tl_unit_partner(A, s, s2)) continue;                // the partner's wave encodes this mono stream's frame with its own
            s2 = __builtin_amdgcn_readfirstlane(s2);
            if (s2 >= 0) { tl_main_pair<PSY>(wl, B, sh.enw_s, &sh.pack, A, s, s2, f); continue; }
        }
        tl_main_unit<PSY>(wl, B, sh.enw_s, &sh.pack, A, s, f);
    }
}

// Models 1 and 3: psy model and encoder of a (stream, frame) unit by the same wave, one after the other (tl_frame_unit).
// Registers: the model needs 118, the encoder 154 -- the kernel needs the larger, not the sum, because the model runs BEFORE
// the filterbank fills its 72 sample registers (round 1's fused kernel ran it in the middle: 256).  LDS: the union of the
// two phases' blocks; both table sets (dB sums 7.8 KB, encoder 12.1 KB) once per workgroup.  The PCM the model read is still
// in L2 when the encoder stages it.
static_assert((sizeof(double) * 1258 + sizeof(TlMainShared) + TL_MAIN_WAVES * sizeof(TlFrameLds) + TL_LDS_GRANULE - 1) / TL_LDS_GRANULE <= 128, "the workgroup fits a CU");
template <int PSY, bool PAIRS>
__global__ void __launch_bounds__(64 * TL_MAIN_WAVES) __attribute__((amdgpu_waves_per_eu(TL_MAIN_WPE, TL_MAIN_WPE))) tl_frame_kernel(TlLaunch A)
{
    __shared__ __attribute__((aligned(16))) double dbt[1258];     // dB-sum table + glibc's log table (TlTables::dblog)
    __shared__ TlMainShared sh;
    __shared__ TlFrameLds lds[TL_MAIN_WAVES];
    {
        for (int i = (int)threadIdx.x; i < 1258; i += 64 * TL_MAIN_WAVES) dbt[i] = A.tables->dblog[i];
        const double *src = (const double *)&A.tables->shared.scalefactor[0];
        double *dst = (double *)&sh.bytes[0];
        for (int i = (int)threadIdx.x; i < (int)(sizeof(sh.bytes) / 8); i += 64 * TL_MAIN_WAVES) dst[i] = src[i];
        for (int i = (int)threadIdx.x; i < 512; i += 64 * TL_MAIN_WAVES) sh.enw_s[i] = A.tables->enwindow_s[i];
        for (int i = (int)threadIdx.x; i < (int)(sizeof(TlPackTables) / 8); i += 64 * TL_MAIN_WAVES) ((double *)&sh.pack)[i] = ((const double *)&A.tables->pack)[i];
    }
    __syncthreads();
    const TlBlockShared *B = (const TlBlockShared *)((const char *)&sh.bytes[0] - offsetof(TlBlockShared, scalefactor));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = (int)blockIdx.x & 7;
    int wave_v = (int)(threadIdx.x >> 6);
    asm volatile("" : "+v"(wave_v));
    TlFrameLds &wl = lds[wave_v];
    for (int hop = 0, k, f, first = ((int)blockIdx.x >> 3) * TL_MAIN_WAVES + wave; tl_take_unit(A.work + TL_HEAD_STRIDE, A.nlist, A.nframes, grp, hop, k, f, first); first = -1) {
        const int s = __builtin_amdgcn_readfirstlane(A.stream_list[k]);
#if defined(__HIP_DEVICE_COMPILE__)
        // each phase reads the launch record afresh from the kernel-argument segment (scalar loads), so that nothing but s
        // and f lives in registers across the two
        TlKArg a1 = (TlKArg)__builtin_amdgcn_kernarg_segment_ptr(), a2 = a1;
        asm volatile("" : "+s"(a1));
        const TlLaunch A1 = *a1;
        if constexpr (PAIRS) {
            int s2;
            if (!tl_unit_partner(A, s, s2)) continue;                // the partner's wave encodes this mono stream's frame with its own
            tl_frame_unit<PSY>(wl, dbt, B, sh.enw_s, &sh.pack, A1, a2, s, f, __builtin_amdgcn_readfirstlane(s2));
        } else tl_frame_unit<PSY>(wl, dbt, B, sh.enw_s, &sh.pack, A1, a2, s, f);
#endif
    }
}

// slot recurrence of the split path (44.1 / 22.05 kHz streams only): one lane per stream (tl_slots_stream)
__global__ void __launch_bounds__(256) tl_slots_kernel(TlLaunch A)
{
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k >= A.nlist) return;
    tl_slots_stream(A, A.stream_list[k]);
}

// finish pass of the split path: one wave per stream (tl_finish_stream)
__global__ void __launch_bounds__(256) tl_finish_kernel(TlLaunch A)
{
    // the unit counters of the list's persistent kernels are dead now: zero them for the next launch (instead of a memset per launch)
    if (blockIdx.x == 0 && threadIdx.x < 9) A.work[threadIdx.x * TL_HEAD_STRIDE] = 0;
    const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (k >= A.nlist) return;
    tl_finish_stream(A, __builtin_amdgcn_readfirstlane(A.stream_list[k]));
}

// Ingest glue of the caller (src/odr-audioenc.cpp:1030-1051 gain + peak, :1139-1152
// de-interleave): interleaved s16le -> planar [2][1152], optional linear gain with the reference's
// double-multiply-and-truncate, positive peak per channel.  Pure streaming kernel: 16 B loads, 8 B stores.
// in  [nframes][nstreams][2304] int16 (L R L R ...; mono streams use the first 1152 values)
// out [nframes][nstreams][2][1152] int16, peaks [nframes][nstreams][2] int16
__global__ void __launch_bounds__(256) tl_ingest_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                       int16_t *__restrict__ peaks, const double *__restrict__ gain,
                                                       const TlConfig *configs, const int32_t *stream_cfg, int nstreams)
{
    const size_t slot = blockIdx.x;
    const int s = (int)(slot % (size_t)nstreams);
    const int nch = configs[stream_cfg[s]].nch;
    const double g = gain[s];
    const int16_t *src = in + slot * 2304;
    int16_t *dst = out + slot * 2304;
    int pk0 = 0, pk1 = 0;
    // the level loop of the reference always walks the buffer as L/R pairs, also in mono (odr-audioenc.cpp:1034-1051)
    const int nquads = nch == 2 ? 288 : 144;                    // 16 bytes = 4 L/R pairs per step
    for (int q = (int)threadIdx.x; q < nquads; q += 256) {
        const uint4 v = ((const uint4 *)src)[q];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        int16_t l[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int a = (int16_t)(w4[k] & 0xffff), b = (int16_t)(w4[k] >> 16);
            if (g != 1.0) { a = (int16_t)(int)((double)a * g); b = (int16_t)(int)((double)b * g); }
            l[k] = (int16_t)a; r[k] = (int16_t)b;
            pk0 = a > pk0 ? a : pk0; pk1 = b > pk1 ? b : pk1;
        }
        if (nch == 2) {
            ((uint2 *)dst)[q] = make_uint2((uint16_t)l[0] | ((uint32_t)(uint16_t)l[1] << 16), (uint16_t)l[2] | ((uint32_t)(uint16_t)l[3] << 16));
            ((uint2 *)(dst + 1152))[q] = make_uint2((uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16), (uint16_t)r[2] | ((uint32_t)(uint16_t)r[3] << 16));
        } else {                                                 // mono: consecutive samples, channel 0 only
            ((uint4 *)dst)[q] = make_uint4((uint16_t)l[0] | ((uint32_t)(uint16_t)r[0] << 16), (uint16_t)l[1] | ((uint32_t)(uint16_t)r[1] << 16),
                                           (uint16_t)l[2] | ((uint32_t)(uint16_t)r[2] << 16), (uint16_t)l[3] | ((uint32_t)(uint16_t)r[3] << 16));
        }
    }
    if (nch == 1) for (int q = (int)threadIdx.x; q < 288; q += 256) ((uint2 *)(dst + 1152))[q] = make_uint2(0u, 0u);
    __shared__ int red[2][4];
    for (int o = 32; o; o >>= 1) { int t0 = __shfl_xor(pk0, o, 64), t1 = __shfl_xor(pk1, o, 64); pk0 = t0 > pk0 ? t0 : pk0; pk1 = t1 > pk1 ? t1 : pk1; }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pk0; red[1][threadIdx.x >> 6] = pk1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        int m = red[threadIdx.x][0];
        for (int k = 1; k < 4; k++) m = red[threadIdx.x][k] > m ? red[threadIdx.x][k] : m;
        peaks[slot * 2 + threadIdx.x] = (int16_t)m;
    }
}

// Silence accounting of the caller (src/odr-audioenc.cpp:1053-1079): a frame whose peaks are both 0
// adds its duration (integer milliseconds, as the reference computes it) to the stream's counter, any other frame resets it.
// One thread per stream, frames in order.
__constant__ int32_t tl_fs_hz[2][3] = {{22050, 24000, 16000}, {44100, 48000, 32000}};     // [MPEG version][sampling_frequency index], common.c:118-144
__global__ void __launch_bounds__(256) tl_silence_kernel(const int16_t *__restrict__ peaks, uint32_t *__restrict__ silence_ms, const TlConfig *configs,
                                                          const int32_t *stream_cfg, int nstreams, int nframes)
{
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= nstreams) return;
    const TlConfig &c = configs[stream_cfg[s]];
    const unsigned long rate = (unsigned long)tl_fs_hz[c.version][c.fs_idx], nch = (unsigned long)c.nch;
    const uint32_t frame_ms = (uint32_t)(1000ul * (1152ul * 2ul * nch) / (2ul * nch * rate));
    uint32_t ms = silence_ms[s];
    for (int f = 0; f < nframes; f++) {
        const size_t slot = (size_t)f * (size_t)nstreams + (size_t)s;
        const int pl = peaks[slot * 2], pr = peaks[slot * 2 + 1];
        ms = (pl > pr ? pl : pr) == 0 ? ms + frame_ms : 0u;
    }
    silence_ms[s] = ms;
}

// ZeroMQ wire format of the step after the path (src/Outputs.h:76-99, Outputs.cpp:101-138):
// packed header {u16 version=1, u16 encoder=2 (MPEG L2), u32 datasize, i16 level_left, i16 level_right} + the unit's bytes.
// One message per UNIT of 3 * bitrate bytes (src/odr-audioenc.cpp:1211-1219): message slot v = f * max_upf + u carries bytes
// [u * unit, (u + 1) * unit) of frame f; a stream with fewer units per frame leaves its surplus slots empty (datasize 0 in an
// all-zero header).  msgs [nframes * max_upf][nstreams][msg_stride]; one block per slot.
__global__ void tl_zmq_frame_kernel(const uint8_t *__restrict__ frames, const int16_t *__restrict__ peaks, uint8_t *__restrict__ msgs,
                                    const TlConfig *configs, const int32_t *stream_cfg, int nstreams, int out_stride, int msg_stride, int max_upf,
                                    const int32_t *__restrict__ frame_len)
{
    const size_t pslot = blockIdx.x;
    const int s = (int)(pslot % (size_t)nstreams), v = (int)(pslot / (size_t)nstreams), f = v / max_upf, u = v - f * max_upf;
    const size_t slot = (size_t)f * (size_t)nstreams + (size_t)s;
    const TlConfig &c = configs[stream_cfg[s]];
    const int n = 3 * c.kbps, upf = c.frame_bytes / n;
    uint8_t *m = msgs + pslot * (size_t)msg_stride;
    if (u >= upf || (frame_len && frame_len[slot] == 0)) { if (threadIdx.x < 3) ((uint32_t *)m)[threadIdx.x] = 0u; return; }   // no unit here / no frame in this slot
    if (threadIdx.x < 3) {
        const int pl = peaks ? peaks[slot * 2] : 0, pr = peaks ? peaks[slot * 2 + 1] : 0;
        const uint32_t w = threadIdx.x == 0 ? (1u | (2u << 16)) : threadIdx.x == 1 ? (uint32_t)n
                                            : ((uint32_t)(uint16_t)pl | ((uint32_t)(uint16_t)pr << 16));
        ((uint32_t *)m)[threadIdx.x] = w;
    }
    const uint32_t *src = (const uint32_t *)(frames + slot * (size_t)out_stride + (size_t)u * n);      // unit sizes are multiples of 4
    for (int i = (int)threadIdx.x; i < (n >> 2); i += (int)blockDim.x) ((uint32_t *)m)[3 + i] = src[i];
}

// EDI AF packets of the step after the path (EDI part; csrc/edi_af.h): one wavefront per packet,
// blockIdx.y = packet slot (frame * max_upf + unit) of the call, four streams per workgroup.
__global__ void __launch_bounds__(256) tl_edi_af_kernel(TlEdiArgs A)
{
    const int s = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (s >= A.nstreams) return;
    tl_edi_af_packet(A, s, (int)blockIdx.y);      // blockIdx.y = packet slot
}

// EDI PFT layer (csrc/edi_pft.h): one wavefront per AF packet, blockIdx.y = frame of the call.  The Reed-Solomon tables
// (11 KB) are copied into LDS once per workgroup: the encoder's look-ups are dependent and must not go to global memory.
__global__ void __launch_bounds__(256) tl_edi_pft_kernel(TlPftArgs A, const TlTables *T)
{
    __shared__ uint8_t s_log[256], s_exp[512], s_mlog[207 * TL_PFT_PARITY];
    __shared__ TlPftScratch scratch[4];
    for (int i = (int)threadIdx.x; i < 256; i += 256) s_log[i] = T->rs_log[i];
    for (int i = (int)threadIdx.x; i < 512; i += 256) s_exp[i] = T->rs_exp[i];
    for (int i = (int)threadIdx.x; i < 207 * TL_PFT_PARITY; i += 256) s_mlog[i] = (&T->rs_mlog[0][0])[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = (int)blockIdx.x * 4 + wave;
    if (s >= A.nstreams) return;
    const TlPftTables R = {s_log, s_exp, s_mlog};
    tl_edi_pft_packet(A, R, s, (int)blockIdx.y, scratch[wave]);
}

// pending frame (big-endian words in the stream state) -> bytes
__global__ void tl_flush_kernel(const TlStreamState *state, const TlConfig *configs, const int32_t *stream_cfg,
                                uint8_t *out, int32_t *out_len, int nstreams, int out_stride)
{
    const int s = (int)blockIdx.x;
    if (s >= nstreams) return;
    const bool any = state[s].frames_done > 0;
    const int n = any ? state[s].pending_len : 0;                   // frame_bytes, or one more (padding slot)
    for (int i = (int)threadIdx.x; i < out_stride; i += (int)blockDim.x)
        out[(size_t)s * out_stride + i] = i < n ? (uint8_t)(state[s].pending[i >> 2] >> (24 - 8 * (i & 3))) : 0;
    if (out_len && threadIdx.x == 0) out_len[s] = n;
}

// ------------------------------------------------------------------------------------------
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
    int32_t *d_work = nullptr;                   // unit counters of the persistent kernels
    bool work_clean = false;                     // ... are zero (tl_finish_kernel zeroes them after use)
    bool broken = false;                         // a launch or a reconfiguration failed half way: stream state, psy-2 state copies and lists may disagree;
                                                 // every further launch is refused (TLB_ERR_HIP) until tlb_reset() has put all streams back to zero
    int num_cu = 256;
};

static hipError_t stage_reserve(tlb_batch *b, int k, size_t bytes)
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

extern "C" {

int tlb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int tlb_lds_bytes_per_stream(void)
{   // per WAVE (= per unit in flight): the largest of the kernels' per-wave blocks
    size_t m = sizeof(TlMainLds);
    if (sizeof(TlPsyLds) > m) m = sizeof(TlPsyLds);
    if (sizeof(TlPsy2Lds) > m) m = sizeof(TlPsy2Lds);
    return (int)m;
}
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
        std::vector<int32_t> partner((size_t)nstreams, -1);
        std::vector<int> open(b->h_configs.size(), -1);              // per configuration: a mono stream still waiting for a partner
        for (int s2 = 0; s2 < nstreams; s2++) {
            const int ci = b->h_stream_cfg[s2];
            if (b->h_configs[(size_t)ci].nch != 1) continue;
            if (open[(size_t)ci] < 0) open[(size_t)ci] = s2;
            else { partner[(size_t)s2] = open[(size_t)ci]; partner[(size_t)open[(size_t)ci]] = s2; open[(size_t)ci] = -1; }
        }
        for (int p = 0; p < 4; p++) b->list_pairs[p] = false;
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
    std::vector<tlb_stream_config> &uniq = b->h_uniq;
    for (int s = 0; s < nstreams; s++) {
        int found = -1;
        for (size_t u = 0; u < uniq.size(); u++)
            if (uniq[u].samplerate == cfgs[s].samplerate && uniq[u].mode == cfgs[s].mode && uniq[u].bitrate == cfgs[s].bitrate &&
                uniq[u].psy_model == cfgs[s].psy_model && uniq[u].pad_len == cfgs[s].pad_len) { found = (int)u; break; }
        if (found < 0) {
            TlConfig c;
            int rc = tl_build_config(&c, cfgs[s].samplerate, cfgs[s].mode, cfgs[s].bitrate, cfgs[s].psy_model, cfgs[s].pad_len);
            if (rc) return rc;
            uniq.push_back(cfgs[s]);
            b->h_configs.push_back(c);
            found = (int)uniq.size() - 1;
        }
        b->h_stream_cfg[s] = found;
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
    int found = -1;
    for (size_t u = 0; u < b->h_uniq.size(); u++)
        if (b->h_uniq[u].samplerate == cfg->samplerate && b->h_uniq[u].mode == cfg->mode && b->h_uniq[u].bitrate == cfg->bitrate &&
            b->h_uniq[u].psy_model == cfg->psy_model && b->h_uniq[u].pad_len == cfg->pad_len) { found = (int)u; break; }
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

static int tlb_launch(tlb_batch *b, const int16_t *d_pcm, int nframes, const uint8_t *d_xpad, const int32_t *d_xpad_len,
                      uint8_t *d_out, TlTaps *d_taps, hipStream_t st, long long *d_stamps = nullptr, int32_t *d_out_len = nullptr)
{
    if (!b || !d_pcm || !d_out || nframes <= 0) return TLB_ERR_ARG;
    for (int p = 0; p < 4; p++) if ((long)b->n_list[p] * nframes > (1L << 30)) return TLB_ERR_ARG;   // unit indices are 32-bit; checked for every model before anything is queued
    if (b->broken) { fprintf(stderr, "libtoolame-dab-hip: this batch had a launch fail half way; tlb_reset() it before encoding on\n"); return TLB_ERR_HIP; }
    HIPCHK(hipSetDevice(b->device));
    // From the first kernel on the streams' state is in motion.  If anything below fails, the psy-2 state copy the next launch would
    // read may never have been written and the unit counters may be non-zero: the flip is taken back, the counters are re-zeroed by
    // the next launch, and the batch refuses further work until tlb_reset().
    struct Guard { tlb_batch *b; int flip; bool ok; ~Guard() { if (!ok) { b->psy2_flip = flip; b->work_clean = false; b->broken = true; } } } guard_{b, b->psy2_flip, false};
    TlLaunch A;
    memset(&A, 0, sizeof A);
    A.tables = b->d_tables; A.configs = b->d_configs; A.stream_cfg = b->d_stream_cfg; A.state = b->d_state;
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
        A.stream_list = b->d_list[p]; A.nlist = b->n_list[p];
        // persistent waves, twelve per CU (three per SIMD) in every kernel; they take their units off a counter
        const long units = (long)b->n_list[p] * nframes;
        A.padbits = b->pads[p] ? (uint8_t *)b->stage[7] : nullptr; A.newlag = b->d_newlag;
        if (b->pads[p]) { hipLaunchKernelGGL(tl_slots_kernel, dim3((unsigned)((b->n_list[p] + 255) / 256)), dim3(256), 0, st, A); HIPCHK(hipGetLastError()); }
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
            if (p == 1 && pr) hipLaunchKernelGGL((tl_frame_kernel<1, true>), dim3((unsigned)mb1), dim3(64 * TL_MAIN_WAVES), 0, st, A);
            else if (p == 1) hipLaunchKernelGGL((tl_frame_kernel<1, false>), dim3((unsigned)mb1), dim3(64 * TL_MAIN_WAVES), 0, st, A);
            else if (pr) hipLaunchKernelGGL((tl_frame_kernel<3, true>), dim3((unsigned)mb1), dim3(64 * TL_MAIN_WAVES), 0, st, A);
            else hipLaunchKernelGGL((tl_frame_kernel<3, false>), dim3((unsigned)mb1), dim3(64 * TL_MAIN_WAVES), 0, st, A);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(tl_finish_kernel, dim3((unsigned)((b->n_list[p] + 3) / 4)), dim3(256), 0, st, A);
            HIPCHK(hipGetLastError());
            b->work_clean = true;                                    // tl_finish_kernel leaves the counters at zero
            continue;
        }
        if (p == 2) hipLaunchKernelGGL(tl_psy2_kernel, dim3((unsigned)qb), dim3(64 * TL_PSY2_WAVES), 0, st, A);
        HIPCHK(hipGetLastError());
        if (p == 2 && b->n_list[p] == b->nstreams) { HIPCHK(hipEventRecord(b->ev_mid, st)); b->have_mid = true; }      // models 2/4 only in the batch: psy | encode split of the time
        long mb = (units + TL_MAIN_WAVES - 1) / TL_MAIN_WAVES;
        if (mb > b->num_cu) mb = b->num_cu;
        const bool pr = b->list_pairs[p] && !d_taps && !d_stamps;
        if (p == 0 && pr) hipLaunchKernelGGL((tl_main_kernel<0, true>), dim3((unsigned)mb), dim3(64 * TL_MAIN_WAVES), 0, st, A);       // model 0: no psy kernel
        else if (p == 0) hipLaunchKernelGGL((tl_main_kernel<0, false>), dim3((unsigned)mb), dim3(64 * TL_MAIN_WAVES), 0, st, A);
        else if (pr) hipLaunchKernelGGL((tl_main_kernel<2, true>), dim3((unsigned)mb), dim3(64 * TL_MAIN_WAVES), 0, st, A);
        else hipLaunchKernelGGL((tl_main_kernel<2, false>), dim3((unsigned)mb), dim3(64 * TL_MAIN_WAVES), 0, st, A);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(tl_finish_kernel, dim3((unsigned)((b->n_list[p] + 3) / 4)), dim3(256), 0, st, A);
        HIPCHK(hipGetLastError());
        b->work_clean = true;
    }
    HIPCHK(hipEventRecord(b->ev1, st));
    if (b->n_list[2]) b->psy2_flip ^= 1;         // only now: every kernel that writes the other copy has been queued
    guard_.ok = true;
    b->last_stream = st; b->timed = true;
    b->frames += nframes;
    return TLB_OK;
}

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
    hipLaunchKernelGGL(tl_ingest_kernel, dim3((unsigned)((size_t)nframes * (size_t)b->nstreams)), dim3(256), 0, (hipStream_t)hip_stream,
                       d_interleaved, d_pcm, d_peaks, b->d_gain, b->d_configs, b->d_stream_cfg, b->nstreams);
    HIPCHK(hipGetLastError());
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
    hipLaunchKernelGGL(tl_silence_kernel, dim3((unsigned)((b->nstreams + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       d_peaks, d_silence_ms, b->d_configs, b->d_stream_cfg, b->nstreams, nframes);
    HIPCHK(hipGetLastError());
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

static int zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream, const int32_t *d_frame_len);
int tlb_zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream)
{
    return zmq_frame_device(b, d_frames, d_peaks, nframes, d_msgs, hip_stream, nullptr);
}
// d_frame_len: int32 [nframes][nstreams] or null -- 0 marks a slot without a frame (a stream just reset inside a tick object): no message
static int zmq_frame_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_peaks, int nframes, uint8_t *d_msgs, void *hip_stream, const int32_t *d_frame_len)
{
    if (!b || !d_frames || !d_msgs || nframes <= 0) return TLB_ERR_ARG;
    if (!b->max_upf) return TLB_ERR_SAMPLERATE;
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(tl_zmq_frame_kernel, dim3((unsigned)((size_t)nframes * (size_t)b->max_upf * (size_t)b->nstreams)), dim3(128), 0, (hipStream_t)hip_stream,
                       d_frames, d_peaks, d_msgs, b->d_configs, b->d_stream_cfg, b->nstreams, b->out_stride, 12 + b->out_stride, b->max_upf, d_frame_len);
    HIPCHK(hipGetLastError());
    return TLB_OK;
}

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

static int edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
                         const char *version, int version_len, uint8_t *d_pkts, int32_t *d_pkt_len, void *hip_stream, const int32_t *d_frame_len);
int tlb_edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
                      const char *version, int version_len, uint8_t *d_pkts, int32_t *d_pkt_len, void *hip_stream)
{
    return edi_af_device(b, d_frames, d_levels, nframes, d_state, version, version_len, d_pkts, d_pkt_len, hip_stream, nullptr);
}
// d_frame_len: int32 [nframes][nstreams] or null -- 0 marks a slot without a frame: no packet, sender state untouched (csrc/edi_af.h)
static int edi_af_device(tlb_batch *b, const uint8_t *d_frames, const int16_t *d_levels, int nframes, tlb_edi_state *d_state,
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
    hipLaunchKernelGGL(tl_edi_af_kernel, dim3((unsigned)((b->nstreams + 3) / 4), (unsigned)(nframes * b->max_upf)), dim3(256), 0, st, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d_state, b->d_edi_state_tmp, sizeof(TlEdiState) * (size_t)b->nstreams, hipMemcpyDeviceToDevice, st));
    return TLB_OK;
}

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
static int pft_shape(int max_af_len, int fec, int chunk_len, int transport, int *max_frags, int *frag_stride)
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
    hipLaunchKernelGGL(tl_edi_pft_kernel, dim3((unsigned)((b->nstreams + 3) / 4), (unsigned)nframes), dim3(256), 0, st, A, (const TlTables *)b->d_tables);
    HIPCHK(hipGetLastError());
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

int tlb_flush_device_len(tlb_batch *b, uint8_t *d_out, int32_t *d_out_len, void *hip_stream)
{
    if (!b || !d_out) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(tl_flush_kernel, dim3(b->nstreams), dim3(128), 0, (hipStream_t)hip_stream, b->d_state, b->d_configs,
                       b->d_stream_cfg, d_out, d_out_len, b->nstreams, b->out_stride);
    HIPCHK(hipGetLastError());
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

}  // extern "C" (the shim's private state and helper have internal linkage: the library exports the nine names of libtoolame-dab.sym and tlb_*, nothing else)

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
};

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
static bool tick_input_free(const tlb_tick *t) { return t && !t->finished && t->ticks - t->waited < 2; }
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
    return tlb_stream_reset(G->b, k);
}
int tlb_tick_stream_finish(tlb_tick *t, int stream, uint8_t *out, size_t out_size)
{
    int k; TickGroup *G = tick_group_of(t, stream, &k);
    if (!G || t->finished) return -TLB_ERR_ARG;
    return tlb_stream_finish(G->b, k, out, out_size);
}
int tlb_tick_stream_reconfigure(tlb_tick *t, int stream, const tlb_stream_config *cfg)
{
    int k; TickGroup *G = tick_group_of(t, stream, &k);
    if (!G || t->finished) return TLB_ERR_ARG;
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

// Queue one tick -- copy-in, ingest, encode, egress, copy-out of every group -- on the input set the caller has just filled, and
// return at once.  tlb_tick_pcm() then points at the OTHER input set: the caller fills the next tick while this one is on its way
// (odr-audioenc decouples capture from encoding with its input queue, src/odr-audioenc.cpp:904-986).  The device buffers of a group
// are single, so across ticks: the next copy-in waits for this tick's ingest kernel, the next kernels for this tick's copy-out --
// the host-to-device link, the limit at large stream counts, never idles between ticks.  At most two ticks may be in flight
// (two host sets): submit, submit, wait, submit, wait, ...
int tlb_tick_submit(tlb_tick *t)
{
    if (!t || t->finished || t->ticks - t->waited >= 2) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(t->device));
    const int set = (int)(t->ticks & 1);                             // == in_set: ticks and input sets alternate together
    const int oset = (int)(t->ticks % 3);                            // output set: the caller may still be reading tick - 2's
    HIPCHK(hipEventRecord(t->ev0[oset], t->s_in));
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
        if (rc) { tick_drain(t); return rc; }
    }
    if (hipEventRecord(t->ev1[oset], t->s_out) != hipSuccess) { tick_drain(t); return TLB_ERR_HIP; }
    t->ticks++;
    t->in_set = (int)(t->ticks & 1);
    return TLB_OK;
}

// Wait for the oldest submitted tick; the read accessors then show ITS results until the next wait (three output sets: neither
// of the two ticks that can be submitted before that wait writes the set this one's results are in).
int tlb_tick_wait(tlb_tick *t)
{
    if (!t || t->waited >= t->ticks) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(t->device));
    const int set = (int)(t->waited % 3);
    if (hipEventSynchronize(t->ev1[set]) != hipSuccess) { tick_drain(t); return TLB_ERR_HIP; }
    t->out_set = set;
    t->waited++;
    return TLB_OK;
}

int tlb_tick_run(tlb_tick *t)
{   // one tick start to end: the accessors show its results when the call returns
    if (!t || t->ticks != t->waited) return TLB_ERR_ARG;
    if (int rc = tlb_tick_submit(t)) return rc;
    return tlb_tick_wait(t);
}

// end of the streams (toolame_finish): the pending frame of every stream through the egress stage; no further run
int tlb_tick_finish(tlb_tick *t)
{
    if (!t || t->finished || t->ticks == 0 || t->ticks != t->waited) return TLB_ERR_ARG;
    HIPCHK(hipSetDevice(t->device));
    const int set = (int)(t->ticks % 3);
    for (auto &G : t->groups) {
        int rc = hipStreamWaitEvent(t->s_run, G.ev_out, 0) == hipSuccess ? TLB_OK : TLB_ERR_HIP;
        if (!rc) rc = tlb_flush_device_len(G.b, G.d_frames, G.d_flen, t->s_run);
        if (!rc) {      // the egress sends the levels of the last run with the last frame (they are in the other host set; the device copy is current)
            rc = tick_egress(t, G, true, set, false);
        }
        if (rc) { tick_drain(t); return rc; }
    }
    if (hipStreamSynchronize(t->s_out) != hipSuccess) { tick_drain(t); return TLB_ERR_HIP; }
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
