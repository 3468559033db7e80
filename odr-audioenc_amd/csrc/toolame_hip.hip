// toolame_hip.hip -- the gfx950 kernels of the library and their launchers (tl_kernels.h).  The C-ABI of include/toolame_batch.h lives in
// tlb_batch.cpp (batches), tlb_egress.cpp (ZeroMQ / EDI / PFT), tlb_tick.cpp (the real-time tick), tlb_node.cpp (all GPUs of a host) and
// toolame_legacy.cpp (the reference's nine functions).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
// -ffp-contract=off is load-bearing: the reference is built -std=c99 (no FMA contraction,
// Makefile.am:68) and every MAC chain / the quantiser d*a+b must round twice.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include "../../include/toolame_batch.h"
#include "mp2_host.h"
#include "mp2_wave.h"
#include "edi_af.h"
#include "edi_pft.h"
#include "tl_kernel_util.h"
#include "tl_kernels.h"

static_assert(TL_MAX_XPAD == TLB_MAX_XPAD, "xpad record size");


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
// psy kernel of models 2 and 4: csrc/toolame_psy2.hip, a translation unit of its own (it wants the IR load / store vectorizer the
// other kernels are built without, see csrc/Makefile).  TL_PSY2_WAVES waves per workgroup.
__global__ void tl_psy2_kernel(TlLaunch A);
// encode kernel of the split path: the tables it needs on dependent-load chains (TlBlockShared without the dB-sum table)
struct TlMainShared { double enw_s[512]; char bytes[sizeof(TlBlockShared) - offsetof(TlBlockShared, scalefactor)]; TlPackTables pack; };
static_assert(offsetof(TlBlockShared, dbtable) == 0 && offsetof(TlBlockShared, scalefactor) == sizeof(double) * 1002, "dbtable leads TlBlockShared");
static_assert((sizeof(TlMainShared) + TL_MAIN_WAVES * sizeof(TlMainLds) + TL_LDS_GRANULE - 1) / TL_LDS_GRANULE <= 128, "the encode workgroup fits a CU");
// PAIRS: the list contains mono streams that share waves in pairs (TlLaunch::partner, tl_encode_pair).  A second instantiation, so
// that the kernel of lists without pairs -- every all-stereo batch -- carries none of the pair code (with it inline the register
// allocation of the stereo path moved from 152 to 168 VGPRs and psy-1 stereo lost 1.1 %).
template <int PSY, bool PAIRS, int NCH = 0>     // 0: model 0 (no psy kernel); 2: models 2 and 4 (after tl_psy2_kernel); NCH = 2: a list of two-channel streams only (as tl_frame_kernel)
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
        const int s = A.stream_list ? __builtin_amdgcn_readfirstlane(A.stream_list[k]) : k;      // no list: the launch's streams are ALL streams, position = id (one round trip to L2 less per unit)
        if constexpr (PAIRS) {
            int s2;
            if (!tl_unit_partner(A, s, s2)) continue;                // the partner's wave encodes this mono stream's frame with its own
            s2 = __builtin_amdgcn_readfirstlane(s2);
            if (s2 >= 0) { tl_main_pair<PSY>(wl, B, sh.enw_s, &sh.pack, A, s, s2, f); continue; }
        }
        tl_main_unit<PSY, NCH>(wl, B, sh.enw_s, &sh.pack, A, s, f);
    }
}

// Models 1 and 3: psy model and encoder of a (stream, frame) unit by the same wave, one after the other (tl_frame_unit).
// Registers: the model needs 118, the encoder 154 -- the kernel needs the larger, not the sum, because the model runs BEFORE
// the filterbank fills its 72 sample registers (round 1's fused kernel ran it in the middle: 256).  LDS: the union of the
// two phases' blocks; both table sets (dB sums 7.8 KB, encoder 12.1 KB) once per workgroup.  The PCM the model read is still
// in L2 when the encoder stages it.
static_assert((sizeof(double) * 1258 + sizeof(TlMainShared) + TL_MAIN_WAVES * sizeof(TlFrameLds) + TL_LDS_GRANULE - 1) / TL_LDS_GRANULE <= 128, "the workgroup fits a CU");
template <int PSY, bool PAIRS, int NCH = 0>      // NCH = 2: the list holds two-channel streams only (tlb_batch.cpp: list_stereo) -- the mono paths are not in the kernel and a unit's
                                                 // first transform does not wait for the stream's configuration record
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
        const int s = A.stream_list ? __builtin_amdgcn_readfirstlane(A.stream_list[k]) : k;      // no list: the launch's streams are ALL streams, position = id (one round trip to L2 less per unit)
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
        } else tl_frame_unit<PSY, NCH>(wl, dbt, B, sh.enw_s, &sh.pack, A1, a2, s, f);
#endif
    }
}

// slot recurrence of the split path (44.1 / 22.05 kHz streams only): one lane per stream (tl_slots_stream)
__global__ void __launch_bounds__(256) tl_slots_kernel(TlLaunch A)
{
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k >= A.nlist) return;
    tl_slots_stream(A, A.stream_list ? A.stream_list[k] : k);
}

// finish pass of the split path: one wave per stream (tl_finish_stream)
__global__ void __launch_bounds__(256) tl_finish_kernel(TlLaunch A)
{
    // the unit counters of the list's persistent kernels are dead now: zero them for the next launch (instead of a memset per launch)
    if (blockIdx.x == 0 && threadIdx.x < 9) A.work[threadIdx.x * TL_HEAD_STRIDE] = 0;
    const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (k >= A.nlist) return;
    tl_finish_stream(A, A.stream_list ? __builtin_amdgcn_readfirstlane(A.stream_list[k]) : k);
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
// launchers (tl_kernels.h): the only way the host translation units reach a kernel
#define TLK_GO(...) do { hipLaunchKernelGGL(__VA_ARGS__); return hipGetLastError(); } while (0)
hipError_t tlk_slots(unsigned blocks, hipStream_t st, const TlLaunch &A) { TLK_GO(tl_slots_kernel, dim3(blocks), dim3(256), 0, st, A); }
hipError_t tlk_frame(int psy, bool pairs, bool stereo, unsigned blocks, hipStream_t st, const TlLaunch &A)
{
    const dim3 g(blocks), t(64 * TL_MAIN_WAVES);
    if (stereo && !pairs && psy == 1) TLK_GO((tl_frame_kernel<1, false, 2>), g, t, 0, st, A);
    if (stereo && !pairs) TLK_GO((tl_frame_kernel<3, false, 2>), g, t, 0, st, A);
    if (psy == 1 && pairs) TLK_GO((tl_frame_kernel<1, true>), g, t, 0, st, A);
    if (psy == 1) TLK_GO((tl_frame_kernel<1, false>), g, t, 0, st, A);
    if (pairs) TLK_GO((tl_frame_kernel<3, true>), g, t, 0, st, A);
    TLK_GO((tl_frame_kernel<3, false>), g, t, 0, st, A);
}
hipError_t tlk_main(int psy, bool pairs, bool stereo, unsigned blocks, hipStream_t st, const TlLaunch &A)
{
    const dim3 g(blocks), t(64 * TL_MAIN_WAVES);
    if (stereo && !pairs && psy == 0) TLK_GO((tl_main_kernel<0, false, 2>), g, t, 0, st, A);
    if (stereo && !pairs) TLK_GO((tl_main_kernel<2, false, 2>), g, t, 0, st, A);
    if (psy == 0 && pairs) TLK_GO((tl_main_kernel<0, true>), g, t, 0, st, A);       // model 0: no psy kernel
    if (psy == 0) TLK_GO((tl_main_kernel<0, false>), g, t, 0, st, A);
    if (pairs) TLK_GO((tl_main_kernel<2, true>), g, t, 0, st, A);
    TLK_GO((tl_main_kernel<2, false>), g, t, 0, st, A);
}
hipError_t tlk_psy2(unsigned blocks, hipStream_t st, const TlLaunch &A) { TLK_GO(tl_psy2_kernel, dim3(blocks), dim3(64 * TL_PSY2_WAVES), 0, st, A); }
hipError_t tlk_finish(unsigned blocks, hipStream_t st, const TlLaunch &A) { TLK_GO(tl_finish_kernel, dim3(blocks), dim3(256), 0, st, A); }
hipError_t tlk_ingest(unsigned blocks, hipStream_t st, const int16_t *in, int16_t *out, int16_t *peaks, const double *gain,
                      const TlConfig *configs, const int32_t *stream_cfg, int nstreams)
{
    TLK_GO(tl_ingest_kernel, dim3(blocks), dim3(256), 0, st, in, out, peaks, gain, configs, stream_cfg, nstreams);
}
hipError_t tlk_silence(unsigned blocks, hipStream_t st, const int16_t *peaks, uint32_t *silence_ms, const TlConfig *configs,
                       const int32_t *stream_cfg, int nstreams, int nframes)
{
    TLK_GO(tl_silence_kernel, dim3(blocks), dim3(256), 0, st, peaks, silence_ms, configs, stream_cfg, nstreams, nframes);
}
hipError_t tlk_zmq_frame(unsigned blocks, hipStream_t st, const uint8_t *frames, const int16_t *peaks, uint8_t *msgs, const TlConfig *configs,
                         const int32_t *stream_cfg, int nstreams, int out_stride, int msg_stride, int max_upf, const int32_t *frame_len)
{
    TLK_GO(tl_zmq_frame_kernel, dim3(blocks), dim3(128), 0, st, frames, peaks, msgs, configs, stream_cfg, nstreams, out_stride, msg_stride, max_upf, frame_len);
}
hipError_t tlk_edi_af(unsigned bx, unsigned by, hipStream_t st, const TlEdiArgs &A) { TLK_GO(tl_edi_af_kernel, dim3(bx, by), dim3(256), 0, st, A); }
hipError_t tlk_edi_pft(unsigned bx, unsigned by, hipStream_t st, const TlPftArgs &A, const TlTables *T) { TLK_GO(tl_edi_pft_kernel, dim3(bx, by), dim3(256), 0, st, A, T); }
hipError_t tlk_flush(unsigned blocks, hipStream_t st, const TlStreamState *state, const TlConfig *configs, const int32_t *stream_cfg,
                     uint8_t *out, int32_t *out_len, int nstreams, int out_stride)
{
    TLK_GO(tl_flush_kernel, dim3(blocks), dim3(128), 0, st, state, configs, stream_cfg, out, out_len, nstreams, out_stride);
}
#undef TLK_GO
size_t tlk_lds_bytes_per_wave(void)
{
    size_t m = sizeof(TlMainLds);
    if (sizeof(TlPsyLds) > m) m = sizeof(TlPsyLds);
    if (sizeof(TlPsy2Lds) > m) m = sizeof(TlPsy2Lds);
    return m;
}
