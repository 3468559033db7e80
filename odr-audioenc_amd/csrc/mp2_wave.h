// mp2_wave.h -- the DAB MP2 (MPEG-1/2 Layer II) frame encoder as ONE WAVEFRONT PER (STREAM, FRAME).
//
// This is the MI355X-native restatement of toolame_encode_frame()
// (/root/reference/libtoolame-dab/toolame.c:267-554): 64 lanes cooperate on one frame of one stream,
// all per-frame working state lives in LDS / registers, HBM sees PCM in and frame bytes out.
// Units of work (tl_frame_unit, tl_main_unit, tl_psy2_chain, tl_finish_stream at the end of the file) are what
// the kernels in toolame_hip.hip hand to their waves.
//
// The file is written in a lane-SPMD style that compiles two ways from the same source:
//   * hipcc --offload-arch=gfx950 : TL_LANES_BEGIN/END open a per-lane scope, cross-lane
//     exchange goes through LDS or wave shuffles, TL_SYNC() is a wavefront-scope fence;
//   * -DTL_EMULATE (g++, tests only): TL_LANES_BEGIN/END are `for (lane = 0..63)` loops and
//     per-lane "registers" are [64] arrays, so the CPU test-suite executes exactly the device
//     algorithm (same arithmetic, same order) and compares it with the oracle.
//
// Exactness rules: fp64 everywhere, no FMA contraction (-ffp-contract=off),
// every reduction that the reference performs sequentially is owned by ONE lane and performed in
// the reference's order; lanes are only ever assigned whole outputs, never partial sums.
#pragma once
#include "mp2_types.h"
#include "tl_libm.h"
#include <math.h>
#include <stddef.h>

// ------------------------------------------------------------------------------------------
#ifdef TL_EMULATE
#define TL_FN static inline
#define TL_LANES_BEGIN for (int lane = 0; lane < 64; ++lane) {
#define TL_LANES_END }
#define TL_SYNC() ((void)0)
#define PV(T, name) T name[64]
#define PA(T, name, n) T name[64][n]
#define L(name) name[lane]
#define PARG(T, name) T (&name)[64]
#define PARGA(T, name, n) T (&name)[64][n]
#define TL_OTHER(name, idx, src) name[src] idx          /* value of another lane's register */
#define TL_ATOMIC_OR(p, v) (*(p) |= (v))
TL_FN uint64_t tlh_ballot(const bool (&p)[64]) { uint64_t m = 0; for (int i = 0; i < 64; i++) if (p[i]) m |= 1ull << i; return m; }
TL_FN uint64_t tlh_min_u64(const uint64_t (&v)[64]) { uint64_t m = v[0]; for (int i = 1; i < 64; i++) if (v[i] < m) m = v[i]; return m; }
TL_FN int tlh_sum_i32(const int (&v)[64]) { int s = 0; for (int i = 0; i < 64; i++) s += v[i]; return s; }
TL_FN uint32_t tlh_xor_u32(const uint32_t (&v)[64]) { uint32_t s = 0; for (int i = 0; i < 64; i++) s ^= v[i]; return s; }
TL_FN void tlh_exscan_i32(int (&d)[64], const int (&v)[64]) { int s = 0; for (int i = 0; i < 64; i++) { d[i] = s; s += v[i]; } }
TL_FN int tlh_argmin_u64(const uint64_t (&v)[64])
{   // lane of the smallest key, ties: channel-0 lanes (even) first, then ascending; -1 when every key is ~0
    const uint64_t m = tlh_min_u64(v);
    if (m == ~0ull) return -1;
    for (int i = 0; i < 64; i += 2) if (v[i] == m) return i;
    for (int i = 1; i < 64; i += 2) if (v[i] == m) return i;
    return -1;
}
TL_FN void tlh_row16_max_f64(double (&d)[64], const double (&v)[64])
{ for (int r = 0; r < 4; r++) { double m = v[16 * r]; for (int i = 1; i < 16; i++) if (m < v[16 * r + i]) m = v[16 * r + i]; for (int i = 0; i < 16; i++) d[16 * r + i] = m; } }
#define TL_ROW16_MAX_F64(dst, src) tlh_row16_max_f64(dst, src)      /* valid in lane 15 of each row of 16 (all lanes here) */
TL_FN void tlh_incl_xscan_u32(uint32_t (&d)[64], const uint32_t (&v)[64]) { uint32_t x = 0; for (int i = 0; i < 64; i++) { x ^= v[i]; d[i] = x; } }
#define TL_WAVE_INCL_XSCAN_U32(dst, src) tlh_incl_xscan_u32(dst, src)
// minimum / sum over the lanes of the caller's parity (the cells of one of two mono streams sharing the wave), delivered to each of them
TL_FN void tlh_par_min_u64(uint64_t (&d)[64], const uint64_t (&v)[64])
{ for (int p = 0; p < 2; p++) { uint64_t m = v[p]; for (int i = p; i < 64; i += 2) if (v[i] < m) m = v[i]; for (int i = p; i < 64; i += 2) d[i] = m; } }
TL_FN void tlh_par_sum_i32(int (&d)[64], const int (&v)[64])
{ for (int p = 0; p < 2; p++) { int m = 0; for (int i = p; i < 64; i += 2) m += v[i]; for (int i = p; i < 64; i += 2) d[i] = m; } }
#define TL_PAR_MIN_U64(dst, src) tlh_par_min_u64(dst, src)
#define TL_PAR_SUM_I32(dst, src) tlh_par_sum_i32(dst, src)
#define TL_BALLOT(name) tlh_ballot(name)
#define TL_SWAP1_U64(dst, src) do { for (int l_ = 0; l_ < 64; l_++) dst[l_] = src[l_ ^ 1]; } while (0)
#define TL_WAVE_ARGMIN_U64(name) tlh_argmin_u64(name)
#define TL_WAVE_MIN_U64(name) tlh_min_u64(name)
#define TL_WAVE_SUM_I32(name) tlh_sum_i32(name)
#define TL_WAVE_XOR_U32(name) tlh_xor_u32(name)
#define TL_WAVE_EXSCAN_I32(dst, src) tlh_exscan_i32(dst, src)
#define TL_UNI_I(x) (x)
#define TL_READLANE_I32(name, l) name[l]
#define TL_RESTRICT
#define TL_SELECT(c, a, b) ((c) ? (a) : (b))
#define TL_LAUNDER(p) ((void)0)
#define TL_KARG const TlLaunch *
#define TL_KEEP(x) ((void)0)
#define TL_PIN(x) ((void)0)
#define TL_TIE(p, v) ((void)0)
#define TL_LD2(p, a, b) do { const double *p_ = (p); (a) = p_[0]; (b) = p_[1]; } while (0)
#else
#define TL_FN __device__ __forceinline__
#define TL_LANES_BEGIN { int lane_ = (int)(threadIdx.x & 63u); asm volatile("" : "+v"(lane_)); __builtin_assume(lane_ >= 0 && lane_ < 64); const int lane = lane_;
#define TL_LANES_END } TL_SYNC();
#define TL_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define PV(T, name) T name
#define PA(T, name, n) T name[n]
#define L(name) name
#define PARG(T, name) T &name
#define PARGA(T, name, n) T (&name)[n]
#define TL_OTHER(name, idx, src) tld_shfl_f64(name idx, src)
#define TL_ATOMIC_OR(p, v) atomicOr((p), (v))
TL_FN double tld_shfl_f64(double v, int src) { return __shfl(v, src, 64); }
TL_FN double tld_swap1_f64(double v) {
    // value of lane^1 (the other channel of the same subband): DPP quad_perm [1,0,3,2], no LDS crossbar
    const uint64_t u = tl_d2u(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, 0xB1, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), 0xB1, 0xf, 0xf, true);
    return tl_u2d(((uint64_t)hi << 32) | lo);
}
TL_FN uint32_t tld_min_u32(uint32_t v) {
    // DPP reduction (gfx9): row_shr 1,2,4,8 -> row minimum in lane 15 of each row; row_bcast15 / row_bcast31
    // carry it across rows; lane 63 holds the wave minimum.  Shifted-in lanes read the identity.
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x118, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
TL_FN uint64_t tld_min_u64(uint64_t v) {
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mhi = tld_min_u32(hi);
    const uint32_t mlo = tld_min_u32(hi == mhi ? lo : 0xffffffffu);
    return ((uint64_t)mhi << 32) | mlo;
}
TL_FN int tld_incl_scan_i32(int v) {
    // inclusive prefix sum over the 64 lanes with DPP only (no ds_bpermute, no per-lane address registers):
    // Hillis-Steele inside each row of 16 (row_shr 1,2,4,8), then row_bcast15 / row_bcast31 across rows.
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
TL_FN uint32_t tld_xor_u32(uint32_t x) {
    int v = (int)x;
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
TL_FN int tld_sum_i32(int v) { return __builtin_amdgcn_readlane(tld_incl_scan_i32(v), 63); }
TL_FN int tld_exscan_i32(int v) { return tld_incl_scan_i32(v) - v; }
TL_FN int tld_argmin_u64(uint64_t v)
{   // as tlh_argmin_u64.  The low words are only reduced when several lanes share the smallest high word.
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mhi = tld_min_u32(hi);
    uint64_t m = (uint64_t)__ballot(hi == mhi);
    if (__builtin_popcountll(m) > 1) {
        const uint32_t mlo = tld_min_u32(hi == mhi ? lo : 0xffffffffu);
        if ((mhi & mlo) == 0xffffffffu) return -1;
        m = (uint64_t)__ballot(hi == mhi && lo == mlo);
    } else if (mhi == 0xffffffffu && (uint32_t)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(m)) == 0xffffffffu) return -1;
    const uint64_t even = m & 0x5555555555555555ull;
    return __builtin_ctzll(even ? even : m);
}
TL_FN double tld_row16_max_f64(double v)
{   // maximum over each row of 16 lanes, valid in the row's lane 15 (row_shr 1,2,4,8; shifted-in lanes keep their own value)
#pragma unroll
    for (int sh = 1; sh <= 8; sh <<= 1) {
        const uint64_t u = tl_d2u(v);
        const int ctl = 0x110 | sh;
        uint32_t lo, hi;
        switch (sh) {
        case 1: lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)u, (int)(uint32_t)u, 0x111, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(u >> 32), (int)(uint32_t)(u >> 32), 0x111, 0xf, 0xf, false); break;
        case 2: lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)u, (int)(uint32_t)u, 0x112, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(u >> 32), (int)(uint32_t)(u >> 32), 0x112, 0xf, 0xf, false); break;
        case 4: lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)u, (int)(uint32_t)u, 0x114, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(u >> 32), (int)(uint32_t)(u >> 32), 0x114, 0xf, 0xf, false); break;
        default: lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)u, (int)(uint32_t)u, 0x118, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(u >> 32), (int)(uint32_t)(u >> 32), 0x118, 0xf, 0xf, false); break;
        }
        (void)ctl;
        const double o = tl_u2d(((uint64_t)hi << 32) | lo);
        v = v < o ? o : v;
    }
    return v;
}
#define TL_ROW16_MAX_F64(dst, src) dst = tld_row16_max_f64(src)
TL_FN uint32_t tld_incl_xscan_u32(uint32_t x)
{   // inclusive XOR prefix over the 64 lanes, same DPP ladder as the integer sum scan
    int v = (int)x;
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (uint32_t)v;
}
#define TL_WAVE_INCL_XSCAN_U32(dst, src) dst = tld_incl_xscan_u32(src)
// Butterflies over the 32 lanes of one parity, the result in every lane (no readlane, no scalar round trip): row_ror 2 / 4 / 8 inside
// the rows of 16, then the two gfx950 row / half swaps (v_permlane16_swap, v_permlane32_swap: with both operands the same register
// the two results hold each lane's value and its counterpart's in the other row / half).
TL_FN uint32_t tld_par_min_u32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false); v = t < v ? t : v;
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = r[0] < r[1] ? r[0] : r[1];
    const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = q[0] < q[1] ? q[0] : q[1];
    return v;
}
TL_FN uint64_t tld_par_min_u64(uint64_t v)
{
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mhi = tld_par_min_u32(hi);
    const uint32_t mlo = tld_par_min_u32(hi == mhi ? lo : 0xffffffffu);
    return ((uint64_t)mhi << 32) | mlo;
}
TL_FN int tld_par_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = (int)(r[0] + r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = (int)(q[0] + q[1]);
    return v;
}
#define TL_PAR_MIN_U64(dst, src) dst = tld_par_min_u64(src)
#define TL_PAR_SUM_I32(dst, src) dst = tld_par_sum_i32(src)
#define TL_BALLOT(name) ((uint64_t)__ballot(name))
#define TL_SWAP1_U64(dst, src) dst = tl_d2u(tld_swap1_f64(tl_u2d(src)))
#define TL_WAVE_ARGMIN_U64(name) tld_argmin_u64(name)
#define TL_WAVE_MIN_U64(name) tld_min_u64(name)
#define TL_WAVE_SUM_I32(name) tld_sum_i32(name)
#define TL_WAVE_XOR_U32(name) tld_xor_u32(name)
#define TL_WAVE_EXSCAN_I32(dst, src) dst = tld_exscan_i32(src)
#define TL_UNI_I(x) __builtin_amdgcn_readfirstlane(x)
#define TL_READLANE_I32(name, l) __builtin_amdgcn_readlane(name, l)
#define TL_RESTRICT __restrict__
#define TL_SELECT(c, a, b) (__builtin_unpredictable(c) ? (a) : (b))      /* a v_cndmask, never a divergent branch */
#define TL_KEEP(x) asm volatile("" : : "v"(x))             /* x is computed (a load: issued) here, not sunk into a later branch */
typedef double tl_f64x2 __attribute__((ext_vector_type(2)));
#define TL_LD2(p, a, b) do { const tl_f64x2 v_ = *(const tl_f64x2 *)(p); (a) = v_.x; (b) = v_.y; } while (0)      /* two doubles, 16-byte aligned: one ds_read_b128 */
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) struct TlLaunch *TlKArg;     /* the kernel-argument segment: scalar loads */
#else
typedef const struct TlLaunch *TlKArg;                                        /* host pass of the same translation unit */
#endif
#define TL_KARG TlKArg
#define TL_LAUNDER(p) asm volatile("" : "+s"(p))       /* keeps loads through p inside the frame loop (no hoisting into long-lived VGPRs) */
#define TL_TIE(p, v) asm volatile("" : "+s"(p), "+v"(v))   /* loads through p are requested after v has been computed, not before: a software pipeline's order, pinned */
#define TL_PIN(x) asm volatile("" : "+v"(x))           /* a constant made once, here, in a vector register: machine LICM is off (csrc/Makefile), so a literal used inside a hot loop
                                                          is otherwise re-made by a v_mov on every trip */
#endif

// Issue priority of the wave (s_setprio 0..3).  A wave inside a serial, latency-bound piece (a dB-sum chain: one table look-up
// and six dependent operations per step, two to four lanes alive) has ONE instruction ready at a time; behind the long
// independent streams of the other waves of its SIMD every step waits for an issue slot it could have had at once.  Raising
// the priority for such pieces shortens them towards their uncontended latency and costs the throughput phases nothing they
// can notice (they always have another instruction to issue).  TL_PRIO_CHAIN=0 builds without it (measurement).
#ifndef TL_PRIO_CHAIN
#define TL_PRIO_CHAIN 3
#endif
// diagnostic: which of the streaming stages issue at the serial level (A/B builds)
#ifndef TL_PS_FHT
#define TL_PS_FHT 0
#endif
#ifndef TL_PS_THR
#define TL_PS_THR 1          // the threshold walks too: + 0.4 % psy 1, + 0.9 % psy 3 (their trip counts differ from lane to lane)
#endif
#ifndef TL_PS_FB
#define TL_PS_FB 0
#endif
#ifndef TL_PS_Q
#define TL_PS_Q 0
#endif
#ifndef TL_PS_POW
#define TL_PS_POW 1          // the power spectrum since its logarithms come in two halves (filing ballots, one dependent pass per 64 filed lines): psy 1 +- 0, psy 3 + 0.3 %
#endif
#ifndef TL_PRIO_SERIAL
#define TL_PRIO_SERIAL 2             // TL_PRIO_SERIAL=0 builds without it (measurement)
#endif
#ifdef TL_EMULATE
#define TL_PRIO(n) ((void)0)
#define TL_PRIO2(n) ((void)0)
#else
// TL_PRIO(1) ... TL_PRIO(0): a dB-sum chain (highest); it falls back to the serial level, which is what surrounds every chain.
// TL_PRIO2(1) / (0): a stage that is a dependent chain of look-ups, ballots and scans with few instructions to issue (tone labelling,
// compaction, decimation, thresholds, bit allocation, field writers, CRCs) / a stage that streams (transform, power spectrum,
// filterbank, quantiser).  Round 4, A/B on one box: + 1.9 % psy 1, + 1.9 % psy 3, + 1.8 % psy 0.
#define TL_PRIO(n) do { if (TL_PRIO_CHAIN) __builtin_amdgcn_s_setprio((n) ? TL_PRIO_CHAIN : TL_PRIO_SERIAL); } while (0)
#define TL_PRIO2(n) do { if (TL_PRIO_SERIAL) __builtin_amdgcn_s_setprio((n) ? TL_PRIO_SERIAL : 0); } while (0)
#endif
#ifdef TL_EMULATE
#define TL_STAMP(sp, k) ((void)0)
#else
#define TL_STAMP(sp, k) do { if (sp) { long long t_ = (long long)__builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0) (sp)[k] = t_; } } while (0)
#endif

#if defined(TL_EMULATE) && defined(TL_DEBUG_DUMP)
#include <stdio.h>
#include <stdlib.h>
#define TL_DBG_DUMP(tag, ch, nt, nn, x, b) do { if (getenv("TL_DUMP")) { printf("%s ch%d ntone %d nnoise %d:", tag, ch, nt, nn); \
    for (int i_ = 0; i_ < (nt) + (nn); i_++) printf(" (%.17g,%.6f)", (x)[i_], (b)[i_]); printf("\n"); } } while (0)
#define TL_DBG_WALK(ch, lane, cnt) do { if (getenv("TL_DUMP_WALK")) printf("walk ch%d lane %d cnt %d\n", ch, lane, cnt); } while (0)
static long tl_dbg_rounds = 0, tl_dbg_tones = 0, tl_dbg_deadheads = 0, tl_dbg_fronts = 0, tl_dbg_cands = 0;
#define TL_DBG_CAND(n) (tl_dbg_cands += (n))
#define TL_DBG_ROUND() (tl_dbg_rounds++)
#define TL_DBG_TONES(n, dh) (tl_dbg_tones += (n), tl_dbg_deadheads += (dh) ? 1 : 0, tl_dbg_fronts++)
#else
#define TL_DBG_DUMP(tag, ch, nt, nn, x, b) ((void)0)
#define TL_DBG_WALK(ch, lane, cnt) ((void)0)
#define TL_DBG_ROUND() ((void)0)
#define TL_DBG_TONES(n, dh) ((void)0)
#define TL_DBG_CAND(n) ((void)0)
#endif
// Diagnostic builds only (tools/instr_budget.sh): TL_EXP_LEVEL = n removes the last n stages of psy model 1 (results are then
// wrong on purpose); the VALU-instruction counters of successive levels attribute the instructions to the stages.
#ifndef TL_EXP_LEVEL
#define TL_EXP_LEVEL 0
#endif
// The same for the encoder phase (tools/class_budget.sh): TL_ENC_LEVEL = n removes its last n stages -- 1: CRC-16 / ScF-CRC / X-PAD,
// 2: + quantiser and sample packing, 3: + header / bit_alloc / scalefactor fields, 4: + bit allocation, 5: + scalefactors, SMR line and
// transmission pattern (the filterbank alone is left; its samples are kept alive by an empty asm statement).
#ifndef TL_ENC_LEVEL
#define TL_ENC_LEVEL 0
#endif
// And for the psy-2 kernel (tools/class_budget_psy2.sh): TL_P2_LEVEL = n removes the last n stages of a full pass of tl_psy2_pass --
// 1: the 32 subbands, 2: + the per-line thresholds, 3: + spreading / required SNR / permissible noise, 4: + the partition sums,
// 5: + the unpredictability (two sincos, c[]): the pass is then a seed pass, 6: + the polar form (energy, square root, arctangent),
// 7: the whole pass (window + transform too).  TL_P2_SUB picks ONE operation out of the line loop instead (levels 0 only):
// 1: no sincos of the predicted phase, 2: no sincos at all, 3: no arctangent, 4: no square roots and no division.
#ifndef TL_P2_LEVEL
#define TL_P2_LEVEL 0
#endif
#ifndef TL_P2_SUB
#define TL_P2_SUB 0
#endif
// ... and for psy model 3: TL_P3_NOTAIL = 1 leaves out the threshold chains of the eight top subset lines (tl_psy3_back), to measure them
#ifndef TL_P3_NOTAIL
#define TL_P3_NOTAIL 0
#endif
#define TL_DBMIN (-200.0)
#define TL_POWERNORM 90.3090
#define TL_T_NOISE 10
#define TL_T_TONE 20
#define TL_LAST (-1)
#define TL_STOP (-100)

// ------------------------------------------------------------------------------------------
// Per-wave LDS working set.
#define TL_CAND_MAX 256              // local maxima with passing right side (<= 249)
#ifndef TL_FB_BATCH_MAIN
#define TL_FB_BATCH_MAIN 6           // encode kernel of the split path: 36 = 6 x 6 (a smaller scratch, fewer live registers)
#endif
#define TL_PSY_EXT 5                 // tl_encode_frame<TL_PSY_EXT>: SMR from the record the model left in the wave's LDS (models 1 and 3)
#define TL_TONE_MAX 77               // confirmed tones per channel-frame (hard bound 75: a tone erases run lines either side, 20 + 16 + 19 + 19 fit below line 500); sized so that three 4-wave psy workgroups fit one CU's LDS
#define TL_MASKER_MAX 128            // tones + noise components after decimation
// psy 1/3: the FHT needs 1024 doubles, what follows it needs the 513 energies (lower half) and the power spectrum in dB
// (520 entries) side by side -- so the power spectrum lives in the transform's upper half (dead once the energies exist),
// 9 entries longer than the transform: 4 KB less LDS per wave than a separate array.
#define TL_FFT_WORDS (513 + 520)
#define TL_PX(w) ((w).u.fft + 513)
// Energies are stored at i ^ ((i >> 4) & 15) (a permutation inside each group of 16 lines): line-parallel accesses stay
// spread over the banks, and the spike sums -- a lane per subband walking its 16 lines (psycho_1.c:252-257) -- no longer
// collide (16 lanes at stride 16 doubles would share one bank pair; the XOR gives each its own).
#define TL_EX(i) ((i) ^ (((i) >> 4) & 15))
// Per-wave LDS of the encode kernel (the psy models run in their own kernels): PCM staging / frame being
// packed, the filterbank's window-output scratch and the small per-subband arrays.
#define TL_YP_ROW 18
struct TlMainLds {
    static constexpr int kFbBatch = TL_FB_BATCH_MAIN;
    union alignas(16) {
        struct { int16_t pcm[2][TL_HIST + 1152]; } fbk;
        uint32_t frame[2][TL_MAX_FRAME_WORDS + 2];         // the frame being packed (+ 2: tl_put_bits48); [1]: the second unit of a mono pair (tl_encode_pair)
    } u;
    // window outputs of a batch of blocks, dealt by (channel, parity of the index): row 2 c + (j & 1) holds yprime[j] at [j >> 1], rows 18 doubles
    // apart (16-byte aligned, the four rows a wave reads at once on different banks) -- a matrixing lane reads TWO consecutive operands of its
    // chain with one 16-byte load (mp2_fb.h)
    alignas(16) double yp[TL_FB_BATCH_MAIN][TL_YP_ROW * 3 + 16];
    double smr[2][32];                  // models 1 and 3: until the SMR line, the level of the model's record
    double psy_m[2][32];                // models 1 and 3: minimum masking threshold of the model's record
    int16_t ncentre[32];                // (ScF-CRC scratch)
    uint8_t scf[2][3][32];
    uint8_t jscale[3][32];
    uint8_t scfsi[2][32];
    uint8_t balloc[2][32];
    uint8_t minidx[2][32];
    uint8_t xpad[TL_MAX_XPAD];
    typedef double (*YpRows)[TL_YP_ROW * 3 + 16];
#ifdef TL_EMULATE
    YpRows yp_rows() { return yp; }
#else
    __device__ YpRows yp_rows() { return yp; }
#endif
};
// Per-wave LDS of the psy kernel of models 2 and 4: the transform / energies (partition sums in its dead upper half) and c[] / fthr[].
struct alignas(16) TlPsy2Lds {       // (16: the partition sums are read as pairs, tl_psy2_pass)
    struct { double fft[1024]; } u;
    double px[536];                  // (513 lines + one slot of padding per 32: tl_psy2_pass lays fthr[] out for the subband walk)
};
// Per-wave LDS of the psy kernel (models 1 and 3).
struct TlPsyLds {
    struct { double fft[TL_FFT_WORDS]; } u;
    double tone_x[TL_TONE_MAX];
    double nsum[32];
    uint32_t cinfo[TL_CAND_MAX];
    int16_t conf_c[TL_TONE_MAX];
    int16_t conf_nxt[TL_TONE_MAX];
    int16_t tlist[TL_TONE_MAX];
    int16_t ncentre[32];
    int16_t bandoff[40];
    uint8_t ptype[520];
};
// masker lists / thresholds live in the low half of the FHT buffer once the energies are no longer needed
#define TL_MK_X(w) ((w).u.fft)                         /* [TL_MASKER_MAX] */
#define TL_MK_BARK(w) ((w).u.fft + TL_MASKER_MAX)      /* [TL_MASKER_MAX] */
#define TL_LTG(w) ((w).u.fft + 2 * TL_MASKER_MAX)      /* [136] */
// per-masker constants of the threshold loops, computed once per masker instead of once per (masker, line)
// av = level term (psycho_1.c:493,512), g = 0.4x+6, n = 17-0.15x; c17 = 17.0 sits BETWEEN g and n so that the pair of slopes a
// line needs is one 16-byte window of the record: (g, 17) for a masker above the line, (17, n) for one below (tl_mask_term_w)
struct TlMasker { double bark, av, g, c17, n; };
#define TL_MK4(w) ((TlMasker *)((w).u.fft + 2 * TL_MASKER_MAX + 136))   /* [TL_MASKER_MAX], ends at fft[1032] */
static_assert(2 * TL_MASKER_MAX + 136 + 5 * TL_MASKER_MAX <= TL_FFT_WORDS, "masker records fit the transform buffer");



// Emulation only: how often the power spectrum's deferral list ran FULL (64 filed lines put through the logarithm's other branch before the
// line loop was done, mp2_psy13.h) -- tests/test_emu_parity.py crafts a spectrum that takes that path and wants to see that it did.
#ifdef TL_EMULATE
static long tl_emu_near1_full = 0;
#define TL_DBG_NEAR1_FULL() (tl_emu_near1_full++)
#else
#define TL_DBG_NEAR1_FULL() ((void)0)
#endif

// ---- the stages, one file each.  They are FRAGMENTS of this header: they rely on the macros and LDS blocks above and on each
// other in this order, and are not meant to be included on their own. ----
#define MP2_WAVE_PARTS 1
#include "mp2_dbsum.h"
#include "mp2_fht.h"
#include "mp2_psy13.h"
#include "mp2_psy24.h"
#include "mp2_fb.h"
#include "mp2_alloc.h"
#include "mp2_pack.h"
#include "mp2_units.h"
#undef MP2_WAVE_PARTS
