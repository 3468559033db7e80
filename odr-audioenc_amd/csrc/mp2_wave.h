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
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) struct TlLaunch *TlKArg;     /* the kernel-argument segment: scalar loads */
#else
typedef const struct TlLaunch *TlKArg;                                        /* host pass of the same translation unit */
#endif
#define TL_KARG TlKArg
#define TL_LAUNDER(p) asm volatile("" : "+s"(p))       /* keeps loads through p inside the frame loop (no hoisting into long-lived VGPRs) */
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
#define TL_PS_POW 0
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
static long tl_dbg_rounds = 0;
#define TL_DBG_ROUND() (tl_dbg_rounds++)
#else
#define TL_DBG_DUMP(tag, ch, nt, nn, x, b) ((void)0)
#define TL_DBG_WALK(ch, lane, cnt) ((void)0)
#define TL_DBG_ROUND() ((void)0)
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
struct TlMainLds {
    static constexpr int kFbBatch = TL_FB_BATCH_MAIN;
    union alignas(16) {
        struct { int16_t pcm[2][TL_HIST + 1152]; } fbk;
        uint32_t frame[2][TL_MAX_FRAME_WORDS + 2];         // the frame being packed (+ 2: tl_put_bits48); [1]: the second unit of a mono pair (tl_encode_pair)
    } u;
    double yp[TL_FB_BATCH_MAIN][2][34];
    double smr[2][32];                  // models 1 and 3: until the SMR line, the level of the model's record
    double psy_m[2][32];                // models 1 and 3: minimum masking threshold of the model's record
    int16_t ncentre[32];                // (ScF-CRC scratch)
    uint8_t scf[2][3][32];
    uint8_t jscale[3][32];
    uint8_t scfsi[2][32];
    uint8_t balloc[2][32];
    uint8_t minidx[2][32];
    uint8_t xpad[TL_MAX_XPAD];
    typedef double (*YpRows)[2][34];
#ifdef TL_EMULATE
    YpRows yp_rows() { return yp; }
#else
    __device__ YpRows yp_rows() { return yp; }
#endif
};
// Per-wave LDS of the psy kernel of models 2 and 4: the transform / energies (partition sums in its dead upper half) and c[] / fthr[].
struct TlPsy2Lds {
    struct { double fft[1024]; } u;
    double px[520];
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



// ------------------------------------------------------------------------------------------
TL_FN double tl_add_db(const double *TL_RESTRICT dbtable, double a, double b)
{   // psycho_1.c:180-205 == psycho_3.c:44-69, written without branches (every lane of a wave walks its own
    // chain) and with nothing but the final add behind the table read.  Inside |fdiff| <= 990 the index is the
    // reference's (int)fdiff; beyond it the reference returns the larger operand unchanged, which is
    // operand + table[1000] with table[1000] = -0.0.
    const double fdiff = 10.0 * (a - b);
    const double af = __builtin_fabs(fdiff);
    const int mag = (int)af;                                        // == |(int)fdiff|: truncation is symmetric
    const int idx = TL_SELECT(af > 990.0, 1000, mag);
    const double base = TL_SELECT(fdiff > -1.0, a, b);              // (int)fdiff >= 0
    return base + dbtable[idx];
}
// Two independent dB sums at once: both table entries are requested before either is used.
TL_FN void tl_add_db2(const double *TL_RESTRICT dbtable, double &a0, double b0, double &a1, double b1)
{
    const double f0 = 10.0 * (a0 - b0), f1 = 10.0 * (a1 - b1);
    const double g0 = __builtin_fabs(f0), g1 = __builtin_fabs(f1);
    int i0 = TL_SELECT(g0 > 990.0, 1000, (int)g0), i1 = TL_SELECT(g1 > 990.0, 1000, (int)g1);
    const double s0 = TL_SELECT(f0 > -1.0, a0, b0), s1 = TL_SELECT(f1 > -1.0, a1, b1);
    TL_KEEP(i0); TL_KEEP(i1);
    const double t0 = dbtable[i0], t1 = dbtable[i1];
    a0 = s0 + t0; a1 = s1 + t1;
}
TL_FN void tl_add_db2_k(const double *TL_RESTRICT dbtable, int k1000, double &a0, double b0, double &a1, double b1)
{
    const double f0 = 10.0 * (a0 - b0), f1 = 10.0 * (a1 - b1);
    const double g0 = __builtin_fabs(f0), g1 = __builtin_fabs(f1);
    int i0 = TL_SELECT(g0 > 990.0, k1000, (int)g0), i1 = TL_SELECT(g1 > 990.0, k1000, (int)g1);
    const double s0 = TL_SELECT(f0 > -1.0, a0, b0), s1 = TL_SELECT(f1 > -1.0, a1, b1);
    TL_KEEP(i0); TL_KEEP(i1);
    const double t0 = dbtable[i0], t1 = dbtable[i1];
    a0 = s0 + t0; a1 = s1 + t1;
}
TL_FN uint64_t tl_mnr_key(double mnr)
{   // order-preserving map double -> u64 for the allocation arg-min; ~0 = never chosen (encode_new.c:1068: small = 999999.0)
    uint64_t u = tl_d2u(mnr + 0.0);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    return 999999.0 > mnr ? u : ~0ull;
}
// One step of a threshold chain: x (+) the masking of one masker at distance dz bark, if it reaches the line at all
// (psycho_1.c:489-517 == psycho_3.c:352-394).  The reference's masking function
//   dz < -1: 17*(dz+1) - g     dz < 0: g*dz     dz < 1: -17*dz     else: -(dz-1)*n - 17        (g = 0.4x+6, n = 17-0.15x)
// is, with a = |dz|,  -(A*(a - B) + C):  inside |dz| < 1  A = (dz<0 ? g : 17), B = C = 0;  outside  A = (dz<0 ? 17 : n), B = 1,
// C = (dz<0 ? g : 17) -- the same roundings (negating an operand or a result changes no rounding; adding or subtracting a
// zero changes no bit; at dz = -1 and dz = 1 both neighbouring pieces give the same value), and level + vf = level - (...).
// C is the inside A times B (a product with 0.0 or 1.0 is exact): 64-bit selects cost two instructions, a product one.
// A masker out of reach (dz outside [-3, 8)) enters the dB sum as a level below -65536 dB, which leaves the sum as it
// is (|difference| > 99 dB: the reference returns the larger operand, tl_add_db adds its -0.0 entry).
// The literals of the threshold walk, made once per walk (TL_PIN) instead of once per masker and line.
struct TlMaskK { uint32_t c17_hi, one_hi, far_hi; int k1000; };
TL_FN TlMaskK tl_mask_consts()
{
    TlMaskK k;
    k.c17_hi = 0x40310000u; k.one_hi = 0x3ff00000u; k.far_hi = 0xC0F00000u; k.k1000 = 1000;
    TL_PIN(k.c17_hi); TL_PIN(k.one_hi); TL_PIN(k.far_hi); TL_PIN(k.k1000);
    return k;
}
TL_FN double tl_mask_term(double dz, double av, double g, double n, bool live = true)
{
    const double ad = __builtin_fabs(dz);
    const bool s = dz < 0.0, o = ad >= 1.0;
    const double G = TL_SELECT(s, g, 17.0), H = TL_SELECT(s, 17.0, n);
    const double A = TL_SELECT(o, H, G), Bc = TL_SELECT(o, 1.0, 0.0);
    const double term = av - (A * (ad - Bc) + G * Bc);
    const bool in = live && dz >= -3.0 && dz < 8.0;
    const uint64_t tu = tl_d2u(term);
    const uint32_t hi = TL_SELECT(in, (uint32_t)(tu >> 32), 0xC0F00000u);
    return tl_u2d(((uint64_t)hi << 32) | (tu & 0xffffffffull));
}
// The same term with the walk's pinned literals.  Same operations on the same values: 17.0 = {c17_hi, 0}, 1.0 = {one_hi, 0}.
TL_FN double tl_mask_term_k(const TlMaskK &k, double dz, double av, double g, double n, bool live = true)
{
    const double ad = __builtin_fabs(dz);
    const bool s = dz < 0.0, o = ad >= 1.0;
    const uint64_t gu = tl_d2u(g), nu = tl_d2u(n);
    const uint32_t Gh = TL_SELECT(s, (uint32_t)(gu >> 32), k.c17_hi), Gl = TL_SELECT(s, (uint32_t)gu, 0u);
    const uint32_t Hh = TL_SELECT(s, k.c17_hi, (uint32_t)(nu >> 32)), Hl = TL_SELECT(s, 0u, (uint32_t)nu);
    const uint32_t Ah = TL_SELECT(o, Hh, Gh), Al = TL_SELECT(o, Hl, Gl);
    const double G = tl_u2d(((uint64_t)Gh << 32) | Gl), A = tl_u2d(((uint64_t)Ah << 32) | Al);
    const double Bc = tl_u2d((uint64_t)TL_SELECT(o, k.one_hi, 0u) << 32);
    const double term = av - (A * (ad - Bc) + G * Bc);
    const bool in = live && dz >= -3.0 && dz < 8.0;
    const uint64_t tu = tl_d2u(term);
    const uint32_t hi = TL_SELECT(in, (uint32_t)(tu >> 32), k.far_hi);
    return tl_u2d(((uint64_t)hi << 32) | (tu & 0xffffffffull));
}
// 1 for a negative x, else 0.  On the device one shift of the high word, opaque to the compiler (which otherwise folds it into the
// address arithmetic that follows as shift + and + add: three instructions where shift + shift-add do).
TL_FN int tl_sign_bit(double x)
{
#ifdef TL_EMULATE
    return (int)(tl_d2u(x) >> 63);
#else
    int r;
    asm("v_lshrrev_b32 %0, 31, %1" : "=v"(r) : "v"((uint32_t)(tl_d2u(x) >> 32)));
    return r;
#endif
}
// The same term without a select for its shape.  dzp = masker bark - line bark = -dz (exactly: negation commutes with rounding).
//  * which pair of slopes (inner G, outer H): dz < 0 -> (g, 17), else (17, n) -- the 16-byte window of the masker's record at
//    &g + (dzp < 0): one address computed from the sign bit, one LDS read.  At dz = 0 either window serves (both products are 0).
//  * inside / outside |dz| = 1:  A (ad - Bc) + G Bc  with  (A, Bc) = (H, 1) outside, (G, 0) inside  is  H max(ad - 1, 0) + G min(ad, 1):
//    outside the very same operations (ad - 1.0; G * 1.0 == G); inside G * ad plus a zero in either form, and adding a zero of
//    either sign to a sum changes no bit of it unless the sum is itself a zero -- in which case the term is av - (+-0) = av
//    in either form, av never being -0.0 (a sum of finite non-zero values never rounds to -0).
//  * the reach test -3 <= dz < 8 is -8 < dzp <= 3.
TL_FN double tl_mask_term_w(const TlMasker *TL_RESTRICT m, double dzp, double av, uint32_t far_hi, bool live = true)
{
    const double ad = __builtin_fabs(dzp);
    const double *gh = &m->g + tl_sign_bit(dzp);
    const double G = gh[0], H = gh[1];
    const double t1 = __builtin_fmax(ad - 1.0, 0.0), t2 = __builtin_fmin(ad, 1.0);
    const double term = av - (H * t1 + G * t2);
    const bool in = live && dzp <= 3.0 && dzp > -8.0;
    const uint64_t tu = tl_d2u(term);
    const uint32_t hi = TL_SELECT(in, (uint32_t)(tu >> 32), far_hi);
    return tl_u2d(((uint64_t)hi << 32) | (tu & 0xffffffffull));
}
TL_FN double tl_mask_step(const double *TL_RESTRICT db, double x, const TlMasker *TL_RESTRICT m, double dzp, double av, uint32_t far_hi)
{
    return tl_add_db(db, x, tl_mask_term_w(m, dzp, av, far_hi));
}
TL_FN void tl_masker_consts(TlMasker *TL_RESTRICT mk, const double *TL_RESTRICT mx, const double *TL_RESTRICT mbk, int t, bool tonal)
{
    const double x = mx[t], mb = mbk[t];
    mk[t].bark = mb;
    mk[t].av = tonal ? -1.525 - 0.275 * mb - 4.5 + x : -1.525 - 0.175 * mb - 0.5 + x;
    mk[t].g = 0.4 * x + 6;
    mk[t].c17 = 17.0;
    mk[t].n = 17 - 0.15 * x;
}
// s / d given r = RN(1/d): two residual corrections with fused multiply-adds.  After the first, q is a faithful
// quotient (error ~2u^2 before its rounding); for a faithful q and the correctly rounded reciprocal the second yields the
// correctly rounded quotient (Markstein's theorem; its one exception, a divisor whose significand is all ones, does not
// occur among the divisors used: scalefactors and critical-band widths -- tests/test_emu_parity.py checks them and
// 10^8 quotients incl. the hardest near-midpoint ones).
// No scaling: the encoder's operands are far from the exponent limits.  A zero dividend may come out as +0 where the
// division gives -0; the quantiser adds a non-zero constant next, so no bit depends on it.
TL_FN double tl_div_by(double s, double d, double r)
{
    double q = s * r;
    double e = __builtin_fma(-q, d, s);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-q, d, s);
    return __builtin_fma(e, r, q);
}
// First and last masker with blo < bark <= bhi among the tones [0, ntone) -> a0..a1 and among the noise components
// [ntone, nm) -> b0..b1 (empty: first > last).  Lane-private: called inside a lanes block.
TL_FN void tl_mask_spans(const TlMasker *mk, int nm, int ntone, double blo, double bhi, int &a0, int &a1, int &b0, int &b1)
{
    a0 = nm; a1 = -1; b0 = nm; b1 = -1;
    for (int tb = 0; tb < nm; tb += 32) {                           // 32 maskers -> one hit mask, eight barks per LDS round trip
        uint32_t m = 0;
        for (int t8 = 0; t8 < 32 && tb + t8 < nm; t8 += 8) {
            double mb[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) mb[q] = mk[tb + t8 + q].bark;              // entries past nm (< TL_MASKER_MAX) are masked below
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) m |= (mb[q] > blo && mb[q] <= bhi) ? 1u << (t8 + q) : 0u;
        }
        const int left = nm - tb, tleft = ntone - tb;
        m &= left >= 32 ? ~0u : (1u << left) - 1u;
        const uint32_t tmask = tleft >= 32 ? ~0u : tleft <= 0 ? 0u : (1u << tleft) - 1u;
        const uint32_t mt = m & tmask, mn = m & ~tmask;
        const int ft = tb + __builtin_ctz(mt | 0x80000000u), lt = tb + 31 - __builtin_clz(mt | 1u);
        const int fn = tb + __builtin_ctz(mn | 0x80000000u), ln = tb + 31 - __builtin_clz(mn | 1u);
        a0 = (mt && ft < a0) ? ft : a0; a1 = mt ? lt : a1;
        b0 = (mn && fn < b0) ? fn : b0; b1 = mn ? ln : b1;
    }
}
// The same spans when both lists are ascending in bark -- they are, except after the dead-head replay: tones come in chain order
// (ascending lines), noise components in band order, and bark grows with the line -- by bisection instead of a look at every masker:
// count(B <= v) for v = blo and bhi on the tones mb[0, ntone) and on the noise components mb[ntone, ntone + nnoise), the four
// searches side by side (four reads in flight per level).  STEPS_T / STEPS_N: highest power of two of a count (64: up to 127, 32: up to 63).
// tl_maskers_sorted() decides, wave-uniformly, whether this form may be used.
template <int STEPS_T, int STEPS_N>
TL_FN void tl_mask_spans_sorted(const double *TL_RESTRICT mb, int ntone, int nnoise, double blo, double bhi, int &a0, int &a1, int &b0, int &b1)
{
    int tl = 0, th = 0, nl = 0, nh = 0;
    const double *nbk = mb + ntone;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int step = STEPS_T > STEPS_N ? STEPS_T : STEPS_N; step; step >>= 1) {
        const bool dt = step <= STEPS_T, dn = step <= STEPS_N;
        const int qtl = tl + step, qth = th + step, qnl = nl + step, qnh = nh + step;
        // reads past a list's end stay inside the wave's transform buffer (the masker arrays lie at its start) and are gated by the count tests
        const double vtl = dt ? mb[qtl - 1] : 0.0, vth = dt ? mb[qth - 1] : 0.0;                // (index <= 2 * STEPS_T - 2)
        const double vnl = dn ? nbk[qnl - 1] : 0.0, vnh = dn ? nbk[qnh - 1] : 0.0;
        if (dt) { tl = (qtl <= ntone && vtl <= blo) ? qtl : tl; th = (qth <= ntone && vth <= bhi) ? qth : th; }
        if (dn) { nl = (qnl <= nnoise && vnl <= blo) ? qnl : nl; nh = (qnh <= nnoise && vnh <= bhi) ? qnh : nh; }
    }
    const int nm = ntone + nnoise;
    a0 = th > tl ? tl : nm; a1 = th > tl ? th - 1 : -1;
    b0 = nh > nl ? ntone + nl : nm; b1 = nh > nl ? ntone + nh - 1 : -1;
}
// Are both masker lists ascending in bark?  (wave-uniform; not inside a lanes block)
TL_FN bool tl_maskers_sorted(const double *TL_RESTRICT mb, int ntone, int nnoise)
{
    PV(bool, bad);
    TL_LANES_BEGIN
    bool b = false;
    for (int q = 1 + lane; q < ntone + nnoise; q += 64) b = b || (q != ntone && mb[q] < mb[q - 1]);
    L(bad) = b;
    TL_LANES_END
    return TL_BALLOT(bad) == 0ull;
}
// Running minimum over rows [j0, j0 + n) in the reference's order and with its comparison (`if (m > v) m = v`), four rows
// per LDS round trip; a short last group repeats the last row, which changes nothing.  take_first: m starts as row j0.
TL_FN double tl_min_rows(const double *ltg, int j0, int n, double m, bool take_first)
{
    const int last = j0 + n - 1;
    for (int j = j0; j <= last; j += 4) {
        const double a = ltg[j], b = ltg[j + 1 <= last ? j + 1 : last], c = ltg[j + 2 <= last ? j + 2 : last], d = ltg[j + 3 <= last ? j + 3 : last];
        if (take_first && j == j0) m = a; else if (m > a) m = a;
        if (m > b) m = b;
        if (m > c) m = c;
        if (m > d) m = d;
    }
    return m;
}
// scalefactors transmitted for scfsi 0..3: 3, 2, 1, 2 (encode_new.c:1101, sfsPerScfsi) -- from a constant, not from memory
TL_FN int tl_sfs_count(unsigned scfsi) { return (int)((0x2123u >> (4u * (scfsi & 3u))) & 15u); }
TL_FN unsigned tl_sf_index_ref(const double *TL_RESTRICT sf, double cur_max)
{   // encode_new.c:208-218 as written there (the emulation build checks tl_sf_index against it)
    unsigned i = 32;
    for (unsigned l = 16; l; l >>= 1) { if (cur_max <= sf[i]) i += l; else i -= l; }
    if (cur_max > sf[i]) i--;
    return i;
}
// The same result without the chain of seven dependent table reads.  The table is decreasing, so the search returns
// (number of entries >= cur_max) - 1 (0 when there is none).  Entry i is 2^(1 - i/3) cut to 14 decimals (and [63] = 1e-20): with
// cur_max in [2^e, 2^(e+1)) and i0 = 3(1 - e), every entry above i0 is < 2^e and every entry below i0 - 3 is >= 2^(e+1);
// entry i0 - 3 itself stands for 2^(e+1) but may fall just short of it (the cut), so it is looked at together with the
// three entries in between: four reads, issued together.
TL_FN unsigned tl_sf_index(const double *TL_RESTRICT sf, double cur_max)
{
    const int e = (int)((tl_d2u(cur_max) >> 52) & 0x7ffu) - 1023;
    int i0 = 3 * (1 - e);
    i0 = i0 < 3 ? 3 : i0 > 63 ? 63 : i0;
    const double s3 = sf[i0 - 3], s2 = sf[i0 - 2], s1 = sf[i0 - 1], s0 = sf[i0];
    const int cnt = (i0 - 3) + (cur_max <= s3 ? 1 : 0) + (cur_max <= s2 ? 1 : 0) + (cur_max <= s1 ? 1 : 0) + (cur_max <= s0 ? 1 : 0);
    return (unsigned)(cnt > 0 ? cnt - 1 : 0);
}
TL_FN void tl_put_bits(uint32_t *frame, int pos, uint32_t val, int nbits)
{   // MSB-first bit field at bit offset `pos`; words are big-endian bit order (bitstream.c:130-150)
    if (nbits <= 0) return;
    int w = pos >> 5, o = pos & 31, room = 32 - o;
    if (nbits <= room) TL_ATOMIC_OR(&frame[w], val << (room - nbits));
    else {
        TL_ATOMIC_OR(&frame[w], val >> (nbits - room));
        TL_ATOMIC_OR(&frame[w + 1], val << (32 - (nbits - room)));
    }
}
// The same for a field of 1..48 bits (three codewords of a subband at once), without branches on the field's position:
// the left-aligned value, followed by 32 zero bits, shifted right by the offset inside the first word, is three words.
TL_FN void tl_put_bits48(uint32_t *frame, int pos, uint64_t val, int nbits)
{
    const int w = pos >> 5, o = pos & 31;
    const uint64_t top = val << (64 - nbits);
    TL_ATOMIC_OR(&frame[w], (uint32_t)(top >> (32 + o)));
    TL_ATOMIC_OR(&frame[w + 1], (uint32_t)(top >> o));
    if (o + nbits > 64) TL_ATOMIC_OR(&frame[w + 2], (uint32_t)(((top & 0xffffffffull) << 32) >> o));
}
TL_FN uint32_t tl_get_bit(const uint32_t *frame, int pos) { return (frame[pos >> 5] >> (31 - (pos & 31))) & 1u; }
TL_FN uint32_t tl_bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }
TL_FN unsigned tl_crc_upd(unsigned crc, unsigned data, int len, unsigned poly, unsigned top)
{   // crc.c:43-56 / :99-113
    for (int b = len - 1; b >= 0; b--) {
        unsigned carry = crc & top;
        crc <<= 1;
        if ((!carry) ^ (!((data >> b) & 1u))) crc ^= poly;
    }
    return crc;
}

// ------------------------------------------------------------------------------------------
// K3: 1024-point FHT (fft.c:78-1185), parallel over 64 lanes.  The swap list of fft.c:85-1090 is the 10-bit reversal.
//
// Head in registers: lane L loads the windowed samples i = L + 64*it (it = 0..15); their bit-reversed slots are
// 16*rev6(L) + rev4(it) -- exactly one block of 16 consecutive points, the unit the first pass (groups of four,
// fft.c:1092-1102) and the k=2 pass (fft.c:1104-1184 with k1=4) work on.  So the lane runs both passes on its own sixteen
// values without touching LDS and stores the block once (tl_fht_head / tl_fht_store).
// Layout: logical index i lives at i ^ (i >> 5) (a permutation inside each group of 32 doubles).  With it the stored
// blocks, the k=4/6/8 butterflies and the energy reads spread over the LDS banks (at most ~3 lanes per bank instead of
// up to 32); index fields that do not share bits pass through the map separately: FX(a|b) = FX(a) ^ FX(b).
#define TL_FX(i) ((i) ^ ((i) >> 5))
TL_FN void tl_fht_head(double (&e)[16], const double (*TL_RESTRICT tw)[4])
{
    const double SQRT2 = 1.4142135623730951454746218587388284504414;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int g = 0; g < 16; g += 4) {                               // fft.c:1092-1102
        const double f1 = e[g] - e[g + 1], f0 = e[g] + e[g + 1], f3 = e[g + 2] - e[g + 3], f2 = e[g + 2] + e[g + 3];
        e[g + 2] = f0 - f2; e[g] = f0 + f2; e[g + 3] = f1 - f3; e[g + 1] = f1 + f3;
    }
    {   // k=2 pass, i = 0: fi = block, gi = block + 2 (k1 = 4, k2 = 8, k3 = 12)
        const double f1 = e[0] - e[4], f0 = e[0] + e[4], f3 = e[8] - e[12], f2 = e[8] + e[12];
        e[8] = f0 - f2; e[0] = f0 + f2; e[12] = f1 - f3; e[4] = f1 + f3;
        const double g1 = e[2] - e[6], g0 = e[2] + e[6], g3 = SQRT2 * e[14], g2 = SQRT2 * e[10];
        e[10] = g0 - g2; e[2] = g0 + g2; e[14] = g1 - g3; e[6] = g1 + g3;
    }
    {   // k=2 pass, i = 1: fi = block + 1, gi = block + 3; one twiddle set for every block
        const double c1 = tw[0][0], s1 = tw[0][1], c2 = tw[0][2], s2 = tw[0][3];
        double a, b2, g0, f0, f1, g1, f2, g2, f3, g3;
        b2 = s2 * e[5] - c2 * e[7]; a = c2 * e[5] + s2 * e[7];
        f1 = e[1] - a; f0 = e[1] + a; g1 = e[3] - b2; g0 = e[3] + b2;
        b2 = s2 * e[13] - c2 * e[15]; a = c2 * e[13] + s2 * e[15];
        f3 = e[9] - a; f2 = e[9] + a; g3 = e[11] - b2; g2 = e[11] + b2;
        b2 = s1 * f2 - c1 * g3; a = c1 * f2 + s1 * g3;
        e[9] = f0 - a; e[1] = f0 + a; e[15] = g1 - b2; e[7] = g1 + b2;
        b2 = c1 * g2 - s1 * f3; a = s1 * g2 + c1 * f3;
        e[11] = g0 - a; e[3] = g0 + a; e[13] = f1 - b2; e[5] = f1 + b2;
    }
}
TL_FN int tl_rev6(int lane) { int r = 0; for (int b = 0; b < 6; b++) r |= ((lane >> b) & 1) << (5 - b); return r; }
TL_FN void tl_fht_store(double *x, int lane, const double (&e)[16])
{
    const int l = tl_rev6(lane), base = (16 * l) ^ (l >> 1);         // FX(16*l + t) = base ^ t for t < 16
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int t = 0; t < 16; t++) x[base ^ t] = e[t];
}
// Twiddles (c1,s1,c2,s2) of the (up to) two general butterflies a lane runs in pass K; fetched one pass ahead.
template <int K>
TL_FN void tl_fht_twiddles(double (&t)[8], const TlTables *TL_RESTRICT T, int lane)
{   // rows in lane order (TlTables::fht_tw_lane): one address per lane, no index arithmetic
    const double (*tw)[4] = T->fht_tw_lane[(K - 4) / 2];
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int q = 0; q < 4; q++) t[4 * it + q] = tw[lane + 64 * it][q];
}
template <int K>
TL_FN void tl_fht_pass(double *x, const double (&t)[8], int lane)
{   // fft.c:1104-1184: one pass = 128 independent 8-point butterflies: per block of 4*k1 points one with trivial /
    // sqrt(2) twiddles (i = 0) and kx-1 general ones.  The general ones are dealt densely to the lanes and the
    // trivial ones follow in their own step, so a wave never runs both code paths for one batch of butterflies.
    // Addresses: block, i (or k1-i) and q*k1 occupy disjoint bit fields, so each goes through TL_FX on its own.
    const double SQRT2 = 1.4142135623730951454746218587388284504414;
    constexpr int k1 = 1 << K, k2 = k1 << 1, k4 = k2 << 1, k3 = k2 + k1, kx = k1 >> 1;
    constexpr int NBLK = 128 / kx, NGEN = 128 - NBLK;
    constexpr int q1 = TL_FX(k1), q2 = TL_FX(k2), q3 = TL_FX(k3);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int g = lane + 64 * it;
        if (g >= NGEN) break;
        const int blk = g / (kx - 1), i = 1 + (g - blk * (kx - 1));
        const int pb = TL_FX(blk * k4);
        const double c1 = t[4 * it], s1 = t[4 * it + 1], c2 = t[4 * it + 2], s2 = t[4 * it + 3];
        const int F = pb ^ TL_FX(i), G = pb ^ TL_FX(k1 - i);
        double *f0p = x + F, *f1p = x + (F ^ q1), *f2p = x + (F ^ q2), *f3p = x + (F ^ q3);
        double *g0p = x + G, *g1p = x + (G ^ q1), *g2p = x + (G ^ q2), *g3p = x + (G ^ q3);
        double a, b2, g0, f0, f1, g1, f2, g2, f3, g3;
        b2 = s2 * *f1p - c2 * *g1p; a = c2 * *f1p + s2 * *g1p;
        f1 = *f0p - a; f0 = *f0p + a; g1 = *g0p - b2; g0 = *g0p + b2;
        b2 = s2 * *f3p - c2 * *g3p; a = c2 * *f3p + s2 * *g3p;
        f3 = *f2p - a; f2 = *f2p + a; g3 = *g2p - b2; g2 = *g2p + b2;
        b2 = s1 * f2 - c1 * g3; a = c1 * f2 + s1 * g3;
        *f2p = f0 - a; *f0p = f0 + a; *g3p = g1 - b2; *g1p = g1 + b2;
        b2 = c1 * g2 - s1 * f3; a = s1 * g2 + c1 * f3;
        *g2p = g0 - a; *g0p = g0 + a; *f3p = f1 - b2; *f1p = f1 + b2;
    }
    if (lane < NBLK) {
        const int F = TL_FX(lane * k4), G = F ^ TL_FX(kx);
        double *f0p = x + F, *f1p = x + (F ^ q1), *f2p = x + (F ^ q2), *f3p = x + (F ^ q3);
        double *g0p = x + G, *g1p = x + (G ^ q1), *g2p = x + (G ^ q2), *g3p = x + (G ^ q3);
        double f1 = *f0p - *f1p, f0 = *f0p + *f1p, f3 = *f2p - *f3p, f2 = *f2p + *f3p;
        *f2p = f0 - f2; *f0p = f0 + f2; *f3p = f1 - f3; *f1p = f1 + f3;
        double g1 = *g0p - *g1p, g0 = *g0p + *g1p, g3 = SQRT2 * *g3p, g2 = SQRT2 * *g2p;
        *g2p = g0 - g2; *g0p = g0 + g2; *g3p = g1 - g3; *g1p = g1 + g3;
    }
}

// Hann window of samples [t-192, t+832) + FHT + energy (psycho_1.c:57-76,215-239, fft.c:1278-1293).
// Leaves energy[i] in w.u.fft[TL_EX(i)], i = 0..512.
// A stream's PCM as the kernel sees it in HBM: this frame (planar [2][1152]) and the 480 samples per
// channel that precede it (the stream state on the first frame of a launch, the previous input frame after).
// Per "channel" c of the wave: the two channels of a stereo stream -- or, for a PAIR of mono streams sharing a wave (tl_frame_unit),
// channel 0 of each of the two streams.
struct TlPcmView { const int16_t *cur[2]; const int16_t *hist[2]; };

TL_FN void tl_psy_spectrum(TlPsyLds &w, const TlTables *TL_RESTRICT T, const TlPcmView &pv, int ch, long long *sp)
{
    double *x = w.u.fft;
    long long *sq = (sp && ch == 0) ? sp + 16 : nullptr;      // channel 0's pass-by-pass stamps: slots 24..30 of the frame's record
    TL_STAMP(sq, 0);
    // twiddles travel one pass ahead of their use (twc: k=4 with the window, twb: k=6 during pass 4, twa: k=8 during pass 6): two sets live at most
    PA(double, twa, 8); PA(double, twb, 8); PA(double, twc, 8);
    TL_LANES_BEGIN
    {
        // sample i = lane + 64*it of the analysis window: the last 192 samples of the history (it < 3), then the
        // first 832 of the frame.  The loads are issued in batches ahead of their use.  Slot of i inside the lane's block
        // of sixteen: rev4(it).
        const int16_t *hs = (ch ? pv.hist[1] : pv.hist[0]) + (TL_HIST - 192) + lane;      // (a select, not an indexed array: that would live in scratch)
        const int16_t *cs = (ch ? pv.cur[1] : pv.cur[0]) - 192 + lane;
        const double *hann = T->hann;
        TL_LAUNDER(hann);
        tl_fht_twiddles<4>(L(twc), T, lane);
        double e[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int half = 0; half < 16; half += 8) {                  // eight loads in flight (sixteen would spill)
            int16_t v[8]; double h[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) { const int it = half + q; v[q] = it < 3 ? hs[64 * it] : cs[64 * it]; h[q] = hann[lane + 64 * it]; }
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) {
                const int it = half + q;
                const int r4 = ((it & 1) << 3) | ((it & 2) << 1) | ((it & 4) >> 1) | ((it & 8) >> 3);
                e[r4] = ((double)v[q] / 32768) * h[q];
            }
        }
        tl_fht_head(e, T->fht_tw);
        tl_fht_store(x, lane, e);
    }
    TL_LANES_END
    TL_STAMP(sq, 1);
    TL_STAMP(sq, 2);
    TL_STAMP(sq, 3);
    TL_LANES_BEGIN tl_fht_twiddles<6>(L(twb), T, lane); tl_fht_pass<4>(x, L(twc), lane); TL_LANES_END
    TL_STAMP(sq, 4);
    TL_LANES_BEGIN tl_fht_twiddles<8>(L(twa), T, lane); tl_fht_pass<6>(x, L(twb), lane); TL_LANES_END
    TL_STAMP(sq, 5);
    // Last pass (k=8) and energies (fft.c:1278-1293) in one go: the eight outputs of a k=8 butterfly are x[i+256q] and
    // x[256-i+256q], and line j pairs with 1024-j -- so butterfly i holds both members of the pairs of lines i, 256-i, 256+i
    // and 512-i (the trivial butterfly: lines 0, 128, 256, 384, 512).  The transform is never written back: every input is
    // read first (the energies go to natural positions, which are other lanes' inputs), then each lane squares its own pairs.
    {
        constexpr int k1 = 256, kx = 128;
        constexpr int q1 = TL_FX(256), q2 = TL_FX(512), q3 = TL_FX(768);
        PA(double, fv, 8); PA(double, gv, 8);
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 2; it++) {
            const int g = lane + 64 * it;                            // general butterflies i = 1 + g (g < 127); g = 127: the trivial one
            const int F = g < 127 ? TL_FX(1 + g) : 0, G = g < 127 ? TL_FX(k1 - 1 - g) : TL_FX(kx);
            L(fv)[4 * it] = x[F]; L(fv)[4 * it + 1] = x[F ^ q1]; L(fv)[4 * it + 2] = x[F ^ q2]; L(fv)[4 * it + 3] = x[F ^ q3];
            L(gv)[4 * it] = x[G]; L(gv)[4 * it + 1] = x[G ^ q1]; L(gv)[4 * it + 2] = x[G ^ q2]; L(gv)[4 * it + 3] = x[G ^ q3];
        }
        TL_LANES_END
        TL_LANES_BEGIN
        const double SQRT2 = 1.4142135623730951454746218587388284504414;
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 2; it++) {
            const int g = lane + 64 * it;
            const double fi0 = L(fv)[4 * it], fi1 = L(fv)[4 * it + 1], fi2 = L(fv)[4 * it + 2], fi3 = L(fv)[4 * it + 3];
            const double gi0 = L(gv)[4 * it], gi1 = L(gv)[4 * it + 1], gi2 = L(gv)[4 * it + 2], gi3 = L(gv)[4 * it + 3];
            if (g < 127) {
                const int i = 1 + g;
                const double c1 = L(twa)[4 * it], s1 = L(twa)[4 * it + 1], c2 = L(twa)[4 * it + 2], s2 = L(twa)[4 * it + 3];
                double a, b2, g0, f0, f1, g1, f2, g2, f3, g3;
                b2 = s2 * fi1 - c2 * gi1; a = c2 * fi1 + s2 * gi1;
                f1 = fi0 - a; f0 = fi0 + a; g1 = gi0 - b2; g0 = gi0 + b2;
                b2 = s2 * fi3 - c2 * gi3; a = c2 * fi3 + s2 * gi3;
                f3 = fi2 - a; f2 = fi2 + a; g3 = gi2 - b2; g2 = gi2 + b2;
                b2 = s1 * f2 - c1 * g3; a = c1 * f2 + s1 * g3;
                const double o_f2 = f0 - a, o_f0 = f0 + a, o_g3 = g1 - b2, o_g1 = g1 + b2;     // x[i+512], x[i], x[1024-i], x[512-i]
                b2 = c1 * g2 - s1 * f3; a = s1 * g2 + c1 * f3;
                const double o_g2 = g0 - a, o_g0 = g0 + a, o_f3 = f1 - b2, o_f1 = f1 + b2;     // x[768-i], x[256-i], x[i+768], x[i+256]
                // E[j] = (x[j]^2 + x[1024-j]^2) / 2 with a = x[j] first, as in the reference
                x[TL_EX(i)] = (o_f0 * o_f0 + o_g3 * o_g3) / 2.0;
                x[TL_EX(256 - i)] = (o_g0 * o_g0 + o_f3 * o_f3) / 2.0;
                x[TL_EX(256 + i)] = (o_f1 * o_f1 + o_g2 * o_g2) / 2.0;
                x[TL_EX(512 - i)] = (o_g1 * o_g1 + o_f2 * o_f2) / 2.0;
            } else if (g == 127) {
                double f1 = fi0 - fi1, f0 = fi0 + fi1, f3 = fi2 - fi3, f2 = fi2 + fi3;
                const double o_f2 = f0 - f2, o_f0 = f0 + f2, o_f3 = f1 - f3, o_f1 = f1 + f3;     // x[512], x[0], x[768], x[256]
                double g1 = gi0 - gi1, g0 = gi0 + gi1, g3 = SQRT2 * gi3, g2 = SQRT2 * gi2;
                const double o_g2 = g0 - g2, o_g0 = g0 + g2, o_g3 = g1 - g3, o_g1 = g1 + g3;     // x[640], x[128], x[896], x[384]
                x[0] = o_f0 * o_f0;                                   // TL_EX leaves multiples of 256 where they are
                x[512] = o_f2 * o_f2;
                x[256] = (o_f1 * o_f1 + o_f3 * o_f3) / 2.0;
                x[TL_EX(128)] = (o_g0 * o_g0 + o_g3 * o_g3) / 2.0;
                x[TL_EX(384)] = (o_g1 * o_g1 + o_g2 * o_g2) / 2.0;
            }
        }
        TL_LANES_END
    }
    TL_STAMP(sq, 6);
}

// ------------------------------------------------------------------------------------------
// Candidate record used by the tone labelling of psy 1 and psy 3: bits 0-8 line index, bits 21.. the line's run.  Only candidates
// whose RIGHT neighbours pass are recorded: the right side of a candidate is never touched by an earlier tone (the erasure reach
// of every earlier tone ends below the candidate), so that half of the test is decided in parallel from the original spectrum,
// for all 500 lines.  The left half depends on which earlier candidates were confirmed: the walk that follows reads the left
// neighbours of the CANDIDATES (tl_cand_left: "neighbour j fails" as bit j - 2) and resolves them against its state.

// power density in dB of one line (psycho_1.c:241-248, psycho_3.c:152-160), straight-line so that several lines' logarithms
// (long dependent chains) can be in flight together
// The logarithm is glibc 2.35's own (tl_libm.h: table-driven log, no division, then e_log10.c's recombination) -- bit-equal to
// the reference's libm by construction.  That matters: on degenerate spectra (a lone impulse: hundreds of lines of nearly
// equal level) the tone tests and the allocation compare values that differ in the last bits (GPU soak, round 2).
#define TL_LOGTAB(db) ((const uint64_t *)((db) + 1002))     /* the log table rides behind the dB-sum table in the workgroup's LDS block (TlTables::dblog) */
// The floor (`energy < 1E-20 ? -200 + POWERNORM : ...`, psycho_1.c:243-246) is one maximum: log10 of the double 1E-20 is exactly -20.0
// in glibc and in its restatement here (tests/test_libm_agree.py pins it), so 10 * log10(max(e, 1E-20)) + POWERNORM is the reference's
// value on either side of the test -- (-200.0 + POWERNORM) is the same sum -- and the argument of the logarithm is always normal.
TL_FN double tl_power_db(double e, const uint64_t *lt)
{
    return 10 * tlm_log10_pn(__builtin_fmax(e, 1E-20), lt) + TL_POWERNORM;
}
TL_FN int tl_run_psy1(int c) { return (c < 3 || c > 500) ? 0 : c < 63 ? 2 : c < 127 ? 3 : c < 255 ? 6 : 12; }   // psycho_1.c:289-298
TL_FN int tl_run_psy3(int c) { return c < 63 ? 2 : c < 127 ? 3 : c < 255 ? 6 : 12; }                            // psycho_3.c:212-215

// Tone candidates of one 64-line chunk: local maxima 2..499 whose right-hand neighbours (distance 2..run) pass the
// 7 dB test; the left-hand failures are recorded as a bit mask for the walk that follows (psycho_1.c:267-300,
// psycho_3.c:186-236).  RMAX is the largest run inside the chunk, so the neighbour reads are straight-line code
// and overlap; PSY3 selects psycho_3's strict maximum and its (peak - neighbour) < 7 form of the test.
template <int RMAX, bool PSY3>
TL_FN void tl_cand_chunk(TlPsyLds &w, int c8, int &ncand)
{
    const double *px = TL_PX(w);
    PV(bool, isc); PV(uint32_t, rec);
    TL_LANES_BEGIN
    const int i = 64 * c8 + lane - 1;                               // chunks start one line early: the run lengths change at 63, 127, 255
    const bool inr = i >= 2 && i < 500;
    const int ii = inr ? i : 16;
    // every neighbour is read before the first test (TL_KEEP: otherwise the compiler reads each one only if the
    // tests so far passed -- a chain of dependent LDS round trips)
    double a[RMAX + 1];
    const double pk = px[ii], b1 = px[ii - 1];
#pragma unroll
    for (int j = 1; j <= RMAX; j++) a[j] = px[ii + j];
#pragma unroll
    for (int j = 1; j <= RMAX; j++) TL_KEEP(a[j]);
    bool cnd = inr && pk > b1 && (PSY3 ? pk > a[1] : pk >= a[1]);
    // the run of a line inside the chunk is the chunk's RMAX (the chunks are cut where the run changes), or 0 (psycho_1's lines
    // below 3): one per-lane flag instead of a `j <= run` per neighbour
    const int run = PSY3 ? tl_run_psy3(ii) : tl_run_psy1(ii);
    const bool has = run != 0;
    const double max = pk - 7;
    bool fail = false;
#pragma unroll
    for (int j = 2; j <= RMAX; j++) fail = fail || (PSY3 ? (pk - a[j]) < 7.0 : max < a[j]);
    cnd = cnd && !(has && fail);
    // The LEFT-hand neighbours are not looked at here: what they decide depends on the walk, and the walk looks at them for the
    // candidates alone (tl_cand_left) -- a few dozen lines instead of five hundred.
    L(isc) = cnd; L(rec) = (uint32_t)i | ((uint32_t)run << 21);     // line | run << 21
    TL_LANES_END
    const uint64_t m = TL_BALLOT(isc);
    TL_LANES_BEGIN
    if ((m >> lane) & 1ull) w.cinfo[ncand + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = L(rec);
    TL_LANES_END
    ncand += __builtin_popcountll(m);
}
// "left neighbour j fails the 7 dB test" for j = 2..run of candidate line c with level pk, bit j - 2 (psycho_1.c:289-300,
// psycho_3.c:217-226), from the still-original spectrum.  All eleven neighbours are read whatever the run is (px[] sits behind
// the transform buffer: c - 12 is inside the wave's block for every c >= 2, and what lies there are finite energies); the bits
// beyond the run are masked off.
template <bool PSY3>
TL_FN uint32_t tl_cand_left(const double *px, int c, int run, double pk)
{
    double b[11];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int j = 2; j <= 12; j++) b[j - 2] = px[c - j];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int j = 2; j <= 12; j++) TL_KEEP(b[j - 2]);
    const double max = pk - 7;
    uint32_t lf = 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int j = 2; j <= 12; j++) lf |= (PSY3 ? (pk - b[j - 2]) < 7.0 : max < b[j - 2]) ? 1u << (j - 2) : 0u;
    return lf & (run >= 2 ? (1u << (run - 1)) - 1u : 0u);
}

// psy model 1 (psycho_1.c:22-87, :215-581); result in w.smr[ch][0..sblimit).
//
// Per channel the model is a FRONT (spectrum, power, tone labelling, compaction of the lines each critical band sums),
// the per-band dB-sum CHAINS (sequential by definition: up to 164 dependent table look-ups in the widest band, on 27
// lanes) and a BACK (band centres, decimation, thresholds, SMR).  For two channels the chains of both run side by side
// on the two halves of the wave (tl_psy1_stereo): channel 0's front results wait in registers while channel 1's front
// uses the LDS arrays.
struct TlPsy1Ch { int nconf, nlist; bool dead_head; };

TL_FN TlPsy1Ch tl_psy1_front(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                             const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const double *energy = w.u.fft;                                   // line i at TL_EX(i)
    double *px = TL_PX(w);
    TL_PRIO2(TL_PS_FHT);
    TL_STAMP(sp, 0);
    if (TL_EXP_LEVEL < 8) tl_psy_spectrum(w, T, pv, ch, sp);
    TL_STAMP(sp, 1);
    TL_PRIO2(TL_PS_POW);

    // power density spectrum (psycho_1.c:241-248); spike (psycho_1.c:252-257)
    // The spike sums read 16 consecutive energies per lane; the energies' XOR layout keeps those reads off each other's
    // LDS banks.  Only subbands below sblimit (<= 30) are ever used.
    TL_LANES_BEGIN
    for (int i0 = lane; i0 < (TL_EXP_LEVEL >= 7 ? 0 : 512); i0 += 256) {                     // four lines per lane at a time
        double e[4], v[4];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) e[q] = energy[TL_EX(i0 + 64 * q)];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) v[q] = tl_power_db(e[q], TL_LOGTAB(db));
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) {
            const int i = i0 + 64 * q;
            px[i] = v[q];
            w.ptype[i] = 0;
        }
    }
    TL_LANES_END
    TL_LANES_BEGIN
    if (lane < (TL_EXP_LEVEL >= 7 ? 0 : 30)) {
        double e[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int j = 0; j < 16; j++) e[j] = energy[16 * lane + (j ^ (lane & 15))];      // == energy[TL_EX(16 * lane + j)]
        double sum = 1E-20;
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int j = 0; j < 16; j++) sum += 1073741824 * e[j];
        const double spk = 10.0 * tlm_log10_pn(sum, TL_LOGTAB(db));
        L(rec)[ch] = spk;                                           // final as it is: straight to the record (nothing to park)
    } else if (lane < 32) L(rec)[ch] = 0.0;
    TL_LANES_END
    TL_PRIO2(1);                                                      // from here to the thresholds: serial stages

    // ---- tonal components (psycho_1.c:267-340) ----
    // (1) local maxima 2..499 whose right-hand neighbours pass the 7 dB test, compacted ascending
    int ncand = 0;
    if (TL_EXP_LEVEL < 6) {
    tl_cand_chunk<2, false>(w, 0, ncand);                           // lines -1..62: run 2
    tl_cand_chunk<3, false>(w, 1, ncand);                           // 63..126: run 3
    for (int c8 = 2; c8 < 4; c8++) tl_cand_chunk<6, false>(w, c8, ncand);     // 127..254: run 6
    for (int c8 = 4; c8 < 8; c8++) tl_cand_chunk<12, false>(w, c8, ncand);    // 255..510: run 12
    }
    TL_STAMP(sp, 2);
    // (2) which candidates become tones.  The reference walks its list once, in line order; what a candidate's fate depends on is the
    //     walk's state -- the last confirmed tone (`last`), its run and whether its left neighbour was erased -- and that state only
    //     changes at a CONFIRMATION.  So instead of one scalar iteration per candidate (sixty dependent scalar instructions each: the
    //     walk was 9 % of a frame's cycles with every lane idle) the candidates sit in lanes and every round evaluates all of them
    //     against the current state at once: the first one that passes is the next confirmed tone -- every candidate before it was
    //     rejected under the same state, exactly as the sequential walk rejects them -- the state moves, the lanes after it go on.
    //     One round per confirmed tone (+ 1) instead of one iteration per candidate.  The bookkeeping of the reference's list is
    //     kept: the erasure reach R = last + run(last), the last_but_one relinking (psycho_1.c:313-316).
    int nconf = 0;
    bool any_erased;                                                  // a confirmed tone erased its predecessor: the chain order is not 0..nconf-1
    {
        int last = -1, run_last = 0, last_var = 0;
        any_erased = false;
        for (int kb = 0; kb < (TL_EXP_LEVEL >= 5 ? 0 : ncand); kb += 64) {          // 64 candidates per pass (there are rarely more)
        PV(int, cc); PV(int, crun); PV(int, clf); PV(double, cpx); PV(bool, act); PV(bool, dep);
        TL_LANES_BEGIN
        const bool in = kb + lane < ncand;
        const uint32_t info = w.cinfo[in ? kb + lane : 0];
        const uint32_t pinfo = w.cinfo[in && kb + lane > 0 ? kb + lane - 1 : 0];    // the candidate before this one
        L(cc) = (int)(info & 511u); L(crun) = (int)(info >> 21);
        L(cpx) = px[L(cc)];
        L(clf) = (int)tl_cand_left<false>(px, L(cc), L(crun), L(cpx));
        L(act) = in;
        // A candidate at least run(previous candidate) + run(its own) + 1 lines above the candidate before it is out of every earlier
        // tone's reach whatever the walk's state is when it gets there: `last` is at or below that previous candidate and runs grow
        // with the line, so d - run_last - 1 >= run, every one of its left neighbours is original, and its fate is lfail == 0 -- the
        // value the test below gives it under ANY earlier state (d > run, so neither the summed-level test nor an erasure applies,
        // and its left neighbour is not the end of a reach: var = 0).  Such candidates need no round of their own.
        L(dep) = in && kb + lane > 0 && L(cc) - (int)(pinfo & 511u) < (int)(pinfo >> 21) + L(crun) + 1;
        TL_LANES_END
        for (;;) {
            PV(bool, okv); PV(bool, needx);
            TL_LANES_BEGIN
            const int c = L(cc), run = L(crun);
            const uint32_t lfail = (uint32_t)L(clf);
            bool ok, nx = false;
            if (last < 0) ok = lfail == 0;
            else {
                // neighbours c-j <= R were erased to DBMIN by `last` (they pass), except `last` itself,
                // which carries its summed level; neighbours above R (or below last-run_last) are original
                // bit j-2 set for j in [2, run] with c-j > R (j <= c-R-1) or c-j < last-run_last (j >= c-last+run_last+1)
                const int d = c - last;
                const int hi_j = run < d - run_last - 1 ? run : d - run_last - 1;
                const int lo_j = d + run_last + 1 > 2 ? d + run_last + 1 : 2;
                uint32_t orig = hi_j >= 2 ? (1u << (hi_j - 1)) - 1u : 0u;
                if (lo_j <= run) orig |= ((1u << (run - 1)) - 1u) & ~((1u << (lo_j - 2)) - 1u);
                ok = d > run_last && !(lfail & orig);                // d <= run_last: unlinked by the help loop, psycho_1.c:309-312
                nx = ok && d >= 2 && d <= run;                       // then `last` itself is among its neighbours, with its summed level
            }
            L(okv) = ok && L(act); L(needx) = nx && L(act);
            TL_LANES_END
            if (TL_BALLOT(needx) != 0ull) {                           // rare: a candidate within its run of the last tone
                TL_LANES_BEGIN
                const double xl = tl_add_db(db, px[last], tl_add_db(db, last_var ? TL_DBMIN : px[last - 1], px[last + 1]));
                if (L(needx) && L(cpx) - 7 < xl) L(okv) = false;
                TL_LANES_END
            }
            const uint64_t m = TL_BALLOT(okv);
            if (m == 0ull) break;                                     // everything left of this pass is rejected (only unlinked, psycho_1.c:330-338)
            const int wl = __builtin_ctzll(m);
            const int c = TL_READLANE_I32(cc, wl), run = TL_READLANE_I32(crun, wl);
            // confirmed.  Its left neighbour c-1 was erased iff it is exactly the end of `last`'s reach.
            const int var = (last >= 0 && run_last >= 1 && c - 1 == last + run_last) ? 1 : 0;
            // With it, in the same round: every passing candidate between it and the next state-DEPENDENT candidate still to be
            // decided (see `dep` above) -- their verdicts under the state of this round are their verdicts under any state.
            PV(bool, depact);
            TL_LANES_BEGIN L(depact) = L(dep) && L(act) && lane > wl; TL_LANES_END
            const uint64_t dm = TL_BALLOT(depact);
            const uint64_t upto = dm ? (1ull << __builtin_ctzll(dm)) - 1ull : ~0ull;     // lanes below the next dependent one
            const uint64_t batch = m & upto;                           // wl and the independent passing candidates after it
            const int nb = __builtin_popcountll(batch), wlast = 63 - __builtin_clzll(batch);
            const int i0 = nconf;
            TL_LANES_BEGIN
            if ((batch >> lane) & 1ull) {
                const int i = i0 + __builtin_popcountll(batch & ((1ull << lane) - 1ull));
                if (i < TL_TONE_MAX) {
                    w.conf_c[i] = (int16_t)(L(cc) | ((lane == wl ? var : 0) << 12));
                    w.conf_nxt[i] = (int16_t)((lane == wlast || i + 1 >= TL_TONE_MAX) ? TL_LAST : i + 1);
                }
            }
            TL_LANES_END
            if (i0 < TL_TONE_MAX && i0 > 0) {                         // the round's first tone against the tone before it
                if (c - last <= run) {                                // erases the previous tone, psycho_1.c:313-316,322-326
                    any_erased = true;
                    w.conf_nxt[i0 - 1] = TL_STOP;
                    w.conf_c[i0 - 1] = (int16_t)(w.conf_c[i0 - 1] | (1 << 13));
                    if (i0 >= 2) w.conf_nxt[i0 - 2] = (int16_t)i0;
                } else w.conf_nxt[i0 - 1] = (int16_t)i0;
            }
            nconf = i0 + nb < TL_TONE_MAX ? i0 + nb : (i0 < TL_TONE_MAX ? TL_TONE_MAX : i0);
            last = TL_READLANE_I32(cc, wlast); run_last = TL_READLANE_I32(crun, wlast); last_var = wlast == wl ? var : 0;
            TL_LANES_BEGIN L(act) = L(act) && lane > wlast; TL_LANES_END
            TL_DBG_ROUND();
        }
        }
        TL_SYNC();
    }
    // (3) levels of the confirmed tones from the still-original spectrum (psycho_1.c:317-321)
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {
        const int cc = w.conf_c[i], c = cc & 511, var = (cc >> 12) & 1;
        w.tone_x[i] = tl_add_db(db, px[c], tl_add_db(db, var ? TL_DBMIN : px[c - 1], px[c + 1]));
    }
    TL_LANES_END
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) { const int c = w.conf_c[i] & 511; px[c] = w.tone_x[i]; w.ptype[c] = TL_T_TONE; }
    TL_LANES_END
    // (4) erasures (psycho_1.c:322-326); a tone erased by its successor ends up DBMIN / not TONE
    //     Straight-line: a tone's run is 2, 3, 6 or 12 (it is a line 3..499), so the stores are four nested groups behind three tests,
    //     each store at a constant offset from ONE address per array (px + c - 12 is inside the wave's block: px[] lies behind the
    //     transform) -- instead of a loop of `run` trips with four address computations each.
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {
        const int c = w.conf_c[i] & 511, run = tl_run_psy1(c);
        double *pb = px + (c - 12);
        uint8_t *tb = w.ptype;
#define TL_ERASE1(j) do { pb[12 - (j)] = TL_DBMIN; pb[12 + (j)] = TL_DBMIN; tb[c - (j)] = 0; tb[c + (j)] = 0; } while (0)
        if (run >= 2) {
            TL_ERASE1(1); TL_ERASE1(2);
            if (run >= 3) {
                TL_ERASE1(3);
                if (run >= 6) {
                    TL_ERASE1(4); TL_ERASE1(5); TL_ERASE1(6);
                    if (run >= 12) { TL_ERASE1(7); TL_ERASE1(8); TL_ERASE1(9); TL_ERASE1(10); TL_ERASE1(11); TL_ERASE1(12); }
                }
            }
        }
#undef TL_ERASE1
    }
    TL_LANES_END
    // (5) the tone list in chain order (psycho_1.c list head `*tone`): walk the links, then decimate
    //     in parallel (psycho_1.c:416-428): drop erased tones and tones below the threshold in quiet
    int nlist = 0;
    if (!any_erased) {                                                // every link points to the next tone: the chain is 0..nconf-1
        nlist = nconf;
        TL_LANES_BEGIN
        for (int i = lane; i < nconf; i += 64) w.tlist[i] = (int16_t)i;
        TL_LANES_END
    } else {
        PV(int, nx0); PV(int, nx1);                                   // the links in registers: the walk reads lanes, not LDS
        TL_LANES_BEGIN L(nx0) = w.conf_nxt[lane]; L(nx1) = w.conf_nxt[64 + lane < TL_TONE_MAX ? 64 + lane : 0]; TL_LANES_END
        int i = nconf ? 0 : TL_LAST, guard = 0;
        while (i != TL_LAST && i != TL_STOP && guard++ < TL_TONE_MAX) {
            w.tlist[nlist++] = (int16_t)i;
            i = i < 64 ? TL_READLANE_I32(nx0, i) : TL_READLANE_I32(nx1, i - 64);
        }
        TL_SYNC();
    }
    TL_STAMP(sp, 3);

    // ---- noise components (psycho_1.c:356-376) ----
    // Line-parallel preparation: the lines a band will actually sum (not tonal, not erased) are compacted in
    // ascending order together with their weight terms, so the sequential part is a bare dB-sum chain.
    // vt[] overwrites the energies and vp[] the power spectrum in place (a compacted position is never above its line, and a
    // chunk of 64 lines is read completely before its entries are written).  power[] is gone after this: the dead-head
    // replay rebuilds what it reads (tl_psy1_deadhead).
    const int nbands = C->p1_ncb - 1;
    {
        double *vt = w.u.fft, *vp = px;
        int nvalid = 0;
        PA(uint32_t, linfo, 8); PA(double, lrw, 8);                 // the table reads of all eight chunks in one batch
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int c8 = 0; c8 < 8; c8++) { L(linfo)[c8] = C->p1_lineinfo[64 * c8 + lane]; L(lrw)[c8] = C->p1_linerw[64 * c8 + lane]; }
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int base = 0; base < (TL_EXP_LEVEL >= 4 ? 0 : 512); base += 64) {
            PV(bool, ok); PV(double, tv); PV(double, pvv); PV(int, bnd);
            TL_LANES_BEGIN
            const int j = base + lane;
            const uint32_t info = L(linfo)[base >> 6];
            const int lo = (int)((info >> 8) & 0xfffu), hi = (int)(info >> 20);
            bool v = false; double t = 0, p = 0;
            if (info) {                                             // line inside the bands
                p = px[j];
                v = w.ptype[j] != TL_T_TONE && p != TL_DBMIN;
                t = tl_div_by(1073741824 * energy[TL_EX(j)] * (double)(j - lo), (double)(hi - lo), L(lrw)[base >> 6]);   // == num / (hi - lo)
            }
            L(ok) = v; L(tv) = t; L(pvv) = p; L(bnd) = (info && j == lo) ? (int)(info & 0xffu) : -1;
            TL_LANES_END
            const uint64_t m = TL_BALLOT(ok);
            TL_LANES_BEGIN
            const int pos = nvalid + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (L(bnd) >= 0) w.bandoff[L(bnd)] = (int16_t)pos;     // first line of its band
            if (L(ok)) { vt[pos] = L(tv); vp[pos] = L(pvv); }
            TL_LANES_END
            nvalid += __builtin_popcountll(m);
        }
        w.bandoff[nbands] = (int16_t)nvalid;
        TL_SYNC();
    }
    TlPsy1Ch r;
    r.nconf = nconf; r.nlist = nlist;
    // The reference keeps tones and noise components in ONE linked list field (power[].next).  The two chains only interact
    // when the head of the tone chain is a tone that was erased by its successor (psycho_1.c:313-316 with last_but_one ==
    // LAST): its line is no longer TONE, so a noise centre may land on it and splice the noise chain into the tone chain.
    // That (rare) case is replayed pointer by pointer (tl_psy1_deadhead); otherwise the chains are independent.
    r.dead_head = nconf > 0 && ((w.conf_c[0] >> 13) & 1);
    return r;
}

// weight sums of the bands (psycho_1.c:364-366), ascending line order; lane b < nbands owns band b.  Only used where the
// weights cannot ride along with the dB-sum chain (channel 0 of a stereo frame, whose terms leave LDS before its chain runs).
TL_FN void tl_psy1_weights(TlPsyLds &w, int nbands, PARG(double, wt))
{
    TL_LANES_BEGIN
    double weight = 0.0;
    if (lane < nbands) {
        const double *vt = w.u.fft;
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 16 <= i1; i += 16) {                             // sixteen operands per LDS round trip, summed in order
            double t[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 16; q++) t[q] = vt[i + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 16; q++) weight += t[q];
        }
        for (; i < i1; i++) weight += vt[i];
    }
    L(wt) = weight;
    TL_LANES_END
}

// dB sums and weight sums of the bands of ONE channel (levels at TL_PX, weight terms at fft): lane b < nbands
TL_FN void tl_psy1_chain(TlPsyLds &w, const double *TL_RESTRICT db, int nbands, PARG(double, bsum), PARG(double, wt))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN, weight = 0.0;
    if (lane < nbands) {
        const double *vt = w.u.fft, *vp = TL_PX(w);
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 4 <= i1; i += 4) {                               // operands of four steps in flight per LDS round trip
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            const double t0 = vt[i], t1 = vt[i + 1], t2 = vt[i + 2], t3 = vt[i + 3];
            sum = tl_add_db(db, p0, sum); weight += t0;
            sum = tl_add_db(db, p1, sum); weight += t1;
            sum = tl_add_db(db, p2, sum); weight += t2;
            sum = tl_add_db(db, p3, sum); weight += t3;
        }
        for (; i < i1; i++) { sum = tl_add_db(db, vp[i], sum); weight += vt[i]; }
    }
    L(bsum) = sum; L(wt) = weight;
    TL_LANES_END
}

// dB sums of BOTH channels at once: lanes 0..31 walk channel 0's bands (levels parked at fft[], ranges in r0/r1), lanes 32..63
// channel 1's (levels at TL_PX, ranges from bandoff[]).  Result: lane b holds channel 0's sum, lane 32+b channel 1's.
TL_FN void tl_psy1_chain2(TlPsyLds &w, const double *TL_RESTRICT db, int nbands, PARG(int, r0), PARG(int, r1), PARG(double, bsum))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN;
    const int band = lane & 31;
    if (band < nbands) {
        const bool second = lane >= 32;
        const double *vp = second ? TL_PX(w) : w.u.fft;
        const int i0 = second ? (int)w.bandoff[band] : L(r0), i1 = second ? (int)w.bandoff[band + 1] : L(r1);
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                               // eight steps' operands per LDS round trip
            double p[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) p[q] = vp[i + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) sum = tl_add_db(db, p[q], sum);
        }
        for (; i + 4 <= i1; i += 4) {
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            sum = tl_add_db(db, p0, sum); sum = tl_add_db(db, p1, sum); sum = tl_add_db(db, p2, sum); sum = tl_add_db(db, p3, sum);
        }
        for (; i < i1; i++) sum = tl_add_db(db, vp[i], sum);
    }
    L(bsum) = sum;
    TL_LANES_END
}

// band centres (psycho_1.c:367-388) from the sums and weights of lanes b < nbands; needs ptype[] of the channel
TL_FN void tl_psy1_centres(TlPsyLds &w, const TlConfig *TL_RESTRICT C, int nbands, PARG(double, bsum), PARG(double, wt))
{
    TL_LANES_BEGIN
    if (lane < nbands) {
        const int lo = C->p1_cbound[lane], hi = C->p1_cbound[lane + 1];
        const double sum = L(bsum), weight = L(wt);
        int centre;
        if (sum <= TL_DBMIN) centre = (hi + lo) / 2;
        else {
            double index = weight * tlm_pow10_sl(-0.1 * sum);
            centre = lo + (int)(index * (double)(hi - lo));
        }
        centre = centre < 1 ? 1 : centre > 510 ? 510 : centre;     // out-of-range only on non-finite input (UB in the reference)
        if (w.ptype[centre] == TL_T_TONE) { if (w.ptype[centre + 1] == TL_T_TONE) centre++; else centre--; }
        w.nsum[lane] = sum; w.ncentre[lane] = (int16_t)centre;
    }
    TL_LANES_END
}

// individual + global masking thresholds, minimum per subband, SMR (psycho_1.c:480-581) from the masker lists
TL_FN void tl_psy1_thresholds(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, int ntone, int nnoise, PARGA(double, rec, 4), long long *sp)
{
    TL_STAMP(sp, 5);
    TL_PRIO2(TL_PS_THR);

    TL_DBG_DUMP("psy1", ch, ntone, nnoise, TL_MK_X(w), TL_MK_BARK(w));
    // ---- individual + global masking thresholds on the table lines (psycho_1.c:480-532) ----
    const int sub = C->p1_sub;
    TL_LANES_BEGIN
    for (int t = lane; t < ntone + nnoise; t += 64) tl_masker_consts(TL_MK4(w), TL_MK_X(w), TL_MK_BARK(w), t, t < ntone);
    TL_LANES_END
    // Each lane folds the maskers into two ADJACENT table lines at once (two independent dB-sum chains).  A masker only
    // reaches lines with -3 <= dz < 8 bark, so a lane first finds the first and last masker (tones, then noise, in list
    // order) that reaches either of its lines and walks only that span; the per-line range test stays in the walk, so
    // nothing depends on the lists being sorted.
    const bool srt = ntone < 128 && nnoise < 64 && tl_maskers_sorted(TL_MK_BARK(w), ntone, nnoise);
    for (int base = 1; base < (TL_EXP_LEVEL >= 1 ? 0 : sub); base += 128) {       // 126..132 lines: one full pass + a 4-line tail at most
        TL_LANES_BEGIN
        const int k0 = base + 2 * lane, k1 = k0 + 1;
        const bool h0 = k0 < sub, h1 = k1 < sub;
        if (h0) {
            const double bk0 = C->p1_bark[k0], bk1 = C->p1_bark[h1 ? k1 : k0];
            const double blo = (bk0 < bk1 ? bk0 : bk1) - 8.0, bhi = (bk0 < bk1 ? bk1 : bk0) + 3.0;
            const TlMasker *mk = TL_MK4(w);
            const int nm = ntone + nnoise;
            int a0, a1, b0, b1;                                     // spans inside the tone part and inside the noise part
            if (srt) tl_mask_spans_sorted<64, 32>(TL_MK_BARK(w), ntone, nnoise, blo, bhi, a0, a1, b0, b1);
            else tl_mask_spans(mk, nm, ntone, blo, bhi, a0, a1, b0, b1);
            // one walk over the tone span, then the noise span, two maskers per trip: their four masking terms do not depend on
            // the running sums and are computed while the masker reads and the previous table look-ups are under way
            double x0 = TL_DBMIN, x1 = TL_DBMIN;
            const int nt = a1 >= a0 ? a1 - a0 + 1 : 0, cnt = nt + (b1 >= b0 ? b1 - b0 + 1 : 0);
            int t = nt ? a0 : b0;
            TL_DBG_WALK(ch, lane, cnt);
            const TlMaskK kk = tl_mask_consts();
            for (int i = 0; i < cnt; i += 2) {
                const int tA = t, tB = tA == a1 ? b0 : tA + 1;
                t = tB == a1 ? b0 : tB + 1;
                const TlMasker *pA = &mk[tA & (TL_MASKER_MAX - 1)], *pB = &mk[tB & (TL_MASKER_MAX - 1)];
                const double bA = pA->bark, avA = pA->av, bB = pB->bark, avB = pB->av;
                const bool two = i + 1 < cnt;                           // an odd walk ends with a masker that reaches nothing
                const double mA0 = tl_mask_term_w(pA, bA - bk0, avA, kk.far_hi), mA1 = tl_mask_term_w(pA, bA - bk1, avA, kk.far_hi);
                const double mB0 = tl_mask_term_w(pB, bB - bk0, avB, kk.far_hi, two), mB1 = tl_mask_term_w(pB, bB - bk1, avB, kk.far_hi, two);
                tl_add_db2_k(db, kk.k1000, x0, mA0, x1, mA1);
                tl_add_db2_k(db, kk.k1000, x0, mB0, x1, mB1);
            }
            TL_LTG(w)[k0] = tl_add_db(db, C->br_per_ch < 96 ? C->p1_hear[k0] : C->p1_hear[k0] - 12.0, x0);
            if (h1) TL_LTG(w)[k1] = tl_add_db(db, C->br_per_ch < 96 ? C->p1_hear[k1] : C->p1_hear[k1] - 12.0, x1);
        }
        TL_LANES_END
    }
    TL_STAMP(sp, 6);

    // ---- minimum per subband (psycho_1.c:541-559) and SMR (psycho_1.c:568-581) ----
    TL_LANES_BEGIN
    if (lane < C->sblimit) {
        double m;
        int n = C->p1_mm_n[lane], j0 = C->p1_mm_j0[lane];
        if (n == 0) m = C->p1_hear[sub - 1];
        else {
            m = tl_min_rows(TL_LTG(w), j0, n, 0.0, true);
        }
        L(rec)[2 + ch] = m;                                         // the encoder finishes the line (tl_encode_frame, TL_PSY_EXT): SMR = max(spike, scale level) - m, psycho_1.c:575-580
    } else if (lane < 32) L(rec)[2 + ch] = 0.0;                     // subbands the model leaves alone
    TL_LANES_END
}

// band levels, decimation (psycho_1.c:390-470) and everything after; the regular (not dead-head) case
TL_FN void tl_psy1_back(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, const TlPsy1Ch &st, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1, nlist = st.nlist;
    int ntone = 0, nnoise = 0;
    TL_STAMP(sp, 4);
    if (TL_EXP_LEVEL >= 2) { tl_psy1_thresholds(w, db, C, ch, 0, 0, rec, sp); return; }
    // The reference now writes every band's sum to power[centre] in band order -- a later band overwrites an earlier one
    // that chose the same line, and (through the centre+1 rule above) a centre may even land on a tone's line
    // (psycho_1.c:390-398) -- and the decimation reads the levels back from power[].  The same values without the array:
    // a band's level is the sum of the LAST band with its centre, a tone's level is its own unless a band centre sits on
    // its line.  (Lane reads, no LDS round trips.)
    PV(int, ncen); PV(double, nlev); PV(int, nsh); PV(int, nsl); PV(bool, ontone);
    TL_LANES_BEGIN
    L(ncen) = lane < nbands ? (int)w.ncentre[lane] : -1 - lane;
    const double v = lane < nbands ? w.nsum[lane] : 0.0;
    L(nlev) = v; L(nsh) = (int)(uint32_t)(tl_d2u(v) >> 32); L(nsl) = (int)(uint32_t)tl_d2u(v);
    L(ontone) = lane < nbands && w.ptype[L(ncen)] == TL_T_TONE;
    TL_LANES_END
    const bool centre_on_tone = TL_BALLOT(ontone) != 0ull;            // only then can a tone's level be replaced (rare)
    // Two bands with the same centre are rare too.  Every band writes its index at its centre in a scratch map (the candidate
    // records are dead by now) and reads it back: with all centres distinct every band finds itself; otherwise some band
    // finds another one (whichever write lands last) and the overwrite order is resolved band by band.
    PV(bool, shared_c);
    uint8_t *cmark = (uint8_t *)w.cinfo;                              // centres are 1..510
    TL_LANES_BEGIN if (lane < nbands) cmark[L(ncen)] = (uint8_t)lane; TL_LANES_END
    TL_LANES_BEGIN L(shared_c) = lane < nbands && cmark[L(ncen)] != (uint8_t)lane; TL_LANES_END
    if (TL_BALLOT(shared_c) != 0ull)
        for (int b = 1; b < nbands; b++) {
            const int cb = TL_READLANE_I32(ncen, b);
            const double vb = tl_u2d(((uint64_t)(uint32_t)TL_READLANE_I32(nsh, b) << 32) | (uint32_t)TL_READLANE_I32(nsl, b));
            TL_LANES_BEGIN if (lane < b && L(ncen) == cb) L(nlev) = vb; TL_LANES_END
        }

    // ---- decimation (psycho_1.c:409-470) ----
    {
        // tones: keep if not erased and not below the threshold in quiet (order preserved)
        for (int base = 0; base < nlist; base += 64) {
            PV(bool, keep); PV(double, kx); PV(double, kb); PV(int, tline);
            TL_LANES_BEGIN
            double x = 0; int c = -1000 - lane;
            if (base + lane < nlist) { const int ti = w.tlist[base + lane]; c = w.conf_c[ti] & 511; x = w.tone_x[ti]; }
            L(kx) = x; L(tline) = c;
            TL_LANES_END
            if (centre_on_tone)
                for (int b = 0; b < nbands; b++) {                  // a band centre on the tone's line replaces its level
                    const int cb = TL_READLANE_I32(ncen, b);
                    const double vb = tl_u2d(((uint64_t)(uint32_t)TL_READLANE_I32(nsh, b) << 32) | (uint32_t)TL_READLANE_I32(nsl, b));
                    TL_LANES_BEGIN if (L(tline) == cb) L(kx) = vb; TL_LANES_END
                }
            TL_LANES_BEGIN
            bool kp = false; double bk = 0;
            if (base + lane < nlist) {
                const int cc = w.conf_c[w.tlist[base + lane]], c = cc & 511;
                bk = C->p1_lbark[c];
                kp = !((cc >> 13) & 1) && !(L(kx) < C->p1_lhear[c]);
            }
            L(keep) = kp; L(kb) = bk;
            TL_LANES_END
            const uint64_t m = TL_BALLOT(keep);
            TL_LANES_BEGIN
            if ((m >> lane) & 1ull) {
                const int pos = ntone + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                TL_MK_X(w)[pos] = L(kx); TL_MK_BARK(w)[pos] = L(kb);
            }
            TL_LANES_END
            ntone += __builtin_popcountll(m);
        }
        // tones closer than 0.5 bark: keep the stronger (psycho_1.c:443-469).  The walk compares each tone with the current
        // survivor; as long as no two NEIGHBOURS of the list are that close the survivor is always the previous tone and
        // nothing is merged, which one line-parallel comparison settles.  Only otherwise the sequential walk runs.
        {
            PV(bool, closep);
            TL_LANES_BEGIN
            bool cl = false;
            for (int q = 1 + lane; q < ntone; q += 64) cl = cl || (TL_MK_BARK(w)[q] - TL_MK_BARK(w)[q - 1] < 0.5);
            L(closep) = cl;
            TL_LANES_END
            if (TL_BALLOT(closep) != 0ull) {
                int n = 0;                // compacted in place: entries [0,n) are final, (xi,bi) is the current survivor
                double xi = TL_MK_X(w)[0], bi = TL_MK_BARK(w)[0];
                for (int q = 1; q < ntone; q++) {
                    const double xn = TL_MK_X(w)[q], bn = TL_MK_BARK(w)[q];
                    if (bn - bi < 0.5) {
                        if (xn > xi) { xi = xn; bi = bn; }           // drop i, continue from next
                    } else { TL_MK_X(w)[n] = xi; TL_MK_BARK(w)[n] = bi; n++; xi = xn; bi = bn; }
                }
                TL_MK_X(w)[n] = xi; TL_MK_BARK(w)[n] = bi; n++;
                ntone = n;
                TL_SYNC();
            }
        }
        // noise: band order, keep if not below the threshold in quiet (psycho_1.c:429-442)
        PV(bool, keepn); PV(double, nx); PV(double, nb);
        TL_LANES_BEGIN
        bool kp = false; double x = 0, bk = 0;
        if (lane < nbands) {
            const int c = L(ncen);
            x = L(nlev); bk = C->p1_lbark[c];
            kp = !(x < C->p1_lhear[c]);
        }
        L(keepn) = kp; L(nx) = x; L(nb) = bk;
        TL_LANES_END
        const uint64_t mn = TL_BALLOT(keepn);
        TL_LANES_BEGIN
        if ((mn >> lane) & 1ull) {
            const int pos = ntone + __builtin_popcountll(mn & ((1ull << lane) - 1ull));
            TL_MK_X(w)[pos] = L(nx); TL_MK_BARK(w)[pos] = L(nb);
        }
        TL_LANES_END
        nnoise = __builtin_popcountll(mn);
    }
    tl_psy1_thresholds(w, db, C, ch, ntone, nnoise, rec, sp);
}

// the dead-head replay (see tl_psy1_front): works on power[] (px) and the shared links like the reference
TL_FN void tl_psy1_deadhead(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, const TlPsy1Ch &st, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1, nconf = st.nconf;
    const uint8_t *map = C->p1_map;
    double *px = TL_PX(w);
    int ntone = 0, nnoise = 0;
    TL_STAMP(sp, 4);
    {
        // power[] as the replay needs it.  The array itself was compacted in place (tl_psy1_front), but the replay only ever
        // reads the lines of its chains: a confirmed tone's line holds the tone's summed level (psycho_1.c:317-321) unless
        // its successor erased it (DBMIN, :322-326); every other line the replay can reach is non-tonal and inside the
        // bands, i.e. consumed by its band (DBMIN, psycho_1.c:363).
        TL_LANES_BEGIN
        for (int j = lane; j < 520; j += 64) px[j] = TL_DBMIN;
        TL_LANES_END
        TL_LANES_BEGIN
        for (int i = lane; i < nconf; i += 64) { const int cc = w.conf_c[i]; if (!((cc >> 13) & 1)) px[cc & 511] = w.tone_x[i]; }
        TL_LANES_END
        TL_DBG_DUMP("deadhead", ch, 0, 0, px, px);
        int16_t *pnext = (int16_t *)w.cinfo;                // candidate records are dead by now
        TL_LANES_BEGIN
        for (int i = lane; i < 512; i += 64) pnext[i] = TL_STOP;
        TL_LANES_END
        TL_LANES_BEGIN
        for (int i = lane; i < nconf; i += 64) {
            const int nx = w.conf_nxt[i];
            pnext[w.conf_c[i] & 511] = (int16_t)(nx >= 0 ? (w.conf_c[nx] & 511) : nx);
        }
        TL_LANES_END
        int tone = w.conf_c[0] & 511, noise = 0;
        {   // noise chain in band order (psycho_1.c:390-398)
            int last = TL_LAST;
            for (int i = 0; i < nbands; i++) {
                const int centre = w.ncentre[i];
                if (last == TL_LAST) noise = centre;
                else { pnext[centre] = TL_LAST; pnext[last] = (int16_t)centre; }
                px[centre] = w.nsum[i]; w.ptype[centre] = TL_T_NOISE; last = centre;
            }
        }
        {   // psycho_1.c:409-470 verbatim on the shared links
            int i = tone, old = TL_STOP, guard = 0;
            while (i != TL_LAST && i != TL_STOP && guard++ < 600) {
                if (px[i] < C->p1_hear[map[i]]) {
                    w.ptype[i] = 0; px[i] = TL_DBMIN;
                    if (old == TL_STOP) tone = pnext[i]; else pnext[old] = pnext[i];
                } else old = i;
                i = pnext[i];
            }
            i = noise; old = TL_STOP; guard = 0;
            while (i != TL_LAST && i != TL_STOP && guard++ < 600) {
                if (px[i] < C->p1_hear[map[i]]) {
                    w.ptype[i] = 0; px[i] = TL_DBMIN;
                    if (old == TL_STOP) noise = pnext[i]; else pnext[old] = pnext[i];
                } else old = i;
                i = pnext[i];
            }
            i = tone; old = TL_STOP; guard = 0;
            while (i != TL_LAST && i != TL_STOP && guard++ < 600) {
                const int nx = pnext[i];
                if (nx == TL_LAST) break;
                if (nx == TL_STOP) break;                     // (the reference would index power[-100]; never reached in practice)
                if (C->p1_bark[map[nx]] - C->p1_bark[map[i]] < 0.5) {
                    if (px[nx] > px[i]) {
                        if (old == TL_STOP) tone = nx; else pnext[old] = (int16_t)nx;
                        w.ptype[i] = 0; px[i] = TL_DBMIN; i = nx;
                    } else {
                        w.ptype[nx] = 0; px[nx] = TL_DBMIN;
                        pnext[i] = pnext[nx]; old = i;
                    }
                } else { old = i; i = nx; }
            }
            guard = 0;
            for (int t = tone; t != TL_LAST && t != TL_STOP && ntone < TL_MASKER_MAX - 32 && guard++ < 600; t = pnext[t]) {
                TL_MK_X(w)[ntone] = px[t]; TL_MK_BARK(w)[ntone] = C->p1_bark[map[t]]; ntone++;
            }
            guard = 0;
            for (int t = noise; t != TL_LAST && t != TL_STOP && ntone + nnoise < TL_MASKER_MAX && guard++ < 600; t = pnext[t]) {
                TL_MK_X(w)[ntone + nnoise] = px[t]; TL_MK_BARK(w)[ntone + nnoise] = C->p1_bark[map[t]]; nnoise++;
            }
        }
        TL_SYNC();
    }
    tl_psy1_thresholds(w, db, C, ch, ntone, nnoise, rec, sp);
}

// one channel start to end (mono streams; stereo streams when a dead-head case forces the plain order)
TL_FN void tl_psy1_finish(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, const TlPsy1Ch &st, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1;
    PV(double, wt); PV(double, bsum);
    TL_PRIO(1); tl_psy1_chain(w, db, nbands, bsum, wt); TL_PRIO(0);
    tl_psy1_centres(w, C, nbands, bsum, wt);
    if (st.dead_head) tl_psy1_deadhead(w, db, C, ch, st, rec, sp); else tl_psy1_back(w, db, C, ch, st, rec, sp);
}
TL_FN void tl_psy1(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                   const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const TlPsy1Ch st = tl_psy1_front(w, T, db, C, pv, ch, rec, sp);
    tl_psy1_finish(w, db, C, ch, st, rec, sp);
}

// Both channels of a stereo frame.  Order: front(0) -> park channel 0's front results in registers -> front(1) -> the dB-sum
// chains of both channels side by side -> back(1) -> channel 0's results return to the LDS arrays -> back(0).
// Parked: the compacted levels (<= 466 doubles: 8 per lane), the tone records (conf_c, tlist, tone_x), the spike levels, the
// band ranges and the weight sums.  ptype[] is not parked: after the tone labelling a line is TONE exactly if it is the line
// of a confirmed tone that was not erased by its successor, so it is rebuilt from conf_c.  A dead-head channel (see
// tl_psy1_front) falls back to the plain per-channel order.
TL_FN void tl_psy1_stereo(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                          const TlConfig *TL_RESTRICT C, const TlPcmView &pv, PARGA(double, rec, 4), long long *sp)
{
    const int nbands = C->p1_ncb - 1;
    long long *sp0 = sp ? sp + 8 : nullptr, *sp1 = sp ? sp + 16 : nullptr;
    const TlPsy1Ch s0 = tl_psy1_front(w, T, db, C, pv, 0, rec, sp0);
    if (s0.dead_head) {                                               // plain order for both channels
        tl_psy1_finish(w, db, C, 0, s0, rec, sp0);
        tl_psy1(w, T, db, C, pv, 1, rec, sp1);
        return;
    }
    // ---- park channel 0 ----
    PV(double, wt0); PV(int, r0); PV(int, r1);
    PA(double, pvp, 8); PV(int, pcc); PV(int, ptl); PV(double, ptx0); PV(double, ptx1);
    if (TL_EXP_LEVEL < 3) tl_psy1_weights(w, nbands, wt0);
    TL_LANES_BEGIN
    L(r0) = lane < nbands ? (int)w.bandoff[lane] : 0; L(r1) = lane < nbands ? (int)w.bandoff[lane + 1] : 0;
    const double *vp = TL_PX(w);
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) L(pvp)[k] = lane + 64 * k < 504 ? vp[lane + 64 * k] : 0.0;
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    L(pcc) = (int)((uint32_t)(uint16_t)w.conf_c[lane] | ((uint32_t)(uint16_t)w.conf_c[hi] << 16));
    L(ptl) = (int)((uint32_t)(uint16_t)w.tlist[lane] | ((uint32_t)(uint16_t)w.tlist[hi] << 16));
    L(ptx0) = w.tone_x[lane]; L(ptx1) = w.tone_x[hi];
    TL_LANES_END
    // ---- channel 1's front; a dead-head channel 1 is finished in the plain order first ----
    const TlPsy1Ch s1 = tl_psy1_front(w, T, db, C, pv, 1, rec, sp1);
    PV(double, bsum); PV(double, wt1);
    if (s1.dead_head) {
        tl_psy1_finish(w, db, C, 1, s1, rec, sp1);
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int k = 0; k < 8; k++) if (lane + 64 * k < 504) TL_PX(w)[lane + 64 * k] = L(pvp)[k];
        if (lane <= nbands) w.bandoff[lane] = (int16_t)(lane < nbands ? L(r0) : 0);
        TL_LANES_END
        // bandoff[nbands] = end of the last band
        {
            const int last_end = TL_READLANE_I32(r1, nbands - 1);
            TL_LANES_BEGIN if (lane == 0) w.bandoff[nbands] = (int16_t)last_end; TL_LANES_END
        }
        PV(double, wdummy);
        // the weight terms are gone; tl_psy1_chain's weight output is ignored (the parked sums are used)
        TL_LANES_BEGIN
        for (int i = lane; i < 504; i += 64) { uint64_t z = 0; TL_KEEP(z); w.u.fft[i] = tl_u2d(z); }   // (a zero made here, not a register kept through the frame)
        TL_LANES_END
        TL_PRIO(1); tl_psy1_chain(w, db, nbands, bsum, wdummy); TL_PRIO(0);
    } else {
        // ---- both chains: channel 1's weight sums first (its terms sit where channel 0's levels go) ----
        if (TL_EXP_LEVEL < 3) tl_psy1_weights(w, nbands, wt1);
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int k = 0; k < 8; k++) if (lane + 64 * k < 504) w.u.fft[lane + 64 * k] = L(pvp)[k];
        TL_LANES_END
        TL_STAMP(sp1, 4);
        if (TL_EXP_LEVEL < 3) { TL_PRIO(1); tl_psy1_chain2(w, db, nbands, r0, r1, bsum); TL_PRIO(0); }
        // ---- back(1): its sums move from lanes 32+b to lanes b ----
        PV(double, bsum1);
#ifdef TL_EMULATE
        for (int lane = 0; lane < 64; ++lane) bsum1[lane] = bsum[(lane + 32) & 63];
#else
        bsum1 = __shfl(bsum, (int)((threadIdx.x + 32u) & 63u), 64);
#endif
        if (TL_EXP_LEVEL < 3) tl_psy1_centres(w, C, nbands, bsum1, wt1);
        tl_psy1_back(w, db, C, 1, s1, rec, sp1);
    }
    // ---- channel 0 returns to the LDS arrays ----
    TL_LANES_BEGIN
    for (int i = lane; i < 520; i += 64) w.ptype[i] = 0;
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    w.conf_c[lane] = (int16_t)(L(pcc) & 0xffff); w.tlist[lane] = (int16_t)(L(ptl) & 0xffff); w.tone_x[lane] = L(ptx0);
    if (64 + lane < TL_TONE_MAX) { w.conf_c[hi] = (int16_t)((uint32_t)L(pcc) >> 16); w.tlist[hi] = (int16_t)((uint32_t)L(ptl) >> 16); w.tone_x[hi] = L(ptx1); }
    TL_LANES_END
    TL_LANES_BEGIN
    for (int i = lane; i < s0.nconf; i += 64) { const int cc = w.conf_c[i]; if (!((cc >> 13) & 1)) w.ptype[cc & 511] = TL_T_TONE; }
    TL_LANES_END
    if (TL_EXP_LEVEL < 3) tl_psy1_centres(w, C, nbands, bsum, wt0);
    tl_psy1_back(w, db, C, 0, s0, rec, sp0);
}

// ------------------------------------------------------------------------------------------
// psy model 3 (psycho_3.c:71-432) for channel `ch`; result in w.smr[ch][0..32).
TL_FN int tl_psy3_front(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                   const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const double *energy = w.u.fft;                                   // line i at TL_EX(i)
    double *px = TL_PX(w);
    const double *bark = C->p3_bark, *ath = C->p3_ath;
    TL_PRIO2(TL_PS_FHT);
    TL_STAMP(sp, 0);
    tl_psy_spectrum(w, T, pv, ch, sp);
    TL_STAMP(sp, 1);
    TL_PRIO2(TL_PS_POW);

    // power[1..512] (psycho_3.c:152-160); power[0] is an uninitialised slot in the reference, pinned
    // to 0.0 (oracle/mp2_oracle.c:psy3_run)
    // and the strongest line Xmax of each subband (psycho_3.c:163-183; line 512 is skipped, see oracle) -> the output record.  A subband's 16 lines sit in one row of
    // 16 lanes, so its maximum is a row reduction of the values just computed (no strided re-read of px).
    PA(double, pxa, 8);
    TL_LANES_BEGIN
    for (int h = 0; h < 2; h++) {                                   // four lines per lane at a time
        double e[4], v[4];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) e[q] = energy[TL_EX(lane + 64 * (4 * h + q))];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) v[q] = tl_power_db(e[q], TL_LOGTAB(db));
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int q = 0; q < 4; q++) {
            const int i = lane + 64 * (4 * h + q);
            px[i] = i == 0 ? 0.0 : v[q];
            L(pxa)[4 * h + q] = i == 0 ? TL_DBMIN : v[q];
        }
    }
    TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int it = 0; it < 8; it++) {
        PV(double, pxv); PV(double, pxm);
        TL_LANES_BEGIN L(pxv) = L(pxa)[it]; TL_LANES_END
        TL_ROW16_MAX_F64(pxm, pxv);
        // subband 4 it + r: the maximum sits in lane 16 r + 15; the record keeps it in lane 4 it + r (the encoder takes the
        // maximum with the scalefactor level, psycho_3.c:180-182)
        PV(double, xm);
        TL_LANES_BEGIN L(xm) = TL_DBMIN < L(pxm) ? L(pxm) : TL_DBMIN; TL_LANES_END
        TL_LANES_BEGIN
        {
            const double v = TL_OTHER(xm, , 16 * (lane & 3) + 15);
            if (lane < 32 && (lane >> 2) == it) L(rec)[ch] = v;
        }
        TL_LANES_END
    }
    TL_LANES_BEGIN
    if (lane == 0) px[512] = tl_power_db(energy[512], TL_LOGTAB(db));
    TL_LANES_END
    TL_PRIO2(1);                                                      // from here to the thresholds: serial stages
    // ---- tone labelling (psycho_3.c:186-247) ----
    // (1) local maxima 2..499 whose right-hand neighbours are >= 7 dB down, compacted ascending
    int ncand = 0;
    tl_cand_chunk<2, true>(w, 0, ncand);                           // lines -1..62: run 2
    tl_cand_chunk<3, true>(w, 1, ncand);                           // 63..126: run 3
    for (int c8 = 2; c8 < 4; c8++) tl_cand_chunk<6, true>(w, c8, ncand);     // 127..254: run 6
    for (int c8 = 4; c8 < 8; c8++) tl_cand_chunk<12, true>(w, c8, ncand);    // 255..510: run 12
    TL_STAMP(sp, 2);
    // (2) which candidates become tones: the candidates in lanes, one round per confirmed tone (see tl_psy1_front).  A confirmed tone k
    //     erases lines k-sr..k+sr (itself included) to DBMIN (psycho_3.c:243-244); a later maximum inside that reach R has power
    //     DBMIN and always fails, one above R sees erased left neighbours (always >= 7 dB down) and original ones beyond.
    int nconf = 0;
    {
        int R = -1;
        for (int kb = 0; kb < ncand; kb += 64) {
        PV(int, ck); PV(int, csr); PV(int, clf); PV(bool, act); PV(bool, dep);
        TL_LANES_BEGIN
        const bool in = kb + lane < ncand;
        const uint32_t info = w.cinfo[in ? kb + lane : 0];
        const uint32_t pinfo = w.cinfo[in && kb + lane > 0 ? kb + lane - 1 : 0];    // the candidate before this one
        L(ck) = (int)(info & 511u); L(csr) = (int)(info >> 21);
        L(clf) = (int)tl_cand_left<true>(px, L(ck), L(csr), px[L(ck)]);
        L(act) = in;
        // state-independent candidates as in tl_psy1_front: at least sr(previous candidate) + sr + 1 lines above the candidate before
        // it, a candidate is above every earlier reach R with all its left neighbours original -- its verdict is clf == 0 under any R
        L(dep) = in && kb + lane > 0 && L(ck) - (int)(pinfo & 511u) < (int)(pinfo >> 21) + L(csr) + 1;
        TL_LANES_END
        for (;;) {
            PV(bool, okv);
            TL_LANES_BEGIN
            const int k = L(ck), sr = L(csr);
            const int hi_j = sr < k - R - 1 ? sr : k - R - 1;         // bit j-2 set for j in [2, sr] with k-j > R
            const uint32_t orig = hi_j >= 2 ? (1u << (hi_j - 1)) - 1u : 0u;
            L(okv) = L(act) && k > R && !((uint32_t)L(clf) & orig);
            TL_LANES_END
            const uint64_t m = TL_BALLOT(okv);
            if (m == 0ull) break;
            const int wl = __builtin_ctzll(m);
            // the first passing candidate, and with it every passing state-independent one up to the next dependent candidate
            PV(bool, depact);
            TL_LANES_BEGIN L(depact) = L(dep) && L(act) && lane > wl; TL_LANES_END
            const uint64_t dm = TL_BALLOT(depact);
            const uint64_t batch = m & (dm ? (1ull << __builtin_ctzll(dm)) - 1ull : ~0ull);
            const int nb = __builtin_popcountll(batch), wlast = 63 - __builtin_clzll(batch);
            const int i0 = nconf, Rold = R;
            TL_LANES_BEGIN
            if ((batch >> lane) & 1ull) {
                const int i = i0 + __builtin_popcountll(batch & ((1ull << lane) - 1ull));
                if (i < TL_TONE_MAX) w.conf_c[i] = (int16_t)(L(ck) | ((lane == wl && L(ck) - 1 <= Rold) ? (1 << 12) : 0));
            }
            TL_LANES_END
            nconf = i0 + nb < TL_TONE_MAX ? i0 + nb : (i0 < TL_TONE_MAX ? TL_TONE_MAX : i0);
            R = TL_READLANE_I32(ck, wlast) + TL_READLANE_I32(csr, wlast);
            TL_LANES_BEGIN L(act) = L(act) && lane > wlast; TL_LANES_END
            TL_DBG_ROUND();
        }
        }
        TL_SYNC();
    }
    // (3) tone levels from the still-original spectrum (psycho_3.c:238-239); kept aside until the energies
    //     are dead (the masker lists share the FHT buffer)
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {
        const int cc = w.conf_c[i], k = cc & 511;
        const double temp = tl_add_db(db, (cc >> 12) & 1 ? TL_DBMIN : px[k - 1], px[k]);
        w.tone_x[i] = tl_add_db(db, temp, px[k + 1]);
    }
    TL_LANES_END
    // (4) erasures
    TL_LANES_BEGIN
    for (int i = lane; i < nconf; i += 64) {                         // straight-line, as in tl_psy1_front: sr is 2, 3, 6 or 12
        const int k = w.conf_c[i] & 511, sr = tl_run_psy3(k);
        double *pb = px + (k - 12);
#define TL_ERASE3(j) do { pb[12 - (j)] = TL_DBMIN; pb[12 + (j)] = TL_DBMIN; } while (0)
        pb[12] = TL_DBMIN; TL_ERASE3(1); TL_ERASE3(2);
        if (sr >= 3) {
            TL_ERASE3(3);
            if (sr >= 6) {
                TL_ERASE3(4); TL_ERASE3(5); TL_ERASE3(6);
                if (sr >= 12) { TL_ERASE3(7); TL_ERASE3(8); TL_ERASE3(9); TL_ERASE3(10); TL_ERASE3(11); TL_ERASE3(12); }
            }
        }
#undef TL_ERASE3
    }
    TL_LANES_END
    TL_STAMP(sp, 3);
    // ---- noise per critical band (psycho_3.c:264-304) + decimation (:313-320); one lane per band ----
    const int nb = C->p3_cbands;
    // Line-parallel preparation as in psy 1: the lines that are summed (not erased) are compacted in ascending
    // order -- levels in place in px[], energies in place in the FHT buffer (a compacted position is always below
    // its line), and each entry's distance j - lo from its band's first line (the factor of its centre-of-gravity term
    // (j-lo)*e, psycho_3.c:283-289) as 16 bits in the candidate records' place, which are dead by now -- so the per-band
    // part is three bare chains.
    {
        double *ve = w.u.fft, *vp = px;
        uint16_t *vj = (uint16_t *)w.cinfo;                          // [512]
        static_assert(sizeof(w.cinfo) >= 512 * sizeof(uint16_t), "distance records");
        PV(double, e512);
        TL_LANES_BEGIN L(e512) = energy[512]; TL_LANES_END
        int nvalid = 0;
        PA(uint32_t, linfo, 9);                                     // the table reads of all nine chunks in one batch
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int c8 = 0; c8 < 9; c8++) L(linfo)[c8] = 64 * c8 + lane < 520 ? C->p3_lineinfo[64 * c8 + lane] : 0u;
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int base = 0; base < 576; base += 64) {                // lines 1..512
            PV(bool, ok); PV(double, ev); PV(int, dj); PV(double, pvv); PV(int, bnd);
            TL_LANES_BEGIN
            const int j = base + lane;
            const uint32_t info = L(linfo)[base >> 6];
            const int lo = (int)((info >> 8) & 0xfffu);
            bool v = false; double e = 0, p = 0;
            if (info) {
                p = px[j];
                v = p != TL_DBMIN;
                e = j == 512 ? L(e512) : energy[TL_EX(j)];
            }
            L(ok) = v; L(ev) = e; L(dj) = j - lo; L(pvv) = p; L(bnd) = (info && j == lo) ? (int)(info & 0xffu) : -1;
            TL_LANES_END
            const uint64_t m = TL_BALLOT(ok);
            TL_LANES_BEGIN
            const int pos = nvalid + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (L(bnd) >= 0) w.bandoff[L(bnd)] = (int16_t)pos;     // first line of its band
            if (L(ok)) { ve[pos] = L(ev); vj[pos] = (uint16_t)L(dj); vp[pos] = L(pvv); }
            TL_LANES_END
            nvalid += __builtin_popcountll(m);
        }
        w.bandoff[nb] = (int16_t)nvalid;
        TL_SYNC();
    }
    return nconf;
}

// energy sum and centre-of-gravity sum of the bands (psycho_3.c:283-289), ascending line order; lane b < nb.  Used on their own
// for stereo frames, where only the levels take part in the shared dB-sum chain.
TL_FN void tl_psy3_moments(TlPsyLds &w, int nb, PARG(double, es), PARG(double, cg))
{
    TL_LANES_BEGIN
    double esum = 0, cw = 0;
    if (lane < nb) {
        const double *ve = w.u.fft;
        const uint16_t *vj = (const uint16_t *)w.cinfo;
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                               // operands of eight steps per LDS round trip, summed in order
            double e[8], c[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) { e[q] = ve[i + q]; c[q] = (int)vj[i + q] * e[q]; }     // (j - lo) * e, psycho_3.c:287
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) { esum += e[q]; cw += c[q]; }
        }
        for (; i < i1; i++) { esum += ve[i]; cw += (int)vj[i] * ve[i]; }
    }
    L(es) = esum; L(cg) = cw;
    TL_LANES_END
}

// dB sums, energy sums and centre-of-gravity sums of the bands of ONE channel: lane b < nb
TL_FN void tl_psy3_chain(TlPsyLds &w, const double *TL_RESTRICT db, int nb, PARG(double, bsum), PARG(double, es), PARG(double, cg))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN, esum = 0, cw = 0;
    if (lane < nb) {
        const double *ve = w.u.fft, *vp = TL_PX(w);
        const uint16_t *vj = (const uint16_t *)w.cinfo;
        const int i0 = w.bandoff[lane], i1 = w.bandoff[lane + 1];
        int i = i0;
        for (; i + 4 <= i1; i += 4) {                               // operands of four steps in flight per LDS round trip
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            const double e0 = ve[i], e1 = ve[i + 1], e2 = ve[i + 2], e3 = ve[i + 3];
            const double c0 = (int)vj[i] * e0, c1 = (int)vj[i + 1] * e1, c2 = (int)vj[i + 2] * e2, c3 = (int)vj[i + 3] * e3;
            sum = tl_add_db(db, p0, sum); esum += e0; cw += c0;
            sum = tl_add_db(db, p1, sum); esum += e1; cw += c1;
            sum = tl_add_db(db, p2, sum); esum += e2; cw += c2;
            sum = tl_add_db(db, p3, sum); esum += e3; cw += c3;
        }
        for (; i < i1; i++) { sum = tl_add_db(db, vp[i], sum); esum += ve[i]; cw += (int)vj[i] * ve[i]; }
    }
    L(bsum) = sum; L(es) = esum; L(cg) = cw;
    TL_LANES_END
}

// dB sums of BOTH channels at once: lanes 0..31 walk channel 0's bands (levels parked at fft[], ranges in r0/r1), lanes 32..63
// channel 1's (levels in px[], ranges from bandoff[]).  Lane b holds channel 0's sum, lane 32+b channel 1's.
TL_FN void tl_psy3_chain2(TlPsyLds &w, const double *TL_RESTRICT db, int nb, PARG(int, r0), PARG(int, r1), PARG(double, bsum))
{
    TL_LANES_BEGIN
    double sum = TL_DBMIN;
    const int band = lane & 31;
    if (band < nb) {
        const bool second = lane >= 32;
        const double *vp = second ? TL_PX(w) : w.u.fft;
        const int i0 = second ? (int)w.bandoff[band] : L(r0), i1 = second ? (int)w.bandoff[band + 1] : L(r1);
        int i = i0;
        for (; i + 8 <= i1; i += 8) {                               // eight steps' operands per LDS round trip
            double p[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) p[q] = vp[i + q];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) sum = tl_add_db(db, p[q], sum);
        }
        for (; i + 4 <= i1; i += 4) {
            const double p0 = vp[i], p1 = vp[i + 1], p2 = vp[i + 2], p3 = vp[i + 3];
            sum = tl_add_db(db, p0, sum); sum = tl_add_db(db, p1, sum); sum = tl_add_db(db, p2, sum); sum = tl_add_db(db, p3, sum);
        }
        for (; i < i1; i++) sum = tl_add_db(db, vp[i], sum);
    }
    L(bsum) = sum;
    TL_LANES_END
}

// band centres, decimation, thresholds, SMR (psycho_3.c:290-432) from the sums of lanes b < nb
TL_FN void tl_psy3_back(TlPsyLds &w, const double *TL_RESTRICT db, const TlConfig *TL_RESTRICT C, int ch, int nconf,
                        PARG(double, bsum), PARG(double, es), PARG(double, cg), PARGA(double, rec, 4), long long *sp)
{
    const double *bark = C->p3_bark, *ath = C->p3_ath;
    const int nb = C->p3_cbands;
    PV(bool, keepn); PV(double, nx); PV(double, nbk);
    TL_LANES_BEGIN
    bool kp = false; double xn = 0, bk = 0;
    if (lane < nb) {
        const int lo = C->p3_cbidx[lane], hi = C->p3_cbidx[lane + 1];
        const double sum = L(bsum), esum = L(es), cw = L(cg);
        // esum == 0: the reference indexes with (int)(0/0) and segfaults; defined as the band centre
        int centre = (sum <= TL_DBMIN || esum == 0) ? (lo + hi) / 2 : lo + (int)(cw / esum);
        centre = centre < 1 ? 1 : centre > 512 ? 512 : centre;
        xn = sum; bk = bark[centre];
        kp = !(xn < ath[centre]);
    }
    L(keepn) = kp; L(nx) = xn; L(nbk) = bk;
    TL_LANES_END
    const uint64_t mn = TL_BALLOT(keepn);       // ascending line order == band order (centres stay in their band)
    // tones: decimation against the threshold in quiet (psycho_3.c:321-326), compaction in ascending line order
    int ntone = 0;
    for (int base = 0; base < nconf; base += 64) {
        PV(bool, keep); PV(double, kx); PV(double, kb);
        TL_LANES_BEGIN
        bool kp2 = false; double x = 0, bk2 = 0;
        if (base + lane < nconf) {
            const int k = w.conf_c[base + lane] & 511;
            x = w.tone_x[base + lane]; bk2 = bark[k];
            kp2 = !(x < ath[k]);
        }
        L(keep) = kp2; L(kx) = x; L(kb) = bk2;
        TL_LANES_END
        const uint64_t m = TL_BALLOT(keep);
        TL_LANES_BEGIN
        if ((m >> lane) & 1ull) {
            const int pos = ntone + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            TL_MK_X(w)[pos] = L(kx); TL_MK_BARK(w)[pos] = L(kb);
        }
        TL_LANES_END
        ntone += __builtin_popcountll(m);
    }
    TL_LANES_BEGIN
    if ((mn >> lane) & 1ull) {
        const int pos = ntone + __builtin_popcountll(mn & ((1ull << lane) - 1ull));
        TL_MK_X(w)[pos] = L(nx); TL_MK_BARK(w)[pos] = L(nbk);
    }
    TL_LANES_END
    const int nnoise = __builtin_popcountll(mn);
    TL_STAMP(sp, 4);
    TL_STAMP(sp, 5);
    TL_PRIO2(TL_PS_THR);
    // ---- thresholds on the 136 subsampled lines (psycho_3.c:339-406) ----
    TL_LANES_BEGIN
    for (int t = lane; t < ntone + nnoise; t += 64) tl_masker_consts(TL_MK4(w), TL_MK_X(w), TL_MK_BARK(w), t, t < ntone);
    TL_LANES_END
    const bool srt = ntone < 128 && nnoise < 64 && tl_maskers_sorted(TL_MK_BARK(w), ntone, nnoise);
    // lines 0..127: every lane folds the maskers into two ADJACENT lines (two independent dB-sum chains at a time) and
    // walks only the maskers that can reach one of them (-3 <= dz < 8 bark; the exact test stays in the step)
    TL_LANES_BEGIN
    {
        const int j0 = 2 * lane, j1 = j0 + 1;
        const int line0 = C->p3_subset[j0], line1 = C->p3_subset[j1];
        const double b0 = bark[line0], b1 = bark[line1];
        const TlMasker *mk = TL_MK4(w);
        int ta0, ta1, tb0, tb1;
        if (srt) tl_mask_spans_sorted<64, 32>(TL_MK_BARK(w), ntone, nnoise, (b0 < b1 ? b0 : b1) - 8.0, (b0 < b1 ? b1 : b0) + 3.0, ta0, ta1, tb0, tb1);
        else tl_mask_spans(mk, ntone + nnoise, ntone, (b0 < b1 ? b0 : b1) - 8.0, (b0 < b1 ? b1 : b0) + 3.0, ta0, ta1, tb0, tb1);
        double lt0 = TL_DBMIN, ln0 = TL_DBMIN, lt1 = TL_DBMIN, ln1 = TL_DBMIN;
        uint32_t far_hi = 0xC0F00000u;
        TL_PIN(far_hi);
        for (int t = ta0; t <= ta1; t++) {
            const double mb = mk[t].bark, av = mk[t].av;
            lt0 = tl_mask_step(db, lt0, &mk[t], mb - b0, av, far_hi);
            lt1 = tl_mask_step(db, lt1, &mk[t], mb - b1, av, far_hi);
        }
        for (int t = tb0; t <= tb1; t++) {
            const double mb = mk[t].bark, av = mk[t].av;
            ln0 = tl_mask_step(db, ln0, &mk[t], mb - b0, av, far_hi);
            ln1 = tl_mask_step(db, ln1, &mk[t], mb - b1, av, far_hi);
        }
        const double g0 = tl_add_db(db, ln0, lt0), g1 = tl_add_db(db, ln1, lt1);
        TL_LTG(w)[j0] = tl_add_db(db, C->br_per_ch < 96 ? ath[line0] : ath[line0] - 12.0, g0);
        TL_LTG(w)[j1] = tl_add_db(db, C->br_per_ch < 96 ? ath[line1] : ath[line1] - 12.0, g1);
    }
    TL_LANES_END
    // lines 128..135: the tone sum and the noise sum of a line are independent chains (psycho_3.c:350-395), so
    // lanes 0..7 run the tone chains and lanes 8..15 the noise chains of the eight lines side by side
    TL_LANES_BEGIN
    if (lane < 16) {
        const int j = 128 + (lane & 7), line = C->p3_subset[j];
        const double bj = bark[line];
        const TlMasker *mk = TL_MK4(w);
        int ta0, ta1, tb0, tb1;                                       // only the maskers that can reach the line (it is one of the top eight)
        if (srt) tl_mask_spans_sorted<64, 32>(TL_MK_BARK(w), ntone, nnoise, bj - 8.0, bj + 3.0, ta0, ta1, tb0, tb1);
        else tl_mask_spans(mk, ntone + nnoise, ntone, bj - 8.0, bj + 3.0, ta0, ta1, tb0, tb1);
        const int t0 = lane < 8 ? ta0 : tb0, t1 = lane < 8 ? ta1 : tb1;
        double acc = TL_DBMIN;
        uint32_t far_hi = 0xC0F00000u;
        TL_PIN(far_hi);
        for (int t = t0; t <= t1; t++) {
            const double mb = mk[t].bark, av = mk[t].av;
            acc = tl_mask_step(db, acc, &mk[t], mb - bj, av, far_hi);
        }
        w.nsum[lane] = acc;
    }
    TL_LANES_END
    TL_LANES_BEGIN
    if (lane < 8) {
        const int j = 128 + lane, line = C->p3_subset[j];
        const double g = tl_add_db(db, w.nsum[8 + lane], w.nsum[lane]);
        TL_LTG(w)[j] = tl_add_db(db, C->br_per_ch < 96 ? ath[line] : ath[line] - 12.0, g);
    }
    TL_LANES_END
    TL_STAMP(sp, 6);
    // ---- minimum per subband + SMR (psycho_3.c:409-432); subset rows of subband sb are contiguous ----
    TL_LANES_BEGIN
    if (lane < 32) {
        double m = 999999.9;
        const int j0 = C->p3_sb_j0[lane], n = C->p3_sb_n[lane];
        m = tl_min_rows(TL_LTG(w), j0, n, m, false);
        L(rec)[2 + ch] = m;
    }
    TL_LANES_END
}


TL_FN void tl_psy3(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                   const TlConfig *TL_RESTRICT C, const TlPcmView &pv, int ch, PARGA(double, rec, 4), long long *sp)
{
    const int nconf = tl_psy3_front(w, T, db, C, pv, ch, rec, sp);
    PV(double, bsum); PV(double, es); PV(double, cg);
    TL_PRIO(1); tl_psy3_chain(w, db, C->p3_cbands, bsum, es, cg); TL_PRIO(0);
    tl_psy3_back(w, db, C, ch, nconf, bsum, es, cg, rec, sp);
}

// Both channels of a stereo frame, organised like tl_psy1_stereo: front(0) -> channel 0's compacted levels, tone records, Lsb
// and band moments wait in registers -> front(1) -> both dB-sum chains side by side -> back(1) -> back(0).
TL_FN void tl_psy3_stereo(TlPsyLds &w, const TlTables *TL_RESTRICT T, const double *TL_RESTRICT db,
                          const TlConfig *TL_RESTRICT C, const TlPcmView &pv, PARGA(double, rec, 4), long long *sp)
{
    const int nb = C->p3_cbands;
    long long *sp0 = sp ? sp + 8 : nullptr, *sp1 = sp ? sp + 16 : nullptr;
    const int nconf0 = tl_psy3_front(w, T, db, C, pv, 0, rec, sp0);
    PV(double, es0); PV(double, cg0); PV(int, r0); PV(int, r1);
    PA(double, pvp, 8); PV(int, pcc); PV(double, ptx0); PV(double, ptx1);
    tl_psy3_moments(w, nb, es0, cg0);
    TL_LANES_BEGIN
    L(r0) = lane < nb ? (int)w.bandoff[lane] : 0; L(r1) = lane < nb ? (int)w.bandoff[lane + 1] : 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) L(pvp)[k] = TL_PX(w)[lane + 64 * k];
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    L(pcc) = (int)((uint32_t)(uint16_t)w.conf_c[lane] | ((uint32_t)(uint16_t)w.conf_c[hi] << 16));
    L(ptx0) = w.tone_x[lane]; L(ptx1) = w.tone_x[hi];
    TL_LANES_END
    const int nconf1 = tl_psy3_front(w, T, db, C, pv, 1, rec, sp1);
    PV(double, es1); PV(double, cg1); PV(double, bsum); PV(double, bsum1);
    tl_psy3_moments(w, nb, es1, cg1);
    TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) w.u.fft[lane + 64 * k] = L(pvp)[k];   // channel 1's energies are summed: the buffer's lower half is free
    TL_LANES_END
    TL_PRIO(1); tl_psy3_chain2(w, db, nb, r0, r1, bsum); TL_PRIO(0);
#ifdef TL_EMULATE
    for (int lane = 0; lane < 64; ++lane) bsum1[lane] = bsum[(lane + 32) & 63];
#else
    bsum1 = __shfl(bsum, (int)((threadIdx.x + 32u) & 63u), 64);
#endif
    tl_psy3_back(w, db, C, 1, nconf1, bsum1, es1, cg1, rec, sp1);
    TL_LANES_BEGIN
    const int hi = 64 + lane < TL_TONE_MAX ? 64 + lane : 0;
    w.conf_c[lane] = (int16_t)(L(pcc) & 0xffff); w.tone_x[lane] = L(ptx0);
    if (64 + lane < TL_TONE_MAX) { w.conf_c[hi] = (int16_t)((uint32_t)L(pcc) >> 16); w.tone_x[hi] = L(ptx1); }
    TL_LANES_END
    tl_psy3_back(w, db, C, 0, nconf0, bsum, es0, cg0, rec, sp0);
}

// ------------------------------------------------------------------------------------------
// psy model 2 (psycho_2.c:52-254, psycho_2_fft fft.c:1230-1275), one 576-sample pass of channel `ch`.
// Two passes per frame; a pass needs the 480 samples before its 544 new ones -- the stream's PCM history on pass 0,
// samples 96..575 of the frame on pass 1 (the reference's savebuf shift by 576).
// Line-parallel: FFT, unpredictability (sincos/atan2/sqrt per line), thresholds; partition-parallel:
// grouping, spreading, SNR.  Every sum is one lane's sequential chain in the reference's order.
//
// The prediction state.  The reference keeps r = sqrt(energy) and phi of the two previous passes per line
// (psycho_2.c:111-116, 300-306) -- but there is no recurrence in it: both are functions of that pass's transform alone
// (`lthr`, the one true feedback of the model, is dead for Layer II, psycho_2.c:214-224).  So a run of passes can start
// anywhere: two SEED passes (transform, square root, arctangent -- no unpredictability, nothing after it) over the 1152
// samples before it rebuild exactly the state the chain would have carried there.  During a run the state lives in the wave's
// REGISTERS: line lane + 64 it in slot `it` of r1/p1 (previous pass) and r2/p2 (the pass before), line 512 in four LDS words.
#define TL_P2_L512(w) ((w).px + 516)     /* r1, r2, p1, p2 of line 512 (c[] / fthr[] end at px[512]) */
template <bool SEED>
TL_FN void tl_psy2_pass(TlPsy2Lds &w, const TlTables *TL_RESTRICT T, const TlPsy2Tables *TL_RESTRICT P, const TlPcmView &pv, int ch, int pass,
                        PARGA(double, r1, 8), PARGA(double, r2, 8), PARGA(double, p1, 8), PARGA(double, p2, 8),
                        PARG(double, snr0), double *smr_out, const uint64_t *sct, long long *sq)
{   // sct: glibc's __sincostab (tl_libm.h), the workgroup's LDS copy on the device
    double *x = w.u.fft;
    double *cw = w.px, *ge = w.u.fft + 520;  // c[] (unpredictability), then fthr[]; partition sums in the
    double *gc = ge + 64, *ecb = gc + 64, *nb = ecb + 64;   // dead upper half of the FHT buffer
    double *l5 = TL_P2_L512(w);
    {
        TL_STAMP(sq, 0);
        PA(double, twa, 8); PA(double, twb, 8); PA(double, twc, 8);
        TL_LANES_BEGIN
        {
            // sample i = lane + 64*it of the pass's 1024-sample window (psycho_2.c:84-92); loads in batches of eight ahead
            // of their use; slot of i inside the lane's block of sixteen: rev4(it) (see tl_fht_head)
            const double *win = P->window;
            TL_LAUNDER(win);
            const int16_t *pvh = ch ? pv.hist[1] : pv.hist[0], *pvc = ch ? pv.cur[1] : pv.cur[0];
            tl_fht_twiddles<4>(L(twc), T, lane);
            double e[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int half = 0; half < 16; half += 8) {
                int16_t v[8]; double h[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int q = 0; q < 8; q++) {
                    const int i = lane + 64 * (half + q);
                    if (pass == 0) v[q] = i < TL_HIST ? pvh[i] : pvc[i - TL_HIST];
                    else v[q] = pvc[96 + i];
                    h[q] = win[i];
                }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int q = 0; q < 8; q++) {
                    const int it = half + q;
                    const int r4 = ((it & 1) << 3) | ((it & 2) << 1) | ((it & 4) >> 1) | ((it & 8) >> 3);
                    e[r4] = h[q] * (double)v[q];
                }
            }
            tl_fht_twiddles<6>(L(twb), T, lane);
            tl_fht_head(e, T->fht_tw);
            tl_fht_store(x, lane, e);
        }
        TL_LANES_END
        TL_LANES_BEGIN tl_fht_twiddles<8>(L(twa), T, lane); tl_fht_pass<4>(x, L(twc), lane); TL_LANES_END
        TL_LANES_BEGIN tl_fht_pass<6>(x, L(twb), lane); TL_LANES_END
        TL_LANES_BEGIN tl_fht_pass<8>(x, L(twa), lane); TL_LANES_END
        TL_STAMP(sq, 1);
        // energy + phase (fft.c:1246-1275), unpredictability (psycho_2.c:119-140).
        // 64 lines per step: the transform is read through the layout map first, then the step's energies are written in
        // natural order (their slots hold nothing a later step reads).
        // Lines 0..511 are eight full steps of the wave; line 512 would be a ninth with ONE lane at work, at the price of a full
        // step (two sincos, an atan2, two square roots for every lane).  It needs no arctangent of its own (its phase is 0 or pi,
        // fft.c:1274) and line 0 needs none either and no sincos of its phase (phi = 0, fft.c:1257-1259), so in step 0 lane 0 puts
        // line 512's PREDICTED phase through its first sincos slot and finishes that line with a few extra operations.
        PV(double, e512);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) {
            PV(double, xa); PV(double, xb); PV(double, xc);
            TL_LANES_BEGIN
            const int j = lane + 64 * it;
            L(xa) = x[TL_FX(j)];
            L(xb) = j >= 1 ? x[TL_FX(1024 - j)] : 0.0;
            L(xc) = it == 0 ? x[TL_FX(512)] : 0.0;
            TL_LANES_END
            TL_LANES_BEGIN
            const int j = lane + 64 * it;
            {
                const bool first = j == 0;                           // lane 0 of step 0: lines 0 and 512
                double r_o5 = 0, r_n5 = 0, p_o5 = 0, p_n5 = 0;      // state of line 512
                if (it == 0) { r_o5 = l5[0]; r_n5 = l5[1]; p_o5 = l5[2]; p_n5 = l5[3]; }
                const double a = L(xa), b = L(xb);
                double e = (a * a + b * b) / 2.0;
                const bool low = e < 0.0005;
                double phi = tlm_atan2_sl<false>(-a, b, tlm_atan_cij) + 3.14159265358979 / 4;
                e = TL_SELECT(low, 0.0005, e); phi = TL_SELECT(low, 0.0, phi);
                e = TL_SELECT(first, a * a, e); phi = TL_SELECT(first, 0.0, phi);      // line 0: energy x_real[0]^2, phase 0
                const double rn = sqrt(e);
                double spp5 = 0, cpp5 = 0;
                if (!SEED) {
                    const double r_prime = 2.0 * L(r1)[it] - L(r2)[it];
                    const double phi_prime = 2.0 * L(p1)[it] - L(p2)[it];
                    double sp, cp, spp, cpp;
                    tlm_sincos_sl(TL_SELECT(first, 2.0 * p_o5 - p_n5, phi), &sp, &cp, sct);
                    tlm_sincos_sl(phi_prime, &spp, &cpp, sct);
                    spp5 = sp; cpp5 = cp;                                // sincos of line 512's predicted phase (lane 0 of step 0)
                    sp = TL_SELECT(first, 0.0, sp); cp = TL_SELECT(first, 1.0, cp);         // sincos(0.0)
                    const double t1 = rn * cp - r_prime * cpp;
                    const double t2 = rn * sp - r_prime * spp;
                    const double t3 = rn + fabs(r_prime);
                    cw[j] = t3 != 0 ? sqrt(t1 * t1 + t2 * t2) / t3 : 0;
                    x[j] = e;
                }
                L(r2)[it] = L(r1)[it]; L(r1)[it] = rn; L(p2)[it] = L(p1)[it]; L(p1)[it] = phi;
                if (it == 0) {                                       // line 512 (psycho_2.c:110-140 with fft.c:1274's phase), finished by lane 0
                    const double c5 = L(xc);
                    const double e5 = c5 * c5;
                    const bool neg5 = (tl_d2u(c5) >> 63) != 0;       // atan2(+0.0, x) = pi for x < 0 and x = -0, else +0
                    const double phi5 = neg5 ? tl_u2d(0x400921fb54442d18ull) : 0.0;
                    const double rn5 = sqrt(e5);
                    double c512 = 0;
                    if (!SEED) {
                        const double sp5 = neg5 ? tl_u2d(0x3ca1a62633145c07ull) : 0.0, cp5 = neg5 ? -1.0 : 1.0;   // glibc's sincos of that pi / of 0
                        const double r_prime5 = 2.0 * r_o5 - r_n5;
                        const double t15 = rn5 * cp5 - r_prime5 * cpp5;
                        const double t25 = rn5 * sp5 - r_prime5 * spp5;
                        const double t35 = rn5 + fabs(r_prime5);
                        c512 = t35 != 0 ? sqrt(t15 * t15 + t25 * t25) / t35 : 0;
                    }
                    if (first) { l5[0] = rn5; l5[1] = r_o5; l5[2] = phi5; l5[3] = p_o5; if (!SEED) cw[512] = c512; }
                    L(e512) = e5;                                    // slot 512 of the transform buffer still holds a point step 7 reads
                }
            }
            TL_LANES_END
        }
        if (SEED) return;
        TL_LANES_BEGIN
        if (lane == 0) x[512] = L(e512);
        TL_LANES_END
        TL_STAMP(sq, 2);
        const double *energy = x;
        // grouped energy / weighted unpredictability per partition (psycho_2.c:146-155)
        TL_LANES_BEGIN
        {
            double e = 0, c = 0;
            if (lane < P->npart) {
                const int lo = P->part_lo[lane], hi = P->part_hi[lane];
                int j = lo;
                for (; j + 8 <= hi; j += 8) {                           // eight lines' operands per LDS round trip, summed in line order
                    double ev[8], cv[8];
#ifndef TL_EMULATE
#pragma unroll
#endif
                    for (int q = 0; q < 8; q++) { ev[q] = energy[j + q]; cv[q] = cw[j + q]; }
#ifndef TL_EMULATE
#pragma unroll
#endif
                    for (int q = 0; q < 8; q++) { e += ev[q]; c += ev[q] * cv[q]; }
                }
                for (; j < hi; j++) { e += energy[j]; c += energy[j] * cw[j]; }
            }
            ge[lane] = e; gc[lane] = c;
        }
        TL_LANES_END
        TL_STAMP(sq, 3);
        // spreading (psycho_2.c:161-175), required SNR (:181-193), permissible noise (:200-204)
        TL_LANES_BEGIN
        {
            double e = 0, c = 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k0 = 0; k0 < 64; k0 += 16) {                   // sixteen coefficient loads in flight per round trip
                double sv[16];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int q = 0; q < 16; q++) sv[q] = P->s_t[k0 + q][lane];
#ifndef TL_EMULATE
#pragma unroll
#endif
                // the reference skips zero coefficients (psycho_2.c:165); adding their +-0 products leaves the sums unchanged
                // bit for bit (finite operands, sums start at +0), so the test is dropped instead of branching 64 times
                for (int q = 0; q < 16; q++) { e += sv[q] * ge[k0 + q]; c += sv[q] * gc[k0 + q]; }
            }
            double cb = e != 0 ? c / e : 0;
            if (cb < .05) cb = 0.05; else if (cb > .5) cb = 0.5;
            const double tb = -0.434294482 * tlm_log_pn(cb, tlm_log_tab) - 0.301029996;
            double bc = P->tmn[lane] * tb + 5.5 * (1.0 - tb);
            bc = bc > P->bmaxk[lane] ? bc : P->bmaxk[lane];
            bc = tlm_exp_sl<false>(-bc * 0.2302585093, 0.0);
            ecb[lane] = e;
            nb[lane] = P->den[lane] != 0 ? e * bc / P->den[lane] : 0;
        }
        TL_LANES_END
        TL_STAMP(sq, 4);
        // threshold per line (psycho_2.c:205-224): c[] is dead, reuse it for fthr[]
        TL_LANES_BEGIN
        for (int j = lane; j <= 512; j += 64) {
            const double t = nb[P->partition[j]], a = P->absthr[j];
            cw[j] = t > a ? t : a;
        }
        TL_LANES_END
        TL_STAMP(sq, 5);
        // 32 subbands (psycho_2.c:227-246)
        TL_LANES_BEGIN
        if (lane < 32) {
            const int j = 16 * lane;
            double minthres = lane < 13 ? 60802371420160.0 : 0.0, sum_energy = 0.0;
            for (int k = 0; k < 17; k++) {
                if (lane < 13) { if (minthres > cw[j + k]) minthres = cw[j + k]; }
                else minthres += cw[j + k];
                sum_energy += energy[j + k];
            }
            double snr = lane < 13 ? sum_energy / (minthres * 17.0) : sum_energy / minthres;
            snr = 4.342944819 * tlm_log_pn(snr, tlm_log_tab);
            if (pass == 0) L(snr0) = snr;
            else smr_out[lane] = L(snr0) > snr ? L(snr0) : snr;
        }
        TL_LANES_END
        TL_STAMP(sq, 6);
    }
}

// ---- K1: polyphase filterbank (subband.c:201-310), 36 blocks of 32 samples, for the `nlan` channels staged in w.u.fbk.pcm ----
// (the two channels of a stereo stream, one channel, or channel 0 of each of two mono streams sharing the wave)
// Window stage: lane (ch,i) owns yprime[i] and computes exactly the two window outputs it is made of
// (yprime[0]=y[16]; yprime[i]=y[i+16]+y[16-i], i<=16; y[i+16]-y[80-i], i>=17 -- every y is used by one
// yprime only, so nothing is computed twice), each as the reference's ascending 8-tap chain.
// Matrixing stage: lane (ch,sb) owns the even-k chain s0 (sb<16) or the odd-k chain s1 (sb>=16) of row
// min(sb,31-sb); the two halves swap values (a move, not a re-association): s[i]=s0+s1, s[31-i]=s0-s1.
TL_FN void tl_filterbank(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s, const int nch, PARGA(double, smp, 36))
{
    constexpr int FB = TlMainLds::kFbBatch;
        // the reference scales the sample, (pcm/32768)*C (subband.c:233,249); scaling the coefficient instead is the
        // same real product rounded once (2^-15 is exact, nothing underflows), so the bits are identical: enw_s = C / 32768
        // (host table; the encode kernel of the split path reads its workgroup's LDS copy).
        // Window taps as a rolling register file: tap j of block b is tap j+1 of block b+2 (the window advances 32
        // samples per block, the taps are 64 apart), so each block reads two new samples per lane from LDS instead of
        // sixteen (kept as integers and converted at every use: a window of doubles, converted once, measured slower each
        // time it was tried).  Slot of (b, j): [b & 1][((b >> 1) - j) & 7].
        // The coefficients are NOT kept in registers across batches: a batch fetches the eight of its ya-sums, runs them for
        // all its blocks, then the eight of its yb-sums -- 16 registers live instead of 32 next to the 72 of the samples.
        PA(int, xa, 16); PA(int, xb, 16);
        TL_LANES_BEGIN
        const int c = lane & 1, i = lane >> 1;
        const int ya = i == 0 ? 16 : i + 16, yb = i == 0 ? 16 : (i <= 16 ? 16 - i : 80 - i);
        for (int b = 0; b < 2; b++)
            for (int j = 1; j < 8; j++) {
                L(xa)[8 * b + ((0 - j) & 7)] = c < nch ? w.u.fbk.pcm[c][TL_HIST + 32 * b + 31 - ya - 64 * j] : 0;
                L(xb)[8 * b + ((0 - j) & 7)] = c < nch ? w.u.fbk.pcm[c][TL_HIST + 32 * b + 31 - yb - 64 * j] : 0;
            }
        TL_LANES_END
        TlMainLds::YpRows yp = w.yp_rows();
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int b0 = 0; b0 < 36; b0 += FB) {
            TL_LANES_BEGIN
            const int c = lane & 1, i = lane >> 1;
            if (c < nch) {
                const int ya = i == 0 ? 16 : i + 16, yb = i == 0 ? 16 : (i <= 16 ? 16 - i : 80 - i);
                // yprime = ya-sum (i == 0), ya-sum + yb-sum (i <= 16), ya-sum - yb-sum (i >= 17) as ONE addition: the yb-sum
                // with its sign flipped (a - b == a + (-b)) or replaced by -0.0 (a + (-0.0) == a, for every a)
                const uint64_t keep = i == 0 ? 0ull : ~0ull, flip = (i == 0 || i > 16) ? 0x8000000000000000ull : 0ull;
                // the batch's new samples (two per block) and its first coefficients are all requested before the first block is computed
                int na[FB], nb[FB];
                double cf[8], ta[FB];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) {
                    // X[k] = pcm[t0 + 31 - k], t0 = index of the block's first new sample
                    na[bb] = w.u.fbk.pcm[c][TL_HIST + 32 * (b0 + bb) + 31 - ya];
                    nb[bb] = w.u.fbk.pcm[c][TL_HIST + 32 * (b0 + bb) + 31 - yb];
                }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int j = 0; j < 8; j++) cf[j] = enw_s[ya + 64 * j];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) { TL_KEEP(na[bb]); TL_KEEP(nb[bb]); }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) {
                    const int b = b0 + bb, q = 8 * (b & 1), h = b >> 1;
                    L(xa)[q + (h & 7)] = na[bb];
                    double t = (double)L(xa)[q + (h & 7)] * cf[0];
                    for (int j = 1; j < 8; j++) t += (double)L(xa)[q + ((h - j) & 7)] * cf[j];
                    ta[bb] = t;
                }
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int j = 0; j < 8; j++) cf[j] = enw_s[yb + 64 * j];
#ifndef TL_EMULATE
#pragma unroll
#endif
                for (int bb = 0; bb < FB; bb++) {
                    const int b = b0 + bb, q = 8 * (b & 1), h = b >> 1;
                    L(xb)[q + (h & 7)] = nb[bb];
                    double t = (double)L(xb)[q + (h & 7)] * cf[0];
                    for (int j = 1; j < 8; j++) t += (double)L(xb)[q + ((h - j) & 7)] * cf[j];
                    yp[bb][c][i] = ta[bb] + tl_u2d((tl_d2u(t) & keep) ^ flip);
                }
            }
            TL_LANES_END
            PA(double, part, FB);
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1, par = sb < 16 ? 0 : 1, r = sb < 16 ? sb : 31 - sb;
            double acc[FB];
            for (int bb = 0; bb < FB; bb++) acc[bb] = 0.0;
            if (c < nch)
                for (int k = 0; k < 16; k++) {
                    const double m = B->dct_t[k][par][r];               // shared LDS copy, conflict-free per k
                    for (int bb = 0; bb < FB; bb++) acc[bb] += m * yp[bb][c][2 * k + par];
                }
            for (int bb = 0; bb < FB; bb++) L(part)[bb] = acc[bb];
            TL_LANES_END
            PA(double, oth, FB);
#ifdef TL_EMULATE
            for (int lane = 0; lane < 64; ++lane)
                for (int bb = 0; bb < FB; bb++) oth[lane][bb] = part[2 * (31 - (lane >> 1)) + (lane & 1)][bb];
#else
            {
                const int lane_ = (int)(threadIdx.x & 63u), partner = 2 * (31 - (lane_ >> 1)) + (lane_ & 1);
#pragma unroll
                for (int bb = 0; bb < FB; bb++) oth[bb] = __shfl(part[bb], partner, 64);
            }
#endif
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            for (int bb = 0; bb < FB; bb++)
                L(smp)[b0 + bb] = c < nch ? (sb < 16 ? L(part)[bb] + L(oth)[bb] : L(oth)[bb] - L(part)[bb]) : 0.0;
            TL_LANES_END
        }
}

// a_bit_allocation_new (encode_new.c:1078-1187) for the cells of the wave: the one or two channels of a stream, joint pairs included.
// adb: the frame's bits after header extension and PAD (toolame.c:292-301).  Returns the bits left over.
TL_FN int tl_allocate(const TlBlockShared *TL_RESTRICT B, int adb, int nch, int sblimit, int jsbound, PARG(int, a_ln), PARG(int, a_nbal),
                      PARG(int, a_sfs), PARG(int, a_sfs_o), PARG(double, a_smr), PARG(int, ba))
{
    PV(uint64_t, ukey); PV(uint64_t, ukey2); PV(int, nbits); PV(int, cost); PV(int, cost2);
    PV(int, jpair);                                             // lane belongs to a joint-coded pair (steps with its partner)
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = c < nch && sb < sblimit;
    const int maxa = (1 << L(a_nbal)) - 1;
    L(ukey) = live ? tl_mnr_key(B->snr_line[L(a_ln)][0] - L(a_smr)) : ~0ull;
    L(ukey2) = (live && 1 < maxa) ? tl_mnr_key(B->snr_line[L(a_ln)][1] - L(a_smr)) : ~0ull;
    L(ba) = 0;
    L(nbits) = (sb < sblimit && c < (sb < jsbound ? nch : 1)) ? L(a_nbal) : 0;
    // first step of a cell: samples + scfsi + scalefactors (both channels above jsbound), :1139-1147
    L(cost) = live ? B->bits12_line[L(a_ln)][1] + 2 + L(a_sfs) + ((nch == 2 && sb >= jsbound) ? 2 + L(a_sfs_o) : 0) : 0;
    L(cost2) = live ? B->bits12_line[L(a_ln)][2] - B->bits12_line[L(a_ln)][1] : 0;
    L(jpair) = (live && nch == 2 && sb >= jsbound) ? 1 : 0;
    TL_LANES_END
    const int bbal = TL_WAVE_SUM_I32(nbits);
    const int ad = adb - (bbal + 16 + 32);
    int spent = 0;                                              // bspl + bscf + bsel
    const bool any_pair = nch == 2 && jsbound < sblimit;
    for (; TL_ENC_LEVEL < 4;) {                                 // rounds
        PV(uint64_t, keff); PV(uint64_t, k2eff);
        TL_LANES_BEGIN L(keff) = L(ukey); L(k2eff) = L(ukey2); TL_LANES_END
        if (any_pair) {                                         // a pair acts at the smaller of its two keys
            PV(uint64_t, ok1); PV(uint64_t, ok2);
            TL_SWAP1_U64(ok1, ukey); TL_SWAP1_U64(ok2, ukey2);
            TL_LANES_BEGIN
            if (L(jpair)) { if (L(ok1) < L(keff)) L(keff) = L(ok1); if (L(ok2) < L(k2eff)) L(k2eff) = L(ok2); }
            TL_LANES_END
        }
        const uint64_t M = TL_WAVE_MIN_U64(k2eff);
        PV(bool, inb); PV(int, bcost);
        TL_LANES_BEGIN
        L(inb) = L(keff) < M;
        L(bcost) = (L(inb) && !(L(jpair) && (lane & 1))) ? L(cost) : 0;     // a pair pays once
        TL_LANES_END
        const uint64_t bm = TL_BALLOT(inb);
        if (bm == 0ull) break;
        const int csum = TL_WAVE_SUM_I32(bcost);
        bool last_round = false;
        if (csum > ad - spent) {
            // The round does not fit as a whole: admit its events up to the first refusal.  Each cell adds up the
            // prices of the round's events that come no later than its own (equal keys count as earlier, which can
            // only shorten the admitted prefix); the prefix sums grow along the greedy order, so the cells whose
            // sum still fits are exactly a prefix of it.  The event-by-event loop below deals with the rest.
            // The test is ONE compare, key of the event < the cell's own key + 1 (the key of a cell with an event is never ~0; a cell
            // without one compares against 0 and its sum is not read).  A cell's own event passes it as well -- that is the cell's own
            // price, so the sum starts at 0; the two cells of a joint pair share key and price (same allocation line, the two
            // scalefactor selections added up either way round), and the pair's one event is the own event of both.
            PV(int, pre); PV(int, kh); PV(int, kl); PV(uint64_t, kb);
            TL_LANES_BEGIN L(pre) = 0; L(kh) = (int)(uint32_t)(L(keff) >> 32); L(kl) = (int)(uint32_t)L(keff); L(kb) = L(keff) + 1; TL_LANES_END
            uint64_t pm = bm;
            while (pm) {
                const int j = __builtin_ctzll(pm);
                pm &= pm - 1;
                const int cj = TL_READLANE_I32(bcost, j);
                if (cj == 0) continue;                          // the non-paying lane of a pair
                const uint64_t kj = ((uint64_t)(uint32_t)TL_READLANE_I32(kh, j) << 32) | (uint32_t)TL_READLANE_I32(kl, j);
                TL_LANES_BEGIN
                if (kj < L(kb)) L(pre) += cj;
                TL_LANES_END
            }
            TL_LANES_BEGIN
            L(inb) = L(inb) && L(pre) <= ad - spent;
            L(bcost) = L(inb) ? L(bcost) : 0;
            TL_LANES_END
            if (TL_BALLOT(inb) == 0ull) break;
            spent += TL_WAVE_SUM_I32(bcost);
            last_round = true;
        } else spent += csum;
        TL_LANES_BEGIN
        if (L(inb)) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
        if (last_round) break;
    }
    for (; TL_ENC_LEVEL < 4;) {                                 // one event at a time
        // maxmnr_new (encode_new.c:1061-1077): smallest mnr, first in (ch, sb) order
        PV(uint64_t, key);
        TL_LANES_BEGIN
        L(key) = L(cost) <= ad - spent ? L(ukey) : ~0ull;
        TL_LANES_END
        const int wl = TL_WAVE_ARGMIN_U64(key);                  // ch 0 first, then ascending sb
        if (wl < 0) break;
        const int min_sb = wl >> 1;
        spent += TL_READLANE_I32(cost, wl);
        const bool joint_pair = (min_sb >= jsbound && nch == 2);
        TL_LANES_BEGIN
        if (lane == wl || (joint_pair && (lane ^ 1) == wl)) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
    }
    return ad - spent;
}

// The same allocation for the two mono streams sharing a wave, BOTH AT ONCE: lane = 2*sb + u owns cell sb of unit u.  Every minimum and
// sum is taken over the 32 lanes of one parity and lands in all of them (TL_PAR_*), so a unit's greedy loop advances on its own state
// (`room` = bits it may still spend, `ph` = still in the rounds) held in its own lanes, and what a unit's lanes compute is what
// tl_allocate(unit = u) computes for it; the wave leaves a loop when neither unit has anything left in it.
TL_FN void tl_allocate_pair(const TlBlockShared *TL_RESTRICT B, int adb0, int adb1, int sblimit, PARG(int, a_ln), PARG(int, a_nbal),
                            PARG(int, a_sfs), PARG(double, a_smr), PARG(int, ba))
{
    PV(uint64_t, ukey); PV(uint64_t, ukey2); PV(int, nbits); PV(int, cost); PV(int, cost2); PV(int, room); PV(int, ph); PV(int, bbal);
    TL_LANES_BEGIN
    const bool live = (lane >> 1) < sblimit;
    const int maxa = (1 << L(a_nbal)) - 1;
    L(ukey) = live ? tl_mnr_key(B->snr_line[L(a_ln)][0] - L(a_smr)) : ~0ull;
    L(ukey2) = (live && 1 < maxa) ? tl_mnr_key(B->snr_line[L(a_ln)][1] - L(a_smr)) : ~0ull;
    L(ba) = 0;
    L(nbits) = live ? L(a_nbal) : 0;
    L(cost) = live ? B->bits12_line[L(a_ln)][1] + 2 + L(a_sfs) : 0;
    L(cost2) = live ? B->bits12_line[L(a_ln)][2] - B->bits12_line[L(a_ln)][1] : 0;
    TL_LANES_END
    TL_PAR_SUM_I32(bbal, nbits);
    TL_LANES_BEGIN
    L(room) = ((lane & 1) ? adb1 : adb0) - (L(bbal) + 16 + 32);
    L(ph) = 0;
    TL_LANES_END
    for (; TL_ENC_LEVEL < 4;) {                                 // rounds, as in tl_allocate
        PV(uint64_t, k2); PV(uint64_t, M); PV(bool, inb); PV(int, bcost); PV(int, csum); PV(bool, part);
        TL_LANES_BEGIN L(k2) = L(ph) == 0 ? L(ukey2) : ~0ull; TL_LANES_END
        TL_PAR_MIN_U64(M, k2);
        TL_LANES_BEGIN
        L(inb) = L(ph) == 0 && L(ukey) < L(M);
        L(bcost) = L(inb) ? L(cost) | 0x10000 : 0;              // price, and one count per event (a round's prices stay far below 2^16)
        TL_LANES_END
        if (TL_BALLOT(inb) == 0ull) break;
        TL_PAR_SUM_I32(csum, bcost);
        TL_LANES_BEGIN L(part) = L(ph) == 0 && (L(csum) & 0xffff) > L(room); TL_LANES_END
        if (TL_BALLOT(part) != 0ull) {                          // a unit's round does not fit as a whole: its events up to the first refusal
            PV(int, pre); PV(int, kh); PV(int, kl); PV(bool, pin);
            TL_LANES_BEGIN
            L(pre) = 0; L(kh) = (int)(uint32_t)(L(ukey) >> 32); L(kl) = (int)(uint32_t)L(ukey); L(pin) = L(inb) && L(part);
            TL_LANES_END
            // Unit 0's events first, then unit 1's.  A cell counts event j when j's key is no later than its own and j belongs to its
            // unit: ONE compare against its own key + 1 (no key of a live cell is ~0) or against 0 for the other unit's events; the
            // cell's own event passes the test too, which is its own price (so the sum starts at 0; only cells with an event are read).
            PV(uint64_t, kb);
            for (int u = 0; u < 2; u++) {
                TL_LANES_BEGIN L(kb) = (lane & 1) == u ? L(ukey) + 1 : 0ull; TL_LANES_END
                uint64_t pm = TL_BALLOT(pin) & (0x5555555555555555ull << u);
                while (pm) {
                    const int j = __builtin_ctzll(pm);
                    pm &= pm - 1;
                    const int cj = TL_READLANE_I32(bcost, j) & 0xffff;
                    const uint64_t kj = ((uint64_t)(uint32_t)TL_READLANE_I32(kh, j) << 32) | (uint32_t)TL_READLANE_I32(kl, j);
                    TL_LANES_BEGIN
                    if (kj < L(kb)) L(pre) += cj;
                    TL_LANES_END
                }
            }
            TL_LANES_BEGIN
            if (L(part)) { L(inb) = L(inb) && L(pre) <= L(room); L(bcost) = L(inb) ? L(bcost) : 0; }
            TL_LANES_END
            TL_PAR_SUM_I32(csum, bcost);
        }
        TL_LANES_BEGIN
        if (L(ph) == 0) {
            L(room) -= L(csum) & 0xffff;
            if ((L(csum) >> 16) == 0 || L(part)) L(ph) = 1;      // nothing admitted, or the partial round was the unit's last
        }
        if (L(inb)) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
    }
    for (; TL_ENC_LEVEL < 4;) {                                 // one event per unit at a time
        PV(uint64_t, key); PV(uint64_t, mk); PV(bool, win);
        TL_LANES_BEGIN L(key) = L(cost) <= L(room) ? L(ukey) : ~0ull; TL_LANES_END
        TL_PAR_MIN_U64(mk, key);
        TL_LANES_BEGIN L(win) = L(key) == L(mk) && L(key) != ~0ull; TL_LANES_END
        const uint64_t m = TL_BALLOT(win);
        if (m == 0ull) break;
        const uint64_t m0 = m & 0x5555555555555555ull, m1 = m & 0xaaaaaaaaaaaaaaaaull;
        const int wl0 = m0 ? __builtin_ctzll(m0) : -1, wl1 = m1 ? __builtin_ctzll(m1) : -1;     // ascending sb (maxmnr_new, encode_new.c:1061-1077)
        const int c0 = wl0 >= 0 ? TL_READLANE_I32(cost, wl0) : 0, c1 = wl1 >= 0 ? TL_READLANE_I32(cost, wl1) : 0;
        TL_LANES_BEGIN
        L(room) -= (lane & 1) ? c1 : c0;
        if (lane == wl0 || lane == wl1) {
            const int nba = L(ba) + 1;
            L(ba) = nba;
            L(ukey) = L(ukey2);
            L(cost) = L(cost2);
            L(ukey2) = (nba + 1 >= (1 << L(a_nbal)) - 1) ? ~0ull : tl_mnr_key(B->snr_line[L(a_ln)][(nba + 1) & 15] - L(a_smr));
            L(cost2) = B->bits12_line[L(a_ln)][(nba + 2) & 15] - B->bits12_line[L(a_ln)][(nba + 1) & 15];
        }
        TL_LANES_END
    }
}

// ------------------------------------------------------------------------------------------
// One frame of one stream.  lane = 2*sb + ch owns subband sb of channel ch.  PSY (the psy model) is a compile-time
// parameter: one kernel per model keeps each kernel's code and register footprint to what that model needs.
// Where a frame of the frame-parallel encode kernel goes (exactly one of bytes / words is set): the output slot it waits in
// for its successor's ScF-CRC, or -- the last frame of a launch -- the batch's pending buffer (big-endian words like
// TlStreamState::pending); and the slot for its own ScF-CRC bytes, which tl_finish_stream stores into the frame before it.
struct TlFrameOut { uint8_t *bytes; uint32_t *words; uint8_t *scfcrc; };
template <int PSY>
TL_FN void tl_encode_frame(TlMainLds &w, const TlTables *TL_RESTRICT T, const TlBlockShared *TL_RESTRICT B,
                           const TlConfig *TL_RESTRICT C, const TlPsyOut *TL_RESTRICT PO,
                           const TlPcmView &pv, int xpad_len, const TlFrameOut &fo,
                           const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, int padding, TlTaps *taps, long long *sp)
{
    constexpr int FB = TlMainLds::kFbBatch;
    const int nch = C->nch, sblimit = C->sblimit;
    PA(double, smp, 36);            // sb_sample[ch][gr][bl][sb] of this lane's (sb,ch), b = gr*12+bl
    PA(int, scf, 3);

    TL_STAMP(sp, 0);
    TL_PRIO2(TL_PS_FB);
    // ---- K1: polyphase filterbank ----
    tl_filterbank(w, B, enw_s, nch, smp);

    TL_STAMP(sp, 1);
    TL_PRIO2(1);
#if !defined(TL_EMULATE) && TL_ENC_LEVEL >= 5
    for (int b = 0; b < 36; b++) TL_KEEP(smp[b]);
    scf[0] = scf[1] = scf[2] = 0;
#endif
    // ---- K2: scalefactors (encode_new.c:179-230) + find_sf_max (:260-277) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (TL_ENC_LEVEL >= 5) { }
    else if (c < nch && sb < sblimit) {
        unsigned lo = 63;
        for (int gr = 0; gr < 3; gr++) {
            double m = fabs(L(smp)[gr * 12 + 11]);
            for (int j = 10; j >= 0; j--) { double t = fabs(L(smp)[gr * 12 + j]); if (t > m) m = t; }
            unsigned idx = tl_sf_index(B->scalefactor, m);
            L(scf)[gr] = (int)idx;
            w.scf[c][gr][sb] = (uint8_t)idx;
            if (idx < lo) lo = idx;
        }
        w.minidx[c][sb] = (uint8_t)lo;
    } else {
        L(scf)[0] = L(scf)[1] = L(scf)[2] = 0;
        if (sb >= sblimit || c >= nch) { w.minidx[c][sb] = 63; w.scf[c][0][sb] = w.scf[c][1][sb] = w.scf[c][2][sb] = 0; }
    }
    TL_LANES_END

    // joint stereo: scalefactors of .5*(L+R) (toolame.c:332-337, encode_new.c:237-246)
    if (TL_ENC_LEVEL < 5 && C->mode0 == 1) {
        for (int gr = 0; gr < 3; gr++) {
            PV(double, jm);
            TL_LANES_BEGIN L(jm) = 0.0; TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int j = 11; j >= 0; j--) {
                PV(double, other);
#ifdef TL_EMULATE
                for (int lane = 0; lane < 64; ++lane) other[lane] = smp[lane ^ 1][gr * 12 + j];
#else
                other = tld_swap1_f64(smp[gr * 12 + j]);
#endif
                TL_LANES_BEGIN
                double t = fabs(.5 * (L(smp)[gr * 12 + j] + L(other)));     // ch0 lane: .5*(L+R)
                if (j == 11 || t > L(jm)) L(jm) = t;
                TL_LANES_END
            }
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            if (c == 0 && sb < sblimit) w.jscale[gr][sb] = (uint8_t)tl_sf_index(B->scalefactor, L(jm));
            TL_LANES_END
        }
    }

    if (taps) {
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        for (int b = 0; b < 36; b++) taps->sb_sample[c][b / 12][b % 12][sb] = L(smp)[b];
        for (int gr = 0; gr < 3; gr++) { taps->scalar_pre[c][gr][sb] = w.scf[c][gr][sb]; if (c == 0) taps->j_scale[gr][sb] = C->mode0 == 1 && sb < sblimit ? w.jscale[gr][sb] : 0; }
        taps->max_sc[c][sb] = (c < nch && sb < sblimit) ? B->scalefactor[w.minidx[c][sb]] : 1E-20;
        TL_LANES_END
    }

    TL_STAMP(sp, 2);
    // ---- K3/K4: psychoacoustic model -> SMR (toolame.c:361-452) ----
    if constexpr (TL_ENC_LEVEL >= 5) { }
    else if constexpr (PSY == 0) {                                    // psycho_0.c:52-68
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        if (c < nch) {
            int m = sb < sblimit ? (int)w.minidx[c][sb] : 0;     // scalar[] above sblimit stays 0 (toolame.c:132)
            w.smr[c][sb] = 2.0 * (30.0 - m) - C->p0_athmin[sb];
        }
        TL_LANES_END
    } else if constexpr (PSY == 2) {
        TL_LANES_BEGIN                                           // models 2 and 4: the psy-2 kernel left the SMR itself
        const int c = lane & 1, sb = lane >> 1;
        if (c < nch) w.smr[c][sb] = PO->a[c][sb];
        TL_LANES_END
    } else {
        // models 1 and 3: the model (tl_frame_unit ran it before this frame body) left, per (channel, subband), the level A that
        // competes with the scalefactor level (in smr[]) and the minimum masking threshold m (in psy_m[]); the SMR line itself
        // needs this frame's scalefactors and is finished here:
        // psycho_1.c:575-580 (max = scale level; if (spike > max) max = spike; smr = max - ltmin) and psycho_3.c:180-182,428
        // (Lsb = max(Xmax, scale level); smr = Lsb - ltmin) are the same three operations.
        static_assert(PSY == TL_PSY_EXT, "models 1 and 3");
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        if (c < nch) {
            const double a = w.smr[c][sb], m = w.psy_m[c][sb];
            const double val = C->scale_db[w.minidx[c][sb]];
            const double top = a > val ? a : val;
            w.smr[c][sb] = top - m;
        }
        TL_LANES_END
    }

    TL_STAMP(sp, 3);
    // ---- sf_transmission_pattern (encode_new.c:288-354, ISO Table C.4) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (TL_ENC_LEVEL < 5 && c < nch && sb < sblimit) {
        int s0 = L(scf)[0], s1 = L(scf)[1], s2 = L(scf)[2];
        int d0 = s0 - s1, d1 = s1 - s2;
        int c0 = d0 <= -3 ? 0 : d0 < 0 ? 1 : d0 == 0 ? 2 : d0 < 3 ? 3 : 4;
        int c1 = d1 <= -3 ? 0 : d1 < 0 ? 1 : d1 == 0 ? 2 : d1 < 3 ? 3 : 4;
        // pattern of the class pair: where each transmitted scalefactor comes from, and scfsi (no branches: a lane per cell)
        const unsigned p = B->sfpat[c0 * 5 + c1];
        const int m02 = s0 > s2 ? s2 : s0;                              // pattern 444: the larger scalefactor (smaller index) of the outer two
        const unsigned q0 = p & 3u, q1 = (p >> 2) & 3u, q2 = (p >> 4) & 3u;
        const int n0 = q0 == 0 ? s0 : q0 == 1 ? s1 : q0 == 2 ? s2 : m02;
        const int n1 = q1 == 0 ? s0 : q1 == 1 ? s1 : q1 == 2 ? s2 : m02;
        const int n2 = q2 == 0 ? s0 : q2 == 1 ? s1 : q2 == 2 ? s2 : m02;
        const int sel = (int)(p >> 6);
        s0 = n0; s1 = n1; s2 = n2;
        L(scf)[0] = s0; L(scf)[1] = s1; L(scf)[2] = s2;
        w.scf[c][0][sb] = (uint8_t)s0; w.scf[c][1][sb] = (uint8_t)s1; w.scf[c][2][sb] = (uint8_t)s2;
        w.scfsi[c][sb] = (uint8_t)sel;
    } else w.scfsi[c][sb] = 0;
    w.balloc[c][sb] = 0;
    TL_LANES_END

    // ---- K5: bit allocation (encode_new.c:733-886, :634-705, :1061-1187) ----
    const int lg_frame = C->frame_bytes + padding;                  // availbits.c:64: (whole + extra) slots
    int adb = lg_frame * 8 - (C->dab_ext * 8 + (xpad_len ? xpad_len : 2) * 8);     // toolame.c:292-301
    int mode = C->mode0, mode_ext = C->mode_ext0, jsbound = C->jsbound0;
    // per-lane constants of the allocation loops
    PV(int, a_ln); PV(int, a_nbal); PV(int, a_sfs); PV(int, a_sfs_o); PV(double, a_smr); PV(double, a_smr_o);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = c < nch && sb < sblimit;
    L(a_ln) = live ? C->line[sb] : 0;
    L(a_nbal) = live ? C->nbal[sb] : 0;
    L(a_sfs) = live ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0;
    L(a_sfs_o) = (live && nch == 2) ? 6 * tl_sfs_count(w.scfsi[1 - c][sb]) : 0;
    L(a_smr) = live ? w.smr[c][sb] : 0.0;
    L(a_smr_o) = (live && nch == 2) ? w.smr[1 - c][sb] : 0.0;
    TL_LANES_END
    if (TL_ENC_LEVEL < 4 && C->mode0 == 1) {
        // try plain stereo, then jsbound 16, 12, 8, 4 (encode_new.c:803-819).  What a cell needs for "no audible noise"
        // (bits_for_nonoise_new, encode_new.c:634-705) does not depend on the trial: the SNR column of an allocation line
        // is increasing, so the first allocation that masks the cell's own SMR is the number of allocations that do not,
        // and above jsbound (where the search goes on against the other channel's SMR) it is the larger of the two counts.
        // Both counts and both prices are computed once; a trial only selects and sums.
        PV(int, nz_own); PV(int, nz_jnt);
        TL_LANES_BEGIN
        int bo = 0, bj = 0;
        if (lane < 2 * sblimit) {
            const int ln = L(a_ln), maxAlloc = (1 << L(a_nbal)) - 1;
            double sv[15];
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 15; q++) sv[q] = B->snr_line[ln][q];
            int n1 = 0, n2 = 0;
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int q = 0; q < 15; q++) {
                const bool inr = q < maxAlloc - 1;
                n1 += (inr && !((sv[q] - L(a_smr)) >= 0.0)) ? 1 : 0;
                n2 += (inr && !((sv[q] - L(a_smr_o)) >= 0.0)) ? 1 : 0;
            }
            n2 = n2 > n1 ? n2 : n1;
            bo = n1 > 0 ? B->bits12_line[ln][n1] + 2 + L(a_sfs) : 0;
            bj = n2 > 0 ? B->bits12_line[ln][n2] + 4 + L(a_sfs) + L(a_sfs_o) : 0;
        }
        L(nz_own) = bo; L(nz_jnt) = bj;
        TL_LANES_END
        mode = 0; mode_ext = 0; jsbound = sblimit;
        int tries = 0, try_ext = 4;
        for (;;) {
            PV(int, need);
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            int bitsn = 0;
            if (sb < sblimit && c < (sb < jsbound ? nch : 1))
                bitsn = ((nch == 2 && sb >= jsbound) ? L(nz_jnt) : L(nz_own)) + L(a_nbal);     // + the bbal share of this (sb,ch)
            L(need) = bitsn;
            TL_LANES_END
            int rq = 32 + 16 + TL_WAVE_SUM_I32(need);
            if (tries == 0) {
                if (rq > adb) { mode = 1; } else break;
            } else if (!(rq > adb && try_ext > 0)) { mode_ext = try_ext; break; }
            --try_ext; jsbound = 4 * (try_ext + 1); tries++;          // 16, 12, 8, 4
        }
    }
    int adb_left;
    {   // a_bit_allocation_new (encode_new.c:1078-1187).  Every lane carries its cell's order-preserving mnr key,
        // its ba and the price of its next step; one wave arg-min per iteration.
        // The reference marks a cell used=2 when its next step does not fit.  The bits left only shrink and a
        // cell's price only changes when it wins, so a cell that does not fit now never fits later: such cells
        // are left out of the arg-min right away (same result, no iterations spent on refusals).
        //
        // A cell's mnr only grows with its allocation (the SNR column of an allocation line is increasing), so the
        // greedy order is the merge of the cells' ascending key lists.  Every cell therefore carries the key and the
        // price of its next step AND of the step after it; with M = the smallest second key in the wave, the cells
        // whose next key is below M are exactly the greedy order's next events (no second step can come before
        // them).  If together they still fit, they are all taken in one round; the one-at-a-time loop takes over
        // when a round no longer fits (or is empty), so the refusal rule above is applied event by event.
        PV(int, ba);
        adb_left = tl_allocate(B, adb, nch, sblimit, jsbound, a_ln, a_nbal, a_sfs, a_sfs_o, a_smr, ba);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        w.balloc[c][sb] = (uint8_t)((c < nch && sb < sblimit) ? L(ba) : 0);
        TL_LANES_END
    }

    TL_STAMP(sp, 4);
    // ---- K6: header, CRC, bit_alloc, scfsi, scalefactors, quantised samples -> LDS frame ----
    uint32_t *frame = w.u.frame[0];
    TL_LANES_BEGIN
    for (int i = lane; i < ((lg_frame + 3) >> 2) + 2; i += 64) frame[i] = 0;      // this frame's words (+ 2: tl_put_bits48)
    TL_LANES_END
    PV(int, f_ba); PV(int, f_sel); PV(int, f_scf); PV(int, f_smp);
    PV(int, o_ba); PV(int, o_sel); PV(int, o_scf); PV(int, o_smp);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = c < nch && sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0;
    const bool own = sb < sblimit && c < (sb < jsbound ? nch : 1);     // transmits bit_alloc + samples
    L(f_ba) = own ? L(a_nbal) : 0;
    L(f_sel) = (live && ba) ? 2 : 0;
    L(f_scf) = (live && ba) ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0;
    L(f_smp) = (own && ba) ? B->bits12_line[L(a_ln)][ba] / 12 : 0;     // group * bits of the cell's quantiser class
    TL_LANES_END
    TL_WAVE_EXSCAN_I32(o_ba, f_ba); TL_WAVE_EXSCAN_I32(o_sel, f_sel);
    TL_WAVE_EXSCAN_I32(o_scf, f_scf); TL_WAVE_EXSCAN_I32(o_smp, f_smp);
    const int n_ba = TL_WAVE_SUM_I32(f_ba), n_sel = TL_WAVE_SUM_I32(f_sel);
    const int n_scf = TL_WAVE_SUM_I32(f_scf), n_smp = TL_WAVE_SUM_I32(f_smp);
    const int p_ba = 48, p_sel = p_ba + n_ba, p_scf = p_sel + n_sel, p_smp = p_scf + n_scf;

    if (TL_ENC_LEVEL < 3) {
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (lane == 0) {     // write_header (encode_new.c:356-373)
        uint32_t h = (0xfffu << 20) | ((uint32_t)C->version << 19) | (2u << 17) | (0u << 16)
                   | ((uint32_t)C->br_idx << 12) | ((uint32_t)C->fs_idx << 10) | ((uint32_t)padding << 9) | (0u << 8)
                   | ((uint32_t)mode << 6) | ((uint32_t)mode_ext << 4);
        TL_ATOMIC_OR(&frame[0], h);
    }
    const bool live = c < nch && sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0;
    if (L(f_ba)) tl_put_bits48(frame, p_ba + L(o_ba), (uint64_t)ba, L(f_ba));
    if (L(f_sel)) {
        const unsigned si = w.scfsi[c][sb];
        tl_put_bits48(frame, p_sel + L(o_sel), si, 2);
        // write_scalefactors (encode_new.c:428-443): scfsi 0 -> three, 1/3 -> first and last, 2 -> one; as one field
        const unsigned s0 = (unsigned)L(scf)[0], s1 = (unsigned)L(scf)[1], s2 = (unsigned)L(scf)[2];
        const unsigned f3 = (s0 << 12) | (s1 << 6) | s2, f2 = (s0 << 6) | s2;
        tl_put_bits48(frame, p_scf + L(o_scf), si == 0 ? f3 : si == 2 ? s0 : f2, L(f_scf));
    }
    TL_LANES_END
    }

    TL_PRIO2(TL_PS_Q);
    // quantise (encode_new.c:479-547) + write_samples_new (:560-598): 12 rounds of 3 samples
    if (TL_ENC_LEVEL < 2) {
        const bool any_joint = (nch == 2) && jsbound < sblimit;      // joint-coded subbands exist in this frame
        // per-lane constants of the frame: quantiser class and its coefficients, the three scalefactors
        PV(int, q_ba); PV(int, q_nb); PV(int, q_grp); PV(int, q_s2n); PV(int, q_steps);
        PV(double, q_a); PV(double, q_b); PV(double, q_s2nf); PA(double, q_sf, 3); PA(double, q_rsf, 3);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        const bool own = sb < sblimit && c < (sb < jsbound ? nch : 1);
        const int ba = own ? w.balloc[c][sb] : 0;
        const unsigned qi = ba ? B->qinfo_line[L(a_ln)][ba] : 0u;   // class, bits and grouping from the shared LDS copy
        const int q = (int)(qi & 31u);
        const bool joint = any_joint && sb >= jsbound;
        // The field of a triple (encode_new.c:574-592) is ONE Horner form A + M (v1 + M C): three separate codewords of nb bits are
        // v2 + 2^nb (v1 + 2^nb v0), a grouped codeword is v0 + steps (v1 + steps v2) -- M = 2^nb or steps, (A, C) = (v2, v0) or (v0, v2).
        L(q_ba) = ba; L(q_nb) = (int)((qi >> 5) & 31u); L(q_grp) = ((qi >> 10) & 1u) ? 3 : 1; L(q_s2n) = K->steps2n[q];
        L(q_steps) = ((qi >> 10) & 1u) ? 1 << L(q_nb) : K->steps[q];                                // M
        L(q_a) = K->qa[q]; L(q_b) = K->qb[q]; L(q_s2nf) = K->steps2n_f[q];
        for (int gr = 0; gr < 3; gr++) {
            L(q_sf)[gr] = B->scalefactor[joint ? w.jscale[gr][sb] : L(scf)[gr]];
            L(q_rsf)[gr] = 1.0 / L(q_sf)[gr];                          // one division per granule instead of twelve
        }
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int r = 0; r < 12; r++) {
            const int gr = r >> 2, j0 = (r & 3) * 3;
            PA(double, oth, 3);
#ifdef TL_EMULATE
            for (int lane = 0; lane < 64; ++lane) for (int x = 0; x < 3; x++) oth[lane][x] = smp[lane ^ 1][gr * 12 + j0 + x];
#else
            if (any_joint) {
#pragma unroll
                for (int x = 0; x < 3; x++) oth[x] = tld_swap1_f64(smp[gr * 12 + j0 + x]);
            } else { oth[0] = oth[1] = oth[2] = 0.0; }
#endif
            TL_LANES_BEGIN
            const int c = lane & 1, sb = lane >> 1;
            unsigned v[3] = {0, 0, 0};
            if (L(q_ba)) {
                const bool joint = any_joint && sb >= jsbound;
                const double sfv = L(q_sf)[gr], rsf = L(q_rsf)[gr];
                for (int x = 0; x < 3; x++) {
                    double s = L(smp)[gr * 12 + j0 + x];
                    if (joint) s = .5 * (s + L(oth)[x]);
                    double d = tl_div_by(s, sfv, rsf);                   // == s / sfv (encode_new.c:507,511)
                    d = d * L(q_a) + L(q_b);
                    const bool neg = !(d >= 0);                          // encode_new.c:528-534; d + 0.0 changes no quantised value
                    d += TL_SELECT(neg, 1.0, 0.0);
                    const unsigned qv = (unsigned)(d * L(q_s2nf));
                    v[x] = qv | (neg ? 0u : (unsigned)L(q_s2n));
                }
                // three codewords of nb bits, or one codeword v0 + v1*steps + v2*steps^2 of nb bits (encode_new.c:574-592): one field
                const int nb = L(q_nb);
                const int pos = p_smp + r * n_smp + L(o_smp);
                const bool three = L(q_grp) == 3;
                const unsigned M = (unsigned)L(q_steps);
                const unsigned fa = TL_SELECT(three, v[2], v[0]), fc = TL_SELECT(three, v[0], v[2]);
                const unsigned inner = v[1] + M * fc;                                    // < 2^32: nb <= 16
                tl_put_bits48(frame, pos, (uint64_t)fa + (uint64_t)M * (uint64_t)inner, L(q_grp) * nb);
            }
            if (taps) for (int x = 0; x < 3; x++) taps->subband[c][gr][j0 + x][sb] = (c < nch) ? v[x] : 0;
            TL_LANES_END
        }
    }

    TL_STAMP(sp, 5);
    TL_PRIO2(1);
    // CRC-16 over header bits 16..31, bit_alloc and scfsi fields (crc.c:12-41).
    // Protected message M = frame bits [16,32) then [48,p_scf), n bits.  The register after M with preset I is
    // (I(x) x^n + M(x) x^16) mod P -- linear over GF(2) -- so every lane takes one byte of M (a byte of the frame: the message is
    // byte aligned in it) and adds up bit_k * x^(16 + bits after the byte + k) mod P, starting from a table value and
    // multiplying by x per step; the two bytes of the preset ride on lanes 62/63; one XOR-reduce.
    unsigned crc16 = 0;
    if (TL_ENC_LEVEL < 1) {
    {
        const int n = 16 + (p_scf - 48);
        PV(uint32_t, part);
        TL_LANES_BEGIN
        uint32_t acc = 0;
        const bool preset = lane >= 62;
        const int first = 8 * lane;                                   // message bits [first, first + cnt)
        if (first < n || preset) {
            const int byte = lane < 2 ? lane + 2 : lane + 4;          // frame byte holding them
            const int cnt = preset ? 8 : (n - first < 8 ? n - first : 8);
            const int e0 = preset ? n + 8 * (63 - lane) : 16 + (n - first - cnt);     // exponent of the byte's last bit
            unsigned xp = K->crc_xpow[e0];
            const unsigned v = preset ? 0xffu : ((frame[byte >> 2] >> (24 - 8 * (byte & 3))) & 0xffu) >> (8 - cnt);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 8; k++) {                             // bits past cnt are zero
                acc ^= ((v >> k) & 1u) ? xp : 0u;
                xp = ((xp << 1) & 0xffffu) ^ ((xp & 0x8000u) ? 0x8005u : 0u);
            }
        }
        L(part) = acc;
        TL_LANES_END
        crc16 = TL_WAVE_XOR_U32(part) & 0xffffu;
    }
    // ScF-CRC (crc.c:58-97, toolame.c:527-542).  Every (sb,ch) lane packs the 3 MSBs of the scalefactors it
    // transmits (crc.c:83-96) and folds it on its own; the band groups are combined below.
    const int tail = lg_frame - 2 - C->dab_ext;                     // byte offset of the first ScF-CRC byte
    PV(int, rlen); PV(uint32_t, rcrc);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    uint32_t rec = 0;
    if (c < nch && sb < sblimit && w.balloc[c][sb]) {
        const uint32_t s0 = (uint32_t)L(scf)[0] >> 3, s1 = (uint32_t)L(scf)[1] >> 3, s2 = (uint32_t)L(scf)[2] >> 3;
        switch (w.scfsi[c][sb]) {
        case 0: rec = (9u << 16) | (s0 << 6) | (s1 << 3) | s2; break;
        case 1: case 3: rec = (6u << 16) | (s0 << 3) | s2; break;
        default: rec = (3u << 16) | s0; break;
        }
    }
    if (lane == 0) tl_put_bits(frame, 32, crc16, 16);
    L(rlen) = (int)(rec >> 16); L(rcrc) = rec & 0x1ffu;               // the record's bits; folded below (crc.c:99-113)
    TL_LANES_END
    // The CRC register update is linear over GF(2): the CRC of a band group (records concatenated in (sb,ch) order, crc.c:58-97)
    // is the XOR of rec_l(x) * x^(8 + bits after record l) mod P.  Bits-after from a prefix sum of the lengths, x^e from a table,
    // one XOR scan, then the four group values are differences of that scan at the group boundaries.
    {
        PV(int, lex);
        TL_WAVE_EXSCAN_I32(lex, rlen);
        const int f[5] = {0, 4, 8, 16, 30};
        int gend[4], gfirst[4], glast[4];
        for (int g = 0; g < 4; g++) {
            gfirst[g] = f[g]; glast[g] = f[g + 1] > sblimit ? sblimit : f[g + 1];
            const int e = 2 * glast[g];
            gend[g] = (g < C->dab_ext && glast[g] > gfirst[g]) ? TL_READLANE_I32(lex, e) : 0;      // e <= 60
        }
        PV(uint32_t, part); PV(uint32_t, pscan);
        TL_LANES_BEGIN
        const int sb = lane >> 1;
        const int g = sb < 4 ? 0 : sb < 8 ? 1 : sb < 16 ? 2 : 3;
        const int after = (g == 0 ? gend[0] : g == 1 ? gend[1] : g == 2 ? gend[2] : gend[3]) - L(lex) - L(rlen);
        const int e0 = after + 8;                                    // <= 252 + 8: inside crc8_xpow[]
        unsigned xp = K->crc8_xpow[e0 < 0 ? 0 : e0 > 319 ? 319 : e0];
        unsigned acc = 0;
        const unsigned rb = L(rcrc);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int b = 0; b < 9; b++) {                                // acc = rec(x) * x^(8 + after) mod P, shift-and-add in GF(2)
            acc ^= ((rb >> b) & 1u) ? xp : 0u;
            xp = ((xp << 1) & 0xffu) ^ ((xp & 0x80u) ? 0x1Du : 0u);
        }
        L(part) = (L(rlen) && sb < sblimit) ? acc : 0u;
        TL_LANES_END
        TL_WAVE_INCL_XSCAN_U32(pscan, part);
        unsigned c8g[4];
        for (int g = 0; g < 4; g++) {
            c8g[g] = 0;
            if (g < C->dab_ext && glast[g] > gfirst[g]) {
                c8g[g] = (unsigned)TL_READLANE_I32(pscan, 2 * glast[g] - 1);
                if (gfirst[g] > 0) c8g[g] ^= (unsigned)TL_READLANE_I32(pscan, 2 * gfirst[g] - 1);
            }
        }
        TL_LANES_BEGIN
        if (lane < C->dab_ext) {
            const int grp = C->dab_ext - 1 - lane;                  // transmission order: i = dab_ext-1 .. 0
            const unsigned c8 = (grp == 0 ? c8g[0] : grp == 1 ? c8g[1] : grp == 2 ? c8g[2] : c8g[3]) & 0xffu;
            tl_put_bits(frame, (tail + lane) * 8, c8, 8);
            w.ncentre[lane] = (int16_t)c8;                           // reused as a 4-entry scratch
        }
        TL_LANES_END
    }
    TL_LANES_BEGIN
    // X-PAD + F-PAD bytes (toolame.c:515-524,544-551): xpad[] holds xpad_len bytes in transmission order
    if (xpad_len) {
        const int xstart = lg_frame - C->dab_ext - xpad_len;        // X-PAD sits right before the ScF-CRC
        for (int i = lane; i < xpad_len; i += 64) {
            int bytepos = i < xpad_len - 2 ? xstart + i : lg_frame - 2 + (i - (xpad_len - 2));
            tl_put_bits(frame, bytepos * 8, w.xpad[i], 8);
        }
    }
    TL_LANES_END

    } else {
        TL_LANES_BEGIN
        if (lane < 4) w.ncentre[lane] = 0;
        TL_LANES_END
    }
    if (taps) {
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        taps->smr[c][sb] = (c < nch && (C->psy != 1 || sb < sblimit)) ? w.smr[c][sb] : 0.0;   // psy 1 leaves sb >= sblimit unset
        taps->scfsi[c][sb] = w.scfsi[c][sb]; taps->bit_alloc[c][sb] = w.balloc[c][sb];
        for (int gr = 0; gr < 3; gr++) taps->scalar[c][gr][sb] = w.scf[c][gr][sb];
        if (lane == 0) { taps->adb_left = adb_left; taps->mode = mode; taps->mode_ext = mode_ext; taps->jsbound = jsbound; taps->crc16 = (int)crc16; }
        if (lane < 4) taps->scfcrc[lane] = lane < C->dab_ext ? (uint8_t)w.ncentre[lane] : 0;
        TL_LANES_END
    }

    TL_STAMP(sp, 6);
    // ---- emit (toolame.c:527-542 keeps "one frame in memory" to patch its ScF-CRC slot with the next frame's CRC) ----
    const int nwords = (lg_frame + 3) >> 2;
    // frames of a stream are encoded by different waves in any order: this one only files its frame and its ScF-CRC;
    // tl_finish_stream puts each frame's CRC into the frame before it once the launch's frames are all there
    TL_LANES_BEGIN
    for (int i = lane; i < nwords; i += 64) {
        if (fo.words) fo.words[i] = frame[i];
        else {
            const uint32_t le = tl_bswap(frame[i]);
            const int rem = lg_frame - 4 * i;
            if (rem >= 4) ((uint32_t *)fo.bytes)[i] = le;
            else for (int b = 0; b < rem; b++) fo.bytes[4 * i + b] = (uint8_t)(le >> (8 * b));
        }
    }
    if (lane < 4) fo.scfcrc[lane] = lane < C->dab_ext ? (uint8_t)w.ncentre[lane] : 0;
    TL_LANES_END
    TL_PRIO2(0);
    TL_STAMP(sp, 7);
}

// ------------------------------------------------------------------------------------------
// TWO mono streams of ONE configuration in one wave.  A lone mono frame leaves every second lane idle from the filterbank to the
// packing (lane = 2*sb + ch, ch = 0 only) and costs as many instructions as a stereo frame; here lane = 2*sb + u owns subband sb of
// UNIT u, the u-th of the two streams (same frame index f of the launch).  Filterbank, scalefactors, transmission pattern, quantiser
// and packing run for both units at once, the bit allocation too (tl_allocate_pair); the CRC-16 folds use one half-wave per unit.  What a unit produces is what
// tl_encode_frame produces for it alone: the operations per cell are the same text, and every wave-level sum, scan and minimum is
// taken over the unit's own lanes (a scan over both units carries unit 0 in the low and unit 1 in the high half of a word:
// sums stay below 2^16, XORs never carry).  toolame.c:267-554 twice, the `nch` loop of toolame.c:308-312 turned into lanes.
template <int PSY>
TL_FN void tl_encode_pair(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const TlConfig *TL_RESTRICT C, const TlPsyOut *const (&PO)[2],
                          const int (&xpad_len)[2], const uint8_t *const (&xpad_src)[2], const TlFrameOut (&fo)[2],
                          const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, const int (&padding)[2])
{
    const int sblimit = C->sblimit;
    const int padpk = padding[0] | (padding[1] << 1);             // both units' padding bits in one scalar (an array indexed by the lane would live in scratch)
    PA(double, smp, 36);
    PA(int, scf, 3);
    TL_PRIO2(TL_PS_FB);
    tl_filterbank(w, B, enw_s, 2, smp);
    TL_PRIO2(1);
    // ---- scalefactors (encode_new.c:179-230) + find_sf_max (:260-277) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (sb < sblimit) {
        unsigned lo = 63;
        for (int gr = 0; gr < 3; gr++) {
            double m = fabs(L(smp)[gr * 12 + 11]);
            for (int j = 10; j >= 0; j--) { double t = fabs(L(smp)[gr * 12 + j]); if (t > m) m = t; }
            unsigned idx = tl_sf_index(B->scalefactor, m);
            L(scf)[gr] = (int)idx;
            if (idx < lo) lo = idx;
        }
        w.minidx[c][sb] = (uint8_t)lo;
    } else {
        L(scf)[0] = L(scf)[1] = L(scf)[2] = 0;
        w.minidx[c][sb] = 63;
    }
    TL_LANES_END
    // ---- SMR (toolame.c:361-452), as in tl_encode_frame with c = the unit ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if constexpr (PSY == 0) {
        const int m = sb < sblimit ? (int)w.minidx[c][sb] : 0;
        w.smr[c][sb] = 2.0 * (30.0 - m) - C->p0_athmin[sb];
    } else if constexpr (PSY == 2) {
        w.smr[c][sb] = (c ? PO[1] : PO[0])->a[0][sb];
    } else {
        const double a = w.smr[c][sb], m = w.psy_m[c][sb];
        const double val = C->scale_db[w.minidx[c][sb]];
        const double top = a > val ? a : val;
        w.smr[c][sb] = top - m;
    }
    TL_LANES_END
    // ---- sf_transmission_pattern (encode_new.c:288-354, ISO Table C.4) ----
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    if (sb < sblimit) {
        int s0 = L(scf)[0], s1 = L(scf)[1], s2 = L(scf)[2];
        int d0 = s0 - s1, d1 = s1 - s2;
        int c0 = d0 <= -3 ? 0 : d0 < 0 ? 1 : d0 == 0 ? 2 : d0 < 3 ? 3 : 4;
        int c1 = d1 <= -3 ? 0 : d1 < 0 ? 1 : d1 == 0 ? 2 : d1 < 3 ? 3 : 4;
        const unsigned p = B->sfpat[c0 * 5 + c1];
        const int m02 = s0 > s2 ? s2 : s0;
        const unsigned q0 = p & 3u, q1 = (p >> 2) & 3u, q2 = (p >> 4) & 3u;
        const int n0 = q0 == 0 ? s0 : q0 == 1 ? s1 : q0 == 2 ? s2 : m02;
        const int n1 = q1 == 0 ? s0 : q1 == 1 ? s1 : q1 == 2 ? s2 : m02;
        const int n2 = q2 == 0 ? s0 : q2 == 1 ? s1 : q2 == 2 ? s2 : m02;
        L(scf)[0] = n0; L(scf)[1] = n1; L(scf)[2] = n2;
        w.scfsi[c][sb] = (uint8_t)(p >> 6);
    } else w.scfsi[c][sb] = 0;
    w.balloc[c][sb] = 0;
    TL_LANES_END
    // ---- bit allocation (encode_new.c:733-886, :1061-1187): both units at once, each over its own lanes ----
    int lg_frame[2];
    {
        PV(int, a_ln); PV(int, a_nbal); PV(int, a_sfs); PV(double, a_smr); PV(int, ba);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        const bool live = sb < sblimit;
        L(a_ln) = live ? C->line[sb] : 0;
        L(a_nbal) = live ? C->nbal[sb] : 0;
        L(a_sfs) = live ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0;
        L(a_smr) = live ? w.smr[c][sb] : 0.0;
        TL_LANES_END
        int adb[2];
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int u = 0; u < 2; u++) {
            lg_frame[u] = C->frame_bytes + padding[u];                                       // availbits.c:64
            adb[u] = lg_frame[u] * 8 - (C->dab_ext * 8 + (xpad_len[u] ? xpad_len[u] : 2) * 8);    // toolame.c:292-301
        }
        tl_allocate_pair(B, adb[0], adb[1], sblimit, a_ln, a_nbal, a_sfs, a_smr, ba);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        w.balloc[c][sb] = (uint8_t)(sb < sblimit ? L(ba) : 0);
        TL_LANES_END
    }
    // ---- header, bit_alloc, scfsi, scalefactors, quantised samples -> the two LDS frames ----
    TL_LANES_BEGIN
    for (int i = lane; i < ((lg_frame[0] + 3) >> 2) + 2; i += 64) w.u.frame[0][i] = 0;
    for (int i = lane; i < ((lg_frame[1] + 3) >> 2) + 2; i += 64) w.u.frame[1][i] = 0;
    TL_LANES_END
    PV(int, f_ba); PV(int, f_sel); PV(int, f_scf); PV(int, f_smp); PV(int, a_ln2);
    PV(int, o_ba); PV(int, o_sel); PV(int, o_scf); PV(int, o_smp);
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1;
    const bool live = sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0, sh = 16 * c;
    L(a_ln2) = live ? C->line[sb] : 0;
    L(f_ba) = (live ? (int)C->nbal[sb] : 0) << sh;
    L(f_sel) = (ba ? 2 : 0) << sh;
    L(f_scf) = (ba ? 6 * tl_sfs_count(w.scfsi[c][sb]) : 0) << sh;
    L(f_smp) = (ba ? B->bits12_line[L(a_ln2)][ba] / 12 : 0) << sh;
    TL_LANES_END
    TL_WAVE_EXSCAN_I32(o_ba, f_ba); TL_WAVE_EXSCAN_I32(o_sel, f_sel);
    TL_WAVE_EXSCAN_I32(o_scf, f_scf); TL_WAVE_EXSCAN_I32(o_smp, f_smp);
    const int s_ba = TL_WAVE_SUM_I32(f_ba), s_sel = TL_WAVE_SUM_I32(f_sel), s_scf = TL_WAVE_SUM_I32(f_scf), s_smp = TL_WAVE_SUM_I32(f_smp);
    int p_sel[2], p_scf[2], p_smp[2], n_smp[2];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++) {
        const int sh = 16 * u;
        p_sel[u] = 48 + ((s_ba >> sh) & 0xffff); p_scf[u] = p_sel[u] + ((s_sel >> sh) & 0xffff); p_smp[u] = p_scf[u] + ((s_scf >> sh) & 0xffff);
        n_smp[u] = (s_smp >> sh) & 0xffff;
    }
    TL_LANES_BEGIN
    const int c = lane & 1, sb = lane >> 1, sh = 16 * c;
    uint32_t *frame = w.u.frame[c];
    if (lane < 2) {      // write_header (encode_new.c:356-373), lane u for unit u
        uint32_t h = (0xfffu << 20) | ((uint32_t)C->version << 19) | (2u << 17) | (0u << 16)
                   | ((uint32_t)C->br_idx << 12) | ((uint32_t)C->fs_idx << 10) | ((uint32_t)((padpk >> c) & 1) << 9) | (0u << 8)
                   | ((uint32_t)C->mode0 << 6) | ((uint32_t)C->mode_ext0 << 4);
        TL_ATOMIC_OR(&frame[0], h);
    }
    const bool live = sb < sblimit;
    const int ba = live ? w.balloc[c][sb] : 0;
    const int nb_ba = (L(f_ba) >> sh) & 0xffff;
    if (nb_ba) tl_put_bits48(frame, 48 + ((L(o_ba) >> sh) & 0xffff), (uint64_t)ba, nb_ba);
    if (ba) {
        const unsigned si = w.scfsi[c][sb];
        tl_put_bits48(frame, (c ? p_sel[1] : p_sel[0]) + ((L(o_sel) >> sh) & 0xffff), si, 2);
        const unsigned s0 = (unsigned)L(scf)[0], s1 = (unsigned)L(scf)[1], s2 = (unsigned)L(scf)[2];
        const unsigned f3 = (s0 << 12) | (s1 << 6) | s2, f2 = (s0 << 6) | s2;
        tl_put_bits48(frame, (c ? p_scf[1] : p_scf[0]) + ((L(o_scf) >> sh) & 0xffff), si == 0 ? f3 : si == 2 ? s0 : f2, (L(f_scf) >> sh) & 0xffff);
    }
    TL_LANES_END
    TL_PRIO2(TL_PS_Q);
    // quantise (encode_new.c:479-547) + write_samples_new (:560-598): 12 rounds of 3 samples
    {
        PV(int, q_ba); PV(int, q_nb); PV(int, q_grp); PV(int, q_s2n); PV(int, q_steps); PV(int, q_pos); PV(int, q_rstep);
        PV(double, q_a); PV(double, q_b); PV(double, q_s2nf); PA(double, q_sf, 3); PA(double, q_rsf, 3);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1, sh = 16 * c;
        const int ba = sb < sblimit ? w.balloc[c][sb] : 0;
        const unsigned qi = ba ? B->qinfo_line[L(a_ln2)][ba] : 0u;
        const int q = (int)(qi & 31u);
        L(q_ba) = ba; L(q_nb) = (int)((qi >> 5) & 31u); L(q_grp) = ((qi >> 10) & 1u) ? 3 : 1; L(q_s2n) = K->steps2n[q];
        L(q_steps) = ((qi >> 10) & 1u) ? 1 << L(q_nb) : K->steps[q];                                // M of the field's Horner form (tl_encode_frame)
        L(q_a) = K->qa[q]; L(q_b) = K->qb[q]; L(q_s2nf) = K->steps2n_f[q];
        for (int gr = 0; gr < 3; gr++) {
            L(q_sf)[gr] = B->scalefactor[L(scf)[gr]];
            L(q_rsf)[gr] = 1.0 / L(q_sf)[gr];
        }
        L(q_pos) = (c ? p_smp[1] : p_smp[0]) + ((L(o_smp) >> sh) & 0xffff);
        L(q_rstep) = c ? n_smp[1] : n_smp[0];
        TL_LANES_END
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int r = 0; r < 12; r++) {
            const int gr = r >> 2, j0 = (r & 3) * 3;
            TL_LANES_BEGIN
            if (L(q_ba)) {
                const double sfv = L(q_sf)[gr], rsf = L(q_rsf)[gr];
                unsigned v[3];
                for (int x = 0; x < 3; x++) {
                    double d = tl_div_by(L(smp)[gr * 12 + j0 + x], sfv, rsf);                // == s / sfv (encode_new.c:507,511)
                    d = d * L(q_a) + L(q_b);
                    const bool neg = !(d >= 0);                                            // encode_new.c:528-534
                    d += TL_SELECT(neg, 1.0, 0.0);
                    const unsigned qv = (unsigned)(d * L(q_s2nf));
                    v[x] = qv | (neg ? 0u : (unsigned)L(q_s2n));
                }
                const int nb = L(q_nb);
                const bool three = L(q_grp) == 3;
                const unsigned M = (unsigned)L(q_steps);
                const unsigned fa = TL_SELECT(three, v[2], v[0]), fc = TL_SELECT(three, v[0], v[2]);
                const unsigned inner = v[1] + M * fc;
                tl_put_bits48(w.u.frame[lane & 1], L(q_pos) + r * L(q_rstep), (uint64_t)fa + (uint64_t)M * (uint64_t)inner, L(q_grp) * nb);
            }
            TL_LANES_END
        }
    }
    TL_PRIO2(1);
    // ---- CRC-16 (crc.c:12-41) of both frames: lanes 0..31 fold unit 0's message bytes, lanes 32..63 unit 1's (a mono frame protects
    //      at most 16 + 94 + 60 bits: 22 bytes; the preset's two bytes ride on lanes 30/31 of each half) ----
    {
        PV(uint32_t, part0); PV(uint32_t, part1);
        TL_LANES_BEGIN
        const int u = lane >> 5, l5 = lane & 31;
        const uint32_t *frame = w.u.frame[u];
        const int n = 16 + ((u ? p_scf[1] : p_scf[0]) - 48);
        uint32_t acc = 0;
        const bool preset = l5 >= 30;
        const int first = 8 * l5;
        if (first < n || preset) {
            const int byte = l5 < 2 ? l5 + 2 : l5 + 4;
            const int cnt = preset ? 8 : (n - first < 8 ? n - first : 8);
            const int e0 = preset ? n + 8 * (31 - l5) : 16 + (n - first - cnt);
            unsigned xp = K->crc_xpow[e0];
            const unsigned v = preset ? 0xffu : ((frame[byte >> 2] >> (24 - 8 * (byte & 3))) & 0xffu) >> (8 - cnt);
#ifndef TL_EMULATE
#pragma unroll
#endif
            for (int k = 0; k < 8; k++) {
                acc ^= ((v >> k) & 1u) ? xp : 0u;
                xp = ((xp << 1) & 0xffffu) ^ ((xp & 0x8000u) ? 0x8005u : 0u);
            }
        }
        L(part0) = u == 0 ? acc : 0u; L(part1) = u == 1 ? acc : 0u;
        TL_LANES_END
        const unsigned crc0 = TL_WAVE_XOR_U32(part0) & 0xffffu, crc1 = TL_WAVE_XOR_U32(part1) & 0xffffu;
        TL_LANES_BEGIN
        if (lane < 2) tl_put_bits(w.u.frame[lane], 32, lane ? crc1 : crc0, 16);
        TL_LANES_END
    }
    // ---- ScF-CRC (crc.c:58-97, toolame.c:527-542), both units through ONE sum scan and ONE XOR scan (unit u in bits 16u..16u+15) ----
    {
        PV(int, rlen); PV(uint32_t, rcrc); PV(int, rl2); PV(int, lex);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1;
        uint32_t rec = 0;
        if (sb < sblimit && w.balloc[c][sb]) {
            const uint32_t s0 = (uint32_t)L(scf)[0] >> 3, s1 = (uint32_t)L(scf)[1] >> 3, s2 = (uint32_t)L(scf)[2] >> 3;
            switch (w.scfsi[c][sb]) {
            case 0: rec = (9u << 16) | (s0 << 6) | (s1 << 3) | s2; break;
            case 1: case 3: rec = (6u << 16) | (s0 << 3) | s2; break;
            default: rec = (3u << 16) | s0; break;
            }
        }
        L(rlen) = (int)(rec >> 16); L(rcrc) = rec & 0x1ffu; L(rl2) = L(rlen) << (16 * c);
        TL_LANES_END
        TL_WAVE_EXSCAN_I32(lex, rl2);
        const int f[5] = {0, 4, 8, 16, 30};
        int gend[4], gfirst[4], glast[4];
        for (int g = 0; g < 4; g++) {
            gfirst[g] = f[g]; glast[g] = f[g + 1] > sblimit ? sblimit : f[g + 1];
            gend[g] = (g < C->dab_ext && glast[g] > gfirst[g]) ? TL_READLANE_I32(lex, 2 * glast[g]) : 0;      // both units' sums, packed; lane <= 60
        }
        PV(uint32_t, part); PV(uint32_t, pscan);
        TL_LANES_BEGIN
        const int c = lane & 1, sb = lane >> 1, sh = 16 * c;
        const int g = sb < 4 ? 0 : sb < 8 ? 1 : sb < 16 ? 2 : 3;
        const int ge = ((g == 0 ? gend[0] : g == 1 ? gend[1] : g == 2 ? gend[2] : gend[3]) >> sh) & 0xffff;
        const int after = ge - ((L(lex) >> sh) & 0xffff) - L(rlen);
        const int e0 = after + 8;
        unsigned xp = K->crc8_xpow[e0 < 0 ? 0 : e0 > 319 ? 319 : e0];
        unsigned acc = 0;
        const unsigned rb = L(rcrc);
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int b = 0; b < 9; b++) {
            acc ^= ((rb >> b) & 1u) ? xp : 0u;
            xp = ((xp << 1) & 0xffu) ^ ((xp & 0x80u) ? 0x1Du : 0u);
        }
        L(part) = (L(rlen) && sb < sblimit) ? acc << sh : 0u;
        TL_LANES_END
        TL_WAVE_INCL_XSCAN_U32(pscan, part);
        unsigned c8g[4];                                             // per group: unit 0's CRC in bits 0..7, unit 1's in bits 16..23
        for (int g = 0; g < 4; g++) {
            c8g[g] = 0;
            if (g < C->dab_ext && glast[g] > gfirst[g]) {
                c8g[g] = (unsigned)TL_READLANE_I32(pscan, 2 * glast[g] - 1);
                if (gfirst[g] > 0) c8g[g] ^= (unsigned)TL_READLANE_I32(pscan, 2 * gfirst[g] - 1);
            }
        }
        TL_LANES_BEGIN
        const int u = lane >> 5, l5 = lane & 31;
        if (l5 < C->dab_ext) {
            const int grp = C->dab_ext - 1 - l5;                     // transmission order: i = dab_ext-1 .. 0
            const unsigned c8 = ((grp == 0 ? c8g[0] : grp == 1 ? c8g[1] : grp == 2 ? c8g[2] : c8g[3]) >> (16 * u)) & 0xffu;
            const int tail = (u ? lg_frame[1] : lg_frame[0]) - 2 - C->dab_ext;
            tl_put_bits(w.u.frame[u], (tail + l5) * 8, c8, 8);
            w.ncentre[4 * u + l5] = (int16_t)c8;
        }
        TL_LANES_END
    }
    // ---- X-PAD + F-PAD bytes (toolame.c:515-524,544-551), straight from the launch's X-PAD records ----
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++)
        if (xpad_len[u]) {
            const int xl = xpad_len[u], xstart = lg_frame[u] - C->dab_ext - xl;
            TL_LANES_BEGIN
            for (int i = lane; i < xl; i += 64) {
                const int bytepos = i < xl - 2 ? xstart + i : lg_frame[u] - 2 + (i - (xl - 2));
                tl_put_bits(w.u.frame[u], bytepos * 8, xpad_src[u][i], 8);
            }
            TL_LANES_END
        }
    // ---- emit: each unit files its frame and its ScF-CRC (tl_finish_stream puts the CRC into the frame before it) ----
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++) {
        const int nwords = (lg_frame[u] + 3) >> 2;
        const uint32_t *frame = w.u.frame[u];
        TL_LANES_BEGIN
        for (int i = lane; i < nwords; i += 64) {
            if (fo[u].words) fo[u].words[i] = frame[i];
            else {
                const uint32_t le = tl_bswap(frame[i]);
                const int rem = lg_frame[u] - 4 * i;
                if (rem >= 4) ((uint32_t *)fo[u].bytes)[i] = le;
                else for (int b = 0; b < rem; b++) fo[u].bytes[4 * i + b] = (uint8_t)(le >> (8 * b));
            }
        }
        if (lane < 4) fo[u].scfcrc[lane] = lane < C->dab_ext ? (uint8_t)w.ncentre[4 * u + lane] : 0;
        TL_LANES_END
    }
    TL_PRIO2(0);
}

// ------------------------------------------------------------------------------------------
// One step of the padding recurrence (availbits.c:49-62): does the next frame carry a padding slot?  fp64 as in the reference.
TL_FN int tl_slot_step(double &lag, double frac)
{
    if (frac == 0) return 0;
    if (lag > (frac - 1.0)) { lag -= frac; return 0; }
    lag += (1 - frac);
    return 1;
}
// Split path, 44.1 / 22.05 kHz only: the recurrence is sequential, the frames are not -- so one lane per stream runs it over
// the launch's frames first and leaves every frame's padding bit for the units (and the state after the launch for the finish pass).
TL_FN void tl_slots_stream(const TlLaunch &A, int s)
{
    const double frac = A.configs[A.stream_cfg[s]].pad_frac;
    double lag = A.state[s].slot_lag;
    for (int f = 0; f < A.nframes; f++) A.padbits[(size_t)f * (size_t)A.nstreams + (size_t)s] = (uint8_t)tl_slot_step(lag, frac);
    A.newlag[s] = lag;
}

// A stream's PCM around frame f of a launch: the frame itself and the 480 samples per channel before it (the stream state
// on the first frame of a launch, the previous input frame after).
TL_FN TlPcmView tl_pcm_view(const TlLaunch &A, const TlStreamState *st, int s, int f)
{
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    TlPcmView pv;
    pv.cur[0] = A.pcm + slot * 2304; pv.cur[1] = pv.cur[0] + 1152;
    pv.hist[0] = f == 0 ? &st->hist[0][0] : A.pcm + (slot - (size_t)A.nstreams) * 2304 + (1152 - TL_HIST);
    pv.hist[1] = f == 0 ? &st->hist[1][0] : pv.hist[0] + 1152;
    return pv;
}

// The same for a PAIR of mono streams sharing a wave: "channel" u is channel 0 of stream s[u]
TL_FN TlPcmView tl_pcm_view_pair(const TlLaunch &A, int s0, int s1, int f)
{
    const TlPcmView a = tl_pcm_view(A, &A.state[s0], s0, f), b = tl_pcm_view(A, &A.state[s1], s1, f);
    TlPcmView pv;
    pv.cur[0] = a.cur[0]; pv.hist[0] = a.hist[0]; pv.cur[1] = b.cur[0]; pv.hist[1] = b.hist[0];
    return pv;
}

// One unit of the psy kernel (models 1 and 3): both channels of frame f of stream s -> A.psy_out[f][s].  The model reads
// nothing but PCM (the window of a frame: the last 192 samples before it and its first 832), so units are independent of each
// other -- of other streams AND of other frames of the same stream -- and the kernel runs them in any order on any wave.
template <int PSY>
TL_FN void tl_psy_unit(TlPsyLds &w, const double *TL_RESTRICT db, const TlLaunch &A, int s, int f, PARGA(double, rec, 4), int s2 = -1)
{   // s2 >= 0: a PAIR of mono streams of one configuration -- the model runs its two-channel form on channel 0 of s and of s2
    // rec: the model's result per subband, in the registers of lane = subband: [ch] the level that competes with the scalefactor
    // level, [2 + ch] the minimum masking threshold (SMR = max(level, scale_db[min scalefactor index]) - threshold is the encoder's
    // line: psycho_1.c:568-581 with level = spike level; psycho_3.c:163-183,409-432 with level = strongest line of the subband)
    const TlTables *T = A.tables;
    const TlConfig *C = &A.configs[A.stream_cfg[s]];
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    const TlPcmView pv = s2 >= 0 ? tl_pcm_view_pair(A, s, s2, f) : tl_pcm_view(A, &A.state[s], s, f);
    TL_LANES_BEGIN
    L(rec)[0] = 0.0; L(rec)[1] = 0.0; L(rec)[2] = 0.0; L(rec)[3] = 0.0;     // the model writes every subband of the channels it runs
    TL_LANES_END
#ifdef TL_NO_PSY_STAMPS
    long long *sp = nullptr;
#else
    long long *sp = A.stamps ? A.stamps + slot * 32 : nullptr;
#endif
    TL_STAMP(sp, 15);                                                 // unit begin (slots 8..14 / 16..22: the channels' stages, 24..30: FHT passes)
    if constexpr (TL_EXP_LEVEL >= 9) { }                              // diagnostic build: no model at all (tools/class_budget.sh: what the encoder phase alone issues)
    else if constexpr (PSY == 1) {
        if (C->nch == 2 || s2 >= 0) tl_psy1_stereo(w, T, db, C, pv, rec, sp);
        else tl_psy1(w, T, db, C, pv, 0, rec, sp ? sp + 8 : nullptr);
    } else {
        if (C->nch == 2 || s2 >= 0) tl_psy3_stereo(w, T, db, C, pv, rec, sp);
        else tl_psy3(w, T, db, C, pv, 0, rec, sp ? sp + 8 : nullptr);
    }
    TL_STAMP(sp, 23);                                                 // unit end
}

// Models 2 and 4 on the split path.  One unit = frames [f0, f1) of ONE channel of one stream, in order (the two channels of a
// stream share nothing).  The r/phi prediction state of the run lives in the wave's registers (tl_psy2_pass).  Where a run starts
// at the launch's first frame the state comes from the stream's record (what the previous launch left; the passes before it
// are PCM this launch cannot see); anywhere else two seed passes over frame f0 - 1 rebuild it.  The run that ends the launch
// leaves the state for the next one -- in the OTHER of the record's two copies, so that it can never be read by a run of the
// same launch that starts at frame 0 and is scheduled later.  It leaves the SMR itself in TlPsyOut::a (the model's last line
// needs no scalefactors).
TL_FN void tl_psy2_chain(TlPsy2Lds &w, const TlLaunch &A, int s, int ch, int f0, int f1, const uint64_t *sct)
{
    const TlConfig *C = &A.configs[A.stream_cfg[s]];
    if (ch >= C->nch || f0 >= f1) return;
    const TlPsy2Tables *P = &A.psy2_tables[C->psy2_tab];
    PA(double, r1, 8); PA(double, r2, 8); PA(double, p1, 8); PA(double, p2, 8); PV(double, snr0);
    double *l5 = TL_P2_L512(w);
    if (f0 == 0) {
        const TlPsy2State *S = &A.psy2_state[2 * (size_t)s + (size_t)A.psy2_flip];
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) {
            const int j = lane + 64 * it;
            L(r1)[it] = S->r[ch][0][j]; L(r2)[it] = S->r[ch][1][j]; L(p1)[it] = S->phi[ch][0][j]; L(p2)[it] = S->phi[ch][1][j];
        }
        if (lane == 0) { l5[0] = S->r[ch][0][512]; l5[1] = S->r[ch][1][512]; l5[2] = S->phi[ch][0][512]; l5[3] = S->phi[ch][1][512]; }
        L(snr0) = 0.0;
        TL_LANES_END
    } else {
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) { L(r1)[it] = 0.0; L(r2)[it] = 0.0; L(p1)[it] = 0.0; L(p2)[it] = 0.0; }
        if (lane == 0) { l5[0] = 0.0; l5[1] = 0.0; l5[2] = 0.0; l5[3] = 0.0; }
        L(snr0) = 0.0;
        TL_LANES_END
        const TlPcmView pv = tl_pcm_view(A, &A.state[s], s, f0 - 1);
        tl_psy2_pass<true>(w, A.tables, P, pv, ch, 0, r1, r2, p1, p2, snr0, nullptr, sct, nullptr);
        tl_psy2_pass<true>(w, A.tables, P, pv, ch, 1, r1, r2, p1, p2, snr0, nullptr, sct, nullptr);
    }
    for (int f = f0; f < f1; f++) {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
        const TlPcmView pv = tl_pcm_view(A, &A.state[s], s, f);
        long long *sp = A.stamps ? A.stamps + slot * 32 + 8 + 8 * ch : nullptr;
        tl_psy2_pass<false>(w, A.tables, P, pv, ch, 0, r1, r2, p1, p2, snr0, &A.psy_out[slot].a[ch][0], sct, sp);
        tl_psy2_pass<false>(w, A.tables, P, pv, ch, 1, r1, r2, p1, p2, snr0, &A.psy_out[slot].a[ch][0], sct, nullptr);
    }
    if (f1 == A.nframes) {
        TlPsy2State *S = &A.psy2_state[2 * (size_t)s + (size_t)(1 - A.psy2_flip)];
        TL_LANES_BEGIN
#ifndef TL_EMULATE
#pragma unroll
#endif
        for (int it = 0; it < 8; it++) {
            const int j = lane + 64 * it;
            S->r[ch][0][j] = L(r1)[it]; S->r[ch][1][j] = L(r2)[it]; S->phi[ch][0][j] = L(p1)[it]; S->phi[ch][1][j] = L(p2)[it];
        }
        if (lane == 0) { S->r[ch][0][512] = l5[0]; S->r[ch][1][512] = l5[1]; S->phi[ch][0][512] = l5[2]; S->phi[ch][1][512] = l5[3]; }
        TL_LANES_END
    }
}
// Unit u of the psy-2 kernel's work list -> (chain, first frame, end frame).  The launch's chains (TlLaunch::chain_list: first
// channels of the list's streams, then the second channels of its stereo streams) are dealt to the waves longest first: chains
// [0, p2_nwhole) as ONE unit each, every chain after them cut into p2_k runs of p2_plen frames -- so that the last round of
// waves is as full as the ones before it (the host picks the cut, tl_psy2_plan in mp2_host.cpp).
TL_FN bool tl_psy2_unit(const TlLaunch &A, int u, int &chain, int &f0, int &f1)
{
    if (u < A.p2_nwhole) { chain = u; f0 = 0; f1 = A.nframes; return true; }
    const int v = u - A.p2_nwhole, k = A.p2_k;
    chain = A.p2_nwhole + v / k;
    f0 = (v - (v / k) * k) * A.p2_plen;
    f1 = f0 + A.p2_plen < A.nframes ? f0 + A.p2_plen : A.nframes;
    return f0 < f1;
}

// [history | frame] -> LDS in 8-byte pieces, 120 + 288 per channel.  All of a lane's loads are issued before the first LDS
// write so the HBM latency is paid once per frame, not once per piece.
TL_FN void tl_stage_pcm(TlMainLds &w, const TlPcmView &pv, int nch)
{
    TL_LANES_BEGIN
    {
        constexpr int HP = TL_HIST / 4, CP = 1152 / 4, PER = HP + CP;      // pieces per channel
        constexpr int NIT = (2 * PER + 63) / 64;
        uint64_t v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = lane + 64 * it, ch = i >= PER ? 1 : 0, k = i - ch * PER;
            v[it] = 0;
            if (i < PER * nch)
                v[it] = k < HP ? *(const uint64_t *)((ch ? pv.hist[1] : pv.hist[0]) + 4 * k) : *(const uint64_t *)((ch ? pv.cur[1] : pv.cur[0]) + 4 * (k - HP));
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = lane + 64 * it, ch = i >= PER ? 1 : 0, k = i - ch * PER;
            if (i < PER * nch) *(uint64_t *)&w.u.fbk.pcm[ch][4 * k] = v[it];
        }
    }
    TL_LANES_END
}
// X-PAD bytes of a slot -> LDS; returns the length the frame carries.  The contract is 0 or 2..pad_len (toolame.c:515-516,
// odr-audioenc.cpp:803,830-834); anything else -- more than the stream's toolame_set_pad() length, more than the record
// holds -- is treated as "no PAD this frame" (tl_build_config has made sure that pad_len itself fits into the frame).
TL_FN int tl_stage_xpad(TlMainLds &w, const TlLaunch &A, const TlConfig *C, size_t slot)
{
    if (!A.xpad_len) return 0;
    int xl = A.xpad_len[slot];
    if (xl < 2 || xl > TL_MAX_XPAD || xl > C->dab_length) xl = 0;
    TL_LANES_BEGIN
    for (int i = lane; i < xl; i += 64) w.xpad[i] = A.xpad[slot * TL_MAX_XPAD + i];
    TL_LANES_END
    return xl;
}

// ------------------------------------------------------------------------------------------
// Encode kernel: one unit = frame f of stream s.  Like the psy kernel's units these are
// independent of each other: the filterbank's history is PCM (the previous input frame, or the stream state before frame 0),
// the SMR comes from the psy kernel's record, and the one thing a frame owes its predecessor -- its ScF-CRC, which travels in
// the frame before (toolame.c:527-542) -- is filed aside and put in place by tl_finish_stream.
template <int PSY>     // TL_PSY_EXT: SMR from the psy kernel's record (models 1 and 3); 2: the psy-2 kernel's SMR (models 2 and 4); 0: model 0, which needs nothing but this frame's scalefactors
TL_FN void tl_main_unit(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, const TlLaunch &A, int s, int f)
{
    const TlConfig *C = &A.configs[A.stream_cfg[s]];
    TlStreamState *st = &A.state[s];
    const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
    const TlPcmView pv = tl_pcm_view(A, st, s, f);
#ifdef TL_NO_MAIN_STAMPS
    long long *sp = nullptr;
#else
    long long *sp = A.stamps ? A.stamps + slot * 32 : nullptr;
#endif
    TL_STAMP(sp, 31);
    tl_stage_pcm(w, pv, C->nch);
    const int xl = tl_stage_xpad(w, A, C, slot);
    TlFrameOut fo;
    fo.bytes = f + 1 < A.nframes ? A.out + (slot + (size_t)A.nstreams) * (size_t)A.out_stride : nullptr;     // waits in the next slot
    fo.words = f + 1 < A.nframes ? nullptr : A.newpend + (size_t)s * TL_MAX_FRAME_WORDS;
    fo.scfcrc = A.scfcrc + slot * 4;
    const int padding = A.padbits ? (int)A.padbits[slot] : 0;
    tl_encode_frame<PSY>(w, A.tables, B, C, PSY == 2 ? &A.psy_out[slot] : nullptr, pv, xl, fo, enw_s, K, padding,
                                A.taps ? &A.taps[slot] : nullptr, sp);
}

// The same unit for a PAIR of mono streams s0, s1 of one configuration (TlLaunch::partner): frame f of both by one wave (tl_encode_pair)
template <int PSY>
TL_FN void tl_main_pair(TlMainLds &w, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s, const TlPackTables *TL_RESTRICT K, const TlLaunch &A, int s0, int s1, int f)
{
    const TlConfig *C = &A.configs[A.stream_cfg[s0]];
    const int ss[2] = {s0, s1};
    tl_stage_pcm(w, tl_pcm_view_pair(A, s0, s1, f), 2);
    TlFrameOut fo[2];
    const TlPsyOut *po[2];
    const uint8_t *xsrc[2];
    int xl[2], padding[2];
#ifndef TL_EMULATE
#pragma unroll
#endif
    for (int u = 0; u < 2; u++) {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)ss[u];
        fo[u].bytes = f + 1 < A.nframes ? A.out + (slot + (size_t)A.nstreams) * (size_t)A.out_stride : nullptr;
        fo[u].words = f + 1 < A.nframes ? nullptr : A.newpend + (size_t)ss[u] * TL_MAX_FRAME_WORDS;
        fo[u].scfcrc = A.scfcrc + slot * 4;
        po[u] = PSY == 2 ? &A.psy_out[slot] : nullptr;
        padding[u] = A.padbits ? (int)A.padbits[slot] : 0;
        int x = A.xpad_len ? A.xpad_len[slot] : 0;                  // the contract of tl_stage_xpad
        if (x < 2 || x > TL_MAX_XPAD || x > C->dab_length) x = 0;
        xl[u] = x; xsrc[u] = A.xpad ? A.xpad + slot * TL_MAX_XPAD : nullptr;
    }
    tl_encode_pair<PSY>(w, B, C, po, xl, xsrc, fo, enw_s, K, padding);
}

// Models 1 and 3: one unit = frame f of stream s, psy model first, then the encoder, by the same wave.  The two phases
// share the wave's LDS block (a union: the model's arrays are dead when the encoder starts) and nothing else but the
// model's record, 4 values per subband, which waits in registers until the model is done.
union TlFrameLds { TlPsyLds p; TlMainLds m; };
template <int PSY>
TL_FN void tl_frame_unit(TlFrameLds &w, const double *TL_RESTRICT db, const TlBlockShared *TL_RESTRICT B, const double *TL_RESTRICT enw_s,
                         const TlPackTables *TL_RESTRICT K, const TlLaunch &Apsy, TL_KARG Amain_p, int s, int f, int s2 = -1)
{   // s2 >= 0: frame f of the two mono streams s and s2 (one configuration) as the two "channels" of the wave
    PA(double, rec, 4);
    tl_psy_unit<PSY>(w.p, db, Apsy, s, f, rec, s2);
    TL_SYNC();
    // The encoder phase reads the launch record afresh (device: scalar loads from the kernel-argument segment, issued HERE) and
    // re-derives its pointers from laundered copies of s / f / s2: nothing of the model phase's scalar state stays live across the
    // phases, and nothing of the encoder's is loaded before the model has run.
    TL_LAUNDER(Amain_p); TL_LAUNDER(s); TL_LAUNDER(f); TL_LAUNDER(s2);
    const TlLaunch Amain = *Amain_p;
    // the model's arrays are dead: its record goes where the encoder expects it (its own SMR array and the one beside it)
    TL_LANES_BEGIN
    if (lane < 32) {
        w.m.smr[0][lane] = L(rec)[0]; w.m.smr[1][lane] = L(rec)[1];
        w.m.psy_m[0][lane] = L(rec)[2]; w.m.psy_m[1][lane] = L(rec)[3];
    }
    TL_LANES_END
    if (s2 >= 0) tl_main_pair<TL_PSY_EXT>(w.m, B, enw_s, K, Amain, s, s2, f);
    else tl_main_unit<TL_PSY_EXT>(w.m, B, enw_s, K, Amain, s, f);
}

// Which stream shares a wave with stream s?  TlLaunch::partner[s]: the other mono stream of s's configuration it is paired with, or -1.
// The lower-numbered stream of a pair runs the unit for both (returns true, s2 = the partner), the higher one has nothing to do
// (returns false).  Launches with stage taps or cycle stamps (diagnostics, per frame of one stream) run every stream alone.
TL_FN bool tl_unit_partner(const TlLaunch &A, int s, int &s2)
{
    s2 = -1;
#ifdef TL_NO_PAIRS
    return true;                                                      // diagnostic build: every stream alone (what pairing is measured against)
#endif
    if (!A.partner || A.taps || A.stamps) return true;
    const int p = A.partner[s];
    if (p < 0) return true;
    if (p < s) return false;
    s2 = p;
    return true;
}

// After the units of a launch: for stream s, hand out the frame that was pending before the launch (slot 0), store every
// frame's ScF-CRC into the frame before it, make the launch's last frame the pending one, roll the PCM history forward.
TL_FN void tl_finish_stream(const TlLaunch &A, int s)
{
    const TlConfig *C = &A.configs[A.stream_cfg[s]];
    TlStreamState *st = &A.state[s];
    const int whole = C->frame_bytes, dab_ext = C->dab_ext, nch = C->nch;
    const bool have_prev = st->frames_done > 0;
    const int prev_len = st->pending_len;
    uint8_t *out0 = A.out + (size_t)s * (size_t)A.out_stride;
    // slot 0: the frame that was pending before the launch, with the ScF-CRC of the launch's first frame
    TL_LANES_BEGIN
    if (have_prev)
        for (int i = lane; i < ((prev_len + 3) >> 2); i += 64) {
            const uint32_t le = tl_bswap(st->pending[i]);
            const int rem = prev_len - 4 * i;
            if (rem >= 4) ((uint32_t *)out0)[i] = le;
            else for (int b = 0; b < rem; b++) out0[4 * i + b] = (uint8_t)(le >> (8 * b));
        }
    TL_LANES_END
    // slot f holds frame f-1 (slot 0: the old pending frame); frame f's ScF-CRC goes 2 + dab_ext bytes before the END of the
    // frame in slot f -- whose length (a padding slot more or less at 44.1 / 22.05 kHz) comes from the slot recurrence
    TL_LANES_BEGIN
    for (int f = lane; f < A.nframes; f += 64) {
        const size_t slot = (size_t)f * (size_t)A.nstreams + (size_t)s;
        const int len = f > 0 ? whole + (A.padbits ? (int)A.padbits[slot - (size_t)A.nstreams] : 0) : prev_len;
        if (f > 0 || have_prev) {
            uint8_t *o = A.out + slot * (size_t)A.out_stride + (len - 2 - dab_ext);
            for (int k = 0; k < dab_ext; k++) o[k] = A.scfcrc[slot * 4 + k];
        }
        if (A.out_len) A.out_len[slot] = (f > 0 || have_prev) ? len : 0;
    }
    TL_LANES_END
    const int last_len = whole + (A.padbits ? (int)A.padbits[(size_t)(A.nframes - 1) * (size_t)A.nstreams + (size_t)s] : 0);
    TL_LANES_BEGIN
    for (int i = lane; i < ((last_len + 3) >> 2); i += 64) st->pending[i] = A.newpend[(size_t)s * TL_MAX_FRAME_WORDS + i];
    TL_LANES_END
    {
        const int16_t *last = A.pcm + ((size_t)(A.nframes - 1) * (size_t)A.nstreams + (size_t)s) * 2304;
        TL_LANES_BEGIN
        for (int i = lane; i < (TL_HIST / 2) * 2; i += 64) {
            const int ch = i / (TL_HIST / 2), k = (i % (TL_HIST / 2)) * 2;
            *(uint32_t *)&st->hist[ch][k] = ch < nch ? *(const uint32_t *)(last + ch * 1152 + (1152 - TL_HIST) + k) : 0u;
        }
        if (lane == 0) { st->frames_done += A.nframes; st->pending_len = last_len; if (A.padbits) st->slot_lag = A.newlag[s]; }
        TL_LANES_END
    }
}
