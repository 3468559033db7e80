// tl_kernel_util.h -- what the kernels' translation units (toolame_hip.hip, toolame_psy2.hip) share beside mp2_wave.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TL_LDS_GRANULE 1280u          // LDS is handed out in granules of 1280 bytes on gfx950 (160 KB / 128)
#ifndef TL_PSY2_WAVES
#define TL_PSY2_WAVES 12
#endif

// Next unit of a persistent kernel's work list: ONE returning device-scope atomic add per wave, issued by lane 0 alone.
// The lane mask is narrowed inside the asm statement, not with an `if (lane == 0)`: LLVM threaded such a branch together
// with the equal test of the diagnostic stamps at the end of the previous unit into a loop that some lanes never left; and
// its wave-level atomic optimiser, which folds `atomicAdd(p, 1)` of 64 lanes into one add, only does so while it can prove the
// address uniform -- when it cannot, the 64 adds of two waves interleave and units are handed out twice or never.
// `counter` must be wave-uniform and the call site wave-uniform control flow (lane 0 active: its registers carry the operands).
static __device__ __forceinline__ int tl_next_unit(int32_t *counter)
{
    int u;
    uint64_t saved;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(u), "=&s"(saved) : "v"(0), "v"(1), "s"(counter) : "memory");
    return __builtin_amdgcn_readfirstlane(u);
}
