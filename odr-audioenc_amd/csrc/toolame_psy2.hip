// toolame_psy2.hip -- the psy kernel of models 2 and 4 (psycho_2.c:52-254, psycho_4.c:124-325 through tl_psy2_chain, mp2_wave.h).
// A translation unit of its own because of its build flags: this kernel is faster WITH the IR load / store vectorizer (its
// subband stage reads 17 consecutive doubles per lane at a stride of 16 between lanes: as ds_read_b128 the conflicts are taken
// per 16-lane group, as ds_read_b64 per 32-lane half), every other kernel without it (csrc/Makefile).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mp2_host.h"
#include "mp2_wave.h"
#include "tl_kernel_util.h"

// psy kernel of models 2 and 4: a unit = a run of frames of one channel of one stream (tl_psy2_unit: whole chains first, then
// the chains of the last round of waves cut into runs, so that one stream with many frames fills the chip as well as many
// streams do); the run's r/phi prediction state stays in the wave's registers (tl_psy2_chain).  No table in LDS but glibc's
// sincos table (the model's own tables are read through the caches), 12.1 KB per wave: twelve waves per CU like the other kernels.
static_assert((TL_PSY2_WAVES * sizeof(TlPsy2Lds) + 440 * 8 + TL_LDS_GRANULE - 1) / TL_LDS_GRANULE <= 128, "twelve psy-2 waves per CU");
__global__ void __launch_bounds__(64 * TL_PSY2_WAVES) __attribute__((amdgpu_waves_per_eu(TL_PSY2_WAVES / 4, TL_PSY2_WAVES / 4))) tl_psy2_kernel(TlLaunch A)
{
    __shared__ TlPsy2Lds lds[TL_PSY2_WAVES];
    __shared__ __attribute__((aligned(16))) uint64_t sct[440];      // glibc's sincos table: two 16-byte gathers per sincos stay on the CU
    for (int i = (int)threadIdx.x; i < 440; i += 64 * TL_PSY2_WAVES) sct[2 * (i >> 2) + (i & 1) + 220 * ((i >> 1) & 1)] = tlm_sincostab[i];      // (TL_SCT, mp2_psy24.h)
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int wave_v = (int)(threadIdx.x >> 6);
    asm volatile("" : "+v"(wave_v));
    TlPsy2Lds &wl = lds[wave_v];
    const int nunits = A.p2_nwhole + (A.nchain - A.p2_nwhole) * A.p2_k, nwaves = (int)gridDim.x * TL_PSY2_WAVES;
    for (int u = (int)blockIdx.x * TL_PSY2_WAVES + wave; u < nunits; u = nwaves + tl_next_unit(&A.work[0])) {
        int c, f0, f1;
        if (!tl_psy2_unit(A, u, c, f0, f1)) continue;
        const int e = __builtin_amdgcn_readfirstlane(A.chain_list[c]);
        tl_psy2_chain(wl, A, e & 0x3fffffff, e >> 30, f0, f1, sct);
    }
}

