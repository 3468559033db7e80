// Which XCD does workgroup b run on?  (evidence for the blockIdx % 8 grouping of the unit lists; profiles/xcc_probe_r02.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(768) k(unsigned *o) {
    __shared__ double big[19000];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    big[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x == 0) o[blockIdx.x] = x + (unsigned)(big[5] * 0);
}
int main() {
    unsigned *d; hipMalloc(&d, 4 * 512);
    hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, d);
    unsigned h[512]; hipMemcpy(h, d, 4 * 256, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < 256; i++) { if (i < 24) printf("%u(%x) ", h[i] & 15, h[i]); cnt[h[i] & 15]++; }
    printf("\n");
    for (int i = 0; i < 16; i++) printf("id %d: %d blocks\n", i, cnt[i]);
    return 0;
}
