// Hardware check of the wave_shr:1 DPP control used by pair_fill.hip.h (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int v = lane * 10 + 1;
    int r = __builtin_amdgcn_update_dpp(-7, v, 0x138, 0xf, 0xf, false);
    out[lane] = r;
}
int main() {
    int* d;
    int h[64];
    if (hipMalloc(&d, 256) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        int want = l == 0 ? -7 : (l - 1) * 10 + 1;
        if (h[l] != want) { ++bad; printf("lane %d got %d want %d\n", l, h[l], want); }
    }
    printf("wave_shr:1 %s\n", bad ? "WRONG" : "OK");
    return bad != 0;
}
