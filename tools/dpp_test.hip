// Hardware check of the wave_shr:1 DPP control used by pair_fill.hip.h (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int v = lane * 10 + 1;
    int r = __builtin_amdgcn_update_dpp(-7, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
    out[lane] = r;
    out[64 + lane] = __builtin_amdgcn_update_dpp(-9, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
}
int main() {
    int* d;
    int h[128];
    if (hipMalloc(&d, 512) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        int want = l == 0 ? -7 : (l - 1) * 10 + 1;
        if (h[l] != want) { ++bad; printf("lane %d got %d want %d\n", l, h[l], want); }
    }
    printf("wave_shr:1 %s\n", bad ? "WRONG" : "OK");
    int bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        int want = l == 63 ? -9 : (l + 1) * 10 + 1;
        if (h[64 + l] != want) { ++bad2; printf("shl lane %d got %d want %d\n", l, h[64 + l], want); }
    }
    printf("wave_shl:1 %s\n", bad2 ? "WRONG" : "OK");
    bad += bad2;
    return bad != 0;
}
