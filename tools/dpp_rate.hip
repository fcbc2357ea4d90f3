// Latency / throughput of the cross-lane moves the pair engine could use (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4000
template <int KIND, bool DEP>
__global__ __launch_bounds__(64) void k(int* out, int seed) {
    int a[4];
    for (int i = 0; i < 4; ++i) a[i] = threadIdx.x * 3 + i + seed;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int& x = DEP ? a[0] : a[u & 3];
            if (KIND == 0) x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false) + 1;        // wave_shr:1
            if (KIND == 1) x = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false) + 1;        // row_shr:1
            if (KIND == 2) x = __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false) + 1;        // wave_shl:1
            if (KIND == 3) x = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 63) & 63) * 4, x) + 1;  // ds_bpermute
            if (KIND == 4) x = __builtin_amdgcn_readlane(x, 63) + x + 1;                             // readlane
            if (KIND == 5) x = x + 1;                                                                // plain add (reference)
            if (KIND == 6) x = __builtin_amdgcn_update_dpp(x, x, 0x13C, 0xf, 0xf, false) + 1;        // wave_ror:1
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
}
typedef void (*kfn)(int*, int);
int main() {
    const char* names[] = {"wave_shr:1", "row_shr:1", "wave_shl:1", "ds_bpermute", "readlane+add", "v_add", "wave_ror:1"};
    kfn dep[] = {k<0, true>, k<1, true>, k<2, true>, k<3, true>, k<4, true>, k<5, true>, k<6, true>};
    kfn ind[] = {k<0, false>, k<1, false>, k<2, false>, k<3, false>, k<4, false>, k<5, false>, k<6, false>};
    int* d;
    (void)hipMalloc(&d, 1024 * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int kind = 0; kind < 7; ++kind)
        for (int mode = 0; mode < 2; ++mode) {
            kfn f = mode ? ind[kind] : dep[kind];
            hipLaunchKernelGGL(f, dim3(1024), dim3(64), 0, 0, d, 3);   // one wave per SIMD
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(f, dim3(1024), dim3(64), 0, 0, d, 3);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-14s %-11s %.1f cycles per (op + add) @2.4GHz, one wave per SIMD\n", names[kind], mode ? "independent" : "dependent",
                   ms * 1e-3 * 2.4e9 / (N_ITER * 16.0));
        }
    return 0;
}
