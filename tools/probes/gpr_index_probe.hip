#include <hip/hip_runtime.h>
// probe: VGPR index mode (SRC0_REL) on gfx950 -- out[lane] = d + prof[idx] with prof in four consecutive physical VGPRs
__global__ void k(int* out, const int* in, int idx) {
    const int lane = threadIdx.x;
    int p0 = in[lane], p1 = in[64 + lane], p2 = in[128 + lane], p3 = in[192 + lane];
    int d = in[256 + lane];
    int t;
    asm volatile("s_set_gpr_idx_on %2, 1\n\tv_add_u32 %0, v100, %1\n\ts_set_gpr_idx_off"
                 : "=v"(t) : "v"(d), "s"(idx), "{v100}"(p0), "{v101}"(p1), "{v102}"(p2), "{v103}"(p3) : "m0");
    out[lane] = t;
}
int main() {
    int *din, *dout, h[320], o[64];
    for (int i = 0; i < 320; ++i) h[i] = (i / 64) * 1000 + i % 64;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    int bad = 0;
    for (int idx = 0; idx < 4; ++idx) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, din, idx);
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) if (o[l] != (4000 + l) + (idx * 1000 + l)) ++bad;
        printf("idx %d: out[5] = %d (expect %d)\n", idx, o[5], 4005 + idx * 1000 + 5);
    }
    printf(bad ? "INDEX MODE BROKEN (%d wrong)\n" : "index mode works (%d wrong)\n", bad);
    return bad != 0;
}
