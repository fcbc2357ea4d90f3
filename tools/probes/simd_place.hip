// simd_place.hip -- where do the single-wave workgroups of a launch land?  Every wave records its XCC / SE / CU / SIMD (HW_ID) and runs a
// fixed VALU loop; the host prints, per grid size, how many SIMDs were used, the largest number of waves on one SIMD, and the kernel time.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/simd_place tools/probes/simd_place.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int VG>
__global__ __launch_bounds__(64) void probe(uint32_t* out, int iters) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    int v[VG];
#pragma unroll
    for (int k = 0; k < VG; ++k) v[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < VG; ++k) v[k] = max(v[k] + it, v[(k + 1) % VG]);   // add + max per element: half-rate max
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < VG; ++k) s += v[k];
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = hw;
        out[3 * blockIdx.x + 1] = xcc;
        out[3 * blockIdx.x + 2] = (uint32_t)s;
    }
}

template <int VG>
void run(int grid, int iters, uint32_t* d, std::vector<uint32_t>& h) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<VG>, dim3(grid), dim3(64), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<VG>, dim3(grid), dim3(64), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), d, (size_t)grid * 12, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_simd, per_cu;
    for (int g = 0; g < grid; ++g) {
        const uint32_t hw = h[3 * g], xcc = h[3 * g + 1] & 0xf;
        const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cu_key]++;
        per_simd[(cu_key << 2) | simd]++;
    }
    int mx = 0, mxcu = 0;
    for (auto& kv : per_simd) mx = std::max(mx, kv.second);
    for (auto& kv : per_cu) mxcu = std::max(mxcu, kv.second);
    std::map<int, int> hist;
    for (auto& kv : per_simd) hist[kv.second]++;
    std::printf("VG %3d grid %5d: %4zu CUs, %4zu SIMDs used, max %d waves on a SIMD, max %d on a CU, %.3f ms; SIMDs by wave count:", VG, grid, per_cu.size(),
                per_simd.size(), mx, mxcu, ms);
    for (auto& kv : hist) std::printf(" %dx:%d", kv.first, kv.second);
    std::printf("\n");
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
    uint32_t* d;
    hipMalloc(&d, 65536 * 12);
    std::vector<uint32_t> h(65536 * 3);
    for (int grid : {256, 512, 768, 1024, 1536, 2048, 4096}) run<16>(grid, iters, d, h);
    for (int grid : {256, 512, 1024, 2048}) run<80>(grid, iters / 5, d, h);
    return 0;
}
