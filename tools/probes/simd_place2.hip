// simd_place2.hip -- where do the waves of a launch land when ANOTHER kernel ran just before it?  (r03: the mini-stripe fill took 2.4 ms
// instead of 1.35 ms whenever the walk kernel, or any kernel with a large grid, preceded it.)  A pure-VALU probe kernel records XCC / SE /
// CU / SIMD of every wave, launched (a) as 1024 single-wave workgroups, (b) as 256 workgroups of 4 waves that each ask for 96 KiB of LDS (one
// per CU), right after a streaming read kernel of 4096 x 256 threads.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/simd_place2 tools/probes/simd_place2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void read_all(const uint4* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i].x ^ p[i].w;
    if (acc == 0x12345u) out[0] = acc;
}

__global__ void probe(uint32_t* out, int iters) {
    extern __shared__ int pad[];
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    int v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = max(v[k] + it, v[(k + 1) % 16]);
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[3 * w] = hw;
        out[3 * w + 1] = xcc;
        out[3 * w + 2] = (uint32_t)s;
    }
    if (s == 0x7fffffff) pad[threadIdx.x] = s;
}

int main() {
    uint32_t* d;
    uint8_t* big;
    (void)hipMalloc(&d, 65536 * 12);
    const size_t big_bytes = 6ull << 30;
    if (hipMalloc(&big, big_bytes) != hipSuccess) return 1;
    (void)hipMemset(big, 1, big_bytes);
    std::vector<uint32_t> h(65536 * 3);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int form = 0; form < 4; ++form) {
        const bool wg4 = form & 1, after_read = form & 2;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipDeviceSynchronize();
            if (after_read) hipLaunchKernelGGL(read_all, dim3(4096), dim3(256), 0, 0, (const uint4*)big, big_bytes / 16, d + 60000);
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a);
            if (wg4) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 96 * 1024, 0, d, 20000);
            else hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, d, 20000);
            (void)hipEventRecord(b);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            (void)hipMemcpy(h.data(), d, 1024 * 12, hipMemcpyDeviceToHost);
            std::map<uint32_t, int> per_simd, per_cu, per_xcc;
            for (int g = 0; g < 1024; ++g) {
                const uint32_t hw = h[3 * g], xcc = h[3 * g + 1] & 0xf;
                const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const uint32_t cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
                per_cu[cu_key]++;
                per_simd[(cu_key << 2) | simd]++;
                per_xcc[xcc]++;
            }
            std::map<int, int> hist, hist_cu;
            for (auto& kv : per_simd) hist[kv.second]++;
            for (auto& kv : per_cu) hist_cu[kv.second]++;
            std::printf("%-28s %-18s %.3f ms; %zu CUs, %zu SIMDs; SIMDs by waves:", wg4 ? "256 WGs x 4 waves + 96K LDS" : "1024 single-wave WGs", after_read ? "after a read kernel" : "after idle", ms,
                        per_cu.size(), per_simd.size());
            for (auto& kv : hist) std::printf(" %dx:%d", kv.first, kv.second);
            std::printf("; CUs by waves:");
            for (auto& kv : hist_cu) std::printf(" %dx:%d", kv.first, kv.second);
            std::printf("; per XCC:");
            for (auto& kv : per_xcc) std::printf(" %d", kv.second);
            std::printf("\n");
        }
    }
    return 0;
}
