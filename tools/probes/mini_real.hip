// mini_real.hip -- the product's mini_fill_kernel<10, global, gap-shifted> launched directly on synthetic descriptors: fill time against
// the number of pairs, for random sequences and for constant ones (is the one-wave-per-SIMD slowdown a matter of the DATA?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bioinformatics-algorithms_amd/csrc -o tools/probes/mini_real tools/probes/mini_real.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <unistd.h>
#include <random>
#include <vector>
#include "mini_fill.hip.h"
using namespace pwa;

__global__ void read_all(const uint4* p, size_t n, uint32_t* out) {   // streams through n uint4 with plain loads
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i].x ^ p[i].w;
    if (acc == 0x12345u) out[0] = acc;
}

__global__ void alu_clock(unsigned long long* out, int iters) {   // pure VALU; lane 0 of block 0 reports shader cycles and 100 MHz ticks
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = max(v[k] + it, v[(k + 1) % 8]);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
        out[2] = (unsigned long long)s;
    }
}

static void launch_fill(const PairParams& G) {   // as pwalign.hip launches it: 4-wave workgroups, LDS request = one share of a CU
    const uint32_t n_wg = (G.n_tasks + 3) / 4, per_cu = std::min<uint32_t>(5, (n_wg + 255) / 256);
    static const uint32_t pad_kib[6] = {0, 96, 64, 48, 36, 30};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mini_fill_kernel<10, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pad_kib[per_cu] * 1024));
    hipLaunchKernelGGL((mini_fill_kernel<10, false, false, true>), dim3(std::min<uint32_t>(n_wg, 256 * per_cu)), dim3(256), pad_kib[per_cu] * 1024, 0, G);
}

int main(int argc, char** argv) {
    const int n = 150, m = 10000, max_pairs = 8192;
    const size_t band_one = mini_band_steps(m) * 160;
    uint8_t *d_arena, *d_band;
    PairDesc* d_desc;
    PairResult* d_res;
    StripeBest* d_best;
    uint32_t* d_queue;
    const size_t pat_stride = 160, txt_stride = 10240;
    const size_t arena_bytes = max_pairs * pat_stride + 256 * txt_stride + 4096;
    if (hipMalloc(&d_arena, arena_bytes) != hipSuccess || hipMalloc(&d_band, (size_t)max_pairs * band_one + 65536) != hipSuccess) return 1;
    (void)hipMalloc(&d_desc, max_pairs * sizeof(PairDesc));
    (void)hipMalloc(&d_res, max_pairs * sizeof(PairResult));
    (void)hipMalloc(&d_best, max_pairs * sizeof(StripeBest));
    (void)hipMalloc(&d_queue, 64);
    std::vector<PairDesc> pd(max_pairs);
    for (int p = 0; p < max_pairs; ++p) {
        std::memset(&pd[p], 0, sizeof(PairDesc));
        pd[p].pat = d_arena + (size_t)p * pat_stride;
        pd[p].txt = d_arena + (size_t)max_pairs * pat_stride + (size_t)(p % 256) * txt_stride;
        pd[p].n = n;
        pd[p].m = m;
        pd[p].tb = d_band + (size_t)p * band_one;
        pd[p].res = d_res + p;
        pd[p].first_stripe = p;
        pd[p].n_stripes = 1;
    }
    (void)hipMemcpy(d_desc, pd.data(), pd.size() * sizeof(PairDesc), hipMemcpyHostToDevice);
    std::vector<uint8_t> host(arena_bytes);
    for (const char* data : {"random", "constant", "period-2"}) {
        std::mt19937 rng(7);
        for (size_t o = 0; o < arena_bytes; ++o) host[o] = data[0] == 'r' ? (uint8_t)(rng() & 3) : data[0] == 'c' ? 0 : (uint8_t)(o & 1);
        (void)hipMemcpy(d_arena, host.data(), arena_bytes, hipMemcpyHostToDevice);
        std::printf("%-9s sequences:", data);
        for (int pairs : {1024, 2048, 3072, 4096, 8192}) {
            PairParams G{};
            G.pairs = d_desc;
            G.n_pairs = pairs;
            G.n_tasks = pairs / 4;
            G.queue = d_queue;
            G.best = d_best;
            G.match = 3;   // gap-shifted 1 / -1 / -1: s - 2 gap
            G.mismatch = 1;
            G.gap = 0;
            G.dash = 0x100;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                (void)hipMemsetAsync(d_queue, 0, 16, 0);
                (void)hipEventRecord(a);
                launch_fill(G);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            std::printf("  %5d pairs %6.3f ms", pairs, best);
        }
        std::printf("\n");
    }
    // the same launch after the GPU has been idle for a while (what a library call sees: host work and copies between kernels)
    for (int idle_ms : {0, 1, 3, 10}) {
        PairParams G{};
        G.pairs = d_desc;
        G.n_pairs = 4096;
        G.n_tasks = 1024;
        G.queue = d_queue;
        G.best = d_best;
        G.match = 3;
        G.mismatch = 1;
        G.gap = 0;
        G.dash = 0x100;
        std::printf("4096 pairs, %2d ms of idle before each launch:", idle_ms);
        for (int rep = 0; rep < 5; ++rep) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipDeviceSynchronize();
            if (idle_ms) usleep(idle_ms * 1000);
            (void)hipMemsetAsync(d_queue, 0, 16, 0);
            (void)hipEventRecord(a);
            launch_fill(G);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            std::printf(" %6.3f", ms);
        }
        std::printf(" ms\n");
    }
    // fill, then the product's op-list walk over the bands, then the fill again: what does the walk leave behind?
    {
        uint8_t* d_ops;
        (void)hipMalloc(&d_ops, (size_t)4096 * 10160 + 4096);
        for (int p = 0; p < 4096; ++p) {
            pd[p].ops = d_ops + (size_t)p * 10160;
            pd[p].ops_cap = 10150;
            pd[p].score_bias = -10150;
        }
        (void)hipMemcpy(d_desc, pd.data(), pd.size() * sizeof(PairDesc), hipMemcpyHostToDevice);
        PairParams G{};
        G.pairs = d_desc;
        G.n_pairs = 4096;
        G.n_tasks = 1024;
        G.queue = d_queue;
        G.best = d_best;
        G.match = 3;
        G.mismatch = 1;
        G.gap = 0;
        G.dash = 0x100;
        std::vector<uint8_t> hops((size_t)4096 * 10160);
        uint8_t* d_other;
        (void)hipMalloc(&d_other, (size_t)4096 * band_one);
        (void)hipMemset(d_other, 1, (size_t)4096 * band_one);
        {   // the shader clock right after each kind of kernel: a pure-VALU kernel on every SIMD, cycles / (10 ns ticks)
            unsigned long long* d_clk;
            (void)hipMalloc(&d_clk, 64);
            const char* pre[] = {"idle 5 ms", "the fill", "the walk", "a plain read of 6.6 GB", "a memset of 6.6 GB"};
            for (int kind = 0; kind < 5; ++kind) {
                std::printf("clock of a VALU kernel (0.5 ms, 1024 waves) after %-24s:", pre[kind]);
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipDeviceSynchronize();
                    if (kind == 0) usleep(5000);
                    if (kind == 1) {
                        (void)hipMemsetAsync(d_queue, 0, 16, 0);
                        launch_fill(G);
                    }
                    if (kind == 2) hipLaunchKernelGGL((pair_traceback_kernel<10, false, WALK_OPS, 16>), dim3(4096), dim3(64), 0, 0, G);
                    if (kind == 3) hipLaunchKernelGGL(read_all, dim3(4096), dim3(256), 0, 0, (const uint4*)d_other, (size_t)4096 * band_one / 16, d_queue + 8);
                    if (kind == 4) (void)hipMemsetAsync(d_other, 0, (size_t)4096 * band_one, 0);
                    hipEvent_t a, b;
                    (void)hipEventCreate(&a);
                    (void)hipEventCreate(&b);
                    (void)hipEventRecord(a);
                    hipLaunchKernelGGL(alu_clock, dim3(1024), dim3(64), 0, 0, d_clk, 12000);
                    (void)hipEventRecord(b);
                    (void)hipDeviceSynchronize();
                    unsigned long long h[3];
                    (void)hipMemcpy(h, d_clk, sizeof h, hipMemcpyDeviceToHost);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, a, b);
                    std::printf("  %.3f ms, %.2f GHz", ms, (double)h[0] / ((double)h[1] * 10.0));
                }
                std::printf("\n");
            }
        }
        // F W F F: is only the first fill after a walk slow?
        std::printf("fill, walk, fill, fill, fill:");
        for (int rep = 0; rep < 3; ++rep) {
            float t[4];
            for (int f = 0; f < 4; ++f) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                (void)hipMemsetAsync(d_queue, 0, 16, 0);
                (void)hipEventRecord(a);
                launch_fill(G);
                (void)hipEventRecord(b);
                if (f == 0) hipLaunchKernelGGL((pair_traceback_kernel<10, false, WALK_OPS, 16>), dim3(4096), dim3(64), 0, 0, G);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&t[f], a, b);
            }
            std::printf("  [%.3f | %.3f %.3f %.3f]", t[0], t[1], t[2], t[3]);
        }
        std::printf(" ms\n");
        for (int mode = 4; mode < 7; ++mode) {
            const char* what[] = {"", "", "", "", "fill + plain read of the whole band", "fill + plain read of ANOTHER 6.6 GB buffer", "fill + walk + plain read of another 6.6 GB"};
            std::printf("%-46s fills:", what[mode]);
            for (int rep = 0; rep < 5; ++rep) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                (void)hipMemsetAsync(d_queue, 0, 16, 0);
                (void)hipEventRecord(a);
                launch_fill(G);
                (void)hipEventRecord(b);
                if (mode == 6) hipLaunchKernelGGL((pair_traceback_kernel<10, false, WALK_OPS, 16>), dim3(4096), dim3(64), 0, 0, G);
                hipLaunchKernelGGL(read_all, dim3(4096), dim3(256), 0, 0, (const uint4*)(mode == 4 ? d_band : d_other), (size_t)4096 * band_one / 16, d_queue + 8);
                (void)hipDeviceSynchronize();
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                std::printf(" %6.3f", ms);
            }
            std::printf(" ms\n");
        }
        for (int mode = 0; mode < 4; ++mode) {
            const char* what[] = {"fill only", "fill + walk", "fill + walk + D2H of the ops (41 MB)", "fill + memset of the whole band (6.6 GB)"};
            std::printf("%-42s fills:", what[mode]);
            for (int rep = 0; rep < 5; ++rep) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                (void)hipMemsetAsync(d_queue, 0, 16, 0);
                (void)hipEventRecord(a);
                launch_fill(G);
                (void)hipEventRecord(b);
                if (mode == 1 || mode == 2) hipLaunchKernelGGL((pair_traceback_kernel<10, false, WALK_OPS, 16>), dim3(4096), dim3(64), 0, 0, G);
                if (mode == 2) (void)hipMemcpy(hops.data(), d_ops, hops.size(), hipMemcpyDeviceToHost);
                if (mode == 3) (void)hipMemsetAsync(d_band, 0, (size_t)4096 * band_one, 0);
                (void)hipDeviceSynchronize();
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                std::printf(" %6.3f", ms);
            }
            std::printf(" ms\n");
        }
    }
    // ... and with a host-to-device copy of 40 MB right before it (the arena / descriptor uploads of a call)
    {
        std::vector<uint8_t> big(40u << 20, 1);
        PairParams G{};
        G.pairs = d_desc;
        G.n_pairs = 4096;
        G.n_tasks = 1024;
        G.queue = d_queue;
        G.best = d_best;
        G.match = 3;
        G.mismatch = 1;
        G.gap = 0;
        G.dash = 0x100;
        std::printf("4096 pairs, a 40 MB pageable H2D copy before each launch:");
        for (int rep = 0; rep < 5; ++rep) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipMemcpy(d_band, big.data(), big.size(), hipMemcpyHostToDevice);
            (void)hipMemsetAsync(d_queue, 0, 16, 0);
            (void)hipEventRecord(a);
            launch_fill(G);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            std::printf(" %6.3f", ms);
        }
        std::printf(" ms\n");
    }
    return 0;
}
