// mini_mix.hip -- which part of the mini-stripe step does not scale from 2 to 4 waves per CU?  The step of mini_fill.hip.h (RL = 10, global,
// gap-shifted, no stores) in a loop, with pieces switched off by FLAGS, at 512 / 768 / 1024 / 2048 single-wave workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bioinformatics-algorithms_amd/csrc -o tools/probes/mini_mix tools/probes/mini_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mini_fill.hip.h"
using namespace pwa;

enum { NO_DPP = 1, NO_TABLE = 2, NO_CODES = 4, NO_NORM = 8, NO_SDWA_ADD = 16 };

template <int FLAGS>
__global__ __launch_bounds__(64) void mix(int* out, int n_chunks, uint32_t seed) {
    constexpr int RL = 10, NQ = 3;
    const int lane = threadIdx.x, k = lane & 15;
    uint32_t pk[NQ] = {seed * (lane + 1), seed * (lane + 7), seed ^ 0x07070000u};
    int hl[RL];
#pragma unroll
    for (int r = 0; r < RL; ++r) hl[r] = 1;
    int diag0 = 1, bottom = 0, tch = 0;
    const uint32_t tab_lo = 0x0606060eu, tab_hi = 0x06060606u;
    uint32_t acc = 0;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int tcv = (int)((seed + ch * 0x01010101u) & 0x03030303u);
        static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            int tn, up_in;
            if (FLAGS & NO_DPP) {
                tn = tch ^ tcv;
                up_in = bottom;
            } else {
                tn = mini_row_shr1(mini_pick_lane0<q>(tch, tcv), tch);
                up_in = mini_row_shr1(0, bottom);
            }
            uint32_t s4[NQ];
#pragma unroll
            for (int x = 0; x < NQ; ++x) s4[x] = (FLAGS & NO_TABLE) ? (pk[x] ^ (uint32_t)tn) : __builtin_amdgcn_perm(tab_hi, tab_lo, pk[x] ^ (uint32_t)tn);
            int dg = diag0, up = up_in;
            uint32_t codes[NQ] = {0, 0, 0};
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const int kd = (FLAGS & NO_SDWA_ADD) ? p_addw(dg, (int)s4[r / 4]) : p_addw(dg, (int)(int8_t)(s4[r / 4] >> (8 * (r % 4))));
                const int kl = hl[r];
                const int kk = max(max(kd, up), kl);
                const int base = (FLAGS & NO_NORM) ? kk : (kk & ~3);
                if (!(FLAGS & NO_CODES)) {
                    if (r % 4 == 0) codes[r / 4] = tb_first_code(kk);
                    if (r % 4 == 1) tb_put_code<1>(codes[r / 4], kk);
                    if (r % 4 == 2) tb_put_code<2>(codes[r / 4], kk);
                    if (r % 4 == 3) tb_put_code<3>(codes[r / 4], kk);
                } else {
                    codes[r / 4] ^= (uint32_t)kk;
                }
                dg = kl;
                up = base;
                hl[r] = (FLAGS & NO_NORM) ? (base ^ 1) : (base | 1);
            }
            diag0 = p_addw(up_in, 1);
            bottom = up;
            tch = tn;
            acc ^= codes[0] + codes[1] + codes[2];
        });
    }
    if (acc == 0x12345678u && k == 99) out[0] = (int)acc;   // keep the work alive
}

template <int FLAGS>
void run(const char* what, int* d) {
    const int n_chunks = 626;   // = 10 016 steps, the `g` shape
    std::printf("%-44s", what);
    for (int grid : {256, 512, 768, 1024, 2048}) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        hipLaunchKernelGGL(mix<FLAGS>, dim3(grid), dim3(64), 0, 0, d, n_chunks, 0x9e3779b9u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(mix<FLAGS>, dim3(grid), dim3(64), 0, 0, d, n_chunks, 0x9e3779b9u);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        std::printf("  %4d wg: %6.3f ms", grid, ms);
    }
    std::printf("\n");
}

int main() {
    int* d;
    (void)hipMalloc(&d, 64);
    run<0>("full step (no stores, no text loads)", d);
    run<NO_DPP>("without the DPP moves", d);
    run<NO_TABLE>("without v_perm table lookups", d);
    run<NO_CODES>("without SDWA code packing", d);
    run<NO_NORM>("without and/or normalisation", d);
    run<NO_SDWA_ADD>("plain add instead of SDWA add", d);
    run<NO_DPP | NO_TABLE | NO_CODES | NO_SDWA_ADD>("only max3 + and + or + add", d);
    return 0;
}
