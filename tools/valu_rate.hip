// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction per SIMD for the
// instruction kinds the DP kernels use, at 1/2/4 waves per SIMD.  Used to fix the VALU roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N_ITER 2000
#define UNR 32
template <int KIND>
__global__ __launch_bounds__(64) void k(int* out, int seed) {
    int a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i + seed; f[i] = (float)(a[i]); }
    int g = seed | 1;
    float gf = (float)g;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = u & 7;
            if (KIND == 0) a[i] = a[i] + g;                                  // v_add_u32
            if (KIND == 1) a[i] = max(a[i], g + u);                          // v_max_i32
            if (KIND == 2) a[i] = max(max(a[i], a[(i + 1) & 7]), g);         // v_max3_i32
            if (KIND == 3) f[i] = f[i] + gf;                                 // v_add_f32
            if (KIND == 4) f[i] = fmaxf(fmaxf(f[i], f[(i + 1) & 7]), gf);    // v_max3_f32
            if (KIND == 5) f[i] = fmaf(f[i], gf, f[(i + 1) & 7]);            // v_fma_f32
            if (KIND == 6) a[i] = __builtin_amdgcn_perm(a[i], g, a[(i + 1) & 7]);   // v_perm_b32
            if (KIND == 7) a[i] = a[i] + (int)(int8_t)(a[(i + 1) & 7] >> 8); // v_add_u32_sdwa
            if (KIND == 8) {                                                 // v_pk_add_i16
                typedef short s2 __attribute__((ext_vector_type(2)));
                s2 x = __builtin_bit_cast(s2, a[i]), y = __builtin_bit_cast(s2, g);
                a[i] = __builtin_bit_cast(int, (s2)(x + y));
            }
            if (KIND == 9) {                                                 // v_pk_max_i16
                typedef short s2 __attribute__((ext_vector_type(2)));
                s2 x = __builtin_bit_cast(s2, a[i]), y = __builtin_bit_cast(s2, a[(i + 1) & 7]);
                a[i] = __builtin_bit_cast(int, __builtin_elementwise_max(x, y));
            }
            if (KIND == 10) a[i] = a[i] ^ a[(i + 1) & 7];                     // v_xor_b32
            if (KIND == 11) a[i] = (a[(i + 1) & 7] == a[(i + 3) & 7]) ? a[i] : a[(i + 2) & 7];   // v_cmp_eq_u32 + v_cndmask_b32
            if (KIND == 13) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));          // VOP2
            if (KIND == 14) asm volatile("v_max_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));          // VOP2
            if (KIND == 15) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));          // VOP2
            if (KIND == 16) asm volatile("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7])); // VOP3, 2 sources
            if (KIND == 17) asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if (KIND == 18) asm volatile("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                                         : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
            if (KIND == 19) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));   // VOP2 (reads vcc)
            if (KIND == 12) {                                                // v_cmp_eq_u32 alone (result kept in SGPRs)
                unsigned long long m;
                asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                asm volatile("" :: "s"(m));
            }
        }
    }
    int r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (int)f[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
typedef void (*kfn)(int*, int);
int main() {
    const char* names[] = {"v_add_u32", "v_max_i32", "v_max3_i32", "v_add_f32", "v_max3_f32", "v_fma_f32", "v_perm_b32",
                           "v_add_u32_sdwa", "v_pk_add_i16", "v_pk_max_i16", "v_xor_b32", "v_cmp+v_cndmask (2 instr)", "v_cmp_eq_u32 -> SGPR",
                           "asm v_add_u32 (VOP2)", "asm v_max_i32 (VOP2)", "asm v_and_b32 (VOP2)", "asm v_sub_u32_e64 clamp", "asm v_max3_i32 (VOP3)",
                           "asm v_add_u32_sdwa", "asm v_cndmask_b32 (VOP2)"};
    kfn fns[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>};
    int* d;
    hipMalloc(&d, 256 * 64 * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    for (int kind = 0; kind < 20; ++kind)
        for (int wps : {1, 2, 4}) {
            const int blocks = prop.multiProcessorCount * 4 * wps;
            hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(64), 0, 0, d, 3);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(64), 0, 0, d, 3);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = 5.0 * N_ITER * UNR * wps;
            const double ns_per_instr = ms * 1e6 / instr_per_simd;
            printf("%-26s waves/SIMD=%d  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", names[kind], wps,
                   ns_per_instr, ns_per_instr * 2.4);
        }
    return 0;
}
