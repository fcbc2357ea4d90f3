// Issue cost of individual VALU instructions on gfx950 (cycles per wave64 instruction per SIMD at 1/2/4 waves per
// SIMD), each as an inline-asm stream over 8 registers so that the compiler cannot fold anything.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2000
#define UNR 32
#define OP2(name, txt) \
    __global__ __launch_bounds__(64) void name(int* out, int seed) { \
        int a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 3 + i + seed; \
        for (int it = 0; it < N_ITER; ++it) { _Pragma("unroll") for (int u = 0; u < UNR; ++u) { const int i = u & 7; \
            asm volatile(txt : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7])); } } \
        int r = 0; for (int i = 0; i < 8; ++i) r += a[i]; out[blockIdx.x * 64 + threadIdx.x] = r; }
OP2(k_add, "v_add_u32 %0, %1, %2")
OP2(k_sub, "v_sub_u32 %0, %1, %2")
OP2(k_subc, "v_sub_u32_e64 %0, %1, %2 clamp")
OP2(k_and, "v_and_b32 %0, %1, %2")
OP2(k_xor, "v_xor_b32 %0, %1, %2")
OP2(k_or3, "v_or3_b32 %0, %1, %2, %3")
OP2(k_lshl, "v_lshlrev_b32 %0, 3, %1")
OP2(k_lshladd, "v_lshl_add_u32 %0, %1, 2, %2")
OP2(k_add3, "v_add3_u32 %0, %1, %2, %3")
OP2(k_maxi, "v_max_i32 %0, %1, %2")
OP2(k_maxu, "v_max_u32 %0, %1, %2")
OP2(k_mini, "v_min_i32 %0, %1, %2")
OP2(k_max3i, "v_max3_i32 %0, %1, %2, %3")
OP2(k_max3u, "v_max3_u32 %0, %1, %2, %3")
OP2(k_med3, "v_med3_i32 %0, %1, %2, %3")
OP2(k_maxi16, "v_max_i16 %0, %1, %2")
OP2(k_pkmax, "v_pk_max_i16 %0, %1, %2")
OP2(k_pkadd, "v_pk_add_i16 %0, %1, %2")
OP2(k_pksubc, "v_pk_sub_u16 %0, %1, %2 clamp")
OP2(k_perm, "v_perm_b32 %0, %1, %2, %3")
OP2(k_bfe, "v_bfe_i32 %0, %1, 8, 8")
OP2(k_sdwa, "v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
OP2(k_mad24, "v_mad_u32_u24 %0, %1, %2, %3")
OP2(k_sad, "v_sad_u32 %0, %1, %2, %3")
OP2(k_alignbit, "v_alignbit_b32 %0, %1, %2, 8")
OP2(k_bfi, "v_bfi_b32 %0, %1, %2, %3")
OP2(k_mov, "v_mov_b32 %0, %2")
OP2(k_dpp, "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf")
OP2(k_cvt, "v_cvt_f32_i32 %0, %1")
OP2(k_addf, "v_add_f32 %0, %1, %2")
OP2(k_maxf, "v_max_f32 %0, %1, %2")
OP2(k_max3f, "v_max3_f32 %0, %1, %2, %3")
typedef void (*kfn)(int*, int);
int main() {
    struct { const char* n; kfn f; } t[] = {{"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_sub_u32_e64 clamp", k_subc}, {"v_and_b32", k_and},
        {"v_xor_b32", k_xor}, {"v_or3_b32", k_or3}, {"v_lshlrev_b32", k_lshl}, {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3},
        {"v_max_i32", k_maxi}, {"v_max_u32", k_maxu}, {"v_min_i32", k_mini}, {"v_max3_i32", k_max3i}, {"v_max3_u32", k_max3u},
        {"v_med3_i32", k_med3}, {"v_max_i16", k_maxi16}, {"v_pk_max_i16", k_pkmax}, {"v_pk_add_i16", k_pkadd}, {"v_pk_sub_u16 clamp", k_pksubc},
        {"v_perm_b32", k_perm}, {"v_bfe_i32", k_bfe}, {"v_add_u32_sdwa", k_sdwa}, {"v_mad_u32_u24", k_mad24}, {"v_sad_u32", k_sad},
        {"v_alignbit_b32", k_alignbit}, {"v_bfi_b32", k_bfi}, {"v_mov_b32", k_mov}, {"v_mov_b32_dpp wave_shr", k_dpp}, {"v_cvt_f32_i32", k_cvt},
        {"v_add_f32", k_addf}, {"v_max_f32", k_maxf}, {"v_max3_f32", k_max3f}};
    int* d; hipMalloc(&d, 256 * 64 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s CUs %d; cycles per wave64 instruction per SIMD at 1 / 2 / 4 waves per SIMD (2.4 GHz nominal)\n", prop.gcnArchName, prop.multiProcessorCount);
    for (auto& e : t) {
        printf("%-26s", e.n);
        for (int wps : {1, 2, 4}) {
            const int blocks = prop.multiProcessorCount * 4 * wps;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, d, 3); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, d, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %5.2f", ms * 1e6 / (5.0 * N_ITER * UNR * wps) * 2.4);
        }
        printf("\n");
    }
    return 0;
}
