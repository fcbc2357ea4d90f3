// valu_issue.hip -- issue cost of single VALU instructions on gfx950, measured two ways at 1 / 2 / 4 / 8 waves per SIMD:
//   (a) in-kernel: s_memtime around the stream (shader-clock ticks), lane 0 of every wave, averaged  -> cycles per
//       instruction as ONE wave sees them (at w waves per SIMD a wave gets 1/w of the SIMD: SIMD cost = that / w);
//   (b) wall: HIP events around the launch, converted at the NOMINAL 2.4 GHz -> cycles per instruction and SIMD;
//   and their quotient = the clock the chip actually held (ticks per second), so that neither figure rests on an assumed
//   frequency.
// Every stream is inline asm over 8 rotating registers (no instruction reads the result of the one before it), so the
// compiler can neither fold nor reorder anything, and every launch runs >= 5 ms so that launch overhead is < 1 %.
// Replaces the folded rows of profiles/r01_valu_rate_microbench.txt.
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue tools/valu_issue.hip && tools/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define N_ITER 60000
#define UNR 32
#define BODY(txt, T)                                                                                                   \
    T a[8];                                                                                                            \
    for (int i = 0; i < 8; ++i) a[i] = (T)(threadIdx.x * 3 + i + seed);                                                \
    unsigned long long t0, t1;                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");               \
    for (int it = 0; it < N_ITER; ++it) {                                                                              \
        _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                              \
            const int i = u & 7;                                                                                       \
            asm volatile(txt : "=v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));            \
        }                                                                                                              \
    }                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");               \
    T r = 0;                                                                                                           \
    for (int i = 0; i < 8; ++i) r += a[i];                                                                             \
    out[blockIdx.x * 64 + threadIdx.x] = (int)r;                                                                       \
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
#define OP(name, txt) \
    __global__ __launch_bounds__(64) void name(int* out, unsigned long long* ticks, int seed) { BODY(txt, int) }
#define OP64(name, txt) \
    __global__ __launch_bounds__(64) void name(int* out, unsigned long long* ticks, int seed) { BODY(txt, long long) }

OP(k_fma, "v_fma_f32 %0, %1, %2, %3")
OP64(k_pkfma, "v_pk_fma_f32 %0, %1, %2, %3")
OP64(k_pkadd32, "v_pk_add_f32 %0, %1, %2")
OP(k_mulf, "v_mul_f32 %0, %1, %2")
OP(k_addf, "v_add_f32 %0, %1, %2")
OP(k_maxf, "v_max_f32 %0, %1, %2")
OP(k_max3f, "v_max3_f32 %0, %1, %2, %3")
OP(k_max3i, "v_max3_i32 %0, %1, %2, %3")
OP(k_maxi, "v_max_i32 %0, %1, %2")
OP(k_maxi16, "v_max_i16 %0, %1, %2")
OP(k_maxu16, "v_max_u16 %0, %1, %2")
OP(k_add, "v_add_u32 %0, %1, %2")
OP(k_addc, "v_add_i32 %0, %1, %2 clamp")
OP(k_subc, "v_sub_u32_e64 %0, %1, %2 clamp")
OP(k_and, "v_and_b32 %0, %1, %2")
OP(k_or, "v_or_b32 %0, %1, %2")
OP(k_xor, "v_xor_b32 %0, %1, %2")
OP(k_mov, "v_mov_b32 %0, %2")
OP(k_sdwa, "v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
OP(k_perm, "v_perm_b32 %0, %1, %2, %3")
OP(k_dot4i8, "v_dot4_i32_i8 %0, %1, %2, %3")
OP(k_dot4u8, "v_dot4_u32_u8 %0, %1, %2, %3")
OP(k_dot8i4, "v_dot8_i32_i4 %0, %1, %2, %3")
OP(k_dot2i16, "v_dot2_i32_i16 %0, %1, %2, %3")
OP(k_sad8, "v_sad_u8 %0, %1, %2, %3")
OP(k_bfei, "v_bfe_i32 %0, %1, 8, 8")
OP(k_pkmax, "v_pk_max_i16 %0, %1, %2")
OP(k_pkadd, "v_pk_add_i16 %0, %1, %2")
OP(k_pksubc, "v_pk_sub_u16 %0, %1, %2 clamp")
OP(k_cndmask, "v_cndmask_b32_e64 %0, %1, %2, s[2:3]")   // the kernarg pointer pair as an arbitrary lane mask
OP(k_cmp, "v_cmp_eq_u32 vcc, %1, %2\n\tv_mov_b32 %0, %3")   // a compare needs a VGPR result to stay in the rotation: 2 instructions
OP(k_lshl, "v_lshlrev_b32 %0, 3, %1")
OP(k_lshladd, "v_lshl_add_u32 %0, %1, 2, %2")
OP(k_add3, "v_add3_u32 %0, %1, %2, %3")
OP(k_mad24, "v_mad_i32_i24 %0, %1, %2, %3")
OP(k_cvtub, "v_cvt_f32_ubyte1 %0, %1")
OP(k_dpp, "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf")
OP(k_adddpp, "v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf")
OP(k_max3i16, "v_max3_i16 %0, %1, %2, %3")
OP(k_max3u16, "v_max3_u16 %0, %1, %2, %3")
OP(k_min3i16, "v_min3_i16 %0, %1, %2, %3")
OP(k_med3i16, "v_med3_i16 %0, %1, %2, %3")
OP(k_addu16, "v_add_u16 %0, %1, %2")
OP(k_subu16c, "v_sub_u16_e64 %0, %1, %2 clamp")
OP(k_addi16c, "v_add_i16 %0, %1, %2 clamp")
OP(k_madu16, "v_mad_u16 %0, %1, %2, %3")
OP(k_max3f16, "v_max3_f16 %0, %1, %2, %3")
OP(k_maxf16, "v_max_f16 %0, %1, %2")
OP(k_addu16sdwa, "v_add_u16_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
OP(k_addcou32, "v_addc_co_u32 %0, vcc, %1, %2, vcc")
OP(k_subrev, "v_subrev_u32 %0, %1, %2")
OP(k_bfi2, "v_bfi_b32 %0, %1, %2, %3")
OP(k_andor, "v_and_or_b32 %0, %1, %2, %3")
// VGPR index mode (the profile form of the batched kernels): 4 indexed full-rate adds between one s_set_gpr_idx_on / _off pair,
// and the VALU of four SW cells (4 adds + 6 max3 + 4 sub-clamp) with the pair and one s_bfe -- against the table form's
// (4 sdwa adds + xor + perm + the same).  Counted per CELL.  (Scalar temporaries are compiler-allocated operands, m0 is declared.)
#define BODY_X(txt)                                                                                                    \
    int a[8];                                                                                                          \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 3 + i + seed;                                                     \
    int st = seed;                                                                                                     \
    unsigned long long t0, t1;                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");               \
    for (int it = 0; it < N_ITER; ++it) {                                                                              \
        _Pragma("unroll") for (int u = 0; u < UNR / 4; ++u) {                                                          \
            const int i = u & 7;                                                                                       \
            asm volatile(txt : "=&v"(a[i]), "+s"(st) : "v"(a[(i + 3) & 7]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]) : "m0", "scc"); \
        }                                                                                                              \
    }                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");               \
    int r = st;                                                                                                        \
    for (int i = 0; i < 8; ++i) r += a[i];                                                                             \
    out[blockIdx.x * 64 + threadIdx.x] = r;                                                                            \
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
#define OPX(name, txt) \
    __global__ __launch_bounds__(64) void name(int* out, unsigned long long* ticks, int seed) { BODY_X(txt) }
OPX(k_idx4, "s_set_gpr_idx_on 0, 1\n\tv_add_u32 %0, %2, %3\n\tv_add_u32 %0, %2, %4\n\tv_add_u32 %0, %3, %4\n\tv_add_u32 %0, %4, %3\n\ts_set_gpr_idx_off")
OPX(k_cell4, "s_bfe_u32 %1, %1, 0x80008\n\ts_set_gpr_idx_on 0, 1\n\tv_add_u32 %0, %2, %3\n\tv_add_u32 %0, %2, %4\n\tv_add_u32 %0, %3, %4\n\tv_add_u32 %0, %4, %3\n\ts_set_gpr_idx_off\n\tv_max3_i32 %0, %2, %3, %4\n\tv_sub_u32_e64 %0, %2, %3 clamp\n\tv_max3_i32 %0, %2, %3, %4\n\tv_sub_u32_e64 %0, %2, %4 clamp\n\tv_max3_i32 %0, %4, %3, %2\n\tv_sub_u32_e64 %0, %3, %2 clamp\n\tv_max3_i32 %0, %2, %4, %3\n\tv_sub_u32_e64 %0, %4, %2 clamp\n\tv_max3_i32 %0, %3, %4, %2\n\tv_max3_i32 %0, %3, %2, %4")
OPX(k_cell4old, "v_add_u32_sdwa %0, %2, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_add_u32_sdwa %0, %2, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\tv_add_u32_sdwa %0, %3, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\tv_add_u32_sdwa %0, %4, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\tv_xor_b32 %0, %2, %3\n\tv_perm_b32 %0, %2, %3, %4\n\tv_max3_i32 %0, %2, %3, %4\n\tv_sub_u32_e64 %0, %2, %3 clamp\n\tv_max3_i32 %0, %2, %3, %4\n\tv_sub_u32_e64 %0, %2, %4 clamp\n\tv_max3_i32 %0, %4, %3, %2\n\tv_sub_u32_e64 %0, %3, %2 clamp\n\tv_max3_i32 %0, %2, %4, %3\n\tv_sub_u32_e64 %0, %4, %2 clamp\n\tv_max3_i32 %0, %3, %4, %2\n\tv_max3_i32 %0, %3, %2, %4")
OP(k_nop, "s_nop 0")

// dependent chains (every instruction reads the result of the one before it) and chains diluted with independent fillers:
// what ONE wave alone on its SIMD sustains -- the single-pair fill kernel's regime (one stripe = one wave per SIMD)
#define OPDEP(name, txt) \
    __global__ __launch_bounds__(64) void name(int* out, unsigned long long* ticks, int seed) { \
        int a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 3 + i + seed; \
        unsigned long long t0, t1; \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
        for (int it = 0; it < N_ITER; ++it) { _Pragma("unroll") for (int u = 0; u < UNR / 4; ++u) { \
            asm volatile(txt : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "v"(a[6]), "v"(a[7])); } } \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
        int r = 0; for (int i = 0; i < 8; ++i) r += a[i]; out[blockIdx.x * 64 + threadIdx.x] = r; \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0; }
// %0 is the chain register; %1..%5 are filler registers; %6, %7 read-only.  Each macro body = 4 chain instructions (+ fillers).
OPDEP(d_add, "v_add_u32 %0, %0, %6\n\tv_add_u32 %0, %0, %7\n\tv_add_u32 %0, %0, %6\n\tv_add_u32 %0, %0, %7")
OPDEP(d_and, "v_and_b32 %0, %0, %6\n\tv_or_b32 %0, %0, %7\n\tv_and_b32 %0, %0, %6\n\tv_or_b32 %0, %0, %7")
OPDEP(d_max3, "v_max3_i32 %0, %0, %6, %7\n\tv_max3_i32 %0, %0, %7, %6\n\tv_max3_i32 %0, %0, %6, %7\n\tv_max3_i32 %0, %0, %7, %6")
OPDEP(d_sdwa, "v_add_u32_sdwa %0, %0, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\tv_add_u32_sdwa %0, %0, sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\tv_add_u32_sdwa %0, %0, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\tv_add_u32_sdwa %0, %0, sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
OPDEP(d_dpp, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
// the fill kernel's chain: max3 -> and -> add, undiluted; then with one / two independent fillers after every chain instruction
OPDEP(d_mix, "v_max3_i32 %0, %0, %6, %7\n\tv_and_b32 %0, %0, %6\n\tv_add_u32 %0, %0, %7\n\tv_max3_i32 %0, %0, %7, %6")
OPDEP(d_mix1, "v_max3_i32 %0, %0, %6, %7\n\tv_add_u32 %1, %1, %6\n\tv_and_b32 %0, %0, %6\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %0, %0, %7\n\tv_add_u32 %3, %3, %6\n\tv_max3_i32 %0, %0, %7, %6\n\tv_add_u32 %4, %4, %6")
OPDEP(d_mix1s, "v_max3_i32 %0, %0, %6, %7\n\tv_perm_b32 %1, %1, %6, %7\n\tv_and_b32 %0, %0, %6\n\tv_perm_b32 %2, %2, %6, %7\n\tv_add_u32 %0, %0, %7\n\tv_perm_b32 %3, %3, %6, %7\n\tv_max3_i32 %0, %0, %7, %6\n\tv_perm_b32 %4, %4, %6, %7")
OPDEP(d_mix2, "v_max3_i32 %0, %0, %6, %7\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %5, %5, %6\n\tv_and_b32 %0, %0, %6\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %1, %1, %7\n\tv_add_u32 %0, %0, %7\n\tv_add_u32 %3, %3, %6\n\tv_add_u32 %2, %2, %7\n\tv_max3_i32 %0, %0, %7, %6\n\tv_add_u32 %4, %4, %6\n\tv_add_u32 %3, %3, %7")

typedef void (*kfn)(int*, unsigned long long*, int);
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;   // run only the rows whose name contains this
    struct { const char* n; kfn f; int per; } t[] = {
        {"v_fma_f32", k_fma, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_add_f32", k_pkadd32, 1}, {"v_mul_f32", k_mulf, 1}, {"v_add_f32", k_addf, 1},
        {"v_max_f32", k_maxf, 1}, {"v_max3_f32", k_max3f, 1}, {"v_max3_i32", k_max3i, 1}, {"v_max_i32", k_maxi, 1}, {"v_max_i16", k_maxi16, 1},
        {"v_max_u16", k_maxu16, 1}, {"v_add_u32", k_add, 1}, {"v_add_i32 clamp", k_addc, 1}, {"v_sub_u32 clamp", k_subc, 1}, {"v_and_b32", k_and, 1},
        {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1}, {"v_mov_b32", k_mov, 1}, {"v_add_u32_sdwa sext(byte)", k_sdwa, 1}, {"v_perm_b32", k_perm, 1},
        {"v_pk_max_i16", k_pkmax, 1}, {"v_pk_add_i16", k_pkadd, 1}, {"v_pk_sub_u16 clamp", k_pksubc, 1}, {"v_cndmask_b32 (sgpr mask)", k_cndmask, 1},
        {"v_cmp_eq_u32 + v_mov_b32", k_cmp, 2}, {"v_lshlrev_b32", k_lshl, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_add3_u32", k_add3, 1},
        {"v_mad_i32_i24", k_mad24, 1}, {"v_cvt_f32_ubyte1", k_cvtub, 1}, {"v_mov_b32_dpp wave_shr:1", k_dpp, 1}, {"v_add_u32_dpp row_shr:1", k_adddpp, 1},
        {"v_max3_i16", k_max3i16, 1}, {"v_max3_u16", k_max3u16, 1}, {"v_min3_i16", k_min3i16, 1}, {"v_med3_i16", k_med3i16, 1}, {"v_add_u16", k_addu16, 1},
        {"v_sub_u16 clamp", k_subu16c, 1}, {"v_add_i16 clamp", k_addi16c, 1}, {"v_mad_u16", k_madu16, 1}, {"v_max3_f16", k_max3f16, 1}, {"v_max_f16", k_maxf16, 1},
        {"v_add_u16_sdwa sext(byte)", k_addu16sdwa, 1}, {"v_addc_co_u32", k_addcou32, 1}, {"v_subrev_u32", k_subrev, 1}, {"v_bfi_b32", k_bfi2, 1},
        {"v_and_or_b32", k_andor, 1}, {"idx_on+4 v_add+idx_off /add", k_idx4, 1},
        {"v_dot4_i32_i8", k_dot4i8, 1}, {"v_dot4_u32_u8", k_dot4u8, 1}, {"v_dot8_i32_i4", k_dot8i4, 1}, {"v_dot2_i32_i16", k_dot2i16, 1},
        {"v_sad_u8", k_sad8, 1}, {"v_bfe_i32", k_bfei, 1},
        {"SW cell, profile form /cell", k_cell4, 1}, {"SW cell, table form   /cell", k_cell4old, 1}, {"s_nop 0", k_nop, 1}};
    struct { const char* n; kfn f; int chain, total; } dt_[] = {
        {"v_add_u32 chain", d_add, 4, 4}, {"v_and/v_or chain", d_and, 4, 4}, {"v_max3_i32 chain", d_max3, 4, 4}, {"v_add_u32_sdwa chain", d_sdwa, 4, 4},
        {"v_mov_b32_dpp wave_shr chain", d_dpp, 4, 4}, {"max3->and->add chain", d_mix, 4, 4}, {"... + 1 v_add filler each", d_mix1, 4, 8},
        {"... + 1 v_perm filler each", d_mix1s, 4, 8}, {"... + 2 v_add fillers each", d_mix2, 4, 12}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int max_blocks = prop.multiProcessorCount * 4 * 8;
    int* d;
    unsigned long long* dt;
    hipMalloc(&d, (size_t)max_blocks * 64 * 4);
    hipMalloc(&dt, (size_t)max_blocks * 8);
    std::vector<unsigned long long> ht(max_blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("device %s, %d CUs.  Per instruction stream, at w = 1 / 2 / 4 / 8 waves per SIMD (every SIMD of the chip busy):\n", prop.gcnArchName,
           prop.multiProcessorCount);
    printf("  cyc/SIMD(ticks) = s_memtime ticks per instruction of one wave / w;  cyc/SIMD(wall@2.4) = wall time x 2.4 GHz per instruction / w ... and the\n"
           "  clock they imply (GHz = ticks / wall).  %d x %d instructions per wave and launch.\n", N_ITER, UNR);
    printf("%-28s | %-31s | %-31s | %s\n", "instruction", "cyc/SIMD (s_memtime ticks)", "cyc/SIMD (wall at 2.4 GHz)", "ms per launch (w=1), tick rate GHz (w=1, w=8)");
    for (auto& e : t) {
        if (only && !strstr(e.n, only)) continue;
        double ct[4], cw[4], ghz[4], ms1 = 0;
        int q = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = prop.multiProcessorCount * 4 * wps;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, d, dt, 3);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, d, dt, 3);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(ht.data(), dt, (size_t)blocks * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            for (int b = 0; b < blocks; ++b) sum += (double)ht[b];
            const double n_instr = (double)N_ITER * UNR * e.per;
            ct[q] = sum / blocks / n_instr / wps;
            cw[q] = ms * 1e-3 * 2.4e9 / n_instr / wps;
            ghz[q] = (sum / blocks) / (ms * 1e-3) / 1e9;
            if (wps == 1) ms1 = ms;
            ++q;
        }
        fflush(stdout);
        printf("%-28s | %6.2f  %6.2f  %6.2f  %6.2f | %6.2f  %6.2f  %6.2f  %6.2f | %6.2f ms  %5.3f  %5.3f\n", e.n, ct[0], ct[1], ct[2], ct[3], cw[0], cw[1],
               cw[2], cw[3], ms1, ghz[0], ghz[3]);
    }
    printf("\nONE wave per SIMD (w = 1), dependent chains: cycles (s_memtime ticks) per CHAIN instruction, and per instruction overall\n");
    for (auto& e : dt_) {
        if (only) break;
        const int blocks = prop.multiProcessorCount * 4;
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(64), 0, 0, d, dt, 3);
        hipDeviceSynchronize();
        hipMemcpy(ht.data(), dt, (size_t)blocks * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < blocks; ++b) sum += (double)ht[b];
        const double groups = (double)N_ITER * (UNR / 4);
        printf("%-32s | %6.2f per chain instruction | %6.2f per instruction (%d of %d on the chain)\n", e.n, sum / blocks / groups / e.chain,
               sum / blocks / groups / e.total, e.chain, e.total);
    }
    return 0;
}
