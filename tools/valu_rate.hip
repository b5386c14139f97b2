// Issue rate of the vector instructions the streaming front kernel is made of, per SIMD, by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
// Every kernel runs ITERS x 64 instructions of one kind on 8 independent registers; blocks of 256 threads put one
// wave on each SIMD of a CU, so a grid of 256 x k blocks gives k waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
constexpr int ITERS = 4000;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {             \
        unsigned r[8], b = seed + threadIdx.x, c = seed * 3u + 1u;                          \
        for (int k = 0; k < 8; k++) r[k] = seed + k + threadIdx.x;                          \
        for (int it = 0; it < ITERS; it++) {                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                  \
                _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(ASM : "+v"(r[k]) : "v"(b), "v"(c)); \
            }                                                                                \
        }                                                                                    \
        unsigned s = 0;                                                                      \
        for (int k = 0; k < 8; k++) s ^= r[k];                                               \
        if (s == 0x12345678u) out[0] = s;                                                    \
    }
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_sub_sat, "v_pk_sub_u16 %0, %0, %1 clamp")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL(k_mov_dpp_wave, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_mov_dpp_row, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 8, %1")
KERNEL(k_mad_u64, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_pk_mul_lo, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")

KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0")
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL(k_min_i32, "v_min_i32 %0, %0, %1")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_add_u16, "v_add_u16 %0, %0, %1")
KERNEL(k_max_i16, "v_max_i16 %0, %0, %1")
KERNEL(k_mad_u16, "v_mad_u16 %0, %0, %1, %2")
KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_add_dpp, "v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(k_pk_lshr, "v_pk_lshrrev_b16 %0, 8, %0 op_sel_hi:[0,1]")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6")
KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_max3, "v_max3_u32 %0, %0, %1, %2")
KERNEL(k_sub_i32_vop3, "v_sub_i32 %0, %0, %1")
KERNEL(k_add_i32_clamp, "v_add_i32 %0, %0, %1 clamp")
__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned *out, unsigned seed) {
    double r[8]; double b = seed + threadIdx.x, c = seed * 3.0 + 1.0;
    for (int k = 0; k < 8; k++) r[k] = seed + k + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
        _Pragma("unroll") for (int u = 0; u < 8; u++) {
            _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(c));
        }
    }
    double s = 0; for (int k = 0; k < 8; k++) s += r[k];
    if (s == 12345.678) out[0] = 1;
}

#define KERNEL2(NAME, ASM_A, ASM_B)                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {             \
        unsigned r[8], b = seed + threadIdx.x, c = seed * 3u + 1u;                          \
        for (int k = 0; k < 8; k++) r[k] = seed + k + threadIdx.x;                          \
        for (int it = 0; it < ITERS; it++) {                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                  \
                _Pragma("unroll") for (int k = 0; k < 8; k++) {                              \
                    asm volatile(ASM_A : "+v"(r[k]) : "v"(b), "v"(c));                       \
                    asm volatile(ASM_B : "+v"(r[(k + 4) & 7]) : "v"(b), "v"(c));             \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
        unsigned s = 0;                                                                      \
        for (int k = 0; k < 8; k++) s ^= r[k];                                               \
        if (s == 0x12345678u) out[0] = s;                                                    \
    }
KERNEL2(k_mix_add_dot4, "v_add_u32 %0, %0, %1", "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL2(k_mix_add_pk, "v_add_u32 %0, %0, %1", "v_pk_add_u16 %0, %0, %1")
KERNEL2(k_mix_add_add, "v_add_u32 %0, %0, %1", "v_sub_u32 %0, %0, %1")
KERNEL2(k_mix_add_salu, "v_add_u32 %0, %0, %1", "s_add_u32 s20, s20, 1")
KERNEL2(k_mix_pk_salu, "v_pk_add_u16 %0, %0, %1", "s_add_u32 s20, s20, 1")
KERNEL2(k_mix_add_nop, "v_add_u32 %0, %0, %1", "s_nop 0")
KERNEL2(k_mix_pk_nop, "v_pk_add_u16 %0, %0, %1", "s_nop 0")
KERNEL2(k_mix_add_dpp, "v_add_u32 %0, %0, %1", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")

#define KERNELR(NAME, ASM_A, ASM_B, RUN)                                                     \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {             \
        unsigned r[8], b = seed + threadIdx.x, c = seed * 3u + 1u;                          \
        for (int k = 0; k < 8; k++) r[k] = seed + k + threadIdx.x;                          \
        for (int it = 0; it < ITERS; it++) {                                                 \
            _Pragma("unroll") for (int u = 0; u < 32 / RUN; u++) {                           \
                _Pragma("unroll") for (int k = 0; k < RUN; k++) asm volatile(ASM_A : "+v"(r[k & 7]) : "v"(b), "v"(c)); \
                _Pragma("unroll") for (int k = 0; k < RUN; k++) asm volatile(ASM_B : "+v"(r[k & 7]) : "v"(b), "v"(c)); \
            }                                                                                \
        }                                                                                    \
        unsigned s = 0;                                                                      \
        for (int k = 0; k < 8; k++) s ^= r[k];                                               \
        if (s == 0x12345678u) out[0] = s;                                                    \
    }
KERNELR(k_run2, "v_add_u32 %0, %0, %1", "v_dot4_u32_u8 %0, %0, %1, %2", 2)
KERNELR(k_run4, "v_add_u32 %0, %0, %1", "v_dot4_u32_u8 %0, %0, %1, %2", 4)
KERNELR(k_run8, "v_add_u32 %0, %0, %1", "v_dot4_u32_u8 %0, %0, %1, %2", 8)
KERNELR(k_run16, "v_add_u32 %0, %0, %1", "v_dot4_u32_u8 %0, %0, %1, %2", 16)
KERNELR(k_run32, "v_add_u32 %0, %0, %1", "v_dot4_u32_u8 %0, %0, %1, %2", 32)
typedef void (*kern_t)(unsigned *, unsigned);
int main() {
    unsigned *out;
    CK(hipMalloc(&out, 64));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char *name; kern_t k; } ks[] = {
        {"v_add_u32", k_add_u32}, {"v_fma_f32", k_fma_f32}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_mad_u16", k_pk_mad_u16},
        {"v_pk_sub_u16 clamp", k_pk_sub_sat}, {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_mul_lo_u16", k_pk_mul_lo}, {"v_dot4_u32_u8", k_dot4},
        {"v_alignbyte_b32", k_alignbyte}, {"v_perm_b32", k_perm}, {"v_bfi_b32", k_bfi}, {"v_lshl_add_u32", k_lshl_add},
        {"v_mad_u32_u24", k_mad_u64}, {"v_mov_dpp wave_shr", k_mov_dpp_wave}, {"v_mov_dpp row_shr", k_mov_dpp_row},
        {"v_sub_u32", k_sub_u32},
        {"v_and_b32", k_and},
        {"v_or_b32", k_or},
        {"v_xor_b32", k_xor},
        {"v_lshlrev_b32", k_lshl},
        {"v_lshrrev_b32", k_lshr},
        {"v_ashrrev_i32", k_ashr},
        {"v_max_u32", k_max_u32},
        {"v_min_i32", k_min_i32},
        {"v_mul_u32_u24", k_mul_u24},
        {"v_mov_b32", k_mov},
        {"v_add3_u32", k_add3},
        {"v_and_or_b32", k_and_or},
        {"v_lshl_or_b32", k_lshl_or},
        {"v_or3_b32", k_or3},
        {"v_bfe_u32", k_bfe},
        {"v_add_u16", k_add_u16},
        {"v_max_i16", k_max_i16},
        {"v_mad_u16", k_mad_u16},
        {"v_sad_u8", k_sad_u8},
        {"v_add_u32 sdwa", k_add_sdwa},
        {"v_add_u32 dpp wave_shr", k_add_dpp},
        {"v_pk_fma_f32 (64-bit)", k_pk_fma_f32},
        {"v_cvt_f32_ubyte1", k_cvt_ub},
        {"v_mul_f32", k_mul_f32},
        {"v_pk_lshrrev_b16", k_pk_lshr},
        {"v_bitop3_b32", k_bitop3},
        {"v_med3_i32", k_med3},
        {"v_max3_u32", k_max3},
        {"v_sub_i32 (VOP3)", k_sub_i32_vop3},
        {"v_add_i32 clamp", k_add_i32_clamp},
        {"mix add+dot4 (per instr)", k_mix_add_dot4}, {"mix add+pk_add", k_mix_add_pk}, {"mix add+sub", k_mix_add_add},
        {"mix add+s_add (per instr)", k_mix_add_salu}, {"mix pk_add+s_add", k_mix_pk_salu}, {"mix add+s_nop0", k_mix_add_nop},
        {"mix pk_add+s_nop0", k_mix_pk_nop}, {"mix add+dpp", k_mix_add_dpp},
        {"runs of 2: add | dot4", k_run2}, {"runs of 4", k_run4}, {"runs of 8", k_run8}, {"runs of 16", k_run16}, {"runs of 32", k_run32},
    };
    printf("%d CUs, clock %d kHz; cycles per wave64 instruction per SIMD at the nominal clock, by waves per SIMD\n", cus, prop.clockRate);
    printf("%-24s %8s %8s %8s %8s\n", "instruction", "1", "2", "4", "8");
    for (auto &k : ks) {
        printf("%-24s", k.name);
        for (int w : {1, 2, 4, 8}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k.k, dim3(cus * w), dim3(256), 0, 0, out, 7u);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)ITERS * 64 * w;
            const double cyc = best * 1e-3 * (double)prop.clockRate * 1e3 / instr_per_simd;
            printf(" %8.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
